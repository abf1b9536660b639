cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $C --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc_$C -o p -- python /root/repo/bench.py --steps 20 --warmup 2 --no-roofline --no-cpu-baseline --streams 1 > /dev/null 2>&1
done
ls /root/repo/gpurun_out/pmc_FETCH_SIZE /root/repo/gpurun_out/pmc_WRITE_SIZE
