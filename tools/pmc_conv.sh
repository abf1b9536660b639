cd /tmp && export TMPDIR=/tmp
python /root/repo/tools/bench_conv.py --iters 50
for L in y3x3_128_256_52 y1x1_256_128_52 y3x3_32_64_s2_416; do
for sp in 1; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc_$L -o p -- python /root/repo/tools/bench_conv.py --iters 5 --only $L --splits $sp > /dev/null 2>&1
done; done
ls /root/repo/gpurun_out/pmc_*/
