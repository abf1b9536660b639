#!/bin/bash
# HBM-side traffic of the conv kernels: two separate rocprofv3 PMC passes (kernel-trace only), then the summary JSON.
#   [BP_PMC_KERNEL=conv_pl] tools/pmc_traffic.sh [extra bench.py args, e.g. --precision f16] ; output: gpurun_out/pmc_traffic.json
#   (BP_PMC_KERNEL: substring of the kernel to summarise; default conv_igemm, the bf16x3 / fp32 modes' kernels)
EXTRA="$*"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /root/repo/gpurun_out/pmc_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /root/repo/gpurun_out/pmc_$C -o p -- python /root/repo/bench.py --steps 20 --warmup 2 --no-roofline --no-cpu-baseline --no-served-legs --no-flip-rate --other-modes "" --repeats 1 --streams 1 $EXTRA > /dev/null 2>&1
done
cd /root/repo && BP_BENCH_EXTRA="$EXTRA" python tools/pmc_traffic_summarize.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_traffic.json ${BP_PMC_KERNEL:-conv_igemm}
