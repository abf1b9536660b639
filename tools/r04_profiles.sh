#!/bin/bash
# round-4 profile set (gpurun_out/r04_*): rocprofv3 kernel stats of the default command, PMC traffic per launch for the two
# dominant conv kernels and per frame, matrix-pipe occupancy, wave stall breakdown, the halo kernels alone, in-situ layer table
REPO="$(cd "$(dirname "$0")/.." && pwd)"; OUT=$REPO/gpurun_out; cd $REPO
tools/trace_headline.sh r04 > $OUT/r04_trace_summary_stdout.txt 2>&1
BP_PMC_KERNEL=conv_igemm_bdk2 tools/pmc_traffic.sh > /dev/null 2>&1 && cp $OUT/pmc_traffic.json $OUT/r04_pmc_traffic.json
BP_PMC_KERNEL=conv_halo tools/pmc_traffic.sh > /dev/null 2>&1 && cp $OUT/pmc_traffic.json $OUT/r04_pmc_traffic_halo.json
tools/pmc_frame_traffic.sh 1 > $OUT/r04_pmc_frame_traffic.json 2>/dev/null
tools/pmc_mfma_busy.sh > /dev/null 2>&1 && cp $OUT/pmc_mfma_busy.json $OUT/r04_pmc_mfma_busy.json
tools/pmc_wave_stalls.sh > /dev/null 2>&1 && cp $OUT/pmc_wave_stalls.json $OUT/r04_pmc_wave_stalls.json
{ echo "# tools/bench_halo.py: filters-direct kernel against the halo tiles, one kernel at a time, best slice count each (us)"; python tools/bench_halo.py 2>/dev/null; echo "# --batch 28"; python tools/bench_halo.py --batch 28 2>/dev/null; } > $OUT/r04_halo_kernels.txt
python bench.py --no-cpu-baseline --no-served-legs --no-flip-rate --other-modes= --steps 200 --warmup 20 --insitu $OUT/r04_insitu_layer_times.txt > $OUT/r04_bench_insitu.json 2>/dev/null
ls -la $OUT | grep r04
rm -rf $OUT/trace_headline $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmcf_* $OUT/pmc_stalls $OUT/pmc_mfma* $OUT/pmc_busy* 2>/dev/null
du -sh $OUT
