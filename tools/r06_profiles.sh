#!/bin/bash
# round-6 profile set (gpurun_out/r06_*; copy what you keep into profiles/): rocprofv3 kernel stats of the default command; the
# configs[2] shape (28 frames per launch, f16 and f16r) as rocprofv3 stats + per-kernel table, HBM-side traffic per frame and matrix-pipe
# occupancy (verdict item 4); PMC traffic per launch of the dominant kernel and per frame at batch 1; in-situ layer table; per-op tables.
REPO="$(cd "$(dirname "$0")/.." && pwd)"; OUT=$REPO/gpurun_out; cd $REPO
tools/trace_headline.sh r06 > $OUT/r06_trace_summary_stdout.txt 2>&1
for P in f16 f16r bf16x3; do tools/trace_batch28.sh $P r06 > /dev/null 2>&1; done
tools/pmc_frame_traffic.sh 1 --batch 28 --precision f16 --steps 6 --warmup 2 > $OUT/r06_pmc_frame_traffic_batch28_f16.json 2>/dev/null
tools/pmc_frame_traffic.sh 1 --batch 28 --precision f16r --steps 6 --warmup 2 > $OUT/r06_pmc_frame_traffic_batch28_f16r.json 2>/dev/null
tools/pmc_mfma_busy.sh --precision f16r --batch 28 --steps 3 --warmup 1 > /dev/null 2>&1 && cp $OUT/pmc_mfma_busy.json $OUT/r06_pmc_mfma_busy_batch28_f16r.json
BP_PMC_KERNEL=conv_igemm_bdk2 tools/pmc_traffic.sh > /dev/null 2>&1 && cp $OUT/pmc_traffic.json $OUT/r06_pmc_traffic.json
tools/pmc_frame_traffic.sh 1 > $OUT/r06_pmc_frame_traffic.json 2>/dev/null
tools/pmc_wave_stalls.sh --batch 28 --precision f16r --steps 4 --warmup 1 > /dev/null 2>&1 && cp $OUT/pmc_wave_stalls.json $OUT/r06_pmc_wave_stalls_batch28_f16r.json
tools/pmc_mfma_busy.sh > /dev/null 2>&1 && cp $OUT/pmc_mfma_busy.json $OUT/r06_pmc_mfma_busy.json
python tools/per_op.py > $OUT/r06_per_op_b1_bf16x3.txt 2>/dev/null
python tools/per_op.py --batch 28 --precision f16r --iters 5 > $OUT/r06_per_op_b28_f16r.txt 2>/dev/null
python tools/per_op.py --batch 28 --precision f16 --iters 5 > $OUT/r06_per_op_b28_f16.txt 2>/dev/null
# the persistent 3x3 kernel (conv_p3.hip) against the plan without it: per-op table, one kernel at a time, the configs[2] A/B on this box, clock / power probe
BP_NO_P3=1 python tools/per_op.py --batch 28 --precision f16r --iters 5 > $OUT/r06_per_op_b28_f16r_no_p3.txt 2>/dev/null
python tools/bench_p3.py 28 > $OUT/r06_bench_p3_f16.txt 2>/dev/null
BP_CONV_F16R=1 python tools/bench_p3.py 28 > $OUT/r06_bench_p3_f16r.txt 2>/dev/null
tools/ab_p3.sh 2 > $OUT/r06_ab_p3.txt 2>&1
python tools/p3_clock_probe.py > $OUT/r06_p3_clock_probe.txt 2>/dev/null
python tools/bench_p1.py 28 > $OUT/r06_bench_p1.txt 2>/dev/null
python tools/bench_stem7.py > $OUT/r06_bench_stem7.txt 2>/dev/null
tools/energy_configs2.sh > $OUT/r06_energy_configs2.txt 2>&1
tools/energy_modes.sh > $OUT/r06_energy_modes.txt 2>&1
tools/ab_stem7.sh > $OUT/r06_ab_stem7.txt 2>&1
python bench.py --no-cpu-baseline --no-served-legs --no-flip-rate --other-modes= --steps 200 --warmup 20 --insitu $OUT/r06_insitu_layer_times.txt > $OUT/r06_bench_insitu.json 2>/dev/null
rm -rf $OUT/trace_headline $OUT/trace_b28 $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmcf_* $OUT/pmc_stalls $OUT/pmc_mfma* $OUT/pmc_busy* 2>/dev/null
ls $OUT | grep r06
