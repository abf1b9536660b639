#!/bin/bash
# Regenerates everything under profiles/ for one round on an MI355X box (writes into gpurun_out/, copy what you keep).
#   tools/reproduce_profiles.sh [round tag, default r01]
# Each block is independent; PMC passes are separate rocprofv3 runs with --kernel-trace only.
set -u
TAG=${1:-r01}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

# 1. headline bench line (+ other_precisions, roofline, cpu_baseline) and the opt-in modes on their own
python $REPO/bench.py                                   > $OUT/${TAG}_bench_default.json   2>/dev/null
python $REPO/bench.py --precision f16    --no-cpu-baseline --other-modes "" > $OUT/${TAG}_bench_f16.json    2>/dev/null
python $REPO/bench.py --precision bf16x3 --no-cpu-baseline --other-modes "" > $OUT/${TAG}_bench_bf16x3.json 2>/dev/null

# 2. rocprofv3 kernel statistics of the same command (kernels are serialised under the profiler: compare AverageNs of
#    the conv kernel with roofline.avg_launch_us, not the frames/s)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof32 -o p -- \
    python $REPO/bench.py --steps 200 --warmup 20 --no-cpu-baseline --other-modes "" > $OUT/${TAG}_bench_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof16 -o p -- \
    python $REPO/bench.py --steps 200 --warmup 20 --no-cpu-baseline --other-modes "" --precision f16 > $OUT/${TAG}_bench_f16_under_rocprof.json 2>/dev/null
cp $OUT/prof32/p_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
cp $OUT/prof32/p_domain_stats.csv $OUT/${TAG}_bench_domain_stats.csv
cp $OUT/prof16/p_kernel_stats.csv $OUT/${TAG}_bench_f16_kernel_stats.csv

# 3. HBM-side traffic of the conv kernels (FETCH_SIZE / WRITE_SIZE, two passes each)
cd $REPO
tools/pmc_traffic.sh                 && cp $OUT/pmc_traffic.json $OUT/${TAG}_pmc_traffic.json
tools/pmc_traffic.sh --precision f16 && cp $OUT/pmc_traffic.json $OUT/${TAG}_pmc_traffic_f16.json

# 4. per-shape split-K tuning tables (engine.cpp embeds the winners), per-op times, LDS micro-benchmark
python tools/tune_conv.py        > $OUT/${TAG}_splitk_tuning.txt        2>&1
python tools/tune_conv.py --f16  > $OUT/${TAG}_splitk_tuning_f16.txt    2>&1
python tools/tune_conv.py --b3   > $OUT/${TAG}_splitk_tuning_bf16x3.txt 2>&1
python tools/profile_ops.py      > $OUT/${TAG}_per_op_times.txt         2>/dev/null
tools/micro/run_lds_bw.sh        > $OUT/${TAG}_lds_bandwidth.txt        2>/dev/null

# 5. experiments recorded in DESIGN.md
python tools/partition_probe.py  > $OUT/${TAG}_cu_partition_probe.txt   2>/dev/null
tools/partition_sweep.sh         > $OUT/${TAG}_cu_partition_sweep.txt   2>/dev/null
tools/f16_sweep.sh               > $OUT/${TAG}_f16_sweep.txt            2>/dev/null
python tools/host_launch_cost.py > $OUT/${TAG}_host_launch_cost.txt     2>/dev/null
for pr in f32 bf16x3 f16; do python tools/soak_determinism.py --iters 4000 --precision $pr; done > $OUT/${TAG}_soak.txt 2>/dev/null
ls -la $OUT | tail -30
