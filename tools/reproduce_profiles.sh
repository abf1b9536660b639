#!/bin/bash
# Regenerates everything under profiles/ for one round on an MI355X box (writes into gpurun_out/, copy what you keep).
#   tools/reproduce_profiles.sh [round tag, default r02]
# Each block is independent; PMC passes are separate rocprofv3 runs with --kernel-trace only.
set -u
TAG=${1:-r02}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$REPO/gpurun_out
mkdir -p "$OUT"
cd $REPO

# 1. headline bench line (+ other_precisions, roofline, latency, h2d_inclusive, cpu_baseline) and the other modes alone
python bench.py                                   > $OUT/${TAG}_bench_default.json 2>/dev/null
python bench.py --precision f32 --no-cpu-baseline --other-modes "" > $OUT/${TAG}_bench_f32.json 2>/dev/null
python bench.py --precision f16 --no-cpu-baseline --other-modes "" > $OUT/${TAG}_bench_f16.json 2>/dev/null
for cfg in "bf16x3 28 2" "f16 28 3" "f32 28 2" "bf16x3 2 4" "bf16x3 4 3"; do set -- $cfg
  python bench.py --precision $1 --batch $2 --streams $3 --steps 60 --warmup 6 --no-cpu-baseline --other-modes "" --no-roofline --no-side-runs 2>/dev/null
done > $OUT/${TAG}_bench_batched.jsonl
for st in 1 2 3 4 5 6 8; do
  python bench.py --streams $st --steps 300 --warmup 20 --no-cpu-baseline --other-modes "" --no-roofline --no-side-runs 2>/dev/null
done > $OUT/${TAG}_bench_streams.jsonl

# 2. rocprofv3 kernel statistics of the default command (kernels are serialised under the profiler: compare AverageNs
#    of the dominant conv kernel with roofline.isolated.avg_launch_us, not the frames/s)
tools/trace_headline.sh $TAG > /dev/null

# 3. HBM-side traffic: the dominant kernel per launch, and one whole frame (FETCH_SIZE / WRITE_SIZE, two passes each)
tools/pmc_traffic.sh && cp $OUT/pmc_traffic.json $OUT/${TAG}_pmc_traffic.json
tools/pmc_traffic.sh --precision f32 && cp $OUT/pmc_traffic.json $OUT/${TAG}_pmc_traffic_f32.json
tools/pmc_frame_traffic.sh 1 > $OUT/${TAG}_pmc_frame_traffic.json
tools/pmc_mfma_busy.sh > /dev/null && cp $OUT/pmc_mfma_busy.json $OUT/${TAG}_pmc_mfma_busy.json
tools/pmc_wave_stalls.sh > /dev/null && cp $OUT/pmc_wave_stalls.json $OUT/${TAG}_pmc_wave_stalls.json

# 4. per-shape kernel / slice tuning tables (engine.cpp embeds the winners), per-op times
python tools/tune_conv.py            > $OUT/${TAG}_tune_b3.txt          2>&1
python tools/tune_conv.py --kg       > $OUT/${TAG}_tune_b3_kernels.txt  2>&1   # 64x64 / filters-direct / K-group / register-direct
python tools/tune_conv.py --f16      > $OUT/${TAG}_tune_f16.txt         2>&1
python tools/tune_conv.py --fp32     > $OUT/${TAG}_tune_f32.txt         2>&1
python tools/tune_conv.py --batch 4  > $OUT/${TAG}_tune_b3_batch4.txt   2>&1
python tools/tune_conv.py --batch 28 > $OUT/${TAG}_tune_b3_batch28.txt  2>&1
python tools/tune_conv.py --f16 --batch 28 > $OUT/${TAG}_tune_f16_batch28.txt 2>&1
python tools/profile_ops.py                 > $OUT/${TAG}_per_op_times.txt     2>/dev/null
python tools/profile_ops.py --precision f32 > $OUT/${TAG}_per_op_times_f32.txt 2>/dev/null

# 5. determinism soak (bit-identical records over 4000 frames under 4-stream load, every precision)
for pr in bf16x3 f32 f16; do python tools/soak_determinism.py --iters 4000 --precision $pr; done > $OUT/${TAG}_soak.txt 2>/dev/null

# 6. what bounds the batch-1 pipeline: parts of the conv kernels compiled out (timing only), the launch-chain floor,
#    BASELINE configs[4] (eight resident objects, (frame, object) units) against eight single-object runs
bash tools/ablate_pipeline.sh > /dev/null && cp $OUT/ablate_pipeline.txt $OUT/${TAG}_ablate_pipeline.txt
python tools/launch_floor.py 202 > $OUT/${TAG}_launch_floor.txt 2>/dev/null
python tools/occlusion_scale.py --frames 384 > $OUT/${TAG}_occlusion_8obj.txt 2>&1
for b in 1 2 4; do python evaluate.py --synthetic 768 --outdir /tmp/ev --fused --streams 4 --detbatch $b 2>&1 | grep frames/sec; done > $OUT/${TAG}_evaluate_fused.txt
ls -la $OUT | tail -40
