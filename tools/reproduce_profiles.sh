#!/bin/bash
# Regenerates everything under profiles/ for one round on an MI355X box (writes into gpurun_out/, copy what you keep).
#   tools/reproduce_profiles.sh [round tag, default r03] [blocks, default "1 2 3 4 5 6 7"]
# Each block is independent; PMC passes are separate rocprofv3 runs with --kernel-trace only.  The experimental library
# (python -m betapose_amd.build --experimental) is needed by block 5 (A/B against the round-2 data path).
set -u
TAG=${1:-r03}
BLOCKS=${2:-"1 2 3 4 5 6 7"}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$REPO/gpurun_out
EXP=$REPO/betapose_amd/libbetapose_hip_exp.so
mkdir -p "$OUT"
cd $REPO
has() { [[ " $BLOCKS " == *" $1 "* ]]; }
line() { python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "frames/s", d["value"], "ms/step", d["ms_per_step"])' "$1"; }
quick() { tag=$1; shift; python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-served-legs --no-flip-rate --no-roofline --other-modes , --no-side-runs --repeats 1 "$@" 2>/dev/null | tail -1 | line "$tag"; }

if has 1; then
# 1. headline bench line (+ repeats, other_precisions, roofline with hbm / layer classes / in-situ layer table, latency,
#    h2d_inclusive, cpu_baseline) and the other modes alone; batched and stream-count sweeps
python bench.py --insitu $OUT/${TAG}_insitu_layer_times.txt > $OUT/${TAG}_bench_default.json 2>/dev/null
python bench.py --precision f32 --no-cpu-baseline --no-served-legs --no-flip-rate --other-modes , > $OUT/${TAG}_bench_f32.json 2>/dev/null
python bench.py --precision f16 --no-cpu-baseline --no-served-legs --no-flip-rate --other-modes , --insitu $OUT/${TAG}_insitu_layer_times_f16.txt > $OUT/${TAG}_bench_f16.json 2>/dev/null
for cfg in "bf16x3 28 2" "f16 28 3" "f32 28 2" "bf16x3 2 4" "bf16x3 4 3"; do set -- $cfg
  python bench.py --precision $1 --batch $2 --streams $3 --steps 60 --warmup 6 --no-cpu-baseline --no-served-legs --no-flip-rate --other-modes , --no-roofline --no-side-runs --repeats 3 2>/dev/null
done > $OUT/${TAG}_bench_batched.jsonl
for st in 1 2 3 4 5 6 8; do
  python bench.py --streams $st --steps 300 --warmup 20 --no-cpu-baseline --no-served-legs --no-flip-rate --other-modes , --no-roofline --no-side-runs --repeats 1 2>/dev/null
done > $OUT/${TAG}_bench_streams.jsonl
fi

if has 2; then
# 2. rocprofv3 kernel statistics of the default command (the tracer serialises the streams: AverageNs of the dominant conv
#    kernel = its isolated duration, compare with roofline.isolated.avg_launch_us, not the frames/s)
tools/trace_headline.sh $TAG > /dev/null
fi

if has 3; then
# 3. HBM-side traffic (FETCH_SIZE / WRITE_SIZE, two passes each): the dominant kernel per launch in the bf16x3 and fp16
#    modes, one whole frame; matrix-pipe occupancy; wave stall breakdown
tools/pmc_traffic.sh && cp $OUT/pmc_traffic.json $OUT/${TAG}_pmc_traffic.json
BP_PMC_KERNEL=conv_pl tools/pmc_traffic.sh --precision f16 && cp $OUT/pmc_traffic.json $OUT/${TAG}_pmc_traffic_f16.json
tools/pmc_traffic.sh --precision f32 && cp $OUT/pmc_traffic.json $OUT/${TAG}_pmc_traffic_f32.json
tools/pmc_frame_traffic.sh 1 > $OUT/${TAG}_pmc_frame_traffic.json
tools/pmc_frame_traffic.sh 1 --precision f16 > $OUT/${TAG}_pmc_frame_traffic_f16.json
tools/pmc_mfma_busy.sh > /dev/null && cp $OUT/pmc_mfma_busy.json $OUT/${TAG}_pmc_mfma_busy.json
tools/pmc_mfma_busy.sh --precision f16 --batch 28 --steps 3 --warmup 1 > /dev/null && cp $OUT/pmc_mfma_busy.json $OUT/${TAG}_pmc_mfma_busy_f16_batch28.json
tools/pmc_wave_stalls.sh > /dev/null && cp $OUT/pmc_wave_stalls.json $OUT/${TAG}_pmc_wave_stalls.json
fi

if has 4; then
# 4. per-shape kernel / slice tuning tables (engine.cpp embeds the winners), kernel-against-kernel timings, per-op times
python tools/tune_conv.py --pl              > $OUT/${TAG}_tune_pl_b3.txt          2>&1   # plane path, bf16x3 (BP_B3_PLANES)
python tools/tune_conv.py --pl --f16        > $OUT/${TAG}_tune_pl_f16.txt         2>&1
python tools/tune_conv.py --pl --f16 --big --batch 28 > $OUT/${TAG}_tune_pl_f16_batch28.txt 2>&1
for m in f16 b3; do
  python tools/bench_pl.py --batch 28 --mode $m --splits 1 --only y3x3 --tiles pl64,pl128x64,pl128,pl256x128 --engine-like 2>&1 | grep -v amdgpu.ids
done > $OUT/${TAG}_bench_pl_batch28.txt
python tools/bench_pl.py --batch 1 --mode b3 --tiles bd,pl64 --engine-like 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_bench_pl_batch1.txt
python tools/profile_ops.py                 > $OUT/${TAG}_per_op_times.txt     2>/dev/null
python tools/profile_ops.py --precision f16 > $OUT/${TAG}_per_op_times_f16.txt 2>/dev/null
python tools/profile_ops.py --precision f16 --batch 28 > $OUT/${TAG}_per_op_times_f16_batch28.txt 2>/dev/null
tools/micro/run_dma_bw.sh 2>/dev/null | grep "B/clk" > $OUT/${TAG}_dma_bw.txt
fi

if has 5; then
# 5. A/B of the whole pipeline on ONE box: the product, the bf16x3 mode forced onto the operand-plane path, and the round-2
#    data path in every mode (experimental library, BP_LEGACY=1)
{
for r in 1 2 3; do
  quick "product bf16x3 (fp32 activations, filters-direct kernel)"
  BP_B3_PLANES=1 quick "bf16x3 on the operand-plane path (BP_B3_PLANES=1)"
  [ -f $EXP ] && BP_LIB=$EXP BP_LEGACY=1 quick "round-2 kernels, bf16x3 (experimental library)"
done
quick "product fp16 (operand-plane path)" --precision f16
[ -f $EXP ] && BP_LIB=$EXP BP_LEGACY=1 quick "round-2 kernels, fp16" --precision f16
quick "product fp16, batch 28 x 3 streams" --precision f16 --batch 28 --streams 3 --steps 30
[ -f $EXP ] && BP_LIB=$EXP BP_LEGACY=1 quick "round-2 kernels, fp16, batch 28 x 3 streams" --precision f16 --batch 28 --streams 3 --steps 30
quick "product bf16x3, batch 28 x 2 streams" --batch 28 --streams 2 --steps 30
BP_B3_PLANES=1 quick "bf16x3 on the plane path, batch 28 x 2 streams" --batch 28 --streams 2 --steps 30
quick "product bf16x3, one frame at a time" --streams 1
BP_B3_PLANES=1 quick "bf16x3 on the plane path, one frame at a time" --streams 1
BP_KEEP_F32=1 quick "product fp16, fp32 stores kept everywhere (BP_KEEP_F32=1)" --precision f16
} > $OUT/${TAG}_ab_pipeline.txt
fi

if has 6; then
# 6. determinism soak (bit-identical records over 4000 frames under 4-stream load, every precision), BASELINE configs[4]
#    (eight resident objects, (frame, object) units), files-on-disk harness
{ for pr in bf16x3 f32 f16; do python tools/soak_determinism.py --iters 4000 --precision $pr; done; for pr in bf16x3 f16; do python tools/soak_determinism.py --iters 4000 --precision $pr --latency-mode; done; } 2>/dev/null | grep "^soak" > $OUT/${TAG}_soak.txt
python tools/occlusion_scale.py --frames 384 > $OUT/${TAG}_occlusion_8obj.txt 2>&1
for b in 1 2 4; do python evaluate.py --synthetic 768 --outdir /tmp/ev --fused --streams 4 --detbatch $b 2>&1 | grep frames/sec; done > $OUT/${TAG}_evaluate_fused.txt
fi
if has 7; then
# 7. what the chip does meanwhile: clocks and power by mode, first-touch fetch rates, block -> XCD map,
#    per-layer SQ counters of the plane kernels, latency-mode A/B by frames in flight, the cache-resident-filters bound
clk() { tag=$1; shift; python bench.py --steps 100 --warmup 20 --no-roofline --no-cpu-baseline --no-served-legs --no-flip-rate --other-modes , --repeats 2 "$@" 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], json.dumps(d.get("clocks_under_load")))' "$tag"; }
{
clk "b3 4 streams"; clk "b3 1 stream" --streams 1; clk "b3 2 streams" --streams 2; clk "b3 3 streams" --streams 3
BP_B3_PLANES=1 clk "b3 planes 4 streams"; clk "f16 4 streams" --precision f16; clk "f32 4 streams" --precision f32
clk "b3 batch 28 x2" --batch 28 --streams 2 --steps 30; clk "f16 batch 28 x3" --batch 28 --streams 3 --steps 30 --precision f16
[ -f $EXP ] && BP_LIB=$EXP clk "exp b3 4 streams fixed box" --fixed-box
[ -f $EXP ] && BP_LIB=$EXP BP_ALIAS_WEIGHTS=1 clk "exp alias b3 4 streams fixed box (every layer reads ONE filter buffer: wrong results)" --fixed-box
} > $OUT/${TAG}_clocks_power.txt
tools/micro/run_cold_fetch.sh > $OUT/${TAG}_cold_fetch.txt 2>/dev/null
tools/micro/run_xcc_map.sh 2>/dev/null | grep "^stream\|^checked\|^grid" > $OUT/${TAG}_xcc_map.txt
tools/pmc_layer.sh y3x3_128_256_52 pl128 f16 28 pl128_f16 > /dev/null && cp $OUT/pmc_layer_pl128_f16.json $OUT/${TAG}_pmc_layer_pl128_f16.json
tools/pmc_layer.sh y3x3_128_256_52 pl256x128 f16 28 pl256_f16 > /dev/null && cp $OUT/pmc_layer_pl256_f16.json $OUT/${TAG}_pmc_layer_pl256_f16.json
tools/pmc_layer.sh y3x3_128_256_52 pl128 b3 28 pl128_b3 > /dev/null && cp $OUT/pmc_layer_pl128_b3.json $OUT/${TAG}_pmc_layer_pl128_b3.json
{
for st in 1 2 4; do
  quick "bf16x3, $st in flight, default layout" --streams $st
  quick "bf16x3, $st in flight, latency mode (xcd_home + filter prefetch)" --streams $st --prefetch
  BP_NO_XCD_HOME=1 quick "bf16x3, $st in flight, latency mode without xcd_home (no prefetch either: it needs the layout)" --streams $st --prefetch
done
quick "fp16, 4 in flight, default layout" --precision f16; quick "fp16, 4 in flight, latency mode" --precision f16 --prefetch
quick "fp16, one at a time, default layout" --precision f16 --streams 1; quick "fp16, one at a time, latency mode" --precision f16 --streams 1 --prefetch
} > $OUT/${TAG}_prefetch_ab.txt
[ -f $EXP ] && for m in b3 f16; do BP_LIB=$EXP python tools/bench_pl.py --mode $m --batch 1 --tiles pl64,pl64k2 --splits 1,2,3,4,5,6,8,10 --all-splits --iters 20 2>&1 | grep -v amdgpu.ids; done > $OUT/${TAG}_pl64_kgroups.txt
fi
ls -la $OUT | tail -40
