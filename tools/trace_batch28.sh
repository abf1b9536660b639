#!/bin/bash
# rocprofv3 kernel trace of the configs[2] shape (28 frames per launch, fp16 operands; PREC = f16 | f16r), one stream so the
# durations are isolated: per (kernel, grid, LDS) launches per step, average us, share -- the layers that decide configs[2].
#   tools/trace_batch28.sh [f16r] [tag]   -> gpurun_out/<tag>_batch28_<prec>_kernels.txt, <tag>_bench_kernel_stats_batch28_<prec>.csv (rocprofv3 --stats)
PREC=${1:-f16r}
TAG=${2:-r05}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace_b28
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_b28 -o p -- \
    python $REPO/bench.py --batch 28 --streams 1 --precision $PREC --steps 12 --warmup 4 --no-cpu-baseline --no-served-legs --no-flip-rate --other-modes "" --no-roofline --no-side-runs --repeats 1 > $OUT/${TAG}_batch28_${PREC}_under_rocprof.json 2>/dev/null
cp $(find $OUT/trace_b28 -name "p_kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats_batch28_${PREC}.csv
python - <<PY > $OUT/${TAG}_batch28_${PREC}_kernels.txt
import csv, glob
f = glob.glob("$OUT/trace_b28/**/p_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * 0.5):]                      # the timed steps
per = {}
for r in rows:
    k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * max(1, int(r.get("Grid_Size_Y", 1) or 1)), int(r["LDS_Block_Size"]))
    a = per.setdefault(k, [0, 0]); a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in per.values())
# steps in the window = launches of a kernel that runs once per step (round-4 verdict: a hard-coded 6 was wrong)
steps = float(max(1, sum(1 for r in rows if "heatmap_argmax" in r["Kernel_Name"])))
print("# batch 28 x 1 stream, precision $PREC, %d launches in the window, %.3f ms of kernel time per 28-frame step" % (len(rows), tot / 1e6 / steps))
print("# kernel | blocks | LDS B | launches/step | avg us | ms/step | share")
for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%-58s %7d %6d %6.1f %9.1f %8.3f %6.1f%%" % (k[0][:58], k[1], k[2], v[0] / steps, v[1] / v[0] / 1e3, v[1] / 1e6 / steps, 100.0 * v[1] / tot))
PY
rm -rf $OUT/trace_b28
cat $OUT/${TAG}_batch28_${PREC}_kernels.txt
