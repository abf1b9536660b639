#!/bin/bash
# rocprofv3 kernel trace (+ stats) of the HEADLINE configuration (default bench: bf16x3, 4 frames in flight).  The tracer
# SERIALISES the four streams (mean concurrency ~1, frames/s roughly halved): read the per-kernel AverageNs as ISOLATED
# durations (they agree with roofline.isolated of bench.py); durations with the frames overlapping come from the kernels' own
# s_memtime stamps (bench.py --insitu -> profiles/r03_insitu_layer_times.txt).
#   tools/trace_headline.sh [round tag]   -> gpurun_out/<tag>_bench_kernel_stats.csv, <tag>_bench_trace_summary.json
TAG=${1:-r02}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace_headline
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_headline -o p -- \
    python $REPO/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-served-legs --no-flip-rate --other-modes "" --no-roofline --no-side-runs --repeats 1 > $OUT/${TAG}_bench_under_rocprof.json 2>/dev/null
cp $OUT/trace_headline/p_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
python - <<PY
import csv, json
rows = list(csv.DictReader(open("$OUT/trace_headline/p_kernel_trace.csv")))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in rows)
# the timed region = the last 200 frames' worth of launches: take the last 60 % of the trace
t_lo = ev[int(len(ev) * 0.4)][0]
ev = [e for e in ev if e[0] >= t_lo]
span = max(e[1] for e in ev) - ev[0][0]
busy, cur_s, cur_e, conc = 0, None, None, 0
for s, e, _ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _ in ev)
per = {}
for s, e, k in ev:
    a = per.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
top = sorted(per.items(), key=lambda kv: -kv[1][1])[:8]
out = {"kernels": len(ev), "span_ms": span / 1e6, "device_busy_frac": busy / span, "mean_concurrency": tot / span,
       "top": [{"kernel": k, "launches": v[0], "avg_us": v[1] / v[0] / 1e3, "share_of_kernel_time": v[1] / tot} for k, v in top]}
json.dump(out, open("$OUT/${TAG}_bench_trace_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
