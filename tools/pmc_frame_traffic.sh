#!/bin/bash
# HBM-side traffic of ONE FRAME (all kernels) at a given number of frames in flight: two rocprofv3 PMC passes
# (kernel-trace only), summed over every kernel of the timed run.  tools/pmc_frame_traffic.sh <streams> [bench args]
ST=${1:-4}; shift
EXTRA="$*"
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $REPO/gpurun_out/pmcf_$C
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $REPO/gpurun_out/pmcf_$C -o p -- python $REPO/bench.py --steps 40 --warmup 4 --no-roofline --no-cpu-baseline --no-served-legs --no-flip-rate --other-modes "" --no-side-runs --repeats 1 --streams $ST $EXTRA > /dev/null 2>&1
done
python - <<PY
import csv, glob, json, os
def tot(d, c):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    per = {}
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == c:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            a = per.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
    return per
F = tot("$REPO/gpurun_out/pmcf_FETCH_SIZE", "FETCH_SIZE"); W = tot("$REPO/gpurun_out/pmcf_WRITE_SIZE", "WRITE_SIZE")
import re
extra = "$EXTRA"
def arg(name, dflt):
    m = re.findall(r"--%s[ =](\d+)" % name, extra)
    return int(m[-1]) if m else dflt
frames = float((arg("steps", 40) + arg("warmup", 4)) * arg("batch", 1))     # (every frame of the run, warm-up included, is under the counters)
fb = sum(v[1] for v in F.values()) * 1024 * 2 / frames; wb = sum(v[1] for v in W.values()) * 1024 / frames
print(json.dumps({"streams": $ST, "frames_counted": frames, "bench_args": extra, "fetch_MB_per_frame(x2 corrected)": fb / 1e6, "write_MB_per_frame": wb / 1e6,
                  "per_kernel_fetch_MB_per_frame": {k: round(v[1] * 2048 / frames / 1e6, 1) for k, v in sorted(F.items(), key=lambda kv: -kv[1][1])[:16]},
                  "per_kernel_write_MB_per_frame": {k: round(v[1] * 1024 / frames / 1e6, 1) for k, v in sorted(W.items(), key=lambda kv: -kv[1][1])[:16]}}, indent=1))
PY
