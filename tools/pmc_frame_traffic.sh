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
# set-up work of the process is not a frame's traffic (round 6: the arena's zero fill and the filter packing of a 28-frame engine were 49 MB
# "per frame" of a 224-frame run): runtime fills (no memset is on the path; the frame copies of the bench loop stay in), filter conversion and packing, the dispatch probe -- counted apart
SETUP = ("__amd_rocclr_fillBuffer", "pack_wpl", "f32_to_bf16x3_staged", "f32_to_f16_staged", "f32_to_f16_kernel", "f32_to_bf16x3_kernel", "probe_placement")
def split(per):
    run = {k: v for k, v in per.items() if not any(t in k for t in SETUP)}
    return run, {k: v for k, v in per.items() if k not in run}
F, FS = split(F); W, WS = split(W)
import re
extra = "$EXTRA"
def arg(name, dflt):
    m = re.findall(r"--%s[ =](\d+)" % name, extra)
    return int(m[-1]) if m else dflt
frames = float((arg("steps", 40) + arg("warmup", 4)) * arg("batch", 1))     # (every frame of the run, warm-up included, is under the counters)
fb = sum(v[1] for v in F.values()) * 1024 * 2 / frames; wb = sum(v[1] for v in W.values()) * 1024 / frames
print(json.dumps({"streams": $ST, "frames_counted": frames, "bench_args": extra, "fetch_MB_per_frame(x2 corrected)": fb / 1e6, "write_MB_per_frame": wb / 1e6,
                  "setup_kernels_excluded_MB_total": {"fetch": round(sum(v[1] for v in FS.values()) * 2048 / 1e6, 1), "write": round(sum(v[1] for v in WS.values()) * 1024 / 1e6, 1),
                                                      "kernels": sorted(set(FS) | set(WS))},
                  "per_kernel_fetch_MB_per_frame": {k: round(v[1] * 2048 / frames / 1e6, 1) for k, v in sorted(F.items(), key=lambda kv: -kv[1][1])[:16]},
                  "per_kernel_write_MB_per_frame": {k: round(v[1] * 1024 / frames / 1e6, 1) for k, v in sorted(W.items(), key=lambda kv: -kv[1][1])[:16]}}, indent=1))
PY
