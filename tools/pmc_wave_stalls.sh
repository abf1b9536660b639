#!/bin/bash
# Where the waves of the conv kernels spend their cycles: SQ_WAVE_CYCLES = SQ_WAIT_ANY (parked on s_waitcnt / barrier)
# + SQ_WAIT_INST_ANY (issue stall) + SQ_ACTIVE_INST_ANY, per kernel name, one rocprofv3 PMC pass (kernel-trace only) over
# a single-stream run of the default bench.   tools/pmc_wave_stalls.sh ; output: gpurun_out/pmc_wave_stalls.json
EXTRA="$*"
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
rm -rf $REPO/gpurun_out/pmc_stalls
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_stalls -o p -- \
    python $REPO/bench.py --steps 20 --warmup 2 --no-roofline --no-cpu-baseline --no-served-legs --no-flip-rate --other-modes "" --no-side-runs --repeats 1 --streams 1 $EXTRA > /dev/null 2>&1
python - <<PY
import csv, glob, json, os
fs = glob.glob(os.path.join("$REPO/gpurun_out/pmc_stalls", "**", "*counter_collection.csv"), recursive=True)
per = {}
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d = per.setdefault(k, {})
    a = d.setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
out = {}
for k, d in per.items():
    if "SQ_WAVE_CYCLES" not in d: continue
    wc = d["SQ_WAVE_CYCLES"][1]
    out[k] = {"launches": d["SQ_WAVE_CYCLES"][0], "wave_quad_cycles_per_launch": wc / d["SQ_WAVE_CYCLES"][0]}
    for c, v in d.items():
        if c != "SQ_WAVE_CYCLES": out[k][c + "_frac"] = round(v[1] / wc, 4) if wc else None
json.dump(out, open("$REPO/gpurun_out/pmc_wave_stalls.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if "conv" in k}, indent=1))
PY
