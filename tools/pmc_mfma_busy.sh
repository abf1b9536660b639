#!/bin/bash
# Matrix-pipe occupancy of the conv kernels: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs) per kernel name, one
# rocprofv3 PMC pass (kernel-trace only) over a single-stream run of the default bench.
#   tools/pmc_mfma_busy.sh [bench args] ; output: gpurun_out/pmc_mfma_busy.json
EXTRA="$*"
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
rm -rf $REPO/gpurun_out/pmc_mfma
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $REPO/gpurun_out/pmc_mfma -o p -- \
    python $REPO/bench.py --steps 20 --warmup 2 --no-roofline --no-cpu-baseline --no-served-legs --no-flip-rate --other-modes "" --no-side-runs --repeats 1 --streams 1 $EXTRA > /dev/null 2>&1
python - <<PY
import csv, glob, json, os
f = glob.glob(os.path.join("$REPO/gpurun_out/pmc_mfma", "**", "*counter_collection.csv"), recursive=True)[0]
per = {}
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d = per.setdefault(k, {})
    a = d.setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
out = {}
for k, d in per.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in d or "GRBM_GUI_ACTIVE" not in d: continue
    n = d["GRBM_GUI_ACTIVE"][0]
    mf, act = d["SQ_VALU_MFMA_BUSY_CYCLES"][1], d["GRBM_GUI_ACTIVE"][1]
    # SQ_VALU_MFMA_BUSY_CYCLES: cycles the matrix pipe is busy, summed over the chip's 1024 SIMDs (= 32 per
    # v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md constants table); GRBM_GUI_ACTIVE: active cycles summed over the
    # 8 XCDs (a 12.9 us launch reads ~8 x its wall cycles), so wall cycles = GUI_ACTIVE / 8 -- an upper bound on the
    # kernel's own duration under the profiler
    out[k] = {"launches": n, "mfma_busy_cycles_per_launch": mf / n, "gui_active_cycles_per_launch_all_xcds": act / n,
              "mfma_busy_frac_of_1024_simds": mf / (act / 8.0 * 1024.0) if act else None}
json.dump(out, open("$REPO/gpurun_out/pmc_mfma_busy.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if "conv" in k}, indent=1))
PY
