#!/bin/bash
# Per-kernel SQ counters for ONE layer of tools/bench_pl.py (its conv_pl kernel alone): where the waves' cycles go,
# instruction mix, LDS conflicts.   tools/pmc_layer.sh <layer> <tile> <mode> <batch> [tag]   -> gpurun_out/pmc_layer_<tag>.json
L=${1:-y3x3_128_256_52}; T=${2:-pl128}; MODE=${3:-f16}; B=${4:-28}; TAG=${5:-$T_$MODE}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  n=$1; shift
  rm -rf $REPO/gpurun_out/pl_$n
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $REPO/gpurun_out/pl_$n -o p -- \
    python $REPO/tools/bench_pl.py --mode $MODE --batch $B --only $L --tiles $T --splits 1 --iters 5 > /dev/null 2>&1 || echo "pass $n failed"
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
run b SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS
run c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_LDS_DATA_FIFO_FULL
python - <<PY
import csv, glob, json, os
out = {}
for n in "abc":
    fs = glob.glob(os.path.join("$REPO/gpurun_out/pl_" + n, "**", "*counter_collection.csv"), recursive=True)
    if not fs: continue
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "conv_pl_kernel" not in k: continue
        a = out.setdefault(k, {}).setdefault(r["Counter_Name"], [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
res = {k: {c: v[1] / v[0] for c, v in d.items()} for k, d in out.items()}
json.dump(res, open("$REPO/gpurun_out/pmc_layer_$TAG.json", "w"), indent=1)
for k, d in res.items():
    print(k); wc = d.get("SQ_WAVE_CYCLES", 0)
    for c, v in sorted(d.items()): print("  %-28s %14.0f %s" % (c, v, ("%.3f of wave cycles" % (v / wc)) if wc and c.startswith("SQ_") else ""))
PY
rm -rf $REPO/gpurun_out/pl_a $REPO/gpurun_out/pl_b $REPO/gpurun_out/pl_c
