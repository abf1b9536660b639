#!/bin/bash
# PMC evidence for conv_p3.hip against the halo plane tile, one kernel at a time on the six 3x3 shapes of the 28-frame pass (tools/bench_p3.py):
# LDS bank conflicts / LDS-array cycles / unaligned stalls (pass A) and matrix-pipe busy / wave cycles / waits (pass B), per kernel name.
#   tools/pmc_p3.sh > gpurun_out/r06_pmc_p3.txt
REPO="$(cd "$(dirname "$0")/.." && pwd)"; OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_p3_A $OUT/pmc_p3_B
BP_CONV_F16R=1 timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_p3_A -o p -- python $REPO/tools/bench_p3.py 28 > /dev/null 2>&1
BP_CONV_F16R=1 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_p3_B -o p -- python $REPO/tools/bench_p3.py 28 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for tag in ("A", "B"):
    fs = glob.glob("$OUT/pmc_p3_%s/**/p_counter_collection.csv" % tag, recursive=True)
    if not fs:
        print("pass", tag, ": no counter file"); continue
    per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not ("conv_p3" in k or "conv_pl" in k): continue
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen: seen.add(key); n[k] += 1
    print("== pass %s: counter sums per launch (all XCDs, all SEs), by kernel" % tag)
    for k in sorted(per):
        print("%-62s launches %4d  " % (k[:62], n[k]) + "  ".join("%s %.4g" % (c, v / n[k]) for c, v in sorted(per[k].items())))
        c = per[k]
        if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"] > 0:
            print("    bank-conflict cycles / LDS-array cycles = %.4f" % (c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]))
        if "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"] > 0:
            print("    MFMA-busy cycles / SQ-busy cycles = %.4f ; waits / wave cycles = %.4f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"], c["SQ_WAIT_INST_ANY"] / max(1.0, c["SQ_WAVE_CYCLES"])))
PY
rm -rf $OUT/pmc_p3_A $OUT/pmc_p3_B
