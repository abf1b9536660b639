#!/bin/bash
# LDS bank conflicts per kernel of a bench run (one stream): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, LDS instructions, unaligned stalls.
#   tools/pmc_lds.sh [bench args] > gpurun_out/r06_pmc_lds.txt
REPO="$(cd "$(dirname "$0")/.." && pwd)"; OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_lds
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_lds -o p -- \
    python $REPO/bench.py --steps 12 --warmup 4 --no-roofline --no-cpu-baseline --no-served-legs --no-flip-rate --other-modes "" --no-side-runs --repeats 1 --streams 1 "$@" > /dev/null 2>&1
python - <<PY
import csv, glob, collections
fs = glob.glob("$OUT/pmc_lds/**/p_counter_collection.csv", recursive=True)
per = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); seen = set()
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen: seen.add(key); n[k] += 1
print("# bench.py %s, one stream: per kernel -- launches | LDS-array cycles per launch | bank-conflict share | LDS-wait share of wave cycles" % "$*")
for k in sorted(per, key=lambda k: -per[k]["SQ_LDS_IDX_ACTIVE"]):
    c = per[k]
    if c["SQ_LDS_IDX_ACTIVE"] <= 0: continue
    print("%-64s %6d  %12.0f  conflicts %.3f  waiting on LDS %.3f" % (k[:64], n[k], c["SQ_LDS_IDX_ACTIVE"] / n[k], c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], c["SQ_WAIT_INST_LDS"] / max(1.0, c["SQ_WAVE_CYCLES"])))
PY
rm -rf $OUT/pmc_lds
