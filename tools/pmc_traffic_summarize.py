#!/usr/bin/env python3
"""Summarise the two rocprofv3 PMC passes of tools/pmc_traffic.sh (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace
only) into profiles/<round>_pmc_traffic[_<precision>].json: HBM-side bytes per launch of the dominant conv kernel.
Corrections as prescribed in MI355X_MICROARCH.md (HBM / rocprofv3 section): counters are in KB; on gfx950 FETCH_SIZE
reports half the bytes of 16-B/lane coalesced reads -> doubled.
    python tools/pmc_traffic_summarize.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE out.json [kernel-substring]"""
import csv, glob, json, os, sys

def load(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert f, "no counter_collection.csv under " + d
    per = {}
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != counter:
            continue
        per.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    return per

fd, wd, out = sys.argv[1:4]
want = sys.argv[4] if len(sys.argv) > 4 else "conv_igemm"
F, Wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
# dominant = the matching kernel with the most launches
name = max((k for k in F if want in k), key=lambda k: len(F[k]))
fk = sum(F[name]) / len(F[name])
wk = sum(Wr[name]) / len(Wr[name])
res = {"kernel": name.replace("void ", "").split("(")[0],
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --steps 20 "
                  "--warmup 2 --no-roofline --no-cpu-baseline --no-served-legs --no-flip-rate --other-modes \"\" --streams 1" + (" " + os.environ.get("BP_BENCH_EXTRA", "")).rstrip(),
       "launches": len(F[name]), "FETCH_SIZE_KB_mean_raw": fk, "WRITE_SIZE_KB_mean_raw": wk,
       "correction": "gfx950 rocprofv3 FETCH_SIZE reports half the bytes of 16-B/lane coalesced reads "
                     "(MI355X_MICROARCH.md HBM): fetch bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE taken as KB",
       "fetch_bytes_per_launch": 2 * fk * 1024, "write_bytes_per_launch": wk * 1024,
       "traffic_bytes_per_launch": 2 * fk * 1024 + wk * 1024}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
