#!/usr/bin/env python3
"""Greedy in-situ tuning of the K-slice counts: for the conv shapes that occur most often, try neighbouring slice counts
in the WHOLE 4-frames-in-flight pipeline (BP_PLAN_FILE, no rebuild) and keep a change only when frames/s improve by more
than the run-to-run noise.  A kernel timed alone prefers fewer slices than the pipeline does.
    python tools/tune_splits_insitu.py            # prints the accepted changes and the final plan lines"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "betapose_amd", "csrc", "engine.cpp")).read()
a = src.index("static const PlanEntry kPlanB3[] = {")
TILE_ID = {"TILE_64x64_BD": 12, "TILE_BD_K2": 24, "TILE_HALO64K2": 23, "TILE_HALO128": 22, "TILE_HALO64": 21}
rows = re.findall(r"\{\s*(\d+),\s*(\d+),\s*(\d+),\s*(TILE_\w+),\s*(\d+)\}", src[a:src.index("};", a)])
plan = {}
for M, cp, n, t, sp in rows:          # the FIRST row of a shape is the one the engine takes (halo rows precede the filters-direct ones)
    if int(M) and (int(M), int(cp), int(n)) not in plan:
        plan[(int(M), int(cp), int(n))] = (TILE_ID[t], int(sp))
CLASSES = [(320, 256, 72), (320, 256, 32), (320, 1024, 8), (2704, 256, 36), (2704, 128, 8), (676, 512, 72), (676, 256, 16),
           (169, 1024, 144), (169, 512, 32), (1280, 128, 36), (1280, 128, 16), (5120, 64, 18), (80, 512, 144), (320, 1024, 16)]


def fps(p, steps=400):
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        for (M, cp, n), (t, s) in p.items():
            f.write("%d %d %d %d %d\n" % (M, cp, n, t, s))
    env = dict(os.environ, BP_PLAN_FILE=f.name)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "40", "--no-side-runs",
                          "--no-cpu-baseline", "--no-served-legs", "--no-flip-rate", "--no-roofline", "--other-modes", ""], capture_output=True, text=True, env=env).stdout
    os.unlink(f.name)
    return json.loads(out.strip().splitlines()[-1])["value"]


base = max(fps(plan), fps(plan))
print("baseline %.1f frames/s" % base, flush=True)
for key in CLASSES:
    if key not in plan:
        continue
    tile0, s0 = plan[key]
    nch = key[2] // (9 if tile0 in (21, 22, 23) else 1)          # halo tiles cut K by channel groups
    cands = sorted({max(1, s0 - 2), max(1, s0 - 1), s0 + 1, s0 + 2, min(16, s0 * 2)} - {s0})
    best_s, best = s0, base
    for s in cands:
        if s > 1 and nch // s < 2:
            continue
        trial = dict(plan); trial[key] = (tile0, s)
        v = fps(trial)
        print("  %s: %d -> %d slices: %.1f" % (key, s0, s, v), flush=True)
        if v > best * 1.008:
            v2 = fps(trial)                       # confirm
            if v2 > best * 1.005:
                best_s, best = s, min(v, v2)
    if best_s != s0:
        plan[key] = (tile0, best_s)
        base = best
        print("ACCEPT %s: %d -> %d slices, now %.1f frames/s" % (key, s0, best_s, base), flush=True)
print("final %.1f frames/s" % base)
NAME = {v: k for k, v in TILE_ID.items()}
for (M, cp, n), (t, s) in sorted(plan.items()):
    print("    {%6d, %5d, %4d, %s, %2d}," % (M, cp, n, NAME[t], s))
