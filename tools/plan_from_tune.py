#!/usr/bin/env python3
"""tools/tune_conv.py output -> plan lines ("M CoutPad nchunks tile splits", one per distinct shape) for BP_PLAN_FILE, or
with --cpp the initializer rows of engine.cpp's kPlanB3.  `--only rd4,kg4` keeps a new tile only where it wins by
--margin percent over the 64x64-block kernel (isolated timings are noisy at the 2-3 % level)."""
import re
import sys

args = sys.argv[1:]
cpp = "--cpp" in args
margin = float(args[args.index("--margin") + 1]) if "--margin" in args else 0.0
path = [a for a in args if not a.startswith("--") and not a.replace(".", "").isdigit()][0]
rows = {}
for line in open(path):
    m = re.match(r"\s*\{\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)\},\s*//\s*x\d+\s+([\d.]+) us (\S+) \| (.*)", line)
    if not m:
        continue
    M, cp, nch, tile, sp = (int(m.group(i)) for i in range(1, 6))
    per = dict((t.split()[0], t.split()[1]) for t in m.group(8).split("  ") if t.strip())
    base_sp, base_us = per["64x64"].split(":")
    if tile != 0 and float(m.group(6)) > float(base_us) * (1.0 - margin / 100.0):
        tile, sp = 0, int(base_sp)
    rows[(M, cp, nch)] = (tile, sp)
for (M, cp, nch), (tile, sp) in sorted(rows.items()):
    print(("    {%6d, %5d, %4d, %d, %2d}," if cpp else "%d %d %d %d %d") % (M, cp, nch, tile, sp))
