#!/usr/bin/env python3
"""Halo tiles (conv_halo.hip) against the filters-direct kernel on the 3x3 / stride-1 shapes of both networks, one kernel at a
time: best slice count per kernel.  python tools/bench_halo.py [--batch B]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from betapose_amd import ops

BATCH = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 1
# (H, W, Cin, Cout, count per frame)
SHAPES = [(208, 208, 32, 64, 1), (104, 104, 64, 128, 2), (52, 52, 128, 256, 11), (26, 26, 256, 512, 11), (13, 13, 512, 1024, 7),
          (80, 64, 64, 64, 3), (40, 32, 128, 128, 3), (20, 16, 256, 256, 22), (10, 8, 512, 512, 2),
          (20, 16, 512, 1024, 1), (40, 32, 256, 512, 1), (80, 64, 128, 50, 1)]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
tot = {"bd": 0.0, "best": 0.0}
for (h, w_, cin, co, cnt) in SHAPES:
    x = torch.randn(BATCH, h, w_, cin, generator=g).to(dev)
    wt = torch.randn(co, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    res = torch.randn(BATCH, h, w_, co, generator=g).to(dev)
    line = []
    best_all = {}
    for tile in ("bd", "halo64", "halo128", "halo64k2"):
        if tile == "halo128" and ((co + 63) // 64 * 64) % 128:
            continue
        if tile != "bd" and w_ > (95 if tile == "halo128" else 79):
            continue
        tb = (1e9, None)
        cands = (1, 2, 3, 4, 5, 6, 8, 10, 12, 16) if tile == "bd" else [s for s in (1, 2, 3, 4, 6, 8, 16) if s <= cin // 32]
        for sp in cands:
            try:
                ms = ops.conv2d_nhwc(x, wt, None, stride=1, pad=1, act="leaky", res=res, res_after_act=True, splits=sp, iters=30, tile=tile + "_b3")[-1]
            except Exception as e:
                print("  ", tile, sp, "failed:", str(e)[:100]); continue
            if ms * 1e3 < tb[0]:
                tb = (ms * 1e3, sp)
        best_all[tile] = tb
        line.append("%s %s:%.1f" % (tile, tb[1], tb[0]))
    tot["bd"] += best_all["bd"][0] * cnt
    tot["best"] += min(v[0] for v in best_all.values()) * cnt
    print("%dx%d %d->%d x%d | %s" % (h, w_, cin, co, cnt, "  ".join(line)), flush=True)
print("sum per frame: filters-direct %.1f us, best-of %.1f us" % (tot["bd"], tot["best"]))
