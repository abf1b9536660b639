#!/usr/bin/env python3
"""The filters-direct 64x64 tile against its two-K-group form (TILE_BD_K2) on the 1x1 / stride-2 / small 3x3 shapes the plan keeps on it,
one kernel at a time, best slice count each.  python tools/bench_bdk2.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from betapose_amd import ops
# (H, W, Cin, Cout, k, stride, count per frame)
SHAPES = [(52, 52, 256, 128, 1, 1, 10), (26, 26, 512, 256, 1, 1, 10), (13, 13, 1024, 512, 1, 1, 7), (104, 104, 128, 64, 1, 1, 2), (208, 208, 64, 32, 1, 1, 1),
          (20, 16, 1024, 256, 1, 1, 22), (20, 16, 256, 1024, 1, 1, 23), (20, 16, 256, 256, 3, 1, 22), (40, 32, 512, 128, 1, 1, 3), (40, 32, 128, 512, 1, 1, 4),
          (80, 64, 64, 256, 1, 1, 4), (80, 64, 256, 64, 1, 1, 2), (10, 8, 2048, 512, 1, 1, 2), (10, 8, 512, 2048, 1, 1, 3), (10, 8, 512, 512, 3, 1, 2),
          (104, 104, 64, 128, 3, 1, 2), (208, 208, 32, 64, 3, 1, 1), (416, 416, 32, 64, 3, 2, 1), (208, 208, 64, 128, 3, 2, 1), (104, 104, 128, 256, 3, 2, 1),
          (52, 52, 256, 512, 3, 2, 1), (26, 26, 512, 1024, 3, 2, 1), (26, 26, 768, 256, 1, 1, 1), (52, 52, 384, 128, 1, 1, 1)]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
TILES = sys.argv[1].split(",") if len(sys.argv) > 1 else ["bd", "bdk2"]      # e.g. bd,bdk2,bdk2d2,bdk2d3,bdk2d4
tot = {t: 0.0 for t in TILES}
tot["best"] = 0.0
for (h, w_, cin, co, k, st, cnt) in SHAPES:
    x = torch.randn(1, h, w_, cin, generator=g).to(dev)
    wt = torch.randn(co, cin, k, k, generator=g) / np.sqrt(cin * k * k)
    oh = (h + 2 * (k // 2) - k) // st + 1; ow = (w_ + 2 * (k // 2) - k) // st + 1
    res = torch.randn(1, oh, ow, co, generator=g).to(dev)
    nch = cin * k * k // 32
    best = {}
    for tile in TILES:
        tb = (1e9, None)
        for sp in (1, 2, 3, 4, 5, 6, 8, 10, 12):
            if sp > 1 and nch // sp < (2 if tile == "bd" else 4):
                continue
            ms = ops.conv2d_nhwc(x, wt, None, stride=st, pad=k // 2, act="leaky", res=res, res_after_act=True, splits=sp, iters=30, tile=tile + "_b3")[-1]
            if ms * 1e3 < tb[0]:
                tb = (ms * 1e3, sp)
        best[tile] = tb
    for t in TILES:
        tot[t] += best[t][0] * cnt
    tot["best"] += min(best[t][0] for t in TILES) * cnt
    M = oh * ow; cpad = (co + 63) // 64 * 64
    print("{%6d, %5d, %4d}  k%d s%d %dx%d %d->%d x%d | " % (M, cpad, nch, k, st, h, w_, cin, co, cnt) + "  ".join("%s %d:%.1f" % (t, best[t][1], best[t][0]) for t in TILES), flush=True)
print("sum per frame (us): " + ", ".join("%s %.1f" % (t, tot[t]) for t in TILES) + ", best-of %.1f" % tot["best"])
