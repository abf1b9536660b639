#!/usr/bin/env python3
"""The fp16 plane tiles against the halo form (TILE_PLH128) on the 3x3 / stride-1 layers of both networks at batch 28 (configs[2]) and
batch 1, one kernel at a time.  python tools/bench_plh.py [batch]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from betapose_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 28
# (H, W, Cin, Cout, launches per frame)
SHAPES = [(20, 16, 512, 1024, 1), (40, 32, 256, 512, 1), (52, 52, 128, 256, 11), (26, 26, 256, 512, 11), (13, 13, 512, 1024, 7), (20, 16, 256, 256, 22), (10, 8, 512, 512, 2), (40, 32, 128, 128, 3)]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
tot = {}
for (h, w_, cin, co, cnt) in SHAPES:
    x = torch.randn(B, h, w_, cin, generator=g).to(dev)
    wt = torch.randn(co, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    res = torch.randn(B, h, w_, co, generator=g).to(dev)
    line = []
    for tile in ("pl64", "pl128", "pl256x128", "plh128"):
        best = (1e9, 0)
        for sp in ((1,) if B > 4 else (1, 2, 4)):
            if cin // 32 < sp:
                continue
            ms = ops.conv2d_nhwc(x, wt, None, pad=1, act="leaky", res=res, res_after_act=True, splits=sp, iters=20, tile=tile + "_f16")[-1]
            if ms * 1e3 < best[0]:
                best = (ms * 1e3, sp)
        tot[tile] = tot.get(tile, 0.0) + best[0] * cnt
        line.append("%s %d:%.1f" % (tile, best[1], best[0]))
    print("{%7d, %5d, %4d} %dx%d %d->%d x%d | " % (B * h * w_, co, cin * 9 // 32, h, w_, cin, co, cnt) + "  ".join(line), flush=True)
print("sum per frame-batch (us): " + ", ".join("%s %.1f" % kv for kv in tot.items()))
