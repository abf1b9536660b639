#!/usr/bin/env python3
"""The halo plane tile (TILE_PLH128) with K slices at 28 frames per launch, one kernel at a time: do the layers whose 128x128 tiles
do not fill the chip (13x13: 296 tiles of 144 stages; 20x16: 140 tiles of 72) gain from K slices?  python tools/bench_plh_splits.py [batch]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from betapose_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 28
SHAPES = [(13, 13, 512, 1024, 8), (20, 16, 256, 256, 22), (10, 8, 512, 512, 2), (40, 32, 128, 128, 3), (26, 26, 256, 512, 11), (52, 52, 128, 256, 11), (20, 16, 512, 1024, 1), (40, 32, 256, 512, 1)]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for (h, w_, cin, co, cnt) in SHAPES:
    x = torch.randn(B, h, w_, cin, generator=g).to(dev)
    wt = torch.randn(co, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    res = torch.randn(B, h, w_, co, generator=g).to(dev)
    line = []
    for tile in ("plh128", "pl128", "pl64"):
        for sp in (1, 2, 3, 4, 6, 8):
            if cin // 32 < sp:
                continue
            try:
                ms = ops.conv2d_nhwc(x, wt, None, pad=1, act="leaky", res=res, res_after_act=True, splits=sp, iters=20, tile=tile + "_f16")[-1]
            except Exception as e:
                line.append("%s/%d:err" % (tile, sp))
                continue
            line.append("%s/%d:%.1f" % (tile, sp, ms * 1e3))
    print("%dx%d %d->%d x%d | " % (h, w_, cin, co, cnt) + "  ".join(line), flush=True)
