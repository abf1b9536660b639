#!/usr/bin/env python3
"""The 1x1 form of the persistent kernel (TILE_P3, conv_p3.hip KSZ = 1) against the plane tile (TILE_PL64) and the streaming kernel (TILE_S1) on the
1x1 / stride-1 layers of both networks at 28 frames per launch, one kernel at a time, launched as the f16r plan launches them.
BP_CONV_F16R=1 python tools/bench_p1.py [batch]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from betapose_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 28
# (H, W, Cin, Cout, skip connection, launches per frame)
SHAPES = [(20, 16, 1024, 256, False, 22), (20, 16, 256, 1024, True, 22), (13, 13, 1024, 512, False, 7), (26, 26, 512, 256, False, 10), (52, 52, 256, 128, False, 10),
          (40, 32, 512, 128, False, 3), (40, 32, 128, 512, True, 3), (10, 8, 2048, 512, False, 2), (10, 8, 512, 2048, True, 2), (80, 64, 256, 128, False, 1), (52, 52, 384, 128, False, 1)]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
os.environ["BP_S1_K512"] = "1"
f16r = os.environ.get("BP_CONV_F16R") is not None
for (h, w_, cin, co, r_, cnt) in SHAPES:
    x = torch.randn(B, h, w_, cin, generator=g).to(dev)
    wt = torch.randn(co, cin, 1, 1, generator=g) / np.sqrt(cin)
    res = torch.randn(B, h, w_, co, generator=g).to(dev) if r_ else None
    line = []
    for tile in ("pl64", "s1", "p3"):
        try:
            r = ops.conv2d_nhwc(x, wt, None, pad=0, act="relu", res=res, res_after_act=False, splits=1, iters=20, tile=tile + "_f16", planes=f16r)
            line.append("%s %.1f us" % (tile, r[-1] * 1e3))
        except Exception as e:
            line.append("%s -" % tile)
    print("%dx%d %d->%d%s x%d | " % (h, w_, cin, co, " +skip" if r_ else "", cnt) + "   ".join(line), flush=True)
