#!/usr/bin/env python3
"""The persistent 3x3 kernel (TILE_P3, conv_p3.hip) against the halo plane tile (TILE_PLH128) on the 3x3 / stride-1 layers of both networks at
28 frames per launch (configs[2]), one kernel at a time.  python tools/bench_p3.py [batch] [--nores]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from betapose_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 28
NORES = "--nores" in sys.argv
# (H, W, Cin, Cout, launches per frame)
SHAPES = [(52, 52, 128, 256, 11), (26, 26, 256, 512, 11), (13, 13, 512, 1024, 8), (20, 16, 256, 256, 22), (40, 32, 128, 128, 3), (104, 104, 64, 128, 2)]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
tot = {}
for (h, w_, cin, co, cnt) in SHAPES:
    x = torch.randn(B, h, w_, cin, generator=g).to(dev)
    wt = torch.randn(co, cin, 3, 3, generator=g) / np.sqrt(cin * 9)
    res = None if NORES else torch.randn(B, h, w_, co, generator=g).to(dev)
    line = []
    outs = {}
    for tile in ("plh128", "p3"):
        # BP_CONV_F16R=1: as the engine's f16r plan launches these layers (skip connection from its fp16 plane, fp16 plane out, no fp32 store)
        f16r = os.environ.get("BP_CONV_F16R") is not None
        r = ops.conv2d_nhwc(x, wt, None, pad=1, act="leaky", res=res, res_after_act=True, splits=1, iters=20, tile=tile + "_f16", planes=f16r)
        outs[tile] = r[1] if f16r else r[0]
        us = r[-1] * 1e3
        tot[tile] = tot.get(tile, 0.0) + us * cnt
        fl = 2.0 * B * h * w_ * co * cin * 9
        line.append("%s %.1f us (%.0f TFLOP/s)" % (tile, us, fl / us / 1e6))
    same = bool(torch.equal(outs["plh128"], outs["p3"]))
    print("%dx%d %d->%d x%d | " % (h, w_, cin, co, cnt) + "   ".join(line) + "   bit-identical: %s" % same, flush=True)
print("sum per 28-frame pass (us): " + ", ".join("%s %.1f" % kv for kv in tot.items()))
