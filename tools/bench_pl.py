#!/usr/bin/env python3
"""Kernel-against-kernel timing of single convolutions: the operand-plane kernels (conv_pl.hip) beside the round-2 ones.
    python tools/bench_pl.py [--batch 1] [--mode b3|f16] [--tiles pl64,pl128,...] [--iters 30] [--shapes big|all]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from betapose_amd import ops

SHAPES = {
    # name: (H, W, Cin, Cout, k, stride)
    "y3x3_32_64_s2_416": (416, 416, 32, 64, 3, 2),
    "y3x3_64_128_104": (104, 104, 64, 128, 3, 1),
    "y1x1_256_128_52": (52, 52, 256, 128, 1, 1),
    "y3x3_128_256_52": (52, 52, 128, 256, 3, 1),
    "y3x3_256_512_26": (26, 26, 256, 512, 3, 1),
    "y1x1_1024_512_13": (13, 13, 1024, 512, 1, 1),
    "y3x3_512_1024_13": (13, 13, 512, 1024, 3, 1),
    "k1x1_1024_256_20x16": (20, 16, 1024, 256, 1, 1),
    "k3x3_256_256_20x16": (20, 16, 256, 256, 3, 1),
    "k1x1_256_1024_20x16": (20, 16, 256, 1024, 1, 1),
    "k3x3_128_128_40x32": (40, 32, 128, 128, 3, 1),
    "kduc1_512_1024_20x16": (20, 16, 512, 1024, 3, 1),
}
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--mode", default="b3")
ap.add_argument("--tiles", default="")
ap.add_argument("--only", default="")
ap.add_argument("--splits", default="1,2,3,4,5,6,8,10")
ap.add_argument("--engine-like", action="store_true", help="residual added after the activation + operand planes emitted, as inside the networks")
ap.add_argument("--all-splits", action="store_true", help="print every slice count, not only the best")
a = ap.parse_args()
tiles = a.tiles.split(",") if a.tiles else (["bd", "pl64", "pl128x64", "pl128"] if a.mode == "b3" else ["64x64", "w2x2", "pl64", "pl128x64", "pl128", "pl256x128"])
BMN = {"bd": (64, 64), "64x64": (64, 64), "w2x2": (128, 128), "pl64": (64, 64), "pl128": (128, 128), "pl128x64": (128, 64), "pl256x128": (256, 128), "pl128s": (128, 128), "pl64k2": (64, 64), "pl64bd": (64, 64), "128x64": (128, 64), "plh128": (128, 128), "s1": (32, 128)}
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
peak = 2500.0 / (6 if a.mode == "b3" else 1)
for name, (H, W, Cin, Cout, k, st) in SHAPES.items():
    if a.only and a.only not in name:
        continue
    x = torch.randn(a.batch, H, W, Cin, generator=g).to(dev)
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    OH, OW = (H + 2 * (k // 2) - k) // st + 1, (W + 2 * (k // 2) - k) // st + 1
    M = a.batch * OH * OW
    nch = Cin * k * k // 32
    fl = 2.0 * M * Cout * Cin * k * k
    line = "%-22s M=%7d N=%5d K=%5d |" % (name, M, Cout, Cin * k * k)
    kw = {}
    if a.engine_like:
        kw = dict(res=torch.randn(a.batch, OH, OW, Cout, generator=g).to(dev), res_after_act=True, planes=True)
    for t in tiles:
        bm, bn = BMN[t]
        blocks = -(-M // bm) * -(-((Cout + 63) // 64 * 64) // bn)
        best = (1e9, 0)
        for sp in [int(s) for s in a.splits.split(",")]:
            if sp > 1 and (nch // sp < 2 or blocks * sp > 2048):
                continue
            try:
                ms = ops.conv2d_nhwc(x, w, b, stride=st, pad=k // 2, act="leaky", tile=t + "_" + a.mode, splits=sp, iters=a.iters, **kw)[-1]
                if a.all_splits:
                    line += " s%d:%.1f" % (sp, ms * 1e3)
            except Exception as e:
                ms = float("nan")
            if ms < best[0]:
                best = (ms, sp)
        line += " %s %7.1f us (s%d, %5.1f%%)" % (t, best[0] * 1e3, best[1], 100 * fl / (best[0] * 1e-3) / 1e12 / peak)
    print(line, flush=True)
