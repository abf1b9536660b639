#!/usr/bin/env python3
"""A/B of the convolution kernels on the layer classes of the two networks:
    python tools/bench_w64.py [--iters 50] [--only NAME] [--tiles 64x64,64x64_b3,w1x2_b3,...] [--splits 0,1,3]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from betapose_amd import ops

LAYERS = {
    # name: (H, W, Cin, Cout, k, stride)
    "y3x3_32_64_s2_416": (416, 416, 32, 64, 3, 2),
    "y1x1_64_32_208": (208, 208, 64, 32, 1, 1),
    "y3x3_64_128_104": (104, 104, 64, 128, 3, 1),
    "y1x1_256_128_52": (52, 52, 256, 128, 1, 1),
    "y3x3_128_256_52": (52, 52, 128, 256, 3, 1),
    "y3x3_256_512_26": (26, 26, 256, 512, 3, 1),
    "y1x1_1024_512_13": (13, 13, 1024, 512, 1, 1),
    "y3x3_512_1024_13": (13, 13, 512, 1024, 3, 1),
    "k1x1_1024_256_20x16": (20, 16, 1024, 256, 1, 1),
    "k3x3_256_256_20x16": (20, 16, 256, 256, 3, 1),
    "k1x1_256_1024_20x16": (20, 16, 256, 1024, 1, 1),
    "kduc1_512_1024_20x16": (20, 16, 512, 1024, 3, 1),
}
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--only", default="")
ap.add_argument("--tiles", default="64x64,64x64_b3,w1x1_b3,w1x2_b3,w2x1_b3,w2x2_b3")
ap.add_argument("--splits", default="0")
ap.add_argument("--batch", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for name, (H, W, Cin, Cout, k, st) in LAYERS.items():
    if a.only and a.only not in name:
        continue
    x = torch.randn(a.batch, H, W, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k))
    b = torch.randn(Cout, generator=g)
    for tile in a.tiles.split(","):
        for sp in [int(v) for v in a.splits.split(",")]:
            try:
                out, ms = ops.conv2d_nhwc(x, w, b, stride=st, pad=(k - 1) // 2, act="leaky", tile=tile, splits=sp, iters=a.iters)
            except Exception as e:   # noqa: BLE001
                print("%-24s %-10s splits=%d: %s" % (name, tile, sp, str(e)[:60]))
                continue
            OH, OW = out.shape[1], out.shape[2]
            fl = 2.0 * a.batch * OH * OW * Cout * Cin * k * k
            print("%-24s M=%6d N=%5d K=%5d %-10s splits=%2d %8.2f us  %6.1f TF/s" % (
                name, a.batch * OH * OW, Cout, Cin * k * k, tile, sp, ms * 1e3, fl / ms / 1e9), flush=True)
