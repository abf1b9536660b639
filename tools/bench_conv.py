#!/usr/bin/env python3
"""Single fused-conv micro-benchmark (the layer classes of SURVEY §8 d').
    python tools/bench_conv.py [--iters 50] [--only NAME] [--tile auto] [--splits 0]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from betapose_amd import ops

LAYERS = {
    # name: (H, W, Cin, Cout, k, stride)
    "y3x3_3_32_416": (416, 416, 3, 32, 3, 1),
    "k7x7_3_64_s2_320x256": (320, 256, 3, 64, 7, 2),
    "y3x3_32_64_s2_416": (416, 416, 32, 64, 3, 2),
    "y3x3_64_128_104": (104, 104, 64, 128, 3, 1),
    "y1x1_256_128_52": (52, 52, 256, 128, 1, 1),
    "y3x3_128_256_52": (52, 52, 128, 256, 3, 1),
    "y3x3_256_512_26": (26, 26, 256, 512, 3, 1),
    "y1x1_1024_512_13": (13, 13, 1024, 512, 1, 1),
    "y3x3_512_1024_13": (13, 13, 512, 1024, 3, 1),
    "k3x3_256_256_20x16": (20, 16, 256, 256, 3, 1),
    "k1x1_256_1024_20x16": (20, 16, 256, 1024, 1, 1),
    "kduc1_512_1024_20x16": (20, 16, 512, 1024, 3, 1),
}
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--only", default="")
ap.add_argument("--tile", default="auto")
ap.add_argument("--splits", type=int, default=0)
ap.add_argument("--batch", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for name, (H, W, Cin, Cout, k, st) in LAYERS.items():
    if a.only and a.only != name:
        continue
    x = torch.randn(a.batch, H, W, Cin, generator=g).to(dev)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k))
    b = torch.randn(Cout, generator=g)
    out, ms = ops.conv2d_nhwc(x, w, b, stride=st, pad=(k - 1) // 2, act="leaky", tile=a.tile, splits=a.splits, iters=a.iters)
    OH, OW = out.shape[1], out.shape[2]
    fl = 2.0 * a.batch * OH * OW * Cout * Cin * k * k
    print("%-24s M=%6d N=%5d K=%5d  %8.2f us  %6.1f TF/s" % (name, a.batch * OH * OW, Cout, Cin * k * k, ms * 1e3, fl / ms / 1e9), flush=True)
