#!/usr/bin/env python3
"""Is the batched 3x3 kernel bound by the matrix pipe or by the power budget?  The 52x52 128 -> 256 layer at 28 frames per launch, 3 000 launches
back to back, with random and with zero-filled operands (MI355X_MICROARCH.md "DVFS give-back": the chip clocks to its power budget), shader clock
and socket power sampled meanwhile.  python tools/p3_clock_probe.py [tile ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from betapose_amd import ops
tiles = [a for a in sys.argv[1:]] or ["p3", "plh128"]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, h, w_, cin, co = 28, 52, 52, 128, 256
for data in ("random", "zeros"):
    x = (torch.randn(B, h, w_, cin, generator=g) if data == "random" else torch.zeros(B, h, w_, cin)).to(dev)
    wt = torch.randn(co, cin, 3, 3, generator=g) / np.sqrt(cin * 9) if data == "random" else torch.zeros(co, cin, 3, 3)
    for tile in tiles:
        ops.conv2d_nhwc(x, wt, None, pad=1, act="leaky", splits=1, iters=200, tile=tile + "_f16")
        with bench.ClockSampler(period=0.25) as cs:
            us = ops.conv2d_nhwc(x, wt, None, pad=1, act="leaky", splits=1, iters=40000, tile=tile + "_f16")[-1] * 1e3
        s = cs.summary() or {}
        fl = 2.0 * B * h * w_ * co * cin * 9
        print("%-7s %-7s %6.1f us  %5.0f TFLOP/s  sclk p50 %s MHz (min %s)  socket %s W" % (tile, data, us, fl / us / 1e6, s.get("sclk_MHz_p50"), s.get("sclk_MHz_min"), s.get("package_W_p50")), flush=True)
