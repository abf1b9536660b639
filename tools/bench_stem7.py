#!/usr/bin/env python3
"""The key-point detector's 7x7 / stride-2 RGB stem at 28 frames per launch: the fp32 MFMA kernel against the fp16 im2col kernel (TILE_STEM7)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from betapose_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 28
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for cin in (3, 4):      # 3: the engine's crop tensor (three floats per pixel); 4: a padded one
    x = torch.randn(B, 320, 256, cin, generator=g).to(dev)
    w = torch.randn(64, cin, 7, 7, generator=g) / 14
    for tile in ("64x64", "stem7"):
        us = ops.conv2d_nhwc(x, w, None, stride=2, pad=3, act="relu", iters=20, tile=tile)[-1] * 1e3
        print("cin %d  %-6s %.1f us" % (cin, tile, us), flush=True)
