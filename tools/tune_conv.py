#!/usr/bin/env python3
"""Per-shape split-K tuning: time every distinct conv shape of both networks alone with each candidate slice count
and print the table engine.cpp embeds (kSplitTable).  Run on an MI355X: python tools/tune_conv.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from betapose_amd import cfg as C, ops, weights as W

shapes = {}
# YOLO
blocks = C.parse_cfg_text(C.yolov3_single_cfg_text())
H = 416; hw = []
cur = 416; chans = 3
sizes = {}
outs = []
for i, b in enumerate(blocks):
    if b["type"] == "convolutional":
        k, st, co = int(b["size"]), int(b["stride"]), int(b["filters"])
        cin = chans
        key = (cur, cur, cin, co, k, st)
        shapes[key] = shapes.get(key, 0) + 1
        cur = (cur + 2 * ((k - 1) // 2) - k) // st + 1
        chans = co
    elif b["type"] == "route":
        ls = [int(a) for a in b["layers"].split(",")]
        if len(ls) == 1:
            cur, chans = outs[i + ls[0]]
        else:
            cur = outs[i + ls[0]][0]; chans = outs[i + ls[0]][1] + outs[ls[1]][1]
    elif b["type"] == "upsample":
        cur *= 2
    outs.append((cur, chans))
# KPD (FastPose) shapes
def add(h, w, cin, co, k, st):
    key = (h, w, cin, co, k, st); shapes[key] = shapes.get(key, 0) + 1
h, w_, inpl = 80, 64, 64
for planes, nb, st in W.FASTPOSE_STAGES:
    for bi in range(nb):
        s = st if bi == 0 else 1
        add(h, w_, inpl, planes, 1, 1); add(h, w_, planes, planes, 3, s)
        h2, w2 = h // s, w_ // s
        add(h2, w2, planes, planes * 4, 1, 1)
        if bi == 0: add(h, w_, inpl, planes * 4, 1, s)
        h, w_, inpl = h2, w2, planes * 4
add(20, 16, 512, 1024, 3, 1); add(40, 32, 256, 512, 3, 1); add(80, 64, 128, 50, 3, 1)

F16 = "--f16" in sys.argv          # tune the fp16-MFMA kernel instead
B3 = "--b3" in sys.argv            # ... or the bf16x3 kernel
TILE = "64x64_f16" if F16 else ("64x64_b3" if B3 else "64x64")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
print("// {M, CoutPad, nchunks, splits}  (count, us_best, us_default)")
tot_best = tot_def = 0.0
for (h, w_, cin, co, k, st), cnt in sorted(shapes.items()):
    if cin % 32:
        continue
    x = torch.randn(1, h, w_, cin, generator=g).to(dev)
    wt = torch.randn(co, cin, k, k, generator=g) / np.sqrt(cin * k * k)
    oh = (h + 2 * ((k - 1) // 2) - k) // st + 1; ow = (w_ + 2 * ((k - 1) // 2) - k) // st + 1
    M = oh * ow; nch = cin * k * k // 32; cpad = (co + 63) // 64 * 64
    res = {}
    for sp in (0, 1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
        if sp > 1 and nch // sp < 2:
            continue
        _, ms = ops.conv2d_nhwc(x, wt, None, stride=st, pad=(k - 1) // 2, act="leaky", splits=sp, iters=30, tile=TILE)
        res[sp] = ms * 1e3
    best = min((v, s) for s, v in res.items() if s > 0)
    tot_best += best[0] * cnt; tot_def += res[0] * cnt
    print("    {%6d, %5d, %4d, %2d},   // x%d  %.1f us (auto %.1f)  %s" % (M, cpad, nch, best[1], cnt, best[0], res[0],
          " ".join("%d:%.1f" % (s, v) for s, v in sorted(res.items()))))
print("// sum best %.1f us, sum auto %.1f us" % (tot_best, tot_def))
