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

F16 = "--f16" in sys.argv          # tune the fp16-operand kernels instead of the bf16x3 (fp32-accurate) ones
FP32 = "--fp32" in sys.argv        # ... or the fp32-MFMA kernel (64x64 tile, slices only)
SUF = "_f16" if F16 else "_b3"
TILES = ["64x64"] if FP32 else ["64x64" + SUF, "w1x1" + SUF, "w1x2" + SUF, "w2x1" + SUF, "w2x2" + SUF] + (["128x64_f16", "bd_f16"] if F16 else ["bd_b3"])
if "--kg" in sys.argv:             # batch-1 candidates only: the 64x64-block kernel against the K-group kernels
    TILES = ["64x64_b3", "bd_b3", "kg2_b3", "rd4_b3"]
PL = "--pl" in sys.argv            # the operand-plane kernels (conv_pl.hip), the product's 16-bit kernels
if PL:
    TILES = [t + SUF for t in (["pl64", "pl128x64"] + (["pl128", "pl256x128", "pl128s"] if "--big" in sys.argv else []))]
ENGINE_LIKE = PL                   # residual after the activation + operand planes emitted, as most layers of the networks run
TILE_ID = {"pl64": 13, "pl128": 14, "pl128x64": 15, "pl256x128": 16, "pl128s": 17, "pl64k2": 18, "pl64bd": 19, "64x64": 0, "128x64": 1, "w1x1": 2, "w1x2": 3, "w2x1": 5, "w2x2": 6, "kg1": 7, "kg2": 8, "kg4": 9, "rd4": 10, "rd8": 11, "bd": 12}
BMN = {"pl64": (64, 64), "pl128": (128, 128), "pl128x64": (128, 64), "pl256x128": (256, 128), "pl128s": (128, 128), "pl64k2": (64, 64), "pl64bd": (64, 64), "64x64": (64, 64), "128x64": (128, 64), "w1x1": (64, 64), "w1x2": (64, 128), "w2x1": (128, 64), "w2x2": (128, 128),
       "kg1": (64, 64), "kg2": (64, 64), "kg4": (64, 64), "rd4": (64, 64), "rd8": (64, 64), "bd": (64, 64)}
BATCH = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 1
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
print("// {M, CoutPad, nchunks, tile, splits}  (count, us_best | per tile: best splits:us)")
tot_best = 0.0
for (h, w_, cin, co, k, st), cnt in sorted(shapes.items()):
    if cin % 32:
        continue
    x = torch.randn(BATCH, h, w_, cin, generator=g).to(dev)
    wt = torch.randn(co, cin, k, k, generator=g) / np.sqrt(cin * k * k)
    oh = (h + 2 * ((k - 1) // 2) - k) // st + 1; ow = (w_ + 2 * ((k - 1) // 2) - k) // st + 1
    M = BATCH * oh * ow; nch = cin * k * k // 32; cpad = (co + 63) // 64 * 64
    best = (1e9, None, None); per_tile = []
    res = torch.randn(BATCH, oh, ow, co, generator=g).to(dev) if ENGINE_LIKE else None
    for tile in TILES:
        base = tile.split("_")[0]
        bm, bn = BMN[base]
        if bn > 64 and cpad < bn and base != "w1x2":
            continue                       # a tile twice as wide as the layer
        tiles = -(-M // bm) * -(-cpad // bn)
        tb = (1e9, None)
        for sp in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 24):
            if sp > 1 and base == "pl128s":
                continue                   # (the wave-specialised tile takes no K slices)
            if sp > 1 and (nch // sp < 2 or tiles * sp > 1400 or BATCH > 4):
                continue
            kw = dict(res=res, res_after_act=True, planes=True) if ENGINE_LIKE else {}
            ms = ops.conv2d_nhwc(x, wt, None, stride=st, pad=(k - 1) // 2, act="leaky", splits=sp, iters=30, tile=tile, **kw)[-1]
            if ms * 1e3 < tb[0]:
                tb = (ms * 1e3, sp)
        per_tile.append("%s %d:%.1f" % (base, tb[1], tb[0]))
        if tb[0] < best[0]:
            best = (tb[0], base, tb[1])
    tot_best += best[0] * cnt
    print("    {%6d, %5d, %4d, %d, %2d},   // x%d  %.1f us %s | %s" % (M, cpad, nch, TILE_ID[best[1]], best[2], cnt, best[0], best[1],
          "  ".join(per_tile)), flush=True)
print("// sum best %.1f us" % tot_best)
