#!/usr/bin/env python3
"""Per-op device time of both networks (HIP events, eager): where the frame time goes.
    python tools/profile_ops.py [--batch 1] [--sk-target 512] [--sk-min 4] [--tile -1]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from betapose_amd import cfg as C, synth
from betapose_amd.darknet import Darknet
from betapose_amd.kpd import FastPoseHIP

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--sk-target", type=int, default=512)
ap.add_argument("--sk-min", type=int, default=4)
ap.add_argument("--tile", type=int, default=-1)
ap.add_argument("--sk-max", type=int, default=8)
ap.add_argument("--top", type=int, default=400)
ap.add_argument("--precision", default="bf16x3")
a = ap.parse_args()
blocks = C.parse_cfg_text(C.yolov3_single_cfg_text())
det = Darknet("yolov3-single.cfg", max_batch=a.batch).load_stream(synth.synth_yolo_stream(1, blocks)).cuda()
pose = FastPoseHIP(synth.synth_fastpose_state_dict(2), max_batch=a.batch).cuda()
for name, net in (("yolo", det), ("kpd", pose)):
    net.set_precision(a.precision)
    net.set_policy(a.sk_target, a.sk_min, a.sk_max, a.tile)
    ms, info = net.profile(a.batch, 20)
    fl, by = net.op_stats()
    tot = ms.sum()
    conv = info[:, 0] == 1
    print("== %s: %d ops, %.3f ms total, conv %.3f ms, %.1f GFLOP -> %.1f TF/s on convs" % (
        name, len(ms), tot, ms[conv].sum(), fl.sum() * a.batch / 1e9, fl[conv].sum() * a.batch / ms[conv].sum() / 1e9))
    for i in range(min(len(ms), a.top)):
        tf = fl[i] * a.batch / (ms[i] * 1e-3) / 1e12 if ms[i] > 0 else 0
        print("%3d conv=%d tile=%d vec=%d splits=%2d  %8.1f us  %7.3f GFLOP  %6.1f TF/s  %6.2f MB" % (
            i, info[i, 0], info[i, 1], info[i, 2], info[i, 3], ms[i] * 1e3, fl[i] * a.batch / 1e9, tf, by[i] / 1e6))
