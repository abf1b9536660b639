#!/usr/bin/env python3
"""Soak test of the in-kernel split-K hand-off (write-through slabs + ticket) under load: N frames in flight on
separate streams, thousands of replays of the same frames; every result record must be bit-identical to the first
pass (slices are summed in slice order, so any difference is a stale/torn slab read)."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from betapose_amd import cfg as C, synth
from betapose_amd.darknet import Darknet
from betapose_amd.kpd import FastPoseHIP
from betapose_amd.pipeline import FramePipeline

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=3000)
ap.add_argument("--streams", type=int, default=4)
ap.add_argument("--precision", default="bf16x3", choices=["f32", "bf16x3", "f16", "f16r"])
ap.add_argument("--latency-mode", action="store_true",
                help="bp_*_set_prefetch: split-K hand-off inside one XCD's L2 + filter prefetch blocks; the reference pass is taken WITHOUT it, "
                     "so the soak also proves the mode bit-identical to the default under load")
a = ap.parse_args()
dev = torch.device("cuda:0")
blocks = C.parse_cfg_text(C.yolov3_single_cfg_text())
det = Darknet("yolov3-single.cfg").load_stream(synth.synth_yolo_stream(1, blocks)).cuda()
pose = FastPoseHIP(synth.synth_fastpose_state_dict(2)).cuda()
det.set_precision(a.precision)
pose.set_precision(a.precision)
S = a.streams
dets = [det] + [det.clone() for _ in range(S - 1)]
poses = [pose] + [pose.clone() for _ in range(S - 1)]
pipes = [FramePipeline(dets[k], poses[k], 480, 640, keep_heatmaps=True) for k in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
frames = [torch.from_numpy(synth.synth_frame(1234 + k)[None]).to(dev) for k in range(S)]
ref_rec, ref_hm = [], []
for k in range(S):
    pipes[k].frames.copy_(frames[k]); pipes[k].enqueue(); torch.cuda.synchronize()
    ref_rec.append(pipes[k].results.clone()); ref_hm.append(pipes[k].heatmaps.clone())
if a.latency_mode:
    for d_, p_ in zip(dets, poses):
        d_.set_prefetch(True)
        p_.set_prefetch(True)
bad = 0
for it in range(a.iters):
    for k in range(S):
        with torch.cuda.stream(streams[k]):
            pipes[k].frames.copy_(frames[k], non_blocking=True)
            pipes[k].enqueue(streams[k].cuda_stream)
    if it % 50 == 49 or it == a.iters - 1:
        torch.cuda.synchronize()
        for k in range(S):
            if not torch.equal(pipes[k].results, ref_rec[k]) or not torch.equal(pipes[k].heatmaps, ref_hm[k]):
                bad += 1
                d = (pipes[k].heatmaps - ref_hm[k]).abs().max().item()
                print("MISMATCH iter", it, "stream", k, "max |d hm|", d, flush=True)
print("soak[%s%s]: %d iterations x %d streams, mismatching checks: %d" % (a.precision, ", latency mode" if a.latency_mode else "", a.iters, S, bad))
sys.exit(1 if bad else 0)
