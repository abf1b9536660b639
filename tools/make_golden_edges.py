#!/usr/bin/env python3
"""Edge-case golden vectors from the REFERENCE's own functions (build container only; shims as tools/make_golden.py):

  * ``crop_from_dets`` / ``cropBox`` (dataloader.py:794-835, KPD/src/utils/img.py:242-262) on 64 seeded boxes of every
    flavour -- tiny, huge, touching or crossing the frame border, fractional corners, both sides of the width-100
    pad-rule switch.  Stored per box: pt1, pt2, crop sum / abs-sum and 96 sampled crop values.  Boxes the reference
    itself cannot process (it prints and leaves the slot empty, :825-831, or raises) are recorded as such.
  * ``dynamic_write_results`` (yolo/util.py:104-223) on 48 seeded prediction tensors: single / multi class, images
    without a candidate above the threshold, several images per batch.  (No exact objectness ties: the reference ranks
    with an unstable sort, so its pick among ties is unspecified.)

Writes tests/golden/edges.npz."""
import os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_shims  # noqa: E402

ref_shims.install()
import torch  # noqa: E402

os.chdir(ref_shims.REF)
sys.path.insert(0, ref_shims.REF)
sys.argv = [sys.argv[0]]
from opt import opt  # noqa: E402
import dataloader as ref_dl  # noqa: E402
from yolo.util import dynamic_write_results  # noqa: E402
from KPD.src.utils.img import im_to_torch  # noqa: E402
from betapose_amd import synth  # noqa: E402

out = {}
g = np.random.Generator(np.random.PCG64(77))

# ---------------------------------------------------------------- crop
frame = synth.synth_frame(4321)                       # BGR u8 480x640
boxes = []
for _ in range(40):
    cx, cy = g.uniform(-30, 670), g.uniform(-30, 510)
    w, h = np.exp(g.uniform(np.log(3), np.log(650))), np.exp(g.uniform(np.log(3), np.log(470)))
    boxes.append([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2])
for w in (99.0, 100.0, 100.5, 101.0):
    for x0 in (0.0, 33.3, 539.5):
        boxes.append([x0, 120.25, x0 + w, 300.5])
for _ in range(12):
    x0, y0 = int(g.integers(0, 600)), int(g.integers(0, 440))
    boxes.append([float(x0), float(y0), float(x0 + g.integers(1, 200)), float(y0 + g.integers(1, 200))])
boxes = np.array(boxes, np.float32)
samp = g.choice(3 * 320 * 256, 96, replace=False)
ok, pt1s, pt2s, sums, asums, samples = [], [], [], [], [], []
for b in boxes:
    img = im_to_torch(np.ascontiguousarray(frame[:, :, ::-1]))          # RGB float CHW /255, as dataloader.py:452
    inps = torch.full((1, 3, opt.inputResH, opt.inputResW), float("nan"))
    pt1, pt2 = torch.zeros(1, 2), torch.zeros(1, 2)
    try:
        ref_dl.crop_from_dets(img, torch.from_numpy(b[None]), inps, pt1, pt2)
        good = bool(torch.isfinite(inps).all())
    except Exception as e:                                              # noqa: BLE001
        good = False
    ok.append(good)
    pt1s.append(pt1[0].numpy().copy()); pt2s.append(pt2[0].numpy().copy())
    flat = inps[0].reshape(-1).double()
    sums.append(float(flat.sum()) if good else 0.0)
    asums.append(float(flat.abs().sum()) if good else 0.0)
    samples.append(inps[0].reshape(-1)[samp].numpy().copy() if good else np.zeros(96, np.float32))
out["crop_frame_seed"] = np.array(4321)
out["crop_boxes"], out["crop_ok"] = boxes, np.array(ok)
out["crop_pt1"], out["crop_pt2"] = np.array(pt1s), np.array(pt2s)
out["crop_sum"], out["crop_abs_sum"] = np.array(sums), np.array(asums)
out["crop_samp_idx"], out["crop_samples"] = samp, np.array(samples)
print("crop: %d boxes, reference handled %d" % (len(boxes), int(np.sum(ok))))

# ---------------------------------------------------------------- select
n_cases = 48
out["sel_n"] = np.array(n_cases)
kept = 0
for t in range(n_cases):
    B = int(g.integers(1, 4))
    rows = 60
    ncls = int(g.choice([1, 3]))
    pred = np.zeros((B, rows, 5 + ncls), np.float32)
    pred[..., 0:2] = g.uniform(0, 416, (B, rows, 2))
    pred[..., 2:4] = g.uniform(2, 300, (B, rows, 2))
    pred[..., 4] = g.uniform(0, 1, (B, rows)) ** 3
    pred[..., 5:] = g.uniform(0, 1, (B, rows, ncls))
    conf = float(g.choice([0.01, 0.5, 0.9]))
    if t % 4 == 0:
        pred[0, :, 4] *= conf * 0.5
    if t % 7 == 0:
        pred[:, :, 4] *= conf * 0.5                       # nothing anywhere -> int 0
    res = dynamic_write_results(torch.from_numpy(pred.copy()), conf, 80, nms=True, nms_conf=0.6)
    out["sel%d_pred" % t] = pred
    out["sel%d_conf" % t] = np.array(conf)
    if isinstance(res, int):
        out["sel%d_out" % t] = np.zeros((0, 8), np.float32)
    else:
        out["sel%d_out" % t] = res.numpy().astype(np.float32)
        kept += 1
print("select: %d cases, %d with detections" % (n_cases, kept))
# ---------------------------------------------------------------- getPrediction on planted heat-maps
from KPD.src.utils.eval import getPrediction  # noqa: E402
n, K, H, Wd = 2, 50, 80, 64
hm = g.normal(0, 0.3, (n, K, H, Wd)).astype(np.float32)
for k, (y, x) in enumerate([(0, 0), (0, 63), (79, 0), (79, 63), (0, 30), (79, 31), (40, 0), (41, 63)]):
    hm[0, k, y, x] = 5.0                                   # maxima on corners / borders: no quarter-pixel shift
for k in range(8, 16):                                     # two equal maxima
    a, b = sorted(g.choice(H * Wd, 2, replace=False))
    hm[0, k].reshape(-1)[[a, b]] = 4.0
hm[1, :10] = -np.abs(hm[1, :10]) - 0.01                    # all negative -> zeroed key point
hm[1, 10:12] = 0.0                                         # all zero
for k in range(12, 20):                                    # equal left/right neighbours: sign(0) = 0
    y, x = int(g.integers(1, H - 1)), int(g.integers(1, Wd - 1))
    hm[1, k, y, x] = 6.0
    hm[1, k, y, x - 1] = hm[1, k, y, x + 1] = 1.5
    hm[1, k, y - 1, x], hm[1, k, y + 1, x] = 0.5, 2.5
hm = hm.astype(np.float16).astype(np.float32)              # stored as fp16: those values ARE the input
pt1 = g.uniform(0, 200, (n, 2)).astype(np.float32)
pt2 = pt1 + g.uniform(20, 300, (n, 2)).astype(np.float32)
a_, b_, c_ = getPrediction(torch.from_numpy(hm), torch.from_numpy(pt1), torch.from_numpy(pt2), 320, 256, 80, 64)
out["gpe_hms"], out["gpe_pt1"], out["gpe_pt2"] = hm.astype(np.float16), pt1, pt2
out["gpe_preds_hm"], out["gpe_preds_img"], out["gpe_maxval"] = a_.numpy(), b_.numpy(), c_.numpy()
print("getPrediction planted:", a_.shape)
# ---------------------------------------------------------------- pose_nms on seeded candidate sets
import pPose_nms as ref_nms  # noqa: E402
n_nms = 24
out["nms_n"] = np.array(n_nms)
for t in range(n_nms):
    npose = int(g.integers(1, 7))
    base = g.uniform(100, 300, (50, 2)).astype(np.float32)
    poses, boxes_, bsc_ = [], [], []
    for j in range(npose):
        kind = int(g.integers(0, 4))
        if kind == 0:   pose = base + g.normal(0, 0.7, (50, 2))                 # near-duplicate of the base pose
        elif kind == 1: pose = base + g.normal(0, 6.0, (50, 2))                 # loosely similar
        elif kind == 2: pose = base + g.uniform(40, 120)                        # shifted away
        else:           pose = g.uniform(50, 400, (50, 2))                      # unrelated
        poses.append(pose.astype(np.float32))
        lo, hi = pose.min(0) - 5, pose.max(0) + 5
        boxes_.append([lo[0], lo[1], hi[0], hi[1]])
        bsc_.append([float(g.uniform(0.3, 0.99))])
    poses = np.stack(poses).astype(np.float32)
    psc = g.uniform(0.05, 0.95, (npose, 50, 1)).astype(np.float32)
    if t % 5 == 0:
        psc[int(g.integers(0, npose))] *= 0.2                                   # a weak pose (max score < 0.3)
    if t % 6 == 0:
        psc[0, :5] = 0.0                                                        # exact zeros -> 1e-5 (pPose_nms.py:36)
    bxs = np.array(boxes_, np.float32); bsc = np.array(bsc_, np.float32)
    res = ref_nms.pose_nms(torch.from_numpy(bxs.copy()), torch.from_numpy(bsc.copy()), torch.from_numpy(poses.copy()),
                           torch.from_numpy(psc.copy()))
    out["nms%d_boxes" % t], out["nms%d_bsc" % t], out["nms%d_poses" % t], out["nms%d_psc" % t] = bxs, bsc, poses, psc
    out["nms%d_n" % t] = np.array(len(res))
    for j, r in enumerate(res):
        out["nms%d_o%d_kp" % (t, j)] = r["keypoints"].numpy()
        out["nms%d_o%d_score" % (t, j)] = r["kp_score"].numpy()
        out["nms%d_o%d_prop" % (t, j)] = np.array(float(r["proposal_score"]))
        out["nms%d_o%d_bbox" % (t, j)] = np.asarray(r["bbox"].numpy() if hasattr(r["bbox"], "numpy") else r["bbox"])
print("pose_nms:", n_nms, "cases, outputs", [int(out["nms%d_n" % t]) for t in range(n_nms)])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "edges.npz"), **out)
print(os.path.getsize(os.path.join(ROOT, "tests", "golden", "edges.npz")), "bytes")
