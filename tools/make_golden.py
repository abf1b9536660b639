#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python.

Build-container only (needs /root/reference; see tools/ref_shims.py for how the
reference is imported without copying it).  The reference's real stage classes
(``ImageLoader`` -> ``DetectionLoader`` -> ``DetectionProcessor``), its
``Darknet`` / ``FastPose`` modules and ``getPrediction`` / ``pose_nms`` /
``write_json`` are driven on seeded synthetic frames and weights
(betapose_amd.synth); what they produce is stored as small fixtures:

  pipeline.npz   per frame: resized YOLO input samples, head rows, arg-max index,
                 detection row, rescaled box, crop window, crop samples,
                 heat-map samples, key-point arg-max pixels / maxima / neighbours,
                 getPrediction outputs, pose_nms outputs
  formats.npz    .weights round trip through the reference loader, cfg digest
  post.npz       getPrediction / pose_nms (n=3) / metrics on random inputs
  MANIFEST.json  what produced each file, seeds, which third-party steps are
                 restated rather than run

Usage: python tools/make_golden.py [--frames 4]
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=4)
args_cli = ap.parse_args()

import ref_shims  # noqa: E402

ref_shims.install()

import torch  # noqa: E402
from PIL import Image  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)

from betapose_amd import synth, weights as W, cfg as C  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
TMP = tempfile.mkdtemp(prefix="golden_")

YOLO_SEED, KPD_SEED, FRAME_SEED = 1, 2, 1234

# ---------------------------------------------------------------------------
# cv2 stand-ins that are pure data movement (decode, channel swap)
# ---------------------------------------------------------------------------
import cv2  # the stub  # noqa: E402


def _imread(path):
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])


cv2.imread = _imread
cv2.COLOR_BGR2RGB = 4
cv2.cvtColor = lambda img, code: np.ascontiguousarray(img[:, :, ::-1])

# ---------------------------------------------------------------------------
# reference imports
# ---------------------------------------------------------------------------
from opt import opt  # noqa: E402
import dataloader as ref_dl  # noqa: E402
from yolo.darknet import Darknet as RefDarknet, parse_cfg as ref_parse_cfg  # noqa: E402
from yolo.util import dynamic_write_results  # noqa: E402
from KPD.src.models.FastPose import FastPose as RefFastPose  # noqa: E402
from KPD.src.utils.eval import getPrediction  # noqa: E402
import pPose_nms as ref_nms  # noqa: E402
from utils import metrics as ref_metrics  # noqa: E402

opt.inputpath = os.path.join(TMP, "frames")
os.makedirs(opt.inputpath)
manifest = {"generator": "tools/make_golden.py", "torch": torch.__version__,
            "numpy": np.__version__, "pillow": Image.__version__ if hasattr(Image, "__version__") else "",
            "seeds": {"yolo": YOLO_SEED, "kpd": KPD_SEED, "frames": FRAME_SEED},
            "restated_third_party": ["torchsample.SpecialCrop/Pad (tools/ref_shims.py)",
                                     "cv2.imread/cvtColor -> PIL decode + channel swap"],
            "not_run": ["cv2.solvePnP/Rodrigues (OpenCV not installable here)"]}

# ---------------------------------------------------------------------------
# formats: cfg digest + .weights round trip through the reference loader
# ---------------------------------------------------------------------------
ref_cfg_path = os.path.join(ref_shims.REF, "yolo/cfg/yolov3-single.cfg")
ref_blocks = ref_parse_cfg(ref_cfg_path)
cfg_digest = hashlib.sha256(json.dumps(ref_blocks, sort_keys=True).encode()).hexdigest()
blocks = C.parse_cfg_text(C.yolov3_single_cfg_text())
assert [dict(b) for b in ref_blocks] == blocks, "generated cfg != reference cfg"

stream = synth.synth_yolo_stream(YOLO_SEED, blocks)
wpath = os.path.join(TMP, "01.weights")
W.write_darknet_weights(wpath, stream)
ref_net = RefDarknet(ref_cfg_path, reso=416)
ref_net.load_weights(wpath)
ref_net.eval()
convs = W.split_darknet_stream(blocks, stream)
fmt = {"cfg_sha256": np.array(cfg_digest), "stream_sha256": np.array(hashlib.sha256(stream.tobytes()).hexdigest()),
       "stream_size": np.array(stream.size)}
probe_idx = [0, 1, 43, 58, 81, 93, 105]
for c in convs:
    i = c["index"]
    mod = ref_net.module_list[i]
    assert np.array_equal(mod[0].weight.detach().numpy(), c["weight"]), i
    if c["bn"]:
        assert np.array_equal(mod[1].bias.detach().numpy(), c["bn_bias"])
        assert np.array_equal(mod[1].weight.detach().numpy(), c["bn_weight"])
        assert np.array_equal(mod[1].running_mean.numpy(), c["bn_mean"])
        assert np.array_equal(mod[1].running_var.numpy(), c["bn_var"])
    else:
        assert np.array_equal(mod[0].bias.detach().numpy(), c["bias"])
    if i in probe_idx:   # a few values as loaded BY THE REFERENCE, for the reader test
        fmt["w%d_first8" % i] = mod[0].weight.detach().numpy().ravel()[:8].copy()
        fmt["w%d_sum" % i] = np.array(mod[0].weight.detach().double().sum().item())
print("weights round trip through reference loader: OK")

# tiny 5-conv cfg + file, committed whole (reader/writer test without the 246 MB blob)
tiny_cfg = """[convolutional]
batch_normalize=1
filters=8
size=3
stride=1
pad=1
activation=leaky

[convolutional]
batch_normalize=1
filters=16
size=3
stride=2
pad=1
activation=leaky

[convolutional]
batch_normalize=1
filters=8
size=1
stride=1
pad=1
activation=leaky

[convolutional]
batch_normalize=1
filters=16
size=3
stride=1
pad=1
activation=leaky

[shortcut]
from=-3
activation=linear

[convolutional]
size=1
stride=1
pad=1
filters=18
activation=linear

[yolo]
mask = 0,1,2
anchors = 10,13,  16,30,  33,23,  30,61,  62,45,  59,119,  116,90,  156,198,  373,326
classes=1
num=9
jitter=.5
ignore_thresh = .7
truth_thresh = 1
random=1
"""
tiny_blocks = C.parse_cfg_text(tiny_cfg)
tiny_stream = synth.synth_yolo_stream(5, tiny_blocks, head_gain=1.0)
tiny_cfg_path = os.path.join(TMP, "tiny.cfg")
open(tiny_cfg_path, "w").write(tiny_cfg)
tiny_w_path = os.path.join(TMP, "tiny.weights")
W.write_darknet_weights(tiny_w_path, tiny_stream, seen=7)
tiny_net = RefDarknet(tiny_cfg_path, reso=64)
tiny_net.load_weights(tiny_w_path)
tiny_net.eval()
tx = torch.from_numpy(np.random.Generator(np.random.PCG64(11)).uniform(0, 1, (2, 3, 64, 64)).astype(np.float32))
with torch.no_grad():
    tiny_out = tiny_net(tx)
fmt["tiny_cfg"] = np.array(tiny_cfg)
fmt["tiny_weights_bytes"] = np.frombuffer(open(tiny_w_path, "rb").read(), dtype=np.uint8)
fmt["tiny_in"] = tx.numpy()
fmt["tiny_out"] = tiny_out.numpy()
np.savez_compressed(os.path.join(GOLD, "formats.npz"), **fmt)

# ---------------------------------------------------------------------------
# pipeline: run the reference's stage classes on synthetic frames
# ---------------------------------------------------------------------------
nF = args_cli.frames
frames = synth.synth_frames(nF, FRAME_SEED)
names = []
for i, fr in enumerate(frames):
    name = "%04d.png" % i
    Image.fromarray(fr[:, :, ::-1].copy()).save(os.path.join(opt.inputpath, name))
    names.append(name)

# DetectionLoader builds its own Darknet from hard-coded relative paths
# (dataloader.py:289-293); hand it the reference network we already loaded.
class _Factory:
    def __call__(self, cfg_path, reso=416):
        ref_net.load_weights = lambda path: None
        return ref_net


ref_dl.Darknet = _Factory()
captured = {}
_orig_forward = ref_net.forward


def _spy(x, y_true=None):
    out = _orig_forward(x)
    captured.setdefault("yolo_in", []).append(x.clone())
    captured.setdefault("pred", []).append(out.clone())
    return out


ref_net.forward = _spy

data_loader = ref_dl.ImageLoader(names, batchSize=1, format="yolo", reso=416).start()
det_loader = ref_dl.DetectionLoader(data_loader, 1, batchSize=1).start()
det_proc = ref_dl.DetectionProcessor(det_loader).start()

sd_np = synth.synth_fastpose_state_dict(KPD_SEED)
pose_model = RefFastPose()
missing = pose_model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=False)
assert not missing.unexpected_keys and all(k.endswith("num_batches_tracked") for k in missing.missing_keys), missing
pose_model.eval()

rng = np.random.Generator(np.random.PCG64(99))
row_samp = np.sort(rng.choice(10647, 64, replace=False))
crop_samp = rng.integers(0, 3 * 320 * 256, 256)
hm_samp = rng.integers(0, 50 * 80 * 64, 512)
in_samp = rng.integers(0, 3 * 416 * 416, 256)

pipe = {"row_samp": row_samp, "crop_samp": crop_samp, "hm_samp": hm_samp, "in_samp": in_samp,
        "n_frames": np.array(nF)}
results_for_json = []
for i in range(nF):
    (inps, orig_img, im_name, boxes, scores, pt1, pt2) = det_proc.read()
    assert boxes is not None, "synthetic weights produced no detection for frame %d" % i
    pred = captured["pred"][i]
    yin = captured["yolo_in"][i]
    dets = dynamic_write_results(pred, opt.confidence, opt.num_classes, nms=True, nms_conf=opt.nms_thesh)
    k = "f%d_" % i
    pipe[k + "yolo_in_samp"] = yin.numpy().ravel()[in_samp]
    pipe[k + "yolo_in_u8sum"] = np.array(int(torch.round(yin * 255).long().sum()))
    pipe[k + "pred_rows"] = pred[0].numpy()[row_samp]
    pipe[k + "pred_colsum"] = pred[0].double().sum(0).numpy()
    pipe[k + "obj_argmax"] = np.array(int(torch.argmax(pred[0, :, 4])))
    pipe[k + "obj_top2"] = torch.topk(pred[0, :, 4], 2).values.numpy()
    pipe[k + "det_row"] = dets.numpy()
    pipe[k + "boxes"] = boxes.numpy()
    pipe[k + "scores"] = scores.numpy()
    pipe[k + "pt1"] = pt1.numpy()
    pipe[k + "pt2"] = pt2.numpy()
    pipe[k + "crop_samp"] = inps.numpy().ravel()[crop_samp]
    pipe[k + "crop_sum"] = np.array(inps.double().sum().item())
    pipe[k + "crop_abs_sum"] = np.array(inps.double().abs().sum().item())
    with torch.no_grad():
        hm = pose_model(inps).narrow(1, 0, 50)
    flat = hm.view(50, -1)
    mx, idx = flat.max(1)
    nb = np.zeros((50, 4), np.float32)
    for j in range(50):
        x, y = int(idx[j]) % 64, int(idx[j]) // 64
        if 0 < x < 63 and 0 < y < 79:
            nb[j] = [hm[0, j, y, x - 1], hm[0, j, y, x + 1], hm[0, j, y - 1, x], hm[0, j, y + 1, x]]
    srt = torch.sort(flat, 1, descending=True)[0]
    pipe[k + "hm_samp"] = hm.numpy().ravel()[hm_samp]
    pipe[k + "hm_sum"] = np.array(hm.double().sum().item())
    pipe[k + "kp_idx"] = idx.numpy()
    pipe[k + "kp_max"] = mx.numpy()
    pipe[k + "kp_margin"] = (srt[:, 0] - srt[:, 1]).numpy()
    pipe[k + "kp_nb"] = nb
    preds_hm, preds_img, preds_scores = getPrediction(hm, pt1, pt2, opt.inputResH, opt.inputResW,
                                                      opt.outputResH, opt.outputResW)
    pipe[k + "preds_hm"] = preds_hm.numpy()
    pipe[k + "preds_img"] = preds_img.numpy()
    pipe[k + "preds_scores"] = preds_scores.numpy()
    res = ref_nms.pose_nms(boxes.clone(), scores.clone(), preds_img.clone(), preds_scores.clone())
    pipe[k + "nms_n"] = np.array(len(res))
    if res:
        pipe[k + "nms_kp"] = res[0]["keypoints"].numpy()
        pipe[k + "nms_score"] = res[0]["kp_score"].numpy()
        pipe[k + "nms_prop"] = np.array(float(res[0]["proposal_score"]))
        pipe[k + "nms_bbox"] = res[0]["bbox"].numpy()
    Rfake = np.eye(3) + 0.01 * i
    tfake = np.array([[0.01 * i], [0.02], [0.9]])
    results_for_json.append({"imgname": im_name.split("/")[-1], "result": res, "cam_R": Rfake, "cam_t": tfake})
    print("frame", i, "argmax", int(pipe[k + "obj_argmax"]), "box", boxes.numpy().round(1),
          "kp_max[:3]", mx[:3].numpy().round(3), "min margin %.2e" % float(pipe[k + "kp_margin"].min()))

outdir = os.path.join(TMP, "json")
os.makedirs(outdir)
opt.format = None
ref_nms.write_json(results_for_json, outdir)
pipe["json_text"] = np.array(open(os.path.join(outdir, "Betapose-results.json")).read())
np.savez_compressed(os.path.join(GOLD, "pipeline.npz"), **pipe)

# ---------------------------------------------------------------------------
# post: getPrediction / pose_nms with several poses / metrics on random inputs
# ---------------------------------------------------------------------------
post = {}
g = np.random.Generator(np.random.PCG64(321))
hms = torch.from_numpy(g.normal(0, 0.3, (3, 50, 80, 64)).astype(np.float32))
hms[:, 10:] = 0.0   # keep the fixture small: 10 random maps per sample, rest flat zero
# force a few special cases: border maxima, all-negative map
hms[0, 0, 0, 5] = 9.0
hms[0, 1, 79, 63] = 9.0
hms[0, 2, 40, 0] = 9.0
hms[1, 3] = -hms[1, 3].abs() - 0.1
p1 = torch.tensor([[100.0, 80.0], [10.5, 20.25], [300.0, 200.0]])
p2 = torch.tensor([[260.0, 300.0], [90.75, 260.5], [420.0, 330.0]])
post["gp_hms"] = hms.numpy().astype(np.float16)   # stored compactly: the fp16-rounded maps ARE the input
hms16 = torch.from_numpy(post["gp_hms"].astype(np.float32))
a, b, c = getPrediction(hms16, p1.clone(), p2.clone(), 320, 256, 80, 64)
post["gp_pt1"], post["gp_pt2"] = p1.numpy(), p2.numpy()
post["gp_preds_hm"], post["gp_preds_img"], post["gp_maxval"] = a.numpy(), b.numpy(), c.numpy()

# pose_nms with 4 candidates (two near-duplicates, one weak, one distinct)
base = torch.from_numpy(g.uniform(100, 300, (50, 2)).astype(np.float32))
poses = torch.stack([base, base + 0.5, base + 80.0, base + torch.from_numpy(g.normal(0, 3, (50, 2)).astype(np.float32))])
pscores = torch.from_numpy(g.uniform(0.2, 0.9, (4, 50, 1)).astype(np.float32))
pscores[2] *= 0.2
bxs = torch.tensor([[90.0, 90, 310, 310], [91, 91, 311, 311], [170, 170, 390, 390], [88, 92, 312, 308]])
bsc = torch.tensor([[0.9], [0.8], [0.7], [0.6]])
post["nms_in_boxes"], post["nms_in_scores"] = bxs.numpy(), bsc.numpy()
post["nms_in_poses"], post["nms_in_pscores"] = poses.numpy(), pscores.numpy()
res = ref_nms.pose_nms(bxs.clone(), bsc.clone(), poses.clone(), pscores.clone())
post["nms_out_n"] = np.array(len(res))
for j, r in enumerate(res):
    post["nms_out%d_kp" % j] = r["keypoints"].numpy()
    post["nms_out%d_score" % j] = r["kp_score"].numpy()
    post["nms_out%d_prop" % j] = np.array(float(r["proposal_score"]))
    post["nms_out%d_bbox" % j] = r["bbox"].numpy()

# metrics
model_pts = g.uniform(-0.05, 0.05, (200, 3))
def _pose(rv, t):
    th = np.linalg.norm(rv); kx = rv / th
    K = np.array([[0, -kx[2], kx[1]], [kx[2], 0, -kx[0]], [-kx[1], kx[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K
    P = np.eye(4); P[:3, :3] = R; P[:3, 3] = t
    return P
gt = _pose(np.array([0.3, -0.2, 0.5]), np.array([0.02, -0.01, 0.8]))
est = _pose(np.array([0.31, -0.19, 0.48]), np.array([0.021, -0.012, 0.81]))
post["m_model"], post["m_gt"], post["m_est"] = model_pts, gt, est
post["m_add"] = np.array(ref_metrics.add_err(gt, est, model_pts))
post["m_proj"] = np.array(ref_metrics.projection_error_2d(gt, est, model_pts, synth.CAM_K))
post["m_iou"] = np.array([ref_metrics.iou([10, 10, 110, 210], [30, 40, 100, 260]),
                          ref_metrics.iou([10, 10, 50, 50], [60, 60, 80, 80])])
np.savez_compressed(os.path.join(GOLD, "post.npz"), **post)

json.dump(manifest, open(os.path.join(GOLD, "MANIFEST.json"), "w"), indent=1)
for f in sorted(os.listdir(GOLD)):
    print(f, os.path.getsize(os.path.join(GOLD, f)))
