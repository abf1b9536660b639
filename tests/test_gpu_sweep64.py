"""BASELINE configs[0] at its own size: the 64 evaluation frames of tests/golden/sweep64.npz -- compact records written by the REFERENCE's
own stage classes (tools/make_golden_sweep64.py: ImageLoader -> DetectionLoader -> DetectionProcessor -> FastPose -> getPrediction ->
pose_nms -> write_json) -- through the fused HIP pipeline.

Integer-exact bar (north star): YOLO box index and the 50 KPD arg-max pixels identical to the reference's on every frame, in the
fp32-accurate arithmetics (bf16x3 = the headline, f32).  "Identical" is asserted wherever the reference's own best-vs-second margin
exceeds the float tolerance of that stage (2 x 2e-4 for heat-maps: 2 984 of the 3 200 key points once the three frames with a coin-toss crop
corner are set aside; 2e-5 for objectness: all 64 frames);
below it the pixel the pipeline picked must be as high as the reference's maximum within that tolerance.  Float bars as everywhere else:
boxes and crop window 5e-3 px, maxima 2e-4, JSON key points 5e-3 px.  The fp16 modes are stated-tolerance modes: their flips against the
reference's fp32 run are COUNTED (printed with -s), one frame per launch and at configs[2]'s 28 frames per launch (28 REFERENCE frames, not
random crops).  With these seeded random weights 34 % of the reference's own best-vs-second margins are below the fp16 modes' heat-map
tolerance of 1e-2 (19 % below 5e-3), and the fp16 detector's box (<= 0.25 px) re-samples the crop: measured 7-8 % of the 3 200 key points move (259 / 265 one frame per launch, 192 / 199 of 2 800 at 28 per launch),
and where the box moves a crop corner across an integer the whole crop shifts by a pixel; the box index moves on one of the 64 frames.
Asserted: <= 10 % of the key points, <= 2 box indices.  (The <= 2 % of tests/test_gpu_nets.py is the key-point detector ALONE on the reference's crops.)"""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import helpers  # noqa: E402
from betapose_amd import synth  # noqa: E402
from betapose_amd.darknet import Darknet  # noqa: E402
from betapose_amd.kpd import FastPoseHIP  # noqa: E402
from betapose_amd.pipeline import FramePipeline, finish_record  # noqa: E402

HM_TOL = 2e-4
PROB_TOL = 2e-5


@pytest.fixture(scope="module")
def gold():
    g = helpers.golden("sweep64.npz")
    assert int(g["n_frames"]) == 64 and int(g["frame_seed"]) == helpers.FRAME_SEED
    return g


@pytest.fixture(scope="module")
def frames64():
    return synth.synth_frames(64, helpers.FRAME_SEED)


def _engines(max_batch):
    det = Darknet("yolo/cfg/yolov3-single.cfg", reso=416, max_batch=max_batch).load_stream(helpers.yolo_stream()).cuda()
    pose = FastPoseHIP(helpers.kpd_state_dict(), n_classes=50, max_batch=max_batch).cuda()
    return det, pose


def _compare(rec, g, i, exact):
    """One frame's 316-float record against reference frame i.  Returns (key points checked exactly, key points below the margin bar,
    key-point flips, box-index flip)."""
    idx = int(rec[:1].view(np.int32)[0])
    obj_margin = float(g["obj_top2"][i, 0] - g["obj_top2"][i, 1])
    idx_flip = idx != int(g["obj_argmax"][i])
    got = rec[16:].reshape(50, 6)
    kp = got[:, 0].copy().view(np.int32)
    ref = g["kp_idx"][i].astype(np.int32)
    sure = g["kp_margin"][i] > 2 * HM_TOL
    # cropBox truncates the window's corners to integers (KPD/src/utils/img.py:242-262): a corner the reference has within the box tolerance of
    # an integer (not ON one: clamped corners are exact) is a coin toss between two crops one pixel apart -- frame 47's x2 = 388.00043.  Such a
    # frame is held to the box / window bars only (the 28-frames-per-launch plan, whose accumulation order differs, lands on the other side)
    pts = np.concatenate([g["pt1"][i], g["pt2"][i]])
    frac = np.abs(pts - np.round(pts))
    if exact and bool(((frac > 0) & (frac < 5e-3)).any()):
        np.testing.assert_allclose(rec[12:16], g["boxes"][i], rtol=0, atol=5e-3, err_msg="frame %d: box" % i)
        np.testing.assert_allclose(rec[8:12], pts, rtol=0, atol=5e-3, err_msg="frame %d: crop window" % i)
        assert not idx_flip
        return 0, 50, int((kp != ref).sum()), 0
    if exact:
        if obj_margin > PROB_TOL:
            assert not idx_flip, "frame %d: YOLO box index %d != reference %d" % (i, idx, int(g["obj_argmax"][i]))
        np.testing.assert_allclose(rec[12:16], g["boxes"][i], rtol=0, atol=5e-3, err_msg="frame %d: box" % i)
        assert abs(float(rec[5]) - float(g["scores"][i])) <= PROB_TOL
        np.testing.assert_allclose(rec[8:12], np.concatenate([g["pt1"][i], g["pt2"][i]]), rtol=0, atol=5e-3, err_msg="frame %d: crop window" % i)
        assert np.array_equal(kp[sure], ref[sure]), "frame %d: KPD arg-max pixels %s" % (i, np.nonzero(kp != ref)[0])
        # a key point the reference itself separates by less than the tolerance may legitimately land on the runner-up, but must be as high
        assert np.all(g["kp_max"][i] - got[:, 1] <= 2 * HM_TOL) and np.all(np.abs(got[sure, 1] - g["kp_max"][i][sure]) <= HM_TOL)
    return int(sure.sum()), int((~sure).sum()), int((kp != ref).sum()), int(idx_flip)


@pytest.mark.parametrize("mode", ["bf16x3", "f32"])
def test_64_reference_frames_integer_exact(cuda, gold, frames64, mode):
    det, pose = _engines(1)
    det.set_precision(mode)
    pose.set_precision(mode)
    pipe = FramePipeline(det, pose, 480, 640, batch=1)
    kp3d, cam_K = synth.synth_kp3d(50), synth.CAM_K
    ref_json = json.loads(bytes(gold["json_utf8"]).decode("utf-8"))
    assert len(ref_json) == 64
    checked = skipped = flips = 0
    for i, frame in enumerate(frames64):
        rec = pipe.run(frame)[0]
        c, s, f, _ = _compare(rec, gold, i, exact=True)
        checked, skipped, flips = checked + c, skipped + s, flips + f
        out = finish_record(rec, "%04d.png" % i, kp3d, cam_K)
        assert len(out["result"]) == int(gold["nms_n"][i]) == 1
        # the JSON line of the frame: (x, y, score) x 50 as the reference's write_json printed them
        r_ = ref_json[i]
        assert r_["image_id"] == "%04d.png" % i
        kps = np.asarray(r_["keypoints"], np.float64).reshape(50, 3)
        low = gold["kp_margin"][i] <= 2 * HM_TOL           # (a runner-up pixel moves the key point by a pixel or more: compared above, not here)
        np.testing.assert_allclose(np.asarray(out["result"][0]["keypoints"])[~low], kps[~low, :2], rtol=1e-4, atol=5e-3)
        np.testing.assert_allclose(np.asarray(out["result"][0]["kp_score"])[:, 0], kps[:, 2], rtol=0, atol=2 * HM_TOL)
        assert abs(float(out["result"][0]["proposal_score"]) - float(r_["score"])) < 1e-3
    assert checked >= 2950 and skipped <= 250, (checked, skipped)      # (2 984 / 216: 71 of the 3 200 reference margins are below 2 x 2e-4; three frames have a crop corner within 5e-3 of an integer)
    assert flips <= skipped          # every difference sits on a margin below the tolerance


@pytest.mark.parametrize("mode", ["f16", "f16r"])
def test_64_reference_frames_fp16_flips_are_counted(cuda, gold, frames64, mode):
    """Stated-tolerance modes (BASELINE configs[2]): against the REFERENCE's fp32 run <= 10 % of the 3 200 key points (measured 6-7 %) and <= 2 of
    the 64 box indices may move (module docstring: low margins of the random-weight maps + a crop window whose integer corners move with
    the fp16 detector's box)."""
    det, pose = _engines(1)
    det.set_precision(mode)
    pose.set_precision(mode)
    pipe = FramePipeline(det, pose, 480, 640, batch=1)
    flips = idx_flips = 0
    for i, frame in enumerate(frames64):
        rec = pipe.run(frame)[0]
        _, _, f, jf = _compare(rec, gold, i, exact=False)
        flips, idx_flips = flips + f, idx_flips + jf
    print("fp16 flips, 64 reference frames, one per launch, %s: %d of 3200 key points (%.1f %%), %d of 64 box indices" % (mode, flips, flips / 32.0, idx_flips))
    assert idx_flips <= 2, idx_flips
    assert flips <= 0.10 * 3200, flips


@pytest.mark.parametrize("mode,exact", [("bf16x3", True), ("f16", False), ("f16r", False)])
def test_batch_28_of_reference_frames(cuda, gold, frames64, mode, exact):
    """configs[2]'s shape on 28 REFERENCE frames per launch (frames 0-27 and 28-55): the whole fused pipeline at 28 frames per launch --
    plan tables of batch 28, halo / streaming / fused kernels, tiles spanning images -- against the reference's records.  bf16x3: the
    integer-exact bar; fp16 modes: flips counted, <= 10 % (measured 6.9-7.1 %)."""
    det, pose = _engines(28)
    det.set_precision(mode)
    pose.set_precision(mode)
    pipe = FramePipeline(det, pose, 480, 640, batch=28)
    flips = idx_flips = skipped = 0
    for lo in (0, 28):
        recs = pipe.run(np.stack(frames64[lo:lo + 28]))
        assert np.array_equal(pipe.run(np.stack(frames64[lo:lo + 28])), recs)      # bit-reproducible
        for b in range(28):
            _, s, f, jf = _compare(recs[b], gold, lo + b, exact=exact)
            flips, idx_flips, skipped = flips + f, idx_flips + jf, skipped + s
    print("28 reference frames per launch, %s: %d of 2800 key points differ (%.1f %%), %d of 56 box indices" % (mode, flips, flips / 28.0, idx_flips))
    if exact:
        assert flips <= skipped and idx_flips == 0
    else:
        assert idx_flips <= 2 and flips <= 0.10 * 56 * 50, (idx_flips, flips)
