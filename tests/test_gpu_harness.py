"""The drop-in stage classes and the evaluation harness end to end (synthetic frames/weights): the staged pipeline
(reference structure) and the fused hipGraph pipeline must produce the same records as the reference's golden run."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import helpers  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", [[], ["--fused"], ["--detbatch", "3"]])   # staged, fused, staged with a ragged last batch
def test_evaluate_synthetic_matches_reference_json(tmp_path, mode):
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "evaluate.py"), "--synthetic", "4", "--outdir", str(out),
                        "--sp", "--profile"] + mode, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    got = json.loads(open(out / "Betapose-results.json").read())
    ref = json.loads(str(helpers.golden("pipeline.npz")["json_text"]))
    assert [g["image_id"] for g in got] == [r_["image_id"] for r_ in ref]
    for g, r_ in zip(got, ref):
        np.testing.assert_allclose(g["keypoints"], r_["keypoints"], rtol=1e-4, atol=5e-3)   # (x, y, score) x 50
        assert abs(g["score"] - r_["score"]) < 1e-3
        assert len(g["cam_R"]) == 9 and len(g["cam_t"]) == 3


def test_dynamic_write_results_convention(cuda):
    import torch
    from betapose_amd.yolo_util import dynamic_write_results
    pred = torch.zeros(2, 100, 6)
    pred[0, 17] = torch.tensor([50.0, 60.0, 20.0, 10.0, 0.9, 0.8])
    pred[0, 5] = torch.tensor([10.0, 10.0, 4.0, 4.0, 0.9, 0.7])      # tie on objectness: first index wins
    dets = dynamic_write_results(pred, 0.5, 80)
    assert dets.shape == (1, 8) and dets[0, 0] == 0
    np.testing.assert_allclose(dets[0, 1:].numpy(), [8, 8, 12, 12, 0.9, 0.7, 0], rtol=1e-6)
    assert dynamic_write_results(pred, 0.95, 80) == 0


@pytest.mark.parametrize("occlusion,staged", [(False, False), (True, False), (False, True)])
def test_evaluate_dataset_layout_closed_loop(tmp_path, cuda, occlusion, staged):
    """The non-synthetic route of the harness (--indir frames + --sixd_base ground truth, LineMod and Occlusion
    protocol): the ground-truth tree is written from the pipeline's own poses and boxes, so the three printed
    numbers must all be 1.000 -- this exercises frame files -> engines -> JSON -> gt.yml / models / kpmodels
    readers -> metric loop end to end."""
    import re
    from PIL import Image
    from betapose_amd import synth
    from betapose_amd.darknet import Darknet
    from betapose_amd.kpd import FastPoseHIP
    from betapose_amd.pipeline import FramePipeline, finish_record
    from betapose_amd.weights import fastpose_stream_from_state_dict

    obj_id, left = 1, (10 if occlusion else 50)
    frames = helpers.frames(3)
    indir = tmp_path / "rgb"
    indir.mkdir()
    for i, fr in enumerate(frames):
        Image.fromarray(fr[:, :, ::-1].copy()).save(indir / ("%04d.png" % i))
    kp_mm = np.round(synth.synth_kp3d(50) * 1000.0, 6)          # what the .ply holds after the %.6f write
    kp3d = kp_mm / 1000.0

    det = Darknet("yolo/cfg/yolov3-single.cfg", reso=416).load_stream(helpers.yolo_stream()).cuda()
    pose = FastPoseHIP.from_stream(fastpose_stream_from_state_dict(helpers.kpd_state_dict(), 50), n_classes=50).cuda()
    pipe = FramePipeline(det, pose, 480, 640, batch=1, confidence=0.01)
    gt = {}
    for i, fr in enumerate(frames):
        out = finish_record(pipe.run(fr)[0], "%04d.png" % i, kp3d, synth.CAM_K, left)
        assert out["boxes"] is not None and len(out["result"]) == 1
        x1, y1, x2, y2 = [float(v) for v in out["result"][0]["bbox"]]
        mine = (obj_id, out["cam_R"], np.asarray(out["cam_t"]).reshape(3) * 1000.0, [x1, y1, x2 - x1, y2 - y1])
        other = (7, np.eye(3), np.array([0.0, 0.0, 800.0]), [5, 5, 20, 20])
        gt[i] = [other, mine] if occlusion else [mine]
    del pipe, det, pose
    rng = np.random.default_rng(0)
    model_mm = rng.normal(size=(300, 3)) * 30.0
    synth.write_sixd_tree(str(tmp_path / "sixd"), 2 if occlusion else obj_id, gt, {obj_id: model_mm},
                          {obj_id: kp_mm}, {obj_id: 100.0})

    script = "occlusion_evaluate.py" if occlusion else "evaluate.py"
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(ROOT, script), "--indir", str(indir), "--outdir", str(out),
                        "--sixd_base", str(tmp_path / "sixd"), "--synth_weights"] + (["--sp"] if staged else ["--fused"]) +
                       ["--obj_id", str(obj_id), "--left_keypoints", "10"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    nums = dict(re.findall(r"(Mean add accuracy|2d reprojection accuracy|Mean IoU) for seq \d+ is: ([\d.nan]+)", r.stdout))
    assert nums == {"Mean add accuracy": "1.000", "2d reprojection accuracy": "1.000", "Mean IoU": "1.000"}, r.stdout
    assert len(json.loads(open(out / "Betapose-results.json").read())) == 3


def test_evaluate_f16_precision_close_to_reference_json(tmp_path):
    """--precision f16 end to end: same frames detected; fp16 operand rounding moves the box by a fraction of a pixel
    (re-sampling the crop) and heat-map values by ~1e-3, so a key point whose two best pixels are closer than that
    flips.  The seeded random-weight KPD produces nearly flat heat-maps (best-vs-second margins down to 3.6e-4 at a
    scale of 2.3, tests/golden), the worst case for this: measured 20 of 200 key points move; with the crop held
    fixed none does (tests/test_gpu_nets.py).  Stated bound here: >= 85 % of the key points within 1.5 px, scores
    within 5e-2 for those (the sub-pixel box shift re-samples the crop, which moves heat-map values more than the
    rounding itself)."""
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "evaluate.py"), "--synthetic", "4", "--outdir", str(out),
                        "--fused", "--precision", "f16"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    got = json.loads(open(out / "Betapose-results.json").read())
    ref = json.loads(str(helpers.golden("pipeline.npz")["json_text"]))
    assert [g["image_id"] for g in got] == [r_["image_id"] for r_ in ref]
    far = total = 0
    for g, r_ in zip(got, ref):
        kg, kr = np.array(g["keypoints"]).reshape(50, 3), np.array(r_["keypoints"]).reshape(50, 3)
        near = np.abs(kg[:, :2] - kr[:, :2]).max(axis=1) < 1.5
        far += int((~near).sum())
        total += 50
        assert np.abs(kg[near, 2] - kr[near, 2]).max() < 5e-2
    assert far <= total * 15 // 100, (far, total)


@pytest.mark.parametrize("batch", [1, 2])
def test_streamed_runner_order_and_failure_cleanup(tmp_path, cuda, batch):
    """Frames-in-flight driver over the native loader: records come back in list order and equal the one-at-a-time
    pipeline's; a broken frame in the middle surfaces as an exception after the frames in flight have drained, and the
    loader can still be closed (no slot left checked out)."""
    from PIL import Image
    from betapose_amd import _lib, synth
    from betapose_amd.darknet import Darknet
    from betapose_amd.frame_loader import FrameLoader
    from betapose_amd.kpd import FastPoseHIP
    from betapose_amd.pipeline import FramePipeline, StreamedRunner
    from betapose_amd.weights import fastpose_stream_from_state_dict
    frames = synth.synth_frames(7, 321)
    paths = []
    for i, fr in enumerate(frames):
        p = tmp_path / ("%04d.png" % i)
        Image.fromarray(fr[:, :, ::-1].copy()).save(p, compress_level=1)
        paths.append(str(p))
    det = Darknet("yolo/cfg/yolov3-single.cfg", reso=416, max_batch=batch).load_stream(helpers.yolo_stream()).cuda()
    pose = FastPoseHIP.from_stream(fastpose_stream_from_state_dict(helpers.kpd_state_dict(), 50), n_classes=50,
                                   max_batch=batch).cuda()
    runner = StreamedRunner(det, pose, 480, 640, streams=3, batch=batch)   # batch 2: 7 frames = 3 launches + a ragged one
    got = {}
    ld = FrameLoader(paths, threads=2, depth=8)
    assert runner.run(ld, lambda i, rec: got.__setitem__(i, rec)) == 7
    ld.close()
    assert list(got) == list(range(7))
    single = FramePipeline(det.clone(), pose.clone(), 480, 640)
    for i in (0, 3, 6):
        if batch == 1:
            np.testing.assert_array_equal(single.run(frames[i])[0], got[i])  # same kernels, same plan: bit-identical
        else:                                                                # two frames per launch: another summation order
            ref = single.run(frames[i])[0]
            bits = lambda r: np.ascontiguousarray(np.concatenate([r[:1], r[16::6][:50]])).view(np.int32)
            np.testing.assert_array_equal(bits(ref), bits(got[i]))           # same detector row, same 50 arg-max pixels
    # broken frame in the middle
    open(tmp_path / "bad.png", "wb").write(open(paths[2], "rb").read()[:4000])
    ld = FrameLoader(paths[:3] + [str(tmp_path / "bad.png")] + paths[3:], threads=2, depth=8)
    seen = []
    with pytest.raises(_lib.BetaposeHipError, match="bad.png"):
        runner.run(ld, lambda i, rec: seen.append(i))
    assert seen == sorted(seen) and set(seen) <= {0, 1, 2}
    ld.close()                                                                 # must not hang


def test_video_detection_loader_flow(tmp_path, cuda):
    """f4: frame sequence -> VideoDetectionLoader (letterbox, detector, box un-letterboxing) -> one box per frame inside
    the frame, equal to the oracle's box for the same letterboxed input."""
    import torch
    from PIL import Image
    from betapose_amd import video
    from betapose_amd.darknet import Darknet
    from betapose_amd.opt import opt
    fr = helpers.frames(3)
    d = tmp_path / "seq"
    d.mkdir()
    for i, f in enumerate(fr):
        Image.fromarray(f[:, :, ::-1].copy()).save(d / ("%04d.png" % i))
    det = Darknet("yolo/cfg/yolov3-single.cfg", reso=416, max_batch=2).load_stream(helpers.yolo_stream()).cuda()
    old = (opt.inp_dim, opt.confidence)
    opt.inp_dim, opt.confidence = "416", 0.01
    try:
        vd = video.VideoDetectionLoader(str(d), batchSize=2, det_model=det).start()
        assert vd.length() == 3
        got = []
        for i in range(3):
            inp, orig, boxes, scores = vd.read()
            assert tuple(inp.shape) == (3, 480, 640) and np.array_equal(orig, fr[i])
            assert boxes.shape == (1, 4) and scores.shape == (1, 1)
            b = boxes[0].numpy()
            assert 0 <= b[0] < b[2] <= 640 and 0 <= b[1] < b[3] <= 480
            got.append((boxes.clone(), scores.clone()))
    finally:
        opt.inp_dim, opt.confidence = old
    # ... and the boxes ARE the oracle's: the same letterboxed tensor (prep_frame, pinned to the reference's in
    # tests/test_video.py) through the oracle detector, dynamic_write_results and the oracle restatement of the
    # reference's un-letterboxing (dataloader.py:548-560)
    from betapose_amd import cfg as C, weights as W
    from oracle import post_ref, yolo_ref
    blocks = C.parse_cfg_text(C.yolov3_single_cfg_text())
    convs = W.split_darknet_stream(blocks, helpers.yolo_stream())
    for i, f in enumerate(fr):
        t, _, dim = video.prep_frame(f, 416)
        pred = yolo_ref.darknet_forward(blocks, convs, t)
        dets = yolo_ref.write_results(pred, 0.01, 80)
        ref = post_ref.unletterbox_boxes_ref(dets, torch.tensor([dim], dtype=torch.float32).repeat(1, 2), 416)
        assert got[i][0].shape == ref[:, 1:5].shape
        assert float((got[i][0] - ref[:, 1:5]).abs().max()) < 5e-3          # box corners, frame pixels (fp32 detector noise)
        assert float((got[i][1] - ref[:, 5:6]).abs().max()) < 2e-5


def test_fused_frames_per_launch_matches_one_frame_per_launch(tmp_path):
    """``--fused --detbatch 3`` (three frames per launch and stream, ragged last launch) against ``--detbatch 1``: same
    frames in the JSON, same arg-max pixels (key points within the float tolerance of a different summation order)."""
    import json
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    outs = {}
    for b in (1, 3):
        od = tmp_path / ("b%d" % b)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "evaluate.py"), "--synthetic", "7", "--outdir", str(od), "--fused",
                            "--detbatch", str(b), "--streams", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert r.returncode == 0, r.stdout + r.stderr
        outs[b] = json.load(open(od / "Betapose-results.json"))
    assert [e["image_id"] for e in outs[1]] == [e["image_id"] for e in outs[3]] and len(outs[1]) == 7
    for e1, e3 in zip(outs[1], outs[3]):
        k1, k3 = np.asarray(e1["keypoints"]).reshape(-1, 3), np.asarray(e3["keypoints"]).reshape(-1, 3)
        assert np.abs(k1[:, :2] - k3[:, :2]).max() <= 5e-3 and np.abs(k1[:, 2] - k3[:, 2]).max() <= 2e-4
