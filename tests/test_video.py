"""Video / webcam loaders and visualisation (dataloader.py:192-282,468-647, yolo/preprocess.py:18-60,
fn.py vis_frame) over the frame sources this image can decode.  CPU only (the detector-backed loader is in the GPU
suite)."""
import os

import numpy as np
import pytest

from betapose_amd import synth, video
from betapose_amd.opt import opt


def _frames(n=5):
    return synth.synth_frames(n, 77)


def test_mjpeg_roundtrip_and_frame_source(tmp_path):
    fr = _frames(4)
    w = video.MJPEGWriter(str(tmp_path / "v.avi"), fps=30, frame_size=(640, 480), quality=95)
    for f in fr:
        w.write(f)
    w.release()
    src = video.FrameSource(str(tmp_path / "v.avi"))
    assert src.isOpened() and src.frame_count == 4 and src.frame_size == (640, 480) and abs(src.fps - 30) < 1e-3
    for f in fr:
        ok, g = src.read()
        assert ok and g.shape == f.shape and g.dtype == np.uint8
        assert np.abs(g.astype(int) - f.astype(int)).mean() < 12          # JPEG loss on noise-like frames
    assert src.read() == (False, None)
    with pytest.raises(IOError):
        video.FrameSource(str(tmp_path / "missing.mp4"))
    (tmp_path / "x.mp4").write_bytes(b"\x00\x00\x00\x18ftypmp42")
    with pytest.raises(IOError, match="Motion-JPEG"):
        video.FrameSource(str(tmp_path / "x.mp4"))


def test_letterbox_and_prep_frame_geometry():
    f = _frames(1)[0]                                   # 480 x 640
    lb = video.letterbox_image(f, (416, 416))
    assert lb.shape == (416, 416, 3)
    new_h = int(480 * min(416 / 640, 416 / 480))        # 312
    top = (416 - new_h) // 2
    assert (lb[:top] == 128).all() and (lb[top + new_h:] == 128).all()          # grey bars (preprocess.py:27)
    assert not (lb[top:top + new_h] == 128).all()
    t, orig, dim = video.prep_frame(f, 416)
    assert tuple(t.shape) == (1, 3, 416, 416) and dim == (640, 480) and orig is f
    assert abs(float(t[0, :, 0, 0].mean()) - 128 / 255) < 1e-6                   # RGB 0..1 of the grey border
    # cubic resize restatement: identity size is exact, a constant image stays constant, down-scaling a ramp stays a ramp
    assert np.array_equal(video.cv_resize_cubic(f, 640, 480), f)
    assert (video.cv_resize_cubic(np.full((40, 60, 3), 77, np.uint8), 33, 21) == 77).all()
    ramp = np.tile(np.arange(0, 240, 2, dtype=np.uint8)[None, :, None], (8, 1, 3))
    r = video.cv_resize_cubic(ramp, 60, 8)[4, 2:-2, 0].astype(int)
    assert (np.diff(r) >= 3).all() and (np.diff(r) <= 5).all()


def test_video_loader_batches_in_order(tmp_path):
    from PIL import Image
    fr = _frames(5)
    d = tmp_path / "seq"
    d.mkdir()
    for i, f in enumerate(fr):
        Image.fromarray(f[:, :, ::-1].copy()).save(d / ("%04d.png" % i))
    old = opt.inp_dim
    opt.inp_dim = "416"
    try:
        vl = video.VideoLoader(str(d), batchSize=2).start()
        assert vl.length() == 5 and vl.videoinfo()[2] == (640, 480)
        seen = []
        for _ in range(3):
            img, orig, names, dims = vl.getitem()
            assert img.shape[1:] == (3, 416, 416) and len(orig) == img.shape[0] == dims.shape[0]
            assert dims[0].tolist() == [640.0, 480.0, 640.0, 480.0]
            seen += names
            for o, n in zip(orig, names):
                assert np.array_equal(o, fr[int(n.split(".")[0])])
        assert seen == ["0.jpg", "1.jpg", "2.jpg", "3.jpg", "4.jpg"]
        with pytest.raises(IOError, match="capture library"):
            video.WebcamLoader(0)
        wc = video.WebcamLoader(str(d), queueSize=8).start()
        img, orig, inp, dims = wc.read()
        assert tuple(img.shape) == (1, 3, 416, 416) and tuple(inp.shape) == (3, 480, 640) and dims.shape == (1, 4)
    finally:
        opt.inp_dim = old


def test_vis_frame_and_data_writer_video(tmp_path):
    f = _frames(1)[0]
    res = {"imgname": "0.png", "result": [{"bbox": np.array([100., 120., 300., 330.]),
                                           "keypoints": np.array([[150., 200.], [250., 300.]]),
                                           "kp_score": np.array([[0.9], [0.01]])}]}
    out = video.vis_frame(f, res)
    assert out.shape == f.shape and out.dtype == np.uint8
    assert not np.array_equal(out[198:203, 148:153], f[198:203, 148:153])       # drawn key point
    assert np.array_equal(out[298:303, 248:253], f[298:303, 248:253])           # score <= 0.05: skipped (fn.py:175)
    assert (out[120, 100:301, 1] == 255).all()                                  # box edge
    from betapose_amd.dataloader import DataWriter
    dw = DataWriter(synth.CAM_K, 50, synth.synth_kp3d(50), save_video=True, savepath=str(tmp_path / "o" / "1.avi"),
                    fps=25, frameSize=(640, 480)).start()
    dw.save(None, None, None, None, None, f, "0.png")
    while dw.running():
        pass
    dw.stop()
    assert video.FrameSource(str(tmp_path / "o" / "1.avi")).frame_count == 1


def test_data_writer_survives_a_failing_item():
    """An item that makes the writer thread raise (here: a box with no heat-maps) must still leave the pending count, so that
    ``while writer.running()`` ends, and the exception reaches the caller through stop() (round-4 advisor finding)."""
    import time
    import pytest
    from betapose_amd.dataloader import DataWriter
    dw = DataWriter(synth.CAM_K, 50, synth.synth_kp3d(50)).start()
    dw.save(np.zeros((1, 4), np.float32), np.ones((1, 1), np.float32), None, None, None, None, "bad.png")
    dw.save(None, None, None, None, None, None, "empty.png")          # a frame without detections behind it is still drained
    t0 = time.time()
    while dw.running():
        assert time.time() - t0 < 20, "running() never turned false after the writer raised"
    with pytest.raises(Exception):
        dw.stop()
    assert dw.results() == []


def test_prep_frame_equals_the_reference_on_its_golden_frames(golden_dir):
    """``letterbox_image`` / ``prep_frame`` against vectors produced by the reference's OWN functions (yolo/preprocess.py:
    18-60, tools/make_golden_video.py) with the stated bicubic stand-in for its one ``cv2.resize`` call: canvas size and
    colour, where the picture sits, the truncated new_w / new_h, BGR -> RGB, / 255, layout, the returned (w, h)."""
    import torch
    from betapose_amd import video
    g = np.load(os.path.join(golden_dir, "video.npz"))
    D = int(g["inp_dim"])
    for i in range(4):
        f = g["frame%d" % i]
        lb = video.letterbox_image(f, (D, D))
        assert lb.dtype == np.uint8 and np.array_equal(lb, g["canvas%d" % i])
        t, orig, dim = video.prep_frame(f, D)
        assert orig is f and tuple(dim) == tuple(int(v) for v in g["dim%d" % i])
        assert torch.equal(t[0], torch.from_numpy(g["tensor_u8_%d" % i]).float().div(255.0))


def test_unletterbox_boxes_equal_the_oracle():
    """The vectorised box transform against the oracle's statement-for-statement restatement of dataloader.py:548-560
    (bit-identical: the same float operations per element), frames of every aspect, boxes inside, across and beyond
    the picture."""
    import torch
    from betapose_amd import video
    from oracle import post_ref
    rng = np.random.default_rng(5)
    dims = torch.tensor([[640., 480.], [75., 125.], [139., 77.], [416., 416.]]).repeat(1, 2)
    for D in (416, 608, 96):
        n = 64
        dets = torch.zeros(n, 8)
        dets[:, 0] = torch.from_numpy(rng.integers(0, 4, n)).float()
        dets[:, 1:5] = torch.from_numpy(rng.uniform(-40, D + 40, (n, 4))).float()
        dets[:, 5:] = torch.from_numpy(rng.random((n, 3))).float()
        got = video.unletterbox_boxes(dets, dims, D)
        ref = post_ref.unletterbox_boxes_ref(dets, dims, D)
        assert torch.equal(got, ref)
        assert torch.equal(dets[:, 1:5], torch.from_numpy(np.asarray(dets[:, 1:5])))   # the input is left alone
