#!/usr/bin/env python3
"""Golden vectors for the video input stage from the REFERENCE's own ``letterbox_image`` / ``prep_frame``
(3_6Dpose_estimator/yolo/preprocess.py:18-60), imported in place (build container only; shims as tools/make_golden.py).

``cv2`` cannot be installed here, and the reference's letterbox calls ``cv2.resize(img, (w, h), interpolation=
cv2.INTER_CUBIC)``.  The stub handed to the reference for THAT ONE CALL is a stated stand-in -- the restatement of
OpenCV's 8-bit bicubic in betapose_amd/video.py (``cv_resize_cubic``: 4 taps, a = -0.75, 11-bit fixed-point weights,
replicated border) -- so these vectors pin everything around the resize as the reference computes it (the truncated
new_w / new_h, the grey canvas and where the picture sits on it, BGR -> RGB, / 255, tensor layout, the returned (w, h)),
and NOT the interpolation itself, which stays "parity unpinned" against a real cv2.

Writes tests/golden/video.npz."""
import os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_shims  # noqa: E402

ref_shims.install()
import torch  # noqa: E402
from betapose_amd import synth  # noqa: E402
from betapose_amd.video import cv_resize_cubic  # noqa: E402

cv2 = sys.modules["cv2"]
cv2.INTER_CUBIC = 2
cv2.resize = lambda img, size, interpolation=None: cv_resize_cubic(np.asarray(img, dtype=np.uint8), int(size[0]), int(size[1]))

os.chdir(ref_shims.REF)
from yolo import preprocess as ref_pre  # noqa: E402

INP = 96
base = synth.synth_frame(777)                                    # BGR u8 480 x 640
frames = [np.ascontiguousarray(base[::4, ::4]),                  # 120 x 160: wider than tall (borders top / bottom)
          np.ascontiguousarray(base[:125 * 3:3, :75 * 3:3]),     # 125 x 75: taller than wide (borders left / right)
          np.ascontiguousarray(base[100:177, 200:339]),          # 77 x 139: sizes that do not divide anything
          np.ascontiguousarray(base[:96, :96])]                  # already the network size
out = {"inp_dim": np.int32(INP)}
for i, f in enumerate(frames):
    t, orig, dim = ref_pre.prep_frame(f, INP)
    lb = ref_pre.letterbox_image(f, (INP, INP))
    assert orig is f and tuple(t.shape) == (1, 3, INP, INP)
    u8 = torch.round(t[0] * 255).to(torch.uint8)
    assert torch.equal(u8.float().div(255.0), t[0])              # the tensor is exactly (u8 canvas) / 255: store the u8
    out["frame%d" % i] = f
    out["tensor_u8_%d" % i] = u8.numpy()
    out["canvas%d" % i] = np.asarray(lb).astype(np.uint8)
    out["dim%d" % i] = np.asarray(dim, np.int32)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "video.npz"), **out)
print("wrote tests/golden/video.npz", {k: getattr(v, "shape", v) for k, v in out.items()})
