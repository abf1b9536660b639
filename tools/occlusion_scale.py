#!/usr/bin/env python3
"""BASELINE configs[4] at its own shape on ONE GPU: the eight Occlusion-LineMod objects resident, every frame decoded once
and handed to all eight object pipelines ((frame, object) units), seeded synthetic weights and frames, dummy ground truth
(the printed accuracies mean nothing here; tests/test_gpu_multirank.py holds the closed-loop parity test).  Prints the
harness' own units/sec line and, for comparison, the reference's protocol: the same eight objects as eight separate
single-object runs over the same frames.

    python tools/occlusion_scale.py [--frames 96] [--streams 4] [--skip-single]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from betapose_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=96)
ap.add_argument("--streams", type=int, default=4)
ap.add_argument("--skip-single", action="store_true")
a = ap.parse_args()
OBJS = [1, 5, 6, 8, 9, 10, 11, 12]          # the objects of Occlusion-LineMod (sequence 02)
from PIL import Image  # noqa: E402

with tempfile.TemporaryDirectory() as tmp:
    indir = os.path.join(tmp, "sixd", "test", "02", "rgb")
    os.makedirs(indir)
    for i, fr in enumerate(synth.synth_frames(a.frames)):
        Image.fromarray(fr[:, :, ::-1].copy()).save(os.path.join(indir, "%04d.png" % i))
    rng = np.random.default_rng(0)
    kp_mm = {o: np.round(synth.synth_kp3d(50, seed=7 + o) * 1000.0, 6) for o in OBJS}
    gt = {i: [(o, np.eye(3), np.array([0.0, 0.0, 800.0]), [200, 150, 100, 100]) for o in OBJS] for i in range(a.frames)}
    synth.write_sixd_tree(os.path.join(tmp, "sixd"), 2, gt, {o: rng.normal(size=(300, 3)) * 30.0 for o in OBJS}, kp_mm,
                          {o: 100.0 for o in OBJS})
    common = [sys.executable, os.path.join(ROOT, "occlusion_evaluate.py"), "--indir", indir, "--sixd_base",
              os.path.join(tmp, "sixd"), "--synth_weights", "--fused", "--left_keypoints", "10", "--streams", str(a.streams)]
    t0 = time.time()
    r = subprocess.run(common + ["--obj_ids", ",".join(map(str, OBJS)), "--outdir", os.path.join(tmp, "multi")],
                       capture_output=True, text=True, cwd=ROOT)
    t_multi = time.time() - t0
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if "units/sec" in l]
    print("multi-object run (%d frames x %d objects = %d units, weights resident, frame decoded once): %s | process wall %.1f s"
          % (a.frames, len(OBJS), a.frames * len(OBJS), line[0].strip() if line else "?", t_multi))
    if not a.skip_single:
        t0 = time.time()
        rates = []
        for o in OBJS:
            r = subprocess.run(common + ["--obj_id", str(o), "--outdir", os.path.join(tmp, "s%d" % o)], capture_output=True,
                               text=True, cwd=ROOT)
            assert r.returncode == 0, r.stdout + r.stderr
            m = re.search(r"([\d.]+) (?:frames|units)/sec", r.stdout)
            rates.append(float(m.group(1)) if m else float("nan"))
        t_single = time.time() - t0
        print("reference protocol (8 single-object processes over the same frames): per-process frames/sec %s | total wall %.1f s"
              % (" ".join("%.0f" % v for v in rates), t_single))
