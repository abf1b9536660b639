#!/usr/bin/env python3
"""Golden vectors for the key-point model thinning step (utils/model.py:29-46 ``Model3D.refine``), produced by the
REFERENCE's own class imported in place (build container only; shims as in tools/make_golden.py).  Inputs are seeded
point sets, outputs what ``refine`` leaves.  Writes tests/golden/refine.npz."""
import os, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_shims  # noqa: E402

ref_shims.install()
os.chdir(ref_shims.REF)
sys.path.insert(0, ref_shims.REF)
from utils.model import Model3D  # noqa: E402

g = np.random.Generator(np.random.PCG64(2024))
out = {}
cases = {
    "a": (g.uniform(-0.06, 0.06, (64, 3)), 50),                      # metres, as the harness uses it (kp .ply / 1000)
    "b": (g.uniform(-0.06, 0.06, (23, 3)), 10),                      # SURVEY fixture (6): thin to 10
    "c": (np.round(g.uniform(-0.05, 0.05, (30, 3)), 2), 12),         # coarse grid: many exactly tied distances
    "d": (g.uniform(-400.0, 400.0, (12, 3)), 5),                     # millimetre-scale, spread: distances above the
}                                                                    # hard-wired 100.0 start value (model.py:36)
for k, (pts, keep) in cases.items():
    m = Model3D()
    m.vertices = pts.copy()
    m.refine(total_kp=keep)
    out[k + "_in"], out[k + "_keep"], out[k + "_out"] = pts, np.array(keep), np.asarray(m.vertices)
    print(k, pts.shape, "->", np.asarray(m.vertices).shape)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "refine.npz"), **out)
