"""SIXD benchmark reader with the reference's structure (utils/sixd.py:43-111): ``load_sixd(base_path, seq, nr_frames)``
returns a ``Benchmark`` whose ``frames[i].gt`` is a list of ``(obj_id, pose 4x4 in metres, [x, y, w, h])``, ``cam`` the
3x3 intrinsics from ``camera.yml`` (identity without it) and ``diameter`` a list indexed by object id (entry 0 is a
placeholder, as in the reference).  Images and meshes are not loaded (the reference has those lines commented out)."""
from __future__ import annotations

import os

import numpy as np


class Frame:
    def __init__(self):
        self.nr = None
        self.color = None
        self.depth = None
        self.cam = np.identity(3)
        self.gt = []
        self.path = None


class Benchmark:
    def __init__(self):
        self.cam = np.identity(3)
        self.models = {}
        self.kpmodels = {}
        self.frames = []
        self.diameter = []
        self.scale_to_meters = 0.001


def _yaml(*parts):
    import yaml
    with open(os.path.join(*parts), "r") as f:
        return yaml.safe_load(f)


def _intrinsics(base_path):
    """camera.yml (fx, fy, cx, cy) -> 3x3 K; identity when the tree has no camera file."""
    K = np.identity(3)
    if os.path.exists(os.path.join(base_path, "camera.yml")):
        c = _yaml(base_path, "camera.yml")
        K[[0, 1, 0, 1], [0, 1, 2, 2]] = c["fx"], c["fy"], c["cx"], c["cy"]
    return K


def _gt_tuple(entry, to_metres):
    """One gt.yml record -> (obj_id, 4x4 model-to-camera pose with the translation in metres, [x, y, w, h])."""
    T = np.identity(4)
    T[:3, :3] = np.asarray(entry["cam_R_m2c"], dtype=np.float64).reshape(3, 3)
    T[:3, 3] = np.asarray(entry["cam_t_m2c"], dtype=np.float64).reshape(3) * to_metres
    return entry["obj_id"], T, entry["obj_bb"]


def _frame(seq_dir, nr, gt_entries, info, to_metres):
    fr = Frame()
    fr.nr, fr.path = nr, "%srgb/%04d.png" % (seq_dir, nr)
    fr.gt = [_gt_tuple(e, to_metres) for e in gt_entries]
    if "cam_K" in info:
        fr.cam = np.asarray(info["cam_K"], dtype=np.float64).reshape(3, 3)
    return fr


def load_sixd(base_path, seq, nr_frames=0, load_mesh=True):
    """``seq`` None: intrinsics and diameters only.  ``nr_frames`` 0: every frame of the sequence."""
    bench = Benchmark()
    bench.cam = _intrinsics(base_path)
    # diameters indexed by object id; ids start at 1, so slot 0 holds the reference's placeholder (utils/sixd.py:73)
    bench.diameter = [10000.0] + [v["diameter"] for v in _yaml(base_path, "models", "models_info.yml").values()]
    if seq is not None:
        seq_dir = os.path.join(base_path, "test/%02d/" % seq)
        infos, gts = _yaml(seq_dir, "info.yml"), _yaml(seq_dir, "gt.yml")
        count = nr_frames if nr_frames > 0 else len(infos)
        bench.frames = [_frame(seq_dir, i, gts[i], infos[i], bench.scale_to_meters) for i in range(count)]
    return bench
