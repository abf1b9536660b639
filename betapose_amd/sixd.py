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


def _load_yaml(path):
    import yaml
    with open(path, "r") as f:
        return yaml.safe_load(f)


def load_sixd(base_path, seq, nr_frames=0, load_mesh=True):
    bench = Benchmark()
    bench.scale_to_meters = 0.001
    cam_path = os.path.join(base_path, "camera.yml")
    if os.path.exists(cam_path):
        c = _load_yaml(cam_path)
        bench.cam[0, 0], bench.cam[0, 2], bench.cam[1, 1], bench.cam[1, 2] = c["fx"], c["cx"], c["fy"], c["cy"]
    info = _load_yaml(os.path.join(base_path, "models", "models_info.yml"))
    bench.diameter.append(10000.0)                     # index 0 is unused: object ids start at 1 (sixd.py:73)
    for _, val in info.items():
        bench.diameter.append(val["diameter"])
    if seq is None:
        return bench
    path = os.path.join(base_path, "test/{:02d}/".format(seq))
    frame_info = _load_yaml(os.path.join(path, "info.yml"))
    gts = _load_yaml(os.path.join(path, "gt.yml"))
    nr_frames = nr_frames if nr_frames > 0 else len(frame_info)
    for i in range(nr_frames):
        fr = Frame()
        fr.nr = i
        fr.path = path + "rgb/" + "{:04d}".format(i) + ".png"
        for gt in gts[i]:
            pose = np.identity(4)
            pose[:3, :3] = np.array(gt["cam_R_m2c"], dtype=np.float64).reshape(3, 3)
            pose[:3, 3] = np.squeeze(np.array(gt["cam_t_m2c"], dtype=np.float64)) * bench.scale_to_meters
            fr.gt.append((gt["obj_id"], pose, gt["obj_bb"]))
        if "cam_K" in frame_info[i]:
            fr.cam = np.array(frame_info[i]["cam_K"], dtype=np.float64).reshape(3, 3)
        bench.frames.append(fr)
    return bench
