"""Evaluation metrics and ground-truth helpers of the harness (utils/metrics.py:10-22,77-127;
utils/model.py:29-46,79-85; utils/sixd.py:60-111), numpy f64."""
from __future__ import annotations

import os
from typing import Dict, List

import numpy as np


def add_err(gt_pose, est_pose, model):
    a = model @ gt_pose[:3, :3].T + gt_pose[:3, 3]
    b = model @ est_pose[:3, :3].T + est_pose[:3, 3]
    return float(np.mean(np.linalg.norm(a - b, axis=1)))


def projection_error_2d(gt_pose, est_pose, model, cam):
    m = np.concatenate((model, np.ones((model.shape[0], 1))), axis=1)
    g = cam @ gt_pose[:3] @ m.T
    e = cam @ est_pose[:3] @ m.T
    g, e = g / g[2], e / e[2]
    return float(np.mean(np.linalg.norm(g[:2].T - e[:2].T, axis=1)))


def rot_error(gt_pose, est_pose):
    """Angle of the relative rotation in degrees, 0..180 (utils/metrics.py:35-67 computes the same angle through
    quaternions)."""
    R = np.asarray(gt_pose)[:3, :3] @ np.asarray(est_pose)[:3, :3].T
    return float(np.degrees(np.arccos(np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0))))


def trans_error(gt_pose, est_pose):
    """(norm, per-axis absolute) translation error (utils/metrics.py:70-74)."""
    d = np.asarray(gt_pose)[:3, 3] - np.asarray(est_pose)[:3, 3]
    return float(np.linalg.norm(d)), np.abs(d)


def iou(gt_box, est_box):
    xA, yA = max(gt_box[0], est_box[0]), max(gt_box[1], est_box[1])
    xB, yB = min(gt_box[2], est_box[2]), min(gt_box[3], est_box[3])
    if xB <= xA or yB <= yA:
        return 0.0
    inter = (xB - xA) * (yB - yA)
    A = (gt_box[2] - gt_box[0]) * (gt_box[3] - gt_box[1])
    B = (est_box[2] - est_box[0]) * (est_box[3] - est_box[1])
    return inter / float(A + B - inter)


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}


def load_ply_vertices(path: str) -> np.ndarray:
    """Vertex x, y, z of a .ply file as float64 [n, 3] -- what ``Model3D.load`` takes from plyfile
    (utils/model.py:79-85).  ASCII and binary (little / big endian) files; the vertex element must come first, as in
    the SIXD models and the designator's key-point files."""
    with open(path, "rb") as f:
        head = b""
        while not head.rstrip().endswith(b"end_header"):
            line = f.readline()
            if not line:
                raise ValueError("%s: no end_header" % path)
            head += line
        lines = [ln.strip() for ln in head.decode("ascii", "replace").split("\n") if ln.strip()]
        if not lines or lines[0] != "ply":
            raise ValueError("%s: not a PLY file" % path)
        fmt, n, props, elem_order, cur = None, 0, [], [], None
        for ln in lines[1:]:
            tok = ln.split()
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                cur = tok[1]
                elem_order.append(cur)
                if cur == "vertex":
                    n = int(tok[2])
            elif tok[0] == "property" and cur == "vertex":
                if tok[1] == "list":
                    raise ValueError("%s: list property in the vertex element" % path)
                if tok[1] not in _PLY_TYPES:
                    raise ValueError("%s: unknown property type %s" % (path, tok[1]))
                props.append((tok[2], _PLY_TYPES[tok[1]]))
        if fmt is None or not elem_order or elem_order[0] != "vertex":
            raise ValueError("%s: the vertex element must be the first element" % path)
        names = [p_[0] for p_ in props]
        if not all(a in names for a in ("x", "y", "z")):
            raise ValueError("%s: vertex element without x, y, z" % path)
        if fmt == "ascii":
            ix = [names.index(a) for a in ("x", "y", "z")]
            rows = []
            for _ in range(n):
                tok = f.readline().split()
                if len(tok) < len(names):
                    raise ValueError("%s: truncated vertex list" % path)
                rows.append([float(tok[i]) for i in ix])
            return np.array(rows, dtype=np.float64).reshape(n, 3)
        if fmt not in ("binary_little_endian", "binary_big_endian"):
            raise ValueError("%s: unknown PLY format %s" % (path, fmt))
        end = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(nm, end + t) for nm, t in props])
        buf = f.read(n * dt.itemsize)
        if len(buf) < n * dt.itemsize:
            raise ValueError("%s: truncated vertex data" % path)
        v = np.frombuffer(buf, dtype=dt, count=n)
        return np.stack([v["x"], v["y"], v["z"]], axis=1).astype(np.float64)


def refine_keypoints(vertices: np.ndarray, keep: int) -> np.ndarray:
    """``Model3D.refine`` (utils/model.py:29-46): ``len - keep`` times, delete the first point (in row-major pair
    order) of the closest pair.  Two quirks of the reference are kept: the running minimum of a round starts at the
    hard-wired 100.0, and the index to delete is carried over from the previous round (initially 0) -- so when every
    pair is at least 100 apart (millimetre-scale, spread-out points) the previous round's index is deleted again."""
    v = np.array(vertices, dtype=np.float64)
    min_index = 0
    for _ in range(max(0, len(v) - keep)):
        diff = v[:, None, :] - v[None, :, :]
        d = np.sqrt(np.sum(np.square(diff), axis=2))
        d[np.diag_indices(len(v))] = np.inf
        if d.min() < 100.0:
            min_index = int(np.unravel_index(np.argmin(d), d.shape)[0])
        v = np.delete(v, min_index, axis=0)
    return v


def evaluate_results(final_result: List[dict], gt_frames: Dict[int, dict], model_vertices, cam_K, diameter_mm,
                     pixel_thresh: float = 5.0):
    """The metric loop of betapose_evaluate.py:204-266.  ``gt_frames[nr]`` = list of ``{'pose': 4x4, 'bbox': [x, y, w, h]}`` (one per ground-truth
    annotation compared; a bare dict is accepted for one).
    Returns dict(mean_add, mean_2d_acc, mean_iou, n)."""
    add_errs, adds, proj, ious = [], [], [], []
    for f in final_result:
        nr = int(os.path.basename(f["imgname"])[0:-4])
        if nr not in gt_frames:
            continue
        entries = gt_frames[nr]
        if isinstance(entries, dict):
            entries = [entries]
        for gt in entries:
            if len(f["result"]) < 1 or len(f["result"][0]) < 1:
                continue
            x, y, w, h = gt["bbox"]
            gt_box = [x, y, x + w, y + h]
            pred_box = np.asarray(f["result"][0]["bbox"]).tolist()
            i = iou(gt_box, pred_box)
            ious.append(i)
            pose = np.eye(4)
            pose[:3, :3] = f["cam_R"]
            pose[:3, 3] = np.asarray(f["cam_t"])[:, 0]
            if i >= 0.5:
                a = add_err(gt["pose"], pose, model_vertices) * 1000
                add_errs.append(a)
                adds.append(a < diameter_mm / 10)
                proj.append(projection_error_2d(gt["pose"], pose, model_vertices, cam_K))
    return {"mean_add": float(np.mean(adds)) if adds else float("nan"),
            "mean_2d_acc": float(np.mean(np.array(proj) < pixel_thresh)) if proj else float("nan"),
            "mean_iou": float(np.mean(np.array(ious) > 0.5)) if ious else float("nan"),
            "mean_add_err_mm": float(np.mean(add_errs)) if add_errs else float("nan"), "n": len(ious)}


class Model3D:
    """The two methods of the reference's ``Model3D`` the harness uses (utils/model.py:29-46,79-85;
    betapose_evaluate.py:65-81): ``load(path, scale=...)`` of an ASCII .ply into ``vertices`` and ``refine(total_kp)``."""

    def __init__(self, file_to_load=None):
        self.vertices = None
        if file_to_load:
            self.load(file_to_load)

    def load(self, path, demean=False, scale=1.0):
        self.vertices = load_ply_vertices(path) * scale

    def refine(self, total_kp=30, save=False, save_path="test.ply"):
        self.vertices = refine_keypoints(self.vertices, total_kp)
