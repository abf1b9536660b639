"""Seeded synthetic inputs and weights (no LineMod data, ``.weights`` or ``.pkl``
ship with the reference -- SURVEY.md §0 F5, §8(d)).

Everything here draws from ``numpy.random.Generator(PCG64(seed))`` whose stream
is stable across numpy versions, so the same arrays are regenerated on the GPU
box, in the oracle tools and in the tests.  The distributions are chosen so
activations stay O(1) through 75 / 104 conv layers (residual branches are
damped) and the detector's objectness / the heat-map maxima are not saturated.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np

from .cfg import parse_cfg_text, yolov3_single_cfg_text
from .weights import darknet_stream_layout, fastpose_modules

CAM_K = np.array([[572.4114, 0.0, 325.2611],
                  [0.0, 573.57043, 242.04899],
                  [0.0, 0.0, 1.0]])  # betapose_evaluate.py:59


def _rng(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(seed))


def _bn(rng, c: int, gamma_lo=0.8, gamma_hi=1.2):
    beta = rng.uniform(-0.1, 0.1, c).astype(np.float32)
    gamma = rng.uniform(gamma_lo, gamma_hi, c).astype(np.float32)
    mean = rng.uniform(-0.1, 0.1, c).astype(np.float32)
    var = rng.uniform(0.5, 1.5, c).astype(np.float32)
    return beta, gamma, mean, var


def synth_yolo_stream(seed: int = 1, blocks=None, head_gain: float = 0.25) -> np.ndarray:
    """fp32 stream (payload of a ``.weights`` file) for the YOLOv3 cfg."""
    if blocks is None:
        blocks = parse_cfg_text(yolov3_single_cfg_text())
    rng = _rng(seed)
    table = darknet_stream_layout(blocks)
    last = table[-1]
    flat = np.empty(last["weight"] + last["cout"] * last["cin"] * last["k"] ** 2, np.float32)
    for ent in table:
        co, ci, k = ent["cout"], ent["cin"], ent["k"]
        fan_in = ci * k * k
        idx = ent["index"]
        before_shortcut = idx + 1 < len(blocks) and blocks[idx + 1]["type"] == "shortcut"
        if ent["bn"]:
            lo, hi = (0.1, 0.2) if before_shortcut else (0.8, 1.2)
            beta, gamma, mean, var = _bn(rng, co, lo, hi)
            flat[ent["bn_bias"]:ent["bn_bias"] + co] = beta
            flat[ent["bn_weight"]:ent["bn_weight"] + co] = gamma
            flat[ent["bn_mean"]:ent["bn_mean"] + co] = mean
            flat[ent["bn_var"]:ent["bn_var"] + co] = var
            std = np.sqrt(2.0 / fan_in)
        else:
            # detection head: linear 1x1 conv with bias; objectness biased low so
            # that only a handful of the 10 647 candidates clear the threshold
            bias = rng.uniform(-0.5, 0.5, co).astype(np.float32)
            bias[4::6] -= 3.0
            flat[ent["bias"]:ent["bias"] + co] = bias
            std = head_gain * np.sqrt(1.0 / fan_in)
        w = rng.standard_normal(co * fan_in, dtype=np.float32) * np.float32(std)
        flat[ent["weight"]:ent["weight"] + co * fan_in] = w
    return flat


def synth_fastpose_state_dict(seed: int = 2, n_classes: int = 50,
                              out_gain: float = 1.0) -> Dict[str, np.ndarray]:
    """Random FastPose ``state_dict`` (numpy arrays, ``state_dict()`` key names)."""
    rng = _rng(seed)
    sd: Dict[str, np.ndarray] = {}
    for m in fastpose_modules(n_classes):
        n = m["name"]
        if m["kind"] == "conv":
            co, ci, k = m["cout"], m["cin"], m["k"]
            fan_in = ci * k * k
            if m["bn"]:
                damp = n.endswith(".conv3")
                lo, hi = (0.1, 0.2) if damp else (0.8, 1.2)
                beta, gamma, mean, var = _bn(rng, co, lo, hi)
                b = m["bn"]
                sd[b + ".weight"], sd[b + ".bias"] = gamma, beta
                sd[b + ".running_mean"], sd[b + ".running_var"] = mean, var
                std = np.sqrt(2.0 / fan_in)
                if n.endswith("downsample.0"):
                    std = np.sqrt(1.0 / fan_in)
            else:
                sd[n + ".bias"] = rng.uniform(0.2, 0.6, co).astype(np.float32)
                std = out_gain * np.sqrt(1.0 / fan_in)
            w = rng.standard_normal(co * fan_in, dtype=np.float32) * np.float32(std)
            sd[n + ".weight"] = w.reshape(co, ci, k, k)
        else:
            co, ci = m["cout"], m["cin"]
            w = rng.standard_normal(co * ci, dtype=np.float32) * np.float32(np.sqrt(1.0 / ci))
            sd[n + ".weight"] = w.reshape(co, ci)
            sd[n + ".bias"] = rng.uniform(-0.2, 0.2, co).astype(np.float32)
    return sd


def object_seeds(obj_id: int):
    """(detector seed, key-point-net seed) of the seeded synthetic weights of object ``obj_id``: (1, 2) for object 1
    (the fixtures' weights), distinct streams for the others so several resident objects really differ."""
    return (1, 2) if int(obj_id) == 1 else (1 + 16 * int(obj_id), 2 + 16 * int(obj_id))


def synth_frame(seed: int = 1234, h: int = 480, w: int = 640) -> np.ndarray:
    """LineMod-shaped BGR u8 frame: smooth low-res field + noise."""
    rng = _rng(seed)
    gh, gw = h // 32 + 2, w // 32 + 2
    low = rng.uniform(0, 255, (gh, gw, 3))
    ys = np.linspace(0, gh - 1.001, h)
    xs = np.linspace(0, gw - 1.001, w)
    y0 = np.floor(ys).astype(int)
    x0 = np.floor(xs).astype(int)
    fy = (ys - y0)[:, None, None]
    fx = (xs - x0)[None, :, None]
    a = low[y0][:, x0]
    b = low[y0][:, x0 + 1]
    c = low[y0 + 1][:, x0]
    d = low[y0 + 1][:, x0 + 1]
    img = (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy
    img = img + rng.normal(0, 8, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def synth_frames(n: int, seed: int = 1234) -> List[np.ndarray]:
    return [synth_frame(seed + i) for i in range(n)]


def synth_kp3d(n: int = 50, seed: int = 7, radius: float = 0.06) -> np.ndarray:
    """Non-planar 3-D key points in metres (stand-in for the designated
    key points ``assets/sifts/*.ply`` x0.001; LineMod objects are ~0.1 m)."""
    rng = _rng(seed)
    p = rng.uniform(-1, 1, (n, 3))
    return (p * np.array([radius, radius * 0.7, radius * 0.5])).astype(np.float64)


def _write_ply(path: str, pts: np.ndarray) -> None:
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                "end_header\n" % len(pts))
        for x, y, z in pts:
            f.write("%.6f %.6f %.6f\n" % (x, y, z))


def write_sixd_tree(base: str, seq: int, gt_by_frame: Dict[int, list], models_mm: Dict[int, np.ndarray],
                    kpmodels_mm: Dict[int, np.ndarray], diameters_mm: Dict[int, float], cam_K=None) -> None:
    """Lay out a SIXD-format ground-truth tree the harness reads (utils/sixd.py:60-111; betapose_evaluate.py:60-75):
    ``camera.yml``, ``models/models_info.yml``, ``models/obj_XX.ply``, ``kpmodels/obj_XX.ply`` (millimetres) and
    ``test/<seq>/{gt.yml, info.yml}``.  ``gt_by_frame[nr]`` = list of ``(obj_id, R[3,3], t_mm[3], bbox[x,y,w,h])``."""
    import os
    import yaml
    K = CAM_K if cam_K is None else np.asarray(cam_K, dtype=np.float64)
    os.makedirs(os.path.join(base, "models"), exist_ok=True)
    os.makedirs(os.path.join(base, "kpmodels"), exist_ok=True)
    sdir = os.path.join(base, "test", "%02d" % seq)
    os.makedirs(os.path.join(sdir, "rgb"), exist_ok=True)
    with open(os.path.join(base, "camera.yml"), "w") as f:
        yaml.safe_dump({"fx": float(K[0, 0]), "fy": float(K[1, 1]), "cx": float(K[0, 2]), "cy": float(K[1, 2]),
                        "depth_scale": 1.0, "width": 640, "height": 480}, f)
    with open(os.path.join(base, "models", "models_info.yml"), "w") as f:
        yaml.safe_dump({int(k): {"diameter": float(v)} for k, v in sorted(diameters_mm.items())}, f)
    for oid, pts in models_mm.items():
        _write_ply(os.path.join(base, "models", "obj_%02d.ply" % oid), np.asarray(pts))
    for oid, pts in kpmodels_mm.items():
        _write_ply(os.path.join(base, "kpmodels", "obj_%02d.ply" % oid), np.asarray(pts))
    gt, info = {}, {}
    for nr, objs in sorted(gt_by_frame.items()):
        gt[int(nr)] = [{"obj_id": int(o), "cam_R_m2c": [float(v) for v in np.asarray(R).reshape(9)],
                        "cam_t_m2c": [float(v) for v in np.asarray(t).reshape(3)],
                        "obj_bb": [float(v) for v in bb]} for (o, R, t, bb) in objs]
        info[int(nr)] = {"cam_K": [float(v) for v in K.reshape(9)], "depth_scale": 1.0}
    with open(os.path.join(sdir, "gt.yml"), "w") as f:
        yaml.safe_dump(gt, f)
    with open(os.path.join(sdir, "info.yml"), "w") as f:
        yaml.safe_dump(info, f)
