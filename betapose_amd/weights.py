"""Weight-file formats of the hot path.

* Darknet ``.weights`` (detector): header ``int32[major, minor, revision]`` then
  ``seen`` as int32 (``major*10+minor < 2``) or uint64 (otherwise,
  train_YOLO/src/parser.c:1161-1174), then a flat fp32 stream, per
  ``[convolutional]`` block in cfg order: ``bn.bias, bn.weight(scale),
  running_mean, running_var`` (or ``conv.bias`` when the block has no
  ``batch_normalize``), then ``conv.weight`` in ``[out_c, in_c, k, k]`` order
  (yolo/darknet.py:365-432; parser.c:1103-1146).  The reference's Python
  loader always skips exactly 16 bytes; we accept both header sizes.
* KPD ``.pkl``: ``torch.save(FastPose().state_dict())``
  (KPD/src/main_fast_inference.py:29-37).  For the C-ABI the state dict is
  flattened into the same kind of stream ("KPD stream", documented in
  include/betapose_hip.h): convs in module-definition order with Darknet's
  per-conv field order, SE linears as ``fc0.weight, fc0.bias, fc2.weight,
  fc2.bias``.
"""
from __future__ import annotations

import struct
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np

from .cfg import conv_blocks

# ----------------------------------------------------------------------------
# Darknet .weights
# ----------------------------------------------------------------------------


def read_darknet_weights(path: str) -> Tuple[np.ndarray, int, np.ndarray]:
    """Return ``(version[3] int32, seen, flat fp32 stream)``."""
    with open(path, "rb") as f:
        raw = f.read()
    if len(raw) < 16:
        raise ValueError("%s: truncated .weights header" % path)
    major, minor, revision = struct.unpack_from("<iii", raw, 0)
    if major * 10 + minor >= 2:
        if len(raw) < 20:
            raise ValueError("%s: truncated .weights header" % path)
        (seen,) = struct.unpack_from("<Q", raw, 12)
        off = 20
    else:
        (seen,) = struct.unpack_from("<i", raw, 12)
        off = 16
    if (len(raw) - off) % 4:
        raise ValueError("%s: payload is not a whole number of fp32" % path)
    flat = np.frombuffer(raw, dtype="<f4", offset=off).copy()
    return np.array([major, minor, revision], dtype=np.int32), int(seen), flat


def write_darknet_weights(path: str, flat: np.ndarray, seen: int = 0,
                          version: Sequence[int] = (0, 1, 0)) -> None:
    major, minor, revision = [int(v) for v in version]
    with open(path, "wb") as f:
        f.write(struct.pack("<iii", major, minor, revision))
        if major * 10 + minor >= 2:
            f.write(struct.pack("<Q", seen))
        else:
            f.write(struct.pack("<i", seen))
        f.write(np.ascontiguousarray(flat, dtype="<f4").tobytes())


def darknet_stream_layout(blocks) -> List[dict]:
    """Offset table of the fp32 stream for a parsed cfg: one entry per conv
    block with field offsets (in floats)."""
    table = []
    ptr = 0
    for idx, b, cin, cout in conv_blocks(blocks):
        k = int(b["size"])
        bn = int(b.get("batch_normalize", 0)) > 0
        ent = {"index": idx, "cin": cin, "cout": cout, "k": k,
               "stride": int(b["stride"]), "bn": bn, "activation": b["activation"]}
        if bn:
            for name in ("bn_bias", "bn_weight", "bn_mean", "bn_var"):
                ent[name] = ptr
                ptr += cout
        else:
            ent["bias"] = ptr
            ptr += cout
        ent["weight"] = ptr
        ptr += cout * cin * k * k
        table.append(ent)
    return table


def darknet_stream_size(blocks) -> int:
    t = darknet_stream_layout(blocks)
    last = t[-1]
    return last["weight"] + last["cout"] * last["cin"] * last["k"] ** 2


def split_darknet_stream(blocks, flat: np.ndarray) -> List[dict]:
    """Per-conv arrays (views into ``flat``)."""
    need = darknet_stream_size(blocks)
    if flat.size < need:
        raise ValueError("weights stream too short: %d < %d floats" % (flat.size, need))
    out = []
    for ent in darknet_stream_layout(blocks):
        co, ci, k = ent["cout"], ent["cin"], ent["k"]
        d = dict(ent)
        if ent["bn"]:
            for name in ("bn_bias", "bn_weight", "bn_mean", "bn_var"):
                d[name] = flat[ent[name]:ent[name] + co]
        else:
            d["bias"] = flat[ent["bias"]:ent["bias"] + co]
        d["weight"] = flat[ent["weight"]:ent["weight"] + co * ci * k * k].reshape(co, ci, k, k)
        out.append(d)
    return out


# ----------------------------------------------------------------------------
# FastPose (SE-ResNet-101 + DUC) structure -- KPD/src/models/FastPose.py:13-35,
# layers/SE_Resnet.py:6-99, layers/SE_module.py:4-19, layers/DUC.py:5-23
# ----------------------------------------------------------------------------

FASTPOSE_STAGES = ((64, 3, 1), (128, 4, 2), (256, 23, 2), (512, 3, 2))  # planes, blocks, stride


def fastpose_modules(n_classes: int = 50) -> List[dict]:
    """Ordered module list (state-dict order) of FastPose.

    Entries: ``{"kind": "conv", "name", "cin", "cout", "k", "stride", "pad",
    "bn": bn_name|None, "bias": bool}`` or ``{"kind": "linear", "name", "cin",
    "cout"}``.
    """
    mods: List[dict] = []

    def conv(name, cin, cout, k, stride, pad, bn=None, bias=False):
        mods.append({"kind": "conv", "name": name, "cin": cin, "cout": cout, "k": k,
                     "stride": stride, "pad": pad, "bn": bn, "bias": bias})

    conv("preact.conv1", 3, 64, 7, 2, 3, bn="preact.bn1")
    inplanes = 64
    for li, (planes, nblocks, stride) in enumerate(FASTPOSE_STAGES, start=1):
        for bi in range(nblocks):
            p = "preact.layer%d.%d" % (li, bi)
            s = stride if bi == 0 else 1
            first = bi == 0  # make_layer: first block always has a downsample here
            conv(p + ".conv1", inplanes, planes, 1, 1, 0, bn=p + ".bn1")
            conv(p + ".conv2", planes, planes, 3, s, 1, bn=p + ".bn2")
            conv(p + ".conv3", planes, planes * 4, 1, 1, 0, bn=p + ".bn3")
            if first:
                mods.append({"kind": "linear", "name": p + ".se.fc.0", "cin": planes * 4, "cout": planes * 4})
                mods.append({"kind": "linear", "name": p + ".se.fc.2", "cin": planes * 4, "cout": planes * 4})
                conv(p + ".downsample.0", inplanes, planes * 4, 1, s, 0, bn=p + ".downsample.1")
            inplanes = planes * 4
    conv("duc1.conv", 512, 1024, 3, 1, 1, bn="duc1.bn")
    conv("duc2.conv", 256, 512, 3, 1, 1, bn="duc2.bn")
    conv("conv_out", 128, n_classes, 3, 1, 1, bn=None, bias=True)
    return mods


def fastpose_state_dict_keys(n_classes: int = 50) -> List[Tuple[str, Tuple[int, ...]]]:
    """(key, shape) in ``state_dict()`` order, ``num_batches_tracked`` omitted."""
    keys = []
    for m in fastpose_modules(n_classes):
        if m["kind"] == "conv":
            keys.append((m["name"] + ".weight", (m["cout"], m["cin"], m["k"], m["k"])))
            if m["bias"]:
                keys.append((m["name"] + ".bias", (m["cout"],)))
            if m["bn"]:
                for f in ("weight", "bias", "running_mean", "running_var"):
                    keys.append((m["bn"] + "." + f, (m["cout"],)))
        else:
            keys.append((m["name"] + ".weight", (m["cout"], m["cin"])))
            keys.append((m["name"] + ".bias", (m["cout"],)))
    return keys


def _np(t) -> np.ndarray:
    if hasattr(t, "detach"):
        t = t.detach().cpu().numpy()
    return np.ascontiguousarray(t, dtype=np.float32)


def fastpose_stream_from_state_dict(sd: Dict[str, object], n_classes: int = 50) -> np.ndarray:
    """Flatten a FastPose state dict into the KPD stream the C-ABI consumes."""
    chunks = []
    for m in fastpose_modules(n_classes):
        n = m["name"]
        if m["kind"] == "conv":
            w = _np(sd[n + ".weight"])
            if w.shape != (m["cout"], m["cin"], m["k"], m["k"]):
                raise ValueError("%s.weight has shape %s" % (n, w.shape))
            if m["bn"]:
                b = m["bn"]
                chunks += [_np(sd[b + ".bias"]), _np(sd[b + ".weight"]),
                           _np(sd[b + ".running_mean"]), _np(sd[b + ".running_var"])]
            else:
                chunks.append(_np(sd[n + ".bias"]))
            chunks.append(w.ravel())
        else:
            chunks += [_np(sd[n + ".weight"]).ravel(), _np(sd[n + ".bias"])]
    return np.concatenate([c.ravel() for c in chunks]).astype(np.float32)


def fastpose_stream_size(n_classes: int = 50) -> int:
    n = 0
    for m in fastpose_modules(n_classes):
        if m["kind"] == "conv":
            n += m["cout"] * m["cin"] * m["k"] ** 2 + (4 * m["cout"] if m["bn"] else m["cout"])
        else:
            n += m["cout"] * m["cin"] + m["cout"]
    return n


def load_kpd_pkl(path: str) -> Dict[str, np.ndarray]:
    """``torch.load`` of a FastPose ``state_dict`` -> numpy dict (host only).

    Checkpoints are loaded with torch's restricted unpickler (tensors and plain containers only).  A file that needs
    arbitrary pickled classes -- e.g. a whole pickled module -- is refused unless ``BP_TRUST_PKL=1`` is set, because
    unpickling executes code from the file (the reference's ``torch.load`` does so unconditionally)."""
    import os
    import pickle
    import torch
    try:
        sd = torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as e:
        if os.environ.get("BP_TRUST_PKL") != "1":
            raise ValueError("%s is not a plain tensor state_dict (%s); set BP_TRUST_PKL=1 to unpickle it anyway "
                             "(runs code from the file)" % (path, str(e).splitlines()[0])) from e
        sd = torch.load(path, map_location="cpu", weights_only=False)
    if hasattr(sd, "state_dict"):
        sd = sd.state_dict()
    return {k: v.detach().cpu().numpy() for k, v in sd.items() if hasattr(v, "detach")}


# ----------------------------------------------------------------------------
# BN folding (host side; the engine does the same in C++ at load time)
# ----------------------------------------------------------------------------

def fold_bn(weight: np.ndarray, gamma, beta, mean, var, eps: float = 1e-5,
            darknet_eps: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """Fold an eval-mode BatchNorm into the preceding conv.

    PyTorch path: ``(x-mean)/sqrt(var+1e-5)*gamma+beta`` (yolo/darknet.py:256).
    Darknet-C path (``darknet_eps``): ``(x-mean)/(sqrt(var)+1e-6)``
    (train_YOLO/src/blas.c:136, network.c:827-834).
    """
    w = weight.astype(np.float64)
    g, b, m, v = [np.asarray(a, dtype=np.float64) for a in (gamma, beta, mean, var)]
    s = g / (np.sqrt(v) + 1e-6) if darknet_eps else g / np.sqrt(v + eps)
    wf = w * s.reshape(-1, *([1] * (w.ndim - 1)))
    bf = b - m * s
    return wf.astype(np.float32), bf.astype(np.float32)
