"""Heat-map decoding on the host side of the boundary (vectorised numpy, fp32).

``getPrediction`` keeps the reference's signature and return triple
(KPD/src/utils/eval.py:113-147) but needs only, per key point, the arg-max pixel,
its value and the four neighbours -- exactly the 6-float record the device arg-max
kernel returns (``bp_kpd_forward_argmax`` / the pipeline result row), so the
50x80x64 maps never have to leave the GPU.  ``decode_keypoints`` is that reduced
form; ``getPrediction`` builds the records from full maps and calls it.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def kp_records_from_heatmaps(hms: np.ndarray) -> np.ndarray:
    """hms [n,K,H,W] f32 -> records [n,K,6] f32 (idx as int32 bits, max, l, r, u, d) -- host twin
    of heatmap_argmax_kernel (first maximum wins, neighbours only for interior maxima)."""
    hms = np.ascontiguousarray(hms, dtype=F32)
    n, K, H, W = hms.shape
    flat = hms.reshape(n, K, H * W)
    idx = flat.argmax(axis=2)
    mx = np.take_along_axis(flat, idx[..., None], axis=2)[..., 0]
    x, y = idx % W, idx // W
    inner = (x > 0) & (x < W - 1) & (y > 0) & (y < H - 1)
    rec = np.zeros((n, K, 6), F32)
    rec[..., 0] = idx.astype(np.int32).view(F32)
    rec[..., 1] = mx

    def nb(off):
        j = np.clip(idx + off, 0, H * W - 1)
        return np.where(inner, np.take_along_axis(flat, j[..., None], axis=2)[..., 0], F32(0))

    rec[..., 2], rec[..., 3], rec[..., 4], rec[..., 5] = nb(-1), nb(1), nb(-W), nb(W)
    return rec


def transformBoxInvert_batch(pt: np.ndarray, ul: np.ndarray, br: np.ndarray, inpH, inpW, resH, resW) -> np.ndarray:
    """KPD/src/utils/img.py:216-239, fp32 op-for-op.  pt [n,K,2], ul/br [n,2]."""
    pt, ul, br = pt.astype(F32), ul.astype(F32), br.astype(F32)
    center = (br - F32(1) - ul) / F32(2)
    size = br - ul
    size[:, 0] = size[:, 0] * F32(inpH / inpW)
    lenH = size.max(axis=1)
    lenW = lenH * F32(inpW / inpH)
    _pt = (pt * lenH[:, None, None]) / F32(resH)
    dx = np.maximum((lenW - F32(1)) / F32(2) - center[:, 0], F32(0))
    dy = np.maximum((lenH - F32(1)) / F32(2) - center[:, 1], F32(0))
    out = np.empty_like(_pt)
    out[:, :, 0] = (_pt[:, :, 0] - dx[:, None]) + ul[:, 0][:, None]
    out[:, :, 1] = (_pt[:, :, 1] - dy[:, None]) + ul[:, 1][:, None]
    return out


def decode_keypoints(rec: np.ndarray, pt1: np.ndarray, pt2: np.ndarray, inpH=320, inpW=256, resH=80, resW=64):
    """rec [n,K,6] device/host arg-max records -> (preds_hm [n,K,2], preds_img [n,K,2], maxval [n,K,1])."""
    rec = np.ascontiguousarray(rec, dtype=F32)
    idx = rec[..., 0].view(np.int32)
    maxval = rec[..., 1]
    x = (idx % resW).astype(F32)
    y = (idx // resW).astype(F32)
    pos = maxval > 0
    x = np.where(pos, x, F32(0))
    y = np.where(pos, y, F32(0))
    inner = (x > 0) & (x < resW - 1) & (y > 0) & (y < resH - 1)
    sx = np.sign(rec[..., 3] - rec[..., 2]).astype(F32)
    sy = np.sign(rec[..., 5] - rec[..., 4]).astype(F32)
    x = x + np.where(inner, sx * F32(0.25), F32(0))
    y = y + np.where(inner, sy * F32(0.25), F32(0))
    preds = np.stack((x, y), axis=-1) + F32(0.2)
    preds_img = transformBoxInvert_batch(preds, np.asarray(pt1), np.asarray(pt2), inpH, inpW, resH, resW)
    return preds, preds_img, maxval[..., None].copy()


def getPrediction(hms, pt1, pt2, inpH, inpW, resH, resW):
    """Reference signature; accepts torch tensors or numpy arrays, returns torch tensors when given tensors."""
    is_torch = hasattr(hms, "detach")
    h = hms.detach().cpu().numpy() if is_torch else np.asarray(hms)
    assert h.ndim == 4, 'Score maps should be 4-dim'
    p1 = pt1.detach().cpu().numpy() if hasattr(pt1, "detach") else np.asarray(pt1)
    p2 = pt2.detach().cpu().numpy() if hasattr(pt2, "detach") else np.asarray(pt2)
    a, b, c = decode_keypoints(kp_records_from_heatmaps(h), p1, p2, inpH, inpW, resH, resW)
    if is_torch:
        import torch
        return torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(c)
    return a, b, c
