"""Low-margin flip rate of the arg-max stages under each matrix-core arithmetic (hard part (i)).

The path's integer outputs are arg-maxes: the YOLO box index = arg-max objectness over the candidates (yolo/util.py:210 with
NMS off), the key-point pixels = arg-max of each heat-map (KPD/src/utils/eval.py:113-131).  A flip against exact arithmetic is
possible only where the best and the second-best candidate are closer than the rounding of the convolution that produced them.
This module PLANTS such pairs in the layer that feeds each arg-max -- two candidates whose receptive fields are DIFFERENT vectors
(the filter's direction plus independent noise orthogonal to it) with fp64 responses exactly (1 - margin) apart -- and counts, per arithmetic, how often the kernel's arg-max differs from an fp64 convolution
of the same fp32 inputs.  No oracle import: the fp64 reference is torch's CPU conv2d on the same tensors.
"""
from __future__ import annotations

import numpy as np

MARGINS = (1e-3, 1e-4, 1e-5, 1e-6, 3e-7, 1e-7)
# product kernels per arithmetic: (3x3 conv_out class, 1x1 head class)
# 'f16r' (fp16 skip connections) differs from 'f16' in what the RESIDUAL adds read and in which tensors keep an fp32 copy; the two layers
# that feed the arg-maxes (conv_out, the YOLO heads) have no skip connection and read the same fp16 plane in both modes, so at this level the
# two columns are the same launches -- the column is there so that the bench line says so explicitly (round-4 verdict)
MODES = {"f32_mfma": ("64x64", "64x64"), "bf16x3": ("halo64_b3", "bd_b3"), "f16": ("pl64_f16", "pl64_f16"), "f16r": ("pl64_f16", "pl64_f16")}


def _ortho_noise(wvec, g, scale):
    """fp64 noise orthogonal to ``wvec`` with norm ``scale`` * |wvec|: adds nothing to the planted channel in exact arithmetic and
    rounding noise in every finite one."""
    import torch
    r = torch.randn(wvec.shape, generator=g, dtype=torch.float64)
    wv = wvec.double()
    r = r - (r * wv).sum() / (wv * wv).sum() * wv
    return r * (scale * wv.norm() / r.norm())


def _plant_pair(wvec, margin, g, alpha=6.0):
    """Two fp32 receptive fields a, b with fp64 responses o_a > o_b = o_a (1 - margin) to the filter ``wvec`` (up to the fp32
    rounding of b's rescale, ~1e-8 relative): different vectors, so finite arithmetic accumulates different rounding on them."""
    wv = wvec.double()
    a = (alpha * wv + _ortho_noise(wvec, g, alpha)).float()
    b = (alpha * wv + _ortho_noise(wvec, g, alpha)).float()
    o_a, o_b = (a.double() * wv).sum(), (b.double() * wv).sum()
    b = (b.double() * (o_a * (1.0 - margin) / o_b)).float()
    return a, b


def _planted_heatmap_case(margin: float, seed: int):
    """conv_out class: 3x3, 128 -> 50, 80x64, NCHW heat-maps.  For every channel c two 3x3 receptive fields respond to filter_c
    with the channel's two largest values, ``margin`` (relative) apart."""
    import torch
    g = torch.Generator().manual_seed(seed)
    C, K, H, W = 128, 50, 80, 64
    x = 0.05 * torch.randn(1, H, W, C, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) / np.sqrt(C * 9)
    spots = [(4 * i + 2, 4 * j + 2) for i in range(H // 4 - 1) for j in range(W // 4 - 1)]
    perm = torch.randperm(len(spots), generator=g).tolist()
    for c in range(K):
        (ya, xa), (yb, xb) = spots[perm[2 * c]], spots[perm[2 * c + 1]]
        pa, pb = _plant_pair(w[c].permute(1, 2, 0).contiguous(), margin, g)           # [3,3,C], NHWC
        x[0, ya - 1:ya + 2, xa - 1:xa + 2, :] = pa
        x[0, yb - 1:yb + 2, xb - 1:xb + 2, :] = pb
    return x, w


def _planted_head_case(margin: float, seed: int):
    """YOLO head class: 1x1, 1024 -> 18 on 13x13; the objectness logits are channels 4, 10, 16.  Two cells respond to filter_4 with
    the frame's two largest objectness logits, ``margin`` apart."""
    import torch
    g = torch.Generator().manual_seed(seed)
    C, H, W = 1024, 13, 13
    x = 0.05 * torch.randn(1, H, W, C, generator=g)
    w = torch.randn(18, C, 1, 1, generator=g) / np.sqrt(C)
    cells = torch.randperm(H * W, generator=g)[:2].tolist()
    va, vb = _plant_pair(w[4, :, 0, 0].contiguous(), margin, g)
    x[0, cells[0] // W, cells[0] % W, :] = va
    x[0, cells[1] // W, cells[1] % W, :] = vb
    return x, w


def _argmax_heat(y_nchw):
    return y_nchw.reshape(y_nchw.shape[1], -1).argmax(1)          # first maximum, as getPrediction's torch.max


def _argmax_obj(y_nhwc):
    o = y_nhwc[0][:, :, [4, 10, 16]].permute(2, 0, 1).reshape(-1)   # DetectionLayer row order: anchor, gy, gx
    return o.argmax()


def measure(device="cuda:0", trials: int = 4, modes=None):
    """{mode: {"heatmap": {margin: flips of trials*50}, "objectness": {margin: flips of trials*8}}} against fp64 convolutions."""
    import torch
    import torch.nn.functional as F
    from . import ops
    modes = modes or list(MODES)
    out = {m: {"heatmap": {}, "objectness": {}} for m in modes}
    counts = {"heatmap": trials * 50, "objectness": trials * 8}
    for margin in MARGINS:
        flips = {m: [0, 0] for m in modes}
        for t in range(trials):
            x, w = _planted_heatmap_case(margin, 9000 + t)
            ref = _argmax_heat(F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), padding=1))
            for m in modes:
                y = ops.conv2d_nhwc(x.to(device), w, None, pad=1, store="nchw", tile=MODES[m][0], splits=1).cpu()
                flips[m][0] += int((_argmax_heat(y) != ref).sum())
            for u in range(8):
                x, w = _planted_head_case(margin, 9500 + 8 * t + u)
                ref = _argmax_obj(F.conv2d(x.double().permute(0, 3, 1, 2), w.double()).permute(0, 2, 3, 1))
                for m in modes:
                    y = ops.conv2d_nhwc(x.to(device), w, None, tile=MODES[m][1], splits=4).cpu()
                    flips[m][1] += int(_argmax_obj(y) != ref)
        for m in modes:
            out[m]["heatmap"]["%g" % margin] = flips[m][0]
            out[m]["objectness"]["%g" % margin] = flips[m][1]
    return {"margins_relative": list(MARGINS), "candidates": counts, "flips_vs_fp64": out,
            "f16r_note": "the layers that feed the arg-maxes have no skip connection: 'f16r' runs them exactly as 'f16' does (same launches, same counts); "
                         "the mode's effect on whole networks is in tests/test_gpu_nets.py::test_f16r_mode_fp16_skip_connections",
            "definition": "per arg-max two planted candidates (filter direction + independent orthogonal noise) whose fp64 responses are (1 - margin) apart; flips = arg-max of the kernel's "
                          "output != arg-max of an fp64 convolution of the same fp32 inputs (torch CPU)"}
