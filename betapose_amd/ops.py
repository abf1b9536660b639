"""Stand-alone device stages of the hot path (thin ctypes wrappers, torch tensors
as device memory): fused conv (unit-test / benchmark hook), crop, Pillow-exact
bicubic resize, PnP."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib

ACT = {"linear": 0, "leaky": 1, "relu": 2}
STORE = {"nhwc": 0, "up2": 1, "pixshuf": 2, "nchw": 3}
# + 256: fp16 operands, + 512: bf16x3 (fp32-accurate) operands.  pl<BM>[x<BN>] = conv_pl.hip (operand planes + LDS-DMA);
# w<WM>x<WN> = conv_w64.hip with WM x WN waves of 64x64; kg / rd / bd: the round-2 experiments (csrc/bp_common.h ConvTile)
_F16, _B3 = 256, 512
TILE = {"auto": -1, "64x64": 0, "128x64": 1, "stem3": 20, "stem7": 28, "64x64_f16": _F16, "128x64_f16": _F16 + 1, "64x64_b3": _B3, "bd_b3": _B3 + 12,
        "bd_f16": _F16 + 12, "halo64_b3": _B3 + 21, "halo128_b3": _B3 + 22, "halo64k2_b3": _B3 + 23, "bdk2_b3": _B3 + 24}
for _n, _i in (("w1x1", 2), ("w1x2", 3), ("w2x1", 5), ("w2x2", 6), ("pl64", 13), ("pl128", 14), ("pl128x64", 15), ("pl256x128", 16), ("pl128s", 17), ("pl64k2", 18), ("pl64bd", 19), ("plh128", 25), ("s1", 26), ("p3", 27)):
    TILE[_n + "_f16"] = _F16 + _i
    TILE[_n + "_b3"] = _B3 + _i
for _n, _i in (("kg1", 7), ("kg2", 8), ("kg4", 9), ("rd4", 10), ("rd8", 11)):
    TILE[_n + "_b3"] = _B3 + _i


def conv2d_nhwc(x, weight, bias=None, stride: int = 1, pad: int = 0, act: str = "linear", store: str = "nhwc",
                res=None, res_after_act: bool = False, tile: str = "auto", splits: int = 0, iters: int = 0,
                planes: bool = False):
    """One fused convolution.  ``x``: cuda f32 [N,H,W,Cin] (NHWC); ``weight``: host
    numpy/torch [Cout,Cin,k,k]; returns the output tensor laid out per ``store`` and,
    when ``iters`` > 0, also the measured ms per launch.  ``planes``: also return the operand planes the epilogue
    emits for the next layer (int16 tensor [np, *out.shape]: np = 1 fp16 bits / 3 bf16 bits)."""
    import torch
    _lib.require_gpu()
    w = np.ascontiguousarray(weight.detach().cpu().numpy() if hasattr(weight, "detach") else weight, dtype=np.float32)
    b = None
    if bias is not None:
        b = np.ascontiguousarray(bias.detach().cpu().numpy() if hasattr(bias, "detach") else bias, dtype=np.float32)
    N, H, W, Cin = x.shape
    Cout, _, k, _ = w.shape
    OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    x = x.contiguous()
    if store == "nhwc":
        out = torch.empty((N, OH, OW, Cout), device=x.device, dtype=torch.float32)
    elif store == "up2":
        out = torch.empty((N, 2 * OH, 2 * OW, Cout), device=x.device, dtype=torch.float32)
    elif store == "pixshuf":
        out = torch.empty((N, 2 * OH, 2 * OW, Cout // 4), device=x.device, dtype=torch.float32)
    else:
        out = torch.empty((N, Cout, OH, OW), device=x.device, dtype=torch.float32)
    ms = C.c_float(0)
    pl = None
    if planes:
        assert store != "nchw", "NCHW outputs (the heat-maps) have no operand planes"
        pl = torch.zeros((1 if TILE[tile] < _B3 else 3,) + tuple(out.shape), device=x.device, dtype=torch.int16)
    _lib.check(_lib.lib().bp_conv2d_planes(x.data_ptr(), N, H, W, Cin, w.ctypes.data, b.ctypes.data if b is not None else None,
                                           Cout, k, stride, pad, ACT[act], STORE[store],
                                           res.contiguous().data_ptr() if res is not None else None, int(res_after_act),
                                           TILE[tile], int(splits), out.data_ptr(), pl.data_ptr() if planes else None,
                                           int(iters), C.byref(ms), _lib.current_stream()))
    ret = (out,) + ((pl,) if planes else ()) + ((ms.value,) if iters > 0 else ())
    return ret if len(ret) > 1 else out


def crop(frames_bgr_u8, sel=None, boxes=None, reso: int = 416, oh: int = 320, ow: int = 256, nchw: bool = True):
    """Device crop (dataloader.py:354-364,794-835; img.py:242-262).  ``frames``: cuda u8
    [B,H,W,3] BGR; ``sel`` [B,8] (YOLO-input-pixel boxes) or ``boxes`` [B,4] (frame pixels).
    Returns (inps [B,3,oh,ow] or [B,oh,ow,3], pts [B,8] = pt1.x,pt1.y,pt2.x,pt2.y, box x1,y1,x2,y2)."""
    import torch
    _lib.require_gpu()
    f = frames_bgr_u8.contiguous()
    B, H, W, _ = f.shape
    out = torch.empty((B, 3, oh, ow) if nchw else (B, oh, ow, 3), device=f.device, dtype=torch.float32)
    pts = torch.empty((B, 8), device=f.device, dtype=torch.float32)
    _lib.check(_lib.lib().bp_crop(f.data_ptr(), B, H, W, sel.contiguous().data_ptr() if sel is not None else None, reso,
                                  boxes.contiguous().data_ptr() if boxes is not None else None,
                                  out.data_ptr() if nchw else None, None if nchw else out.data_ptr(), pts.data_ptr(),
                                  oh, ow, _lib.current_stream()))
    return out, pts


def resize_bicubic(frames_u8, oh: int = 416, ow: int = 416, swap_rb: bool = True, want: str = "f32"):
    """Pillow-exact antialiased bicubic (dataloader.py:94-99).  ``frames``: cuda u8 [B,H,W,3].
    ``want`` 'u8' -> u8 [B,oh,ow,3]; 'f32' -> f32 NHWC /255."""
    import torch
    _lib.require_gpu()
    f = frames_u8.contiguous()
    B, H, W, _ = f.shape
    if want == "u8":
        out = torch.empty((B, oh, ow, 3), device=f.device, dtype=torch.uint8)
        _lib.check(_lib.lib().bp_resize_bicubic(f.data_ptr(), B, H, W, oh, ow, int(swap_rb), out.data_ptr(), None,
                                                _lib.current_stream()))
    else:
        out = torch.empty((B, oh, ow, 3), device=f.device, dtype=torch.float32)
        _lib.check(_lib.lib().bp_resize_bicubic(f.data_ptr(), B, H, W, oh, ow, int(swap_rb), None, out.data_ptr(),
                                                _lib.current_stream()))
    return out


def solve_pnp(points_3d, points_2d, K, method: str = "iterative"):
    """utils/utils.py:17-41 ``pnp``: returns (R [3,3], t [3,1]) f64.  Host-only (no GPU needed).
    ``method``: 'iterative' = the restatement of cv2.solvePnP's SOLVEPNP_ITERATIVE (what the reference calls);
    'refined' = conditioned DLT + converged minimiser (opt-in, csrc/host_post.cpp)."""
    p3 = np.ascontiguousarray(points_3d, dtype=np.float64)
    p2 = np.ascontiguousarray(np.asarray(points_2d)[:, :2], dtype=np.float64)
    assert p3.shape[0] == p2.shape[0], "points 3D and points 2D must have same number of vertices"
    Kc = np.ascontiguousarray(K, dtype=np.float64)
    R = np.empty((3, 3), np.float64)
    t = np.empty(3, np.float64)
    fn = {"iterative": _lib.lib().bp_solve_pnp, "refined": _lib.lib().bp_solve_pnp_refined}[method]
    _lib.check(fn(p3.ctypes.data, p2.ctypes.data, p3.shape[0], Kc.ctypes.data, R.ctypes.data, t.ctypes.data))
    return R, t.reshape(3, 1)


def solve_pnp_ransac(points_3d, points_2d, K, reprojection_error: float = 12.0, iterations: int = 100,
                     confidence: float = 0.99):
    """The variant utils/utils.py:32-36 keeps commented out (cv2.solvePnPRansac, reprojectionError=12.0): returns
    (R [3,3], t [3,1], inlier mask [n] bool).  Host-only."""
    p3 = np.ascontiguousarray(points_3d, dtype=np.float64)
    p2 = np.ascontiguousarray(np.asarray(points_2d)[:, :2], dtype=np.float64)
    assert p3.shape[0] == p2.shape[0], "points 3D and points 2D must have same number of vertices"
    Kc = np.ascontiguousarray(K, dtype=np.float64)
    R = np.empty((3, 3), np.float64)
    t = np.empty(3, np.float64)
    inl = np.zeros(p3.shape[0], np.uint8)
    _lib.check(_lib.lib().bp_solve_pnp_ransac(p3.ctypes.data, p2.ctypes.data, p3.shape[0], Kc.ctypes.data,
                                              float(reprojection_error), int(iterations), float(confidence),
                                              R.ctypes.data, t.ctypes.data, inl.ctypes.data))
    return R, t.reshape(3, 1), inl.astype(bool)


def heatmap_argmax(hm):
    """Device arg-max records of a heat-map tensor: cuda f32 [B,K,H,W] -> [B,K,6] (idx as int bits, max, l, r, u, d)."""
    import torch
    _lib.require_gpu()
    hm = hm.contiguous().float()
    B, K, H, W = hm.shape
    kp = torch.empty((B, K, 6), device=hm.device, dtype=torch.float32)
    _lib.check(_lib.lib().bp_heatmap_argmax(hm.data_ptr(), B, K, H, W, kp.data_ptr(), _lib.current_stream()))
    return kp
