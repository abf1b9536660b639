"""ctypes binding of libbetapose_hip.so (include/betapose_hip.h).

The library is the product: if it is missing it is built in-tree with hipcc, and
if that is impossible the import of any engine class FAILS LOUDLY -- there is no
CPU or eager-PyTorch fallback on the product path.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# BP_LIB: an alternative build of the same library (tools only: `python -m betapose_amd.build --experimental` makes
# libbetapose_hip_exp.so with the measured-and-rejected kernels and the timing ablations compiled in)
LIB_PATH = os.environ.get("BP_LIB") or os.path.join(_HERE, "libbetapose_hip.so")
_lock = threading.Lock()
_lib = None

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/betapose_hip.h declares
PROTOTYPES = {
    "bp_last_error": (C.c_char_p, []),
    "bp_version": (C.c_int, []),
    "bp_device_count": (C.c_int, []),
    "bp_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "bp_yolo_create": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "bp_yolo_create_from_memory": (C.c_int, [C.c_char_p, vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "bp_yolo_clone": (C.c_int, [vp, C.POINTER(vp)]),
    "bp_kpd_clone": (C.c_int, [vp, C.POINTER(vp)]),
    "bp_yolo_destroy": (None, [vp]),
    "bp_yolo_rows": (C.c_int, [vp]),
    "bp_yolo_attrs": (C.c_int, [vp]),
    "bp_yolo_forward": (C.c_int, [vp, vp, C.c_int, vp, vp]),
    "bp_yolo_forward_select": (C.c_int, [vp, vp, C.c_int, C.c_float, C.c_int, vp, vp, vp]),
    "bp_yolo_select": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, vp, vp]),
    "bp_yolo_tap_count": (C.c_int, [vp]),
    "bp_yolo_tap_info": (C.c_int, [vp, C.c_int, C.c_char_p, C.c_int, c_int_p, c_int_p, c_int_p]),
    "bp_yolo_tap_copy": (C.c_int, [vp, C.c_int, C.c_int, vp, vp]),
    "bp_kpd_create": (C.c_int, [vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "bp_kpd_destroy": (None, [vp]),
    "bp_kpd_forward": (C.c_int, [vp, vp, C.c_int, vp, vp]),
    "bp_kpd_forward_argmax": (C.c_int, [vp, vp, C.c_int, vp, vp, vp]),
    "bp_kpd_tap_count": (C.c_int, [vp]),
    "bp_kpd_tap_info": (C.c_int, [vp, C.c_int, C.c_char_p, C.c_int, c_int_p, c_int_p, c_int_p]),
    "bp_kpd_tap_copy": (C.c_int, [vp, C.c_int, C.c_int, vp, vp]),
    "bp_yolo_set_policy": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "bp_kpd_set_policy": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "bp_yolo_set_precision": (C.c_int, [vp, C.c_int]),
    "bp_kpd_set_precision": (C.c_int, [vp, C.c_int]),
    "bp_calibrate_ticks": (C.c_int, [C.c_longlong, c_float_p, vp]),
    "bp_yolo_set_prefetch": (C.c_int, [vp, C.c_int]),
    "bp_kpd_set_prefetch": (C.c_int, [vp, C.c_int]),
    "bp_yolo_set_fusion": (C.c_int, [vp, C.c_int]),
    "bp_kpd_set_fusion": (C.c_int, [vp, C.c_int]),
    "bp_yolo_fused_launches": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int)]),
    "bp_kpd_fused_launches": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int)]),
    "bp_yolo_xcd_errors": (C.c_int, [vp, C.POINTER(C.c_int), vp]),
    "bp_kpd_xcd_errors": (C.c_int, [vp, C.POINTER(C.c_int), vp]),
    "bp_yolo_set_stamps": (C.c_int, [vp, vp, C.c_int]),
    "bp_kpd_set_stamps": (C.c_int, [vp, vp, C.c_int]),
    "bp_yolo_op_name": (C.c_int, [vp, C.c_int, C.c_char_p, C.c_int]),
    "bp_kpd_op_name": (C.c_int, [vp, C.c_int, C.c_char_p, C.c_int]),
    "bp_yolo_op_stats": (C.c_int, [vp, c_double_p, c_double_p, C.c_int]),
    "bp_kpd_op_stats": (C.c_int, [vp, c_double_p, c_double_p, C.c_int]),
    "bp_yolo_device_bytes": (C.c_size_t, [vp]),
    "bp_kpd_device_bytes": (C.c_size_t, [vp]),
    "bp_crop": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int, vp]),
    "bp_resize_bicubic": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "bp_conv2d": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                            C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, c_float_p, vp]),
    "bp_conv2d_planes": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, c_float_p, vp]),
    "bp_pipeline_create": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, vp, vp, vp, C.POINTER(vp)]),
    "bp_pipeline_kernel_count": (C.c_int, [vp]),
    "bp_yolo_profile": (C.c_int, [vp, C.c_int, C.c_int, c_float_p, c_int_p, C.c_int, vp]),
    "bp_kpd_profile": (C.c_int, [vp, C.c_int, C.c_int, c_float_p, c_int_p, C.c_int, vp]),
    "bp_pipeline_destroy": (None, [vp]),
    "bp_pipeline_frames": (vp, [vp]),
    "bp_pipeline_results": (vp, [vp]),
    "bp_pipeline_heatmaps": (vp, [vp]),
    "bp_pipeline_set_fixed_box": (C.c_int, [vp, vp]),
    "bp_pipeline_run": (C.c_int, [vp, C.c_int, vp]),
    "bp_pipeline_prepare": (C.c_int, [vp]),
    "bp_pipeline_latency_faults": (C.c_int, [vp]),
    "bp_heatmap_argmax": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "bp_solve_pnp": (C.c_int, [vp, vp, C.c_int, vp, vp, vp]),
    "bp_solve_pnp_refined": (C.c_int, [vp, vp, C.c_int, vp, vp, vp]),
    "bp_solve_pnp_ransac": (C.c_int, [vp, vp, C.c_int, vp, C.c_double, C.c_int, C.c_double, vp, vp, vp]),
    "bp_pose_nms": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp]),
    "bp_darknet_last_error": (C.c_char_p, []),
    "bp_darknet_create": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(vp)]),
    "bp_darknet_destroy": (None, [vp]),
    "bp_darknet_width": (C.c_int, [vp]),
    "bp_darknet_height": (C.c_int, [vp]),
    "bp_darknet_classes": (C.c_int, [vp]),
    "bp_darknet_detect_rgb": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_float, C.c_float, vp, C.c_int]),
    "bp_darknet_detect_png": (C.c_int, [vp, vp, C.c_size_t, C.c_float, C.c_float, vp, C.c_int]),
    "bp_darknet_detect_image": (C.c_int, [vp, vp, C.c_size_t, C.c_float, C.c_float, vp, C.c_int]),
    "bp_image_decode_rgb": (C.c_int, [vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bp_darknet_detect_file": (C.c_int, [vp, C.c_char_p, C.c_float, C.c_float, vp, C.c_int]),
    "bp_yolo_create_darknet": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "bp_stream_create_masked": (C.c_int, [vp, C.c_int, C.POINTER(vp)]),
    "bp_stream_destroy": (C.c_int, [vp]),
    "bp_probe_placement": (C.c_int, [C.c_int, vp, vp, vp]),
    "bp_png_info": (C.c_int, [vp, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bp_png_decode_bgr": (C.c_int, [vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "bp_loader_create": (C.c_int, [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.POINTER(vp)]),
    "bp_loader_destroy": (None, [vp]),
    "bp_loader_next": (C.c_int, [vp, C.POINTER(C.c_longlong), C.POINTER(vp)]),
    "bp_loader_release": (C.c_int, [vp, C.c_longlong]),
    "bp_upload": (C.c_int, [vp, vp, C.c_size_t, vp]),
}

RESULT_FLOATS = 316


class BetaposeHipError(RuntimeError):
    pass


def lib():
    """Load (building first if needed) the HIP library.  Raises on failure."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        # torch ships its own libamdhip64; load it FIRST so this library binds to the same HIP runtime
        # (loading /opt/rocm's copy first leaves two runtimes in the process and ours then sees no device)
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            from .build import build
            build(verbose=False)
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:
            raise BetaposeHipError("cannot load %s: %s (run `python -m betapose_amd.build`)" % (LIB_PATH, e))
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)   # AttributeError if the .so is stale/incomplete
            fn.restype = res
            fn.argtypes = args
        _lib = L
        return L


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().bp_last_error()
        raise BetaposeHipError(msg.decode() if msg else "libbetapose_hip error %d" % rc)


def require_gpu() -> None:
    """Product entry points call this: no GPU / no HIP extension => hard error."""
    import torch
    if not torch.cuda.is_available():
        raise BetaposeHipError("betapose_amd needs an AMD GPU (gfx950): torch.cuda.is_available() is False; "
                               "there is no CPU fallback on the product path")
    if lib().bp_device_count() <= 0:
        raise BetaposeHipError("libbetapose_hip.so sees no HIP device")


def ptr(t) -> int:
    """Device/host pointer of a contiguous torch tensor or numpy array."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        if not t.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return t.data_ptr()
    return t.ctypes.data


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
