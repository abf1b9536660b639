"""Python view of the Darknet-API-compatible detector (include/betapose_hip.h ``bp_darknet_*``; the plain
``init / detect_image / detect_mat / dispose`` symbols of yolo_v2_class.hpp live in the same library and are meant for
C/C++/C# callers).  ``DarknetDetector(cfg_with_net_block, weights).detect(image)`` returns what the reference's
``Detector::detect`` does: a list of ``(x, y, w, h, prob, obj_id)`` in image pixels."""
from __future__ import annotations

import ctypes as C
from typing import List, Tuple

import numpy as np

from . import _lib


class BBox(C.Structure):            # yolo_v2_class.hpp:16-22
    _fields_ = [("x", C.c_uint), ("y", C.c_uint), ("w", C.c_uint), ("h", C.c_uint), ("prob", C.c_float),
                ("obj_id", C.c_uint), ("track_id", C.c_uint), ("frames_counter", C.c_uint)]


def _check(rc: int) -> int:
    if rc < 0:
        msg = _lib.lib().bp_darknet_last_error()
        raise _lib.BetaposeHipError(msg.decode() if msg else "darknet-compat error %d" % rc)
    return rc


class DarknetDetector:
    def __init__(self, cfg_path: str, weights_path: str, device: int = 0):
        _lib.require_gpu()
        h = C.c_void_p()
        _check(_lib.lib().bp_darknet_create(cfg_path.encode(), weights_path.encode(), int(device), C.byref(h)))
        self._h = h
        self.width = _lib.lib().bp_darknet_width(h)
        self.height = _lib.lib().bp_darknet_height(h)
        self.classes = _lib.lib().bp_darknet_classes(h)

    def close(self):
        if getattr(self, "_h", None) is not None:
            _lib.lib().bp_darknet_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _list(buf, n) -> List[Tuple[int, int, int, int, float, int]]:
        return [(b.x, b.y, b.w, b.h, float(b.prob), b.obj_id) for b in buf[:min(n, len(buf))]]

    def detect(self, planar_rgb: np.ndarray, thresh: float = 0.2, nms: float = 0.4, cap: int = 1000):
        """``planar_rgb``: float32 [3,h,w] in 0..1 (Darknet's ``image``)."""
        im = np.ascontiguousarray(planar_rgb, dtype=np.float32)
        assert im.ndim == 3 and im.shape[0] == 3
        buf = (BBox * cap)()
        n = _check(_lib.lib().bp_darknet_detect_rgb(self._h, im.ctypes.data, im.shape[2], im.shape[1], float(thresh),
                                                    float(nms), buf, cap))
        return self._list(buf, n)

    def detect_file(self, png_path: str, thresh: float = 0.2, nms: float = 0.4, cap: int = 1000):
        buf = (BBox * cap)()
        n = _check(_lib.lib().bp_darknet_detect_file(self._h, png_path.encode(), float(thresh), float(nms), buf, cap))
        return self._list(buf, n)
