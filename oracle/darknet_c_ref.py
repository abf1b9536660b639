"""Oracle (test infrastructure, see oracle/__init__.py): the reference's OWN Darknet-C inference path, compiled from
its sources by oracle/Makefile into oracle/_ref/libdarknet_ref.so, driven through ctypes.

Entry points used (3_6Dpose_estimator/train_YOLO/src): ``load_network_custom`` network.c:34, ``network_predict_image``
network.c:652 (-> ``network_predict`` :534 -> ``forward_network`` :193), ``get_network_boxes`` network.c:635,
``free_detections`` :642.  Darknet-C emits detections cell-major / anchor-minor with boxes relative to the image
(0..1) and folds BN with ``sqrt(var)+1e-6``; ``predict_rows`` converts to the Python path's layout
(head -> anchor -> gy -> gx, centre-size pixels) so the two can be compared row by row.
"""
from __future__ import annotations

import ctypes as C
import os
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_ref", "libdarknet_ref.so")

NET_BLOCK = ("[net]\nbatch=1\nsubdivisions=1\nwidth=%d\nheight=%d\nchannels=3\nmomentum=0.9\ndecay=0.0005\n"
             "learning_rate=0.001\nburn_in=1000\nmax_batches=500200\npolicy=steps\nsteps=400000,450000\nscales=.1,.1\n\n")


class _Box(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("w", C.c_float), ("h", C.c_float)]


class _Detection(C.Structure):      # box.h:26-33
    _fields_ = [("bbox", _Box), ("classes", C.c_int), ("prob", C.POINTER(C.c_float)), ("mask", C.POINTER(C.c_float)),
                ("objectness", C.c_float), ("sort_class", C.c_int)]


class _Image(C.Structure):          # image.h
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("c", C.c_int), ("data", C.POINTER(C.c_float))]


def available() -> bool:
    return os.path.exists(LIB)


def load_image_color(path: str) -> np.ndarray:
    """The reference's image loader on its own: ``load_image_color`` (image.c:1877 -> load_image_stb :1820 -> the
    vendored stb_image v2.16): planar RGB float32 [3,h,w] / 255.  Needs no network."""
    L = C.CDLL(LIB)
    L.load_image_color.restype = _Image
    L.load_image_color.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.free_image.argtypes = [_Image]
    im = L.load_image_color(path.encode(), 0, 0)
    arr = np.ctypeslib.as_array(im.data, shape=(im.c, im.h, im.w)).copy()
    L.free_image(im)
    return arr


class DarknetC:
    def __init__(self, cfg_text_without_net: str, weights_path: str, reso: int = 416):
        self.lib = C.CDLL(LIB)
        L = self.lib
        L.load_network_custom.restype = C.c_void_p
        L.load_network_custom.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        L.network_predict_image.restype = C.POINTER(C.c_float)
        L.network_predict_image.argtypes = [C.c_void_p, _Image]
        L.get_network_boxes.restype = C.POINTER(_Detection)
        L.get_network_boxes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int,
                                        C.POINTER(C.c_int), C.c_int]
        L.free_detections.argtypes = [C.POINTER(_Detection), C.c_int]
        self.reso = reso
        self._tmp = tempfile.mkdtemp(prefix="dkref_")
        cfg_path = os.path.join(self._tmp, "net.cfg")
        with open(cfg_path, "w") as f:
            f.write(NET_BLOCK % (reso, reso) + cfg_text_without_net)
        self.net = L.load_network_custom(cfg_path.encode(), weights_path.encode(), 0, 1)
        if not self.net:
            raise RuntimeError("load_network_custom failed")

    def predict_rows(self, img_chw: np.ndarray, grids=(13, 26, 52), n_anchor: int = 3) -> np.ndarray:
        """img_chw: f32 [3,R,R] RGB 0..1 (planar, what ``load_image`` produces).  Returns [sum 3 g^2, 5+C] rows in the
        PYTHON path's order and units (cx, cy, w, h in input pixels, objectness, class probabilities)."""
        img = np.ascontiguousarray(img_chw, dtype=np.float32)
        im = _Image(self.reso, self.reso, 3, img.ctypes.data_as(C.POINTER(C.c_float)))
        self.lib.network_predict_image(self.net, im)
        num = C.c_int(0)
        dets = self.lib.get_network_boxes(self.net, self.reso, self.reso, -1.0, 0.0, None, 1, C.byref(num), 0)
        n = num.value
        classes = dets[0].classes
        out = np.zeros((n, 5 + classes), np.float32)
        for i in range(n):
            d = dets[i]
            out[i, 0:4] = (d.bbox.x * self.reso, d.bbox.y * self.reso, d.bbox.w * self.reso, d.bbox.h * self.reso)
            out[i, 4] = d.objectness
            for c in range(classes):
                # yolo_layer.c:get_yolo_detections stores prob = objectness * class_prob
                out[i, 5 + c] = d.prob[c] / d.objectness if d.objectness > 0 else 0.0
        self.lib.free_detections(dets, n)
        # cell-major/anchor-minor per head -> anchor-major
        rows, off = [], 0
        for g in grids:
            blk = out[off:off + g * g * n_anchor].reshape(g * g, n_anchor, -1).transpose(1, 0, 2).reshape(g * g * n_anchor, -1)
            rows.append(blk)
            off += g * g * n_anchor
        assert off == n
        return np.concatenate(rows)

    def detect(self, img_chw: np.ndarray, thresh: float = 0.2, nms: float = 0.4):
        """The reference's ``Detector::detect`` chain (yolo_v2_class.cpp:239-317) through its own C functions:
        ``network_predict_image`` (resize_image + network_predict, network.c:652-660) -> ``get_network_boxes`` with
        relative boxes -> ``do_nms_sort`` (box.c:331) -> bbox conversion restated from yolo_v2_class.cpp:293-311.
        ``img_chw``: f32 [3,h,w] RGB 0..1 at the IMAGE's size.  Returns [(x, y, w, h, prob, obj_id)]."""
        img = np.ascontiguousarray(img_chw, dtype=np.float32)
        _, h, w = img.shape
        L = self.lib
        L.do_nms_sort.argtypes = [C.POINTER(_Detection), C.c_int, C.c_int, C.c_float]
        L.do_nms_sort.restype = None
        im = _Image(w, h, 3, img.ctypes.data_as(C.POINTER(C.c_float)))
        L.network_predict_image(self.net, im)
        num = C.c_int(0)
        dets = L.get_network_boxes(self.net, w, h, float(thresh), 0.5, None, 1, C.byref(num), 0)
        n = num.value
        out = []
        if n:
            classes = dets[0].classes
            if nms:
                L.do_nms_sort(dets, n, classes, float(nms))
            for i in range(n):
                d = dets[i]
                probs = [d.prob[c] for c in range(classes)]
                obj_id = int(np.argmax(probs))
                prob = probs[obj_id]
                if prob > thresh:
                    b = d.bbox
                    out.append((int(max(0.0, (b.x - b.w / 2.) * w)), int(max(0.0, (b.y - b.h / 2.) * h)),
                                int(np.float32(b.w) * np.float32(w)), int(np.float32(b.h) * np.float32(h)), float(prob), obj_id))
        L.free_detections(dets, n)
        return out

    def load_image(self, path: str) -> np.ndarray:
        """``load_image_color`` (image.c: stb decode -> planar RGB float / 255)."""
        L = self.lib
        L.load_image_color.restype = _Image
        L.load_image_color.argtypes = [C.c_char_p, C.c_int, C.c_int]
        L.free_image.argtypes = [_Image]
        im = L.load_image_color(path.encode(), 0, 0)
        arr = np.ctypeslib.as_array(im.data, shape=(im.c, im.h, im.w)).copy()
        L.free_image(im)
        return arr
