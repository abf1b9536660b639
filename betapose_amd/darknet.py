"""``Darknet`` -- the detector, same call surface as the reference's
``yolo/darknet.py:Darknet`` (3_6Dpose_estimator/yolo/darknet.py:209-432) so that
``DetectionLoader`` (dataloader.py:285-301) runs with only its import changed:

    det_model = Darknet("yolo/cfg/yolov3-single.cfg", reso=416)
    det_model.load_weights("models/yolo/01.weights")
    det_model.net_info['height'] = 416
    det_model.cuda(); det_model.eval()
    prediction = det_model(img)            # f32[B,3,R,R] -> f32[B, sum 3g^2, 5+C]

All arithmetic runs in libbetapose_hip.so (hand-written HIP for gfx950); this class
only parses the cfg, reads the ``.weights`` file and moves pointers.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import _lib
from .cfg import parse_cfg, parse_cfg_text, yolov3_single_cfg_text
from .weights import darknet_stream_size, read_darknet_weights


def _cfg_text(blocks) -> str:
    out = []
    for b in blocks:
        out.append("[%s]" % b["type"])
        for k, v in b.items():
            if k != "type":
                out.append("%s=%s" % (k, v))
        out.append("")
    return "\n".join(out)


class Darknet:
    def __init__(self, cfgfile: str, reso: int = 416, max_batch: int = 1, device: Optional[int] = None):
        self.blocks = parse_cfg(cfgfile)
        self.reso = int(reso)
        self.net_info = self.blocks[0]          # aliases block 0, as in the reference (darknet.py:232)
        self.max_batch = int(max_batch)
        self._device = device
        self._stream: Optional[np.ndarray] = None
        self._h = None
        self.header = None
        self.seen = 0
        self.training = False

    # ---- nn.Module-like surface used by the reference callers
    def load_weights(self, path: str, cutoff=None):
        if cutoff is not None:
            raise NotImplementedError("cutoff is not used on the inference path")
        self.header, self.seen, flat = read_darknet_weights(path)
        self.load_stream(flat)
        return self

    def load_stream(self, flat: np.ndarray):
        need = darknet_stream_size([b for b in self.blocks if b["type"] != "net"])
        if flat.size < need:
            raise ValueError("weights stream has %d floats, cfg needs %d" % (flat.size, need))
        self._stream = np.ascontiguousarray(flat[:need], dtype=np.float32)
        self._destroy()
        return self

    def eval(self):
        self.training = False
        return self

    def cuda(self, device=None):
        if device is not None:
            self._device = int(device) if not hasattr(device, "index") else device.index
        self._ensure()
        return self

    def to(self, device):
        return self.cuda(device)

    def __call__(self, x, y_true=None):
        return self.forward(x)

    # ---- engine
    def _ensure(self):
        if self._h is not None:
            return
        import torch
        _lib.require_gpu()
        if self._stream is None:
            raise RuntimeError("call load_weights() before the first forward")
        if self._device is None:
            self._device = torch.cuda.current_device()
        h = C.c_void_p()
        blocks = [b for b in self.blocks if b["type"] != "net"]
        _lib.check(_lib.lib().bp_yolo_create_from_memory(
            _cfg_text(blocks).encode(), self._stream.ctypes.data, self._stream.size, self.reso, self.max_batch,
            self._device, C.byref(h)))
        self._h = h
        self.rows = _lib.lib().bp_yolo_rows(h)
        self.attrs = _lib.lib().bp_yolo_attrs(h)

    def _destroy(self):
        if self._h is not None:
            _lib.lib().bp_yolo_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    @property
    def handle(self):
        self._ensure()
        return self._h

    def _prep(self, x):
        import torch
        self._ensure()
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != self.reso or x.shape[3] != self.reso:
            raise ValueError("expected [B,3,%d,%d], got %s" % (self.reso, self.reso, tuple(x.shape)))
        if x.shape[0] > self.max_batch:
            raise ValueError("batch %d > max_batch %d" % (x.shape[0], self.max_batch))
        return x.to(device="cuda:%d" % self._device, dtype=torch.float32).contiguous()

    def forward(self, x):
        """f32[B,3,R,R] (RGB 0..1) -> f32[B, rows, 5+C] in DetectionLayer row order."""
        import torch
        x = self._prep(x)
        pred = torch.empty((x.shape[0], self.rows, self.attrs), device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().bp_yolo_forward(self._h, x.data_ptr(), x.shape[0], pred.data_ptr(), _lib.current_stream()))
        return pred

    def forward_select(self, x, confidence: float = 0.01, num_classes: int = 80, want_pred: bool = False):
        """Fused forward + ``dynamic_write_results`` (yolo/util.py:104-223, NMS off):
        returns ``sel`` f32[B,8] = (idx as int bits, x1,y1,x2,y2,obj,cls_conf,cls_idx); idx=-1 -> no detection."""
        import torch
        x = self._prep(x)
        sel = torch.empty((x.shape[0], 8), device=x.device, dtype=torch.float32)
        pred = torch.empty((x.shape[0], self.rows, self.attrs), device=x.device, dtype=torch.float32) if want_pred else None
        _lib.check(_lib.lib().bp_yolo_forward_select(self._h, x.data_ptr(), x.shape[0], float(confidence), int(num_classes),
                                                     pred.data_ptr() if want_pred else None, sel.data_ptr(),
                                                     _lib.current_stream()))
        return (sel, pred) if want_pred else sel

    # ---- inspection hooks (tests)
    def taps(self):
        self._ensure()
        L = _lib.lib()
        out = []
        name = C.create_string_buffer(64)
        c, h, w = C.c_int(), C.c_int(), C.c_int()
        for i in range(L.bp_yolo_tap_count(self._h)):
            _lib.check(L.bp_yolo_tap_info(self._h, i, name, 64, C.byref(c), C.byref(h), C.byref(w)))
            out.append((name.value.decode(), c.value, h.value, w.value))
        return out

    def tap(self, i: int, batch: int = 1):
        import torch
        name, c, h, w = self.taps()[i]
        t = torch.empty((batch, c, h, w), device="cuda:%d" % self._device, dtype=torch.float32)
        _lib.check(_lib.lib().bp_yolo_tap_copy(self._h, i, batch, t.data_ptr(), _lib.current_stream()))
        return t

    def set_policy(self, sk_target_blocks: int = 512, sk_min_chunks: int = 4, sk_max_splits: int = 8,
                   force_tile: int = -1):
        self._ensure()
        _lib.check(_lib.lib().bp_yolo_set_policy(self._h, sk_target_blocks, sk_min_chunks, sk_max_splits, force_tile))

    def set_precision(self, precision: str = "bf16x3"):
        """'f32' (fp32 MFMA), 'bf16x3' (fp32-accurate: exact 3-way bf16 operand split on the bf16 MFMA) or 'f16'
        (fp16 operands, fp32 accumulate: carries fp16 rounding); 'f16r' = 'f16' with fp16 skip connections (residuals read from the fp16
        operand planes, fp32 copies of tensors that only convolutions and residual adds read are dropped)."""
        self._ensure()
        _lib.check(_lib.lib().bp_yolo_set_precision(self._h, {"f32": 0, "f16": 1, "bf16x3": 2, "f16r": 3}[precision]))
        self._precision = precision
        return self

    def clone(self):
        """Second engine over the same device filters (own activations): one per concurrent stream."""
        import copy
        self._ensure()
        h = C.c_void_p()
        _lib.check(_lib.lib().bp_yolo_clone(self._h, C.byref(h)))   # first: a failed clone must not leave a copy owning self._h
        other = copy.copy(self)
        other._h = h
        return other

    def profile(self, batch: int = 1, iters: int = 10):
        """Eager pass with hipEvent pairs per op -> (ms[n_ops], info[n_ops,4] = is_conv, tile, vec, splits)."""
        self._ensure()
        L = _lib.lib()
        n = L.bp_yolo_profile(self._h, batch, iters, None, None, 0, _lib.current_stream())
        ms = (C.c_float * n)()
        info = (C.c_int * (4 * n))()
        rc = L.bp_yolo_profile(self._h, batch, iters, ms, info, n, _lib.current_stream())
        if rc < 0:
            _lib.check(rc)
        return np.array(ms, dtype=np.float64), np.array(info, dtype=np.int64).reshape(n, 4)

    def set_prefetch(self, on: bool = True):
        """Lone-frame latency mode (include/betapose_hip.h bp_*_set_prefetch): split-K hand-off inside one XCD's L2 +
        prefetch of the next layer's filters.  Bit-identical results; pays with one frame at a time, costs with several in flight."""
        self._ensure()
        _lib.check(_lib.lib().bp_yolo_set_prefetch(self._h, int(bool(on))))
        self._latency_mode = bool(on)

    def set_fusion(self, on: bool = True):
        """Conv -> conv fusion of whole residual / bottleneck blocks (include/betapose_hip.h bp_*_set_fusion; default on)."""
        self._ensure()
        _lib.check(_lib.lib().bp_yolo_set_fusion(self._h, int(bool(on))))

    def fused_launches(self, batch: int = 1) -> int:
        self._ensure()
        n = C.c_int(0)
        _lib.check(_lib.lib().bp_yolo_fused_launches(self._h, int(batch), C.byref(n)))
        return int(n.value)

    def xcd_errors(self) -> int:
        """Non-zero when a launch of the latency mode found a K slice on the wrong XCD since the last call (include/betapose_hip.h
        bp_*_xcd_errors): its tile was not stored, the frame must be run again with the mode off.  Waits for the current stream."""
        if not getattr(self, "_latency_mode", False) or self._h is None:
            return 0
        n = C.c_int(0)
        _lib.check(_lib.lib().bp_yolo_xcd_errors(self._h, C.byref(n), _lib.current_stream()))
        return int(n.value)

    def set_stamps(self, buf=None, slots: int = 0):
        """In-situ conv timing (include/betapose_hip.h bp_*_set_stamps): ``buf`` a cuda int64 tensor of
        n_convs * slots * 8 elements, or None to switch it off."""
        self._ensure()
        _lib.check(_lib.lib().bp_yolo_set_stamps(self._h, buf.data_ptr() if buf is not None else None, int(slots)))

    def op_names(self):
        """[(layer name, is_convolution)] in op order."""
        self._ensure()
        n = _lib.lib().bp_yolo_op_stats(self._h, None, None, 0)
        name = C.create_string_buffer(96)
        out = []
        for i in range(n):
            is_conv = _lib.lib().bp_yolo_op_name(self._h, i, name, 96)
            out.append((name.value.decode(), bool(is_conv == 1)))
        return out

    def op_stats(self):
        self._ensure()
        n = _lib.lib().bp_yolo_op_stats(self._h, None, None, 0)
        f = (C.c_double * n)()
        b = (C.c_double * n)()
        _lib.lib().bp_yolo_op_stats(self._h, f, b, n)
        return np.array(f), np.array(b)


def sel_to_dets(sel) -> "object":
    """sel f32[B,8] (device or cpu) -> what ``dynamic_write_results`` returns:
    int ``0`` when no image has a detection, else f32[n,8] rows
    (batch_idx, x1,y1,x2,y2, obj, cls_conf, cls_idx)  (yolo/util.py:206-222)."""
    import torch
    s = sel.detach().cpu()
    idx = s[:, 0].contiguous().view(torch.int32)
    keep = idx >= 0
    if not bool(keep.any()):
        return 0
    rows = []
    for b in torch.nonzero(keep).flatten().tolist():
        rows.append(torch.cat((torch.tensor([float(b)]), s[b, 1:8])))
    return torch.stack(rows)
