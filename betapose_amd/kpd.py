"""Key-point detector: ``InferenNet_fast`` with the reference's call surface
(3_6Dpose_estimator/KPD/src/main_fast_inference.py:26-46):

    pose_model = InferenNet_fast(4 * 1 + 1, obj_id, pose_dataset)
    pose_model.cuda(); pose_model.eval()
    hm = pose_model(inps)                  # f32[B,3,320,256] -> f32[B,50,80,64]

The network is FastPose = SE-ResNet-101 + PixelShuffle + 2 x DUC + conv_out
(KPD/src/models/FastPose.py:13-35); its ``.pkl`` state dict is flattened by
``weights.fastpose_stream_from_state_dict`` and handed to libbetapose_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np

from . import _lib
from .weights import fastpose_stream_from_state_dict, fastpose_stream_size, load_kpd_pkl

# main_fast_inference.py:29-32
ALLPATHS = ['NULL', 'seq1_model', 'seq2_model', 'NULL', 'seq4_model', 'seq5_model', 'seq6_model', 'NULL',
            'seq8_model', 'seq9_model', 'Semmetry_obj10', 'seq11_model', 'seq12_model', 'seq13_model',
            'seq14_model', 'seq15_model']


class FastPoseHIP:
    """Engine wrapper.  ``state_dict`` values may be numpy arrays or torch tensors."""

    def __init__(self, state_dict: Dict[str, object], n_classes: int = 50, max_batch: int = 1,
                 device: Optional[int] = None):
        self.n_classes = int(n_classes)
        self.max_batch = int(max_batch)
        self._device = device
        self._stream = fastpose_stream_from_state_dict(state_dict, self.n_classes)
        assert self._stream.size == fastpose_stream_size(self.n_classes)
        self._h = None
        self.training = False

    @classmethod
    def from_stream(cls, stream: np.ndarray, n_classes: int = 50, max_batch: int = 1, device=None):
        self = cls.__new__(cls)
        self.n_classes, self.max_batch, self._device = int(n_classes), int(max_batch), device
        self._stream = np.ascontiguousarray(stream, dtype=np.float32)
        self._h = None
        self.training = False
        return self

    def _ensure(self):
        if self._h is not None:
            return
        import torch
        _lib.require_gpu()
        if self._device is None:
            self._device = torch.cuda.current_device()
        h = C.c_void_p()
        _lib.check(_lib.lib().bp_kpd_create(self._stream.ctypes.data, self._stream.size, self.n_classes,
                                            self.max_batch, self._device, C.byref(h)))
        self._h = h

    def _destroy(self):
        if self._h is not None:
            _lib.lib().bp_kpd_destroy(self._h)
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

    def cuda(self, device=None):
        if device is not None:
            self._device = int(device) if not hasattr(device, "index") else device.index
        self._ensure()
        return self

    def eval(self):
        self.training = False
        return self

    def _prep(self, x):
        import torch
        self._ensure()
        if x.dim() != 4 or tuple(x.shape[1:]) != (3, 320, 256):
            raise ValueError("expected [B,3,320,256], got %s" % (tuple(x.shape),))
        if x.shape[0] > self.max_batch:
            raise ValueError("batch %d > max_batch %d" % (x.shape[0], self.max_batch))
        return x.to(device="cuda:%d" % self._device, dtype=torch.float32).contiguous()

    def forward(self, x):
        import torch
        x = self._prep(x)
        nout = min(self.n_classes, 50)
        hm = torch.empty((x.shape[0], nout, 80, 64), device=x.device, dtype=torch.float32)
        _lib.check(_lib.lib().bp_kpd_forward(self._h, x.data_ptr(), x.shape[0], hm.data_ptr(), _lib.current_stream()))
        return hm

    __call__ = forward

    def forward_argmax(self, x, want_hm: bool = False):
        """-> kp f32[B,50,6] = (argmax idx as int bits, max, left, right, up, down) -- the part of
        ``getPrediction`` (KPD/src/utils/eval.py:113-141) that needs the heat-map."""
        import torch
        x = self._prep(x)
        nout = min(self.n_classes, 50)
        kp = torch.empty((x.shape[0], nout, 6), device=x.device, dtype=torch.float32)
        hm = torch.empty((x.shape[0], nout, 80, 64), device=x.device, dtype=torch.float32) if want_hm else None
        _lib.check(_lib.lib().bp_kpd_forward_argmax(self._h, x.data_ptr(), x.shape[0],
                                                    hm.data_ptr() if want_hm else None, kp.data_ptr(),
                                                    _lib.current_stream()))
        return (kp, hm) if want_hm else kp

    def taps(self):
        self._ensure()
        L = _lib.lib()
        out = []
        name = C.create_string_buffer(64)
        c, h, w = C.c_int(), C.c_int(), C.c_int()
        for i in range(L.bp_kpd_tap_count(self._h)):
            _lib.check(L.bp_kpd_tap_info(self._h, i, name, 64, C.byref(c), C.byref(h), C.byref(w)))
            out.append((name.value.decode(), c.value, h.value, w.value))
        return out

    def tap(self, i: int, batch: int = 1):
        import torch
        name, c, h, w = self.taps()[i]
        t = torch.empty((batch, c, h, w), device="cuda:%d" % self._device, dtype=torch.float32)
        _lib.check(_lib.lib().bp_kpd_tap_copy(self._h, i, batch, t.data_ptr(), _lib.current_stream()))
        return t

    def set_policy(self, sk_target_blocks: int = 512, sk_min_chunks: int = 4, sk_max_splits: int = 8,
                   force_tile: int = -1):
        self._ensure()
        _lib.check(_lib.lib().bp_kpd_set_policy(self._h, sk_target_blocks, sk_min_chunks, sk_max_splits, force_tile))

    def set_precision(self, precision: str = "bf16x3"):
        """'f32' (fp32 MFMA), 'bf16x3' (fp32-accurate: exact 3-way bf16 operand split on the bf16 MFMA) or 'f16'
        (fp16 operands, fp32 accumulate: carries fp16 rounding)."""
        self._ensure()
        _lib.check(_lib.lib().bp_kpd_set_precision(self._h, {"f32": 0, "f16": 1, "bf16x3": 2, "f16r": 3}[precision]))
        self._precision = precision
        return self

    def clone(self):
        """Second engine over the same device filters (own activations): one per concurrent stream."""
        import copy
        self._ensure()
        h = C.c_void_p()
        _lib.check(_lib.lib().bp_kpd_clone(self._h, C.byref(h)))   # first: a failed clone must not leave a copy owning self._h
        other = copy.copy(self)
        other._h = h
        return other

    def profile(self, batch: int = 1, iters: int = 10):
        """Eager pass with hipEvent pairs per op -> (ms[n_ops], info[n_ops,4] = is_conv, tile, vec, splits)."""
        self._ensure()
        L = _lib.lib()
        n = L.bp_kpd_profile(self._h, batch, iters, None, None, 0, _lib.current_stream())
        ms = (C.c_float * n)()
        info = (C.c_int * (4 * n))()
        rc = L.bp_kpd_profile(self._h, batch, iters, ms, info, n, _lib.current_stream())
        if rc < 0:
            _lib.check(rc)
        return np.array(ms, dtype=np.float64), np.array(info, dtype=np.int64).reshape(n, 4)

    def set_prefetch(self, on: bool = True):
        """Lone-frame latency mode (include/betapose_hip.h bp_*_set_prefetch), see Darknet.set_prefetch."""
        self._ensure()
        _lib.check(_lib.lib().bp_kpd_set_prefetch(self._h, int(bool(on))))
        self._latency_mode = bool(on)

    def set_fusion(self, on: bool = True):
        """Conv -> conv fusion of whole residual / bottleneck blocks (include/betapose_hip.h bp_*_set_fusion; default on)."""
        self._ensure()
        _lib.check(_lib.lib().bp_kpd_set_fusion(self._h, int(bool(on))))

    def fused_launches(self, batch: int = 1) -> int:
        self._ensure()
        n = C.c_int(0)
        _lib.check(_lib.lib().bp_kpd_fused_launches(self._h, int(batch), C.byref(n)))
        return int(n.value)

    def xcd_errors(self) -> int:
        """Non-zero when a launch of the latency mode found a K slice on the wrong XCD since the last call (include/betapose_hip.h
        bp_*_xcd_errors): its tile was not stored, the frame must be run again with the mode off.  Waits for the current stream."""
        if not getattr(self, "_latency_mode", False) or self._h is None:
            return 0
        n = C.c_int(0)
        _lib.check(_lib.lib().bp_kpd_xcd_errors(self._h, C.byref(n), _lib.current_stream()))
        return int(n.value)

    def set_stamps(self, buf=None, slots: int = 0):
        """In-situ conv timing (include/betapose_hip.h bp_*_set_stamps): ``buf`` a cuda int64 tensor of
        n_convs * slots * 8 elements, or None to switch it off."""
        self._ensure()
        _lib.check(_lib.lib().bp_kpd_set_stamps(self._h, buf.data_ptr() if buf is not None else None, int(slots)))

    def op_names(self):
        """[(layer name, is_convolution)] in op order."""
        self._ensure()
        n = _lib.lib().bp_kpd_op_stats(self._h, None, None, 0)
        name = C.create_string_buffer(96)
        out = []
        for i in range(n):
            is_conv = _lib.lib().bp_kpd_op_name(self._h, i, name, 96)
            out.append((name.value.decode(), bool(is_conv == 1)))
        return out

    def op_stats(self):
        self._ensure()
        n = _lib.lib().bp_kpd_op_stats(self._h, None, None, 0)
        f = (C.c_double * n)()
        b = (C.c_double * n)()
        _lib.lib().bp_kpd_op_stats(self._h, f, b, n)
        return np.array(f), np.array(b)


class InferenNet_fast:
    """Same constructor as the reference: loads ``./exp/final_model/<name>.pkl`` for ``obj_id``."""

    def __init__(self, kernel_size, obj_id, dataset, n_classes: int = 50, max_batch: int = 1,
                 model_dir: str = "./exp/final_model/", state_dict=None):
        path = os.path.join(model_dir, ALLPATHS[obj_id] + ".pkl")
        if state_dict is None:
            print("Loading pose model from {}".format(path))
            state_dict = load_kpd_pkl(path)
        self.pyranet = FastPoseHIP(state_dict, n_classes=n_classes, max_batch=max_batch)
        self.dataset = dataset

    def cuda(self, device=None):
        self.pyranet.cuda(device)
        return self

    def eval(self):
        return self

    def forward(self, x):
        return self.pyranet.forward(x)     # conv_out already narrowed to the first 50 maps in the engine

    __call__ = forward
