"""Frame input for the fused pipeline -- the counterpart of ``ImageLoader``'s ``cv2.imread`` thread
(dataloader.py:150-179): PNG files are decoded by native worker threads (csrc/frame_io.cpp) straight into
pinned host slots, in list order, ahead of the consumer.  Other formats (the reference also accepts .jpg) go
through a small PIL thread pool with the same interface."""
from __future__ import annotations

import ctypes as C
from typing import Iterator, List, Optional, Tuple

import numpy as np

from . import _lib


def decode_png(data: bytes) -> np.ndarray:
    """PNG bytes -> BGR u8 [h,w,3] exactly as ``cv2.imread`` returns it (host only, no GPU needed)."""
    L = _lib.lib()
    h, w, c = C.c_int(), C.c_int(), C.c_int()
    buf = (C.c_ubyte * len(data)).from_buffer_copy(data)
    _lib.check(L.bp_png_info(buf, len(data), C.byref(h), C.byref(w), C.byref(c)))
    out = np.empty((h.value, w.value, 3), np.uint8)
    _lib.check(L.bp_png_decode_bgr(buf, len(data), out.ctypes.data, out.nbytes, C.byref(h), C.byref(w)))
    return out


def png_size(path: str) -> Tuple[int, int]:
    with open(path, "rb") as f:
        head = f.read(64)
    h, w, c = C.c_int(), C.c_int(), C.c_int()
    buf = (C.c_ubyte * len(head)).from_buffer_copy(head)
    _lib.check(_lib.lib().bp_png_info(buf, len(head), C.byref(h), C.byref(w), C.byref(c)))
    return h.value, w.value


class FrameLoader:
    """Iterate ``(index, frame, slot_address)`` over ``paths`` in order; call ``release(index)`` when the
    frame (or the asynchronous upload reading it) is done with the slot.  ``depth`` slots are decoded ahead."""

    def __init__(self, paths: List[str], height: Optional[int] = None, width: Optional[int] = None, threads: int = 8,
                 depth: int = 16, pinned: bool = True):
        self.paths = list(paths)
        self._h = None
        self._pil = None
        native = len(self.paths) > 0 and all(p.lower().endswith(".png") for p in self.paths)
        if native and (height is None or width is None):
            height, width = png_size(self.paths[0])
        if not native:
            self._init_pil(height, width, threads, depth)
            return
        self.height, self.width = int(height), int(width)
        arr = (C.c_char_p * len(self.paths))(*[p.encode() for p in self.paths])
        h = C.c_void_p()
        _lib.check(_lib.lib().bp_loader_create(arr, len(self.paths), self.height, self.width, int(threads), int(depth),
                                               int(pinned), C.byref(h)))
        self._h = h

    # -- non-PNG inputs: PIL on Python threads (decode releases the GIL), same slot discipline
    def _init_pil(self, height, width, threads, depth):
        from concurrent.futures import ThreadPoolExecutor
        from .img import load_frame_bgr
        self._pil = ThreadPoolExecutor(max_workers=max(1, int(threads)))
        self._depth = max(2, int(depth))
        self._load = load_frame_bgr
        self._futs = {}
        self._held = {}
        self._submitted = 0
        if self.paths and (height is None or width is None):
            height, width = load_frame_bgr(self.paths[0]).shape[:2]
        self.height, self.width = height, width

    def __len__(self):
        return len(self.paths)

    def __iter__(self) -> Iterator[Tuple[int, np.ndarray, int]]:
        if self._pil is not None:
            n = len(self.paths)
            for i in range(n):
                while self._submitted < n and self._submitted < i + self._depth:
                    self._futs[self._submitted] = self._pil.submit(self._load, self.paths[self._submitted])
                    self._submitted += 1
                fr = np.ascontiguousarray(self._futs.pop(i).result())
                if fr.shape[:2] != (self.height, self.width):
                    raise ValueError("%s: frame is %dx%d, expected %dx%d" % (self.paths[i], fr.shape[1], fr.shape[0],
                                                                          self.width, self.height))
                self._held[i] = fr
                yield i, fr, fr.ctypes.data
            return
        L = _lib.lib()
        nbytes = self.height * self.width * 3
        while True:
            idx, p = C.c_longlong(), C.c_void_p()
            rc = L.bp_loader_next(self._h, C.byref(idx), C.byref(p))
            if rc == 1:
                return
            if rc != 0:
                msg = L.bp_last_error()
                L.bp_loader_release(self._h, idx.value)
                raise _lib.BetaposeHipError(msg.decode() if msg else "frame %d failed" % idx.value)
            view = np.ctypeslib.as_array((C.c_ubyte * nbytes).from_address(p.value)).reshape(self.height, self.width, 3)
            yield idx.value, view, p.value

    def release(self, index: int) -> None:
        if self._pil is not None:
            self._held.pop(index, None)
            return
        _lib.check(_lib.lib().bp_loader_release(self._h, int(index)))

    def close(self) -> None:
        if self._h is not None:
            _lib.lib().bp_loader_destroy(self._h)
            self._h = None
        if self._pil is not None:
            self._pil.shutdown(wait=False)
            self._pil = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
