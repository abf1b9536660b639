"""HIP streams confined to a subset of the chip (``hipExtStreamCreateWithCUMask``).

MI355X = 8 XCDs x 32 CUs.  In the driver's numbering mask bit ``i`` is CU ``i // 8`` of XCD ``i % 8``
(tools/partition_probe.py).  The hardware hands the workgroups of every dispatch round-robin to ALL XCDs, so a queue
cannot be confined to one XCD (a mask that leaves an XCD without CUs is ignored by the driver); what a mask can do
is give a queue its own slice of CUs inside every XCD.  The throughput configuration gives every in-flight frame
such a slice: its kernels never compete with another frame's for a CU."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np

from . import _lib

N_XCD = 8
CUS_PER_XCD = 32
N_CUS = N_XCD * CUS_PER_XCD


def slice_mask_words(cu_lo: int, cu_hi: int) -> np.ndarray:
    """Mask words selecting CUs ``cu_lo .. cu_hi-1`` (per-XCD CU index) of every XCD."""
    if not 0 <= cu_lo < cu_hi <= CUS_PER_XCD:
        raise ValueError("CU slice [%d, %d) out of range" % (cu_lo, cu_hi))
    words = np.zeros(N_CUS // 32, np.uint32)
    for i in range(cu_lo * N_XCD, cu_hi * N_XCD):
        words[i // 32] |= np.uint32(1) << np.uint32(i % 32)
    return words


def partition_cus(groups: int) -> List[Tuple[int, int]]:
    """Split the 32 CUs of every XCD into ``groups`` equal slices."""
    if groups < 1 or CUS_PER_XCD % groups:
        raise ValueError("groups must divide %d" % CUS_PER_XCD)
    per = CUS_PER_XCD // groups
    return [(g * per, (g + 1) * per) for g in range(groups)]


class MaskedStream:
    """A HIP stream restricted to a CU slice of every XCD, usable as a torch stream (``.torch``) and as a raw handle
    (``.handle``).  ``n_cus`` = CUs the stream can use."""

    def __init__(self, cu_lo: int, cu_hi: int, device: int = 0):
        import torch
        _lib.require_gpu()
        self.slice = (int(cu_lo), int(cu_hi))
        self.n_cus = (cu_hi - cu_lo) * N_XCD
        w = slice_mask_words(cu_lo, cu_hi)
        h = C.c_void_p()
        with torch.cuda.device(device):
            _lib.check(_lib.lib().bp_stream_create_masked(w.ctypes.data, len(w), C.byref(h)))
        self.handle = h.value
        self.torch = torch.cuda.ExternalStream(self.handle, device=torch.device("cuda", device))

    def probe(self, blocks: int = 4096):
        """(XCD id, raw HW_ID) of every workgroup of a ``blocks``-wide launch on this stream."""
        x, hw = np.zeros(blocks, np.int32), np.zeros(blocks, np.int32)
        _lib.check(_lib.lib().bp_probe_placement(blocks, x.ctypes.data, hw.ctypes.data, self.handle))
        return x, hw

    def places(self, blocks: int = 4096) -> int:
        """Number of distinct CUs a wide launch on this stream touched."""
        x, hw = self.probe(blocks)
        return len(set(zip(x.tolist(), ((hw >> 13) & 7).tolist(), ((hw >> 8) & 15).tolist())))

    def close(self):
        if self.handle:
            _lib.lib().bp_stream_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
