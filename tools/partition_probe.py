#!/usr/bin/env python3
"""Where do workgroups land under a CU mask?  For a few mask shapes prints the histogram of XCD ids and the number of
distinct (XCD, SE, CU) places a 4096-block launch touched."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from betapose_amd import _lib

L = _lib.lib()
_lib.require_gpu()

def run(name, bits):
    words = np.zeros(8, np.uint32)
    for i in bits:
        words[i // 32] |= np.uint32(1) << np.uint32(i % 32)
    h = C.c_void_p()
    _lib.check(L.bp_stream_create_masked(words.ctypes.data, 8, C.byref(h)))
    n = 4096
    x = np.zeros(n, np.int32); hw = np.zeros(n, np.int32)
    _lib.check(L.bp_probe_placement(n, x.ctypes.data, hw.ctypes.data, h))
    cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7
    places = set(zip(x.tolist(), se.tolist(), cu.tolist()))
    print("%-34s XCD hist %s  distinct (xcd,se,cu) %d" % (name, np.bincount(x, minlength=8).tolist(), len(places)))
    L.bp_stream_destroy(h)

run("all 256 bits", range(256))
run("bits 0..31 (contiguous)", range(32))
run("bits 32..63", range(32, 64))
run("bits i%8==0", range(0, 256, 8))
run("bits i%8==3", range(3, 256, 8))
run("bits 0..7", range(8))
run("bit 0 only", [0])
run("bit 1 only", [1])
run("bit 8 only", [8])
run("bits 0..127", range(128))
