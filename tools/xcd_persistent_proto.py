import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np, helpers
from betapose_amd import _lib
from betapose_amd.darknet import Darknet
dev = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 64
SK = int(sys.argv[3]) if len(sys.argv) > 3 else 64
net = Darknet("yolo/cfg/yolov3-single.cfg", reso=416, max_batch=1).load_stream(helpers.yolo_stream()).cuda().eval()
nets = [net] + [net.clone() for _ in range(N - 1)]
for n_ in nets:
    n_.set_policy(SK, 4, 8, -1)
xs = [helpers.yolo_input_from_frame(f).to(dev) for f in helpers.frames(4)]
refs = []
for k, n_ in enumerate(nets):
    n_(xs[k % 4])                     # ordinary path: leaves the input in place and computes every activation
    torch.cuda.synchronize()
    refs.append([n_.tap(i).clone() for i in range(len(n_.taps()))])
# poison the head tensors' sources by re-running mega from the inputs: every conv output is recomputed in place
H = (C.c_void_p * N)(*[n_._h for n_ in nets])
L = _lib.lib()
assert hasattr(L, "bp_mega_yolo_convs_stamped"), "needs the experimental library: python -m betapose_amd.build --experimental; BP_LIB=betapose_amd/libbetapose_hip_exp.so"
L.bp_mega_yolo_convs_stamped.restype = C.c_int
L.bp_mega_yolo_convs_stamped.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
ms, err = C.c_float(0), C.c_uint(0)
us = (C.c_float * 128)(); info = (C.c_int * 384)()
_lib.check(L.bp_mega_yolo_convs_stamped(H, N, NB, 20, C.byref(ms), C.byref(err), us, info, 128, _lib.current_stream()))
if "--ops" in sys.argv:
    names = [n for n, c in nets[0].op_names() if c]
    prof_ms, _ = nets[0].profile(batch=1, iters=5)
    conv_ms = [m for m, (n, c) in zip(prof_ms, nets[0].op_names()) if c]
    tot = 0.0
    for i, nm in enumerate(names):
        tot += us[i]
        print("  %-40s type %d items %4d slices %d  %7.1f us in the xcd launch | %6.1f us alone on the chip" % (nm, info[3 * i], info[3 * i + 1], info[3 * i + 2], us[i], conv_ms[i] * 1e3))
    print("  sum %.1f us" % tot)
torch.cuda.synchronize()
print("mega: %d frames, %d blocks/XCD, sk_target %d: %.3f ms per launch -> %.1f frames/s for the YOLO convolutions, err word %#x" % (N, NB, SK, ms.value, N / ms.value * 1e3, err.value))
bad = 0
for k, n_ in enumerate(nets):
    for i in range(len(n_.taps())):
        t = n_.tap(i)
        if not torch.equal(t, refs[k][i]):
            bad += 1
            if bad < 6:
                print("  frame", k, "tap", n_.taps()[i], "max |d|", float((t - refs[k][i]).abs().max()))
print("taps differing from the ordinary path:", bad, "of", N * len(nets[0].taps()))
# the ordinary path alone, for scale: 4 streams
