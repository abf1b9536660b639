import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np, helpers
from betapose_amd.darknet import Darknet
from betapose_amd.kpd import FastPoseHIP
dev = "cuda:0"
net = Darknet("yolo/cfg/yolov3-single.cfg", reso=416, max_batch=2).load_stream(helpers.yolo_stream()).cuda().eval()
x = torch.cat([helpers.yolo_input_from_frame(f) for f in helpers.frames(2)])
out = {}
for m in ("f32", "f16", "f16r"):
    net.set_precision(m); out[m] = net(x.to(dev)).cpu()
for m in ("f16", "f16r"):
    d = (out[m] - out["f32"]).abs()
    print("yolo", m, "centre", float(d[..., :2].max()), "wh rel", float((d[..., 2:4] / (0.05 + 2e-2 * out["f32"][..., 2:4].abs())).max()), "prob", float(d[..., 4:].max()),
          "argmax same", [int(out[m][b, :, 4].argmax()) == int(out["f32"][b, :, 4].argmax()) for b in range(2)])
kpd = FastPoseHIP(helpers.kpd_state_dict(), n_classes=50, max_batch=8).cuda().eval()
g = torch.Generator().manual_seed(11)
inps = torch.rand(8, 3, 320, 256, generator=g) - 0.45
hm = {}
for m in ("f32", "f16", "f16r"):
    kpd.set_precision(m); hm[m] = kpd(inps.to(dev)).cpu()
for m in ("f16", "f16r"):
    a = hm[m].reshape(8, 50, -1).argmax(2); a32 = hm["f32"].reshape(8, 50, -1).argmax(2)
    print("kpd", m, "max |d|", float((hm[m] - hm["f32"]).abs().max()), "scale", float(hm["f32"].abs().max()), "flips of 400", int((a != a32).sum()))
