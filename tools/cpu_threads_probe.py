import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from betapose_amd import cfg as C, synth, weights as W
from oracle import yolo_ref, kpd_ref
blocks = C.parse_cfg_text(C.yolov3_single_cfg_text())
convs = W.split_darknet_stream(blocks, synth.synth_yolo_stream(1, blocks))
sd = {k: torch.from_numpy(v) for k, v in synth.synth_fastpose_state_dict(2).items()}
x = torch.rand(1, 3, 416, 416); xi = torch.rand(1, 3, 320, 256)
for nt in (8, 16, 32, 64):
    torch.set_num_threads(nt)
    t0 = time.time(); yolo_ref.darknet_forward(blocks, convs, x); kpd_ref.fastpose_forward(sd, xi); t1 = time.time()
    yolo_ref.darknet_forward(blocks, convs, x); kpd_ref.fastpose_forward(sd, xi); t2 = time.time()
    print("threads", nt, "first %.2f s  second %.2f s" % (t1 - t0, t2 - t1), flush=True)
