#!/usr/bin/env python3
"""Host-side cost of one frame-graph launch (hipGraphLaunch of ~200 kernel nodes) vs the device time of the frame."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from betapose_amd import synth
from betapose_amd.darknet import Darknet
from betapose_amd.kpd import FastPoseHIP
from betapose_amd.pipeline import FramePipeline
from betapose_amd.weights import fastpose_stream_from_state_dict

det = Darknet("yolo/cfg/yolov3-single.cfg", reso=416).load_stream(synth.synth_yolo_stream(1)).cuda()
pose = FastPoseHIP.from_stream(fastpose_stream_from_state_dict(synth.synth_fastpose_state_dict(2)), n_classes=50).cuda()
S = 4
dets = [det] + [det.clone() for _ in range(S - 1)]
poses = [pose] + [pose.clone() for _ in range(S - 1)]
pipes = [FramePipeline(dets[k], poses[k], 480, 640) for k in range(S)]
streams = [torch.cuda.Stream() for _ in range(S)]
fr = torch.from_numpy(synth.synth_frame(1)[None]).cuda()
for k in range(S):
    pipes[k].frames.copy_(fr)
    for _ in range(3):
        pipes[k].enqueue(streams[k].cuda_stream)
torch.cuda.synchronize()
# 1) host cost when the stream is idle each time (launch, then wait)
t_host = 0.0
for _ in range(50):
    t = time.perf_counter(); pipes[0].enqueue(streams[0].cuda_stream); t_host += time.perf_counter() - t
    torch.cuda.synchronize()
print("graph launch, idle stream: host %.3f ms per launch" % (t_host / 50 * 1e3))
# 2) back-to-back launches round-robin over 4 streams without waiting: host time per launch and total
N = 400
t0 = time.perf_counter(); t_host = 0.0
for i in range(N):
    t = time.perf_counter(); pipes[i % S].enqueue(streams[i % S].cuda_stream); t_host += time.perf_counter() - t
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("4 streams, %d launches: host in enqueue %.3f ms/launch, issue loop %.3f ms/launch, wall %.3f ms/launch (%.0f fps)" % (
    N, t_host / N * 1e3, t_issue / N * 1e3, t_all / N * 1e3, N / t_all))
# 3) the same from 4 host threads, one per stream
import threading
def worker(k, n):
    for _ in range(n):
        pipes[k].enqueue(streams[k].cuda_stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
th = [threading.Thread(target=worker, args=(k, N // S)) for k in range(S)]
[t.start() for t in th]; [t.join() for t in th]
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("4 host threads: issue %.3f ms/launch, wall %.3f ms/launch (%.0f fps)" % (t_issue / N * 1e3, t_all / N * 1e3, N / t_all))
