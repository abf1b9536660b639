"""Split-K hand-off under load (write-through slabs + ticket counter, conv_igemm.hip): four frames in flight on four
streams, replayed; results must be bit-identical every time (slices are summed in slice order, so any difference
would be a stale or torn slab read)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import helpers  # noqa: E402
from betapose_amd import synth  # noqa: E402
from betapose_amd.darknet import Darknet  # noqa: E402
from betapose_amd.kpd import FastPoseHIP  # noqa: E402
from betapose_amd.pipeline import FramePipeline  # noqa: E402


def test_concurrent_replay_is_bit_reproducible(cuda):
    det = Darknet("yolo/cfg/yolov3-single.cfg").load_stream(helpers.yolo_stream()).cuda()
    pose = FastPoseHIP(helpers.kpd_state_dict()).cuda()
    S = 4
    dets = [det] + [det.clone() for _ in range(S - 1)]
    poses = [pose] + [pose.clone() for _ in range(S - 1)]
    pipes = [FramePipeline(dets[k], poses[k], 480, 640, keep_heatmaps=True) for k in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    frames = [torch.from_numpy(f[None]).to(cuda) for f in helpers.frames(S)]
    ref = []
    for k in range(S):
        pipes[k].frames.copy_(frames[k])
        pipes[k].enqueue()
        torch.cuda.synchronize()
        ref.append((pipes[k].results.clone(), pipes[k].heatmaps.clone()))
    # clones share filters but nothing else: each stream's first pass equals the single-engine golden run
    gold = helpers.golden("pipeline.npz")
    for k in range(S):
        assert int(ref[k][0][0, :1].cpu().view(torch.int32)) == int(gold["f%d_obj_argmax" % k])
    for it in range(300):
        for k in range(S):
            with torch.cuda.stream(streams[k]):
                pipes[k].frames.copy_(frames[k], non_blocking=True)
                pipes[k].enqueue(streams[k].cuda_stream)
        if it % 10 == 9:
            torch.cuda.synchronize()
            for k in range(S):
                assert torch.equal(pipes[k].results, ref[k][0]), (it, k)
                assert torch.equal(pipes[k].heatmaps, ref[k][1]), (it, k)
