"""Where a fused block's launch spends its time: per-block s_memrealtime marks of the conv_fused.hip launches of one eager pass
(entry | first 1x1 done | halo parked | 3x3 done | trailing 1x1 done | stores done), mean over the blocks, in us since the block's entry,
and the launch's span (first entry -> last store).   python tools/fused_stamps.py"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from betapose_amd import _lib, cfg as C, synth
    from betapose_amd.darknet import Darknet
    from betapose_amd.kpd import FastPoseHIP
    blocks = C.parse_cfg_text(C.yolov3_single_cfg_text())
    det = Darknet("yolo/cfg/yolov3-single.cfg", reso=416, max_batch=1).load_stream(synth.synth_yolo_stream(1, blocks)).cuda()
    pose = FastPoseHIP(synth.synth_fastpose_state_dict(2), n_classes=50, max_batch=1).cuda()
    ms = ctypes.c_float(0)
    TICKS = 2_000_000
    _lib.check(_lib.lib().bp_calibrate_ticks(TICKS, ctypes.byref(ms), _lib.current_stream()))
    per_us = TICKS / (ms.value * 1e3)          # ticks per microsecond
    SLOTS = 4096
    for name, net, x in (("yolo", det, torch.rand(1, 3, 416, 416).cuda()), ("kpd", pose, (torch.rand(1, 3, 320, 256) - 0.45).cuda())):
        names = net.op_names()
        convs = [n for n, c in names if c]
        buf = torch.zeros(len(convs) * SLOTS * 8, dtype=torch.int64, device="cuda")
        net.set_stamps(buf, SLOTS)
        for _ in range(3):
            buf.zero_()
            net(x)
            torch.cuda.synchronize()
        a = buf.cpu().numpy().reshape(len(convs), SLOTS, 8)
        info = net.profile(1, 1)[1]
        conv_idx = [i for i, (_, c) in enumerate(names) if c]
        for c, nm in enumerate(convs):
            if info[conv_idx[c], 1] != 40:
                continue
            blk = a[c][a[c, :, 0] != 0]
            e = blk[:, 0:1]
            rel = lambda k: float(((blk[:, k] - blk[:, 0])[blk[:, k] != 0]).mean() / per_us) if (blk[:, k] != 0).any() else float("nan")
            span = (blk[:, 4].max() - blk[:, 0].min()) / per_us
            late = (blk[:, 0].max() - blk[:, 0].min()) / per_us
            print("%-44s blocks %4d  span %6.2f us (last block enters at %5.2f) | 1x1 done %5.2f | halo parked %5.2f | 3x3 done %5.2f | "
                  "last 1x1 done %5.2f | stores done %5.2f" % (nm, len(blk), span, late, rel(1), rel(2), rel(3), rel(5), rel(4)))
        net.set_stamps(None, 0)


if __name__ == "__main__":
    main()
