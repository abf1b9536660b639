"""Per-op device times of one eager pass (HIP events around every launch): python tools/per_op.py [--batch B] [--precision P] [--no-fusion]
Prints name, tile id (40 = a whole block in one launch, -1 = computed inside another launch), K slices, us, GFLOP, TFLOP/s."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--precision", default="bf16x3")
    ap.add_argument("--no-fusion", action="store_true")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from betapose_amd import cfg as C, synth
    from betapose_amd.darknet import Darknet
    from betapose_amd.kpd import FastPoseHIP
    blocks = C.parse_cfg_text(C.yolov3_single_cfg_text())
    det = Darknet("yolo/cfg/yolov3-single.cfg", reso=416, max_batch=a.batch).load_stream(synth.synth_yolo_stream(1, blocks)).cuda()
    pose = FastPoseHIP(synth.synth_fastpose_state_dict(2), n_classes=50, max_batch=a.batch).cuda()
    for name, net in (("yolo", det), ("kpd", pose)):
        net.set_precision(a.precision)
        if a.no_fusion:
            net.set_fusion(False)
        ms, info = net.profile(a.batch, a.iters)
        flops, _ = net.op_stats()
        names = net.op_names()
        print("== %s: %d ops, %d launches, %.1f us total, fused groups %d" % (name, len(ms), int((info[:, 1] != -1).sum()), ms.sum() * 1e3, net.fused_launches(a.batch)))
        carry = 0.0
        for i in range(len(ms)):
            gf = flops[i] * a.batch / 1e9
            if info[i, 1] == -1:
                carry += gf if info[i, 0] else 0.0
                print("%3d %-34s tile %3d                (inside the next launch)" % (i, names[i][0], info[i, 1]))
                continue
            gf += carry if info[i, 0] else 0.0
            carry = 0.0 if info[i, 0] else carry
            print("%3d %-34s tile %3d splits %2d  %8.1f us  %7.3f GFLOP  %6.1f TFLOP/s" % (
                i, names[i][0], info[i, 1], info[i, 3], ms[i] * 1e3, gf, gf / ms[i] if ms[i] > 0 else 0.0))


if __name__ == "__main__":
    main()
