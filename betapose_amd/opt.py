"""Flags of the hot path -- same names and defaults as the reference's argparse singleton
(3_6Dpose_estimator/opt.py:1-150), restricted to what the inference path reads.  ``opt`` is a module-level
namespace like the reference's; ``parse_args`` refreshes it from a command line."""
from __future__ import annotations

import argparse


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Betapose per-frame inference on MI355X")
    p.add_argument('--left_keypoints', default=10, type=int, help='key points kept for PnP on Occlusion-LineMod')
    p.add_argument('--obj_id', default=5, type=int)
    p.add_argument('--sp', default=False, action='store_true', help='single process (threads); the only mode here')
    p.add_argument('--profile', default=False, action='store_true')
    p.add_argument('--nClasses', default=50, type=int)
    p.add_argument('--fast_inference', default=True, type=bool)
    p.add_argument('--inputResH', default=320, type=int)
    p.add_argument('--inputResW', default=256, type=int)
    p.add_argument('--outputResH', default=80, type=int)
    p.add_argument('--outputResW', default=64, type=int)
    p.add_argument('--indir', dest='inputpath', default='')
    p.add_argument('--list', dest='inputlist', default='')
    p.add_argument('--mode', dest='mode', default='normal')
    p.add_argument('--outdir', dest='outputpath', default='examples/res/')
    p.add_argument('--inp_dim', dest='inp_dim', type=str, default='416')
    p.add_argument('--conf', dest='confidence', type=float, default=0.01)
    p.add_argument('--nms', dest='nms_thesh', type=float, default=0.6)
    p.add_argument('--save_img', default=False, action='store_true')
    p.add_argument('--vis', default=False, action='store_true')
    p.add_argument('--format', type=str, default=None)
    p.add_argument('--detbatch', type=int, default=1)
    p.add_argument('--posebatch', type=int, default=80)
    p.add_argument('--save_video', dest='save_video', default=False, action='store_true')
    # additions of this implementation
    p.add_argument('--occlusion', default=False, action='store_true',
                   help='Occlusion-LineMod protocol (occlusion_betapose_evaluate.py): GT sequence 02, every GT object of a '
                        'frame, --left_keypoints for PnP, 20 px reprojection threshold')
    p.add_argument('--fused', default=False, action='store_true', help='one hipGraph per frame instead of stage threads')
    p.add_argument('--synthetic', type=int, default=0, help='run on N seeded synthetic frames / weights')
    p.add_argument('--synth_weights', default=False, action='store_true',
                   help='seeded synthetic weights with real frames / ground truth (plumbing runs without checkpoints)')
    p.add_argument('--precision', choices=['f32', 'f16'], default='f32',
                   help='matrix-core operand precision (f16: fp16 MFMA operands, fp32 accumulate; see DESIGN.md)')
    p.add_argument('--streams', type=int, default=4, help='--fused: frames in flight (HIP streams / engine clones)')
    p.add_argument('--load_threads', type=int, default=8, help='--fused: PNG decode threads')
    p.add_argument('--sixd_base', default='/media/data_2/SIXD/hinterstoisser')
    p.add_argument('--yolo_weights', default='')
    p.add_argument('--kpd_weights', default='')
    return p


opt = build_parser().parse_args([])
opt.num_classes = 80            # opt.py:150


def parse_args(argv=None):
    global opt
    ns = build_parser().parse_args(argv)
    ns.num_classes = 80
    opt.__dict__.update(ns.__dict__)
    return opt
