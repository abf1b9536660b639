"""Flags of the hot path -- same names and defaults as the reference's argparse singleton
(3_6Dpose_estimator/opt.py:1-150): the flags the inference path reads, plus the remaining ones accepted and ignored.  ``opt`` is a module-level
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
    p.add_argument('--obj_ids', default='', type=str,
                   help='occlusion_evaluate.py: comma-separated object ids evaluated in ONE run, e.g. 1,5,6,8,9,10,11,12 -- '
                        'units of work are (frame, object) pairs, every object keeps its weights resident, frames are '
                        'decoded once')
    p.add_argument('--fused', default=False, action='store_true', help='one hipGraph per frame instead of stage threads')
    p.add_argument('--synthetic', type=int, default=0, help='run on N seeded synthetic frames / weights')
    p.add_argument('--synth_weights', default=False, action='store_true',
                   help='seeded synthetic weights with real frames / ground truth (plumbing runs without checkpoints)')
    p.add_argument('--precision', choices=['f32', 'bf16x3', 'f16'], default='bf16x3',
                   help='matrix-core operand precision: bf16x3 (default: fp32-accurate 3-way bf16 split), f32 (fp32 MFMA), f16 (fp16 operands, '
                        'fp32 accumulate); see DESIGN.md 3.1b/c')
    p.add_argument('--streams', type=int, default=4, help='--fused: frames in flight (HIP streams / engine clones)')
    p.add_argument('--load_threads', type=int, default=8, help='--fused: PNG decode threads')
    p.add_argument('--sixd_base', default='/media/data_2/SIXD/hinterstoisser')
    p.add_argument('--yolo_weights', default='')
    p.add_argument('--kpd_weights', default='')
    # the rest of the reference's flag set (training, visualisation, video input: opt.py:9-150) -- accepted with the
    # reference's names, types and defaults so that existing command lines and scripts keep parsing; nothing on the
    # inference path reads them
    for name, default, typ in _COMPAT_FLAGS:
        kw = {"default": default, "help": argparse.SUPPRESS}
        if typ is not None:
            kw["type"] = typ
        p.add_argument(name, **kw)
    p.add_argument('--dist', dest='dist', type=int, default=1, help=argparse.SUPPRESS)
    p.add_argument('--backend', dest='backend', type=str, default='gloo', help=argparse.SUPPRESS)
    p.add_argument('--port', dest='port', default=None, help=argparse.SUPPRESS)
    p.add_argument('--net', dest='demo_net', default='res152', help=argparse.SUPPRESS)
    p.add_argument('--video', dest='video', default="", help=argparse.SUPPRESS)
    p.add_argument('--webcam', dest='webcam', type=str, default='0', help=argparse.SUPPRESS)
    p.add_argument('--vis_fast', dest='vis_fast', default=False, action='store_true', help=argparse.SUPPRESS)
    return p


_COMPAT_FLAGS = [
    ('--expID', 'default', str), ('--dataset', 'coco', str), ('--nThreads', 40, int), ('--debug', False, bool),
    ('--snapshot', 1, int), ('--addDPG', False, bool), ('--netType', 'hgPRM', str), ('--loadModel', None, str),
    ('--Continue', False, bool), ('--nFeats', 256, int), ('--nStack', 4, int), ('--use_pyranet', True, bool),
    ('--LR', 2.5e-4, float), ('--momentum', 0, float), ('--weightDecay', 0, float), ('--crit', 'MSE', str),
    ('--optMethod', 'rmsprop', str), ('--nEpochs', 200, int), ('--epoch', 0, int), ('--trainBatch', 40, int),
    ('--validBatch', 20, int), ('--trainIters', 0, int), ('--valIters', 0, int), ('--init', None, str),
    ('--scale', 0.25, float), ('--rotate', 30, float), ('--hmGauss', 1, int), ('--baseWidth', 9, int),
    ('--cardinality', 5, int), ('--nResidual', 1, int),
]


opt = build_parser().parse_args([])
opt.num_classes = 80            # opt.py:150


def parse_args(argv=None):
    global opt
    ns = build_parser().parse_args(argv)
    ns.num_classes = 80
    opt.__dict__.update(ns.__dict__)
    return opt
