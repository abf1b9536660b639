#!/usr/bin/env python3
"""Occlusion-LineMod harness -- counterpart of the reference's ``occlusion_betapose_evaluate.py`` (run as in
``occlusion.sh``: ``--nClasses 50 --indir ... --outdir ... --sp --profile --conf 0.9 --obj_id N``).

    python occlusion_evaluate.py --indir <seq02/rgb> --outdir <out> --obj_id 5            # one object, as the reference
    python occlusion_evaluate.py --indir <seq02/rgb> --outdir <out> --obj_ids 1,5,6,8,9,10,11,12
    python -m torch.distributed.run --nproc-per-node 8 occlusion_evaluate.py --obj_ids ...   # units sharded over GPUs

The reference evaluates ONE object per process: eight runs over the same 1214 frames of sequence 02, each decoding
every frame again and loading one detector + one key-point net (occlusion_betapose_evaluate.py:89-90,131-139).  With
``--obj_ids`` the unit of work is a (frame, object) pair: every object's two weight sets stay resident in
HBM (8 x 1.2 GB of 288 GB), a frame is decoded once and handed to every object's graph, units are sharded
``u % world`` over the ranks (u = frame * n_objects + object), the 316-float records are gathered per unit, and rank 0
prints the reference's three numbers per object -- ADD accuracy, 2-D reprojection accuracy at 20 px with the
``--left_keypoints`` best key points, IoU (occlusion_betapose_evaluate.py:204-260) -- and writes one
``obj_XX/Betapose-results.json`` per object.  Per object the results equal a single-object run of that object.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import evaluate  # noqa: E402


def main():
    if "--occlusion" not in sys.argv:
        sys.argv.append("--occlusion")
    from betapose_amd.opt import parse_args
    args = parse_args()
    if not args.obj_ids:
        return evaluate.main()          # the reference's own protocol: one object per run

    from betapose_amd import _lib, dist as bpd, metrics, synth
    from betapose_amd.darknet import Darknet
    from betapose_amd.frame_loader import FrameLoader
    from betapose_amd.kpd import ALLPATHS, FastPoseHIP
    from betapose_amd.pipeline import MultiObjectRunner, finish_record
    from betapose_amd.pPose_nms import write_json
    from betapose_amd.weights import fastpose_stream_from_state_dict, load_kpd_pkl, read_darknet_weights

    _lib.require_gpu()
    bpd.limit_host_threads()
    rank, world, local = bpd.init_from_env()
    obj_ids = [int(v) for v in args.obj_ids.split(",") if v.strip()]
    assert len(obj_ids) == len(set(obj_ids)) and obj_ids, "--obj_ids: distinct object ids"
    K = len(obj_ids)
    left_number = args.left_keypoints
    os.makedirs(args.outputpath, exist_ok=True)
    if len(args.inputlist):
        im_names = [l.strip() for l in open(args.inputlist)]
    elif len(args.inputpath) and args.inputpath != '/':
        im_names = sorted(f for f in os.listdir(args.inputpath) if f.lower().endswith((".png", ".jpg")))
    else:
        raise IOError('Error: must contain either --indir/--list')
    n_units = len(im_names) * K
    print("Betapose begin running now.  Occlusion objects", obj_ids, "| %d frames -> %d (frame, object) units | "
          "key points for PnP: %d" % (len(im_names), n_units, left_number))

    def owned(u):
        return bpd.owner_of(u, world) == rank
    my_objs = [o for oi, o in enumerate(obj_ids) if any(owned(f * K + oi) for f in range(min(len(im_names), world)))]
    my_frames = [f for f in range(len(im_names)) if any(owned(f * K + oi) for oi in range(K))]

    # ---- ground truth / models per object (rank 0 evaluates)
    gt = {}
    if rank == 0:
        for o in obj_ids:
            frames_gt, model, kp3d, diameter, cam = evaluate.load_sixd_gt(args.sixd_base, o, 2)
            gt[o] = (frames_gt, model, metrics.refine_keypoints(kp3d, 50) if len(kp3d) > 50 else kp3d, diameter, cam)

    # ---- weights: rank 0 reads every object's two streams, all ranks receive them, each builds the engines it needs
    engines = {}
    t0 = time.time()
    for o in obj_ids:
        ys = ks = None
        if rank == 0:
            if args.synth_weights and not args.yolo_weights:
                sy, sk = synth.object_seeds(o)
                ys = synth.synth_yolo_stream(sy)
                ks = fastpose_stream_from_state_dict(synth.synth_fastpose_state_dict(sk, args.nClasses), args.nClasses)
            else:
                ys = read_darknet_weights('models/yolo/{:02d}.weights'.format(o))[2]
                ks = fastpose_stream_from_state_dict(load_kpd_pkl('./exp/final_model/' + ALLPATHS[o] + '.pkl'), args.nClasses)
        ys, ks = bpd.broadcast_stream(ys), bpd.broadcast_stream(ks)
        if o in my_objs:
            det = Darknet("yolo/cfg/yolov3-single.cfg", reso=int(args.inp_dim), max_batch=1, device=local)
            det.load_stream(ys).cuda()
            pose = FastPoseHIP.from_stream(ks, n_classes=args.nClasses, max_batch=1, device=local).cuda()
            det.set_precision(args.precision)
            pose.set_precision(args.precision)
            engines[o] = (det, pose)
        del ys, ks
    print("rank %d: %d object engine pairs resident (%s), %.1f s" % (rank, len(engines), sorted(engines), time.time() - t0))

    # ---- run this rank's units
    recs = {}
    t_dev = time.time()
    if my_frames:
        threads = max(1, min(args.load_threads, (os.cpu_count() or 8) // max(1, world)))
        loader = FrameLoader([os.path.join(args.inputpath, im_names[f]) for f in my_frames], threads=threads,
                             depth=max(16, 2 * args.streams + threads))
        runner = MultiObjectRunner(engines, obj_ids, loader.height, loader.width, streams=args.streams,
                                   confidence=args.confidence, num_classes=args.num_classes)
        runner.run(loader, my_frames, owned, lambda u, rec: recs.__setitem__(u, rec))
        loader.close()
    t_dev = time.time() - t_dev
    mine = sorted(recs)
    print("rank %d: %d units over %d decoded frames, %.1f units/sec (%d in flight)" % (
        rank, len(mine), len(my_frames), len(mine) / max(t_dev, 1e-9), args.streams))
    mine_recs = np.stack([recs[u] for u in mine]) if mine else np.zeros((0, 316), np.float32)
    allrec = bpd.gather_records(mine_recs, mine, n_units)

    if rank == 0:
        for oi, o in enumerate(obj_ids):
            frames_gt, model, kp3d, diameter, cam = gt[o]
            final_result = []
            for f, name in enumerate(im_names):
                out = finish_record(allrec[f * K + oi], name, kp3d, synth.CAM_K, left_number)
                if out["boxes"] is not None:
                    final_result.append(out)
            odir = os.path.join(args.outputpath, "obj_%02d" % o)
            os.makedirs(odir, exist_ok=True)
            write_json(final_result, odir)
            m = metrics.evaluate_results(final_result, frames_gt, model, cam, diameter, 20.0)
            print("Mean add accuracy for seq %02d is: %.3f" % (o, m["mean_add"]))
            print("2d reprojection accuracy with leftkeypoints %d for seq %02d is: %.3f" % (left_number, o, m["mean_2d_acc"]))
            print("Mean IoU for seq %02d is: %.3f" % (o, m["mean_iou"]))
    bpd.finalize()


if __name__ == "__main__":
    main()
