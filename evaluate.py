#!/usr/bin/env python3
"""LineMod single-object evaluation harness on the HIP engines -- the counterpart of the reference's
``betapose_evaluate.py`` (3_6Dpose_estimator/betapose_evaluate.py:86-266) with the same flags (betapose_amd/opt.py):

    python evaluate.py --nClasses 50 --indir <frames> --outdir <out> --sp [--profile] [--obj_id N]
    python evaluate.py --synthetic 16 --outdir /tmp/out [--fused]          # no LineMod needed
    python -m torch.distributed.run --nproc-per-node 8 evaluate.py --fused ...   # frames sharded over GPUs

Default = the reference's staged pipeline (ImageLoader -> DetectionLoader -> DetectionProcessor -> KPD main loop ->
DataWriter, threads + queues).  ``--fused`` = one hipGraph per frame (betapose_amd/pipeline.py), frames round-robined
over ranks, records gathered to rank 0.  Both write ``Betapose-results.json`` and, when LineMod ground truth is
available under --sixd_base, print the reference's three numbers (ADD accuracy, 2-D reprojection accuracy, IoU).
"""
from __future__ import annotations

import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from betapose_amd import dist as bpd, metrics, synth  # noqa: E402
from betapose_amd.opt import parse_args  # noqa: E402


def load_sixd_gt(base, obj_id, seq_id=None):
    """Minimal ``load_sixd`` (utils/sixd.py:60-111): camera, model, key-point model, per-frame GT of object ``obj_id``
    in sequence ``seq_id`` (LineMod: the object's own sequence; Occlusion-LineMod: always sequence 02, where every
    frame lists several objects -- occlusion_betapose_evaluate.py:204,218-220)."""
    import yaml
    seq = os.path.join(base, "test", "%02d" % (obj_id if seq_id is None else seq_id))
    gt = yaml.safe_load(open(os.path.join(seq, "gt.yml")))
    info = yaml.safe_load(open(os.path.join(base, "models", "models_info.yml")))
    frames = {}
    for nr, objs in gt.items():
        # LineMod looks at the frame's FIRST annotation only and skips the frame when that is another object
        # (betapose_evaluate.py:219-221); Occlusion-LineMod walks every annotation of the frame
        # (occlusion_betapose_evaluate.py:218-220)
        cand = objs if seq_id is not None else objs[:1]
        ent = []
        for o in cand:
            if o["obj_id"] == obj_id:
                pose = np.eye(4)
                pose[:3, :3] = np.array(o["cam_R_m2c"]).reshape(3, 3)
                pose[:3, 3] = np.array(o["cam_t_m2c"]).reshape(3) / 1000.0
                ent.append({"pose": pose, "bbox": list(o["obj_bb"])})
        frames[int(nr)] = ent
    model = metrics.load_ply_vertices(os.path.join(base, "models", "obj_%02d.ply" % obj_id)) / 1000.0
    kp = metrics.load_ply_vertices(os.path.join(base, "kpmodels", "obj_%02d.ply" % obj_id)) / 1000.0
    # camera of the 2-D reprojection metric: camera.yml when the dataset has one, identity otherwise
    # (utils/sixd.py:53-68 -- ``Benchmark.cam``); PnP always uses the hard-coded LineMod K (betapose_evaluate.py:59)
    cam = np.identity(3)
    if os.path.exists(os.path.join(base, "camera.yml")):
        c = yaml.safe_load(open(os.path.join(base, "camera.yml")))
        cam[0, 0], cam[0, 2], cam[1, 1], cam[1, 2] = c["fx"], c["cx"], c["fy"], c["cy"]
    # the reference indexes a list built in models_info.yml key order (sixd.py:72-76), which equals a lookup by
    # object id for the 1..N keys every SIXD dataset has
    return frames, model, kp, float(info[obj_id]["diameter"]), cam


def main():
    args = parse_args()
    from betapose_amd import _lib
    from betapose_amd.darknet import Darknet
    from betapose_amd.kpd import ALLPATHS, FastPoseHIP
    from betapose_amd.pPose_nms import write_json
    from betapose_amd.weights import fastpose_stream_from_state_dict, load_kpd_pkl, read_darknet_weights

    _lib.require_gpu()
    bpd.limit_host_threads()
    rank, world, local = bpd.init_from_env()
    obj_id = args.obj_id
    # key points handed to PnP: all 50 on LineMod (betapose_evaluate.py:139), the --left_keypoints best on Occlusion
    left_number = args.left_keypoints if args.occlusion else 50
    pixel_thresh = 20.0 if args.occlusion else 5.0          # occlusion_betapose_evaluate.py:255 vs betapose_evaluate.py:257
    print("Betapose begin running now.  Test object", obj_id, "| key points for PnP:", left_number)
    os.makedirs(args.outputpath, exist_ok=True)

    # ---- inputs
    gt_frames, model_vertices, diameter, metric_cam = None, None, None, None
    if args.synthetic:
        from PIL import Image
        args.inputpath = tempfile.mkdtemp(prefix="bp_frames_")
        im_names = []
        if rank == 0 or True:
            for i, fr in enumerate(synth.synth_frames(args.synthetic, 1234)):
                name = "%04d.png" % i
                Image.fromarray(fr[:, :, ::-1].copy()).save(os.path.join(args.inputpath, name))
                im_names.append(name)
        cam_K, kp3d = synth.CAM_K, synth.synth_kp3d(50)
    else:
        if len(args.inputlist):
            im_names = [l.strip() for l in open(args.inputlist)]
        elif len(args.inputpath) and args.inputpath != '/':
            im_names = sorted(f for f in os.listdir(args.inputpath) if f.lower().endswith((".png", ".jpg")))
        else:
            raise IOError('Error: must contain either --indir/--list')
        cam_K = synth.CAM_K
        gt_frames, model_vertices, kp3d, diameter, metric_cam = load_sixd_gt(args.sixd_base, obj_id, 2 if args.occlusion else None)
        kp3d = metrics.refine_keypoints(kp3d, 50) if len(kp3d) > 50 else kp3d

    # ---- weights: rank 0 reads the files, the fp32 streams are broadcast (RCCL)
    ys = ks = None
    if rank == 0:
        if (args.synthetic or args.synth_weights) and not args.yolo_weights:
            sy, sk = synth.object_seeds(obj_id) if not args.synthetic else (1, 2)
            ys = synth.synth_yolo_stream(sy)
            ks = fastpose_stream_from_state_dict(synth.synth_fastpose_state_dict(sk, args.nClasses), args.nClasses)
        else:
            ys = read_darknet_weights(args.yolo_weights or 'models/yolo/{:02d}.weights'.format(obj_id))[2]
            ks = fastpose_stream_from_state_dict(
                load_kpd_pkl(args.kpd_weights or './exp/final_model/' + ALLPATHS[obj_id] + '.pkl'), args.nClasses)
    ys, ks = bpd.broadcast_stream(ys), bpd.broadcast_stream(ks)
    det = Darknet("yolo/cfg/yolov3-single.cfg", reso=int(args.inp_dim), max_batch=max(1, args.detbatch), device=local)
    det.load_stream(ys).cuda()
    pose_model = FastPoseHIP.from_stream(ks, n_classes=args.nClasses, max_batch=max(1, args.detbatch) if args.fused else 1,
                                        device=local).cuda()
    det.set_precision(args.precision)
    pose_model.set_precision(args.precision)

    t0 = time.time()
    if args.fused:
        from betapose_amd.frame_loader import FrameLoader
        from betapose_amd.pipeline import StreamedRunner, finish_record
        mine = bpd.shard_indices(len(im_names), rank, world)
        recs = np.zeros((len(mine), 316), np.float32)

        def keep(j, rec):
            recs[j] = rec
        t_dev = time.time()
        if len(mine):   # a rank beyond the frame count has nothing to load (its share of the gather is empty)
            # decode threads are a per-GPU budget: ranks of one node share the host cores
            threads = max(1, min(args.load_threads, (os.cpu_count() or 8) // max(1, world)))
            loader = FrameLoader([os.path.join(args.inputpath, im_names[i]) for i in mine], threads=threads,
                                 depth=max(16, 2 * args.streams * max(1, args.detbatch) + threads))
            runner = StreamedRunner(det, pose_model, loader.height, loader.width, streams=args.streams,
                                    confidence=args.confidence, num_classes=args.num_classes, batch=args.detbatch)
            runner.run(loader, keep)
            loader.close()
        t_dev = time.time() - t_dev
        print("rank %d: %d frames, files -> records %.1f frames/sec (%d launches of %d frame(s) in flight, %d decode threads)" % (
            rank, len(mine), len(mine) / max(t_dev, 1e-9), args.streams, max(1, args.detbatch), args.load_threads))
        allrec = bpd.gather_records(recs, mine, len(im_names))
        final_result = []
        if rank == 0:
            for i, name in enumerate(im_names):
                out = finish_record(allrec[i], name, kp3d, cam_K, left_number)
                if out["boxes"] is not None:
                    final_result.append(out)
    else:
        assert world == 1, "the staged pipeline is single-GPU; use --fused to shard frames over ranks"
        from betapose_amd.dataloader import DataWriter, DetectionLoader, DetectionProcessor, ImageLoader
        data_loader = ImageLoader(im_names, batchSize=args.detbatch, format='yolo', reso=int(args.inp_dim)).start()
        det_loader = DetectionLoader(data_loader, obj_id, batchSize=args.detbatch, det_model=det).start()
        det_processor = DetectionProcessor(det_loader).start()
        writer = DataWriter(cam_K, left_number, kp3d).start()
        prof = {'dt': [], 'pt': [], 'pn': []}
        for i in range(data_loader.length()):
            t_s = time.time()
            (inps, orig_img, im_name, boxes, scores, pt1, pt2) = det_processor.read()
            if boxes is None or boxes.nelement() == 0:
                writer.save(None, None, None, None, None, orig_img, im_name.split('/')[-1])
                continue
            t_d = time.time()
            hm = pose_model(inps.cuda()).cpu()
            t_p = time.time()
            writer.save(boxes, scores, hm, pt1, pt2, orig_img, im_name.split('/')[-1])
            prof['dt'].append(t_d - t_s); prof['pt'].append(t_p - t_d); prof['pn'].append(time.time() - t_p)
        while writer.running():
            pass
        writer.stop()
        final_result = writer.results()
        if args.profile:
            print('det time: {:.4f} | pose time: {:.4f} | post processing: {:.5f}'.format(
                np.mean(prof['dt']), np.mean(prof['pt']), np.mean(prof['pn'])))
    if rank == 0:
        print('===========================> Finish Model Running: %d frames, %d with a pose, %.2f s' % (
            len(im_names), sum(len(f['result']) > 0 for f in final_result), time.time() - t0))
        write_json(final_result, args.outputpath)
        if gt_frames is not None:
            m = metrics.evaluate_results(final_result, gt_frames, model_vertices, metric_cam, diameter, pixel_thresh)
            print("Mean add accuracy for seq %02d is: %.3f" % (obj_id, m["mean_add"]))
            print("2d reprojection accuracy for seq %02d is: %.3f" % (obj_id, m["mean_2d_acc"]))
            print("Mean IoU for seq %02d is: %.3f" % (obj_id, m["mean_iou"]))
    bpd.finalize()


if __name__ == "__main__":
    main()
