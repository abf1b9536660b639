"""Parametric pose NMS and result serialisation -- host side, numpy fp32.

Same constants, call signature and result dicts as the reference's ``pose_nms`` /
``write_json`` (3_6Dpose_estimator/pPose_nms.py:13-20,24-122,284-371).  The detector
emits exactly one box per frame (yolo/util.py:181,210-211), so n = 1 is the hot
case; the greedy cluster / merge for any n runs in C++ behind the C-ABI
(``bp_pose_nms``, csrc/host_post.cpp).  Unlike the reference, inputs are not
modified in place (pPose_nms.py:33,78-82,254 side effects).
"""
from __future__ import annotations

import json
import os

import numpy as np

F32 = np.float32
delta1, mu, delta2, gamma = 1, 1.7, 2.65, 22.48
scoreThreds, matchThreds, areaThres, alpha = 0.3, 5, 0, 0.1


def _np(a):
    return a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)


def pose_nms(bboxes, bbox_scores, pose_preds, pose_scores):
    """bboxes [n,4], bbox_scores [n,1], pose_preds [n,K,2], pose_scores [n,K,1] -> list of dicts with
    numpy arrays: 'bbox' [4], 'keypoints' [K,2], 'kp_score' [K,1], 'proposal_score' (1-elem array).
    The greedy cluster / merge runs in ``bp_pose_nms`` (csrc/host_post.cpp, f32) for every n."""
    import ctypes as C
    from . import _lib
    bboxes = np.ascontiguousarray(_np(bboxes), dtype=F32).reshape(-1, 4)
    bbox_scores = np.ascontiguousarray(_np(bbox_scores), dtype=F32).reshape(-1)
    preds = np.ascontiguousarray(_np(pose_preds), dtype=F32)
    n, K = preds.shape[0], preds.shape[1]
    scores = np.ascontiguousarray(_np(pose_scores), dtype=F32).reshape(n, K)
    pick = np.zeros(max(n, 1), np.int32)
    out_pose = np.zeros((max(n, 1), K, 2), F32)
    out_score = np.zeros((max(n, 1), K), F32)
    out_prop = np.zeros(max(n, 1), F32)
    m = _lib.lib().bp_pose_nms(bboxes.ctypes.data, bbox_scores.ctypes.data, preds.ctypes.data, scores.ctypes.data, n, K,
                               pick.ctypes.data, out_pose.ctypes.data, out_score.ctypes.data, out_prop.ctypes.data)
    if m < 0:
        _lib.check(m)
    return [{"bbox": bboxes[0].copy(),                                   # always the first box (pPose_nms.py:116)
             "keypoints": out_pose[j].copy(),
             "kp_score": out_score[j].reshape(K, 1).copy(),
             "proposal_score": out_prop[j:j + 1].copy()} for j in range(m)]


def results_to_json_list(all_results, for_eval=False):
    out = []
    for im_res in all_results:
        im_name, cam_R, cam_t = im_res["imgname"], im_res["cam_R"], im_res["cam_t"]
        for human in im_res["result"]:
            r = {}
            if for_eval:
                r["image_id"] = int(im_name.split('/')[-1].split('.')[0].split('_')[-1])
            else:
                r["image_id"] = im_name.split('/')[-1]
            if len(cam_R) > 0:
                r["cam_R"] = np.array(cam_R).reshape((9, 1))[:, 0].tolist()
                r["cam_t"] = np.array(cam_t).reshape((3, 1))[:, 0].tolist()
            kp, sc = _np(human["keypoints"]), _np(human["kp_score"]).reshape(-1)
            flat = []
            for k in range(sc.shape[0]):
                flat += [float(kp[k, 0]), float(kp[k, 1]), float(sc[k])]
            r["keypoints"] = flat
            r["score"] = float(np.asarray(_np(human["proposal_score"])).reshape(-1)[0])
            out.append(r)
    return out


def write_json(all_results, outputpath, for_eval=False, form=None):
    """Default ('coco'-like list) format of the reference's write_json; the 'cmu'/'open' body-pose
    re-mappings (pPose_nms.py:316-347) index 17/18 human joints and do not apply to 50 object key points."""
    if form is None:   # the reference reads opt.format (pPose_nms.py:287)
        from .opt import opt as _opt
        form = getattr(_opt, "format", None)
    if form in ("cmu", "open"):
        raise NotImplementedError("cmu/open formats are human-pose layouts; not used on the 6D path")
    text = json.dumps(results_to_json_list(all_results, for_eval))
    path = os.path.join(outputpath, "Betapose-results.json")
    with open(path, "w") as f:
        f.write(text)
    print("Results have been written to", path)
    return path
