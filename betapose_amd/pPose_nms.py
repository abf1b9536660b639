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


# The 'cmu' / 'open' layouts of the reference's write_json (pPose_nms.py:316-347): per image a dict with a version string and
# a list of bodies / people, each a flat (x, y, score) list of 18 joints picked from the result's key-point list after
# three derived values have been appended to it.  The picks are positions in the FLAT list (3 per key point): written for
# 17 COCO joints (position 51 = the first appended value = a neck), they run unchanged on whatever key points there are.
_BODY_LAYOUTS = {"cmu": ("Betapose v1.0", "bodies", "joints"), "open": ("Betapose v0.2", "people", "pose_keypoints_2d")}
_BODY_PICKS = (0, 51, 18, 24, 30, 15, 21, 27, 36, 42, 48, 33, 39, 45, 6, 3, 12, 9)
_BODY_MEANS = ((15, 18), (16, 19), (50, 20))       # appended values: means of these flat positions, in this order


def body_layout_results(all_results, form, for_eval=False):
    """{image_id: {"version": ..., "bodies" | "people": [{"joints" | "pose_keypoints_2d": [54 floats]}, ...]}}."""
    version, list_key, joints_key = _BODY_LAYOUTS[form]
    per_image = {}
    for r in results_to_json_list(all_results, for_eval):
        flat = list(r["keypoints"])
        for a, b in _BODY_MEANS:          # each mean is appended before the next is taken, as the reference does
            flat.append((flat[a] + flat[b]) / 2)
        joints = [flat[i + d] for i in _BODY_PICKS for d in (0, 1, 2)]
        entry = per_image.setdefault(r["image_id"], {"version": version, list_key: []})
        entry[list_key].append({joints_key: joints})
    return per_image


def write_json(all_results, outputpath, for_eval=False, form=None):
    """pPose_nms.py:284-371: ``Betapose-results.json`` in the default list layout, or -- opt.format 'cmu' / 'open' -- the
    per-image body layouts plus one file per image under ``sep-json/``."""
    if form is None:   # the reference reads opt.format (pPose_nms.py:287)
        from .opt import opt as _opt
        form = getattr(_opt, "format", None)
    path = os.path.join(outputpath, "Betapose-results.json")
    if form in _BODY_LAYOUTS:
        if for_eval:   # the reference fails here too (pPose_nms.py:358: int image ids have no .split) -- same outcome, said clearly
            raise AttributeError("write_json: the 'cmu' / 'open' layouts name their per-image files after the image name; "
                                 "for_eval=True turns it into an int (the reference raises at pPose_nms.py:358)")
        per_image = body_layout_results(all_results, form, for_eval)
        with open(path, "w") as f:
            f.write(json.dumps(per_image))
        sep = os.path.join(outputpath, "sep-json")
        os.makedirs(sep, exist_ok=True)
        for name, entry in per_image.items():
            with open(os.path.join(sep, str(name).split('.')[0] + ".json"), "w") as f:
                f.write(json.dumps(entry))
    else:
        with open(path, "w") as f:
            f.write(json.dumps(results_to_json_list(all_results, for_eval)))
    print("Results have been written to", path)
    return path
