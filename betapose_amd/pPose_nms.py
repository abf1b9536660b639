"""Parametric pose NMS and result serialisation -- host side, numpy fp32.

Same constants, call signature and result dicts as the reference's ``pose_nms`` /
``write_json`` (3_6Dpose_estimator/pPose_nms.py:13-20,24-122,284-371).  The detector
emits exactly one box per frame (yolo/util.py:181,210-211), so n = 1 is the hot
case and has a closed form (the merge of a pose with itself is the identity); the
general greedy cluster/merge path is kept for n > 1.  Unlike the reference, inputs
are not modified in place (pPose_nms.py:33,78-82,254 side effects).
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


def _dist(ref, allp):
    d = ref[None, :, :] - allp
    return np.sqrt((d * d).sum(axis=2, dtype=F32))


def _merge(ref_pose, cluster_preds, cluster_scores, ref_dist):
    dist = _dist(ref_pose, cluster_preds)
    mask = dist <= min(ref_dist, 15)
    masked = cluster_scores * mask[..., None].astype(F32)
    normed = masked / masked.sum(axis=0, dtype=F32)
    final_pose = (cluster_preds * normed).sum(axis=0, dtype=F32)
    final_score = (masked * normed).sum(axis=0, dtype=F32)
    return final_pose, final_score


def pose_nms(bboxes, bbox_scores, pose_preds, pose_scores):
    """bboxes [n,4], bbox_scores [n,1], pose_preds [n,K,2], pose_scores [n,K,1] -> list of dicts with
    numpy arrays: 'bbox' [4], 'keypoints' [K,2], 'kp_score' [K,1], 'proposal_score' (1-elem array)."""
    bboxes = _np(bboxes).astype(F32)
    bbox_scores = _np(bbox_scores).astype(F32).reshape(-1, 1)
    preds = _np(pose_preds).astype(F32)
    scores = _np(pose_scores).astype(F32).copy()
    scores[scores == 0] = F32(1e-5)
    n, K = preds.shape[0], preds.shape[1]
    ref_dists = F32(alpha) * np.maximum(bboxes[:, 2] - bboxes[:, 0], bboxes[:, 3] - bboxes[:, 1])

    if n == 1:
        pick, merge_ids = [0], [np.array([0])]
    else:
        human_scores = scores.mean(axis=1, dtype=F32)[:, 0]
        ids = np.arange(n)
        cur_p, cur_s = preds, scores
        pick, merge_ids = [], []
        while ids.size:
            pid = int(np.argmax(human_scores))
            pick.append(int(ids[pid]))
            ref_dist = float(ref_dists[ids[pid]])
            dist = _dist(cur_p[pid], cur_p)
            m = dist <= 1
            sd = np.where(m, np.tanh(cur_s[pid, :, 0][None, :] / F32(delta1)) * np.tanh(cur_s[:, :, 0] / F32(delta1)), F32(0))
            simi = sd.sum(axis=1, dtype=F32) + F32(mu) * np.exp(-dist / F32(delta2)).sum(axis=1, dtype=F32)
            nmatch = (dist / F32(min(ref_dist, 7)) <= 1).sum(axis=1)
            dele = np.nonzero((simi > gamma) | (nmatch >= matchThreds))[0]
            if dele.size == 0:
                dele = np.array([pid])
            merge_ids.append(ids[dele])
            keep = np.setdiff1d(np.arange(ids.size), dele)
            cur_p, cur_s, ids, human_scores = cur_p[keep], cur_s[keep], ids[keep], human_scores[keep]

    out = []
    for j, pk in enumerate(pick):
        if scores[pk, :, 0].max() < scoreThreds:
            continue
        mid = merge_ids[j]
        if n == 1:
            merge_pose, merge_score = preds[0].copy(), scores[0].copy()
        else:
            merge_pose, merge_score = _merge(preds[pk], preds[mid], scores[mid], float(ref_dists[pk]))
        if merge_score[:K].max() < scoreThreds:
            continue
        w = merge_pose[:, 0].max() - merge_pose[:, 0].min()
        h = merge_pose[:, 1].max() - merge_pose[:, 1].min()
        if 1.5 ** 2 * w * h < areaThres:
            continue
        out.append({
            "bbox": bboxes[0].copy(),                                   # always the first box (pPose_nms.py:116)
            "keypoints": merge_pose - F32(0.3),
            "kp_score": merge_score,
            "proposal_score": merge_score.mean(dtype=F32) + bbox_scores[pk] + F32(1.25) * merge_score.max(),
        })
    return out


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
