"""Oracle (test infrastructure, see oracle/__init__.py): everything around the two
networks -- crop, heat-map decoding, pPose-NMS, PnP cross-check, JSON, metrics.

Restates (torch-CPU / numpy, written for clarity not speed):
  * ``im_to_torch``                      KPD/src/utils/img.py:13-18
  * ``crop_from_dets``                   dataloader.py:794-835
  * ``cropBox``                          KPD/src/utils/img.py:242-262 (+ torchsample SpecialCrop/Pad,
                                         third-party, restated: top-left crop, centred zero pad)
  * ``getPrediction``                    KPD/src/utils/eval.py:113-147
  * ``transformBoxInvert_batch``         KPD/src/utils/img.py:216-239
  * ``pose_nms`` + helpers               pPose_nms.py:24-122,204-281
  * ``write_json`` (default format)      pPose_nms.py:284-371
  * ``add_err`` / ``projection_error_2d`` / ``iou``   utils/metrics.py:10-22,77-127
  * key-point pruning of ``DataWriter.update``        dataloader.py:715-727
"""
from __future__ import annotations

import json
import math
from typing import List

import numpy as np
import torch
import torch.nn.functional as F

# pPose_nms.py:13-20
DELTA1, MU, DELTA2, GAMMA, SCORE_THREDS, MATCH_THREDS, AREA_THRES, ALPHA = 1, 1.7, 2.65, 22.48, 0.3, 5, 0, 0.1


# ----------------------------------------------------------------------------- crop (a7)
def im_to_torch(img_rgb_u8: np.ndarray) -> torch.Tensor:
    t = torch.from_numpy(np.transpose(img_rgb_u8, (2, 0, 1)).copy()).float()
    if t.max() > 1:
        t /= 255
    return t


def crop_box(img: torch.Tensor, ul: torch.Tensor, br: torch.Tensor, resH: int, resW: int) -> torch.Tensor:
    """img [3,H,W] f32; ul/br float tensors (x, y).  torch>=1.x division semantics (true division)."""
    ul = ul.int()
    br = br.int()
    a = br[1] - ul[1]
    b = (br[0] - ul[0]) * resH / resW
    lenH = b if b > a else a            # python max(a, b)
    lenW = lenH * resW / resH
    PH, PW = int(lenH), int(lenW)
    sub = img[:, int(ul[1]):, int(ul[0]):]
    ch, cw = int(br[1] - ul[1]), int(br[0] - ul[0])
    sub = sub[:, 0:ch, 0:cw]            # SpecialCrop(size, 1): top-left
    dh = max(PH - sub.shape[1], 0)
    dw = max(PW - sub.shape[2], 0)
    pad = (int(math.ceil(dw / 2.0)), int(math.floor(dw / 2.0)), int(math.ceil(dh / 2.0)), int(math.floor(dh / 2.0)))
    sub = F.pad(sub, pad, mode="constant", value=0.0)   # Pad(size): centred zeros, never crops
    out = F.interpolate(sub.unsqueeze(0), size=(int(resH), int(resW)), mode="bilinear", align_corners=True)
    return out[0]


def crop_from_dets(img: torch.Tensor, boxes: torch.Tensor, resH: int = 320, resW: int = 256):
    """img [3,H,W] RGB 0..1 (modified in place like the reference); boxes [n,4] frame pixels."""
    imght, imgwidth = img.size(1), img.size(2)
    img[0].add_(-0.406)
    img[1].add_(-0.457)
    img[2].add_(-0.480)
    n = boxes.size(0)
    inps = torch.zeros(n, 3, resH, resW)
    pt1 = torch.zeros(n, 2)
    pt2 = torch.zeros(n, 2)
    for i, box in enumerate(boxes):
        upLeft = torch.Tensor((float(box[0]), float(box[1])))
        bottomRight = torch.Tensor((float(box[2]), float(box[3])))
        ht = bottomRight[1] - upLeft[1]
        width = bottomRight[0] - upLeft[0]
        scaleRate = 0.2 if width > 100 else 0.3
        upLeft[0] = max(0, upLeft[0] - width * scaleRate / 2)
        upLeft[1] = max(0, upLeft[1] - ht * scaleRate / 2)
        bottomRight[0] = max(min(imgwidth - 1, bottomRight[0] + width * scaleRate / 2), upLeft[0] + 5)
        bottomRight[1] = max(min(imght - 1, bottomRight[1] + ht * scaleRate / 2), upLeft[1] + 5)
        inps[i] = crop_box(img, upLeft, bottomRight, resH, resW)
        pt1[i] = upLeft
        pt2[i] = bottomRight
    return inps, pt1, pt2


def crop_from_dets_frame(frame_bgr_u8: np.ndarray, boxes: torch.Tensor, resH: int = 320, resW: int = 256):
    """DetectionProcessor.update (dataloader.py:452-453): BGR->RGB, im_to_torch, crop."""
    inp = im_to_torch(np.ascontiguousarray(frame_bgr_u8[:, :, ::-1]))
    return crop_from_dets(inp, boxes, resH, resW)


# ----------------------------------------------------------------------------- heat-map decoding (a9)
def unletterbox_boxes_ref(dets: torch.Tensor, im_dim_list: torch.Tensor, det_inp_dim: int) -> torch.Tensor:
    """3_6Dpose_estimator/dataloader.py:548-560 (inside VideoDetectionLoader.update), statement for statement on a
    copy: detections (frame index, x1, y1, x2, y2, ...) of the LETTERBOXED detector input back to frame pixels -- remove
    the grey border, divide by the letterbox scale, clamp box by box to the frame."""
    dets = dets.clone()
    im_dim_list = torch.index_select(im_dim_list, 0, dets[:, 0].long())
    scaling_factor = torch.min(det_inp_dim / im_dim_list, 1)[0].view(-1, 1)
    dets[:, [1, 3]] -= (det_inp_dim - scaling_factor * im_dim_list[:, 0].view(-1, 1)) / 2
    dets[:, [2, 4]] -= (det_inp_dim - scaling_factor * im_dim_list[:, 1].view(-1, 1)) / 2
    dets[:, 1:5] /= scaling_factor
    for j in range(dets.shape[0]):
        dets[j, [1, 3]] = torch.clamp(dets[j, [1, 3]], 0.0, float(im_dim_list[j, 0]))
        dets[j, [2, 4]] = torch.clamp(dets[j, [2, 4]], 0.0, float(im_dim_list[j, 1]))
    return dets


def transform_box_invert_batch(pt, ul, br, inpH, inpW, resH, resW):
    center = (br - 1 - ul) / 2
    size = br - ul
    size[:, 0] *= (inpH / inpW)
    lenH, _ = torch.max(size, dim=1)
    lenW = lenH * (inpW / inpH)
    _pt = (pt * lenH[:, None, None]) / resH
    K = pt.shape[1]
    _pt[:, :, 0] = _pt[:, :, 0] - ((lenW[:, None].repeat(1, K) - 1) / 2 - center[:, 0].unsqueeze(-1).repeat(1, K)).clamp(min=0)
    _pt[:, :, 1] = _pt[:, :, 1] - ((lenH[:, None].repeat(1, K) - 1) / 2 - center[:, 1].unsqueeze(-1).repeat(1, K)).clamp(min=0)
    new_point = torch.zeros(pt.size())
    new_point[:, :, 0] = _pt[:, :, 0] + ul[:, 0].unsqueeze(-1).repeat(1, K)
    new_point[:, :, 1] = _pt[:, :, 1] + ul[:, 1].unsqueeze(-1).repeat(1, K)
    return new_point


def get_prediction(hms, pt1, pt2, inpH=320, inpW=256, resH=80, resW=64):
    assert hms.dim() == 4
    n, K, H, Wd = hms.shape
    maxval, idx = torch.max(hms.view(n, K, -1), 2)
    maxval = maxval.view(n, K, 1)
    idx = idx.view(n, K, 1) + 1
    preds = idx.repeat(1, 1, 2).float()
    preds[:, :, 0] = (preds[:, :, 0] - 1) % Wd
    preds[:, :, 1] = torch.floor((preds[:, :, 1] - 1) / Wd)
    preds *= maxval.gt(0).repeat(1, 1, 2).float()
    for i in range(n):
        for j in range(K):
            hm = hms[i][j]
            pX, pY = int(round(float(preds[i][j][0]))), int(round(float(preds[i][j][1])))
            if 0 < pX < resW - 1 and 0 < pY < resH - 1:
                diff = torch.Tensor((hm[pY][pX + 1] - hm[pY][pX - 1], hm[pY + 1][pX] - hm[pY - 1][pX]))
                preds[i][j] += diff.sign() * 0.25
    preds += 0.2
    preds_tf = transform_box_invert_batch(preds, pt1.clone(), pt2.clone(), inpH, inpW, resH, resW)
    return preds, preds_tf, maxval


# ----------------------------------------------------------------------------- pPose-NMS (a10)
def _pairwise_dist(ref, allp):
    return torch.sqrt(torch.sum(torch.pow(ref[None, :] - allp, 2), dim=2))


def _parametric_distance(i, all_preds, keypoint_scores, ref_dist):
    pick_preds = all_preds[i]
    pred_scores = keypoint_scores[i]
    dist = _pairwise_dist(pick_preds, all_preds)
    mask = dist <= 1
    K = all_preds.shape[1]
    score_dists = torch.zeros(all_preds.shape[0], K)
    ks = keypoint_scores.reshape(all_preds.shape[0], K)
    ps = pred_scores.reshape(K, 1).repeat(1, all_preds.shape[0]).transpose(0, 1)
    score_dists[mask] = torch.tanh(ps[mask] / DELTA1) * torch.tanh(ks[mask] / DELTA1)
    point_dist = torch.exp((-1) * dist / DELTA2)
    return torch.sum(score_dists, dim=1) + MU * torch.sum(point_dist, dim=1)


def _pck_match(pick_pred, all_preds, ref_dist):
    dist = _pairwise_dist(pick_pred, all_preds)
    ref_dist = min(ref_dist, 7)
    return torch.sum(dist / ref_dist <= 1, dim=1)


def _merge_fast(ref_pose, cluster_preds, cluster_scores, ref_dist):
    dist = _pairwise_dist(ref_pose, cluster_preds)
    ref_dist = min(ref_dist, 15)
    mask = dist <= ref_dist
    if cluster_preds.dim() == 2:
        cluster_preds = cluster_preds.unsqueeze(0)
        cluster_scores = cluster_scores.unsqueeze(0)
    if mask.dim() == 1:
        mask = mask.unsqueeze(0)
    masked = cluster_scores.mul(mask.float().unsqueeze(-1))
    normed = masked / torch.sum(masked, dim=0)
    final_pose = torch.mul(cluster_preds, normed.repeat(1, 1, 2)).sum(dim=0)
    final_score = torch.mul(masked, normed).sum(dim=0)
    return final_pose, final_score


def pose_nms(bboxes, bbox_scores, pose_preds, pose_scores):
    """bboxes [n,4], bbox_scores [n,1], pose_preds [n,K,2], pose_scores [n,K,1] (torch, not modified)."""
    bboxes, bbox_scores = bboxes.clone(), bbox_scores.clone()
    pose_preds, pose_scores = pose_preds.clone(), pose_scores.clone()
    pose_scores[pose_scores == 0] = 1e-5
    K = pose_preds.shape[1]
    ori_bbox_scores, ori_pose_preds, ori_pose_scores = bbox_scores.clone(), pose_preds.clone(), pose_scores.clone()
    widths = bboxes[:, 2] - bboxes[:, 0]
    heights = bboxes[:, 3] - bboxes[:, 1]
    ref_dists = ALPHA * np.maximum(widths.numpy(), heights.numpy())
    human_scores = pose_scores.mean(dim=1)
    human_ids = np.arange(bboxes.shape[0])
    pick, merge_ids = [], []
    while human_scores.shape[0] != 0:
        pick_id = int(torch.argmax(human_scores))
        pick.append(int(human_ids[pick_id]))
        ref_dist = float(ref_dists[human_ids[pick_id]])
        simi = _parametric_distance(pick_id, pose_preds, pose_scores, ref_dist)
        nmatch = _pck_match(pose_preds[pick_id], pose_preds, ref_dist)
        sel = ((simi > GAMMA) | (nmatch >= MATCH_THREDS)).numpy()
        delete_ids = np.arange(human_scores.shape[0])[sel]
        if delete_ids.shape[0] == 0:
            delete_ids = np.array([pick_id])
        merge_ids.append(human_ids[delete_ids])
        keep = np.setdiff1d(np.arange(human_scores.shape[0]), delete_ids)
        pose_preds, pose_scores = pose_preds[keep], pose_scores[keep]
        human_ids, human_scores, bbox_scores = human_ids[keep], human_scores[keep], bbox_scores[keep]
    results = []
    for j, pk in enumerate(pick):
        if torch.max(ori_pose_scores[pk, :, 0]) < SCORE_THREDS:
            continue
        mid = merge_ids[j]
        merge_pose, merge_score = _merge_fast(ori_pose_preds[pk], ori_pose_preds[mid], ori_pose_scores[mid],
                                              float(ref_dists[pk]))
        if torch.max(merge_score[:K]) < SCORE_THREDS:
            continue
        xmax, xmin = max(merge_pose[:, 0]), min(merge_pose[:, 0])
        ymax, ymin = max(merge_pose[:, 1]), min(merge_pose[:, 1])
        if 1.5 ** 2 * (xmax - xmin) * (ymax - ymin) < AREA_THRES:
            continue
        results.append({"bbox": bboxes[0],
                        "keypoints": merge_pose - 0.3,
                        "kp_score": merge_score,
                        "proposal_score": torch.mean(merge_score) + ori_bbox_scores[pk] + 1.25 * max(merge_score)})
    return results


# ----------------------------------------------------------------------------- key-point pruning + PnP cross-check (a11)
def prune_keypoints(kp_2d, kp_3d, kp_score, keep: int):
    kp_2d, kp_3d, kp_score = np.array(kp_2d), np.array(kp_3d), np.array(kp_score)
    while len(kp_2d) > keep:
        d = int(np.argmin(kp_score, axis=0))
        kp_score = np.delete(kp_score, d)
        kp_2d = np.delete(kp_2d, d, axis=0)
        kp_3d = np.delete(kp_3d, d, axis=0)
    return kp_2d, kp_3d, kp_score


def pnp_least_squares(points_3d, points_2d, K, R0=None, t0=None):
    """Independent reference for the pose solve: scipy Levenberg-Marquardt on the same pixel
    reprojection objective cv2.solvePnP(SOLVEPNP_ITERATIVE) minimises (utils/utils.py:25-29)."""
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation as Rot
    P = np.asarray(points_3d, np.float64)
    U = np.asarray(points_2d, np.float64)[:, :2]
    K = np.asarray(K, np.float64)
    if R0 is None:
        R0, t0 = np.eye(3), np.array([0, 0, 1.0])
    x0 = np.r_[Rot.from_matrix(R0).as_rotvec(), np.asarray(t0).ravel()]

    def res(x):
        Y = P @ Rot.from_rotvec(x[:3]).as_matrix().T + x[3:]
        p = Y @ K.T
        return (p[:, :2] / p[:, 2:] - U).ravel()

    s = least_squares(res, x0, method="lm", xtol=1e-15, ftol=1e-15, gtol=1e-15)
    return Rot.from_rotvec(s.x[:3]).as_matrix(), s.x[3:].reshape(3, 1), float((s.fun ** 2).sum())


def _rodrigues_vec_to_mat(r):
    """cvRodrigues2, vector input, with dR/dr_i (3 matrices) -- the closed form OpenCV differentiates."""
    r = np.asarray(r, np.float64)
    th = float(np.linalg.norm(r))
    I = np.eye(3)

    def skew(v):
        return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])
    if th < 1e-12:
        return I + skew(r), [skew(e) for e in I]
    u = r / th
    c, s_, c1 = np.cos(th), np.sin(th), 1 - np.cos(th)
    uut, ux = np.outer(u, u), skew(u)
    R = c * I + c1 * uut + s_ * ux
    dR = []
    for i in range(3):
        du = (I[i] - u * u[i]) / th
        dR.append(-s_ * u[i] * I + s_ * u[i] * uut + c1 * (np.outer(du, u) + np.outer(u, du)) + c * u[i] * ux + s_ * skew(du))
    return R, dR


def _rodrigues_mat_to_vec(R):
    from scipy.spatial.transform import Rotation as Rot
    U, _, Vt = np.linalg.svd(np.asarray(R, np.float64))
    return Rot.from_matrix(U @ Vt).as_rotvec()


def solve_pnp_iterative_ref(points_3d, points_2d, K):
    """Independent numpy restatement of cv2.solvePnP(flags=SOLVEPNP_ITERATIVE) + cv2.Rodrigues as the reference calls
    it (utils/utils.py:17-41; OpenCV calib3d cvFindExtrinsicCameraParams2, 2.4 ... 4.6): planar / DLT initialisation,
    CvLevMarq on (rvec, tvec) with <= 20 accepted steps and FLT_EPSILON.  Same steps as csrc/host_post.cpp, different
    linear algebra (LAPACK SVD instead of Jacobi sweeps, OpenCV's closed-form dR/dr instead of the SO(3) right
    Jacobian).  OpenCV itself is not available here: "parity unpinned" against a real cv2 build."""
    P = np.asarray(points_3d, np.float64)
    U = np.asarray(points_2d, np.float64)[:, :2]
    K = np.asarray(K, np.float64)
    n = P.shape[0]
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    mn = np.stack([(U[:, 0] - cx) / fx, (U[:, 1] - cy) / fy], 1)
    Mc = P.mean(0)
    MM = (P - Mc).T @ (P - Mc)
    _, W, Vt = np.linalg.svd(MM)
    if W[2] / W[1] < 1e-3:
        Rt = Vt.copy()
        if Rt[2, 0] ** 2 + Rt[2, 1] ** 2 < 1e-10:
            Rt = np.eye(3)
        if np.linalg.det(Rt) < 0:
            Rt = -Rt
        Tt = -Rt @ Mc
        xy = (P @ Rt.T + Tt)[:, :2]

        def norm_pts(q):
            c = q.mean(0)
            d = np.linalg.norm(q - c, axis=1).mean()
            s = np.sqrt(2) / d
            return (q - c) * s, np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1.0]])
        a, T0 = norm_pts(xy)
        b, T1 = norm_pts(mn)
        rows = []
        for (x, y), (u, v) in zip(a, b):
            rows.append([x, y, 1, 0, 0, 0, -u * x, -u * y, -u])
            rows.append([0, 0, 0, x, y, 1, -v * x, -v * y, -v])
        H = np.linalg.svd(np.asarray(rows))[2][-1].reshape(3, 3)
        H = np.linalg.inv(T1) @ H @ T0
        H = H / H[2, 2]
        h1, h2, h3 = H[:, 0], H[:, 1], H[:, 2]
        n1, n2 = np.linalg.norm(h1), np.linalg.norm(h2)
        t = h3 * 2.0 / (n1 + n2)
        h1, h2 = h1 / n1, h2 / n2
        Hm = np.stack([h1, h2, np.cross(h1, h2)], 1)
        Hm = _rodrigues_vec_to_mat(_rodrigues_mat_to_vec(Hm))[0]
        t = Hm @ Tt + t
        R = Hm @ Rt
    else:
        L = np.zeros((2 * n, 12))
        for i in range(n):
            x, y = mn[i]
            X = np.r_[P[i], 1.0]
            L[2 * i, 0:4], L[2 * i, 8:12] = X, -x * X
            L[2 * i + 1, 4:8], L[2 * i + 1, 8:12] = X, -y * X
        v = np.linalg.svd(L.T @ L)[2][-1].reshape(3, 4)
        RR, tt = v[:, :3], v[:, 3]
        if np.linalg.det(RR) < 0:
            RR, tt = -RR, -tt
        sc = np.linalg.norm(RR)
        Uu, _, Vv = np.linalg.svd(RR)
        R = Uu @ Vv
        t = tt * (np.linalg.norm(R) / sc)
    prm = np.r_[_rodrigues_mat_to_vec(R), t]

    def project(prm, want_J):
        R, dR = _rodrigues_vec_to_mat(prm[:3])
        Y = P @ R.T + prm[3:]
        iz = np.where(Y[:, 2] != 0, 1.0 / np.where(Y[:, 2] != 0, Y[:, 2], 1.0), 1.0)
        err = np.stack([fx * Y[:, 0] * iz + cx - U[:, 0], fy * Y[:, 1] * iz + cy - U[:, 1]], 1).ravel()
        if not want_J:
            return err, None
        J = np.zeros((2 * n, 6))
        dproj = np.zeros((n, 2, 3))
        dproj[:, 0, 0], dproj[:, 0, 2] = fx * iz, -fx * Y[:, 0] * iz * iz
        dproj[:, 1, 1], dproj[:, 1, 2] = fy * iz, -fy * Y[:, 1] * iz * iz
        for i in range(3):
            dY = P @ dR[i].T
            J[:, i] = np.einsum("nij,nj->ni", dproj, dY).ravel()
            J[:, 3 + i] = dproj[:, :, i].ravel()
        return err, J

    lam10, iters, prev_err = -3, 0, 0.0
    while True:
        err, J = project(prm, True)
        JtJ, JtE = J.T @ J, J.T @ err
        prev = prm.copy()
        if iters == 0:
            prev_err = np.linalg.norm(err)

        def step():
            A = JtJ.copy()
            A[np.diag_indices(6)] *= 1.0 + 10.0 ** lam10
            return prev - np.linalg.solve(A, JtE)
        prm = step()
        while True:
            err_n = np.linalg.norm(project(prm, False)[0])
            if err_n > prev_err:
                lam10 += 1
                if lam10 <= 16:
                    prm = step()
                    continue
            break
        lam10 = max(lam10 - 1, -16)
        iters += 1
        if iters >= 20 or np.linalg.norm(prm - prev) / max(np.linalg.norm(prev), 1e-300) < float(np.finfo(np.float32).eps):
            break
        prev_err = err_n
    return _rodrigues_vec_to_mat(prm[:3])[0], prm[3:].reshape(3, 1)


# ----------------------------------------------------------------------------- JSON + metrics (a12)
def results_to_json(all_results) -> str:
    out = []
    for im_res in all_results:
        im_name, cam_R, cam_t = im_res["imgname"], im_res["cam_R"], im_res["cam_t"]
        for human in im_res["result"]:
            r = {"image_id": im_name.split("/")[-1]}
            if len(cam_R) > 0:
                r["cam_R"] = np.array(cam_R).reshape((9, 1))[:, 0].tolist()
                r["cam_t"] = np.array(cam_t).reshape((3, 1))[:, 0].tolist()
            kp, sc = human["keypoints"], human["kp_score"]
            flat = []
            for n in range(sc.shape[0]):
                flat += [float(kp[n, 0]), float(kp[n, 1]), float(sc[n])]
            r["keypoints"] = flat
            r["score"] = float(human["proposal_score"])
            out.append(r)
    return json.dumps(out)


def add_err(gt_pose, est_pose, model):
    a = model @ gt_pose[:3, :3].T + gt_pose[:3, 3]
    b = model @ est_pose[:3, :3].T + est_pose[:3, 3]
    return float(np.mean(np.linalg.norm(a - b, axis=1)))


def projection_error_2d(gt_pose, est_pose, model, cam):
    m = np.concatenate((model, np.ones((model.shape[0], 1))), axis=1)
    g = cam @ gt_pose[:3] @ m.T
    e = cam @ est_pose[:3] @ m.T
    g, e = g / g[2], e / e[2]
    return float(np.mean(np.linalg.norm(g[:2].T - e[:2].T, axis=1)))


def iou(gt_box, est_box):
    xA, yA = max(gt_box[0], est_box[0]), max(gt_box[1], est_box[1])
    xB, yB = min(gt_box[2], est_box[2]), min(gt_box[3], est_box[3])
    if xB <= xA or yB <= yA:
        return 0.0
    inter = (xB - xA) * (yB - yA)
    A = (gt_box[2] - gt_box[0]) * (gt_box[3] - gt_box[1])
    B = (est_box[2] - est_box[0]) * (est_box[3] - est_box[1])
    return inter / float(A + B - inter)
