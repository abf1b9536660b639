"""Product host post-processing (numpy) against the reference's golden outputs. CPU only."""
import json

import numpy as np
import pytest
import torch

import helpers
from betapose_amd import eval as bp_eval, pPose_nms as bp_nms


def test_get_prediction_matches_reference():
    post = helpers.golden("post.npz")
    hms = post["gp_hms"].astype(np.float32)
    a, b, c = bp_eval.getPrediction(hms, post["gp_pt1"], post["gp_pt2"], 320, 256, 80, 64)
    np.testing.assert_array_equal(a, post["gp_preds_hm"])
    np.testing.assert_allclose(b, post["gp_preds_img"], rtol=1e-6, atol=2e-5)
    np.testing.assert_array_equal(c, post["gp_maxval"])
    # torch in -> torch out
    ta, tb, tc = bp_eval.getPrediction(torch.from_numpy(hms), torch.from_numpy(post["gp_pt1"]),
                                       torch.from_numpy(post["gp_pt2"]), 320, 256, 80, 64)
    assert isinstance(ta, torch.Tensor) and tuple(tb.shape) == (3, 50, 2) and tuple(tc.shape) == (3, 50, 1)


def test_decode_from_records_matches_pipeline_golden():
    pipe = helpers.golden("pipeline.npz")
    for i in range(int(pipe["n_frames"])):
        k = "f%d_" % i
        rec = np.zeros((1, 50, 6), np.float32)
        rec[0, :, 0] = pipe[k + "kp_idx"].astype(np.int32).view(np.float32)
        rec[0, :, 1] = pipe[k + "kp_max"]
        rec[0, :, 2:] = pipe[k + "kp_nb"]
        a, b, c = bp_eval.decode_keypoints(rec, pipe[k + "pt1"], pipe[k + "pt2"])
        np.testing.assert_array_equal(a, pipe[k + "preds_hm"])
        np.testing.assert_allclose(b, pipe[k + "preds_img"], rtol=1e-6, atol=2e-5)
        np.testing.assert_array_equal(c, pipe[k + "preds_scores"])
        res = bp_nms.pose_nms(pipe[k + "boxes"], pipe[k + "scores"], b, c)
        assert len(res) == int(pipe[k + "nms_n"]) == 1
        np.testing.assert_allclose(res[0]["keypoints"], pipe[k + "nms_kp"], rtol=1e-6, atol=2e-5)
        np.testing.assert_allclose(res[0]["kp_score"], pipe[k + "nms_score"], rtol=1e-6)
        assert abs(float(res[0]["proposal_score"][0]) - float(pipe[k + "nms_prop"])) < 1e-5
        np.testing.assert_array_equal(res[0]["bbox"], pipe[k + "nms_bbox"])


def test_pose_nms_multi_matches_reference():
    post = helpers.golden("post.npz")
    res = bp_nms.pose_nms(post["nms_in_boxes"], post["nms_in_scores"], post["nms_in_poses"], post["nms_in_pscores"])
    assert len(res) == int(post["nms_out_n"])
    for j, r in enumerate(res):
        np.testing.assert_allclose(r["keypoints"], post["nms_out%d_kp" % j], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(r["kp_score"], post["nms_out%d_score" % j], rtol=1e-5, atol=1e-6)
        assert abs(float(r["proposal_score"][0]) - float(post["nms_out%d_prop" % j])) < 1e-5


def test_pose_nms_drops_weak_pose():
    boxes = np.array([[0, 0, 100, 100]], np.float32)
    res = bp_nms.pose_nms(boxes, np.array([[0.9]], np.float32), np.zeros((1, 50, 2), np.float32),
                          np.full((1, 50, 1), 0.1, np.float32))
    assert res == []


def test_write_json_matches_reference(tmp_path):
    pipe = helpers.golden("pipeline.npz")
    results = []
    for i in range(int(pipe["n_frames"])):
        k = "f%d_" % i
        res = bp_nms.pose_nms(pipe[k + "boxes"], pipe[k + "scores"], pipe[k + "preds_img"], pipe[k + "preds_scores"])
        R = np.eye(3) + 0.01 * i
        t = np.array([[0.01 * i], [0.02], [0.9]])
        results.append({"imgname": "%04d.png" % i, "result": res, "cam_R": R, "cam_t": t})
    path = bp_nms.write_json(results, str(tmp_path))
    got = json.loads(open(path).read())
    ref = json.loads(str(pipe["json_text"]))
    assert len(got) == len(ref)
    for g, r in zip(got, ref):
        assert g["image_id"] == r["image_id"]
        np.testing.assert_allclose(g["cam_R"], r["cam_R"])
        np.testing.assert_allclose(g["cam_t"], r["cam_t"])
        np.testing.assert_allclose(g["keypoints"], r["keypoints"], rtol=1e-6, atol=2e-5)
        assert abs(g["score"] - r["score"]) < 1e-5


def test_pose_nms_seeded_candidate_sets_match_reference():
    """24 candidate sets (1-6 poses: near-duplicates, loosely similar, shifted, unrelated; weak poses; exact-zero key-point
    scores) through the reference's pose_nms (tools/make_golden_edges.py): same number of surviving poses, same merged
    key points / scores / proposal scores / boxes, for the product's numpy pose_nms and for the oracle."""
    import torch
    from betapose_amd.pPose_nms import pose_nms
    from oracle import post_ref
    g = helpers.golden("edges.npz")
    counts = []
    for t in range(int(g["nms_n"])):
        args = (g["nms%d_boxes" % t], g["nms%d_bsc" % t], g["nms%d_poses" % t], g["nms%d_psc" % t])
        keep = [a.copy() for a in args]
        for fn in (pose_nms, lambda *a: post_ref.pose_nms(*[torch.from_numpy(x.copy()) for x in a])):
            res = fn(*args)
            n = int(g["nms%d_n" % t])
            assert len(res) == n, (t, len(res), n)
            for j, r in enumerate(res):
                np.testing.assert_allclose(np.asarray(r["keypoints"]), g["nms%d_o%d_kp" % (t, j)], atol=2e-4, rtol=0)
                np.testing.assert_allclose(np.asarray(r["kp_score"]), g["nms%d_o%d_score" % (t, j)], atol=2e-6, rtol=0)
                assert abs(float(np.asarray(r["proposal_score"]).reshape(-1)[0]) - float(g["nms%d_o%d_prop" % (t, j)])) < 1e-5
                np.testing.assert_allclose(np.asarray(r["bbox"]), g["nms%d_o%d_bbox" % (t, j)], atol=1e-5, rtol=0)
        for a, k in zip(args, keep):
            np.testing.assert_array_equal(a, k)          # inputs are not mutated (the reference squeezes/deletes in place)
        counts.append(n)
    assert min(counts) == 1 and max(counts) >= 4


def test_write_json_layouts_equal_the_reference(tmp_path, golden_dir):
    """pPose_nms.py:284-371 reads opt.format: the default list and the 'cmu' / 'open' per-image layouts (+ sep-json files)
    against the files the reference's own write_json wrote for the same seeded results (tools/make_golden_json.py), text
    for text; for_eval=True with a body layout fails, as in the reference."""
    import os
    import torch
    from betapose_amd import pPose_nms
    from betapose_amd.opt import opt
    g = np.load(os.path.join(golden_dir, "json_formats.npz"))
    results, n = [], 0
    for i, name in enumerate(g["names"]):
        humans = []
        for _ in range(int(g["per_image"][i])):
            humans.append({"keypoints": torch.from_numpy(g["kp"][n]), "kp_score": torch.from_numpy(g["sc"][n]),
                           "proposal_score": torch.tensor([float(g["prop"][n])])})
            n += 1
        results.append({"imgname": "some/dir/" + str(name), "result": humans, "cam_R": g["R"][i] if i != 1 else [],
                        "cam_t": g["t"][i] if i != 1 else []})
    old = getattr(opt, "format", None)
    try:
        for form in (None, "cmu", "open"):
            for for_eval in ((False, True) if form is None else (False,)):
                opt.format = form
                d = tmp_path / ("%s_%d" % (form, for_eval))
                d.mkdir()
                pPose_nms.write_json(results, str(d), for_eval=for_eval)
                key = "%s_%d" % (form or "default", int(for_eval))
                assert (d / "Betapose-results.json").read_text() == str(g["main_" + key])
                if form:
                    assert sorted(os.listdir(d / "sep-json")) == list(g["sepnames_" + key])
                    for nm, txt in zip(g["sepnames_" + key], g["sep_" + key]):
                        assert (d / "sep-json" / str(nm)).read_text() == str(txt)
        opt.format = "cmu"
        with pytest.raises(AttributeError):
            pPose_nms.write_json(results, str(tmp_path), for_eval=True)
        opt.format = old
        pPose_nms.write_json([], str(tmp_path))
        assert (tmp_path / "Betapose-results.json").read_text() == "[]"
    finally:
        opt.format = old


def test_image_loader_forwards_decode_errors(tmp_path):
    """A corrupt frame must surface in the consumer instead of leaving it blocked on the queue (dataloader.py:150-179
    dies silently in its thread)."""
    from betapose_amd.dataloader import ImageLoader
    (tmp_path / "bad.png").write_bytes(b"\x89PNG\r\n\x1a\nnot a png")
    from betapose_amd.opt import opt
    old = opt.inputpath
    opt.inputpath = str(tmp_path)
    try:
        loader = ImageLoader(["bad.png"], batchSize=1, format="yolo", reso=416).start()
        with pytest.raises(RuntimeError, match="frame input failed"):
            loader.getitem()
    finally:
        opt.inputpath = old
