"""Pins the CPU oracle (oracle/) against golden vectors produced by the REFERENCE's own
Python (tools/make_golden.py).  Runs without a GPU.  torch-CPU kernels may differ between
machines in the last bits, so float comparisons carry a small tolerance; integer results
(arg-max indices, key-point pixels, u8 images) must be exact."""
import json

import numpy as np
import pytest
import torch

import helpers
from betapose_amd import weights as W
from oracle import kpd_ref, post_ref, yolo_ref


@pytest.fixture(scope="module")
def pipe():
    return helpers.golden("pipeline.npz")


@pytest.fixture(scope="module")
def post():
    return helpers.golden("post.npz")


def test_oracle_yolo_matches_reference(pipe):
    blocks = helpers.yolo_blocks()
    convs = W.split_darknet_stream(blocks, helpers.yolo_stream())
    fr = helpers.frames()[0]
    x = helpers.yolo_input_from_frame(fr)
    assert np.allclose(x.numpy().ravel()[pipe["in_samp"]], pipe["f0_yolo_in_samp"], atol=0)
    assert int(torch.round(x * 255).long().sum()) == int(pipe["f0_yolo_in_u8sum"])
    pred = yolo_ref.darknet_forward(blocks, convs, x)
    np.testing.assert_allclose(pred[0].numpy()[pipe["row_samp"]], pipe["f0_pred_rows"], rtol=1e-5, atol=1e-4)
    assert int(torch.argmax(pred[0, :, 4])) == int(pipe["f0_obj_argmax"])
    assert int(yolo_ref.select_index(pred, 0.01)[0]) == int(pipe["f0_obj_argmax"])
    dets = yolo_ref.write_results(pred, 0.01, 80)
    np.testing.assert_allclose(dets.numpy(), pipe["f0_det_row"], rtol=1e-5, atol=1e-4)
    im_dim = torch.tensor([[640.0, 480.0, 640.0, 480.0]])
    boxes, scores = yolo_ref.rescale_boxes(dets, im_dim, 416)
    np.testing.assert_allclose(boxes.numpy(), pipe["f0_boxes"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(scores.numpy(), pipe["f0_scores"], rtol=1e-5)
    assert yolo_ref.write_results(pred, 0.9999, 80) == 0


def test_oracle_crop_matches_reference(pipe):
    for i, fr in enumerate(helpers.frames(int(pipe["n_frames"]))):
        k = "f%d_" % i
        inps, pt1, pt2 = post_ref.crop_from_dets_frame(fr, torch.from_numpy(pipe[k + "boxes"]))
        np.testing.assert_array_equal(pt1.numpy(), pipe[k + "pt1"])
        np.testing.assert_array_equal(pt2.numpy(), pipe[k + "pt2"])
        np.testing.assert_allclose(inps.numpy().ravel()[pipe["crop_samp"]], pipe[k + "crop_samp"], atol=1e-6)
        assert abs(float(inps.double().sum()) - float(pipe[k + "crop_sum"])) < 1e-2


def test_oracle_kpd_matches_reference(pipe):
    sd = helpers.kpd_state_dict()
    inps, _, _ = post_ref.crop_from_dets_frame(helpers.frames()[0], torch.from_numpy(pipe["f0_boxes"]))
    hm = kpd_ref.fastpose_forward(sd, inps)
    np.testing.assert_allclose(hm.numpy().ravel()[pipe["hm_samp"]], pipe["f0_hm_samp"], atol=2e-5)
    assert np.array_equal(hm.view(50, -1).argmax(1).numpy(), pipe["f0_kp_idx"])


def test_oracle_get_prediction_and_nms_match_reference(pipe, post):
    hms = torch.from_numpy(post["gp_hms"].astype(np.float32))
    a, b, c = post_ref.get_prediction(hms, torch.from_numpy(post["gp_pt1"]), torch.from_numpy(post["gp_pt2"]))
    np.testing.assert_array_equal(a.numpy(), post["gp_preds_hm"])
    np.testing.assert_allclose(b.numpy(), post["gp_preds_img"], rtol=1e-6, atol=1e-5)
    np.testing.assert_array_equal(c.numpy(), post["gp_maxval"])
    res = post_ref.pose_nms(torch.from_numpy(post["nms_in_boxes"]), torch.from_numpy(post["nms_in_scores"]),
                            torch.from_numpy(post["nms_in_poses"]), torch.from_numpy(post["nms_in_pscores"]))
    assert len(res) == int(post["nms_out_n"])
    for j, r in enumerate(res):
        np.testing.assert_allclose(r["keypoints"].numpy(), post["nms_out%d_kp" % j], rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(r["kp_score"].numpy(), post["nms_out%d_score" % j], rtol=1e-6, atol=1e-6)
        assert abs(float(r["proposal_score"]) - float(post["nms_out%d_prop" % j])) < 1e-5
        np.testing.assert_array_equal(r["bbox"].numpy(), post["nms_out%d_bbox" % j])
    # the n = 1 pipeline case
    for i in range(int(pipe["n_frames"])):
        k = "f%d_" % i
        res = post_ref.pose_nms(torch.from_numpy(pipe[k + "boxes"]), torch.from_numpy(pipe[k + "scores"]),
                                torch.from_numpy(pipe[k + "preds_img"]), torch.from_numpy(pipe[k + "preds_scores"]))
        assert len(res) == int(pipe[k + "nms_n"])
        if res:
            np.testing.assert_allclose(res[0]["keypoints"].numpy(), pipe[k + "nms_kp"], rtol=1e-6, atol=1e-5)
            np.testing.assert_allclose(res[0]["kp_score"].numpy(), pipe[k + "nms_score"], rtol=1e-6, atol=1e-6)


def test_oracle_metrics_match_reference(post):
    assert abs(post_ref.add_err(post["m_gt"], post["m_est"], post["m_model"]) - float(post["m_add"])) < 1e-12
    from betapose_amd.synth import CAM_K
    assert abs(post_ref.projection_error_2d(post["m_gt"], post["m_est"], post["m_model"], CAM_K) - float(post["m_proj"])) < 1e-9
    assert abs(post_ref.iou([10, 10, 110, 210], [30, 40, 100, 260]) - float(post["m_iou"][0])) < 1e-12
    assert post_ref.iou([10, 10, 50, 50], [60, 60, 80, 80]) == float(post["m_iou"][1]) == 0.0


@pytest.fixture(scope="module")
def edges():
    return helpers.golden("edges.npz")


def test_oracle_crop_edge_boxes_match_reference(edges):
    """64 boxes through the reference's own crop_from_dets / cropBox (tools/make_golden_edges.py): tiny, huge, crossing
    the border, fractional corners, both sides of the width-100 pad-rule switch."""
    from betapose_amd import synth
    fr = synth.synth_frame(int(edges["crop_frame_seed"]))
    assert bool(edges["crop_ok"].all())                      # the reference handled every one of them
    for i, b in enumerate(edges["crop_boxes"]):
        inps, pt1, pt2 = post_ref.crop_from_dets_frame(fr, torch.from_numpy(b[None]))
        np.testing.assert_array_equal(pt1.numpy()[0], edges["crop_pt1"][i], err_msg=str(b))
        np.testing.assert_array_equal(pt2.numpy()[0], edges["crop_pt2"][i], err_msg=str(b))
        flat = inps[0].reshape(-1)
        assert np.abs(flat.numpy()[edges["crop_samp_idx"]] - edges["crop_samples"][i]).max() <= 1e-6, b
        assert abs(float(flat.double().sum()) - float(edges["crop_sum"][i])) < 0.02
        assert abs(float(flat.double().abs().sum()) - float(edges["crop_abs_sum"][i])) < 0.02


def test_oracle_select_matches_reference(edges):
    """48 prediction tensors through the reference's dynamic_write_results: multi-class rows, images / whole batches
    without a candidate (int 0)."""
    n_empty = 0
    for t in range(int(edges["sel_n"])):
        pred = torch.from_numpy(edges["sel%d_pred" % t])
        want = edges["sel%d_out" % t]
        got = yolo_ref.write_results(pred.clone(), float(edges["sel%d_conf" % t]), 80)
        if len(want) == 0:
            assert isinstance(got, int) and got == 0
            n_empty += 1
        else:
            assert got.shape == want.shape
            np.testing.assert_allclose(got.numpy(), want, rtol=1e-6, atol=1e-5, err_msg="case %d" % t)
    assert 3 <= n_empty <= 20


def test_get_prediction_planted_cases_match_reference(edges):
    """Heat-maps with maxima on corners / borders, two equal maxima, all-negative and all-zero maps, equal neighbours:
    the reference's getPrediction output (tools/make_golden_edges.py) against the oracle AND the product's host decode
    of arg-max records (betapose_amd.eval), all exact."""
    from betapose_amd.eval import decode_keypoints, kp_records_from_heatmaps
    hm = edges["gpe_hms"].astype(np.float32)
    a, b, c = post_ref.get_prediction(torch.from_numpy(hm), torch.from_numpy(edges["gpe_pt1"]),
                                      torch.from_numpy(edges["gpe_pt2"]))
    np.testing.assert_array_equal(a.numpy(), edges["gpe_preds_hm"])
    np.testing.assert_allclose(b.numpy(), edges["gpe_preds_img"], rtol=0, atol=1e-4)
    np.testing.assert_array_equal(c.numpy(), edges["gpe_maxval"])
    a2, b2, c2 = decode_keypoints(kp_records_from_heatmaps(hm), edges["gpe_pt1"], edges["gpe_pt2"])
    np.testing.assert_array_equal(a2, edges["gpe_preds_hm"])
    np.testing.assert_allclose(b2, edges["gpe_preds_img"], rtol=0, atol=1e-4)
    np.testing.assert_array_equal(c2, edges["gpe_maxval"])


# ---- BASELINE configs[0] at its own size: tests/golden/sweep64.npz holds compact records of 64 frames through the reference's own stage
# classes (tools/make_golden_sweep64.py).  The oracle is pinned to a spread of them here (the whole 64 run through the HIP pipeline in
# tests/test_gpu_sweep64.py); ~1.5 s of torch-CPU per frame keeps it to six
@pytest.mark.parametrize("i", [4, 17, 29, 41, 52, 63])
def test_oracle_matches_reference_on_sweep64_frames(i):
    g = helpers.golden("sweep64.npz")
    torch.set_num_threads(min(16, torch.get_num_threads()))
    blocks = helpers.yolo_blocks()
    convs = W.split_darknet_stream(blocks, helpers.yolo_stream())
    from betapose_amd import synth
    fr = synth.synth_frame(helpers.FRAME_SEED + i)
    pred = yolo_ref.darknet_forward(blocks, convs, helpers.yolo_input_from_frame(fr))
    assert int(torch.argmax(pred[0, :, 4])) == int(g["obj_argmax"][i])
    np.testing.assert_allclose(torch.topk(pred[0, :, 4], 2).values.numpy(), g["obj_top2"][i], rtol=0, atol=2e-6)
    dets = yolo_ref.write_results(pred, 0.01, 80)
    np.testing.assert_allclose(dets.numpy()[0], g["det_row"][i], rtol=1e-5, atol=1e-4)
    boxes, scores = yolo_ref.rescale_boxes(dets, torch.tensor([[640.0, 480.0, 640.0, 480.0]]), 416)
    np.testing.assert_allclose(boxes.numpy()[0], g["boxes"][i], rtol=1e-5, atol=1e-4)
    inps, pt1, pt2 = post_ref.crop_from_dets_frame(fr, torch.from_numpy(g["boxes"][i:i + 1]))
    np.testing.assert_array_equal(pt1.numpy()[0], g["pt1"][i])
    np.testing.assert_array_equal(pt2.numpy()[0], g["pt2"][i])
    hm = kpd_ref.fastpose_forward(helpers.kpd_state_dict(), inps)
    flat = hm.view(50, -1)
    sure = g["kp_margin"][i] > 4e-5          # (the restatement and the reference module differ by <= 2e-5 per heat-map value)
    assert np.array_equal(flat.argmax(1).numpy()[sure], g["kp_idx"][i].astype(np.int64)[sure])
    np.testing.assert_allclose(flat.max(1).values.numpy(), g["kp_max"][i], rtol=0, atol=2e-5)
    _, preds_img, preds_scores = post_ref.get_prediction(hm, pt1, pt2)
    np.testing.assert_allclose(preds_img.numpy()[0][sure], g["preds_img"][i][sure], rtol=1e-6, atol=1e-4)
    res = post_ref.pose_nms(boxes, scores, torch.from_numpy(g["preds_img"][i:i + 1]), torch.from_numpy(g["preds_scores"][i:i + 1, :, None]))
    assert len(res) == int(g["nms_n"][i]) == 1
    np.testing.assert_allclose(res[0]["keypoints"].numpy(), g["nms_kp"][i], rtol=1e-6, atol=1e-5)
    np.testing.assert_allclose(res[0]["kp_score"].numpy()[:, 0], g["nms_score"][i], rtol=1e-6, atol=1e-6)
