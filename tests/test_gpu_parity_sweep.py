"""Wider parity sweep than the four golden frames: 12 further seeded frames through the fused pipeline (fp32 mode)
against the oracle on the same inputs.  Integer results (YOLO box index, KPD arg-max pixel) must be identical
wherever the oracle's own best-vs-second margin exceeds the float tolerance of that stage; float results within the
tolerances of tests/test_gpu_nets.py."""
import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu

import helpers  # noqa: E402
from betapose_amd import synth, weights as W  # noqa: E402
from betapose_amd.darknet import Darknet  # noqa: E402
from betapose_amd.kpd import FastPoseHIP  # noqa: E402
from betapose_amd.pipeline import FramePipeline  # noqa: E402
from oracle import kpd_ref, post_ref, yolo_ref  # noqa: E402

HM_TOL = 2e-4
PROB_TOL = 2e-5


def test_twelve_more_frames_match_the_oracle(cuda):
    torch.set_num_threads(16)
    blocks = helpers.yolo_blocks()
    convs = W.split_darknet_stream(blocks, helpers.yolo_stream())
    sd = {k: torch.from_numpy(v) for k, v in helpers.kpd_state_dict().items()}
    det = Darknet("yolo/cfg/yolov3-single.cfg", reso=416).load_stream(helpers.yolo_stream()).cuda()
    pose = FastPoseHIP(helpers.kpd_state_dict(), n_classes=50).cuda()
    pipe = FramePipeline(det, pose, 480, 640, batch=1, keep_heatmaps=True)
    checked_kp = skipped_kp = 0
    for seed in range(5000, 5012):
        frame = synth.synth_frame(seed)
        rec = pipe.run(frame)[0]
        hm_gpu = pipe.heatmaps.cpu()[0]
        # oracle
        img = Image.fromarray(np.ascontiguousarray(frame[:, :, ::-1])).resize((416, 416), 3)
        x = torch.from_numpy(np.asarray(img, dtype=np.uint8).transpose(2, 0, 1).copy()).float().div(255).unsqueeze(0)
        pred = yolo_ref.darknet_forward(blocks, convs, x)
        obj = pred[0, :, 4]
        top2 = torch.topk(obj, 2).values
        idx = int(rec[:1].view(np.int32)[0])
        if float(top2[0] - top2[1]) > PROB_TOL:
            assert idx == int(obj.argmax()), "seed %d: YOLO box index" % seed
        else:
            assert abs(float(obj[idx]) - float(top2[0])) <= PROB_TOL
        dets = yolo_ref.write_results(pred, 0.01, 80)
        boxes, _ = yolo_ref.rescale_boxes(dets, torch.tensor([[640.0, 480.0, 640.0, 480.0]]), 416)
        assert np.abs(boxes.numpy() - rec[12:16]).max() < 5e-3
        inps, _, _ = post_ref.crop_from_dets_frame(frame, boxes)
        hm = kpd_ref.fastpose_forward(sd, inps)[0]
        assert float((hm_gpu - hm).abs().max()) <= HM_TOL, "seed %d: heat-map" % seed
        flat = hm.reshape(50, -1)
        t2 = torch.topk(flat, 2, dim=1).values
        margin = (t2[:, 0] - t2[:, 1]).numpy()
        got = rec[16:].reshape(50, 6)[:, 0].copy().view(np.int32)
        ref = flat.argmax(1).numpy()
        sure = margin > 2 * HM_TOL
        assert np.array_equal(got[sure], ref[sure]), "seed %d: KPD arg-max pixels" % seed
        # a pixel the oracle itself separates by less than the tolerance may legitimately differ, but must be as high
        for k in np.nonzero(~sure)[0]:
            assert float(flat[k, ref[k]] - flat[k, got[k]]) <= 2 * HM_TOL
        checked_kp += int(sure.sum())
        skipped_kp += int((~sure).sum())
    assert checked_kp >= 550 and skipped_kp <= 50, (checked_kp, skipped_kp)
