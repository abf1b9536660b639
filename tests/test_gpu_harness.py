"""The drop-in stage classes and the evaluation harness end to end (synthetic frames/weights): the staged pipeline
(reference structure) and the fused hipGraph pipeline must produce the same records as the reference's golden run."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import helpers  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", [[], ["--fused"]])
def test_evaluate_synthetic_matches_reference_json(tmp_path, mode):
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "evaluate.py"), "--synthetic", "4", "--outdir", str(out),
                        "--sp", "--profile"] + mode, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    got = json.loads(open(out / "Betapose-results.json").read())
    ref = json.loads(str(helpers.golden("pipeline.npz")["json_text"]))
    assert [g["image_id"] for g in got] == [r_["image_id"] for r_ in ref]
    for g, r_ in zip(got, ref):
        np.testing.assert_allclose(g["keypoints"], r_["keypoints"], rtol=1e-4, atol=5e-3)   # (x, y, score) x 50
        assert abs(g["score"] - r_["score"]) < 1e-3
        assert len(g["cam_R"]) == 9 and len(g["cam_t"]) == 3


def test_dynamic_write_results_convention(cuda):
    import torch
    from betapose_amd.yolo_util import dynamic_write_results
    pred = torch.zeros(2, 100, 6)
    pred[0, 17] = torch.tensor([50.0, 60.0, 20.0, 10.0, 0.9, 0.8])
    pred[0, 5] = torch.tensor([10.0, 10.0, 4.0, 4.0, 0.9, 0.7])      # tie on objectness: first index wins
    dets = dynamic_write_results(pred, 0.5, 80)
    assert dets.shape == (1, 8) and dets[0, 0] == 0
    np.testing.assert_allclose(dets[0, 1:].numpy(), [8, 8, 12, 12, 0.9, 0.7, 0], rtol=1e-6)
    assert dynamic_write_results(pred, 0.95, 80) == 0
