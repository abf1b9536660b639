"""The N>1 drivers on a real GPU: two ranks launched by torch.distributed.run exactly as the scaling bench does,
sharing the one GPU of the test box with the collectives routed through gloo (BP_DIST_BACKEND=gloo; on a multi-GPU
node the same code runs one rank per GPU over RCCL).  Checks the weight broadcast -> engines -> sharded frames ->
record gather -> rank-0 output flow of bench.py and evaluate.py --fused."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _launch(nproc, script_args, timeout=900):
    env = dict(os.environ, BP_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_port())] + script_args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


def test_bench_two_ranks(cuda):
    r = _launch(2, ["bench.py", "--gpus", "2", "--steps", "24", "--warmup", "4", "--no-cpu-baseline", "--no-roofline"])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 24 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 2 and out["value"] > 0
    assert out["records_gathered"] == 48 and out["detections"] == 24     # rank 0's own frames all detected
    assert abs(out["value"] - 2 * 24 / (out["ms_per_step"] * 24 / 1e3)) / out["value"] < 1e-3


def test_evaluate_fused_two_ranks_equals_one(tmp_path, cuda):
    outs = []
    for n in (1, 2):
        out = tmp_path / ("out%d" % n)
        r = _launch(n, ["evaluate.py", "--synthetic", "7", "--outdir", str(out), "--fused"])
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        outs.append(json.loads(open(out / "Betapose-results.json").read()))
    assert [g["image_id"] for g in outs[0]] == [g["image_id"] for g in outs[1]] and len(outs[0]) == 7
    for a, b in zip(*outs):
        assert a["keypoints"] == b["keypoints"] and a["cam_R"] == b["cam_R"] and a["cam_t"] == b["cam_t"]
