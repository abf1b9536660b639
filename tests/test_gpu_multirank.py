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


def _launch(nproc, script_args, timeout=900, backend="gloo"):
    env = dict(os.environ, BP_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
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
    assert out["rccl"]["ranks"] == 2 and out["rccl"]["backend"] == "gloo" and len(out["rccl"]["per_rank_frames_per_sec"]) == 2
    assert out["rccl"]["weight_broadcast_ms"] > 0 and 0 < out["latency_ms"]["p50"] <= out["latency_ms"]["p95"]


def test_bench_one_rank_under_the_launcher_equals_the_plain_line(cuda):
    """Round-5 verdict item 8: the driver's SCALE run starts with N = 1 -- `python bench.py --gpus 1` -- and continues with the launcher.
    The launcher's N = 1 (WORLD_SIZE 1: no process group, no `rccl` object) must be the same measurement as the plain line: same
    schema, same step count, frames/s within the run-to-run noise of a 40-step region."""
    flags = ["--gpus", "1", "--steps", "40", "--warmup", "10", "--no-cpu-baseline", "--no-roofline", "--no-served-legs",
             "--no-flip-rate", "--no-side-runs", "--other-modes", "", "--repeats", "1"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    plain = subprocess.run([sys.executable, "bench.py"] + flags, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert plain.returncode == 0, plain.stdout[-3000:] + plain.stderr[-3000:]
    a = json.loads([l for l in plain.stdout.splitlines() if l.startswith("{")][0])
    r = _launch(1, ["bench.py"] + flags, backend="nccl")          # exactly the driver's command shape, backend left at its default
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    b = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert a["n_gpus"] == b["n_gpus"] == 1 and "rccl" not in a and "rccl" not in b
    assert set(a) == set(b) and a["config"] == b["config"] and a["records_gathered"] == b["records_gathered"] == 40
    assert abs(a["value"] - b["value"]) / a["value"] < 0.15, (a["value"], b["value"])


def test_bench_two_ranks_rccl(cuda):
    """First contact with a multi-GPU node: the SAME command the driver's scaling run issues, collectives over RCCL
    (backend nccl, one rank per GPU).  Skips on a one-GPU box -- there the gloo test above covers everything but the
    transport."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (the 1-GPU test box runs the gloo variant)")
    r = _launch(2, ["bench.py", "--gpus", "2", "--steps", "24", "--warmup", "4", "--no-cpu-baseline", "--no-roofline"],
                backend="nccl")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["rccl"]["backend"] == "nccl" and out["records_gathered"] == 48
    r = _launch(2, ["evaluate.py", "--synthetic", "7", "--outdir", "/tmp/bp_rccl_eval", "--fused"], backend="nccl")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


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


def test_occlusion_multi_object_units(tmp_path):
    """Occlusion-LineMod as (frame, object) units (BASELINE configs[4]): a synthetic SIXD tree with three
    objects per frame whose ground truth is written from each object's own single-object pipeline (closed loop).
    One run with --obj_ids must (a) print 1.000 for every object, (b) give per-object JSON identical to that
    object's single-object run, and (c) give identical JSON whether the units run on 1 rank or are sharded over 2."""
    import re
    import torch
    from PIL import Image
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    from betapose_amd import synth
    from betapose_amd.darknet import Darknet
    from betapose_amd.kpd import FastPoseHIP
    from betapose_amd.pipeline import FramePipeline, finish_record
    from betapose_amd.weights import fastpose_stream_from_state_dict
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    objs, left = [1, 5, 6], 10
    frames = helpers.frames(3)
    indir = tmp_path / "rgb"
    indir.mkdir()
    for i, fr in enumerate(frames):
        Image.fromarray(fr[:, :, ::-1].copy()).save(indir / ("%04d.png" % i))
    kp_mm = {o: np.round(synth.synth_kp3d(50, seed=7 + o) * 1000.0, 6) for o in objs}
    gt = {i: [(7, np.eye(3), np.array([0.0, 0.0, 800.0]), [5, 5, 20, 20])] for i in range(len(frames))}
    for o in objs:
        sy, sk = synth.object_seeds(o)
        det = Darknet("yolo/cfg/yolov3-single.cfg", reso=416).load_stream(synth.synth_yolo_stream(sy)).cuda()
        pose = FastPoseHIP.from_stream(fastpose_stream_from_state_dict(synth.synth_fastpose_state_dict(sk, 50), 50), n_classes=50).cuda()
        pipe = FramePipeline(det, pose, 480, 640, batch=1, confidence=0.01)
        for i, fr in enumerate(frames):
            out = finish_record(pipe.run(fr)[0], "%04d.png" % i, kp_mm[o] / 1000.0, synth.CAM_K, left)
            assert out["boxes"] is not None and len(out["result"]) == 1
            x1, y1, x2, y2 = [float(v) for v in out["result"][0]["bbox"]]
            gt[i].append((o, out["cam_R"], np.asarray(out["cam_t"]).reshape(3) * 1000.0, [x1, y1, x2 - x1, y2 - y1]))
        del pipe, det, pose
    rng = np.random.default_rng(0)
    synth.write_sixd_tree(str(tmp_path / "sixd"), 2, gt, {o: rng.normal(size=(300, 3)) * 30.0 for o in objs}, kp_mm,
                          {o: 100.0 for o in objs})
    common = ["--indir", str(indir), "--sixd_base", str(tmp_path / "sixd"), "--synth_weights", "--fused",
              "--left_keypoints", str(left), "--streams", "2"]
    script = os.path.join(ROOT, "occlusion_evaluate.py")

    def run(nproc, extra, outdir, backend="gloo"):
        if nproc == 1:
            r = subprocess.run([sys.executable, script] + common + extra + ["--outdir", str(outdir)], capture_output=True,
                               text=True, timeout=900, cwd=ROOT)
        else:
            r = _launch(nproc, [script] + common + extra + ["--outdir", str(outdir)], backend=backend)
        assert r.returncode == 0, r.stdout + r.stderr
        return r.stdout
    out1 = run(1, ["--obj_ids", "1,5,6"], tmp_path / "m1")
    for o in objs:
        nums = dict(re.findall(r"(Mean add accuracy|2d reprojection accuracy with leftkeypoints \d+|Mean IoU) for seq %02d is: ([\d.nan]+)" % o, out1))
        assert list(nums.values()) == ["1.000"] * 3, out1
    run(2, ["--obj_ids", "1,5,6"], tmp_path / "m2")
    if torch.cuda.device_count() >= 2:      # a multi-GPU node: the same units over RCCL, one rank per GPU (BASELINE configs[4])
        run(2, ["--obj_ids", "1,5,6"], tmp_path / "m2n", backend="nccl")
        for o in objs:
            assert open(tmp_path / "m2" / ("obj_%02d" % o) / "Betapose-results.json").read() == \
                open(tmp_path / "m2n" / ("obj_%02d" % o) / "Betapose-results.json").read()
    for o in objs:
        j1 = open(tmp_path / "m1" / ("obj_%02d" % o) / "Betapose-results.json").read()
        assert j1 == open(tmp_path / "m2" / ("obj_%02d" % o) / "Betapose-results.json").read()       # 1 rank == 2 ranks
        run(1, ["--obj_id", str(o)], tmp_path / ("s%d" % o))
        assert j1 == open(tmp_path / ("s%d" % o) / "Betapose-results.json").read()                   # == single-object run
        assert len(json.loads(j1)) == 3
