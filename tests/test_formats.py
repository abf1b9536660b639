"""Weight/cfg formats and the C-ABI surface.  CPU only."""
import ctypes
import hashlib
import json
import os
import re

import numpy as np
import pytest

import helpers
from betapose_amd import _lib, cfg as C, weights as W, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_generated_cfg_equals_reference_cfg():
    fmt = helpers.golden("formats.npz")
    blocks = C.parse_cfg_text(C.yolov3_single_cfg_text())
    digest = hashlib.sha256(json.dumps(blocks, sort_keys=True).encode()).hexdigest()
    assert digest == str(fmt["cfg_sha256"])          # digest of the REFERENCE's parse of its own cfg
    kinds = [b["type"] for b in blocks]
    assert kinds.count("convolutional") == 75 and kinds.count("shortcut") == 23
    assert kinds.count("route") == 4 and kinds.count("upsample") == 2 and kinds.count("yolo") == 3


def test_cfg_parser_edge_cases():
    blocks = C.parse_cfg_text("# c\n\n[net]\nwidth = 416 \n[convolutional]\nfilters=8\nsize = 3\n  stride=1\npad=1\nactivation=leaky\n")
    assert blocks[0] == {"type": "net", "width": "416"}
    assert blocks[1]["size"] == "3" and blocks[1]["activation"] == "leaky"
    with pytest.raises(ValueError):
        C.parse_cfg_text("[convolutional]\nbroken line\n")
    assert C.parse_cfg("yolo/cfg/yolov3-single.cfg") == C.parse_cfg_text(C.yolov3_single_cfg_text())
    with pytest.raises(FileNotFoundError):
        C.parse_cfg("/nonexistent/other.cfg")


def test_synthetic_stream_is_reproducible():
    fmt = helpers.golden("formats.npz")
    stream = helpers.yolo_stream()
    assert stream.size == int(fmt["stream_size"]) == W.darknet_stream_size(helpers.yolo_blocks())
    assert hashlib.sha256(stream.tobytes()).hexdigest() == str(fmt["stream_sha256"])
    # values the REFERENCE loader placed into its conv modules from the file we wrote
    convs = {c["index"]: c for c in W.split_darknet_stream(helpers.yolo_blocks(), stream)}
    probed = sorted(int(k[1:-7]) for k in fmt.files if k.endswith("_first8"))
    assert len(probed) >= 5
    for i in probed:
        np.testing.assert_array_equal(convs[i]["weight"].ravel()[:8], fmt["w%d_first8" % i])
        assert abs(float(convs[i]["weight"].astype(np.float64).sum()) - float(fmt["w%d_sum" % i])) < 1e-6


def test_weights_file_round_trip(tmp_path):
    fmt = helpers.golden("formats.npz")
    raw = fmt["tiny_weights_bytes"].tobytes()
    path = tmp_path / "tiny.weights"
    path.write_bytes(raw)
    ver, seen, flat = W.read_darknet_weights(str(path))
    assert list(ver) == [0, 1, 0] and seen == 7
    blocks = C.parse_cfg_text(str(fmt["tiny_cfg"]))
    assert flat.size == W.darknet_stream_size(blocks)
    out = tmp_path / "again.weights"
    W.write_darknet_weights(str(out), flat, seen=7)
    assert out.read_bytes() == raw
    # 20-byte header variant (major*10+minor >= 2 -> 64-bit seen; parser.c:1161-1174): same payload
    out2 = tmp_path / "v2.weights"
    W.write_darknet_weights(str(out2), flat, seen=123456789012, version=(0, 2, 0))
    assert out2.stat().st_size == len(raw) + 4
    ver2, seen2, flat2 = W.read_darknet_weights(str(out2))
    assert list(ver2) == [0, 2, 0] and seen2 == 123456789012 and np.array_equal(flat2, flat)
    with pytest.raises(ValueError):
        (tmp_path / "bad.weights").write_bytes(b"\0" * 8)
        W.read_darknet_weights(str(tmp_path / "bad.weights"))
    with pytest.raises(ValueError):
        W.split_darknet_stream(blocks, flat[:-1])


def test_fastpose_stream_layout():
    sd = helpers.kpd_state_dict()
    keys = W.fastpose_state_dict_keys(50)
    assert len(keys) == 654 - 106      # 654 tensors in the reference state dict, 106 of them num_batches_tracked
    assert sum(int(np.prod(s)) for _, s in keys) == W.fastpose_stream_size(50) == 59716338
    assert set(sd) == {k for k, _ in keys}
    for k, shape in keys:
        assert sd[k].shape == shape, k
    stream = W.fastpose_stream_from_state_dict(sd)
    assert stream.size == W.fastpose_stream_size(50)
    # first conv: bn.bias, bn.weight, mean, var, then the 7x7 filter
    np.testing.assert_array_equal(stream[:64], sd["preact.bn1.bias"])
    np.testing.assert_array_equal(stream[64:128], sd["preact.bn1.weight"])
    np.testing.assert_array_equal(stream[256:256 + 64 * 3 * 49], sd["preact.conv1.weight"].ravel())
    np.testing.assert_array_equal(stream[-128 * 9 * 50:], sd["conv_out.weight"].ravel())
    bad = dict(sd)
    bad["conv_out.weight"] = bad["conv_out.weight"][:10]
    with pytest.raises(ValueError):
        W.fastpose_stream_from_state_dict(bad)


def test_bn_folding_matches_torch():
    import torch
    import torch.nn.functional as F
    g = np.random.Generator(np.random.PCG64(3))
    w = g.normal(size=(6, 4, 3, 3)).astype(np.float32)
    gamma, beta = g.uniform(0.5, 1.5, 6).astype(np.float32), g.normal(size=6).astype(np.float32)
    mean, var = g.normal(size=6).astype(np.float32), g.uniform(0.5, 2, 6).astype(np.float32)
    x = torch.from_numpy(g.normal(size=(1, 4, 8, 8)).astype(np.float32))
    ref = F.batch_norm(F.conv2d(x, torch.from_numpy(w), padding=1), torch.from_numpy(mean), torch.from_numpy(var),
                       torch.from_numpy(gamma), torch.from_numpy(beta), False, 0.1, 1e-5)
    wf, bf = W.fold_bn(w, gamma, beta, mean, var)
    got = F.conv2d(x, torch.from_numpy(wf), torch.from_numpy(bf), padding=1)
    assert float((got - ref).abs().max()) < 1e-5
    wd, bd = W.fold_bn(w, gamma, beta, mean, var, darknet_eps=True)   # (x-mean)/(sqrt(var)+1e-6), blas.c:136
    assert not np.array_equal(wd, wf)


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads (no GPU needed) and exports exactly what include/betapose_hip.h declares."""
    header = open(os.path.join(ROOT, "include", "betapose_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(bp_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 35
    lib = _lib.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), "libbetapose_hip.so does not export %s" % name
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert lib.bp_version() >= 100
    assert isinstance(lib.bp_last_error(), bytes)
    # include/yolo_v2_class_compat.h: the reference's six Darknet-library entry points (yolo_v2_class.hpp:49-54)
    compat = open(os.path.join(ROOT, "include", "yolo_v2_class_compat.h")).read()
    compat = re.sub(r"/\*.*?\*/", "", compat, flags=re.S)
    names = set(re.findall(r'extern "C" int (\w+)\(', compat))
    assert names == {"init", "detect_image", "detect_mat", "dispose", "get_device_count", "get_device_name"}
    for name in names:
        assert hasattr(lib, name), "libbetapose_hip.so does not export %s" % name


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from betapose_amd.darknet import Darknet
    net = Darknet("yolo/cfg/yolov3-single.cfg").load_stream(helpers.yolo_stream())
    with pytest.raises(_lib.BetaposeHipError):
        net.cuda()
    with pytest.raises(_lib.BetaposeHipError):
        net(torch.zeros(1, 3, 416, 416))


def test_product_package_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under betapose_amd/ may import it."""
    pkg = os.path.join(ROOT, "betapose_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f


def test_kpd_pkl_loader_is_restricted(tmp_path, monkeypatch):
    """.pkl checkpoints: a tensor state_dict loads (counters included); a pickled module is refused unless the operator
    opts in, since unpickling it runs code from the file."""
    import torch
    ok = tmp_path / "ok.pkl"
    torch.save({"conv.weight": torch.ones(2, 3), "bn.num_batches_tracked": torch.tensor(3)}, ok)
    sd = W.load_kpd_pkl(str(ok))
    assert set(sd) == {"conv.weight", "bn.num_batches_tracked"} and sd["conv.weight"].shape == (2, 3)
    mod = tmp_path / "module.pkl"
    torch.save(torch.nn.Linear(2, 2), mod)
    monkeypatch.delenv("BP_TRUST_PKL", raising=False)
    with pytest.raises(ValueError, match="BP_TRUST_PKL"):
        W.load_kpd_pkl(str(mod))
    monkeypatch.setenv("BP_TRUST_PKL", "1")
    sd = W.load_kpd_pkl(str(mod))                       # a module: its state_dict is taken
    assert set(sd) == {"weight", "bias"}


def test_headers_compile_as_plain_c_and_as_a_reference_consumer(tmp_path):
    """include/betapose_hip.h is a C header (gcc -std=c99 -pedantic -Werror) and include/yolo_v2_class_compat.h serves
    a C++ consumer of the reference's detector library; both programs link the .so and run without a GPU."""
    import subprocess
    _lib.lib()                                                   # make sure the library is built
    libdir = os.path.join(ROOT, "betapose_amd")
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    for cc, std, src in (("gcc", "-std=c99", "c_abi_check.c"), ("g++", "-std=c++11", "compat_check.cpp")):
        exe = str(tmp_path / src.split(".")[0])
        cmd = [cc, std, "-pedantic", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
               os.path.join(ROOT, "examples", src), "-o", exe, "-L" + libdir, "-lbetapose_hip", "-Wl,-rpath," + libdir]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode == 0, (src, r.returncode, r.stdout, r.stderr)


def test_every_reference_flag_parses_with_its_default():
    """The reference's whole flag set (opt.py:9-150; names, dests and defaults recorded here) is accepted, so existing
    command lines keep parsing; defaults are the reference's."""
    from betapose_amd.opt import build_parser
    ref = {"expID": "default", "dataset": "coco", "nThreads": 40, "debug": False, "snapshot": 1, "addDPG": False,
           "netType": "hgPRM", "loadModel": None, "Continue": False, "nFeats": 256, "nClasses": 50, "nStack": 4,
           "fast_inference": True, "use_pyranet": True, "LR": 2.5e-4, "momentum": 0, "weightDecay": 0, "crit": "MSE",
           "optMethod": "rmsprop", "nEpochs": 200, "epoch": 0, "trainBatch": 40, "validBatch": 20, "trainIters": 0,
           "valIters": 0, "init": None, "inputResH": 320, "inputResW": 256, "outputResH": 80, "outputResW": 64,
           "scale": 0.25, "rotate": 30, "hmGauss": 1, "baseWidth": 9, "cardinality": 5, "nResidual": 1, "dist": 1,
           "backend": "gloo", "port": None, "demo_net": "res152", "inputpath": "", "inputlist": "", "mode": "normal",
           "outputpath": "examples/res/", "inp_dim": "416", "confidence": 0.01, "nms_thesh": 0.6, "save_img": False,
           "vis": False, "profile": False, "format": None, "detbatch": 1, "posebatch": 80, "video": "", "webcam": "0",
           "save_video": False, "vis_fast": False, "sp": False, "obj_id": 5, "left_keypoints": 10}
    ns = build_parser().parse_args([])
    for k, v in ref.items():
        assert hasattr(ns, k), k
        assert getattr(ns, k) == v, (k, getattr(ns, k), v)
    ns = build_parser().parse_args("--nClasses 50 --indir in --outdir out --sp --profile --conf 0.9 --obj_id 9 "
                                   "--nThreads 4 --vis_fast --net res50 --webcam 1".split())
    assert (ns.inputpath, ns.outputpath, ns.confidence, ns.obj_id, ns.demo_net, ns.webcam) == ("in", "out", 0.9, 9, "res50", "1")
