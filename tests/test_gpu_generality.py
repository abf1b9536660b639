"""Generality and error behaviour of the engine through the C-ABI: another cfg (the reference's Darknet class ran
it: tests/golden/formats.npz), other input sizes / class counts / batch limits, malformed inputs."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import helpers  # noqa: E402
from betapose_amd import _lib, cfg as Cfg, ops, synth, weights as W  # noqa: E402
from betapose_amd.darknet import Darknet  # noqa: E402
from betapose_amd.kpd import FastPoseHIP  # noqa: E402
from oracle import kpd_ref, yolo_ref  # noqa: E402


def test_tiny_cfg_matches_reference_darknet(tmp_path, cuda):
    """5-conv cfg with a shortcut and one yolo head, batch 2, 64x64 input: output of the REFERENCE's Darknet class."""
    fmt = helpers.golden("formats.npz")
    cfg_path, w_path = tmp_path / "tiny.cfg", tmp_path / "tiny.weights"
    cfg_path.write_text(str(fmt["tiny_cfg"]))
    w_path.write_bytes(fmt["tiny_weights_bytes"].tobytes())
    net = Darknet(str(cfg_path), reso=64, max_batch=2).load_weights(str(w_path)).cuda()
    assert net.seen == 7
    out = net(torch.from_numpy(fmt["tiny_in"])).cpu().numpy()
    ref = fmt["tiny_out"]
    assert out.shape == ref.shape == (2, 3 * 32 * 32, 6)
    assert bool((np.abs(out[..., :4] - ref[..., :4]) <= 2e-3 + 3e-5 * np.abs(ref[..., :4])).all())
    assert np.abs(out[..., 4:] - ref[..., 4:]).max() <= 2e-5
    # the C entry point that reads cfg + .weights itself (yolo_v2_class-style init) gives the same engine
    h = C.c_void_p()
    _lib.check(_lib.lib().bp_yolo_create(str(cfg_path).encode(), str(w_path).encode(), 64, 2, 0, C.byref(h)))
    x = torch.from_numpy(fmt["tiny_in"]).cuda()
    pred = torch.empty((2, _lib.lib().bp_yolo_rows(h), _lib.lib().bp_yolo_attrs(h)), device="cuda")
    _lib.check(_lib.lib().bp_yolo_forward(h, x.data_ptr(), 2, pred.data_ptr(), _lib.current_stream()))
    torch.cuda.synchronize()
    assert np.array_equal(pred.cpu().numpy(), out)
    _lib.lib().bp_yolo_destroy(h)


def test_cfg_with_net_block_and_multi_class(cuda):
    """Darknet-C style cfg ([net] first) with 3 classes at reso 320 against the oracle."""
    text = "[net]\nwidth=320\nheight=320\nchannels=3\n\n" + Cfg.yolov3_single_cfg_text(classes=3)
    blocks = [b for b in Cfg.parse_cfg_text(text) if b["type"] != "net"]
    stream = synth.synth_yolo_stream(9, blocks)
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".cfg", delete=False) as f:
        f.write(text)
    net = Darknet(f.name, reso=320, max_batch=1)
    net.blocks = [b for b in net.blocks if b["type"] != "net"]
    net.net_info = net.blocks[0]
    net.load_stream(stream).cuda()
    x = torch.rand(1, 3, 320, 320, generator=torch.Generator().manual_seed(4))
    out = net(x).cpu()
    ref = yolo_ref.darknet_forward(blocks, W.split_darknet_stream(blocks, stream), x, reso=320)
    assert out.shape == ref.shape == (1, 3 * (10 * 10 + 20 * 20 + 40 * 40), 8)
    assert bool(((out[..., :4] - ref[..., :4]).abs() <= 2e-3 + 3e-5 * ref[..., :4].abs()).all())
    assert float((out[..., 4:] - ref[..., 4:]).abs().max()) <= 2e-5
    os.unlink(f.name)


def test_kpd_more_classes_is_narrowed_to_50(cuda):
    """nClasses > 50: InferenNet_fast keeps the first 50 maps (main_fast_inference.py:44)."""
    sd = synth.synth_fastpose_state_dict(5, n_classes=60)
    net = FastPoseHIP(sd, n_classes=60).cuda()
    x = torch.rand(1, 3, 320, 256, generator=torch.Generator().manual_seed(1)) - 0.45
    hm = net(x).cpu()
    ref = kpd_ref.fastpose_forward(sd, x, n_keep=50)
    assert hm.shape == ref.shape == (1, 50, 80, 64)
    assert float((hm - ref).abs().max()) <= 2e-4


def test_errors_are_reported_not_fatal(tmp_path, cuda):
    L = _lib.lib()
    h = C.c_void_p()
    assert L.bp_yolo_create(b"/nonexistent.cfg", b"/nonexistent.weights", 416, 1, 0, C.byref(h)) != 0
    assert b"cannot open" in L.bp_last_error()
    cfg = tmp_path / "c.cfg"
    cfg.write_text(Cfg.yolov3_single_cfg_text())
    short = tmp_path / "short.weights"
    W.write_darknet_weights(str(short), np.zeros(1000, np.float32))
    assert L.bp_yolo_create(str(cfg).encode(), str(short).encode(), 416, 1, 0, C.byref(h)) != 0
    assert b"too short" in L.bp_last_error()
    assert L.bp_yolo_create(str(cfg).encode(), str(short).encode(), 400, 1, 0, C.byref(h)) != 0   # reso % 32
    bad = tmp_path / "bad.cfg"
    bad.write_text("[maxpool]\nsize=2\nstride=2\n")
    assert L.bp_yolo_create_from_memory(bad.read_text().encode(), np.zeros(4, np.float32).ctypes.data, 4, 416, 1, 0,
                                        C.byref(h)) != 0
    assert b"unsupported cfg block" in L.bp_last_error()
    k = C.c_void_p()
    assert L.bp_kpd_create(np.zeros(10, np.float32).ctypes.data, 10, 50, 1, 0, C.byref(k)) != 0
    # python-level argument checks
    net = Darknet("yolo/cfg/yolov3-single.cfg", max_batch=1).load_stream(helpers.yolo_stream()).cuda()
    with pytest.raises(ValueError):
        net(torch.zeros(2, 3, 416, 416))          # batch > max_batch
    with pytest.raises(ValueError):
        net(torch.zeros(1, 3, 320, 320))          # wrong resolution
    with pytest.raises(ValueError):
        Darknet("yolo/cfg/yolov3-single.cfg").load_stream(np.zeros(10, np.float32))


@pytest.mark.parametrize("hw", [(240, 320), (480, 640), (1080, 1920), (97, 131)])
def test_crop_and_resize_other_frame_sizes(cuda, hw):
    from PIL import Image
    from oracle import post_ref
    H, Wd = hw
    rng = np.random.Generator(np.random.PCG64(H))
    frame = rng.integers(0, 256, (H, Wd, 3), dtype=np.uint8)
    got = ops.resize_bicubic(torch.from_numpy(frame[None]).to(cuda), 416, 416, swap_rb=False, want="u8").cpu().numpy()[0]
    assert np.array_equal(got, np.asarray(Image.fromarray(frame).resize((416, 416), 3)))
    box = torch.tensor([[0.2 * Wd, 0.1 * H, 0.7 * Wd, 0.9 * H]])
    ref, pt1, pt2 = post_ref.crop_from_dets_frame(frame, box)
    out, pts = ops.crop(torch.from_numpy(frame[None]).to(cuda), boxes=box.to(cuda))
    np.testing.assert_array_equal(pts.cpu().numpy()[0, :4], np.r_[pt1.numpy()[0], pt2.numpy()[0]])
    assert float((out.cpu() - ref).abs().max()) <= 2e-6


def test_masked_stream_confines_work_to_its_cu_slice(cuda):
    """hipExtStreamCreateWithCUMask: mask bit i = CU i//8 of XCD i%8; a slice of k CUs per XCD gives 8k places and the
    workgroups still go round-robin to all 8 XCDs."""
    import torch
    from betapose_amd.streams import MaskedStream, partition_cus, slice_mask_words
    assert partition_cus(4) == [(0, 8), (8, 16), (16, 24), (24, 32)]
    w = slice_mask_words(8, 16)
    assert w.tolist() == [0, 0, 0xFFFFFFFF, 0xFFFFFFFF, 0, 0, 0, 0]
    seen = []
    for lo, hi in partition_cus(4):
        st = MaskedStream(lo, hi)
        x, hw = st.probe(4096)
        assert sorted(set(x.tolist())) == list(range(8))
        places = set(zip(x.tolist(), ((hw >> 13) & 7).tolist(), ((hw >> 8) & 15).tolist()))
        assert len(places) == 64 == st.n_cus
        seen.append(places)
        # the stream works as a torch stream
        with torch.cuda.stream(st.torch):
            y = torch.arange(1000, device="cuda").float().sum()
        st.torch.synchronize()
        assert float(y) == 499500.0
        st.close()
    for i in range(4):
        for j in range(i):
            assert not (seen[i] & seen[j])          # slices do not overlap


def test_engines_of_several_objects_coexist(cuda):
    """BASELINE configs[2] serves all LineMod objects: one detector + key-point engine per object, each with its own
    filters, side by side in one process (no hidden globals).  Three differently seeded KPD engines at batch 28 in the
    fp16 mode, called interleaved on two streams, must each return what they return when run alone."""
    import torch
    from betapose_amd import synth
    from betapose_amd.kpd import FastPoseHIP
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(28, 3, 320, 256, generator=g) - 0.45).to(cuda)
    engines, alone = [], []
    for obj, seed in enumerate((2, 12, 22)):
        e = FastPoseHIP(synth.synth_fastpose_state_dict(seed), n_classes=50, max_batch=28).cuda()
        e.set_precision("f16")
        engines.append(e)
        alone.append(e(x).clone())
    torch.cuda.synchronize()
    assert float((alone[0] - alone[1]).abs().max()) > 1e-2          # different objects really differ
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = [None] * 3
    for rep in range(2):
        for i, e in enumerate(engines):
            with torch.cuda.stream(s1 if (i + rep) % 2 else s2):
                outs[i] = e(x).clone()
    torch.cuda.synchronize()
    for i in range(3):
        assert torch.equal(outs[i], alone[i]), "object %d" % i


def test_inferennet_fast_loads_the_reference_pkl_layout(tmp_path, cuda, monkeypatch):
    """``InferenNet_fast(kernel_size, obj_id, dataset)`` as the harness constructs it (betapose_evaluate.py:124-130):
    reads ``./exp/final_model/<name of obj_id>.pkl`` = ``torch.save(model.state_dict())`` of the reference's FastPose,
    i.e. torch tensors including the 106 ``num_batches_tracked`` counters, which must be ignored."""
    from betapose_amd.kpd import ALLPATHS, InferenNet_fast
    sd = helpers.kpd_state_dict()
    full = {}
    for k, v in sd.items():
        full[k] = torch.from_numpy(np.asarray(v))
        if k.endswith("running_var"):
            full[k[:-len("running_var")] + "num_batches_tracked"] = torch.tensor(12345, dtype=torch.long)
    assert sum(k.endswith("num_batches_tracked") for k in full) == 106
    d = tmp_path / "exp" / "final_model"
    d.mkdir(parents=True)
    torch.save(full, d / (ALLPATHS[6] + ".pkl"))
    monkeypatch.chdir(tmp_path)
    net = InferenNet_fast(4 * 1 + 1, 6, None).cuda().eval()
    x = torch.rand(1, 3, 320, 256, generator=torch.Generator().manual_seed(4)) - 0.45
    ref = FastPoseHIP(sd).cuda()(x).cpu()
    assert torch.equal(net(x).cpu(), ref)
    with pytest.raises((FileNotFoundError, OSError)):
        InferenNet_fast(5, 1, None)                      # no checkpoint for object 1 in this directory


@pytest.mark.parametrize("bad", [
    "[convolutional]\nfilters=8\nsize=3\nstride=1\npad=1\nactivation=leaky\n[shortcut]\nfrom=-5\nactivation=linear\n",
    "[yolo]\nmask=0,1,2\nanchors=1,2,3,4,5,6\nclasses=1\n",
    "[convolutional]\nfilters=18\nsize=1\nstride=1\nactivation=linear\n[yolo]\nmask=0,-1,2\nanchors=1,2,3,4,5,6\nclasses=1\n",
    "[convolutional]\nfilters=8\nsize=0\nstride=1\nactivation=leaky\n",
])
def test_malformed_cfg_is_a_clean_error(cuda, bad):
    """Out-of-range shortcut sources, a [yolo] at layer 0, negative anchor masks: bp::Error through the C-ABI, no
    out-of-bounds indexing on the host."""
    h = C.c_void_p()
    stream = np.zeros(4096, np.float32)
    rc = _lib.lib().bp_yolo_create_from_memory(bad.encode(), stream.ctypes.data, stream.size, 64, 1, 0, C.byref(h))
    assert rc != 0 and _lib.lib().bp_last_error()
