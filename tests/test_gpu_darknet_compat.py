"""The Darknet-API-compatible detector (bp_darknet_*, and the six yolo_v2_class symbols) against the reference's OWN
compiled Darknet-C (oracle/_ref) running the same chain -- load_image / resize_image / network_predict /
get_network_boxes / do_nms_sort -- on the same cfg ([net] block), .weights file and PNG frame."""
import ctypes as C
import os

import numpy as np
import pytest
from PIL import Image

pytestmark = pytest.mark.gpu

import helpers  # noqa: E402
from betapose_amd import _lib, cfg as CFG, weights as W  # noqa: E402
from betapose_amd.darknet_compat import BBox, DarknetDetector  # noqa: E402
from oracle import darknet_c_ref  # noqa: E402

needs_ref = pytest.mark.skipif(not darknet_c_ref.available(), reason="oracle/_ref/libdarknet_ref.so not built")


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("dk")
    cfg = d / "yolov3-single.cfg"
    cfg.write_text(darknet_c_ref.NET_BLOCK % (416, 416) + CFG.yolov3_single_cfg_text())
    wts = d / "01.weights"
    W.write_darknet_weights(str(wts), helpers.yolo_stream())
    pngs = []
    for i, fr in enumerate(helpers.frames(2)):
        p = d / ("%04d.png" % i)
        Image.fromarray(fr[:, :, ::-1].copy()).save(p)
        pngs.append(str(p))
    return str(cfg), str(wts), pngs


def _match(got, ref, tol_px=1, tol_p=2e-3, ordered=True):
    """Same detections: equal counts, and a one-to-one pairing with boxes within 1 px, probabilities within 2e-3, same
    class.  The list ORDER is by probability and may differ between candidates whose probabilities agree to ~1e-7
    (the two implementations differ by fp32 rounding), so the pairing is order-free; the order itself is checked to be
    non-increasing in probability up to that noise."""
    assert len(got) == len(ref), (len(got), len(ref), got[:3], ref[:3])
    g = np.array([d[:4] for d in got], dtype=np.int64).reshape(-1, 4)
    gp = np.array([d[4] for d in got])
    used = np.zeros(len(got), bool)
    for r in ref:
        ok = (~used) & (np.abs(g - np.array(r[:4])).max(axis=1) <= tol_px) & (np.abs(gp - r[4]) <= tol_p)
        idx = np.nonzero(ok)[0]
        assert len(idx), ("unmatched reference detection", r)
        best = idx[np.argmin(np.abs(gp[idx] - r[4]))]
        assert got[best][5] == r[5]
        used[best] = True
    assert used.all()
    if ordered:          # with NMS the list is sorted by probability; without, it stays in candidate order
        assert bool((np.diff(gp) <= 1e-5).all())
    else:                # same candidate order: head -> cell -> anchor (yolo_layer.c:371-389)
        assert all(max(abs(int(a) - int(b)) for a, b in zip(x[:4], y[:4])) <= tol_px for x, y in zip(got, ref))


@needs_ref
def test_detect_matches_reference_darknet_c(cuda, files):
    cfg, wts, pngs = files
    ref = darknet_c_ref.DarknetC(CFG.yolov3_single_cfg_text(), wts, 416)
    det = DarknetDetector(cfg, wts)
    assert (det.width, det.height, det.classes) == (416, 416, 1)
    for path in pngs:
        im = ref.load_image(path)                               # the reference's own stb decode
        assert im.shape == (3, 480, 640)
        for thresh, nms in ((0.05, 0.4), (0.02, 0.4), (0.05, 0.0), (0.2, 0.4)):
            want = ref.detect(im, thresh, nms)
            got = det.detect(im, thresh, nms, cap=20000)
            _match(got, want, ordered=nms > 0)
            got_file = det.detect_file(path, thresh, nms, cap=20000)       # own PNG decode + /255
            _match(got_file, want, ordered=nms > 0)
        assert len(ref.detect(im, 0.02, 0.0)) > len(ref.detect(im, 0.02, 0.4)) > 0   # the NMS actually prunes here
    # JPEG and BMP files: the detector's own decoders (csrc/jpeg_bmp.cpp) against the reference's stb loader + Darknet-C
    from PIL import Image
    import os
    for ext, kw in ((".jpg", {"quality": 85, "subsampling": 2}), (".jpg", {"quality": 92, "subsampling": 0}), (".bmp", {})):
        path = os.path.splitext(pngs[0])[0] + "_x" + str(kw.get("subsampling", 9)) + ext
        Image.open(pngs[0]).convert("RGB").save(path, **kw)
        im = ref.load_image(path)
        for thresh in (0.2, 0.1):       # list lengths are compared exactly: stay clear of candidates sitting on a threshold
            want = ref.detect(im, thresh, 0.4)
            if min(abs(d[4] - thresh) for d in ref.detect(im, thresh * 0.5, 0.0)) < 1e-4:
                continue
            _match(det.detect_file(path, thresh, 0.4, cap=20000), want)
    # network-sized input: no resize branch (yolo_v2_class.cpp:263-266)
    small = np.ascontiguousarray(ref.load_image(pngs[0])[:, :416, :416])
    _match(det.detect(small, 0.05, 0.4, cap=20000), ref.detect(small, 0.05, 0.4))
    det.close()


def test_yolo_v2_class_symbols(cuda, files):
    """init / detect_image / detect_mat / dispose / get_device_count / get_device_name as a C# or C++ caller of the
    reference's library would use them (ctypes passes the container by pointer, which is what a C++ reference is)."""
    cfg, wts, pngs = files
    L = _lib.lib()

    class Container(C.Structure):
        _fields_ = [("candidates", BBox * 1000)]

    for name in ("init", "detect_image", "detect_mat", "dispose", "get_device_count", "get_device_name"):
        assert hasattr(L, name), name
    assert L.get_device_count() >= 1
    buf = C.create_string_buffer(256)
    assert L.get_device_name(0, buf) == 1 and b"gfx" in buf.value
    box = Container()
    L.detect_image.argtypes = [C.c_char_p, C.POINTER(Container)]
    L.detect_mat.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Container)]
    assert L.detect_image(pngs[0].encode(), C.byref(box)) < 0            # before init(): error, not a crash
    assert L.init(cfg.encode(), wts.encode(), 0) == 1
    n = L.detect_image(pngs[0].encode(), C.byref(box))
    assert n >= 0
    det = DarknetDetector(cfg, wts)
    want = det.detect_file(pngs[0], 0.2, 0.4)                              # the library's defaults: thresh 0.2, nms 0.4
    assert n == len(want)
    for i in range(n):
        b = box.candidates[i]
        assert (b.x, b.y, b.w, b.h, b.obj_id) == (want[i][0], want[i][1], want[i][2], want[i][3], want[i][5])
    data = open(pngs[0], "rb").read()
    assert L.detect_mat(data, len(data), C.byref(box)) == n
    assert L.detect_mat(b"not an image", 12, C.byref(box)) < 0
    assert L.detect_image(b"/nonexistent.png", C.byref(box)) < 0
    assert L.dispose() == 1
    assert L.detect_image(pngs[0].encode(), C.byref(box)) < 0
    det.close()


def test_create_errors(cuda, files, tmp_path):
    cfg, wts, _ = files
    bare = tmp_path / "bare.cfg"
    bare.write_text(CFG.yolov3_single_cfg_text())                          # no [net] block
    with pytest.raises(_lib.BetaposeHipError, match=r"\[net\]"):
        DarknetDetector(str(bare), wts)
    with pytest.raises(_lib.BetaposeHipError, match="cannot open"):
        DarknetDetector(cfg, str(tmp_path / "missing.weights"))
