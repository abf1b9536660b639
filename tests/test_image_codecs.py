"""JPEG / BMP decoders of the Darknet-API-compatible detector (csrc/jpeg_bmp.cpp) against the REFERENCE's own image
loader: Darknet-C's load_image_color -> stb_image v2.16, compiled from /root/reference into oracle/_ref (travels to the
GPU box as a built .so).  Bit-exact pixels for every sampling layout Pillow can write.  CPU only."""
import ctypes as C
import io

import numpy as np
import pytest
from PIL import Image

from betapose_amd import _lib, synth
from betapose_amd.darknet_compat import _check
from oracle import darknet_c_ref

pytestmark = pytest.mark.skipif(not darknet_c_ref.available(), reason="oracle/_ref (reference Darknet-C) not built")


def _decode(data: bytes) -> np.ndarray:
    h, w = C.c_int(), C.c_int()
    buf = (C.c_ubyte * len(data)).from_buffer_copy(data)
    _check(_lib.lib().bp_image_decode_rgb(buf, len(data), None, 0, C.byref(h), C.byref(w)))
    out = np.empty((h.value, w.value, 3), np.uint8)
    _check(_lib.lib().bp_image_decode_rgb(buf, len(data), out.ctypes.data, out.nbytes, C.byref(h), C.byref(w)))
    return out


def _stb(path) -> np.ndarray:
    """The reference's load_image_color: planar float RGB / 255 -> interleaved u8."""
    im = darknet_c_ref.load_image_color(str(path))
    return np.rint(im.transpose(1, 2, 0) * 255.0).astype(np.uint8)


def _pictures():
    rng = np.random.default_rng(3)
    yield "frame", synth.synth_frame(5)[:, :, ::-1].copy()                       # 480 x 640 synthetic frame
    yield "odd", rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)             # sizes that are no multiple of the MCU
    yy, xx = np.mgrid[0:95, 0:131]
    yield "smooth", np.stack([(xx * 2) % 256, (yy * 3) % 256, (xx + yy) % 256], -1).astype(np.uint8)
    yield "tiny", rng.integers(0, 256, (1, 1, 3), dtype=np.uint8)
    yield "row", rng.integers(0, 256, (3, 17, 3), dtype=np.uint8)


@pytest.mark.parametrize("subsampling", [0, 1, 2])          # 4:4:4, 4:2:2, 4:2:0
@pytest.mark.parametrize("quality", [35, 75, 95])
def test_baseline_jpeg_pixels_equal_the_reference_loader(tmp_path, subsampling, quality):
    for name, rgb in _pictures():
        p = tmp_path / ("%s_%d_%d.jpg" % (name, subsampling, quality))
        Image.fromarray(rgb).save(p, format="JPEG", quality=quality, subsampling=subsampling, optimize=(quality == 35))
        got, ref = _decode(p.read_bytes()), _stb(p)
        assert got.shape == ref.shape == rgb.shape
        assert np.array_equal(got, ref), (name, int(np.abs(got.astype(int) - ref.astype(int)).max()))


def test_grey_jpeg_restart_intervals_and_bmp(tmp_path):
    rgb = synth.synth_frame(9)[:, :, ::-1].copy()
    g = tmp_path / "grey.jpg"
    Image.fromarray(rgb).convert("L").save(g, format="JPEG", quality=80)
    assert np.array_equal(_decode(g.read_bytes()), _stb(g))
    # restart markers: rewrite a 4:2:0 stream with DRI = 5 MCUs through Pillow when it can, else skip that part
    r = tmp_path / "rst.jpg"
    try:
        Image.fromarray(rgb).save(r, format="JPEG", quality=70, subsampling=2, restart_marker_blocks=5)
        assert b"\xff\xdd" in r.read_bytes()
        assert np.array_equal(_decode(r.read_bytes()), _stb(r))
    except TypeError:
        pass
    for mode in ("RGB", "P", "RGBA"):
        b = tmp_path / ("x_%s.bmp" % mode)
        Image.fromarray(rgb[:61, :83]).convert(mode).save(b, format="BMP")
        assert np.array_equal(_decode(b.read_bytes()), _stb(b)), mode


@pytest.mark.parametrize("subsampling", [0, 1, 2])
@pytest.mark.parametrize("quality", [35, 75, 95])
def test_progressive_jpeg_pixels_equal_the_reference_loader(tmp_path, subsampling, quality):
    """SOF2 streams as libjpeg writes them: interleaved DC first scan, per-component AC bands, successive-approximation
    refinement scans with end-of-band runs (stb_image decodes these, train_YOLO/src/image.c:1820)."""
    for name, rgb in _pictures():
        p = tmp_path / ("%s_%d_%d_p.jpg" % (name, subsampling, quality))
        Image.fromarray(rgb).save(p, format="JPEG", quality=quality, subsampling=subsampling, progressive=True)
        data = p.read_bytes()
        assert b"\xff\xc2" in data and data.count(b"\xff\xda") > 3           # really progressive, several scans
        got, ref = _decode(data), _stb(p)
        assert got.shape == ref.shape == rgb.shape
        assert np.array_equal(got, ref), (name, int(np.abs(got.astype(int) - ref.astype(int)).max()))


def test_progressive_grey_restarts_and_flat_pictures(tmp_path):
    rgb = synth.synth_frame(4)[:, :, ::-1].copy()
    g = tmp_path / "grey_p.jpg"
    Image.fromarray(rgb).convert("L").save(g, format="JPEG", quality=85, progressive=True)
    assert np.array_equal(_decode(g.read_bytes()), _stb(g))
    flat = np.full((40, 72, 3), 117, np.uint8)                                 # long end-of-band runs, no AC at all
    flat[8:24, 16:40] = (250, 3, 90)
    f = tmp_path / "flat_p.jpg"
    Image.fromarray(flat).save(f, format="JPEG", quality=90, progressive=True, subsampling=2)
    assert np.array_equal(_decode(f.read_bytes()), _stb(f))
    r = tmp_path / "rst_p.jpg"
    try:
        Image.fromarray(rgb).save(r, format="JPEG", quality=70, subsampling=2, progressive=True, restart_marker_blocks=3)
    except TypeError:
        return
    assert b"\xff\xdd" in r.read_bytes()
    assert np.array_equal(_decode(r.read_bytes()), _stb(r))


def test_unsupported_streams_fail_loudly(tmp_path):
    rgb = synth.synth_frame(2)[:64, :64, ::-1].copy()
    buf = io.BytesIO()
    Image.fromarray(rgb).save(buf, format="JPEG", progressive=True)
    with pytest.raises(_lib.BetaposeHipError, match="unsupported image format"):
        _decode(b"GIF89a" + b"\0" * 64)
    with pytest.raises(_lib.BetaposeHipError):
        _decode(buf.getvalue()[:200])
    arith = bytearray(buf.getvalue())
    i = arith.index(b"\xff\xc2")
    arith[i + 1] = 0xCA                                                        # progressive, arithmetic-coded
    with pytest.raises(_lib.BetaposeHipError, match="arithmetic"):
        _decode(bytes(arith))


def test_malformed_jpegs_are_errors_not_overreads(tmp_path):
    """Truncated segments and mutated streams must come back as clean errors (every fixed-size field is length-checked
    before it is read; a DC category above 11 is rejected)."""
    def rc_of(data: bytes) -> int:
        h, w = C.c_int(), C.c_int()
        buf = (C.c_ubyte * max(1, len(data))).from_buffer_copy(data if data else b"\0")
        out = np.empty(1 << 20, np.uint8)
        return _lib.lib().bp_image_decode_rgb(buf, len(data), out.ctypes.data, out.nbytes, C.byref(h), C.byref(w))
    assert rc_of(bytes.fromhex("ffd8ffc4000300")) != 0                 # DHT of 1 byte (the advisor's ASAN repro)
    assert rc_of(bytes.fromhex("ffd8ffdb000300")) != 0                 # DQT without its table
    assert rc_of(bytes.fromhex("ffd8ffc000030800")) != 0               # SOF cut before the component count
    assert rc_of(bytes.fromhex("ffd8ffdd000200")) != 0                 # DRI without its interval
    rgb = synth.synth_frame(11)[:120, :160, ::-1].copy()
    p = tmp_path / "ok.jpg"
    Image.fromarray(rgb).save(p, format="JPEG", quality=70)
    good = p.read_bytes()
    assert rc_of(good) == 0
    pp = tmp_path / "ok_p.jpg"
    Image.fromarray(rgb).save(pp, format="JPEG", quality=70, progressive=True)
    rng = np.random.default_rng(0)
    for good in (good, pp.read_bytes()):
        assert rc_of(good) == 0
        for cut in (3, 5, 20, 100, 180, len(good) // 2):
            rc_of(good[:cut])                                          # any return code; must not crash
        for _ in range(300):
            bad = bytearray(good)
            for _ in range(int(rng.integers(1, 6))):
                bad[int(rng.integers(2, len(bad)))] = int(rng.integers(0, 256))
            rc_of(bytes(bad))
