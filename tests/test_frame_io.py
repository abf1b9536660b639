"""Frame input: the native PNG decoder must return exactly what the reference's ``cv2.imread`` does
for LineMod frames -- BGR u8 of the stored pixels (PNG is lossless, so PIL's decode of the same file is the oracle)
-- for every colour type / filter the format allows, and the threaded loader must deliver frames in order, with
back-pressure, and report broken files.  Host only."""
import io
import os
import zlib
import struct

import numpy as np
import pytest
from PIL import Image

from betapose_amd import _lib, synth
from betapose_amd.frame_loader import FrameLoader, decode_png, png_size


def _png_bytes(img, **kw):
    b = io.BytesIO()
    img.save(b, format="PNG", **kw)
    return b.getvalue()


def _bgr(img):
    return np.asarray(img.convert("RGB"))[:, :, ::-1]


def test_rgb_frame_matches_pil_exactly():
    fr = synth.synth_frame(1234)                      # BGR u8 480x640x3
    data = _png_bytes(Image.fromarray(fr[:, :, ::-1].copy()))
    out = decode_png(data)
    assert out.dtype == np.uint8 and out.shape == (480, 640, 3)
    np.testing.assert_array_equal(out, fr)
    for level in (0, 1, 9):                           # stored / fast / best deflate, different IDAT splits
        np.testing.assert_array_equal(decode_png(_png_bytes(Image.fromarray(fr[:, :, ::-1].copy()), compress_level=level)), fr)


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "LA", "P", "1"])
def test_colour_types(mode):
    rng = np.random.default_rng(5)
    base = Image.fromarray(rng.integers(0, 256, (37, 53, 3), dtype=np.uint8))   # odd sizes: partial bytes at depth 1
    img = base.convert(mode)
    np.testing.assert_array_equal(decode_png(_png_bytes(img)), _bgr(img))


def _raw_png(w, h, depth, ctype, rows, filters, plte=None):
    """Hand-assembled PNG with a chosen filter type per row (PIL picks filters itself)."""
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    raw = b"".join(bytes([f]) + r for f, r in zip(filters, rows))
    comp = zlib.compress(raw, 6)
    mid = len(comp) // 2
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
    if plte is not None:
        out += chunk(b"PLTE", plte)
    return out + chunk(b"tEXt", b"k\0v") + chunk(b"IDAT", comp[:mid]) + chunk(b"IDAT", comp[mid:]) + chunk(b"IEND", b"")


def _filter_rows(img, bpp, ftypes):
    """Apply the PNG forward filters to an [h, rowbytes] u8 array."""
    h, rb = img.shape
    rows = []
    for y in range(h):
        cur = img[y].astype(np.int32)
        up = img[y - 1].astype(np.int32) if y else np.zeros(rb, np.int32)
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        ul = np.concatenate([np.zeros(bpp, np.int32), up[:-bpp]])
        f = ftypes[y]
        if f == 0: pred = 0
        elif f == 1: pred = left
        elif f == 2: pred = up
        elif f == 3: pred = (left + up) // 2
        else:
            p = left + up - ul
            pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
        rows.append(((cur - pred) & 255).astype(np.uint8).tobytes())
    return rows


def test_every_filter_type_and_16_bit():
    rng = np.random.default_rng(9)
    h, w = 10, 17
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    ft = [0, 1, 2, 3, 4, 4, 3, 2, 1, 0]
    data = _raw_png(w, h, 8, 2, _filter_rows(rgb.reshape(h, -1), 3, ft), ft)
    np.testing.assert_array_equal(decode_png(data), rgb[:, :, ::-1])
    np.testing.assert_array_equal(_bgr(Image.open(io.BytesIO(data))), rgb[:, :, ::-1])    # the file itself is valid
    # 16-bit RGBA: cv2 keeps the high byte and drops alpha
    rgba16 = rng.integers(0, 65536, (h, w, 4), dtype=np.uint16)
    be = rgba16.astype(">u2").view(np.uint8).reshape(h, -1)
    data = _raw_png(w, h, 16, 6, _filter_rows(be, 8, ft), ft)
    np.testing.assert_array_equal(decode_png(data), (rgba16[:, :, :3] >> 8).astype(np.uint8)[:, :, ::-1])
    # 2-bit grey, 4-bit palette
    g2 = rng.integers(0, 4, (h, w), dtype=np.uint8)
    packed = np.zeros((h, (w * 2 + 7) // 8), np.uint8)
    for x in range(w):
        packed[:, x // 4] |= g2[:, x] << (6 - 2 * (x % 4))
    data = _raw_png(w, h, 2, 0, _filter_rows(packed, 1, [0] * h), [0] * h)
    np.testing.assert_array_equal(decode_png(data), np.repeat((g2 * 85)[:, :, None], 3, 2))
    p4 = rng.integers(0, 16, (h, w), dtype=np.uint8)
    packed = np.zeros((h, (w * 4 + 7) // 8), np.uint8)
    for x in range(w):
        packed[:, x // 2] |= p4[:, x] << (4 - 4 * (x % 2))
    pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    data = _raw_png(w, h, 4, 3, _filter_rows(packed, 1, [1] * h), [1] * h, plte=pal.tobytes())
    np.testing.assert_array_equal(decode_png(data), pal[p4][:, :, ::-1])


def test_malformed_files_are_errors_not_crashes():
    good = _png_bytes(Image.fromarray(np.zeros((8, 8, 3), np.uint8)))
    for bad in (b"", b"not a png at all", good[:20], good[:len(good) // 2], good[:-12],
                good.replace(b"IHDR", b"IHDX"), good[:24] + bytes([3]) + good[25:]):       # bit depth 3
        with pytest.raises(_lib.BetaposeHipError):
            decode_png(bad)
    inter = _png_bytes(Image.fromarray(np.zeros((8, 8, 3), np.uint8)))
    inter = inter[:28] + b"\x01" + inter[29:]                                                 # interlace flag
    with pytest.raises(_lib.BetaposeHipError, match="interlaced"):
        decode_png(inter)
    corrupt = bytearray(_png_bytes(Image.fromarray(np.random.default_rng(0).integers(0, 256, (32, 32, 3), dtype=np.uint8))))
    i = corrupt.index(b"IDAT") + 12
    corrupt[i:i + 8] = b"\xff" * 8
    with pytest.raises(_lib.BetaposeHipError):
        decode_png(bytes(corrupt))


def test_loader_order_backpressure_and_errors(tmp_path):
    frames = synth.synth_frames(12, 77)
    paths = []
    for i, fr in enumerate(frames):
        p = tmp_path / ("%04d.png" % i)
        Image.fromarray(fr[:, :, ::-1].copy()).save(p, compress_level=1)
        paths.append(str(p))
    assert png_size(paths[0]) == (480, 640)
    ld = FrameLoader(paths, threads=3, depth=4, pinned=False)          # depth < len: slots are recycled
    assert (ld.height, ld.width, len(ld)) == (480, 640, 12)
    held = []
    for idx, view, addr in ld:
        np.testing.assert_array_equal(view, frames[idx])
        assert addr == view.ctypes.data
        held.append(idx)
        if len(held) == 3:                                               # hold 3 of the 4 slots, then let go
            for j in held:
                ld.release(j)
            held = []
    for j in held:
        ld.release(j)
    ld.close()

    # a wrong-size frame and a broken file are reported for exactly that frame; the rest still arrives
    Image.fromarray(np.zeros((100, 100, 3), np.uint8)).save(tmp_path / "small.png")
    open(tmp_path / "broken.png", "wb").write(open(paths[1], "rb").read()[:5000])
    mixed = [paths[0], str(tmp_path / "small.png"), paths[2], str(tmp_path / "broken.png"), str(tmp_path / "missing.png"),
             paths[3]]
    ld = FrameLoader(mixed, height=480, width=640, threads=2, depth=3, pinned=False)
    it, got, errs = iter(ld), [], []
    while True:
        try:
            idx, view, _ = next(it)
            got.append(idx)
            ld.release(idx)
        except StopIteration:
            break
        except _lib.BetaposeHipError as e:
            errs.append(str(e))
            it = iter(ld)
    assert got == [0, 2, 5] and len(errs) == 3
    assert "small.png" in errs[0] and "expected 640x480" in errs[0]
    assert "broken.png" in errs[1] and "missing.png" in errs[2]
    ld.close()


def test_loader_non_png_inputs_go_through_pil(tmp_path):
    fr = synth.synth_frame(5)
    p = tmp_path / "a.jpg"
    Image.fromarray(fr[:, :, ::-1].copy()).save(p, quality=95)
    ld = FrameLoader([str(p), str(p)], threads=2, depth=2)
    out = [(i, v.copy()) for i, v, _ in ld]
    assert [i for i, _ in out] == [0, 1] and out[0][1].shape == (480, 640, 3)
    np.testing.assert_array_equal(out[0][1], _bgr(Image.open(p)))
    ld.close()
