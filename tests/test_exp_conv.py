"""The measured-and-superseded convolution kernels of rounds 1-2 (conv_igemm_h: fp32 activations converted in the K loop,
staged and filters-direct; conv_w64.hip; conv_kg.hip; conv_rd.hip).  They are NOT in the product library: these tests run
only against the experimental build,

    python -m betapose_amd.build --experimental
    BP_LIB=$PWD/betapose_amd/libbetapose_hip_exp.so python -m pytest tests/test_exp_conv.py -m experimental

and are deliberately not marked ``gpu`` (the product's GPU suite counts only what the product runs)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.experimental

from betapose_amd import ops  # noqa: E402
from test_gpu_conv import CASES, F16_CASES, _check, _ref  # noqa: E402


# ---- fp16-MFMA variant (BASELINE configs[2]): operands rounded to fp16 (RNE) on the way into LDS / at weight upload,
# fp32 accumulation and fp32 outputs.  Against a torch conv on the SAME rounded operands only the accumulation order
# differs, so the fp32 tolerance applies; against the unrounded fp32 conv the distance is the fp16 rounding (~1e-3).


@pytest.mark.parametrize("case", F16_CASES)
@pytest.mark.parametrize("tile", ["64x64_f16", "128x64_f16"])
@pytest.mark.parametrize("splits", [1, 3])
def test_conv_f16_operands(cuda, case, tile, splits):
    N, H, W, Cin, Cout, k, st, pad, act = case
    g = torch.Generator().manual_seed(300 + CASES.index(case))
    x = torch.randn(N, H, W, Cin, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    res = torch.randn(N, (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1, Cout, generator=g)
    ref16 = _ref(x.half().float(), w.half().float(), b, st, pad, act, res, False)
    ref32 = _ref(x, w, b, st, pad, act, res, False)
    if splits > 1 and Cin * k * k // 32 < 2 * splits:
        pytest.skip("too few K-chunks to split")
    out = ops.conv2d_nhwc(x.to(cuda), w, b, stride=st, pad=pad, act=act, res=res.to(cuda), tile=tile, splits=splits)
    out = out.cpu().permute(0, 3, 1, 2)
    _check(out, ref16)                      # same operands: accumulation order only
    assert float((out - ref32).abs().max()) < 2e-2 and float((out - ref32).abs().max()) > 1e-6   # really fp16 operands


# ---- bf16x3 variant: fp32 operands split exactly into three bf16 terms, six partial products on the bf16 MFMA.
# It must be AS ACCURATE AS the fp32-MFMA kernel: same tolerance against the fp32 conv, and its distance to an fp64
# conv must not exceed the fp32 kernel's.
@pytest.mark.parametrize("case", F16_CASES)
@pytest.mark.parametrize("splits", [1, 3])
def test_conv_bf16x3_is_fp32_accurate(cuda, case, splits):
    N, H, W, Cin, Cout, k, st, pad, act = case
    g = torch.Generator().manual_seed(500 + CASES.index(case))
    x = torch.randn(N, H, W, Cin, generator=g) * torch.exp(2 * torch.randn(N, H, W, 1, generator=g))   # wide dynamic range
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    if splits > 1 and Cin * k * k // 32 < 2 * splits:
        pytest.skip("too few K-chunks to split")
    ref64 = _ref(x.double(), w.double(), b.double(), st, pad, act, None, False)
    ref32 = ref64.float()
    out3 = ops.conv2d_nhwc(x.to(cuda), w, b, stride=st, pad=pad, act=act, tile="64x64_b3", splits=splits)
    out32 = ops.conv2d_nhwc(x.to(cuda), w, b, stride=st, pad=pad, act=act, tile="64x64", splits=splits)
    out3, out32 = out3.cpu().permute(0, 3, 1, 2), out32.cpu().permute(0, 3, 1, 2)
    scale = float(ref64.abs().mean())
    e3 = float((out3.double() - ref64).abs().max()) / scale
    e32 = float((out32.double() - ref64).abs().max()) / scale
    assert e3 <= max(1.5 * e32, 2e-6), (e3, e32)           # no less accurate than the fp32-MFMA kernel
    _check(out3, ref32, tol=2e-5 * max(1.0, scale))


# ---- conv_w64.hip: 64x64 accumulator tile per wave, WM x WN waves per block, filters by LDS-DMA.  Same bars as the
# kernels above: bf16x3 must be fp32-accurate, fp16 must match a conv on fp16-rounded operands to accumulation order.
W64_TILES = ["w1x1", "w1x2", "w2x1", "w2x2"]


@pytest.mark.parametrize("case", F16_CASES)
@pytest.mark.parametrize("tile", W64_TILES)
@pytest.mark.parametrize("splits", [1, 3])
def test_conv_w64_bf16x3_is_fp32_accurate(cuda, case, tile, splits):
    N, H, W, Cin, Cout, k, st, pad, act = case
    g = torch.Generator().manual_seed(700 + CASES.index(case))
    x = torch.randn(N, H, W, Cin, generator=g) * torch.exp(2 * torch.randn(N, H, W, 1, generator=g))   # wide dynamic range
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    if splits > 1 and Cin * k * k // 32 < 2 * splits:
        pytest.skip("too few K-chunks to split")
    OH, OW = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    res = torch.randn(N, OH, OW, Cout, generator=g)
    ref64 = _ref(x.double(), w.double(), b.double(), st, pad, act, res.double(), True)
    out3 = ops.conv2d_nhwc(x.to(cuda), w, b, stride=st, pad=pad, act=act, res=res.to(cuda), res_after_act=True,
                           tile=tile + "_b3", splits=splits)
    out32 = ops.conv2d_nhwc(x.to(cuda), w, b, stride=st, pad=pad, act=act, res=res.to(cuda), res_after_act=True,
                            tile="64x64", splits=splits)
    out3, out32 = out3.cpu().permute(0, 3, 1, 2), out32.cpu().permute(0, 3, 1, 2)
    scale = float(ref64.abs().mean())
    e3 = float((out3.double() - ref64).abs().max()) / scale
    e32 = float((out32.double() - ref64).abs().max()) / scale
    assert e3 <= max(1.5 * e32, 2e-6), (e3, e32)
    _check(out3, ref64.float(), tol=2e-5 * max(1.0, scale))


@pytest.mark.parametrize("case", F16_CASES)
@pytest.mark.parametrize("tile", W64_TILES)
def test_conv_w64_f16_operands(cuda, case, tile):
    N, H, W, Cin, Cout, k, st, pad, act = case
    g = torch.Generator().manual_seed(900 + CASES.index(case))
    x = torch.randn(N, H, W, Cin, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    ref16 = _ref(x.half().float(), w.half().float(), b, st, pad, act, None, False)
    out = ops.conv2d_nhwc(x.to(cuda), w, b, stride=st, pad=pad, act=act, tile=tile + "_f16", splits=1)
    _check(out.cpu().permute(0, 3, 1, 2), ref16)


@pytest.mark.parametrize("tile", W64_TILES)
def test_conv_w64_store_modes(cuda, tile):
    g = torch.Generator().manual_seed(15)
    x = torch.randn(2, 10, 8, 64, generator=g)
    w = torch.randn(128, 64, 3, 3, generator=g) / 24
    b = torch.randn(128, generator=g)
    ref = _ref(x, w, b, 1, 1, "relu", None, False)
    xd = x.to(cuda)
    t = tile + "_b3"
    up = ops.conv2d_nhwc(xd, w, b, pad=1, act="relu", store="up2", tile=t).cpu().permute(0, 3, 1, 2)
    _check(up, F.interpolate(ref, scale_factor=2, mode="nearest"))
    ps = ops.conv2d_nhwc(xd, w, b, pad=1, act="relu", store="pixshuf", tile=t).cpu().permute(0, 3, 1, 2)
    _check(ps, F.pixel_shuffle(ref, 2))
    nc = ops.conv2d_nhwc(xd, w, b, pad=1, act="relu", store="nchw", tile=t, splits=2).cpu()
    _check(nc, ref)


def test_conv_w64_full_size_layers(cuda):
    """Full-size layers of both networks on their planned tiles: spot-check against the definition (fp64) and the
    size-independent linearity property."""
    g = torch.Generator().manual_seed(21)
    for (H, W, Cin, Cout, k, tile) in [(52, 52, 128, 256, 3, "w1x2"), (13, 13, 512, 1024, 3, "w1x2"),
                                        (104, 104, 64, 128, 3, "w2x2"), (20, 16, 1024, 256, 1, "w1x2"),
                                        (208, 208, 64, 32, 1, "w2x1")]:
        x1 = torch.randn(1, H, W, Cin, generator=g)
        x2 = torch.randn(1, H, W, Cin, generator=g)
        w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
        y1 = ops.conv2d_nhwc(x1.to(cuda), w, None, pad=k // 2, tile=tile + "_b3")
        y2 = ops.conv2d_nhwc(x2.to(cuda), w, None, pad=k // 2, tile=tile + "_b3")
        y3 = ops.conv2d_nhwc((2.0 * x1 + x2).to(cuda), w, None, pad=k // 2, tile=tile + "_b3")
        assert float((y3 - (2.0 * y1 + y2)).abs().max()) < 5e-5
        ref = F.conv2d(x1.double().permute(0, 3, 1, 2), w.double(), padding=k // 2).float()
        _check(y1.cpu().permute(0, 3, 1, 2), ref)


# ---- conv_kg.hip: 64x64 tile, K split over G groups of four waves INSIDE the block (partial sums meet in LDS), with and
# without cross-block slices on top.  Same bars: fp32-accurate, every epilogue / store mode, bit-reproducible.
KG_TILES = ["kg1", "kg2", "kg4", "rd4", "rd8", "bd"]   # rd<W>: conv_rd.hip, operands global -> registers -> MFMA, W K ranges per block; bd: conv_igemm.hip's 64x64 kernel with the filter fragments straight into registers


@pytest.mark.parametrize("case", F16_CASES)
@pytest.mark.parametrize("tile", KG_TILES)
@pytest.mark.parametrize("splits", [1, 3])
def test_conv_kg_bf16x3_is_fp32_accurate(cuda, case, tile, splits):
    N, H, W, Cin, Cout, k, st, pad, act = case
    g = torch.Generator().manual_seed(1100 + CASES.index(case))
    x = torch.randn(N, H, W, Cin, generator=g) * torch.exp(2 * torch.randn(N, H, W, 1, generator=g))   # wide dynamic range
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    if splits > 1 and Cin * k * k // 32 < 2 * splits:
        pytest.skip("too few K-chunks to split")
    OH, OW = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
    res = torch.randn(N, OH, OW, Cout, generator=g)
    for after in (False, True):
        ref64 = _ref(x.double(), w.double(), b.double(), st, pad, act, res.double(), after)
        out3 = ops.conv2d_nhwc(x.to(cuda), w, b, stride=st, pad=pad, act=act, res=res.to(cuda), res_after_act=after,
                               tile=tile + "_b3", splits=splits)
        out32 = ops.conv2d_nhwc(x.to(cuda), w, b, stride=st, pad=pad, act=act, res=res.to(cuda), res_after_act=after,
                                tile="64x64", splits=splits)
        again = ops.conv2d_nhwc(x.to(cuda), w, b, stride=st, pad=pad, act=act, res=res.to(cuda), res_after_act=after,
                                tile=tile + "_b3", splits=splits)
        assert torch.equal(out3, again)                         # fixed summation order: bit-reproducible
        out3, out32 = out3.cpu().permute(0, 3, 1, 2), out32.cpu().permute(0, 3, 1, 2)
        scale = float(ref64.abs().mean())
        e3 = float((out3.double() - ref64).abs().max()) / scale
        e32 = float((out32.double() - ref64).abs().max()) / scale
        assert e3 <= max(1.5 * e32, 2e-6), (e3, e32)
        _check(out3, ref64.float(), tol=2e-5 * max(1.0, scale))


@pytest.mark.parametrize("tile", KG_TILES)
def test_conv_kg_store_modes(cuda, tile):
    g = torch.Generator().manual_seed(16)
    x = torch.randn(2, 10, 8, 64, generator=g)
    w = torch.randn(128, 64, 3, 3, generator=g) / 24
    b = torch.randn(128, generator=g)
    ref = _ref(x, w, b, 1, 1, "relu", None, False)
    xd = x.to(cuda)
    t = tile + "_b3"
    up = ops.conv2d_nhwc(xd, w, b, pad=1, act="relu", store="up2", tile=t).cpu().permute(0, 3, 1, 2)
    _check(up, F.interpolate(ref, scale_factor=2, mode="nearest"))
    ps = ops.conv2d_nhwc(xd, w, b, pad=1, act="relu", store="pixshuf", tile=t).cpu().permute(0, 3, 1, 2)
    _check(ps, F.pixel_shuffle(ref, 2))
    nc = ops.conv2d_nhwc(xd, w, b, pad=1, act="relu", store="nchw", tile=t, splits=2).cpu()
    _check(nc, ref)
    # Cout below the tile (detection head class): columns past Cout are never stored
    w18 = torch.randn(18, 64, 1, 1, generator=g) / 8
    b18 = torch.randn(18, generator=g)
    hd = ops.conv2d_nhwc(xd, w18, b18, tile=t).cpu().permute(0, 3, 1, 2)
    _check(hd, _ref(x, w18, b18, 1, 0, "linear", None, False))


def test_conv_kg_full_size_layers(cuda):
    """Full-size layers of both networks: against the definition (fp64) and the size-independent linearity property;
    K ranges that do not divide over the groups (odd stage counts, groups left without work)."""
    g = torch.Generator().manual_seed(22)
    for (H, W, Cin, Cout, k, tile, splits) in [(52, 52, 128, 256, 3, "kg4", 1), (13, 13, 512, 1024, 3, "kg4", 3),
                                               (104, 104, 64, 128, 3, "kg2", 1), (20, 16, 1024, 256, 1, "kg4", 1),
                                               (208, 208, 64, 32, 1, "kg4", 1), (26, 26, 32, 64, 1, "kg4", 1),
                                               (13, 13, 96, 64, 3, "kg4", 2), (40, 32, 160, 64, 1, "kg2", 1)]:
        x1 = torch.randn(1, H, W, Cin, generator=g)
        x2 = torch.randn(1, H, W, Cin, generator=g)
        w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
        y1 = ops.conv2d_nhwc(x1.to(cuda), w, None, pad=k // 2, tile=tile + "_b3", splits=splits)
        y2 = ops.conv2d_nhwc(x2.to(cuda), w, None, pad=k // 2, tile=tile + "_b3", splits=splits)
        y3 = ops.conv2d_nhwc((2.0 * x1 + x2).to(cuda), w, None, pad=k // 2, tile=tile + "_b3", splits=splits)
        assert float((y3 - (2.0 * y1 + y2)).abs().max()) < 5e-5
        ref = F.conv2d(x1.double().permute(0, 3, 1, 2), w.double(), padding=k // 2).float()
        _check(y1.cpu().permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize("case", F16_CASES)
@pytest.mark.parametrize("splits", [1, 3])
@pytest.mark.parametrize("mode", ["b3", "f16"])
def test_conv_filters_direct_is_bit_identical_to_the_staged_kernel(cuda, case, splits, mode):
    """The filters-direct variant (filter fragments global -> registers from the stage-packed copy) feeds the MFMAs the
    same operands in the same order as the LDS-staged bf16x3 kernel: the two data paths must agree bit for bit."""
    N, H, W, Cin, Cout, k, st, pad, act = case
    if splits > 1 and Cin * k * k // 32 < 2 * splits:
        pytest.skip("too few K-chunks to split")
    g = torch.Generator().manual_seed(1300 + CASES.index(case))
    x = (torch.randn(N, H, W, Cin, generator=g) * torch.exp(2 * torch.randn(N, H, W, 1, generator=g))).to(cuda)
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    a1 = ops.conv2d_nhwc(x, w, b, stride=st, pad=pad, act=act, tile="bd_" + mode, splits=splits)
    a2 = ops.conv2d_nhwc(x, w, b, stride=st, pad=pad, act=act, tile="64x64_" + mode, splits=splits)
    assert torch.equal(a1, a2)




# ---- three round-3 forms of conv_pl.hip that were measured and left out of the product: K groups inside the
# block (pl64k2: 8 waves, a ring per group, partial sums combined through LDS), wave specialisation (pl128s: 4 loader waves),
# filter fragments direct from global memory beside activation planes by LDS-DMA (pl64bd)
@pytest.mark.parametrize("tile,splits", [("pl64k2_b3", 1), ("pl64k2_b3", 3), ("pl64k2_f16", 1), ("pl64k2_f16", 2), ("pl128s_b3", 1), ("pl128s_f16", 1),
                                          ("pl64bd_b3", 1), ("pl64bd_b3", 2), ("pl64bd_b3", 5)])
@pytest.mark.parametrize("shape", [(20, 16, 256, 256, 3), (26, 26, 128, 192, 1), (13, 13, 96, 128, 3)])
def test_conv_pl_round3_experiments(cuda, tile, splits, shape):
    H, W, Cin, Cout, k = shape
    g = torch.Generator().manual_seed(5200 + H + Cout + splits)
    x = torch.randn(1, H, W, Cin, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    out = ops.conv2d_nhwc(x.to(cuda), w, b, pad=k // 2, act="relu", tile=tile, splits=splits).cpu().permute(0, 3, 1, 2)
    if tile.endswith("f16"):
        _check(out, _ref(x.half().float(), w.half().float(), b, 1, k // 2, "relu", None, False))
    else:
        _check(out, _ref(x, w, b, 1, k // 2, "relu", None, False))
