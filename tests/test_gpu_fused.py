"""Conv -> conv fusion of whole residual / bottleneck blocks (betapose_amd/csrc/conv_fused.hip, round 5): the Darknet-53 residual
block's 1x1 + 3x3 + shortcut (yolo/darknet.py:319-363, shortcut :338-340) and the bottleneck's conv1 + conv2 (+ conv3 + skip
connection; KPD/src/models/layers/SE_Resnet.py:25-42) as ONE launch on 8 x 8 output patches.

The whole parity suite runs WITH the fusion (it is the default plan: tests/test_gpu_nets.py compares every tap with the oracle);
here the fused plan is held against the unfused one tap by tap, across batch sizes and image borders, and for determinism."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import helpers  # noqa: E402
from betapose_amd.darknet import Darknet  # noqa: E402
from betapose_amd.kpd import FastPoseHIP  # noqa: E402

REL = 3e-5          # the per-layer bar of tests/test_gpu_nets.py (relative to the layer's scale)


def _pair(make):
    a, b = make(), make()
    b.set_fusion(False)
    return a, b


def _close(x, y, what):
    scale = max(1.0, float(y.abs().max()))
    assert float((x - y).abs().max()) <= REL * scale, (what, float((x - y).abs().max()), scale)


def test_yolo_fused_blocks_equal_the_unfused_plan_tap_by_tap(cuda):
    fused, plain = _pair(lambda: Darknet("yolo/cfg/yolov3-single.cfg", reso=416, max_batch=2).load_stream(helpers.yolo_stream()).cuda().eval())
    assert fused.fused_launches(1) == 3 and plain.fused_launches(1) == 0        # the 208x208 block and the two 104x104 blocks
    x = torch.cat([helpers.yolo_input_from_frame(f) for f in helpers.frames(2)])
    for batch in (1, 2):
        pf, pp = fused(x[:batch].to(cuda)).cpu(), plain(x[:batch].to(cuda)).cpu()
        assert torch.equal(fused(x[:batch].to(cuda)).cpu(), pf)                 # fixed summation order
        for i, (name, *_shape) in enumerate(fused.taps()):
            _close(fused.tap(i, batch=batch).cpu(), plain.tap(i, batch=batch).cpu(), "yolo tap " + name)
        for bb in range(batch):
            assert int(pf[bb, :, 4].argmax()) == int(pp[bb, :, 4].argmax())
        assert float((pf[..., 4:] - pp[..., 4:]).abs().max()) <= 2e-5
        assert bool(((pf[..., :4] - pp[..., :4]).abs() <= 2e-3 + 3e-5 * pp[..., :4].abs()).all())
    # a frame of a batch of two against its own launch (patches never span images; the OTHER layers' K slices differ with the batch size,
    # so this is the rows' fp32 bar, not bit equality)
    p2 = fused(x.to(cuda)).cpu()
    for bb in range(2):
        p1 = fused(x[bb:bb + 1].to(cuda)).cpu()[0]
        assert float((p1[:, 4:] - p2[bb][:, 4:]).abs().max()) <= 2e-5
        assert bool(((p1[:, :4] - p2[bb][:, :4]).abs() <= 2e-3 + 3e-5 * p2[bb][:, :4].abs()).all())
    # Net::profile reports the members: two of each group launch nothing (tile -1), the last one carries the block (tile 40)
    info = fused.profile(1, 1)[1]
    assert int((info[:, 1] == 40).sum()) == 3 and int((info[:, 1] == -1).sum()) == 3


def test_kpd_fused_bottlenecks_equal_the_unfused_plan_tap_by_tap(cuda):
    fused, plain = _pair(lambda: FastPoseHIP(helpers.kpd_state_dict(), n_classes=50, max_batch=3).cuda().eval())
    assert fused.fused_launches(1) == 3 and plain.fused_launches(1) == 0        # layer1.0 (conv1 + conv2), layer1.1 / layer1.2 (whole bottlenecks)
    g = torch.Generator().manual_seed(77)
    inps = torch.rand(3, 3, 320, 256, generator=g) - 0.45
    for batch in (1, 3):
        hf, hp = fused(inps[:batch].to(cuda)).cpu(), plain(inps[:batch].to(cuda)).cpu()
        assert torch.equal(fused(inps[:batch].to(cuda)).cpu(), hf)
        for i, (name, *_shape) in enumerate(fused.taps()):
            _close(fused.tap(i, batch=batch).cpu(), plain.tap(i, batch=batch).cpu(), "kpd tap " + name)
        assert float((hf - hp).abs().max()) <= 1e-4
        assert torch.equal(hf.reshape(batch, 50, -1).argmax(2), hp.reshape(batch, 50, -1).argmax(2))
    h3 = fused(inps.to(cuda)).cpu()
    for bb in range(3):
        assert float((fused(inps[bb:bb + 1].to(cuda)).cpu()[0] - h3[bb]).abs().max()) <= 1e-4


def test_fusion_follows_the_plan(cuda):
    """The fp32-MFMA mode keeps one launch per convolution; bf16x3 (fp32 activations) and the fp16 modes (operand planes) fuse the same
    three groups; switching the mode or the fusion flag re-plans, and a clone inherits the flag."""
    net = Darknet("yolo/cfg/yolov3-single.cfg", reso=416, max_batch=1).load_stream(helpers.yolo_stream()).cuda().eval()
    x = helpers.yolo_input_from_frame(helpers.frames(1)[0]).to(cuda)
    assert net.fused_launches(1) == 3
    base = net(x).cpu()
    net.set_precision("f32")
    assert net.fused_launches(1) == 0 and torch.isfinite(net(x)).all()
    for mode in ("f16", "f16r"):
        net.set_precision(mode)
        assert net.fused_launches(1) == 1
        assert torch.isfinite(net(x)).all()
    net.set_precision("bf16x3")
    assert net.fused_launches(1) == 3 and torch.equal(net(x).cpu(), base)
    net.set_fusion(False)
    twin = net.clone()
    assert net.fused_launches(1) == 0 and twin.fused_launches(1) == 0
    unf = net(x).cpu()
    assert torch.equal(twin(x).cpu(), unf)
    net.set_fusion(True)
    assert torch.equal(net(x).cpu(), base)
    assert float((unf[..., 4:] - base[..., 4:]).abs().max()) <= 2e-5


@pytest.mark.parametrize("mode", ["f16", "f16r"])
def test_fp16_fused_blocks_equal_the_unfused_plan(cuda, mode):
    """The fp16 form of the fused block (operand planes in, fp16 plane [+ fp32 tensor] out, fp16 skip connections in 'f16r'): the
    intermediates are rounded to fp16 where the unfused launches round them, so fused and unfused differ by summation order and by the
    fp16 roundings that order flips -- held to the fp16 modes' stated tolerances (tests/test_gpu_nets.py), tap by tap at 1e-2 of the
    layer's scale, batch 1 and 3."""
    fused, plain = _pair(lambda: Darknet("yolo/cfg/yolov3-single.cfg", reso=416, max_batch=3).load_stream(helpers.yolo_stream()).cuda().eval())
    kf, kp = _pair(lambda: FastPoseHIP(helpers.kpd_state_dict(), n_classes=50, max_batch=3).cuda().eval())
    for n in (fused, plain, kf, kp):
        n.set_precision(mode)
    assert fused.fused_launches(3) == 1 and kf.fused_launches(3) == 3 and plain.fused_launches(3) == 0      # (fp16: the 104x104 blocks stay unfused, conv_fused.hip fused_form)
    x = torch.cat([helpers.yolo_input_from_frame(f) for f in helpers.frames(3)])
    g = torch.Generator().manual_seed(78)
    inps = torch.rand(3, 3, 320, 256, generator=g) - 0.45
    for batch in (1, 3):
        pf, pp = fused(x[:batch].to(cuda)).cpu(), plain(x[:batch].to(cuda)).cpu()
        assert torch.equal(fused(x[:batch].to(cuda)).cpu(), pf)
        d = (pf - pp).abs()
        assert float(d[..., :2].max()) < 0.25 and float(d[..., 4:].max()) < 5e-3
        assert bool((d[..., 2:4] <= 0.05 + 2e-2 * pp[..., 2:4].abs()).all())
        for i, (name, *_s) in enumerate(fused.taps()[:12]):
            a, bb = fused.tap(i, batch=batch).cpu(), plain.tap(i, batch=batch).cpu()
            assert float((a - bb).abs().max()) <= 1e-2 * max(1.0, float(bb.abs().max())), ("yolo tap " + name, mode)
        hf, hp = kf(inps[:batch].to(cuda)).cpu(), kp(inps[:batch].to(cuda)).cpu()
        assert torch.equal(kf(inps[:batch].to(cuda)).cpu(), hf)
        assert float((hf - hp).abs().max()) < 1e-2
        for i, (name, *_s) in enumerate(kf.taps()[:5]):
            a, bb = kf.tap(i, batch=batch).cpu(), kp.tap(i, batch=batch).cpu()
            assert float((a - bb).abs().max()) <= 1e-2 * max(1.0, float(bb.abs().max())), ("kpd tap " + name, mode)


def test_fused_blocks_at_other_resolutions(cuda):
    """Other map sizes (reso 320 / 608: 160, 80 / 304, 152 wide maps): patch grids of other shapes,
    fused == unfused at the layer bar."""
    for reso in (320, 608):
        fused, plain = _pair(lambda: Darknet("yolo/cfg/yolov3-single.cfg", reso=reso, max_batch=1).load_stream(helpers.yolo_stream()).cuda().eval())
        g = torch.Generator().manual_seed(reso)
        x = torch.rand(1, 3, reso, reso, generator=g)
        pf, pp = fused(x.to(cuda)).cpu(), plain(x.to(cuda)).cpu()
        assert fused.fused_launches(1) == 3
        assert float((pf[..., 4:] - pp[..., 4:]).abs().max()) <= 2e-5
        assert bool(((pf[..., :4] - pp[..., :4]).abs() <= 2e-3 + 3e-5 * pp[..., :4].abs()).all())
