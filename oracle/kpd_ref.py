"""Oracle (test infrastructure, see oracle/__init__.py): key-point detector.

Restates with stock torch-CPU fp32 ops
  * ``FastPose.forward``      KPD/src/models/FastPose.py:28-35
  * ``SEResnet.forward``      KPD/src/models/layers/SE_Resnet.py:70-76
  * ``Bottleneck.forward``    SE_Resnet.py:25-42
  * ``SELayer.forward``       layers/SE_module.py:15-19
  * ``DUC.forward``           layers/DUC.py:18-23
  * ``InferenNet_fast.forward`` (narrow to 50 maps)  KPD/src/main_fast_inference.py:42-46
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

STAGES = ((64, 3, 1), (128, 4, 2), (256, 23, 2), (512, 3, 2))


_DT = torch.float32      # arithmetic of the restatement: fp32 as the reference runs it; fp64 only for the flip-rate tests' "exact" run


class arithmetic:
    """``with kpd_ref.arithmetic(torch.float64): ...`` -- the same restatement in double precision (inputs must be double too)."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global _DT
        self._old, _DT = _DT, self.dtype

    def __exit__(self, *exc):
        global _DT
        _DT = self._old
        return False


def _t(a) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        return a.to(_DT)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(_DT)


def _bn(x, sd, name):
    return F.batch_norm(x, _t(sd[name + ".running_mean"]), _t(sd[name + ".running_var"]),
                        _t(sd[name + ".weight"]), _t(sd[name + ".bias"]), False, 0.1, 1e-5)


def _bottleneck(x, sd, p, stride, first):
    residual = x
    out = F.relu(_bn(F.conv2d(x, _t(sd[p + ".conv1.weight"])), sd, p + ".bn1"))
    out = F.relu(_bn(F.conv2d(out, _t(sd[p + ".conv2.weight"]), stride=stride, padding=1), sd, p + ".bn2"))
    out = _bn(F.conv2d(out, _t(sd[p + ".conv3.weight"])), sd, p + ".bn3")
    if first:   # reduction=True <=> downsample is not None (SE_Resnet.py:91-94)
        b, c = out.shape[:2]
        y = F.adaptive_avg_pool2d(out, 1).view(b, c)
        y = F.relu(F.linear(y, _t(sd[p + ".se.fc.0.weight"]), _t(sd[p + ".se.fc.0.bias"])))
        y = torch.sigmoid(F.linear(y, _t(sd[p + ".se.fc.2.weight"]), _t(sd[p + ".se.fc.2.bias"])))
        out = out * y.view(b, c, 1, 1)
        residual = _bn(F.conv2d(x, _t(sd[p + ".downsample.0.weight"]), stride=stride), sd, p + ".downsample.1")
    out = out + residual
    return F.relu(out)


def fastpose_forward(sd: Dict[str, object], x: torch.Tensor, n_keep: int = 50,
                     keep: Optional[dict] = None) -> torch.Tensor:
    """[B,3,320,256] -> [B,50,80,64] heat-maps."""
    with torch.no_grad():
        x = F.conv2d(x, _t(sd["preact.conv1.weight"]), stride=2, padding=3)
        x = F.relu(_bn(x, sd, "preact.bn1"))
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        if keep is not None:
            keep["stem"] = x
        for li, (planes, nblocks, stride) in enumerate(STAGES, start=1):
            for bi in range(nblocks):
                p = "preact.layer%d.%d" % (li, bi)
                x = _bottleneck(x, sd, p, stride if bi == 0 else 1, bi == 0)
                if keep is not None:
                    keep[p] = x
        x = F.pixel_shuffle(x, 2)
        for d in ("duc1", "duc2"):
            x = F.conv2d(x, _t(sd[d + ".conv.weight"]), padding=1)
            x = F.relu(_bn(x, sd, d + ".bn"))
            x = F.pixel_shuffle(x, 2)
            if keep is not None:
                keep[d] = x
        x = F.conv2d(x, _t(sd["conv_out.weight"]), _t(sd["conv_out.bias"]), padding=1)
        return x.narrow(1, 0, n_keep)
