"""Oracle (test infrastructure, see oracle/__init__.py): detector half.

Restates, with stock torch-CPU fp32 ops,
  * ``Darknet.forward``          3_6Dpose_estimator/yolo/darknet.py:319-363
  * ``DetectionLayer.forward``   yolo/darknet.py:129-169
  * ``write_results`` / ``dynamic_write_results``  yolo/util.py:104-223
  * box rescale of ``DetectionLoader.update``      dataloader.py:354-364
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F


def _t(a) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def detection_layer(x: torch.Tensor, anchors, reso: int) -> torch.Tensor:
    """yolo/darknet.py:129-169.  ``x``: [B, 3*(5+C), g, g] -> [B, 3*g*g, 5+C]
    in anchor -> gy -> gx row order."""
    bs, ch, gs, _ = x.shape
    nA = len(anchors)
    attrs = ch // nA
    stride = reso // gs
    scaled = torch.tensor([(aw / stride, ah / stride) for aw, ah in anchors], dtype=torch.float32)
    grid_x = torch.arange(gs).repeat(gs, 1).view(1, 1, gs, gs).float()
    grid_y = torch.arange(gs).repeat(gs, 1).t().reshape(1, 1, gs, gs).float()
    aw = scaled[:, 0:1].view(1, nA, 1, 1)
    ah = scaled[:, 1:2].view(1, nA, 1, 1)
    x = x.view(bs, nA, attrs, gs, gs).permute(0, 1, 3, 4, 2).contiguous()
    det = torch.empty(bs, nA, gs, gs, attrs)
    det[..., 0] = torch.sigmoid(x[..., 0]) + grid_x
    det[..., 1] = torch.sigmoid(x[..., 1]) + grid_y
    det[..., 2] = torch.exp(x[..., 2]) * aw
    det[..., 3] = torch.exp(x[..., 3]) * ah
    det[..., :4] *= stride
    det[..., 4] = torch.sigmoid(x[..., 4])
    det[..., 5:] = torch.sigmoid(x[..., 5:])
    return det.view(bs, -1, attrs)


def darknet_forward(blocks: List[Dict[str, str]], convs: List[dict], x: torch.Tensor,
                    reso: int = 416, keep: Optional[dict] = None) -> torch.Tensor:
    """Forward of the cfg network.  ``convs`` = per-conv arrays from
    ``betapose_amd.weights.split_darknet_stream`` (raw, un-folded).  Returns
    [B, sum 3 g^2, 5+C].  ``keep`` (optional dict) receives every layer output
    (NCHW) for per-layer checks."""
    by_index = {c["index"]: c for c in convs}
    outputs: Dict[int, torch.Tensor] = {}
    dets = None
    with torch.no_grad():
        for i, b in enumerate(blocks):
            t = b["type"]
            if t == "convolutional":
                c = by_index[i]
                k = c["k"]
                pad = (k - 1) // 2 if b["pad"] else 0   # string truthiness, darknet.py:250
                w = _t(c["weight"])
                if c["bn"]:
                    x = F.conv2d(x, w, None, stride=c["stride"], padding=pad)
                    x = F.batch_norm(x, _t(c["bn_mean"]), _t(c["bn_var"]), _t(c["bn_weight"]),
                                     _t(c["bn_bias"]), False, 0.1, 1e-5)
                else:
                    x = F.conv2d(x, w, _t(c["bias"]), stride=c["stride"], padding=pad)
                if b["activation"] == "leaky":
                    x = F.leaky_relu(x, 0.1)
                outputs[i] = x
            elif t == "upsample":
                x = F.interpolate(x, scale_factor=int(b["stride"]), mode="nearest")
                outputs[i] = x
            elif t == "shortcut":
                x = outputs[i - 1] + outputs[i + int(b["from"])]
                outputs[i] = x
            elif t == "route":
                layers = [int(a) for a in b["layers"].split(",")]
                if len(layers) == 1:
                    x = outputs[i + layers[0]]
                else:
                    x = torch.cat((outputs[i + layers[0]], outputs[layers[1]]), 1)
                outputs[i] = x
            elif t == "yolo":
                mask = [int(m) for m in b["mask"].split(",")]
                a = [int(v) for v in b["anchors"].split(",")]
                anchors = [(a[2 * j], a[2 * j + 1]) for j in mask]
                d = detection_layer(x, anchors, reso)
                dets = d if dets is None else torch.cat((dets, d), 1)
                outputs[i] = outputs[i - 1]
            else:
                raise NotImplementedError(t)
    if keep is not None:
        keep.update(outputs)
    return dets


def write_results(prediction: torch.Tensor, confidence: float, num_classes: int = 80):
    """yolo/util.py:118-223 with its hard-coded ``nms = False`` (:181): one row
    per image = the highest-objectness candidate above ``confidence`` whose
    arg-max class is 0.  Returns ``0`` (int) when nothing passes, else
    [n, 8] = (batch, x1, y1, x2, y2, obj, cls_conf, cls_idx)."""
    pred = prediction.clone()
    mask = (pred[:, :, 4] > confidence).float().unsqueeze(2)
    pred = pred * mask
    box = pred.new_empty(pred.shape)
    box[:, :, 0] = pred[:, :, 0] - pred[:, :, 2] / 2
    box[:, :, 1] = pred[:, :, 1] - pred[:, :, 3] / 2
    box[:, :, 2] = pred[:, :, 0] + pred[:, :, 2] / 2
    box[:, :, 3] = pred[:, :, 1] + pred[:, :, 3] / 2
    pred[:, :, :4] = box[:, :, :4]
    rows = []
    for ind in range(pred.size(0)):
        ip = pred[ind]
        cls_conf, cls_idx = torch.max(ip[:, 5:5 + num_classes], 1)
        ip = torch.cat((ip[:, :5], cls_conf.float().unsqueeze(1), cls_idx.float().unsqueeze(1)), 1)
        nz = torch.nonzero(ip[:, 4]).squeeze(1)
        ip = ip[nz]
        if ip.shape[0] == 0:
            continue
        ip = ip[ip[:, -1] == 0]          # "if cls != 0: continue"
        if ip.shape[0] == 0:
            continue
        order = torch.sort(ip[:, 4], descending=True)[1]
        ip = ip[order]
        best = int(np.argmax(ip[:, 4].numpy()))
        row = torch.cat((torch.tensor([float(ind)]), ip[best]))
        rows.append(row.view(1, -1))
    if not rows:
        return 0
    return torch.cat(rows)


def select_index(prediction: torch.Tensor, confidence: float) -> np.ndarray:
    """The "YOLO box index" of SURVEY §8 a5: arg-max objectness position in the
    DetectionLayer row order, -1 when nothing exceeds ``confidence``."""
    obj = prediction[:, :, 4]
    idx = torch.argmax(obj, dim=1)
    ok = obj.gather(1, idx[:, None])[:, 0] > confidence
    out = idx.numpy().astype(np.int64)
    out[~ok.numpy()] = -1
    return out


def rescale_boxes(dets: torch.Tensor, im_dim_list: torch.Tensor, reso: int):
    """dataloader.py:354-364 -- stretch (no letterbox) back to frame pixels."""
    dims = torch.index_select(im_dim_list, 0, dets[:, 0].long())
    w, h = dims[:, 0], dims[:, 1]
    boxes = dets[:, 1:5].clone()
    boxes[:, 0] = boxes[:, 0] * (w / reso)
    boxes[:, 1] = boxes[:, 1] * (h / reso)
    boxes[:, 2] = boxes[:, 2] * (w / reso)
    boxes[:, 3] = boxes[:, 3] * (h / reso)
    scores = dets[:, 5:6].clone()
    return boxes, scores
