"""``dynamic_write_results`` with the reference's signature and return convention
(3_6Dpose_estimator/yolo/util.py:104-223): NMS is hard-wired off there (:181), so the
result is one row per image -- the arg-max objectness candidate above ``confidence`` whose
arg-max class is 0 -- or the int ``0`` when no image has one.  The reduction runs in the HIP
select kernel on the device-resident prediction tensor."""
from __future__ import annotations

from . import _lib
from .darknet import sel_to_dets


def write_results(prediction, confidence, num_classes, nms=True, nms_conf=0.4):
    import torch
    _lib.require_gpu()
    pred = prediction if prediction.is_cuda else prediction.cuda()
    pred = pred.contiguous().float()
    B, rows, attrs = pred.shape
    sel = torch.empty((B, 8), device=pred.device, dtype=torch.float32)
    _lib.check(_lib.lib().bp_yolo_select(pred.data_ptr(), B, rows, attrs, float(confidence), int(num_classes),
                                         sel.data_ptr(), _lib.current_stream()))
    return sel_to_dets(sel)


def dynamic_write_results(prediction, confidence, num_classes, nms=True, nms_conf=0.4):
    # the reference re-runs with nms_conf - 0.05 when more than 100 rows survive; with one row per image
    # that needs a batch > 100, where the second pass (NMS still off) returns the same rows
    return write_results(prediction, confidence, num_classes, nms, nms_conf)
