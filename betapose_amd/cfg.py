"""Darknet ``.cfg`` handling for the detector half of the hot path.

``parse_cfg`` keeps the reference's block-list semantics
(3_6Dpose_estimator/yolo/darknet.py:45-74): every ``[section]`` becomes a dict
of *strings* with a ``type`` key, comments (``#``) and blank lines dropped,
keys/values stripped.  ``yolov3_single_cfg_text`` regenerates the text of the
single-class YOLOv3 network the reference ships as
``yolo/cfg/yolov3-single.cfg`` (75 conv, 23 shortcut, 4 route, 2 upsample,
3 yolo; no ``[net]`` block) from a compact description, so the repo does not
have to carry the reference's file; ``tests/test_formats.py`` pins the parsed
block list against a digest of the reference's own cfg.
"""
from __future__ import annotations

import os
from typing import Dict, List

ANCHORS = "10,13,  16,30,  33,23,  30,61,  62,45,  59,119,  116,90,  156,198,  373,326"


def parse_cfg_text(text: str) -> List[Dict[str, str]]:
    lines = [ln.strip() for ln in text.split("\n")]
    lines = [ln for ln in lines if ln and not ln.startswith("#")]
    blocks: List[Dict[str, str]] = []
    block: Dict[str, str] = {}
    for line in lines:
        if line[0] == "[":
            if block:
                blocks.append(block)
                block = {}
            block["type"] = line[1:-1].strip()
        else:
            if "=" not in line:
                raise ValueError("malformed cfg line: %r" % line)
            key, value = line.split("=", 1)
            block[key.strip()] = value.strip()
    if block:
        blocks.append(block)
    return blocks


def parse_cfg(cfgfile: str) -> List[Dict[str, str]]:
    """Same call surface as the reference's ``parse_cfg(cfgfile)``.

    The reference hard-codes ``yolo/cfg/yolov3-single.cfg``
    (dataloader.py:289); when that path does not exist we fall back to the
    generated text so an unmodified ``DetectionLoader`` keeps working.
    """
    if os.path.exists(cfgfile):
        with open(cfgfile, "r") as f:
            return parse_cfg_text(f.read())
    if os.path.basename(cfgfile) == "yolov3-single.cfg":
        return parse_cfg_text(yolov3_single_cfg_text())
    raise FileNotFoundError(cfgfile)


def _conv(filters: int, size: int, stride: int, bn: bool = True, act: str = "leaky") -> str:
    out = ["[convolutional]"]
    if bn:
        out.append("batch_normalize=1")
    if not bn:
        out += ["size=%d" % size, "stride=%d" % stride, "pad=1", "filters=%d" % filters]
    else:
        out += ["filters=%d" % filters, "size=%d" % size, "stride=%d" % stride, "pad=1"]
    out.append("activation=%s" % act)
    return "\n".join(out) + "\n"


def _shortcut() -> str:
    return "[shortcut]\nfrom=-3\nactivation=linear\n"


def _yolo(mask: str, classes: int) -> str:
    return ("[yolo]\nmask = %s\nanchors = %s\nclasses=%d\nnum=9\njitter=.5\n"
            "ignore_thresh = .7\ntruth_thresh = 1\nrandom=1\n" % (mask, ANCHORS, classes))


def yolov3_single_cfg_text(classes: int = 1) -> str:
    """Text of the YOLOv3 (Darknet-53 + 3-scale head) cfg with ``classes`` classes."""
    nf = 3 * (5 + classes)
    parts: List[str] = []
    # Darknet-53 backbone
    parts.append(_conv(32, 3, 1))
    for ch, reps in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
        parts.append(_conv(ch, 3, 2))                      # downsample
        for _ in range(reps):
            parts.append(_conv(ch // 2, 1, 1))
            parts.append(_conv(ch, 3, 1))
            parts.append(_shortcut())
    # head, scale 13
    for _ in range(3):
        parts.append(_conv(512, 1, 1))
        parts.append(_conv(1024, 3, 1))
    parts.append(_conv(nf, 1, 1, bn=False, act="linear"))
    parts.append(_yolo("6,7,8", classes))
    # scale 26
    parts.append("[route]\nlayers = -4\n")
    parts.append(_conv(256, 1, 1))
    parts.append("[upsample]\nstride=2\n")
    parts.append("[route]\nlayers = -1, 61\n")
    for _ in range(3):
        parts.append(_conv(256, 1, 1))
        parts.append(_conv(512, 3, 1))
    parts.append(_conv(nf, 1, 1, bn=False, act="linear"))
    parts.append(_yolo("3,4,5", classes))
    # scale 52
    parts.append("[route]\nlayers = -4\n")
    parts.append(_conv(128, 1, 1))
    parts.append("[upsample]\nstride=2\n")
    parts.append("[route]\nlayers = -1, 36\n")
    for _ in range(3):
        parts.append(_conv(128, 1, 1))
        parts.append(_conv(256, 3, 1))
    parts.append(_conv(nf, 1, 1, bn=False, act="linear"))
    parts.append(_yolo("0,1,2", classes))
    return "\n".join(parts)


def conv_blocks(blocks: List[Dict[str, str]]):
    """Yield ``(index, block, in_channels, out_channels)`` for every conv block,
    tracking channel counts the way ``Darknet.build_model`` does
    (yolo/darknet.py:223-317)."""
    in_ch = 3
    outs: List[int] = []
    for idx, b in enumerate(blocks):
        out_ch = in_ch
        if b["type"] == "convolutional":
            out_ch = int(b["filters"])
            yield idx, b, in_ch, out_ch
        elif b["type"] == "route":
            layers = [int(a) for a in b["layers"].split(",")]
            if len(layers) == 1:
                out_ch = outs[idx + layers[0]]
            else:
                out_ch = outs[idx + layers[0]] + outs[layers[1]]
        elif b["type"] in ("shortcut", "upsample", "yolo", "maxpool", "net"):
            out_ch = in_ch if b["type"] != "net" else 3
        outs.append(out_ch)
        in_ch = out_ch
