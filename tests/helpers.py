"""Shared test inputs: the same seeded synthetic weights / frames the golden
fixtures were generated from (tools/make_golden.py)."""
import functools
import os

import numpy as np

from betapose_amd import cfg as C, synth, weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
YOLO_SEED, KPD_SEED, FRAME_SEED = 1, 2, 1234


@functools.lru_cache(None)
def yolo_blocks():
    return C.parse_cfg_text(C.yolov3_single_cfg_text())


@functools.lru_cache(None)
def yolo_stream():
    return synth.synth_yolo_stream(YOLO_SEED, yolo_blocks())


@functools.lru_cache(None)
def kpd_state_dict():
    return synth.synth_fastpose_state_dict(KPD_SEED)


@functools.lru_cache(None)
def frames(n=4):
    return synth.synth_frames(n, FRAME_SEED)


def golden(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def yolo_input_from_frame(frame_bgr, reso=416):
    """a1 on the host exactly as the reference does it: PIL bicubic stretch + ToTensor
    (dataloader.py:94-99,162)."""
    import torch
    from PIL import Image
    img = Image.fromarray(np.ascontiguousarray(frame_bgr[:, :, ::-1])).resize((reso, reso), 3)
    a = np.asarray(img, dtype=np.uint8).transpose(2, 0, 1).copy()
    return torch.from_numpy(a).float().div(255).unsqueeze(0)
