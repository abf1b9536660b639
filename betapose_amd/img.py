"""Image helpers with the reference's names (KPD/src/utils/img.py:13-18,242-262; dataloader.py:794-835).
The crop itself runs in the HIP crop kernel (``bp_crop``); these are the host-facing wrappers."""
from __future__ import annotations

import numpy as np

from . import ops


def im_to_torch(img):
    """HWC u8/float -> CHW float 0..1 (img.py:13-18)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(np.transpose(img, (2, 0, 1)))).float()
    if t.max() > 1:
        t /= 255
    return t


def load_frame_bgr(path: str) -> np.ndarray:
    """cv2.imread replacement: BGR u8 HWC (8-bit PNG/JPEG decode via Pillow is bit-identical)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] == b"\x89PNG\r\n\x1a\n":   # same decoder as the fused path (16-bit / palette / alpha handled one way)
        from .frame_loader import decode_png
        return decode_png(data)
    import io
    from PIL import Image
    return np.ascontiguousarray(np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))[:, :, ::-1])


def crop_from_dets_frame(frame_bgr_u8, boxes, inputResH: int = 320, inputResW: int = 256):
    """``crop_from_dets`` (dataloader.py:794-835) on the original BGR u8 frame: mean-subtracted RGB crop(s) resized to
    inputResH x inputResW.  frame: numpy [H,W,3] or cuda u8 tensor; boxes: [n,4] frame pixels.
    Returns (inps cuda [n,3,H,W], pt1 cpu [n,2], pt2 cpu [n,2])."""
    import torch
    f = frame_bgr_u8 if hasattr(frame_bgr_u8, "is_cuda") else torch.from_numpy(np.ascontiguousarray(frame_bgr_u8))
    f = f.cuda()
    b = boxes if hasattr(boxes, "is_cuda") else torch.as_tensor(np.asarray(boxes), dtype=torch.float32)
    b = b.float().cuda().reshape(-1, 4)
    n = b.shape[0]
    frames = f.unsqueeze(0).expand(n, -1, -1, -1).contiguous()
    inps, pts = ops.crop(frames, boxes=b, oh=inputResH, ow=inputResW)
    pts = pts.cpu()
    return inps, pts[:, 0:2].clone(), pts[:, 2:4].clone()


def crop_from_dets(img, boxes, inps, pt1, pt2):
    """The reference's signature and side effects (dataloader.py:794-835): ``img`` is the RGB float CHW frame in 0..1
    that ``im_to_torch`` made from the u8 frame; the per-channel means are subtracted from it IN PLACE, ``inps`` /
    ``pt1`` / ``pt2`` (pre-allocated by the caller) are filled per box and returned.  The crop itself runs in the HIP
    kernel on the u8 frame recovered from ``img`` (x * 255 is exact for values that came from u8)."""
    import torch
    frame = (img.detach().cpu() * 255.0).round().clamp_(0, 255).to(torch.uint8)          # RGB u8 CHW
    frame_bgr = frame.flip(0).permute(1, 2, 0).contiguous().numpy()
    img[0].add_(-0.406)
    img[1].add_(-0.457)
    img[2].add_(-0.480)
    out, p1, p2 = crop_from_dets_frame(frame_bgr, boxes, inps.shape[2], inps.shape[3])
    inps.copy_(out.to(inps.device))
    pt1.copy_(p1)
    pt2.copy_(p2)
    return inps, pt1, pt2
