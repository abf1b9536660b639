"""``from KPD.src.utils.img import im_to_torch`` (+ the frame reader)."""
from betapose_amd.img import im_to_torch, load_frame_bgr  # noqa: F401
