"""``from utils.metrics import *`` -> ``add_err``, ``projection_error_2d``, ``iou``, ``rot_error``, ``trans_error``."""
from betapose_amd.metrics import add_err, iou, projection_error_2d, rot_error, trans_error  # noqa: F401

__all__ = ["add_err", "iou", "projection_error_2d", "rot_error", "trans_error"]
