"""``from pPose_nms import pose_nms, write_json``."""
from betapose_amd.pPose_nms import pose_nms, write_json  # noqa: F401
