"""``from yolo.darknet import Darknet``."""
from betapose_amd.cfg import parse_cfg  # noqa: F401
from betapose_amd.darknet import Darknet  # noqa: F401
