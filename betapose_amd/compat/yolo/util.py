"""``from yolo.util import write_results, dynamic_write_results``."""
from betapose_amd.yolo_util import dynamic_write_results, write_results  # noqa: F401
