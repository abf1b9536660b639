"""``from utils.sixd import load_sixd``."""
from betapose_amd.sixd import Benchmark, Frame, load_sixd  # noqa: F401
