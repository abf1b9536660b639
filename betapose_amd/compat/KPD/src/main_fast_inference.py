"""``from KPD.src.main_fast_inference import *`` -> ``InferenNet_fast``."""
from betapose_amd.kpd import InferenNet_fast  # noqa: F401

__all__ = ["InferenNet_fast"]
