"""``from KPD.src.utils.eval import getPrediction``."""
from betapose_amd.eval import getPrediction  # noqa: F401
