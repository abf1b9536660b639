"""``from utils.model import *`` -> ``Model3D``."""
from betapose_amd.metrics import Model3D  # noqa: F401

__all__ = ["Model3D"]
