"""``from utils.utils import pnp``."""
from betapose_amd.ops import solve_pnp as pnp  # noqa: F401
