"""``from opt import opt`` -- parses ``sys.argv`` at import time, as the reference's opt.py does (:150)."""
import sys

from betapose_amd import opt as _opt

_opt.parse_args(sys.argv[1:])
opt = _opt.opt
