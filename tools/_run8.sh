cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/profile_ops.py 2>&1 | grep -v amdgpu.ids > gpurun_out/per_op_pl.txt; head -40 gpurun_out/per_op_pl.txt; grep "^==" gpurun_out/per_op_pl.txt
python - <<'PY'
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from betapose_amd import ops
dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(0)
for (H, W, Cin, Cout, k, sp) in [(52, 52, 128, 256, 3, 2), (52, 52, 256, 128, 1, 1), (20, 16, 256, 1024, 1, 1), (20, 16, 256, 256, 3, 5), (104,104,64,128,3,1)]:
    x = torch.randn(1, H, W, Cin, generator=g).to(dev); w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k); b = torch.randn(Cout, generator=g)
    r0 = ops.conv2d_nhwc(x, w, b, pad=k // 2, act="leaky", tile="pl64_b3", splits=sp, iters=50)
    r1 = ops.conv2d_nhwc(x, w, b, pad=k // 2, act="leaky", tile="pl64_b3", splits=sp, iters=50, planes=True)
    print("%dx%d %d->%d k%d s%d: no planes %.1f us, with planes %.1f us" % (H, W, Cin, Cout, k, sp, r0[-1] * 1e3, r1[-1] * 1e3))
PY
