import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import torch.nn.functional as F
from betapose_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for (N, H, W, Cin, Cout, tile, sp) in [(1, 80, 64, 64, 64, "halo64", 1), (1, 80, 64, 32, 64, "halo64", 1), (1, 80, 64, 64, 64, "halo64", 2), (1, 80, 63, 64, 64, "halo64", 1), (1, 80, 64, 64, 128, "halo128", 1)]:
    x = torch.randn(N, H, W, Cin, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / np.sqrt(Cin * 9)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, padding=1)
    out = ops.conv2d_nhwc(x.to(dev), w, None, pad=1, tile=tile + "_b3", splits=sp).cpu().permute(0, 3, 1, 2)
    d = (out - ref).abs()
    bad = (d > 1e-4).nonzero()
    print(N, H, W, Cin, Cout, tile, sp, "max", float(d.max()), "nbad", len(bad))
    if len(bad):
        ys = sorted(set(bad[:, 2].tolist())); xs = sorted(set(bad[:, 3].tolist())); cs = sorted(set(bad[:, 1].tolist()))
        print("  rows", ys[:20], len(ys), "cols", xs[:20], len(xs), "ch", cs[:10], len(cs))
        m = (bad[:, 2] * W + bad[:, 3]); print("  m%64:", sorted(set((m % 64).tolist()))[:64])
