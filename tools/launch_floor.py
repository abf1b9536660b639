#!/usr/bin/env python3
"""What a chain of N dependent, (nearly) empty kernels costs as a hipGraph, on 1..S streams at once -- the floor under
the per-frame launch chain (202 nodes) that no kernel work can go below.  Uses torch only to get tiny kernels."""
import sys
import time

import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 202
dev = torch.device("cuda:0")
for blocks in (1, 256, 1024):
    for S in (1, 2, 4):
        streams = [torch.cuda.Stream() for _ in range(S)]
        graphs, bufs = [], []
        for s in streams:
            x = torch.zeros(blocks * 256, device=dev)
            bufs.append(x)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(s):
                for _ in range(3):
                    x.add_(1.0)
                torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=s):
                    for _ in range(N):
                        x.add_(1.0)
            graphs.append((g, s))
        torch.cuda.synchronize()
        reps = 50
        for _ in range(5):
            for g, s in graphs:
                with torch.cuda.stream(s):
                    g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for g, s in graphs:
                with torch.cuda.stream(s):
                    g.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("blocks=%4d streams=%d: %.1f us per %d-node graph per stream-slot -> %.3f us per node (aggregate), %.0f graphs/s"
              % (blocks, S, dt / reps * 1e6, N, dt / (reps * S * N) * 1e6, reps * S / dt), flush=True)
