"""N>1 path on CPU: 2 processes over gloo exercise the frame sharding, the weight broadcast and the record
gather that bench.py / evaluate.py use over RCCL on GPUs."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from betapose_amd import dist as bpd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_indices_partition():
    for n in (0, 1, 7, 64, 1001):
        for world in (1, 2, 3, 8):
            seen = sorted(i for r in range(world) for i in bpd.shard_indices(n, r, world))
            assert seen == list(range(n))
            sizes = [len(bpd.shard_indices(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
            for r in range(world):
                assert all(bpd.owner_of(i, world) == r for i in bpd.shard_indices(n, r, world))


def test_single_process_gather_is_identity():
    rec = np.arange(12, dtype=np.float32).reshape(4, 3)
    out = bpd.gather_records(rec, [0, 1, 2, 3], 4)
    assert np.array_equal(out, rec)


WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, %r)
    from betapose_amd import dist as bpd
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    # weights: only rank 0 has them
    w = np.arange(1000, dtype=np.float32) * 0.5 if rank == 0 else None
    got = bpd.broadcast_stream(w, 0)
    assert got.shape == (1000,) and np.array_equal(got, np.arange(1000, dtype=np.float32) * 0.5)
    # every rank "processes" its shard of 13 frames: record = [frame id, frame id squared, rank]
    n = 13
    idx = bpd.shard_indices(n, rank, world)
    rec = np.array([[i, i * i, rank] for i in idx], np.float32).reshape(-1, 3)
    out = bpd.gather_records(rec, idx, n, dst=0)
    if rank == 0:
        assert out.shape == (n, 3)
        assert np.array_equal(out[:, 0], np.arange(n)) and np.array_equal(out[:, 1], np.arange(n) ** 2)
        assert np.array_equal(out[:, 2], np.arange(n) %% world)
        print("GATHER_OK")
    else:
        assert out is None
    assert bpd.max_over_ranks(1.5 + rank) == 1.5 + (world - 1)      # timed regions: slowest rank
    bpd.barrier()
    bpd.finalize()
    assert not bpd.is_dist()
''')


def test_two_rank_broadcast_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK" in outs[0]


def test_bench_under_the_launcher_fails_loudly_without_a_gpu():
    """The driver's scaling command on a box with no GPU (this container): every rank must stop at the product's own loud error --
    no CPU fall-back, no rendezvous left hanging, nothing that looks like a bench line on stdout.  (With a GPU the same command is
    covered by tests/test_gpu_multirank.py: 1 rank == the plain line, 2 ranks over gloo, 2 ranks over RCCL where 2 GPUs exist.)"""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is visible: the gpu-marked launcher tests cover this command")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BP_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "BetaposeHipError" in r.stderr or "no GPU" in r.stderr or "HIP" in r.stderr, r.stderr[-2000:]
