"""Multi-GPU driver pieces: one process per GPU, frames sharded by image.

There is no cross-frame state on the path, so the only communication is
  * a broadcast of the two fp32 weight streams from rank 0 at start-up (RCCL on GPUs, gloo in CPU tests),
  * a gather of the fixed-size per-frame result records (316 floats) at the end of the stream,
  * barriers around timed regions.
Frame i of the (sorted) list goes to rank ``i % world`` -- round-robin keeps ranks balanced on a streamed
split.  Metrics are computed on rank 0 from the gathered records, so they are identical to a 1-GPU run.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np


def limit_host_threads(n: int = 4) -> int:
    """The host side of the path is a handful of small tensor ops per frame (resize to tensor, box arithmetic, key-point
    decoding) issued from several Python threads.  torch's default intra-op pool has one thread per core; on a
    256-thread host every tiny op then fans out and the stage threads fight over the pool -- measured 5.8 frames/s for
    the staged harness against 140 with the pool capped.  Caps the pool at ``n`` unless OMP_NUM_THREADS says otherwise.
    Returns the resulting thread count."""
    import os
    import torch
    if "OMP_NUM_THREADS" not in os.environ and torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return torch.get_num_threads()


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Indices (into the sorted frame list) this rank processes."""
    if not (0 <= rank < world):
        raise ValueError("rank %d not in [0, %d)" % (rank, world))
    return list(range(rank, n_items, world))


def owner_of(index: int, world: int) -> int:
    return index % world


def _dist():
    import torch.distributed as dist
    return dist


def is_dist() -> bool:
    dist = _dist()
    return dist.is_available() and dist.is_initialized()


def init_from_env():
    """Join the job ``torch.distributed.run`` started (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*): binds this
    process to its GPU and, for world > 1, initialises the process group -- RCCL (``nccl``) over xGMI by default;
    ``BP_DIST_BACKEND=gloo`` routes the (tiny) collectives through the host instead, which also lets several ranks
    share one GPU on a single-GPU box.  Returns ``(rank, world, device_index)``."""
    import os
    import torch
    dist = _dist()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("BP_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev <= 0:
        raise RuntimeError("no GPU visible to rank %d" % rank)
    if backend == "nccl" and world > 1 and local >= ndev:
        raise RuntimeError("LOCAL_RANK %d but only %d GPUs visible (one process per GPU)" % (local, ndev))
    device = local % ndev
    torch.cuda.set_device(device)
    if world > 1 and not is_dist():
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, device


def barrier() -> None:
    if is_dist() and _dist().get_world_size() > 1:
        _dist().barrier()


def max_over_ranks(x: float) -> float:
    """The slowest rank's value (timed regions are reported as the max over ranks)."""
    import torch
    dist = _dist()
    if not is_dist() or dist.get_world_size() == 1:
        return float(x)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def gather_floats(x: float) -> List[float]:
    """Every rank's value, in rank order (per-rank throughput lines of the bench)."""
    import torch
    dist = _dist()
    if not is_dist() or dist.get_world_size() == 1:
        return [float(x)]
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([x], dtype=torch.float64, device=dev)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(v[0]) for v in out]


def finalize() -> None:
    if is_dist():
        _dist().barrier()
        _dist().destroy_process_group()


def broadcast_stream(stream: Optional[np.ndarray], src: int = 0, device=None) -> np.ndarray:
    """Broadcast a flat fp32 weight stream (rank ``src`` passes the array, others ``None``)."""
    import torch
    dist = _dist()
    if not is_dist() or dist.get_world_size() == 1:
        return np.ascontiguousarray(stream, dtype=np.float32)
    rank = dist.get_rank()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    n = torch.tensor([stream.size if rank == src else 0], dtype=torch.long, device=dev)
    dist.broadcast(n, src)
    t = torch.from_numpy(np.ascontiguousarray(stream, dtype=np.float32)).to(dev) if rank == src \
        else torch.empty(int(n[0]), dtype=torch.float32, device=dev)
    dist.broadcast(t, src)
    return t.cpu().numpy()


def gather_records(local_records: np.ndarray, local_indices: Sequence[int], n_total: int, dst: int = 0, device=None):
    """Gather per-frame records [n_local, R] to ``dst`` and put them back in global frame order.
    Returns [n_total, R] on ``dst`` (None elsewhere)."""
    import torch
    dist = _dist()
    local_records = np.ascontiguousarray(local_records, dtype=np.float32)
    if not is_dist() or dist.get_world_size() == 1:
        out = np.zeros((n_total, local_records.shape[1]), np.float32)
        out[list(local_indices)] = local_records
        return out
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    R = local_records.shape[1]
    per = (n_total + world - 1) // world           # pad every rank to the same count
    # column R carries (global index + 1) as INT32 BITS inside the fp32 buffer (0 = padding row): exact for every index a
    # 32-bit count can name, where the fp32 VALUE round 2 sent was exact only below 2^24 frames
    buf = torch.zeros((per, R + 1), dtype=torch.float32, device=dev)
    if len(local_indices):
        buf[:len(local_indices), :R] = torch.from_numpy(local_records).to(dev)
        idx = (torch.tensor(list(local_indices), dtype=torch.int64) + 1).to(torch.int32)
        buf[:len(local_indices), R] = idx.view(torch.float32).to(dev)
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    if rank != dst:
        return None
    out = np.zeros((n_total, R), np.float32)
    for b in bufs:
        b = np.ascontiguousarray(b.cpu().numpy())
        tag = np.ascontiguousarray(b[:, R]).view(np.int32)
        valid = tag > 0
        out[tag[valid].astype(np.int64) - 1] = b[valid, :R]
    return out
