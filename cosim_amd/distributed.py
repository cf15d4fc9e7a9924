"""Multi-GPU sharding: one process per GPU, contiguous env blocks, one tiny metrics collective.

The reference is single-process; environments never interact, so sharding needs no data-path
collective.  Rank ``r`` of ``G`` owns global env ids ``[r * N/G, (r+1) * N/G)``; every random stream is keyed by the
*global* env id (``rng.py``), so results do not depend on ``G``.  The only exchange is the per-episode / per-window
metrics that feed ``core/reporter.py`` on rank 0 (reference ``Reporter.write_info``, core/reporter.py:210-218):
an ``all_reduce(SUM)`` of sufficient statistics (count, sum, sum of squares) over ``torch.distributed``
(backend "nccl" == RCCL over xGMI on the GPU node, "gloo" on CPU for the tests).
"""
from __future__ import annotations

import os
from typing import Dict, Tuple

import numpy as np


def shard_range(num_envs_total: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block of global env ids owned by ``rank`` (remainder spread over the first ranks)."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank / world_size")
    base, rem = divmod(num_envs_total, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def init_from_env(backend: str = "nccl"):
    """``torch.distributed`` rendezvous from RANK / WORLD_SIZE / MASTER_* (torchrun contract); returns (rank, world)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


class MetricsAccumulator:
    """Fleet statistics of the reporter's per-step scalars, reducible across ranks.

    Keeps ``count``, ``sum`` and ``sum of squares`` per metric column as one flat tensor so that the cross-GPU exchange
    is a single small all-reduce (a few hundred bytes: latency-bound on xGMI, issued off the critical path).
    """

    def __init__(self, names, device="cpu"):
        import torch
        self.torch = torch
        self.names = list(names)
        self.buf = torch.zeros((3, len(self.names)), dtype=torch.float64, device=device)

    def update(self, values):
        """``values``: tensor ``[N, K]`` (one row per env) of this step's metrics."""
        v = values.to(self.buf.dtype)
        self.buf[0] += v.shape[0]
        self.buf[1] += v.sum(dim=0)
        self.buf[2] += (v * v).sum(dim=0)

    def reduce(self) -> Dict[str, Dict[str, float]]:
        """All-reduce over ranks (no-op for one process) and return mean / std per metric."""
        import torch.distributed as dist
        buf = self.buf.clone()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # fp64 is fine for gloo; RCCL reduces fp64 as well
            dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        b = buf.cpu().numpy()                      # ONE device-to-host copy; the per-metric arithmetic runs on the host
        n = np.maximum(b[0], 1.0)
        mean = b[1] / n
        std = np.sqrt(np.maximum(b[2] / n - mean * mean, 0.0))
        return {k: {"count": float(b[0, i]), "mean": float(mean[i]), "std": float(std[i])} for i, k in enumerate(self.names)}
