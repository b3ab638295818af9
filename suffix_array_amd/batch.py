"""Multi-GPU batch mode: one independent text per device, no data-path collective
(SURVEY.md section 8e).  torch.distributed is only the control plane (barrier, max-over-ranks
time, summed bytes); with backend "nccl" that is RCCL on ROCm, with "gloo" it runs on CPUs."""
from __future__ import annotations

import time
from typing import Callable, List, Sequence


def shard(count: int, world: int, rank: int) -> List[int]:
    """text i is built by rank i mod world (SURVEY.md 8e 'Partitioning')"""
    return [i for i in range(count) if i % world == rank]


def run_sharded(texts: Sequence, build: Callable, dist=None, sync: Callable = lambda: None):
    """Every rank builds its shard of `texts` with `build(text) -> SA`; returns
    (results for this rank as {index: SA}, whole-job bytes, max-over-ranks seconds).
    `dist` is an initialised torch.distributed module or None for a single process."""
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    mine = shard(len(texts), world, rank)
    sync()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    out = {i: build(texts[i]) for i in mine}
    sync()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    nbytes = sum(len(texts[i]) for i in mine)
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        b = torch.tensor([nbytes], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(b, op=dist.ReduceOp.SUM)
        dt, nbytes = float(t.item()), int(b.item())
    return out, nbytes, dt
