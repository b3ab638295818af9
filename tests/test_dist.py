"""CPU suite: the N > 1 path (sharding of independent texts over ranks, barrier + max-over-ranks
timing, summed bytes) under torch.distributed with the gloo backend, world size 2.  The builder
injected here is the CPU oracle -- the host logic is what is under test; the GPU builder is
covered by the -m gpu suite."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from conftest import Oracle
    from suffix_array_amd import batch, corpus
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle()
    texts = [corpus.uniform(20_000 + 1000 * i, 50 + i) for i in range(5)]
    out, nbytes, dt = batch.run_sharded(texts, orc.sais, dist)
    ok = all(orc.verify(texts[i], sa) == 1 for i, sa in out.items())
    q.put((rank, sorted(out), nbytes, dt, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_assignment():
    from suffix_array_amd import batch
    assert batch.shard(8, 8, 3) == [3]
    assert batch.shard(5, 2, 0) == [0, 2, 4] and batch.shard(5, 2, 1) == [1, 3]
    assert sorted(sum((batch.shard(11, 4, r) for r in range(4)), [])) == list(range(11))


def test_world_size_2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, idx0, b0, t0, ok0), (r1, idx1, b1, t1, ok1) = res
    assert idx0 == [0, 2, 4] and idx1 == [1, 3] and ok0 and ok1
    assert b0 == b1 == sum(20_000 + 1000 * i for i in range(5))      # whole-job bytes on every rank
    assert t0 == t1 > 0                                               # max over ranks, identical everywhere


def test_single_process_path(oracle):
    from suffix_array_amd import batch, corpus
    texts = [corpus.dna(3000, 1), corpus.uniform(10, 2)]
    out, nbytes, dt = batch.run_sharded(texts, oracle.sais)
    assert sorted(out) == [0, 1] and nbytes == 3010 and dt > 0
    assert np.array_equal(out[1], oracle.naive(texts[1]))
