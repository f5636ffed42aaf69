"""N > 1 path on CPU: world_size-2 `gloo` process group exercising marqo_amd.parallel (the only collective on the path is
the all_gather of embedding shards) with equal and ragged shards."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from marqo_amd.parallel import gather_embeddings, shard_bounds


def test_shard_bounds():
    assert shard_bounds(10, 2) == [(0, 5), (5, 10)]
    assert shard_bounds(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert shard_bounds(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    assert shard_bounds(0, 2) == [(0, 0), (0, 0)]
    b = shard_bounds(100_000, 8)
    assert b[0] == (0, 12500) and b[-1] == (87500, 100000) and sum(e - s for s, e in b) == 100_000
    with pytest.raises(ValueError):
        shard_bounds(3, 0)


def test_gather_is_identity_without_process_group():
    x = torch.arange(6.0).reshape(2, 3)
    assert gather_embeddings(x) is x


def _worker(rank, world, port, n_items, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        D = 8
        full = torch.arange(n_items * D, dtype=torch.float32).reshape(n_items, D)  # "embeddings" of item i = row i
        bounds = shard_bounds(n_items, world)
        s, e = bounds[rank]
        local = full[s:e].clone()
        out = gather_embeddings(local, counts=[b - a for a, b in bounds])
        ok = torch.equal(out, full)
        eq = full[: (n_items // world) * world]
        s2, e2 = shard_bounds(eq.shape[0], world)[rank]
        ok = ok and torch.equal(gather_embeddings(eq[s2:e2].clone()), eq)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [10, 7, 1])
def test_world_size_2_gather_restores_item_order(n_items):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, True), (1, True)]
