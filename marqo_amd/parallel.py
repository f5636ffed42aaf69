"""Multi-GPU sharding of the hot path.

Every embedding is independent (no cross-item reduction, per-row normalisation), so the path shards
trivially: one process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI on ROCm), weights
replicated on every GPU (<= 0.85 GB bf16 for ViT-L/14 — nothing against 288 GB of HBM), the item list
split across ranks, and ONE collective: an all_gather of the [n_i, D] fp32 embedding shards, straight
from HBM, for the final concat.  xGMI is point-to-point, so this is a single fat message per peer pair
(8 GPUs x 38 MB for 100k x 768 fp32) rather than many small ones.  The reference has nothing to
translate here: it only exposes `device="cuda:N"` (src/marqo/tensor_search/utils.py:90-123).

Two ways to split:
  * `shard_bounds`      contiguous, equal item counts (images after the resize all cost the same);
  * `balanced_shards`   ragged text: items are dealt to ranks by estimated cost (token count), longest first onto the least
                        loaded rank, so that every GPU runs the same number of token rows; `ShardPlan.restore` puts the gathered
                        rows back into request order with one index_select.  The plan is a pure function of the costs, so every
                        rank computes the same one and nothing but the embeddings crosses the fabric.
"""
from __future__ import annotations

import heapq
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch


def shard_bounds(n_items: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced [start, stop) per rank (first `n % world` ranks take one extra)."""
    if world < 1:
        raise ValueError("world must be >= 1")
    base, extra = divmod(n_items, world)
    out, start = [], 0
    for r in range(world):
        size = base + (1 if r < extra else 0)
        out.append((start, start + size))
        start += size
    return out


@dataclass
class ShardPlan:
    """which items each rank encodes (`items[r]`, ascending request index) and how to undo it after the gather"""
    items: List[List[int]]
    n_items: int

    @property
    def counts(self) -> List[int]:
        return [len(x) for x in self.items]

    def restore(self, gathered: torch.Tensor) -> torch.Tensor:
        """rows in rank order (rank 0's items, then rank 1's ...) -> rows in request order"""
        order = [i for part in self.items for i in part]
        if order == list(range(self.n_items)):
            return gathered
        inv = np.empty(self.n_items, dtype=np.int64)   # (host index arithmetic in NumPy: no OpenMP region on the request path)
        inv[np.asarray(order, dtype=np.int64)] = np.arange(self.n_items, dtype=np.int64)
        if gathered.device.type == "cpu":
            return torch.from_numpy(np.ascontiguousarray(gathered.numpy()[inv]))
        return gathered.index_select(0, torch.from_numpy(inv).to(gathered.device))


def balanced_shards(costs: Sequence[float], world: int) -> ShardPlan:
    """Longest-processing-time-first assignment of items to ranks by cost (ties broken by index, so the plan is deterministic).
    Guarantees max load <= 4/3 of the optimum; with thousands of short texts per request the loads differ by < 1 item."""
    if world < 1:
        raise ValueError("world must be >= 1")
    n = len(costs)
    if world == 1:
        return ShardPlan([list(range(n))], n)
    heap = [(0.0, r) for r in range(world)]
    items: List[List[int]] = [[] for _ in range(world)]
    for i in sorted(range(n), key=lambda j: (-float(costs[j]), j)):
        load, r = heapq.heappop(heap)
        items[r].append(i)
        heapq.heappush(heap, (load + max(float(costs[i]), 0.0), r))
    for part in items:
        part.sort()
    return ShardPlan(items, n)


def contiguous_shards(n_items: int, world: int) -> ShardPlan:
    return ShardPlan([list(range(a, b)) for a, b in shard_bounds(n_items, world)], n_items)


def gather_embeddings(local: torch.Tensor, counts: Optional[Sequence[int]] = None, force_collective: bool = False) -> torch.Tensor:
    """All-gather row shards [n_r, D] (fp32) from every rank into [sum n_r, D] in rank order — ONE collective.

    counts: per-rank row counts when shards are ragged (known from the shard plan, so no size exchange is needed).  Equal shards
    use a single all_gather_into_tensor; ragged shards are padded to the max count so it is still ONE collective, then trimmed.
    """
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local
    if dist.get_world_size() == 1 and not force_collective:   # (force_collective: run the 1-rank collective anyway, for RCCL tests)
        return local
    world = dist.get_world_size()
    D = local.shape[1]
    if counts is None or len(set(counts)) == 1:
        out = torch.empty(world * local.shape[0], D, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    if len(counts) != world:
        raise ValueError("counts must have one entry per rank")
    mx = max(counts)
    padded = local
    if local.shape[0] != mx:
        padded = torch.zeros(mx, D, dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
    buf = torch.empty(world * mx, D, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(buf, padded.contiguous())
    return torch.cat([buf[r * mx: r * mx + counts[r]] for r in range(world)], dim=0)


def gather_embeddings_to_root(local: torch.Tensor, counts: Sequence[int], root: int = 0) -> Optional[torch.Tensor]:
    """`gather` (not all_gather) of ragged row shards [n_r, D] onto ONE rank: the root receives [sum n_r, D] in rank order, every other
    rank returns None and keeps nothing — the add_documents stream (BASELINE configs[3]) needs the embeddings in one place (the rank that
    feeds the document store), and an all_gather would move world x the bytes over xGMI and make every rank copy the full matrix to its
    host.  counts: rows per rank, known to every rank from the request plan (no size exchange).  Shards are padded to the largest count so
    that it is ONE collective."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    if len(counts) != world:
        raise ValueError("counts must have one entry per rank")
    D, mx = local.shape[1], max(counts)
    padded = local
    if local.shape[0] != mx:
        padded = torch.zeros(mx, D, dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
    bufs = [torch.empty(mx, D, dtype=local.dtype, device=local.device) for _ in range(world)] if rank == root else None
    dist.gather(padded.contiguous(), bufs, dst=root)
    if rank != root:
        return None
    return torch.cat([bufs[r][: counts[r]] for r in range(world)], dim=0)


def agree_on_shard(ok: bool, width: int, device) -> Tuple[bool, int]:
    """One tiny all_reduce in front of a data collective: (every rank's local encode succeeded?, the embedding width as the ranks that
    encoded something saw it).  A rank whose `vectorise` raised must not simply leave: its peers would sit in the all_gather until the
    RCCL timeout.  MIN over [ok, -width]: ok = 0 as soon as one rank failed, width = the largest reported (0 = nobody had items)."""
    import torch.distributed as dist
    t = torch.tensor([1 if ok else 0, -int(width)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    v = t.tolist()
    return bool(v[0]), int(-v[1])
