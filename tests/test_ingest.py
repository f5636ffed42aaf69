"""BulkVectoriser (caller-side batching): call-count known answers in the style of the reference's
tests/core/vespa_index/test_add_documents_handler.py:221-249, on the `random` fake (CPU) and world-size-2 gloo."""
import os
import socket
from unittest import mock

import numpy as np
import pytest
import torch.multiprocessing as mp

from marqo_amd.ingest import BulkVectoriser
from marqo_amd.s2_inference import s2_inference
from marqo_amd.s2_inference.enums import Modality


def test_one_vectorise_call_per_modality_and_order():
    calls = []

    def fake(model, content, **kw):
        calls.append((kw["modality"], list(content)))
        return np.asarray([[float(len(str(c))), 1.0] for c in content], dtype=np.float32)

    bv = BulkVectoriser("m", "cpu", vectorise_fn=fake)
    for d in range(5):
        bv.add((d, "title"), f"title {d}")
        bv.add((d, "img"), f"http://x/{d}.png", Modality.IMAGE)
        bv.add((d, "body"), f"body of doc {d}")
    assert bv.pending() == 15
    out = bv.flush()
    assert len(calls) == 2 and {c[0] for c in calls} == {Modality.TEXT, Modality.IMAGE}
    assert [len(c[1]) for c in calls] == [10, 5]
    assert set(out) == {(d, f) for d in range(5) for f in ("title", "img", "body")}
    assert out[(3, "body")][0] == len("body of doc 3") and bv.pending() == 0 and bv.flush() == {}


def test_auto_flush_bounds_the_queue():
    fake = mock.MagicMock(side_effect=lambda m, c, **k: np.zeros((len(c), 4), np.float32))
    bv = BulkVectoriser("m", "cpu", max_pending=4, vectorise_fn=fake)
    for i in range(10):
        bv.add(i, f"t{i}")
    assert fake.call_count == 2 and bv.pending() == 2
    assert len(bv.flush()) == 10


def test_against_the_random_model_through_vectorise_ndarray():
    bv = BulkVectoriser("random/small", "cpu")
    for i in range(6):
        bv.add(i, f"text {i}")
    out = bv.flush()
    ref = s2_inference.vectorise_ndarray("random/small", [f"text {i}" for i in range(6)], device="cpu")
    assert np.allclose(np.stack([out[i] for i in range(6)]), ref)


def test_failed_flush_keeps_the_queue():
    """a vectorise error must not lose queued items (of either modality): they stay queued and can be flushed again"""
    state = {"fail": True}

    def flaky(model, content, **kw):
        if kw["modality"] == Modality.IMAGE and state["fail"]:
            raise OSError("image file is truncated")
        return np.asarray([[float(len(str(c)))] for c in content], dtype=np.float32)

    bv = BulkVectoriser("m", "cpu", vectorise_fn=flaky)
    bv.add("t0", "hello")
    bv.add("i0", "bad.png", Modality.IMAGE)
    bv.add("i1", "good.png", Modality.IMAGE)
    with pytest.raises(OSError):
        bv.flush()
    assert bv.pending() == 2                      # the text was encoded; both images are still queued, in order
    assert bv.discard("i0") == 1
    state["fail"] = False
    out = bv.flush()
    assert set(out) == {"t0", "i1"} and out["i1"][0] == len("good.png") and bv.pending() == 0


def test_shard_plans():
    from marqo_amd.ingest import estimate_tokens
    from marqo_amd.parallel import balanced_shards, contiguous_shards
    import torch
    costs = [estimate_tokens("x" * n) for n in (400, 8, 8, 8, 300, 8, 8, 120, 8, 8, 8, 8, 8)]
    plan = balanced_shards(costs, 2)
    assert sorted(plan.items[0] + plan.items[1]) == list(range(13)) and plan.n_items == 13
    loads = [sum(costs[i] for i in part) for part in plan.items]
    assert abs(loads[0] - loads[1]) <= max(costs) * 0.35      # token rows per rank are balanced although the item counts are not
    assert balanced_shards(costs, 2).items == plan.items   # deterministic
    assert balanced_shards([estimate_tokens("x" * n) for n in [800] + [8] * 12], 2).counts == [1, 12]   # one long text vs twelve short ones
    rows = torch.arange(13.0).unsqueeze(1)
    gathered = torch.cat([rows[plan.items[0]], rows[plan.items[1]]])
    assert torch.equal(plan.restore(gathered), rows)
    assert balanced_shards([], 3).counts == [0, 0, 0] and balanced_shards([1.0], 1).items == [[0]]
    c = contiguous_shards(7, 2)
    assert c.items == [[0, 1, 2, 3], [4, 5, 6]] and torch.equal(c.restore(rows[:7]), rows[:7])


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seen = []

        def fake(model, content, **kw):  # embedding of item i = [i, i, i] -> order is checkable after the gather
            seen.append(list(content))
            return np.asarray([[float(c.split()[-1])] * 3 for c in content], dtype=np.float32)

        bv = BulkVectoriser("m", "cpu", vectorise_fn=fake)
        for i in range(7):
            bv.add(i, f"item {i}")
        out = bv.flush()
        ok = all(np.allclose(out[i], [i, i, i]) for i in range(7)) and len(seen) == 1 and len(seen[0]) == (4 if rank == 0 else 3)
        # ragged texts: shards balanced by estimated tokens, results still in request order; images: contiguous halves
        texts = [("w " * (1 + 37 * (i % 5))).strip() + f" {i}" for i in range(11)]
        for i, t in enumerate(texts):
            bv.add(("t", i), t)
        for i in range(5):
            bv.add(("i", i), f"img {100 + i}", Modality.IMAGE)
        seen.clear()
        out = bv.flush()
        ok = ok and all(np.allclose(out[("t", i)], [i] * 3) for i in range(11)) and all(np.allclose(out[("i", i)], [100 + i] * 3) for i in range(5))
        from marqo_amd.ingest import estimate_tokens
        from marqo_amd.parallel import balanced_shards
        plan = balanced_shards([estimate_tokens(t) for t in texts], world)
        ok = ok and seen[0] == [texts[i] for i in plan.items[rank]] and seen[1] == [f"img {100 + i}" for i in (range(0, 3) if rank == 0 else range(3, 5))]
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_sharded_flush_world_size_2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [(0, True), (1, True)]


def _worker_requests(rank, world, port, q, merge_images=0):
    """RequestShardedIngest: ranks own disjoint REQUESTS (nothing inside a request is sharded), one gather onto rank 0"""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from marqo_amd.ingest import RequestShardedIngest
        calls = []

        def fake(model, content, **kw):
            calls.append((kw.get("modality"), list(content)))
            return np.asarray([[float(str(c).split()[-1])] * 4 for c in content], dtype=np.float32)

        ing = RequestShardedIngest("m", "cpu", vectorise_fn=fake, merge_images=merge_images, merge_deadline_ms=0)
        n_req = 7   # ragged: rank 0 owns 4 requests, rank 1 owns 3; requests of different sizes
        for i in range(n_req):
            if not ing.owns(i):
                continue
            items = [((i, "t", j), f"text {1000 * i + j}", Modality.TEXT) for j in range(3 + i % 2)]
            items += [((i, "i", j), f"img {1000 * i + 500 + j}", Modality.IMAGE) for j in range(2)]
            ing.submit(i, items)
        ok = ing.touched == [i for i in range(n_req) if i % world == rank]            # no rank touched a request it does not own
        ok = ok and all(all(int(c.split()[-1]) // 1000 % world == rank for c in content) for _, content in calls)
        if merge_images == 0:
            ok = ok and len(calls) == 2 * len(ing.touched)                               # one call per modality per OWNED request
        else:   # merged: groups of >= 4 images = 2 requests; the odd last request of rank 1 waits for collect()
            ok = ok and ing.groups_launched == ([[0, 2], [4, 6]] if rank == 0 else [[1, 3]]) and len(calls) == 2 * len(ing.groups_launched)
        try:
            ing.submit(rank + 1, [((0, "t", 0), "text 1", Modality.TEXT)])                # a foreign request is refused
            ok = False
        except ValueError:
            pass
        rows = ing.collect()
        if rank == 0:
            ok = ok and sorted(rows) == list(range(n_req))
            for i in range(n_req):
                ok = ok and len(rows[i]) == 3 + i % 2 + 2
                ok = ok and all(np.allclose(rows[i][(i, "t", j)], [1000 * i + j] * 4) for j in range(3 + i % 2))
                ok = ok and all(np.allclose(rows[i][(i, "i", j)], [1000 * i + 500 + j] * 4) for j in range(2))
        else:
            ok = ok and rows == {}                                                        # gather, not all_gather: nothing lands here
        ok = ok and ing.collect() == {}                                                   # the store is reset (an empty collect is still collective-safe)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("merge_images", [0, 4])
def test_request_sharded_ingest_world_size_2(merge_images):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_requests, args=(r, 2, port, q, merge_images)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [(0, True), (1, True)]


def _worker_failure(rank, world, port, q):
    """one rank's shard fails to encode: EVERY rank must raise (nobody is left waiting in the all_gather), every rank re-queues, and the
    flush succeeds on all ranks once the offending item is discarded"""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from marqo_amd.ingest import PeerShardError

        def fake(model, content, **kw):
            if any("bad" in c for c in content):
                raise ValueError("undecodable item")
            return np.asarray([[float(c.split()[-1])] * 5 for c in content], dtype=np.float32)   # width 5: NOT what a registry would say

        bv = BulkVectoriser("m", "cpu", vectorise_fn=fake)
        for i in range(6):
            bv.add(i, f"img {i}" if i != 4 else "bad 4", Modality.IMAGE)   # contiguous halves: item 4 is in rank 1's shard
        err = None
        try:
            bv.flush()
        except BaseException as e:  # noqa: BLE001
            err = e
        ok = isinstance(err, ValueError) if rank == 1 else isinstance(err, PeerShardError)
        ok = ok and bv.pending() == 6                       # re-queued on every rank
        ok = ok and bv.discard(4) == 1
        out = bv.flush()
        ok = ok and sorted(out) == [0, 1, 2, 3, 5] and all(np.allclose(out[i], [i] * 5) for i in out)
        # a rank with an EMPTY shard takes the width from its peers (agreed in the same all_reduce), not from a registry entry
        bv.add("only", "img 7", Modality.IMAGE)
        out = bv.flush()
        ok = ok and np.allclose(out["only"], [7] * 5)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_failed_shard_raises_on_every_rank_world_size_2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_failure, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == [(0, True), (1, True)]


def test_a_failed_request_does_not_poison_the_stream():
    """an undecodable image in request 1 must fail request 1 only — request 2 on the same rank encodes from an empty queue,
    no stale rows leak into it, and collect() reports the failed index"""
    from marqo_amd.ingest import RequestShardedIngest
    calls = []

    def flaky(model, content, **kw):
        calls.append((kw["modality"], list(content)))
        if any(c == "BAD" for c in content):
            raise OSError("image file is truncated")
        return np.asarray([[float(len(str(c))), float(kw["modality"] == Modality.IMAGE)] for c in content], dtype=np.float32)

    ing = RequestShardedIngest("m", "cpu", vectorise_fn=flaky, merge_images=0)
    ing.submit(0, [((0, "t"), "hello", Modality.TEXT), ((0, "i"), "img0", Modality.IMAGE)])
    with pytest.raises(OSError):
        ing.submit(1, [((1, "t"), "text of the bad request", Modality.TEXT), ((1, "i"), "BAD", Modality.IMAGE)])
    assert ing._bulk.pending() == 0 and ing._bulk.flush() == {}
    n_calls = len(calls)
    ing.submit(2, [((2, "t"), "fine", Modality.TEXT), ((2, "i"), "img2", Modality.IMAGE)])
    # request 2 ran exactly its own two items (one call per modality), nothing of request 1 was re-encoded with it
    assert [c[1] for c in calls[n_calls:]] == [["fine"], ["img2"]]
    rows = ing.collect()
    assert sorted(rows) == [0, 2] and set(rows[2]) == {(2, "t"), (2, "i")} and rows[2][(2, "t")][0] == 4.0
    assert ing.failed_requests == [1] and ing.failed == []
    assert ing.collect() == {} and ing.failed_requests == []


def test_one_request_stays_in_flight_and_a_late_device_error_fails_only_that_request():
    """submit(i) enqueues request i and only then copies request i - 1's rows to the host (its host work overlaps the previous request's kernels).
    Rows come out in submission order whatever the depth; an error that surfaces at the deferred copy (an asynchronous device fault) is charged to
    ITS request — recorded when the next one is submitted, raised by drain() — and leaves the queue clean"""
    import torch
    from marqo_amd.ingest import RequestShardedIngest

    class _Faulty(torch.Tensor):           # stands for rows still in HBM whose kernels faulted: the copy raises
        def cpu(self, *a, **k):
            raise RuntimeError("HIP error: an illegal memory access was encountered")

    def fn(model, content, **kw):
        rows = torch.tensor([[float(len(str(c))), float(kw["modality"] == Modality.IMAGE)] for c in content])
        return torch.Tensor._make_subclass(_Faulty, rows) if any("FAULT" in str(c) for c in content) else rows

    def request(i, word):
        return [((i, "t"), word, Modality.TEXT), ((i, "i"), word + "-img", Modality.IMAGE)]

    for depth in (1, 0):
        ing = RequestShardedIngest("m", "cpu", vectorise_fn=fn, merge_images=0)
        ing.pipeline_depth = depth
        ing.submit(0, request(0, "a"))
        assert (ing._inflight is not None) == (depth == 1) and len(ing._rows) == (0 if depth else 2)
        ing.submit(1, request(1, "bb"))
        assert len(ing._rows) == (2 if depth else 4)                  # request 0 was filed while request 1 went in flight
        ing.submit(2, request(2, "ccc"))
        rows = ing.collect()
        assert sorted(rows) == [0, 1, 2] and [rows[i][(i, "t")][0] for i in range(3)] == [1.0, 2.0, 3.0] and ing._inflight is None
    # a late fault
    ing = RequestShardedIngest("m", "cpu", vectorise_fn=fn, merge_images=0)
    ing.submit(0, request(0, "ok"))
    ing.submit(1, request(1, "FAULT"))                                # enqueues fine (the fault is asynchronous) ...
    ing.submit(2, request(2, "fine"))                                 # ... and is charged to request 1 here, without failing request 2's call
    assert ing.failed == [1] and isinstance(ing.errors[1], RuntimeError) and ing._bulk.pending() == 0
    ing.submit(3, request(3, "FAULT"))
    with pytest.raises(RuntimeError):
        ing.drain()                                                   # the synchronous form: wait for the request in flight, raise its error
    rows = ing.collect()
    assert sorted(rows) == [0, 2] and ing.failed_requests == [1, 3] and set(rows[2]) == {(2, "t"), (2, "i")}


def test_flush_async_hands_back_what_flush_would():
    from marqo_amd.ingest import BulkVectoriser
    fn = lambda model, content, **kw: np.asarray([[float(len(c))] for c in content], dtype=np.float32)   # noqa: E731
    bulk = BulkVectoriser("m", "cpu", vectorise_fn=fn)
    bulk.add("a", "x")
    bulk.add("b", "yy", Modality.IMAGE)
    h = bulk.flush_async()
    assert bulk.pending() == 0 and bulk.flush() == {}                 # not visible to another flush before result()
    out = h.result()
    assert set(out) == {"a", "b"} and out["b"][0] == 2.0 and h.result() is out


# ---- cross-request micro-batching (merged groups) -------------------------------------------------------------------------------------------
def _req(i, n_text=2, n_img=2, bad=None, fault=None):
    items = [(("t", j), f"text {1000 * i + j}", Modality.TEXT) for j in range(n_text)]        # keys COLLIDE across requests on purpose
    items += [(("i", j), ("BAD" if bad == j else "FAULT" if fault == j else "img") + f" {1000 * i + 500 + j}", Modality.IMAGE) for j in range(n_img)]
    return items


def _numbered(calls):
    def fn(model, content, **kw):
        calls.append((kw["modality"], list(content)))
        if any(str(c).startswith("BAD") for c in content):
            raise OSError("image file is truncated")
        return np.asarray([[float(str(c).split()[-1]), float(kw["modality"] == Modality.IMAGE)] for c in content], dtype=np.float32)
    return fn


def _check_rows(rows, i, n_text=2, n_img=2):
    assert set(rows[i]) == {("t", j) for j in range(n_text)} | {("i", j) for j in range(n_img)}
    assert all(rows[i][("t", j)][0] == 1000 * i + j for j in range(n_text)) and all(rows[i][("i", j)][0] == 1000 * i + 500 + j for j in range(n_img))


def test_requests_merge_until_the_image_target_and_scatter_back_in_order():
    from marqo_amd.ingest import RequestShardedIngest
    calls = []
    ing = RequestShardedIngest("m", "cpu", vectorise_fn=_numbered(calls), merge_images=6, merge_deadline_ms=0)
    for i in range(7):
        ing.submit(i, _req(i, n_text=1 + i % 3))
    # 2 images per request, target 6: groups of 3 requests; request 6 waits for collect()
    assert ing.groups_launched == [[0, 1, 2], [3, 4, 5]] and len(calls) == 4
    assert [len(c[1]) for c in calls if c[0] == Modality.IMAGE] == [6, 6] and [len(c[1]) for c in calls if c[0] == Modality.TEXT] == [6, 6]
    assert len(ing._rows) == 12          # ONE group in flight: the first group was filed when the second one was launched
    rows = ing.collect()
    assert len(calls) == 6 and sorted(rows) == list(range(7))
    for i in range(7):
        _check_rows(rows, i, n_text=1 + i % 3)
    assert [k for k in rows[4]] == [("t", 0), ("t", 1), ("i", 0), ("i", 1)]         # submission order inside a request
    assert ing.failed_requests == [] and ing.collect() == {}


def test_text_only_requests_merge_on_the_token_target():
    from marqo_amd.ingest import RequestShardedIngest, estimate_tokens
    calls = []
    per_request = 4 * estimate_tokens("text 1000")
    ing = RequestShardedIngest("m", "cpu", vectorise_fn=_numbered(calls), merge_images=512, merge_text_tokens=int(2.5 * per_request), merge_deadline_ms=0)
    for i in range(6):
        ing.submit(i, _req(i, n_text=4, n_img=0))
    assert ing.groups_launched == [[0, 1, 2], [3, 4, 5]] and [len(c[1]) for c in calls] == [12, 12]
    rows = ing.collect()
    for i in range(6):
        _check_rows(rows, i, n_text=4, n_img=0)


def test_a_bad_request_inside_a_merged_group_fails_alone():
    """host-side failure of a merged group: its requests are re-run one by one; the bad one is recorded, the others' rows are kept, in order;
    submit() raises only when the bad request is the one being submitted"""
    from marqo_amd.ingest import RequestShardedIngest
    calls = []
    ing = RequestShardedIngest("m", "cpu", vectorise_fn=_numbered(calls), merge_images=6, merge_deadline_ms=0)
    ing.submit(0, _req(0))
    ing.submit(1, _req(1, bad=1))
    ing.submit(2, _req(2))                        # launches [0, 1, 2]: fails, isolated; request 2 itself is fine -> no raise
    assert ing.failed == [1] and isinstance(ing.errors[1], OSError) and ing._bulk.pending() == 0
    assert [i for i, _ in ing._index] == [0] * 4 + [2] * 4
    ing.submit(3, _req(3))
    ing.submit(4, _req(4))
    with pytest.raises(OSError):
        ing.submit(5, _req(5, bad=0))             # the submitting request is the bad one: raised here
    assert ing.failed == [1, 5]
    ing.submit(6, _req(6))
    rows = ing.collect()
    assert sorted(rows) == [0, 2, 3, 4, 6] and ing.failed_requests == [1, 5]
    for i in (0, 2, 3, 4, 6):
        _check_rows(rows, i)
    assert [i for grp in [[0, 2], [3, 4], [6]] for i in grp] == sorted(rows)


def test_a_late_device_fault_in_a_merged_group_is_isolated_and_order_is_kept():
    import torch
    from marqo_amd.ingest import RequestShardedIngest

    class _Faulty(torch.Tensor):
        def cpu(self, *a, **k):
            raise RuntimeError("HIP error: an illegal memory access was encountered")

    def fn(model, content, **kw):
        rows = torch.tensor([[float(str(c).split()[-1]), float(kw["modality"] == Modality.IMAGE)] for c in content])
        return torch.Tensor._make_subclass(_Faulty, rows) if any("FAULT" in str(c) for c in content) else rows

    ing = RequestShardedIngest("m", "cpu", vectorise_fn=fn, merge_images=4, merge_deadline_ms=0)
    for i in range(6):
        ing.submit(i, _req(i, fault=0 if i == 2 else None))      # groups [0,1] [2,3] [4,5]; the fault surfaces when [4,5] is launched
    assert ing.failed == [2] and [i for i, _ in ing._index] == [0] * 4 + [1] * 4 + [3] * 4
    rows = ing.collect()
    assert sorted(rows) == [0, 1, 3, 4, 5] and ing.failed_requests == [2]
    assert [i for i, _ in sorted(((i, 0) for i in rows))] == [0, 1, 3, 4, 5]
    ing.submit(6, _req(6, fault=1))
    ing.submit(7, _req(7))
    with pytest.raises(RuntimeError):
        ing.drain()                                             # the synchronous form raises the group's first error
    assert ing.failed == [6] and sorted(ing.collect()) == [7]


def test_the_deadline_launches_a_partial_group():
    """a slow producer: the waiting request is launched by the deadline thread, not held until the group is full"""
    import time
    from marqo_amd.ingest import RequestShardedIngest
    calls = []
    ing = RequestShardedIngest("m", "cpu", vectorise_fn=_numbered(calls), merge_images=100, merge_deadline_ms=20)
    ing.submit(0, _req(0))
    assert calls == [] and ing.groups_launched == []
    t0 = time.perf_counter()
    while not ing.groups_launched and time.perf_counter() - t0 < 5:
        time.sleep(0.005)
    assert ing.groups_launched == [[0]] and len(calls) == 2 and time.perf_counter() - t0 < 2
    ing.submit(1, _req(1))
    ing.submit(2, _req(2))                        # both arrive inside one window: merged
    rows = ing.collect()
    assert sorted(rows) == [0, 1, 2] and len(calls) == 4
    for i in range(3):
        _check_rows(rows, i)
    ing.close()
    ing._deadline_thread.join(2)
    assert not ing._deadline_thread.is_alive()


def test_two_groups_in_flight_keep_submission_order():
    """pipeline_depth = 2: rows of group g are filed when group g + 2 is launched; order and failure isolation as with one group in flight"""
    from marqo_amd.ingest import RequestShardedIngest
    calls = []
    ing = RequestShardedIngest("m", "cpu", vectorise_fn=_numbered(calls), merge_images=4, merge_deadline_ms=0)
    ing.pipeline_depth = 2
    for i in range(8):
        if i == 3:
            with pytest.raises(OSError):              # request 3 launches its own group and is the bad one: raised to its submitter
                ing.submit(i, _req(i, bad=0))
        else:
            ing.submit(i, _req(i))
        assert len(ing._inflight_q) <= 2
    # groups [0,1] [2,3] [4,5] [6,7]; [2,3] fails on the host when it is launched: the group in flight is settled first, then 2 runs alone, 3 is recorded
    assert ing.groups_launched == [[0, 1], [2, 3], [4, 5], [6, 7]] and ing.failed == [3]
    assert [i for i, _ in ing._index] == [0] * 4 + [1] * 4 + [2] * 4          # [4,5] and [6,7] are still in flight
    rows = ing.collect()
    assert sorted(rows) == [0, 1, 2, 4, 5, 6, 7] and ing.failed_requests == [3] and ing._inflight is None
    for i in rows:
        _check_rows(rows, i)


def test_settle_overlaps_the_next_launch_without_changing_order_or_isolation():
    """on a GPU the launch of group g + 1 runs on a launcher thread while the submitting thread files group g's rows: same order, same rows, same
    failure isolation (forced here on the CPU fake: the two-thread predicate is what gates it)"""
    from marqo_amd.ingest import RequestShardedIngest
    calls = []
    ing = RequestShardedIngest("m", "cpu", vectorise_fn=_numbered(calls), merge_images=4, merge_deadline_ms=0)
    ing._bulk._two_threads = lambda: True
    assert ing.overlap_settle
    for i in range(9):
        if i == 5:
            with pytest.raises(OSError):
                ing.submit(i, _req(i, bad=1))          # group [4, 5] fails on the launcher thread: isolated, 5 is the submitter's own
        else:
            ing.submit(i, _req(i))
    assert ing._launcher is not None and ing.failed == [5]
    rows = ing.collect()
    assert sorted(rows) == [0, 1, 2, 3, 4, 6, 7, 8] and ing.failed_requests == [5]
    for i in rows:
        _check_rows(rows, i)


def test_collect_moves_blocks_not_rows():
    """round 6: a rank's rows stay in ONE slab (host) / in the groups' device blocks (RCCL) — settled groups are copied once,
    whole; nothing is stacked per row.  On a NON-root rank collect() is: take the slab, concatenate one position array per request, hand both to
    the collectives.  25 000 rows x 512: that local part takes milliseconds and runs no Python statement per row."""
    import sys
    import time
    from marqo_amd import ingest as I

    D, per_req, n_req = 512, 125, 200
    rng = np.random.default_rng(0)
    table = rng.standard_normal((n_req * per_req, D)).astype(np.float32)

    def fake(model, content, **kw):
        return table[[int(c) for c in content]]

    ing = I.RequestShardedIngest("m", "cpu", vectorise_fn=fake, merge_images=512, merge_deadline_ms=0)
    for i in range(n_req):
        items = []
        for j in range(per_req):          # texts and images interleaved, as a document's fields come
            m = Modality.IMAGE if j % 3 == 0 else Modality.TEXT
            items.append(((i, j), str(i * per_req + j), m))
        ing.submit(i, items)
    ing.drain()
    assert ing._nrows == n_req * per_req and len(ing._perm) == n_req and ing._slab.n == ing._nrows
    assert len(ing._slab.chunks) == 1 and ing._slab.chunks[0].shape[1] == D                           # ONE array (a few for long streams), not a list of rows
    # what a non-root rank does inside collect() before the collectives, with a line counter on: no per-row Python
    lines = [0]

    def tracer(frame, event, arg):
        if event == "line":
            lines[0] += 1
        return tracer
    t0 = time.perf_counter()
    sys.settrace(tracer)
    try:
        index, perms, n = ing._index, ing._perm, ing._nrows
        local = [c[:u] for c, u in zip(ing._slab.chunks, ing._slab.used) if u]
        local = local[0] if len(local) == 1 else np.concatenate(local)
        perm = np.concatenate(perms)
        payload = (n, int(local.shape[1]), (index, perm), [])
    finally:
        sys.settrace(None)
    dt = time.perf_counter() - t0
    assert lines[0] < 50 and dt < 5e-3, (lines[0], dt)
    assert local.flags["C_CONTIGUOUS"] and payload[0] == 25000
    # ... and the root's view is right: every key has its row, in submission order per request
    t0 = time.perf_counter()
    rows = ing.collect()
    t_collect = time.perf_counter() - t0
    assert sorted(rows) == list(range(n_req))
    for i in (0, 57, n_req - 1):
        assert list(rows[i]) == [(i, j) for j in range(per_req)]
        for j in (0, 1, 3, per_req - 1):
            assert np.array_equal(rows[i][(i, j)], table[i * per_req + j])
    assert t_collect < 0.5            # (the root / single rank builds 25 000 dictionary entries: tens of milliseconds)
    assert ing.collect() == {} and ing._nrows == 0 and ing._slab.n == 0


def test_row_slab_spills_into_further_chunks_without_recopying(monkeypatch):
    from marqo_amd import ingest as I
    monkeypatch.setattr(I._RowSlab, "CHUNK_ROWS", 50)
    table = np.arange(400 * 8, dtype=np.float32).reshape(400, 8)
    ing = I.RequestShardedIngest("m", "cpu", vectorise_fn=lambda model, content, **kw: table[[int(c) for c in content]], merge_images=8, merge_deadline_ms=0)
    for i in range(40):
        ing.submit(i, [((i, j), str(i * 10 + j), Modality.IMAGE if j % 2 else Modality.TEXT) for j in range(10)])
    ing.drain()
    assert len(ing._slab.chunks) > 3 and sum(ing._slab.used) == 400 and all(u <= 50 for u in ing._slab.used)
    first = ing._slab.chunks[0]
    rows = ing.collect()
    assert np.shares_memory(rows[0][(0, 0)], first)                       # single rank: views of the chunks, nothing copied
    for i in range(40):
        assert list(rows[i]) == [(i, j) for j in range(10)]
        for j in range(10):
            assert np.array_equal(rows[i][(i, j)], table[i * 10 + j])
