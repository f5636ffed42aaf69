"""The native request queue (mq_queue_*, csrc/queue.hip, ABI 14; engine/native_queue.py): concurrent small text calls share tower calls on worker threads
outside the interpreter.  What is checked: a request's rows equal the DIRECT tower call's bit for bit whoever it shared a launch with (same kernel family),
against the golden vectors `transformers` made, under 16 request threads, through `vectorise()`, and that bad requests fail alone.
Reference load being served: 8 indexing + 8 search threads calling vectorise() (/root/reference/src/marqo/api/configs.py:27-28)."""
import ctypes as C
import os
import threading

import numpy as np
import pytest
import torch

from tests import golden_util as G

pytestmark = pytest.mark.gpu

COS_TIGHT = 3e-4


def _cos_err(a, b) -> float:
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((1 - (a * b).sum(-1) / (a.norm(dim=-1) * b.norm(dim=-1))).max())


def _clip_text():
    from marqo_amd.engine import archs as A, towers as T
    sd, z = G.load("clip_text_small")
    V, ctx, W, L, H, F, D = [int(v) for v in z["cfg"]]
    return T.ClipTextTower(A.ClipTextArch(V, ctx, W, L, H, F, D), sd, "cuda"), z


def _bert(pooling="mean"):
    from marqo_amd.engine import archs as A, towers as T
    sd, z = G.load("bert_small")
    V, P, W, L, H, F = [int(v) for v in z["cfg"]]
    return T.BertTower(A.BertArch(vocab=V, max_pos=P, width=W, layers=L, heads=H, mlp_dim=F), sd, "cuda", pooling=pooling), z


def _packed(ids: np.ndarray, lengths: np.ndarray):
    keep = np.arange(ids.shape[1])[None, :] < lengths[:, None]
    return np.ascontiguousarray(ids[keep], dtype=np.int32), np.ascontiguousarray(lengths, dtype=np.int32)


def test_queue_rows_are_the_direct_calls_rows_and_the_golden_ones():
    """one request through mq_queue_encode == the same sequences through mq_encode_clip_text / mq_encode_bert directly (same entry point underneath, same
    packing: bit-identical), and both within the tolerance of the vectors `transformers` produced"""
    from marqo_amd.engine import native_queue as NQ
    tower, z = _clip_text()
    ids = z["ids"].astype(np.int64)
    lengths = ids.argmax(axis=1) + 1
    direct = tower.encode_ids(torch.from_numpy(ids), normalize=False, pack=True).cpu().numpy()
    q = tower._queue(False, clip=True)
    assert isinstance(q, NQ.TextQueue) and q is tower._queue(False, clip=True) and q is not tower._queue(True, clip=True)
    rows = q.encode(*_packed(ids, lengths))
    assert rows.shape == direct.shape and np.array_equal(rows, direct)
    assert _cos_err(rows, z["emb"]) < COS_TIGHT
    st = q.stats()
    assert st["requests"] == 1 and st["calls"] == 1 and st["merged_calls"] == 0 and st["sequences"] == ids.shape[0] and st["failed_calls"] == 0

    for pooling, key in (("mean", "mean_norm"), ("cls", "cls_norm")):
        bert, zb = _bert(pooling)
        bids, mask = zb["ids"].astype(np.int64), zb["mask"].astype(np.int64)
        direct = bert.encode_ids(torch.from_numpy(bids), torch.from_numpy(mask), normalize=True).cpu().numpy()
        rows = bert._queue(True, clip=False).encode(*_packed(bids, mask.sum(axis=1)))
        assert np.array_equal(rows, direct) and _cos_err(rows, zb[key]) < COS_TIGHT


def test_sixteen_threads_share_tower_calls_and_every_request_gets_its_own_rows():
    """16 request threads x 12 requests of 1-4 sequences against ONE queue: every request's rows are those of the same sequences encoded alone (the small-row
    kernel family serves lone and merged calls alike here: bit-identical), the stats show merged tower calls, nothing is lost or swapped"""
    tower, z = _clip_text()
    ids = z["ids"].astype(np.int64)
    n_all = ids.shape[0]
    lengths = ids.argmax(axis=1) + 1
    alone = tower.encode_ids(torch.from_numpy(ids), normalize=True, pack=True).cpu().numpy()
    one_by_one = np.concatenate([tower.encode_ids(torch.from_numpy(ids[i:i + 1]), normalize=True, pack=True).cpu().numpy() for i in range(n_all)])
    q = tower._queue(True, clip=True)
    errs, results = [], {}
    start = threading.Barrier(16)

    def worker(t):
        try:
            rng = np.random.default_rng(t)
            start.wait(30)
            for c in range(12):
                pick = rng.integers(0, n_all, size=int(rng.integers(1, 5)))
                results[(t, c)] = (pick, q.encode(*_packed(ids[pick], lengths[pick])))
        except BaseException as e:  # noqa: BLE001
            errs.append((t, e))
    ts = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not errs, errs
    assert len(results) == 16 * 12
    for pick, rows in results.values():
        assert rows.shape == (len(pick), alone.shape[1])
        assert np.array_equal(rows, alone[pick]) or np.array_equal(rows, one_by_one[pick]) or _cos_err(rows, alone[pick]) < 1e-5
    st = q.stats()
    assert st["requests"] == 16 * 12 and st["failed_calls"] == 0
    assert st["calls"] < st["requests"] and st["merged_calls"] >= 1 and st["max_call_sequences"] > 4, st    # launches were shared
    assert st["max_call_sequences"] <= q.max_seqs


def test_bad_requests_fail_alone_and_on_their_own_thread():
    from marqo_amd import _lib as L
    tower, z = _clip_text()
    ids = z["ids"].astype(np.int64)
    lengths = ids.argmax(axis=1) + 1
    q = tower._queue(True, clip=True)
    good = q.encode(*_packed(ids[:2], lengths[:2]))
    with pytest.raises(L.MarqoHipError, match="outside"):          # an id outside the embedding table would fault on the device for everybody
        bad = ids[:1].copy()
        bad[0, 1] = tower.arch.vocab + 7
        q.encode(*_packed(bad, lengths[:1]))
    with pytest.raises(L.MarqoHipError, match="tokens"):           # longer than the tower's context
        q.encode(np.ones(tower.arch.ctx + 5, dtype=np.int32), np.asarray([tower.arch.ctx + 5], dtype=np.int32))
    with pytest.raises(L.MarqoHipError, match="sequences"):        # more sequences than one merged call carries
        n = q.max_seqs + 1
        q.encode(np.ones(n, dtype=np.int32), np.ones(n, dtype=np.int32))
    with pytest.raises(ValueError):
        q.encode(np.ones(5, dtype=np.int32), np.asarray([4], dtype=np.int32))
    assert q.encode(np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32)).shape == (0, tower.arch.out_dim)
    again = q.encode(*_packed(ids[:2], lengths[:2]))               # the queue is as good as before
    assert np.array_equal(good, again) and q.stats()["failed_calls"] == 0
    # create-time argument checks (no queue comes back)
    lib, h = tower.lib, C.c_void_p()
    cfg = L.QueueCfg(kind=7, device=0, max_seqs=8, max_rows=8 * tower.arch.ctx, normalize=1, depth=1, window_us=0, graphs=0)
    assert lib.mq_queue_create(C.byref(cfg), C.cast(C.byref(tower.cfg), C.c_void_p), C.cast(C.byref(tower.w), C.c_void_p), C.byref(h)) == -1 and not h
    cfg.kind, cfg.max_rows = L.QUEUE_CLIP_TEXT, 3
    assert lib.mq_queue_create(C.byref(cfg), C.cast(C.byref(tower.cfg), C.c_void_p), C.cast(C.byref(tower.w), C.c_void_p), C.byref(h)) == -1 and not h
    assert b"max_rows" in lib.mq_last_error()


def test_destroy_serves_what_is_pending_and_policy_changes_rebuild_the_queue():
    from marqo_amd.engine import native_queue as NQ
    tower, z = _clip_text()
    ids = z["ids"].astype(np.int64)
    lengths = ids.argmax(axis=1) + 1
    q = NQ.TextQueue(tower.lib, 0, tower.cfg, tower.w, 0, tower.arch.out_dim, tower.arch.ctx, True, max_seqs=8, depth=1, window_us=0)
    outs, errs = {}, []

    def worker(t):
        try:
            outs[t] = q.encode(*_packed(ids[t:t + 1], lengths[t:t + 1]))
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
    nt = min(6, ids.shape[0])
    ts = [threading.Thread(target=worker, args=(t,)) for t in range(nt)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    q.close()
    q.close()                                                      # idempotent
    assert not errs and len(outs) == nt
    ref = tower.encode_ids(torch.from_numpy(ids[:nt]), normalize=True).cpu().numpy()
    for t in range(nt):
        assert _cos_err(outs[t], ref[t:t + 1]) < 1e-5
    # a queue closed under a caller (the tower re-creates its queue when its policy changes): that one request takes the regular path, the next a new queue
    qd = tower._queue(True, clip=True)
    qd.close()
    assert tower.queue_rows_ids(ids[:2], True) is None and tower._queue(True, clip=True) is not qd
    assert _cos_err(tower.queue_rows_ids(ids[:2], True), ref[:2]) < 1e-5
    # the tower's queue follows the tower's policy fields: a changed cfg gets a queue of its own (its scratch is sized from the cfg)
    q1 = tower._queue(True, clip=True)
    before = bytes(tower.cfg)
    tower.cfg.enc.residual_stream = 0 if tower.cfg.enc.residual_stream else 1
    try:
        if bytes(tower.cfg) != before:
            q2 = tower._queue(True, clip=True)
            assert q2 is not q1
            assert _cos_err(q2.encode(*_packed(ids[:3], lengths[:3])), ref[:3]) < 1e-3
    finally:
        tower.cfg.enc.residual_stream = 1 - tower.cfg.enc.residual_stream


def test_vectorise_from_request_threads_goes_through_the_queue(monkeypatch):
    """the product path: 12 threads call vectorise() with 1-3 texts each on a registry CLIP model (synthetic weights); the loader's text tower serves them
    through its queue (the Python coalescer steps aside), rows equal the lone calls' within the kernel-family bound, and a LONE single query still takes the
    captured graph"""
    monkeypatch.setenv("MARQO_AMD_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("MARQO_MAX_CUDA_MODEL_MEMORY", "64")
    monkeypatch.delenv("MARQO_AMD_COALESCE_US", raising=False)
    from marqo_amd.s2_inference import coalesce, s2_inference as s2
    from marqo_amd.s2_inference.enums import AvailableModelsKey, Modality
    name = "open_clip/ViT-B-32/laion2b_s34b_b79k"
    props = s2.get_model_properties_from_registry(name)
    kw = dict(device="cuda:0", modality=Modality.TEXT, model_properties=props)
    texts = [f"a photo of object number {i} on a table" for i in range(40)]
    lone = np.concatenate([s2.vectorise_ndarray(name, [t], **kw) for t in texts])
    key = s2._create_model_cache_key(name, "cuda:0", props)
    model = s2.get_available_models()[key][AvailableModelsKey.model]
    assert model.native_queue_takes(texts[:3]) is True and model.native_queue_takes(texts * 3) is False and model.native_queue_takes([object()]) is False
    tower = model.text
    from marqo_amd.engine import native_queue as NQ
    st0 = tower.queue_stats().get(True)
    if NQ.GRAPHS:      # lone single queries go through the queue as well: the worker replays a hipGraph per token count (captured at the second sighting)
        assert st0 is not None and st0["requests"] == len(texts) and st0["merged_calls"] == 0 and st0["graphs"] >= 1 and st0["graph_replays"] >= 1, st0
    else:              # ... or replay the tower's own captured graph, through torch
        assert st0 is None or st0["requests"] == 0
    merged_before = coalesce.get_coalescer().stats["calls"]
    errs, out = [], {}
    start = threading.Barrier(12)

    def worker(t):
        try:
            start.wait(30)
            for c in range(10):
                pick = [(7 * t + 3 * c + j) % len(texts) for j in range(1 + (t + c) % 3)]
                out[(t, c)] = (pick, s2.vectorise_ndarray(name, [texts[i] for i in pick], **kw))
        except BaseException as e:  # noqa: BLE001
            errs.append((t, e))
    ts = [threading.Thread(target=worker, args=(t,)) for t in range(12)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(180)
    assert not errs, errs
    for pick, rows in out.values():
        assert rows.shape == (len(pick), lone.shape[1]) and _cos_err(rows, lone[pick]) < 1e-4
    st = tower.queue_stats()[True]
    base = st0["requests"] if st0 else 0
    assert st["requests"] - base >= 60 and st["failed_calls"] == 0 and st["calls"] - base < st["requests"] - base, st
    assert coalesce.get_coalescer().stats["calls"] == merged_before        # the Python coalescer saw none of them
    # MARQO_AMD_COALESCE_US set explicitly: the operator's choice wins, the coalescer merges as before
    monkeypatch.setenv("MARQO_AMD_COALESCE_US", "500")
    r = s2.vectorise_ndarray(name, texts[:2], **kw)
    assert _cos_err(r, lone[:2]) < 1e-4


def test_a_lone_sequence_replays_a_graph_with_the_eager_calls_bits():
    """graphs = 1: the first call of a token count runs eagerly, the second is captured, later ones are one hipGraphLaunch — the same rows every time,
    the direct call's rows; another token count gets a graph of its own; graphs = 0 never captures"""
    from marqo_amd.engine import native_queue as NQ
    for clip in (True, False):
        if clip:
            tower, z = _clip_text()
            ids = z["ids"].astype(np.int64)
            lengths = ids.argmax(axis=1) + 1
            direct = tower.encode_ids(torch.from_numpy(ids), normalize=True, pack=True).cpu().numpy()
            mk = lambda g: NQ.TextQueue(tower.lib, 0, tower.cfg, tower.w, 0, tower.arch.out_dim, tower.arch.ctx, True, max_seqs=8, depth=1, window_us=0, graphs=g)  # noqa: E731
        else:
            tower, z = _bert("mean")
            ids, mask = z["ids"].astype(np.int64), z["mask"].astype(np.int64)
            lengths = mask.sum(axis=1)
            direct = tower.encode_ids(torch.from_numpy(ids), torch.from_numpy(mask), normalize=True).cpu().numpy()
            mk = lambda g: NQ.TextQueue(tower.lib, 1, tower.cfg, tower.w, 0, tower.out_width, tower.arch.max_pos, True, max_seqs=8, depth=1, window_us=0, graphs=g)  # noqa: E731
        q = mk(True)
        rows = [q.encode(*_packed(ids[0:1], lengths[0:1])) for _ in range(4)]
        st = q.stats()
        assert st["graphs"] == 1 and st["graph_replays"] == 3 and st["calls"] == 4 and st["failed_calls"] == 0, st     # eager, capture + launch, launch, launch
        for r in rows[1:]:
            assert np.array_equal(r, rows[0])
        assert _cos_err(rows[0], direct[0:1]) < 1e-5          # (a lone sequence and the batch may sit in different kernel families: same rows to rounding)
        other = next((i for i in range(1, ids.shape[0]) if lengths[i] != lengths[0]), None)
        if other is not None:
            r2 = [q.encode(*_packed(ids[other:other + 1], lengths[other:other + 1])) for _ in range(3)]
            assert q.stats()["graphs"] == 2 and np.array_equal(r2[0], r2[2]) and _cos_err(r2[0], direct[other:other + 1]) < 1e-5
        two = q.encode(*_packed(ids[0:2], lengths[0:2]))      # a group of two sequences is never a graph
        assert q.stats()["graph_replays"] == (3 if other is None else 5) and _cos_err(two, direct[0:2]) < 1e-5
        q.close()
        q0 = mk(False)
        r0 = [q0.encode(*_packed(ids[0:1], lengths[0:1])) for _ in range(3)]
        assert q0.stats()["graphs"] == 0 and q0.stats()["graph_replays"] == 0 and np.array_equal(r0[0], rows[0])
        q0.close()


def test_preprocessed_images_of_request_threads_share_tower_calls(monkeypatch):
    """MQ_QUEUE_IMAGE_F32: the few tensors `.preprocess` hands a request thread (one document field: add_docs.py:129-141 -> vectorise per field) go to the image
    tower's queue by address; a worker gathers the waiting callers' images into one batch and runs ONE mq_encode_image_f32.  Rows = the batch call's rows;
    larger lists keep the slab path; through vectorise() from 12 threads"""
    monkeypatch.setenv("MARQO_AMD_SYNTHETIC_WEIGHTS", "1")
    monkeypatch.setenv("MARQO_MAX_CUDA_MODEL_MEMORY", "64")
    monkeypatch.delenv("MARQO_AMD_COALESCE_US", raising=False)
    from PIL import Image
    from marqo_amd.engine import native_queue as NQ
    from marqo_amd.s2_inference import coalesce, s2_inference as s2
    from marqo_amd.s2_inference.enums import AvailableModelsKey, Modality
    name = "open_clip/ViT-B-32/laion2b_s34b_b79k"
    props = s2.get_model_properties_from_registry(name)
    model, pre = s2.load_multimodal_model_and_get_preprocessors(name, props, "cuda:0")
    enc = s2.get_available_models()[s2._create_model_cache_key(name, "cuda:0", props)][AvailableModelsKey.model]
    rng = np.random.default_rng(5)
    pil = [Image.fromarray(rng.integers(0, 256, (180 + 5 * i, 240 - 3 * i, 3), dtype=np.uint8)) for i in range(24)]
    views = [pre["image"](p) for p in pil]
    want = enc.encode_image(torch.stack([v.clone() for v in views]))            # the batch form, no queue (a tensor, not a list of views)
    assert enc.native_queue_takes_images(views[:3]) is True and enc.native_queue_takes_images(views) is False
    assert enc.native_queue_takes_images([v.clone() for v in views[:2]]) is False and enc.native_queue_takes_images(views[0]) is False
    got3 = enc.encode_image(views[:3])
    st = enc.vision.queue_stats()[True]
    assert st["requests"] == 1 and st["sequences"] == 3 and st["failed_calls"] == 0
    assert got3.shape == (3, want.shape[1]) and _cos_err(got3, want[:3]) < 1e-4      # (3 images and 24 sit in different kernel families)
    one = [enc.encode_image([views[7]]) for _ in range(3)]                     # a lone image: eager, captured, replayed
    assert np.array_equal(one[0], one[2]) and _cos_err(one[0], want[7:8]) < 1e-4
    if NQ.GRAPHS:
        assert enc.vision.queue_stats()[True]["graph_replays"] >= 2
    big = enc.encode_image(views)                                               # 24 views: the slab path, as before
    assert np.array_equal(big, want) and enc.vision.queue_stats()[True]["requests"] == 4
    kw = dict(device="cuda:0", modality=Modality.IMAGE, model_properties=props)
    before_c = coalesce.get_coalescer().stats["calls"]
    errs, out = [], {}
    start = threading.Barrier(12)

    def worker(t):
        try:
            start.wait(30)
            for c in range(8):
                pick = [(5 * t + 3 * c + j) % len(views) for j in range(1 + (t + c) % 3)]
                out[(t, c)] = (pick, s2.vectorise_ndarray(name, [views[i] for i in pick], **kw))
        except BaseException as e:  # noqa: BLE001
            errs.append((t, e))
    ts = [threading.Thread(target=worker, args=(t,)) for t in range(12)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(180)
    assert not errs, errs
    for pick, rows in out.values():
        assert rows.shape == (len(pick), want.shape[1]) and _cos_err(rows, want[pick]) < 1e-4
    st = enc.vision.queue_stats()[True]
    assert st["requests"] == 4 + 96 and st["failed_calls"] == 0 and st["merged_calls"] >= 1 and st["calls"] < st["requests"], st
    assert coalesce.get_coalescer().stats["calls"] == before_c                  # the Python coalescer saw none of them


def test_stress_depth_two_with_a_window_and_close_under_load():
    """depth 2 + a 50 us window (the paths the default, depth 1, never takes): 24 threads x 40 requests of 1-6 ragged sequences — every row right, both
    workers used; then the queue is closed while 8 threads keep calling: no caller hangs, each call either returns its rows or fails with the queue's
    'being destroyed' / 'null queue' error"""
    from marqo_amd import _lib as L
    from marqo_amd.engine import native_queue as NQ
    tower, z = _bert("mean")
    ids, mask = z["ids"].astype(np.int64), z["mask"].astype(np.int64)
    lengths = mask.sum(axis=1)
    n_all = ids.shape[0]
    ref = tower.encode_ids(torch.from_numpy(ids), torch.from_numpy(mask), normalize=True).cpu().numpy()
    q = NQ.TextQueue(tower.lib, 1, tower.cfg, tower.w, 0, tower.out_width, tower.arch.max_pos, True, max_seqs=16, depth=2, window_us=50, graphs=True)
    errs, bad = [], []
    start = threading.Barrier(24)

    def worker(t):
        try:
            rng = np.random.default_rng(100 + t)
            start.wait(30)
            for _ in range(40):
                pick = rng.integers(0, n_all, size=int(rng.integers(1, 7)))
                rows = q.encode(*_packed(ids[pick], lengths[pick]))
                if rows.shape != (len(pick), ref.shape[1]) or _cos_err(rows, ref[pick]) > 1e-5:
                    bad.append(pick)
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=worker, args=(t,)) for t in range(24)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(180)
    assert not errs and not bad and not any(t.is_alive() for t in ts), (errs[:1], bad[:1])
    st = q.stats()
    assert st["requests"] == 24 * 40 and st["failed_calls"] == 0 and st["merged_calls"] >= 1 and st["max_call_sequences"] <= 16, st

    stop, served, refused, other = threading.Event(), [0], [0], []

    def hammer(t):
        rng = np.random.default_rng(t)
        while not stop.is_set():
            pick = rng.integers(0, n_all, size=2)
            try:
                q.encode(*_packed(ids[pick], lengths[pick]))
                served[0] += 1
            except L.MarqoHipError as e:
                if NQ.gone(e):
                    refused[0] += 1
                else:
                    other.append(e)
    hs = [threading.Thread(target=hammer, args=(t,)) for t in range(8)]
    for t in hs:
        t.start()
    import time
    time.sleep(0.05)
    q.close()                      # serves what is pending, joins the workers; callers that arrive later are refused
    time.sleep(0.05)
    stop.set()
    for t in hs:
        t.join(60)
    assert not any(t.is_alive() for t in hs) and not other and served[0] > 0 and refused[0] > 0, (served, refused, other[:1])
