"""The request path's integer bookkeeping (packing right-padded ids, cu_seqlens, call chunking, shard restore) runs in NumPy, not in
PyTorch CPU kernels: boolean-mask indexing / reductions / index_select enter an OpenMP region with torch.get_num_threads() workers that keep
spinning afterwards, and under a container CPU quota (the GPU boxes: 16 CPUs' worth for 128 torch threads) that throttled a bulk-ingest step
from 7 ms to 35 ms (profiles/r02ae_ingest_phases.txt).  These tests pin the NumPy helpers to the plain restatement of what they compute and
keep torch's CPU kernels out of them."""
import numpy as np
import pytest
import torch

from marqo_amd.engine import towers
from marqo_amd.parallel import ShardPlan, balanced_shards


def _ragged(n, S, seed=0):
    rng = np.random.default_rng(seed)
    lengths = rng.integers(1, S + 1, n)
    ids = rng.integers(1, 30000, (n, S))
    ids[np.arange(S)[None, :] >= lengths[:, None]] = 0
    return ids.astype(np.int64), lengths.astype(np.int64)


@pytest.mark.parametrize("n,S", [(1, 1), (1, 77), (7, 12), (128, 77), (300, 512)])
def test_pack_matches_row_by_row_concatenation(n, S):
    ids, lengths = _ragged(n, S, seed=n)
    packed, cu = towers._pack(ids, lengths)
    assert packed.dtype == torch.int32 and cu.dtype == torch.int32 and packed.is_contiguous()
    want = np.concatenate([ids[i, :lengths[i]] for i in range(n)])
    assert np.array_equal(packed.numpy(), want)
    assert np.array_equal(cu.numpy(), np.concatenate([[0], np.cumsum(lengths)]))


def test_host_i64_accepts_tensors_arrays_and_lists():
    for src in (torch.tensor([[1, 2], [3, 4]], dtype=torch.int32), np.array([[1, 2], [3, 4]], dtype=np.int16), [[1, 2], [3, 4]],
                torch.tensor([[1, 2], [3, 4]]).t().t()):
        out = towers._host_i64(src)
        assert isinstance(out, np.ndarray) and out.dtype == np.int64 and out.flags.c_contiguous and out.tolist() == [[1, 2], [3, 4]]


def test_chunks_cover_everything_within_the_row_cap(monkeypatch):
    base = towers._TextTowerBase.__new__(towers._TextTowerBase)
    base.max_rows_per_call = 100
    lengths = np.array([60, 30, 10, 5, 100, 1, 99, 1, 1, 250, 3], dtype=np.int64)   # (a single over-long sequence still forms its own chunk)
    chunks = list(base._chunks(lengths))
    assert chunks[0][0] == 0 and chunks[-1][1] == len(lengths) and all(a[1] == b[0] for a, b in zip(chunks, chunks[1:]))
    for a, b in chunks:
        assert b > a and (int(lengths[a:b].sum()) <= 100 or b - a == 1)
    assert list(base._chunks(np.zeros(0, dtype=np.int64))) == []
    assert list(base._chunks(np.array([5, 5], dtype=np.int64))) == [(0, 2)]
    # pieces come out balanced: 120 equal sequences of 1 row under a cap of 100 -> 60 + 60, not 100 + 20
    assert list(base._chunks(np.ones(120, dtype=np.int64))) == [(0, 60), (60, 120)]
    # the post-LN BERT tower runs smaller calls (class attribute), everything else keeps the large default
    assert towers.BertTower.max_rows_per_call < towers.ClipTextTower.max_rows_per_call == towers.MAX_ROWS_PER_CALL


def test_request_path_helpers_do_not_call_torch_cpu_kernels(monkeypatch):
    """the helpers must not index / reduce torch CPU tensors (from_numpy views are fine)"""
    def boom(*a, **k):
        raise AssertionError("torch CPU kernel on the request path")
    ids, lengths = _ragged(64, 77)
    for name in ("arange", "cumsum", "nonzero", "searchsorted", "zeros", "full", "masked_select", "index_select"):
        monkeypatch.setattr(torch, name, boom)
    monkeypatch.setattr(torch.Tensor, "__getitem__", boom)
    monkeypatch.setattr(torch.Tensor, "cumsum", boom)
    monkeypatch.setattr(torch.Tensor, "sum", boom)
    monkeypatch.setattr(torch.Tensor, "max", boom)
    monkeypatch.setattr(torch.Tensor, "index_select", boom)
    towers._pack(ids, lengths)
    base = towers._TextTowerBase.__new__(towers._TextTowerBase)
    list(base._chunks(lengths))
    plan = balanced_shards([float(x) for x in lengths], 2)
    rows = np.concatenate([np.asarray(part) for part in plan.items]).astype(np.float32)[:, None]
    back = plan.restore(torch.from_numpy(rows))
    assert back.numpy()[:, 0].tolist() == list(range(64))


def test_shard_plan_restore_is_the_inverse_permutation():
    plan = ShardPlan(items=[[1, 3, 4], [0, 2]], n_items=5)
    gathered = torch.from_numpy(np.asarray([[1.0], [3.0], [4.0], [0.0], [2.0]], dtype=np.float32))
    assert plan.restore(gathered)[:, 0].tolist() == [0.0, 1.0, 2.0, 3.0, 4.0]
    ident = ShardPlan(items=[[0, 1], [2]], n_items=3)
    assert ident.restore(gathered[:3]) is not None


def test_cpu_quota_parsing_and_oversubscription_note(tmp_path):
    from marqo_amd import _lib as L
    (tmp_path / "cpu.max").write_text("1600000 100000\n")
    assert L.cpu_quota(str(tmp_path)) == 16.0
    (tmp_path / "cpu.max").write_text("max 100000\n")
    assert L.cpu_quota(str(tmp_path)) is None
    v1 = tmp_path / "v1"
    (v1 / "cpu").mkdir(parents=True)
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("250000\n")
    (v1 / "cpu" / "cpu.cfs_period_us").write_text("100000\n")
    assert L.cpu_quota(str(v1)) == 2.5
    (v1 / "cpu" / "cpu.cfs_quota_us").write_text("-1\n")
    assert L.cpu_quota(str(v1)) is None
    assert L.cpu_quota(str(tmp_path / "missing")) is None
    assert L.oversubscription_note(128, 16.0) and "set_num_threads(16)" in L.oversubscription_note(128, 16.0)
    assert L.oversubscription_note(16, 16.0) is None and L.oversubscription_note(128, None) is None and L.oversubscription_note(4, 0.5) is None


def test_lone_short_text_prefers_the_host_tokenizer(monkeypatch):
    """the search path's single query: host tokenisation (0.01-0.07 ms) beats staging + three launches + a D2H sync (0.12-0.15 ms)"""
    from marqo_amd.engine import gpu_tokenizers as GT
    assert GT.prefers_host(["a photo of a cat"]) and GT.prefers_host(("q",)) and GT.prefers_host([""])
    assert not GT.prefers_host(["a", "b"]) and not GT.prefers_host([]) and not GT.prefers_host(["x" * (GT.HOST_TOKENIZE_MAX_CHARS + 1)])
    monkeypatch.setattr(GT, "HOST_TOKENIZE_MAX_CHARS", -1)    # MARQO_AMD_HOST_TOKENIZE_MAX_CHARS=-1: always the device route
    assert not GT.prefers_host(["q"])
