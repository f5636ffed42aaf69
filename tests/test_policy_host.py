"""The load-time policies of the towers (engine/towers.py: residual-stream type, fp8 block split + MLP-only blocks + the stream inside an fp8
tower) are searches over a measured error.  Here the measurement is replaced by a SYNTHETIC error model on the CPU — a fake tower whose `run()`
returns the reference rotated by an angle that grows with the e4m3 share and with the bf16 stream — so that the search logic itself is checked
without a GPU: budgets are never exceeded, the largest admissible e4m3 share is found, the stream is taken only inside its share of the
budget, the environment overrides work, and a second load decides the same."""
import math
import types

import numpy as np
import pytest
import torch

from marqo_amd.engine import towers


class _Enc:
    def __init__(self, layers, post_ln=0):
        self.layers, self.post_ln = layers, post_ln
        self.precision = towers.L.MQ_PREC_FP8
        self.fp8_first_layer = self.fp8_mlp_extra = 0
        self.residual_stream = 2


class _FakeTower(towers._TowerBase):
    """policy methods of _TowerBase on top of an error model: 1 - cos = block_cost * (sum over e4m3 blocks of a depth weight) + stream_cost"""

    def __init__(self, layers=24, block_cost=1e-4, stream_cost=8e-5, precision="fp8", post_ln=0):
        self.cfg = types.SimpleNamespace(enc=_Enc(layers, post_ln))
        self.precision = precision
        self._fp8 = types.SimpleNamespace(calibrated=False) if precision == "fp8" else None
        self.block_cost, self.stream_cost = block_cost, stream_cost
        self.calls = 0
        g = torch.Generator().manual_seed(0)
        self.ref = torch.nn.functional.normalize(torch.randn(8, 64, generator=g, dtype=torch.float64), dim=-1)
        self.ortho = torch.nn.functional.normalize(torch.randn(8, 64, generator=g, dtype=torch.float64), dim=-1)
        self.ortho = torch.nn.functional.normalize(self.ortho - (self.ortho * self.ref).sum(-1, keepdim=True) * self.ref, dim=-1)

    def calibrate_fp8(self, run, passes=2, margin=1.0):
        self._fp8.calibrated = True

    def error_model(self) -> float:
        e = self.cfg.enc
        err = 0.0
        if e.precision == towers.L.MQ_PREC_FP8:
            first, extra = e.fp8_first_layer, e.fp8_mlp_extra
            for l in range(e.layers):                     # early blocks cost more (their noise is amplified by every later one)
                w = 1.0 + 2.0 * (e.layers - 1 - l) / e.layers
                if l >= first:
                    err += self.block_cost * w
                elif l >= first - extra:
                    err += 0.45 * self.block_cost * w
        if e.residual_stream == 1:
            err += self.stream_cost
        return err

    def run(self):
        self.calls += 1
        err = self.error_model()
        th = math.acos(max(-1.0, 1.0 - err))
        return (math.cos(th) * self.ref + math.sin(th) * self.ortho).float()


def _share(t):
    return (t.cfg.enc.layers - t.fp8_first_layer) + 2.0 / 3.0 * t.fp8_mlp_extra


def test_fp8_policy_search_respects_the_budget_and_is_maximal(monkeypatch):
    monkeypatch.delenv("MARQO_AMD_RESIDUAL_STREAM", raising=False)
    monkeypatch.delenv("MARQO_AMD_FP8_MLP_ONLY", raising=False)
    for layers, cost, stream_cost in ((24, 1e-4, 8e-5), (12, 5e-5, 1e-5), (24, 1e-5, 0.0), (24, 5e-3, 2e-4), (32, 7e-5, 1.7e-4)):
        t = _FakeTower(layers, cost, stream_cost)
        first = t.tune_fp8(t.run, budget=7e-4)
        enc = t.cfg.enc
        assert first == enc.fp8_first_layer == t.fp8_first_layer and enc.fp8_mlp_extra == t.fp8_mlp_extra <= first
        assert t.error_model() <= 7e-4 * (1 + 1e-6) and abs(t.fp8_calibration_error - t.error_model()) < 1e-9
        # the stream is taken exactly when it alone costs at most its share of the budget
        assert (t.residual_stream == "bf16") == (stream_cost <= t.FP8_STREAM_SHARE * 7e-4)
        assert enc.residual_stream == (1 if t.residual_stream == "bf16" else 2)
        # one more e4m3 block (all of it) would break the budget, unless everything already runs on e4m3
        if first > 0:
            enc.fp8_first_layer, enc.fp8_mlp_extra = first - 1, min(t.fp8_mlp_extra, first - 1)
            assert t.error_model() > 7e-4
            enc.fp8_first_layer, enc.fp8_mlp_extra = first, t.fp8_mlp_extra
        # never a smaller e4m3 share than the best one-dimensional split of the trace
        one_d = [tr for tr in t.fp8_policy_trace if tr[1] == 0][0]
        assert _share(t) >= (layers - one_d[0]) - 1e-9
        # a second load decides the same
        t2 = _FakeTower(layers, cost, stream_cost)
        assert t2.tune_fp8(t2.run, budget=7e-4) == first and t2.fp8_mlp_extra == t.fp8_mlp_extra and t2.residual_stream == t.residual_stream
    t = _FakeTower(24, 1e-6, 0.0)
    assert t.tune_fp8(t.run, budget=7e-4) == 0 and t.fp8_all_blocks_error <= 7e-4          # everything fits: every block on e4m3
    t = _FakeTower(24, 1.0, 0.0)
    assert t.tune_fp8(t.run, budget=7e-4) == 24 and t.fp8_mlp_extra == 0                    # nothing fits: every block stays bf16


def test_fp8_policy_environment_overrides(monkeypatch):
    monkeypatch.setenv("MARQO_AMD_RESIDUAL_STREAM", "fp32")
    t = _FakeTower(24, 1e-4, 0.0)
    t.tune_fp8(t.run, budget=7e-4)
    assert t.residual_stream == "fp32" and t.cfg.enc.residual_stream == 2
    monkeypatch.setenv("MARQO_AMD_RESIDUAL_STREAM", "bf16")
    t = _FakeTower(24, 1e-4, 3e-4)                                                          # forced although it eats 3e-4 of the 7e-4
    t.tune_fp8(t.run, budget=7e-4)
    assert t.residual_stream == "bf16" and t.error_model() <= 7e-4 * (1 + 1e-6)
    monkeypatch.setenv("MARQO_AMD_RESIDUAL_STREAM", "auto")
    monkeypatch.setenv("MARQO_AMD_FP8_MLP_ONLY", "0")
    t = _FakeTower(24, 1e-4, 0.0)
    t.tune_fp8(t.run, budget=7e-4)
    assert t.fp8_mlp_extra == 0 and len(t.fp8_policy_trace) == 1
    post = _FakeTower(12, 1e-4, 0.0, post_ln=1)                                             # post-LN towers: split only, fp32 stream
    post.tune_fp8(post.run, budget=7e-4)
    assert post.fp8_mlp_extra == 0 and post.residual_stream == "fp32"
    with pytest.raises(RuntimeError):
        _FakeTower(12, precision="bf16").tune_fp8(lambda: None)


def test_residual_stream_policy_decision(monkeypatch):
    monkeypatch.delenv("MARQO_AMD_RESIDUAL_STREAM", raising=False)
    for post_ln in (0, 1):                                # pre-LN and (round 3) post-LN towers decide the same way
        ok = _FakeTower(12, precision="bf16", stream_cost=6e-5, post_ln=post_ln)
        ok.cfg.enc.precision = towers.L.MQ_PREC_BF16
        assert ok.tune_residual_stream(ok.run) == "bf16" and ok.cfg.enc.residual_stream == 1 and abs(ok.residual_stream_error - 6e-5) < 1e-9
        bad = _FakeTower(12, precision="bf16", stream_cost=7.8e-3, post_ln=post_ln)
        bad.cfg.enc.precision = towers.L.MQ_PREC_BF16
        assert bad.tune_residual_stream(bad.run) == "fp32" and bad.cfg.enc.residual_stream == 2
        assert ok.tune_residual_stream(ok.run, budget=1e-9) == "fp32"
    f8 = _FakeTower(12, precision="fp8")
    assert f8.tune_residual_stream(f8.run) == "fp32" and f8.calls == 0                       # (an fp8 tower decides inside tune_fp8)
    monkeypatch.setenv("MARQO_AMD_RESIDUAL_STREAM", "bf16")
    forced = _FakeTower(12, precision="bf16", stream_cost=1.0)
    assert forced.tune_residual_stream(forced.run) == "bf16" and forced.calls == 0
    monkeypatch.setenv("MARQO_AMD_RESIDUAL_STREAM", "fp32")
    assert forced.tune_residual_stream(forced.run) == "fp32" and forced.cfg.enc.residual_stream == 2


def test_fp8_share_of_a_tower_that_pools_one_row_does_not_credit_the_last_blocks_mlp(monkeypatch):
    """with the pooled-rows-only last block (towers.hip, last_block_selected) that block's out-proj / MLP never run on e4m3, so a
    (split = layers, extra >= 1) candidate must not be credited an fp8 MLP for it, and a whole last block is worth its QKV GEMM only"""
    monkeypatch.delenv("MARQO_AMD_FP8_MLP_ONLY", raising=False)
    t = _FakeTower(24, 1e-4, 0.0)
    assert t._fp8_share(24, 24, 1) == pytest.approx(2.0 / 3.0) and t._fp8_share(24, 20, 3) == pytest.approx(6.0)
    t.pools_one_row = True
    assert t._fp8_share(24, 24, 1) == pytest.approx(0.0) and t._fp8_share(24, 24, 3) == pytest.approx(4.0 / 3.0)
    assert t._fp8_share(24, 23, 0) == pytest.approx(0.25) and t._fp8_share(24, 20, 3) == pytest.approx(5.25)
    first = t.tune_fp8(t.run, budget=7e-4)       # the search still ends inside the budget with the corrected shares
    assert t.error_model() <= 7e-4 * (1 + 1e-6) and first == t.fp8_first_layer
