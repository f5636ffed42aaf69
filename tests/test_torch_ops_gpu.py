"""torch.ops.marqo_hip.* on the GPU: each op against the C-ABI call it forwards to (bit-identical — same kernel, same arguments) and
against a plain fp32 PyTorch statement of the operation; the towers through both boundaries; stream semantics; argument errors."""
import ctypes as C

import pytest
import torch

from marqo_amd import _lib as L
from oracle import towers as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rand(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def test_building_block_ops_match_the_c_abi_and_fp32_torch():
    ops, lib = L.load_torch_ops(), L.load()
    s = torch.cuda.current_stream().cuda_stream
    M, N, K = 300, 256, 192
    A, W = _rand(M, K, seed=1).to(DEV, torch.bfloat16), _rand(N, K, seed=2, scale=0.05).to(DEV, torch.bfloat16)
    bias, res = _rand(N, seed=3).to(DEV), _rand(M, N, seed=4).to(DEV)
    for flags in (0, L.MQ_EPI_OUT_F32, L.MQ_EPI_BIAS | L.MQ_EPI_GELU, L.MQ_EPI_BIAS | L.MQ_EPI_RESIDUAL | L.MQ_EPI_OUT_F32):
        got = ops.gemm_bf16(A, W, bias if flags & L.MQ_EPI_BIAS else None, res if flags & L.MQ_EPI_RESIDUAL else None, flags)
        want = torch.empty(M, N, dtype=torch.float32 if flags & L.MQ_EPI_OUT_F32 else torch.bfloat16, device=DEV)
        L.check(lib.mq_gemm_bf16(A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), res.data_ptr(), want.data_ptr(), N, M, N, K, flags, s))
        assert got.dtype == want.dtype and torch.equal(got, want), flags
        ref = A.float() @ W.float().t()
        if flags & L.MQ_EPI_BIAS:
            ref = ref + bias
        if flags & L.MQ_EPI_GELU:
            ref = torch.nn.functional.gelu(ref)
        if flags & L.MQ_EPI_RESIDUAL:
            ref = ref + res
        assert torch.allclose(got.float(), ref, atol=2e-2, rtol=2e-2), flags

    x, g, b = _rand(77, 768, seed=5).to(DEV), (_rand(768, seed=6) * 0.1 + 1).to(DEV), _rand(768, seed=7).to(DEV)
    y = ops.layernorm(x, g, b, 1e-5, False)
    assert torch.allclose(y, torch.nn.functional.layer_norm(x, (768,), g, b, 1e-5), atol=2e-5, rtol=1e-5)
    assert ops.layernorm(x, g, b, 1e-5, True).dtype == torch.bfloat16

    e = _rand(9, 512, seed=8).to(DEV)
    assert torch.allclose(ops.l2_normalize(e), torch.nn.functional.normalize(e, dim=-1), atol=1e-6)

    T, heads, Wd = 50, 4, 256
    qkv = _rand(3 * T, 3 * Wd, seed=9).to(DEV, torch.bfloat16)
    out = ops.attention(qkv, None, 3, T, T, heads, L.MQ_MASK_NONE)
    q, k, v = [t.float().reshape(3, T, heads, 64).transpose(1, 2) for t in qkv.split(Wd, dim=1)]
    ref = torch.nn.functional.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(3 * T, Wd)
    assert torch.allclose(out.float(), ref, atol=2e-2, rtol=2e-2)
    cu = torch.tensor([0, 20, 70, 150], dtype=torch.int32, device=DEV)
    ragged = ops.attention(qkv, cu, 3, 0, 80, heads, L.MQ_MASK_CAUSAL)
    assert ragged.shape == (150, Wd) and bool(torch.isfinite(ragged.float()).all())


def _towers(boundary, monkeypatch):
    monkeypatch.setenv("MARQO_AMD_BOUNDARY", boundary)
    from marqo_amd.engine import archs, towers
    return archs, towers


def test_towers_are_bit_identical_through_both_boundaries(monkeypatch):
    """the same HIP launches behind torch.ops.marqo_hip.* (the default) and behind the ctypes binding of the C ABI — batched calls,
    single-request hipGraph replays, image / CLIP-text / BERT towers"""
    from tests import golden_util as G
    results = {}
    for boundary in ("torch_ops", "ctypes"):
        A, T = _towers(boundary, monkeypatch)
        sd, z = G.load("clip_vit_small")
        S, P, W, L_, H, F, D = [int(v) for v in z["cfg"]]
        vit = T.VitTower(A.VitArch(S, P, W, L_, H, F, D), sd, DEV)
        assert (vit._ops is not None) == (boundary == "torch_ops")
        px = torch.from_numpy(z["pixels"]).to(DEV)
        u8 = O.synthetic_images_u8(5, S, seed=3).to(DEV)
        sdt, zt = G.load("clip_text_small")
        V, ctx, Wt, Lt, Ht, Ft, Dt = [int(v) for v in zt["cfg"]]
        txt = T.ClipTextTower(A.ClipTextArch(V, ctx, Wt, Lt, Ht, Ft, Dt), sdt, DEV)
        ids = torch.from_numpy(zt["ids"])
        sdb, zb = G.load("bert_small")
        Vb, Pb, Wb, Lb, Hb, Fb = [int(v) for v in zb["cfg"]]
        bert = T.BertTower(A.BertArch(vocab=Vb, max_pos=Pb, width=Wb, layers=Lb, heads=Hb, mlp_dim=Fb), sdb, DEV, pooling="mean")
        results[boundary] = [vit.encode_f32(px).cpu(), vit.encode_u8(u8).cpu(), vit.encode_u8(u8[:1]).cpu(), txt.encode_ids(ids).cpu(),
                             txt.encode_ids(ids, pack=False).cpu(), txt.encode_ids(ids[:1]).cpu()]
        bids, bmask = torch.from_numpy(zb["ids"]), torch.from_numpy(zb["mask"])
        results[boundary] += [bert.encode_ids(bids, bmask).cpu(), bert.encode_ids(bids[:1], bmask[:1]).cpu()]
    for a, b in zip(results["torch_ops"], results["ctypes"]):
        assert torch.equal(a, b)
    ref = torch.from_numpy(G.load("clip_vit_small")[1]["emb_gelu"])
    cos = torch.nn.functional.cosine_similarity(results["torch_ops"][0].double(), ref.double(), dim=-1)
    assert float((1 - cos).max()) < 3e-4


def test_ops_run_on_the_callers_current_stream():
    ops = L.load_torch_ops()
    x = _rand(4096, 1024, seed=11).to(DEV)
    side = torch.cuda.Stream(DEV)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        y = x * 2.0                      # produced on `side`: an op enqueued on another stream could read it too early
        z = ops.l2_normalize(y)
        done = torch.cuda.Event()
        done.record(side)
    done.synchronize()
    assert torch.allclose(z, torch.nn.functional.normalize(x, dim=-1), atol=1e-6)
    g = torch.cuda.CUDAGraph()           # and they capture into a hipGraph like any aten kernel
    static = x.clone()
    with torch.cuda.graph(g):
        out = ops.l2_normalize(static)
    static.copy_(x * 3.0)
    g.replay()
    torch.cuda.synchronize()
    assert torch.allclose(out, torch.nn.functional.normalize(x, dim=-1), atol=1e-6)


def test_argument_errors_surface_as_exceptions():
    ops = L.load_torch_ops()
    A = torch.ones(8, 100, dtype=torch.bfloat16, device=DEV)   # K must be a multiple of 64: rejected by the C ABI, text from mq_last_error
    with pytest.raises(RuntimeError, match="multiple of 64"):
        ops.gemm_bf16(A, A, None, None, 0)
    with pytest.raises(RuntimeError, match="must be"):
        ops.l2_normalize(torch.ones(2, 8, dtype=torch.float64, device=DEV))
    with pytest.raises(RuntimeError, match="CPU uint8 tensor of"):
        ops.encode_image_u8(torch.zeros(3, dtype=torch.uint8), torch.zeros(3, dtype=torch.uint8),
                            torch.zeros(1, 32, 32, 3, dtype=torch.uint8, device=DEV), torch.zeros(1, 8, device=DEV), True,
                            torch.zeros(16, dtype=torch.uint8, device=DEV))
