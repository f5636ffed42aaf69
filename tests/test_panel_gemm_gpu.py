"""`mq_panel_gemm_ln` (ABI 13, csrc/panel_gemm.hip): the LayerNorm-folded QKV / fc1 GEMMs of the ViT-B/32 image tower with one workgroup per image — against
`mq_gemm_bf16_ln` (bit-identical output: same k order, same epilogue order) and against plain PyTorch fp32.  Reference arithmetic: open_clip's
ResidualAttentionBlock (in_proj(ln_1(x)), gelu(c_fc(ln_2(x)))), reached from
/root/reference/src/marqo/core/inference/embedding_models/open_clip_model.py:249-266."""
import pytest
import torch

from marqo_amd import _lib as L

pytestmark = pytest.mark.gpu

K = 768


def _s():
    return torch.cuda.current_stream().cuda_stream


def _operands(nseq, T, N, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    rows = nseq * T
    x = (torch.randn(rows, K, device="cuda", generator=g) * 1.7 + 0.3).to(torch.bfloat16)
    x[:, 3] += 9.0                                                     # an outlier channel, as the towers' residual streams have
    gam, bet = 1 + 0.2 * torch.randn(K, device="cuda", generator=g), 0.1 * torch.randn(K, device="cuda", generator=g)
    W0 = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    b0 = 0.1 * torch.randn(N, device="cuda", generator=g)
    wf = (W0 * gam.unsqueeze(0)).to(torch.bfloat16)
    bf = (b0 + W0 @ bet).contiguous()
    colsum = wf.float().sum(1).contiguous()
    eps = 1e-5
    xd = x.double()
    stats = torch.stack([xd.mean(1).float(), (1.0 / torch.sqrt(xd.var(1, unbiased=False) + eps)).float()], dim=1).contiguous()
    want = torch.nn.functional.layer_norm(x.float(), (K,), gam, bet, eps) @ W0.t() + b0
    return x, wf, bf, colsum, stats, want


@pytest.mark.parametrize("flags", [L.MQ_EPI_BIAS, L.MQ_EPI_BIAS | L.MQ_EPI_GELU, L.MQ_EPI_BIAS | L.MQ_EPI_QUICKGELU])
@pytest.mark.parametrize("nseq,T,N", [(256, 50, 2304), (256, 50, 3072), (3, 50, 768), (300, 50, 1536), (5, 64, 2304), (7, 17, 3072), (2, 1, 768)])
def test_one_workgroup_per_image_equals_the_tiled_gemm_and_fp32_torch(nseq, T, N, flags):
    lib = L.load()
    rows = nseq * T
    x, wf, bf, colsum, stats, want = _operands(nseq, T, N, 7 * nseq + T + N)
    ldc = N + 64                                                        # a row stride of its own (the towers' qkv / fc1 buffer is wider than some GEMMs)
    tiled = torch.zeros(rows, ldc, device="cuda", dtype=torch.bfloat16)
    panel = torch.zeros(rows, ldc, device="cuda", dtype=torch.bfloat16)
    L.check(lib.mq_gemm_bf16_ln(x.data_ptr(), K, wf.data_ptr(), K, bf.data_ptr(), colsum.data_ptr(), stats.data_ptr(), tiled.data_ptr(), ldc, rows, N, K, flags, _s()))
    L.check(lib.mq_panel_gemm_ln(x.data_ptr(), wf.data_ptr(), bf.data_ptr(), colsum.data_ptr(), stats.data_ptr(), panel.data_ptr(), ldc, nseq, T, N, K, flags, _s()))
    torch.cuda.synchronize()
    assert torch.equal(panel.view(torch.int16), tiled.view(torch.int16))          # incl. the untouched columns past N
    if flags & L.MQ_EPI_GELU:
        want = torch.nn.functional.gelu(want)
    elif flags & L.MQ_EPI_QUICKGELU:
        want = want * torch.sigmoid(1.702 * want)
    got = panel[:, :N].float()
    assert float((got - want).abs().max() / want.abs().max()) < 2e-2
    cos = torch.nn.functional.cosine_similarity(got, want, dim=-1)
    assert float((1 - cos).max()) < 5e-5


def test_repeatable_and_refusals():
    lib = L.load()
    x, wf, bf, colsum, stats, _ = _operands(40, 50, 2304, 5)
    outs = []
    for _ in range(2):
        o = torch.zeros(2000, 2304, device="cuda", dtype=torch.bfloat16)
        L.check(lib.mq_panel_gemm_ln(x.data_ptr(), wf.data_ptr(), bf.data_ptr(), colsum.data_ptr(), stats.data_ptr(), o.data_ptr(), 2304, 40, 50, 2304, K, L.MQ_EPI_BIAS, _s()))
        outs.append(o)
    torch.cuda.synchronize()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    assert lib.mq_panel_gemm_ln_ok(256, 50, 2304, 768) == 1
    for nseq, T, n, k in [(256, 65, 2304, 768), (256, 50, 2304, 1024), (256, 50, 1000, 768), (0, 50, 768, 768)]:
        assert lib.mq_panel_gemm_ln_ok(nseq, T, n, k) == 0
    o = outs[0]
    assert lib.mq_panel_gemm_ln(x.data_ptr(), wf.data_ptr(), bf.data_ptr(), colsum.data_ptr(), stats.data_ptr(), o.data_ptr(), 2304, 40, 50, 2304, K, L.MQ_EPI_RESIDUAL, _s()) == -1


def test_vit_b32_tower_under_the_knob_has_the_same_bits():
    """200 images under mq_tune("panel_gemm", 192) (one round of the 256 CUs filled to 78 %) run the QKV / fc1 GEMMs one workgroup per image; the same images
    without the knob (the default: the form is slower, profiles/r06za) run the tiled kernels: the same embeddings, bit for bit"""
    from marqo_amd.engine import archs, towers
    from oracle import towers as O
    lib = L.load()
    varch, _ = archs.resolve_open_clip("ViT-B-32")
    cfg = O.VitConfig(varch.image_size, varch.patch_size, varch.width, varch.layers, varch.heads, varch.mlp_dim, varch.out_dim)
    sd = O.synthetic_vit_state_dict(cfg, seed=0)
    u8 = O.synthetic_images_u8(200, varch.image_size, seed=4).to("cuda:0")
    tower = towers.VitTower(varch, sd, "cuda:0")
    tiled = tower.encode_u8(u8).cpu()
    try:
        L.check(lib.mq_tune(b"panel_gemm", 192))
        panel = tower.encode_u8(u8).cpu()
    finally:
        L.check(lib.mq_tune(b"panel_gemm", 0))
    assert torch.equal(panel, tiled)
    ref = O.vit_forward(sd, cfg, O.preprocess_u8_exact_size(u8[:4].cpu()))
    c = torch.nn.functional.cosine_similarity(panel[:4].double(), ref.double(), dim=-1)
    assert float((1 - c).max()) < 1e-3
