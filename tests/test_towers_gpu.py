"""Parity of the three towers (HIP, through the C ABI) against (a) the committed golden vectors made
by `transformers` and (b) the CPU fp32 oracle at BASELINE.json's real shapes, on the same seeded inputs.

Tolerance (north_star): cosine >= 1 - 1e-3 per embedding; we also assert a tighter 3e-4 on the
configs measured here so regressions show up early."""
import numpy as np
import pytest
import torch

from oracle import towers as O
from tests import golden_util as G

pytestmark = pytest.mark.gpu

COS_TOL = 1e-3      # north_star tolerance
COS_TIGHT = 3e-4    # what the bf16/fp32-residual design actually holds


def _cos_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().cpu(), b.double().cpu()
    cos = (a * b).sum(-1) / (a.norm(dim=-1) * b.norm(dim=-1))
    return float((1 - cos).max())


def _towers():
    from marqo_amd.engine import towers, archs
    return towers, archs


def test_golden_clip_vit_small():
    T, A = _towers()
    sd, z = G.load("clip_vit_small")
    S, P, W, L, H, F, D = [int(v) for v in z["cfg"]]
    px = torch.from_numpy(z["pixels"])
    for quick, key in ((False, "emb_gelu"), (True, "emb_quick_gelu")):
        tower = T.VitTower(A.VitArch(S, P, W, L, H, F, D, quick_gelu=quick), sd, "cuda")
        out = tower.encode_f32(px, normalize=False)
        ref = torch.from_numpy(z[key])
        assert _cos_err(out, ref) < COS_TIGHT, key
        assert (out.cpu() - ref).abs().max() < 0.03 * ref.abs().max()
        other = torch.from_numpy(z["emb_quick_gelu" if not quick else "emb_gelu"])
        assert _cos_err(out, other) > _cos_err(out, ref)  # the activation flag is honoured


def test_golden_clip_text_small():
    T, A = _towers()
    sd, z = G.load("clip_text_small")
    V, ctx, W, L, H, F, D = [int(v) for v in z["cfg"]]
    tower = T.ClipTextTower(A.ClipTextArch(V, ctx, W, L, H, F, D), sd, "cuda")
    ids = torch.from_numpy(z["ids"])
    ref = torch.from_numpy(z["emb"])
    packed = tower.encode_ids(ids, normalize=False, pack=True)
    padded = tower.encode_ids(ids, normalize=False, pack=False)
    assert _cos_err(packed, ref) < COS_TIGHT
    assert _cos_err(padded, ref) < COS_TIGHT
    assert _cos_err(packed, padded) < 1e-5  # packing to EOT is not an approximation


def test_golden_bert_small():
    T, A = _towers()
    sd, z = G.load("bert_small")
    V, P, W, L, H, F = [int(v) for v in z["cfg"]]
    ids, mask = torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"])
    arch = A.BertArch(vocab=V, max_pos=P, width=W, layers=L, heads=H, mlp_dim=F)
    for pooling, raw, normed in (("mean", "mean", "mean_norm"), ("cls", "cls", "cls_norm")):
        tower = T.BertTower(arch, sd, "cuda", pooling=pooling)
        out = tower.encode_ids(ids, mask, normalize=False)
        assert _cos_err(out, torch.from_numpy(z[raw])) < COS_TIGHT, pooling
        outn = tower.encode_ids(ids, mask, normalize=True)
        assert _cos_err(outn, torch.from_numpy(z[normed])) < COS_TIGHT
        assert torch.allclose(outn.norm(dim=-1).cpu(), torch.ones(ids.shape[0]), atol=1e-5)


def test_vit_b32_full_size_vs_oracle():
    """BASELINE config 2 shape (open_clip ViT-B/32 image tower), synthetic seeded weights, uint8 input."""
    T, A = _towers()
    arch, _ = A.resolve_open_clip("ViT-B-32")
    cfg = O.VitConfig(arch.image_size, arch.patch_size, arch.width, arch.layers, arch.heads, arch.mlp_dim, arch.out_dim)
    sd = O.synthetic_vit_state_dict(cfg, seed=0)
    u8 = O.synthetic_images_u8(6, 224, seed=0)
    ref = O.vit_forward(sd, cfg, O.preprocess_u8_exact_size(u8))
    tower = T.VitTower(arch, sd, "cuda")
    out = tower.encode_u8(u8.cuda())
    assert out.shape == (6, 512)
    assert _cos_err(out, ref) < COS_TIGHT
    assert torch.allclose(out.norm(dim=-1).cpu(), torch.ones(6), atol=1e-5)
    # the float path (what the reference's .preprocess hands over) gives the same embeddings
    out_f = tower.encode_f32(O.preprocess_u8_exact_size(u8))
    assert _cos_err(out_f, out) < 1e-5
    # batching invariance: one image alone == the same image inside a batch
    single = tower.encode_u8(u8[2:3].cuda())
    assert _cos_err(single, out[2:3]) < 1e-4   # a lone image takes the skinny GEMM family: another summation order of a bf16-rounded stream


def test_vit_l14_vs_oracle():
    """BASELINE config 3 image tower shape (ViT-L/14, 257 tokens, K = 588 padded to 640)."""
    T, A = _towers()
    arch, _ = A.resolve_open_clip("ViT-L-14")
    cfg = O.VitConfig(arch.image_size, arch.patch_size, arch.width, arch.layers, arch.heads, arch.mlp_dim, arch.out_dim)
    sd = O.synthetic_vit_state_dict(cfg, seed=1)
    u8 = O.synthetic_images_u8(2, 224, seed=1)
    ref = O.vit_forward(sd, cfg, O.preprocess_u8_exact_size(u8))
    out = T.VitTower(arch, sd, "cuda").encode_u8(u8.cuda())
    assert _cos_err(out, ref) < COS_TIGHT


def test_clip_text_b32_vs_oracle():
    T, A = _towers()
    _, arch = A.resolve_open_clip("ViT-B-32")
    cfg = O.ClipTextConfig(arch.vocab, arch.ctx, arch.width, arch.layers, arch.heads, arch.mlp_dim, arch.out_dim)
    sd = O.synthetic_clip_text_state_dict(cfg, seed=0)
    ids = O.synthetic_clip_ids(8, seed=0)
    ids[0] = O.synthetic_clip_ids(1, seed=9, full_length=True)[0]  # a full 77-token sequence
    ref = O.clip_text_forward(sd, cfg, ids)
    tower = T.ClipTextTower(arch, sd, "cuda")
    assert _cos_err(tower.encode_ids(ids), ref) < COS_TIGHT
    assert _cos_err(tower.encode_ids(ids, pack=False), ref) < COS_TIGHT


def test_e5_base_shape_vs_oracle():
    """BASELINE config 1 inputs: 8 short docs, BERT-base (hf/e5-base-v2 architecture), mean pooling."""
    T, A = _towers()
    arch = A.BertArch()
    cfg = O.BertConfig()
    sd = O.synthetic_bert_state_dict(cfg, seed=0)
    ids, mask = O.synthetic_bert_batch(8, 8, 32, seed=0)
    ref = O.hf_encode(sd, cfg, ids, mask)
    out = T.BertTower(arch, sd, "cuda").encode_ids(ids, mask)
    assert out.shape == (8, 768)
    assert _cos_err(out, ref) < COS_TIGHT
    # a 512-token document (maximum sequence; exercises the 128 KB LDS attention path)
    ids2, mask2 = O.synthetic_bert_batch(2, seed=3, fixed_len=512)
    ref2 = O.hf_encode(sd, cfg, ids2, mask2)
    out2 = T.BertTower(arch, sd, "cuda").encode_ids(ids2, mask2)
    assert _cos_err(out2, ref2) < COS_TOL


def test_tower_input_validation():
    T, A = _towers()
    sd, z = G.load("bert_small")
    V, P, W, L, H, F = [int(v) for v in z["cfg"]]
    tower = T.BertTower(A.BertArch(vocab=V, max_pos=P, width=W, layers=L, heads=H, mlp_dim=F), sd, "cuda")
    ids = torch.ones(2, 4, dtype=torch.int64)
    with pytest.raises(ValueError):
        tower.encode_ids(ids, torch.tensor([[0, 1, 1, 1], [1, 1, 1, 1]]))  # left padding
    with pytest.raises(ValueError):
        tower.encode_ids(ids, torch.zeros(2, 4, dtype=torch.int64))  # empty sequence
    bad = dict(sd)
    bad.pop("embeddings.LayerNorm.weight")
    with pytest.raises(KeyError):
        T.BertTower(A.BertArch(vocab=V, max_pos=P, width=W, layers=L, heads=H, mlp_dim=F), bad, "cuda")
    from marqo_amd._lib import MarqoHipUnavailableError
    with pytest.raises(MarqoHipUnavailableError):
        T.BertTower(A.BertArch(vocab=V, max_pos=P, width=W, layers=L, heads=H, mlp_dim=F), sd, "cpu")


def _tune(key: str, value: int):
    from marqo_amd import _lib as L
    L.check(L.load().mq_tune(key.encode(), value))


def test_row_selected_last_block_is_bit_identical(tiled_gemm_only):
    """The towers run the out-projection / MLP of the LAST block only on the rows that are pooled afterwards
    (class token, EOT, CLS).  That is dead-row elimination, not an approximation: embeddings must be bit-identical to
    the all-rows execution (mq_tune("row_select", 0)), in bf16 and fp8, for every tower that pools single rows."""
    T, A = _towers()
    sd, z = G.load("clip_vit_small")
    S, P, W, L_, H, F, D = [int(v) for v in z["cfg"]]
    px = torch.from_numpy(z["pixels"])
    vit = T.VitTower(A.VitArch(S, P, W, L_, H, F, D), sd, "cuda")
    sdt, zt = G.load("clip_text_small")
    V, ctx, Wt, Lt, Ht, Ft, Dt = [int(v) for v in zt["cfg"]]
    txt = T.ClipTextTower(A.ClipTextArch(V, ctx, Wt, Lt, Ht, Ft, Dt), sdt, "cuda")
    ids = torch.from_numpy(zt["ids"])
    sdb, zb = G.load("bert_small")
    Vb, Pb, Wb, Lb, Hb, Fb = [int(v) for v in zb["cfg"]]
    bert = T.BertTower(A.BertArch(vocab=Vb, max_pos=Pb, width=Wb, layers=Lb, heads=Hb, mlp_dim=Fb), sdb, "cuda", pooling="cls")
    bids, bmask = torch.from_numpy(zb["ids"]), torch.from_numpy(zb["mask"])
    runs = {
        "vit": lambda: vit.encode_f32(px, normalize=False),
        "text_packed": lambda: txt.encode_ids(ids, normalize=False, pack=True),
        "text_padded": lambda: txt.encode_ids(ids, normalize=False, pack=False),
        "bert_cls": lambda: bert.encode_ids(bids, bmask, normalize=False),
    }
    try:
        for name, run in runs.items():
            _tune("row_select", 1)
            sel = run().cpu()
            _tune("row_select", 0)
            full = run().cpu()
            assert torch.isfinite(sel).all()
            assert torch.equal(sel, full), f"{name}: max |diff| = {(sel - full).abs().max().item():.3e}"
    finally:
        _tune("row_select", 1)


def test_row_selected_last_block_full_size_fp8(tiled_gemm_only):
    """Same property at the ViT-B/32 shape on the fp8 path (frozen scales): selected-row and all-row runs agree bit for bit."""
    T, A = _towers()
    arch, _ = A.resolve_open_clip("ViT-B-32")
    cfg = O.VitConfig(arch.image_size, arch.patch_size, arch.width, 2, arch.heads, arch.mlp_dim, arch.out_dim)
    import dataclasses
    arch2 = dataclasses.replace(arch, layers=2)
    sd = O.synthetic_vit_state_dict(cfg, seed=3)
    u8 = O.synthetic_images_u8(9, 224, seed=5).to("cuda")
    for precision in ("bf16", "fp8"):
        tower = T.VitTower(arch2, sd, "cuda", precision=precision)
        if precision == "fp8":
            tower.calibrate_fp8(lambda: tower.encode_u8(u8))
        try:
            _tune("row_select", 1)
            sel = tower.encode_u8(u8).cpu()
            _tune("row_select", 0)
            full = tower.encode_u8(u8).cpu()
        finally:
            _tune("row_select", 1)
        assert torch.equal(sel, full), precision


def test_full_size_batch_properties_config2_and_config3(tiled_gemm_only):
    """BASELINE configs 2 and 3 at their FULL sizes (256 ViT-B/32 images; ViT-L/14 with 128 images + 128 ragged texts), where the
    CPU oracle would take minutes: size-independent properties of an embarrassingly row-parallel map instead —
    every embedding is independent of what else is in the batch (bitwise: permutation equivariance and batch-split invariance),
    unit norm, run-to-run determinism — plus the oracle itself on a small sample of the same batch."""
    T, A = _towers()
    for name, n_img, n_txt, oracle_n in (("ViT-B-32", 256, 0, 4), ("ViT-L-14", 128, 128, 2)):
        varch, tarch = A.resolve_open_clip(name)
        from marqo_amd.engine import synthetic
        sd = synthetic.random_open_clip_state_dict(vision=varch, text=tarch if n_txt else None, seed=0)
        g = torch.Generator().manual_seed(11)
        u8 = torch.randint(0, 256, (n_img, varch.image_size, varch.image_size, 3), generator=g, dtype=torch.uint8)
        vt = T.VitTower(varch, sd, "cuda")
        full = vt.encode_u8(u8.cuda())
        assert torch.equal(full, vt.encode_u8(u8.cuda()))                                   # deterministic
        perm = torch.randperm(n_img, generator=g)
        assert torch.equal(vt.encode_u8(u8[perm].cuda()), full[perm.cuda()])                # permutation equivariance
        halves = torch.cat([vt.encode_u8(u8[:n_img // 2].cuda()), vt.encode_u8(u8[n_img // 2:].cuda())])
        assert torch.equal(halves, full)                                                    # batch-split invariance
        assert torch.equal(vt.encode_u8(u8[5:6].cuda()), full[5:6])                         # a batch of one
        assert torch.allclose(full.norm(dim=-1), torch.ones(n_img, device="cuda"), atol=1e-5)
        cfg = O.VitConfig(varch.image_size, varch.patch_size, varch.width, varch.layers, varch.heads, varch.mlp_dim, varch.out_dim)
        ref = O.vit_forward(sd, cfg, O.preprocess_u8_exact_size(u8[:oracle_n]))
        assert _cos_err(full[:oracle_n], ref) < COS_TIGHT, name
        if n_txt:
            tt = T.ClipTextTower(tarch, sd, "cuda")
            ids = O.synthetic_clip_ids(n_txt, seed=12)                                      # ragged lengths, EOT = max id
            ft = tt.encode_ids(ids)
            assert torch.equal(ft, tt.encode_ids(ids))
            p2 = torch.randperm(n_txt, generator=g)
            assert torch.equal(tt.encode_ids(ids[p2]), ft[p2.cuda()])
            assert torch.equal(torch.cat([tt.encode_ids(ids[:50]), tt.encode_ids(ids[50:])]), ft)
            assert torch.allclose(ft.norm(dim=-1), torch.ones(n_txt, device="cuda"), atol=1e-5)
            tcfg = O.ClipTextConfig(tarch.vocab, tarch.ctx, tarch.width, tarch.layers, tarch.heads, tarch.mlp_dim, tarch.out_dim)
            assert _cos_err(ft[:oracle_n], O.clip_text_forward(sd, tcfg, ids[:oracle_n])) < COS_TIGHT


def test_golden_xlm_roberta_small():
    """multilingual-e5 family (transformers.XLMRobertaModel): BERT encoder, position table used from row 2 on"""
    T, A = _towers()
    sd, z = G.load("xlmr_small")
    V, P, W, L_, H, F = [int(v) for v in z["cfg"]]
    ids, mask = torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"])
    arch = A.BertArch(vocab=V, max_pos=P - 2, width=W, layers=L_, heads=H, mlp_dim=F, ln_eps=1e-5, pos_offset=2)
    tower = T.BertTower(arch, sd, "cuda", pooling="mean")
    assert _cos_err(tower.encode_ids(ids, mask, normalize=False), torch.from_numpy(z["mean"])) < COS_TIGHT
    assert _cos_err(tower.encode_ids(ids, mask, normalize=True), torch.from_numpy(z["mean_norm"])) < COS_TIGHT
    wrong = T.BertTower(A.BertArch(vocab=V, max_pos=P, width=W, layers=L_, heads=H, mlp_dim=F, ln_eps=1e-5, pos_offset=0), sd, "cuda")
    assert _cos_err(wrong.encode_ids(ids, mask, normalize=False), torch.from_numpy(z["mean"])) > 10 * COS_TIGHT  # the offset is honoured


def test_clipa_style_towers():
    """open_clip ViT-L-14-CLIPA-336's forms at a small size: vision = no ln_pre, mean of the PATCH tokens, ln_post after the pooling, projection
    (MQ_VIT_POOL_AVG); text = unmasked transformer with last-position pooling over all ctx positions, un-biased projection — against the
    fp32 oracle, batched and single-item calls, f32 and u8 pixels"""
    T, A = _towers()
    S, P, W, L_, H, F, D = 64, 16, 128, 2, 2, 256, 64
    vcfg = O.VitConfig(S, P, W, L_, H, F, D, ln_pre=False, pool="avg")
    sd = O.synthetic_vit_state_dict(vcfg, seed=21)
    sd = {k: v for k, v in sd.items() if not k.startswith("visual.ln_pre.")}      # a no_ln_pre checkpoint has no such tensors
    arch = A.VitArch(S, P, W, L_, H, F, D, pool="avg", ln_pre=False)
    assert arch.tokens == 17
    tower = T.VitTower(arch, sd, "cuda")
    u8 = O.synthetic_images_u8(5, S, seed=22)
    px = O.preprocess_u8_exact_size(u8)
    ref = O.vit_forward(sd, vcfg, px)
    assert _cos_err(tower.encode_u8(u8.to("cuda")), ref) < COS_TIGHT
    assert _cos_err(tower.encode_f32(px.to("cuda")), ref) < COS_TIGHT
    assert _cos_err(tower.encode_u8(u8[:1].to("cuda")), ref[:1]) < COS_TIGHT and _cos_err(tower.encode_u8(u8[:1].to("cuda")), ref[:1]) < COS_TIGHT
    cls_cfg = O.VitConfig(S, P, W, L_, H, F, D, ln_pre=False, pool="cls")
    assert _cos_err(tower.encode_u8(u8.to("cuda")), O.vit_forward(sd, cls_cfg, px)) > 10 * COS_TIGHT     # not the class-token pooling
    with pytest.raises(ValueError):
        T.VitTower(A.VitArch(S, P, W, L_, H, F, D, pool="avg", ln_pre=True), sd, "cuda")
    tcfg = O.ClipTextConfig(vocab=300, ctx=16, width=W, layers=L_, heads=H, mlp_dim=F, out_dim=D, causal=False)
    tsd = O.synthetic_clip_text_state_dict(tcfg, seed=23)
    txt = T.ClipTextTower(A.ClipTextArch(vocab=300, ctx=16, width=W, layers=L_, heads=H, mlp_dim=F, out_dim=D, causal=False), tsd, "cuda")
    g = torch.Generator().manual_seed(24)
    ids = torch.zeros(6, 16, dtype=torch.int64)
    for i, n in enumerate((3, 16, 7, 1, 12, 9)):
        ids[i, :n] = torch.randint(1, 300, (n,), generator=g)
    tref = O.clip_text_forward(tsd, tcfg, ids)
    assert _cos_err(txt.encode_ids(ids), tref) < COS_TIGHT and _cos_err(txt.encode_ids(ids[2:3]), tref[2:3]) < COS_TIGHT


def test_hf_text_tower_of_multilingual_clip():
    """open_clip CustomTextCLIP with an HF text tower (open_clip/xlm-roberta-base-ViT-B-32, xlm-roberta-large-ViT-H-14): the XLM-RoBERTa
    encoder of the transformers golden (`text.transformer.*`), open_clip's mean pooler over the non-pad tokens and the projection MLP
    inside mq_encode_bert, against the oracle (whose encoder is pinned to transformers.XLMRobertaModel by the same fixture)"""
    T, A = _towers()
    sd0, z = G.load("xlmr_small")
    V, P, W, L_, H, F = [int(v) for v in z["cfg"]]
    D = 64
    bert = A.BertArch(vocab=V, max_pos=P - 2, width=W, layers=L_, heads=H, mlp_dim=F, ln_eps=1e-5, pos_offset=2, type_vocab=1)
    arch = A.HfClipTextArch(bert=bert, out_dim=D, ctx=64)
    g = torch.Generator().manual_seed(12)
    sd = {"text.transformer." + k: v for k, v in sd0.items()}
    sd["text.proj.0.weight"] = torch.randn(arch.proj_hidden, W, generator=g) / W ** 0.5
    sd["text.proj.2.weight"] = torch.randn(D, arch.proj_hidden, generator=g) / arch.proj_hidden ** 0.5
    assert arch.proj_hidden == (W + D) // 2 == 96
    ids = torch.from_numpy(z["ids"]).clone()
    mask = torch.from_numpy(z["mask"])
    assert int((ids[mask == 1] == 1).sum()) == 0          # (no real token equals the pad id)
    ids[mask == 0] = 1                                     # open_clip's HFTokenizer pads with <pad> = 1
    cfg = O.BertConfig(vocab=V, max_pos=P, width=W, layers=L_, heads=H, mlp_dim=F, ln_eps=1e-5, pooling="mean", pos_offset=2)
    tower = T.HfClipTextTower(arch, sd, "cuda")
    for normalize in (True, False):
        ref = O.hf_clip_text_forward(sd, cfg, ids, pad_id=1, normalize=normalize)
        assert _cos_err(tower.encode_padded(ids, normalize=normalize), ref) < COS_TIGHT
    one = tower.encode_padded(ids[1:2])                    # single query: skinny GEMMs + graph replay, projection head included
    assert one.shape == (1, D) and _cos_err(one, O.hf_clip_text_forward(sd, cfg, ids[1:2])) < COS_TIGHT
    assert _cos_err(tower.encode_padded(ids[1:2]), O.hf_clip_text_forward(sd, cfg, ids[1:2])) < COS_TIGHT


def test_golden_mpnet_small():
    """hf/all-mpnet-base-* family (transformers.MPNetModel): the BERT tower with MPNet's checkpoint naming, no token types, positions from
    2 and the relative-position bias added to the attention scores inside the kernel (attention_kernel<..., BIAS>): batched rows of up
    to 200 tokens (8-wave form), a short batch (4-wave form) and the single-query route, against the transformers golden"""
    T, A = _towers()
    sd, z = G.load("mpnet_small")
    V, P, W, L_, H, F = [int(v) for v in z["cfg"]]
    ids, mask = torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"])
    arch = A.BertArch(vocab=V, max_pos=P, width=W, layers=L_, heads=H, mlp_dim=F, ln_eps=1e-5, pos_offset=2, type_vocab=0, rel_buckets=32)
    tower = T.BertTower(arch, sd, "cuda", pooling="mean")
    assert _cos_err(tower.encode_ids(ids, mask, normalize=False), torch.from_numpy(z["mean"])) < COS_TIGHT
    assert _cos_err(tower.encode_ids(ids, mask, normalize=True), torch.from_numpy(z["mean_norm"])) < COS_TIGHT
    short = [0, 1, 2, 3, 5]                                     # rows of <= 33 tokens
    out = tower.encode_ids(ids[short, :33], mask[short, :33])
    assert _cos_err(out, torch.from_numpy(z["mean_norm"])[short]) < COS_TIGHT
    for r in (1, 4):                                            # one query per call (skinny GEMMs, graph replay): 17 and 200 tokens
        n = int(mask[r].sum())
        one = tower.encode_ids(ids[r:r + 1, :n], mask[r:r + 1, :n])
        assert _cos_err(one, torch.from_numpy(z["mean_norm"])[r:r + 1]) < COS_TIGHT
        assert _cos_err(tower.encode_ids(ids[r:r + 1, :n], mask[r:r + 1, :n]), torch.from_numpy(z["mean_norm"])[r:r + 1]) < COS_TIGHT
    nobias = dict(sd)
    nobias["encoder.relative_attention_bias.weight"] = torch.zeros_like(sd["encoder.relative_attention_bias.weight"])
    wrong = T.BertTower(arch, nobias, "cuda", pooling="mean")
    assert _cos_err(wrong.encode_ids(ids, mask, normalize=False), torch.from_numpy(z["mean"])) > 5 * COS_TIGHT    # the bias is honoured


def test_golden_bert_32_wide_heads():
    """12-heads-of-32 checkpoints (e5-small, bge-small, MiniLM) run on the 64-wide attention kernel through zero-padded heads
    (engine/towers.py::_pad_heads); bf16 and fp8, mean and CLS pooling, against the transformers golden"""
    T, A = _towers()
    sd, z = G.load("bert_small_h32")
    V, P, W, L_, H, F = [int(v) for v in z["cfg"]]
    ids, mask = torch.from_numpy(z["ids"]), torch.from_numpy(z["mask"])
    arch = A.BertArch(vocab=V, max_pos=P, width=W, layers=L_, heads=H, mlp_dim=F)
    for pooling, key in (("mean", "mean"), ("cls", "cls")):
        tower = T.BertTower(arch, sd, "cuda", pooling=pooling)
        assert tower.cfg.enc.attn_width == H * 64 and tower.cfg.enc.width == W
        assert _cos_err(tower.encode_ids(ids, mask, normalize=False), torch.from_numpy(z[key])) < COS_TIGHT, pooling
    t8 = T.BertTower(arch, sd, "cuda", pooling="mean", precision="fp8")
    t8.calibrate_fp8(lambda: t8.encode_ids(ids, mask))
    assert _cos_err(t8.encode_ids(ids, mask, normalize=False), torch.from_numpy(z["mean"])) < 1e-2
    with pytest.raises(ValueError):
        T.BertTower(A.BertArch(vocab=V, max_pos=P, width=W, layers=L_, heads=3, mlp_dim=F), sd, "cuda")  # 128 / 3 heads


@pytest.mark.parametrize("name,layers", [("ViT-H-14", 3), ("ViT-g-14", 2), ("ViT-bigG-14", 2), ("ViT-H-14-378", 2)])
def test_vit_wide_heads_vs_oracle(name, layers):
    """ViT-H / g / bigG (model_registry.py:237-256): 16 heads of 80 / 88 / 104 run as zero-padded 96 / 96 / 112-wide heads
    (engine/towers.py::_pad_heads, attention_kernel<HD=128, HS>: 257 tokens = the full 160 KiB of LDS).  Real widths / MLP dims /
    token counts, depth cut to keep the fp32 CPU oracle in seconds; the text towers of these models are plain 64-wide."""
    from dataclasses import replace
    T, A = _towers()
    arch, text = A.resolve_open_clip(name)
    assert text.width == text.heads * 64
    arch = replace(arch, layers=layers)
    cfg = O.VitConfig(arch.image_size, arch.patch_size, arch.width, arch.layers, arch.heads, arch.mlp_dim, arch.out_dim)
    sd = O.synthetic_vit_state_dict(cfg, seed=3)
    u8 = O.synthetic_images_u8(3 if arch.image_size == 224 else 2, arch.image_size, seed=3)  # (-378: 730 tokens, K / V streamed through the LDS)
    ref = O.vit_forward(sd, cfg, O.preprocess_u8_exact_size(u8))
    tower = T.VitTower(arch, sd, "cuda")
    assert tower.cfg.enc.attn_width == arch.heads * (112 if name == "ViT-bigG-14" else 96) and tower.cfg.enc.width == arch.width
    out = tower.encode_u8(u8.cuda())
    assert _cos_err(out, ref) < COS_TIGHT
    # batching invariance: one image alone (257 rows: the small-call kernel family, LayerNorm kernel + un-folded weights) vs inside the batch (tiled
    # family, folded LayerNorm) — across families the documented bound is 1e-4; measured 1.0e-5 here
    assert _cos_err(tower.encode_u8(u8[1:2].cuda()), out[1:2]) < 1e-4
    if name == "ViT-H-14":
        t8 = T.VitTower(arch, sd, "cuda", precision="fp8")
        t8.calibrate_fp8(lambda: t8.encode_u8(u8.cuda()))
        assert _cos_err(t8.encode_u8(u8.cuda()), ref) < 1e-2


def test_golden_siglip_small():
    """SigLIP image (attention-pool head, no class token) and text (no mask, last-token pooling, biased projection) towers against
    the transformers.Siglip* golden (tests/golden/make_golden.py::make_siglip)"""
    T, A = _towers()
    sd, z = G.load("siglip_small")
    S, P, W, Lyr, H, Fd, V, ctx, D = [int(v) for v in z["cfg"]]
    varch = A.VitArch(S, P, W, Lyr, H, Fd, W, ln_eps=1e-6, pool="map")
    vt = T.VitTower(varch, sd, "cuda")
    assert vt.cfg.pool == 1 and varch.tokens == (S // P) ** 2
    px = torch.from_numpy(z["pixels"])
    ref = torch.from_numpy(z["image_emb"])
    # 1e-3 = the north-star tolerance: this fixture's weights are scaled 3x (make_golden._jitter) and a CPU emulation of the bf16
    # dataflow (bf16 GEMM operands / LN outputs / P, fp32 accumulation and residual) already sits at 4.1e-4 on its third image
    assert _cos_err(vt.encode_f32(px, normalize=False), ref) < 1e-3
    out_n = vt.encode_f32(px)
    assert torch.allclose(out_n.norm(dim=-1).cpu(), torch.ones(px.shape[0]), atol=1e-5)
    assert _cos_err(vt.encode_f32(px[1:2]), out_n[1:2]) < 1e-5  # batching invariance
    tarch = A.ClipTextArch(vocab=V, ctx=ctx, width=W, layers=Lyr, heads=H, mlp_dim=Fd, out_dim=D, ln_eps=1e-6, causal=False,
                           proj_bias=True, prefix="text.", pad_id=1)
    tt = T.ClipTextTower(tarch, sd, "cuda")
    # this fixture's 3x-scaled weights put the residual stream two orders of magnitude above the per-block updates: a bf16 stream loses
    # 7.8e-3 here, and the load-time residual-stream policy must see that on its calibration batch and keep fp32
    assert tt.residual_stream == "fp32", (tt.residual_stream, tt.residual_stream_error)
    ids = torch.from_numpy(z["ids"])
    assert _cos_err(tt.encode_ids(ids, normalize=False), torch.from_numpy(z["text_emb"])) < 1e-3
    with pytest.raises(ValueError):
        tt.encode_ids(ids[:, :ctx - 1])  # an unmasked tower must see all ctx positions
    lengths = torch.full((ids.shape[0],), ctx, dtype=torch.int64)
    dev = tt.encode_device(ids.to(torch.int32).to(tt.device), lengths, normalize=False)
    assert _cos_err(dev, torch.from_numpy(z["text_emb"])) < 1e-3


@pytest.mark.parametrize("name,layers,n", [("ViT-B-16-SigLIP", 3, 5), ("ViT-L-16-SigLIP-384", 2, 2), ("ViT-SO400M-14-SigLIP-384", 2, 2)])
def test_siglip_full_size_vs_oracle(name, layers, n):
    """registry shapes (model_registry.py:371-432): 196 tokens x 768, 576 tokens x 1024 (the 8-wave attention path) and SO400M's
    729 tokens x 1152 (72-wide heads run as 96, K / V streamed through the LDS, MLP 4304 zero-padded to 4352), depth cut to keep
    the fp32 CPU oracle in seconds; bf16 and fp8; text tower at ctx 64 over the 32 000-piece vocabulary"""
    from dataclasses import replace
    T, A = _towers()
    varch, tarch = A.resolve_open_clip(name)
    varch, tarch = replace(varch, layers=layers), replace(tarch, layers=layers)
    vcfg = O.SiglipVitConfig(varch.image_size, varch.patch_size, varch.width, layers, varch.heads, varch.mlp_dim)
    tcfg = O.SiglipTextConfig(tarch.vocab, tarch.ctx, tarch.width, layers, tarch.heads, tarch.mlp_dim, tarch.out_dim)
    sd = O.synthetic_siglip_state_dict(vcfg, tcfg, seed=4)
    mean = std = (0.5, 0.5, 0.5)
    u8 = O.synthetic_images_u8(n, varch.image_size, seed=4)
    ref = O.siglip_vit_forward(sd, vcfg, O.preprocess_u8_exact_size(u8, mean, std))
    vt = T.VitTower(varch, sd, "cuda", mean=mean, std=std)
    out = vt.encode_u8(u8.cuda())
    assert _cos_err(out, ref) < COS_TIGHT
    if name == "ViT-SO400M-14-SigLIP-384":
        assert vt.cfg.enc.mlp_dim == 4352 and vt.cfg.map_mlp_dim == 4352 and vt.cfg.enc.attn_width == 16 * 96
        ids = torch.ones(3, tarch.ctx, dtype=torch.int64)
        ids[:, :10] = torch.randint(2, tarch.vocab, (3, 10), generator=torch.Generator().manual_seed(5))
        assert _cos_err(T.ClipTextTower(tarch, sd, "cuda").encode_ids(ids), O.siglip_text_forward(sd, tcfg, ids)) < COS_TIGHT
    if name == "ViT-B-16-SigLIP":
        v8 = T.VitTower(varch, sd, "cuda", mean=mean, std=std, precision="fp8")
        v8.calibrate_fp8(lambda: v8.encode_u8(u8.cuda()))
        assert _cos_err(v8.encode_u8(u8.cuda()), ref) < 1e-2
        g = torch.Generator().manual_seed(4)
        ids = torch.ones(7, tarch.ctx, dtype=torch.int64)
        for i, ln in enumerate([3, 64, 9, 17, 40, 2, 33]):
            ids[i, :ln - 1] = torch.randint(2, tarch.vocab, (ln - 1,), generator=g)
        tref = O.siglip_text_forward(sd, tcfg, ids)
        tt = T.ClipTextTower(tarch, sd, "cuda")
        assert _cos_err(tt.encode_ids(ids), tref) < COS_TIGHT


@pytest.mark.parametrize("name,layers,n", [("coca_ViT-B-32", 12, 9), ("coca_ViT-L-14", 3, 3)])
def test_coca_towers_vs_oracle(name, layers, n):
    """CoCa (model_registry.py:344-370): the ViT trunk + attentional pooler (one learned query over ln_k(tokens), pooler width = embedding width, 8
    heads: 64-wide at B/32, 96-wide at L/14) and the text tower with the appended class embedding (position ctx - 1, pooled row), registry shapes
    (L/14 depth cut to keep the fp32 CPU oracle in seconds), against oracle.coca_*_forward (restated from open_clip 2.24.0; unpinned — no open_clip
    in this image; the class-token mask executes build_cls_mask's own tensor operations: the class token attends the text, the FIRST pad position and
    not itself).  Ragged texts incl. one that fills all 76 positions and one that leaves a single pad; single-text call (graph path); device ids."""
    from dataclasses import replace
    from marqo_amd import _lib as L
    T, A = _towers()
    varch, tarch = A.resolve_open_clip(name)
    varch, tarch = replace(varch, layers=layers), replace(tarch, layers=layers)
    vcfg = O.CocaVitConfig(varch.image_size, varch.patch_size, varch.width, layers, varch.heads, varch.mlp_dim, varch.out_dim, pool_heads=varch.pool_heads,
                           n_queries=256)
    tcfg = O.ClipTextConfig(tarch.vocab, tarch.ctx, tarch.width, layers, tarch.heads, tarch.mlp_dim, tarch.out_dim)
    sd = O.synthetic_coca_state_dict(vcfg, tcfg, seed=6)
    u8 = O.synthetic_images_u8(n, varch.image_size, seed=6)
    ref = O.coca_vit_forward(sd, vcfg, O.preprocess_u8_exact_size(u8))
    vt = T.VitTower(varch, sd, "cuda")
    out = vt.encode_u8(u8.cuda())
    assert out.shape == (n, varch.out_dim) and _cos_err(out, ref) < COS_TIGHT
    assert _cos_err(vt.encode_u8(u8[:1].cuda()), ref[:1]) < COS_TIGHT                      # one image (the small-row kernel families)
    raw = vt.encode_u8(u8.cuda(), normalize=False)
    assert _cos_err(raw, O.coca_vit_forward(sd, vcfg, O.preprocess_u8_exact_size(u8), normalize=False)) < COS_TIGHT
    # text: 76 token positions + the class embedding
    g = torch.Generator().manual_seed(6)
    S = tarch.ctx - 1
    ids = torch.zeros(7, S, dtype=torch.int64)
    for i, ln in enumerate([1, 74, 9, 17, 40, 30, 73]):      # ln random ids between SOT and EOT; 74 fills every position (the class token then sees itself:
                                                              # twin row), 73 leaves exactly one pad position (which the class token attends instead of itself)
        ids[i, 0] = tarch.vocab - 2
        ids[i, 1:1 + ln] = torch.randint(1, tarch.vocab - 2, (ln,), generator=g)
        ids[i, 1 + ln] = tarch.vocab - 1
    tref = O.coca_text_forward(sd, tcfg, ids)
    tt = T.ClipTextTower(tarch, sd, "cuda")
    assert tt.cfg.cls_pos == tarch.ctx - 1 and tt.cfg.vocab == tarch.vocab + 2 and tt.cfg.enc.mask == L.MQ_MASK_CAUSAL_CLS
    tout = tt.encode_ids(ids)
    assert tout.shape == (7, tarch.out_dim) and _cos_err(tout, tref) < COS_TIGHT
    # the mask matters: under the plain causal mask (the class token sees itself instead of the first pad position) the short texts move
    plain = tt.cfg.enc.mask
    try:
        tt.cfg.enc.mask = L.MQ_MASK_CAUSAL
        assert _cos_err(tt.encode_ids(ids)[[0, 2, 3]], tref[[0, 2, 3]]) > 10 * COS_TIGHT
    finally:
        tt.cfg.enc.mask = plain
    assert _cos_err(tt.encode_ids(ids[2:3]), tref[2:3]) < COS_TIGHT                         # single query
    lengths = ids.argmax(dim=1) + 1
    assert _cos_err(tt.encode_device(ids.to(torch.int32).cuda(), lengths), tref) < COS_TIGHT
    with pytest.raises(ValueError):
        tt.encode_ids(torch.zeros(1, tarch.ctx, dtype=torch.int64))                         # 77 token positions leave no room for the class embedding


@pytest.mark.parametrize("name,layers,n", [("EVA02-B-16", 12, 5), ("EVA02-L-14", 2, 3), ("EVA02-L-14-336", 1, 2), ("tiny", 2, 7)])
def test_eva02_tower_vs_oracle(name, layers, n, monkeypatch):
    """EVA02-CLIP vision towers (model_registry.py:441-460; timm Eva behind open_clip's TimmModel): class token + learned positions + 2-D rotary
    positions on the patch tokens' q / k (the 336 px tower: a 24 x 24 grid rescaled to the 16 x 16 pre-training grid), separate q / k / v with
    a bias-free k, LayerNorm between attention and out-projection, SwiGLU with a LayerNorm behind the gate (hidden 2 730 at L/14 -> zero-padded
    to 2 752, statistics over 2 730; 170 -> 192 in the tiny form), norm(class token) -> head with bias; registry shapes (L/14 depth cut to keep
    the fp32 CPU oracle in seconds) against oracle.eva_vit_forward (restated from timm; unpinned — no timm in this image).  One image (the
    small-row kernel families), un-normalised output, and the fused-qkv checkpoint form."""
    from dataclasses import replace
    from marqo_amd import _lib as L
    T, A = _towers()
    if name == "tiny":
        varch = A.VitArch(64, 16, 128, layers, 2, 170, 64, ln_eps=1e-6, ln_pre=False, eva=True)
    else:
        varch = replace(A.resolve_open_clip(name)[0], layers=layers)
    cfg = O.EvaVitConfig(varch.image_size, varch.patch_size, varch.width, layers, varch.heads, varch.mlp_dim, varch.out_dim, ref_grid=varch.rope_ref_grid)
    sd = O.synthetic_eva_state_dict(cfg, seed=8)
    u8 = O.synthetic_images_u8(n, varch.image_size, seed=8)
    ref = O.eva_vit_forward(sd, cfg, O.preprocess_u8_exact_size(u8))
    vt = T.VitTower(varch, sd, "cuda")
    assert vt.cfg.enc.mlp_glu == 2 and vt.cfg.enc.mlp_ln_dim == varch.mlp_dim and vt.cfg.enc.mlp_dim % 64 == 0 and vt.residual_stream in ("bf16", "fp32")
    out = vt.encode_u8(u8.cuda())
    assert out.shape == (n, varch.out_dim) and _cos_err(out, ref) < COS_TIGHT
    assert _cos_err(vt.encode_u8(u8[:1].cuda()), ref[:1]) < COS_TIGHT
    raw = vt.encode_u8(u8.cuda(), normalize=False)
    assert _cos_err(raw, O.eva_vit_forward(sd, cfg, O.preprocess_u8_exact_size(u8), normalize=False)) < COS_TIGHT
    assert float((raw.cpu().norm(dim=-1) / O.eva_vit_forward(sd, cfg, O.preprocess_u8_exact_size(u8), normalize=False).norm(dim=-1) - 1).abs().max()) < 5e-3
    if name in ("tiny", "EVA02-L-14"):
        # the same weights as a fused-qkv checkpoint (timm qkv_fused=True: attn.qkv.weight + q_bias / v_bias): the same tower, the same bits
        fused = dict(sd)
        for i in range(layers):
            p = f"visual.trunk.blocks.{i}.attn."
            fused[p + "qkv.weight"] = torch.cat([fused.pop(p + "q_proj.weight"), fused.pop(p + "k_proj.weight"), fused.pop(p + "v_proj.weight")], dim=0)
            fused[p + "q_bias"], fused[p + "v_bias"] = fused.pop(p + "q_proj.bias"), fused.pop(p + "v_proj.bias")
        assert torch.equal(T.VitTower(varch, fused, "cuda").encode_u8(u8.cuda()), out)
    with pytest.raises(ValueError):
        T.VitTower(varch, sd, "cuda", precision="fp8")
    # both residual-stream forms, forced: fp32 (LayerNorm kernels) and bf16 (norm1 / norm2 folded into the QKV / (up | gate) GEMMs, bf16 epilogues)
    big = O.synthetic_images_u8(max(n, 24), varch.image_size, seed=9)        # enough rows for the tiled / folded GEMM family
    bref = O.eva_vit_forward(sd, cfg, O.preprocess_u8_exact_size(big[:n]))
    for mode in ("fp32", "bf16"):
        monkeypatch.setenv("MARQO_AMD_RESIDUAL_STREAM", mode)
        tw = T.VitTower(varch, sd, "cuda")
        assert tw.residual_stream == mode
        got = tw.encode_u8(big.cuda())
        assert _cos_err(got[:n], bref) < COS_TIGHT, mode
        assert _cos_err(tw.encode_u8(big[:n].cuda()), got[:n]) < COS_TIGHT          # (another batch size = other GEMM kernel families: last bits only)
        if mode == "bf16":
            # round 6 (ABI 12): attn.norm / mlp.norm folded into the out-projection / fc2 GEMMs (default) against the LayerNorm passes (mq_tune("subln_fold", 0))
            assert tw._blocks[0].out_wf and tw._blocks[0].fc2_wf and tw._blocks[0].fc2_sf
            assert torch.equal(tw.encode_u8(big.cuda()), got)                         # deterministic
            try:
                L.check(L.load().mq_tune(b"subln_fold", 0))
                passes = tw.encode_u8(big.cuda())
            finally:
                L.check(L.load().mq_tune(b"subln_fold", 1))
            assert _cos_err(passes[:n], bref) < COS_TIGHT and _cos_err(passes, got) < 1e-4


def test_single_request_graph_replay_is_bit_identical(monkeypatch):
    """one query text / one image per call replays a hipGraph captured per (tower, token count): same kernels, same bits as the
    eager launches; new contents and new lengths go through, batches are untouched"""
    T, A = _towers()
    varch, tarch = A.resolve_open_clip("ViT-B-32")
    from dataclasses import replace
    varch, tarch = replace(varch, layers=2), replace(tarch, layers=2)
    vcfg = O.VitConfig(varch.image_size, varch.patch_size, varch.width, 2, varch.heads, varch.mlp_dim, varch.out_dim)
    tcfg = O.ClipTextConfig(vocab=tarch.vocab, ctx=77, width=tarch.width, layers=2, heads=tarch.heads, mlp_dim=tarch.mlp_dim, out_dim=tarch.out_dim)
    sd = O.synthetic_vit_state_dict(vcfg, seed=7)
    sd.update(O.synthetic_clip_text_state_dict(tcfg, seed=7))
    vt, tt = T.VitTower(varch, sd, "cuda"), T.ClipTextTower(tarch, sd, "cuda")
    bcfg = O.BertConfig(vocab=3000, max_pos=64, width=128, layers=2, heads=2, mlp_dim=256)
    bt = T.BertTower(A.BertArch(vocab=3000, max_pos=64, width=128, layers=2, heads=2, mlp_dim=256), O.synthetic_bert_state_dict(bcfg, seed=7), "cuda")
    u8 = O.synthetic_images_u8(3, 224, seed=7)
    ids = O.synthetic_clip_ids(4, seed=7)
    bids, bmask = O.synthetic_bert_batch(3, vocab=3000, seed=7)

    def run_all():
        outs = [vt.encode_u8(u8[i:i + 1]).cpu() for i in range(3)] + [vt.encode_f32(O.preprocess_u8_exact_size(u8[:1])).cpu()]
        outs += [tt.encode_ids(ids[i:i + 1]).cpu() for i in range(4)] + [tt.encode_ids(ids[1:2], normalize=False).cpu()]
        outs += [tt.encode_device(ids[2:3].to(torch.int32).to(tt.device), ids[2:3].argmax(1) + 1).cpu()]
        for i in range(3):
            n = int(bmask[i].sum())
            outs.append(bt.encode_ids(bids[i:i + 1, :n], bmask[i:i + 1, :n]).cpu())
        return outs
    monkeypatch.setattr(T, "GRAPHS", False)
    eager = run_all()
    assert not vt._graphs and not tt._graphs and not bt._graphs
    monkeypatch.setattr(T, "GRAPHS", True)
    first = run_all()    # captures
    again = run_all()    # replays
    assert vt._graphs and tt._graphs and bt._graphs and not (vt._graphs_off or tt._graphs_off or bt._graphs_off)
    for e, a, b in zip(eager, first, again):
        assert torch.equal(e, a) and torch.equal(e, b)
    # the batch path is the eager one and agrees with the single calls
    assert _cos_err(vt.encode_u8(u8.cuda()), torch.cat(eager[:3])) < 1e-5


# ---- trained-like statistics at full depth (every kernel had only ever seen N(0, s) weights) --------------------------------
def test_vit_l14_full_depth_realistic_weights_bf16():
    """ViT-L/14 at its real depth with LayerNorm gains spread over 0.05..4, outlier output channels, peaky attention and a class-token
    massive activation (~120 in two residual channels from block 2 on, against an rms of ~2): bf16 path vs the fp32 CPU oracle"""
    from marqo_amd.engine import archs, towers
    varch, _ = archs.resolve_open_clip("ViT-L-14")
    cfg = O.VitConfig(224, 14, 1024, 24, 16, 4096, 768)
    sd = O.synthetic_vit_state_dict_realistic(cfg, 0)
    u8 = O.synthetic_images_u8(3, 224, seed=9)
    ref = O.vit_forward(sd, cfg, O.preprocess_u8_exact_size(u8))
    out = towers.VitTower(varch, sd, "cuda:0").encode_u8(u8.to("cuda:0")).cpu()
    e = float((1 - torch.nn.functional.cosine_similarity(out.double(), ref.double(), dim=-1)).max())
    print(f"ViT-L/14 24L realistic weights: bf16 1-cos vs fp32 oracle {e:.2e}")
    assert e < 3e-4


def test_clip_text_full_depth_realistic_weights_bf16():
    from marqo_amd.engine import archs, towers
    _, tarch = archs.resolve_open_clip("ViT-L-14")
    cfg = O.ClipTextConfig(49408, 77, 768, 12, 12, 3072, 768)
    sd = O.synthetic_clip_text_state_dict_realistic(cfg, 0)
    ids = O.synthetic_clip_ids(12, seed=6)
    ref = O.clip_text_forward(sd, cfg, ids)
    out = towers.ClipTextTower(tarch, sd, "cuda:0").encode_ids(ids).cpu()
    e = float((1 - torch.nn.functional.cosine_similarity(out.double(), ref.double(), dim=-1)).max())
    print(f"CLIP text L/14 12L realistic weights (SOT attention-sink massive activation): bf16 1-cos vs fp32 oracle {e:.2e}")
    assert e < 3e-4


def test_bf16_residual_stream_parity(tiled_gemm_only, monkeypatch):
    """The bf16 residual stream (pre-LN towers keep x in bf16 between blocks: half the bytes of every residual epilogue and LayerNorm),
    forced on (MARQO_AMD_RESIDUAL_STREAM=bf16).  Full registry depth, plain and trained-like weights: still inside the 3e-4 the fp32-stream
    form is held to, and the pooled-rows-only last block stays bit-identical to the all-rows execution in this form too (within the tiled
    GEMM family: the fixture keeps the few pooled rows off the skinny kernel, whose split-K summation order differs)."""
    from marqo_amd import _lib as L
    from marqo_amd.engine import archs, towers
    lib = L.load()
    monkeypatch.setenv("MARQO_AMD_RESIDUAL_STREAM", "bf16")
    try:
        for arch_name, cfg in (("ViT-B-32", O.VitConfig(224, 32, 768, 12, 12, 3072, 512)), ("ViT-L-14", O.VitConfig(224, 14, 1024, 24, 16, 4096, 768))):
            varch, tarch = archs.resolve_open_clip(arch_name)
            for weights in ("plain", "realistic"):
                sd = O.synthetic_vit_state_dict(cfg, 0) if weights == "plain" else O.synthetic_vit_state_dict_realistic(cfg, 0)
                u8 = O.synthetic_images_u8(3, 224, seed=9)
                ref = O.vit_forward(sd, cfg, O.preprocess_u8_exact_size(u8))
                tower = towers.VitTower(varch, sd, "cuda:0")
                assert tower.residual_stream == "bf16" and tower.cfg.enc.residual_stream == 1
                out = tower.encode_u8(u8.to("cuda:0"))
                e = float((1 - torch.nn.functional.cosine_similarity(out.cpu().double(), ref.double(), dim=-1)).max())
                print(f"bf16 residual stream, {arch_name} {weights}: 1-cos vs fp32 oracle {e:.2e}")
                assert e < 3e-4
                L.check(lib.mq_tune(b"row_select", 0))
                full = tower.encode_u8(u8.to("cuda:0"))
                L.check(lib.mq_tune(b"row_select", 1))
                assert torch.equal(full, out)
        tcfg = O.ClipTextConfig(49408, 77, 768, 12, 12, 3072, 768)
        _, tarch = archs.resolve_open_clip("ViT-L-14")
        sd = O.synthetic_clip_text_state_dict_realistic(tcfg, 0)
        ids = O.synthetic_clip_ids(12, seed=6)
        out = towers.ClipTextTower(tarch, sd, "cuda:0").encode_ids(ids).cpu()
        e = float((1 - torch.nn.functional.cosine_similarity(out.double(), O.clip_text_forward(sd, tcfg, ids).double(), dim=-1)).max())
        print(f"bf16 residual stream, CLIP text L/14 realistic: 1-cos vs fp32 oracle {e:.2e}")
        assert e < 5e-4
    finally:
        L.check(lib.mq_tune(b"row_select", 1))


def test_residual_stream_policy_is_decided_per_model_at_load(monkeypatch):
    """auto (the default): the fixed seeded calibration batch runs through both stream forms at load; bf16 is kept only within the budget of
    the fp32 stream.  Registry-shaped weights take bf16 (and stay inside the towers' 3e-4 bound vs the fp32 oracle); a tower whose residual
    stream dwarfs its per-block updates (the SigLIP-small golden text tower: 7.8e-3 with a bf16 stream) keeps fp32; the decision is a
    pure function of (weights, seed): two loads agree; the environment can force either form."""
    from marqo_amd.engine import archs, towers
    monkeypatch.delenv("MARQO_AMD_RESIDUAL_STREAM", raising=False)
    cfg = O.VitConfig(224, 32, 768, 12, 12, 3072, 512)
    varch, _ = archs.resolve_open_clip("ViT-B-32")
    sd = O.synthetic_vit_state_dict_realistic(cfg, 0)
    t1, t2 = towers.VitTower(varch, sd, "cuda:0"), towers.VitTower(varch, sd, "cuda:0")
    assert t1.residual_stream == t2.residual_stream == "bf16" and t1.residual_stream_error == t2.residual_stream_error
    assert t1.residual_stream_error <= t1.RESIDUAL_STREAM_BUDGET
    u8 = O.synthetic_images_u8(4, 224, seed=3)
    ref = O.vit_forward(sd, cfg, O.preprocess_u8_exact_size(u8))
    e = float((1 - torch.nn.functional.cosine_similarity(t1.encode_u8(u8.to("cuda:0")).cpu().double(), ref.double(), dim=-1)).max())
    assert e < 3e-4, e
    monkeypatch.setenv("MARQO_AMD_RESIDUAL_STREAM", "fp32")
    t3 = towers.VitTower(varch, sd, "cuda:0")
    assert t3.residual_stream == "fp32" and t3.cfg.enc.residual_stream == 2
    e3 = float((1 - torch.nn.functional.cosine_similarity(t3.encode_u8(u8.to("cuda:0")).cpu().double(), ref.double(), dim=-1)).max())
    assert e3 < e + 1e-6 or e3 < 5e-5          # the fp32 stream is the tighter form
    monkeypatch.setenv("MARQO_AMD_RESIDUAL_STREAM_BUDGET", "1e-9")   # an impossible budget: auto must fall back to fp32
    monkeypatch.setenv("MARQO_AMD_RESIDUAL_STREAM", "auto")
    t4 = towers.VitTower(varch, sd, "cuda:0")
    t4.tune_residual_stream(lambda: t4.encode_u8(t4.calibration_images()), budget=1e-9)
    assert t4.residual_stream == "fp32" and t4.cfg.enc.residual_stream == 2


def test_post_ln_bf16_stream_bert_family(tiled_gemm_only, monkeypatch):
    """BERT-family (post-LN) towers on the bf16 stream: the normalised bf16 rows are the residual (in-place read-modify-write epilogues,
    in-place LayerNorm, no fp32 copy of x inside the blocks).  Forced on and off on the golden BERT / MPNet fixtures and on a full-depth
    e5-base-shaped tower: both forms inside the towers' 3e-4 bound of the fp32 oracle, different computations, deterministic; the load-time
    policy decides (auto) and two loads agree; CLS pooling works on the bf16 stream (all rows run in the last block there)."""
    T, A = _towers()
    from marqo_amd.engine import synthetic
    sdb, zb = G.load("bert_small")
    Vb, Pb, Wb, Lb, Hb, Fb = [int(v) for v in zb["cfg"]]
    bids, bmask = torch.from_numpy(zb["ids"]), torch.from_numpy(zb["mask"])
    arch_small = A.BertArch(vocab=Vb, max_pos=Pb, width=Wb, layers=Lb, heads=Hb, mlp_dim=Fb)
    cases = []
    for pooling in ("mean", "cls"):
        ref = O.hf_encode({k: v.float() for k, v in sdb.items()}, O.BertConfig(vocab=Vb, max_pos=Pb, width=Wb, layers=Lb, heads=Hb, mlp_dim=Fb, pooling=pooling),
                          bids, bmask)
        cases.append((f"bert_small/{pooling}", lambda p=pooling: T.BertTower(arch_small, sdb, "cuda", pooling=p), lambda t: t.encode_ids(bids, bmask), ref))
    b = A.HF_BERT_ARCHS["intfloat/e5-base-v2"]
    bsd = synthetic.random_bert_state_dict(b, seed=0)
    ids = torch.randint(1000, b.vocab, (24, 40), generator=torch.Generator().manual_seed(5))
    mask = torch.ones(24, 40, dtype=torch.int64)
    for i in range(24):
        mask[i, 8 + i:] = 0
    ref = O.hf_encode(bsd, O.BertConfig(vocab=b.vocab, max_pos=b.max_pos, width=b.width, layers=b.layers, heads=b.heads, mlp_dim=b.mlp_dim, ln_eps=b.ln_eps),
                      ids, mask)
    cases.append(("e5-base 12L", lambda: T.BertTower(b, bsd, "cuda"), lambda t: t.encode_ids(ids, mask), ref))
    for name, make, run, ref in cases:
        outs = {}
        for mode in ("fp32", "bf16", "auto"):
            monkeypatch.setenv("MARQO_AMD_RESIDUAL_STREAM", mode)
            t = make()
            out = run(t)
            assert torch.equal(run(t), out)
            e = float((1 - torch.nn.functional.cosine_similarity(out.cpu().double(), ref.double(), dim=-1)).max())
            print(f"post-LN stream, {name}, MARQO_AMD_RESIDUAL_STREAM={mode}: stream {t.residual_stream} (bf16 vs fp32 on the calibration batch "
                  f"{t.residual_stream_error}), 1-cos vs fp32 oracle {e:.2e}")
            assert e < 3e-4, (name, mode, e)
            assert t.cfg.enc.residual_stream == (1 if t.residual_stream == "bf16" else 2)
            if mode != "auto":
                assert t.residual_stream == mode
            else:
                t2 = make()
                assert t2.residual_stream == t.residual_stream and t2.residual_stream_error == t.residual_stream_error
            outs[mode] = out
        assert not torch.equal(outs["fp32"], outs["bf16"])
        assert torch.equal(outs["auto"], outs["bf16"] if t.residual_stream == "bf16" else outs["fp32"])


def test_weight_prefetch_changes_no_bit():
    """The LayerNorm kernels of a tower whose block weights outlive the Infinity Cache (>= 128 MB: ViT-B/32 here) also touch the next GEMMs'
    weights (rowops.hip, LnExtra).  Loads whose values are discarded: embeddings are bit-identical with the knob off, at every prefetch
    granularity, in bf16 and on the fp8 policy."""
    T, A = _towers()
    from marqo_amd.engine import synthetic
    v, t = A.resolve_open_clip("ViT-B-32")
    sd = synthetic.random_open_clip_state_dict(vision=v, text=t, seed=0)
    u8 = O.synthetic_images_u8(24, v.image_size, seed=9).cuda()          # 1200 rows: above the prefetch's row threshold
    for precision in ("bf16", "fp8"):
        tower = T.VitTower(v, sd, "cuda", precision=precision)
        if precision == "fp8":
            tower.tune_fp8_default()
        outs = []
        try:
            for knob in (0, 1, 2, 3):
                _tune("ln_prefetch", knob)
                outs.append(tower.encode_u8(u8))
        finally:
            _tune("ln_prefetch", 1)
        assert all(torch.equal(outs[0], o) for o in outs[1:]), precision
