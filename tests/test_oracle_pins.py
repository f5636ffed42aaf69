"""The pieces of the EVA02 / CoCa oracle restatements (oracle/towers.py) that CAN be pinned in this image, pinned: each is compared, live, with the
installed third-party implementation of the same published operator (torch.nn / transformers) on seeded inputs.  What stays unpinned afterwards is
the ORDER in which timm's EvaBlock / open_clip's CoCa compose them (neither package is installed) — stated in DESIGN.md §5.

  CoCa attentional pooler   == torch.nn.MultiheadAttention(embed_dim, heads, kdim = vdim = width, batch_first = True)   (open_clip AttentionalPooler.attn)
  CoCa class-token mask     == what torch.nn.MultiheadAttention computes under open_clip's 3-D additive mask (causal + build_cls_mask, per head)
  CoCa packed rows          == the engine's packing ([text, one pad row, class] / [text, class, twin], MQ_MASK_CAUSAL_CLS) against the padded form
  EVA02 rotary              == transformers.models.gptj.modeling_gptj.apply_rotary_pos_emb / rotate_every_two (interleaved pairs)
  EVA02 SwiGLU gate         == transformers LlamaMLP (gate_proj / up_proj with biases, identity down_proj)
  sub-LayerNorms            == torch.nn.LayerNorm
Registry rows: /root/reference/src/marqo/s2_inference/model_registry.py:344-370 (CoCa), :441-460 (EVA02)."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import towers as O

TOL = 2e-5


def test_coca_pooler_is_torch_multihead_attention_with_kdim_vdim():
    g = torch.Generator().manual_seed(1)
    for (B, T, W, D, H, Q) in ((3, 50, 96, 64, 8, 5), (2, 257, 128, 96, 8, 3), (1, 7, 80, 64, 1, 1)):
        mha = torch.nn.MultiheadAttention(D, H, kdim=W, vdim=W, batch_first=True).eval()
        with torch.no_grad():
            for p in mha.parameters():
                p.copy_(torch.randn(p.shape, generator=g) * 0.2)
        assert not mha._qkv_same_embed_dim
        sd = {"a.q_proj_weight": mha.q_proj_weight, "a.k_proj_weight": mha.k_proj_weight, "a.v_proj_weight": mha.v_proj_weight,
              "a.in_proj_bias": mha.in_proj_bias, "a.out_proj.weight": mha.out_proj.weight, "a.out_proj.bias": mha.out_proj.bias}
        q = torch.randn(Q, D, generator=g)
        kx = torch.randn(B, T, W, generator=g)
        with torch.no_grad():
            want = mha(q[None].expand(B, -1, -1), kx, kx, need_weights=False)[0]
            got = O.coca_pooler_attention(sd, "a.", q, kx, H)
        assert got.shape == want.shape == (B, Q, D)
        assert float((got - want).abs().max()) < TOL


def _open_clip_text_mask(ids, ctx, heads, pad_id=0):
    """open_clip 2.24.0 TextTransformer.forward with embed_cls, as quoted in oracle.coca_text_forward -> the 3-D mask nn.MultiheadAttention takes"""
    seq_len = ids.shape[1] + 1
    causal = torch.full((ctx, ctx), float("-inf")).triu(1)
    cls_mask = (ids != pad_id).unsqueeze(1)
    cls_mask = F.pad(cls_mask, (1, 0, cls_mask.shape[2], 0), value=True)
    additive = torch.zeros(cls_mask.shape).masked_fill(~cls_mask, float("-inf"))
    additive = torch.repeat_interleave(additive, heads, 0)
    return causal[None, :seq_len, :seq_len] + additive[:, :seq_len, :seq_len]


def _coca_ids(vocab, S, lens, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros(len(lens), S, dtype=torch.int64)
    for i, ln in enumerate(lens):
        ids[i, 0] = vocab - 2
        ids[i, 1:1 + ln] = torch.randint(1, vocab - 2, (ln,), generator=g)
        ids[i, 1 + ln] = vocab - 1
    return ids


def test_coca_text_tower_under_the_cls_mask_is_what_torch_mha_computes():
    """one block of the oracle's text tower (its own _mha with the 4-D broadcast mask) against torch.nn.MultiheadAttention fed open_clip's 3-D
    [B * heads, L, L] mask, and the class row of that mask spelled out: key 0, key j + 1 where text[j] != pad — the first pad position, not itself"""
    vocab, ctx, W, H = 100, 12, 32, 4
    S = ctx - 1
    ids = _coca_ids(vocab, S, [1, S - 2, S - 3, 4], seed=2)     # S - 2 fills all positions, S - 3 leaves one pad
    m3 = _open_clip_text_mask(ids, ctx, H)
    assert m3.shape == (4 * H, ctx, ctx)
    for b, ln in enumerate([1, S - 2, S - 3, 4]):
        L_ = ln + 2                                             # SOT .. EOT
        row = m3[b * H, S]                                      # the class token's query row
        allowed = set(torch.nonzero(row == 0).flatten().tolist())
        want = set(range(0, min(L_, S - 1) + 1)) | ({S} if L_ == S else set())
        assert allowed == want, (b, sorted(allowed), sorted(want))
        assert bool((m3[b * H, :S] == torch.full((ctx, ctx), float("-inf")).triu(1)[:S]).all())   # every other row: plain causal
    g = torch.Generator().manual_seed(3)
    mha = torch.nn.MultiheadAttention(W, H, batch_first=True).eval()
    with torch.no_grad():
        for p in mha.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.2)
    x = torch.randn(4, ctx, W, generator=g)
    with torch.no_grad():
        want = mha(x, x, x, attn_mask=m3, need_weights=False)[0]
        m4 = m3.view(4, H, ctx, ctx)[:, :1]
        got = O._mha(x, mha.in_proj_weight, mha.in_proj_bias, mha.out_proj.weight, mha.out_proj.bias, H, m4)
    assert float((got - want).abs().max()) < TOL


def test_engine_packing_of_coca_texts_equals_the_padded_form():
    """the engine never runs padding rows: it packs  text, ONE pad row, class  (or  text, class, class twin  behind a full-length text) and masks the last
    row's own key (MQ_MASK_CAUSAL_CLS).  Evaluated here in fp32 with the oracle's blocks, row by row, against the padded reference form."""
    vocab, ctx, W, H, layers = 120, 16, 32, 4, 3
    S = ctx - 1
    tcfg = O.ClipTextConfig(vocab, ctx, W, layers, H, 64, 24)
    vcfg = O.CocaVitConfig(32, 16, 32, 1, 4, 64, 24, pool_heads=4, n_queries=4)
    sd = O.synthetic_coca_state_dict(vcfg, tcfg, seed=4)
    lens = [1, S - 2, S - 3, 5, S - 4]
    ids = _coca_ids(vocab, S, lens, seed=5)
    ref = O.coca_text_forward(sd, tcfg, ids, normalize=False)
    tok, pos, cls = sd["text.token_embedding.weight"], sd["text.positional_embedding"], sd["text.cls_emb"]
    for b, ln in enumerate(lens):
        L_ = ln + 2
        if L_ < S:
            rows = torch.cat([tok[ids[b, :L_]] + pos[:L_], (tok[0] + pos[L_])[None], (cls + pos[ctx - 1])[None]])
        else:
            rows = torch.cat([tok[ids[b, :S]] + pos[:S], (cls + pos[ctx - 1])[None], (cls + pos[ctx - 1])[None]])
        n = rows.shape[0]
        mask = torch.full((n, n), float("-inf")).triu(1)
        mask[n - 1, n - 1] = float("-inf")                      # the last row does not see its own key
        x = O._clip_resblocks(rows[None], sd, "text.transformer.", layers, H, False, tcfg.ln_eps, mask[None, None])
        pooled = F.layer_norm(x[0, -1], (W,), sd["text.ln_final.weight"], sd["text.ln_final.bias"], tcfg.ln_eps) @ sd["text.text_projection"]
        assert float((pooled - ref[b]).abs().max()) < TOL * 5, (b, ln)


def test_eva_rotary_is_gptj_rotary():
    from transformers.models.gptj.modeling_gptj import apply_rotary_pos_emb, rotate_every_two
    cfg = O.EvaVitConfig(64, 16, 128, 1, 2, 170, 64)
    sin, cos = O.eva_rope(cfg)                                  # [patches, head_dim], every band twice
    g = torch.Generator().manual_seed(6)
    B, H, T, hd = 2, 2, sin.shape[0], 64
    x = torch.randn(B, H, T, hd, generator=g)
    assert torch.equal(O._eva_rot(x), rotate_every_two(x))
    # GPT-J takes [B, T, H, hd] tensors and HALF-width tables that it repeats pairwise itself
    want = apply_rotary_pos_emb(x.transpose(1, 2), sin[None, :, 0::2], cos[None, :, 0::2]).transpose(1, 2)
    got = O.eva_apply_rope(x, sin, cos)
    assert torch.equal(sin[:, 0::2], sin[:, 1::2]) and torch.equal(got, want)


def test_eva_swiglu_gate_is_llama_mlp():
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaMLP
    W, Fh = 48, 80
    mlp = LlamaMLP(LlamaConfig(hidden_size=W, intermediate_size=Fh, mlp_bias=True, hidden_act="silu", num_attention_heads=1, num_key_value_heads=1)).eval()
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for p in mlp.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.2)
        # an identity "down projection" exposes the gated product itself: square it up with a [Fh, Fh] identity
        mlp.down_proj = torch.nn.Identity()
        h = torch.randn(5, 9, W, generator=g)
        want = mlp(h)
        got = O.eva_swiglu_gate(h, mlp.gate_proj.weight, mlp.gate_proj.bias, mlp.up_proj.weight, mlp.up_proj.bias)
    assert got.shape == (5, 9, Fh) and float((got - want).abs().max()) < 1e-6


def test_sub_layernorms_are_torch_layernorm():
    g = torch.Generator().manual_seed(8)
    for dim, eps in ((170, 1e-6), (2730, 1e-6), (768, 1e-5)):
        ln = torch.nn.LayerNorm(dim, eps=eps).eval()
        with torch.no_grad():
            ln.weight.copy_(1 + 0.1 * torch.randn(dim, generator=g))
            ln.bias.copy_(0.05 * torch.randn(dim, generator=g))
            x = torch.randn(4, 7, dim, generator=g) * 3 + 0.5
            assert torch.equal(F.layer_norm(x, (dim,), ln.weight, ln.bias, eps), ln(x))


def test_eva_block_composed_from_the_pinned_pieces():
    """the oracle's EVA forward is the pinned pieces in timm's EvaBlock order (the ORDER is the unpinned part): recompute one tower from nn modules"""
    cfg = O.EvaVitConfig(64, 16, 128, 2, 2, 170, 64)
    sd = O.synthetic_eva_state_dict(cfg, seed=9)
    px = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(10))
    ref = O.eva_vit_forward(sd, cfg, px, normalize=False)
    from transformers.models.gptj.modeling_gptj import apply_rotary_pos_emb
    t, W, H = "visual.trunk.", cfg.width, cfg.heads
    hd = W // H
    sin, cos = O.eva_rope(cfg)
    with torch.no_grad():
        x = F.conv2d(px, sd[t + "patch_embed.proj.weight"], sd[t + "patch_embed.proj.bias"], stride=cfg.patch_size).flatten(2).transpose(1, 2)
        x = torch.cat([sd[t + "cls_token"].expand(2, 1, W), x], 1) + sd[t + "pos_embed"]
        T = x.shape[1]
        for i in range(cfg.layers):
            p = f"{t}blocks.{i}."
            n1 = torch.nn.LayerNorm(W, eps=cfg.ln_eps); n1.weight.copy_(sd[p + "norm1.weight"]); n1.bias.copy_(sd[p + "norm1.bias"])
            h = n1(x)
            q = F.linear(h, sd[p + "attn.q_proj.weight"], sd[p + "attn.q_proj.bias"]).view(2, T, H, hd)
            k = F.linear(h, sd[p + "attn.k_proj.weight"]).view(2, T, H, hd)
            v = F.linear(h, sd[p + "attn.v_proj.weight"], sd[p + "attn.v_proj.bias"]).view(2, T, H, hd)
            rot = lambda u: torch.cat([u[:, :1], apply_rotary_pos_emb(u[:, 1:], sin[None, :, 0::2], cos[None, :, 0::2])], 1)
            a = F.scaled_dot_product_attention(rot(q).transpose(1, 2), rot(k).transpose(1, 2), v.transpose(1, 2)).transpose(1, 2).reshape(2, T, W)
            a = F.layer_norm(a, (W,), sd[p + "attn.norm.weight"], sd[p + "attn.norm.bias"], cfg.ln_eps)
            x = x + F.linear(a, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
            h = F.layer_norm(x, (W,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], cfg.ln_eps)
            m = F.silu(F.linear(h, sd[p + "mlp.fc1_g.weight"], sd[p + "mlp.fc1_g.bias"])) * F.linear(h, sd[p + "mlp.fc1_x.weight"], sd[p + "mlp.fc1_x.bias"])
            m = F.layer_norm(m, (cfg.mlp_dim,), sd[p + "mlp.norm.weight"], sd[p + "mlp.norm.bias"], cfg.ln_eps)
            x = x + F.linear(m, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
        x = F.layer_norm(x, (W,), sd[t + "norm.weight"], sd[t + "norm.bias"], cfg.ln_eps)
        out = F.linear(x[:, 0], sd[t + "head.weight"], sd[t + "head.bias"])
    assert float((out - ref).abs().max()) < 1e-4
