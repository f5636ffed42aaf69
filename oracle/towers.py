"""ORACLE — test infrastructure only.  Never imported by the product (marqo_amd/*).

CPU fp32 restatement of the arithmetic behind Marqo's ``s2_inference.vectorise()`` hot path.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker / reported baseline.

Where the algorithm lives
-------------------------
The reference (marqo-ai/marqo v2.13.0) authors no tensor arithmetic; its wrappers call into
un-vendored wheels pinned in ``requirements.dev.txt``: ``open_clip_torch==2.24.0`` (ViT image
tower and CLIP text tower), ``transformers==4.41.2`` (``BertModel`` behind ``AutoModel``),
``torch==1.12.1``.  The functions below restate the published forward passes of those modules and
follow the reference's own call sites for everything around them:

* ``vit_forward``        <- ``OPEN_CLIP.encode_image`` -> ``model.encode_image``
                            (src/marqo/core/inference/embedding_models/open_clip_model.py:249-266;
                            fp32 path = the ``device == 'cpu'`` branch :259-260)
* ``clip_text_forward``  <- ``OPEN_CLIP.encode_text`` -> ``model.encode_text`` (:268-286)
* ``bert_forward`` / ``hf_encode`` <- ``HuggingFaceModel.encode`` + ``_average_pool_func`` /
                            ``_cls_pool_func`` + ``F.normalize``
                            (src/marqo/core/inference/embedding_models/hugging_face_model.py:172-214)
* ``l2_normalize_clip``  <- ``outputs /= self.normalize(outputs)`` (open_clip_model.py:262-265,
                            abstract_clip_model.py:83-85)
* ``siglip_vit_forward`` / ``siglip_text_forward`` <- the same two call sites for the SigLIP registry entries
                            (model_registry.py:371-432, 489-494: ``ViT-*-SigLIP*/webli``, ``Marqo/marqo-fashionSigLIP``):
                            open_clip builds them as ``TimmModel`` (timm ``vit_*_siglip_*``: no class token, no ln_pre,
                            attention-pool head) + ``TextTransformer(no_causal_mask, pool_type='last', proj_bias)``

Pinning status
--------------
* BERT path: PINNED against ``transformers.BertModel`` (the class the reference instantiates through
  ``AutoModel``) run in this container with shared weights; golden vectors committed under
  ``tests/golden/`` by ``tests/golden/make_golden.py`` (transformers 5.15 here vs 4.41.2 pinned by
  the reference: same BertModel arithmetic).
* CLIP ViT / text towers: pinned against ``transformers.CLIPVisionModelWithProjection`` /
  ``CLIPTextModelWithProjection`` (an independent implementation of the same published
  architecture; ``open_clip`` itself is not installable here).  The reference holds NO golden vector
  for ``open_clip/ViT-B-32/laion2b_s34b_b79k`` or ``ViT-L-14`` and real checkpoints are unavailable
  offline -> for real open_clip weights parity remains **unpinned** (SURVEY.md §8c).

* SigLIP towers: pinned against ``transformers.SiglipVisionModel`` / ``SiglipTextModel`` (``hidden_act='gelu'``: timm's
  SigLIP ViTs use nn.GELU) with the weights renamed to the open_clip / timm checkpoint naming; ``timm`` itself is not installed.

State-dict conventions: open_clip names for CLIP (``visual.*``, ``transformer.resblocks.*``,
``token_embedding.weight`` ...), HuggingFace names for BERT (``embeddings.*``, ``encoder.layer.*``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, replace, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)  # clip_utils.py:32
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)  # clip_utils.py:33


# --------------------------------------------------------------------------------------------
# configs
# --------------------------------------------------------------------------------------------
@dataclass
class VitConfig:
    image_size: int = 224
    patch_size: int = 32
    width: int = 768
    layers: int = 12
    heads: int = 12
    mlp_dim: int = 3072
    out_dim: int = 512
    quick_gelu: bool = False
    ln_eps: float = 1e-5
    ln_pre: bool = True       # False: open_clip `no_ln_pre` (CLIPA)
    pool: str = "cls"         # "avg": pool_type 'avg' with final_ln_after_pool (CLIPA): mean of the patch tokens -> ln_post -> proj

    @property
    def tokens(self) -> int:
        return (self.image_size // self.patch_size) ** 2 + 1


@dataclass
class ClipTextConfig:
    vocab: int = 49408
    ctx: int = 77
    width: int = 512
    layers: int = 12
    heads: int = 8
    mlp_dim: int = 2048
    out_dim: int = 512
    quick_gelu: bool = False
    ln_eps: float = 1e-5
    causal: bool = True   # False: `no_causal_mask` with pool_type 'last' (CLIPA text tower: every position attends to every other, the
    #                       pooled row is the LAST of the ctx positions; un-prefixed keys and an un-biased projection, unlike SigLIP)


@dataclass
class BertConfig:
    vocab: int = 30522
    max_pos: int = 512
    width: int = 768
    layers: int = 12
    heads: int = 12
    mlp_dim: int = 3072
    ln_eps: float = 1e-12
    pooling: str = "mean"  # "mean" | "cls"
    pos_offset: int = 0    # XLM-RoBERTa / RoBERTa: position ids start at padding_idx + 1 = 2 (0 for BERT)


# --------------------------------------------------------------------------------------------
# shared pieces
# --------------------------------------------------------------------------------------------
@dataclass
class SiglipVitConfig:
    """timm ``vit_{base,large}_patch16_siglip_*`` behind open_clip's TimmModel (pool 'map', proj 'none': out dim = width)"""
    image_size: int = 224
    patch_size: int = 16
    width: int = 768
    layers: int = 12
    heads: int = 12
    mlp_dim: int = 3072
    ln_eps: float = 1e-6

    @property
    def tokens(self) -> int:
        return (self.image_size // self.patch_size) ** 2


@dataclass
class SiglipTextConfig:
    vocab: int = 32000
    ctx: int = 64
    width: int = 768
    layers: int = 12
    heads: int = 12
    mlp_dim: int = 3072
    out_dim: int = 768
    ln_eps: float = 1e-6


def _act(x: Tensor, quick: bool) -> Tensor:
    if quick:
        return x * torch.sigmoid(1.702 * x)  # open_clip QuickGELU
    return F.gelu(x)  # nn.GELU (erf)


def _mha(x: Tensor, in_w: Tensor, in_b: Tensor, out_w: Tensor, out_b: Tensor, heads: int,
         attn_mask: Optional[Tensor]) -> Tensor:
    """nn.MultiheadAttention(batch_first) forward, self-attention, no dropout.  x: [B, T, W];
    attn_mask: additive, broadcastable to [B, heads, T, T]."""
    B, T, W = x.shape
    hd = W // heads
    qkv = F.linear(x, in_w, in_b)
    q, k, v = qkv.split(W, dim=-1)
    q = q.view(B, T, heads, hd).transpose(1, 2)
    k = k.view(B, T, heads, hd).transpose(1, 2)
    v = v.view(B, T, heads, hd).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    if attn_mask is not None:
        s = s + attn_mask
    p = torch.softmax(s, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, T, W)
    return F.linear(o, out_w, out_b)


def _clip_resblocks(x: Tensor, sd: Dict[str, Tensor], prefix: str, layers: int, heads: int, quick: bool,
                    eps: float, attn_mask: Optional[Tensor]) -> Tensor:
    """open_clip ResidualAttentionBlock stack (pre-LN):
    x = x + attn(ln_1(x)); x = x + mlp(ln_2(x))."""
    W = x.shape[-1]
    for i in range(layers):
        p = f"{prefix}resblocks.{i}."
        h = F.layer_norm(x, (W,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], eps)
        x = x + _mha(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"],
                     sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"], heads, attn_mask)
        h = F.layer_norm(x, (W,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], eps)
        h = _act(F.linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"]), quick)
        x = x + F.linear(h, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
    return x


def l2_normalize_clip(x: Tensor) -> Tensor:
    """open_clip_model.py:262-265: outputs /= outputs.norm(dim=-1, keepdim=True)"""
    return x / x.norm(dim=-1, keepdim=True)


# --------------------------------------------------------------------------------------------
# towers
# --------------------------------------------------------------------------------------------
@torch.no_grad()
def vit_forward(sd: Dict[str, Tensor], cfg: VitConfig, pixels: Tensor, normalize: bool = True) -> Tensor:
    """open_clip VisionTransformer.forward + encode_image's projection.
    pixels: fp32 [B, 3, S, S], already preprocessed (clip_utils.py:48-67)."""
    W = cfg.width
    x = F.conv2d(pixels, sd["visual.conv1.weight"], None, stride=cfg.patch_size)  # [B, W, G, G], no bias
    B = x.shape[0]
    x = x.reshape(B, W, -1).permute(0, 2, 1)  # [B, np, W]
    cls = sd["visual.class_embedding"].to(x.dtype).expand(B, 1, W)
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"]
    if cfg.ln_pre:
        x = F.layer_norm(x, (W,), sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"], cfg.ln_eps)
    x = _clip_resblocks(x, sd, "visual.transformer.", cfg.layers, cfg.heads, cfg.quick_gelu, cfg.ln_eps, None)
    # open_clip transformer.py VisionTransformer.forward (open_clip_torch 2.24.0, un-vendored): `final_ln_after_pool` pools first
    # (_global_pool: 'avg' = x[:, 1:].mean(dim=1), the patch tokens) and applies ln_post to the pooled row; else ln_post(class token)
    pooled = x[:, 1:].mean(dim=1) if cfg.pool == "avg" else x[:, 0]
    pooled = F.layer_norm(pooled, (W,), sd["visual.ln_post.weight"], sd["visual.ln_post.bias"], cfg.ln_eps)
    out = pooled @ sd["visual.proj"]
    return l2_normalize_clip(out) if normalize else out


@torch.no_grad()
def clip_text_forward(sd: Dict[str, Tensor], cfg: ClipTextConfig, ids: Tensor, normalize: bool = True) -> Tensor:
    """open_clip CLIP.encode_text: token embedding + positional, causal pre-LN transformer, ln_final,
    row at argmax(id) (EOT), @ text_projection.  ids: int64 [B, ctx] zero-padded."""
    B, T = ids.shape
    W = cfg.width
    x = sd["token_embedding.weight"][ids] + sd["positional_embedding"][:T]
    mask = torch.full((T, T), float("-inf")).triu(1) if cfg.causal else None
    x = _clip_resblocks(x, sd, "transformer.", cfg.layers, cfg.heads, cfg.quick_gelu, cfg.ln_eps, mask)
    x = F.layer_norm(x, (W,), sd["ln_final.weight"], sd["ln_final.bias"], cfg.ln_eps)
    pooled = x[torch.arange(B), ids.argmax(dim=-1)] if cfg.causal else x[:, -1]   # text_global_pool: 'argmax' / 'last
    out = pooled @ sd["text_projection"]
    return l2_normalize_clip(out) if normalize else out


# --------------------------------------------------------------------------------------------
# CoCa (open_clip coca_model.py + transformer.py, open_clip_torch 2.24.0 — un-vendored; RESTATED, UNPINNED: neither open_clip nor timm is in
# this image, so these two functions follow the published source from memory of its structure and are pinned to nothing but themselves.
# The reference only names the models (model_registry.py:344-370) and calls model.encode_image / encode_text on them
# (core/inference/embedding_models/open_clip_model.py:249-286).)
# --------------------------------------------------------------------------------------------
@dataclass
class CocaVitConfig(VitConfig):
    pool_heads: int = 8       # attn_pooler_heads
    n_queries: int = 256      # attn_pooler_queries (only query 0 reaches the contrastive embedding)


def coca_pooler_attention(sd: Dict[str, Tensor], a: str, q: Tensor, kx: Tensor, heads: int) -> Tensor:
    """The attention of open_clip's AttentionalPooler = torch.nn.MultiheadAttention(embed_dim = D, heads, kdim = vdim = W, batch_first = True) with its
    separate projection weights (`q_proj_weight` [D, D], `k_proj_weight` / `v_proj_weight` [D, W], one `in_proj_bias` [3 D]): q [Q, D] (shared by the batch),
    kx [B, T, W] -> [B, Q, D].  PINNED to the installed torch.nn.MultiheadAttention (tests/test_oracle_pins.py)."""
    B, D = kx.shape[0], q.shape[-1]
    hd = D // heads
    bq, bk, bv = sd[a + "in_proj_bias"].split(D)
    qh = F.linear(q, sd[a + "q_proj_weight"], bq).view(1, -1, heads, hd).transpose(1, 2)                   # [1, H, Q, hd]
    kh = F.linear(kx, sd[a + "k_proj_weight"], bk).view(B, -1, heads, hd).transpose(1, 2)                  # [B, H, T, hd]
    vh = F.linear(kx, sd[a + "v_proj_weight"], bv).view(B, -1, heads, hd).transpose(1, 2)
    p = torch.softmax((qh @ kh.transpose(-1, -2)) / math.sqrt(hd), dim=-1)                                 # [B, H, Q, T]
    o = (p @ vh).transpose(1, 2).reshape(B, -1, D)
    return F.linear(o, sd[a + "out_proj.weight"], sd[a + "out_proj.bias"])


@torch.no_grad()
def coca_vit_forward(sd: Dict[str, Tensor], cfg: CocaVitConfig, pixels: Tensor, normalize: bool = True) -> Tensor:
    """VisionTransformer(attentional_pool=True, output_dim = embed_dim) as CoCa builds it: conv1 -> class token + positions -> ln_pre -> blocks;
    x = attn_pool(x) with AttentionalPooler(d_model = output_dim, context_dim = width): MultiheadAttention(embed_dim = d_model, kdim = vdim =
    context_dim) of q = ln_q(query) [n_queries, d_model] over k = v = ln_k(x); x = ln_post(x) [B, n_queries, d_model]; pooled = x[:, 0]
    ('tok' pooling of the pooler's outputs), tokens = x[:, 1:] (the captioning decoder's input, not computed here); pooled @ proj.
    CoCa.encode_image L2-normalises by default."""
    W, D = cfg.width, cfg.out_dim
    x = F.conv2d(pixels, sd["visual.conv1.weight"], None, stride=cfg.patch_size)
    B = x.shape[0]
    x = x.reshape(B, W, -1).permute(0, 2, 1)
    x = torch.cat([sd["visual.class_embedding"].to(x.dtype).expand(B, 1, W), x], dim=1) + sd["visual.positional_embedding"]
    x = F.layer_norm(x, (W,), sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"], cfg.ln_eps)
    x = _clip_resblocks(x, sd, "visual.transformer.", cfg.layers, cfg.heads, cfg.quick_gelu, cfg.ln_eps, None)
    a = "visual.attn_pool."
    kx = F.layer_norm(x, (W,), sd[a + "ln_k.weight"], sd[a + "ln_k.bias"], cfg.ln_eps)                     # [B, T, W]
    q = F.layer_norm(sd[a + "query"], (D,), sd[a + "ln_q.weight"], sd[a + "ln_q.bias"], cfg.ln_eps)        # [n_queries, D]
    o = coca_pooler_attention(sd, a + "attn.", q, kx, cfg.pool_heads)
    o = F.layer_norm(o, (D,), sd["visual.ln_post.weight"], sd["visual.ln_post.bias"], cfg.ln_eps)
    out = o[:, 0] @ sd["visual.proj"]
    return l2_normalize_clip(out) if normalize else out


@torch.no_grad()
def coca_text_forward(sd: Dict[str, Tensor], cfg: ClipTextConfig, ids: Tensor, pad_id: int = 0, normalize: bool = True) -> Tensor:
    """TextTransformer(embed_cls=True) as CoCa builds it (keys under `text.`): ids int64 [B, ctx - 1] (SOT ... EOT, zero-padded); the learned class
    embedding is appended BEHIND the padding -> ctx positions; attention mask = causal + build_cls_mask; blocks; pooled = the LAST position (the
    class token), ln_final applied to the pooled row, @ text_projection.

    The class-token mask is written with the SAME tensor operations as open_clip 2.24.0 (transformer.py, TextTransformer.build_cls_mask / forward):
        cls_mask = (text != self.pad_id).unsqueeze(1)
        cls_mask = F.pad(cls_mask, (1, 0, cls_mask.shape[2], 0), value=True)
        additive_mask = zeros.masked_fill(~cls_mask, -inf)          # [B, S + 1, S + 1], repeated per head
        attn_mask = self.attn_mask[None, :seq_len, :seq_len] + cls_mask[:, :seq_len, :seq_len]
    The key axis is padded on the LEFT: the class row (the only row that is not all-True) allows key 0 and key j + 1 wherever text[j] != pad, i.e. the text,
    the FIRST pad position, and the class token itself only when the text fills all S positions.  (Round 5 restated this mask as "text keys + itself";
    that was wrong against this code — ADVICE r5 — and the pretrained coca_* checkpoints were trained with the shifted form.)  open_clip is not in this
    image: the four lines above are quoted from memory of the published source, the arithmetic below executes them literally; unpinned."""
    B, T = ids.shape
    W = cfg.width
    if T + 1 != cfg.ctx:
        raise ValueError(f"CoCa text towers take {cfg.ctx - 1} token positions + the class embedding")
    x = torch.cat([sd["text.token_embedding.weight"][ids], sd["text.cls_emb"].expand(B, 1, W)], dim=1) + sd["text.positional_embedding"][:T + 1]
    seq_len = T + 1
    causal = torch.full((cfg.ctx, cfg.ctx), float("-inf")).triu(1)                                         # TextTransformer.build_causal_mask
    cls_mask = (ids != pad_id).unsqueeze(1)                                                                # [B, 1, S]
    cls_mask = F.pad(cls_mask, (1, 0, cls_mask.shape[2], 0), value=True)                                   # [B, S + 1, S + 1]
    additive = torch.zeros(cls_mask.shape).masked_fill(~cls_mask, float("-inf"))
    mask = (causal[None, :seq_len, :seq_len] + additive[:, :seq_len, :seq_len])[:, None]                   # [B, 1, S + 1, S + 1] (broadcast over heads)
    x = _clip_resblocks(x, sd, "text.transformer.", cfg.layers, cfg.heads, cfg.quick_gelu, cfg.ln_eps, mask)
    pooled = F.layer_norm(x[:, -1], (W,), sd["text.ln_final.weight"], sd["text.ln_final.bias"], cfg.ln_eps)
    out = pooled @ sd["text.text_projection"]
    return l2_normalize_clip(out) if normalize else out


def synthetic_coca_state_dict(vcfg: CocaVitConfig, tcfg: ClipTextConfig, seed: int = 0) -> Dict[str, Tensor]:
    """seeded CoCa checkpoint in open_clip's naming (visual.* incl. attn_pool, text.* incl. cls_emb; the captioning decoder is left out)"""
    g = _g(seed + 4000)
    sd = synthetic_vit_state_dict(replace(vcfg, out_dim=vcfg.out_dim), seed)
    W, D = vcfg.width, vcfg.out_dim
    a = "visual.attn_pool."
    sd[a + "query"] = torch.randn(vcfg.n_queries, D, generator=g)
    for n_, dim in (("ln_q", D), ("ln_k", W)):
        sd[a + n_ + ".weight"] = 1 + 0.1 * torch.randn(dim, generator=g)
        sd[a + n_ + ".bias"] = 0.05 * torch.randn(dim, generator=g)
    sd[a + "attn.q_proj_weight"] = torch.randn(D, D, generator=g) / math.sqrt(D)
    sd[a + "attn.k_proj_weight"] = torch.randn(D, W, generator=g) / math.sqrt(W)
    sd[a + "attn.v_proj_weight"] = torch.randn(D, W, generator=g) / math.sqrt(W)
    sd[a + "attn.in_proj_bias"] = 0.02 * torch.randn(3 * D, generator=g)
    sd[a + "attn.out_proj.weight"] = torch.randn(D, D, generator=g) / math.sqrt(D)
    sd[a + "attn.out_proj.bias"] = 0.02 * torch.randn(D, generator=g)
    sd["visual.ln_post.weight"] = 1 + 0.1 * torch.randn(D, generator=g)      # over the pooler's width
    sd["visual.ln_post.bias"] = 0.05 * torch.randn(D, generator=g)
    sd["visual.proj"] = torch.randn(D, D, generator=g) / math.sqrt(D)
    Wt = tcfg.width
    std = 1.0 / math.sqrt(Wt)
    sd["text.token_embedding.weight"] = 0.5 * torch.randn(tcfg.vocab, Wt, generator=g)
    sd["text.positional_embedding"] = 0.3 * torch.randn(tcfg.ctx, Wt, generator=g)
    sd["text.cls_emb"] = 0.5 * torch.randn(Wt, generator=g)
    _blocks_clip(sd, "text.transformer.", tcfg.layers, Wt, tcfg.mlp_dim, g, 0.6 * std)
    sd["text.ln_final.weight"] = 1 + 0.1 * torch.randn(Wt, generator=g)
    sd["text.ln_final.bias"] = 0.05 * torch.randn(Wt, generator=g)
    sd["text.text_projection"] = std * torch.randn(Wt, tcfg.out_dim, generator=g)
    return sd


# --------------------------------------------------------------------------------------------
# EVA02-CLIP vision tower (timm models/eva.py + layers/pos_embed_sincos.py + layers/mlp.py, behind open_clip 2.24.0's TimmModel — un-vendored;
# RESTATED, UNPINNED: timm is not in this image and transformers holds no model of this block form, so the functions below follow the published
# source from memory of its structure and are pinned to nothing but themselves.  The reference only names the models
# (model_registry.py:441-460: EVA02-L-14-336 / EVA02-B-16 / EVA02-L-14) and calls model.encode_image on them
# (core/inference/embedding_models/open_clip_model.py:249-266).  The text towers of these entries are the plain CLIP form (clip_text_forward with
# the `text.` prefix stripped).)
# --------------------------------------------------------------------------------------------
@dataclass
class EvaVitConfig(VitConfig):
    ln_eps: float = 1e-6      # timm LayerNorm default
    ref_grid: int = 16        # ref_feat_shape = (16, 16): the pre-training grid the rotary positions are rescaled to
    rope_theta: float = 10000.0


def eva_rope(cfg: EvaVitConfig) -> Tuple[Tensor, Tensor]:
    """timm RotaryEmbeddingCat(dim = head_dim, in_pixels = False, feat_shape = grid, ref_feat_shape) -> (sin, cos), each [patches, head_dim]:
    build_rotary_pos_embed -> build_fourier_pos_embed(num_bands = head_dim // 4, bands = freq_bands = 1 / theta ** (arange(0, nb) / nb),
    t = arange(grid) / grid * ref_grid per axis, meshgrid 'ij', pos = grid[..., None] * bands) -> sin / cos reshaped [patches, 2 * nb] (y bands then
    x bands) and repeat_interleave(2) along the last axis."""
    G, hd = cfg.image_size // cfg.patch_size, cfg.width // cfg.heads
    nb = hd // 4
    bands = 1.0 / (cfg.rope_theta ** (torch.arange(0, nb, 1, dtype=torch.int64).to(torch.float32) / nb))
    t = [torch.arange(G, dtype=torch.float32) / G * cfg.ref_grid for _ in range(2)]
    grid = torch.stack(torch.meshgrid(t, indexing="ij"), dim=-1).unsqueeze(-1)
    pos = grid * bands
    sin = pos.sin().reshape(G * G, -1).repeat_interleave(2, -1)
    cos = pos.cos().reshape(G * G, -1).repeat_interleave(2, -1)
    return sin, cos


def _eva_rot(x: Tensor) -> Tensor:
    """timm `rot`: (x0, x1, x2, x3, ...) -> (-x1, x0, -x3, x2, ...)"""
    return torch.stack([-x[..., 1::2], x[..., ::2]], -1).reshape(x.shape)


def eva_apply_rope(x: Tensor, sin: Tensor, cos: Tensor) -> Tensor:
    """timm apply_rot_embed_cat on [..., tokens, head_dim]: x * cos + rot(x) * sin with the interleaved-pair rotation — the same function as GPT-J's rotary
    (rotate_every_two).  PINNED to transformers.models.gptj.modeling_gptj.apply_rotary_pos_emb (tests/test_oracle_pins.py)."""
    return x * cos + _eva_rot(x) * sin


def eva_swiglu_gate(h: Tensor, wg: Tensor, bg: Tensor, wx: Tensor, bx: Tensor) -> Tensor:
    """timm SwiGLU up to its inner norm: silu(fc1_g(h)) * fc1_x(h) — the gate / up pair of a Llama MLP.  PINNED to transformers' LlamaMLP
    (gate_proj = fc1_g, up_proj = fc1_x, identity down_proj; tests/test_oracle_pins.py)."""
    return F.silu(F.linear(h, wg, bg)) * F.linear(h, wx, bx)


@torch.no_grad()
def eva_vit_forward(sd: Dict[str, Tensor], cfg: EvaVitConfig, pixels: Tensor, normalize: bool = True) -> Tensor:
    """timm Eva.forward as open_clip's TimmModel(pool='token', proj=None) runs it (keys under `visual.trunk.`): patch_embed (conv WITH bias) ->
    class token -> + pos_embed -> blocks (EvaBlock without layer scale: x = x + attn(norm1(x), rope); x = x + mlp(norm2(x))) -> norm -> class token
    -> head (Linear with bias = the projection to the embedding width).
    EvaAttention (qkv_fused = False): q = q_proj(x), k = k_proj(x) (no bias), v = v_proj(x); the PATCH tokens' q and k are rotated,
    q[:, :, 1:] = q * cos + rot(q) * sin; softmax(q k^T / sqrt(hd)) v; x = attn.norm(x) (scale_attn_inner); proj.
    SwiGLU (scale_mlp): fc2(norm(silu(fc1_g(x)) * fc1_x(x)))."""
    t = "visual.trunk."
    W, H = cfg.width, cfg.heads
    hd = W // H
    x = F.conv2d(pixels, sd[t + "patch_embed.proj.weight"], sd[t + "patch_embed.proj.bias"], stride=cfg.patch_size)
    B = x.shape[0]
    x = x.reshape(B, W, -1).permute(0, 2, 1)
    x = torch.cat([sd[t + "cls_token"].expand(B, 1, W), x], dim=1) + sd[t + "pos_embed"]
    T = x.shape[1]
    sin, cos = eva_rope(cfg)
    for i in range(cfg.layers):
        p = f"{t}blocks.{i}."
        ln = lambda v, name, dim: F.layer_norm(v, (dim,), sd[p + name + ".weight"], sd[p + name + ".bias"], cfg.ln_eps)
        h = ln(x, "norm1", W)
        q = F.linear(h, sd[p + "attn.q_proj.weight"], sd[p + "attn.q_proj.bias"]).view(B, T, H, hd).transpose(1, 2)
        k = F.linear(h, sd[p + "attn.k_proj.weight"], None).view(B, T, H, hd).transpose(1, 2)
        v = F.linear(h, sd[p + "attn.v_proj.weight"], sd[p + "attn.v_proj.bias"]).view(B, T, H, hd).transpose(1, 2)
        q = torch.cat([q[:, :, :1], eva_apply_rope(q[:, :, 1:], sin, cos)], dim=2)
        k = torch.cat([k[:, :, :1], eva_apply_rope(k[:, :, 1:], sin, cos)], dim=2)
        a = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1) @ v
        a = ln(a.transpose(1, 2).reshape(B, T, W), "attn.norm", W)
        x = x + F.linear(a, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        h = ln(x, "norm2", W)
        m = eva_swiglu_gate(h, sd[p + "mlp.fc1_g.weight"], sd[p + "mlp.fc1_g.bias"], sd[p + "mlp.fc1_x.weight"], sd[p + "mlp.fc1_x.bias"])
        m = ln(m, "mlp.norm", cfg.mlp_dim)
        x = x + F.linear(m, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    x = F.layer_norm(x, (W,), sd[t + "norm.weight"], sd[t + "norm.bias"], cfg.ln_eps)
    out = F.linear(x[:, 0], sd[t + "head.weight"], sd[t + "head.bias"])
    return l2_normalize_clip(out) if normalize else out


def synthetic_eva_state_dict(cfg: EvaVitConfig, seed: int = 0) -> Dict[str, Tensor]:
    """seeded EVA02-CLIP vision checkpoint in timm's naming under open_clip's `visual.trunk.`"""
    g = _g(seed + 5000)
    W, P, Fh, t = cfg.width, cfg.patch_size, cfg.mlp_dim, "visual.trunk."
    std = 0.6 / math.sqrt(W)
    sd: Dict[str, Tensor] = {}

    def lin(name, out_f, in_f, scale, bias=True):
        sd[name + ".weight"] = scale * torch.randn(out_f, in_f, generator=g)
        if bias:
            sd[name + ".bias"] = 0.02 * torch.randn(out_f, generator=g)

    def norm(name, dim):
        sd[name + ".weight"] = 1 + 0.1 * torch.randn(dim, generator=g)
        sd[name + ".bias"] = 0.05 * torch.randn(dim, generator=g)
    sd[t + "patch_embed.proj.weight"] = torch.randn(W, 3, P, P, generator=g) / math.sqrt(3 * P * P)
    sd[t + "patch_embed.proj.bias"] = 0.05 * torch.randn(W, generator=g)
    sd[t + "cls_token"] = 0.5 * torch.randn(1, 1, W, generator=g)
    sd[t + "pos_embed"] = 0.3 * torch.randn(1, cfg.tokens, W, generator=g)
    for i in range(cfg.layers):
        p = f"{t}blocks.{i}."
        norm(p + "norm1", W)
        lin(p + "attn.q_proj", W, W, std)
        lin(p + "attn.k_proj", W, W, std, bias=False)
        lin(p + "attn.v_proj", W, W, std)
        norm(p + "attn.norm", W)
        lin(p + "attn.proj", W, W, std)
        norm(p + "norm2", W)
        lin(p + "mlp.fc1_g", Fh, W, 2 * std)
        lin(p + "mlp.fc1_x", Fh, W, 2 * std)
        norm(p + "mlp.norm", Fh)
        lin(p + "mlp.fc2", W, Fh, std / 2)
    norm(t + "norm", W)
    lin(t + "head", cfg.out_dim, W, 1.0 / math.sqrt(W))
    return sd


@torch.no_grad()
def siglip_vit_forward(sd: Dict[str, Tensor], cfg: SiglipVitConfig, pixels: Tensor, normalize: bool = True) -> Tensor:
    """timm VisionTransformer (class_token=False, global_pool='map', pre-LN blocks, nn.GELU, LayerNorm eps 1e-6) as open_clip's
    ``visual.trunk``: patch_embed (conv WITH bias) + pos_embed -> blocks -> norm -> AttentionPoolLatent -> (head = Identity).
    AttentionPoolLatent: q = Linear(latent) [1 query], k, v = Linear(x).chunk(2), softmax(q k^T / sqrt(hd)) v -> proj;
    x = x + mlp(norm(x)); pooled = x[:, 0].   pixels: fp32 [B, 3, S, S] preprocessed."""
    W, H = cfg.width, cfg.heads
    t = "visual.trunk."
    x = F.conv2d(pixels, sd[t + "patch_embed.proj.weight"], sd[t + "patch_embed.proj.bias"], stride=cfg.patch_size)
    B = x.shape[0]
    x = x.reshape(B, W, -1).permute(0, 2, 1) + sd[t + "pos_embed"]  # [B, N, W]
    for i in range(cfg.layers):
        p = f"{t}blocks.{i}."
        h = F.layer_norm(x, (W,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], cfg.ln_eps)
        x = x + _mha(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"], sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"], H, None)
        h = F.layer_norm(x, (W,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], cfg.ln_eps)
        h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        x = x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    x = F.layer_norm(x, (W,), sd[t + "norm.weight"], sd[t + "norm.bias"], cfg.ln_eps)
    a = t + "attn_pool."
    N, hd = x.shape[1], W // H
    q = F.linear(sd[a + "latent"].expand(B, -1, -1), sd[a + "q.weight"], sd[a + "q.bias"]).view(B, 1, H, hd).transpose(1, 2)
    kv = F.linear(x, sd[a + "kv.weight"], sd[a + "kv.bias"]).view(B, N, 2, H, hd).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    o = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1) @ v           # [B, H, 1, hd]
    y = F.linear(o.transpose(1, 2).reshape(B, 1, W), sd[a + "proj.weight"], sd[a + "proj.bias"])
    h = F.layer_norm(y, (W,), sd[a + "norm.weight"], sd[a + "norm.bias"], cfg.ln_eps)
    y = y + F.linear(F.gelu(F.linear(h, sd[a + "mlp.fc1.weight"], sd[a + "mlp.fc1.bias"])), sd[a + "mlp.fc2.weight"], sd[a + "mlp.fc2.bias"])
    out = y[:, 0]
    return l2_normalize_clip(out) if normalize else out


@torch.no_grad()
def siglip_text_forward(sd: Dict[str, Tensor], cfg: SiglipTextConfig, ids: Tensor, normalize: bool = True) -> Tensor:
    """open_clip TextTransformer(no_causal_mask=True, pool_type='last', proj_bias=True) under CustomTextCLIP (``text.*`` keys):
    every one of the ctx positions (padding included — there is no padding mask) attends to every other; the pooled row is the
    LAST position; text_projection is a Linear with bias.  ids: int64 [B, ctx] padded with the pad id (1)."""
    B, T = ids.shape
    W = cfg.width
    x = sd["text.token_embedding.weight"][ids] + sd["text.positional_embedding"][:T]
    x = _clip_resblocks(x, sd, "text.transformer.", cfg.layers, cfg.heads, False, cfg.ln_eps, None)
    x = F.layer_norm(x, (W,), sd["text.ln_final.weight"], sd["text.ln_final.bias"], cfg.ln_eps)
    out = F.linear(x[:, -1], sd["text.text_projection.weight"], sd["text.text_projection.bias"])
    return l2_normalize_clip(out) if normalize else out


@torch.no_grad()
def bert_forward(sd: Dict[str, Tensor], cfg: BertConfig, ids: Tensor, attention_mask: Tensor) -> Tensor:
    """transformers BertModel forward (absolute positions, token_type_ids = 0, eval mode)
    -> last_hidden_state [B, S, W].  With cfg.pos_offset = 2 it is transformers XLMRobertaModel / RobertaModel: the same
    encoder, position ids = cumsum(mask) * mask + padding_idx (= offset + index for the real tokens of a right-padded row; the
    padded rows are masked out of the attention and of the pooling, so their position does not matter)."""
    B, S = ids.shape
    W, H = cfg.width, cfg.heads
    x = (sd["embeddings.word_embeddings.weight"][ids]
         + sd["embeddings.token_type_embeddings.weight"][torch.zeros_like(ids)]
         + sd["embeddings.position_embeddings.weight"][cfg.pos_offset:cfg.pos_offset + S])
    x = F.layer_norm(x, (W,), sd["embeddings.LayerNorm.weight"], sd["embeddings.LayerNorm.bias"], cfg.ln_eps)
    # additive key-padding mask: (1 - mask) * finfo.min
    add_mask = (1.0 - attention_mask[:, None, None, :].to(x.dtype)) * torch.finfo(x.dtype).min
    hd = W // H
    for i in range(cfg.layers):
        p = f"encoder.layer.{i}."
        q = F.linear(x, sd[p + "attention.self.query.weight"], sd[p + "attention.self.query.bias"])
        k = F.linear(x, sd[p + "attention.self.key.weight"], sd[p + "attention.self.key.bias"])
        v = F.linear(x, sd[p + "attention.self.value.weight"], sd[p + "attention.self.value.bias"])
        q = q.view(B, S, H, hd).transpose(1, 2)
        k = k.view(B, S, H, hd).transpose(1, 2)
        v = v.view(B, S, H, hd).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) / math.sqrt(hd) + add_mask
        a = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, S, W)
        a = F.linear(a, sd[p + "attention.output.dense.weight"], sd[p + "attention.output.dense.bias"])
        x = F.layer_norm(a + x, (W,), sd[p + "attention.output.LayerNorm.weight"],
                         sd[p + "attention.output.LayerNorm.bias"], cfg.ln_eps)
        h = F.gelu(F.linear(x, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
        h = F.linear(h, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
        x = F.layer_norm(h + x, (W,), sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], cfg.ln_eps)
    return x


# ---- MPNet (sentence-transformers all-mpnet-base-*: the reference loads it through AutoModel, hugging_face_model.py:125-130) ------------------
MPNET_REL_BUCKETS, MPNET_REL_MAX_DISTANCE = 32, 128


def mpnet_relative_position_bucket(relative_position: Tensor, num_buckets: int = MPNET_REL_BUCKETS,
                                   max_distance: int = MPNET_REL_MAX_DISTANCE) -> Tensor:
    """transformers MPNetEncoder.relative_position_bucket (third-party, un-vendored: transformers==4.41.2 in the reference's
    requirements; modeling_mpnet.py), restated: T5's bidirectional log-spaced buckets of (key position - query position)."""
    ret = 0
    n = -relative_position
    num_buckets //= 2
    ret += (n < 0).to(torch.long) * num_buckets
    n = torch.abs(n)
    max_exact = num_buckets // 2
    is_small = n < max_exact
    val_if_large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact)
                                * (num_buckets - max_exact)).to(torch.long)
    val_if_large = torch.min(val_if_large, torch.full_like(val_if_large, num_buckets - 1))
    return ret + torch.where(is_small, n, val_if_large)


def mpnet_forward(sd: Dict[str, Tensor], cfg: BertConfig, ids: Tensor, attention_mask: Tensor) -> Tensor:
    """transformers MPNetModel forward (eval mode) -> last_hidden_state [B, S, W]: a post-LN BERT-style encoder WITHOUT token types,
    position ids = padding_idx + 1 + index (cfg.pos_offset = 2, as RoBERTa), and ONE relative-position bias table
    (`encoder.relative_attention_bias.weight` [32 buckets, heads]) shared by every layer and added to the scaled q.k scores before the
    key-padding mask and the softmax (MPNetSelfAttention.forward: scores / sqrt(d) + position_bias + mask)."""
    B, S = ids.shape
    W, H = cfg.width, cfg.heads
    x = sd["embeddings.word_embeddings.weight"][ids] + sd["embeddings.position_embeddings.weight"][cfg.pos_offset:cfg.pos_offset + S]
    x = F.layer_norm(x, (W,), sd["embeddings.LayerNorm.weight"], sd["embeddings.LayerNorm.bias"], cfg.ln_eps)
    add_mask = (1.0 - attention_mask[:, None, None, :].to(x.dtype)) * torch.finfo(x.dtype).min
    pos = torch.arange(S, dtype=torch.long)
    bucket = mpnet_relative_position_bucket(pos[None, :] - pos[:, None])                 # memory - context = key - query
    bias = sd["encoder.relative_attention_bias.weight"][bucket].permute(2, 0, 1)[None]    # [1, H, S(query), S(key)]
    hd = W // H
    for i in range(cfg.layers):
        p = f"encoder.layer.{i}."
        q, k, v = (F.linear(x, sd[p + f"attention.attn.{n}.weight"], sd[p + f"attention.attn.{n}.bias"]).view(B, S, H, hd).transpose(1, 2)
                   for n in ("q", "k", "v"))
        s = (q @ k.transpose(-1, -2)) / math.sqrt(hd) + bias + add_mask
        a = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, S, W)
        a = F.linear(a, sd[p + "attention.attn.o.weight"], sd[p + "attention.attn.o.bias"])
        x = F.layer_norm(a + x, (W,), sd[p + "attention.LayerNorm.weight"], sd[p + "attention.LayerNorm.bias"], cfg.ln_eps)
        h = F.gelu(F.linear(x, sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"]))
        h = F.linear(h, sd[p + "output.dense.weight"], sd[p + "output.dense.bias"])
        x = F.layer_norm(h + x, (W,), sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], cfg.ln_eps)
    return x


def synthetic_mpnet_state_dict(cfg: BertConfig, seed: int = 0) -> Dict[str, Tensor]:
    """seeded weights in the MPNetModel checkpoint naming (biases / LayerNorm perturbed off their init so a dropped term cannot pass)"""
    g = torch.Generator().manual_seed(seed)
    W, F_ = cfg.width, cfg.mlp_dim
    rn = lambda *shape, std=0.02: torch.randn(*shape, generator=g) * std
    sd = {"embeddings.word_embeddings.weight": rn(cfg.vocab, W, std=0.05),
          "embeddings.position_embeddings.weight": rn(cfg.max_pos + cfg.pos_offset, W, std=0.05),
          "embeddings.LayerNorm.weight": 1 + rn(W, std=0.1), "embeddings.LayerNorm.bias": rn(W, std=0.1),
          "encoder.relative_attention_bias.weight": rn(MPNET_REL_BUCKETS, cfg.heads, std=0.5)}
    for i in range(cfg.layers):
        p = f"encoder.layer.{i}."
        for n in ("q", "k", "v", "o"):
            sd[p + f"attention.attn.{n}.weight"], sd[p + f"attention.attn.{n}.bias"] = rn(W, W, std=0.06), rn(W, std=0.1)
        sd[p + "attention.LayerNorm.weight"], sd[p + "attention.LayerNorm.bias"] = 1 + rn(W, std=0.1), rn(W, std=0.1)
        sd[p + "intermediate.dense.weight"], sd[p + "intermediate.dense.bias"] = rn(F_, W, std=0.06), rn(F_, std=0.1)
        sd[p + "output.dense.weight"], sd[p + "output.dense.bias"] = rn(W, F_, std=0.06), rn(W, std=0.1)
        sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"] = 1 + rn(W, std=0.1), rn(W, std=0.1)
    return sd


def hf_clip_text_forward(sd: Dict[str, Tensor], cfg: BertConfig, ids: Tensor, pad_id: int = 1, normalize: bool = True) -> Tensor:
    """Text tower of open_clip's CustomTextCLIP with an HF encoder (open_clip/xlm-roberta-base-ViT-B-32, xlm-roberta-large-ViT-H-14; the
    reference calls it through `self.model.encode_text(text)` + its own normalisation, open_clip_model.py:268-286).  open_clip_torch
    2.24.0 hf_model.py (third-party, un-vendored) HFTextEncoder.forward, restated: attention mask = ids != pad; the HF encoder
    (`text.transformer.*`, XLM-RoBERTa = bert_forward with positions from 2); MeanPooler = sum of the masked last_hidden_state /
    number of real tokens; `text.proj` = Linear(W, (W + D) // 2, bias=False) -> GELU -> Linear(., D, bias=False)."""
    t = "text.transformer."
    enc = {k[len(t):]: v for k, v in sd.items() if k.startswith(t)}
    mask = (ids != pad_id).to(torch.int64)
    last = bert_forward(enc, cfg, ids, mask)
    pooled = (last * mask[..., None].to(last.dtype)).sum(dim=1) / mask.sum(dim=-1, keepdim=True)
    out = F.linear(F.gelu(F.linear(pooled, sd["text.proj.0.weight"])), sd["text.proj.2.weight"])
    return l2_normalize_clip(out) if normalize else out


def mclip_text_forward(sd: Dict[str, Tensor], cfg: BertConfig, ids: Tensor, attention_mask: Tensor, normalize: bool = True) -> Tensor:
    """Text encoder of the reference's multilingual_clip loader (clip_utils.py:553-565 -> `self.textual_model.forward(sentence, tokenizer)`
    + its own normalisation).  multilingual_clip's pt_multilingual_clip.MultilingualCLIP.forward (third-party, un-vendored), restated:
    embs = transformer(**tokens)[0]; embs = (embs * att[..., None]).sum(1) / att.sum(1)[:, None]; LinearTransformation(embs)."""
    t = "transformer."
    enc = {k[len(t):]: v for k, v in sd.items() if k.startswith(t)}
    last = bert_forward(enc, cfg, ids, attention_mask)
    att = attention_mask.to(last.dtype)
    pooled = (last * att[..., None]).sum(dim=1) / att.sum(dim=1)[:, None]
    out = F.linear(pooled, sd["LinearTransformation.weight"], sd["LinearTransformation.bias"])
    return l2_normalize_clip(out) if normalize else out


@dataclass
class NewModelConfig:
    """Alibaba-NLP/new-impl `NewModel` (custom remote code: the reference's hf_stella loader runs it through AutoModel with
    trust_remote_code, hugging_face_stella_model.py:9-23; stella_en_400M_v5 and gte-*-en-v1.5 use it).  PARITY UNPINNED: the remote
    code is un-vendored and not available offline; this restates its published forward pass — rotary positions (rotate_half
    convention, NTK-scaled base), packed qkv_proj with bias, o_proj, gated-GELU MLP (up_gate_proj without bias, down_proj with
    bias), post-LN (attn_ln / mlp_ln), embeddings = word + token_type(0) -> LayerNorm."""
    vocab: int = 30528
    max_pos: int = 8192
    width: int = 1024
    layers: int = 24
    heads: int = 16
    mlp_dim: int = 4096
    ln_eps: float = 1e-12
    rope_theta: float = 160000.0
    rope_ntk_factor: Optional[float] = 2.0
    pooling: str = "mean"


def new_model_inv_freq(cfg: NewModelConfig) -> Tensor:
    d = cfg.width // cfg.heads
    base = cfg.rope_theta * (cfg.rope_ntk_factor or 1.0)
    inv = 1.0 / (base ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    return inv / (cfg.rope_ntk_factor ** (2.0 / d)) if cfg.rope_ntk_factor else inv


def rotate_half(x: Tensor) -> Tensor:
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


@torch.no_grad()
def new_model_forward(sd: Dict[str, Tensor], cfg: NewModelConfig, ids: Tensor, attention_mask: Tensor) -> Tensor:
    """-> last_hidden_state [B, S, W]"""
    B, S = ids.shape
    W, H = cfg.width, cfg.heads
    hd = W // H
    x = sd["embeddings.word_embeddings.weight"][ids]
    if "embeddings.token_type_embeddings.weight" in sd:
        x = x + sd["embeddings.token_type_embeddings.weight"][torch.zeros_like(ids)]
    x = F.layer_norm(x, (W,), sd["embeddings.LayerNorm.weight"], sd["embeddings.LayerNorm.bias"], cfg.ln_eps)
    freqs = torch.arange(S, dtype=torch.float32)[:, None] * new_model_inv_freq(cfg)[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    cos, sin = emb.cos()[None, None], emb.sin()[None, None]          # [1, 1, S, hd]
    add_mask = (1.0 - attention_mask[:, None, None, :].to(x.dtype)) * torch.finfo(x.dtype).min
    for i in range(cfg.layers):
        p = f"encoder.layer.{i}."
        qkv = F.linear(x, sd[p + "attention.qkv_proj.weight"], sd[p + "attention.qkv_proj.bias"])
        q, k, v = qkv.split(W, dim=-1)
        q = q.view(B, S, H, hd).transpose(1, 2)
        k = k.view(B, S, H, hd).transpose(1, 2)
        v = v.view(B, S, H, hd).transpose(1, 2)
        q, k = q * cos + rotate_half(q) * sin, k * cos + rotate_half(k) * sin
        s = (q @ k.transpose(-1, -2)) / math.sqrt(hd) + add_mask
        a = (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B, S, W)
        a = F.linear(a, sd[p + "attention.o_proj.weight"], sd[p + "attention.o_proj.bias"])
        x = F.layer_norm(x + a, (W,), sd[p + "attn_ln.weight"], sd[p + "attn_ln.bias"], cfg.ln_eps)
        up_gate = F.linear(x, sd[p + "mlp.up_gate_proj.weight"])
        up, gate = up_gate.split(cfg.mlp_dim, dim=-1)
        h = F.linear(F.gelu(gate) * up, sd[p + "mlp.down_proj.weight"], sd[p + "mlp.down_proj.bias"])
        x = F.layer_norm(x + h, (W,), sd[p + "mlp_ln.weight"], sd[p + "mlp_ln.bias"], cfg.ln_eps)
    return x


@torch.no_grad()
def new_model_encode(sd: Dict[str, Tensor], cfg: NewModelConfig, ids: Tensor, attention_mask: Tensor, normalize: bool = True) -> Tensor:
    """HuggingFaceModel.encode (hugging_face_model.py:187-197) over NewModel: mean (or CLS) pool of last_hidden_state + F.normalize"""
    last = new_model_forward(sd, cfg, ids, attention_mask)
    if cfg.pooling == "mean":
        last = last.masked_fill(~attention_mask[..., None].bool(), 0.0)
        emb = last.sum(dim=1) / attention_mask.sum(dim=1)[..., None]
    else:
        emb = last[:, 0]
    return F.normalize(emb, p=2, dim=1) if normalize else emb


def synthetic_new_model_state_dict(cfg: NewModelConfig, seed: int = 0) -> Dict[str, Tensor]:
    g = _g(seed + 9000)
    W, Fd = cfg.width, cfg.mlp_dim
    std = 0.6 / math.sqrt(W)
    sd: Dict[str, Tensor] = {}
    sd["embeddings.word_embeddings.weight"] = 0.5 * torch.randn(cfg.vocab, W, generator=g)
    sd["embeddings.token_type_embeddings.weight"] = 0.3 * torch.randn(2, W, generator=g)
    sd["embeddings.LayerNorm.weight"] = 1 + 0.1 * torch.randn(W, generator=g)
    sd["embeddings.LayerNorm.bias"] = 0.05 * torch.randn(W, generator=g)
    for i in range(cfg.layers):
        p = f"encoder.layer.{i}."
        sd[p + "attention.qkv_proj.weight"] = std * torch.randn(3 * W, W, generator=g)
        sd[p + "attention.qkv_proj.weight"][: 2 * W] *= 2.0      # so that the rotary phases matter to the softmax
        sd[p + "attention.qkv_proj.bias"] = 0.02 * torch.randn(3 * W, generator=g)
        sd[p + "attention.o_proj.weight"] = std * torch.randn(W, W, generator=g)
        sd[p + "attention.o_proj.bias"] = 0.02 * torch.randn(W, generator=g)
        sd[p + "attn_ln.weight"] = 1 + 0.1 * torch.randn(W, generator=g)
        sd[p + "attn_ln.bias"] = 0.05 * torch.randn(W, generator=g)
        sd[p + "mlp.up_gate_proj.weight"] = std * torch.randn(2 * Fd, W, generator=g)
        sd[p + "mlp.down_proj.weight"] = (0.6 / math.sqrt(Fd)) * torch.randn(W, Fd, generator=g)
        sd[p + "mlp.down_proj.bias"] = 0.02 * torch.randn(W, generator=g)
        sd[p + "mlp_ln.weight"] = 1 + 0.1 * torch.randn(W, generator=g)
        sd[p + "mlp_ln.bias"] = 0.05 * torch.randn(W, generator=g)
    return sd


@torch.no_grad()
def hf_encode(sd: Dict[str, Tensor], cfg: BertConfig, ids: Tensor, attention_mask: Tensor,
              normalize: bool = True) -> Tensor:
    """HuggingFaceModel.encode after tokenisation (hugging_face_model.py:187-197)."""
    mpnet = "encoder.relative_attention_bias.weight" in sd
    last = (mpnet_forward if mpnet else bert_forward)(sd, cfg, ids, attention_mask)
    if cfg.pooling == "mean":  # _average_pool_func :205-209
        last = last.masked_fill(~attention_mask[..., None].bool(), 0.0)
        emb = last.sum(dim=1) / attention_mask.sum(dim=1)[..., None]
    elif cfg.pooling == "cls":  # _cls_pool_func :211-214
        emb = last[:, 0]
    else:
        raise ValueError(cfg.pooling)
    if normalize:
        emb = F.normalize(emb, p=2, dim=1)
    return emb


# --------------------------------------------------------------------------------------------
# synthetic weights (seeded) in the checkpoint naming the loaders consume
# --------------------------------------------------------------------------------------------
def _g(seed: int) -> torch.Generator:
    return torch.Generator().manual_seed(seed)


def _blocks_clip(sd, prefix, layers, W, F_, g, std):
    for i in range(layers):
        p = f"{prefix}resblocks.{i}."
        sd[p + "ln_1.weight"] = 1 + 0.1 * torch.randn(W, generator=g)
        sd[p + "ln_1.bias"] = 0.05 * torch.randn(W, generator=g)
        sd[p + "attn.in_proj_weight"] = std * torch.randn(3 * W, W, generator=g)
        sd[p + "attn.in_proj_bias"] = 0.02 * torch.randn(3 * W, generator=g)
        sd[p + "attn.out_proj.weight"] = std * torch.randn(W, W, generator=g)
        sd[p + "attn.out_proj.bias"] = 0.02 * torch.randn(W, generator=g)
        sd[p + "ln_2.weight"] = 1 + 0.1 * torch.randn(W, generator=g)
        sd[p + "ln_2.bias"] = 0.05 * torch.randn(W, generator=g)
        sd[p + "mlp.c_fc.weight"] = std * torch.randn(F_, W, generator=g)
        sd[p + "mlp.c_fc.bias"] = 0.02 * torch.randn(F_, generator=g)
        sd[p + "mlp.c_proj.weight"] = std * torch.randn(W, F_, generator=g)
        sd[p + "mlp.c_proj.bias"] = 0.02 * torch.randn(W, generator=g)


def synthetic_vit_state_dict(cfg: VitConfig, seed: int = 0) -> Dict[str, Tensor]:
    """Seeded weights with non-trivial LN affine / biases (so a dropped bias or swapped gamma/beta
    cannot pass), scaled so activations stay O(1) through the stack."""
    g = _g(seed)
    W = cfg.width
    std = 1.0 / math.sqrt(W)
    sd: Dict[str, Tensor] = {}
    sd["visual.conv1.weight"] = torch.randn(W, 3, cfg.patch_size, cfg.patch_size, generator=g) / math.sqrt(3 * cfg.patch_size ** 2)
    sd["visual.class_embedding"] = 0.5 * torch.randn(W, generator=g)
    sd["visual.positional_embedding"] = 0.3 * torch.randn(cfg.tokens, W, generator=g)
    sd["visual.ln_pre.weight"] = 1 + 0.1 * torch.randn(W, generator=g)
    sd["visual.ln_pre.bias"] = 0.05 * torch.randn(W, generator=g)
    _blocks_clip(sd, "visual.transformer.", cfg.layers, W, cfg.mlp_dim, g, 0.6 * std)
    sd["visual.ln_post.weight"] = 1 + 0.1 * torch.randn(W, generator=g)
    sd["visual.ln_post.bias"] = 0.05 * torch.randn(W, generator=g)
    sd["visual.proj"] = std * torch.randn(W, cfg.out_dim, generator=g)
    return sd


def _realistic_blocks(sd: Dict[str, Tensor], prefix: str, layers: int, W: int, F_: int, g: torch.Generator) -> None:
    """Rewrites a resblock stack with trained-like statistics (what N(0, s) weights never show a kernel):
      * LayerNorm gamma log-normal (sigma 0.4) with a handful of channels at 0.05 and at 4.0, beta ~ N(0, 0.3);
      * per-layer weight scale varying over 0.7-1.6x, 1 % of the output channels of every linear 4x larger (outlier channels),
        biases ~ N(0, 0.1);
      * query / key rows 2x larger, so attention logits have a standard deviation of a few units (peaky softmax) instead of the
        near-uniform attention of small random weights."""
    base = 0.6 / math.sqrt(W)
    for i in range(layers):
        p = f"{prefix}resblocks.{i}."
        for ln in ("ln_1", "ln_2"):
            gam = torch.exp(0.4 * torch.randn(W, generator=g))
            idx = torch.randperm(W, generator=g)
            gam[idx[:8]] = 0.05
            gam[idx[8:16]] = 4.0
            sd[p + ln + ".weight"] = gam
            sd[p + ln + ".bias"] = 0.3 * torch.randn(W, generator=g)
        scale = 0.7 + 0.9 * float(torch.rand(1, generator=g))
        for name, (n_out, n_in) in (("attn.in_proj_", (3 * W, W)), ("attn.out_proj.", (W, W)), ("mlp.c_fc.", (F_, W)), ("mlp.c_proj.", (W, F_))):
            std = base * scale * (math.sqrt(W / n_in) if n_in != W else 1.0)
            w = std * torch.randn(n_out, n_in, generator=g)
            out_idx = torch.randperm(n_out, generator=g)[: max(1, n_out // 100)]
            w[out_idx] *= 4.0
            if name == "attn.in_proj_":
                w[: 2 * W] *= 2.0
            sd[p + name + "weight"] = w
            sd[p + name + "bias"] = 0.1 * torch.randn(n_out, generator=g)


def _massive_unit(sd: Dict[str, Tensor], prefix: str, layers: int, layer: int, c0: int, targets: Sequence[int], unit: int, value: float,
                  later_gain: float) -> None:
    """MLP hidden unit `unit` of block `layer` becomes a detector of the token whose residual channel c0 is dominant (the class
    token / SOT token, see the callers) and writes `value` x GELU(~5) into the residual channels `targets` of that token only: a
    token-specific massive activation that every later LayerNorm, GEMM and quantiser has to live with, as in trained ViTs / LMs.
    As in trained models the network downstream is NOT sensitive to the exact massive value: every later LayerNorm has a small gain
    (`later_gain`) on the outlier channels, so their main effect is on the row statistics (everything else in that row shrinks)."""
    p = f"{prefix}resblocks.{layer}."
    for i in range(layers):
        for ln in ("ln_1", "ln_2"):
            q = f"{prefix}resblocks.{i}.{ln}."
            sd[q + "weight"][c0] = later_gain
            sd[q + "bias"][c0] = 0.0
            if i > layer:
                for c in targets:
                    sd[q + "weight"][c] = later_gain
                    sd[q + "bias"][c] = 0.0
    sd[p + "ln_2.weight"][c0] = 1.0
    sd[p + "mlp.c_fc.weight"][unit] = 0.0
    sd[p + "mlp.c_fc.weight"][unit, c0] = 1.0
    sd[p + "mlp.c_fc.bias"][unit] = -8.0
    sd[p + "mlp.c_proj.weight"][:, unit] = 0.0
    for c in targets:
        sd[p + "mlp.c_proj.weight"][c, unit] = value


def synthetic_vit_state_dict_realistic(cfg: VitConfig, seed: int = 0, massive_layer: int = 2, massive_value: float = 30.0,
                                       later_gain: float = 0.05) -> Dict[str, Tensor]:
    """ViT checkpoint with trained-like statistics (see _realistic_blocks) plus a class-token-specific massive activation: the class
    embedding carries one dominant channel (20 against ~0.5), block `massive_layer`'s MLP turns it into ~5 x massive_value (= 150)
    in two residual channels of the class-token row from that block on.  Used by the full-depth parity tests (bf16 and fp8)."""
    g = _g(seed + 7000)
    sd = synthetic_vit_state_dict(cfg, seed)
    W = cfg.width
    _realistic_blocks(sd, "visual.transformer.", cfg.layers, W, cfg.mlp_dim, g)
    c0, targets = 5, (W // 3, W // 2 + 1)
    sd["visual.class_embedding"][c0] = 20.0
    sd["visual.positional_embedding"][0, c0] = 0.0
    sd["visual.ln_pre.weight"][c0] = 1.0
    sd["visual.ln_pre.bias"][c0] = 0.0
    if cfg.layers > massive_layer:
        _massive_unit(sd, "visual.transformer.", cfg.layers, massive_layer, c0, targets, unit=7, value=massive_value, later_gain=later_gain)
    sd["visual.ln_post.weight"] = torch.exp(0.4 * torch.randn(W, generator=g))
    sd["visual.ln_post.bias"] = 0.3 * torch.randn(W, generator=g)
    for c in (c0,) + tuple(targets):
        sd["visual.ln_post.weight"][c] = later_gain
    return sd


def synthetic_clip_text_state_dict_realistic(cfg: ClipTextConfig, seed: int = 0, massive_layer: int = 1, massive_value: float = 30.0,
                                             later_gain: float = 0.05) -> Dict[str, Tensor]:
    """CLIP text checkpoint with trained-like statistics and an attention-sink style massive activation on the SOT token (id
    vocab - 2): its embedding has one dominant channel, block `massive_layer` turns it into ~150 in two residual channels."""
    g = _g(seed + 8000)
    sd = synthetic_clip_text_state_dict(cfg, seed)
    W = cfg.width
    _realistic_blocks(sd, "transformer.", cfg.layers, W, cfg.mlp_dim, g)
    c0, targets = 5, (W // 3, W // 2 + 1)
    sd["token_embedding.weight"][cfg.vocab - 2] *= 0.2
    sd["token_embedding.weight"][cfg.vocab - 2, c0] = 20.0
    sd["positional_embedding"][0, c0] = 0.0
    if cfg.layers > massive_layer:
        _massive_unit(sd, "transformer.", cfg.layers, massive_layer, c0, targets, unit=7, value=massive_value, later_gain=later_gain)
    for c in (c0,) + tuple(targets):
        sd["ln_final.weight"][c] = later_gain
    return sd


def synthetic_clip_text_state_dict(cfg: ClipTextConfig, seed: int = 0) -> Dict[str, Tensor]:
    g = _g(seed + 1000)
    W = cfg.width
    std = 1.0 / math.sqrt(W)
    sd: Dict[str, Tensor] = {}
    sd["token_embedding.weight"] = 0.5 * torch.randn(cfg.vocab, W, generator=g)
    sd["positional_embedding"] = 0.3 * torch.randn(cfg.ctx, W, generator=g)
    _blocks_clip(sd, "transformer.", cfg.layers, W, cfg.mlp_dim, g, 0.6 * std)
    sd["ln_final.weight"] = 1 + 0.1 * torch.randn(W, generator=g)
    sd["ln_final.bias"] = 0.05 * torch.randn(W, generator=g)
    sd["text_projection"] = std * torch.randn(W, cfg.out_dim, generator=g)
    return sd


def synthetic_siglip_state_dict(vcfg: SiglipVitConfig, tcfg: Optional[SiglipTextConfig] = None, seed: int = 0) -> Dict[str, Tensor]:
    """random-init SigLIP towers in the open_clip / timm checkpoint naming (visual.trunk.*, text.*)"""
    g = _g(seed)
    W, F_, std = vcfg.width, vcfg.mlp_dim, 0.02
    rn = lambda *shape, sc=std: torch.randn(*shape, generator=g) * sc
    t = "visual.trunk."
    sd = {t + "patch_embed.proj.weight": rn(W, 3, vcfg.patch_size, vcfg.patch_size), t + "patch_embed.proj.bias": rn(W, sc=0.1),
          t + "pos_embed": rn(1, vcfg.tokens, W), t + "norm.weight": 1.0 + rn(W, sc=0.1), t + "norm.bias": rn(W, sc=0.1)}
    for i in range(vcfg.layers):
        p = f"{t}blocks.{i}."
        sd.update({p + "norm1.weight": 1.0 + rn(W, sc=0.1), p + "norm1.bias": rn(W, sc=0.1),
                   p + "attn.qkv.weight": rn(3 * W, W), p + "attn.qkv.bias": rn(3 * W, sc=0.1),
                   p + "attn.proj.weight": rn(W, W), p + "attn.proj.bias": rn(W, sc=0.1),
                   p + "norm2.weight": 1.0 + rn(W, sc=0.1), p + "norm2.bias": rn(W, sc=0.1),
                   p + "mlp.fc1.weight": rn(F_, W), p + "mlp.fc1.bias": rn(F_, sc=0.1),
                   p + "mlp.fc2.weight": rn(W, F_), p + "mlp.fc2.bias": rn(W, sc=0.1)})
    a = t + "attn_pool."
    sd.update({a + "latent": rn(1, 1, W, sc=W ** -0.5), a + "q.weight": rn(W, W, sc=0.05), a + "q.bias": rn(W, sc=0.1),
               a + "kv.weight": rn(2 * W, W, sc=0.05), a + "kv.bias": rn(2 * W, sc=0.1),
               a + "proj.weight": rn(W, W, sc=0.05), a + "proj.bias": rn(W, sc=0.1),
               a + "norm.weight": 1.0 + rn(W, sc=0.1), a + "norm.bias": rn(W, sc=0.1),
               a + "mlp.fc1.weight": rn(F_, W), a + "mlp.fc1.bias": rn(F_, sc=0.1),
               a + "mlp.fc2.weight": rn(W, F_), a + "mlp.fc2.bias": rn(W, sc=0.1)})
    if tcfg is not None:
        Wt = tcfg.width
        sd.update({"text.token_embedding.weight": rn(tcfg.vocab, Wt), "text.positional_embedding": rn(tcfg.ctx, Wt, sc=0.01),
                   "text.ln_final.weight": 1.0 + rn(Wt, sc=0.1), "text.ln_final.bias": rn(Wt, sc=0.1),
                   "text.text_projection.weight": rn(tcfg.out_dim, Wt, sc=Wt ** -0.5), "text.text_projection.bias": rn(tcfg.out_dim, sc=0.1)})
        _blocks_clip(sd, "text.transformer.", tcfg.layers, Wt, tcfg.mlp_dim, g, std)
    return sd


def synthetic_bert_state_dict(cfg: BertConfig, seed: int = 0) -> Dict[str, Tensor]:
    g = _g(seed + 2000)
    W, F_ = cfg.width, cfg.mlp_dim
    std = 0.6 / math.sqrt(W)
    sd: Dict[str, Tensor] = {}
    sd["embeddings.word_embeddings.weight"] = 0.5 * torch.randn(cfg.vocab, W, generator=g)
    sd["embeddings.position_embeddings.weight"] = 0.3 * torch.randn(cfg.max_pos, W, generator=g)
    sd["embeddings.token_type_embeddings.weight"] = 0.3 * torch.randn(2, W, generator=g)
    sd["embeddings.LayerNorm.weight"] = 1 + 0.1 * torch.randn(W, generator=g)
    sd["embeddings.LayerNorm.bias"] = 0.05 * torch.randn(W, generator=g)
    for i in range(cfg.layers):
        p = f"encoder.layer.{i}."
        for nm in ("query", "key", "value"):
            sd[p + f"attention.self.{nm}.weight"] = std * torch.randn(W, W, generator=g)
            sd[p + f"attention.self.{nm}.bias"] = 0.02 * torch.randn(W, generator=g)
        sd[p + "attention.output.dense.weight"] = std * torch.randn(W, W, generator=g)
        sd[p + "attention.output.dense.bias"] = 0.02 * torch.randn(W, generator=g)
        sd[p + "attention.output.LayerNorm.weight"] = 1 + 0.1 * torch.randn(W, generator=g)
        sd[p + "attention.output.LayerNorm.bias"] = 0.05 * torch.randn(W, generator=g)
        sd[p + "intermediate.dense.weight"] = std * torch.randn(F_, W, generator=g)
        sd[p + "intermediate.dense.bias"] = 0.02 * torch.randn(F_, generator=g)
        sd[p + "output.dense.weight"] = std * torch.randn(W, F_, generator=g)
        sd[p + "output.dense.bias"] = 0.02 * torch.randn(W, generator=g)
        sd[p + "output.LayerNorm.weight"] = 1 + 0.1 * torch.randn(W, generator=g)
        sd[p + "output.LayerNorm.bias"] = 0.05 * torch.randn(W, generator=g)
    return sd


# --------------------------------------------------------------------------------------------
# synthetic inputs
# --------------------------------------------------------------------------------------------
def synthetic_images_u8(n: int, size: int = 224, seed: int = 0) -> torch.Tensor:
    """uint8 HWC [n, size, size, 3] ~ U{0..255}."""
    return torch.randint(0, 256, (n, size, size, 3), generator=_g(seed + 3000), dtype=torch.uint8)


def synthetic_natural_images_u8(n: int, height: int = 224, width: int = 224, seed: int = 0) -> torch.Tensor:
    """uint8 HWC [n, height, width, 3] with natural-image statistics instead of white noise: a 1/f amplitude spectrum (random phases), strongly
    correlated colour channels, a smooth illumination gradient and a few flat 'objects' with sharp edges; mean ~0.45, contrast ~0.22.  Held-out
    inputs for the tolerance tests: the towers' load-time policies calibrate on U{0..255} pixels, real requests look like this."""
    import numpy as np
    rng = np.random.default_rng(seed + 9000)
    fy = np.fft.fftfreq(height)[:, None]
    fx = np.fft.rfftfreq(width)[None, :]
    f = np.sqrt(fy * fy + fx * fx)
    f[0, 0] = 1.0
    amp = 1.0 / f
    amp[0, 0] = 0.0
    out = np.empty((n, height, width, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:height, 0:width]
    for i in range(n):
        base = np.fft.irfft2(amp * np.exp(2j * np.pi * rng.random(amp.shape)), s=(height, width))
        base /= base.std() + 1e-12
        img = np.empty((height, width, 3))
        for c in range(3):
            own = np.fft.irfft2(amp * np.exp(2j * np.pi * rng.random(amp.shape)), s=(height, width))
            own /= own.std() + 1e-12
            img[..., c] = 0.85 * base + 0.3 * own
        gy, gx = rng.uniform(-0.6, 0.6, 2)
        img += (gy * (yy / height - 0.5) + gx * (xx / width - 0.5))[..., None]
        img = 0.45 + 0.22 * img / (img.std() + 1e-12)
        for _ in range(int(rng.integers(2, 6))):     # flat patches: edges and saturated regions
            h0, w0 = int(rng.integers(0, height - 8)), int(rng.integers(0, width - 8))
            h1, w1 = min(height, h0 + int(rng.integers(8, height // 2))), min(width, w0 + int(rng.integers(8, width // 2)))
            img[h0:h1, w0:w1] = 0.6 * img[h0:h1, w0:w1] + 0.4 * rng.random(3)
        out[i] = np.clip(img * 255.0 + 0.5, 0, 255).astype(np.uint8)
    return torch.from_numpy(out)


def preprocess_u8_exact_size(images_u8: Tensor, mean: Sequence[float] = OPENAI_DATASET_MEAN,
                             std: Sequence[float] = OPENAI_DATASET_STD) -> Tensor:
    """The tail of clip_utils.py:61-66 for images already at model resolution (Resize and CenterCrop
    are identities): ToTensor (/255, HWC->CHW) then Normalize(mean, std).  fp32 [n, 3, S, S]."""
    x = images_u8.permute(0, 3, 1, 2).to(torch.float32).div(255)
    m = torch.tensor(mean, dtype=torch.float32).view(1, 3, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, 3, 1, 1)
    return (x - m) / s


def synthetic_clip_ids(n: int, ctx: int = 77, vocab: int = 49408, seed: int = 0, full_length: bool = False
                       ) -> torch.Tensor:
    """int64 [n, ctx]: SOT (vocab-2), L_i tokens in [1, vocab-3], EOT (vocab-1, the max id), zero pad.
    full_length=True puts EOT at the last position (every sequence runs all ctx tokens)."""
    g = _g(seed + 4000)
    ids = torch.zeros(n, ctx, dtype=torch.int64)
    for i in range(n):
        L = ctx - 2 if full_length else int(torch.randint(5, ctx - 1, (1,), generator=g))
        ids[i, 0] = vocab - 2
        ids[i, 1:1 + L] = torch.randint(1, vocab - 2, (L,), generator=g)
        ids[i, 1 + L] = vocab - 1
    return ids


def synthetic_bert_batch(n: int, min_len: int = 8, max_len: int = 32, vocab: int = 30522, seed: int = 0,
                         fixed_len: Optional[int] = None):
    """ids int64 [n, S] right-padded with 0 + attention_mask: [CLS]=101 ... [SEP]=102."""
    g = _g(seed + 5000)
    lens = [fixed_len or int(torch.randint(min_len, max_len + 1, (1,), generator=g)) for _ in range(n)]
    S = max(lens)
    ids = torch.zeros(n, S, dtype=torch.int64)
    mask = torch.zeros(n, S, dtype=torch.int64)
    lo = min(1000, vocab // 2)
    for i, L in enumerate(lens):
        ids[i, 0] = 101 if vocab > 102 else 1
        ids[i, 1:L - 1] = torch.randint(lo, vocab, (L - 2,), generator=g)
        ids[i, L - 1] = 102 if vocab > 102 else 2
        mask[i, :L] = 1
    return ids, mask
