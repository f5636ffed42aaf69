"""Architecture tables for the towers the engine runs.

Shapes are the published open_clip 2.24.0 model configs / HF BERT configs (third-party, not in the
reference tree; the reference only pins names and output dims in
src/marqo/s2_inference/model_registry.py:76-610,616-880).  Only towers whose attention head dim is
64 are runnable by the gfx950 attention kernel; the others are listed so that the error is explicit.
"""
from __future__ import annotations

from dataclasses import dataclass, replace
from typing import Optional, Tuple

OPENAI_DATASET_MEAN = (0.48145466, 0.4578275, 0.40821073)  # clip_utils.py:32
OPENAI_DATASET_STD = (0.26862954, 0.26130258, 0.27577711)  # clip_utils.py:33


@dataclass(frozen=True)
class VitArch:
    image_size: int
    patch_size: int
    width: int
    layers: int
    heads: int
    mlp_dim: int
    out_dim: int
    quick_gelu: bool = False
    ln_eps: float = 1e-5
    pool: str = "cls"  # "cls": open_clip VisionTransformer (class token, ln_pre, ln_post(class token) @ proj);
    #                    "map": timm SigLIP ViT behind open_clip's TimmModel (no class token, no ln_pre, attention-pool head, no proj);
    #                    "avg": open_clip VisionTransformer with pool_type 'avg' + final_ln_after_pool (CLIPA): class token kept in the sequence,
    #                           mean of the PATCH tokens -> ln_post -> proj
    #                    "query": open_clip VisionTransformer + AttentionalPooler (CoCa): class token and ln_pre as in CLIP, every token runs every
    #                           block, one learned query attends over ln_k(tokens) in a pooler of width out_dim -> ln_post -> proj
    ln_pre: bool = True          # False: `no_ln_pre` (CLIPA)
    preprocessor: Optional[str] = None   # open_clip preprocess config of the registry entry when it is not the default ("CLIPA")
    pool_heads: int = 8          # pool "query": heads of the attentional pooler (attn_pooler_heads)
    # timm EVA02 CLIP towers behind open_clip's TimmModel (`visual.trunk.*`, model_registry.py:441-460; timm eva.py eva02_*_clip_*): class token, learned
    # absolute positions AND 2-D rotary positions on the patch tokens' q / k, conv bias, no ln_pre; blocks with separate q / k / v projections (no k
    # bias), a LayerNorm between attention and out-projection, SwiGLU MLP (hidden = mlp_dim, e.g. int(1024 * 4 * 2 / 3) = 2730) with a LayerNorm
    # behind the gate; norm(class token) -> head (a Linear WITH bias = the projection)
    eva: bool = False
    rope_ref_grid: int = 16      # eva: `ref_feat_shape` — rotary positions are grid index / grid * rope_ref_grid (the pre-training grid)
    rope_theta: float = 10000.0

    @property
    def tokens(self) -> int:
        return (self.image_size // self.patch_size) ** 2 + (0 if self.pool == "map" else 1)

    def rope_table(self):
        """eva: fp32 [patches, 2, head_dim] = (cos | sin) per patch position, timm's RotaryEmbeddingCat for in_pixels = False with `ref_feat_shape`
        (build_rotary_pos_embed / build_fourier_pos_embed / freq_bands, the torch ops in their order): head_dim // 4 bands
        1 / theta ** (k / (head_dim // 4)); positions t = arange(grid) / grid * ref_grid per axis; angles [y bands | x bands], each repeated twice
        (the interleaved pairs of apply_rot_embed_cat)."""
        import torch
        G, hd = self.image_size // self.patch_size, self.width // self.heads
        nb = hd // 4
        bands = 1.0 / (self.rope_theta ** (torch.arange(0, nb, 1, dtype=torch.int64).to(torch.float32) / nb))
        t = torch.arange(G, dtype=torch.float32) / G * self.rope_ref_grid
        grid = torch.stack(torch.meshgrid(t, t, indexing="ij"), dim=-1).unsqueeze(-1)     # [G, G, 2, 1]
        ang = grid * bands                                                                # [G, G, 2, nb]
        sin = ang.sin().reshape(G * G, -1).repeat_interleave(2, -1)
        cos = ang.cos().reshape(G * G, -1).repeat_interleave(2, -1)
        return torch.stack([cos, sin], dim=1).contiguous()

    @property
    def gflop_per_image(self) -> float:
        """Algorithmic FLOPs (2MNK per GEMM, attention 4 T^2 W per layer) — SURVEY.md §8(d)."""
        T, W, F = self.tokens, self.width, self.mlp_dim
        patch = 2 * (self.image_size // self.patch_size) ** 2 * W * 3 * self.patch_size ** 2
        layer = 2 * T * W * (3 * W) + 2 * T * W * W + 2 * (3 if self.eva else 2) * T * W * F + 4 * T * T * W
        if self.pool == "query":  # keys | values of every token at the pooler's width, one query, out-projection, final projection
            D = self.out_dim
            head = 2 * T * W * (2 * D) + 4 * T * D + 2 * D * D + 2 * D * D
        elif self.pool == "map":  # keys | values of every token, one query per head, proj + MLP on the pooled row
            head = 2 * T * W * (2 * W) + 4 * T * W + 2 * W * W + 2 * 2 * W * F
        else:
            head = 2 * W * self.out_dim
        return (patch + self.layers * layer + head) / 1e9


@dataclass(frozen=True)
class ClipTextArch:
    vocab: int
    ctx: int
    width: int
    layers: int
    heads: int
    mlp_dim: int
    out_dim: int
    quick_gelu: bool = False
    ln_eps: float = 1e-5
    causal: bool = True       # False: SigLIP (no_causal_mask; every one of the ctx positions is run, pooled row = the last one)
    proj_bias: bool = False   # SigLIP: text_projection is a Linear with bias
    prefix: str = ""          # checkpoint key prefix: "" for CLIP, "text." under open_clip's CustomTextCLIP (SigLIP)
    pad_id: int = 0
    cls_embed: bool = False   # CoCa (open_clip TextTransformer embed_cls): ctx - 1 token positions + a learned class embedding appended behind the
    #                           padding (position ctx - 1); the class row is the pooled one, ln_final runs on it; keys under `text.`
    hf_tokenizer: Optional[str] = None   # open_clip HFTokenizer(<name>) instead of the CLIP BPE (CLIPA: "bert-base-uncased")
    strip_sep: bool = False              # HFTokenizer(strip_sep_token=True): [SEP] ids are replaced by 0 after tokenisation

    def gflop_per_text(self, tokens: Optional[int] = None) -> float:
        T, W, F = tokens or self.ctx, self.width, self.mlp_dim
        layer = 2 * T * W * (3 * W) + 2 * T * W * W + 2 * 2 * T * W * F + 4 * T * T * W
        return (self.layers * layer + 2 * W * self.out_dim) / 1e9


@dataclass(frozen=True)
class BertArch:
    vocab: int = 30522
    max_pos: int = 512
    width: int = 768
    layers: int = 12
    heads: int = 12
    mlp_dim: int = 3072
    ln_eps: float = 1e-12
    pos_offset: int = 0  # XLM-RoBERTa / RoBERTa checkpoints: position ids start at padding_idx + 1 = 2; max_pos counts USABLE positions
    # "NewModel" encoders (Alibaba-NLP/new-impl: stella_en_400M_v5 = the reference's hf_stella entry, gte-*-en-v1.5): rotary positions
    # instead of a learned table, packed qkv_proj, gated-GELU MLP (up_gate_proj [2F, W] without bias), post-LN
    rope_theta: Optional[float] = None       # None: learned absolute positions (BERT)
    rope_ntk_factor: Optional[float] = None  # rope_scaling {"type": "ntk", "factor": f}
    glu: bool = False
    type_vocab: int = 2
    # MPNet (sentence-transformers all-mpnet-base-*): the BERT post-LN encoder without token types, RoBERTa-style position offset, and one
    # T5-style relative-position bias table [rel_buckets, heads] shared by all layers (0 = no relative bias)
    rel_buckets: int = 0
    rel_max_distance: int = 128

    def rel_bias_table(self, weight):
        """`encoder.relative_attention_bias.weight` [rel_buckets, heads] -> fp32 [heads, 2 * max_pos - 1] indexed by (key - query) +
        max_pos - 1 and multiplied by sqrt(head_dim) (= divided by the softmax scale the attention kernel applies to the sum): MPNet's
        `compute_position_bias`, bucketed exactly as transformers' MPNetEncoder.relative_position_bucket (torch ops in the same order)."""
        import math
        import torch
        span = self.max_pos
        rel = torch.arange(-(span - 1), span, dtype=torch.long)           # key position - query position
        n = -rel
        nb = self.rel_buckets // 2
        ret = (n < 0).to(torch.long) * nb
        n = torch.abs(n)
        max_exact = nb // 2
        large = max_exact + (torch.log(n.float() / max_exact) / math.log(self.rel_max_distance / max_exact) * (nb - max_exact)).to(torch.long)
        large = torch.min(large, torch.full_like(large, nb - 1))
        bucket = ret + torch.where(n < max_exact, n, large)
        return (weight.detach().to(torch.float32)[bucket].t() * math.sqrt(self.width // self.heads)).contiguous()

    def rope_inv_freq(self):
        """[head_dim / 2] inverse frequencies exactly as NewModel builds them: base^-(2i/d); with NTK scaling the module re-derives
        its cache for max_pos * factor > max_pos at construction, which fixes base *= factor and inv_freq /= factor^(2/d) for every
        sequence length (NTKScalingRotaryEmbedding.__init__ -> _set_cos_sin_cache, mixed_b = None)."""
        import torch
        d = self.width // self.heads
        base = float(self.rope_theta) * (self.rope_ntk_factor or 1.0)
        inv = 1.0 / (base ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
        if self.rope_ntk_factor:
            inv = inv / (self.rope_ntk_factor ** (2.0 / d))
        return inv

    def gflop_per_text(self, tokens: int) -> float:
        T, W, F = tokens, self.width, self.mlp_dim
        layer = 2 * T * W * (3 * W) + 2 * T * W * W + 2 * (3 if self.glu else 2) * T * W * F + 4 * T * T * W
        return self.layers * layer / 1e9


@dataclass(frozen=True)
class HfClipTextArch:
    """open_clip's HFTextEncoder text tower (CustomTextCLIP checkpoints `text.transformer.*` + `text.proj.*`): a Hugging Face encoder
    (XLM-RoBERTa here), masked-mean pooler, projection MLP Linear(W, hidden) -> GELU -> Linear(hidden, out_dim) without biases,
    hidden = (W + out_dim) // 2 (open_clip hf_model.py: proj "mlp", pooler "mean_pooler").  Texts are tokenised by the HF tokenizer
    padded to ctx = 77 with <pad>; attention mask = ids != pad_id."""
    bert: "BertArch"
    out_dim: int
    ctx: int = 77
    pad_id: int = 1
    quick_gelu: bool = False   # (the towers' activation is the HF encoder's own GELU; kept so that resolve_open_clip can `replace` it)
    causal: bool = False

    @property
    def proj_hidden(self) -> int:
        return (self.bert.width + self.out_dim) // 2

    @property
    def vocab(self) -> int:
        return self.bert.vocab

    @property
    def width(self) -> int:
        return self.bert.width

    def gflop_per_text(self, tokens: Optional[int] = None) -> float:
        return self.bert.gflop_per_text(tokens or self.ctx) + 2 * (self.bert.width + self.out_dim) * self.proj_hidden / 1e9


_TEXT_B = ClipTextArch(vocab=49408, ctx=77, width=512, layers=12, heads=8, mlp_dim=2048, out_dim=512)
_TEXT_L = ClipTextArch(vocab=49408, ctx=77, width=768, layers=12, heads=12, mlp_dim=3072, out_dim=768)
_TEXT_B_PLUS = ClipTextArch(vocab=49408, ctx=77, width=640, layers=12, heads=10, mlp_dim=2560, out_dim=640)
_TEXT_H = ClipTextArch(vocab=49408, ctx=77, width=1024, layers=24, heads=16, mlp_dim=4096, out_dim=1024)
_TEXT_BIGG = ClipTextArch(vocab=49408, ctx=77, width=1280, layers=32, heads=20, mlp_dim=5120, out_dim=1280)

# SigLIP (open_clip model configs ViT-{B,L}-16-SigLIP*: timm vit_{base,large}_patch16_siglip_* + TextTransformer, ctx 64,
# 32k sentencepiece vocabulary, LayerNorm eps 1e-6 everywhere)
def _siglip(image_size: int, large: bool = False, so400m: bool = False) -> Tuple[VitArch, ClipTextArch]:
    W, Lyr, H, Fd = (1024, 24, 16, 4096) if large else (768, 12, 12, 3072)
    if so400m:  # shape-optimised 400M: 72-wide heads (run as 96), MLP 4304 (zero-padded to 4352 at load), patch 14
        W, Lyr, H, Fd = 1152, 27, 16, 4304
    return (VitArch(image_size, 14 if so400m else 16, W, Lyr, H, Fd, W, ln_eps=1e-6, pool="map"),
            ClipTextArch(vocab=32000, ctx=64, width=W, layers=Lyr, heads=H, mlp_dim=Fd, out_dim=W, ln_eps=1e-6, causal=False,
                         proj_bias=True, prefix="text.", pad_id=1))


# open_clip architecture name -> (vision, text)
OPEN_CLIP_ARCHS = {
    "ViT-B-32": (VitArch(224, 32, 768, 12, 12, 3072, 512), _TEXT_B),
    "ViT-B-32-256": (VitArch(256, 32, 768, 12, 12, 3072, 512), _TEXT_B),
    "ViT-B-16": (VitArch(224, 16, 768, 12, 12, 3072, 512), _TEXT_B),
    "ViT-B-16-plus-240": (VitArch(240, 16, 896, 12, 14, 3584, 640), _TEXT_B_PLUS),
    "ViT-L-14": (VitArch(224, 14, 1024, 24, 16, 4096, 768), _TEXT_L),
    "ViT-L-14-336": (VitArch(336, 14, 1024, 24, 16, 4096, 768), _TEXT_L),
    # 16 heads of 80 / 88 / 104: zero-padded to 96 / 96 / 112-wide heads at load (engine/towers.py::_pad_heads)
    "ViT-H-14": (VitArch(224, 14, 1280, 32, 16, 5120, 1024), _TEXT_H),
    # CLIPs with a Hugging Face text tower: the ViT towers above with a RoBERTa / XLM-RoBERTa encoder (open_clip model configs
    # roberta-ViT-B-32, xlm-roberta-base-ViT-B-32, xlm-roberta-large-ViT-H-14; model_registry.py:257-273 in the reference)
    # CLIPA (open_clip ViT-L-14-CLIPA-336: no ln_pre, average pooling of the patch tokens with the final LayerNorm after it; unmasked text
    # tower with last-position pooling over bert-base-uncased WordPiece ids without [SEP], 32 positions; bilinear-squash preprocessing)
    "ViT-L-14-CLIPA-336": (VitArch(336, 14, 1024, 24, 16, 4096, 768, pool="avg", ln_pre=False, preprocessor="CLIPA"),
                           ClipTextArch(vocab=32000, ctx=32, width=768, layers=12, heads=12, mlp_dim=3072, out_dim=768, causal=False,
                                        hf_tokenizer="bert-base-uncased", strip_sep=True)),
    "roberta-ViT-B-32": (VitArch(224, 32, 768, 12, 12, 3072, 512, quick_gelu=True), "roberta-base:512"),   # (its model config sets quick_gelu)
    "xlm-roberta-base-ViT-B-32": (VitArch(224, 32, 768, 12, 12, 3072, 512), "xlmr-base:512"),
    "xlm-roberta-large-ViT-H-14": (VitArch(224, 14, 1280, 32, 16, 5120, 1024), "xlmr-large:1024"),
    "ViT-H-14-378": (VitArch(378, 14, 1280, 32, 16, 5120, 1024), _TEXT_H),  # 730 tokens: K / V stream through the LDS in pieces
    "ViT-g-14": (VitArch(224, 14, 1408, 40, 16, 6144, 1024), _TEXT_H),
    "ViT-bigG-14": (VitArch(224, 14, 1664, 48, 16, 8192, 1280), _TEXT_BIGG),
    # CoCa (model_registry.py:344-370; open_clip model configs coca_ViT-B-32 / coca_ViT-L-14): the contrastive half — the captioning decoder is not
    # part of an embedding
    "coca_ViT-B-32": (VitArch(224, 32, 768, 12, 12, 3072, 512, pool="query", pool_heads=8),
                      ClipTextArch(49408, 77, 512, 12, 8, 2048, 512, prefix="text.", cls_embed=True)),
    "coca_ViT-L-14": (VitArch(224, 14, 1024, 24, 16, 4096, 768, pool="query", pool_heads=8),
                      ClipTextArch(49408, 77, 768, 12, 12, 3072, 768, prefix="text.", cls_embed=True)),
    # EVA02-CLIP (model_registry.py:441-460; open_clip model configs EVA02-B-16 / EVA02-L-14 / EVA02-L-14-336: timm trunks eva02_base_patch16_clip_224,
    # eva02_large_patch14_clip_224 / _336; custom_text towers of the CLIP form under `text.`)
    "EVA02-B-16": (VitArch(224, 16, 768, 12, 12, 2048, 512, ln_eps=1e-6, ln_pre=False, eva=True),
                   ClipTextArch(49408, 77, 512, 12, 8, 2048, 512, prefix="text.")),
    "EVA02-L-14": (VitArch(224, 14, 1024, 24, 16, 2730, 768, ln_eps=1e-6, ln_pre=False, eva=True),
                   ClipTextArch(49408, 77, 768, 12, 12, 3072, 768, prefix="text.")),
    "EVA02-L-14-336": (VitArch(336, 14, 1024, 24, 16, 2730, 768, ln_eps=1e-6, ln_pre=False, eva=True),
                       ClipTextArch(49408, 77, 768, 12, 12, 3072, 768, prefix="text.")),
    "ViT-B-16-SigLIP": _siglip(224), "ViT-B-16-SigLIP-256": _siglip(256), "ViT-B-16-SigLIP-384": _siglip(384),
    "ViT-B-16-SigLIP-512": _siglip(512),
    "ViT-SO400M-14-SigLIP": _siglip(224, so400m=True), "ViT-SO400M-14-SigLIP-384": _siglip(384, so400m=True),
    "ViT-L-16-SigLIP-256": _siglip(256, large=True), "ViT-L-16-SigLIP-384": _siglip(384, large=True),
}
# architectures the registry names but which are not ViT / CLIP-text / BERT-family towers (ResNet, ConvNeXt, NLLB text towers ...)
UNSUPPORTED_HINT = ("this open_clip architecture is not runnable by the marqo_amd engine yet "
                    "(supported: " + ", ".join(sorted(OPEN_CLIP_ARCHS)) + " and their -quickgelu variants)")

# hf-hub repos the reference registry names (model_registry.py:483-494) -> the open_clip architecture their open_clip_config.json
# describes (used when that file is not on disk, e.g. with synthetic weights)
KNOWN_HF_HUB_ARCHS = {"hf-hub:Marqo/marqo-fashionCLIP": "ViT-B-16", "hf-hub:Marqo/marqo-fashionSigLIP": "ViT-B-16-SigLIP"}

# OpenAI `clip` names (clip_utils.py:295-492) -> open_clip architecture (always QuickGELU)
OPENAI_CLIP_NAMES = {"ViT-B/32": "ViT-B-32", "ViT-B/16": "ViT-B-16", "ViT-L/14": "ViT-L-14", "ViT-L/14@336px": "ViT-L-14-336"}


def resolve_open_clip(arch_name: str, pretrained: Optional[str] = None) -> Tuple[VitArch, ClipTextArch]:
    """'ViT-B-32' / 'ViT-B-32-quickgelu' (+ pretrained tag) -> (vision, text) arch or KeyError."""
    quick = False
    base = arch_name
    if base.endswith("-quickgelu"):
        base, quick = base[: -len("-quickgelu")], True
    if pretrained == "openai":  # OpenAI checkpoints were trained with QuickGELU
        quick = True
    if base not in OPEN_CLIP_ARCHS:
        raise KeyError(f"{arch_name}: {UNSUPPORTED_HINT}")
    v, t = OPEN_CLIP_ARCHS[base]
    quick = quick or v.quick_gelu
    if isinstance(t, str):   # HF text tower: "xlmr-<size>:<out_dim>" (BertArch is defined further down)
        size, out_dim = t.split(":")
        xl = {"roberta-base": BertArch(vocab=50265, max_pos=512, ln_eps=1e-5, pos_offset=2, type_vocab=1),
              "xlmr-base": BertArch(vocab=250002, max_pos=512, ln_eps=1e-5, pos_offset=2, type_vocab=1),
              "xlmr-large": BertArch(vocab=250002, max_pos=512, width=1024, layers=24, heads=16, mlp_dim=4096, ln_eps=1e-5, pos_offset=2,
                                     type_vocab=1)}[size]
        t = HfClipTextArch(bert=xl, out_dim=int(out_dim))
    return replace(v, quick_gelu=quick), replace(t, quick_gelu=quick)


# HF repo id -> BERT arch for the BERT-family registry entries (model_registry.py:616-880)
_BERT_BASE = BertArch()
_BERT_SMALL = BertArch(width=384, layers=12, heads=12, mlp_dim=1536)  # 12 heads of 32 (zero-padded to 64 at load)
_BERT_LARGE = BertArch(width=1024, layers=24, heads=16, mlp_dim=4096)
_MINILM_L6 = BertArch(width=384, layers=6, heads=12, mlp_dim=1536)   # 12 heads of 32
_MPNET_BASE = BertArch(vocab=30527, max_pos=512, ln_eps=1e-5, pos_offset=2, type_vocab=0, rel_buckets=32)
HF_BERT_ARCHS = {
    "intfloat/e5-base-v2": _BERT_BASE, "intfloat/e5-base": _BERT_BASE,
    "intfloat/e5-small-v2": _BERT_SMALL, "intfloat/e5-small": _BERT_SMALL,
    "intfloat/e5-large-v2": _BERT_LARGE, "intfloat/e5-large": _BERT_LARGE,
    "BAAI/bge-base-en": _BERT_BASE, "BAAI/bge-base-en-v1.5": _BERT_BASE,
    "BAAI/bge-small-en": _BERT_SMALL, "BAAI/bge-small-en-v1.5": _BERT_SMALL,
    "BAAI/bge-large-en": _BERT_LARGE, "BAAI/bge-large-en-v1.5": _BERT_LARGE,
    "sentence-transformers/all-MiniLM-L6-v1": _MINILM_L6, "sentence-transformers/all-MiniLM-L6-v2": _MINILM_L6,
    "sentence-transformers/all-MiniLM-L12-v2": _BERT_SMALL,
    "flax-sentence-embeddings/all_datasets_v3_MiniLM-L12": _BERT_SMALL, "flax-sentence-embeddings/all_datasets_v4_MiniLM-L12": _BERT_SMALL,
    "flax-sentence-embeddings/all_datasets_v3_MiniLM-L6": _MINILM_L6, "flax-sentence-embeddings/all_datasets_v4_MiniLM-L6": _MINILM_L6,
    "intfloat/e5-small-unsupervised": _BERT_SMALL, "intfloat/e5-base-unsupervised": _BERT_BASE, "intfloat/e5-large-unsupervised": _BERT_LARGE,
    "Snowflake/snowflake-arctic-embed-m": _BERT_BASE, "Snowflake/snowflake-arctic-embed-m-v1.5": _BERT_BASE,
    "Snowflake/snowflake-arctic-embed-l": _BERT_LARGE, "llmrails/ember-v1": _BERT_LARGE, "avsolatorio/GIST-large-Embedding-v0": _BERT_LARGE,
    # Chinese BGE: BERT with the 21128-entry Chinese WordPiece vocabulary (small: 4 layers of width 512)
    "BAAI/bge-small-zh-v1.5": BertArch(vocab=21128, width=512, layers=4, heads=8, mlp_dim=2048),
    "BAAI/bge-base-zh-v1.5": BertArch(vocab=21128), "BAAI/bge-large-zh-v1.5": BertArch(vocab=21128, width=1024, layers=24, heads=16, mlp_dim=4096),
    # MPNet encoders (relative-position attention bias, no token types, position ids from 2; vocabulary = BERT's + <s> <pad> </s> <unk> ... <mask>)
    "sentence-transformers/all-mpnet-base-v1": _MPNET_BASE, "sentence-transformers/all-mpnet-base-v2": _MPNET_BASE,
    "flax-sentence-embeddings/all_datasets_v3_mpnet-base": _MPNET_BASE, "flax-sentence-embeddings/all_datasets_v4_mpnet-base": _MPNET_BASE,
    # XLM-RoBERTa encoders
    "sentence-transformers/stsb-xlm-r-multilingual": BertArch(vocab=250002, max_pos=512, ln_eps=1e-5, pos_offset=2),
    "intfloat/multilingual-e5-small": BertArch(vocab=250037, max_pos=512, width=384, layers=12, heads=12, mlp_dim=1536, ln_eps=1e-5, pos_offset=2),
    "intfloat/multilingual-e5-base": BertArch(vocab=250002, max_pos=512, ln_eps=1e-5, pos_offset=2),
    "intfloat/multilingual-e5-large": BertArch(vocab=250002, max_pos=512, width=1024, layers=24, heads=16, mlp_dim=4096, ln_eps=1e-5, pos_offset=2),
    "intfloat/multilingual-e5-large-instruct": BertArch(vocab=250002, max_pos=512, width=1024, layers=24, heads=16, mlp_dim=4096, ln_eps=1e-5, pos_offset=2),
}


# the reference's hf_stella registry entry (model_registry.py:898-904): NewModel-large, rope theta 160000 with NTK factor 2, 8192 positions
STELLA_EN_400M = BertArch(vocab=30528, max_pos=8192, width=1024, layers=24, heads=16, mlp_dim=4096, rope_theta=160000.0, rope_ntk_factor=2.0,
                          glu=True)
HF_BERT_ARCHS["Marqo/dunzhang-stella_en_400M_v5"] = STELLA_EN_400M


def bert_arch_from_hf_config(cfg: dict) -> BertArch:
    """A local HF `config.json` -> BertArch.  model_type bert, or xlm-roberta / roberta: the same encoder with the position ids
    shifted by padding_idx + 1 (the multilingual-e5 family)."""
    mtype = cfg.get("model_type", "bert")
    if mtype == "new":  # Alibaba-NLP/new-impl NewModel (custom remote code in the reference: hf_stella)
        if cfg.get("position_embedding_type", "rope") != "rope" or cfg.get("hidden_act", "gelu") != "gelu":
            raise KeyError("NewModel checkpoints are supported with rotary positions and the gated-GELU MLP")
        if cfg.get("layer_norm_type", "layer_norm") != "layer_norm" or not cfg.get("pack_qkv", True):
            raise KeyError("NewModel variants with rms_norm / unpacked qkv are not supported")
        rs = cfg.get("rope_scaling") or {}
        if rs and rs.get("type") != "ntk" or rs.get("mixed_b") is not None:
            raise KeyError(f"rope_scaling={rs} unsupported (ntk without mixed_b only)")
        return BertArch(vocab=cfg["vocab_size"], max_pos=cfg["max_position_embeddings"], width=cfg["hidden_size"],
                        layers=cfg["num_hidden_layers"], heads=cfg["num_attention_heads"], mlp_dim=cfg["intermediate_size"],
                        ln_eps=cfg.get("layer_norm_eps", 1e-12), rope_theta=float(cfg.get("rope_theta", 10000.0)),
                        rope_ntk_factor=float(rs["factor"]) if rs else None, glu=True, type_vocab=int(cfg.get("type_vocab_size", 2)))
    if mtype == "mpnet":   # sentence-transformers all-mpnet-base-v1 / -v2, flax all_datasets_v{3,4}_mpnet-base
        if cfg.get("hidden_act", "gelu") != "gelu":
            raise KeyError(f"hidden_act={cfg.get('hidden_act')} unsupported")
        off = int(cfg.get("pad_token_id", 1)) + 1
        return BertArch(vocab=cfg["vocab_size"], max_pos=cfg["max_position_embeddings"] - off, width=cfg["hidden_size"],
                        layers=cfg["num_hidden_layers"], heads=cfg["num_attention_heads"], mlp_dim=cfg["intermediate_size"],
                        ln_eps=cfg.get("layer_norm_eps", 1e-5), pos_offset=off, type_vocab=0,
                        rel_buckets=int(cfg.get("relative_attention_num_buckets", 32)))
    if mtype not in ("bert", "xlm-roberta", "roberta"):
        raise KeyError(f"model_type={mtype} is not a BERT-family encoder")
    if cfg.get("position_embedding_type", "absolute") != "absolute":
        raise KeyError("only absolute position embeddings are supported")
    if cfg.get("hidden_act", "gelu") != "gelu":
        raise KeyError(f"hidden_act={cfg.get('hidden_act')} unsupported")
    off = 0 if mtype == "bert" else int(cfg.get("pad_token_id", 1)) + 1
    return BertArch(vocab=cfg["vocab_size"], max_pos=cfg["max_position_embeddings"] - off, width=cfg["hidden_size"],
                    layers=cfg["num_hidden_layers"], heads=cfg["num_attention_heads"],
                    mlp_dim=cfg["intermediate_size"], ln_eps=cfg.get("layer_norm_eps", 1e-12 if mtype == "bert" else 1e-5),
                    pos_offset=off)
