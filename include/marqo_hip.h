/*
 * marqo_hip.h — C ABI of libmarqo_hip.so, the MI355X (gfx950) engine underneath
 * Marqo's s2_inference.vectorise() hot path.
 *
 * The reference (marqo-ai/marqo @ v2.13.0) has NO FFI for this path: its "plugin"
 * interface is Python duck-typing through the loader map
 * (src/marqo/s2_inference/model_registry.py:2133-2145) and the arithmetic lives in
 * un-vendored wheels (open_clip_torch 2.24.0 / transformers 4.41.2).  This header is
 * therefore the boundary the reference WOULD bind (via ctypes) to
 * replace, one for one:
 *
 *   mq_encode_image_*   <- OPEN_CLIP.encode_image -> open_clip model.encode_image
 *                          src/marqo/core/inference/embedding_models/open_clip_model.py:249-266
 *                          (+ the preprocess transform src/marqo/s2_inference/clip_utils.py:48-67
 *                             when raw uint8 pixels are handed over)
 *   mq_encode_clip_text <- OPEN_CLIP.encode_text -> open_clip model.encode_text
 *                          .../open_clip_model.py:268-286
 *   mq_encode_bert      <- HuggingFaceModel.encode -> AutoModel forward + pooling + F.normalize
 *                          .../hugging_face_model.py:172-214
 *   mq_chunk_grid_u8    <- PatchifySimple / chunk_image  src/marqo/s2_inference/processing/image.py:46-151
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer (HBM) owned by the caller (the Python
 *     host allocates through PyTorch-ROCm; the library never allocates device memory —
 *     the one exception is mq_queue_create, ABI 14, which sizes its own staging once);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every entry
 *     point only ENQUEUES work on it and returns; results are valid after the caller
 *     synchronises that stream;
 *   - no torch types, no C++ types: plain pointers, ints and POD structs;
 *   - return value: 0 = MQ_OK, negative = error; mq_last_error() gives the text
 *     (thread-local).
 *   - bf16 = the 16 high bits of an IEEE fp32 (round-to-nearest-even when produced here).
 */
#ifndef MARQO_HIP_H
#define MARQO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MQ_OK 0
#define MQ_ERR_INVALID (-1)   /* bad argument / unsupported shape */
#define MQ_ERR_HIP (-2)       /* a HIP runtime call or launch failed */
#define MQ_ERR_WORKSPACE (-3) /* caller's workspace too small */
#define MQ_ERR_UNSUPPORTED (-4) /* ABI 11: the device is not the one this library is built for (mq_check_device) */

#define MQ_ABI_VERSION 14

/* ---- activation / mask / pooling selectors ---------------------------------------- */
#define MQ_ACT_NONE 0
#define MQ_ACT_GELU 1      /* erf GELU (open_clip nn.GELU, HF "gelu") */
#define MQ_ACT_QUICKGELU 2 /* x * sigmoid(1.702 x) (OpenAI / *-quickgelu weights) */
#define MQ_ACT_SILU 3      /* x * sigmoid(x): the gate of the SwiGLU MLP (EVA02 towers; gated MLPs only) */

#define MQ_MASK_NONE 0   /* ViT: full attention inside a sequence */
#define MQ_MASK_CAUSAL 1 /* CLIP text tower */
#define MQ_MASK_CAUSAL_CLS 2 /* ABI 11 — CoCa text tower (open_clip TextTransformer embed_cls): causal, and the LAST row of a sequence — the appended class
                              * token — does not attend its own key.  open_clip 2.24.0 build_cls_mask pads the key axis of the class row's mask on the LEFT
                              * (F.pad(cls_mask, (1, 0, S, 0), value=True)): the class row sees key 0 and key j + 1 wherever text[j] != pad — i.e. the text, the
                              * FIRST pad position, and itself only when the text fills all S positions.  The caller packs [text, one pad row, class row]
                              * (text shorter than S) or [text, class row, class row] (full text: the twin stands for "itself"). */
/* key-padding masks (BERT) are expressed by packing: only real tokens are rows. */

#define MQ_PREC_BF16 0
#define MQ_PREC_FP8 1

#define MQ_POOL_MEAN 0 /* hugging_face_model.py:205-209 */
#define MQ_POOL_CLS 1  /* hugging_face_model.py:211-214 */

#define MQ_VIT_POOL_CLS 0 /* open_clip VisionTransformer: class token */
#define MQ_VIT_POOL_MAP 1 /* timm SigLIP ViT: attention-pool ('map') head */
#define MQ_VIT_POOL_AVG 2 /* open_clip VisionTransformer pool_type 'avg' with final_ln_after_pool (CLIPA): mean of the patch tokens -> ln_post -> proj;
                           * ln_pre_g / ln_pre_b may be NULL (no_ln_pre) */
#define MQ_VIT_POOL_QUERY 3 /* open_clip VisionTransformer with an AttentionalPooler behind it (CoCa, model_registry.py:344-370): class token and ln_pre as
                             * in CLIP; every token runs every block; k | v = ln_k(x) @ kv_w^T + kv_b of width pool_dim, ONE query row (query 0 of the
                             * learned queries, projected at load) -> out-projection -> ln_post -> proj.  mq_map_head carries the pooler (fc1 / fc2 NULL) */

/* GEMM epilogue flags for mq_gemm_bf16 */
#define MQ_EPI_BIAS 1      /* + bias[n] (fp32) */
#define MQ_EPI_GELU 2      /* erf GELU after bias */
#define MQ_EPI_QUICKGELU 4 /* quick GELU after bias */
#define MQ_EPI_RESIDUAL 8  /* + residual[m,n] (fp32, leading dim ldc) */
#define MQ_EPI_OUT_F32 16  /* write fp32 instead of bf16 */
#define MQ_EPI_OUT_FP8 32  /* (mq_gemm_fp8 only) write e4m3 codes = value / out_scale, saturating at +-448 */

/* ---- transformer encoder description ---------------------------------------------- */

/* One residual block.  Linear weights are bf16, row-major [out_features, in_features]
 * (PyTorch nn.Linear layout); biases and LayerNorm parameters are fp32. */
typedef struct mq_block_weights {
    const float* ln1_g; const float* ln1_b;   /* [W]  pre-LN: before attention; post-LN: after attention */
    const void*  qkv_w; const float* qkv_b;   /* [3W, W], [3W]  rows: q | k | v */
    const void*  out_w; const float* out_b;   /* [W, W], [W] */
    const float* ln2_g; const float* ln2_b;   /* [W] */
    const void*  fc1_w; const float* fc1_b;   /* [F, W], [F] */
    const void*  fc2_w; const float* fc2_b;   /* [W, F], [W] */
    /* fp8 path (precision == MQ_PREC_FP8, else ignored / NULL): e4m3 codes in the same [out, in] layout and the
     * per-output-channel fp32 scales written by mq_quantize_weights_fp8 */
    const void*  qkv_w8; const float* qkv_ws; /* [3W, W], [3W] */
    const void*  out_w8; const float* out_ws; /* [W, W],  [W]  */
    const void*  fc1_w8; const float* fc1_ws; /* [F, W],  [F]  */
    const void*  fc2_w8; const float* fc2_ws; /* [W, F],  [W]  */
    /* LayerNorm folding (pre-LN bf16 encoders; all six NULL = off): the LayerNorm in front of the QKV / fc1 GEMM folded into
     * that GEMM, LN(x) @ W^T = rstd * (x @ (g*W)^T - mean * colsum(g*W)) + (b + W @ beta):
     * *_wf = bf16(g[k] * W[n,k]) [out, in], *_sf = fp32 sum_k of those bf16 values [out], *_bf = fp32 b + W @ beta [out]. */
    const void*  qkv_wf; const float* qkv_sf; const float* qkv_bf;
    const void*  fc1_wf; const float* fc1_sf; const float* fc1_bf;
    /* sub-LayerNorms of the EVA02 blocks (timm eva.py; ABI 10; NULL = none): attn_ln over the attention output [attn_width], in front of the
     * out-projection (`scale_attn_inner`); mlp_ln over the gated hidden row up * silu(gate) [F] in front of fc2 (`scale_mlp`; mean and variance
     * over the first mq_encoder_cfg.mlp_ln_dim columns — the checkpoint's hidden width — when F is that width zero-padded to a multiple of 64). */
    const float* attn_ln_g; const float* attn_ln_b;
    const float* mlp_ln_g;  const float* mlp_ln_b;
    /* ABI 12 — the sub-LayerNorms folded into the GEMMs behind them (bf16 stream, tiled GEMM family; all six NULL = the LayerNorm kernels run):
     * out_wf = bf16(attn_ln_g[k] * out_w[n,k]) [W, attn_width], out_sf its row sums, out_bf = out_b + out_w @ attn_ln_b;
     * fc2_wf = bf16(mlp_ln_g[k] * fc2_w[n,k]) [W, F], fc2_sf, fc2_bf likewise.  The rows' (mean, rstd) come from the launch that wrote them: the
     * attention kernel's per-head sums (mq_attention_stats), the gated epilogue's per-slot sums (mq_gemm_bf16_lnrs with MQ_EPI_GLU). */
    const void*  out_wf; const float* out_sf; const float* out_bf;
    const void*  fc2_wf; const float* fc2_sf; const float* fc2_bf;
} mq_block_weights;

typedef struct mq_encoder_cfg {
    int32_t width;      /* W, multiple of 64 */
    int32_t layers;
    int32_t heads;      /* attention runs 64-, 96-, 112- or 128-wide heads: heads * that == attn_width (or width) */
    int32_t mlp_dim;    /* F, multiple of 64 */
    int32_t act;        /* MQ_ACT_* */
    int32_t post_ln;    /* 0: pre-LN (CLIP); 1: post-LN (BERT) */
    int32_t mask;       /* MQ_MASK_* */
    float   ln_eps;
    int32_t precision;  /* MQ_PREC_BF16 (0) or MQ_PREC_FP8 (width and mlp_dim multiples of 128) */
    int32_t attn_width; /* 0 = width.  heads * {64, 96, 112, 128} when the checkpoint's heads are narrower and were zero-padded at load
                         * (QKV weights [3*attn_width, W], out-projection [W, attn_width]) */
    int32_t fp8_first_layer; /* MQ_PREC_FP8: blocks [0, fp8_first_layer) run their GEMMs on bf16 operands, blocks from here on on e4m3.
                              * Quantisation noise injected in EARLY blocks is amplified by every later one (measured: the first 12 of
                              * ViT-L/14's 24 blocks cost 2-7x the cosine error of the last 12), so the loaders pick the smallest value
                              * that keeps the calibrated error inside the budget (engine/towers.py::tune_fp8); 0 = every block fp8 */
    int32_t mlp_glu;         /* 1: gated MLP: fc1_w is [2F, W] = (up | gate) rows, hidden = up * act(gate), fc2 takes the F-wide product.  2 (ABI 11, pre-LN
                              * blocks with MQ_ACT_SILU): the same with fc1_w / fc1_b (and the folded copies) interleaved 16 rows at a time as MQ_EPI_GLU
                              * takes them — the product is formed in the GEMM's epilogue.  bf16 path only:
                              * post-LN = the "NewModel" encoders (stella_en_400M_v5, gte-*-en-v1.5); pre-LN = the EVA02 vision blocks (SwiGLU:
                              * act = MQ_ACT_SILU, up = fc1_x, gate = fc1_g, optional mlp_ln). */
    /* fp8 path: static per-tensor activation scales, device fp32 [layers][2] = (attention output, MLP hidden), and the
     * calibration accumulator of the same shape (NULL = frozen scales; non-NULL = fold max|value| of this pass into it) */
    const float* d_fp8_act_scale;
    float*       d_fp8_act_amax;
    /* rotary position embedding (NULL: none — learned absolute positions are added by the embedding kernel instead): device fp32
     * [head_dim / 2] inverse frequencies (theta scaling, NTK factor etc. are folded in by the loader).  Applied to the Q and K columns
     * of the QKV buffer in place, HF "rotate_half" convention: (x1, x2) = (x[d], x[d + hd/2]) -> (x1 cos - x2 sin, x2 cos + x1 sin),
     * angle = position_in_sequence * inv_freq[d].  bf16 path only. */
    const float* d_rope_inv_freq;
    /* relative-position attention bias (NULL: none).  MPNet (sentence-transformers all-mpnet-base-*, transformers MPNetModel: one T5-style
     * bucketed bias table shared by every layer, added to the attention scores before the softmax): device fp32
     * [heads][2 * rel_span - 1], entry (h, d + rel_span - 1) = bias(h, key - query == d) * sqrt(head_dim) (i.e. divided by the softmax
     * scale, which the kernel applies to the sum), rel_span >= the longest sequence run.  bf16 path, 64-wide heads, unmasked attention. */
    const float* d_rel_bias;
    int32_t      rel_span;
    int32_t      residual_stream; /* 0 = the process default (mq_tune("residual_bf16") / MQ_RESIDUAL_BF16 for pre-LN bf16 encoders, fp32 unless set),
                                   * 1 = bf16 residual stream, 2 = fp32.  Pre-LN encoders (bf16, and e4m3 towers whose policy asks for it): x is
                                   * kept in bf16 between blocks (half the bytes of every residual epilogue and LayerNorm).  Post-LN bf16 encoders
                                   * (BERT family): the normalised bf16 rows are the residual, updated in place; the last LayerNorm writes the fp32
                                   * rows that are pooled.  The loaders decide 1 / 2 per MODEL at load on a fixed seeded batch
                                   * (engine/towers.py::tune_residual_stream / tune_fp8: bf16 only where it stays within a 1 - cos budget of the
                                   * fp32 stream).  (Took the slot of the former reserved0.) */
    int32_t      fp8_mlp_extra;   /* MQ_PREC_FP8, pre-LN encoders: the `fp8_mlp_extra` blocks in FRONT of fp8_first_layer run only their MLP half on e4m3
                                   * (LayerNorm 2 -> e4m3 rows, fc1 + activation -> e4m3, fc2 + residual) and keep LayerNorm 1 / QKV / attention /
                                   * out-projection on bf16 operands: two thirds of a block's GEMM FLOPs for the rounding noise of two of its four
                                   * GEMMs.  0 = none (every block is all-bf16 or all-e4m3).  Must be <= fp8_first_layer.  (ABI 6) */
    int32_t      rope_prefix;     /* d_rope_table: rows at the head of every sequence that are NOT rotated (1 = the class token) */
    /* 2-D rotary position embedding of the EVA02 vision towers (timm RotaryEmbeddingCat; NULL: none; ABI 10): device fp32
     * [T - rope_prefix][2][head_dim] = (cos row | sin row) per rotated position, the same for every head.  Applied to the Q and K columns of the QKV
     * buffer in place, INTERLEAVED pairs: (y[2i], y[2i+1]) = (x[2i] cos[2i] - x[2i+1] sin[2i], x[2i+1] cos[2i+1] + x[2i] sin[2i+1]).  Fixed-length
     * sequences (images), pre-LN bf16 path only. */
    const float* d_rope_table;
    int32_t      mlp_ln_dim;      /* mq_block_weights.mlp_ln_*: the un-padded hidden width the statistics run over (0 = mlp_dim) */
    int32_t      reserved2;
} mq_encoder_cfg;

/* ---- towers ------------------------------------------------------------------------ */

/* timm AttentionPoolLatent, the 'map' pooling head of the SigLIP ViTs (open_clip TimmModel, pool = "map"): one learned query
 * attends over all tokens, then x = x + mlp(norm(x)) */
typedef struct mq_map_head {
    const float* q;                            /* fp32 [W]: Linear_q(latent), computed once at load, times 1/sqrt(W / heads) */
    const void*  kv_w;   const float* kv_b;    /* bf16 [2W, W], fp32 [2W]  (keys | values) */
    const void*  proj_w; const float* proj_b;  /* bf16 [W, W], fp32 [W] */
    const float* ln_g;   const float* ln_b;    /* fp32 [W] */
    const void*  fc1_w;  const float* fc1_b;   /* bf16 [F, W], fp32 [F] */
    const void*  fc2_w;  const float* fc2_b;   /* bf16 [W, F], fp32 [W] */
} mq_map_head;

typedef struct mq_vit_weights {
    const void*  patch_w;      /* bf16 [W, Kp]  conv1 weight flattened (c, ky, kx), K zero-padded to Kp = ceil64(3*P*P) */
    const float* cls;          /* [W]   class embedding (MQ_VIT_POOL_MAP: NULL, there is no class token) */
    const float* pos;          /* [T, W] positional embedding, T = 1 + (S/P)^2 (MQ_VIT_POOL_MAP: T = (S/P)^2, patch bias added in) */
    const float* ln_pre_g; const float* ln_pre_b;   /* NULL: no pre-LayerNorm (MQ_VIT_POOL_MAP; CLIPA; the EVA02 towers) */
    const mq_block_weights* blocks;  /* host array, `layers` entries */
    const float* ln_post_g; const float* ln_post_b; /* CLIP: on the class token; MQ_VIT_POOL_MAP: the trunk's final norm, all tokens */
    const void*  proj_w;       /* bf16 [D, W]  (= visual.proj transposed); MQ_VIT_POOL_MAP: NULL (D == W, no projection) */
    const mq_map_head* map;    /* MQ_VIT_POOL_MAP only, else NULL */
    const float* proj_b;       /* fp32 [D] or NULL: bias of the projection (MQ_VIT_POOL_CLS; the timm EVA02 towers project through their classifier
                                * head, a Linear WITH bias; ABI 10) */
} mq_vit_weights;

typedef struct mq_vit_cfg {
    mq_encoder_cfg enc;
    int32_t image_size;  /* S: 224 */
    int32_t patch_size;  /* P: 32 / 14 / 16 */
    int32_t out_dim;     /* D */
    float mean[3];       /* preprocessing normalisation (clip_utils.py:32-33 by default) */
    float std[3];
    int32_t pool;        /* MQ_VIT_POOL_CLS (0): open_clip VisionTransformer — class token, ln_pre, ln_post(class token) @ proj;
                          * MQ_VIT_POOL_MAP (1): timm SigLIP ViT — no class token, no ln_pre, norm(all tokens) -> attention-pool head */
    int32_t map_mlp_dim; /* F of the attention-pool head's MLP (MQ_VIT_POOL_MAP) */
    int32_t pool_dim;    /* MQ_VIT_POOL_QUERY: width Dp of the attentional pooler (keys, values, output; <= enc.width); else 0 */
    int32_t pool_heads;  /* MQ_VIT_POOL_QUERY: its heads (Dp / pool_heads = 64 / 96 / 128 ...) */
} mq_vit_cfg;

typedef struct mq_clip_text_weights {
    const float* tok_emb;      /* fp32 [V, W] token embedding */
    const float* pos;          /* fp32 [ctx, W] */
    const mq_block_weights* blocks;
    const float* ln_final_g; const float* ln_final_b;
    const void*  proj_w;       /* bf16 [D, W] (= text_projection transposed) */
    const float* proj_b;       /* fp32 [D] or NULL (SigLIP text towers: text_projection is a Linear with bias) */
} mq_clip_text_weights;

typedef struct mq_clip_text_cfg {
    mq_encoder_cfg enc;
    int32_t vocab;
    int32_t ctx;      /* 77 */
    int32_t out_dim;
    int32_t cls_pos;  /* > 0 (CoCa text towers, open_clip TextTransformer embed_cls): the LAST row of every sequence is the appended class
                       * embedding — it takes position `cls_pos` (= the text context length) instead of its index; sequences may then hold
                       * ctx + 1 rows (MQ_MASK_CAUSAL_CLS: a full-length text carries the class row twice); 0 = off */
} mq_clip_text_cfg;

typedef struct mq_bert_weights {
    const float* word_emb;   /* fp32 [V, W] */
    const float* pos_emb;    /* fp32 [P, W] */
    const float* type_emb;   /* fp32 [2, W] (row 0 is used: token_type_ids are all zero) */
    const float* emb_ln_g; const float* emb_ln_b;
    const mq_block_weights* blocks;
    /* optional projection head on the pooled row (NULL: none) — open_clip's HFTextEncoder with proj "mlp" (the text tower of
     * open_clip/xlm-roberta-base-ViT-B-32 and xlm-roberta-large-ViT-H-14): Linear(W, proj_hidden) -> GELU -> Linear(proj_hidden, out_dim),
     * both without bias in open_clip (proj1_b: fp32 [proj_hidden], zeros then).  bf16 row-major [out_features, in_features].
     * proj2_w == NULL with proj1_w set: ONE biased Linear(W, out_dim) instead (multilingual_clip's `LinearTransformation`, the M-CLIP text
     * encoders of the reference's multilingual_clip loader: clip_utils.py:521-565); mq_bert_cfg.proj_hidden is 0 then. */
    const void*  proj1_w;
    const float* proj1_b;
    const void*  proj2_w;
} mq_bert_weights;

typedef struct mq_bert_cfg {
    mq_encoder_cfg enc;
    int32_t vocab;
    int32_t max_pos;
    int32_t pool;     /* MQ_POOL_* */
    int32_t proj_hidden; /* hidden width of the MLP head (multiple of 64), 0 for the single-Linear head / no head */
    int32_t out_dim;     /* 0: no projection head, d_out is [nseq, W]; else d_out is [nseq, out_dim] */
} mq_bert_cfg;

/* ---- library info ------------------------------------------------------------------ */
int         mq_abi_version(void);
const char* mq_last_error(void);
/* name of the code object arch this library was built for ("gfx950") */
const char* mq_build_arch(void);
/* ABI 11 — MQ_OK when `device` (-1 = the current one) is what the kernels are laid out for: gfx950 with 256 CUs (one whole MI355X, 8 XCDs).  The
 * persistent GEMM grids, the XCD-aware tile order and the in-kernel tail are sized for exactly that; a DPX / CPX partition or another part is
 * refused (MQ_ERR_UNSUPPORTED) by every tiled GEMM launch and, earlier, by the Python loaders at load(). */
int mq_check_device(int device);

/* Host-side staging helper (no GPU work): copy n host buffers to h_dst + h_dst_off[i] with up to `threads` copy threads, in one call
 * (the Python loaders pack a request's decoded images into one pinned buffer this way: one GIL release per request). */
int mq_host_gather(const void* const* h_src, const int64_t* h_bytes, const int64_t* h_dst_off, int64_t n, void* h_dst, int32_t threads);
/* the same with the destination's capacity in bytes: an item that would leave [0, dst_bytes) is refused (MQ_ERR_INVALID) before any copy */
int mq_host_gather_checked(const void* const* h_src, const int64_t* h_bytes, const int64_t* h_dst_off, int64_t n, void* h_dst,
                           int64_t dst_bytes, int32_t threads);

/* ---- workspace sizing (bytes of device scratch the caller must provide) ------------- */
/* rows = total token rows in the call (images: n*T; text: sum of sequence lengths);
 * nseq = number of sequences. */
size_t mq_encoder_workspace_bytes(const mq_encoder_cfg* cfg, int64_t rows, int64_t nseq);
size_t mq_vit_workspace_bytes(const mq_vit_cfg* cfg, int64_t n_images);
size_t mq_clip_text_workspace_bytes(const mq_clip_text_cfg* cfg, int64_t rows, int64_t nseq);
size_t mq_bert_workspace_bytes(const mq_bert_cfg* cfg, int64_t rows, int64_t nseq);

/* ---- hot-path entry points ----------------------------------------------------------- */

/* Image tower on raw pixels already at model resolution.
 * d_pixels: uint8 [n, S, S, 3] (HWC, RGB).  ToTensor (/255) + Normalize(mean,std) are fused
 * into the patch gather.  d_out: fp32 [n, D]; L2-normalised when normalize != 0. */
int mq_encode_image_u8(const mq_vit_cfg* cfg, const mq_vit_weights* w,
                       const uint8_t* d_pixels, int64_t n, float* d_out, int normalize,
                       void* d_workspace, size_t workspace_bytes, void* stream);

/* Image tower on already-preprocessed tensors (what the reference's `.preprocess`
 * returns, add_docs.py:130-134): d_pixels fp32 [n, 3, S, S] (CHW, normalised). */
int mq_encode_image_f32(const mq_vit_cfg* cfg, const mq_vit_weights* w,
                        const float* d_pixels, int64_t n, float* d_out, int normalize,
                        void* d_workspace, size_t workspace_bytes, void* stream);

/* CLIP text tower.  Sequences are PACKED: d_ids int32 [rows] holds the tokens of sequence
 * s at [cu_seqlens[s], cu_seqlens[s+1]) where each sequence is SOT ... EOT (the tokens after
 * EOT never influence the pooled EOT row under the causal mask, so they are not run).
 * d_cu_seqlens: int32 [nseq+1] on device; h_cu_seqlens: the same array on the host.
 * The pooled row is the LAST row of each sequence (the EOT token = argmax id,
 * open_clip text pooling).  To reproduce the reference's padded 77-token execution
 * exactly, pass every sequence with its full 77 ids and d_pool_rows = argmax rows. */
int mq_encode_clip_text(const mq_clip_text_cfg* cfg, const mq_clip_text_weights* w,
                        const int32_t* d_ids, const int32_t* d_cu_seqlens,
                        const int32_t* h_cu_seqlens, int64_t nseq,
                        const int32_t* d_pool_rows, /* int32 [nseq] absolute row to pool, or NULL = last row */
                        float* d_out, int normalize,
                        void* d_workspace, size_t workspace_bytes, void* stream);

/* BERT-family text tower + pooling (+ L2).  Packed like mq_encode_clip_text: only the
 * attention_mask==1 tokens of each sequence are rows (results are identical to the
 * reference's pad-to-longest execution because padded keys are masked out and padded rows
 * are excluded from the pool: hugging_face_model.py:205-209). */
int mq_encode_bert(const mq_bert_cfg* cfg, const mq_bert_weights* w,
                   const int32_t* d_ids, const int32_t* d_cu_seqlens,
                   const int32_t* h_cu_seqlens, int64_t nseq,
                   float* d_out, int normalize,
                   void* d_workspace, size_t workspace_bytes, void* stream);

/* ---- image preprocessing on device (K10 / K11) ----------------------------------------- */
/* All images are uint8 HWC RGB, dense (row stride = w*3), image i at d_src + h_src_off[i] with
 * size h_heights[i] x h_widths[i] (host arrays: the planning — Pillow's coefficient tables — is host
 * work, the pixel passes are device work).  Results are bit-identical to Pillow's
 * Image.resize(..., BICUBIC) (antialiased two-pass 8-bit resampler), which is what the reference's
 * transforms run on the CPU. */

/* CLIP transform up to uint8  (clip_utils.py:61-63: Resize(S, BICUBIC) -> CenterCrop(S)):
 * d_out uint8 [n, S, S, 3].  Feed it to mq_encode_image_u8 (which fuses ToTensor + Normalize) or to
 * mq_to_tensor_normalize. */
size_t mq_clip_resize_workspace_bytes(const int32_t* h_heights, const int32_t* h_widths, int64_t n, int32_t S);
int mq_clip_resize_crop_u8(const uint8_t* d_src, const int64_t* h_src_off, const int32_t* h_heights,
                           const int32_t* h_widths, int64_t n, int32_t S, uint8_t* d_out,
                           void* d_workspace, size_t workspace_bytes, void* stream);

/* Plain PIL.Image.resize((out_w, out_h), BICUBIC) of every image (no aspect preservation, no crop): the
 * chunker's working image (processing/image.py:143).  d_out uint8 [n, out_h, out_w, 3]. */
size_t mq_resize_workspace_bytes(const int32_t* h_heights, const int32_t* h_widths, int64_t n, int32_t out_h, int32_t out_w);
int mq_resize_u8(const uint8_t* d_src, const int64_t* h_src_off, const int32_t* h_heights, const int32_t* h_widths,
                 int64_t n, int32_t out_h, int32_t out_w, uint8_t* d_out, void* d_workspace, size_t workspace_bytes,
                 void* stream);

/* The same with the resampling filter chosen: filter = 2 (PIL.Image.BILINEAR) or 3 (PIL.Image.BICUBIC).  CLIPA checkpoints are
 * preprocessed with a BILINEAR squash to S x S (open_clip's _apcfg(), which the reference selects for
 * image_preprocessor = "CLIPA": core/inference/embedding_models/open_clip_model.py:87-97). */
size_t mq_resize_filter_workspace_bytes(const int32_t* h_heights, const int32_t* h_widths, int64_t n, int32_t out_h, int32_t out_w,
                                        int32_t filter);
int mq_resize_filter_u8(const uint8_t* d_src, const int64_t* h_src_off, const int32_t* h_heights, const int32_t* h_widths,
                        int64_t n, int32_t out_h, int32_t out_w, int32_t filter, uint8_t* d_out, void* d_workspace,
                        size_t workspace_bytes, void* stream);

/* Grid chunker (PatchifySimple, processing/image.py:120-151; 'simple' / 'overlap' patch methods):
 * every image is resized to 240x240 (no aspect preservation), cut into the whole image + the
 * generate_boxes(hn, wn, overlap) grid (image_utils.py:165-202), and every crop is put through the CLIP
 * transform above.  d_out uint8 [n * count, S, S, 3] with count = mq_chunk_grid_count(hn, wn, overlap);
 * h_boxes (host, may be NULL) float [n * count, 4] = (x1, y1, x2, y2) in ORIGINAL pixel coordinates
 * (rescale_box, image_utils.py:141-163). */
int    mq_chunk_grid_count(int32_t hn, int32_t wn, int32_t overlap);
size_t mq_chunk_grid_workspace_bytes(const int32_t* h_heights, const int32_t* h_widths, int64_t n,
                                     int32_t hn, int32_t wn, int32_t overlap, int32_t S);
int mq_chunk_grid_u8(const uint8_t* d_src, const int64_t* h_src_off, const int32_t* h_heights,
                     const int32_t* h_widths, int64_t n, int32_t hn, int32_t wn, int32_t overlap, int32_t S,
                     uint8_t* d_out, float* h_boxes, void* d_workspace, size_t workspace_bytes, void* stream);

/* Resize by the source image's MODE, as Pillow itself resizes it.  The reference's transform calls PIL's Image.resize on whatever mode
 * the decoder produced and converts to RGB only afterwards (src/marqo/s2_inference/clip_utils.py:61-64; open_clip image_transform,
 * open_clip_model.py:84-97), and Image.resize is mode dependent:
 *   MQ_IMG_RGB      uint8 [h, w, 3]: the given filter (2 = Image.BILINEAR, 3 = Image.BICUBIC) — what mq_clip_resize_crop_u8 /
 *                   mq_resize_filter_u8 do; also right for "L" images replicated to RGB by the caller (the filter is per band);
 *   MQ_IMG_NEAREST  uint8 [h, w, 3] converted from a palette ("P") or bilevel ("1") image: Pillow forces NEAREST for these modes
 *                   (nearest sampling commutes with the palette lookup, so the caller converts first);
 *   MQ_IMG_RGBA     uint8 [h, w, 4] (RGBA; LA expanded to RGBA by the caller), images 4-byte aligned: premultiply by alpha, resample
 *                   all four bands, un-premultiply (Pillow's "RGBa" round trip), alpha dropped = the transform's .convert("RGB").
 *                   An image that needs no resize at all passes its colour bytes through untouched, as in the reference.
 * crop != 0: Resize(out_h) on the shorter side + CenterCrop(out_h) (the CLIP transform; out_w == out_h); crop == 0: plain
 * Image.resize((out_w, out_h)).  d_out uint8 [n, out_h, out_w, 3] in every mode; bit-identical to Pillow. */
#define MQ_IMG_RGB 0
#define MQ_IMG_NEAREST 1
#define MQ_IMG_RGBA 2
size_t mq_resize_mode_workspace_bytes(const int32_t* h_heights, const int32_t* h_widths, int64_t n, int32_t out_h, int32_t out_w,
                                      int32_t filter, int32_t crop, int32_t mode);
int mq_resize_mode_u8(const uint8_t* d_src, const int64_t* h_src_off, const int32_t* h_heights, const int32_t* h_widths, int64_t n,
                      int32_t out_h, int32_t out_w, int32_t filter, int32_t crop, int32_t mode, uint8_t* d_out,
                      void* d_workspace, size_t workspace_bytes, void* stream);

/* Decoded Pillow images hold 4 bytes per pixel (R, G, B, pad: Imaging "RGB" storage); the loaders stage those bytes as they are (a
 * zero-copy view through Pillow's Arrow export instead of PIL's 4 -> 3 byte repack on the host, which is what Image.tobytes /
 * np.asarray — and torchvision's ToTensor in the reference, clip_utils.py:65 — spend their time on) and repack on the device.
 * d_staging: one buffer holding the pixel bytes and, at jobs_off (multiple of 8), n records {int64 src_off, int64 dst_off, int64 npix}
 * (byte offsets into d_staging / d_rgb, both multiples of 256).  d_rgb receives packed RGB. */
int mq_unpack_rgbx(const uint8_t* d_staging, int64_t jobs_off, int64_t n, int64_t max_npix, uint8_t* d_rgb, void* stream);

/* ToTensor + Normalize (clip_utils.py:65-66): uint8 [n, S, S, 3] -> fp32 [n, 3, S, S]; this is the
 * tensor the reference's `.preprocess` hands to add_docs.py:130-134. mean/std: host float[3]. */
int mq_to_tensor_normalize(const uint8_t* d_u8, float* d_out, int64_t n, int32_t S,
                           const float* mean, const float* std, void* stream);

/* Pillow's fixed-point bicubic coefficient table of one axis (host only; exported for parity tests):
 * for output positions [first, first+count) of an in_size -> out_size resize, h_bounds int32
 * [count, 2] = (first source index, tap count), h_kk int32 [count, ksize] 22-bit fixed-point weights,
 * ksize = mq_resample_ksize(in_size, out_size). */
int mq_resample_ksize(int32_t in_size, int32_t out_size);
int mq_resample_coeffs(int32_t in_size, int32_t out_size, int32_t first, int32_t count,
                       int32_t* h_bounds, int32_t* h_kk);

/* ---- building blocks (exported for parity tests and for callers that compose) -------- */

/* fp8 (OCP e4m3) GEMM, K13:  out[M,N] = epilogue( (A8[M,K] @ W8[N,K]^T) * a_scale * w_scale[n] ).
 * A8, W8: e4m3 codes, row-major (lda, ldw in bytes, multiples of 16); K % 128 == 0.  d_a_scale: fp32 [M] when
 * a_scale_per_row != 0 (written by mq_layernorm_fp8), else one device scalar (static per-tensor scale).
 * d_w_scale fp32 [N] per output channel (mq_quantize_weights_fp8).  flags: one of
 *   OUT_F32 | BIAS (bf16 out) | BIAS|GELU|OUT_FP8 | BIAS|QUICKGELU|OUT_FP8 | BIAS|RESIDUAL|OUT_F32.
 * With OUT_FP8 the result is divided by *d_out_scale (device scalar) before conversion and, when d_amax != NULL,
 * max|value| is atomically folded into *d_amax (calibration of the static scale). */
int mq_gemm_fp8(const void* d_A8, int64_t lda, const void* d_W8, int64_t ldw, const float* d_a_scale,
                int a_scale_per_row, const float* d_w_scale, const float* d_bias, const float* d_residual,
                void* d_out, int64_t ldc, const float* d_out_scale, float* d_amax,
                int64_t M, int64_t N, int64_t K, int flags, void* stream);

/* W bf16 [N,K] -> e4m3 codes [N,K] + per-row scale[N] = absmax / 448 (done once at model load). */
int mq_quantize_weights_fp8(const void* d_W_bf16, int64_t ldw, void* d_W8, int64_t ld8, float* d_scale,
                            int64_t N, int64_t K, void* stream);

/* LayerNorm with e4m3 output and a dynamic per-row scale: q[r,:] = LN(x[r,:]) / s[r], s[r] = max|LN(x[r,:])| / 448. */
int mq_layernorm_fp8(const float* d_x, const float* d_g, const float* d_b, void* d_out_fp8, float* d_row_scale,
                     float* d_out_f32 /* optional fp32 copy of LN(x) (post-LN models), may be NULL */,
                     int64_t rows, int32_t W, float eps, void* stream);

/* the same with the input rows in bf16 (x_bf16 != 0: the bf16 residual stream of an fp8 tower; d_out_f32 must then be NULL) */
int mq_layernorm_fp8_ex(const void* d_x, int x_bf16, const float* d_g, const float* d_b, void* d_out_fp8, float* d_row_scale,
                        float* d_out_f32, int64_t rows, int32_t W, float eps, void* stream);

/* Per-row e4m3 quantisation without normalisation: q[r,:] = x[r,:] / s[r], s[r] = max|x[r,:]| / 448 (the GEMM operand of the
 * first block of a post-LN fp8 encoder). */
int mq_rowquant_fp8(const float* d_x, void* d_out_fp8, float* d_row_scale, int64_t rows, int32_t W, void* stream);

/* out[M,N] = epilogue(A[M,K] @ W[N,K]^T).  A, W bf16 row-major (lda, ldw in elements);
 * K % 64 == 0, N % 4 == 0.  bias fp32 [N]; residual fp32 [M, ldc]; out bf16 or fp32 [M, ldc]
 * (residual may alias out when MQ_EPI_OUT_F32). */
int mq_gemm_bf16(const void* d_A, int64_t lda, const void* d_W, int64_t ldw,
                 const float* d_bias, const float* d_residual, void* d_out, int64_t ldc,
                 int64_t M, int64_t N, int64_t K, int flags, void* stream);

/* The search path (tensor_search.py:1876-1911 -> vectorise() with ONE query; s2_inference.py:135-146 with a one-item batch): GEMMs of
 * M <= 80 rows (a query text; one ViT-B/32 image = 50 tokens).  Column-sliced skinny kernels (csrc/gemm_small.hip): one workgroup per 16 output columns, its 4 waves split K, so the
 * weight matrix is streamed once by the whole chip.  mq_gemm_bf16 routes such calls here by itself (mq_tune("small_m", rows), 0 = off);
 * the entry points are public for tests and for callers that hold a normalisation to fuse.
 *   mq_gemm_small_bf16:    flags as mq_gemm_bf16 (BIAS | RESIDUAL without OUT_F32 = bf16 residual in / out); K % 32 == 0, N % 4 == 0.
 *   mq_ln_gemm_small_bf16: out = act(LayerNorm(x) @ W^T + bias), the LayerNorm computed in the kernel's prologue (x fp32 [M, ldx], or the
 *                          bf16 residual stream when x_bf16 != 0); M <= 32, K <= 1280; flags BIAS [| GELU | QUICKGELU]; bf16 out.
 *                          d_ln_out (may be NULL; fp32 [M, K]; must not alias d_x) receives the normalised rows — what a post-LN
 *                          encoder (BERT) carries on as its residual. */
int mq_gemm_small_bf16(const void* d_A, int64_t lda, const void* d_W, int64_t ldw, const float* d_bias, const void* d_residual,
                       void* d_out, int64_t ldc, int64_t M, int64_t N, int64_t K, int flags, void* stream);
int mq_ln_gemm_small_bf16(const void* d_x, int64_t ldx, int x_bf16, const float* d_ln_g, const float* d_ln_b, float eps, const void* d_W,
                          int64_t ldw, const float* d_bias, void* d_out, int64_t ldc, int64_t M, int64_t N, int64_t K, int flags,
                          float* d_ln_out, void* stream);

/* The LayerNorm of a pre-LN block folded into the GEMM behind it (csrc/gemm_epilogue.h; the towers' QKV / fc1 GEMMs on the bf16 residual stream):
 *   mq_row_stats:    d_stats fp32 [rows][2] = (mean, rstd) of every bf16 row of d_x [rows, W] (W % 8 == 0, W <= 2048) — ONE read pass, 8 bytes written
 *                    per row; the LayerNorm launch it replaces read AND wrote the whole stream.
 *   mq_gemm_bf16_ln: out bf16 [M, N] = act( LN(A) @ W0^T + b0 ) computed as act( rstd * (A @ d_W^T - mean * d_colsum) + d_bias ) with
 *                    d_W = bf16(gamma * W0) [N, K], d_bias = b0 + W0 @ beta, d_colsum[n] = sum_k d_W[n, k], d_rowstats from mq_row_stats;
 *                    d_A: the UN-normalised bf16 rows (K = the normalised width).  flags: MQ_EPI_BIAS [| MQ_EPI_GELU | MQ_EPI_QUICKGELU].
 * Replaces what open_clip's ResidualAttentionBlock computes as ln_1 -> attn.in_proj / ln_2 -> mlp.c_fc
 * (reached from /root/reference/src/marqo/core/inference/embedding_models/open_clip_model.py:249-266).
 *   mq_gemm_bf16_rs + mq_row_stats_finalize: when the rows were just written by a residual GEMM of the bf16 stream, that GEMM can leave
 *                    per-row partial sums behind (d_partials fp32 [ceil(N/64)][M][2], SLOT-major since ABI 12: (sum, sum of squares) of the bf16 values it stored per
 *                    64-column slot) and the statistics pass shrinks to a finalise over those partials — same d_stats layout as mq_row_stats.
 *                    flags of mq_gemm_bf16_rs: MQ_EPI_BIAS | MQ_EPI_RESIDUAL (bf16 in / out, in place). */
#define MQ_EPI_ROW_STATS 64
#define MQ_EPI_LN_APPLY 128
/* ABI 11 — MQ_EPI_GLU (mq_gemm_bf16 / mq_gemm_bf16_ln with MQ_EPI_BIAS): the gated MLP's product in the (up | gate) GEMM's epilogue.  d_W's N = 2 F rows
 * come INTERLEAVED 16 at a time — rows 32 j .. 32 j + 15 = up units 16 j .. 16 j + 15, rows 32 j + 16 .. 32 j + 31 = the same units' gate rows — so one lane
 * holds up AND gate of the same hidden units; out bf16 [M, F] at row stride ldc: out[m, u] = (up + b_up) * silu(gate + b_gate).  N % 32 == 0. */
#define MQ_EPI_GLU 256
int mq_gemm_bf16_rs(const void* d_A, int64_t lda, const void* d_W, int64_t ldw, const float* d_bias, const void* d_residual, void* d_out, int64_t ldc,
                    int64_t M, int64_t N, int64_t K, int flags, float* d_partials, void* stream);
int mq_row_stats_finalize(const float* d_partials, int32_t nslots, float* d_stats, int64_t rows, int32_t W, float eps, void* stream);
/* ABI 12 — LN_APPLY and ROW_STATS in one launch (csrc/gemm_bf16.hip; the EVA02 sub-LayerNorms, timm eva.py EvaAttention.norm / SwiGLU.norm, reached from
 * /root/reference/src/marqo/core/inference/embedding_models/open_clip_model.py:249-266): out = [residual +] LN(A) @ W0^T + b0 with d_W / d_bias / d_colsum
 * folded as for mq_gemm_bf16_ln, d_rowstats = (mean, rstd) of A's rows, and d_partials = (sum, sum of squares) per row and 64-column slot of what the
 * launch stores.  flags: MQ_EPI_BIAS | MQ_EPI_RESIDUAL (bf16 residual read-modify-write; d_partials [ceil(N/64)][M][2]) or MQ_EPI_BIAS | MQ_EPI_GLU
 * (d_residual NULL; out [M, N/2]; a slot = 64 GEMM columns = 32 hidden units: d_partials [ceil(N/64)][M][2] of the rounded products).
 * mq_attention_stats = mq_attention that also writes (sum, sum of squares) of every output row's rounded values per head: d_row_part [heads][rows][2] (slot-major; rows = the call's token rows). */
int mq_gemm_bf16_lnrs(const void* d_A, int64_t lda, const void* d_W, int64_t ldw, const float* d_bias, const float* d_colsum, const float* d_rowstats,
                      const void* d_residual, void* d_out, int64_t ldc, int64_t M, int64_t N, int64_t K, int flags, float* d_partials, void* stream);
int mq_attention_stats(const void* d_qkv, void* d_out, const int32_t* d_cu_seqlens, int64_t nseq, int32_t fixed_len, int32_t max_len, int32_t W,
                       int32_t heads, int32_t mask, float* d_row_part, int64_t rows, void* stream);
/* ABI 13 — attention + out-projection + residual + the LayerNorm statistics behind it in ONE launch (csrc/attn_proj.hip), for the short fixed-length
 * sequences of the ViT-B/32 image tower (the x = x + out_proj(attention(qkv)) half of open_clip's residual attention block, reached from
 * /root/reference/src/marqo/core/inference/embedding_models/open_clip_model.py:249-266): one workgroup per sequence keeps the attention output in LDS as
 * the GEMM's A operand, streams the weight, adds bias and residual in place and — holding complete rows — writes their (mean, rstd).
 *   d_qkv bf16 [rows, 3 W] (q | k | v, head-major), d_w bf16 [W, W] row-major, d_bias fp32 [W], d_x bf16 [rows, W] updated in place, d_rowstats fp32
 *   [rows][2] = (mean, rstd) with the LayerNorm's eps, as mq_row_stats_finalize leaves them (NULL: not written); rows = nseq * fixed_len, no mask.
 * The rows of d_x carry the same bits as mq_attention + mq_gemm_bf16(MQ_EPI_BIAS | MQ_EPI_RESIDUAL) would leave (same operations, same order), and the
 * statistics the bits of mq_gemm_bf16_rs + mq_row_stats_finalize (same per-lane chains, lane folds and slot order).  mq_attention_proj_ok: 1 for the shapes it takes
 * (1..64 tokens, W = 768, 12 heads), which is what the towers ask before planning a block around it (mq_tune("attn_proj", 0) turns that off).
 * d_pf_a / d_pf_b (may be NULL): weight ranges of the GEMMs behind the launch, touched one dword per 128-byte line so that they sit in the Infinity
 * Cache when those GEMMs start (what mq_gemm_bf16_rsf's d_pf_* are to the finalise this launch replaces). */
int mq_attention_proj_ok(int64_t nseq, int32_t fixed_len, int32_t W, int32_t heads);
int mq_attention_proj(const void* d_qkv, const void* d_w, const float* d_bias, void* d_x, float* d_rowstats, int64_t nseq, int32_t fixed_len, int32_t W,
                      int32_t heads, float eps, const void* d_pf_a, size_t pf_a_bytes, const void* d_pf_b, size_t pf_b_bytes, void* stream);
/* ABI 13 — mq_gemm_bf16_ln for fixed-length sequences of <= 64 rows, one workgroup per sequence (csrc/panel_gemm.hip: the QKV and fc1 GEMMs of the ViT-B/32
 * image tower at chip-filling batches; same reference call site as mq_attention_proj): the sequence's rows are staged once in LDS as the MFMA token operand
 * and the weight streams through per-wave rings in passes of 768 output columns.  Operands as mq_gemm_bf16_ln (d_W / d_bias / d_colsum folded, d_rowstats =
 * (mean, rstd) per row), K = 768, N a multiple of 768, rows = nseq * fixed_len; flags MQ_EPI_BIAS [| MQ_EPI_GELU | MQ_EPI_QUICKGELU]; d_out bf16 [rows, ldc].
 * Output bits = mq_gemm_bf16_ln's (same k order, same epilogue order).  Measured 17-30 % slower than the tiled kernel at the tower's shapes (3.5 / 4.7 MB of
 * weight per CU and image; profiles/r06za_panel_gemm_ab.txt): an opt-in — mq_encoder_forward takes it only under mq_tune("panel_gemm", n) / MQ_PANEL_GEMM=n,
 * from n sequences up when they fill the chip's 256 CUs in whole rounds to >= 3/4 (default 0 = never). */
int mq_panel_gemm_ln_ok(int64_t nseq, int32_t fixed_len, int64_t N, int64_t K);
int mq_panel_gemm_ln(const void* d_x, const void* d_W, const float* d_bias, const float* d_colsum, const float* d_rowstats, void* d_out, int64_t ldc,
                     int64_t nseq, int32_t fixed_len, int64_t N, int64_t K, int flags, void* stream);
/* ABI 11 — mq_gemm_bf16_rsf = mq_gemm_bf16_rs + the finalise, in ONE launch: on return (stream order) d_stats holds (mean, rstd) of every row of d_out,
 * bit for bit what mq_row_stats_finalize would have written (same slot order, same expression).  The last wave to arrive at a row band's counter sums
 * the band's partials inside the GEMM's own launch (csrc/gemm_epilogue.h, GemmLn::band_ctr); d_band_ctr: mq_gemm_band_counters(M) 32-bit counters, all
 * zero on entry, all zero again when the launch has finished (NULL: the finalise runs as a second launch, as before ABI 11).  d_pf_a / d_pf_b
 * (nullable): weight ranges of the GEMMs behind this one, touched one dword per 128-byte line on the way out (the prefetch the finalise launch carried). */
int64_t mq_gemm_band_counters(int64_t M);
int mq_gemm_bf16_rsf(const void* d_A, int64_t lda, const void* d_W, int64_t ldw, const float* d_bias, const void* d_residual, void* d_out, int64_t ldc,
                     int64_t M, int64_t N, int64_t K, int flags, float* d_partials, float* d_stats, float eps, uint32_t* d_band_ctr,
                     const void* d_pf_a, size_t pf_a_bytes, const void* d_pf_b, size_t pf_b_bytes, void* stream);
int mq_row_stats(const void* d_x_bf16, float* d_stats, int64_t rows, int32_t W, float eps, void* stream);
int mq_gemm_bf16_ln(const void* d_A, int64_t lda, const void* d_W, int64_t ldw, const float* d_bias, const float* d_colsum, const float* d_rowstats,
                    void* d_out, int64_t ldc, int64_t M, int64_t N, int64_t K, int flags, void* stream);

/* y = LayerNorm(x) * g + b over the last dim.  x fp32 [rows, W] gathered through an optional
 * row index (d_row_idx int32 [rows], NULL = identity).  Writes bf16 (d_out_bf16) and/or fp32
 * (d_out_f32); either may be NULL. */
int mq_layernorm(const float* d_x, const int32_t* d_row_idx, const float* d_g, const float* d_b,
                 void* d_out_bf16, float* d_out_f32, int64_t rows, int32_t W, float eps, void* stream);
/* the same with the input rows in bf16 (x_bf16 != 0): the LayerNorm of the towers' bf16 residual-stream form (mq_tune("residual_bf16", 1)) */
int mq_layernorm_ex(const void* d_x, int x_bf16, const int32_t* d_row_idx, const float* d_g, const float* d_b,
                    void* d_out_bf16, float* d_out_f32, int64_t rows, int32_t W, float eps, void* stream);

/* Multi-head attention over packed sequences.  d_qkv bf16 [rows, 3W] (q | k | v, head h at
 * columns h*hd .. h*hd+hd-1 of each third, hd = W / heads in {64, 96, 112, 128}; softmax scale 1/sqrt(hd)).
 * d_out bf16 [rows, W].  Either pass fixed_len > 0 (all sequences have that length, nseq = rows / fixed_len)
 * or d_cu_seqlens int32 [nseq+1] with max_len = the longest sequence (<= 8192; sequences beyond 640 keys at
 * hd = 64 / 320 keys at wider heads stream their K / V through the LDS in pieces). */
int mq_attention(const void* d_qkv, void* d_out, const int32_t* d_cu_seqlens, int64_t nseq,
                 int32_t fixed_len, int32_t max_len, int32_t W, int32_t heads, int32_t mask,
                 void* stream);

/* mq_attention (unmasked, bf16 out, 64-wide heads) with an additive relative-position bias on the scores: d_rel_bias as described at
 * mq_encoder_cfg.d_rel_bias. */
int mq_attention_bias(const void* d_qkv, void* d_out, const int32_t* d_cu_seqlens, int64_t nseq, int32_t fixed_len,
                      int32_t max_len, int32_t W, int32_t heads, const float* d_rel_bias, int32_t rel_span, void* stream);

/* mq_attention with an optional e4m3 output (out_fp8 != 0: d_out holds codes = value / *d_out_scale; max|value| is
 * folded into *d_amax when it is non-NULL). */
int mq_attention_ex(const void* d_qkv, void* d_out, const int32_t* d_cu_seqlens, int64_t nseq,
                    int32_t fixed_len, int32_t max_len, int32_t W, int32_t heads, int32_t mask,
                    int32_t out_fp8, const float* d_out_scale, float* d_amax, void* stream);

/* One full encoder stack, in place on the fp32 residual stream d_x [rows, W]. */
int mq_encoder_forward(const mq_encoder_cfg* cfg, const mq_block_weights* blocks,
                       float* d_x, int64_t rows, const int32_t* d_cu_seqlens, int64_t nseq,
                       int32_t fixed_len, int32_t max_len,
                       void* d_workspace, size_t workspace_bytes, void* stream);

/* mq_encoder_forward when only the rows listed in d_out_rows (int32 [n_out_rows], absolute row numbers) are read
 * afterwards — class-token / EOT / CLS pooling.  Every row still feeds the last block's keys and values, but its
 * out-projection, MLP and LayerNorms run on the listed rows only; those rows come out bit-identical to
 * mq_encoder_forward, every other row of d_x is unspecified on return.  The towers use this internally. */
int mq_encoder_forward_rows(const mq_encoder_cfg* cfg, const mq_block_weights* blocks,
                            float* d_x, int64_t rows, const int32_t* d_cu_seqlens, int64_t nseq,
                            int32_t fixed_len, int32_t max_len,
                            const int32_t* d_out_rows, int64_t n_out_rows,
                            void* d_workspace, size_t workspace_bytes, void* stream);

/* rows of x: out[r,:] = x[r,:] / ||x[r,:]||_2   (in place allowed) */
int mq_l2_normalize(const float* d_x, float* d_out, int64_t rows, int32_t D, void* stream);

/* ---- text tokenisation on device (K14) ------------------------------------------------------------------- */
/* The reference tokenises on the host with third-party code (open_clip SimpleTokenizer at
 * src/marqo/core/inference/embedding_models/open_clip_model.py:277, transformers BertTokenizer at
 * .../hugging_face_model.py:179-185).  These entry points run the same published algorithms (one GPU thread per text for the
 * splitting, one per word for the vocabulary work) on any UTF-8 text.  Character handling is table-driven (d_unicode: one 64-bit
 * entry per code point below 0x30000 — flags drop / whitespace / isolate / needs-host / letter / number / Hangul plus the
 * lower-cased, accent-stripped output code points — built on the host from Python's unicodedata / str.lower / regex, so it agrees
 * with the host tokenisers by construction).  Texts whose treatment depends on context the table cannot express (a flagged code
 * point, a code point >= 0x30000, a word longer than the per-thread scratch) come back with d_status[i] = 1, d_lens[i] = 0 and
 * are tokenised by the host tokeniser; every other text gets ids identical to the host ones.
 * Texts are UTF-8 bytes packed back to back: text i = d_text[d_offsets[i] .. d_offsets[i+1]).
 * The hash tables are built once per vocabulary by the host (marqo_amd/engine/gpu_tokenizers.py); layouts: tokenize_algo.h. */
typedef struct mq_wordpiece_vocab {
    const void*    d_slots;   /* mq_wp_entry[n_slots]: {u64 fnv1a hash, i32 id (-1 empty), u32 (pool_off << 8 | cont << 7 | len)} */
    const uint8_t* d_pool;    /* piece bytes (without the "##" prefix) */
    uint32_t n_slots;         /* power of two */
    int32_t unk_id, cls_id, sep_id, pad_id;
    int32_t lower;            /* do_lower_case (informational: the behaviour is in d_unicode) */
    int32_t max_word_chars;   /* 100 */
    const uint64_t* d_unicode;  /* [0x30000] character table for this vocabulary's basic tokenisation (tokenize_algo.h) */
} mq_wordpiece_vocab;

typedef struct mq_clip_bpe_vocab {
    const void*     d_slots;        /* mq_bpe_entry[n_slots]: {u32 key = a << 16 | b (0xffffffff empty), u32 rank, u32 merged id, u32 pad} */
    const uint16_t* d_byte_id;      /* [256] id of the one-byte symbol */
    const uint16_t* d_byte_end_id;  /* [256] id of the word-final one-byte symbol (unit + "</w>") */
    uint32_t n_slots;               /* power of two */
    int32_t sot_id, eot_id;
    int32_t lower;
    const uint64_t* d_unicode;      /* [0x30000] character table: lower-casing + the \s / \p{L} / \p{N} classes of the pre-tokeniser regex */
} mq_clip_bpe_vocab;

/* Scratch of one tokenisation call (spans, counts, per-byte piece slots): total_bytes = d_offsets[n], cap_tokens = max_length
 * (WordPiece) or ctx (CLIP). */
size_t mq_tokenize_workspace_bytes(int64_t n, int64_t total_bytes, int32_t cap_tokens);

/* BERT WordPiece: d_ids int32 [n, ld] rows = [CLS] pieces... [SEP] then pad_id (truncated to max_length like
 * truncation=True, max_length=...), d_lens[i] = tokens in row i including CLS / SEP. */
int mq_tokenize_wordpiece(const mq_wordpiece_vocab* v, const uint8_t* d_text, const int64_t* d_offsets, int64_t n,
                          int64_t total_bytes, int32_t max_length, int32_t* d_ids, int64_t ld, int32_t* d_lens,
                          int32_t* d_status, void* d_workspace, size_t workspace_bytes, void* stream);

/* CLIP byte-level BPE: d_ids int32 [n, ctx] rows = SOT ids... EOT, zero padded; over-long texts are cut to ctx with EOT in
 * the last position (open_clip tokenize).  d_lens[i] = SOT..EOT length (what mq_encode_clip_text packs to). */
int mq_tokenize_clip_bpe(const mq_clip_bpe_vocab* v, const uint8_t* d_text, const int64_t* d_offsets, int64_t n,
                         int64_t total_bytes, int32_t ctx, int32_t* d_ids, int32_t* d_lens, int32_t* d_status,
                         void* d_workspace, size_t workspace_bytes, void* stream);

/* SentencePiece unigram (XLM-RoBERTa checkpoints of the multilingual-e5 family; T5-style vocabularies of the SigLIP towers): the model's
 * own normaliser as a per-code-point map + whitespace rules, then the unigram Viterbi search (tokenize_algo.h).  Rows are
 * [prefix_id] ids + id_offset ... suffix_id pad_id..., truncated to max_length; <unk> is written as unk_out. */
typedef struct mq_sentencepiece_vocab {
    const void*     d_slots;      /* mq_sp_entry[n_slots]: every piece and every proper prefix of a piece */
    const uint8_t*  d_pool;       /* piece bytes */
    const float*    d_score;      /* [vocab] */
    const uint32_t* d_nmap;       /* [0x30000] (npool offset << 8) | normalised bytes; 0xff = needs host */
    const uint8_t*  d_npool;
    const uint8_t*  d_ccc;        /* [0x30000] canonical combining classes */
    uint32_t n_slots;             /* power of two */
    int32_t unk_id;
    float   unk_score;            /* min piece score - 10 */
    int32_t add_dummy_prefix, remove_extra_ws, max_piece_bytes;
    int32_t prefix_id, suffix_id, pad_id, id_offset, unk_out;
} mq_sentencepiece_vocab;

size_t mq_tokenize_sentencepiece_workspace_bytes(int64_t n, int64_t total_bytes, int32_t max_length);
int mq_tokenize_sentencepiece(const mq_sentencepiece_vocab* v, const uint8_t* d_text, const int64_t* d_offsets, int64_t n,
                              int64_t total_bytes, int32_t max_length, int32_t* d_ids, int64_t ld, int32_t* d_lens,
                              int32_t* d_status, void* d_workspace, size_t workspace_bytes, void* stream);

/* padded id rows -> the packed layout of the text towers: d_packed[cu[s] + j] = d_padded[s * ld + j], j < cu[s+1] - cu[s] */
int mq_pack_ids(const int32_t* d_padded, int64_t ld, const int32_t* d_cu_seqlens, int64_t nseq, int32_t* d_packed,
                void* stream);

/* Weighted combination of sub-embeddings (multimodal-combination fields and weighted multi-term queries):
 *   out[g,:] = mean over the group's terms t of ( d_weights[t] * d_emb[d_rows[t], :] ),  t in [d_cu_terms[g], d_cu_terms[g+1])
 * replacing the host numpy of  src/marqo/core/inference/tensor_fields_container.py:346-365  (mode MQ_COMBINE_NORMALIZE:
 * divide by the L2 norm unconditionally, a zero vector gives NaN like numpy) and
 * src/marqo/tensor_search/tensor_search.py:1954-1963  (mode MQ_COMBINE_NORMALIZE_IF_NONZERO).  Accumulation is fp64 like
 * the reference; d_emb fp32 [*, ld]; d_rows int32 [terms] (NULL = term t reads row t); d_out fp32 [n_groups, D]. */
#define MQ_COMBINE_RAW 0
#define MQ_COMBINE_NORMALIZE 1
#define MQ_COMBINE_NORMALIZE_IF_NONZERO 2
int mq_weighted_combine(const float* d_emb, int64_t ld, const int32_t* d_rows, const float* d_weights,
                        const int32_t* d_cu_terms, int64_t n_groups, int32_t D, int32_t mode, float* d_out,
                        void* stream);

/* ---- native request queue: cross-request batching of the request threads' small calls (ABI 14) -----------------------------------------------------
 * The reference runs up to 8 indexing + 8 search request threads (src/marqo/api/configs.py:27-28), each calling vectorise() with one query
 * (src/marqo/tensor_search/tensor_search.py, the query vectorisation) or the few chunks of one document field
 * (src/marqo/core/inference/tensor_fields_container.py:179-223).  A queue merges what concurrent callers hand over into ONE
 * mq_encode_clip_text / mq_encode_bert call per group on worker threads of its own — no interpreter lock anywhere between a caller's hand-over and
 * its wake-up (the Python-level coalescer, marqo_amd/s2_inference/coalesce.py, pays that lock on every hand-off; csrc/queue.hip).
 *   - a request = nseq sequences, PACKED host token ids (h_ids int32 [sum of h_lens], sequence s at its running offset) + h_lens int32 [nseq];
 *     mq_queue_encode blocks until the request's rows are in h_out (fp32 [nseq, out_dim]); any number of threads may call it on one queue;
 *   - "natural" batching: a lone caller's request runs at once; requests that arrive while a merged call executes form the next group (up to
 *     max_seqs sequences / max_rows token rows per tower call).  depth = worker threads = merged calls in flight (each with its own HIP stream, device
 *     scratch and pinned staging, all allocated in mq_queue_create and nowhere else); window_us = how long a group that is not full is held back for
 *     company WHILE another merged call is executing (0 = never);
 *   - tower_cfg / tower_weights: mq_clip_text_cfg + mq_clip_text_weights (MQ_QUEUE_CLIP_TEXT) or mq_bert_cfg + mq_bert_weights (MQ_QUEUE_BERT); the queue
 *     keeps the POINTERS (a tower whose policy fields change after load — fp8 calibration, residual stream — is seen as it is at each call): both
 *     structs and everything they point to must outlive the queue;
 *   - a request is validated on its caller's thread (lengths within the tower's context, ids inside the embedding table): a bad request fails alone,
 *     before it can join a group.  If a merged call itself fails every request of that group gets its status and text;
 *   - embeddings are those of the merged call: rows of a batch are independent in these towers, so a request's rows do not depend on its company
 *     as long as lone and merged call take the same kernel family (the small-row families end at 320 token rows). */
#define MQ_QUEUE_CLIP_TEXT 0
#define MQ_QUEUE_BERT 1
#define MQ_QUEUE_IMAGE_F32 2 /* an image tower (mq_vit_cfg + mq_vit_weights) on preprocessed images: a request = n DEVICE pointers to fp32 [3, S, S] tensors
                              * (what the reference's `.preprocess` returns, add_docs.py:130-134) anywhere in HBM; the worker gathers a group's images into
                              * one batch (device-to-device) and runs ONE mq_encode_image_f32.  max_rows = max_seqs (an image is one "row" here).  The
                              * images must be complete when mq_queue_encode_images is entered (the workers' streams are not ordered behind any other) */
typedef struct mq_queue mq_queue;   /* opaque */
typedef struct mq_queue_cfg {
    int32_t kind;        /* MQ_QUEUE_* */
    int32_t device;      /* HIP device ordinal the tower's weights live on */
    int32_t max_seqs;    /* sequences per merged tower call (and per request) */
    int32_t max_rows;    /* token rows per merged tower call (and per request): >= the tower's context length */
    int32_t normalize;   /* != 0: L2-normalised rows */
    int32_t depth;       /* worker threads (lanes) = merged calls in flight: 1..4.  EQUAL lanes cost heavy load its large groups (measured: depth 1 > 4 > 3 > 2,
                          * profiles/r08d_queue_depth_window_sweep.txt: a small-row tower call is host-launch-bound whatever its size); the loaders use 2 with
                          * helper_seqs = 4 — the second lane only works under light load */
    int32_t window_us;   /* see above; 0 = a group never waits */
    int32_t graphs;      /* != 0: a group of ONE sequence (the search path's lone query) replays a hipGraph of its token count, captured at the second
                          * call of that count on a worker (~100 dependent small launches: 0.45 ms enqueued one by one, 0.3 ms as a graph) */
    int32_t helper_seqs; /* > 0 (with depth > 1): the lanes beyond the first are HELPERS — they take what is waiting only while that is at most this many
                          * sequences AND the group the first lane is executing is that small too.  Light load (two or three request threads: never more than
                          * one request waiting, nothing to merge) then runs its calls side by side instead of one behind the other; under heavy load the first
                          * lane's groups are large, the helpers sleep and it forms them alone.  0 = every lane takes whatever waits */
    int32_t reserved;
} mq_queue_cfg;
typedef struct mq_queue_stats {
    uint64_t requests;            /* served (mq_queue_encode calls that reached a worker) */
    uint64_t calls;               /* tower calls */
    uint64_t merged_calls;        /* ... of more than one request */
    uint64_t failed_calls;
    uint64_t sequences, rows;     /* totals over all calls */
    uint64_t max_call_sequences;  /* the largest group so far */
    uint64_t graphs;              /* single-sequence launch sequences captured */
    uint64_t graph_replays;       /* tower calls that were one hipGraphLaunch */
} mq_queue_stats;
int mq_queue_create(const mq_queue_cfg* cfg, const void* tower_cfg, const void* tower_weights, mq_queue** out);
int mq_queue_encode(mq_queue* q, const int32_t* h_ids, const int32_t* h_lens, int64_t nseq, float* h_out);
int mq_queue_encode_images(mq_queue* q, const float* const* d_images, int64_t n, float* h_out);   /* MQ_QUEUE_IMAGE_F32: d_images = HOST array of n device pointers */
int mq_queue_get_stats(mq_queue* q, mq_queue_stats* out);
/* serves what is still pending, joins the workers, frees the staging; no mq_queue_encode may be entered after this starts */
int mq_queue_destroy(mq_queue* q);


/* Run-time selection of a kernel variant (benchmark A/B and parity tests of every variant in one process).  TEST / BENCH ONLY: the
 * knobs are relaxed atomics (csrc/common.h, mq_knob) that the launch code of every request thread reads — a new value takes effect from
 * the next launch that reads it, so set them while no request is in flight if one call must run under one setting; the loaders never
 * touch them.
 * keys: "gemm_mt" (0 = auto, else GEMM tile height in 32-row units), "gemm_cgroup" (column tiles per L2 group, 0 = row-major walk),
 * "gemm_nh" (0 = the default tile plan, 1 = the (32 MT) x 128 tiles only, 3 = the 8-wave 256 x 256 tile on every row wherever N >= 256,
 * 4 = the eager row-split plan), "gemm_tail" (1 = the big tile's in-kernel split-K tail for a ragged last row of tiles),
 * "gemm_wd" (csrc/gemm_wd.hip, the W-operand-from-global main loop: 0 = off, 2 / 3 = on with that many LDS stages of A, 6 / 7 = the same
 * with both k-halves' W loads issued together), "rs_finalize" (1 = mq_gemm_bf16_rsf finalises the row statistics inside the GEMM's launch),
 * "gemm_addr_limit_mb" (bytes / 2^20 one launch may address per operand, 0 = the 4 GiB of a 32-bit buffer offset: taller matrices go in
 * row chunks), "row_select" (0 = the towers run their last block on every row instead of the pooled rows only), "ln_fold" (0 = LayerNorm
 * kernels, 1 = folded into the QKV GEMM, 2 = and into fc1; needs the folded weights), "subln_fold" (ABI 12: 0 = the EVA02 sub-LayerNorms run as
 * LayerNorm passes instead of inside the out-projection / fc2 GEMMs; MQ_SUBLN_FOLD), "residual_bf16", "small_m" / "small_m_grouped"
 * (row limits of the skinny GEMM kernels), "ln_prefetch", "xcd_band", "attn_waves" (0 = auto, 4 / 8 wave64s per attention workgroup),
 * "attn_proj" (ABI 13: fewest fixed-length sequences from which a ViT-B/32-shaped block runs mq_attention_proj instead of attention + out-projection +
 * finalise, 0 = never; default 128, MQ_ATTN_PROJ), "panel_gemm" (ABI 13: fewest fixed-length sequences from which the folded QKV / fc1 GEMMs of a 768-wide
 * tower run as mq_panel_gemm_ln; default 0 = never, MQ_PANEL_GEMM).
 * Initial values come from the environment (MQ_GEMM_MT, MQ_GEMM_CGROUP, MQ_GEMM_NH, MQ_GEMM_TAIL, MQ_GEMM_WD, MQ_GEMM_RS_FIN, MQ_ROW_SELECT,
 * MQ_LN_FOLD, ...). */
int mq_tune(const char* key, int value);

/* ---- per-kernel timing (bench.py roofline) ------------------------------------------- */
/* When enabled, every launch of a kernel family is bracketed by hipEvents on its stream.
 * mq_profile_collect synchronises, sums the elapsed time per family and resets.
 * family ids: 0 gemm, 1 layernorm, 2 attention, 3 embed/gather, 4 pool/head, 5 preprocess */
#define MQ_PROF_FAMILIES 6
int mq_profile_enable(int on);
int mq_profile_collect(double* ms_per_family /* [MQ_PROF_FAMILIES] */,
                       int64_t* launches_per_family /* [MQ_PROF_FAMILIES] */,
                       double* gemm_flops /* total 2*M*N*K over gemm launches */);

/* ---- measurement support (bench.py `roofline.peak_sustained_measured`; not on the product path) ---------------------------- */
/* A register-resident v_mfma_f32_16x16x32_bf16 burn (no LDS, no memory; 256 CUs x 4 SIMDs x 2 waves, pseudo-random operand bits) of
 * about target_ms on `stream`; waits for it.  *tflops = the dense bf16 rate this chip sustains at the clock it holds under full MFMA
 * load, *shader_mhz = that clock (s_memtime ticks / wall time).  d_scratch: >= 32 KiB of device memory.  No reference counterpart. */
int mq_probe_mfma_peak(double target_ms, void* d_scratch, int64_t scratch_bytes, double* tflops, double* shader_mhz, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MARQO_HIP_H */
