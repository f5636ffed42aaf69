// Host-side orchestration of the three towers on one HIP stream (no allocation, no sync):
//   ViT image tower          (open_clip VisionTransformer forward,   SURVEY.md §3.5)
//   CLIP text tower          (open_clip TextTransformer forward)
//   BERT encoder + pooling   (transformers BertModel + hugging_face_model.py:172-214)
// Residual stream is fp32 (what fp16-autocast keeps it as in the reference's cuda path:
// open_clip_model.py:255-260), GEMM inputs bf16, accumulation / LN statistics / softmax fp32.
#include <stdlib.h>
#include "common.h"

// kernels in the sibling translation units
int mq_cast_bf16(const float* d_x, void* d_out, int64_t n, hipStream_t s);
int mq_patchify(const void* d_in, bool is_u8, void* d_out, int64_t n, int S, int P, int Kp,
                const float* mean, const float* std, hipStream_t s);
int mq_vit_assemble(const float* d_patch_out, const float* cls, const float* pos, const float* g, const float* b,
                    float* d_x, int64_t n, int T, int W, float eps, hipStream_t s, int x_bf16 = 0);
int mq_embed_tokens(const int32_t* d_ids, const int32_t* d_cu, int64_t nseq, const float* tok, const float* pos,
                    const float* type0, const float* g, const float* b, float* d_x, void* d_xb, int W, int vocab,
                    float eps, hipStream_t s, int last_pos = 0);
int mq_pool(const float* d_x, const int32_t* d_cu, int64_t nseq, float* d_out, int W, int pool, int normalize,
            hipStream_t s);
int mq_last_rows(const int32_t* d_cu, int32_t* d_rows, int64_t nseq, hipStream_t s);
int mq_cls_rows(int32_t* d_rows, int64_t n, int T, hipStream_t s);
int mq_map_pool(const void* d_kv, const float* d_q, void* d_out, int64_t n, int T, int W, int heads, hipStream_t s);
int mq_avg_tokens(const void* d_x, int x_bf16, float* d_out, int64_t n, int T, int first, int W, hipStream_t s);
int mq_move_rows(void* d_sparse, const int32_t* d_idx, void* d_dense, int64_t n, int64_t row_bytes, bool scatter, hipStream_t s);
int mq_rope(void* d_qkv, const int32_t* d_cu, int64_t nseq, int fixed_len, int Wa, int heads, const float* d_inv_freq, hipStream_t s);
int mq_glu(void* d_buf, int64_t rows, int F, int act, hipStream_t s, int interleaved = 0);
int mq_glu_ln(void* d_buf, int64_t rows, int F, int Ft, int act, const float* g, const float* b, float eps, hipStream_t s, int mode = 0);
int mq_rope_table(void* d_qkv, int64_t rows, int T, int prefix, int Wa, int heads, const float* d_table, hipStream_t s);
extern "C" int mq_rowquant_fp8(const float* d_x, void* d_out_fp8, float* d_row_scale, int64_t rows, int32_t W, void* stream);

// layouts the ctypes binding (marqo_amd/_lib.py) and tests/test_abi.py assume
static_assert(sizeof(mq_block_weights) == 36 * 8, "mq_block_weights layout");
static_assert(sizeof(mq_encoder_cfg) == 112, "mq_encoder_cfg layout");
static_assert(sizeof(mq_vit_cfg) == 168 && sizeof(mq_clip_text_cfg) == 128 && sizeof(mq_bert_cfg) == 136, "tower cfg layouts");
static_assert(sizeof(mq_vit_weights) == 11 * 8 && sizeof(mq_map_head) == 11 * 8 && sizeof(mq_clip_text_weights) == 7 * 8, "tower weight layouts");

// mq_tune("row_select", 0) runs the last block on every row (A/B and parity tests of the pooled-rows-only last block)
mq_knob mq_tower_row_select{getenv("MQ_ROW_SELECT") ? atoi(getenv("MQ_ROW_SELECT")) : 1};
// LayerNorm folding (gemm_epilogue.h, MQ_EPI_LN_APPLY): on the bf16 residual stream the QKV / fc1 GEMMs of a pre-LN block read the stream itself and
// apply the LayerNorm in their epilogue ((mean, rstd) per row handed in) whenever the block carries the folded tensors (*_wf / *_bf / *_sf,
// engine/towers.py) and the call is large enough for the tiled GEMM.  mq_tune("ln_fold", v) / MQ_LN_FOLD=v:
// 0 = LayerNorm kernels; 1 = folded, statistics from a read pass over the stream; 2 = folded, statistics from the residual GEMMs' partial sums
mq_knob mq_tower_ln_fold{getenv("MQ_LN_FOLD") ? atoi(getenv("MQ_LN_FOLD")) : 2};
// the EVA02 sub-LayerNorms (attn.norm in front of the out-projection, mlp.norm in front of fc2) folded into those GEMMs (round 6, ABI 12; block_eva):
// mq_tune("subln_fold", 0) / MQ_SUBLN_FOLD=0 keeps them as LayerNorm passes over the attention output / the gated product
mq_knob mq_tower_subln_fold{getenv("MQ_SUBLN_FOLD") ? atoi(getenv("MQ_SUBLN_FOLD")) : 1};
// attention + out-projection + residual + statistics in one launch (attn_proj.hip) from this many fixed-length sequences up (0 = never).  One workgroup
// per image: below ~half the chip's 256 CUs the three launches it replaces win (measured, profiles/r06x_attn_proj_batch_ab.txt: +3.4 % at 256 images,
// +2 % at 128, -1 % at 96, -4.5 % at 64)
mq_knob mq_tower_attn_proj{getenv("MQ_ATTN_PROJ") ? atoi(getenv("MQ_ATTN_PROJ")) : 128};
// (one workgroup per image: a second round of the 256 CUs has to be filled to 3/4 — 300 images would run 44 workgroups behind 256 — or the launches win)
static bool attn_proj_fill_ok(int64_t nseq) {
    const int64_t rounds = (nseq + 255) / 256;
    return mq_tower_attn_proj > 0 && nseq >= mq_tower_attn_proj && (rounds == 1 || nseq * 4 >= rounds * 256 * 3);
}
// the folded QKV / fc1 GEMMs one workgroup per image (panel_gemm.hip) from this many fixed-length sequences up, when whole rounds of the 256 CUs are filled
// to >= 3/4.  0 = never, the default: bit-identical and 17-30 % slower than the tiled kernel at these shapes (profiles/r06za_panel_gemm_ab.txt)
mq_knob mq_tower_panel_gemm{getenv("MQ_PANEL_GEMM") ? atoi(getenv("MQ_PANEL_GEMM")) : 0};
static bool panel_fill_ok(int64_t nseq) {
    const int64_t rounds = (nseq + 255) / 256;
    return mq_tower_panel_gemm > 0 && nseq >= mq_tower_panel_gemm && nseq * 4 >= rounds * 256 * 3;
}
// bf16 residual stream for the pre-LN bf16 towers (mq_tune("residual_bf16", 1) / MQ_RESIDUAL_BF16=1): x is kept in bf16 between
// blocks.  The residual GEMMs of a K = 768 tower are memory-bound on their epilogue (out-proj: 15 GFLOP against 39 MB read + 39 MB
// written of fp32 residual) and every LayerNorm re-reads the stream: bf16 halves those bytes.  Cost: one bf16 rounding per residual
// add (measured 1 - cos 5e-5 .. 1.2e-4 against the fp32 oracle at full depth, well inside the 1e-3 tolerance; the reference's own GPU path
// keeps its activations in fp16 under autocast, open_clip_model.py:255-260).  Post-LN (BERT) and fp8 towers keep the fp32 stream.
mq_knob mq_tower_residual_bf16{getenv("MQ_RESIDUAL_BF16") ? atoi(getenv("MQ_RESIDUAL_BF16")) : 0};
extern "C" int mq_layernorm_ex(const void* d_x, int x_bf16, const int32_t* d_row_idx, const float* d_g, const float* d_b, void* d_out_bf16,
                               float* d_out_f32, int64_t rows, int32_t W, float eps, void* stream);
// search path (rows <= 80): LayerNorm fused into the skinny GEMM's prologue (gemm_small.hip)
bool mq_gemm_small_ok(int64_t M, int64_t N, int64_t K, bool ln);
int mq_ln_gemm_small(const void* d_x, int64_t ldx, int x_bf16, const float* ln_g, const float* ln_b, float eps, const void* d_W, int64_t ldw,
                     const float* d_bias, void* d_out, int64_t ldc, int64_t M, int64_t N, int64_t K, int flags, float* d_ln_out,
                     const int32_t* d_rows, hipStream_t s);
// h = LN(x) ; out = epi(h @ W^T): one fused launch when the rows fit the skinny kernel, else LayerNorm kernel + tiled GEMM
// (the LayerNorm kernel also pulls W and `next_w` — the weights of the GEMM after this one — into the Infinity Cache: rowops.hip, LnExtra)
int mq_layernorm_pf(const void* d_x, int x_bf16, const int32_t* d_row_idx, const float* d_g, const float* d_b, void* d_out_bf16, float* d_out_f32,
                    int64_t rows, int32_t W, float eps, const void* pf_a, size_t bytes_a, const void* pf_b, size_t bytes_b, hipStream_t s);
// The prefetch pays only when the weights do not survive in the 256 MB Infinity Cache from one step to the next: the blocks' bf16 weights
// (x layers) at least half of it.  Measured (profiles/r03r_ln_prefetch_ab.txt): ViT-B/32 image (170 MB) GEMMs -4 %, step +2 %; ViT-L/14 (604 MB)
// +3 %; CLIP text B/32 (38 MB) and BERT-base (85 MB, ~480 MB of activations per layer) neutral to -1 % -> off there.  Decided per encoder call.
static thread_local bool t_ln_prefetch = false;
static bool weights_outlive_cache(const mq_encoder_cfg* c, int Wa) {
    const double per_layer = ((double)4 * Wa * c->width + (double)(c->mlp_glu ? 3 : 2) * c->width * c->mlp_dim) * 2.0;
    return per_layer * c->layers >= 128.0 * 1024 * 1024;
}
static inline const void* pf(const void* w) { return t_ln_prefetch ? w : nullptr; }
int mq_layernorm_fp8_pf(const void* d_x, int x_bf16, const float* d_g, const float* d_b, void* d_out_fp8, float* d_row_scale, float* d_out_f32,
                        int64_t rows, int32_t W, float eps, const void* pf_a, size_t bytes_a, const void* pf_b, size_t bytes_b, hipStream_t s);
bool mq_gemm_small_grouped_ok(int64_t M, int64_t N, int64_t K);
extern "C" int mq_gemm_bf16_ln(const void* d_A, int64_t lda, const void* d_W, int64_t ldw, const float* d_bias, const float* d_colsum,
                               const float* d_rowstats, void* d_out, int64_t ldc, int64_t M, int64_t N, int64_t K, int flags, void* stream);
extern "C" int mq_gemm_bf16_rs(const void* d_A, int64_t lda, const void* d_W, int64_t ldw, const float* d_bias, const void* d_residual, void* d_out,
                               int64_t ldc, int64_t M, int64_t N, int64_t K, int flags, float* d_partials, void* stream);
extern "C" int mq_gemm_bf16_rsf(const void* d_A, int64_t lda, const void* d_W, int64_t ldw, const float* d_bias, const void* d_residual, void* d_out,
                                int64_t ldc, int64_t M, int64_t N, int64_t K, int flags, float* d_partials, float* d_stats, float eps,
                                uint32_t* d_band_ctr, const void* d_pf_a, size_t pf_a_bytes, const void* d_pf_b, size_t pf_b_bytes, void* stream);
extern "C" int64_t mq_gemm_band_counters(int64_t M);
bool mq_gemm_rs_in_launch();
int mq_row_stats_finalize_pf(const float* d_partials, int32_t nslots, float* d_stats, int64_t rows, int32_t W, float eps, const void* pf_a, size_t bytes_a,
                             const void* pf_b, size_t bytes_b, hipStream_t s);
bool mq_row_stats_ok(int32_t W);
int mq_row_stats_pf(const void* d_x_bf16, float* d_stats, int64_t rows, int32_t W, float eps, const void* pf_a, size_t bytes_a, const void* pf_b,
                    size_t bytes_b, hipStream_t s);
// folded LayerNorm + GEMM: the bf16 stream is the A operand (no LayerNorm launch, no normalised copy)
static bool fold_ok(int xb, const void* wf, const float* bf, const float* sf, int64_t rows, int N, int K) {
    return mq_tower_ln_fold && xb && wf && bf && sf && mq_row_stats_ok(K) && !mq_gemm_small_ok(rows, N, K, false) && !mq_gemm_small_grouped_ok(rows, N, K);
}
static int ln_gemm(const void* d_x, int xb, const float* g, const float* b, float eps, void* h, const void* W, const float* bias, void* out,
                   int64_t rows, int N, int K, int flags, hipStream_t s, const void* next_w = nullptr, size_t next_bytes = 0,
                   const void* wf = nullptr, const float* bf = nullptr, const float* sf = nullptr, float* row_stats = nullptr,
                   bool stats_ready = false /* the residual GEMM that wrote d_x left its rows' (mean, rstd) in row_stats (mq_gemm_bf16_rsf) */) {
    if (mq_gemm_small_ok(rows, N, K, true)) return mq_ln_gemm_small(d_x, K, xb, g, b, eps, W, K, bias, out, N, rows, N, K, flags, nullptr, nullptr, s);
    if (row_stats && fold_ok(xb, wf, bf, sf, rows, N, K)) {
        // folded: (mean, rstd) per row — left behind by the residual GEMM in front (finalised inside its launch, which also carried the weight
        // prefetch), else from ONE read pass over the stream (which carries the prefetch) — then the GEMM reads the stream itself
        if (!stats_ready) MQ_TRY(mq_row_stats_pf(d_x, row_stats, rows, K, eps, pf(wf), (size_t)N * K * 2, pf(next_w), next_bytes, s));
        return mq_gemm_bf16_ln(d_x, K, wf, K, bf, sf, row_stats, out, N, rows, N, K, flags, s);
    }
    MQ_TRY(mq_layernorm_pf(d_x, xb, nullptr, g, b, h, nullptr, rows, K, eps, pf(W), (size_t)N * K * 2, pf(next_w), next_bytes, s));
    return mq_gemm_bf16(h, K, W, K, bias, nullptr, out, N, rows, N, K, flags, s);
}
// ln_gemm's folded form for fixed-length image sequences that fill the chip: one workgroup per image (panel_gemm.hip; same output bits).  false = not taken
static int ln_gemm_panel(bool& taken, const void* d_x, int xb, float eps, void* out, int64_t rows, int64_t nseq, int32_t fixed_len, const int32_t* d_cu_seqlens, int N, int K,
                         int flags, hipStream_t s, const void* next_w, size_t next_bytes, const void* wf, const float* bf, const float* sf, float* row_stats, bool stats_ready) {
    taken = false;
    if (!(row_stats && xb && !d_cu_seqlens && fixed_len > 0 && rows == nseq * fixed_len && panel_fill_ok(nseq) && fold_ok(xb, wf, bf, sf, rows, N, K) &&
          mq_panel_gemm_ln_ok(nseq, fixed_len, N, K)))
        return MQ_OK;
    if (!stats_ready) MQ_TRY(mq_row_stats_pf(d_x, row_stats, rows, K, eps, pf(wf), (size_t)N * K * 2, pf(next_w), next_bytes, s));
    MQ_TRY(mq_panel_gemm_ln(d_x, wf, bf, sf, row_stats, out, N, nseq, fixed_len, N, K, flags, s));
    taken = true;
    return MQ_OK;
}
// true when a tower with this encoder config keeps its residual stream in bf16
static bool stream_bf16(const mq_encoder_cfg* c) {
    // per-model policy, else the process default (bf16 towers only: an fp8 tower takes the bf16 stream when its load-time policy asks for it)
    const bool want = c->residual_stream == 1 || (c->residual_stream == 0 && mq_tower_residual_bf16 && c->precision == MQ_PREC_BF16);
    return want && (c->precision == MQ_PREC_BF16 || c->precision == MQ_PREC_FP8) && !c->post_ln && !c->d_rope_inv_freq;
}
// the EVA02 vision blocks (timm eva.py EvaBlock): pre-LN with any of — 2-D rotary positions on the patch tokens' Q / K, a LayerNorm between attention and
// out-projection, a gated (SwiGLU) MLP with a LayerNorm behind the gate.  Every row runs every block; residual stream and LayerNorm folding as in the
// plain pre-LN blocks (LN1 into the QKV GEMM, LN2 into the (up | gate) GEMM; the two sub-LayerNorms are kernels).
static bool eva_form(const mq_encoder_cfg* c) { return !c->post_ln && (c->mlp_glu || c->d_rope_table); }

// post-LN encoders (BERT family) on the bf16 stream: the normalised bf16 rows `h` ARE the residual — the out-projection / fc2 epilogues add
// into them in place (bf16 read-modify-write), the LayerNorm normalises them in place, and the fp32 copy of x disappears from the block
// (18 -> 8 bytes per element and sublayer); the last LayerNorm writes the fp32 rows the pooling reads.  Decided per model at load like the
// pre-LN form (mq_encoder_cfg.residual_stream == 1); plain bf16 encoders only (no e4m3 blocks, no gated MLP / rotary positions).
static bool stream_post16(const mq_encoder_cfg* c) {
    return c->residual_stream == 1 && c->post_ln && c->precision == MQ_PREC_BF16 && !c->mlp_glu && !c->d_rope_inv_freq;
}

namespace {
// attention of one block: with the model's relative-position bias when the encoder has one (MPNet), else the plain kernel
inline int attn_bf16(const mq_encoder_cfg* cfg, const void* qf, void* a, const int32_t* d_cu_seqlens, int64_t nseq, int32_t fixed_len,
                     int32_t max_len, int32_t Wa, hipStream_t s) {
    if (cfg->d_rel_bias) return mq_attention_bias(qf, a, d_cu_seqlens, nseq, fixed_len, max_len, Wa, cfg->heads, cfg->d_rel_bias, cfg->rel_span, s);
    return mq_attention(qf, a, d_cu_seqlens, nseq, fixed_len, max_len, Wa, cfg->heads, cfg->mask, s);
}
}  // namespace

namespace {

constexpr size_t WS_ALIGN = 256;

// bump allocator over the caller's workspace: take() returns the (256-B aligned) OFFSET
struct Off {
    size_t off = 0;
    size_t take(size_t bytes) {
        off = align_up(off, WS_ALIGN);
        const size_t o = off;
        off += bytes;
        return o;
    }
    size_t end() const { return align_up(off, WS_ALIGN); }
};

int check_encoder_cfg(const mq_encoder_cfg* c) {
    MQ_CHECK_ARG(c, "encoder cfg is null");
    MQ_CHECK_ARG(c->width >= 64 && c->width % 64 == 0, "encoder width %d must be a multiple of 64", c->width);
    const int wa = c->attn_width ? c->attn_width : c->width;
    MQ_CHECK_ARG(c->heads >= 1 && (wa == c->heads * 64 || wa == c->heads * 96 || wa == c->heads * 112 || wa == c->heads * 128),
                 "encoder attention width must be heads * {64, 96, 112, 128} (attention width %d, heads %d)", wa, c->heads);
    MQ_CHECK_ARG(c->mlp_dim >= 64 && c->mlp_dim % 64 == 0, "encoder mlp_dim %d must be a multiple of 64", c->mlp_dim);
    MQ_CHECK_ARG(c->layers >= 0, "encoder layers < 0");
    MQ_CHECK_ARG(c->act == MQ_ACT_GELU || c->act == MQ_ACT_QUICKGELU || (c->act == MQ_ACT_SILU && c->mlp_glu), "encoder act %d unsupported (MQ_ACT_SILU: gated MLPs only)", c->act);
    MQ_CHECK_ARG(c->mlp_glu >= 0 && c->mlp_glu <= 2 && (c->mlp_glu != 2 || (!c->post_ln && c->act == MQ_ACT_SILU && c->mlp_dim % 16 == 0)),
                 "mlp_glu = 2 (fc1 rows interleaved for MQ_EPI_GLU): pre-LN blocks with MQ_ACT_SILU and mlp_dim %% 16 == 0");
    MQ_CHECK_ARG(c->precision == MQ_PREC_BF16 || c->precision == MQ_PREC_FP8, "encoder precision %d unsupported", c->precision);
    if (c->precision == MQ_PREC_FP8) {
        MQ_CHECK_ARG(c->width % 128 == 0 && c->mlp_dim % 128 == 0 && wa % 128 == 0, "fp8 path needs width / mlp_dim multiples of 128");
        MQ_CHECK_ARG(c->d_fp8_act_scale, "fp8 path needs d_fp8_act_scale");
        MQ_CHECK_ARG(c->fp8_first_layer >= 0, "fp8_first_layer < 0");
        MQ_CHECK_ARG(c->fp8_mlp_extra >= 0 && c->fp8_mlp_extra <= c->fp8_first_layer && (c->fp8_mlp_extra == 0 || !c->post_ln),
                     "fp8_mlp_extra = %d must lie in [0, fp8_first_layer = %d] (pre-LN encoders only)", c->fp8_mlp_extra, c->fp8_first_layer);
        MQ_CHECK_ARG(!c->mlp_glu && !c->d_rope_inv_freq && !c->d_rope_table, "the gated-MLP / rotary encoder variants run on the bf16 path only");
    }
    if (c->d_rope_inv_freq)
        MQ_CHECK_ARG(c->post_ln == 1, "d_rope_inv_freq: the rotary positions of the post-LN NewModel family (EVA02 towers: d_rope_table)");
    if (c->mlp_glu || c->d_rope_inv_freq || c->d_rope_table)
        MQ_CHECK_ARG(wa == c->width, "the gated-MLP / rotary encoder variants run un-padded heads");
    if (c->d_rope_table)
        MQ_CHECK_ARG(!c->post_ln && !c->d_rel_bias && c->rope_prefix >= 0 && c->mask == MQ_MASK_NONE, "d_rope_table: pre-LN, unmasked, fixed-length sequences (the EVA02 vision towers)");
    MQ_CHECK_ARG(c->mlp_ln_dim >= 0 && c->mlp_ln_dim <= c->mlp_dim, "mlp_ln_dim %d must lie in [0, mlp_dim = %d]", c->mlp_ln_dim, c->mlp_dim);
    if (c->d_rel_bias)
        MQ_CHECK_ARG(c->precision == MQ_PREC_BF16 && c->mask == MQ_MASK_NONE && wa == c->width && wa == c->heads * 64 && c->rel_span >= 1,
                     "the relative-position attention bias (MPNet) runs on the bf16 path with un-padded 64-wide heads and no mask");
    return MQ_OK;
}

// attention width: heads * {64, 96, 112, 128}.  Equal to the model width for the B / L CLIP towers and BERT-base/large; models with
// other head dims (e5-small, bge-small, MiniLM: 12 heads of 32; ViT-H / g / bigG: 16 heads of 80 / 88 / 104) are loaded with their
// Q/K/V rows and out-projection columns zero-padded to 64 / 96 / 96 / 112 per head (engine/towers.py), so QKV is [rows, 3*Wa], the attention output [rows, Wa] and the out-projection has K = Wa.
int attn_width(const mq_encoder_cfg* c) { return c->attn_width ? c->attn_width : c->width; }

constexpr int64_t SMALL_LN_ROWS = 32;  // rows up to which the skinny GEMMs fuse the LayerNorm (gemm_small.hip)

// slots per row of the partial-sums buffer: a residual GEMM's 64-column slots; for the EVA02 sub-LayerNorm folds also the attention's heads and the gated
// GEMM's 64-column slots over (up | gate)
size_t part_slots(const mq_encoder_cfg* c) {
    size_t n = (size_t)(c->width + 63) / 64;
    if (c->mlp_glu == 2) {
        const size_t g = (size_t)(2 * c->mlp_dim + 63) / 64, hd = (size_t)c->heads;
        n = n > g ? n : g;
        n = n > hd ? n : hd;
    }
    return n;
}

// scratch of one encoder pass: h bf16 [rows,W] | a bf16 [rows,Wa] | big bf16 [rows, max(3Wa,F)]
size_t encoder_ws(const mq_encoder_cfg* c, int64_t rows) {
    Off cv;
    const int wa = attn_width(c);
    const int fcols = c->mlp_glu ? 2 * c->mlp_dim : c->mlp_dim;   // gated MLP: fc1 writes (up | gate)
    const size_t big = (size_t)(3 * wa > fcols ? 3 * wa : fcols);
    cv.take((size_t)rows * c->width * 2);
    cv.take((size_t)rows * wa * 2);
    cv.take((size_t)rows * big * 2);
    cv.take((size_t)rows * 4);  // per-row activation scales of the fp8 path
    cv.take((size_t)rows * 8);  // (mean, rstd) per row: the statistics of a folded LayerNorm
    cv.take((size_t)rows * part_slots(c) * 8);  // ... and the partial sums a GEMM / the attention leaves for them: (sum, sum of squares) per row and slot
    cv.take((size_t)(rows < SMALL_LN_ROWS ? rows : SMALL_LN_ROWS) * c->width * 4);  // search path, post-LN: the normalised residual (fp32)
    cv.take((size_t)mq_gemm_band_counters(rows) * 4);  // arrival counters of the in-launch statistics finalise
    return cv.end();
}

int ceil64(int v) { return (v + 63) / 64 * 64; }

}  // namespace

extern "C" size_t mq_encoder_workspace_bytes(const mq_encoder_cfg* cfg, int64_t rows, int64_t nseq) {
    (void)nseq;
    if (!cfg || rows <= 0) return 0;
    return encoder_ws(cfg, rows);
}

namespace {

// The LAST block when only `nsel` rows of the residual stream are read afterwards (class-token / EOT / CLS pooling).
// Every row still feeds K and V, so LN1 (pre-LN), the QKV GEMM and the attention run on all rows; the out-projection,
// the MLP and their LayerNorms are row-wise, so they run on the gathered rows only and are scattered back.  The
// selected rows come out bit-identical to the full block (same k-order of MFMAs per output element); the other rows
// of d_x are left as they were after the previous block.
// Scratch reuse: a_sel -> head of `h`, LN2 output -> head of `a`, fc1 output -> head of `qf`, x_sel (fp32) -> `qf`
// behind the fc1 output (all of h / a / qkv are dead by the time they are overwritten).
int last_block_selected(const mq_encoder_cfg* cfg, const mq_block_weights& b, int l, float* d_x, int64_t rows,
                        const int32_t* d_cu_seqlens, int64_t nseq, int32_t fixed_len, int32_t max_len,
                        const int32_t* d_sel, int64_t nsel, void* h, void* a, void* qf, float* row_scale, float* row_stats, bool stats_ready, float* x_sel,
                        bool f8, hipStream_t s) {
    const int W = cfg->width, F = cfg->mlp_dim, Wa = attn_width(cfg);
    const int act_flag = cfg->act == MQ_ACT_QUICKGELU ? MQ_EPI_QUICKGELU : MQ_EPI_GELU;
    const int res_flags = MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_OUT_F32;
    if (f8) {
        const float* s_attn = cfg->d_fp8_act_scale + 2 * l;
        const float* s_mlp = s_attn + 1;
        const int act8 = act_flag | MQ_EPI_BIAS | MQ_EPI_OUT_FP8;
        const int xb = stream_bf16(cfg) ? 1 : 0;
        const int rflags = xb ? (MQ_EPI_BIAS | MQ_EPI_RESIDUAL) : res_flags;
        MQ_TRY(mq_layernorm_fp8_pf(d_x, xb, b.ln1_g, b.ln1_b, h, row_scale, nullptr, rows, W, cfg->ln_eps, pf(b.qkv_w8), (size_t)3 * Wa * W, nullptr, 0, s));
        MQ_TRY(mq_gemm_fp8(h, W, b.qkv_w8, W, row_scale, 1, b.qkv_ws, b.qkv_b, nullptr, qf, 3 * Wa, nullptr, nullptr, rows, 3 * Wa, W,
                           MQ_EPI_BIAS, s));
        MQ_TRY(mq_attention_ex(qf, a, d_cu_seqlens, nseq, fixed_len, max_len, Wa, cfg->heads, cfg->mask, 1, s_attn, nullptr, s));
        MQ_TRY(mq_move_rows(a, d_sel, h, nsel, (int64_t)Wa, false, s));
        MQ_TRY(mq_move_rows(d_x, d_sel, x_sel, nsel, (int64_t)W * (xb ? 2 : 4), false, s));
        MQ_TRY(mq_gemm_fp8(h, Wa, b.out_w8, Wa, s_attn, 0, b.out_ws, b.out_b, x_sel, x_sel, W, nullptr, nullptr, nsel, W, Wa, rflags, s));
        MQ_TRY(mq_layernorm_fp8_ex(x_sel, xb, b.ln2_g, b.ln2_b, a, row_scale, nullptr, nsel, W, cfg->ln_eps, s));
        MQ_TRY(mq_gemm_fp8(a, W, b.fc1_w8, W, row_scale, 1, b.fc1_ws, b.fc1_b, nullptr, qf, F, s_mlp, nullptr, nsel, F, W, act8, s));
        MQ_TRY(mq_gemm_fp8(qf, F, b.fc2_w8, F, s_mlp, 0, b.fc2_ws, b.fc2_b, x_sel, x_sel, W, nullptr, nullptr, nsel, W, F, rflags, s));
    } else if (!cfg->post_ln) {
        const int xb = stream_bf16(cfg) ? 1 : 0;                 // bf16 residual stream: rows of 2 bytes per element, bf16 RMW epilogues
        const int rflags = xb ? (MQ_EPI_BIAS | MQ_EPI_RESIDUAL) : res_flags;
        const int64_t xrow = (int64_t)W * (xb ? 2 : 4);
        MQ_TRY(ln_gemm(d_x, xb, b.ln1_g, b.ln1_b, cfg->ln_eps, h, b.qkv_w, b.qkv_b, qf, rows, 3 * Wa, W, MQ_EPI_BIAS, s, nullptr, 0, b.qkv_wf, b.qkv_bf, b.qkv_sf, row_stats, stats_ready));
        MQ_TRY(attn_bf16(cfg, qf, a, d_cu_seqlens, nseq, fixed_len, max_len, Wa, s));
        MQ_TRY(mq_move_rows(a, d_sel, h, nsel, (int64_t)Wa * 2, false, s));
        MQ_TRY(mq_move_rows(d_x, d_sel, x_sel, nsel, xrow, false, s));
        MQ_TRY(mq_gemm_bf16(h, Wa, b.out_w, Wa, b.out_b, x_sel, x_sel, W, nsel, W, Wa, rflags, s));
        MQ_TRY(ln_gemm(x_sel, xb, b.ln2_g, b.ln2_b, cfg->ln_eps, a, b.fc1_w, b.fc1_b, qf, nsel, F, W, MQ_EPI_BIAS | act_flag, s));
        MQ_TRY(mq_gemm_bf16(qf, F, b.fc2_w, F, b.fc2_b, x_sel, x_sel, W, nsel, W, F, rflags, s));
    } else if (stream_post16(cfg)) {
        // post-LN on the bf16 stream (h = the normalised rows = the residual).  Scratch: a_sel -> head of `qf` (the QKV output is dead after the
        // attention; the fc1 output overwrites it later), h_sel (bf16) -> where x_sel lives, the final fp32 rows -> head of `a` (dead after the gather)
        const int rflags = MQ_EPI_BIAS | MQ_EPI_RESIDUAL;
        void* h_sel = x_sel;
        float* out_sel = (float*)a;
        MQ_TRY(mq_gemm_bf16(h, W, b.qkv_w, W, b.qkv_b, nullptr, qf, 3 * Wa, rows, 3 * Wa, W, MQ_EPI_BIAS, s));
        MQ_TRY(attn_bf16(cfg, qf, a, d_cu_seqlens, nseq, fixed_len, max_len, Wa, s));
        MQ_TRY(mq_move_rows(a, d_sel, qf, nsel, (int64_t)Wa * 2, false, s));
        MQ_TRY(mq_move_rows(h, d_sel, h_sel, nsel, (int64_t)W * 2, false, s));
        MQ_TRY(mq_gemm_bf16(qf, Wa, b.out_w, Wa, b.out_b, (const float*)h_sel, h_sel, W, nsel, W, Wa, rflags, s));
        MQ_TRY(mq_layernorm_ex(h_sel, 1, nullptr, b.ln1_g, b.ln1_b, h_sel, nullptr, nsel, W, cfg->ln_eps, s));
        MQ_TRY(mq_gemm_bf16(h_sel, W, b.fc1_w, W, b.fc1_b, nullptr, qf, F, nsel, F, W, MQ_EPI_BIAS | act_flag, s));
        MQ_TRY(mq_gemm_bf16(qf, F, b.fc2_w, F, b.fc2_b, (const float*)h_sel, h_sel, W, nsel, W, F, rflags, s));
        MQ_TRY(mq_layernorm_ex(h_sel, 1, nullptr, b.ln2_g, b.ln2_b, nullptr, out_sel, nsel, W, cfg->ln_eps, s));
        MQ_TRY(mq_move_rows(d_x, d_sel, out_sel, nsel, (int64_t)W * 4, true, s));
        return MQ_OK;
    } else {
        MQ_TRY(mq_gemm_bf16(h, W, b.qkv_w, W, b.qkv_b, nullptr, qf, 3 * Wa, rows, 3 * Wa, W, MQ_EPI_BIAS, s));
        MQ_TRY(attn_bf16(cfg, qf, a, d_cu_seqlens, nseq, fixed_len, max_len, Wa, s));
        MQ_TRY(mq_move_rows(a, d_sel, h, nsel, (int64_t)Wa * 2, false, s));
        MQ_TRY(mq_move_rows(d_x, d_sel, x_sel, nsel, (int64_t)W * 4, false, s));
        MQ_TRY(mq_gemm_bf16(h, Wa, b.out_w, Wa, b.out_b, x_sel, x_sel, W, nsel, W, Wa, res_flags, s));
        MQ_TRY(mq_layernorm(x_sel, nullptr, b.ln1_g, b.ln1_b, a, x_sel, nsel, W, cfg->ln_eps, s));
        MQ_TRY(mq_gemm_bf16(a, W, b.fc1_w, W, b.fc1_b, nullptr, qf, F, nsel, F, W, MQ_EPI_BIAS | act_flag, s));
        MQ_TRY(mq_gemm_bf16(qf, F, b.fc2_w, F, b.fc2_b, x_sel, x_sel, W, nsel, W, F, res_flags, s));
        MQ_TRY(mq_layernorm(x_sel, nullptr, b.ln2_g, b.ln2_b, nullptr, x_sel, nsel, W, cfg->ln_eps, s));
    }
    MQ_TRY(mq_move_rows(d_x, d_sel, x_sel, nsel, (int64_t)W * (stream_bf16(cfg) ? 2 : 4), true, s));
    return MQ_OK;
}

// One pass of the encoder over `rows` token rows: the scratch buffers carved from the caller's workspace, the decisions that hold for the whole pass,
// and ONE member function per block dataflow (round 5: split out of what had grown into a 250-line loop of eight variants; each body is the launch
// sequence of that variant, nothing else).  `x_has_partials` is the only state a block leaves for the next one.
struct EncoderPass {
    const mq_encoder_cfg* cfg;
    const mq_block_weights* blocks;
    float* d_x;
    int64_t rows;
    const int32_t* d_cu_seqlens;
    int64_t nseq;
    int32_t fixed_len, max_len;
    const int32_t* d_sel;
    int64_t nsel;
    hipStream_t s;
    int W, F, Wa;
    void *h, *a, *qf;                // h bf16 [rows, W] | a bf16 [rows, Wa] | qf bf16 [rows, max(3 Wa, fc1 columns)]: qkv, then the fc1 output
    float *row_scale, *row_stats, *row_part, *xn;
    uint32_t* band_ctr;              // arrival counters of the residual GEMMs' in-launch statistics finalise (mq_gemm_bf16_rsf), zeroed once per pass
    bool x_has_partials;             // row_stats holds (mean, rstd) of the current d_x (the residual GEMM that wrote it left them behind)
    int act_flag, res_flags, first8;

    int block_fp8(const mq_block_weights& b, int l);            // e4m3 GEMM operands, post-LN or pre-LN
    int block_eva(const mq_block_weights& b, int l);            // pre-LN with rotary table / sub-LayerNorms / gated MLP (EVA02)
    int block_pre_ln(const mq_block_weights& b, int l);         // CLIP blocks: fp32 or bf16 stream, folded LayerNorms, MLP-only e4m3
    int block_post_ln_small(const mq_block_weights& b, int l);  // BERT blocks on the search path: LayerNorms inside the skinny GEMMs
    int block_post_ln_bf16(const mq_block_weights& b, int l);   // BERT blocks, the normalised bf16 rows are the residual
    int block_post_ln(const mq_block_weights& b, int l);        // BERT / NewModel blocks on the fp32 stream (rotary positions, gated MLP)
};

int EncoderPass::block_fp8(const mq_block_weights& b, int l) {
    // same dataflow with e4m3 GEMM operands: h / a / fc1-out are fp8 (h with a dynamic per-row scale from the LN,
    // a and fc1-out with static per-tensor scales), qkv stays bf16 for the attention MFMAs, x stays fp32
    const float* s_attn = cfg->d_fp8_act_scale + 2 * l;
    const float* s_mlp = s_attn + 1;
    float* m_attn = cfg->d_fp8_act_amax ? cfg->d_fp8_act_amax + 2 * l : nullptr;
    float* m_mlp = m_attn ? m_attn + 1 : nullptr;
    const int act8 = (cfg->act == MQ_ACT_QUICKGELU ? MQ_EPI_QUICKGELU : MQ_EPI_GELU) | MQ_EPI_BIAS | MQ_EPI_OUT_FP8;
    if (cfg->post_ln) {
        // x = ln1(x + out(attn(qkv(x)))) ; x = ln2(x + fc2(act(fc1(x)))): each LayerNorm rewrites x (fp32, in place:
        // a wave holds its whole row before it stores) and leaves the e4m3 row + scale for the next GEMM
        MQ_TRY(mq_gemm_fp8(h, W, b.qkv_w8, W, row_scale, 1, b.qkv_ws, b.qkv_b, nullptr, qf, 3 * Wa, nullptr, nullptr, rows, 3 * Wa, W,
                           MQ_EPI_BIAS, s));
        MQ_TRY(mq_attention_ex(qf, a, d_cu_seqlens, nseq, fixed_len, max_len, Wa, cfg->heads, cfg->mask, 1, s_attn, m_attn, s));
        MQ_TRY(mq_gemm_fp8(a, Wa, b.out_w8, Wa, s_attn, 0, b.out_ws, b.out_b, d_x, d_x, W, nullptr, nullptr, rows, W, Wa, res_flags, s));
        MQ_TRY(mq_layernorm_fp8(d_x, b.ln1_g, b.ln1_b, h, row_scale, d_x, rows, W, cfg->ln_eps, s));
        MQ_TRY(mq_gemm_fp8(h, W, b.fc1_w8, W, row_scale, 1, b.fc1_ws, b.fc1_b, nullptr, qf, F, s_mlp, m_mlp, rows, F, W, act8, s));
        MQ_TRY(mq_gemm_fp8(qf, F, b.fc2_w8, F, s_mlp, 0, b.fc2_ws, b.fc2_b, d_x, d_x, W, nullptr, nullptr, rows, W, F, res_flags, s));
        MQ_TRY(mq_layernorm_fp8(d_x, b.ln2_g, b.ln2_b, h, row_scale, d_x, rows, W, cfg->ln_eps, s));
        return MQ_OK;
    }
    const int xb = stream_bf16(cfg) ? 1 : 0;     // the stream itself may be bf16 (decided per model at load): bf16 RMW epilogues, bf16-in LN
    const int rflags = xb ? (MQ_EPI_BIAS | MQ_EPI_RESIDUAL) : res_flags;
    MQ_TRY(mq_layernorm_fp8_pf(d_x, xb, b.ln1_g, b.ln1_b, h, row_scale, nullptr, rows, W, cfg->ln_eps, pf(b.qkv_w8), (size_t)3 * Wa * W,
                               pf(b.out_w8), (size_t)W * Wa, s));
    MQ_TRY(mq_gemm_fp8(h, W, b.qkv_w8, W, row_scale, 1, b.qkv_ws, b.qkv_b, nullptr, qf, 3 * Wa, nullptr, nullptr, rows, 3 * Wa, W,
                       MQ_EPI_BIAS, s));
    MQ_TRY(mq_attention_ex(qf, a, d_cu_seqlens, nseq, fixed_len, max_len, Wa, cfg->heads, cfg->mask, 1, s_attn, m_attn, s));
    MQ_TRY(mq_gemm_fp8(a, Wa, b.out_w8, Wa, s_attn, 0, b.out_ws, b.out_b, d_x, d_x, W, nullptr, nullptr, rows, W, Wa, rflags, s));
    MQ_TRY(mq_layernorm_fp8_pf(d_x, xb, b.ln2_g, b.ln2_b, h, row_scale, nullptr, rows, W, cfg->ln_eps, pf(b.fc1_w8), (size_t)F * W,
                               pf(b.fc2_w8), (size_t)W * F, s));
    MQ_TRY(mq_gemm_fp8(h, W, b.fc1_w8, W, row_scale, 1, b.fc1_ws, b.fc1_b, nullptr, qf, F, s_mlp, m_mlp, rows, F, W, act8, s));
    MQ_TRY(mq_gemm_fp8(qf, F, b.fc2_w8, F, s_mlp, 0, b.fc2_ws, b.fc2_b, d_x, d_x, W, nullptr, nullptr, rows, W, F, rflags, s));
    return MQ_OK;
}

int EncoderPass::block_eva(const mq_block_weights& b, int l) {
    // x += out(ln_attn(attn(rope(qkv(ln1(x)))))) ; x += fc2(ln_mlp(up * silu(gate)))  with (up | gate) = fc1(ln2(x))     (x fp32, or bf16 in the bf16-stream form)
    MQ_CHECK_ARG(!cfg->d_rope_table || (fixed_len > 0 && !d_cu_seqlens && rows == nseq * fixed_len && fixed_len > cfg->rope_prefix),
                 "mq_encoder_forward: d_rope_table needs fixed-length sequences longer than rope_prefix");
    MQ_CHECK_ARG(!b.attn_ln_g == !b.attn_ln_b && !b.mlp_ln_g == !b.mlp_ln_b && (!b.mlp_ln_g || cfg->mlp_glu), "mq_encoder_forward: layer %d: sub-LayerNorm weights must come in pairs (mlp_ln: gated MLPs only)", l);
    const int xb = stream_bf16(cfg) ? 1 : 0;
    const int rflags = xb ? (MQ_EPI_BIAS | MQ_EPI_RESIDUAL) : res_flags;
    const int fc1_cols = cfg->mlp_glu ? 2 * F : F;
    // the sub-LayerNorms folded into the GEMMs behind them (mq_gemm_bf16_lnrs): the rows' statistics come from the launch that wrote the rows — the
    // attention kernel's per-head sums, the gated epilogue's per-slot sums — through the finalise kernel; no pass over [rows, Wa] / [rows, F] of its own
    const bool sub_fold = mq_tower_subln_fold && mq_tower_ln_fold >= 2 && xb;
    const bool fold_attn_ln = sub_fold && b.attn_ln_g && b.out_wf && b.out_sf && b.out_bf && !cfg->d_rel_bias && !mq_gemm_small_ok(rows, W, Wa, false) &&
                              !mq_gemm_small_grouped_ok(rows, W, Wa);
    const void* out_w_run = fold_attn_ln ? b.out_wf : b.out_w;
    MQ_TRY(ln_gemm(d_x, xb, b.ln1_g, b.ln1_b, cfg->ln_eps, h, b.qkv_w, b.qkv_b, qf, rows, 3 * Wa, W, MQ_EPI_BIAS, s, out_w_run, (size_t)W * Wa * 2,
                   b.qkv_wf, b.qkv_bf, b.qkv_sf, row_stats, x_has_partials));
    x_has_partials = false;
    if (cfg->d_rope_table) MQ_TRY(mq_rope_table(qf, rows, fixed_len, cfg->rope_prefix, Wa, cfg->heads, cfg->d_rope_table, s));
    const bool fold_mlp = mq_tower_ln_fold >= 2 && fold_ok(xb, b.fc1_wf, b.fc1_bf, b.fc1_sf, rows, fc1_cols, W) && !mq_gemm_small_ok(rows, W, Wa, false) &&
                          !mq_gemm_small_grouped_ok(rows, W, Wa);
    if (fold_attn_ln) {
        MQ_TRY(mq_attention_stats(qf, a, d_cu_seqlens, nseq, fixed_len, max_len, Wa, cfg->heads, cfg->mask, row_part, rows, s));
        MQ_TRY(mq_row_stats_finalize_pf(row_part, cfg->heads, row_stats, rows, Wa, cfg->ln_eps, nullptr, 0, nullptr, 0, s));
        MQ_TRY(mq_gemm_bf16_lnrs(a, Wa, b.out_wf, Wa, b.out_bf, b.out_sf, row_stats, d_x, d_x, W, rows, W, Wa, rflags, row_part, s));
        if (fold_mlp) MQ_TRY(mq_row_stats_finalize_pf(row_part, (W + 63) / 64, row_stats, rows, W, cfg->ln_eps, pf(b.fc1_wf), (size_t)fc1_cols * W * 2, pf(b.fc2_w), (size_t)W * F * 2, s));
    } else {
        MQ_TRY(attn_bf16(cfg, qf, a, d_cu_seqlens, nseq, fixed_len, max_len, Wa, s));
        if (b.attn_ln_g) MQ_TRY(mq_layernorm_ex(a, 1, nullptr, b.attn_ln_g, b.attn_ln_b, a, nullptr, rows, Wa, cfg->ln_eps, s));   // (in place: a wave holds its row before it stores)
        if (fold_mlp) MQ_TRY(mq_gemm_bf16_rsf(a, Wa, b.out_w, Wa, b.out_b, d_x, d_x, W, rows, W, Wa, rflags, row_part, row_stats, cfg->ln_eps, band_ctr, pf(b.fc1_wf),
                                              (size_t)fc1_cols * W * 2, pf(b.fc2_w), (size_t)W * F * 2, s));
        else MQ_TRY(mq_gemm_bf16(a, Wa, b.out_w, Wa, b.out_b, (const float*)d_x, d_x, W, rows, W, Wa, rflags, s));
    }
    // mlp_glu == 2: (up, gate) rows interleaved 16 by 16 — the tiled GEMM forms up * silu(gate) in its epilogue (MQ_EPI_GLU: the (up | gate) tensor is never
    // written: -206 MB written and -206 MB read per block at EVA02-B/16 x 256), the sub-LayerNorm then normalises the product in place; a call of a few
    // rows (the skinny kernels have no gated epilogue) multiplies behind the GEMM, from the interleaved columns.  Row stride 2 F either way.
    const bool il = cfg->mlp_glu == 2;
    const bool glu_epi = il && !mq_gemm_small_ok(rows, fc1_cols, W, true) && !mq_gemm_small_ok(rows, fc1_cols, W, false) && !mq_gemm_small_grouped_ok(rows, fc1_cols, W);
    const mq_block_weights* nbk = l + 1 < cfg->layers ? &blocks[l + 1] : nullptr;
    const bool fold_next = mq_tower_ln_fold >= 2 && nbk && fold_ok(xb, nbk->qkv_wf, nbk->qkv_bf, nbk->qkv_sf, rows, 3 * Wa, W) && !mq_gemm_small_ok(rows, W, F, false) &&
                           !mq_gemm_small_grouped_ok(rows, W, F);
    const bool fold_mlp_ln = sub_fold && fold_mlp && glu_epi && b.mlp_ln_g && b.fc2_wf && b.fc2_sf && b.fc2_bf && !mq_gemm_small_ok(rows, W, F, false) &&
                             !mq_gemm_small_grouped_ok(rows, W, F);
    if (fold_mlp_ln) {
        // (up | gate) GEMM: norm2 applied, up * silu(gate) stored, the product's row sums left per 32-unit slot -> (mean, rstd) of mlp.norm -> fc2 applies it
        const int Fln = cfg->mlp_ln_dim ? cfg->mlp_ln_dim : F;
        MQ_TRY(mq_gemm_bf16_lnrs(d_x, W, b.fc1_wf, W, b.fc1_bf, b.fc1_sf, row_stats, nullptr, qf, fc1_cols, rows, fc1_cols, W, MQ_EPI_BIAS | MQ_EPI_GLU, row_part, s));
        MQ_TRY(mq_row_stats_finalize_pf(row_part, (fc1_cols + 63) / 64, row_stats, rows, Fln, cfg->ln_eps, pf(b.fc2_wf), (size_t)W * F * 2, nullptr, 0, s));
        MQ_TRY(mq_gemm_bf16_lnrs(qf, fc1_cols, b.fc2_wf, F, b.fc2_bf, b.fc2_sf, row_stats, d_x, d_x, W, rows, W, F, rflags, row_part, s));
        if (fold_next) MQ_TRY(mq_row_stats_finalize_pf(row_part, (W + 63) / 64, row_stats, rows, W, cfg->ln_eps, pf(nbk->qkv_wf), (size_t)3 * Wa * W * 2, pf(nbk->out_w),
                                                       (size_t)W * Wa * 2, s));
        x_has_partials = fold_next;
        return MQ_OK;
    }
    MQ_TRY(ln_gemm(d_x, xb, b.ln2_g, b.ln2_b, cfg->ln_eps, h, b.fc1_w, b.fc1_b, qf, rows, fc1_cols, W, MQ_EPI_BIAS | (cfg->mlp_glu ? (glu_epi ? MQ_EPI_GLU : 0) : act_flag), s,
                   b.fc2_w, (size_t)W * F * 2, b.fc1_wf, b.fc1_bf, b.fc1_sf, row_stats, fold_mlp));
    if (cfg->mlp_glu) {
        const int mode = glu_epi ? 2 : il ? 1 : 0;
        if (b.mlp_ln_g) MQ_TRY(mq_glu_ln(qf, rows, F, cfg->mlp_ln_dim ? cfg->mlp_ln_dim : F, cfg->act, b.mlp_ln_g, b.mlp_ln_b, cfg->ln_eps, s, mode));
        else if (!glu_epi) MQ_TRY(mq_glu(qf, rows, F, cfg->act, s, il ? 1 : 0));
    }
    // fc2 (reads the F-wide product at the (up | gate) buffer's row stride) writes the x the NEXT block's QKV normalises
    if (fold_next) MQ_TRY(mq_gemm_bf16_rsf(qf, fc1_cols, b.fc2_w, F, b.fc2_b, d_x, d_x, W, rows, W, F, rflags, row_part, row_stats, cfg->ln_eps, band_ctr, pf(nbk->qkv_wf),
                                           (size_t)3 * Wa * W * 2, pf(nbk->out_w), (size_t)W * Wa * 2, s));
    else MQ_TRY(mq_gemm_bf16(qf, fc1_cols, b.fc2_w, F, b.fc2_b, (const float*)d_x, d_x, W, rows, W, F, rflags, s));
    x_has_partials = fold_next;
    return MQ_OK;
}

int EncoderPass::block_pre_ln(const mq_block_weights& b, int l) {
    // x += out(attn(qkv(ln1(x)))) ; x += fc2(act(fc1(ln2(x))))   (x fp32, or bf16 in the bf16-stream form)
    const int xb = stream_bf16(cfg) ? 1 : 0;
    const int rflags = xb ? (MQ_EPI_BIAS | MQ_EPI_RESIDUAL) : res_flags;
    bool panel = false;
    MQ_TRY(ln_gemm_panel(panel, d_x, xb, cfg->ln_eps, qf, rows, nseq, fixed_len, d_cu_seqlens, 3 * Wa, W, MQ_EPI_BIAS, s, b.out_w, (size_t)W * Wa * 2, b.qkv_wf, b.qkv_bf,
                         b.qkv_sf, row_stats, x_has_partials));
    if (!panel)
        MQ_TRY(ln_gemm(d_x, xb, b.ln1_g, b.ln1_b, cfg->ln_eps, h, b.qkv_w, b.qkv_b, qf, rows, 3 * Wa, W, MQ_EPI_BIAS, s, b.out_w, (size_t)W * Wa * 2,
                       b.qkv_wf, b.qkv_bf, b.qkv_sf, row_stats, x_has_partials));
    x_has_partials = false;
    // the residual GEMMs leave the rows' partial sums behind whenever the GEMM after them folds its LayerNorm (tiled family, bf16 stream)
    const bool last_pooled = d_sel && nsel > 0 && l == cfg->layers - 1;
    const bool mlp_fp8 = cfg->precision == MQ_PREC_FP8 && l >= first8 - cfg->fp8_mlp_extra;
    const bool fold_mlp = mq_tower_ln_fold >= 2 && !last_pooled && !mlp_fp8 && fold_ok(xb, b.fc1_wf, b.fc1_bf, b.fc1_sf, rows, F, W) &&
                          !mq_gemm_small_ok(rows, W, Wa, false) && !mq_gemm_small_grouped_ok(rows, W, Wa);
    // short fixed-length image sequences (ViT-B/32: 50 tokens): attention, out-projection, residual and the statistics of norm2 in ONE launch, one workgroup
    // per image (attn_proj.hip) — the rows of x carry the same bits as the three launches below leave
    if (mq_tower_attn_proj && fold_mlp && xb && !d_cu_seqlens && fixed_len > 0 && rows == nseq * fixed_len && Wa == W && cfg->mask == MQ_MASK_NONE && !cfg->d_rel_bias &&
        attn_proj_fill_ok(nseq) && mq_attention_proj_ok(nseq, fixed_len, W, cfg->heads)) {
        MQ_TRY(mq_attention_proj(qf, b.out_w, b.out_b, d_x, row_stats, nseq, fixed_len, W, cfg->heads, cfg->ln_eps, pf(b.fc1_wf), (size_t)F * W * 2, pf(b.fc2_w), (size_t)W * F * 2, s));
    } else {
    MQ_TRY(attn_bf16(cfg, qf, a, d_cu_seqlens, nseq, fixed_len, max_len, Wa, s));
    if (fold_mlp) MQ_TRY(mq_gemm_bf16_rsf(a, Wa, b.out_w, Wa, b.out_b, d_x, d_x, W, rows, W, Wa, rflags, row_part, row_stats, cfg->ln_eps, band_ctr, pf(b.fc1_wf),
                                          (size_t)F * W * 2, pf(b.fc2_w), (size_t)W * F * 2, s));
    else MQ_TRY(mq_gemm_bf16(a, Wa, b.out_w, Wa, b.out_b, (const float*)d_x, d_x, W, rows, W, Wa, rflags, s));
    }
    if (mlp_fp8) {
        // MLP-only e4m3 block (fp8_mlp_extra): the attention half above ran on bf16 operands
        MQ_CHECK_ARG(b.fc1_w8 && b.fc1_ws && b.fc2_w8 && b.fc2_ws, "mq_encoder_forward: layer %d has no fp8 MLP weights", l);
        const float* s_mlp = cfg->d_fp8_act_scale + 2 * l + 1;
        float* m_mlp = cfg->d_fp8_act_amax ? cfg->d_fp8_act_amax + 2 * l + 1 : nullptr;
        const int act8 = (cfg->act == MQ_ACT_QUICKGELU ? MQ_EPI_QUICKGELU : MQ_EPI_GELU) | MQ_EPI_BIAS | MQ_EPI_OUT_FP8;
        MQ_TRY(mq_layernorm_fp8_pf(d_x, xb, b.ln2_g, b.ln2_b, h, row_scale, nullptr, rows, W, cfg->ln_eps, pf(b.fc1_w8), (size_t)F * W,
                                   pf(b.fc2_w8), (size_t)W * F, s));
        MQ_TRY(mq_gemm_fp8(h, W, b.fc1_w8, W, row_scale, 1, b.fc1_ws, b.fc1_b, nullptr, qf, F, s_mlp, m_mlp, rows, F, W, act8, s));
        MQ_TRY(mq_gemm_fp8(qf, F, b.fc2_w8, F, s_mlp, 0, b.fc2_ws, b.fc2_b, (const float*)d_x, d_x, W, nullptr, nullptr, rows, W, F, rflags, s));
        return MQ_OK;
    }
    // (the LAST block of a call that reads only pooled rows never folds its MLP: the pooled rows take the small-call kernels — LayerNorm
    // kernel + un-folded weights — and dead-row elimination stays bit-identical to this all-rows form, tests/test_towers_gpu.py)
    panel = false;
    if (!last_pooled)
        MQ_TRY(ln_gemm_panel(panel, d_x, xb, cfg->ln_eps, qf, rows, nseq, fixed_len, d_cu_seqlens, F, W, MQ_EPI_BIAS | act_flag, s, b.fc2_w, (size_t)W * F * 2, b.fc1_wf, b.fc1_bf,
                             b.fc1_sf, row_stats, fold_mlp));
    if (!panel)
        MQ_TRY(ln_gemm(d_x, xb, b.ln2_g, b.ln2_b, cfg->ln_eps, h, b.fc1_w, b.fc1_b, qf, rows, F, W, MQ_EPI_BIAS | act_flag, s, b.fc2_w, (size_t)W * F * 2,
                       last_pooled ? nullptr : b.fc1_wf, b.fc1_bf, b.fc1_sf, row_stats, fold_mlp));
    // fc2 writes the x the NEXT block's QKV normalises
    const mq_block_weights* nbk = l + 1 < cfg->layers && l + 1 < first8 ? &blocks[l + 1] : nullptr;
    const bool fold_next = mq_tower_ln_fold >= 2 && nbk && fold_ok(xb, nbk->qkv_wf, nbk->qkv_bf, nbk->qkv_sf, rows, 3 * Wa, W) && !mq_gemm_small_ok(rows, W, F, false) &&
                           !mq_gemm_small_grouped_ok(rows, W, F);
    if (fold_next) MQ_TRY(mq_gemm_bf16_rsf(qf, F, b.fc2_w, F, b.fc2_b, d_x, d_x, W, rows, W, F, rflags, row_part, row_stats, cfg->ln_eps, band_ctr, pf(nbk->qkv_wf),
                                           (size_t)3 * Wa * W * 2, pf(nbk->out_w), (size_t)W * Wa * 2, s));
    else MQ_TRY(mq_gemm_bf16(qf, F, b.fc2_w, F, b.fc2_b, (const float*)d_x, d_x, W, rows, W, F, rflags, s));
    x_has_partials = fold_next;
    return MQ_OK;
}

int EncoderPass::block_post_ln_small(const mq_block_weights& b, int l) {
    // search path: both LayerNorms ride in the prologue of the GEMM that consumes them (gemm_small.hip).  d_x holds the pre-LN sums
    // t, xn the normalised rows (the residual): t1 = r + out(attn(qkv(r))) ; fc1 normalises t1 -> xn ; t2 = xn + fc2(..) ;
    // the NEXT block's QKV GEMM normalises t2 -> xn (its ln2 belongs to this block); after the last block a plain LayerNorm.
    const float* res = d_x;      // block 0: the embedding LayerNorm's output is the residual, h its bf16 copy
    if (l == 0) MQ_TRY(mq_gemm_bf16(h, W, b.qkv_w, W, b.qkv_b, nullptr, qf, 3 * Wa, rows, 3 * Wa, W, MQ_EPI_BIAS, s));
    else {
        const mq_block_weights& pb = blocks[l - 1];
        MQ_TRY(mq_ln_gemm_small(d_x, W, 0, pb.ln2_g, pb.ln2_b, cfg->ln_eps, b.qkv_w, W, b.qkv_b, qf, 3 * Wa, rows, 3 * Wa, W, MQ_EPI_BIAS, xn, nullptr, s));
        res = xn;
    }
    MQ_TRY(attn_bf16(cfg, qf, a, d_cu_seqlens, nseq, fixed_len, max_len, Wa, s));
    MQ_TRY(mq_gemm_bf16(a, Wa, b.out_w, Wa, b.out_b, res, d_x, W, rows, W, Wa, res_flags, s));
    MQ_TRY(mq_ln_gemm_small(d_x, W, 0, b.ln1_g, b.ln1_b, cfg->ln_eps, b.fc1_w, W, b.fc1_b, qf, F, rows, F, W, MQ_EPI_BIAS | act_flag, xn, nullptr, s));
    MQ_TRY(mq_gemm_bf16(qf, F, b.fc2_w, F, b.fc2_b, xn, d_x, W, rows, W, F, res_flags, s));
    if (l == cfg->layers - 1) MQ_TRY(mq_layernorm(d_x, nullptr, b.ln2_g, b.ln2_b, h, d_x, rows, W, cfg->ln_eps, s));
    return MQ_OK;
}

int EncoderPass::block_post_ln_bf16(const mq_block_weights& b, int l) {
    // h = ln1(h + out(attn(qkv(h)))) ; h = ln2(h + fc2(act(fc1(h)))), all in place on the bf16 rows; the LAST LayerNorm writes fp32 x
    const int rflags = MQ_EPI_BIAS | MQ_EPI_RESIDUAL;
    const bool last = l == cfg->layers - 1;
    MQ_TRY(mq_gemm_bf16(h, W, b.qkv_w, W, b.qkv_b, nullptr, qf, 3 * Wa, rows, 3 * Wa, W, MQ_EPI_BIAS, s));
    MQ_TRY(attn_bf16(cfg, qf, a, d_cu_seqlens, nseq, fixed_len, max_len, Wa, s));
    MQ_TRY(mq_gemm_bf16(a, Wa, b.out_w, Wa, b.out_b, (const float*)h, h, W, rows, W, Wa, rflags, s));
    MQ_TRY(mq_layernorm_pf(h, 1, nullptr, b.ln1_g, b.ln1_b, h, nullptr, rows, W, cfg->ln_eps, pf(b.fc1_w), (size_t)F * W * 2, pf(b.fc2_w),
                           (size_t)W * F * 2, s));
    MQ_TRY(mq_gemm_bf16(h, W, b.fc1_w, W, b.fc1_b, nullptr, qf, F, rows, F, W, MQ_EPI_BIAS | act_flag, s));
    MQ_TRY(mq_gemm_bf16(qf, F, b.fc2_w, F, b.fc2_b, (const float*)h, h, W, rows, W, F, rflags, s));
    const mq_block_weights* nb = !last ? &blocks[l + 1] : nullptr;
    MQ_TRY(mq_layernorm_pf(h, 1, nullptr, b.ln2_g, b.ln2_b, last ? nullptr : h, last ? d_x : nullptr, rows, W, cfg->ln_eps,
                           pf(nb ? nb->qkv_w : nullptr), (size_t)3 * Wa * W * 2, pf(nb ? nb->out_w : nullptr), (size_t)W * Wa * 2, s));
    return MQ_OK;
}

int EncoderPass::block_post_ln(const mq_block_weights& b, int l) {
    // x = ln1(x + out(attn(qkv(x)))) ; x = ln2(x + fc2(act(fc1(x))))
    MQ_TRY(mq_gemm_bf16(h, W, b.qkv_w, W, b.qkv_b, nullptr, qf, 3 * Wa, rows, 3 * Wa, W, MQ_EPI_BIAS, s));
    if (cfg->d_rope_inv_freq) MQ_TRY(mq_rope(qf, d_cu_seqlens, nseq, fixed_len, Wa, cfg->heads, cfg->d_rope_inv_freq, s));
    MQ_TRY(attn_bf16(cfg, qf, a, d_cu_seqlens, nseq, fixed_len, max_len, Wa, s));
    MQ_TRY(mq_gemm_bf16(a, Wa, b.out_w, Wa, b.out_b, d_x, d_x, W, rows, W, Wa, res_flags, s));
    MQ_TRY(mq_layernorm_pf(d_x, 0, nullptr, b.ln1_g, b.ln1_b, h, d_x, rows, W, cfg->ln_eps, pf(b.fc1_w), (size_t)(cfg->mlp_glu ? 2 : 1) * F * W * 2,
                           pf(b.fc2_w), (size_t)W * F * 2, s));
    if (cfg->mlp_glu) {
        // gated MLP: fc1 = (up | gate) rows [2F, W] (bias optional), hidden = up * act(gate) in place, fc2 reads it with lda = 2F
        MQ_TRY(mq_gemm_bf16(h, W, b.fc1_w, W, b.fc1_b, nullptr, qf, 2 * F, rows, 2 * F, W, b.fc1_b ? MQ_EPI_BIAS : 0, s));
        MQ_TRY(mq_glu(qf, rows, F, cfg->act, s));
        MQ_TRY(mq_gemm_bf16(qf, 2 * F, b.fc2_w, F, b.fc2_b, d_x, d_x, W, rows, W, F, res_flags, s));
    } else {
        MQ_TRY(mq_gemm_bf16(h, W, b.fc1_w, W, b.fc1_b, nullptr, qf, F, rows, F, W, MQ_EPI_BIAS | act_flag, s));
        MQ_TRY(mq_gemm_bf16(qf, F, b.fc2_w, F, b.fc2_b, d_x, d_x, W, rows, W, F, res_flags, s));
    }
    const mq_block_weights* nb = l + 1 < cfg->layers && l + 1 < first8 ? &blocks[l + 1] : nullptr;   // the next block's QKV / out-proj weights
    MQ_TRY(mq_layernorm_pf(d_x, 0, nullptr, b.ln2_g, b.ln2_b, h, d_x, rows, W, cfg->ln_eps, pf(nb ? nb->qkv_w : nullptr), (size_t)3 * Wa * W * 2,
                           pf(nb ? nb->out_w : nullptr), (size_t)W * Wa * 2, s));
    return MQ_OK;
}

int encoder_forward_impl(const mq_encoder_cfg* cfg, const mq_block_weights* blocks, float* d_x, int64_t rows,
                         const int32_t* d_cu_seqlens, int64_t nseq, int32_t fixed_len, int32_t max_len,
                         const int32_t* d_sel, int64_t nsel, void* d_workspace, size_t workspace_bytes, hipStream_t s) {
    MQ_TRY(check_encoder_cfg(cfg));
    MQ_CHECK_ARG(d_x && (blocks || cfg->layers == 0), "mq_encoder_forward: null pointer");
    if (rows <= 0 || cfg->layers == 0) return MQ_OK;
    if (workspace_bytes < encoder_ws(cfg, rows)) {
        mq_set_error("mq_encoder_forward: workspace %zu < required %zu", workspace_bytes, encoder_ws(cfg, rows));
        return MQ_ERR_WORKSPACE;
    }
    const int W = cfg->width, F = cfg->mlp_dim, Wa = attn_width(cfg);
    const int fcols = cfg->mlp_glu ? 2 * F : F;
    const size_t big = (size_t)(3 * Wa > fcols ? 3 * Wa : fcols);
    Off cv;
    char* wsb = (char*)d_workspace;
    void* h = wsb + cv.take((size_t)rows * W * 2);
    void* a = wsb + cv.take((size_t)rows * Wa * 2);
    void* qf = wsb + cv.take((size_t)rows * big * 2);  // qkv [rows,3W] then fc1 output [rows,F]
    float* row_scale = (float*)(wsb + cv.take((size_t)rows * 4));
    float* row_stats = (float*)(wsb + cv.take((size_t)rows * 8));
    float* row_part = (float*)(wsb + cv.take((size_t)rows * part_slots(cfg) * 8));
    float* xn = (float*)(wsb + cv.take((size_t)(rows < SMALL_LN_ROWS ? rows : SMALL_LN_ROWS) * W * 4));
    uint32_t* band_ctr = (uint32_t*)(wsb + cv.take((size_t)mq_gemm_band_counters(rows) * 4));
    // (the in-launch finalise is opt-in, mq_tune("rs_finalize", 1): without it the residual GEMMs get no counters and a finalise launch follows them)
    if (stream_bf16(cfg) && mq_gemm_rs_in_launch()) MQ_CHECK_HIP(hipMemsetAsync(band_ctr, 0, (size_t)mq_gemm_band_counters(rows) * 4, s));   // (every launch leaves them zeroed again)
    else band_ctr = nullptr;

    // pooled-rows-only last block: worth it when it at least halves the row count; not during fp8 calibration (the
    // activation maxima must see every row); x_sel must fit behind the fc1 output inside `qf`
    const size_t xsel_off = align_up((size_t)(nsel > 0 ? nsel : 0) * F * 2, WS_ALIGN);
    // (not on the search path either: a call of a few rows is bound by its launch count, and the selection costs 4 launches more)
    const bool select_last = d_sel && nsel > 0 && nsel * 2 <= rows && mq_tower_row_select && !cfg->mlp_glu && !cfg->d_rope_inv_freq && !cfg->d_rope_table &&
                             !mq_gemm_small_ok(rows, W, W, false) &&
                             !(cfg->precision == MQ_PREC_FP8 && (cfg->d_fp8_act_amax || cfg->post_ln)) &&
                             xsel_off + (size_t)nsel * W * 4 <= (size_t)rows * big * 2;

    // mixed precision: blocks [0, first8) on bf16 operands, [first8, layers) on e4m3 (mq_encoder_cfg.fp8_first_layer)
    t_ln_prefetch = weights_outlive_cache(cfg, Wa);
    const int first8 = cfg->precision == MQ_PREC_FP8 ? (cfg->fp8_first_layer < cfg->layers ? cfg->fp8_first_layer : cfg->layers) : cfg->layers;

    // post-LN on the search path: LayerNorms fused into the skinny GEMMs (plain bf16 encoder only)
    const bool small_post_ln = cfg->post_ln && first8 >= cfg->layers && !cfg->mlp_glu && !cfg->d_rope_inv_freq && rows <= SMALL_LN_ROWS &&
                               mq_gemm_small_ok(rows, 3 * Wa, W, true) && mq_gemm_small_ok(rows, F, W, true) && mq_gemm_small_ok(rows, W, F, false);
    // (the skinny-GEMM family — a search query alone or inside a small batch — keeps the fp32 form and its fused-LayerNorm path: the same
    // bits for a query whatever shares its call, like the LayerNorm families of rowops.hip)
    const bool post16 = stream_post16(cfg) && !small_post_ln && !mq_gemm_small_ok(rows, W, W, false);
    // block input as GEMM operand (post-LN: afterwards every LayerNorm leaves it behind)
    if (cfg->post_ln && first8 == 0) MQ_TRY(mq_rowquant_fp8(d_x, h, row_scale, rows, W, s));
    else if (cfg->post_ln) MQ_TRY(mq_cast_bf16(d_x, h, rows * W, s));

    EncoderPass p{cfg, blocks, d_x, rows, d_cu_seqlens, nseq, fixed_len, max_len, d_sel, nsel, s, W, F, Wa, h, a, qf, row_scale, row_stats, row_part, xn, band_ctr,
                  /*x_has_partials*/ false, cfg->act == MQ_ACT_QUICKGELU ? MQ_EPI_QUICKGELU : MQ_EPI_GELU, MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_OUT_F32, first8};
    for (int l = 0; l < cfg->layers; ++l) {
        const mq_block_weights& b = blocks[l];
        const bool f8 = l >= first8;
        MQ_CHECK_ARG(b.qkv_w && b.out_w && b.fc1_w && b.fc2_w && b.ln1_g && b.ln2_g, "mq_encoder_forward: layer %d has null weights", l);
        if (f8)
            MQ_CHECK_ARG(b.qkv_w8 && b.qkv_ws && b.out_w8 && b.out_ws && b.fc1_w8 && b.fc1_ws && b.fc2_w8 && b.fc2_ws,
                         "mq_encoder_forward: layer %d has no fp8 weights", l);
        // post-LN: the previous (bf16) block left its output as a bf16 operand; the first e4m3 block wants e4m3 rows + row scales
        if (cfg->post_ln && f8 && l == first8 && l > 0) MQ_TRY(mq_rowquant_fp8(d_x, h, row_scale, rows, W, s));
        if (select_last && l == cfg->layers - 1) {
            MQ_TRY(last_block_selected(cfg, b, l, d_x, rows, d_cu_seqlens, nseq, fixed_len, max_len, d_sel, nsel, h, a, qf, row_scale, row_stats, p.x_has_partials,
                                       (float*)((char*)qf + xsel_off), f8, s));
            break;
        }
        if (f8) MQ_TRY(p.block_fp8(b, l));
        else if (eva_form(cfg)) MQ_TRY(p.block_eva(b, l));
        else if (!cfg->post_ln) MQ_TRY(p.block_pre_ln(b, l));
        else if (small_post_ln) MQ_TRY(p.block_post_ln_small(b, l));
        else if (post16) MQ_TRY(p.block_post_ln_bf16(b, l));
        else MQ_TRY(p.block_post_ln(b, l));
    }
    return MQ_OK;
}
}  // namespace

extern "C" int mq_encoder_forward(const mq_encoder_cfg* cfg, const mq_block_weights* blocks, float* d_x, int64_t rows,
                                  const int32_t* d_cu_seqlens, int64_t nseq, int32_t fixed_len, int32_t max_len,
                                  void* d_workspace, size_t workspace_bytes, void* stream) {
    return encoder_forward_impl(cfg, blocks, d_x, rows, d_cu_seqlens, nseq, fixed_len, max_len, nullptr, 0, d_workspace, workspace_bytes,
                                (hipStream_t)stream);
}

extern "C" int mq_encoder_forward_rows(const mq_encoder_cfg* cfg, const mq_block_weights* blocks, float* d_x, int64_t rows,
                                       const int32_t* d_cu_seqlens, int64_t nseq, int32_t fixed_len, int32_t max_len,
                                       const int32_t* d_out_rows, int64_t n_out_rows, void* d_workspace, size_t workspace_bytes,
                                       void* stream) {
    MQ_CHECK_ARG(d_out_rows && n_out_rows > 0, "mq_encoder_forward_rows: empty row selection");
    return encoder_forward_impl(cfg, blocks, d_x, rows, d_cu_seqlens, nseq, fixed_len, max_len, d_out_rows, n_out_rows, d_workspace,
                                workspace_bytes, (hipStream_t)stream);
}

// =============================== ViT image tower =================================================
namespace {
struct VitPlan {
    int T, np, Kp; int64_t rows;
    size_t off_x, off_rows, off_cls, off_enc, off_patches, off_patch_out, total;
    size_t off_map_y, off_map_z, off_map_h, off_tok, off_kv;  // MQ_VIT_POOL_MAP (off_tok / off_kv inside the encoder's region)
};
VitPlan vit_plan(const mq_vit_cfg* c, int64_t n) {
    VitPlan p;
    const int G = c->image_size / c->patch_size;
    const bool map = c->pool == MQ_VIT_POOL_MAP || c->pool == MQ_VIT_POOL_QUERY;   // (the query pooler uses the MAP head's buffers; its tokens keep the class token)
    p.np = G * G; p.T = p.np + (c->pool == MQ_VIT_POOL_MAP ? 0 : 1); p.Kp = ceil64(3 * c->patch_size * c->patch_size); p.rows = n * p.T;
    const int W = c->enc.width;
    Off cv;
    p.off_x = cv.take((size_t)p.rows * W * 4);
    p.off_rows = cv.take((size_t)n * 4);
    p.off_cls = cv.take((size_t)n * W * 2);       // CLIP: ln_post(class token) bf16; MAP: pooled attention output bf16
    p.off_map_y = p.off_map_z = p.off_map_h = p.off_tok = p.off_kv = 0;
    if (map) {
        p.off_map_y = cv.take((size_t)n * W * 4);                 // fp32 [n, W] head stream
        p.off_map_z = cv.take((size_t)n * W * 2);                 // bf16 norm(y)
        p.off_map_h = cv.take((size_t)n * c->map_mlp_dim * 2);    // bf16 MLP hidden
    }
    // encoder scratch and the patch buffers are never live together -> they share one region (so do, after the encoder, the
    // MAP head's normalised tokens bf16 [rows, W] and their keys | values bf16 [rows, 2W])
    p.off_enc = cv.end();
    size_t enc = encoder_ws(&c->enc, p.rows);
    if (map) {
        Off mv;
        p.off_tok = p.off_enc + mv.take((size_t)p.rows * W * 2);
        p.off_kv = p.off_enc + mv.take((size_t)p.rows * 2 * W * 2);
        if (mv.end() > enc) enc = mv.end();
    }
    Off pv;
    p.off_patches = p.off_enc + pv.take((size_t)n * p.np * p.Kp * 2);
    p.off_patch_out = p.off_enc + pv.take((size_t)n * p.np * W * 4);
    const size_t patches = pv.end();
    p.total = p.off_enc + (enc > patches ? enc : patches);
    return p;
}

int encode_image_impl(const mq_vit_cfg* cfg, const mq_vit_weights* w, const void* d_pixels, bool is_u8, int64_t n,
                      float* d_out, int normalize, void* ws, size_t ws_bytes, hipStream_t s) {
    MQ_CHECK_ARG(cfg && w && d_out, "mq_encode_image: null pointer");
    MQ_TRY(check_encoder_cfg(&cfg->enc));
    MQ_CHECK_ARG(cfg->patch_size > 0 && cfg->image_size >= cfg->patch_size, "mq_encode_image: image %d smaller than patch %d",
                 cfg->image_size, cfg->patch_size);  // floor(S / P) patches per side, like the strided conv (trailing pixels unread)
    MQ_CHECK_ARG(cfg->out_dim >= 4 && cfg->out_dim % 4 == 0, "mq_encode_image: out_dim %d must be a multiple of 4", cfg->out_dim);
    const bool map = cfg->pool == MQ_VIT_POOL_MAP;
    const bool avg = cfg->pool == MQ_VIT_POOL_AVG;
    const bool qpool = cfg->pool == MQ_VIT_POOL_QUERY;
    MQ_CHECK_ARG(cfg->pool == MQ_VIT_POOL_CLS || map || avg || qpool, "mq_encode_image: bad pool %d", cfg->pool);
    MQ_CHECK_ARG(w->patch_w && w->pos && w->ln_post_g && w->ln_post_b, "mq_encode_image: null weight pointer");
    const int W = cfg->enc.width;
    if (map) {
        const mq_map_head* m = w->map;
        MQ_CHECK_ARG(m && m->q && m->kv_w && m->kv_b && m->proj_w && m->proj_b && m->ln_g && m->ln_b && m->fc1_w && m->fc1_b && m->fc2_w && m->fc2_b,
                     "mq_encode_image: MQ_VIT_POOL_MAP needs the attention-pool head's weights");
        MQ_CHECK_ARG(!w->cls && !w->ln_pre_g && !w->ln_pre_b, "mq_encode_image: MQ_VIT_POOL_MAP has no class token / ln_pre");
        MQ_CHECK_ARG(cfg->map_mlp_dim >= 64 && cfg->map_mlp_dim % 64 == 0, "mq_encode_image: map_mlp_dim %d must be a multiple of 64", cfg->map_mlp_dim);
        MQ_CHECK_ARG(w->proj_w || cfg->out_dim == W, "mq_encode_image: without a projection out_dim (%d) must equal the width (%d)", cfg->out_dim, W);
    } else {
        MQ_CHECK_ARG(w->cls && w->proj_w && (avg || eva_form(&cfg->enc) || (w->ln_pre_g && w->ln_pre_b)) && (!w->ln_pre_g == !w->ln_pre_b),
                     "mq_encode_image: null weight pointer (ln_pre may be absent only with MQ_VIT_POOL_AVG and in the EVA02 form)");
        MQ_CHECK_ARG(!w->proj_b || cfg->pool == MQ_VIT_POOL_CLS, "mq_encode_image: proj_b goes with MQ_VIT_POOL_CLS");
    }
    if (qpool) {
        const mq_map_head* m = w->map;
        MQ_CHECK_ARG(m && m->q && m->kv_w && m->kv_b && m->proj_w && m->proj_b && m->ln_g && m->ln_b, "mq_encode_image: MQ_VIT_POOL_QUERY needs the pooler's weights");
        MQ_CHECK_ARG(cfg->pool_dim >= 64 && cfg->pool_dim % 64 == 0 && cfg->pool_dim <= W && cfg->pool_heads >= 1 && cfg->pool_dim % cfg->pool_heads == 0,
                     "mq_encode_image: pool_dim %d / pool_heads %d unsupported (pool_dim a multiple of 64, <= the width %d)", cfg->pool_dim, cfg->pool_heads, W);
    }
    if (n <= 0) return MQ_OK;
    MQ_CHECK_ARG(d_pixels && ws, "mq_encode_image: null input / workspace");
    const VitPlan p = vit_plan(cfg, n);
    if (ws_bytes < p.total) { mq_set_error("mq_encode_image: workspace %zu < required %zu", ws_bytes, p.total); return MQ_ERR_WORKSPACE; }
    char* base = (char*)ws;
    float* x = (float*)(base + p.off_x);
    int32_t* rows_idx = (int32_t*)(base + p.off_rows);
    void* cls_ln = base + p.off_cls;
    void* patches = base + p.off_patches;
    float* patch_out = (float*)(base + p.off_patch_out);

    // K10 (normalise) + K1: im2col gather, conv-as-GEMM, class token + pos + ln_pre (MAP: + pos, which carries the conv bias)
    MQ_TRY(mq_patchify(d_pixels, is_u8, patches, n, cfg->image_size, cfg->patch_size, p.Kp, cfg->mean, cfg->std, s));
    MQ_TRY(mq_gemm_bf16(patches, p.Kp, w->patch_w, p.Kp, nullptr, nullptr, patch_out, W, n * p.np, W, p.Kp, MQ_EPI_OUT_F32, s));
    const int xb = stream_bf16(&cfg->enc) ? 1 : 0;   // residual stream in bf16 (same buffer, half of it used)
    MQ_TRY(mq_vit_assemble(patch_out, w->cls, w->pos, w->ln_pre_g, w->ln_pre_b, x, n, p.T, W, cfg->enc.ln_eps, s, xb));
    if (map) {
        // K2-K5 x layers on every token, then the trunk's norm on every token and the attention-pool head:
        //   k | v = norm(x) @ kv_w^T + kv_b;  o = softmax(q k^T) v per head (one learned query);  y = o @ proj^T + b;
        //   y = y + fc2(gelu(fc1(LN(y))))  -> [n, W]  (timm AttentionPoolLatent; pooled = y, no further projection)
        const mq_map_head* m = w->map;
        const int F = cfg->map_mlp_dim;
        MQ_TRY(encoder_forward_impl(&cfg->enc, w->blocks, x, p.rows, nullptr, n, p.T, p.T, nullptr, 0, base + p.off_enc,
                                    ws_bytes - p.off_enc, s));
        void* tok = base + p.off_tok;
        void* kv = base + p.off_kv;
        float* y = (float*)(base + p.off_map_y);
        void* z = base + p.off_map_z;
        void* hmid = base + p.off_map_h;
        MQ_TRY(mq_layernorm_ex(x, xb, nullptr, w->ln_post_g, w->ln_post_b, tok, nullptr, p.rows, W, cfg->enc.ln_eps, s));
        MQ_TRY(mq_gemm_bf16(tok, W, m->kv_w, W, m->kv_b, nullptr, kv, 2 * W, p.rows, 2 * W, W, MQ_EPI_BIAS, s));
        MQ_TRY(mq_map_pool(kv, m->q, cls_ln, n, p.T, W, cfg->enc.heads, s));
        MQ_CHECK_HIP(hipMemsetAsync(y, 0, (size_t)n * W * 4, s));   // bias-only epilogue = the residual epilogue over zeros
        MQ_TRY(mq_gemm_bf16(cls_ln, W, m->proj_w, W, m->proj_b, y, y, W, n, W, W, MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_OUT_F32, s));
        MQ_TRY(mq_layernorm(y, nullptr, m->ln_g, m->ln_b, z, nullptr, n, W, cfg->enc.ln_eps, s));
        MQ_TRY(mq_gemm_bf16(z, W, m->fc1_w, W, m->fc1_b, nullptr, hmid, F, n, F, W, MQ_EPI_BIAS | MQ_EPI_GELU, s));
        MQ_TRY(mq_gemm_bf16(hmid, F, m->fc2_w, F, m->fc2_b, y, y, W, n, W, F, MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_OUT_F32, s));
        if (w->proj_w) {
            MQ_TRY(mq_cast_bf16(y, z, (int64_t)n * W, s));
            MQ_TRY(mq_gemm_bf16(z, W, w->proj_w, W, nullptr, nullptr, d_out, cfg->out_dim, n, cfg->out_dim, W, MQ_EPI_OUT_F32, s));
            if (normalize) MQ_TRY(mq_l2_normalize(d_out, d_out, n, cfg->out_dim, s));
        } else if (normalize) {
            MQ_TRY(mq_l2_normalize(y, d_out, n, W, s));
        } else {
            MQ_CHECK_HIP(hipMemcpyAsync(d_out, y, (size_t)n * W * 4, hipMemcpyDeviceToDevice, s));
        }
        return MQ_OK;
    }
    if (qpool) {
        // open_clip VisionTransformer + AttentionalPooler (CoCa): every token (class token included) runs every block;
        //   k | v = ln_k(x) @ kv_w^T + kv_b  (width Dp);  o = softmax(q0 k^T) v per pooler head (q0 = the first learned query, projected at load);
        //   y = o @ out_proj^T + b;  embedding = ln_post(y) @ proj  — the other 255 queries feed CoCa's captioning decoder only
        const mq_map_head* m = w->map;
        const int Dp = cfg->pool_dim;
        MQ_TRY(encoder_forward_impl(&cfg->enc, w->blocks, x, p.rows, nullptr, n, p.T, p.T, nullptr, 0, base + p.off_enc, ws_bytes - p.off_enc, s));
        void* tok = base + p.off_tok;
        void* kv = base + p.off_kv;
        float* y = (float*)(base + p.off_map_y);
        void* z = base + p.off_map_z;
        MQ_TRY(mq_layernorm_ex(x, xb, nullptr, w->ln_post_g, w->ln_post_b, tok, nullptr, p.rows, W, cfg->enc.ln_eps, s));      // ln_k
        MQ_TRY(mq_gemm_bf16(tok, W, m->kv_w, W, m->kv_b, nullptr, kv, 2 * Dp, p.rows, 2 * Dp, W, MQ_EPI_BIAS, s));
        MQ_TRY(mq_map_pool(kv, m->q, cls_ln, n, p.T, Dp, cfg->pool_heads, s));
        MQ_CHECK_HIP(hipMemsetAsync(y, 0, (size_t)n * Dp * 4, s));   // bias-only epilogue = the residual epilogue over zeros
        MQ_TRY(mq_gemm_bf16(cls_ln, Dp, m->proj_w, Dp, m->proj_b, y, y, Dp, n, Dp, Dp, MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_OUT_F32, s));
        MQ_TRY(mq_layernorm(y, nullptr, m->ln_g, m->ln_b, z, nullptr, n, Dp, cfg->enc.ln_eps, s));                                // ln_post
        MQ_TRY(mq_gemm_bf16(z, Dp, w->proj_w, Dp, nullptr, nullptr, d_out, cfg->out_dim, n, cfg->out_dim, Dp, MQ_EPI_OUT_F32, s));
        if (normalize) MQ_TRY(mq_l2_normalize(d_out, d_out, n, cfg->out_dim, s));
        return MQ_OK;
    }
    if (avg) {
        // open_clip VisionTransformer with pool_type 'avg' + final_ln_after_pool (the CLIPA towers): every token runs every block, the PATCH
        // tokens are averaged, ln_post normalises the pooled row, then the projection
        MQ_TRY(encoder_forward_impl(&cfg->enc, w->blocks, x, p.rows, nullptr, n, p.T, p.T, nullptr, 0, base + p.off_enc, ws_bytes - p.off_enc, s));
        float* pooled = (float*)(base + p.off_enc);   // (the encoder's scratch is free again)
        MQ_TRY(mq_avg_tokens(x, xb, pooled, n, p.T, 1, W, s));
        MQ_TRY(mq_layernorm(pooled, nullptr, w->ln_post_g, w->ln_post_b, cls_ln, nullptr, n, W, cfg->enc.ln_eps, s));
        MQ_TRY(mq_gemm_bf16(cls_ln, W, w->proj_w, W, nullptr, nullptr, d_out, cfg->out_dim, n, cfg->out_dim, W, MQ_EPI_OUT_F32, s));
        if (normalize) MQ_TRY(mq_l2_normalize(d_out, d_out, n, cfg->out_dim, s));
        return MQ_OK;
    }
    // K2-K5 x layers (only the class-token rows are read afterwards -> the last block's row-wise half runs on them alone)
    MQ_TRY(mq_cls_rows(rows_idx, n, p.T, s));
    MQ_TRY(encoder_forward_impl(&cfg->enc, w->blocks, x, p.rows, nullptr, n, p.T, p.T, rows_idx, n, base + p.off_enc,
                                ws_bytes - p.off_enc, s));
    // K6: ln_post(class token) @ proj, L2
    if (mq_gemm_small_ok(n, cfg->out_dim, W, true))   // search path: ln_post rides in the head GEMM's prologue
        MQ_TRY(mq_ln_gemm_small(x, W, xb, w->ln_post_g, w->ln_post_b, cfg->enc.ln_eps, w->proj_w, W, w->proj_b, d_out, cfg->out_dim, n, cfg->out_dim,
                                W, MQ_EPI_OUT_F32 | (w->proj_b ? MQ_EPI_BIAS : 0), nullptr, rows_idx, s));
    else {
        MQ_TRY(mq_layernorm_ex(x, xb, rows_idx, w->ln_post_g, w->ln_post_b, cls_ln, nullptr, n, W, cfg->enc.ln_eps, s));
        MQ_TRY(mq_gemm_bf16(cls_ln, W, w->proj_w, W, w->proj_b, nullptr, d_out, cfg->out_dim, n, cfg->out_dim, W, MQ_EPI_OUT_F32 | (w->proj_b ? MQ_EPI_BIAS : 0), s));
    }
    if (normalize) MQ_TRY(mq_l2_normalize(d_out, d_out, n, cfg->out_dim, s));
    return MQ_OK;
}
}  // namespace

extern "C" size_t mq_vit_workspace_bytes(const mq_vit_cfg* cfg, int64_t n_images) {
    if (!cfg || n_images <= 0 || cfg->patch_size <= 0) return 0;
    return vit_plan(cfg, n_images).total;
}

extern "C" int mq_encode_image_u8(const mq_vit_cfg* cfg, const mq_vit_weights* w, const uint8_t* d_pixels, int64_t n,
                                  float* d_out, int normalize, void* d_workspace, size_t workspace_bytes, void* stream) {
    return encode_image_impl(cfg, w, d_pixels, true, n, d_out, normalize, d_workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int mq_encode_image_f32(const mq_vit_cfg* cfg, const mq_vit_weights* w, const float* d_pixels, int64_t n,
                                   float* d_out, int normalize, void* d_workspace, size_t workspace_bytes, void* stream) {
    return encode_image_impl(cfg, w, d_pixels, false, n, d_out, normalize, d_workspace, workspace_bytes, (hipStream_t)stream);
}

// =============================== CLIP text tower ==================================================
namespace {
struct TextPlan { size_t off_x, off_rows, off_pool, off_enc, total; };
TextPlan text_plan(const mq_encoder_cfg* enc, int64_t rows, int64_t nseq) {
    TextPlan p;
    Off cv;
    p.off_x = cv.take((size_t)rows * enc->width * 4);
    p.off_rows = cv.take((size_t)nseq * 4);
    p.off_pool = cv.take((size_t)nseq * enc->width * 4);
    p.off_enc = cv.end();
    p.total = p.off_enc + encoder_ws(enc, rows);
    return p;
}
int max_seq_len(const int32_t* h_cu, int64_t nseq, int64_t* rows_out) {
    int mx = 0;
    for (int64_t i = 0; i < nseq; ++i) {
        const int l = h_cu[i + 1] - h_cu[i];
        if (l < 1) return -1;
        if (l > mx) mx = l;
    }
    *rows_out = h_cu[nseq] - h_cu[0];
    return mx;
}
}  // namespace

extern "C" size_t mq_clip_text_workspace_bytes(const mq_clip_text_cfg* cfg, int64_t rows, int64_t nseq) {
    if (!cfg || rows <= 0 || nseq <= 0) return 0;
    return text_plan(&cfg->enc, rows, nseq).total;
}

extern "C" int mq_encode_clip_text(const mq_clip_text_cfg* cfg, const mq_clip_text_weights* w, const int32_t* d_ids,
                                   const int32_t* d_cu_seqlens, const int32_t* h_cu_seqlens, int64_t nseq,
                                   const int32_t* d_pool_rows, float* d_out, int normalize, void* d_workspace,
                                   size_t workspace_bytes, void* stream) {
    MQ_CHECK_ARG(cfg && w && d_out, "mq_encode_clip_text: null pointer");
    MQ_TRY(check_encoder_cfg(&cfg->enc));
    MQ_CHECK_ARG(cfg->out_dim >= 4 && cfg->out_dim % 4 == 0, "mq_encode_clip_text: out_dim %d must be a multiple of 4", cfg->out_dim);
    MQ_CHECK_ARG(w->tok_emb && w->pos && w->ln_final_g && w->ln_final_b && w->proj_w, "mq_encode_clip_text: null weight pointer");
    if (nseq <= 0) return MQ_OK;
    MQ_CHECK_ARG(d_ids && d_cu_seqlens && h_cu_seqlens && d_workspace, "mq_encode_clip_text: null input / workspace");
    MQ_CHECK_ARG(h_cu_seqlens[0] == 0, "mq_encode_clip_text: cu_seqlens[0] must be 0");
    int64_t rows = 0;
    const int maxl = max_seq_len(h_cu_seqlens, nseq, &rows);
    MQ_CHECK_ARG(maxl >= 1 && maxl <= cfg->ctx + (cfg->cls_pos > 0 ? 1 : 0), "mq_encode_clip_text: sequence lengths must be in [1, ctx=%d]", cfg->ctx);
    MQ_CHECK_ARG(cfg->cls_pos >= 0 && cfg->cls_pos < cfg->ctx, "mq_encode_clip_text: cls_pos %d outside the position table [0, %d)", cfg->cls_pos, cfg->ctx);
    const TextPlan p = text_plan(&cfg->enc, rows, nseq);
    if (workspace_bytes < p.total) { mq_set_error("mq_encode_clip_text: workspace %zu < required %zu", workspace_bytes, p.total); return MQ_ERR_WORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    char* base = (char*)d_workspace;
    float* x = (float*)(base + p.off_x);
    int32_t* rows_idx = (int32_t*)(base + p.off_rows);
    void* pooled = base + p.off_pool;  // bf16 [nseq, W]
    const int W = cfg->enc.width;

    const int xb = stream_bf16(&cfg->enc) ? 1 : 0;   // residual stream in bf16: the embedding kernel writes its bf16 output into x
    MQ_TRY(mq_embed_tokens(d_ids, d_cu_seqlens, nseq, w->tok_emb, w->pos, nullptr, nullptr, nullptr, xb ? nullptr : x, xb ? (void*)x : nullptr, W,
                           cfg->vocab, 0.f, s, cfg->cls_pos));
    const int32_t* pool_rows = d_pool_rows;
    if (!pool_rows) { MQ_TRY(mq_last_rows(d_cu_seqlens, rows_idx, nseq, s)); pool_rows = rows_idx; }
    MQ_TRY(encoder_forward_impl(&cfg->enc, w->blocks, x, rows, d_cu_seqlens, nseq, 0, maxl, pool_rows, nseq, base + p.off_enc,
                                workspace_bytes - p.off_enc, s));
    if (!w->proj_b && mq_gemm_small_ok(nseq, cfg->out_dim, W, true)) {   // search path: ln_final rides in the head GEMM's prologue
        MQ_TRY(mq_ln_gemm_small(x, W, xb, w->ln_final_g, w->ln_final_b, cfg->enc.ln_eps, w->proj_w, W, nullptr, d_out, cfg->out_dim, nseq,
                                cfg->out_dim, W, MQ_EPI_OUT_F32, nullptr, pool_rows, s));
        if (normalize) MQ_TRY(mq_l2_normalize(d_out, d_out, nseq, cfg->out_dim, s));
        return MQ_OK;
    }
    MQ_TRY(mq_layernorm_ex(x, xb, pool_rows, w->ln_final_g, w->ln_final_b, pooled, nullptr, nseq, W, cfg->enc.ln_eps, s));
    if (w->proj_b) {  // SigLIP: Linear with bias = the residual epilogue over zeros
        MQ_CHECK_HIP(hipMemsetAsync(d_out, 0, (size_t)nseq * cfg->out_dim * 4, s));
        MQ_TRY(mq_gemm_bf16(pooled, W, w->proj_w, W, w->proj_b, d_out, d_out, cfg->out_dim, nseq, cfg->out_dim, W,
                            MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_OUT_F32, s));
    } else
        MQ_TRY(mq_gemm_bf16(pooled, W, w->proj_w, W, nullptr, nullptr, d_out, cfg->out_dim, nseq, cfg->out_dim, W, MQ_EPI_OUT_F32, s));
    if (normalize) MQ_TRY(mq_l2_normalize(d_out, d_out, nseq, cfg->out_dim, s));
    return MQ_OK;
}

// =============================== BERT + pooling ====================================================
extern "C" size_t mq_bert_workspace_bytes(const mq_bert_cfg* cfg, int64_t rows, int64_t nseq) {
    if (!cfg || rows <= 0 || nseq <= 0) return 0;
    return text_plan(&cfg->enc, rows, nseq).total;
}

extern "C" int mq_encode_bert(const mq_bert_cfg* cfg, const mq_bert_weights* w, const int32_t* d_ids,
                              const int32_t* d_cu_seqlens, const int32_t* h_cu_seqlens, int64_t nseq, float* d_out,
                              int normalize, void* d_workspace, size_t workspace_bytes, void* stream) {
    MQ_CHECK_ARG(cfg && w && d_out, "mq_encode_bert: null pointer");
    MQ_TRY(check_encoder_cfg(&cfg->enc));
    MQ_CHECK_ARG(cfg->enc.post_ln == 1 && cfg->enc.mask == MQ_MASK_NONE, "mq_encode_bert: BERT is post-LN with full attention");
    MQ_CHECK_ARG(cfg->pool == MQ_POOL_MEAN || cfg->pool == MQ_POOL_CLS, "mq_encode_bert: bad pooling %d", cfg->pool);
    if (cfg->out_dim)
        MQ_CHECK_ARG(cfg->out_dim >= 4 && cfg->out_dim % 4 == 0 && w->proj1_w && w->proj1_b &&
                     (w->proj2_w ? (cfg->proj_hidden >= 64 && cfg->proj_hidden % 64 == 0 && cfg->proj_hidden <= 3 * cfg->enc.width) : cfg->proj_hidden == 0),
                     "mq_encode_bert: projection head needs out_dim %% 4 == 0, proj1_w / proj1_b, and either proj2_w with proj_hidden %% 64 == 0 "
                     "(<= 3 W: the MLP head) or neither (one biased Linear)");
    MQ_CHECK_ARG(w->word_emb && (w->pos_emb || cfg->enc.d_rope_inv_freq) && w->emb_ln_g && w->emb_ln_b, "mq_encode_bert: null weight pointer");
    if (nseq <= 0) return MQ_OK;
    MQ_CHECK_ARG(d_ids && d_cu_seqlens && h_cu_seqlens && d_workspace, "mq_encode_bert: null input / workspace");
    MQ_CHECK_ARG(h_cu_seqlens[0] == 0, "mq_encode_bert: cu_seqlens[0] must be 0");
    int64_t rows = 0;
    const int maxl = max_seq_len(h_cu_seqlens, nseq, &rows);
    MQ_CHECK_ARG(maxl >= 1 && maxl <= cfg->max_pos, "mq_encode_bert: sequence lengths must be in [1, max_pos=%d]", cfg->max_pos);
    const TextPlan p = text_plan(&cfg->enc, rows, nseq);
    if (workspace_bytes < p.total) { mq_set_error("mq_encode_bert: workspace %zu < required %zu", workspace_bytes, p.total); return MQ_ERR_WORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    char* base = (char*)d_workspace;
    float* x = (float*)(base + p.off_x);
    const int W = cfg->enc.width;

    MQ_TRY(mq_embed_tokens(d_ids, d_cu_seqlens, nseq, w->word_emb, w->pos_emb, w->type_emb, w->emb_ln_g, w->emb_ln_b, x, nullptr, W,
                           cfg->vocab, cfg->enc.ln_eps, s));
    // CLS pooling reads only the first row of every sequence (= d_cu_seqlens[0..nseq-1]); mean pooling reads them all
    const int32_t* sel = cfg->pool == MQ_POOL_CLS ? d_cu_seqlens : nullptr;
    MQ_TRY(encoder_forward_impl(&cfg->enc, w->blocks, x, rows, d_cu_seqlens, nseq, 0, maxl, sel, sel ? nseq : 0, base + p.off_enc,
                                workspace_bytes - p.off_enc, s));
    if (!cfg->out_dim) {
        MQ_TRY(mq_pool(x, d_cu_seqlens, nseq, d_out, W, cfg->pool, normalize, s));
        return MQ_OK;
    }
    // projection head of open_clip's HFTextEncoder (pooler -> Linear -> GELU -> Linear, then the caller's L2 normalisation): the
    // pooled rows go through two small GEMMs; their operands live in the encoder's scratch, which is free by now
    float* pooled = (float*)(base + p.off_pool);
    bf16_t* pb = (bf16_t*)(base + p.off_enc);
    bf16_t* h1 = pb + (size_t)nseq * W;
    MQ_TRY(mq_pool(x, d_cu_seqlens, nseq, pooled, W, cfg->pool, 0, s));
    MQ_TRY(mq_cast_bf16(pooled, pb, (int64_t)nseq * W, s));
    if (!w->proj2_w) {   // M-CLIP: one biased Linear(W, out_dim) on the pooled row (multilingual_clip's LinearTransformation)
        MQ_TRY(mq_gemm_bf16(pb, W, w->proj1_w, W, w->proj1_b, nullptr, d_out, cfg->out_dim, nseq, cfg->out_dim, W, MQ_EPI_BIAS | MQ_EPI_OUT_F32, s));
    } else {
        MQ_TRY(mq_gemm_bf16(pb, W, w->proj1_w, W, w->proj1_b, nullptr, h1, cfg->proj_hidden, nseq, cfg->proj_hidden, W, MQ_EPI_BIAS | MQ_EPI_GELU, s));
        MQ_TRY(mq_gemm_bf16(h1, cfg->proj_hidden, w->proj2_w, cfg->proj_hidden, nullptr, nullptr, d_out, cfg->out_dim, nseq, cfg->out_dim,
                            cfg->proj_hidden, MQ_EPI_OUT_F32, s));
    }
    if (normalize) MQ_TRY(mq_l2_normalize(d_out, d_out, nseq, cfg->out_dim, s));
    return MQ_OK;
}
