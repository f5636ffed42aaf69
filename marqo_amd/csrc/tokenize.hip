// K14 — text tokenisation on the device (BASELINE north_star: "text tokenisation staged on-GPU").
// The algorithms are the host+device functions of tokenize_algo.h, in three launches per call: A splits every text into
// word / pre-token spans (one thread per text, no table access), B runs the vocabulary work — greedy WordPiece matching / BPE
// merging, dependent hash-table lookups in L2 — one thread per WORD (a 1024-text batch is ~60 k threads instead of the 1 k
// of a thread-per-text form, which measured 3 ms per batch), C concatenates per text.  Latency-bound byte and integer work:
// per-thread scratch (the current word / its BPE symbols) lives in LDS, lane-strided so that lane t touches bank
// (i*64 + t) and the waves never conflict.  Deliberately not reshaped into anything matrix-like.
#include "common.h"
#include "tokenize_algo.h"

static_assert(sizeof(mq_wp_entry) == 16 && sizeof(mq_bpe_entry) == 16, "hash-table entry layouts (mirrored in engine/gpu_tokenizers.py)");
static_assert(sizeof(mq_wordpiece_vocab) == 56 && sizeof(mq_clip_bpe_vocab) == 48 && sizeof(mq_sentencepiece_vocab) == 96 && sizeof(mq_sp_entry) == 16,
              "tokeniser vocabulary structs (mirrored in _lib.py)");

namespace {

constexpr int TOK_THREADS = 64;

// workspace of one call: spans u64 [n, cap] | totals i32 [n] | counts i16 [n, cap] | norm u8 (normalised text: text i at 3 * offsets[i]
// + 16 i) | pieces (one slot per normalised byte: i32 WordPiece, u16 BPE)
struct TokWs {
    uint64_t* spans; int32_t* totals; int16_t* counts; uint8_t* norm; void* pieces; size_t bytes;
};
__host__ __device__ inline int64_t norm_base(int64_t byte_offset, int64_t t) { return 3 * byte_offset + 16 * t; }
TokWs tok_ws(void* base, int64_t n, int64_t total_bytes, int cap, int piece_size) {
    TokWs w;
    size_t off = 0;
    auto take = [&](size_t b) { const size_t o = off; off = align_up(off + b, 256); return o; };
    const size_t norm_bytes = (size_t)norm_base(total_bytes, n) + 16;
    const size_t o_sp = take((size_t)n * cap * 8), o_tot = take((size_t)n * 4), o_cnt = take((size_t)n * cap * 2), o_nm = take(norm_bytes),
                 o_pc = take(norm_bytes * piece_size);
    w.spans = (uint64_t*)((char*)base + o_sp); w.totals = (int32_t*)((char*)base + o_tot); w.counts = (int16_t*)((char*)base + o_cnt);
    w.norm = (uint8_t*)base + o_nm; w.pieces = (char*)base + o_pc; w.bytes = off;
    return w;
}

// ---- phase A: one thread per text, no table access ---------------------------------------------------------------------
template <bool BPE>
__global__ __launch_bounds__(TOK_THREADS) void split_kernel(mq_uni_table U, const uint8_t* __restrict__ text, const int64_t* __restrict__ offsets, int n,
                                                            int cap, uint64_t* __restrict__ spans, uint8_t* __restrict__ norm,
                                                            int32_t* __restrict__ totals, int32_t* __restrict__ status) {
    const int t = blockIdx.x * TOK_THREADS + threadIdx.x;
    if (t >= n) return;
    const int64_t b0 = offsets[t], b1 = offsets[t + 1];
    int st;
    const int cnt = BPE ? mq_clip_split(U, text + b0, (int)(b1 - b0), cap, spans + (int64_t)t * cap, norm + norm_base(b0, t), &st)
                        : mq_wp_split(U, text + b0, (int)(b1 - b0), cap, spans + (int64_t)t * cap, norm + norm_base(b0, t), &st);
    totals[t] = cnt;
    status[t] = st;
}

// ---- phase B: one thread per word / pre-token (the table lookups); scratch in LDS, lane-strided -------------------------
__global__ __launch_bounds__(TOK_THREADS) void wp_pieces_kernel(mq_wp_table T, const uint8_t* __restrict__ norm, const int64_t* __restrict__ offsets,
                                                                int cap, int blocks_per_text, const uint64_t* __restrict__ spans,
                                                                const int32_t* __restrict__ totals, int16_t* __restrict__ counts,
                                                                int32_t* __restrict__ pieces, int32_t* __restrict__ status) {
    __shared__ uint8_t word[MQ_WP_MAX_WORD * TOK_THREADS];
    const int t = blockIdx.x / blocks_per_text;
    const int j = (blockIdx.x - t * blocks_per_text) * TOK_THREADS + threadIdx.x;
    const int nw = min(totals[t], cap);
    if (j >= nw) return;
    const int64_t nb = norm_base(offsets[t], t);
    const uint64_t span = spans[(int64_t)t * cap + j];
    const int c = mq_wp_pieces(T, norm + nb, span, pieces + nb + mq_span_start(span), word + threadIdx.x, TOK_THREADS);
    if (c < 0) status[t] = MQ_TOK_NEEDS_HOST;   // (benign race: every writer stores the same value)
    counts[(int64_t)t * cap + j] = (int16_t)(c < 0 ? 0 : c);
}

__global__ __launch_bounds__(TOK_THREADS) void bpe_merge_kernel(mq_bpe_table T, const uint8_t* __restrict__ norm, const int64_t* __restrict__ offsets,
                                                                int cap, int blocks_per_text, const uint64_t* __restrict__ spans,
                                                                const int32_t* __restrict__ totals, int16_t* __restrict__ counts,
                                                                uint16_t* __restrict__ syms) {
    __shared__ uint16_t sym[MQ_BPE_MAX_SYMS * TOK_THREADS];
    const int t = blockIdx.x / blocks_per_text;
    const int j = (blockIdx.x - t * blocks_per_text) * TOK_THREADS + threadIdx.x;
    const int nw = min(totals[t], cap);
    if (j >= nw) return;
    const int64_t nb = norm_base(offsets[t], t);
    const uint64_t span = spans[(int64_t)t * cap + j];
    counts[(int64_t)t * cap + j] = (int16_t)mq_clip_merge_span(T, norm + nb, span, syms + nb + mq_span_start(span), sym + threadIdx.x, TOK_THREADS);
}

// ---- phase C: one thread per text, copies only ----------------------------------------------------------------------------
__global__ __launch_bounds__(TOK_THREADS) void wp_gather_kernel(mq_wp_table T, const int64_t* __restrict__ offsets, int n, int cap, int max_tokens,
                                                                const uint64_t* __restrict__ spans, const int32_t* __restrict__ totals,
                                                                const int16_t* __restrict__ counts, const int32_t* __restrict__ pieces,
                                                                int32_t* __restrict__ ids, int64_t ld, int32_t* __restrict__ lens,
                                                                const int32_t* __restrict__ status) {
    const int t = blockIdx.x * TOK_THREADS + threadIdx.x;
    if (t >= n) return;
    const int ok = status[t] == MQ_TOK_OK;
    const int len = mq_wp_gather(T, spans + (int64_t)t * cap, counts + (int64_t)t * cap, pieces + norm_base(offsets[t], t), ok ? min(totals[t], cap) : 0,
                                 max_tokens, ids + (int64_t)t * ld, (int)ld);
    lens[t] = ok ? len : 0;
}

__global__ __launch_bounds__(TOK_THREADS) void bpe_gather_kernel(mq_bpe_table T, const int64_t* __restrict__ offsets, int n, int cap, int ctx,
                                                                 const uint64_t* __restrict__ spans, const int32_t* __restrict__ totals,
                                                                 const int16_t* __restrict__ counts, const uint16_t* __restrict__ syms,
                                                                 int32_t* __restrict__ ids, int32_t* __restrict__ lens,
                                                                 const int32_t* __restrict__ status) {
    const int t = blockIdx.x * TOK_THREADS + threadIdx.x;
    if (t >= n) return;
    const int ok = status[t] == MQ_TOK_OK;
    const int tot = ok ? totals[t] : 0;
    const int len = mq_clip_gather(T, spans + (int64_t)t * cap, counts + (int64_t)t * cap, syms + norm_base(offsets[t], t), min(tot, cap), tot, ctx,
                                   ids + (int64_t)t * ctx);
    lens[t] = ok ? len : 0;
}

// ---- SentencePiece unigram: one thread per text (normalise -> Viterbi -> frame); the search is a dependent chain per text -------------
struct SpWs {
    uint8_t* norm; float* best; int32_t* bstart; int32_t* bid; int32_t* pieces; size_t bytes;
};
SpWs sp_ws(void* base, int64_t n, int64_t total_bytes, int max_length) {
    SpWs w;
    size_t off = 0;
    auto take = [&](size_t b) { const size_t o = off; off = align_up(off + b, 256); return o; };
    const size_t cells = (size_t)norm_base(total_bytes, n) + 16 + (size_t)n;   // (+1 cell per text: positions 0 .. length)
    const size_t o_nm = take(cells), o_b = take(cells * 4), o_s = take(cells * 4), o_i = take(cells * 4), o_p = take((size_t)n * max_length * 4);
    w.norm = (uint8_t*)base + o_nm; w.best = (float*)((char*)base + o_b); w.bstart = (int32_t*)((char*)base + o_s);
    w.bid = (int32_t*)((char*)base + o_i); w.pieces = (int32_t*)((char*)base + o_p); w.bytes = off;
    return w;
}

__global__ __launch_bounds__(TOK_THREADS) void sp_kernel(mq_sp_table T, mq_sp_frame F, const uint8_t* __restrict__ text, const int64_t* __restrict__ offsets,
                                                         int n, int max_length, uint8_t* __restrict__ norm, float* __restrict__ best,
                                                         int32_t* __restrict__ bstart, int32_t* __restrict__ bid, int32_t* __restrict__ pieces,
                                                         int32_t* __restrict__ ids, int64_t ld, int32_t* __restrict__ lens, int32_t* __restrict__ status) {
    const int t = blockIdx.x * TOK_THREADS + threadIdx.x;
    if (t >= n) return;
    const int64_t b0 = offsets[t], b1 = offsets[t + 1];
    const int64_t nb = norm_base(b0, t) + t;
    int st;
    const int nl = mq_sp_normalize(T, text + b0, (int)(b1 - b0), norm + nb, &st);
    int total = 0;
    int32_t* pc = pieces + (int64_t)t * max_length;
    if (st == MQ_TOK_OK) total = mq_sp_viterbi(T, norm + nb, nl, best + nb, bstart + nb, bid + nb, pc, max_length);
    const int len = mq_sp_gather(T, F, pc, st == MQ_TOK_OK ? total : 0, max_length, max_length, ids + (int64_t)t * ld, (int)ld);
    lens[t] = st == MQ_TOK_OK ? len : 0;
    status[t] = st;
}

// packed[cu[s] + j] = padded[s, j] for j < cu[s+1] - cu[s]
__global__ __launch_bounds__(256) void pack_ids_kernel(const int32_t* __restrict__ padded, int64_t ld, const int32_t* __restrict__ cu,
                                                       int32_t* __restrict__ packed) {
    const int s = blockIdx.x;
    const int c0 = cu[s], len = cu[s + 1] - c0;
    for (int j = threadIdx.x; j < len; j += 256) packed[c0 + j] = padded[(int64_t)s * ld + j];
}

bool pow2(uint32_t v) { return v && !(v & (v - 1)); }

}  // namespace

extern "C" size_t mq_tokenize_workspace_bytes(int64_t n, int64_t total_bytes, int32_t cap_tokens) {
    if (n <= 0 || cap_tokens <= 0 || total_bytes < 0) return 0;
    return tok_ws(nullptr, n, total_bytes, cap_tokens, 4).bytes;  // sized for the larger (WordPiece) piece type
}

extern "C" int mq_tokenize_wordpiece(const mq_wordpiece_vocab* v, const uint8_t* d_text, const int64_t* d_offsets, int64_t n,
                                     int64_t total_bytes, int32_t max_length, int32_t* d_ids, int64_t ld, int32_t* d_lens,
                                     int32_t* d_status, void* d_workspace, size_t workspace_bytes, void* stream) {
    MQ_CHECK_ARG(v && v->d_slots && v->d_pool && v->d_unicode && pow2(v->n_slots), "mq_tokenize_wordpiece: bad vocabulary table");
    MQ_CHECK_ARG(v->max_word_chars >= 1 && v->max_word_chars <= MQ_WP_MAX_WORD, "mq_tokenize_wordpiece: max_word_chars %d unsupported",
                 v->max_word_chars);
    MQ_CHECK_ARG(max_length >= 2 && ld >= max_length, "mq_tokenize_wordpiece: need 2 <= max_length (%d) <= ld (%ld)", max_length, (long)ld);
    if (n <= 0) return MQ_OK;
    MQ_CHECK_ARG(d_text && d_offsets && d_ids && d_lens && d_status && d_workspace, "mq_tokenize_wordpiece: null pointer");
    MQ_CHECK_ARG(n < (1LL << 24) && total_bytes >= 0 && total_bytes < (1LL << 40), "mq_tokenize_wordpiece: too many texts / bytes");
    const int max_tokens = max_length - 2;
    const int cap = max_tokens > 0 ? max_tokens : 1;
    const TokWs w = tok_ws(d_workspace, n, total_bytes, cap, 4);
    if (workspace_bytes < w.bytes) { mq_set_error("mq_tokenize_wordpiece: workspace %zu < required %zu", workspace_bytes, w.bytes); return MQ_ERR_WORKSPACE; }
    mq_wp_table T;
    T.slots = (const mq_wp_entry*)v->d_slots; T.pool = v->d_pool; T.mask = v->n_slots - 1;
    T.unk_id = v->unk_id; T.cls_id = v->cls_id; T.sep_id = v->sep_id; T.pad_id = v->pad_id;
    T.lower = v->lower; T.max_word_chars = v->max_word_chars;
    hipStream_t s = (hipStream_t)stream;
    MqProfScope prof(3, s);
    const unsigned per_text = (unsigned)cdiv64(n, TOK_THREADS);
    const int bpt = (cap + TOK_THREADS - 1) / TOK_THREADS;
    const mq_uni_table U{v->d_unicode};
    hipLaunchKernelGGL(split_kernel<false>, dim3(per_text), dim3(TOK_THREADS), 0, s, U, d_text, d_offsets, (int)n, cap, w.spans, w.norm, w.totals,
                       d_status);
    hipLaunchKernelGGL(wp_pieces_kernel, dim3((unsigned)(n * bpt)), dim3(TOK_THREADS), 0, s, T, w.norm, d_offsets, cap, bpt, w.spans, w.totals, w.counts,
                       (int32_t*)w.pieces, d_status);
    hipLaunchKernelGGL(wp_gather_kernel, dim3(per_text), dim3(TOK_THREADS), 0, s, T, d_offsets, (int)n, cap, max_tokens, w.spans, w.totals, w.counts,
                       (const int32_t*)w.pieces, d_ids, ld, d_lens, d_status);
    MQ_CHECK_LAUNCH("mq_tokenize_wordpiece");
    return MQ_OK;
}

extern "C" int mq_tokenize_clip_bpe(const mq_clip_bpe_vocab* v, const uint8_t* d_text, const int64_t* d_offsets, int64_t n,
                                    int64_t total_bytes, int32_t ctx, int32_t* d_ids, int32_t* d_lens, int32_t* d_status,
                                    void* d_workspace, size_t workspace_bytes, void* stream) {
    MQ_CHECK_ARG(v && v->d_slots && v->d_byte_id && v->d_byte_end_id && v->d_unicode && pow2(v->n_slots), "mq_tokenize_clip_bpe: bad merge table");
    MQ_CHECK_ARG(ctx >= 2, "mq_tokenize_clip_bpe: context length %d < 2", ctx);
    if (n <= 0) return MQ_OK;
    MQ_CHECK_ARG(d_text && d_offsets && d_ids && d_lens && d_status && d_workspace, "mq_tokenize_clip_bpe: null pointer");
    MQ_CHECK_ARG(n < (1LL << 24) && total_bytes >= 0 && total_bytes < (1LL << 40), "mq_tokenize_clip_bpe: too many texts / bytes");
    const int cap = ctx;
    const TokWs w = tok_ws(d_workspace, n, total_bytes, cap, 2);
    if (workspace_bytes < w.bytes) { mq_set_error("mq_tokenize_clip_bpe: workspace %zu < required %zu", workspace_bytes, w.bytes); return MQ_ERR_WORKSPACE; }
    mq_bpe_table T;
    T.slots = (const mq_bpe_entry*)v->d_slots; T.byte_id = v->d_byte_id; T.byte_end_id = v->d_byte_end_id; T.mask = v->n_slots - 1;
    T.sot_id = v->sot_id; T.eot_id = v->eot_id; T.lower = v->lower;
    hipStream_t s = (hipStream_t)stream;
    MqProfScope prof(3, s);
    const unsigned per_text = (unsigned)cdiv64(n, TOK_THREADS);
    const int bpt = (cap + TOK_THREADS - 1) / TOK_THREADS;
    const mq_uni_table U{v->d_unicode};
    hipLaunchKernelGGL(split_kernel<true>, dim3(per_text), dim3(TOK_THREADS), 0, s, U, d_text, d_offsets, (int)n, cap, w.spans, w.norm, w.totals,
                       d_status);
    hipLaunchKernelGGL(bpe_merge_kernel, dim3((unsigned)(n * bpt)), dim3(TOK_THREADS), 0, s, T, w.norm, d_offsets, cap, bpt, w.spans, w.totals, w.counts,
                       (uint16_t*)w.pieces);
    hipLaunchKernelGGL(bpe_gather_kernel, dim3(per_text), dim3(TOK_THREADS), 0, s, T, d_offsets, (int)n, cap, ctx, w.spans, w.totals, w.counts,
                       (const uint16_t*)w.pieces, d_ids, d_lens, d_status);
    MQ_CHECK_LAUNCH("mq_tokenize_clip_bpe");
    return MQ_OK;
}

extern "C" size_t mq_tokenize_sentencepiece_workspace_bytes(int64_t n, int64_t total_bytes, int32_t max_length) {
    if (n <= 0 || max_length <= 0 || total_bytes < 0) return 0;
    return sp_ws(nullptr, n, total_bytes, max_length).bytes;
}

extern "C" int mq_tokenize_sentencepiece(const mq_sentencepiece_vocab* v, const uint8_t* d_text, const int64_t* d_offsets, int64_t n,
                                         int64_t total_bytes, int32_t max_length, int32_t* d_ids, int64_t ld, int32_t* d_lens,
                                         int32_t* d_status, void* d_workspace, size_t workspace_bytes, void* stream) {
    MQ_CHECK_ARG(v && v->d_slots && v->d_pool && v->d_score && v->d_nmap && v->d_npool && v->d_ccc && pow2(v->n_slots),
                 "mq_tokenize_sentencepiece: bad vocabulary tables");
    MQ_CHECK_ARG(max_length >= 2 && ld >= max_length && v->max_piece_bytes >= 1, "mq_tokenize_sentencepiece: need 2 <= max_length (%d) <= ld (%ld)",
                 max_length, (long)ld);
    if (n <= 0) return MQ_OK;
    MQ_CHECK_ARG(d_text && d_offsets && d_ids && d_lens && d_status && d_workspace, "mq_tokenize_sentencepiece: null pointer");
    MQ_CHECK_ARG(n < (1LL << 24) && total_bytes >= 0 && total_bytes < (1LL << 28), "mq_tokenize_sentencepiece: too many texts / bytes");
    const SpWs w = sp_ws(d_workspace, n, total_bytes, max_length);
    if (workspace_bytes < w.bytes) { mq_set_error("mq_tokenize_sentencepiece: workspace %zu < required %zu", workspace_bytes, w.bytes); return MQ_ERR_WORKSPACE; }
    mq_sp_table T;
    T.slots = (const mq_sp_entry*)v->d_slots; T.pool = v->d_pool; T.score = v->d_score; T.nmap = v->d_nmap; T.npool = v->d_npool; T.ccc = v->d_ccc;
    T.mask = v->n_slots - 1; T.unk_id = v->unk_id; T.unk_score = v->unk_score; T.add_dummy_prefix = v->add_dummy_prefix;
    T.remove_extra_ws = v->remove_extra_ws; T.max_piece_bytes = v->max_piece_bytes;
    const mq_sp_frame F{v->prefix_id, v->suffix_id, v->pad_id, v->id_offset, v->unk_out};
    hipStream_t s = (hipStream_t)stream;
    MqProfScope prof(3, s);
    hipLaunchKernelGGL(sp_kernel, dim3((unsigned)cdiv64(n, TOK_THREADS)), dim3(TOK_THREADS), 0, s, T, F, d_text, d_offsets, (int)n, max_length, w.norm,
                       w.best, w.bstart, w.bid, w.pieces, d_ids, ld, d_lens, d_status);
    MQ_CHECK_LAUNCH("mq_tokenize_sentencepiece");
    return MQ_OK;
}

extern "C" int mq_pack_ids(const int32_t* d_padded, int64_t ld, const int32_t* d_cu_seqlens, int64_t nseq, int32_t* d_packed,
                           void* stream) {
    if (nseq <= 0) return MQ_OK;
    MQ_CHECK_ARG(d_padded && d_cu_seqlens && d_packed && ld >= 1, "mq_pack_ids: bad argument");
    hipStream_t s = (hipStream_t)stream;
    MqProfScope prof(3, s);
    hipLaunchKernelGGL(pack_ids_kernel, dim3((unsigned)nseq), dim3(256), 0, s, d_padded, ld, d_cu_seqlens, d_packed);
    MQ_CHECK_LAUNCH("mq_pack_ids");
    return MQ_OK;
}
