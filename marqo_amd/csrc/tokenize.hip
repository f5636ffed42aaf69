// K14 — text tokenisation on the device (BASELINE north_star: "text tokenisation staged on-GPU").
// The algorithms are the host+device functions of tokenize_algo.h; here they run one GPU thread per text (texts are
// independent; a 1024-text batch is 16 wave64s).  This is HBM/latency-bound byte and integer work: per-thread scratch (the
// current word / the BPE symbols of the current pre-token) lives in LDS, lane-strided so that lane t touches bank
// (i*64 + t) and the waves never conflict; vocabulary hash tables (0.5-2 MB) sit in L2.  Deliberately not reshaped into
// anything matrix-like.
#include "common.h"
#include "tokenize_algo.h"

static_assert(sizeof(mq_wp_entry) == 16 && sizeof(mq_bpe_entry) == 16, "hash-table entry layouts (mirrored in engine/gpu_tokenizers.py)");

namespace {

constexpr int TOK_THREADS = 64;

__global__ __launch_bounds__(TOK_THREADS) void wordpiece_kernel(mq_wp_table T, const uint8_t* __restrict__ text,
                                                                const int64_t* __restrict__ offsets, int n, int max_length,
                                                                int32_t* __restrict__ ids, int64_t ld, int32_t* __restrict__ lens,
                                                                int32_t* __restrict__ status) {
    __shared__ uint8_t word[MQ_WP_MAX_WORD * TOK_THREADS];
    const int t = blockIdx.x * TOK_THREADS + threadIdx.x;
    if (t >= n) return;
    const int64_t b0 = offsets[t], b1 = offsets[t + 1];
    int32_t* row = ids + (int64_t)t * ld;
    const int max_tokens = max_length - 2 > 0 ? max_length - 2 : 0;
    int st;
    const int cnt = mq_wordpiece_text(T, text + b0, (int)(b1 - b0), max_tokens, row + 1, 1, word + threadIdx.x, TOK_THREADS, &st);
    row[0] = T.cls_id;
    row[1 + cnt] = T.sep_id;
    for (int j = cnt + 2; j < ld; ++j) row[j] = T.pad_id;
    lens[t] = st == MQ_TOK_OK ? cnt + 2 : 0;
    status[t] = st;
}

__global__ __launch_bounds__(TOK_THREADS) void clip_bpe_kernel(mq_bpe_table T, const uint8_t* __restrict__ text,
                                                               const int64_t* __restrict__ offsets, int n, int ctx,
                                                               int32_t* __restrict__ ids, int32_t* __restrict__ lens,
                                                               int32_t* __restrict__ status) {
    __shared__ uint16_t sym[MQ_BPE_MAX_SYMS * TOK_THREADS];
    const int t = blockIdx.x * TOK_THREADS + threadIdx.x;
    if (t >= n) return;
    const int64_t b0 = offsets[t], b1 = offsets[t + 1];
    int st;
    const int cnt = mq_clip_bpe_text(T, text + b0, (int)(b1 - b0), ctx, ids + (int64_t)t * ctx, 1, sym + threadIdx.x, TOK_THREADS, &st);
    lens[t] = st == MQ_TOK_OK ? cnt : 0;
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

extern "C" int mq_tokenize_wordpiece(const mq_wordpiece_vocab* v, const uint8_t* d_text, const int64_t* d_offsets, int64_t n,
                                     int32_t max_length, int32_t* d_ids, int64_t ld, int32_t* d_lens, int32_t* d_status,
                                     void* stream) {
    MQ_CHECK_ARG(v && v->d_slots && v->d_pool && pow2(v->n_slots), "mq_tokenize_wordpiece: bad vocabulary table");
    MQ_CHECK_ARG(v->max_word_chars >= 1 && v->max_word_chars <= MQ_WP_MAX_WORD - 4, "mq_tokenize_wordpiece: max_word_chars %d unsupported",
                 v->max_word_chars);
    MQ_CHECK_ARG(max_length >= 2 && ld >= max_length, "mq_tokenize_wordpiece: need 2 <= max_length (%d) <= ld (%ld)", max_length, (long)ld);
    if (n <= 0) return MQ_OK;
    MQ_CHECK_ARG(d_text && d_offsets && d_ids && d_lens && d_status, "mq_tokenize_wordpiece: null pointer");
    MQ_CHECK_ARG(n < (1LL << 30), "mq_tokenize_wordpiece: too many texts");
    mq_wp_table T;
    T.slots = (const mq_wp_entry*)v->d_slots; T.pool = v->d_pool; T.mask = v->n_slots - 1;
    T.unk_id = v->unk_id; T.cls_id = v->cls_id; T.sep_id = v->sep_id; T.pad_id = v->pad_id;
    T.lower = v->lower; T.max_word_chars = v->max_word_chars;
    hipStream_t s = (hipStream_t)stream;
    MqProfScope prof(3, s);
    hipLaunchKernelGGL(wordpiece_kernel, dim3((unsigned)cdiv64(n, TOK_THREADS)), dim3(TOK_THREADS), 0, s, T, d_text, d_offsets, (int)n,
                       max_length, d_ids, ld, d_lens, d_status);
    MQ_CHECK_LAUNCH("mq_tokenize_wordpiece");
    return MQ_OK;
}

extern "C" int mq_tokenize_clip_bpe(const mq_clip_bpe_vocab* v, const uint8_t* d_text, const int64_t* d_offsets, int64_t n,
                                    int32_t ctx, int32_t* d_ids, int32_t* d_lens, int32_t* d_status, void* stream) {
    MQ_CHECK_ARG(v && v->d_slots && v->d_byte_id && v->d_byte_end_id && pow2(v->n_slots), "mq_tokenize_clip_bpe: bad merge table");
    MQ_CHECK_ARG(ctx >= 2, "mq_tokenize_clip_bpe: context length %d < 2", ctx);
    if (n <= 0) return MQ_OK;
    MQ_CHECK_ARG(d_text && d_offsets && d_ids && d_lens && d_status, "mq_tokenize_clip_bpe: null pointer");
    MQ_CHECK_ARG(n < (1LL << 30), "mq_tokenize_clip_bpe: too many texts");
    mq_bpe_table T;
    T.slots = (const mq_bpe_entry*)v->d_slots; T.byte_id = v->d_byte_id; T.byte_end_id = v->d_byte_end_id; T.mask = v->n_slots - 1;
    T.sot_id = v->sot_id; T.eot_id = v->eot_id; T.lower = v->lower;
    hipStream_t s = (hipStream_t)stream;
    MqProfScope prof(3, s);
    hipLaunchKernelGGL(clip_bpe_kernel, dim3((unsigned)cdiv64(n, TOK_THREADS)), dim3(TOK_THREADS), 0, s, T, d_text, d_offsets, (int)n, ctx,
                       d_ids, d_lens, d_status);
    MQ_CHECK_LAUNCH("mq_tokenize_clip_bpe");
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
