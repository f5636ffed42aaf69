// ORACLE-SIDE TEST SHIM — test infrastructure only, never loaded by the product.
// Compiles the product's tokenisation algorithms (marqo_amd/csrc/tokenize_algo.h — the very functions the HIP kernels
// instantiate) for the HOST with g++, so that tests/test_gpu_tokenizers.py can pin them without a GPU against the Python
// tokenisers (marqo_amd/engine/tokenizers.py), which tests/test_tokenizers.py pins against `transformers`.
#include <stdint.h>
#include "../marqo_amd/csrc/tokenize_algo.h"

#include <vector>

extern "C" {

void tokhost_wordpiece(const void* slots, const uint8_t* pool, uint32_t n_slots, int32_t unk_id, int32_t cls_id, int32_t sep_id,
                       int32_t pad_id, int32_t lower, int32_t max_word_chars, const uint64_t* unicode, const uint8_t* text, const int64_t* offsets,
                       int64_t n, int32_t max_length, int32_t* ids, int64_t ld, int32_t* lens, int32_t* status) {
    const mq_uni_table U{unicode};
    mq_wp_table T;
    T.slots = (const mq_wp_entry*)slots; T.pool = pool; T.mask = n_slots - 1;
    T.unk_id = unk_id; T.cls_id = cls_id; T.sep_id = sep_id; T.pad_id = pad_id; T.lower = lower; T.max_word_chars = max_word_chars;
    uint8_t word[MQ_WP_MAX_WORD];
    const int max_tokens = max_length - 2 > 0 ? max_length - 2 : 0;
    const int cap = max_tokens > 0 ? max_tokens : 1;
    std::vector<uint64_t> spans(cap);
    std::vector<int16_t> counts(cap);
    for (int64_t t = 0; t < n; ++t) {  // the three phases of tokenize.hip, one text at a time
        const uint8_t* tx = text + offsets[t];
        const int nb = (int)(offsets[t + 1] - offsets[t]);
        std::vector<uint8_t> norm(mq_norm_capacity(nb));
        std::vector<int32_t> pieces(mq_norm_capacity(nb));
        int st;
        const int total = mq_wp_split(U, tx, nb, cap, spans.data(), norm.data(), &st);                             // A
        int nw = st == MQ_TOK_OK ? (total < cap ? total : cap) : 0;
        for (int j = 0; j < nw; ++j) {                                                                              // B
            const int c = mq_wp_pieces(T, norm.data(), spans[j], pieces.data() + mq_span_start(spans[j]), word, 1);
            if (c < 0) { st = MQ_TOK_NEEDS_HOST; nw = 0; break; }
            counts[j] = (int16_t)c;
        }
        const int len = mq_wp_gather(T, spans.data(), counts.data(), pieces.data(), nw, max_tokens, ids + t * ld, (int)ld);        // C
        lens[t] = st == MQ_TOK_OK ? len : 0;
        status[t] = st;
    }
}

void tokhost_clip_bpe(const void* slots, const uint16_t* byte_id, const uint16_t* byte_end_id, uint32_t n_slots, int32_t sot_id,
                      int32_t eot_id, int32_t lower, const uint64_t* unicode, const uint8_t* text, const int64_t* offsets, int64_t n, int32_t ctx,
                      int32_t* ids, int32_t* lens, int32_t* status) {
    const mq_uni_table U{unicode};
    mq_bpe_table T;
    T.slots = (const mq_bpe_entry*)slots; T.byte_id = byte_id; T.byte_end_id = byte_end_id; T.mask = n_slots - 1;
    T.sot_id = sot_id; T.eot_id = eot_id; T.lower = lower;
    uint16_t sym[MQ_BPE_MAX_SYMS];
    const int cap = ctx;
    std::vector<uint64_t> spans(cap);
    std::vector<int16_t> counts(cap);
    for (int64_t t = 0; t < n; ++t) {
        const uint8_t* tx = text + offsets[t];
        const int nb = (int)(offsets[t + 1] - offsets[t]);
        std::vector<uint8_t> norm(mq_norm_capacity(nb));
        std::vector<uint16_t> syms(mq_norm_capacity(nb));
        int st;
        const int total = mq_clip_split(U, tx, nb, cap, spans.data(), norm.data(), &st);                            // A
        const int tot = st == MQ_TOK_OK ? total : 0;
        const int nw = tot < cap ? tot : cap;
        for (int j = 0; j < nw; ++j)                                                                                  // B
            counts[j] = (int16_t)mq_clip_merge_span(T, norm.data(), spans[j], syms.data() + mq_span_start(spans[j]), sym, 1);
        const int len = mq_clip_gather(T, spans.data(), counts.data(), syms.data(), nw, tot, ctx, ids + t * ctx);                      // C
        lens[t] = st == MQ_TOK_OK ? len : 0;
        status[t] = st;
    }
}

}  // extern "C"
