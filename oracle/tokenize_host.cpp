// ORACLE-SIDE TEST SHIM — test infrastructure only, never loaded by the product.
// Compiles the product's tokenisation algorithms (marqo_amd/csrc/tokenize_algo.h — the very functions the HIP kernels
// instantiate) for the HOST with g++, so that tests/test_gpu_tokenizers.py can pin them without a GPU against the Python
// tokenisers (marqo_amd/engine/tokenizers.py), which tests/test_tokenizers.py pins against `transformers`.
#include <stdint.h>
#include "../marqo_amd/csrc/tokenize_algo.h"

#include <vector>

static inline int64_t norm_base_host(int64_t byte_offset, int64_t t) { return 3 * byte_offset + 16 * t; }

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

extern "C" void tokhost_sentencepiece(const void* slots, const uint8_t* pool, const float* score, const uint32_t* nmap, const uint8_t* npool,
                                      const uint8_t* ccc, uint32_t n_slots, int32_t unk_id, float unk_score, int32_t add_dummy_prefix, int32_t remove_extra_ws,
                                      int32_t max_piece_bytes, int32_t prefix_id, int32_t suffix_id, int32_t pad_id, int32_t id_offset, int32_t unk_out,
                                      const uint8_t* text, const int64_t* offsets, int64_t n, int32_t max_length, int32_t* ids, int64_t ld,
                                      int32_t* lens, int32_t* status, uint8_t* norm_out, int32_t* norm_len) {
    mq_sp_table T;
    T.slots = (const mq_sp_entry*)slots; T.pool = pool; T.score = score; T.nmap = nmap; T.npool = npool; T.ccc = ccc; T.mask = n_slots - 1;
    T.unk_id = unk_id; T.unk_score = unk_score; T.add_dummy_prefix = add_dummy_prefix; T.remove_extra_ws = remove_extra_ws;
    T.max_piece_bytes = max_piece_bytes;
    const mq_sp_frame F{prefix_id, suffix_id, pad_id, id_offset, unk_out};
    for (int64_t t = 0; t < n; ++t) {
        const uint8_t* tx = text + offsets[t];
        const int nb = (int)(offsets[t + 1] - offsets[t]);
        const int64_t cap_n = mq_norm_capacity(nb);
        std::vector<uint8_t> norm(cap_n);
        std::vector<float> best(cap_n + 1);
        std::vector<int32_t> bstart(cap_n + 1), bid(cap_n + 1), pieces(max_length > 0 ? max_length : 1);
        int st;
        const int nl = mq_sp_normalize(T, tx, nb, norm.data(), &st);                                               // A
        int total = 0;
        if (st == MQ_TOK_OK) total = mq_sp_viterbi(T, norm.data(), nl, best.data(), bstart.data(), bid.data(), pieces.data(), max_length);  // B
        const int len = mq_sp_gather(T, F, pieces.data(), st == MQ_TOK_OK ? total : 0, max_length, max_length, ids + t * ld, (int)ld);     // C
        lens[t] = st == MQ_TOK_OK ? len : 0;
        status[t] = st;
        if (norm_out) {  // (tests: the normalised text itself is compared with SentencePieceProcessor.Normalize)
            for (int k = 0; k < nl && st == MQ_TOK_OK; ++k) norm_out[norm_base_host(offsets[t], t) + k] = norm[k];
            norm_len[t] = st == MQ_TOK_OK ? nl : -1;
        }
    }
}
