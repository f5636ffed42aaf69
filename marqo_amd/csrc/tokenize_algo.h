// Text tokenisation (K14) — the per-text algorithms, written once as host+device functions:
//   * tokenize.hip instantiates them with one GPU thread per text (LDS scratch, lane-strided);
//   * oracle/tokenize_host.cpp compiles the SAME functions with g++ so that tests can pin them on the CPU against the
//     Python tokenisers (marqo_amd/engine/tokenizers.py, themselves pinned to `transformers`).
//
// Reference behaviour being reproduced (third-party, un-vendored — SURVEY.md §8c):
//   WordPiece  transformers 4.41.2 BertTokenizer (basic tokenisation + greedy longest-match-first WordPiece), called at
//              src/marqo/core/inference/embedding_models/hugging_face_model.py:179-185 (padding=True, truncation=True)
//   CLIP BPE   open_clip 2.24.0 SimpleTokenizer (lower-case, regex pre-split, byte-level BPE merges by rank), called at
//              src/marqo/core/inference/embedding_models/open_clip_model.py:277
//
// Scope of the device path: texts made only of printable ASCII plus \t \n \r (the host routes every other text — and texts
// that spell a special token or an HTML entity — through the Python tokenisers, exactly as the reference tokenises on the
// host).  Inside that scope the functions are EXACT: every table hit is verified byte for byte.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MQ_TOK_FN __host__ __device__ __forceinline__
#else
#define MQ_TOK_FN static inline
#endif

// status codes written per text
#define MQ_TOK_OK 0
#define MQ_TOK_NEEDS_HOST 1  // byte outside the device scope, or a token longer than the scratch

// ---------------------------------------------------------------------------------------------------------------
// WordPiece vocabulary: open-addressing hash table (linear probing), keyed by FNV-1a-64 of the piece bytes (seeded
// differently for "##" continuation pieces).  A hit is confirmed against the string pool.
// ---------------------------------------------------------------------------------------------------------------
struct mq_wp_entry {
    uint64_t hash;
    int32_t id;        // -1 = empty slot
    uint32_t off_len;  // (pool offset << 8) | (continuation << 7) | length   (length <= 127)
};

struct mq_wp_table {
    const mq_wp_entry* slots;
    const uint8_t* pool;
    uint32_t mask;  // slots - 1 (power of two)
    int32_t unk_id, cls_id, sep_id, pad_id;
    int32_t lower;           // do_lower_case
    int32_t max_word_chars;  // 100
};

#define MQ_FNV_OFFSET 0xcbf29ce484222325ULL
#define MQ_FNV_PRIME 0x100000001b3ULL
#define MQ_WP_CONT_SEED 0x9e3779b97f4a7c15ULL
#define MQ_WP_MAX_WORD 104  // scratch bytes per text for the current word (max_word_chars = 100)

MQ_TOK_FN uint64_t mq_wp_seed(int cont) { return cont ? (MQ_FNV_OFFSET ^ MQ_WP_CONT_SEED) : MQ_FNV_OFFSET; }
MQ_TOK_FN uint64_t mq_wp_step(uint64_t h, uint8_t c) { return (h ^ (uint64_t)c) * MQ_FNV_PRIME; }

MQ_TOK_FN int mq_is_ws(uint8_t c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }
MQ_TOK_FN int mq_is_ascii_punct(uint8_t c) {
    return (c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126);
}
MQ_TOK_FN uint8_t mq_lower(uint8_t c) { return (c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c; }
MQ_TOK_FN int mq_in_scope(uint8_t c) { return (c >= 0x20 && c <= 0x7e) || c == '\t' || c == '\n' || c == '\r'; }

// word bytes live in scratch `w` with element stride `ws` (LDS lane-strided on the device, 1 on the host)
MQ_TOK_FN int32_t mq_wp_lookup(const mq_wp_table& T, uint64_t h, const uint8_t* w, int ws, int start, int len, int cont) {
    uint32_t slot = (uint32_t)(h ^ (h >> 32)) & T.mask;
    for (;;) {
        const mq_wp_entry e = T.slots[slot];
        if (e.id < 0) return -1;
        if (e.hash == h && (int)(e.off_len & 127u) == len && (int)((e.off_len >> 7) & 1u) == cont) {
            const uint8_t* p = T.pool + (e.off_len >> 8);
            int same = 1;
            for (int i = 0; i < len; ++i)
                if (p[i] != w[(start + i) * ws]) { same = 0; break; }
            if (same) return e.id;
        }
        slot = (slot + 1) & T.mask;
    }
}

// Greedy longest-match-first WordPiece of the word w[0..L): appends ids at out[*cnt ...] (only positions < cap are
// written, *cnt always advances); a word with an unmatchable remainder becomes ONE unk token.
MQ_TOK_FN void mq_wp_word(const mq_wp_table& T, const uint8_t* w, int ws, int L, int32_t* out, int os, int cap, int* cnt) {
    const int c0 = *cnt;
    if (L > T.max_word_chars) {
        if (c0 < cap) out[c0 * os] = T.unk_id;
        *cnt = c0 + 1;
        return;
    }
    int start = 0, n = c0;
    while (start < L) {
        uint64_t h = mq_wp_seed(start > 0);
        int best_end = -1;
        int32_t best_id = -1;
        for (int e = start; e < L; ++e) {
            h = mq_wp_step(h, w[e * ws]);
            const int32_t id = mq_wp_lookup(T, h, w, ws, start, e + 1 - start, start > 0);
            if (id >= 0) { best_end = e + 1; best_id = id; }
        }
        if (best_end < 0) {  // whole word -> [UNK]
            if (c0 < cap) out[c0 * os] = T.unk_id;
            *cnt = c0 + 1;
            return;
        }
        if (n < cap) out[n * os] = best_id;
        ++n;
        start = best_end;
    }
    *cnt = n;
}

// One text -> ids (WITHOUT [CLS]/[SEP]); returns the untruncated... no: returns min(count, max_tokens) after the
// reference's truncation ids[:max_tokens]; *status = MQ_TOK_NEEDS_HOST when a byte is outside the device scope.
// Basic tokenisation for in-scope bytes: whitespace splits, every ASCII punctuation char is its own word, the rest are
// words (lower-cased when T.lower); accent stripping / NFC / CJK / control-char removal never fire for these bytes.
MQ_TOK_FN int mq_wordpiece_text(const mq_wp_table& T, const uint8_t* text, int nbytes, int max_tokens, int32_t* out, int os,
                                uint8_t* word, int ws, int* status) {
    int cnt = 0, i = 0;
    *status = MQ_TOK_OK;
    while (i < nbytes && cnt < max_tokens) {  // whole words only: a word is finished before the truncation test
        const uint8_t c = text[i];
        if (!mq_in_scope(c)) { *status = MQ_TOK_NEEDS_HOST; return 0; }
        if (mq_is_ws(c)) { ++i; continue; }
        int L = 0;
        if (mq_is_ascii_punct(c)) {
            word[0] = c;
            L = 1;
            ++i;
        } else {
            int total = 0;  // true word length (may exceed the scratch: only "is it > max_word_chars" matters then)
            while (i < nbytes) {
                const uint8_t d = text[i];
                if (!mq_in_scope(d)) { *status = MQ_TOK_NEEDS_HOST; return 0; }
                if (mq_is_ws(d) || mq_is_ascii_punct(d)) break;
                if (total < MQ_WP_MAX_WORD) word[total * ws] = T.lower ? mq_lower(d) : d;
                ++total;
                ++i;
            }
            L = total;
        }
        mq_wp_word(T, word, ws, L, out, os, max_tokens, &cnt);
    }
    // bytes after the cut-off are not tokenised, but they still decide whether the text is in scope
    for (; i < nbytes; ++i)
        if (!mq_in_scope(text[i])) { *status = MQ_TOK_NEEDS_HOST; return 0; }
    return cnt < max_tokens ? cnt : max_tokens;
}

// ---------------------------------------------------------------------------------------------------------------
// CLIP byte-level BPE.  Symbols are vocabulary ids (< 65536); the merge table maps a pair (a, b) to (rank, merged id).
// ---------------------------------------------------------------------------------------------------------------
struct mq_bpe_entry {
    uint32_t key;     // (a << 16) | b ; 0xffffffff = empty
    uint32_t rank;    // merge priority (lower first)
    uint32_t merged;  // id of the concatenated symbol
    uint32_t pad;
};

struct mq_bpe_table {
    const mq_bpe_entry* slots;
    const uint16_t* byte_id;  // [256] id of the single-byte symbol; the word-final variant ("</w>") is byte_end_id[b]
    const uint16_t* byte_end_id;  // [256]
    uint32_t mask;
    int32_t sot_id, eot_id;
    int32_t lower;
};

#define MQ_BPE_MAX_SYMS 96  // scratch symbols per text (longest pre-token handled on the device)
#define MQ_BPE_EMPTY 0xffffffffu

MQ_TOK_FN int mq_bpe_lookup(const mq_bpe_table& T, uint32_t a, uint32_t b, uint32_t* rank, uint32_t* merged) {
    const uint32_t key = (a << 16) | b;
    uint32_t slot = (key * 0x9e3779b1u) >> 7 & T.mask;
    for (;;) {
        const mq_bpe_entry e = T.slots[slot];
        if (e.key == MQ_BPE_EMPTY) return 0;
        if (e.key == key) { *rank = e.rank; *merged = e.merged; return 1; }
        slot = (slot + 1) & T.mask;
    }
}

// BPE of one pre-token whose byte symbols are already in sym[0..L) (last one the word-final variant): repeatedly merge the
// lowest-ranked adjacent pair, all its occurrences left to right (SimpleTokenizer.bpe).  Returns the new length.
MQ_TOK_FN int mq_bpe_merge(const mq_bpe_table& T, uint16_t* sym, int ss, int L) {
    while (L > 1) {
        uint32_t best_rank = 0xffffffffu, best_merged = 0, ba = 0, bb = 0;
        for (int j = 0; j + 1 < L; ++j) {
            uint32_t r, m;
            const uint32_t a = sym[j * ss], b = sym[(j + 1) * ss];
            if (mq_bpe_lookup(T, a, b, &r, &m) && r < best_rank) { best_rank = r; best_merged = m; ba = a; bb = b; }
        }
        if (best_rank == 0xffffffffu) break;
        int w = 0, j = 0;
        while (j < L) {
            if (j + 1 < L && sym[j * ss] == ba && sym[(j + 1) * ss] == bb) {
                sym[w * ss] = (uint16_t)best_merged;
                j += 2;
            } else {
                sym[w * ss] = sym[j * ss];
                j += 1;
            }
            ++w;
        }
        L = w;
    }
    return L;
}

MQ_TOK_FN int mq_is_alpha(uint8_t c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }
MQ_TOK_FN int mq_is_digit(uint8_t c) { return c >= '0' && c <= '9'; }

// length of the contraction ('s 't 're 've 'm 'll 'd, case-insensitive) starting at text[i] == '\'', or 0
MQ_TOK_FN int mq_contraction(const uint8_t* text, int i, int n) {
    if (i + 1 >= n) return 0;
    const uint8_t a = mq_lower(text[i + 1]);
    if (a == 's' || a == 't' || a == 'm' || a == 'd') return 2;
    if (i + 2 >= n) return 0;
    const uint8_t b = mq_lower(text[i + 2]);
    if ((a == 'r' && b == 'e') || (a == 'v' && b == 'e') || (a == 'l' && b == 'l')) return 3;
    return 0;
}

// One text -> row[0..ctx): SOT, BPE ids, EOT, zero padding; over-long inputs are truncated to ctx and the last kept
// position overwritten with EOT (open_clip tokenize()).  Returns the sequence length including SOT / EOT.
// Pre-tokenisation = the SimpleTokenizer regex restricted to ASCII (the text is lower-cased first when T.lower):
//   contraction | letters+ | one digit | (not whitespace / letter / digit)+      scanned left to right.
MQ_TOK_FN int mq_clip_bpe_text(const mq_bpe_table& T, const uint8_t* text, int nbytes, int ctx, int32_t* row, int rs,
                               uint16_t* sym, int ss, int* status) {
    *status = MQ_TOK_OK;
    int cnt = 1, i = 0;
    row[0] = T.sot_id;
    while (i < nbytes) {
        const uint8_t c = text[i];
        if (!mq_in_scope(c)) { *status = MQ_TOK_NEEDS_HOST; return 0; }
        if (mq_is_ws(c)) { ++i; continue; }
        int len;
        if (c == '\'' && (len = mq_contraction(text, i, nbytes)) > 0) {
        } else if (mq_is_alpha(c)) {
            len = 1;
            while (i + len < nbytes && mq_is_alpha(text[i + len])) ++len;
        } else if (mq_is_digit(c)) {
            len = 1;
        } else {
            len = 1;
            while (i + len < nbytes) {
                const uint8_t d = text[i + len];
                if (!mq_in_scope(d)) { *status = MQ_TOK_NEEDS_HOST; return 0; }
                if (mq_is_ws(d) || mq_is_alpha(d) || mq_is_digit(d)) break;
                ++len;
            }
        }
        if (len > MQ_BPE_MAX_SYMS) { *status = MQ_TOK_NEEDS_HOST; return 0; }
        if (cnt < ctx) {  // tokens past the context window cannot change the kept ones
            for (int j = 0; j < len; ++j) {
                const uint8_t b = T.lower ? mq_lower(text[i + j]) : text[i + j];
                sym[j * ss] = (j == len - 1) ? T.byte_end_id[b] : T.byte_id[b];
            }
            const int L = mq_bpe_merge(T, sym, ss, len);
            for (int j = 0; j < L; ++j) {
                if (cnt < ctx) row[cnt * rs] = (int32_t)sym[j * ss];
                ++cnt;
            }
        } else {
            ++cnt;  // at least one more id: the row is already full
        }
        i += len;
    }
    ++cnt;  // EOT
    if (cnt > ctx) {
        row[(ctx - 1) * rs] = T.eot_id;
        cnt = ctx;
    } else {
        row[(cnt - 1) * rs] = T.eot_id;
        for (int j = cnt; j < ctx; ++j) row[j * rs] = 0;
    }
    return cnt;
}
