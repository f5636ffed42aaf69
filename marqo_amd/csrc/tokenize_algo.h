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

// The work of one text is split into three phases so that the expensive part (hash-table lookups) runs one GPU thread per
// WORD instead of per text:
//   A  mq_wp_split   (per text)  basic tokenisation -> word spans  (start << 8 | min(len, 255)); no table access
//   B  mq_wp_pieces  (per word)  lower-case + greedy WordPiece -> piece ids, stored at the word's own byte positions
//   C  mq_wp_gather  (per text)  concatenate the pieces, truncate, add [CLS] / [SEP], pad
// Basic tokenisation for in-scope bytes: whitespace splits, every ASCII punctuation char is its own word, the rest are words;
// accent stripping / NFC / CJK / control-char removal never fire for these bytes.

// A: returns the number of words found, at most `cap` of them written (a word yields >= 1 id, so words beyond max_tokens can
// never reach the output); *status = MQ_TOK_NEEDS_HOST when a byte is outside the device scope.
MQ_TOK_FN int mq_wp_split(const uint8_t* text, int nbytes, int cap, uint32_t* spans, int* status) {
    int cnt = 0, i = 0;
    *status = MQ_TOK_OK;
    while (i < nbytes) {
        const uint8_t c = text[i];
        if (!mq_in_scope(c)) { *status = MQ_TOK_NEEDS_HOST; return 0; }
        if (mq_is_ws(c)) { ++i; continue; }
        const int start = i;
        if (mq_is_ascii_punct(c)) {
            ++i;
        } else {
            while (i < nbytes) {
                const uint8_t d = text[i];
                if (!mq_in_scope(d)) { *status = MQ_TOK_NEEDS_HOST; return 0; }
                if (mq_is_ws(d) || mq_is_ascii_punct(d)) break;
                ++i;
            }
        }
        const int len = i - start;
        if (cnt < cap) spans[cnt] = ((uint32_t)start << 8) | (uint32_t)(len < 255 ? len : 255);
        ++cnt;
    }
    return cnt;
}

// B: pieces of one word (span from A) -> out[0 .. count) (out has room for one id per byte of the word); returns count.
MQ_TOK_FN int mq_wp_pieces(const mq_wp_table& T, const uint8_t* text, uint32_t span, int32_t* out, uint8_t* word, int ws) {
    const int start = (int)(span >> 8), len = (int)(span & 255u);
    if (len > T.max_word_chars) { out[0] = T.unk_id; return 1; }  // (255 stands for "255 or more")
    for (int j = 0; j < len; ++j) word[j * ws] = T.lower ? mq_lower(text[start + j]) : text[start + j];
    int cnt = 0;
    mq_wp_word(T, word, ws, len, out, 1, len, &cnt);
    return cnt;
}

// C: row = [CLS] pieces... [SEP] pad...; returns the row length.  piece_cnt[j] / pieces at piece_buf[start_j ...] are B's output.
MQ_TOK_FN int mq_wp_gather(const mq_wp_table& T, const uint32_t* spans, const uint8_t* piece_cnt, const int32_t* piece_buf, int nwords,
                           int max_tokens, int32_t* row, int ld) {
    int cnt = 0;
    row[0] = T.cls_id;
    for (int j = 0; j < nwords && cnt < max_tokens; ++j) {
        const int32_t* src = piece_buf + (spans[j] >> 8);
        const int k = piece_cnt[j];
        for (int e = 0; e < k && cnt < max_tokens; ++e) row[1 + cnt++] = src[e];
    }
    row[1 + cnt] = T.sep_id;
    for (int j = cnt + 2; j < ld; ++j) row[j] = T.pad_id;
    return cnt + 2;
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

// Three phases like WordPiece (A per text: regex pre-split into spans; B per pre-token: byte symbols + merges, the surviving
// symbols stored at the pre-token's own byte positions; C per text: SOT, ids, EOT, zero padding / truncation).
// Pre-tokenisation = the SimpleTokenizer regex restricted to ASCII (the text is lower-cased first when T.lower):
//   contraction | letters+ | one digit | (not whitespace / letter / digit)+      scanned left to right.

// A: returns the number of pre-tokens, at most `cap` written as (start << 8 | len); pre-tokens longer than the BPE scratch or
// bytes outside the device scope set *status = MQ_TOK_NEEDS_HOST.
MQ_TOK_FN int mq_clip_split(const uint8_t* text, int nbytes, int cap, uint32_t* spans, int* status) {
    *status = MQ_TOK_OK;
    int cnt = 0, i = 0;
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
        if (cnt < cap) {  // pre-tokens past the context window cannot change the kept ids (each yields >= 1 id)
            if (len > MQ_BPE_MAX_SYMS) { *status = MQ_TOK_NEEDS_HOST; return 0; }
            spans[cnt] = ((uint32_t)i << 8) | (uint32_t)len;
        }
        ++cnt;
        i += len;
    }
    return cnt;
}

// B: BPE of one pre-token -> out[0 .. count) (room for one symbol per byte); returns count.
MQ_TOK_FN int mq_clip_merge_span(const mq_bpe_table& T, const uint8_t* text, uint32_t span, uint16_t* out, uint16_t* sym, int ss) {
    const int start = (int)(span >> 8), len = (int)(span & 255u);
    for (int j = 0; j < len; ++j) {
        const uint8_t b = T.lower ? mq_lower(text[start + j]) : text[start + j];
        sym[j * ss] = (j == len - 1) ? T.byte_end_id[b] : T.byte_id[b];
    }
    const int L = mq_bpe_merge(T, sym, ss, len);
    for (int j = 0; j < L; ++j) out[j] = sym[j * ss];
    return L;
}

// C: row[0..ctx) = SOT ids... EOT 0...; over-long inputs are truncated to ctx and the last kept position overwritten with EOT
// (open_clip tokenize()).  `total` = A's return value (pre-tokens beyond the written ones each stand for >= 1 more id).
MQ_TOK_FN int mq_clip_gather(const mq_bpe_table& T, const uint32_t* spans, const uint8_t* sym_cnt, const uint16_t* sym_buf, int nwritten,
                             int total, int ctx, int32_t* row) {
    int cnt = 1;
    row[0] = T.sot_id;
    for (int j = 0; j < nwritten; ++j) {
        const uint16_t* src = sym_buf + (spans[j] >> 8);
        const int k = sym_cnt[j];
        for (int e = 0; e < k; ++e) {
            if (cnt < ctx) row[cnt] = (int32_t)src[e];
            ++cnt;
        }
    }
    cnt += total - nwritten;  // >= one id each: only "the row overflows" matters
    ++cnt;                    // EOT
    if (cnt > ctx) {
        row[ctx - 1] = T.eot_id;
        cnt = ctx;
    } else {
        row[cnt - 1] = T.eot_id;
        for (int j = cnt; j < ctx; ++j) row[j] = 0;
    }
    return cnt;
}
