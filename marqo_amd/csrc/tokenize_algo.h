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
// Scope of the device path: any UTF-8 text.  Character handling (whitespace / punctuation / CJK isolation / control removal /
// lower-casing / accent stripping by canonical decomposition / letter and number classes) is table-driven: one 64-bit entry per
// code point below MQ_UNI_LIMIT, built on the host from Python's own `unicodedata` / `str.lower` / `regex` so that it agrees with the
// host tokenisers by construction (engine/gpu_tokenizers.py::build_unicode_table).  The few code points whose treatment depends on
// their NEIGHBOURS (Greek capital sigma's final form, combining marks under a cased vocabulary's NFC step, ...) carry a flag that
// hands the whole text back to the host tokeniser — as do texts that spell a special token or an HTML entity.  Inside that scope
// the functions are EXACT: every vocabulary hit is verified byte for byte.
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
#define MQ_WP_MAX_WORD 256  // scratch bytes per thread for the current word (max_word_chars = 100 characters; longer byte strings go to the host)

MQ_TOK_FN uint64_t mq_wp_seed(int cont) { return cont ? (MQ_FNV_OFFSET ^ MQ_WP_CONT_SEED) : MQ_FNV_OFFSET; }
MQ_TOK_FN uint64_t mq_wp_step(uint64_t h, uint8_t c) { return (h ^ (uint64_t)c) * MQ_FNV_PRIME; }

MQ_TOK_FN uint8_t mq_lower(uint8_t c) { return (c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c; }

// ---------------------------------------------------------------------------------------------------------------
// Unicode character table: entry(cp) for cp < limit
//   bits 0-7   flags
//   bits 8-9   number of output code points (0..2) after the tokeniser's per-character normalisation
//   bits 10-30 first output code point, bits 31-51 second output code point
// ---------------------------------------------------------------------------------------------------------------
#define MQ_UNI_LIMIT 0x30000u
#define MQ_U_DROP 0x01u     // removed from the text (control characters, U+0000, U+FFFD; stripped accents)
#define MQ_U_WS 0x02u       // whitespace: separates words
#define MQ_U_ISOLATE 0x04u  // a word of its own (punctuation, CJK ideographs)
#define MQ_U_HOST 0x08u     // context-dependent treatment: the whole text goes to the host tokeniser
#define MQ_U_LETTER 0x10u   // \p{L}   (CLIP pre-tokenisation)
#define MQ_U_NUMBER 0x20u   // \p{N}
#define MQ_U_HANGUL 0x40u   // precomposed Hangul syllable under an accent-stripping vocabulary: arithmetic decomposition into jamo
#define MQ_U_STRIP 0x80u    // CLIP: removed by Python's str.strip() at the ends of the text although the regex's \\s does not match it
                            // (U+001C..U+001F); regex-whitespace characters are strippable too

struct mq_uni_table {
    const uint64_t* e;
};

MQ_TOK_FN uint32_t mq_u_flags(uint64_t e) { return (uint32_t)(e & 0xffu); }
MQ_TOK_FN int mq_u_nout(uint64_t e) { return (int)((e >> 8) & 3u); }
MQ_TOK_FN uint32_t mq_u_cp0(uint64_t e) { return (uint32_t)((e >> 10) & 0x1fffffu); }
MQ_TOK_FN uint32_t mq_u_cp1(uint64_t e) { return (uint32_t)((e >> 31) & 0x1fffffu); }

// next code point of well-formed UTF-8: returns its length in bytes (1..4), 0 when malformed / truncated
MQ_TOK_FN int mq_utf8_next(const uint8_t* s, int n, int i, uint32_t* cp) {
    const uint8_t c = s[i];
    if (c < 0x80) { *cp = c; return 1; }
    int len;
    uint32_t v;
    if ((c & 0xe0) == 0xc0) { len = 2; v = c & 0x1fu; }
    else if ((c & 0xf0) == 0xe0) { len = 3; v = c & 0x0fu; }
    else if ((c & 0xf8) == 0xf0) { len = 4; v = c & 0x07u; }
    else return 0;
    if (i + len > n) return 0;
    for (int k = 1; k < len; ++k) {
        const uint8_t d = s[i + k];
        if ((d & 0xc0) != 0x80) return 0;
        v = (v << 6) | (d & 0x3fu);
    }
    *cp = v;
    return len;
}

MQ_TOK_FN int mq_utf8_put(uint32_t cp, uint8_t* o) {
    if (cp < 0x80) { o[0] = (uint8_t)cp; return 1; }
    if (cp < 0x800) { o[0] = (uint8_t)(0xc0 | (cp >> 6)); o[1] = (uint8_t)(0x80 | (cp & 0x3f)); return 2; }
    if (cp < 0x10000) { o[0] = (uint8_t)(0xe0 | (cp >> 12)); o[1] = (uint8_t)(0x80 | ((cp >> 6) & 0x3f)); o[2] = (uint8_t)(0x80 | (cp & 0x3f)); return 3; }
    o[0] = (uint8_t)(0xf0 | (cp >> 18)); o[1] = (uint8_t)(0x80 | ((cp >> 12) & 0x3f)); o[2] = (uint8_t)(0x80 | ((cp >> 6) & 0x3f));
    o[3] = (uint8_t)(0x80 | (cp & 0x3f));
    return 4;
}

// word / pre-token span in the NORMALISED text of one input text: (first byte << 32) | (bytes << 16) | characters
MQ_TOK_FN uint64_t mq_span(int start, int nbytes, int nchars) {
    return ((uint64_t)(uint32_t)start << 32) | ((uint64_t)(nbytes < 0xffff ? nbytes : 0xffff) << 16) | (uint64_t)(nchars < 0xffff ? nchars : 0xffff);
}
MQ_TOK_FN int mq_span_start(uint64_t s) { return (int)(s >> 32); }
MQ_TOK_FN int mq_span_bytes(uint64_t s) { return (int)((s >> 16) & 0xffffu); }
MQ_TOK_FN int mq_span_chars(uint64_t s) { return (int)(s & 0xffffu); }
// normalised bytes one input text of nbytes can produce: a 3-byte Hangul syllable becomes three 3-byte jamo, a 2-byte letter may
// lower-case / decompose into two 3-byte code points
MQ_TOK_FN int64_t mq_norm_capacity(int64_t nbytes) { return 3 * nbytes + 16; }

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

// Greedy longest-match-first WordPiece of the word w[0..L) (UTF-8 bytes, `nchars` characters): appends ids at out[*cnt ...] (only
// positions < cap are written, *cnt always advances); a word with an unmatchable remainder becomes ONE unk token.  Pieces start and
// end on character boundaries (the host matches substrings of the character string).
MQ_TOK_FN void mq_wp_word(const mq_wp_table& T, const uint8_t* w, int ws, int L, int nchars, int32_t* out, int os, int cap, int* cnt) {
    const int c0 = *cnt;
    if (nchars > T.max_word_chars) {
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
            if (e + 1 < L && (w[(e + 1) * ws] & 0xc0) == 0x80) continue;   // inside a character
            if (e + 1 - start > 127) break;                                   // no vocabulary piece is that long
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
//   A  mq_wp_split   (per text)  basic tokenisation: decode UTF-8, per-character table (drop / whitespace / isolate / lower-case +
//                                accent strip), write the NORMALISED bytes and the word spans; no vocabulary access
//   B  mq_wp_pieces  (per word)  greedy WordPiece -> piece ids, stored at the word's own (normalised) byte positions
//   C  mq_wp_gather  (per text)  concatenate the pieces, truncate, add [CLS] / [SEP], pad

// A: returns the number of words found, at most `cap` of them written (a word yields >= 1 id, so words beyond max_tokens can
// never reach the output); *status = MQ_TOK_NEEDS_HOST when the text needs the host tokeniser.  norm has mq_norm_capacity(nbytes) bytes.
MQ_TOK_FN int mq_wp_split(const mq_uni_table& U, const uint8_t* text, int nbytes, int cap, uint64_t* spans, uint8_t* norm, int* status) {
    int cnt = 0, i = 0, pos = 0, wstart = -1, wchars = 0;
    *status = MQ_TOK_OK;
    while (i < nbytes) {
        uint32_t cp;
        const int len = mq_utf8_next(text, nbytes, i, &cp);
        if (len == 0 || cp >= MQ_UNI_LIMIT) { *status = MQ_TOK_NEEDS_HOST; return 0; }
        i += len;
        const uint64_t e = U.e[cp];
        const uint32_t f = mq_u_flags(e);
        if (f & MQ_U_HOST) { *status = MQ_TOK_NEEDS_HOST; return 0; }
        if (f & MQ_U_DROP) continue;
        if (f & (MQ_U_WS | MQ_U_ISOLATE)) {
            if (wstart >= 0) {
                if (cnt < cap) spans[cnt] = mq_span(wstart, pos - wstart, wchars);
                ++cnt;
                wstart = -1;
            }
            if (f & MQ_U_ISOLATE) {
                const int s0 = pos;
                pos += mq_utf8_put(mq_u_cp0(e), norm + pos);
                if (cnt < cap) spans[cnt] = mq_span(s0, pos - s0, 1);
                ++cnt;
            }
            continue;
        }
        if (wstart < 0) { wstart = pos; wchars = 0; }
        if (f & MQ_U_HANGUL) {  // NFD of a precomposed syllable: L V (T) conjoining jamo
            const uint32_t sidx = cp - 0xac00u, tj = sidx % 28u;
            pos += mq_utf8_put(0x1100u + sidx / 588u, norm + pos);
            pos += mq_utf8_put(0x1161u + (sidx % 588u) / 28u, norm + pos);
            wchars += 2;
            if (tj) { pos += mq_utf8_put(0x11a7u + tj, norm + pos); ++wchars; }
            continue;
        }
        const int no = mq_u_nout(e);
        if (no >= 1) { pos += mq_utf8_put(mq_u_cp0(e), norm + pos); ++wchars; }
        if (no >= 2) { pos += mq_utf8_put(mq_u_cp1(e), norm + pos); ++wchars; }
    }
    if (wstart >= 0) {
        if (cnt < cap) spans[cnt] = mq_span(wstart, pos - wstart, wchars);
        ++cnt;
    }
    return cnt;
}

// B: pieces of one word (span from A, bytes in `norm`) -> out[0 .. count) (out has room for one id per byte of the word); returns
// count, or -1 when the word does not fit the per-thread scratch (the text then goes to the host)
MQ_TOK_FN int mq_wp_pieces(const mq_wp_table& T, const uint8_t* norm, uint64_t span, int32_t* out, uint8_t* word, int ws) {
    const int start = mq_span_start(span), len = mq_span_bytes(span), nchars = mq_span_chars(span);
    if (nchars > T.max_word_chars) { out[0] = T.unk_id; return 1; }
    if (len > MQ_WP_MAX_WORD) return -1;
    for (int j = 0; j < len; ++j) word[j * ws] = norm[start + j];
    int cnt = 0;
    mq_wp_word(T, word, ws, len, nchars, out, 1, len, &cnt);
    return cnt;
}

// C: row = [CLS] pieces... [SEP] pad...; returns the row length.  piece_cnt[j] / pieces at piece_buf[start_j ...] are B's output.
MQ_TOK_FN int mq_wp_gather(const mq_wp_table& T, const uint64_t* spans, const int16_t* piece_cnt, const int32_t* piece_buf, int nwords,
                           int max_tokens, int32_t* row, int ld) {
    int cnt = 0;
    row[0] = T.cls_id;
    for (int j = 0; j < nwords && cnt < max_tokens; ++j) {
        const int32_t* src = piece_buf + mq_span_start(spans[j]);
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

#define MQ_BPE_MAX_SYMS 128  // scratch symbols per thread (longest pre-token, in UTF-8 bytes, handled on the device)
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

// Three phases like WordPiece (A per text: lower-case + regex pre-split into spans of the normalised text; B per pre-token: byte
// symbols + merges, the surviving symbols stored at the pre-token's own byte positions; C per text: SOT, ids, EOT, zero padding /
// truncation).  Pre-tokenisation = the SimpleTokenizer regex on the lower-cased text, scanned left to right:
//   's 't 're 've 'm 'll 'd | \p{L}+ | \p{N} | [^\s\p{L}\p{N}]+
// with \s, \p{L}, \p{N} taken from the character table (built from the `regex` module the host tokeniser runs).

// class of an OUTPUT character: 0 whitespace, 1 letter, 2 number, 3 other
MQ_TOK_FN int mq_clip_class(uint32_t f) { return (f & MQ_U_WS) ? 0 : (f & MQ_U_LETTER) ? 1 : (f & MQ_U_NUMBER) ? 2 : 3; }

// A: returns the number of pre-tokens, at most `cap` written; pre-tokens longer than the BPE scratch or characters that need the
// host set *status = MQ_TOK_NEEDS_HOST.  norm has mq_norm_capacity(nbytes) bytes; lower-cased characters are written there.
MQ_TOK_FN int mq_clip_split(const mq_uni_table& U, const uint8_t* text, int nbytes, int cap, uint64_t* spans, uint8_t* norm, int* status) {
    *status = MQ_TOK_OK;
    int cnt = 0, i = 0, pos = 0;
    int tstart = -1, tclass = -1, tchars = 0;   // open pre-token (class 1 = letters, 3 = other run)
    // the host cleans the text with str.strip() first: characters Python calls whitespace vanish at both ENDS even where the regex
    // would treat them as ordinary symbols (U+001C..U+001F)
    {
        int k = 0, first = -1, last_end = 0;
        while (k < nbytes) {
            uint32_t cp;
            const int len = mq_utf8_next(text, nbytes, k, &cp);
            if (len == 0 || cp >= MQ_UNI_LIMIT) { *status = MQ_TOK_NEEDS_HOST; return 0; }
            if (!(mq_u_flags(U.e[cp]) & (MQ_U_STRIP | MQ_U_WS))) {
                if (first < 0) first = k;
                last_end = k + len;
            }
            k += len;
        }
        if (first < 0) return 0;
        i = first;
        nbytes = last_end;
    }
    while (i < nbytes) {
        uint32_t cp;
        const int len = mq_utf8_next(text, nbytes, i, &cp);
        if (len == 0 || cp >= MQ_UNI_LIMIT) { *status = MQ_TOK_NEEDS_HOST; return 0; }
        const uint64_t e = U.e[cp];
        const uint32_t f = mq_u_flags(e);
        if (f & MQ_U_HOST) { *status = MQ_TOK_NEEDS_HOST; return 0; }
        const uint32_t oc = mq_u_cp0(e);        // lower-cased character (exactly one: multi-character lowerings are HOST)
        const int cls = mq_clip_class(f);
        // a contraction is tried wherever a pre-token could START with an apostrophe: at the beginning of a match, i.e. when no
        // letters / "other" run is open, or right after a letters run (the regex then closes the run and starts a new match)
        if (oc == '\'' && (tstart < 0 || tclass == 1)) {
            // look ahead in the RAW text: the next one or two characters, lower-cased through the table (a character that needs the
            // host is met by the main loop right after and sends the text there)
            uint32_t a = 0, b = 0;
            int la = 0, lb = 0;
            if (i + len < nbytes) {
                uint32_t c1;
                la = mq_utf8_next(text, nbytes, i + len, &c1);
                if (la && c1 < MQ_UNI_LIMIT) a = mq_u_cp0(U.e[c1]);
                if (la && i + len + la < nbytes) {
                    uint32_t c2;
                    lb = mq_utf8_next(text, nbytes, i + len + la, &c2);
                    if (lb && c2 < MQ_UNI_LIMIT) b = mq_u_cp0(U.e[c2]);
                }
            }
            int clen = 0;   // characters after the apostrophe that belong to the contraction (the regex is IGNORECASE: a cased
                            // vocabulary keeps 'S as typed)
            const uint32_t al = (a >= 'A' && a <= 'Z') ? a + 32 : a, bl = (b >= 'A' && b <= 'Z') ? b + 32 : b;
            if (al == 's' || al == 't' || al == 'm' || al == 'd') clen = 1;
            else if ((al == 'r' && bl == 'e') || (al == 'v' && bl == 'e') || (al == 'l' && bl == 'l')) clen = 2;
            if (clen) {
                if (tstart >= 0) {
                    if (cnt < cap) { if (pos - tstart > MQ_BPE_MAX_SYMS) { *status = MQ_TOK_NEEDS_HOST; return 0; } spans[cnt] = mq_span(tstart, pos - tstart, tchars); }
                    ++cnt;
                    tstart = -1;
                }
                const int s0 = pos;
                norm[pos++] = '\'';
                norm[pos++] = (uint8_t)a;
                if (clen == 2) norm[pos++] = (uint8_t)b;
                if (cnt < cap) spans[cnt] = mq_span(s0, pos - s0, 1 + clen);
                ++cnt;
                i += len + la + (clen == 2 ? lb : 0);
                continue;
            }
        }
        i += len;
        if (cls == 0) {
            if (tstart >= 0) {
                if (cnt < cap) { if (pos - tstart > MQ_BPE_MAX_SYMS) { *status = MQ_TOK_NEEDS_HOST; return 0; } spans[cnt] = mq_span(tstart, pos - tstart, tchars); }
                ++cnt;
                tstart = -1;
            }
            continue;
        }
        if (tstart >= 0 && (cls != tclass || cls == 2)) {   // class change closes the run; every number is a pre-token of its own
            if (cnt < cap) { if (pos - tstart > MQ_BPE_MAX_SYMS) { *status = MQ_TOK_NEEDS_HOST; return 0; } spans[cnt] = mq_span(tstart, pos - tstart, tchars); }
            ++cnt;
            tstart = -1;
        }
        if (tstart < 0) { tstart = pos; tclass = cls; tchars = 0; }
        pos += mq_utf8_put(oc, norm + pos);
        ++tchars;
        if (cls == 2) {   // single number
            if (cnt < cap) spans[cnt] = mq_span(tstart, pos - tstart, 1);
            ++cnt;
            tstart = -1;
        }
    }
    if (tstart >= 0) {
        if (cnt < cap) { if (pos - tstart > MQ_BPE_MAX_SYMS) { *status = MQ_TOK_NEEDS_HOST; return 0; } spans[cnt] = mq_span(tstart, pos - tstart, tchars); }
        ++cnt;
    }
    return cnt;
}

// B: BPE of one pre-token (bytes in `norm`) -> out[0 .. count) (room for one symbol per byte); returns count.
MQ_TOK_FN int mq_clip_merge_span(const mq_bpe_table& T, const uint8_t* norm, uint64_t span, uint16_t* out, uint16_t* sym, int ss) {
    const int start = mq_span_start(span), len = mq_span_bytes(span);
    for (int j = 0; j < len; ++j) {
        const uint8_t b = norm[start + j];
        sym[j * ss] = (j == len - 1) ? T.byte_end_id[b] : T.byte_id[b];
    }
    const int L = mq_bpe_merge(T, sym, ss, len);
    for (int j = 0; j < L; ++j) out[j] = sym[j * ss];
    return L;
}

// C: row[0..ctx) = SOT ids... EOT 0...; over-long inputs are truncated to ctx and the last kept position overwritten with EOT
// (open_clip tokenize()).  `total` = A's return value (pre-tokens beyond the written ones each stand for >= 1 more id).
MQ_TOK_FN int mq_clip_gather(const mq_bpe_table& T, const uint64_t* spans, const int16_t* sym_cnt, const uint16_t* sym_buf, int nwritten,
                             int total, int ctx, int32_t* row) {
    int cnt = 1;
    row[0] = T.sot_id;
    for (int j = 0; j < nwritten; ++j) {
        const uint16_t* src = sym_buf + mq_span_start(spans[j]);
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

// ---------------------------------------------------------------------------------------------------------------
// SentencePiece unigram (XLM-RoBERTa: the multilingual-e5 family; T5-style vocabularies: SigLIP).  Reference behaviour being
// reproduced (third-party, un-vendored): sentencepiece's normaliser (precompiled character map + whitespace rules) followed by the
// unigram model's EncodeOptimized Viterbi search, as called through transformers at hugging_face_model.py:179-185.
//   A  mq_sp_normalize  per-character normalisation through a table built on the host from SentencePieceProcessor.Normalize itself
//                       (so it IS the model's own character map), then the normaliser's whitespace rules: leading / trailing /
//                       repeated spaces removed, the remaining ones escaped to U+2581, a dummy U+2581 prefix
//   B  mq_sp_viterbi    best segmentation under the piece scores: for every start (character boundary, left to right) every piece
//                       that begins there proposes `best[start] + score` to its end position, strictly-greater wins (first proposal
//                       wins ties, as in sentencepiece); a start with no one-character piece proposes <unk> with min_score - 10.
//                       Pieces are found by extending a prefix through a hash table that holds every piece AND every proper prefix of
//                       a piece (trie semantics: the extension stops at the first absent prefix).  Consecutive <unk> are merged.
// Characters the per-character table cannot express (combining marks and conjoining jamo, which the NFKC-based map composes with their
// neighbour; mappings that expand more than 3x) flag the text for the host tokeniser.
// ---------------------------------------------------------------------------------------------------------------
struct mq_sp_entry {
    uint64_t hash;
    int32_t id;        // >= 0: piece id; -2: proper prefix of some piece only; -1: empty slot
    uint32_t off_len;  // (pool offset << 8) | length in bytes (<= 255)
};

struct mq_sp_table {
    const mq_sp_entry* slots;
    const uint8_t* pool;
    const float* score;      // [vocab] (used for id >= 0)
    const uint32_t* nmap;    // [MQ_UNI_LIMIT] per code point: (pool2 offset << 8) | normalised byte length; 0xff = needs host; 0 = removed
    const uint8_t* npool;    // normalised strings
    const uint8_t* ccc;      // [MQ_UNI_LIMIT] canonical combining class (two adjacent marks in descending class order would be reordered by
                             // the NFKC-based map: such a text goes to the host)
    uint32_t mask;
    int32_t unk_id;          // SentencePiece's own id of <unk>
    float unk_score;         // min piece score - 10
    int32_t add_dummy_prefix, remove_extra_ws;
    int32_t max_piece_bytes;
};

#define MQ_SP_HOST 0xffu

// A: -> normalised length in bytes (norm has mq_norm_capacity(nbytes)); *status = MQ_TOK_NEEDS_HOST for flagged characters
MQ_TOK_FN int mq_sp_normalize(const mq_sp_table& T, const uint8_t* text, int nbytes, uint8_t* norm, int* status) {
    *status = MQ_TOK_OK;
    int i = 0, pos = 0;
    int started = 0;       // a non-space character has been emitted
    int pending_space = 0; // a (collapsed) space waits for the next non-space character
    int prev_ccc = 0;
    while (i < nbytes) {
        uint32_t cp;
        const int len = mq_utf8_next(text, nbytes, i, &cp);
        if (len == 0 || cp >= MQ_UNI_LIMIT) { *status = MQ_TOK_NEEDS_HOST; return 0; }
        i += len;
        const int cc = T.ccc[cp];
        if (cc != 0 && prev_ccc > cc) { *status = MQ_TOK_NEEDS_HOST; return 0; }   // canonical reordering
        prev_ccc = cc;
        const uint32_t m = T.nmap[cp];
        const int nl = (int)(m & 0xffu);
        if (nl == (int)MQ_SP_HOST) { *status = MQ_TOK_NEEDS_HOST; return 0; }
        const uint8_t* src = T.npool + (m >> 8);
        for (int k = 0; k < nl; ++k) {
            const uint8_t b = src[k];
            if (b == ' ') {
                if (T.remove_extra_ws) { if (started) pending_space = 1; }       // leading spaces vanish, runs collapse, trailing ones never flush
                else { norm[pos++] = 0xe2; norm[pos++] = 0x96; norm[pos++] = 0x81; }
                continue;
            }
            if (!started) {
                started = 1;
                if (T.add_dummy_prefix) { norm[pos++] = 0xe2; norm[pos++] = 0x96; norm[pos++] = 0x81; }
            }
            if (pending_space) { norm[pos++] = 0xe2; norm[pos++] = 0x96; norm[pos++] = 0x81; pending_space = 0; }
            norm[pos++] = b;
        }
    }
    return pos;
}

MQ_TOK_FN int32_t mq_sp_find(const mq_sp_table& T, uint64_t h, const uint8_t* s, int len, int* found) {
    uint32_t slot = (uint32_t)(h ^ (h >> 32)) & T.mask;
    for (;;) {
        const mq_sp_entry e = T.slots[slot];
        if (e.id == -1) { *found = 0; return -1; }
        if (e.hash == h && (int)(e.off_len & 0xffu) == len) {
            const uint8_t* p = T.pool + (e.off_len >> 8);
            int same = 1;
            for (int k = 0; k < len; ++k)
                if (p[k] != s[k]) { same = 0; break; }
            if (same) { *found = 1; return e.id; }
        }
        slot = (slot + 1) & T.mask;
    }
}

MQ_TOK_FN int mq_utf8_len(uint8_t c) { return c < 0x80 ? 1 : ((c & 0xe0) == 0xc0 ? 2 : ((c & 0xf0) == 0xe0 ? 3 : 4)); }

// B: Viterbi over norm[0..n).  Scratch: best [n+1] floats, bstart [n+1] int32 (-1 = unreached), bid [n+1] int32.
// Writes the piece ids (SentencePiece numbering, consecutive <unk> merged) to out[0..count) for count <= cap and returns the TOTAL count.
MQ_TOK_FN int mq_sp_viterbi(const mq_sp_table& T, const uint8_t* norm, int n, float* best, int32_t* bstart, int32_t* bid, int32_t* out, int cap) {
    if (n <= 0) return 0;
    for (int k = 0; k <= n; ++k) bstart[k] = -1;
    best[0] = 0.f;
    bstart[0] = 0;
    int s = 0;
    while (s < n) {
        const float here = best[s];
        const int mblen = mq_utf8_len(norm[s]) < n - s ? mq_utf8_len(norm[s]) : n - s;
        int has_single = 0;
        uint64_t h = MQ_FNV_OFFSET;
        int e = s;
        while (e < n && e - s < T.max_piece_bytes) {
            const int cl = mq_utf8_len(norm[e]);
            if (e + cl > n) break;
            for (int k = 0; k < cl; ++k) h = mq_wp_step(h, norm[e + k]);
            e += cl;
            int found;
            const int32_t id = mq_sp_find(T, h, norm + s, e - s, &found);
            if (!found) break;                       // no piece continues this prefix
            if (id >= 0) {
                const float cand = T.score[id] + here;
                if (bstart[e] < 0 || cand > best[e]) { best[e] = cand; bstart[e] = s; bid[e] = id; }
                if (e - s == mblen) has_single = 1;
            }
        }
        if (!has_single) {
            const int t = s + mblen;
            const float cand = T.unk_score + here;
            if (bstart[t] < 0 || cand > best[t]) { best[t] = cand; bstart[t] = s; bid[t] = T.unk_id; }
        }
        s += mblen;
    }
    // backtrack: count the pieces (consecutive <unk> merged), then write them front to back
    int total = 0, prev_unk = 0;
    for (int e = n; e > 0; e = bstart[e]) {
        const int unk = bid[e] == T.unk_id;
        if (!(unk && prev_unk)) ++total;
        prev_unk = unk;
    }
    int idx = total;
    prev_unk = 0;
    for (int e = n; e > 0; e = bstart[e]) {
        const int unk = bid[e] == T.unk_id;
        if (!(unk && prev_unk)) {
            --idx;
            if (idx < cap) out[idx] = bid[e];
        }
        prev_unk = unk;
    }
    return total;
}

// framing of a row: [prefix_id] ids... [suffix_id] pad...   (XLM-R: <s> ... </s> <pad>; T5 / SigLIP: ... </s>, padded with </s>)
struct mq_sp_frame {
    int32_t prefix_id;   // -1: none
    int32_t suffix_id;
    int32_t pad_id;
    int32_t id_offset;   // added to every SentencePiece id (fairseq layout of XLM-R: 1)
    int32_t unk_out;     // output id of <unk> (XLM-R: 3)
};

// C: pieces[0..min(total, cap)) -> row of `ld` ids, truncated to max_length including the specials; returns the row length
MQ_TOK_FN int mq_sp_gather(const mq_sp_table& T, const mq_sp_frame& F, const int32_t* pieces, int total, int cap, int max_length, int32_t* row, int ld) {
    const int specials = (F.prefix_id >= 0 ? 1 : 0) + 1;
    int keep = total < cap ? total : cap;
    if (keep > max_length - specials) keep = max_length - specials > 0 ? max_length - specials : 0;
    int cnt = 0;
    if (F.prefix_id >= 0) row[cnt++] = F.prefix_id;
    for (int j = 0; j < keep; ++j) row[cnt++] = pieces[j] == T.unk_id ? F.unk_out : pieces[j] + F.id_offset;
    row[cnt++] = F.suffix_id;
    for (int j = cnt; j < ld; ++j) row[j] = F.pad_id;
    return cnt;
}
