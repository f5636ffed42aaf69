#!/usr/bin/env python
"""K14 micro-benchmark: device tokenisation (csrc/tokenize.hip) vs the host Python tokenisers on the same texts.
No real vocabulary exists offline: a BPE merge list / WordPiece vocabulary is trained on a synthetic corpus of pseudo-words
(a few thousand entries — the tables are smaller than the real 49k / 30k ones, the per-text work is comparable).
usage: python tools/tokenize_bench.py [--texts 1024] [--words 60]"""
import argparse
import collections
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from marqo_amd.engine import gpu_tokenizers as GT
from marqo_amd.engine.tokenizers import ClipBpeTokenizer, WordPieceTokenizer, _byte_to_unicode


def corpus(rng, n_words):
    syll = ["ka", "to", "mi", "ra", "sen", "lo", "vi", "pa", "chu", "ne", "ing", "er", "st", "qu", "an", "the", "or", "ti", "ex", "po"]
    return ["".join(rng.choice(syll, size=int(rng.integers(1, 5)))) for _ in range(n_words)]


def train_bpe(words, n_merges):
    b2u = _byte_to_unicode()
    vocab = collections.Counter()
    for w in words:
        sym = [b2u[b] for b in w.encode()]
        sym[-1] += "</w>"
        vocab[tuple(sym)] += 1
    merges = []
    for _ in range(n_merges):
        pairs = collections.Counter()
        for sym, c in vocab.items():
            for p in zip(sym[:-1], sym[1:]):
                pairs[p] += c
        if not pairs:
            break
        best = max(sorted(pairs), key=lambda p: pairs[p])
        merges.append(best)
        new = collections.Counter()
        for sym, c in vocab.items():
            out, i = [], 0
            while i < len(sym):
                if i + 1 < len(sym) and (sym[i], sym[i + 1]) == best:
                    out.append(sym[i] + sym[i + 1]); i += 2
                else:
                    out.append(sym[i]); i += 1
            new[tuple(out)] += c
        vocab = new
    return merges


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--texts", type=int, default=1024)
    ap.add_argument("--words", type=int, default=60)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    words = sorted(set(corpus(rng, 4000)))
    merges = train_bpe(words, 1500)
    clip = ClipBpeTokenizer(merges, context_length=77)
    pieces = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list("abcdefghijklmnopqrstuvwxyz0123456789.,!?") + \
             ["##" + c for c in "abcdefghijklmnopqrstuvwxyz0123456789"] + words[:1500] + ["##" + w for w in words[1500:2500]]
    bert = WordPieceTokenizer({t: i for i, t in enumerate(dict.fromkeys(pieces))})
    texts = [" ".join(rng.choice(words, size=a.words)) + "." for _ in range(a.texts)]
    dclip, dbert = GT.DeviceClipBpeTokenizer(clip, "cuda"), GT.DeviceWordPieceTokenizer(bert, "cuda")
    assert np.array_equal(dclip(texts), clip(texts))
    ref = bert(texts, max_length=128)
    got = dbert(texts, max_length=128)
    assert np.array_equal(got["input_ids"], ref["input_ids"])
    nbytes = sum(len(t) for t in texts)

    def host_clip_cold():
        clip._cache.clear(); clip(texts)

    def host_bert_cold():
        bert._cache.clear(); bert(texts, max_length=128)
    rows = [("clip bpe  host (cold word cache)", timeit(host_clip_cold, 2)), ("clip bpe  host (warm word cache)", timeit(lambda: clip(texts), 3)),
            ("clip bpe  device (encode_device)", timeit(lambda: dclip.encode_device(texts), 10)),
            ("wordpiece host (cold word cache)", timeit(host_bert_cold, 2)), ("wordpiece host (warm word cache)", timeit(lambda: bert(texts, max_length=128), 3)),
            ("wordpiece device (encode_device)", timeit(lambda: dbert.encode_device(texts, 128), 10))]
    print(f"{a.texts} texts, {nbytes / a.texts:.0f} bytes each, {a.words} words each; ids identical on both routes")
    for name, ms in rows:
        print(f"  {name:36s} {ms:9.2f} ms  {a.texts / ms * 1e3:12.0f} texts/s")
    # kernel-only time (texts already staged)
    d_blob, d_off, total = dclip._stage(texts)
    ws = dclip._workspace(a.texts, total, 128)
    import ctypes as C
    from marqo_amd import _lib as L
    ids = torch.zeros(a.texts, 77, dtype=torch.int32, device="cuda")
    meta = torch.zeros(2, a.texts, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    run = lambda: L.check(dclip.lib.mq_tokenize_clip_bpe(C.byref(dclip.vocab), d_blob.data_ptr(), d_off.data_ptr(), a.texts, total, 77, ids.data_ptr(),
                                                          meta[0].data_ptr(), meta[1].data_ptr(), ws.data_ptr(), ws.numel(), s))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    run(); e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    print(f"  clip bpe kernels (split + merge + gather) alone: {e0.elapsed_time(e1) / 20 * 1e3:.0f} us per launch ({nbytes / (e0.elapsed_time(e1) / 20 * 1e-3) / 1e9:.2f} GB/s of text)")
    ids2 = torch.zeros(a.texts, 128, dtype=torch.int32, device="cuda")
    run2 = lambda: L.check(dbert.lib.mq_tokenize_wordpiece(C.byref(dbert.vocab), d_blob.data_ptr(), d_off.data_ptr(), a.texts, total, 128, ids2.data_ptr(), 128,
                                                            meta[0].data_ptr(), meta[1].data_ptr(), ws.data_ptr(), ws.numel(), s))
    run2(); e0.record()
    for _ in range(20):
        run2()
    e1.record(); torch.cuda.synchronize()
    print(f"  wordpiece kernels (split + pieces + gather) alone: {e0.elapsed_time(e1) / 20 * 1e3:.0f} us per launch")


if __name__ == "__main__":
    main()
