"""K14 — tokenisation on the device.

CPU part (not gpu): the PRODUCT algorithm source (marqo_amd/csrc/tokenize_algo.h, the functions the HIP kernels instantiate)
is compiled for the host by oracle/Makefile (oracle/tokenize_host.cpp) and pinned, with the product's own table builders,
against the Python tokenisers (which tests/test_tokenizers.py pins against `transformers`) and against `transformers`
directly — on fixed edge cases and on seeded random ASCII texts.
GPU part: the kernels through the C ABI give the same ids, and the routing wrapper (device for in-scope texts, host for the
rest) equals the host tokeniser on every text."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from marqo_amd.engine import gpu_tokenizers as GT
from marqo_amd.engine.tokenizers import ClipBpeTokenizer, WordPieceTokenizer
from tests.test_tokenizers import CORPUS, SENTENCES, _bert_vocab, _train_bpe

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ASCII_CASES = [
    "a photo of a cat", "The Quick  Brown fox, jumps over the lazy dog!", "it's built for images and text... isn't it?",
    "don't you'll we've they're I'M he'D 'tis 'station rock'n'roll ''s !'s 's's 1's x'", "'", "''", " ' s", "'S 'T 'RE 'Ve 'LL",
    "query: how much protein should a female eat", "", " ", "\t\n\r", "unseenwordzzz qqq 1234567890 007", "tab\tand\nnewline   spaces",
    "word " * 200, "x" * 90, "y" * 101 + " z", "a" * 96 + "!", "hello_world foo-bar (baz) [qux] {quux} <tag> #hash ##double @at",
    "UPPER lower MiXeD 123abc abc123 1a2b3c", "...!!!???", "a.b,c;d:e!f?g", "trailing space ", " leading", "the" * 30,
    "jumps jumping jumped photograph photographs synthetic documents", "$100 50% 3.14 1,000 a+b=c a/b\\c ~tilde^caret`tick|pipe",
]


def _random_texts(seed, n, alphabet=None):
    rng = np.random.default_rng(seed)
    words = [w for w in CORPUS if w.isascii()] + ["it's", "don't", "we'll", "x", "zzz", "Photo", "DOG!", "(cat)", "1234", "a1b2", "'re", "''",
                                                    "jumps", "photographs", "unseenword", "q" * 40, "##x", "#", "&amp", "e-mail", "U.S.A."]
    seps = [" ", " ", " ", "  ", "\t", "\n", ", ", ".", "! ", "'", "-"]
    out = []
    for _ in range(n):
        k = int(rng.integers(0, 40))
        s = "".join(words[int(rng.integers(len(words)))] + seps[int(rng.integers(len(seps)))] for _ in range(k))
        if rng.random() < 0.3:  # raw printable noise
            s += "".join(chr(int(c)) for c in rng.integers(0x20, 0x7f, size=int(rng.integers(1, 30))))
        out.append(s)
    return out


@pytest.fixture(scope="module")
def tokhost():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_build", "libtokhost.so"))
    return lib


@pytest.fixture(scope="module")
def bert_tok():
    return WordPieceTokenizer(_bert_vocab())


@pytest.fixture(scope="module")
def clip_tok():
    return ClipBpeTokenizer(_train_bpe(CORPUS, 150), context_length=77)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _host_wordpiece(lib, tok, texts, max_length):
    t = GT.build_wordpiece_table(tok)
    blob, off = GT.pack_texts(texts)
    n = len(texts)
    ids = np.zeros((n, max_length), np.int32)
    lens, status = np.zeros(n, np.int32), np.zeros(n, np.int32)
    uni = GT.build_unicode_table("wordpiece", bool(tok.lower))
    lib.tokhost_wordpiece(_p(t["slots"]), _p(t["pool"]), C.c_uint32(t["n_slots"]), t["unk_id"], t["cls_id"], t["sep_id"], t["pad_id"],
                          t["lower"], t["max_word_chars"], _p(uni), _p(blob), _p(off), C.c_int64(n), max_length, _p(ids), C.c_int64(max_length),
                          _p(lens), _p(status))
    return ids, lens, status


def _host_clip(lib, tok, texts, ctx):
    t = GT.build_bpe_table(tok)
    blob, off = GT.pack_texts(texts)
    n = len(texts)
    ids = np.zeros((n, ctx), np.int32)
    lens, status = np.zeros(n, np.int32), np.zeros(n, np.int32)
    uni = GT.build_unicode_table("clip", bool(tok.lower))
    lib.tokhost_clip_bpe(_p(t["slots"]), _p(t["byte_id"]), _p(t["byte_end_id"]), C.c_uint32(t["n_slots"]), t["sot_id"], t["eot_id"], t["lower"],
                         _p(uni), _p(blob), _p(off), C.c_int64(n), ctx, _p(ids), _p(lens), _p(status))
    return ids, lens, status


def test_scope_routing_follows_the_vocabularys_own_specials():
    """special-token spellings go to the host tokeniser whatever they look like: MPNet's <s> / </s> / <mask> as well as BERT's [SEP]"""
    from tests.test_tokenizers import _bert_vocab
    toks = ["<s>", "<pad>", "</s>", "<unk>"] + [t for t in _bert_vocab() if t not in ("[PAD]", "[CLS]", "[SEP]", "[MASK]")] + ["<mask>"]
    mp = WordPieceTokenizer({t: i for i, t in enumerate(toks)}, unk="[UNK]", cls="<s>", sep="</s>", pad="<pad>", mask="<mask>")
    assert GT.wordpiece_in_scope(mp, "the fox < the dog > [x]") and GT.wordpiece_in_scope(mp, "plain text")
    for s in ("the fox </s> the dog", "<s> again", "a <mask> b", "an [UNK] word", "<pad>"):
        assert not GT.wordpiece_in_scope(mp, s), s


def test_scope_routing(bert_tok, clip_tok):
    """host-side routing: only special-token spellings, `&` (CLIP: html.unescape), un-encodable and huge texts skip the device"""
    assert GT.wordpiece_in_scope(bert_tok, "plain ascii, with punct!") and GT.clip_in_scope(clip_tok, "plain ascii <b> 'quoted'")
    for ok in ("naïve", "東京", "bell\x07", "del\x7f", "vt\x0b", "Ελληνικά", "한국어", "हिन्दी", "😀 emoji"):
        assert GT.wordpiece_in_scope(bert_tok, ok) and GT.clip_in_scope(clip_tok, ok)
    assert not GT.wordpiece_in_scope(bert_tok, "the fox [SEP] dog") and GT.wordpiece_in_scope(bert_tok, "the fox [sep] dog [x]")
    assert not GT.clip_in_scope(clip_tok, "fish &amp; chips") and not GT.clip_in_scope(clip_tok, "x <START_OF_TEXT> y")
    assert not GT.wordpiece_in_scope(bert_tok, "lone \ud800 surrogate") and not GT.clip_in_scope(clip_tok, "lone \udfff surrogate")


@pytest.mark.parametrize("max_length", [512, 16, 3, 2])
def test_wordpiece_algorithm_matches_python_tokenizer(tokhost, bert_tok, max_length):
    texts = [t for t in ASCII_CASES + _random_texts(1, 300) if GT.wordpiece_in_scope(bert_tok, t)]
    assert len(texts) > 250
    ids, lens, status = _host_wordpiece(tokhost, bert_tok, texts, max_length)
    assert (status == 0).all()
    for i, t in enumerate(texts):
        ref = bert_tok.encode(t, max_length)
        assert lens[i] == len(ref) and ids[i, :lens[i]].tolist() == ref, repr(t)
        assert (ids[i, lens[i]:] == bert_tok.pad_id).all()


def test_wordpiece_algorithm_matches_transformers(tokhost, bert_tok):
    from transformers import BertTokenizer
    hf = BertTokenizer(vocab=_bert_vocab(), do_lower_case=True)
    texts = [t for t in ASCII_CASES + _random_texts(2, 100) if GT.wordpiece_in_scope(bert_tok, t)]
    ids, lens, _ = _host_wordpiece(tokhost, bert_tok, texts, 64)
    for i, t in enumerate(texts):
        assert ids[i, :lens[i]].tolist() == hf(t, truncation=True, max_length=64)["input_ids"], repr(t)


def test_wordpiece_cased_vocab(tokhost):
    vocab = {t: i for i, t in enumerate(["[PAD]", "[UNK]", "[CLS]", "[SEP]", "The", "the", "Fox", "fox", "##es", "F", "##ox"])}
    tok = WordPieceTokenizer(vocab, do_lower_case=False)
    texts = ["The fox Foxes the Fox", "THE FOX"]
    ids, lens, _ = _host_wordpiece(tokhost, tok, texts, 32)
    for i, t in enumerate(texts):
        assert ids[i, :lens[i]].tolist() == tok.encode(t, 32)


def test_context_dependent_characters_are_flagged(tokhost, bert_tok, clip_tok):
    """characters whose treatment depends on their neighbours (final sigma), code points beyond the table and malformed UTF-8 hand the
    text back to the host; everything else is tokenised"""
    texts = ["ok text", "caf\u00e9 x", "ΟΔΟΣ", "\U00030000 far plane", "plain"]
    ids, lens, status = _host_wordpiece(tokhost, bert_tok, texts, 16)
    assert status.tolist() == [0, 0, 1, 1, 0] and lens[2] == 0 and lens[1] > 0
    ids, lens, status = _host_clip(tokhost, clip_tok, texts + ["\u0130stanbul", "long\u017f"], 77)
    assert status.tolist() == [0, 0, 1, 1, 0, 1, 1]
    # malformed UTF-8 (cannot come from a Python str, but the C ABI takes bytes)
    blob = np.frombuffer(b"ok\xff\xfe bad\0", dtype=np.uint8)
    off = np.array([0, 8], dtype=np.int64)
    t = GT.build_wordpiece_table(bert_tok)
    ids, lens, status = np.zeros((1, 16), np.int32), np.zeros(1, np.int32), np.zeros(1, np.int32)
    tokhost.tokhost_wordpiece(_p(t["slots"]), _p(t["pool"]), C.c_uint32(t["n_slots"]), t["unk_id"], t["cls_id"], t["sep_id"], t["pad_id"],
                              t["lower"], t["max_word_chars"], _p(GT.build_unicode_table("wordpiece", True)), _p(blob), _p(off), C.c_int64(1), 16,
                              _p(ids), C.c_int64(16), _p(lens), _p(status))
    assert status.tolist() == [1]


@pytest.mark.parametrize("ctx", [77, 8, 2])
def test_clip_bpe_algorithm_matches_python_tokenizer(tokhost, clip_tok, ctx):
    texts = [t for t in ASCII_CASES + _random_texts(3, 300) if GT.clip_in_scope(clip_tok, t)]
    assert len(texts) > 250
    ids, lens, status = _host_clip(tokhost, clip_tok, texts, ctx)
    ref = clip_tok(texts, ctx)
    for i, t in enumerate(texts):
        if status[i]:  # a pre-token longer than the device scratch is handed back to the host, never mis-tokenised
            assert max(len(w) for w in t.split()) > 90, repr(t)
            continue
        assert ids[i].tolist() == ref[i].tolist(), repr(t)
        assert lens[i] == int(ref[i].argmax()) + 1
    assert status.sum() <= 3


def test_clip_bpe_algorithm_matches_transformers(tokhost, clip_tok):
    from transformers import CLIPTokenizer
    hf_vocab = {{"<start_of_text>": "<|startoftext|>", "<end_of_text>": "<|endoftext|>"}.get(t, t): i for t, i in clip_tok.encoder.items()}
    hf = CLIPTokenizer(vocab=hf_vocab, merges=[tuple(m) for m in clip_tok.merges])
    texts = [t for t in ASCII_CASES[:12] + _random_texts(4, 60) if GT.clip_in_scope(clip_tok, t) and len(t) < 150]
    ids, lens, status = _host_clip(tokhost, clip_tok, texts, 300)
    for i, t in enumerate(texts):
        if not status[i]:
            assert ids[i, :lens[i]].tolist() == hf(t)["input_ids"], repr(t)


def test_clip_uncased_and_duplicate_merges(tokhost):
    merges = _train_bpe(CORPUS, 60)
    merges = merges + [merges[3]]  # a duplicated merge line: python's dict keeps the LAST rank
    tok = ClipBpeTokenizer(merges, context_length=32, lower=False)
    texts = ["The QUICK brown Fox it'S", "photo of a DOG"]
    ids, lens, status = _host_clip(tokhost, tok, texts, 32)
    assert (status == 0).all() and np.array_equal(ids, tok(texts, 32).astype(np.int32))


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_device_wordpiece_equals_host(bert_tok):
    dev = GT.DeviceWordPieceTokenizer(bert_tok, "cuda")
    texts = ASCII_CASES + SENTENCES + _random_texts(5, 2000) + ["the fox [SEP] the dog [MASK]", "bell\x07char"]
    for max_length in (512, 16):
        got = dev(texts, max_length=max_length)
        ref = bert_tok(texts, max_length=max_length)
        assert np.array_equal(got["input_ids"], ref["input_ids"]) and np.array_equal(got["attention_mask"], ref["attention_mask"])
    ids, lens = dev.encode_device(["a photo of a cat"], 32)
    assert ids.is_cuda and ids.shape == (1, 32) and ids[0, :int(lens[0])].tolist() == bert_tok.encode("a photo of a cat", 32)
    assert dev([], max_length=8)["input_ids"].shape[0] == 0
    # Unicode on the device: multi-script vocabulary, >= 10k fuzz texts over both casings; the wrapper equals the host tokeniser on every
    # text (the few context-dependent ones take the host route inside it)
    for lower, seed in ((True, 31), (False, 32)):
        tok = WordPieceTokenizer(_multiscript_vocab(), do_lower_case=lower)
        d = GT.DeviceWordPieceTokenizer(tok, "cuda")
        texts = _unicode_texts(seed, 5200) + ["ΟΔΟΣ final sigma", "the fox [SEP] dog", "\U00030000 far", "lone \ud800"]
        got, ref = d(texts, max_length=48), tok(texts, max_length=48)
        assert np.array_equal(got["input_ids"], ref["input_ids"]) and np.array_equal(got["attention_mask"], ref["attention_mask"])


@pytest.mark.gpu
def test_device_clip_bpe_equals_host(clip_tok):
    dev = GT.DeviceClipBpeTokenizer(clip_tok, "cuda")
    texts = ASCII_CASES + SENTENCES + _random_texts(6, 2000) + ["fish &amp; chips", "x <start_of_text> y", "z" * 300]
    for ctx in (77, 8):
        assert np.array_equal(dev(texts, ctx), clip_tok(texts, ctx))
    ids, lens = dev.encode_device(texts)
    ref = clip_tok(texts)
    assert np.array_equal(lens.numpy(), ref.argmax(1) + 1)
    corpus = " ".join(_unicode_texts(21, 400)).split()
    tok = ClipBpeTokenizer(_train_bpe([w for w in corpus if w][:3000], 200), context_length=77)
    d = GT.DeviceClipBpeTokenizer(tok, "cuda")
    texts = _unicode_texts(33, 10500) + ["İstanbul", "ΟΔΟΣ", "fish &amp; chips", "x <start_of_text> y", "\x1c'm \x1d"]
    assert np.array_equal(d(texts, 77), tok(texts, 77))
    with pytest.raises(UnicodeEncodeError):   # a lone surrogate has no UTF-8 bytes: the host tokenizer (and open_clip's) raises, so does the device route
        d(["lone \udfff"], 77)


def test_device_tokenizers_refuse_cpu(bert_tok):
    from marqo_amd._lib import MarqoHipUnavailableError
    with pytest.raises(MarqoHipUnavailableError):
        GT.DeviceWordPieceTokenizer(bert_tok, "cpu")


# ---- Unicode: multi-script fuzz of the product algorithm (host build) against the Python tokenisers -----------------------------------
_SCRIPTS = {
    "latin": "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ",
    "latin_ext": "àáâãäåæçèéêëìíîïñòóôõöøùúûüýÿÀÁÂÃÄÅÆÇÈÉÊËÌÍÎÏÑÒÓÔÕÖØÙÚÛÜÝßœŒšŠžŽłŁđĐığĞşŞőŐ",
    "greek": "αβγδεζηθικλμνξοπρστυφχψωάέήίόύώΑΒΓΔΕΖΗΘΙΚΛΜΝΞΟΠΡΤΥΦΧΨΩς",       # (capital sigma is the context-dependent one: added separately)
    "cyrillic": "абвгдежзийклмнопрстуфхцчшщъыьэюяАБВГДЕЖЗИЙКЛМНОПРСТУФХЦЧШЩЪЫЬЭЮЯёЁ",
    "cjk": "東京都日本語中文汉字學校愛国",
    "kana": "あいうえおかがきぎくぐけげこごさざしじすずせぜそぞたアイウエオカガキギクグパピプペポ",
    "hangul": "한국어안녕하세요서울대학교감사합니다",
    "devanagari": "हिन्दीकखगघचछजझटठडढणतथदधनपफबभमयरलवशषसह़ािीुूेैोौ्",
    "arabic": "العربيةمرحبابكمفيهذاالنصًٌٍَُِّْ",
    "hebrew": "שלוםעולםעבריתבְּרֵאשִׁית",
    "thai": "ภาษาไทยสวัสดีครับขอบคุณ",
    "digits": "0123456789٠١٢٣٤٥٦٧٨٩０１２３²³½",
    "punct": ".,;:!?'\"()[]{}<>-–—…«»“”‘’·•¿¡。、「」！？",
    "symbols": "$€£¥©®™°±×÷=+*/\\^~`|@#%_",
    "emoji": "😀😂🎉🔥👍🏽❤️🇯🇵",
    "marks": "゙゚̧̣́̀̈̂̃̊̇",
    "space": " \t\n\r  　  ​­﻿\x0b\x0c\x1c\x85\x00\x07\x7f�",
}


def _unicode_texts(seed, n):
    rng = np.random.default_rng(seed)
    names = list(_SCRIPTS)
    out = []
    for _ in range(n):
        k = int(rng.integers(0, 14))
        parts = []
        main = names[int(rng.integers(len(names)))]
        for _ in range(k):
            script = main if rng.random() < 0.6 else names[int(rng.integers(len(names)))]
            chars = _SCRIPTS[script]
            L = int(rng.integers(1, 9))
            w = "".join(chars[int(rng.integers(len(chars)))] for _ in range(L))
            r = rng.random()
            if r < 0.08:
                w += "'" + ["s", "t", "re", "ve", "m", "ll", "d", "S", "x", ""][int(rng.integers(10))]
            elif r < 0.12:
                w += _SCRIPTS["marks"][int(rng.integers(len(_SCRIPTS["marks"])))]
            parts.append(w)
            parts.append([" ", " ", " ", "", "  ", ", ", ". ", "\n", "-", "'"][int(rng.integers(10))])
        out.append("".join(parts))
    return out


def _multiscript_vocab():
    chars = sorted({c for v in _SCRIPTS.values() for c in v} | {c.lower() for v in _SCRIPTS.values() for c in v})
    import unicodedata
    extra = set()
    for c in chars:  # what lower + NFD leaves behind (base letters, jamo, kana without dakuten ...)
        extra.update(unicodedata.normalize("NFD", c.lower()))
    chars = sorted(set(chars) | extra)
    toks = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + chars + ["##" + c for c in chars]
    toks += ["ab", "##cd", "the", "abc", "東京", "αβ", "##γδ", "при", "##вет", "한", "ᄒ", "하", "한", "か", "##き", "हि", "مر", "##حب", "שלום", "ภา", "##ษา", "éa", "ea", "e", "naive", "cafe"]
    seen, out = set(), []
    for t in toks:
        if t not in seen and t not in ("##",):
            seen.add(t)
            out.append(t)
    return {t: i for i, t in enumerate(out)}


@pytest.mark.parametrize("lower", [True, False])
def test_wordpiece_unicode_fuzz_matches_python_tokenizer(tokhost, lower):
    """>= 5k texts per casing over a dozen scripts, combining marks, exotic whitespace and control characters: ids identical to the
    Python tokeniser for every text the device path accepts, and it accepts (almost) all of them"""
    tok = WordPieceTokenizer(_multiscript_vocab(), do_lower_case=lower)
    texts = _unicode_texts(11 if lower else 12, 5500) + ["naïve café über straße", "東京 photos 2024", "İstanbul ǅ ǆ ﬁ ŉ ẞ", "ｆｕｌｌｗｉｄｔｈ Ａ１",
                                                          "é vs é", "が ga が", "한국어 한", "ﬁ Ω Å µ ẛ̣"]
    texts = [t for t in texts if GT.wordpiece_in_scope(tok, t)]
    ids, lens, status = _host_wordpiece(tokhost, tok, texts, 64)
    flagged = 0
    for i, t in enumerate(texts):
        if status[i]:
            flagged += 1
            continue
        ref = tok.encode(t, 64)
        assert lens[i] == len(ref) and ids[i, :lens[i]].tolist() == ref, (repr(t), ids[i, :lens[i]].tolist(), ref)
    assert len(texts) > 5000
    # lower-casing vocabularies (every registry model): only final-sigma-capable text is handed back.  Cased vocabularies run NFC, which
    # is contextual: every text holding a combining mark (most Devanagari / Arabic-with-harakat / Hebrew-with-niqqud / Thai texts of this
    # mark-heavy fuzz) goes back to the host
    assert flagged < (0.02 if lower else 0.75) * len(texts), flagged


def test_wordpiece_unicode_fuzz_matches_transformers(tokhost):
    """the same algorithm against transformers' own BertTokenizer (slow tokenizer, the implementation the reference calls) on the
    multi-script vocabulary"""
    from transformers import BertTokenizer
    vocab = _multiscript_vocab()
    tok = WordPieceTokenizer(vocab, do_lower_case=True)
    hf = BertTokenizer(vocab=vocab, do_lower_case=True)
    texts = [t for t in _unicode_texts(13, 1500) if GT.wordpiece_in_scope(tok, t)]
    ids, lens, status = _host_wordpiece(tokhost, tok, texts, 48)
    checked = 0
    for i, t in enumerate(texts):
        if status[i]:
            continue
        assert ids[i, :lens[i]].tolist() == hf(t, truncation=True, max_length=48)["input_ids"], repr(t)
        checked += 1
    assert checked > 1300


def test_clip_bpe_unicode_fuzz_matches_python_tokenizer(tokhost):
    corpus = " ".join(_unicode_texts(21, 400)).split()
    tok = ClipBpeTokenizer(_train_bpe([w for w in corpus if w][:3000], 200), context_length=77)
    texts = _unicode_texts(22, 5500) + ["naïve café über straße", "東京 photos 2024", "it's l'été d'accord 'S 'RE", "ǅ ǆ ﬁ ẞ ｆｕｌｌ １２３ ²³½"]
    texts = [t for t in texts if GT.clip_in_scope(tok, t)]
    ids, lens, status = _host_clip(tokhost, tok, texts, 77)
    ref = tok(texts, 77)
    flagged = 0
    for i, t in enumerate(texts):
        if status[i]:
            flagged += 1
            continue
        assert ids[i].tolist() == ref[i].tolist(), (repr(t), ids[i, :12].tolist(), ref[i, :12].tolist())
        assert lens[i] == int(ref[i].argmax()) + 1
    assert len(texts) > 5000 and flagged < 0.02 * len(texts), flagged


# ---- SentencePiece unigram ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def sp_model(tmp_path_factory):
    import sentencepiece as spm
    d = tmp_path_factory.mktemp("spm")
    corpus = [" ".join(CORPUS)] * 20 + SENTENCES[:6] * 5 + _unicode_texts(41, 600)
    (d / "c.txt").write_text("\n".join(t.replace("\n", " ").replace("\x00", "") for t in corpus), encoding="utf-8")
    spm.SentencePieceTrainer.train(input=str(d / "c.txt"), model_prefix=str(d / "sentencepiece.bpe"), vocab_size=600, model_type="unigram",
                                   character_coverage=0.98, hard_vocab_limit=False, minloglevel=2)
    return d


def _host_sentencepiece(lib, T, texts, max_length, frame):
    blob, off = GT.pack_texts(texts)
    n = len(texts)
    ids, lens, st = np.zeros((n, max_length), np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    normout = np.zeros(3 * len(blob) + 16 * n + 64, np.uint8)
    nlen = np.zeros(n, np.int32)
    lib.tokhost_sentencepiece(_p(T["slots"]), _p(T["pool"]), _p(T["scores"]), _p(T["nmap"]), _p(T["npool"]), _p(T["ccc"]), C.c_uint32(T["n_slots"]),
                              T["unk_id"], C.c_float(T["unk_score"]), T["add_dummy_prefix"], T["remove_extra_ws"], T["max_piece_bytes"], *frame,
                              _p(blob), _p(off), C.c_int64(n), max_length, _p(ids), C.c_int64(max_length), _p(lens), _p(st), _p(normout), _p(nlen))
    norms = [bytes(normout[3 * int(off[i]) + 16 * i: 3 * int(off[i]) + 16 * i + nlen[i]]).decode("utf-8") if not st[i] else None for i in range(n)]
    return ids, lens, st, norms


def test_sentencepiece_unigram_algorithm_matches_sentencepiece(tokhost, sp_model):
    """the product's normaliser + Viterbi (host build) against the sentencepiece library itself and against the XLM-R wrapper: the
    normalised text equals SentencePieceProcessor.Normalize, the ids equal XlmRobertaTokenizer.encode, on >= 5k multi-script texts"""
    from marqo_amd.engine.tokenizers import XlmRobertaTokenizer
    tok = XlmRobertaTokenizer(str(sp_model))
    T = GT.build_sentencepiece_table(tok.sp)
    texts = _unicode_texts(42, 5500) + ["hello world", "  Hello   World\t x ", "", "   ", "ｆｕｌｌ ㍿ ﬁ", "東京都 x", "naïve café", "▁ literal", "á decomposed"]
    texts = [t for t in texts if GT._encodable(t)]
    for max_length in (64, 6):
        ids, lens, st, norms = _host_sentencepiece(tokhost, T, texts, max_length, (0, 2, 1, 1, 3))
        flagged = 0
        for i, t in enumerate(texts):
            if st[i]:
                flagged += 1
                continue
            assert norms[i] == tok.sp.normalize(t), repr(t)
            ref = tok.encode(t, max_length)
            assert lens[i] == len(ref) and ids[i, :lens[i]].tolist() == ref, (repr(t), ids[i, :lens[i]].tolist(), ref)
            assert (ids[i, lens[i]:] == 1).all()
        # this fuzz appends a raw combining accent to ~1 word in 25 and strings marks together at random: those texts (composition /
        # canonical reordering is contextual) are the ones handed back
        assert len(texts) > 5000 and flagged < 0.5 * len(texts)
    plain = [t for t in texts if all(ord(c) < 0x300 or 0x370 <= ord(c) < 0x3000 and not __import__("unicodedata").combining(c) for c in t)]
    ids, lens, st, _ = _host_sentencepiece(tokhost, T, plain[:800], 64, (0, 2, 1, 1, 3))
    assert st.sum() <= 0.02 * len(st)          # precomposed Latin / Greek / Cyrillic text stays on the device


def test_sentencepiece_table_properties(sp_model):
    import sentencepiece as spm
    sp = spm.SentencePieceProcessor(model_file=str(sp_model / "sentencepiece.bpe.model"))
    T = GT.build_sentencepiece_table(sp)
    used = T["slots"]["id"] != -1
    assert used.sum() >= sum(1 for i in range(len(sp)) if sp.IsUnknown(i) is False and not sp.IsControl(i)) and (T["n_slots"] & (T["n_slots"] - 1)) == 0
    assert T["unk_score"] == pytest.approx(min(sp.GetScore(i) for i in range(len(sp)) if not sp.IsControl(i) and not sp.IsUnknown(i)) - 10.0)
    nm = T["nmap"]
    assert (nm[0x0301] & 0xff) == 0xff and (nm[0x3099] & 0xff) == 0xff and (nm[0x2581] & 0xff) == 0xff     # composing marks, U+2581 -> host
    assert (nm[0x05B4] & 0xff) != 0xff and (nm[ord("é")] & 0xff) == 2 and (nm[ord(" ")] & 0xff) == 1         # Hebrew point: context-free
    assert bytes(T["npool"][nm[0x3000] >> 8: (nm[0x3000] >> 8) + 1]) == b" "                                 # ideographic space -> space


@pytest.mark.gpu
def test_device_sentencepiece_equals_host(sp_model, tmp_path):
    from marqo_amd.engine.tokenizers import SiglipTokenizer, XlmRobertaTokenizer
    tok = XlmRobertaTokenizer(str(sp_model))
    dev = GT.DeviceSentencePieceTokenizer(tok, "cuda")
    texts = _unicode_texts(43, 10500) + ["hello world", "", "   ", "ｆｕｌｌ ㍿ ﬁ", "á decomposed", "\U00030000 far", "lone \ud800"]
    texts = [t for t in texts if GT._encodable(t)] + ["word " * 300]
    for max_length in (64, 8):
        got, ref = dev(texts, max_length=max_length), tok(texts, max_length=max_length)
        assert np.array_equal(got["input_ids"], ref["input_ids"]) and np.array_equal(got["attention_mask"], ref["attention_mask"])
    # SigLIP framing over a T5-style model: canonicalize -> pieces -> </s>, padded with </s> to the context length
    import shutil
    shutil.copy(str(sp_model / "sentencepiece.bpe.model"), str(tmp_path / "spiece.model"))
    stok = SiglipTokenizer(str(tmp_path / "spiece.model"), context_length=16)
    sdev = GT.DeviceSentencePieceTokenizer(stok, "cuda")
    sample = texts[:3000] + ["A red_dress, size: M (new!)", "UPPER lower"]
    assert np.array_equal(sdev(sample), stok(sample))
