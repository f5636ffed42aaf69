"""The C-ABI shared library loads WITHOUT a GPU and exports every entry point include/marqo_hip.h declares; the ctypes
binding covers the same set; POD struct layouts match the header (no compute is launched here)."""
import ctypes as C
import os
import re

from marqo_amd import _lib as L

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "marqo_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mq_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    names = _declared()
    assert len(names) >= 25
    lib = L.load()
    raw = C.CDLL(str(L.LIB_PATH))
    for n in names:
        assert hasattr(raw, n), f"libmarqo_hip.so does not export {n}"
        assert n in L.EXPORTED_SYMBOLS, f"ctypes binding is missing {n}"
    assert set(L.EXPORTED_SYMBOLS) <= set(names), set(L.EXPORTED_SYMBOLS) - set(names)
    assert lib.mq_abi_version() == L.ABI_VERSION == 14 and lib.mq_build_arch() == b"gfx950"


def test_struct_layouts_match_header():
    assert C.sizeof(L.BlockWeights) == 36 * 8           # ABI 10: + attn_ln_g / _b, mlp_ln_g / _b (the EVA02 blocks' sub-LayerNorms); ABI 12: + their folded out / fc2 tensors
    # ... + d_rel_bias, rel_span, residual_stream (ABI 5: reserved0), fp8_mlp_extra, rope_prefix (ABI 6: reserved1) + ABI 10: d_rope_table, mlp_ln_dim, reserved2
    ENC = 8 * 4 + 4 * 4 + 3 * 8 + 8 + 2 * 4 + 2 * 4 + 8 + 2 * 4
    assert C.sizeof(L.EncoderCfg) == ENC
    assert C.sizeof(L.VitCfg) == ENC + 3 * 4 + 6 * 4 + 4 * 4 + 4  # + tail padding to 8
    assert C.sizeof(L.VitWeights) == 11 * 8 and C.sizeof(L.MapHead) == 11 * 8
    assert C.sizeof(L.ClipTextCfg) == ENC + 4 * 4 and C.sizeof(L.ClipTextWeights) == 7 * 8
    assert C.sizeof(L.BertCfg) == ENC + 5 * 4 + 4 and C.sizeof(L.BertWeights) == 9 * 8   # + proj_hidden, out_dim / proj1_w, proj1_b, proj2_w
    assert C.sizeof(L.QueueCfg) == 10 * 4 and C.sizeof(L.QueueStats) == 9 * 8   # ABI 14: the native request queue


def test_argument_errors_are_reported_without_a_gpu():
    lib = L.load()
    assert lib.mq_gemm_bf16(None, 0, None, 0, None, None, None, 0, 1, 4, 64, 0, None) == -1
    assert b"null operand" in lib.mq_last_error()
    cfg = L.EncoderCfg(width=100, layers=1, heads=1, mlp_dim=64, act=1, post_ln=0, mask=0, ln_eps=1e-5)
    assert lib.mq_encoder_forward(C.byref(cfg), None, None, 0, None, 0, 0, 0, None, 0, None) == -1
    assert b"multiple of 64" in lib.mq_last_error()
    assert lib.mq_encoder_workspace_bytes(C.byref(L.EncoderCfg(width=768, layers=12, heads=12, mlp_dim=3072, act=1)), 100, 2) > 0
    assert lib.mq_chunk_grid_count(3, 3, 0) == 10 and lib.mq_chunk_grid_count(3, 3, 1) == 14 and lib.mq_chunk_grid_count(0, 3, 0) == 0
    assert lib.mq_resample_ksize(640, 224) == 13 and lib.mq_resample_ksize(80, 224) == 5
    # ABI 14, the native request queue: arguments are judged before any device is asked for
    h = C.c_void_p()
    assert lib.mq_queue_create(None, None, None, C.byref(h)) == -1 and b"null pointer" in lib.mq_last_error() and not h
    qc = L.QueueCfg(kind=L.QUEUE_BERT, device=0, max_seqs=0, max_rows=512, normalize=1, depth=2, window_us=0, graphs=0)
    bc, bw = L.BertCfg(), L.BertWeights()
    assert lib.mq_queue_create(C.byref(qc), C.cast(C.byref(bc), C.c_void_p), C.cast(C.byref(bw), C.c_void_p), C.byref(h)) == -1 and not h
    assert b"max_seqs" in lib.mq_last_error()
    assert lib.mq_queue_encode(None, None, None, 1, None) == -1 and b"null queue" in lib.mq_last_error()
    assert lib.mq_queue_get_stats(None, None) == -1 and lib.mq_queue_destroy(None) == 0


def test_product_never_imports_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dp, _, files in os.walk(os.path.join(root, "marqo_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)
                assert "liboracle" not in src


def test_host_gather_checked_bounds_and_copies():
    """mq_host_gather_checked: the destination's capacity is validated before any byte moves; within bounds it is mq_host_gather"""
    import ctypes as C
    import numpy as np
    from marqo_amd import _lib as L
    lib = L.load()
    srcs_np = [np.arange(n, dtype=np.uint8) for n in (1000, 0, 5_000_000, 777)]
    offs = np.asarray([0, 1024, 1024, 5_001_200], dtype=np.int64)
    nbytes = np.asarray([a.nbytes for a in srcs_np], dtype=np.int64)
    dst = np.zeros(5_002_000, dtype=np.uint8)
    srcs = (C.c_void_p * 4)(*[a.ctypes.data for a in srcs_np])
    L.check(lib.mq_host_gather_checked(srcs, nbytes.ctypes.data, offs.ctypes.data, 4, dst.ctypes.data, dst.nbytes, 4))
    for a, o in zip(srcs_np, offs):
        assert np.array_equal(dst[o:o + a.nbytes], a)
    rc = lib.mq_host_gather_checked(srcs, nbytes.ctypes.data, offs.ctypes.data, 4, dst.ctypes.data, 5_001_900, 4)   # the last item would overrun
    assert rc != 0 and b"leaves" in lib.mq_last_error()
