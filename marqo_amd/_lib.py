"""ctypes binding of libmarqo_hip.so (the C ABI declared in include/marqo_hip.h).

This is the ONLY way the Python host reaches the GPU arithmetic: PyTorch-ROCm is used for
device memory, streams and torch.distributed, never for the math of the hot path.  There is
no CPU fallback: if the shared library is missing or cannot be loaded, importing callers get
a loud ``MarqoHipUnavailableError``.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess
import sys
import sysconfig
import threading
from typing import Optional
from pathlib import Path

PKG_DIR = Path(__file__).resolve().parent
CSRC_DIR = PKG_DIR / "csrc"
LIB_DIR = PKG_DIR / "lib"
LIB_PATH = Path(os.environ["MARQO_AMD_LIB"]) if os.environ.get("MARQO_AMD_LIB") else LIB_DIR / "libmarqo_hip.so"  # override: diagnostic builds
HEADER_PATH = PKG_DIR.parent / "include" / "marqo_hip.h"
TORCH_OPS_SRC = CSRC_DIR / "torch_ops.cpp"
TORCH_OPS_PATH = LIB_DIR / "libmarqo_torch_ops.so"   # torch.ops.marqo_hip.*: the PyTorch custom-op face of the same C ABI
STAGE_SRC = CSRC_DIR / "py_stage.cpp"
STAGE_PATH = LIB_DIR / "_mq_stage.so"                # CPython extension: a batch of Pillow images -> the pinned staging buffer in one call

MQ_OK = 0
NO_SCRATCH_UNITS = ("rowops", "gemm_bf16", "gemm_wd", "gemm_fp8", "gemm_small", "attention", "attn_proj", "panel_gemm", "embed")  # build() refuses register spills in these
ABI_VERSION = 14
MQ_PREC_BF16, MQ_PREC_FP8 = 0, 1
MQ_ACT_GELU, MQ_ACT_QUICKGELU, MQ_ACT_SILU = 1, 2, 3
MQ_MASK_NONE, MQ_MASK_CAUSAL, MQ_MASK_CAUSAL_CLS = 0, 1, 2
MQ_POOL_MEAN, MQ_POOL_CLS = 0, 1
MQ_VIT_POOL_CLS, MQ_VIT_POOL_MAP, MQ_VIT_POOL_AVG, MQ_VIT_POOL_QUERY = 0, 1, 2, 3
MQ_EPI_BIAS, MQ_EPI_GELU, MQ_EPI_QUICKGELU, MQ_EPI_RESIDUAL, MQ_EPI_OUT_F32, MQ_EPI_OUT_FP8 = 1, 2, 4, 8, 16, 32
MQ_EPI_ROW_STATS, MQ_EPI_LN_APPLY = 64, 128
MQ_COMBINE_RAW, MQ_COMBINE_NORMALIZE, MQ_COMBINE_NORMALIZE_IF_NONZERO = 0, 1, 2
MQ_IMG_RGB, MQ_IMG_NEAREST, MQ_IMG_RGBA = 0, 1, 2
MQ_PROF_FAMILIES = 6
PROF_FAMILY_NAMES = ("gemm", "layernorm", "attention", "embed", "pool_head", "preprocess")


class MarqoHipUnavailableError(RuntimeError):
    """The HIP extension is missing / unloadable.  There is deliberately no fallback."""


class MarqoHipError(RuntimeError):
    """A libmarqo_hip entry point returned an error code."""


# ---- POD structs (mirror include/marqo_hip.h exactly) ------------------------------------------
class BlockWeights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "ln1_g", "ln1_b", "qkv_w", "qkv_b", "out_w", "out_b",
        "ln2_g", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
        "qkv_w8", "qkv_ws", "out_w8", "out_ws", "fc1_w8", "fc1_ws", "fc2_w8", "fc2_ws",
        "qkv_wf", "qkv_sf", "qkv_bf", "fc1_wf", "fc1_sf", "fc1_bf",
        "attn_ln_g", "attn_ln_b", "mlp_ln_g", "mlp_ln_b",
        "out_wf", "out_sf", "out_bf", "fc2_wf", "fc2_sf", "fc2_bf")]


class EncoderCfg(C.Structure):
    _fields_ = [("width", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32), ("mlp_dim", C.c_int32),
                ("act", C.c_int32), ("post_ln", C.c_int32), ("mask", C.c_int32), ("ln_eps", C.c_float),
                ("precision", C.c_int32), ("attn_width", C.c_int32), ("fp8_first_layer", C.c_int32), ("mlp_glu", C.c_int32),
                ("d_fp8_act_scale", C.c_void_p), ("d_fp8_act_amax", C.c_void_p), ("d_rope_inv_freq", C.c_void_p),
                ("d_rel_bias", C.c_void_p), ("rel_span", C.c_int32), ("residual_stream", C.c_int32),
                ("fp8_mlp_extra", C.c_int32), ("rope_prefix", C.c_int32), ("d_rope_table", C.c_void_p), ("mlp_ln_dim", C.c_int32),
                ("reserved2", C.c_int32)]


class MapHead(C.Structure):
    """mq_map_head: timm AttentionPoolLatent (SigLIP 'map' pooling)"""
    _fields_ = [("q", C.c_void_p), ("kv_w", C.c_void_p), ("kv_b", C.c_void_p), ("proj_w", C.c_void_p), ("proj_b", C.c_void_p),
                ("ln_g", C.c_void_p), ("ln_b", C.c_void_p), ("fc1_w", C.c_void_p), ("fc1_b", C.c_void_p),
                ("fc2_w", C.c_void_p), ("fc2_b", C.c_void_p)]


class VitWeights(C.Structure):
    _fields_ = [("patch_w", C.c_void_p), ("cls", C.c_void_p), ("pos", C.c_void_p),
                ("ln_pre_g", C.c_void_p), ("ln_pre_b", C.c_void_p), ("blocks", C.POINTER(BlockWeights)),
                ("ln_post_g", C.c_void_p), ("ln_post_b", C.c_void_p), ("proj_w", C.c_void_p), ("map", C.POINTER(MapHead)),
                ("proj_b", C.c_void_p)]


class VitCfg(C.Structure):
    _fields_ = [("enc", EncoderCfg), ("image_size", C.c_int32), ("patch_size", C.c_int32), ("out_dim", C.c_int32),
                ("mean", C.c_float * 3), ("std", C.c_float * 3), ("pool", C.c_int32), ("map_mlp_dim", C.c_int32), ("pool_dim", C.c_int32),
                ("pool_heads", C.c_int32)]


class ClipTextWeights(C.Structure):
    _fields_ = [("tok_emb", C.c_void_p), ("pos", C.c_void_p), ("blocks", C.POINTER(BlockWeights)),
                ("ln_final_g", C.c_void_p), ("ln_final_b", C.c_void_p), ("proj_w", C.c_void_p), ("proj_b", C.c_void_p)]


class ClipTextCfg(C.Structure):
    _fields_ = [("enc", EncoderCfg), ("vocab", C.c_int32), ("ctx", C.c_int32), ("out_dim", C.c_int32), ("cls_pos", C.c_int32)]


class BertWeights(C.Structure):
    _fields_ = [("word_emb", C.c_void_p), ("pos_emb", C.c_void_p), ("type_emb", C.c_void_p),
                ("emb_ln_g", C.c_void_p), ("emb_ln_b", C.c_void_p), ("blocks", C.POINTER(BlockWeights)),
                ("proj1_w", C.c_void_p), ("proj1_b", C.c_void_p), ("proj2_w", C.c_void_p)]


class BertCfg(C.Structure):
    _fields_ = [("enc", EncoderCfg), ("vocab", C.c_int32), ("max_pos", C.c_int32), ("pool", C.c_int32),
                ("proj_hidden", C.c_int32), ("out_dim", C.c_int32)]


class WordPieceVocab(C.Structure):
    _fields_ = [("d_slots", C.c_void_p), ("d_pool", C.c_void_p), ("n_slots", C.c_uint32), ("unk_id", C.c_int32), ("cls_id", C.c_int32),
                ("sep_id", C.c_int32), ("pad_id", C.c_int32), ("lower", C.c_int32), ("max_word_chars", C.c_int32), ("d_unicode", C.c_void_p)]


class ClipBpeVocab(C.Structure):
    _fields_ = [("d_slots", C.c_void_p), ("d_byte_id", C.c_void_p), ("d_byte_end_id", C.c_void_p), ("n_slots", C.c_uint32),
                ("sot_id", C.c_int32), ("eot_id", C.c_int32), ("lower", C.c_int32), ("d_unicode", C.c_void_p)]


class SentencePieceVocab(C.Structure):
    _fields_ = [("d_slots", C.c_void_p), ("d_pool", C.c_void_p), ("d_score", C.c_void_p), ("d_nmap", C.c_void_p), ("d_npool", C.c_void_p),
                ("d_ccc", C.c_void_p), ("n_slots", C.c_uint32), ("unk_id", C.c_int32), ("unk_score", C.c_float), ("add_dummy_prefix", C.c_int32),
                ("remove_extra_ws", C.c_int32), ("max_piece_bytes", C.c_int32), ("prefix_id", C.c_int32), ("suffix_id", C.c_int32),
                ("pad_id", C.c_int32), ("id_offset", C.c_int32), ("unk_out", C.c_int32)]


class QueueCfg(C.Structure):
    """mq_queue_cfg (ABI 14): the native request queue of a text tower (csrc/queue.hip)"""
    _fields_ = [("kind", C.c_int32), ("device", C.c_int32), ("max_seqs", C.c_int32), ("max_rows", C.c_int32), ("normalize", C.c_int32),
                ("depth", C.c_int32), ("window_us", C.c_int32), ("graphs", C.c_int32), ("helper_seqs", C.c_int32), ("reserved", C.c_int32)]


class QueueStats(C.Structure):
    _fields_ = [("requests", C.c_uint64), ("calls", C.c_uint64), ("merged_calls", C.c_uint64), ("failed_calls", C.c_uint64),
                ("sequences", C.c_uint64), ("rows", C.c_uint64), ("max_call_sequences", C.c_uint64), ("graphs", C.c_uint64), ("graph_replays", C.c_uint64)]


QUEUE_CLIP_TEXT, QUEUE_BERT, QUEUE_IMAGE_F32 = 0, 1, 2

_P = C.c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)
    "mq_abi_version": (C.c_int, []),
    "mq_last_error": (C.c_char_p, []),
    "mq_build_arch": (C.c_char_p, []),
    "mq_encoder_workspace_bytes": (C.c_size_t, [C.POINTER(EncoderCfg), C.c_int64, C.c_int64]),
    "mq_vit_workspace_bytes": (C.c_size_t, [C.POINTER(VitCfg), C.c_int64]),
    "mq_clip_text_workspace_bytes": (C.c_size_t, [C.POINTER(ClipTextCfg), C.c_int64, C.c_int64]),
    "mq_bert_workspace_bytes": (C.c_size_t, [C.POINTER(BertCfg), C.c_int64, C.c_int64]),
    "mq_encode_image_u8": (C.c_int, [C.POINTER(VitCfg), C.POINTER(VitWeights), _P, C.c_int64, _P, C.c_int, _P, C.c_size_t, _P]),
    "mq_encode_image_f32": (C.c_int, [C.POINTER(VitCfg), C.POINTER(VitWeights), _P, C.c_int64, _P, C.c_int, _P, C.c_size_t, _P]),
    "mq_encode_clip_text": (C.c_int, [C.POINTER(ClipTextCfg), C.POINTER(ClipTextWeights), _P, _P, _P, C.c_int64, _P, _P,
                                      C.c_int, _P, C.c_size_t, _P]),
    "mq_encode_bert": (C.c_int, [C.POINTER(BertCfg), C.POINTER(BertWeights), _P, _P, _P, C.c_int64, _P, C.c_int, _P,
                                 C.c_size_t, _P]),
    "mq_gemm_bf16": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                               C.c_int, _P]),
    "mq_gemm_small_bf16": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, _P]),
    "mq_ln_gemm_small_bf16": (C.c_int, [_P, C.c_int64, C.c_int, _P, _P, C.c_float, _P, C.c_int64, _P, _P, C.c_int64, C.c_int64, C.c_int64,
                                        C.c_int64, C.c_int, _P, _P]),
    "mq_gemm_bf16_ln": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, _P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, _P]),
    "mq_row_stats": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_float, _P]),
    "mq_row_stats_finalize": (C.c_int, [_P, C.c_int32, _P, C.c_int64, C.c_int32, C.c_float, _P]),
    "mq_gemm_bf16_lnrs": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, _P, _P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, _P, _P]),
    "mq_attention_stats": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int64, _P]),
    "mq_attention_proj_ok": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "mq_panel_gemm_ln_ok": (C.c_int, [C.c_int64, C.c_int32, C.c_int64, C.c_int64]),
    "mq_panel_gemm_ln": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, C.c_int64, C.c_int32, C.c_int64, C.c_int64, C.c_int, _P]),
    "mq_attention_proj": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_float, _P, C.c_size_t, _P, C.c_size_t, _P]),
    "mq_gemm_bf16_rs": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, _P, _P]),
    "mq_check_device": (C.c_int, [C.c_int]),
    "mq_gemm_band_counters": (C.c_int64, [C.c_int64]),
    "mq_gemm_bf16_rsf": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, _P, _P, C.c_float, _P,
                                   _P, C.c_size_t, _P, C.c_size_t, _P]),
    "mq_gemm_fp8": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, C.c_int, _P, _P, _P, _P, C.c_int64, _P, _P, C.c_int64, C.c_int64,
                              C.c_int64, C.c_int, _P]),
    "mq_quantize_weights_fp8": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, C.c_int64, C.c_int64, _P]),
    "mq_layernorm_fp8": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, C.c_int32, C.c_float, _P]),
    "mq_layernorm_fp8_ex": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, C.c_int64, C.c_int32, C.c_float, _P]),
    "mq_rowquant_fp8": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, _P]),
    "mq_layernorm": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, C.c_int32, C.c_float, _P]),
    "mq_layernorm_ex": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, C.c_int64, C.c_int32, C.c_float, _P]),
    "mq_attention_ex": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    "mq_attention_bias": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P]),
    "mq_attention": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "mq_encoder_forward": (C.c_int, [C.POINTER(EncoderCfg), C.POINTER(BlockWeights), _P, C.c_int64, _P, C.c_int64,
                                     C.c_int32, C.c_int32, _P, C.c_size_t, _P]),
    "mq_encoder_forward_rows": (C.c_int, [C.POINTER(EncoderCfg), C.POINTER(BlockWeights), _P, C.c_int64, _P, C.c_int64,
                                          C.c_int32, C.c_int32, _P, C.c_int64, _P, C.c_size_t, _P]),
    "mq_l2_normalize": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P]),
    "mq_clip_resize_workspace_bytes": (C.c_size_t, [_P, _P, C.c_int64, C.c_int32]),
    "mq_clip_resize_crop_u8": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, _P, _P, C.c_size_t, _P]),
    "mq_resize_workspace_bytes": (C.c_size_t, [_P, _P, C.c_int64, C.c_int32, C.c_int32]),
    "mq_resize_u8": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P, C.c_size_t, _P]),
    "mq_resize_filter_workspace_bytes": (C.c_size_t, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "mq_resize_filter_u8": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_size_t, _P]),
    "mq_resize_mode_workspace_bytes": (C.c_size_t, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "mq_resize_mode_u8": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_size_t, _P]),
    "mq_chunk_grid_count": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "mq_chunk_grid_workspace_bytes": (C.c_size_t, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "mq_chunk_grid_u8": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P,
                                   C.c_size_t, _P]),
    "mq_host_gather": (C.c_int, [_P, _P, _P, C.c_int64, _P, C.c_int32]),
    "mq_host_gather_checked": (C.c_int, [_P, _P, _P, C.c_int64, _P, C.c_int64, C.c_int32]),
    "mq_unpack_rgbx": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int64, _P, _P]),
    "mq_to_tensor_normalize": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float), _P]),
    "mq_resample_ksize": (C.c_int, [C.c_int32, C.c_int32]),
    "mq_resample_coeffs": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "mq_tokenize_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32]),
    "mq_tokenize_wordpiece": (C.c_int, [C.POINTER(WordPieceVocab), _P, _P, C.c_int64, C.c_int64, C.c_int32, _P, C.c_int64, _P, _P, _P,
                                        C.c_size_t, _P]),
    "mq_tokenize_clip_bpe": (C.c_int, [C.POINTER(ClipBpeVocab), _P, _P, C.c_int64, C.c_int64, C.c_int32, _P, _P, _P, _P, C.c_size_t, _P]),
    "mq_tokenize_sentencepiece_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32]),
    "mq_tokenize_sentencepiece": (C.c_int, [C.POINTER(SentencePieceVocab), _P, _P, C.c_int64, C.c_int64, C.c_int32, _P, C.c_int64, _P, _P, _P,
                                            C.c_size_t, _P]),
    "mq_pack_ids": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, _P]),
    "mq_weighted_combine": (C.c_int, [_P, C.c_int64, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P]),
    "mq_queue_create": (C.c_int, [C.POINTER(QueueCfg), _P, _P, C.POINTER(_P)]),
    "mq_queue_encode": (C.c_int, [_P, _P, _P, C.c_int64, _P]),
    "mq_queue_encode_images": (C.c_int, [_P, _P, C.c_int64, _P]),
    "mq_queue_get_stats": (C.c_int, [_P, C.POINTER(QueueStats)]),
    "mq_queue_destroy": (C.c_int, [_P]),
    "mq_tune": (C.c_int, [C.c_char_p, C.c_int]),
    "mq_profile_enable": (C.c_int, [C.c_int]),
    "mq_profile_collect": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "mq_probe_mfma_peak": (C.c_int, [C.c_double, _P, C.c_int64, C.POINTER(C.c_double), C.POINTER(C.c_double), _P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None
_lib_lock = threading.Lock()


def sources() -> list:
    return sorted(str(p) for p in CSRC_DIR.glob("*.hip"))


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile csrc/*.hip for gfx950 into lib/libmarqo_hip.so (hipcc cross-compiles without a GPU).
    One object per source, compiled in parallel and cached by mtime under csrc/.obj/, then linked."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = sources()
    common = sorted(str(p) for p in CSRC_DIR.glob("*.h")) + [str(HEADER_PATH)]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    obj_dir = CSRC_DIR / ".obj"
    obj_dir.mkdir(parents=True, exist_ok=True)
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    common_mtime = max(os.path.getmtime(p) for p in common)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
             "-Rpass-analysis=kernel-resource-usage"]  # the remarks feed the no-scratch check below

    def compile_one(src: str):
        obj = obj_dir / (Path(src).stem + ".o")
        if not force and obj.exists() and os.path.getmtime(obj) >= max(os.path.getmtime(src), common_mtime):
            return str(obj), False
        cmd = [hipcc, *flags, "-c", src, "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise MarqoHipUnavailableError(f"hipcc failed on {src} ({res.returncode}):\n{res.stdout}\n{res.stderr}")
        # The LDS-DMA / MFMA kernels must not use scratch: they sit at the 256-VGPR cap of 2 workgroups per CU, and a variant
        # that spilled (fp8 GEMM, 192-row tile with a whole-tile residual prefetch) returned wrong tiles on MI355X.
        spilled = [ln.strip() for ln in res.stderr.splitlines()
                   if any(int(v) > 0 for v in re.findall(r"(?:VGPRs Spill|ScratchSize \[bytes/lane\]): (\d+)", ln))]
        if spilled and Path(src).stem in NO_SCRATCH_UNITS:
            os.unlink(obj)
            raise MarqoHipUnavailableError(f"{src}: a kernel spills registers to scratch (not allowed, see _lib.build):\n" + "\n".join(spilled[:6]))
        return str(obj), True

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(compile_one, srcs))
    objs = [o for o, _ in results]
    if any(changed for _, changed in results) or not LIB_PATH.exists() or force:
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", str(LIB_PATH)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise MarqoHipUnavailableError(f"link failed ({res.returncode}):\n{res.stdout}\n{res.stderr}")
    return LIB_PATH


def build_torch_ops(force: bool = False, verbose: bool = False) -> Path:
    """Compile csrc/torch_ops.cpp (host code only: TORCH_LIBRARY registrations that forward to the C ABI on PyTorch's current HIP
    stream) with g++ against the installed torch headers and link it to lib/libmarqo_hip.so -> lib/libmarqo_torch_ops.so."""
    import torch
    deps = [str(TORCH_OPS_SRC), str(HEADER_PATH)]
    if not force and TORCH_OPS_PATH.exists() and os.path.getmtime(TORCH_OPS_PATH) >= max(os.path.getmtime(p) for p in deps):
        return TORCH_OPS_PATH
    if not LIB_PATH.exists():
        raise MarqoHipUnavailableError(f"{LIB_PATH} must be built before the torch ops library")
    ti = Path(torch.__file__).resolve().parent
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", f"-I{sysconfig.get_paths()['include']}", f"-I{ti}/include",
           f"-I{ti}/include/torch/csrc/api/include",
           f"-I{rocm}/include", str(TORCH_OPS_SRC), "-o", str(TORCH_OPS_PATH), f"-L{ti}/lib", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip",
           "-ltorch_hip", f"-L{LIB_DIR}", "-lmarqo_hip", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{ti}/lib"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise MarqoHipUnavailableError(f"g++ failed on {TORCH_OPS_SRC} ({res.returncode}):\n{res.stdout}\n{res.stderr[-4000:]}")
    return TORCH_OPS_PATH


def build_stage(force: bool = False, verbose: bool = False) -> Path:
    """Compile csrc/py_stage.cpp (host code only: CPython C API + the Arrow C data interface) -> lib/_mq_stage.so"""
    deps = [STAGE_SRC, CSRC_DIR / "copy_pool.h"]
    if not force and STAGE_PATH.exists() and os.path.getmtime(STAGE_PATH) >= max(os.path.getmtime(d) for d in deps):
        return STAGE_PATH
    LIB_DIR.mkdir(parents=True, exist_ok=True)
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", f"-I{sysconfig.get_paths()['include']}",
           str(STAGE_SRC), "-o", str(STAGE_PATH), "-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise MarqoHipUnavailableError(f"g++ failed on {STAGE_SRC} ({res.returncode}):\n{res.stdout}\n{res.stderr[-4000:]}")
    return STAGE_PATH


_stage = None
_stage_tried = False


def load_stage():
    """The `_mq_stage` extension module, or None when it was not built / MARQO_AMD_NATIVE_STAGE=0.  It is a host-side accelerator of the
    Pillow -> pinned-buffer copy with byte-identical results (the GPU path is the same either way), so unlike the HIP library its absence
    is not an error: PackedImages then exports image by image through pyarrow, as before."""
    global _stage, _stage_tried
    if _stage_tried:
        return _stage
    with _lib_lock:
        if _stage_tried:
            return _stage
        mod = None
        if os.environ.get("MARQO_AMD_NATIVE_STAGE", "1") != "0" and STAGE_PATH.exists():
            import importlib.machinery
            import importlib.util
            try:
                loader = importlib.machinery.ExtensionFileLoader("_mq_stage", str(STAGE_PATH))
                spec = importlib.util.spec_from_loader("_mq_stage", loader)
                mod = importlib.util.module_from_spec(spec)
                loader.exec_module(mod)
            except (ImportError, OSError) as e:
                import logging
                logging.getLogger(__name__).warning("cannot load %s (%s): Pillow images are staged one by one", STAGE_PATH, e)
                mod = None
        _stage, _stage_tried = mod, True
        return _stage


_ops = None
_warned_no_ops = False


def boundary() -> str:
    """'torch_ops' (default): the towers call torch.ops.marqo_hip.* (PyTorch-ROCm custom ops over the C ABI); 'ctypes': they call
    the C ABI directly (MARQO_AMD_BOUNDARY=ctypes, non-torch hosts, and whenever MARQO_AMD_LIB points at a diagnostic build, which
    the ops library is not linked against)."""
    b = os.environ.get("MARQO_AMD_BOUNDARY")
    if b not in (None, "torch_ops", "ctypes"):
        raise ValueError(f"MARQO_AMD_BOUNDARY must be 'torch_ops' or 'ctypes', got {b!r}")
    if os.environ.get("MARQO_AMD_LIB"):
        return "ctypes"
    if b is None:
        # default: the custom ops — unless only the C-ABI library was built (a host without a C++ toolchain for torch_ops.cpp): the
        # same kernels are then reached through the direct binding (both are native paths; the loud failure is a missing libmarqo_hip.so)
        if not TORCH_OPS_PATH.exists():
            global _warned_no_ops
            if not _warned_no_ops:
                _warned_no_ops = True
                import logging
                logging.getLogger(__name__).warning("%s not built: the towers call the C ABI through ctypes (MARQO_AMD_BOUNDARY=ctypes)", TORCH_OPS_PATH)
            return "ctypes"
        return "torch_ops"
    return b


def load_torch_ops():
    """Register torch.ops.marqo_hip.* (after libmarqo_hip.so itself, so both resolve to the one mapped copy) and return the namespace.
    Missing library -> MarqoHipUnavailableError: like the C ABI itself, the custom ops have no fallback."""
    global _ops
    if _ops is not None:
        return _ops
    load()
    with _lib_lock:
        if _ops is not None:
            return _ops
        import torch
        if not TORCH_OPS_PATH.exists():
            raise MarqoHipUnavailableError(
                f"{TORCH_OPS_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(or run with MARQO_AMD_BOUNDARY=ctypes to call the C ABI directly).")
        try:
            torch.ops.load_library(str(TORCH_OPS_PATH))
        except OSError as e:
            raise MarqoHipUnavailableError(f"cannot load {TORCH_OPS_PATH}: {e}") from e
        ops = torch.ops.marqo_hip
        if ops.abi_version() != ABI_VERSION:
            raise MarqoHipUnavailableError(f"ABI version mismatch: torch ops library {ops.abi_version()} != binding {ABI_VERSION}")
        _ops = ops
        return _ops


def struct_blob(st):
    """CPU uint8 tensor ALIASING a ctypes struct (zero copy: later edits of the struct — fp8 policy, scales — are seen by the ops):
    how the POD descriptors of the C ABI travel through torch.ops.marqo_hip.*"""
    import torch
    return torch.frombuffer(st, dtype=torch.uint8)


def load():
    """Load libmarqo_hip.so (after torch, so both share torch's HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        import torch  # noqa: F401  (must be first: pins libamdhip64.so.7 to the copy PyTorch ships)
        if not LIB_PATH.exists():
            raise MarqoHipUnavailableError(
                f"{LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"(hipcc --offload-arch=gfx950). There is no CPU fallback for the marqo_amd engine.")
        try:
            lib = C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL)
        except OSError as e:
            raise MarqoHipUnavailableError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise MarqoHipUnavailableError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        if lib.mq_abi_version() != ABI_VERSION:
            raise MarqoHipUnavailableError(f"ABI version mismatch: library {lib.mq_abi_version()} != binding {ABI_VERSION}")
        _lib = lib
        note_cpu_quota_once()
        return _lib


def cpu_quota(root: str = "/sys/fs/cgroup") -> Optional[float]:
    """CPUs' worth of time the container grants this process (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us), None when
    unlimited or unknown"""
    try:
        with open(os.path.join(root, "cpu.max")) as f:
            quota, period = f.read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        with open(os.path.join(root, "cpu", "cpu.cfs_quota_us")) as f:
            quota = float(f.read())
        with open(os.path.join(root, "cpu", "cpu.cfs_period_us")) as f:
            period = float(f.read())
        return None if quota <= 0 or period <= 0 else quota / period
    except (OSError, ValueError):
        return None


def oversubscription_note(torch_threads: int, quota: Optional[float]) -> Optional[str]:
    """Advice for the operator when PyTorch's intra-op pool is much larger than the container's CPU quota (the engine itself keeps
    PyTorch CPU kernels off the request path, DESIGN.md §6.5 "host bookkeeping"; the host's OTHER torch CPU work — the reference's own
    CPU models, its post-processing — would still wake the whole pool)."""
    if quota is None or torch_threads <= max(4.0, 2.0 * quota):
        return None
    return (f"torch.get_num_threads() = {torch_threads} but the container's CPU quota is {quota:g} CPUs: an OpenMP region of that size keeps "
            f"spinning after small jobs and gets the whole process throttled; consider torch.set_num_threads({max(1, int(quota))}) or "
            f"OMP_NUM_THREADS={max(1, int(quota))} in this service")


_quota_checked = False


def note_cpu_quota_once() -> None:
    global _quota_checked
    if _quota_checked:
        return
    _quota_checked = True
    try:
        import torch
        note = oversubscription_note(torch.get_num_threads(), cpu_quota())
    except Exception:  # noqa: BLE001 - advice only
        return
    if note:
        import logging
        logging.getLogger(__name__).warning(note)


def check(rc: int, what: str = "libmarqo_hip") -> None:
    if rc != MQ_OK:
        msg = load().mq_last_error()
        raise MarqoHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def ptr(t) -> int:
    """Device (or host) address of a tensor / None -> NULL."""
    return 0 if t is None else t.data_ptr()


def current_stream_handle(device=None) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream
