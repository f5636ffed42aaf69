"""torch.ops.marqo_hip.* — the PyTorch-ROCm custom-op face of the C ABI (csrc/torch_ops.cpp).  CPU half: the ops library loads without a
GPU, registers every op with the schema the host code calls, is GPU-only by construction (no CPU kernel to fall back to) and carries the
POD descriptors of include/marqo_hip.h as zero-copy byte tensors."""
import ctypes as C

import pytest
import torch

from marqo_amd import _lib as L

OPS = ("encode_image_u8", "encode_image_f32", "encode_clip_text", "encode_bert", "clip_resize_crop_u8", "clip_resize_workspace_bytes",
       "gemm_bf16", "layernorm", "attention", "l2_normalize", "abi_version")


def test_ops_library_loads_and_registers_every_op():
    ops = L.load_torch_ops()
    assert ops.abi_version() == L.ABI_VERSION == L.load().mq_abi_version()
    for name in OPS:
        assert hasattr(ops, name), name
    s = str(ops.encode_image_u8.default._schema)
    assert "Tensor(a!) out" in s and "Tensor(b!) workspace" in s and "bool normalize" in s
    assert str(ops.gemm_bf16.default._schema).endswith("-> Tensor")


def test_ops_have_no_cpu_kernel():
    ops = L.load_torch_ops()
    with pytest.raises(NotImplementedError):
        ops.l2_normalize(torch.ones(2, 8))
    with pytest.raises(NotImplementedError):
        ops.gemm_bf16(torch.ones(4, 64, dtype=torch.bfloat16), torch.ones(8, 64, dtype=torch.bfloat16), None, None, 0)


def test_host_side_planning_op_matches_the_c_abi():
    ops, lib = L.load_torch_ops(), L.load()
    h = torch.tensor([480, 33, 224], dtype=torch.int32)
    w = torch.tensor([640, 900, 224], dtype=torch.int32)
    assert ops.clip_resize_workspace_bytes(h, w, 224) == lib.mq_clip_resize_workspace_bytes(h.data_ptr(), w.data_ptr(), 3, 224) > 0


def test_struct_blob_aliases_the_ctypes_struct():
    cfg = L.VitCfg(image_size=224, patch_size=32, out_dim=512)
    blob = L.struct_blob(cfg)
    assert blob.dtype == torch.uint8 and blob.numel() == C.sizeof(L.VitCfg) and blob.data_ptr() == C.addressof(cfg)
    before = blob.clone()
    cfg.enc.fp8_first_layer = 7          # a later edit of the descriptor (the fp8 policy does this) is what the ops see
    assert not torch.equal(before, blob)


def test_boundary_selection(monkeypatch):
    monkeypatch.delenv("MARQO_AMD_BOUNDARY", raising=False)
    monkeypatch.delenv("MARQO_AMD_LIB", raising=False)
    assert L.boundary() == "torch_ops"
    monkeypatch.setenv("MARQO_AMD_BOUNDARY", "ctypes")
    assert L.boundary() == "ctypes"
    monkeypatch.setenv("MARQO_AMD_BOUNDARY", "eager")
    with pytest.raises(ValueError):
        L.boundary()
    monkeypatch.setenv("MARQO_AMD_BOUNDARY", "torch_ops")
    monkeypatch.setenv("MARQO_AMD_LIB", "/tmp/libmarqo_hip_diag1.so")   # diagnostic builds are reached through ctypes only
    assert L.boundary() == "ctypes"
