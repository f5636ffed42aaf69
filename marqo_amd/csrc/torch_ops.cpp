// PyTorch-ROCm custom-op face of the C ABI: torch.ops.marqo_hip.* (north-star wording: "Python host code calls hand-written CDNA4 HIP
// kernels through PyTorch-ROCm custom ops").  Every op is a thin, allocation-free shim: it checks device / dtype / contiguity of its
// tensors, takes the CURRENT HIP stream of the tensors' device from PyTorch (so the ops compose with torch.cuda.stream(...) contexts,
// events and hipGraph capture like any aten kernel) and calls the matching extern "C" entry point of libmarqo_hip.so
// (include/marqo_hip.h).  No arithmetic lives here, and there is no CPU implementation: the ops are registered for the CUDA (= HIP on
// ROCm) dispatch key only, so a CPU tensor fails loudly in the dispatcher.
//
// The POD descriptors of the C ABI (mq_vit_cfg, mq_vit_weights, ...) travel as CPU uint8 tensors holding the struct bytes (the Python
// host builds them with ctypes, marqo_amd/_lib.py); their device pointers stay owned by the tensors the towers keep alive.
//
// Reference interface these replace: the loader methods the reference calls through duck typing —
// OPEN_CLIP.encode_image / encode_text (src/marqo/core/inference/embedding_models/open_clip_model.py:249-286) and
// HuggingFaceModel.encode (hugging_face_model.py:172-214); see INTEGRATION.md §3.
//
// Built by marqo_amd/_lib.py::build_torch_ops() with g++ against the installed torch headers (host code only: nothing here is device
// code, the kernels are in the .hip translation units).
#include <Python.h>
#include <stdlib.h>

#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include "../../include/marqo_hip.h"

namespace {

void* stream_of(const at::Tensor& t) {
    return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream();
}

// The tower ops enqueue ~100-800 kernel launches (0.5 - 2 ms of host time) and touch no Python object meanwhile.  torch's Python binding
// already drops the GIL around an operator call (so PyGILState_Check() is false here and this guard is a no-op on that route); the guard
// only matters for callers that invoke the ops from C++ while holding the GIL (MARQO_AMD_OPS_RELEASE_GIL=1).  Measured with 4 concurrent
// 256-image callers (profiles/r02t_gil_boundary_ab.txt, r02z_chain_large_calls_ab.txt): 62-73 k embeddings/s with or without it and
// through the ctypes boundary alike (run-to-run spread).
const bool g_release_gil = [] { const char* v = getenv("MARQO_AMD_OPS_RELEASE_GIL"); return v && v[0] == '1'; }();
struct NoGil {
    PyThreadState* st;
    NoGil() : st(g_release_gil && Py_IsInitialized() && PyGILState_Check() ? PyEval_SaveThread() : nullptr) {}
    ~NoGil() { if (st) PyEval_RestoreThread(st); }
    NoGil(const NoGil&) = delete;
    NoGil& operator=(const NoGil&) = delete;
};

void check_rc(int rc, const char* what) {
    TORCH_CHECK(rc == MQ_OK, "marqo_hip::", what, " failed (", rc, "): ", mq_last_error());
}

void need_dev(const at::Tensor& t, at::ScalarType dt, const char* name) {
    TORCH_CHECK(t.is_cuda(), "marqo_hip: ", name, " must live on the GPU (there is no CPU path)");
    TORCH_CHECK(t.scalar_type() == dt, "marqo_hip: ", name, " must be ", dt, ", got ", t.scalar_type());
    TORCH_CHECK(t.is_contiguous(), "marqo_hip: ", name, " must be contiguous");
}

template <typename T>
const T* blob(const at::Tensor& t, const char* name) {
    TORCH_CHECK(t.device().is_cpu() && t.scalar_type() == at::kByte && t.is_contiguous() && (size_t)t.numel() == sizeof(T),
                "marqo_hip: ", name, " must be a CPU uint8 tensor of ", sizeof(T), " bytes (the ", name, " struct of include/marqo_hip.h)");
    return (const T*)t.data_ptr();
}

// ---- towers (the hot-path entry points) -------------------------------------------------------------------------------------------
void encode_image_u8(const at::Tensor& cfg, const at::Tensor& weights, const at::Tensor& pixels, at::Tensor out, bool normalize,
                     at::Tensor workspace) {
    need_dev(pixels, at::kByte, "pixels");
    need_dev(out, at::kFloat, "out");
    need_dev(workspace, at::kByte, "workspace");
    TORCH_CHECK(pixels.dim() == 4 && out.dim() == 2 && out.size(0) == pixels.size(0), "marqo_hip::encode_image_u8: pixels [n,S,S,3], out [n,D]");
    NoGil nogil;
    check_rc(mq_encode_image_u8(blob<mq_vit_cfg>(cfg, "mq_vit_cfg"), blob<mq_vit_weights>(weights, "mq_vit_weights"),
                                (const uint8_t*)pixels.data_ptr(), pixels.size(0), (float*)out.data_ptr(), normalize ? 1 : 0,
                                workspace.data_ptr(), (size_t)workspace.numel(), stream_of(pixels)), "encode_image_u8");
}

void encode_image_f32(const at::Tensor& cfg, const at::Tensor& weights, const at::Tensor& pixels, at::Tensor out, bool normalize,
                      at::Tensor workspace) {
    need_dev(pixels, at::kFloat, "pixels");
    need_dev(out, at::kFloat, "out");
    need_dev(workspace, at::kByte, "workspace");
    TORCH_CHECK(pixels.dim() == 4 && out.dim() == 2 && out.size(0) == pixels.size(0), "marqo_hip::encode_image_f32: pixels [n,3,S,S], out [n,D]");
    NoGil nogil;
    check_rc(mq_encode_image_f32(blob<mq_vit_cfg>(cfg, "mq_vit_cfg"), blob<mq_vit_weights>(weights, "mq_vit_weights"),
                                 (const float*)pixels.data_ptr(), pixels.size(0), (float*)out.data_ptr(), normalize ? 1 : 0,
                                 workspace.data_ptr(), (size_t)workspace.numel(), stream_of(pixels)), "encode_image_f32");
}

void check_packed(const at::Tensor& ids, const at::Tensor& cu, const at::Tensor& cu_host, const at::Tensor& out) {
    need_dev(ids, at::kInt, "ids");
    need_dev(cu, at::kInt, "cu_seqlens");
    need_dev(out, at::kFloat, "out");
    TORCH_CHECK(cu_host.device().is_cpu() && cu_host.scalar_type() == at::kInt && cu_host.is_contiguous() && cu_host.numel() == cu.numel(),
                "marqo_hip: cu_seqlens_host must be the CPU int32 copy of cu_seqlens");
    TORCH_CHECK(cu.numel() >= 1 && out.dim() == 2 && out.size(0) == cu.numel() - 1, "marqo_hip: out must be [nseq, D] with nseq = len(cu_seqlens) - 1");
}

void encode_clip_text(const at::Tensor& cfg, const at::Tensor& weights, const at::Tensor& ids, const at::Tensor& cu,
                      const at::Tensor& cu_host, const c10::optional<at::Tensor>& pool_rows, at::Tensor out, bool normalize,
                      at::Tensor workspace) {
    check_packed(ids, cu, cu_host, out);
    need_dev(workspace, at::kByte, "workspace");
    const int32_t* pr = nullptr;
    if (pool_rows.has_value()) {
        need_dev(*pool_rows, at::kInt, "pool_rows");
        pr = (const int32_t*)pool_rows->data_ptr();
    }
    NoGil nogil;
    check_rc(mq_encode_clip_text(blob<mq_clip_text_cfg>(cfg, "mq_clip_text_cfg"), blob<mq_clip_text_weights>(weights, "mq_clip_text_weights"),
                                 (const int32_t*)ids.data_ptr(), (const int32_t*)cu.data_ptr(), (const int32_t*)cu_host.data_ptr(),
                                 cu.numel() - 1, pr, (float*)out.data_ptr(), normalize ? 1 : 0, workspace.data_ptr(),
                                 (size_t)workspace.numel(), stream_of(ids)), "encode_clip_text");
}

void encode_bert(const at::Tensor& cfg, const at::Tensor& weights, const at::Tensor& ids, const at::Tensor& cu, const at::Tensor& cu_host,
                 at::Tensor out, bool normalize, at::Tensor workspace) {
    check_packed(ids, cu, cu_host, out);
    need_dev(workspace, at::kByte, "workspace");
    NoGil nogil;
    check_rc(mq_encode_bert(blob<mq_bert_cfg>(cfg, "mq_bert_cfg"), blob<mq_bert_weights>(weights, "mq_bert_weights"),
                            (const int32_t*)ids.data_ptr(), (const int32_t*)cu.data_ptr(), (const int32_t*)cu_host.data_ptr(),
                            cu.numel() - 1, (float*)out.data_ptr(), normalize ? 1 : 0, workspace.data_ptr(), (size_t)workspace.numel(),
                            stream_of(ids)), "encode_bert");
}

// ---- image preprocessing (K10) ----------------------------------------------------------------------------------------------------
void clip_resize_crop_u8(const at::Tensor& packed, const at::Tensor& offsets, const at::Tensor& heights, const at::Tensor& widths,
                         int64_t S, at::Tensor out, at::Tensor workspace) {
    need_dev(packed, at::kByte, "packed");
    need_dev(out, at::kByte, "out");
    need_dev(workspace, at::kByte, "workspace");
    const int64_t n = heights.numel();
    TORCH_CHECK(offsets.device().is_cpu() && offsets.scalar_type() == at::kLong && offsets.is_contiguous() && offsets.numel() == n &&
                heights.device().is_cpu() && heights.scalar_type() == at::kInt && heights.is_contiguous() &&
                widths.device().is_cpu() && widths.scalar_type() == at::kInt && widths.is_contiguous() && widths.numel() == n,
                "marqo_hip::clip_resize_crop_u8: offsets int64 [n], heights / widths int32 [n] on the CPU (the resampling plan is host work)");
    TORCH_CHECK(out.dim() == 4 && out.size(0) == n && out.size(1) == S && out.size(2) == S && out.size(3) == 3, "marqo_hip::clip_resize_crop_u8: out uint8 [n,S,S,3]");
    NoGil nogil;
    check_rc(mq_clip_resize_crop_u8((const uint8_t*)packed.data_ptr(), (const int64_t*)offsets.data_ptr(), (const int32_t*)heights.data_ptr(),
                                    (const int32_t*)widths.data_ptr(), n, (int32_t)S, (uint8_t*)out.data_ptr(), workspace.data_ptr(),
                                    (size_t)workspace.numel(), stream_of(packed)), "clip_resize_crop_u8");
}

int64_t clip_resize_workspace_bytes(const at::Tensor& heights, const at::Tensor& widths, int64_t S) {
    TORCH_CHECK(heights.device().is_cpu() && heights.scalar_type() == at::kInt && heights.is_contiguous() && widths.device().is_cpu() &&
                widths.scalar_type() == at::kInt && widths.is_contiguous() && widths.numel() == heights.numel(), "heights / widths: CPU int32 [n]");
    return (int64_t)mq_clip_resize_workspace_bytes((const int32_t*)heights.data_ptr(), (const int32_t*)widths.data_ptr(), heights.numel(), (int32_t)S);
}

// ---- building blocks (parity tests, composition) ----------------------------------------------------------------------------------
at::Tensor gemm_bf16(const at::Tensor& A, const at::Tensor& W, const c10::optional<at::Tensor>& bias,
                     const c10::optional<at::Tensor>& residual, int64_t flags) {
    need_dev(A, at::kBFloat16, "A");
    need_dev(W, at::kBFloat16, "W");
    TORCH_CHECK(A.dim() == 2 && W.dim() == 2 && A.size(1) == W.size(1), "marqo_hip::gemm_bf16: A [M,K], W [N,K]");
    const int64_t M = A.size(0), K = A.size(1), N = W.size(0);
    const bool f32 = (flags & MQ_EPI_OUT_F32) != 0;
    const float* b = nullptr;
    const void* r = nullptr;
    if (flags & MQ_EPI_BIAS) {
        TORCH_CHECK(bias.has_value(), "marqo_hip::gemm_bf16: MQ_EPI_BIAS without bias");
        need_dev(*bias, at::kFloat, "bias");
        TORCH_CHECK(bias->numel() == N, "bias must be [N]");
        b = (const float*)bias->data_ptr();
    }
    if (flags & MQ_EPI_RESIDUAL) {
        TORCH_CHECK(residual.has_value(), "marqo_hip::gemm_bf16: MQ_EPI_RESIDUAL without residual");
        need_dev(*residual, f32 ? at::kFloat : at::kBFloat16, "residual");
        TORCH_CHECK(residual->dim() == 2 && residual->size(0) == M && residual->size(1) == N, "residual must be [M,N]");
        r = residual->data_ptr();
    }
    at::Tensor out = at::empty({M, N}, A.options().dtype(f32 ? at::kFloat : at::kBFloat16));
    check_rc(mq_gemm_bf16(A.data_ptr(), K, W.data_ptr(), K, b, (const float*)r, out.data_ptr(), N, M, N, K, (int)flags, stream_of(A)), "gemm_bf16");
    return out;
}

at::Tensor layernorm(const at::Tensor& x, const at::Tensor& gamma, const at::Tensor& beta, double eps, bool out_bf16) {
    need_dev(x, at::kFloat, "x");
    need_dev(gamma, at::kFloat, "gamma");
    need_dev(beta, at::kFloat, "beta");
    TORCH_CHECK(x.dim() == 2 && gamma.numel() == x.size(1) && beta.numel() == x.size(1), "marqo_hip::layernorm: x [rows,W], gamma / beta [W]");
    at::Tensor out = at::empty_like(x, x.options().dtype(out_bf16 ? at::kBFloat16 : at::kFloat));
    check_rc(mq_layernorm((const float*)x.data_ptr(), nullptr, (const float*)gamma.data_ptr(), (const float*)beta.data_ptr(),
                          out_bf16 ? out.data_ptr() : nullptr, out_bf16 ? nullptr : (float*)out.data_ptr(), x.size(0), (int32_t)x.size(1),
                          (float)eps, stream_of(x)), "layernorm");
    return out;
}

at::Tensor attention(const at::Tensor& qkv, const c10::optional<at::Tensor>& cu_seqlens, int64_t nseq, int64_t fixed_len, int64_t max_len,
                     int64_t heads, int64_t mask) {
    need_dev(qkv, at::kBFloat16, "qkv");
    TORCH_CHECK(qkv.dim() == 2 && qkv.size(1) % 3 == 0, "marqo_hip::attention: qkv [rows, 3W]");
    const int64_t W = qkv.size(1) / 3;
    const int32_t* cu = nullptr;
    if (cu_seqlens.has_value()) {
        need_dev(*cu_seqlens, at::kInt, "cu_seqlens");
        TORCH_CHECK(cu_seqlens->numel() == nseq + 1, "cu_seqlens must be [nseq + 1]");
        cu = (const int32_t*)cu_seqlens->data_ptr();
    }
    at::Tensor out = at::empty({qkv.size(0), W}, qkv.options());
    check_rc(mq_attention(qkv.data_ptr(), out.data_ptr(), cu, nseq, (int32_t)fixed_len, (int32_t)max_len, (int32_t)W, (int32_t)heads,
                          (int32_t)mask, stream_of(qkv)), "attention");
    return out;
}

at::Tensor l2_normalize(const at::Tensor& x) {
    need_dev(x, at::kFloat, "x");
    TORCH_CHECK(x.dim() == 2, "marqo_hip::l2_normalize: x [rows, D]");
    at::Tensor out = at::empty_like(x);
    check_rc(mq_l2_normalize((const float*)x.data_ptr(), (float*)out.data_ptr(), x.size(0), (int32_t)x.size(1), stream_of(x)), "l2_normalize");
    return out;
}

int64_t abi_version() { return mq_abi_version(); }

}  // namespace

TORCH_LIBRARY(marqo_hip, m) {
    m.def("encode_image_u8(Tensor cfg, Tensor weights, Tensor pixels, Tensor(a!) out, bool normalize, Tensor(b!) workspace) -> ()");
    m.def("encode_image_f32(Tensor cfg, Tensor weights, Tensor pixels, Tensor(a!) out, bool normalize, Tensor(b!) workspace) -> ()");
    m.def("encode_clip_text(Tensor cfg, Tensor weights, Tensor ids, Tensor cu_seqlens, Tensor cu_seqlens_host, Tensor? pool_rows, "
          "Tensor(a!) out, bool normalize, Tensor(b!) workspace) -> ()");
    m.def("encode_bert(Tensor cfg, Tensor weights, Tensor ids, Tensor cu_seqlens, Tensor cu_seqlens_host, Tensor(a!) out, bool normalize, "
          "Tensor(b!) workspace) -> ()");
    m.def("clip_resize_crop_u8(Tensor packed, Tensor offsets, Tensor heights, Tensor widths, int S, Tensor(a!) out, Tensor(b!) workspace) -> ()");
    m.def("clip_resize_workspace_bytes(Tensor heights, Tensor widths, int S) -> int", &clip_resize_workspace_bytes);
    m.def("gemm_bf16(Tensor A, Tensor W, Tensor? bias, Tensor? residual, int flags) -> Tensor");
    m.def("layernorm(Tensor x, Tensor gamma, Tensor beta, float eps, bool out_bf16) -> Tensor");
    m.def("attention(Tensor qkv, Tensor? cu_seqlens, int nseq, int fixed_len, int max_len, int heads, int mask) -> Tensor");
    m.def("l2_normalize(Tensor x) -> Tensor");
    m.def("abi_version() -> int", &abi_version);
}

// GPU tensors only: the CUDA dispatch key is the HIP one on PyTorch-ROCm
TORCH_LIBRARY_IMPL(marqo_hip, CUDA, m) {
    m.impl("encode_image_u8", &encode_image_u8);
    m.impl("encode_image_f32", &encode_image_f32);
    m.impl("encode_clip_text", &encode_clip_text);
    m.impl("encode_bert", &encode_bert);
    m.impl("clip_resize_crop_u8", &clip_resize_crop_u8);
    m.impl("gemm_bf16", &gemm_bf16);
    m.impl("layernorm", &layernorm);
    m.impl("attention", &attention);
    m.impl("l2_normalize", &l2_normalize);
}
