// Shared device/host helpers for libmarqo_hip (gfx950 only: wave64, MFMA, 160 KB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include "../../include/marqo_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t bf16_t;  // storage type; arithmetic always goes through fp32

#define MQ_WAVE 64

// ---- bf16 <-> fp32 (round-to-nearest-even, NaN preserved) -----------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
    return __uint_as_float(((uint32_t)v) << 16);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
// two fp32 -> packed bf16x2 (RNE).  Written as a native __bf16 vector conversion so hipcc emits one
// v_cvt_pk_bf16_f32 instead of ~12 integer ops (the hand-rolled rounding above is kept for scalars).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2_t f = {lo, hi};
    const bf16x2_t v = __builtin_convertvector(f, bf16x2_t);
    return __builtin_bit_cast(uint32_t, v);
}

// ---- activations (fp32) ------------------------------------------------------------------
// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32-roundoff class): 1 v_rcp + 1 v_exp +
// 6 FMAs instead of libm erff's ~40-instruction branchy polynomial — the GELU epilogue of the fc1 GEMM
// was costing as much as its MFMA main loop at K = 768.
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float y = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(y, x);
}
// erf-GELU without transcendentals: gelu(x) = x * Phi(x), Phi(x) = 0.5 + xc * Q(xc^2) with xc = clamp(x, -4.5, 4.5) and Q a degree-8
// weighted-minimax polynomial (tools/fit_gelu.py regenerates the coefficients).  |error| <= 7e-5 absolute in fp32 Horner form over all
// x (|relative| <= 1e-4 for x > 0.5; beyond the clamp Phi stays at Phi(+-4.5) = 1 - 3.4e-6 / 3.4e-6), an order of magnitude below
// the bf16 rounding of the value it feeds.  11 VALU operations, all of them packable two lanes-elements at a time (v_pk_mul_f32 /
// v_pk_fma_f32), against 1 v_rcp + 1 v_exp + ~15 VALU of the Abramowitz-Stegun form above: the GELU was 6.8 k of the fc1 tile
// epilogue's 12.4 k cycles (profiles/r01d_gemm_phase_trace.txt) and the only place where the tower GEMMs trailed the vendor
// library's bias-only kernels (profiles/r02g_vendor_gemm_calibration.txt).  fast_erf stays for callers that need erf itself.
__device__ __forceinline__ f32x2_t gelu_erf2(f32x2_t x) {
#ifdef MQ_GELU_AS   // A/B build (tools/probes/build_gelu_ab.sh): the Abramowitz-Stegun form this replaced
    return f32x2_t{0.5f * x[0] * (1.0f + fast_erf(x[0] * 0.70710678118654752440f)), 0.5f * x[1] * (1.0f + fast_erf(x[1] * 0.70710678118654752440f))};
#endif
    f32x2_t xc = {__builtin_amdgcn_fmed3f(x[0], -4.5f, 4.5f), __builtin_amdgcn_fmed3f(x[1], -4.5f, 4.5f)};
    const f32x2_t s = xc * xc;
    f32x2_t q = {3.036425686e-11f, 3.036425686e-11f};
    q = __builtin_elementwise_fma(q, s, (f32x2_t){-3.314666682e-09f, -3.314666682e-09f});
    q = __builtin_elementwise_fma(q, s, (f32x2_t){1.590999455e-07f, 1.590999455e-07f});
    q = __builtin_elementwise_fma(q, s, (f32x2_t){-4.451734438e-06f, -4.451734438e-06f});
    q = __builtin_elementwise_fma(q, s, (f32x2_t){8.142522511e-05f, 8.142522511e-05f});
    q = __builtin_elementwise_fma(q, s, (f32x2_t){-1.037591685e-03f, -1.037591685e-03f});
    q = __builtin_elementwise_fma(q, s, (f32x2_t){9.585204319e-03f, 9.585204319e-03f});
    q = __builtin_elementwise_fma(q, s, (f32x2_t){-6.598465809e-02f, -6.598465809e-02f});
    q = __builtin_elementwise_fma(q, s, (f32x2_t){3.987126077e-01f, 3.987126077e-01f});
    const f32x2_t ph = __builtin_elementwise_fma(xc, q, (f32x2_t){0.5f, 0.5f});
    return x * ph;
}
__device__ __forceinline__ float gelu_erf(float x) {
    return gelu_erf2((f32x2_t){x, x})[0];
}
__device__ __forceinline__ f32x4 gelu_erf4(f32x4 v) {
    const f32x2_t lo = gelu_erf2((f32x2_t){v[0], v[1]}), hi = gelu_erf2((f32x2_t){v[2], v[3]});
    return f32x4{lo[0], lo[1], hi[0], hi[1]};
}
// x * sigmoid(1.702 x) (open_clip QuickGELU, the OpenAI CLIP checkpoints).  The quotient goes through v_rcp_f32 (1 ulp) instead of an IEEE
// division (v_div_scale x 2 + v_rcp + 6 FMAs + v_div_fmas + v_div_fixup per element: 80 elements per lane and fc1 tile); the value is rounded to bf16
// or e4m3 right after.  exp overflow (x << 0) gives rcp(inf) = 0 -> -0, the limit of the function.
__device__ __forceinline__ float quick_gelu(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
}

__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }   // (as quick_gelu: v_rcp_f32, rounded to bf16 right after)

// ---- XCD-banded block order -----------------------------------------------------------------
// Block b of a launch runs on XCD b % 8 (observed placement, used for speed only).  The tiled GEMMs give XCD k the k-th contiguous
// band of output tiles, i.e. of activation ROWS; the row-wise kernels between them (LayerNorm, attention) use the same banding, so the
// rows one XCD writes are the rows the same XCD reads in the next launch and find them in its own L2 (a kernel boundary writes dirty
// lines back but keeps them) instead of fetching them from the Infinity Cache behind another die's L2.  Bijective for any grid size.
// Run-time knobs (mq_tune, test / bench only) are relaxed atomics: request threads read them in their launch code while a test thread may set them
typedef std::atomic<int> mq_knob;
extern mq_knob mq_xcd_band;   // runtime.hip: mq_tune("xcd_band", 0 | 1)
__device__ __forceinline__ unsigned xcd_banded_block(unsigned b, unsigned nb, int on) {
    if (!on) return b;
    const unsigned q = nb >> 3, r = nb & 7, xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---- wave reductions ---------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// (mean, rstd) of a row from its (sum, sum of squares): ONE expression for every kernel that finalises row statistics (row_stats_finalize_kernel and the
// residual GEMM's in-launch finalise must give the same bits); the explicit fma is the contraction hipcc had picked for the stand-alone kernel
__device__ __forceinline__ float2 mq_finalize_stats(float s1, float s2, float inv_w, float eps) {
    const float mean = s1 * inv_w;
    const float var = __builtin_fmaf(-mean, mean, s2 * inv_w);
    return make_float2(mean, rsqrtf(fmaxf(var, 0.f) + eps));
}

// ---- host-side error plumbing --------------------------------------------------------------
void mq_set_error(const char* fmt, ...);

#define MQ_CHECK_ARG(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            mq_set_error(__VA_ARGS__);          \
            return MQ_ERR_INVALID;              \
        }                                       \
    } while (0)

#define MQ_CHECK_LAUNCH(name)                                                       \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            mq_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));    \
            return MQ_ERR_HIP;                                                      \
        }                                                                           \
    } while (0)

#define MQ_CHECK_HIP(expr)                                                          \
    do {                                                                            \
        hipError_t e__ = (expr);                                                    \
        if (e__ != hipSuccess) {                                                    \
            mq_set_error("%s: %s", #expr, hipGetErrorString(e__));                  \
            return MQ_ERR_HIP;                                                      \
        }                                                                           \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute of a kernel: `done` holds one bit per device ordinal, so a
// process that serves several GPUs ('cuda:N' device strings) sets it once per (kernel instantiation, device); concurrent request
// threads may race to set it (idempotent), never skip it.
static inline hipError_t mq_ensure_dyn_lds(const void* fn, int bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

#define MQ_TRY(expr)                 \
    do {                             \
        int rc__ = (expr);           \
        if (rc__ != MQ_OK) return rc__; \
    } while (0)

// ---- per-family launch timing (see mq_profile_* in marqo_hip.h) -----------------------------
struct MqProfScope {
    MqProfScope(int family, hipStream_t s, double flops = 0.0);
    ~MqProfScope();
    int family_; hipStream_t stream_; hipEvent_t start_; bool on_;
};

// ---- shared row LayerNorm body: one wave64 per row, CH float4 chunks per lane -----------------
// v[] holds the row (already loaded, possibly summed with other rows); returns normalised values in v.
template <int CH>
__device__ __forceinline__ void ln_normalize_row(f32x4 (&v)[CH], int lane, int nch, int W, float eps) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i)
        if (lane + i * 64 < nch) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    const float mean = wave_sum(s) / (float)W;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i)
        if (lane + i * 64 < nch) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; s2 += d * d; }
        }
    const float rstd = rsqrtf(wave_sum(s2) / (float)W + eps);
#pragma unroll
    for (int i = 0; i < CH; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = (v[i][e] - mean) * rstd;
}

// dispatch a kernel template on the number of float4 chunks per lane (W <= 2048)
#define MQ_DISPATCH_CH(W, CALL)                                   \
    do {                                                          \
        const int ch__ = ((W) / 4 + 63) / 64;                     \
        switch (ch__) {                                           \
            case 1: { constexpr int CH = 1; CALL; } break;        \
            case 2: { constexpr int CH = 2; CALL; } break;        \
            case 3: { constexpr int CH = 3; CALL; } break;        \
            case 4: { constexpr int CH = 4; CALL; } break;        \
            case 5: case 6: { constexpr int CH = 6; CALL; } break; \
            default: { constexpr int CH = 8; CALL; } break;       \
        }                                                         \
    } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
