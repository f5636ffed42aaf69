// Shared device/host helpers for libmarqo_hip (gfx950 only: wave64, MFMA, 160 KB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
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
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

// ---- activations (fp32) ------------------------------------------------------------------
__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float quick_gelu(float x) {
    return x / (1.0f + __expf(-1.702f * x));
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

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
