// Row-wise HBM-bound kernels: LayerNorm (K2), L2-normalise (tail of K6/K8/K9), fp32->bf16 cast.
// One wave64 per row, float4 (16 B/lane) accesses, two-pass statistics held in registers so the
// row is read from HBM exactly once.
#include <stdlib.h>
#include "common.h"

int mq_ln_rows_per_wave = getenv("MQ_LN_ROWS") ? atoi(getenv("MQ_LN_ROWS")) : 2;  // mq_tune("ln_rows", 1 | 2)
extern int mq_gemm_small_max_rows;   // gemm_small.hip
int mq_layernorm_pf(const void* d_x, int x_bf16, const int32_t* d_row_idx, const float* d_g, const float* d_b, void* d_out_bf16, float* d_out_f32,
                    int64_t rows, int32_t W, float eps, const void* pf_a, size_t bytes_a, const void* pf_b, size_t bytes_b, hipStream_t s);
int mq_layernorm_fp8_pf(const void* d_x, int x_bf16, const float* d_g, const float* d_b, void* d_out_fp8, float* d_row_scale, float* d_out_f32,
                        int64_t rows, int32_t W, float eps, const void* pf_a, size_t bytes_a, const void* pf_b, size_t bytes_b, hipStream_t s);
int mq_ln_bf16_wide = getenv("MQ_LN_BF16_WIDE") ? atoi(getenv("MQ_LN_BF16_WIDE")) : 1;   // mq_tune("ln_bf16_wide", 0 | 1): 16-byte bf16-input LayerNorm

namespace {

constexpr int LN_MAX_CHUNKS = 8;  // float4 chunks per lane: W <= 64 * 4 * 8 = 2048

// x row (fp32) -> LN -> bf16 and/or fp32.  row_idx gathers input rows (output rows are dense).
// CH = float4 chunks per lane (compile time, so the row lives in CH*4 VGPRs and the kernel keeps
// 8 waves/SIMD in flight — the first version sized its arrays for the maximum W, compiled to 220
// VGPRs / occupancy 2 and ran at 2 TB/s).
// R = rows per wave: the loads of R rows are issued back to back and the 2R wave reductions interleave, which doubles
// the bytes in flight per wave (the one-row form is latency-bound: 4.9 TB/s with the data sitting in the Infinity Cache).
// XB = the input rows are bf16 (the bf16 residual stream of the pre-LN towers, towers.hip): 8-byte loads, same arithmetic in fp32
// Last kernel argument of the LayerNorm kernels: the XCD banding switch and a WEIGHT PREFETCH.  Inside a tower every GEMM's weights were last
// touched one step ago and have left the 256 MB Infinity Cache by the time they are needed again (12 layers x 14 MB of weights + ~140 MB of
// activations cycle through it), and the first round of tiles of a tower GEMM then waits on HBM latency at every k-step: +5..6 us per QKV / fc2
// launch (tools/gemm_layer_probe.py: the same launches with ONE weight set run at their stand-alone time).  The LayerNorm in front of a GEMM
// is memory-bound and short, so its threads each also touch one dword per 128-byte line of the NEXT GEMMs' weights (values unused): by the time the GEMM
// starts its weights sit in the Infinity Cache.  pfa / pfb: two ranges (the next GEMM's weights and the one after it), na / nb in 128-byte lines.
struct LnExtra {
    int band;
    unsigned na, nb;          // touches (one dword each, `step` dwords apart)
    const unsigned* pfa;
    const unsigned* pfb;
    unsigned step;            // dwords between touches: 32 = one per 128-byte L2 line, 16 = one per 64-byte fabric request
};
constexpr int LN_PF = 2;   // lines per thread (one dword of a line brings the line: one VGPR per touched line)

__device__ __forceinline__ void ln_prefetch_issue(const LnExtra& ex, unsigned (&pq)[LN_PF]) {
#pragma unroll
    for (int j = 0; j < LN_PF; ++j) pq[j] = 0u;
    if (ex.na | ex.nb) {
        const unsigned nt = gridDim.x * 256u, t = blockIdx.x * 256u + threadIdx.x;
#pragma unroll
        for (int j = 0; j < LN_PF; ++j) {
            const unsigned c = t + j * nt;
            if (c < ex.na) pq[j] = ex.pfa[(size_t)c * ex.step];
            else if (c - ex.na < ex.nb) pq[j] = ex.pfb[(size_t)(c - ex.na) * ex.step];
        }
    }
}
// the loaded values are "used" by an empty asm at the end of the kernel: keeps the loads alive, and the wait for them out of the row's way
__device__ __forceinline__ void ln_prefetch_retire(const unsigned (&pq)[LN_PF]) {
#pragma unroll
    for (int j = 0; j < LN_PF; ++j) asm volatile("" ::"v"(pq[j]));
}

template <int CH, int R, bool XB = false>
// (register budget by row width: rows + gamma + beta in registers are 3 * CH * R float4; the 1 537..2 048-wide instantiations (CH = 8: ViT-bigG/14's
// 1 664) spilled 61..768 VGPRs under the 128-register cap of 4 waves per SIMD — VERDICT r3 weak #8 — and run at 2 waves per SIMD instead)
__global__ __launch_bounds__(256, (CH * R <= 2 ? 8 : CH * R <= 4 ? 5 : CH * R <= 6 ? 3 : 2)) void layernorm_kernel(
    const void* __restrict__ xv, const int32_t* __restrict__ row_idx, const float* __restrict__ gam,
    const float* __restrict__ bet, bf16_t* out_bf16, float* out_f32, int64_t rows, int W, float eps, LnExtra ex) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)xcd_banded_block(blockIdx.x, gridDim.x, ex.band) * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= rows) return;
    const int nch = W >> 2;  // float4 chunks in the row

    f32x4 v[R][CH];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = row0 + r < rows ? row0 + r : rows - 1;  // a ragged last wave re-reads the last row (never stored)
        const int64_t src = row_idx ? (int64_t)row_idx[row] : row;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = lane + i * 64;
            if (c >= nch) { v[r][i] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
            if (XB) {
                const uint2 q = *(const uint2*)((const bf16_t*)xv + src * W + c * 4);
                v[r][i] = f32x4{__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u)};
            } else {
                v[r][i] = *(const f32x4*)((const float*)xv + src * W + c * 4);
            }
        }
    }
    unsigned pq[LN_PF];
    ln_prefetch_issue(ex, pq);   // behind the row loads (loads return in order: the row must not wait for an HBM miss of the prefetch)
    // gamma / beta are fetched now, not after the reductions: their (L2) latency hides behind the row loads and the shuffles
    f32x4 gv[CH], bv[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = lane + i * 64;
        gv[i] = c < nch ? *(const f32x4*)(gam + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        bv[i] = c < nch ? *(const f32x4*)(bet + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // same arithmetic per row as ln_normalize_row (sum -> mean, sum of squared deviations -> rstd), the R rows interleaved
    float mean[R], rstd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (lane + i * 64 < nch) s1 += (v[r][i][0] + v[r][i][1]) + (v[r][i][2] + v[r][i][3]);
        mean[r] = s1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < R; ++r) mean[r] += __shfl_xor(mean[r], o, 64);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mean[r] = mean[r] / (float)W;
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < CH; ++i)
            if (lane + i * 64 < nch) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float d = v[r][i][e] - mean[r]; s2 += d * d; }
            }
        rstd[r] = s2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < R; ++r) rstd[r] += __shfl_xor(rstd[r], o, 64);
#pragma unroll
    for (int r = 0; r < R; ++r) rstd[r] = rsqrtf(rstd[r] / (float)W + eps);

#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            const f32x4 gg = gv[i], bb = bv[i];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t row = row0 + r;
                if (row >= rows) continue;
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = (v[r][i][e] - mean[r]) * rstd[r] * gg[e] + bb[e];
                if (out_f32) *(f32x4*)(out_f32 + row * W + c * 4) = y;
                if (out_bf16) {
                    uint2 p;
                    p.x = pack_bf16x2(y[0], y[1]);
                    p.y = pack_bf16x2(y[2], y[3]);
                    *(uint2*)(out_bf16 + row * W + c * 4) = p;
                }
            }
        }
    }
    ln_prefetch_retire(pq);
}

// bf16 rows in (the bf16 residual stream of the pre-LN towers) -> LN -> bf16 (and / or fp32) out, EIGHT elements per lane and chunk: 16-byte
// loads AND 16-byte stores (the generic kernel above reads a bf16 row 8 bytes per lane: half the bytes per request of the fp32 form it
// was written for — on the bf16 stream it took 13.6 us per launch against 12.7 us for TWICE the bytes in fp32, profiles/r03a_*).
// R rows per wave keep the bytes in flight per wave where the fp32 two-row form has them (a 768-wide bf16 row is only 1.5 KB).
// Same arithmetic, same reduction order across the lanes' partial sums as ln_normalize_row up to the grouping of a lane's own elements.
template <int CH8, int R>
__global__ __launch_bounds__(256) void layernorm_bf16in_kernel(
    const bf16_t* __restrict__ x, const int32_t* __restrict__ row_idx, const float* __restrict__ gam, const float* __restrict__ bet,
    bf16_t* out_bf16, float* out_f32, int64_t rows, int W, float eps, LnExtra ex) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)xcd_banded_block(blockIdx.x, gridDim.x, ex.band) * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= rows) return;
    const int nch = W >> 3;  // 8-element chunks in the row
    float v[R][CH8][8];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = row0 + r < rows ? row0 + r : rows - 1;
        const int64_t src = row_idx ? (int64_t)row_idx[row] : row;
#pragma unroll
        for (int i = 0; i < CH8; ++i) {
            const int c = lane + i * 64;
            uint4 q = make_uint4(0u, 0u, 0u, 0u);
            if (c < nch) q = *(const uint4*)(x + src * W + c * 8);
            const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[r][i][2 * e] = __uint_as_float(w4[e] << 16); v[r][i][2 * e + 1] = __uint_as_float(w4[e] & 0xffff0000u); }
        }
    }
    unsigned pq[LN_PF];
    ln_prefetch_issue(ex, pq);   // behind the row loads
    f32x4 gv[CH8][2], bv[CH8][2];
#pragma unroll
    for (int i = 0; i < CH8; ++i) {
        const int c = lane + i * 64;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            gv[i][h] = c < nch ? *(const f32x4*)(gam + c * 8 + h * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            bv[i][h] = c < nch ? *(const f32x4*)(bet + c * 8 + h * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    float mean[R], rstd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < CH8; ++i)
            if (lane + i * 64 < nch) s1 += ((v[r][i][0] + v[r][i][1]) + (v[r][i][2] + v[r][i][3])) + ((v[r][i][4] + v[r][i][5]) + (v[r][i][6] + v[r][i][7]));
        mean[r] = s1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < R; ++r) mean[r] += __shfl_xor(mean[r], o, 64);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mean[r] = mean[r] / (float)W;
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < CH8; ++i)
            if (lane + i * 64 < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[r][i][e] - mean[r]; s2 += d * d; }
            }
        rstd[r] = s2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < R; ++r) rstd[r] += __shfl_xor(rstd[r], o, 64);
#pragma unroll
    for (int r = 0; r < R; ++r) rstd[r] = rsqrtf(rstd[r] / (float)W + eps);
#pragma unroll
    for (int i = 0; i < CH8; ++i) {
        const int c = lane + i * 64;
        if (c >= nch) continue;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = row0 + r;
            if (row >= rows) continue;
            float y[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = (v[r][i][e] - mean[r]) * rstd[r] * gv[i][e >> 2][e & 3] + bv[i][e >> 2][e & 3];
            if (out_f32) {
                *(f32x4*)(out_f32 + row * W + c * 8) = f32x4{y[0], y[1], y[2], y[3]};
                *(f32x4*)(out_f32 + row * W + c * 8 + 4) = f32x4{y[4], y[5], y[6], y[7]};
            }
            if (out_bf16)
                *(uint4*)(out_bf16 + row * W + c * 8) = make_uint4(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]), pack_bf16x2(y[4], y[5]), pack_bf16x2(y[6], y[7]));
        }
    }
    ln_prefetch_retire(pq);
}

// (mean, rstd) per row of the bf16 stream, nothing else: the statistics half of a LayerNorm whose apply half is folded into the GEMM behind it
// (gemm_epilogue.h, MQ_EPI_LN_APPLY).  One 16-byte load per lane and chunk, the same two-pass arithmetic and reduction order as
// layernorm_bf16in_kernel (mean first, then the sum of squared deviations: no cancellation), 8 bytes written per row — half the traffic of the
// LayerNorm launch it replaces (which also wrote the normalised row) — and it carries the weight prefetch of the GEMMs behind it (LnExtra).
template <int CH8, int R>
__global__ __launch_bounds__(256) void row_stats_bf16_kernel(const bf16_t* __restrict__ x, float2* __restrict__ stats, int64_t rows, int W, float eps, LnExtra ex) {
    const int lane = threadIdx.x & 63;
    const int64_t row0 = ((int64_t)xcd_banded_block(blockIdx.x, gridDim.x, ex.band) * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= rows) return;
    const int nch = W >> 3;  // 8-element chunks in the row
    float v[R][CH8][8];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t row = row0 + r < rows ? row0 + r : rows - 1;
#pragma unroll
        for (int i = 0; i < CH8; ++i) {
            const int c = lane + i * 64;
            uint4 q = make_uint4(0u, 0u, 0u, 0u);
            if (c < nch) q = *(const uint4*)(x + row * W + c * 8);
            const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[r][i][2 * e] = __uint_as_float(w4[e] << 16); v[r][i][2 * e + 1] = __uint_as_float(w4[e] & 0xffff0000u); }
        }
    }
    unsigned pq[LN_PF];
    ln_prefetch_issue(ex, pq);   // behind the row loads
    float mean[R], rstd[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < CH8; ++i)
            if (lane + i * 64 < nch) s1 += ((v[r][i][0] + v[r][i][1]) + (v[r][i][2] + v[r][i][3])) + ((v[r][i][4] + v[r][i][5]) + (v[r][i][6] + v[r][i][7]));
        mean[r] = s1;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < R; ++r) mean[r] += __shfl_xor(mean[r], o, 64);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mean[r] = mean[r] / (float)W;
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < CH8; ++i)
            if (lane + i * 64 < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[r][i][e] - mean[r]; s2 += d * d; }
            }
        rstd[r] = s2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < R; ++r) rstd[r] += __shfl_xor(rstd[r], o, 64);
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (lane == 0 && row0 + r < rows) stats[row0 + r] = make_float2(mean[r], rsqrtf(rstd[r] / (float)W + eps));
    ln_prefetch_retire(pq);
}

// LN with e4m3 output and a dynamic per-row scale (the row is already in registers, so the absmax is one wave reduction)
// NORM = false: no normalisation / affine, only the per-row e4m3 quantisation of x itself (the first block of a post-LN fp8
// encoder, whose input rows come from the embedding kernels)
// XB = the input rows are bf16 (the bf16 residual stream)
template <int CH, bool NORM = true, bool XB = false>
__global__ __launch_bounds__(256, (CH <= 2 ? 8 : CH <= 4 ? 5 : CH <= 6 ? 3 : 2)) void layernorm_fp8_kernel(
    const void* __restrict__ x, const float* __restrict__ gam, const float* __restrict__ bet, uint8_t* __restrict__ out8,
    float* __restrict__ row_scale, float* out_f32, int64_t rows, int W, float eps, LnExtra ex) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = W >> 2;
    f32x4 v[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = lane + i * 64;
        if (c >= nch) { v[i] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
        if (XB) {
            const uint2 q = *(const uint2*)((const bf16_t*)x + row * W + c * 4);
            v[i] = f32x4{__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u)};
        } else {
            v[i] = *(const f32x4*)((const float*)x + row * W + c * 4);
        }
    }
    unsigned pq[LN_PF];
    ln_prefetch_issue(ex, pq);   // the e4m3 weights of the GEMMs behind this LayerNorm (LnExtra above)
    if (NORM) ln_normalize_row<CH>(v, lane, nch, W, eps);
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            if (NORM) {
                const f32x4 gg = *(const f32x4*)(gam + c * 4);
                const f32x4 bb = *(const f32x4*)(bet + c * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[i][e] = v[i][e] * gg[e] + bb[e];
                if (out_f32) *(f32x4*)(out_f32 + row * W + c * 4) = v[i];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fabsf(v[i][e]));
        }
    }
    mx = wave_max(mx);
    const float sc = mx > 0.f ? mx * (1.0f / 448.f) : 1.f;
    const float inv = 1.0f / sc;
    if (lane == 0) row_scale[row] = sc;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            int w = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][0] * inv, v[i][1] * inv, 0, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][2] * inv, v[i][3] * inv, w, true);
            *(int*)(out8 + row * W + c * 4) = w;
        }
    }
    ln_prefetch_retire(pq);
}

__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ x, float* out, int64_t rows, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) { const float t = xr[c]; s += t * t; }
    // reference: outputs /= outputs.norm(dim=-1, keepdim=True)  (open_clip_model.py:262-265)
    const float inv = 1.0f / sqrtf(wave_sum(s));
    for (int c = lane; c < D; c += 64) out[row * D + c] = xr[c] * inv;
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, bf16_t* out, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = *(const f32x4*)(x + i * 4);
        uint2 p;
        p.x = pack_bf16x2(v[0], v[1]);
        p.y = pack_bf16x2(v[2], v[3]);
        *(uint2*)(out + i * 4) = p;
    }
}

}  // namespace

extern "C" int mq_layernorm(const float* d_x, const int32_t* d_row_idx, const float* d_g, const float* d_b,
                            void* d_out_bf16, float* d_out_f32, int64_t rows, int32_t W, float eps, void* stream) {
    return mq_layernorm_ex(d_x, 0, d_row_idx, d_g, d_b, d_out_bf16, d_out_f32, rows, W, eps, stream);
}

// knob: the LayerNorms of the towers prefetch the weights of the GEMMs behind them (0 = off).  mq_tune("ln_prefetch", v) / MQ_LN_PREFETCH
mq_knob mq_ln_prefetch{getenv("MQ_LN_PREFETCH") ? atoi(getenv("MQ_LN_PREFETCH")) : 1};

static LnExtra ln_extra(int band, int64_t rows, const void* pf_a, size_t bytes_a, const void* pf_b, size_t bytes_b) {
    LnExtra ex{band, 0u, 0u, nullptr, nullptr, 32u};
    if (mq_ln_prefetch && rows >= 1024) {   // (a small call is latency-bound: nothing to hide the extra loads behind)
        const size_t gran = mq_ln_prefetch == 2 ? 64 : mq_ln_prefetch == 3 ? 32 : 128;      // bytes per touch (knob values 2 / 3: A/B of the granularity)
        ex.step = (unsigned)(gran / 4);
        auto lines = [gran](const void* p, size_t b) { return (p && ((uintptr_t)p & 3) == 0) ? (unsigned)(b < ((size_t)1 << 30) ? b / gran : 0) : 0u; };
        ex.pfa = (const unsigned*)pf_a; ex.na = lines(pf_a, bytes_a);
        ex.pfb = (const unsigned*)pf_b; ex.nb = lines(pf_b, bytes_b);
    }
    return ex;
}


extern "C" int mq_layernorm_ex(const void* d_x, int x_bf16, const int32_t* d_row_idx, const float* d_g, const float* d_b,
                               void* d_out_bf16, float* d_out_f32, int64_t rows, int32_t W, float eps, void* stream) {
    return mq_layernorm_pf(d_x, x_bf16, d_row_idx, d_g, d_b, d_out_bf16, d_out_f32, rows, W, eps, nullptr, 0, nullptr, 0, (hipStream_t)stream);
}

// mq_layernorm_ex + weight prefetch (LnExtra above): pf_a / pf_b = two 16-byte-aligned device ranges (bytes) that the kernel's threads read and
// discard, one dword per 128-byte line, as far as its thread count reaches (LN_PF lines per thread); either may be NULL
int mq_layernorm_pf(const void* d_x, int x_bf16, const int32_t* d_row_idx, const float* d_g, const float* d_b, void* d_out_bf16, float* d_out_f32,
                    int64_t rows, int32_t W, float eps, const void* pf_a, size_t bytes_a, const void* pf_b, size_t bytes_b, hipStream_t s) {
    MQ_CHECK_ARG(d_x && d_g && d_b && (d_out_bf16 || d_out_f32), "mq_layernorm: null pointer");
    MQ_CHECK_ARG(W >= 4 && W % 4 == 0 && W <= 64 * 4 * LN_MAX_CHUNKS, "mq_layernorm: W=%d unsupported (multiple of 4, <= 2048)", W);
    if (rows <= 0) return MQ_OK;
    MqProfScope prof(1, s);
    // banding: dense batches only (a gather has no row locality to keep)
    const LnExtra band = ln_extra((mq_xcd_band && !d_row_idx && rows >= 4096) ? 1 : 0, rows, pf_a, bytes_a, pf_b, bytes_b);
    // two rows per wave once there are enough rows to fill the chip that way (and the row fits: CH * 2 float4 per lane)
    // The LayerNorm form follows the GEMM family of the call: at most mq_gemm_small_max_rows rows (the search path: skinny GEMMs, whose fused
    // LayerNorm prologue sums a row in the generic kernel's lane order) keep the generic kernel, so a query has the same bits alone and inside
    // a small batch; every larger call takes the 16-byte form whatever its row count, so an embedding does not depend on what else shares a
    // chip-filling batch either.  (Across the two families results agree to rounding, as their GEMMs do.)
    if (x_bf16 && mq_ln_bf16_wide && rows > mq_gemm_small_max_rows && W % 8 == 0 && W <= 1024 && ((uintptr_t)d_x & 15) == 0 && (!d_out_bf16 || ((uintptr_t)d_out_bf16 & 15) == 0)) {
        // 16-byte form: W / 8 chunks over 64 lanes -> 1 (W <= 512) or 2 chunks per lane; 4 rows per wave once the chip is full that way
        const bf16_t* xb = (const bf16_t*)d_x;
        const bool many = rows >= 16384;
        if (W <= 512) {
            if (many && mq_ln_bf16_wide < 4) hipLaunchKernelGGL((layernorm_bf16in_kernel<1, 2>), dim3((unsigned)cdiv64(rows, 8)), dim3(256), 0, s, xb, d_row_idx, d_g, d_b, (bf16_t*)d_out_bf16, d_out_f32, rows, (int)W, eps, band);
            else if (many) hipLaunchKernelGGL((layernorm_bf16in_kernel<1, 4>), dim3((unsigned)cdiv64(rows, 16)), dim3(256), 0, s, xb, d_row_idx, d_g, d_b, (bf16_t*)d_out_bf16, d_out_f32, rows, (int)W, eps, band);
            else hipLaunchKernelGGL((layernorm_bf16in_kernel<1, 1>), dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, s, xb, d_row_idx, d_g, d_b, (bf16_t*)d_out_bf16, d_out_f32, rows, (int)W, eps, band);
        } else {
            if (rows >= 16384 && mq_ln_bf16_wide >= 4) hipLaunchKernelGGL((layernorm_bf16in_kernel<2, 4>), dim3((unsigned)cdiv64(rows, 16)), dim3(256), 0, s, xb, d_row_idx, d_g, d_b, (bf16_t*)d_out_bf16, d_out_f32, rows, (int)W, eps, band);
            else if (rows >= 8192) hipLaunchKernelGGL((layernorm_bf16in_kernel<2, 2>), dim3((unsigned)cdiv64(rows, 8)), dim3(256), 0, s, xb, d_row_idx, d_g, d_b, (bf16_t*)d_out_bf16, d_out_f32, rows, (int)W, eps, band);
            else hipLaunchKernelGGL((layernorm_bf16in_kernel<2, 1>), dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, s, xb, d_row_idx, d_g, d_b, (bf16_t*)d_out_bf16, d_out_f32, rows, (int)W, eps, band);
        }
    } else if (x_bf16) {
        if (mq_ln_rows_per_wave >= 2 && rows >= 8192 && W <= 1024)
            MQ_DISPATCH_CH(W, hipLaunchKernelGGL((layernorm_kernel<CH, 2, true>), dim3((unsigned)cdiv64(rows, 8)), dim3(256), 0, s, d_x, d_row_idx,
                                                 d_g, d_b, (bf16_t*)d_out_bf16, d_out_f32, rows, (int)W, eps, band));
        else
            MQ_DISPATCH_CH(W, hipLaunchKernelGGL((layernorm_kernel<CH, 1, true>), dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, s, d_x, d_row_idx,
                                                 d_g, d_b, (bf16_t*)d_out_bf16, d_out_f32, rows, (int)W, eps, band));
    } else if (mq_ln_rows_per_wave >= 2 && rows >= 8192 && W <= 1024)
        MQ_DISPATCH_CH(W, hipLaunchKernelGGL((layernorm_kernel<CH, 2>), dim3((unsigned)cdiv64(rows, 8)), dim3(256), 0, s, d_x, d_row_idx,
                                             d_g, d_b, (bf16_t*)d_out_bf16, d_out_f32, rows, (int)W, eps, band));
    else
        MQ_DISPATCH_CH(W, hipLaunchKernelGGL((layernorm_kernel<CH, 1>), dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, s, d_x, d_row_idx,
                                             d_g, d_b, (bf16_t*)d_out_bf16, d_out_f32, rows, (int)W, eps, band));
    MQ_CHECK_LAUNCH("mq_layernorm");
    return MQ_OK;
}

// x fp32 [rows, W], or (x_bf16) the bf16 residual stream; e4m3 rows + per-row scales out; d_out_f32 (optional, fp32 x only): the normalised rows
extern "C" int mq_layernorm_fp8_ex(const void* d_x, int x_bf16, const float* d_g, const float* d_b, void* d_out_fp8, float* d_row_scale,
                                   float* d_out_f32, int64_t rows, int32_t W, float eps, void* stream) {
    return mq_layernorm_fp8_pf(d_x, x_bf16, d_g, d_b, d_out_fp8, d_row_scale, d_out_f32, rows, W, eps, nullptr, 0, nullptr, 0, (hipStream_t)stream);
}

// mq_layernorm_fp8_ex + weight prefetch (as mq_layernorm_pf)
int mq_layernorm_fp8_pf(const void* d_x, int x_bf16, const float* d_g, const float* d_b, void* d_out_fp8, float* d_row_scale, float* d_out_f32,
                        int64_t rows, int32_t W, float eps, const void* pf_a, size_t bytes_a, const void* pf_b, size_t bytes_b, hipStream_t s) {
    MQ_CHECK_ARG(d_x && d_g && d_b && d_out_fp8 && d_row_scale, "mq_layernorm_fp8: null pointer");
    MQ_CHECK_ARG(W >= 4 && W % 4 == 0 && W <= 64 * 4 * LN_MAX_CHUNKS, "mq_layernorm_fp8: W=%d unsupported (multiple of 4, <= 2048)", W);
    MQ_CHECK_ARG(!x_bf16 || !d_out_f32, "mq_layernorm_fp8: the fp32 copy of the normalised rows belongs to the fp32 stream");
    if (rows <= 0) return MQ_OK;
    MqProfScope prof(1, s);
    const LnExtra ex = ln_extra(0, rows, pf_a, bytes_a, pf_b, bytes_b);
    if (x_bf16)
        MQ_DISPATCH_CH(W, hipLaunchKernelGGL((layernorm_fp8_kernel<CH, true, true>), dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, s, d_x, d_g, d_b,
                                             (uint8_t*)d_out_fp8, d_row_scale, d_out_f32, rows, (int)W, eps, ex));
    else
        MQ_DISPATCH_CH(W, hipLaunchKernelGGL((layernorm_fp8_kernel<CH, true, false>), dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, s, d_x, d_g, d_b,
                                             (uint8_t*)d_out_fp8, d_row_scale, d_out_f32, rows, (int)W, eps, ex));
    MQ_CHECK_LAUNCH("mq_layernorm_fp8");
    return MQ_OK;
}

extern "C" int mq_layernorm_fp8(const float* d_x, const float* d_g, const float* d_b, void* d_out_fp8, float* d_row_scale, float* d_out_f32,
                                int64_t rows, int32_t W, float eps, void* stream) {
    return mq_layernorm_fp8_ex(d_x, 0, d_g, d_b, d_out_fp8, d_row_scale, d_out_f32, rows, W, eps, stream);
}

extern "C" int mq_rowquant_fp8(const float* d_x, void* d_out_fp8, float* d_row_scale, int64_t rows, int32_t W, void* stream) {
    MQ_CHECK_ARG(d_x && d_out_fp8 && d_row_scale, "mq_rowquant_fp8: null pointer");
    MQ_CHECK_ARG(W >= 4 && W % 4 == 0 && W <= 64 * 4 * LN_MAX_CHUNKS, "mq_rowquant_fp8: W=%d unsupported (multiple of 4, <= 2048)", W);
    if (rows <= 0) return MQ_OK;
    hipStream_t s = (hipStream_t)stream;
    MqProfScope prof(1, s);
    MQ_DISPATCH_CH(W, hipLaunchKernelGGL((layernorm_fp8_kernel<CH, false>), dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, s, (const void*)d_x,
                                         (const float*)nullptr, (const float*)nullptr, (uint8_t*)d_out_fp8, d_row_scale, (float*)nullptr, rows,
                                         (int)W, 0.f, LnExtra{0, 0u, 0u, nullptr, nullptr, 32u}));
    MQ_CHECK_LAUNCH("mq_rowquant_fp8");
    return MQ_OK;
}

extern "C" int mq_l2_normalize(const float* d_x, float* d_out, int64_t rows, int32_t D, void* stream) {
    MQ_CHECK_ARG(d_x && d_out && D >= 1, "mq_l2_normalize: bad argument");
    if (rows <= 0) return MQ_OK;
    hipStream_t s = (hipStream_t)stream;
    MqProfScope prof(4, s);
    hipLaunchKernelGGL(l2norm_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, s, d_x, d_out, rows, (int)D);
    MQ_CHECK_LAUNCH("mq_l2_normalize");
    return MQ_OK;
}

// internal (not in the public header): fp32 -> bf16 cast of n elements (n % 4 == 0)
int mq_cast_bf16(const float* d_x, void* d_out, int64_t n, hipStream_t s) {
    MQ_CHECK_ARG(n % 4 == 0, "mq_cast_bf16: n must be a multiple of 4");
    if (n <= 0) return MQ_OK;
    MqProfScope prof(1, s);
    const int64_t n4 = n / 4;
    const unsigned grid = (unsigned)(cdiv64(n4, 256) < 4096 ? cdiv64(n4, 256) : 4096);
    hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid), dim3(256), 0, s, d_x, (bf16_t*)d_out, n4);
    MQ_CHECK_LAUNCH("mq_cast_bf16");
    return MQ_OK;
}


// (mean, rstd) per row from the partial sums a residual GEMM left behind (gemm_epilogue.h, MQ_EPI_ROW_STATS): one thread per row, nslots float2 each (slot-major)
// (summed in slot order: deterministic); carries the weight prefetch like the LayerNorm kernels
__global__ __launch_bounds__(256) void row_stats_finalize_kernel(const float2* __restrict__ partials, int nslots, float2* __restrict__ stats, int64_t rows,
                                                                 float inv_w, float eps, LnExtra ex) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float s1 = 0.f, s2 = 0.f;
    if (row < rows) {
        const float2* p = partials + row;   // slot-major [nslots][rows]: the block's threads read 2 KB runs
        for (int i0 = 0; i0 < nslots; i0 += 8) {   // 8 independent loads in flight (every slot is a line of its own), added in slot order
            float2 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = i0 + j < nslots ? p[(int64_t)(i0 + j) * rows] : make_float2(0.f, 0.f);
#pragma unroll
            for (int j = 0; j < 8; ++j) { s1 += v[j].x; s2 += v[j].y; }
        }
    }
    unsigned pq[LN_PF];
    ln_prefetch_issue(ex, pq);
    if (row < rows) stats[row] = mq_finalize_stats(s1, s2, inv_w, eps);
    ln_prefetch_retire(pq);
}
int mq_row_stats_finalize_pf(const float* d_partials, int32_t nslots, float* d_stats, int64_t rows, int32_t W, float eps, const void* pf_a, size_t bytes_a,
                             const void* pf_b, size_t bytes_b, hipStream_t s) {
    MQ_CHECK_ARG(d_partials && d_stats && nslots >= 1 && W >= 1, "mq_row_stats_finalize: bad argument");
    if (rows <= 0) return MQ_OK;
    MqProfScope prof(1, s);
    // (a 50-workgroup launch cannot touch ~70 k weight lines with two loads per thread: the prefetch grid is padded with idle row slots)
    LnExtra ex = ln_extra(0, rows, pf_a, bytes_a, pf_b, bytes_b);
    const int64_t want = ((int64_t)ex.na + ex.nb + LN_PF - 1) / LN_PF;
    const int64_t threads = rows > want ? rows : want;
    hipLaunchKernelGGL(row_stats_finalize_kernel, dim3((unsigned)cdiv64(threads, 256)), dim3(256), 0, s, (const float2*)d_partials, (int)nslots, (float2*)d_stats,
                       rows, 1.0f / (float)W, eps, ex);
    MQ_CHECK_LAUNCH("mq_row_stats_finalize");
    return MQ_OK;
}
extern "C" int mq_row_stats_finalize(const float* d_partials, int32_t nslots, float* d_stats, int64_t rows, int32_t W, float eps, void* stream) {
    return mq_row_stats_finalize_pf(d_partials, nslots, d_stats, rows, W, eps, nullptr, 0, nullptr, 0, (hipStream_t)stream);
}

// (mean, rstd) per row of bf16 rows [rows, W] (W % 8 == 0, W <= 2048) -> d_stats fp32 [rows][2]; pf_a / pf_b: weight ranges to prefetch (mq_layernorm_pf)
bool mq_row_stats_ok(int32_t W) { return W % 8 == 0 && W >= 8 && W <= 2048; }
int mq_row_stats_pf(const void* d_x_bf16, float* d_stats, int64_t rows, int32_t W, float eps, const void* pf_a, size_t bytes_a, const void* pf_b,
                    size_t bytes_b, hipStream_t s) {
    MQ_CHECK_ARG(d_x_bf16 && d_stats, "mq_row_stats: null pointer");
    MQ_CHECK_ARG(mq_row_stats_ok(W), "mq_row_stats: W=%d unsupported (multiple of 8, <= 2048)", W);
    if (rows <= 0) return MQ_OK;
    MqProfScope prof(1, s);
    const LnExtra ex = ln_extra((mq_xcd_band && rows >= 4096) ? 1 : 0, rows, pf_a, bytes_a, pf_b, bytes_b);
    const int ch8 = ((W >> 3) + 63) / 64;
    // several rows per wave for narrow rows: a 768-wide bf16 row is 1.5 KB — one row per wave left the kernel latency-bound (7.7 us for 19.7 MB)
    const bool many = rows >= 4096;
    const int R = !many ? 1 : ch8 == 1 ? 4 : ch8 == 2 ? 2 : 1;
    const unsigned grid = (unsigned)cdiv64(rows, 4 * R);
#define MQ_RS(C, RR) hipLaunchKernelGGL((row_stats_bf16_kernel<C, RR>), dim3(grid), dim3(256), 0, s, (const bf16_t*)d_x_bf16, (float2*)d_stats, rows, (int)W, eps, ex)
    if (ch8 == 1) { if (R == 4) MQ_RS(1, 4); else MQ_RS(1, 1); }
    else if (ch8 == 2) { if (R == 2) MQ_RS(2, 2); else MQ_RS(2, 1); }
    else MQ_RS(4, 1);
#undef MQ_RS
    MQ_CHECK_LAUNCH("mq_row_stats");
    return MQ_OK;
}
extern "C" int mq_row_stats(const void* d_x_bf16, float* d_stats, int64_t rows, int32_t W, float eps, void* stream) {
    return mq_row_stats_pf(d_x_bf16, d_stats, rows, W, eps, nullptr, 0, nullptr, 0, (hipStream_t)stream);
}
