// Shared epilogue of the bf16 MFMA GEMM kernels (gemm_bf16.hip, gemm_ring.hip).
// Accumulator layout (operands are fed swapped, mfma(Wfrag, Afrag)): for sub-tile (mt, nt)
//   D[i][j] = sum_k Wfrag[i][k] * Afrag[j][k]: column j = lane & 15 -> m, row i = 4*g + reg -> n,
// so a lane owns out[m][n .. n+3]: bias / residual / out are 8- or 16-byte vector accesses.
#pragma once
#include "common.h"

template <int FLAGS, int MT>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[MT][4], const float* __restrict__ bias, const float* residual, void* out,
                                              int64_t ldc, int M, int N, int wave_m0, int wave_n0, int l15, int g) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = wave_m0 + mt * 16 + l15;
        if (m >= M) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = wave_n0 + nt * 16 + g * 4;
            if (n >= N) continue;
            f32x4 v = acc[mt][nt];
            if (FLAGS & MQ_EPI_BIAS) {
                const f32x4 b = *(const f32x4*)(bias + n);
                v += b;
            }
            if (FLAGS & MQ_EPI_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            }
            if (FLAGS & MQ_EPI_QUICKGELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
            }
            const int64_t o = (int64_t)m * ldc + n;
            if (FLAGS & MQ_EPI_RESIDUAL) {
                const f32x4 rr = *(const f32x4*)(residual + o);
                v += rr;
            }
            if (FLAGS & MQ_EPI_OUT_F32) {
                *(f32x4*)((float*)out + o) = v;
            } else {
                uint2 p;
                p.x = pack_bf16x2(v[0], v[1]);
                p.y = pack_bf16x2(v[2], v[3]);
                *(uint2*)((bf16_t*)out + o) = p;
            }
        }
    }
}
