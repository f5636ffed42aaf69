// Skinny GEMM for the search path (one query / a handful of rows per vectorise() call: tensor_search.py:1876-1911 in the reference).
//
//   out[M, N] = epi( A[M, K] @ W[N, K]^T ),  M <= 80,   optionally A = LayerNorm(x) computed in the kernel's prologue
//
// The tiled kernel of gemm_bf16.hip gives such a call N/128 workgroups (4 .. 24 of 256 CUs), each walking the whole K serially: a
// 10-token CLIP text costs ~7 us per GEMM and 0.68 ms per query although it is < 0.1 GFLOP.  Here the work is cut the other way:
//   * one workgroup per 16 output COLUMNS (N/16 = 32 .. 256 workgroups: the weight matrix is streamed once, by the whole chip), all rows;
//   * its 4 waves split K; each wave walks 32-deep chunks: the W fragment (16 columns x 32 k) goes global -> VGPR directly (16 B per lane,
//     no LDS: a weight byte is used by this wave only), the A fragments come from global (L2-resident: M x K is a few hundred KB) or,
//     with the fused LayerNorm, from the normalised bf16 rows the workgroup just wrote to LDS;
//   * partial sums meet in LDS and are added in wave order 0..3 (deterministic), then the usual epilogue: bias / GELU / QuickGELU /
//     residual (fp32 or bf16 stream) with 16-byte (fp32) / 8-byte (bf16) stores.
//   * GEMMs with few column slices (N <= 1024: the out-projection and fc2, 32 .. 64 workgroups) run 8 waves per workgroup instead of 4, so
//     the K walk of a slice is cut twice as fine and twice as many weight loads are in flight per CU; every wave fetches the fragments of
//     a group of k-chunks (4, or 2 for M > 32) before the first MFMA of the group: one exposed memory latency per group, not per chunk.
// LayerNorm fusion: every workgroup normalises all M rows itself (M x K fp32 reads from L2, ~0.1 us) — redundant arithmetic, but it removes
// the LayerNorm launch and its dependent-kernel boundary (~1.5 us each, MI355X_MICROARCH.md price list), 24 of them per CLIP text query.
// Post-LN encoders (BERT) need the normalised rows as the next residual too: workgroup 0 also writes them out in fp32 (ln_out, a buffer
// other than x: the other workgroups are still reading x).
#include "common.h"

namespace {

constexpr int SM_BN = 16;        // output columns per workgroup
constexpr int SM_MAX_MT = 5;     // 16-row tiles: M <= 80
constexpr int SM_LDS_LIMIT = 150 * 1024;
constexpr int SM_WIDE_WG_MAX_N = 1024;  // up to 64 column slices: 8 waves per workgroup
constexpr int SM_LN_CH = 5;     // fused LayerNorm: float4 chunks per lane, K <= 1280
constexpr int SM_LN_MAX_MT = 2; // fused LayerNorm: M <= 32 (a wave keeps its 4 * MT rows in registers)

// LDS image of the normalised rows: [M_pad][K] bf16 with a 16-byte pad per row (fragment reads of 16 lanes = 16 rows at one k then land in
// different banks)
__host__ __device__ inline int ln_row_bytes(int K) { return K * 2 + 16; }

template <int FLAGS, int MT, bool LN, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_small_kernel(const void* __restrict__ Av, int64_t lda, int a_bf16_stream,
                                                             const bf16_t* __restrict__ Wt, int64_t ldw, const float* __restrict__ bias,
                                                             const void* residual, void* out, int64_t ldc, int M, int N, int K,
                                                             const float* __restrict__ ln_g, const float* __restrict__ ln_b, float eps,
                                                             float* ln_out, const int32_t* __restrict__ ln_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool BF16_OUT = !(FLAGS & MQ_EPI_OUT_F32);
    constexpr bool RES_BF16 = (FLAGS & MQ_EPI_RESIDUAL) && BF16_OUT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * SM_BN;
    if (!LN && gridDim.y > 1) {
        // row groups (81 .. mq_gemm_small_group_rows rows): workgroup (x, y) owns the 16 columns of slice x for rows [y * 16 MT, (y + 1) * 16 MT);
        // the weight slice is fetched once from HBM and then from L2 by the other groups
        const int64_t r0 = (int64_t)blockIdx.y * (MT * 16);
        Av = (const bf16_t*)Av + r0 * lda;
        if (FLAGS & MQ_EPI_RESIDUAL) residual = (const char*)residual + r0 * ldc * (RES_BF16 ? 2 : 4);
        out = (char*)out + r0 * ldc * (BF16_OUT ? 2 : 4);
        M = min(M - (int)r0, MT * 16);
    }
    const int rowb = ln_row_bytes(K);
    // partial sums: [NW waves][MT][64 lanes] f32x4 — behind the LN image when there is one
    f32x4* part = (f32x4*)(smem + (LN ? ((MT * 16 * rowb + 255) & ~255) : 0));

    if (LN) {
        // ---- prologue: LayerNorm of rows wave, wave + 4, ... of x [M, K] (fp32, or the bf16 residual stream) into the LDS image.  All of a
        // wave's rows are fetched before the first reduction (one exposed L2 latency, not one per row); same arithmetic as layernorm_kernel
        // (rowops.hip): sum -> mean, sum of squared deviations -> rstd ----
        constexpr int RPW = (MT * 16 + NW - 1) / NW;  // rows per wave
        const int nch = K >> 2;                       // float4 chunks per row
        f32x4 v[RPW][SM_LN_CH];
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int r = wave + NW * i;
#pragma unroll
            for (int j = 0; j < SM_LN_CH; ++j) {
                const int c = lane + j * 64;
                if (r >= M || c >= nch) { v[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; continue; }
                const int64_t rs = ln_rows ? (int64_t)ln_rows[r] : (int64_t)r;   // pooled rows (class token / EOT) are gathered
                if (a_bf16_stream) {
                    const uint2 q = *(const uint2*)((const bf16_t*)Av + rs * lda + c * 4);
                    v[i][j] = f32x4{__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u)};
                } else {
                    v[i][j] = *(const f32x4*)((const float*)Av + rs * lda + c * 4);
                }
            }
        }
        f32x4 gv[SM_LN_CH], bv[SM_LN_CH];
#pragma unroll
        for (int j = 0; j < SM_LN_CH; ++j) {
            const int c = lane + j * 64;
            gv[j] = c < nch ? *(const f32x4*)(ln_g + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            bv[j] = c < nch ? *(const f32x4*)(ln_b + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float mean[RPW], rstd[RPW];
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            float s1 = 0.f;
#pragma unroll
            for (int j = 0; j < SM_LN_CH; ++j)
                if (lane + j * 64 < nch) s1 += (v[i][j][0] + v[i][j][1]) + (v[i][j][2] + v[i][j][3]);
            mean[i] = s1;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int i = 0; i < RPW; ++i) mean[i] += __shfl_xor(mean[i], o, 64);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            mean[i] = mean[i] / (float)K;
            float s2 = 0.f;
#pragma unroll
            for (int j = 0; j < SM_LN_CH; ++j)
                if (lane + j * 64 < nch) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float d = v[i][j][e] - mean[i]; s2 += d * d; }
                }
            rstd[i] = s2;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int i = 0; i < RPW; ++i) rstd[i] += __shfl_xor(rstd[i], o, 64);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int r = wave + NW * i;              // rows >= M hold zeros: they normalise to beta, and are never stored by the epilogue
            if (r >= MT * 16) continue;
            rstd[i] = rsqrtf(rstd[i] / (float)K + eps);
            char* dst = smem + (size_t)r * rowb;
#pragma unroll
            for (int j = 0; j < SM_LN_CH; ++j) {
                const int c = lane + j * 64;
                if (c >= nch) continue;
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = (v[i][j][e] - mean[i]) * rstd[i] * gv[j][e] + bv[j][e];
                uint2 q;
                q.x = pack_bf16x2(y[0], y[1]);
                q.y = pack_bf16x2(y[2], y[3]);
                *(uint2*)(dst + c * 8) = q;
                if (ln_out && blockIdx.x == 0 && r < M) *(f32x4*)(ln_out + (int64_t)r * K + c * 4) = y;   // post-LN: the new residual
            }
        }
        __syncthreads();
    }

    // ---- main loop: this wave's share of the 32-deep k-chunks, G chunks per group (all of a group's loads are issued before its MFMAs) ----
    constexpr int G = MT <= 2 ? 4 : 2;
    const int nchunks = K >> 5;
    const int per = (nchunks + NW - 1) / NW;
    const int c0 = wave * per, c1 = min(c0 + per, nchunks);
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int wn = n0 + l15;
    const bf16_t* wrow = Wt + (int64_t)(wn < N ? wn : N - 1) * ldw + g * 8;
    for (int kc = c0; kc < c1; kc += G) {
        bf16x8 wf[G], af[G][MT];
#pragma unroll
        for (int j = 0; j < G; ++j) wf[j] = kc + j < c1 ? *(const bf16x8*)(wrow + (int64_t)(kc + j) * 32) : bf16x8{};   // the HBM stream
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int m = t * 16 + l15;
                const int kk = (kc + j < c1 ? kc + j : c0) * 32 + g * 8;     // (a chunk past the end multiplies a zero weight fragment)
                if (LN) af[j][t] = *(const bf16x8*)(smem + (size_t)m * rowb + kk * 2);
                else af[j][t] = m < M ? *(const bf16x8*)((const bf16_t*)Av + (int64_t)m * lda + kk) : bf16x8{};
            }
#pragma unroll
        for (int j = 0; j < G; ++j)
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[j][t], acc[t], 0, 0, 0);
    }
    // ---- cross-wave reduction in LDS, fixed order ----
#pragma unroll
    for (int t = 0; t < MT; ++t) part[(wave * MT + t) * 64 + lane] = acc[t];
    __syncthreads();
    // ---- epilogue: wave w finishes row tiles w, w + NW, ...; a lane owns out[m][n .. n+3], m = t*16 + l15, n = n0 + 4g ----
    const int n = n0 + g * 4;
    f32x4 bias_v = f32x4{0.f, 0.f, 0.f, 0.f};
    if ((FLAGS & MQ_EPI_BIAS) && n < N) bias_v = *(const f32x4*)(bias + n);
    for (int t = wave; t < MT; t += NW) {
        const int m = t * 16 + l15;
        f32x4 v = part[(0 * MT + t) * 64 + lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += part[(w * MT + t) * 64 + lane];
        if (m >= M || n >= N) continue;
        if (FLAGS & MQ_EPI_BIAS) v += bias_v;
        if (FLAGS & MQ_EPI_GELU) {
            v = gelu_erf4(v);
        }
        if (FLAGS & MQ_EPI_QUICKGELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
        }
        const int64_t o = (int64_t)m * ldc + n;
        if (FLAGS & MQ_EPI_RESIDUAL) {
            if (RES_BF16) {
                const uint2 q = *(const uint2*)((const bf16_t*)residual + o);
                v += f32x4{__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u)};
            } else {
                v += *(const f32x4*)((const float*)residual + o);
            }
        }
        if (BF16_OUT) {
            uint2 q;
            q.x = pack_bf16x2(v[0], v[1]);
            q.y = pack_bf16x2(v[2], v[3]);
            *(uint2*)((bf16_t*)out + o) = q;
        } else {
            *(f32x4*)((float*)out + o) = v;
        }
    }
}

template <int FLAGS, int MT, bool LN, int NW>
int launch_small(const void* A, int64_t lda, int a_stream16, const void* W, int64_t ldw, const float* bias, const void* residual, void* out,
                 int64_t ldc, int M, int N, int K, const float* ln_g, const float* ln_b, float eps, float* ln_out, const int32_t* ln_rows, hipStream_t s) {
    const size_t lds = (LN ? (((size_t)MT * 16 * ln_row_bytes(K) + 255) & ~(size_t)255) : 0) + (size_t)NW * MT * 64 * 16;
    static std::atomic<uint64_t> attr_done{0};
    if (lds > 64 * 1024) {
        if (hipError_t e = mq_ensure_dyn_lds((const void*)gemm_small_kernel<FLAGS, MT, LN, NW>, 160 * 1024, attr_done); e != hipSuccess) {
            mq_set_error("mq_gemm_small: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return MQ_ERR_HIP;
        }
    }
    const unsigned groups = LN ? 1u : (unsigned)((M + MT * 16 - 1) / (MT * 16));
    hipLaunchKernelGGL((gemm_small_kernel<FLAGS, MT, LN, NW>), dim3((unsigned)((N + SM_BN - 1) / SM_BN), groups), dim3(NW * 64), lds, s, A, lda, a_stream16,
                       (const bf16_t*)W, ldw, bias, residual, out, ldc, M, N, K, ln_g, ln_b, eps, ln_out, ln_rows);
    MQ_CHECK_LAUNCH("mq_gemm_small");
    return MQ_OK;
}

template <int FLAGS, bool LN, int NW>
int dispatch_mt(const void* A, int64_t lda, int a_stream16, const void* W, int64_t ldw, const float* bias, const void* residual, void* out, int64_t ldc,
                int M, int N, int K, const float* ln_g, const float* ln_b, float eps, float* ln_out, const int32_t* ln_rows, hipStream_t s) {
#define MQ_SM_MT(T) return launch_small<FLAGS, T, LN, NW>(A, lda, a_stream16, W, ldw, bias, residual, out, ldc, M, N, K, ln_g, ln_b, eps, ln_out, ln_rows, s)
    if constexpr (LN) {  // mq_gemm_small_ok(.., ln = true) admits M <= 32 only
        if (M <= 16) MQ_SM_MT(1);
        MQ_SM_MT(2);
    } else {
        // more than 80 rows: the fewest row groups of at most 5 row tiles, then the smallest height that covers them (256 rows = 4 x 64)
        const int mt_all = (M + 15) / 16, groups = (mt_all + SM_MAX_MT - 1) / SM_MAX_MT;
        const int mt = (mt_all + groups - 1) / groups;   // row tiles are guarded: the next instantiated height
        if (mt <= 1) MQ_SM_MT(1);
        if (mt <= 2) MQ_SM_MT(2);
        if (mt <= 3) MQ_SM_MT(3);
        if (mt <= 4) MQ_SM_MT(4);
        // (taller slices measured slower than the tiled kernel: 257 rows — one ViT-L/14 image — 3.97 ms vs 3.2 ms per image; every
        // workgroup re-reads all of A through its CU's 64 B/clk vector-memory path.  profiles/r02d_latency.txt)
        MQ_SM_MT(5);
    }
#undef MQ_SM_MT
}

// few column slices (N <= 1024): 8 waves per workgroup
template <int FLAGS, bool LN>
int dispatch_nw(const void* A, int64_t lda, int a_stream16, const void* W, int64_t ldw, const float* bias, const void* residual, void* out, int64_t ldc,
                int M, int N, int K, const float* ln_g, const float* ln_b, float eps, float* ln_out, const int32_t* ln_rows, hipStream_t s) {
    if (N <= SM_WIDE_WG_MAX_N && K >= 512)
        return dispatch_mt<FLAGS, LN, 8>(A, lda, a_stream16, W, ldw, bias, residual, out, ldc, M, N, K, ln_g, ln_b, eps, ln_out, ln_rows, s);
    return dispatch_mt<FLAGS, LN, 4>(A, lda, a_stream16, W, ldw, bias, residual, out, ldc, M, N, K, ln_g, ln_b, eps, ln_out, ln_rows, s);
}

}  // namespace

// knob: rows up to which the towers take the skinny path (0 = never).  mq_tune("small_m", v) / MQ_SMALL_M
mq_knob mq_gemm_small_max_rows{getenv("MQ_SMALL_M") ? atoi(getenv("MQ_SMALL_M")) : 80};

// knob: rows up to which a plain GEMM call (no fused LayerNorm) still takes the skinny kernel, in row groups of <= 80 (0 = no grouping).
// mq_tune("small_m_grouped", v) / MQ_SMALL_M_GROUPED.  This is what the pooled-rows-only last block of a 256-item batch runs (M = 256: the
// tiled kernel gave those calls 12-48 workgroups, 11-28 us each) and what a request of a few items runs in every block.
mq_knob mq_gemm_small_group_rows{getenv("MQ_SMALL_M_GROUPED") ? atoi(getenv("MQ_SMALL_M_GROUPED")) : 320};

bool mq_gemm_small_grouped_ok(int64_t M, int64_t N, int64_t K) {
    if (mq_gemm_small_max_rows <= 0 || M <= mq_gemm_small_max_rows || M <= SM_MAX_MT * 16 || M > mq_gemm_small_group_rows) return false;
    return N % 4 == 0 && K % 32 == 0 && K >= 32;
}

// can this call run on the skinny kernel in ONE row group?  (shape rules + LDS budget of the fused LayerNorm)  The towers also read this as
// "is this call on the search path" (LayerNorm fusion, graph replay, no row selection), which is why the grouped form has its own predicate.
bool mq_gemm_small_ok(int64_t M, int64_t N, int64_t K, bool ln) {
    if (mq_gemm_small_max_rows <= 0 || M < 1 || M > mq_gemm_small_max_rows || M > SM_MAX_MT * 16) return false;
    if (N % 4 != 0 || K % 32 != 0 || K < 32) return false;
    if (ln) {
        const size_t mt = (size_t)((M + 15) / 16);
        if (mt > SM_LN_MAX_MT || K > SM_LN_CH * 256 || mt * 16 * ln_row_bytes((int)K) + 8 * mt * 64 * 16 + 256 > (size_t)SM_LDS_LIMIT) return false;
    }
    return true;
}

// out = epi(A @ W^T): A bf16 [M, K].  flags as mq_gemm_bf16 (incl. MQ_EPI_BIAS | MQ_EPI_RESIDUAL = bf16 residual in / out).
int mq_gemm_small(const void* d_A, int64_t lda, const void* d_W, int64_t ldw, const float* d_bias, const void* d_residual, void* d_out,
                  int64_t ldc, int64_t M, int64_t N, int64_t K, int flags, hipStream_t s) {
    MQ_CHECK_ARG(mq_gemm_small_ok(M, N, K, false) || mq_gemm_small_grouped_ok(M, N, K), "mq_gemm_small: shape M=%ld N=%ld K=%ld unsupported", (long)M,
                 (long)N, (long)K);
    MqProfScope prof(0, s, 2.0 * (double)M * (double)N * (double)K);
#define MQ_SM_CASE(F) \
    case (F): return dispatch_nw<(F), false>(d_A, lda, 0, d_W, ldw, d_bias, d_residual, d_out, ldc, (int)M, (int)N, (int)K, nullptr, nullptr, 0.f, nullptr, nullptr, s)
    switch (flags) {
        MQ_SM_CASE(0);
        MQ_SM_CASE(MQ_EPI_OUT_F32);
        MQ_SM_CASE(MQ_EPI_BIAS | MQ_EPI_OUT_F32);
        MQ_SM_CASE(MQ_EPI_BIAS);
        MQ_SM_CASE(MQ_EPI_BIAS | MQ_EPI_GELU);
        MQ_SM_CASE(MQ_EPI_BIAS | MQ_EPI_QUICKGELU);
        MQ_SM_CASE(MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_OUT_F32);
        MQ_SM_CASE(MQ_EPI_BIAS | MQ_EPI_RESIDUAL);
        default:
            mq_set_error("mq_gemm_small: unsupported epilogue flag combination 0x%x", flags);
            return MQ_ERR_INVALID;
    }
#undef MQ_SM_CASE
}

// out = epi(LayerNorm(x) @ W^T): x fp32 [M, K] (x_bf16 != 0: the bf16 residual stream), gamma / beta fp32 [K].  bf16 out.
// d_ln_out (optional, fp32 [M, K], must not alias d_x): the normalised rows, written once (post-LN encoders: the next residual).
// d_rows (optional, int32 [M]): row m of the GEMM is row d_rows[m] of x (the heads: class token / EOT rows).
int mq_ln_gemm_small(const void* d_x, int64_t ldx, int x_bf16, const float* ln_g, const float* ln_b, float eps, const void* d_W, int64_t ldw,
                     const float* d_bias, void* d_out, int64_t ldc, int64_t M, int64_t N, int64_t K, int flags, float* d_ln_out,
                     const int32_t* d_rows, hipStream_t s) {
    MQ_CHECK_ARG(mq_gemm_small_ok(M, N, K, true), "mq_ln_gemm_small: shape M=%ld N=%ld K=%ld unsupported", (long)M, (long)N, (long)K);
    MQ_CHECK_ARG(d_x && ln_g && ln_b && d_W && d_out, "mq_ln_gemm_small: null pointer");
    MQ_CHECK_ARG((const void*)d_ln_out != d_x, "mq_ln_gemm_small: d_ln_out must not alias d_x (every workgroup reads x while workgroup 0 writes)");
    MqProfScope prof(0, s, 2.0 * (double)M * (double)N * (double)K);
#define MQ_SM_CASE(F) \
    case (F): return dispatch_nw<(F), true>(d_x, ldx, x_bf16, d_W, ldw, d_bias, nullptr, d_out, ldc, (int)M, (int)N, (int)K, ln_g, ln_b, eps, d_ln_out, d_rows, s)
    switch (flags) {
        MQ_SM_CASE(MQ_EPI_OUT_F32);   // the towers' heads: proj(ln(pooled row)), fp32 out
        MQ_SM_CASE(MQ_EPI_BIAS | MQ_EPI_OUT_F32);   // ... with a bias (the EVA02 towers project through a Linear with bias)
        MQ_SM_CASE(MQ_EPI_BIAS);
        MQ_SM_CASE(MQ_EPI_BIAS | MQ_EPI_GELU);
        MQ_SM_CASE(MQ_EPI_BIAS | MQ_EPI_QUICKGELU);
        default:
            mq_set_error("mq_ln_gemm_small: unsupported epilogue flag combination 0x%x", flags);
            return MQ_ERR_INVALID;
    }
#undef MQ_SM_CASE
}

// C-ABI test hooks (tests/test_small_m_gpu.py): the two entry points above with plain pointers
extern "C" int mq_gemm_small_bf16(const void* d_A, int64_t lda, const void* d_W, int64_t ldw, const float* d_bias, const void* d_residual,
                                  void* d_out, int64_t ldc, int64_t M, int64_t N, int64_t K, int flags, void* stream) {
    MQ_CHECK_ARG(d_A && d_W && d_out, "mq_gemm_small_bf16: null operand");
    MQ_CHECK_ARG(!(flags & MQ_EPI_BIAS) || d_bias, "mq_gemm_small_bf16: MQ_EPI_BIAS without bias");
    MQ_CHECK_ARG(!(flags & MQ_EPI_RESIDUAL) || d_residual, "mq_gemm_small_bf16: MQ_EPI_RESIDUAL without residual");
    MQ_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0, "mq_gemm_small_bf16: leading dims must keep 16-byte rows");
    return mq_gemm_small(d_A, lda, d_W, ldw, d_bias, d_residual, d_out, ldc, M, N, K, flags, (hipStream_t)stream);
}

extern "C" int mq_ln_gemm_small_bf16(const void* d_x, int64_t ldx, int x_bf16, const float* d_ln_g, const float* d_ln_b, float eps, const void* d_W,
                                     int64_t ldw, const float* d_bias, void* d_out, int64_t ldc, int64_t M, int64_t N, int64_t K, int flags,
                                     float* d_ln_out, void* stream) {
    MQ_CHECK_ARG(!(flags & MQ_EPI_BIAS) || d_bias, "mq_ln_gemm_small_bf16: MQ_EPI_BIAS without bias");
    MQ_CHECK_ARG(ldx % 4 == 0 && ldw % 8 == 0 && ldc % 4 == 0, "mq_ln_gemm_small_bf16: leading dims must keep 16-byte rows");
    return mq_ln_gemm_small(d_x, ldx, x_bf16, d_ln_g, d_ln_b, eps, d_W, ldw, d_bias, d_out, ldc, M, N, K, flags, d_ln_out, nullptr, (hipStream_t)stream);
}
