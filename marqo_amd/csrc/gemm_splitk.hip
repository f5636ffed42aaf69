// Split-K GEMM for a HANDFUL of rows — the rows the big-tile row split of gemm_bf16.hip leaves over (round 5).
//
// The 8-wave 256 x 256 tile is 16-17 % faster than the narrow tiles inside the ViT-L/14 tower, but the rows that do not fill its rounds (128 of
// 32 896 at 128 images: 257 tokens per image never make a multiple of 256) needed a second launch of the narrow kernel that is pure latency — 16
// workgroups walking K / 64 dependent k-steps: 13-33 us, as much as the big tile had saved (profiles/r05g_l14_row_split_per_kernel.txt).  Such a problem
// has no FLOPs to speak of (128 x 1024 x 4096: 1 GFLOP) and one scarce thing, the length of its k-chain.  So K is cut S ways over the whole chip:
//   1. gemm_splitk_partial_kernel: workgroup (column tile, row tile, split s) multiplies a 128 x 128 tile over its K / S slice — fragments straight from
//      global memory into the MFMA operands (no LDS, no barrier: 2-8 k-steps per workgroup), fp32 partial tile -> workspace[s];
//   2. gemm_splitk_finish_kernel<FLAGS>: sums the S partials in split order (deterministic) INTO THE ACCUMULATOR LAYOUT of the tile kernels and runs the
//      very same gemm_epilogue<FLAGS> — every fused epilogue (bias / GELU / residual / LayerNorm apply / row sums) without a second implementation.
// The sum over k is associated differently from the tile kernels' (S slices instead of one chain): rows that take this path agree with them to fp32
// rounding of the accumulation, not bit for bit (tests/test_gemm_variants_gpu.py bounds it); mq_tune("gemm_splitk_rem", 0) keeps the narrow kernel.
#include <map>
#include <mutex>
#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int TM = 128, TN = 128;   // tile of a workgroup (4 waves as 2 x 2, 64 x 64 per wave = 4 x 4 accumulators of 16 x 16)

__global__ __launch_bounds__(256) void gemm_splitk_partial_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ Wt, int64_t ldw,
                                                                  float* __restrict__ ws, int M, int N, int K, int splits) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * TN + wn * 64, m0 = blockIdx.y * TM + wm * 64, s = blockIdx.z;
    const int steps = K / 32;                                      // k-steps of one MFMA depth
    const int k_lo = (int)((int64_t)s * steps / splits) * 32, k_hi = (int)((int64_t)(s + 1) * steps / splits) * 32;
    // lane (l15, g) feeds row base + l15, k elements [k + 8 g, k + 8 g + 8) of a fragment — the tile kernels' fragment layout; rows past the edge are
    // clamped (their products land in accumulator rows / columns the finish kernel never stores)
    const bf16_t* ap[4];
    const bf16_t* wp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int m = m0 + t * 16 + l15; m = m < M ? m : M - 1;
        int n = n0 + t * 16 + l15; n = n < N ? n : N - 1;
        ap[t] = A + (int64_t)m * lda + g * 8;
        wp[t] = Wt + (int64_t)n * ldw + g * 8;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = k_lo; k < k_hi; k += 32) {
        bf16x8 af[4], wf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            af[t] = *(const bf16x8*)(ap[t] + k);
            wf[t] = *(const bf16x8*)(wp[t] + k);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[mt][nt], 0, 0, 0);
    }
    // partial tile -> ws[s][m][n .. n + 3] (accumulator layout: column lane & 15 -> m, row 4 g + reg -> n), rows / columns of the problem only
    float* out = ws + (int64_t)s * M * N;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + mt * 16 + l15;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + nt * 16 + g * 4;
            if (m < M && n < N) *(f32x4*)(out + (int64_t)m * N + n) = acc[mt][nt];
        }
    }
}

template <int FLAGS>
__global__ __launch_bounds__(256) void gemm_splitk_finish_kernel(const float* __restrict__ ws, int splits, const float* __restrict__ bias, const float* residual,
                                                                 void* out, int64_t ldc, int M, int N, int wide_store, GemmLn ln) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * TN + wn * 64, m0 = blockIdx.y * TM + wm * 64;
    f32x4 acc[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int m = m0 + mt * 16 + l15, n = n0 + nt * 16 + g * 4;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (m < M && n < N)
                for (int s = 0; s < splits; ++s) v += *(const f32x4*)(ws + ((int64_t)s * M + m) * N + n);   // split order: deterministic
            acc[mt][nt] = v;
        }
    gemm_epilogue<FLAGS, 4, 4, false>(acc, bias, residual, out, ldc, M, N, m0, n0, l15, g, wide_store != 0, &ln, nullptr);
}

// ---- workspace: the S partial tiles of one launch, one block per HIP stream (launches on a stream are ordered; request threads own their streams).
// Allocated on a stream's first split-K launch and kept for the process — the ONE place where this library allocates device memory itself
// (the C ABI's GEMM entry points take no workspace).
constexpr size_t SPLITK_WS_BYTES = 32u << 20;
std::mutex g_ws_mu;
std::map<std::pair<int, hipStream_t>, float*> g_ws;

float* splitk_workspace(hipStream_t s) {
    int dev = 0;
    hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(g_ws_mu);
    float*& p = g_ws[{dev, s}];
    if (!p) {
        void* q = nullptr;
        if (hipMalloc(&q, SPLITK_WS_BYTES) != hipSuccess) return nullptr;
        p = (float*)q;
    }
    return p;
}

template <int FLAGS>
int launch_finish(const float* ws, int splits, const float* bias, const float* residual, void* out, int64_t ldc, int M, int N, const GemmLn& ln, hipStream_t s) {
    const int wide = (!(FLAGS & MQ_EPI_OUT_F32) && ldc % 8 == 0 && ((uintptr_t)out & 15) == 0) ? 1 : 0;
    hipLaunchKernelGGL(gemm_splitk_finish_kernel<FLAGS>, dim3((N + TN - 1) / TN, (M + TM - 1) / TM), dim3(256), 0, s, ws, splits, bias, residual, out, ldc, M, N, wide, ln);
    MQ_CHECK_LAUNCH("mq_gemm_bf16 (split-K finish)");
    return MQ_OK;
}

}  // namespace

int mq_gemm_splitk_enabled = getenv("MQ_GEMM_SPLITK_REM") ? atoi(getenv("MQ_GEMM_SPLITK_REM")) : 1;   // mq_tune("gemm_splitk_rem", v)

// can the left-over rows of a row split take the split-K path?  (a handful of rows, a k-chain worth cutting, partials inside the workspace)
bool mq_gemm_splitk_ok(int64_t M, int64_t N, int64_t K) {
    return mq_gemm_splitk_enabled && M >= 1 && M <= 256 && K >= 512 && K % 32 == 0 && N % 4 == 0;
}

// out[M, N] = epi(A[M, K] @ W[N, K]^T) for M <= 256 rows; `flags` = the MQ_EPI_* combination of the calling tile kernel (incl. MQ_EPI_ROW_STATS /
// MQ_EPI_LN_APPLY with `ln` filled in)
int mq_gemm_splitk(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* residual, void* out, int64_t ldc,
                   int M, int N, int K, int flags, const GemmLn& ln, hipStream_t s) {
    const int tiles = ((N + TN - 1) / TN) * ((M + TM - 1) / TM);
    int splits = 256 / tiles;                                       // one workgroup per CU
    const int max_by_k = K / 128;                                   // >= 4 MFMA depths per split
    const int max_by_ws = (int)(SPLITK_WS_BYTES / ((size_t)M * N * 4));
    splits = splits < 1 ? 1 : splits;
    splits = splits > max_by_k ? max_by_k : splits;
    splits = splits > max_by_ws ? max_by_ws : splits;
    if (splits < 1) { mq_set_error("mq_gemm_bf16 (split-K): %d x %d partials exceed the workspace", M, N); return MQ_ERR_INVALID; }
    float* ws = splitk_workspace(s);
    if (!ws) { mq_set_error("mq_gemm_bf16 (split-K): workspace allocation failed"); return MQ_ERR_HIP; }
    hipLaunchKernelGGL(gemm_splitk_partial_kernel, dim3((N + TN - 1) / TN, (M + TM - 1) / TM, splits), dim3(256), 0, s, (const bf16_t*)A, lda, (const bf16_t*)W, ldw, ws,
                       M, N, K, splits);
    MQ_CHECK_LAUNCH("mq_gemm_bf16 (split-K partial)");
#define MQ_SK_CASE(F) \
    case (F): return launch_finish<(F)>(ws, splits, bias, residual, out, ldc, M, N, ln, s)
    switch (flags) {
        MQ_SK_CASE(0);
        MQ_SK_CASE(MQ_EPI_OUT_F32);
        MQ_SK_CASE(MQ_EPI_BIAS | MQ_EPI_OUT_F32);
        MQ_SK_CASE(MQ_EPI_BIAS);
        MQ_SK_CASE(MQ_EPI_BIAS | MQ_EPI_GELU);
        MQ_SK_CASE(MQ_EPI_BIAS | MQ_EPI_QUICKGELU);
        MQ_SK_CASE(MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_OUT_F32);
        MQ_SK_CASE(MQ_EPI_BIAS | MQ_EPI_RESIDUAL);
        MQ_SK_CASE(MQ_EPI_BIAS | MQ_EPI_LN_APPLY);
        MQ_SK_CASE(MQ_EPI_BIAS | MQ_EPI_GELU | MQ_EPI_LN_APPLY);
        MQ_SK_CASE(MQ_EPI_BIAS | MQ_EPI_QUICKGELU | MQ_EPI_LN_APPLY);
        MQ_SK_CASE(MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_ROW_STATS);
        default:
            mq_set_error("mq_gemm_bf16 (split-K): unsupported epilogue flag combination 0x%x", flags);
            return MQ_ERR_INVALID;
    }
#undef MQ_SK_CASE
}
