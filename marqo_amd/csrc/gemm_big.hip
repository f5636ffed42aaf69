// bf16 MFMA GEMM, CU-sized tile:  out[M,N] = epi( A[M,K] @ W[N,K]^T )   (same contract as gemm_bf16.hip)
//
// Measurements on the (32*MT)x128 kernel (2 workgroups/CU) show throughput proportional to 1/(bytes staged per FLOP):
// the global->LDS path sustains ~24 B/clk/CU, and two independent 160x128 tiles per CU ask for 57.6 B/clk at full
// MFMA rate.  This kernel gives ONE workgroup the whole CU: 8 wave64s as 2(M) x 4(N), tile (32*MT) x 256 x 64,
// each wave a (16*MT) x 64 sub-tile (MT x 4 v_mfma_f32_16x16x32_bf16 accumulators, MT = 4 / 5 / 6 / 8).  The W stage
// is shared by both wave rows and the A stage by all four wave columns, so a 256x256 tile needs 32 B/clk — and every
// wave issues half as many LDS-DMA pieces per MFMA.  2 stages x (BM*128 B + 32 KiB) = 96..128 KiB of LDS.
// LDS image, swizzle and epilogue are the ones of gemm_bf16.hip.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BN = 256, BK = 64;
constexpr int W_TILE_BYTES = BN * BK * 2;  // 32 KiB

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int FLAGS, int MT>
__global__ __launch_bounds__(512, 2) void gemm_big_kernel(
    const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ Wt, int64_t ldw,
    const float* __restrict__ bias, const float* residual, void* out, int64_t ldc,
    int M, int N, int K, int tiles_n, int num_tiles) {
    constexpr int BM = 32 * MT;
    constexpr int A_TILE_BYTES = BM * BK * 2;
    constexpr int STAGE_BYTES = A_TILE_BYTES + W_TILE_BYTES;
    constexpr int NA = 4 * MT;                  // 1-KiB (8 rows x 128 B) pieces of the A stage, dealt round-robin to the 8 waves
    constexpr int PA = (NA + 7) / 8, PW = 4;    // pieces per wave per stage (odd MT: waves 4..7 own one A piece fewer)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- XCD-aware, bijective block -> tile map -------------------------------------------
    const int bid = blockIdx.x;
    const int q = num_tiles >> 3, r = num_tiles & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, g = lane >> 4;

    // ---- staging: wave w owns A pieces w, w+8, w+16, ... (piece p = rows [8p, 8p+8)) and W rows [32w, 32w+32), 8 rows per LDS-DMA
    const int srow = lane >> 3;
    const bf16_t* a_src[PA];
    const bf16_t* w_src[PW];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = (i * 8 + wave) * 8 + srow;
        int gm = m0 + row; gm = gm < M ? gm : M - 1;
        a_src[i] = A + (int64_t)gm * lda + ((lane & 7) ^ (row & 7)) * 8;
    }
#pragma unroll
    for (int i = 0; i < PW; ++i) {
        const int row = wave * 32 + i * 8 + srow;
        int gn = n0 + row; gn = gn < N ? gn : N - 1;
        w_src[i] = Wt + (int64_t)gn * ldw + ((lane & 7) ^ (row & 7)) * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* sa = smem + buf * STAGE_BYTES + wave * (8 * 128);
        char* sw = smem + buf * STAGE_BYTES + A_TILE_BYTES + wave * (32 * 128);
#pragma unroll
        for (int i = 0; i < PA; ++i)
            if (i * 8 + wave < NA) glds16(a_src[i] + (int64_t)kt * BK, sa + i * (64 * 128));  // wave-uniform guard
#pragma unroll
        for (int i = 0; i < PW; ++i) glds16(w_src[i] + (int64_t)kt * BK, sw + i * (8 * 128));
    };

    int a_off[MT], w_off[4];
#pragma unroll
    for (int t = 0; t < MT; ++t) a_off[t] = (wm * (16 * MT) + t * 16 + l15) * 128;
#pragma unroll
    for (int t = 0; t < 4; ++t) w_off[t] = (wn * 64 + t * 16 + l15) * 128;
    const int sw0 = ((g) ^ (l15 & 7)) << 4;      // kk = 0
    const int sw1 = ((g + 4) ^ (l15 & 7)) << 4;  // kk = 1

    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- role-split k-loop ----------------------------------------------------------------------------------------
    // A k-step is four barrier-separated segments per wave:  R0 | M0 | R1 | M1 |   (R = ds_read the fragments of a 32-deep
    // half [+ in R0: issue this wave's LDS-DMA pieces of the NEXT stage], M = that half's 4*MT MFMAs under s_setprio 1).
    // Waves 4..7 (the second wave of every SIMD) run ONE segment behind waves 0..3 (one extra barrier up front), so on each
    // SIMD one wave is always in an M segment while the other is in an R segment: the matrix pipe sees a pure MFMA stream
    // and every LDS / VMEM issue stall happens beside it instead of in front of it.
    //   slot:      4k      4k+1    4k+2    4k+3    4k+4
    //   waves 0-3  R0(k)   M0(k)   R1(k)   M1(k)   R0(k+1)
    //   waves 4-7  M1(k-1) R0(k)   M0(k)   R1(k)   M1(k)
    // Stage k+1 goes into the buffer of stage k-1, whose last reader is waves 4-7's R1(k-1) in slot 4k-1, so issuing from
    // slot 4k on is safe; every wave retires its own pieces (vmcnt(0)) at the end of its R1, i.e. by the end of slot 4k+3,
    // before the first reader (waves 0-3's R0(k+1) in slot 4k+4).  Reads are drained (lgkmcnt(0)) before the barrier that
    // ends an R segment, so no ds_read is pending when another wave's DMA may target that buffer.
    auto wg_barrier = [] { asm volatile("s_barrier" ::: "memory"); };
    const int nk = K / BK;
    const bool late = wave >= 4;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();
    if (late) wg_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        const char* sa = smem + (kt & 1) * STAGE_BYTES;
        const char* sw = sa + A_TILE_BYTES;
        const bool prefetch = kt + 1 < nk;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            // ---- R segment
            const int swz = kk ? sw1 : sw0;
            bf16x8 af[MT], wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) wf[t] = *(const bf16x8*)(sw + w_off[t] + swz);
#pragma unroll
            for (int t = 0; t < MT; ++t) af[t] = *(const bf16x8*)(sa + a_off[t] + swz);
            if (kk == 0) {
                if (prefetch) stage((kt + 1) & 1, kt + 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            wg_barrier();
            // ---- M segment
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            wg_barrier();
        }
    }
    if (!late) wg_barrier();  // balance the stagger barrier

    gemm_epilogue<FLAGS, MT, 2>(acc, bias, residual, out, ldc, M, N, m0 + wm * (16 * MT), n0 + wn * 64, l15, g);
}

template <int FLAGS, int MT>
int launch_big_mt(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* residual,
                  void* out, int64_t ldc, int M, int N, int K, hipStream_t s) {
    constexpr int BM = 32 * MT;
    constexpr int LDS = 2 * (BM * BK * 2 + W_TILE_BYTES);
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = mq_ensure_dyn_lds((const void*)gemm_big_kernel<FLAGS, MT>, LDS, attr_done); e != hipSuccess) {
        mq_set_error("mq_gemm_bf16(big): hipFuncSetAttribute: %s", hipGetErrorString(e));
        return MQ_ERR_HIP;
    }
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int num_tiles = tiles_m * tiles_n;
    hipLaunchKernelGGL((gemm_big_kernel<FLAGS, MT>), dim3(num_tiles), dim3(512), LDS, s, (const bf16_t*)A, lda, (const bf16_t*)W, ldw,
                       bias, residual, out, ldc, M, N, K, tiles_n, num_tiles);
    MQ_CHECK_LAUNCH("mq_gemm_bf16(big)");
    return MQ_OK;
}

}  // namespace

// called from gemm_bf16.hip's dispatcher; mt in {4, 5, 6, 8}
template <int FLAGS>
int mq_launch_gemm_big(int mt, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* residual,
                       void* out, int64_t ldc, int M, int N, int K, hipStream_t s) {
    switch (mt) {
        case 4: return launch_big_mt<FLAGS, 4>(A, lda, W, ldw, bias, residual, out, ldc, M, N, K, s);
        case 5: return launch_big_mt<FLAGS, 5>(A, lda, W, ldw, bias, residual, out, ldc, M, N, K, s);
        case 6: return launch_big_mt<FLAGS, 6>(A, lda, W, ldw, bias, residual, out, ldc, M, N, K, s);
        default: return launch_big_mt<FLAGS, 8>(A, lda, W, ldw, bias, residual, out, ldc, M, N, K, s);
    }
}

#define MQ_BIG_INST(F)                                                                                                   \
    template int mq_launch_gemm_big<(F)>(int, const void*, int64_t, const void*, int64_t, const float*, const float*, void*, \
                                         int64_t, int, int, int, hipStream_t)
MQ_BIG_INST(0);
MQ_BIG_INST(MQ_EPI_OUT_F32);
MQ_BIG_INST(MQ_EPI_BIAS | MQ_EPI_OUT_F32);
MQ_BIG_INST(MQ_EPI_BIAS);
MQ_BIG_INST(MQ_EPI_BIAS | MQ_EPI_GELU);
MQ_BIG_INST(MQ_EPI_BIAS | MQ_EPI_QUICKGELU);
MQ_BIG_INST(MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_OUT_F32);
MQ_BIG_INST(MQ_EPI_BIAS | MQ_EPI_RESIDUAL);
