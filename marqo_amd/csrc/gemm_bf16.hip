// bf16 MFMA GEMM with fused epilogues for gfx950 (K3 / K5 / K1-GEMM / K6 / K8 of SURVEY.md §8a).
//
//   out[M,N] = epi( A[M,K] @ W[N,K]^T )         A, W bf16, K-contiguous ("NT": W is the
//                                                PyTorch nn.Linear weight as stored)
//
// Design (MI355X-first, not a CUDA tiling):
//   * 128x128x64 block tile, 4 wave64s as 2x2, each wave a 64x64 sub-tile = 4x4
//     v_mfma_f32_16x16x32_bf16 accumulators (64 fp32 acc VGPRs / lane).
//   * global -> LDS by direct LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction),
//     2 stages x (16 KiB A + 16 KiB W) = 64 KiB LDS -> 2 workgroups per CU.
//   * LDS tiles are [128 rows][128 B]; the 16-B chunk index is XOR-swizzled with (row & 7).
//     LDS-DMA writes lane-linear, so the swizzle is applied to each lane's GLOBAL source
//     address and again on the ds_read_b128 side (same involution) -> conflict-free reads.
//   * operands are fed swapped (mfma(Wfrag, Afrag)) so each lane ends up owning 4
//     CONSECUTIVE n of one output row: bias/residual/out are 8/16-byte vector accesses.
//   * workgroup -> tile map is XCD-aware: the 8 XCDs (block b runs on XCD b % 8) each take a
//     contiguous range of tiles, so tiles sharing an A row-panel hit the same private L2.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;      // 16 KiB (A or W tile)
constexpr int STAGE_BYTES = 2 * TILE_BYTES;  // A + W
constexpr int GEMM_LDS = 2 * STAGE_BYTES;    // 64 KiB

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    // 16 B per lane, LDS destination = wave-uniform base + lane * 16.
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int FLAGS>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(
    const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ Wt, int64_t ldw,
    const float* __restrict__ bias, const float* residual, void* out, int64_t ldc,
    int M, int N, int K, int tiles_n, int num_tiles) {
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
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;

    // ---- staging: wave w owns rows [32w, 32w+32) of both tiles, 8 rows per LDS-DMA ---------
    // lane -> (row = base + lane/8, physical chunk = lane%8); it fetches logical chunk
    // (lane%8) ^ (row&7) of that row, so physical chunk p of row r holds logical chunk p^(r&7).
    const int srow = lane >> 3;
    const bf16_t* a_src[4];
    const bf16_t* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + srow;
        const int chunk = (lane & 7) ^ (row & 7);
        int gm = m0 + row; gm = gm < M ? gm : M - 1;
        int gn = n0 + row; gn = gn < N ? gn : N - 1;
        a_src[i] = A + (int64_t)gm * lda + chunk * 8;
        w_src[i] = Wt + (int64_t)gn * ldw + chunk * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* sa = smem + buf * STAGE_BYTES + wave * (32 * 128);
        char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            glds16(a_src[i] + (int64_t)kt * BK, sa + i * (8 * 128));
            glds16(w_src[i] + (int64_t)kt * BK, sw + i * (8 * 128));
        }
    };

    // ---- fragment read offsets (bytes inside a tile), fixed per lane ------------------------
    // logical chunk for k-half kk is g + 4*kk; (row & 7) == (l15 & 7) because sub-tile bases are
    // multiples of 16.
    int a_off[4], w_off[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        a_off[t] = (wm * 64 + t * 16 + l15) * 128;
        w_off[t] = (wn * 64 + t * 16 + l15) * 128;
    }
    const int sw0 = ((g) ^ (l15 & 7)) << 4;      // kk = 0
    const int sw1 = ((g + 4) ^ (l15 & 7)) << 4;  // kk = 1

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = K / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed for every wave, and every wave is done reading buffer (kt+1)&1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);

        const char* sa = smem + (kt & 1) * STAGE_BYTES;
        const char* sw = sa + TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int swz = kk ? sw1 : sw0;
            bf16x8 af[4], wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[t] = *(const bf16x8*)(sa + a_off[t] + swz);
                wf[t] = *(const bf16x8*)(sw + w_off[t] + swz);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[mt][nt], 0, 0, 0);
        }
    }

    // ---- epilogue: lane owns out[m][n .. n+3] for each (mt, nt) -----------------------------
    // D[i][j] = sum_k Wfrag[i][k] * Afrag[j][k]: column j = lane & 15 -> m, row i = 4*g + reg -> n.
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wm * 64 + mt * 16 + l15;
        if (m >= M) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + wn * 64 + nt * 16 + g * 4;
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

template <int FLAGS>
int launch_gemm(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                const float* residual, void* out, int64_t ldc, int M, int N, int K, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_kernel<FLAGS>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS);
        if (e != hipSuccess) {
            mq_set_error("mq_gemm_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return MQ_ERR_HIP;
        }
        attr_set = true;
    }
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int num_tiles = tiles_m * tiles_n;
    hipLaunchKernelGGL(gemm_nt_kernel<FLAGS>, dim3(num_tiles), dim3(256), GEMM_LDS, s,
                       (const bf16_t*)A, lda, (const bf16_t*)W, ldw, bias, residual, out, ldc,
                       M, N, K, tiles_n, num_tiles);
    MQ_CHECK_LAUNCH("mq_gemm_bf16");
    return MQ_OK;
}

}  // namespace

extern "C" int mq_gemm_bf16(const void* d_A, int64_t lda, const void* d_W, int64_t ldw,
                            const float* d_bias, const float* d_residual, void* d_out, int64_t ldc,
                            int64_t M, int64_t N, int64_t K, int flags, void* stream) {
    MQ_CHECK_ARG(d_A && d_W && d_out, "mq_gemm_bf16: null operand");
    MQ_CHECK_ARG(M >= 1 && N >= 4 && K >= BK, "mq_gemm_bf16: bad shape M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
    MQ_CHECK_ARG(K % BK == 0, "mq_gemm_bf16: K=%ld must be a multiple of %d", (long)K, BK);
    MQ_CHECK_ARG(N % 4 == 0, "mq_gemm_bf16: N=%ld must be a multiple of 4", (long)N);
    MQ_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0, "mq_gemm_bf16: leading dims must keep 16-byte rows");
    MQ_CHECK_ARG(M < (1 << 30) && N < (1 << 30) && K < (1 << 30), "mq_gemm_bf16: shape too large");
    MQ_CHECK_ARG(!(flags & MQ_EPI_BIAS) || d_bias, "mq_gemm_bf16: MQ_EPI_BIAS without bias");
    MQ_CHECK_ARG(!(flags & MQ_EPI_RESIDUAL) || d_residual, "mq_gemm_bf16: MQ_EPI_RESIDUAL without residual");
    hipStream_t s = (hipStream_t)stream;
    MqProfScope prof(0, s, 2.0 * (double)M * (double)N * (double)K);
    const int m = (int)M, n = (int)N, k = (int)K;
#define MQ_GEMM_CASE(F) \
    case (F): return launch_gemm<(F)>(d_A, lda, d_W, ldw, d_bias, d_residual, d_out, ldc, m, n, k, s)
    switch (flags) {
        MQ_GEMM_CASE(0);
        MQ_GEMM_CASE(MQ_EPI_OUT_F32);
        MQ_GEMM_CASE(MQ_EPI_BIAS);
        MQ_GEMM_CASE(MQ_EPI_BIAS | MQ_EPI_OUT_F32);
        MQ_GEMM_CASE(MQ_EPI_BIAS | MQ_EPI_GELU);
        MQ_GEMM_CASE(MQ_EPI_BIAS | MQ_EPI_QUICKGELU);
        MQ_GEMM_CASE(MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_OUT_F32);
        MQ_GEMM_CASE(MQ_EPI_RESIDUAL | MQ_EPI_OUT_F32);
        default:
            mq_set_error("mq_gemm_bf16: unsupported epilogue flag combination 0x%x", flags);
            return MQ_ERR_INVALID;
    }
#undef MQ_GEMM_CASE
}
