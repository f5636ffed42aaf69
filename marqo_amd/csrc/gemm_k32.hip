// bf16 MFMA GEMM, short-k-step / many-workgroups form:  out[M,N] = epi( A[M,K] @ W[N,K]^T )   (same contract as gemm_bf16.hip)
//
// The phase trace of the (32*MT)x128x64 kernel (tools/probes/gemm_trace.py) shows a workgroup spending 25-50 % of
// its time in its epilogue and 15-25 % parked on the one stage of prefetch, with only ONE other workgroup on the CU to fill the
// matrix pipe meanwhile (2 x 72 KB of LDS).  This variant halves the k-step (BK = 32: 64-byte LDS rows, 2 stages of
// (BM + 128) x 64 B = 32 KB at BM = 128) and trims the registers (one 32-deep fragment set, 4 x 4 accumulators) so that THREE or
// FOUR workgroups share a CU: each SIMD then holds 3-4 waves, and an epilogue or a load wait of one is covered by the others.
//
// LDS image: rows of 64 B = four 16-byte chunks; chunk c of row r is stored at c ^ F[(r >> 2) & 3], F = {0, 2, 3, 1}.  With the
// 16x16x32 fragment map (lane = 16 g + l15 reads chunk g of row l15) each ds_read_b128 lane group — {0-3,12-15,20-27}, ... —
// then touches 16 distinct 16-byte slots of the 256-byte bank row (derivation in DESIGN.md §6.4).  LDS-DMA writes lane-linear
// (16 rows x 64 B per instruction), so the same permutation is applied to the per-lane SOURCE address.
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BN = 128, BK = 32;
constexpr int ROW_BYTES = BK * 2;              // 64
constexpr int W_TILE_BYTES = BN * ROW_BYTES;   // 8 KiB

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// chunk permutation of a row quad (4 consecutive rows share it): F = {0, 2, 3, 1} packed two bits each in 0x78
static_assert(((0x78 >> 0) & 3) == 0 && ((0x78 >> 2) & 3) == 2 && ((0x78 >> 4) & 3) == 3 && ((0x78 >> 6) & 3) == 1, "F = {0,2,3,1}");

template <int FLAGS, int MT, int WGS>
__global__ __launch_bounds__(256, WGS) void gemm_k32_kernel(
    const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ Wt, int64_t ldw,
    const float* __restrict__ bias, const float* residual, void* out, int64_t ldc,
    int M, int N, int K, int tiles_n, int num_tiles, int cgroup, int band_rows, int wide_store) {
    static_assert(MT % 2 == 0, "a wave stages 8*MT rows in 16-row LDS-DMA pieces");
    constexpr int BM = 32 * MT;
    constexpr int A_TILE_BYTES = BM * ROW_BYTES;
    constexpr int STAGE_BYTES = A_TILE_BYTES + W_TILE_BYTES;
    constexpr int PA = MT / 2, PW = 2;  // 1-KiB (16 rows x 64 B) pieces per wave per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto F = [](int quad) { return (0x78 >> ((quad & 3) * 2)) & 3; };

    const int q = num_tiles >> 3, r = num_tiles & 7;
    const int tiles_m = (M + BM - 1) / BM;
    auto tile_origin = [&](int vbid, int& m0, int& n0) {  // XCD-aware + L2-blocked map, see gemm_bf16.hip
        const int xcd = vbid & 7, idx = vbid >> 3;
        const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        int tm, tn;
        if (cgroup > 0) {
            const int band_tiles = band_rows * tiles_n;
            const int band = tile / band_tiles, rb = tile - band * band_tiles;
            const int rows_here = min(band_rows, tiles_m - band * band_rows);
            const int full = rows_here * cgroup, ncg_full = tiles_n / cgroup;
            int cg = rb / full, r2 = rb - cg * full, cw = cgroup;
            if (cg >= ncg_full) { cg = ncg_full; r2 = rb - ncg_full * full; cw = tiles_n - ncg_full * cgroup; }
            const int rr = r2 / cw;
            tm = band * band_rows + rr;
            tn = cg * cgroup + (r2 - rr * cw);
        } else {
            tm = tile / tiles_n;
            tn = tile - tm * tiles_n;
        }
        m0 = tm * BM;
        n0 = tn * BN;
    };

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;

    // ---- staging: wave w owns A rows [8*MT*w, 8*MT*(w+1)) and W rows [32w, 32w+32), 16 rows per LDS-DMA.
    // lane -> (row = piece base + lane/4, physical chunk = lane%4) fetches logical chunk (lane%4) ^ F[quad of that row]
    // (piece bases are multiples of 16, so the row's quad index is (lane/4) / 4 = lane >> 4)
    const int srow = lane >> 2;
    const int schunk = ((lane & 3) ^ F(lane >> 4)) * 8;  // element offset of the logical chunk inside the 32-deep k-step
    const bf16_t* a_src[PA];
    const bf16_t* w_src[PW];
    auto set_sources = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < PA; ++i) {
            int gm = m0 + wave * (8 * MT) + i * 16 + srow; gm = gm < M ? gm : M - 1;
            a_src[i] = A + (int64_t)gm * lda + schunk;
        }
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            int gn = n0 + wave * 32 + i * 16 + srow; gn = gn < N ? gn : N - 1;
            w_src[i] = Wt + (int64_t)gn * ldw + schunk;
        }
    };
    auto stage = [&](int buf, int kt) {
        char* sa = smem + buf * STAGE_BYTES + wave * (8 * MT * ROW_BYTES);
        char* sw = smem + buf * STAGE_BYTES + A_TILE_BYTES + wave * (32 * ROW_BYTES);
#pragma unroll
        for (int i = 0; i < PA; ++i) glds16(a_src[i] + (int64_t)kt * BK, sa + i * 1024);
#pragma unroll
        for (int i = 0; i < PW; ++i) glds16(w_src[i] + (int64_t)kt * BK, sw + i * 1024);
    };

    f32x4 acc[MT][4];
    const int nk = K / BK;
    int vbid = blockIdx.x;
    int m0, n0;
    tile_origin(vbid, m0, n0);
    set_sources(m0, n0);
    stage(0, 0);
    int buf = 0;
    for (;;) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // fragment read offsets: lane (l15, g) reads physical chunk g ^ F[(l15 >> 2) & 3] of row base16 + l15 (recomputed per tile
        // from a laundered lane id so that they are not live across the epilogue)
        int l15f = l15, gf = g;
        asm volatile("" : "+v"(l15f), "+v"(gf));
        const int coff = (gf ^ F(l15f >> 2)) << 4;
        int a_off[MT], w_off[4];
#pragma unroll
        for (int t = 0; t < MT; ++t) a_off[t] = (wm * (16 * MT) + t * 16 + l15f) * ROW_BYTES + coff;
#pragma unroll
        for (int t = 0; t < 4; ++t) w_off[t] = (wn * 64 + t * 16 + l15f) * ROW_BYTES + coff;

        auto kstep = [&](int cur, int64_t koff, auto prefetch_tag) {
            constexpr bool PREFETCH = decltype(prefetch_tag)::value;
            const char* sa = smem + cur * STAGE_BYTES;
            const char* sw = sa + A_TILE_BYTES;
            char* na = smem + (cur ^ 1) * STAGE_BYTES + wave * (8 * MT * ROW_BYTES);
            char* nw = smem + (cur ^ 1) * STAGE_BYTES + A_TILE_BYTES + wave * (32 * ROW_BYTES);
            bf16x8 af[MT], wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) wf[t] = *(const bf16x8*)(sw + w_off[t]);
#pragma unroll
            for (int t = 0; t < MT; ++t) af[t] = *(const bf16x8*)(sa + a_off[t]);
            constexpr int NL = PA + PW, NM = 4 * MT, GAP = NM / NL;
            int issued = 0;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[mt][nt], 0, 0, 0);
                    const int done = mt * 4 + nt + 1;
                    if (PREFETCH && done % GAP == 0 && issued < NL) {
                        if (issued < PA) glds16(a_src[issued] + koff, na + issued * 1024);
                        else glds16(w_src[issued - PA] + koff, nw + (issued - PA) * 1024);
                        ++issued;
                    }
                }
            if (PREFETCH) {
#pragma unroll
                for (int i = 0; i < NL; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, GAP, 0);
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                }
            }
        };

        for (int kt = 0; kt < nk - 1; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            kstep(buf, (int64_t)(kt + 1) * BK, std::true_type{});
            buf ^= 1;
        }
        const int cm0 = m0, cn0 = n0;
        vbid += gridDim.x;
        const bool more = vbid < num_tiles;
        if (more) {
            tile_origin(vbid, m0, n0);
            set_sources(m0, n0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (more) kstep(buf, 0, std::true_type{});
        else kstep(buf, 0, std::false_type{});
        buf ^= 1;
        gemm_epilogue<FLAGS, MT, 2>(acc, bias, residual, out, ldc, M, N, cm0 + wm * (16 * MT), cn0 + wn * 64, l15, g, wide_store != 0);
        if (!more) break;
    }
}

template <int FLAGS, int MT, int WGS>
int launch_k32(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* residual, void* out, int64_t ldc,
               int M, int N, int K, int cgroup_knob, int wide_knob, hipStream_t s) {
    constexpr int BM = 32 * MT;
    constexpr int LDS = 2 * (BM * ROW_BYTES + W_TILE_BYTES);
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int num_tiles = tiles_m * tiles_n;
    const int cgroup = (cgroup_knob > 0 && tiles_n > cgroup_knob && tiles_m >= 16) ? cgroup_knob : 0;
    const int band_rows = (tiles_m + 7) / 8;
    const int wide = (wide_knob && !(FLAGS & MQ_EPI_OUT_F32) && ldc % 8 == 0 && ((uintptr_t)out & 15) == 0) ? 1 : 0;
    const int slots = 256 * WGS;
    const int grid = num_tiles > slots ? slots : num_tiles;
    hipLaunchKernelGGL((gemm_k32_kernel<FLAGS, MT, WGS>), dim3(grid), dim3(256), LDS, s, (const bf16_t*)A, lda, (const bf16_t*)W, ldw, bias,
                       residual, out, ldc, M, N, K, tiles_n, num_tiles, cgroup, band_rows, wide);
    MQ_CHECK_LAUNCH("mq_gemm_bf16(k32)");
    return MQ_OK;
}

}  // namespace

// called from gemm_bf16.hip's dispatcher; wgs in {3, 4} workgroups per CU, K % 32 == 0
template <int FLAGS>
int mq_launch_gemm_k32(int wgs, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* residual,
                       void* out, int64_t ldc, int M, int N, int K, int cgroup_knob, int wide_knob, hipStream_t s) {
    (void)wgs;  // three workgroups per CU (160 VGPRs); the four-per-CU form (128 VGPRs) spills and is not built
    return launch_k32<FLAGS, 4, 3>(A, lda, W, ldw, bias, residual, out, ldc, M, N, K, cgroup_knob, wide_knob, s);
}

#define MQ_K32_INST(F)                                                                                                    \
    template int mq_launch_gemm_k32<(F)>(int, const void*, int64_t, const void*, int64_t, const float*, const float*, void*, \
                                         int64_t, int, int, int, int, int, hipStream_t)
MQ_K32_INST(0);
MQ_K32_INST(MQ_EPI_OUT_F32);
MQ_K32_INST(MQ_EPI_BIAS | MQ_EPI_OUT_F32);
MQ_K32_INST(MQ_EPI_BIAS);
MQ_K32_INST(MQ_EPI_BIAS | MQ_EPI_GELU);
MQ_K32_INST(MQ_EPI_BIAS | MQ_EPI_QUICKGELU);
MQ_K32_INST(MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_OUT_F32);
MQ_K32_INST(MQ_EPI_BIAS | MQ_EPI_RESIDUAL);
