// bf16 MFMA GEMM, "W-direct" main loop (round 6 experiment, the A/B the round-5 review asked for): the same (32*MT) x 128 x 64 tile, 4 wave64s as
// 2 x 2, persistent XCD-aware tile walk, epilogues (gemm_epilogue.h) and k-order of MFMAs as gemm_nt_kernel (gemm_bf16.hip) — bit-identical results —
// but ONLY THE A OPERAND goes through the LDS ring.  The W fragments are loaded global -> VGPR directly in the MFMA operand layout
// (buffer_load_dwordx4: lane (l15, g) holds W[n0 + 16 t + l15][k0 + 8 (g + 4 kk) .. + 8], 16 B along K), one k-step ahead, double-buffered by k-step
// parity (the k-loop is unrolled by two so that the register sets alternate statically).
// Reference call site: /root/reference/src/marqo/core/inference/embedding_models/open_clip_model.py:249-266 (encode_image / encode_text).
//
// What it changes against the LDS-both loop, per workgroup and k-step at MT = 5:
//   LDS-DMA pieces 36 -> 20 (the A tile only), LDS fragment reads 72 KB -> 40 KB, LDS stage 36 KB -> 20 KB (so THREE stages fit two workgroups per CU:
//   NS = 3 gives the LDS-DMA 2.5 k-steps to land instead of 1.5), the W fragments need no barrier (they are private to the wave);
//   price: the two M-waves of a workgroup each fetch the same W rows (32 KB of vector-memory traffic per k-step instead of 16 KB; the second fetch
//   hits the CU's L1 or the XCD's L2), and 64 more live registers (4 W sets x 16).
// Wait discipline (vector-memory operations retire in issue order; the W loads are inline asm, so every wait is placed by hand):
//   first half of step s : MFMA(W[p][0], af0) || A reads kk = 1 -> af1 || W loads kk = 0 of step s + 1 -> W[p^1][0]
//   mid-step             : vmcnt(NTW [+ NPA when NS = 3]) -> W[p][1] and the A stage s + 1 have landed; lgkmcnt(0); barrier
//   second half          : MFMA(W[p][1], af1) || W loads kk = 1 of step s + 1 -> W[p^1][1], THEN the NPA LDS-DMA pieces of stage s + NS into the
//                          buffer step s just finished with || A reads kk = 0 of stage s + 1 -> af0
//   top of step s + 1    : vmcnt(NTW + NPA) -> W[p^1][0] landed; lgkmcnt(0)
// Selected by mq_tune("gemm_wd", 2 | 3) / MQ_GEMM_WD (0 = off, the default until it wins): tools/gemm_bench.py --ab "base:gemm_wd=0;wd2:gemm_wd=2;wd3:gemm_wd=3".
#include <stdlib.h>
#include "common.h"
#include "gemm_epilogue.h"
#include "gemm_loop.h"

namespace {

template <int OFF>
__device__ __forceinline__ bf16x8 gload16(i32x4 rs, unsigned voff, unsigned soff) {
    i32x4 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=&v"(v) : "v"(voff), "s"(rs), "s"(soff), "n"(OFF));
    return __builtin_bit_cast(bf16x8, v);
}

// PAIR: both k-halves' W loads of the next step are issued together in the FIRST half (a lane's two 16-B pieces of a 128-B line go out back to back, so
// the second finds the line in flight in the CU's L1 instead of fetching it from L2 again half a k-step later)
template <int FLAGS, int MT, int NS, bool PAIR>
__global__ __launch_bounds__(256, 2) void gemm_wd_kernel(
    const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ Wt, int64_t ldw,
    const float* __restrict__ bias, const float* residual, void* out, int64_t ldc,
    int M, int N, int K, int tiles_n, int num_tiles, int cgroup, int band_rows, int wide_store,
    unsigned a_bytes, unsigned w_bytes, GemmLn ln) {
    constexpr int BM = 32 * MT, BN = 128;
    constexpr int NTW = 4;
    constexpr int A_TILE_BYTES = BM * BK * 2;
    constexpr int NPA = MT;           // LDS-DMA pieces (8 rows = 1 KiB each) per wave per stage
    constexpr int NM = NTW * MT;      // MFMAs per wave per k-half
    constexpr int ERG = !(FLAGS & MQ_EPI_RESIDUAL) ? MT : (FLAGS & MQ_EPI_OUT_F32) ? (MT <= 3 ? MT : (MT + 1) / 2) : (MT <= 4 ? MT : 3);
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- XCD-aware, bijective (virtual) block -> tile map, L2-blocked order inside an XCD's share (the map of gemm_nt_kernel) ----------------------
    const int q = num_tiles >> 3, r = num_tiles & 7;
    const int tiles_m = (M + BM - 1) / BM;
    auto tile_origin = [&](int vbid, int& m0, int& n0) {
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

    // ---- A staging (as gemm_nt_kernel): wave w owns A rows [8*MT*w, 8*MT*(w+1)) of a stage, 8 rows per LDS-DMA piece; lane -> (row = base + lane/8,
    // physical 16-B chunk = lane%8), fetching logical chunk (lane%8) ^ (row&7): the swizzle lives on the SOURCE address
    const int srow = lane >> 3;
    const unsigned chunk_off = (unsigned)(((lane & 7) ^ (srow & 7)) * 16);
    unsigned a_vo[NPA];
    auto set_a_sources = [&](int m0) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int gm = m0 + wave * (8 * MT) + i * 8 + srow; gm = gm < M ? gm : M - 1;
            a_vo[i] = (unsigned)gm * (unsigned)lda * 2u + chunk_off;
        }
    };
    // ---- W fragments: lane (l15, g) of sub-tile t reads 16 B at row n0 + 64 wn + 16 t + l15, k-chunk g (+ 4 for kk = 1: the instruction's offset:64)
    unsigned w_vo[NTW];
    auto set_w_sources = [&](int n0) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            int gn = n0 + wn * 64 + t * 16 + l15; gn = gn < N ? gn : N - 1;
            w_vo[t] = (unsigned)gn * (unsigned)ldw * 2u + (unsigned)(g * 16);
        }
    };
    const int nk = K / BK;   // even (the launcher checks)
    // two cursors walk the workgroup's tiles: the A cursor NS stages ahead of the MFMAs, the W cursor one k-step ahead.  Past the last tile the
    // descriptors' sizes drop to 0 (out-of-range requests: no traffic, zeros), which keeps the k-step one straight-line body
    int d_vbid = blockIdx.x, d_k = 0;
    int w_vbid = blockIdx.x, w_k = 0;
    unsigned a_rec = a_bytes, w_rec = w_bytes;
    {
        int m0, n0;
        tile_origin(d_vbid, m0, n0);
        set_a_sources(m0);
        set_w_sources(n0);
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned dma_a0 = lds0 + (unsigned)wave * (8 * MT * 128);
    const unsigned w_lo = (unsigned)(uintptr_t)Wt, w_hi = (unsigned)((uintptr_t)Wt >> 32) & 0xffffu;
    auto issue_a_piece = [&](int i, unsigned bufoff) {
        dma16(__builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_rec, 0x00020000), a_vo[i], (unsigned)d_k * (BK * 2), dma_a0 + bufoff + (unsigned)i * 1024u);
    };
    auto advance_a = [&]() {
        if (++d_k == nk) {
            d_k = 0;
            d_vbid += gridDim.x;
            if (d_vbid < num_tiles) {
                int m0, n0;
                tile_origin(d_vbid, m0, n0);
                set_a_sources(m0);
            } else a_rec = 0;
        }
    };
    auto advance_w = [&]() {
        if (++w_k == nk) {
            w_k = 0;
            w_vbid += gridDim.x;
            if (w_vbid < num_tiles) {
                int m0, n0;
                tile_origin(w_vbid, m0, n0);
                set_w_sources(n0);
            } else w_rec = 0;
        }
    };

    // ---- A fragment read addresses (LDS byte offsets), fixed per lane: logical chunk for k-half kk is g + 4*kk, (row & 7) == (l15 & 7)
    const unsigned sw0 = (unsigned)((g ^ (l15 & 7)) << 4), sw1 = (unsigned)(((g + 4) ^ (l15 & 7)) << 4);
    const unsigned a_row = lds0 + (unsigned)((wm * (16 * MT) + l15) * 128);
    const unsigned aB0 = a_row + sw0, aB1 = a_row + sw1;

    f32x4 acc[MT][4];
    bf16x8 af0[MT], af1[MT];
    bf16x8 wq[2][2][NTW];   // [k-step parity][k-half][sub-tile]
    auto read_a = [&](auto t_tag, unsigned abase, bf16x8 (&af)[MT]) {
        constexpr int T = decltype(t_tag)::value;
        af[T] = lds_read16<T * 2048>(abase);
    };
    auto load_w = [&](auto t_tag, auto kk_tag, bf16x8 (&w)[NTW]) {
        constexpr int T = decltype(t_tag)::value, KK = decltype(kk_tag)::value;
        const i32x4 rs = {(int)w_lo, (int)w_hi, (int)w_rec, 0x00020000};
        w[T] = gload16<KK * 64>(rs, w_vo[T], (unsigned)w_k * (BK * 2));
    };
    auto next_buf = [&](unsigned b) { return b + (unsigned)A_TILE_BYTES == (unsigned)(NS * A_TILE_BYTES) ? 0u : b + (unsigned)A_TILE_BYTES; };

    // ---- prologue: A stage 0, the W fragments of step 0, A stages 1 .. NS-1; then the kk = 0 A fragments of stage 0
    unsigned bufoff = 0;   // LDS byte offset of the stage the next k-step consumes
    {
#pragma unroll
        for (int i = 0; i < NPA; ++i) issue_a_piece(i, 0);
        advance_a();
        static_for<NTW>([&](auto t) { load_w(t, std::integral_constant<int, 0>{}, wq[0][0]); });
        static_for<NTW>([&](auto t) { load_w(t, std::integral_constant<int, 1>{}, wq[0][1]); });
        advance_w();
#pragma unroll
        for (int st = 1; st < NS; ++st) {
#pragma unroll
            for (int i = 0; i < NPA; ++i) issue_a_piece(i, (unsigned)(st * A_TILE_BYTES));
            advance_a();
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NTW + (NS - 1) * NPA) : "memory");   // stage 0 landed (loads retire in issue order)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        static_for<MT>([&](auto t) { read_a(t, aB0, af0); });
    }

    int c_vbid = blockIdx.x;

    // one k-step on the stage at `bufoff` with the W set of parity P; on entry af0 holds (or is about to receive) its kk = 0 A fragments
    auto kstep = [&](auto p_tag) {
        constexpr int P = decltype(p_tag)::value;
        // -------- top: W[P][0] (issued in the previous step's first half) and af0 have landed
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PAIR ? NPA : NTW + NPA) : "memory");
#pragma unroll
        for (int t = 0; t < NTW; ++t) landed(wq[P][0][t]);
#pragma unroll
        for (int t = 0; t < MT; ++t) landed(af0[t]);
        __builtin_amdgcn_sched_barrier(0);
        {
            // -------- first half: MFMAs on (W[P][0], af0); side work: the MT A reads of kk = 1 and the NTW W loads of kk = 0 for the NEXT step
            const unsigned ab = aB1 + bufoff;
            constexpr int NSIDE = MT + (PAIR ? 2 * NTW : NTW);
            static_for<NM>([&](auto idx_tag) {
                constexpr int idx = decltype(idx_tag)::value, mt = idx / NTW, nt = idx % NTW;
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[P][0][nt], af0[mt], acc[mt][nt], 0, 0, 0);
                constexpr int lo = idx == 0 ? 0 : (idx * NSIDE) / NM;
                constexpr int hi = idx == NM - 1 ? NSIDE : ((idx + 1) * NSIDE) / NM;
                static_for<hi - lo>([&](auto it_tag) {
                    constexpr int it = lo + decltype(it_tag)::value;
                    __builtin_amdgcn_sched_barrier(0);
                    // reads first (their data is needed at the mid-step), the W loads (needed a whole step later) behind them
                    if constexpr (it < MT) read_a(std::integral_constant<int, it>{}, ab, af1);
                    else if constexpr (!PAIR) load_w(std::integral_constant<int, it - MT>{}, std::integral_constant<int, 0>{}, wq[P ^ 1][0]);
                    else if constexpr (((it - MT) & 1) == 0) load_w(std::integral_constant<int, (it - MT) / 2>{}, std::integral_constant<int, 0>{}, wq[P ^ 1][0]);
                    else load_w(std::integral_constant<int, (it - MT) / 2>{}, std::integral_constant<int, 1>{}, wq[P ^ 1][1]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        }
        // -------- mid-step: W[P][1] and (my pieces of) the A stage after this one have landed, my reads of this buffer are done; after the barrier
        // both hold for every wave: the next stage may be read, this buffer may be refilled.  NS = 3: the pieces issued in the previous step belong
        // to the stage after next and may stay in flight
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((PAIR ? 2 * NTW : NTW) + (NS == 2 ? 0 : NPA)) : "memory");
#pragma unroll
        for (int t = 0; t < NTW; ++t) landed(wq[P][1][t]);
#pragma unroll
        for (int t = 0; t < MT; ++t) landed(af1[t]);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        {
            // -------- second half: MFMAs on (W[P][1], af1); side work, in this order of VECTOR-MEMORY issue: the NTW W loads of kk = 1 for the next
            // step, then the NPA LDS-DMA pieces of stage s + NS (into the buffer this step just finished with); the MT A reads of the next stage's
            // kk = 0 fragments alternate with the W loads
            const unsigned nb = next_buf(bufoff);
            const unsigned ab = aB0 + nb;
            constexpr int NSIDE = (PAIR ? 0 : NTW) + MT + NPA;
            static_for<NM>([&](auto idx_tag) {
                constexpr int idx = decltype(idx_tag)::value, mt = idx / NTW, nt = idx % NTW;
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[P][1][nt], af1[mt], acc[mt][nt], 0, 0, 0);
                constexpr int lo = idx == 0 ? 0 : (idx * NSIDE) / NM;
                constexpr int hi = idx == NM - 1 ? NSIDE : ((idx + 1) * NSIDE) / NM;
                static_for<hi - lo>([&](auto it_tag) {
                    constexpr int it = lo + decltype(it_tag)::value;
                    __builtin_amdgcn_sched_barrier(0);
                    // items 0 .. 2 NTW - 1: W load / A read alternating; then the remaining A reads; then the DMA pieces
                    if constexpr (PAIR) {
                        if constexpr (it < MT) read_a(std::integral_constant<int, it>{}, ab, af0);
                        else issue_a_piece(it - MT, bufoff);
                    } else if constexpr (it < 2 * NTW) {
                        if constexpr ((it & 1) == 0) load_w(std::integral_constant<int, it / 2>{}, std::integral_constant<int, 1>{}, wq[P ^ 1][1]);
                        else read_a(std::integral_constant<int, it / 2>{}, ab, af0);
                    } else if constexpr (it < NTW + MT) read_a(std::integral_constant<int, it - NTW>{}, ab, af0);
                    else issue_a_piece(it - NTW - MT, bufoff);
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            bufoff = nb;
        }
        advance_a();
        advance_w();
    };

    for (;;) {
        int cm0, cn0;
        tile_origin(c_vbid, cm0, cn0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; kt += 2) {
            kstep(std::integral_constant<int, 0>{});
            kstep(std::integral_constant<int, 1>{});
        }
        // the compiler takes an asm's outputs as valid once the statement has executed: retire the next tile's first fragments (all the W loads; the
        // LDS-DMA pieces issued behind them may stay in flight) before any code it may place behind the loop (register copies) can touch them
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NPA) : "memory");
#pragma unroll
        for (int t = 0; t < NTW; ++t) { landed(wq[0][0][t]); landed(wq[0][1][t]); }
#pragma unroll
        for (int t = 0; t < MT; ++t) landed(af0[t]);

        gemm_epilogue<FLAGS, MT, ERG, true>(acc, bias, residual, out, ldc, M, N, cm0 + wm * (16 * MT), cn0 + wn * 64, l15, g, wide_store != 0, &ln, nullptr);

        c_vbid += gridDim.x;
        if (c_vbid >= num_tiles) break;
    }
    // the trailing (out-of-range) requests must have retired before the workgroup's LDS can be handed to another one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#ifdef MQ_GEMM_PROBE   // compile-and-inspect builds (tests/test_gemm_isa.py): ONE instantiation, hipcc -DMQ_GEMM_PROBE=<flags> -DMQ_GEMM_PROBE_MT=<mt> -DMQ_GEMM_PROBE_NS=<ns> -S
#ifndef MQ_GEMM_PROBE_NS
#define MQ_GEMM_PROBE_NS 3
#endif
#ifndef MQ_GEMM_PROBE_PAIR
#define MQ_GEMM_PROBE_PAIR true
#endif
__attribute__((used)) void* mq_gemm_wd_probe() { return (void*)gemm_wd_kernel<MQ_GEMM_PROBE, MQ_GEMM_PROBE_MT, MQ_GEMM_PROBE_NS, MQ_GEMM_PROBE_PAIR>; }
}  // namespace
#else

template <int FLAGS, int MT, int NS, bool PAIR>
int launch_wd(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* residual, void* out, int64_t ldc,
              int M, int N, int K, int tiles_n, int num_tiles, int cgroup, int band_rows, int grid, int wide, unsigned a_bytes, unsigned w_bytes,
              const GemmLn& ln, hipStream_t s) {
    constexpr int LDS = NS * 32 * MT * BK * 2;
    static std::atomic<uint64_t> attr_done{0};
    auto kern = gemm_wd_kernel<FLAGS, MT, NS, PAIR>;
    if (hipError_t e = mq_ensure_dyn_lds((const void*)kern, LDS, attr_done); e != hipSuccess) {
        mq_set_error("mq_gemm_bf16 (W-direct): hipFuncSetAttribute: %s", hipGetErrorString(e));
        return MQ_ERR_HIP;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, s, (const bf16_t*)A, lda, (const bf16_t*)W, ldw, bias, residual, out, ldc, M, N, K, tiles_n,
                       num_tiles, cgroup, band_rows, wide, a_bytes, w_bytes, ln);
    MQ_CHECK_LAUNCH("mq_gemm_bf16 (W-direct)");
    return MQ_OK;
}

template <int FLAGS>
int dispatch_wd(int mt, int ns, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* residual, void* out, int64_t ldc,
                int M, int N, int K, int tiles_n, int num_tiles, int cgroup, int band_rows, int grid, int wide, unsigned a_bytes, unsigned w_bytes,
                const GemmLn& ln, hipStream_t s) {
#define MQ_WD(MTV, NSV) \
    if (mt == MTV && ns == NSV) return launch_wd<FLAGS, MTV, NSV % 4, (NSV >= 4)>(A, lda, W, ldw, bias, residual, out, ldc, M, N, K, tiles_n, num_tiles, cgroup, band_rows, grid, wide, a_bytes, w_bytes, ln, s)
    MQ_WD(4, 2); MQ_WD(4, 3); MQ_WD(5, 2); MQ_WD(5, 3); MQ_WD(4, 6); MQ_WD(4, 7); MQ_WD(5, 6); MQ_WD(5, 7);   // ns + 4: the PAIR form
#undef MQ_WD
    return -1;
}

}  // namespace

// One launch of the W-direct kernel with the tile plan gemm_bf16.hip's launch_gemm_mt made (same arguments as its own kernel).  Returns -1 when this
// (flags, tile height, stages) combination is not instantiated or K / 64 is odd: the caller then launches the LDS-both kernel.
int mq_gemm_wd_launch(int flags, int mt, int ns, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* residual, void* out,
                      int64_t ldc, int M, int N, int K, int tiles_n, int num_tiles, int cgroup, int band_rows, int grid, int wide, unsigned a_bytes,
                      unsigned w_bytes, const GemmLn& ln, hipStream_t s) {
    if ((K / BK) % 2 != 0 || K < 2 * BK) return -1;
#define MQ_WD_CASE(F) \
    case (F): return dispatch_wd<(F)>(mt, ns, A, lda, W, ldw, bias, residual, out, ldc, M, N, K, tiles_n, num_tiles, cgroup, band_rows, grid, wide, a_bytes, w_bytes, ln, s)
    switch (flags) {
        MQ_WD_CASE(0);
        MQ_WD_CASE(MQ_EPI_OUT_F32);
        MQ_WD_CASE(MQ_EPI_BIAS);
        MQ_WD_CASE(MQ_EPI_BIAS | MQ_EPI_GELU);
        MQ_WD_CASE(MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_OUT_F32);
        MQ_WD_CASE(MQ_EPI_BIAS | MQ_EPI_RESIDUAL);
        MQ_WD_CASE(MQ_EPI_BIAS | MQ_EPI_LN_APPLY);
        MQ_WD_CASE(MQ_EPI_BIAS | MQ_EPI_GELU | MQ_EPI_LN_APPLY);
        MQ_WD_CASE(MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_ROW_STATS);
        default: return -1;
    }
#undef MQ_WD_CASE
}
#endif  // MQ_GEMM_PROBE
