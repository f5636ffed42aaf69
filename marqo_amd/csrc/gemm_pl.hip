// bf16 MFMA GEMM main loop, software-pipelined inside the wave (round 4) — K3 / K5 / K1-GEMM / K6 / K8 of SURVEY.md §8a;
// reference call site: /root/reference/src/marqo/core/inference/embedding_models/open_clip_model.py:249-266 (encode_image / encode_text).
//
//   out[M,N] = epi( A[M,K] @ W[N,K]^T )        A, W bf16, K-contiguous (W = nn.Linear weight as stored)
//
// Same tile, LDS image, swizzle, accumulator layout and k-order of MFMAs as the round 1-3 kernel (bit-identical results), so the fused
// epilogues (gemm_epilogue.h) are shared.  What changed is the k-loop:
//   * fragments are double-buffered in REGISTERS by k-half: while the 4*MT MFMAs of half h run, the ds_read_b128s of the next half are
//     issued between them, so no MFMA waits on LDS latency (the old loop did barrier -> 18 ds_reads -> 40 MFMAs per k-step: ~150-250
//     cycles of exposed LDS latency per step and per wave);
//   * ONE workgroup barrier per k-step, placed MID-step: [MFMA(kk=0) || read kk=1] -> vmcnt(0) + barrier -> [MFMA(kk=1) || read kk=0 of
//     the next stage || LDS-DMA of the stage after next].  After the mid-step barrier the current LDS buffer is dead (both halves are in
//     registers), so the DMA that refills it is issued a full k-step before its data is needed, and its wait sits behind 4*MT MFMAs;
//   * LDS-DMA through a buffer descriptor (buffer_load_dwordx4 ... lds: 32-bit per-lane voffset + scalar k offset) instead of 64-bit
//     per-lane global addresses: half the address VGPRs and no per-piece 64-bit adds;
//   * the fragment reads are inline-asm ds_read_b128: hipcc's waitcnt pass cannot tell an LDS read from the LDS-DMA writes in flight
//     (no alias scopes on a dynamic __shared__ array) and would put s_waitcnt vmcnt(0) in front of every read that follows a DMA —
//     which would serialise the pipeline; the waits are therefore placed by hand (lgkmcnt(0) before the first consumer, vmcnt(0) only
//     at the mid-step barrier);
//   * the DMA runs on its own cursor (tile, k) two stages ahead of the MFMAs and simply walks on into the workgroup's next tile
//     (persistent grid), for any K / 64 >= 1.
#include <stdlib.h>
#include <string_view>
#include <type_traits>
#include "common.h"
#include "gemm_epilogue.h"

namespace {

constexpr int BN = 128, BK = 64;
constexpr int W_TILE_BYTES = BN * BK * 2;  // 16 KiB

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, unsigned lds_wave_base) {
    // 16 B per lane; LDS destination = wave-uniform base (M0) + lane * 16; source = descriptor base + voff (per lane) + soff (scalar)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(uintptr_t)lds_wave_base, 16, voff, soff, 0, 0);
}
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read16(unsigned addr) {
    i32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return __builtin_bit_cast(bf16x8, v);
}

template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
// f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>): every index is a compile-time constant inside f (register arrays stay
// registers; a run-time counter that the unroller has to fold first sent the fragment arrays to scratch)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// ORD: order of the second half's side work — 0: LDS-DMA pieces first, then the next stage's fragment reads; 1: reads first; 2: alternating
template <int FLAGS, int MT, int ORD>
__global__ __launch_bounds__(256, 2) void gemm_pl_kernel(
    const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ Wt, int64_t ldw,
    const float* __restrict__ bias, const float* residual, void* out, int64_t ldc,
    int M, int N, int K, int tiles_n, int num_tiles, int cgroup, int band_rows, int wide_store,
    unsigned a_bytes, unsigned w_bytes, GemmLn ln) {
    constexpr int BM = 32 * MT;
    constexpr int A_TILE_BYTES = BM * BK * 2;
    constexpr int STAGE_BYTES = A_TILE_BYTES + W_TILE_BYTES;
    constexpr int NL = MT + 4;      // LDS-DMA pieces (1 KiB each) per wave per stage = fragment reads per wave per k-half
    constexpr int NM = 4 * MT;      // MFMAs per wave per k-half
    // residual rows prefetched together by the epilogue (16-row units): the next tile's first fragments are live across it
    constexpr int ERG = !(FLAGS & MQ_EPI_RESIDUAL) ? MT : (FLAGS & MQ_EPI_OUT_F32) ? (MT <= 3 ? MT : (MT + 1) / 2) : (MT <= 5 ? MT : 3);
    // MQ_EPI_LN_APPLY (gemm_epilogue.h): the LayerNorm in front of this GEMM is folded in.  A is the UN-normalised bf16 stream; every row's
    // (sum x, sum x^2) is accumulated from the A tiles as they pass through LDS — thread t owns LOGICAL 16-byte chunk t % 8 of tile rows
    // t / 8 + 32 i (i < MT: BM rows x 8 chunks = 256 * MT chunks), two v_dot2c_f32_bf16 per MFMA of the first half-step — and (mean, rstd) per
    // tile row is left in LDS behind the stages for the epilogue.  The 8 lanes of a row add up in logical-chunk order: a row's statistics do
    // not depend on where in a tile (or in which call) the row sits.
    constexpr bool LN_APPLY = (FLAGS & MQ_EPI_LN_APPLY) != 0;
    constexpr int NSC = LN_APPLY ? MT : 0;   // statistics chunks per thread and stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* const rowstats_lds = (float2*)(smem + 2 * STAGE_BYTES);

    // ---- XCD-aware, bijective (virtual) block -> tile map, L2-blocked order inside an XCD's share (as in rounds 1-3) --------------
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

    // ---- staging: wave w owns A rows [8*MT*w, 8*MT*(w+1)) and W rows [32w, 32w+32) of a stage, 8 rows per LDS-DMA piece.
    // lane -> (row = base + lane/8, physical 16-B chunk = lane%8); it fetches logical chunk (lane%8) ^ (row&7) of that row, so physical
    // chunk p of row r holds logical chunk p ^ (r&7) (the swizzle lives on the SOURCE address; the LDS image is lane-linear).
    const int srow = lane >> 3;
    const unsigned chunk_off = (unsigned)(((lane & 7) ^ (srow & 7)) * 16);   // (row & 7) == (srow & 7): piece bases are multiples of 8 rows
    unsigned a_vo[MT], w_vo[4];
    auto set_sources = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int gm = m0 + wave * (8 * MT) + i * 8 + srow; gm = gm < M ? gm : M - 1;
            a_vo[i] = (unsigned)gm * (unsigned)lda * 2u + chunk_off;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int gn = n0 + wave * 32 + i * 8 + srow; gn = gn < N ? gn : N - 1;
            w_vo[i] = (unsigned)gn * (unsigned)ldw * 2u + chunk_off;
        }
    };
    const int nk = K / BK;
    // DMA cursor: (tile d_vbid, k-step d_k) of the next stage to request; runs two stages ahead of the MFMAs.  Once it has walked past
    // the workgroup's last tile the descriptors' sizes drop to 0: the (two) trailing requests are then out of range for every lane — no
    // memory traffic — which keeps the k-step a single straight-line body without a "nothing left to prefetch" variant.
    int d_vbid = blockIdx.x, d_k = 0;
    unsigned a_rec = a_bytes, w_rec = w_bytes;
    {
        int m0, n0;
        tile_origin(d_vbid, m0, n0);
        set_sources(m0, n0);
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned dma_a0 = lds0 + (unsigned)wave * (8 * MT * 128), dma_w0 = lds0 + A_TILE_BYTES + (unsigned)wave * (32 * 128);   // scalars
    auto issue_piece = [&](int i, unsigned bufoff) {
        const unsigned soff = (unsigned)d_k * (BK * 2);
        if (i < MT) dma16(__builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_rec, 0x00020000), a_vo[i < MT ? i : 0], soff, dma_a0 + bufoff + (unsigned)i * 1024u);
        else dma16(__builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, w_rec, 0x00020000), w_vo[i >= MT ? i - MT : 0], soff, dma_w0 + bufoff + (unsigned)(i - MT) * 1024u);
    };
    auto advance_cursor = [&]() {
        if (++d_k == nk) {
            d_k = 0;
            d_vbid += gridDim.x;
            if (d_vbid < num_tiles) {
                int m0, n0;
                tile_origin(d_vbid, m0, n0);
                set_sources(m0, n0);
            } else {
                a_rec = 0; w_rec = 0;
            }
        }
    };

    // ---- fragment read addresses (LDS byte offsets), fixed per lane: logical chunk for k-half kk is g + 4*kk, (row & 7) == (l15 & 7)
    const unsigned sw0 = (unsigned)((g ^ (l15 & 7)) << 4), sw1 = (unsigned)(((g + 4) ^ (l15 & 7)) << 4);
    const unsigned a_row = lds0 + (unsigned)((wm * (16 * MT) + l15) * 128);
    const unsigned w_row = lds0 + A_TILE_BYTES + (unsigned)((wn * 64 + l15) * 128);
    const unsigned aB0 = a_row + sw0, aB1 = a_row + sw1, wB0 = w_row + sw0, wB1 = w_row + sw1;

    f32x4 acc[MT][4];
    bf16x8 wf0[4], af0[MT], wf1[4], af1[MT];
    // statistics chunks: row (tid >> 3) + 32 i, logical chunk tid & 7 = physical chunk (tid & 7) ^ (row & 7)
    const unsigned scB = lds0 + (unsigned)((tid >> 3) * 128 + (((tid & 7) ^ ((tid >> 3) & 7)) << 4));
    i32x4 sc[LN_APPLY ? MT : 1];
    float st1[LN_APPLY ? MT : 1], st2[LN_APPLY ? MT : 1];
    auto read_chunk = [&](auto i_tag, unsigned base) {
        constexpr int I = decltype(i_tag)::value;
        i32x4 v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(base), "n"(I * 4096));
        sc[I] = v;
    };

    // one fragment read of a k-half: piece p < 4 -> W sub-tile p, else A sub-tile p - 4 (offsets t * 16 rows * 128 B)
    auto read_piece = [&](auto p_tag, unsigned wbase, unsigned abase, bf16x8 (&wf)[4], bf16x8 (&af)[MT]) {
        constexpr int P = decltype(p_tag)::value;
        if constexpr (P < 4) wf[P] = lds_read16<P * 2048>(wbase);
        else af[P - 4] = lds_read16<(P - 4) * 2048>(abase);
    };

    // ---- prologue: the workgroup's first two stages, then the kk = 0 fragments of the first -----------------------------------------
#pragma unroll
    for (int i = 0; i < NL; ++i) issue_piece(i, 0);
    advance_cursor();
#pragma unroll
    for (int i = 0; i < NL; ++i) issue_piece(i, STAGE_BYTES);
    advance_cursor();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");   // stage 0 landed (loads retire in issue order)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    static_for<NL>([&](auto p_tag) { read_piece(p_tag, wB0, aB0, wf0, af0); });
    static_for<NSC>([&](auto i_tag) { read_chunk(i_tag, scB); });

    unsigned bufoff = 0;   // LDS byte offset of the stage the next k-step consumes
    int c_vbid = blockIdx.x;

    // one k-step on the stage at `bufoff`; on entry (wf0, af0) hold (or are about to receive) its kk = 0 fragments
    auto kstep = [&]() {
        // -------- first half: MFMAs on (wf0, af0); reads of (wf1, af1) between them
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (wf0, af0) landed
        __builtin_amdgcn_sched_barrier(0);
        {
            const unsigned wb = wB1 + bufoff, ab = aB1 + bufoff;
            constexpr int RG = NM / NL > 0 ? NM / NL : 1;
            static_for<NM>([&](auto idx_tag) {
                constexpr int idx = decltype(idx_tag)::value, mt = idx / 4, nt = idx % 4;
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[nt], af0[mt], acc[mt][nt], 0, 0, 0);
                // reads due after MFMA idx: one per RG MFMAs; the last MFMA flushes whatever is left (NL reads always go out)
                constexpr int lo = idx == 0 ? 0 : (idx / RG < NL ? idx / RG : NL);
                constexpr int hi = idx == NM - 1 ? NL : ((idx + 1) / RG < NL ? (idx + 1) / RG : NL);
                static_for<hi - lo>([&](auto p_tag) {
                    __builtin_amdgcn_sched_barrier(0);
                    read_piece(std::integral_constant<int, lo + decltype(p_tag)::value>{}, wb, ab, wf1, af1);
                    __builtin_amdgcn_sched_barrier(0);
                });
                if constexpr (LN_APPLY) {
                    // two of the stage's 8 * MT statistics operations behind every MFMA (the chunks were read during the previous half-step)
                    static_for<2>([&](auto d_tag) {
                        constexpr int d = 2 * idx + decltype(d_tag)::value, ci = d / 8, w = (d % 8) / 2;
                        const bf16x2_t v = __builtin_bit_cast(bf16x2_t, sc[ci][w]);
                        if constexpr (d % 2 == 0) st1[ci] = __builtin_amdgcn_fdot2_f32_bf16(v, __builtin_bit_cast(bf16x2_t, 0x3f803f80u), st1[ci], false);
                        else st2[ci] = __builtin_amdgcn_fdot2_f32_bf16(v, v, st2[ci], false);
                    });
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        }
        // -------- mid-step: the stage after this one has landed (my pieces), my reads of this buffer are done; after the barrier both
        // hold for every wave: the next stage may be read, this buffer may be refilled
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) — as a builtin: the compiler's own scoreboard must see that nothing is pending
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // -------- second half: MFMAs on (wf1, af1); side work: the NL DMA pieces of the stage after next (into the buffer this step
        // just finished with) and the NL reads of the next stage's kk = 0 fragments (the next tile's first stage at a tile's last step;
        // stale bytes nobody uses at the workgroup's very last step)
        {
            const unsigned nb = bufoff ^ (unsigned)STAGE_BYTES;
            const unsigned wb = wB0 + nb, ab = aB0 + nb;
            constexpr int NSIDE = 2 * NL + NSC;   // (LN_APPLY: the next stage's statistics chunks go last)
            static_for<NM>([&](auto idx_tag) {
                constexpr int idx = decltype(idx_tag)::value, mt = idx / 4, nt = idx % 4;
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[nt], af1[mt], acc[mt][nt], 0, 0, 0);
                constexpr int lo = idx == 0 ? 0 : (idx * NSIDE) / NM;                         // side items due after this MFMA: [lo, hi)
                constexpr int hi = idx == NM - 1 ? NSIDE : ((idx + 1) * NSIDE) / NM;
                static_for<hi - lo>([&](auto it_tag) {
                    constexpr int it = lo + decltype(it_tag)::value;
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (it >= 2 * NL) {
                        read_chunk(std::integral_constant<int, it - 2 * NL>{}, scB + nb);
                    } else {
                        constexpr bool is_dma = ORD == 0 ? it < NL : ORD == 1 ? it >= NL : (it & 1) == 0;
                        constexpr int ord = ORD == 2 ? it / 2 : it % NL;        // its number among the pieces of its kind
                        if constexpr (is_dma) issue_piece(ord, bufoff);
                        else read_piece(std::integral_constant<int, ord>{}, wb, ab, wf0, af0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        }
        advance_cursor();
        bufoff ^= (unsigned)STAGE_BYTES;
    };

    for (;;) {
        int cm0, cn0;
        tile_origin(c_vbid, cm0, cn0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < (LN_APPLY ? MT : 1); ++i) st1[i] = st2[i] = 0.f;
        for (int kt = 0; kt < nk; ++kt) kstep();
        // the compiler takes an asm's outputs as valid once the statement has executed: retire the last fragment reads before any code it
        // may place behind the loop (register copies at the tile boundary) can touch them
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

        if constexpr (LN_APPLY) {
            // the 8 lanes that share a row add up their logical chunks in a fixed order (DPP: xor 1, xor 2, half-row mirror)
            auto dpp_add = [](float v, auto ctrl_tag) {
                constexpr int CTRL = decltype(ctrl_tag)::value;
                return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
            };
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                float a1 = st1[i], a2 = st2[i];
                a1 = dpp_add(a1, std::integral_constant<int, 0xB1>{}); a2 = dpp_add(a2, std::integral_constant<int, 0xB1>{});     // quad_perm [1,0,3,2]
                a1 = dpp_add(a1, std::integral_constant<int, 0x4E>{}); a2 = dpp_add(a2, std::integral_constant<int, 0x4E>{});     // quad_perm [2,3,0,1]
                a1 = dpp_add(a1, std::integral_constant<int, 0x141>{}); a2 = dpp_add(a2, std::integral_constant<int, 0x141>{});   // row_half_mirror
                const float mean = a1 * ln.inv_w;
                const float rstd = rsqrtf(fmaxf(a2 * ln.inv_w - mean * mean, 0.f) + ln.eps);
                if ((tid & 7) == 0) rowstats_lds[(tid >> 3) + 32 * i] = make_float2(mean, rstd);
            }
            __syncthreads();   // (rewritten one whole k-loop — nk barriers — later: no second barrier needed behind the epilogue's reads)
        }
        gemm_epilogue<FLAGS, MT, ERG, true>(acc, bias, residual, out, ldc, M, N, cm0 + wm * (16 * MT), cn0 + wn * 64, l15, g, wide_store != 0, &ln, nullptr,
                                            LN_APPLY ? rowstats_lds + wm * (16 * MT) : nullptr);

        c_vbid += gridDim.x;
        if (c_vbid >= num_tiles) break;
    }
    // the trailing (out-of-range) LDS-DMA requests must have retired before the workgroup's LDS can be handed to another one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

constexpr int RESIDENT_SLOTS = 512;  // 256 CUs x 2 workgroups

struct PlTune {
    int on, ord;
    static int env(const char* k, int d) { const char* v = getenv(k); return v ? atoi(v) : d; }
    PlTune() : on(env("MQ_GEMM_PL", 0)), ord(env("MQ_GEMM_PL_ORD", 0)) {}
};
PlTune g_pl;

template <int FLAGS, int MT, int ORD>
int launch_pl(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* residual, void* out, int64_t ldc,
              int M, int N, int K, int cgroup_knob, int wide_knob, hipStream_t s, const GemmLn& ln) {
    constexpr int BM = 32 * MT;
    constexpr int LDS = 2 * (BM * BK * 2 + W_TILE_BYTES) + ((FLAGS & MQ_EPI_LN_APPLY) ? BM * 8 : 0);
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = mq_ensure_dyn_lds((const void*)gemm_pl_kernel<FLAGS, MT, ORD>, LDS, attr_done); e != hipSuccess) {
        mq_set_error("mq_gemm_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return MQ_ERR_HIP;
    }
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int num_tiles = tiles_m * tiles_n;
    const int cgroup = (cgroup_knob > 0 && tiles_n > cgroup_knob && tiles_m >= 16) ? cgroup_knob : 0;
    const int band_rows = (tiles_m + 7) / 8;
    const bool act = (FLAGS & (MQ_EPI_GELU | MQ_EPI_QUICKGELU)) != 0;
    const int wide = (wide_knob && (wide_knob >= 2 || !act) && !(FLAGS & MQ_EPI_OUT_F32) && ldc % 8 == 0 && ((uintptr_t)out & 15) == 0) ? 1 : 0;
    const int grid = num_tiles > RESIDENT_SLOTS ? RESIDENT_SLOTS : num_tiles;
    const uint64_t a_bytes = ((uint64_t)(M - 1) * (uint64_t)lda + (uint64_t)K) * 2, w_bytes = ((uint64_t)(N - 1) * (uint64_t)ldw + (uint64_t)K) * 2;
    hipLaunchKernelGGL((gemm_pl_kernel<FLAGS, MT, ORD>), dim3(grid), dim3(256), LDS, s, (const bf16_t*)A, lda, (const bf16_t*)W, ldw, bias,
                       residual, out, ldc, M, N, K, tiles_n, num_tiles, cgroup, band_rows, wide, (unsigned)a_bytes, (unsigned)w_bytes, ln);
    MQ_CHECK_LAUNCH("mq_gemm_bf16");
    return MQ_OK;
}

}  // namespace

void mq_gemm_pl_tune(const char* key, int value) {
    const std::string_view k(key);
    if (k == "gemm_pl") g_pl.on = value;
    else if (k == "gemm_pl_ord") g_pl.ord = value;
}
int mq_gemm_pl_mode() { return g_pl.on; }

// operands addressed through 32-bit buffer offsets: rows * leading dimension * 2 B must stay below 4 GiB per launch
bool mq_gemm_pl_fits(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw) {
    const uint64_t lim = 0xffffffffull;
    return ((uint64_t)(M - 1) * (uint64_t)lda + (uint64_t)K) * 2 <= lim && ((uint64_t)(N - 1) * (uint64_t)ldw + (uint64_t)K) * 2 <= lim;
}

template <int FLAGS>
int mq_launch_gemm_pl(int mt, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* residual, void* out,
                      int64_t ldc, int M, int N, int K, int cgroup_knob, int wide_knob, hipStream_t s, const GemmLn& ln) {
    auto run = [&](auto mt_tag) {
        constexpr int T = decltype(mt_tag)::value;
        // second half-step's side work alternates LDS-DMA pieces and fragment reads (ORD = 2: the best of the three orders, profiles/r04a_*)
        return launch_pl<FLAGS, T, 2>(A, lda, W, ldw, bias, residual, out, ldc, M, N, K, cgroup_knob, wide_knob, s, ln);
    };
    if ((FLAGS & MQ_EPI_LN_APPLY) && mt == 6) mt = 5;   // the 192-row tile's stages fill the LDS of two workgroups per CU: no room for the row statistics
    switch (mt) {
        case 2: return run(std::integral_constant<int, 2>{});
        case 5: return run(std::integral_constant<int, 5>{});
        case 6: return run(std::integral_constant<int, 6>{});
        default: return run(std::integral_constant<int, 4>{});
    }
}

#ifdef MQ_PL_PROBE   // compile-and-inspect builds: one instantiation (hipcc -DMQ_PL_PROBE=<flags> -DMQ_PL_PROBE_MT=<mt> -S)
void* mq_gemm_pl_probe() { return (void*)gemm_pl_kernel<MQ_PL_PROBE, MQ_PL_PROBE_MT, 2>; }
#else
#define MQ_PL_INST(F)                                                                                                               \
    template int mq_launch_gemm_pl<(F)>(int, const void*, int64_t, const void*, int64_t, const float*, const float*, void*, int64_t, \
                                        int, int, int, int, int, hipStream_t, const GemmLn&)
MQ_PL_INST(0);
MQ_PL_INST(MQ_EPI_OUT_F32);
MQ_PL_INST(MQ_EPI_BIAS | MQ_EPI_OUT_F32);
MQ_PL_INST(MQ_EPI_BIAS);
MQ_PL_INST(MQ_EPI_BIAS | MQ_EPI_GELU);
MQ_PL_INST(MQ_EPI_BIAS | MQ_EPI_QUICKGELU);
MQ_PL_INST(MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_OUT_F32);
MQ_PL_INST(MQ_EPI_BIAS | MQ_EPI_RESIDUAL);
MQ_PL_INST(MQ_EPI_BIAS | MQ_EPI_LN_APPLY);
MQ_PL_INST(MQ_EPI_BIAS | MQ_EPI_GELU | MQ_EPI_LN_APPLY);
MQ_PL_INST(MQ_EPI_BIAS | MQ_EPI_QUICKGELU | MQ_EPI_LN_APPLY);
#endif
