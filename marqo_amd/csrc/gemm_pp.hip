// bf16 MFMA GEMM, one workgroup per CU with TWO accumulator sets ("pp" = ping-pong accumulators):
//   out[M,N] = epi( A[M,K] @ W[N,K]^T )      same contract, same bits as gemm_bf16.hip (K3 / K5 of SURVEY.md §8a)
//
// Why this form (profiles/r02d/r02f: what bounds the (32*MT)x128 kernel of gemm_bf16.hip at the towers' K = 768 shapes):
//   * two co-resident 160x128 workgroups stage 73.7 KB per k-step per CU through the 64 B/clk L1 -> LDS path and read 8 x 18
//     fragments from the LDS: ~1150 + 576 LDS cycles against 1280 matrix-pipe cycles — the k-loop is bound by the LDS, not by
//     the MFMAs, and every byte moved is power taken from the clock;
//   * a tile's epilogue (24-51 % of its life at K = 768) is only ever covered by the OTHER workgroup's k-loop, which on the
//     single-round shapes runs in lock-step with it.
// Here ONE workgroup of 4 wave64s owns the CU (one wave per SIMD, 512 registers each):
//   * (32*MT) x 256 x 64 tile, waves 2 x 2, each a (16*MT) x 128 sub-tile = MT x 8 v_mfma_f32_16x16x32_bf16 accumulators:
//     at MT = 5 a k-step stages 53 KB (-28 % per FLOP) and reads 4 x 26 fragments (-28 % per FLOP);
//   * the spare half of the register file holds a SECOND accumulator set: the last k-step of a tile writes its MFMA results
//     into set Y while set X restarts from the constant 0, and the tile's epilogue (bias / GELU / residual / convert / store,
//     in units of two 16x16 sub-tiles) is software-interleaved into the first k-steps of the workgroup's NEXT tile — under the
//     MFMAs instead of in front of them.  Stores and residual loads go through buffer descriptors, so row / column guards
//     are out-of-range offsets, not branches (the k-loop stays one basic block per half-step);
//   * 3-stage LDS ring (3 x (BM + 256) x 128 B <= 160 KiB), one barrier per k-step placed in its MIDDLE, counted vmcnt:
//       first half : 8*MT/2.. MFMAs on the kk = 0 fragments | ds_read the kk = 1 fragments | LDS-DMA of stage s+2
//       middle     : s_waitcnt vmcnt(pieces) lgkmcnt(0) ; s_barrier      (stage s+1 landed everywhere; buffer s-1 free)
//       second half: MFMAs on the kk = 1 fragments | ds_read stage s+1's kk = 0 fragments | an epilogue unit of the last tile
//     so a stage has 1.5 k-steps to land and no fragment read waits on the barrier it has just passed.
// LDS image, swizzle, operand order and accumulation order are those of gemm_bf16.hip: results are bit-identical
// (tests/test_gemm_variants_gpu.py).
#include <stdlib.h>
#include <string_view>
#include <type_traits>
#include <utility>
#include "common.h"

namespace {

constexpr int BN = 256, BK = 64;
constexpr int W_BYTES = BN * BK * 2;  // 32 KiB

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int I>
using ic = std::integral_constant<int, I>;

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(ic<Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// s_waitcnt vmcnt(V) lgkmcnt(0), expcnt untouched (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14])
template <int V>
__device__ __forceinline__ void wait_vm_lgkm0() {
    __builtin_amdgcn_s_waitcnt((V & 15) | 0x70 | ((V >> 4) << 14));
}

// PPS = epilogue units (pairs of 16x16 sub-tiles, one 16-byte bf16 store per lane) carried by one k-step of the next tile.
// WN = waves along N: 2 -> 4 waves (one per SIMD, 512 registers each, wave tile (16*MT) x 128), 4 -> 8 waves (two per SIMD, 256 registers
// each, wave tile (16*MT) x 64).  Measured (profiles/r03c_pp_ablation.txt): with ONE wave per SIMD nothing covers the wave's own LDS-DMA
// issue (the CU's address path takes 16 cycles per 1-KiB piece: 26 % of the k-step at 4096^3), its barrier (11 %) or the VALU of the
// interleaved epilogue (fc1 + GELU: +34 %) — an in-order wave cannot issue MFMAs while it is blocked in any of them.  Two waves per
// SIMD give every such slot to the partner's MFMAs, at the price of twice the fragment reads per MFMA.
template <int FLAGS, int MT, int PPS, int WN>
__global__ __launch_bounds__(128 * WN, WN / 2) void gemm_pp_kernel(
    const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ Wt, int64_t ldw,
    const float* __restrict__ bias, const void* residual, void* out, int64_t ldc,
    int M, int N, int K, int tiles_n, int num_tiles, int cgroup, int band_rows, int skew) {
    constexpr int BM = 32 * MT;
    constexpr int A_BYTES = BM * BK * 2;
    constexpr int STAGE = A_BYTES + W_BYTES;
    constexpr int WAVES = 2 * WN, NT = 16 / WN;
    static_assert((4 * MT) % WAVES == 0 && 32 % WAVES == 0, "staging pieces must deal evenly to the waves");
    constexpr int NPA = 4 * MT / WAVES, NPW = 32 / WAVES;
    constexpr int NP = NPA + NPW;         // LDS-DMA pieces (8 rows x 128 B) per wave per stage
    constexpr int NF = MT + NT;           // fragment reads per wave per 32-deep half
    constexpr int UPR = NT / 2;           // epilogue units per 16-row group
    constexpr int UNITS = MT * UPR;       // epilogue units per wave per tile
    constexpr int EC = UNITS / PPS;       // k-steps of the next tile that carry an epilogue unit group
    static_assert(UNITS % PPS == 0, "units must divide");
    constexpr bool HAS_BIAS = (FLAGS & MQ_EPI_BIAS) != 0;
    constexpr bool HAS_RES = (FLAGS & MQ_EPI_RESIDUAL) != 0;
    constexpr bool OUT_F32 = (FLAGS & MQ_EPI_OUT_F32) != 0;
    constexpr bool RES_BF16 = HAS_RES && !OUT_F32;
    constexpr int ES = OUT_F32 ? 4 : 2;   // bytes per output element
#ifdef MQ_PP_DIAG
    bool diag_live = false;   // ablation builds (tools/probes/build_pp_diag.sh): set once the prologue is done
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const bias_lds = (float*)(smem + 3 * STAGE);   // two slots of BN floats

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, g = lane >> 4;
    const int srow = lane >> 3, pch = lane & 7;

    // ---- XCD-aware, bijective block -> tile map with the L2-blocked order of gemm_bf16.hip ----------------------------------
    const int tq = num_tiles >> 3, tr = num_tiles & 7;
    const int tiles_m = (M + BM - 1) / BM;
    auto tile_origin = [&](int vbid, int& m0, int& n0) {
        const int xcd = vbid & 7, idx = vbid >> 3;
        const int tile = (xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq) + idx;
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

    // ---- staging sources: piece p of A (rows [8p, 8p+8)) belongs to wave p % 4, same for the 32 pieces of W.  lane -> (row = 8p + lane/8,
    // physical 16-B chunk = lane % 8) fetches logical chunk (lane % 8) ^ (row & 7): the swizzle lives on the SOURCE address (LDS-DMA
    // writes lane-linear) and again on the fragment reads.  32-bit byte offsets from a wave-uniform tile base.
    const char* a_base; const char* w_base;     // advance by 128 B per staged k-step
    unsigned a_voff[NPA], w_voff[NPW];
    auto set_sources = [&](int m0, int n0) {
        a_base = (const char*)(A + (int64_t)m0 * lda);
        w_base = (const char*)(Wt + (int64_t)n0 * ldw);
        const int mlim = M - 1 - m0, nlim = N - 1 - n0;
#pragma unroll
        for (int i = 0; i < NPA; ++i) {
            const int row = (i * WAVES + wave) * 8 + srow;
            a_voff[i] = (unsigned)min(row, mlim) * (unsigned)(lda * 2) + (unsigned)((pch ^ (row & 7)) * 16);
        }
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const int row = (j * WAVES + wave) * 8 + srow;
            w_voff[j] = (unsigned)min(row, nlim) * (unsigned)(ldw * 2) + (unsigned)((pch ^ (row & 7)) * 16);
        }
    };
    // piece q of this wave's NP (q < MT: A, else W) into ring buffer `buf`
    auto dma_piece = [&](int q, char* stage_base) {
#if defined(MQ_PP_DIAG) && (MQ_PP_DIAG & 1)   // ablation (timing only, wrong results): no global -> LDS traffic in the k-steps
        if (diag_live) return;
#endif
        if (q < NPA) glds16(a_base + a_voff[q], stage_base + (q * WAVES + wave) * 1024);
        else glds16(w_base + w_voff[q - NPA], stage_base + A_BYTES + ((q - NPA) * WAVES + wave) * 1024);
    };

    // ---- fragment read offsets (bytes inside a stage): row (l15), 16-B chunk (g + 4 kk) ^ (row & 7)
    const int sw0 = (g ^ (l15 & 7)) << 4, sw1 = ((g + 4) ^ (l15 & 7)) << 4;
    const int fa0 = (wm * (16 * MT) + l15) * 128 + sw0, fa1 = (wm * (16 * MT) + l15) * 128 + sw1;
    const int fw0 = A_BYTES + (wn * (16 * NT) + l15) * 128 + sw0, fw1 = A_BYTES + (wn * (16 * NT) + l15) * 128 + sw1;

    // ---- epilogue geometry of this lane inside a tile (gemm_epilogue.h): operands are fed swapped, so a lane owns out[m][n..n+3]
    const int m_wave = wm * (16 * MT) + l15;          // + mt * 16
    const int n_lane = wn * (16 * NT) + g * 4;        // + nt * 16
    const int n_wide = wn * (16 * NT) + (g & 1) * 16 + (g >> 1) * 8;   // + p * 32 : after the permlane16 exchange a lane owns 8 consecutive n

    const int nk = K / BK;
    int vb = blockIdx.x;
    int m0, n0;
    tile_origin(vb, m0, n0);
    set_sources(m0, n0);

    f32x4 X[MT][NT], Y[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) { X[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; Y[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    bf16x8 af0[MT], wf0[NT], af1[MT], wf1[NT];

    // previous tile (the one whose results sit in Y): buffer descriptors with num_records = 0 until there is one
    __amdgpu_buffer_rsrc_t p_out = __builtin_amdgcn_make_buffer_rsrc(out, 0, 0, 0x00020000);
    // (the residual epilogues run IN PLACE here: residual == out is a condition of mq_gemm_pp_plan — one descriptor serves both)
    int p_mrem = 0, p_nrem = 0;      // rows / columns of the previous tile inside the matrix
    int p_slot = 0;                   // bias slot of the previous tile
    auto describe_prev = [&](int tm0, int tn0) {
        const int64_t first = (int64_t)tm0 * ldc + tn0;
        const int64_t rem = ((int64_t)M * ldc - first) * ES;
        const int nrec = (int)(rem < 0x7fffffff ? rem : 0x7fffffff);
        p_out = __builtin_amdgcn_make_buffer_rsrc((char*)out + first * ES, 0, nrec, 0x00020000);
        p_mrem = M - tm0;
        p_nrem = N - tn0;
    };

    // Addressing of the epilogue: ONE lane-constant byte offset per access shape (voffset) + a wave-uniform per-unit delta in the buffer
    // instruction's scalar offset — the 16 units' offsets are never materialised in VGPRs (hoisted out of the tile loop as loop
    // invariants they cost 60+ registers and pushed the kernel into scratch).  A lane outside the matrix gets voffset 0x80000000: the
    // range check (voffset against num_records <= 0x7fffffff) drops its store / zeroes its load.
    const unsigned off_lane = (unsigned)(m_wave * (int)ldc + n_lane) * ES;        // sub-tile (0, 0) of this lane
    const unsigned off_wide = (unsigned)(m_wave * (int)ldc + n_wide) * 2u;        // bf16 pair store (after the permlane16 exchange)
    const int ldc16 = (int)ldc * 16;
    constexpr unsigned OOB = 0x80000000u;
    using res_raw_t = std::conditional_t<RES_BF16, u32x2, u32x4>;   // raw residual of one sub-tile (bf16: 4 values in 8 bytes)
    // raw residual of one epilogue unit (two sub-tiles): bf16 -> .x .y of each, fp32 -> all four
    auto res_load = [&](auto u_tag, res_raw_t (&r)[2]) {
        constexpr int U = decltype(u_tag)::value;
        constexpr int mt = U / UPR, p = U % UPR;
        const bool m_ok = m_wave < p_mrem - mt * 16;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool ok = m_ok && n_lane < p_nrem - (2 * p + h) * 16;
            // row-group delta in the scalar offset, column delta as a constant (folds into the instruction's 12-bit immediate)
            const int soff = mt * ldc16 * ES;
            const unsigned voff = (ok ? off_lane : OOB) + (unsigned)((2 * p + h) * 16 * ES);
            if constexpr (RES_BF16) r[h] = __builtin_amdgcn_raw_buffer_load_b64(p_out, voff, soff, 0);
            else r[h] = __builtin_amdgcn_raw_buffer_load_b128(p_out, voff, soff, 0);
        }
    };
    // bias of a unit's two sub-tiles from the previous tile's LDS slot
    auto bias_load = [&](auto u_tag, f32x4 (&b)[2]) {
        constexpr int U = decltype(u_tag)::value;
        constexpr int p = U % UPR;
#pragma unroll
        for (int h = 0; h < 2; ++h) b[h] = *(const f32x4*)(bias_lds + p_slot * BN + n_lane + (2 * p + h) * 16);
    };
    // one epilogue unit: sub-tiles (mt, 2p) and (mt, 2p+1) of Y -> bias / activation / residual -> store
    auto epi_unit = [&](auto u_tag, const res_raw_t (&r)[2], const f32x4 (&b)[2]) {
        constexpr int U = decltype(u_tag)::value;
        constexpr int mt = U / UPR, p = U % UPR;
        const bool m_ok = m_wave < p_mrem - mt * 16;
        f32x4 v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int nt = 2 * p + h;
            v[h] = Y[mt][nt];
            if (HAS_BIAS) v[h] += b[h];
            if (FLAGS & MQ_EPI_GELU) v[h] = gelu_erf4(v[h]);
            if (FLAGS & MQ_EPI_QUICKGELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[h][e] = quick_gelu(v[h][e]);
            }
            if constexpr (RES_BF16) {
                v[h] += f32x4{__uint_as_float(r[h].x << 16), __uint_as_float(r[h].x & 0xffff0000u), __uint_as_float(r[h].y << 16),
                              __uint_as_float(r[h].y & 0xffff0000u)};
            } else if constexpr (HAS_RES) {
                v[h] += f32x4{__uint_as_float(r[h].x), __uint_as_float(r[h].y), __uint_as_float(r[h].z), __uint_as_float(r[h].w)};
            }
        }
        if (OUT_F32) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bool ok = m_ok && n_lane < p_nrem - (2 * p + h) * 16;
                const u32x4 d = {__float_as_uint(v[h][0]), __float_as_uint(v[h][1]), __float_as_uint(v[h][2]), __float_as_uint(v[h][3])};
                __builtin_amdgcn_raw_buffer_store_b128(d, p_out, (ok ? off_lane : OOB) + (unsigned)((2 * p + h) * 64), mt * ldc16 * 4, 0);
            }
        } else {
            unsigned ax = pack_bf16x2(v[0][0], v[0][1]), ay = pack_bf16x2(v[0][2], v[0][3]);
            unsigned bx = pack_bf16x2(v[1][0], v[1][1]), by = pack_bf16x2(v[1][2], v[1][3]);
            const auto r0 = __builtin_amdgcn_permlane16_swap(ax, bx, false, false);
            const auto r1 = __builtin_amdgcn_permlane16_swap(ay, by, false, false);
            // (N % 4 == 0 and n % 8 == 0: a lane's 8 columns are inside the matrix, or exactly its upper 4 are not, or none is)
            const int nrem = p_nrem - p * 32;
            const bool full = m_ok && n_wide + 8 <= nrem, half = m_ok && n_wide + 4 <= nrem;
            const int soff = mt * ldc16 * 2;
            const u32x4 d = {r0[0], r1[0], r0[1], r1[1]};
            __builtin_amdgcn_raw_buffer_store_b128(d, p_out, (full ? off_wide : OOB) + (unsigned)(p * 64), soff, 0);
            const u32x2 dlo = {r0[0], r1[0]};
            __builtin_amdgcn_raw_buffer_store_b64(dlo, p_out, ((half && !full) ? off_wide : OOB) + (unsigned)(p * 64), soff, 0);
        }
    };

    // The four waves leave a barrier together and run identical code: every LDS-DMA piece and fragment read of one wave then collides with
    // the same instruction of the other three at the CU's single address path (16+ cycles per 1-KiB piece), and with one wave per SIMD a
    // wave blocked in VMEM issue is a matrix pipe without work.  `skew` x ~16 cycles x wave id of delay after each barrier de-phases them.
    auto wg_barrier = [&] {
#if defined(MQ_PP_DIAG) && (MQ_PP_DIAG & 4)   // ablation: no workgroup barrier in the k-steps
        if (diag_live) return;
#endif
        __builtin_amdgcn_s_barrier();
        // (one asm block with its own loop: a C++ loop here splits the k-step into more basic blocks and the register allocation with it)
        int cnt = wave * skew;
        asm volatile("s_cmp_eq_u32 %0, 0\n\t"
                     "s_cbranch_scc1 .Lskew_end%=\n"
                     ".Lskew_loop%=:\n\t"
                     "s_nop 7\n\t"
                     "s_sub_u32 %0, %0, 1\n\t"
                     "s_cmp_lg_u32 %0, 0\n\t"
                     "s_cbranch_scc1 .Lskew_loop%=\n"
                     ".Lskew_end%=:"
                     : "+s"(cnt) : : "scc");
    };

    // ---- prologue: stages 0 and 1 of the first tile, then the kk = 0 fragments of stage 0 --------------------------------------------
#pragma unroll
    for (int q = 0; q < NP; ++q) dma_piece(q, smem);
    a_base += 128; w_base += 128;
#pragma unroll
    for (int q = 0; q < NP; ++q) dma_piece(q, smem + STAGE);
    a_base += 128; w_base += 128;
    wait_vm_lgkm0<NP>();
    wg_barrier();
#pragma unroll
    for (int t = 0; t < NT; ++t) wf0[t] = *(const bf16x8*)(smem + fw0 + t * 2048);
#pragma unroll
    for (int t = 0; t < MT; ++t) af0[t] = *(const bf16x8*)(smem + fa0 + t * 2048);

#ifdef MQ_PP_DIAG
    diag_live = true;
#endif
    int sb = 0;                      // ring buffer of the current k-step
    int staged = 2;                  // k-steps of the CURRENT source tile already staged (its stages 0 .. staged-1)
    bool more = false;
    int nm0 = m0, nn0 = n0;

    // one k-step.  E = index of the epilogue unit group it carries (-1: none), FIRST: X restarts from 0, LAST: results go to Y.
    auto kstep = [&](auto e_tag, auto first_tag, auto last_tag, float bias_val, bool park_bias, int c_slot) {
        constexpr int E = decltype(e_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
        const char* cur = smem + sb * STAGE;
        const int sb1 = sb == 2 ? 0 : sb + 1, sb2 = sb == 0 ? 2 : sb - 1;
        const char* nxt = smem + sb1 * STAGE;
        char* dst = smem + sb2 * STAGE;       // buffer of stage s+2 == buffer of stage s-1: free since the previous step's barrier
        // ---- first half: kk = 0 MFMAs | this step's epilogue inputs (bias from LDS, residual) | read the kk = 1 fragments | stage s+2.
        // Source order = issue order wanted: the fragment reads come before the LDS-DMA pieces (LDS reads and LDS-DMA writes may alias
        // for the compiler, so their relative order is fixed by the source).
        res_raw_t rpre[PPS][2];
        f32x4 bpre[PPS][2];
        // (the bf16-residual epilogue of the 8-wave form is the one place the 256-register budget is short by a handful: its bias is
        // read in the second half instead of being carried across the barrier)
        constexpr bool BIAS_PRE = !(RES_BF16 && WN == 4);
        if constexpr (E >= 0 && HAS_BIAS && BIAS_PRE) {
            static_for<PPS>([&](auto i) { bias_load(ic<E * PPS + decltype(i)::value>{}, bpre[decltype(i)::value]); });
        }
        if constexpr (E >= 0 && HAS_RES) {   // ahead of the LDS-DMA pieces: complete at the middle wait
            static_for<PPS>([&](auto i) { res_load(ic<E * PPS + decltype(i)::value>{}, rpre[decltype(i)::value]); });
        }
#if defined(MQ_PP_DIAG) && (MQ_PP_DIAG & 2)   // ablation: no fragment reads in the k-steps (the prologue's fragments are reused)
#pragma unroll
        for (int t = 0; t < NT; ++t) { wf1[t] = wf0[t]; asm volatile("" : "+v"(wf1[t])); }
#pragma unroll
        for (int t = 0; t < MT; ++t) { af1[t] = af0[t]; asm volatile("" : "+v"(af1[t])); }
#else
#pragma unroll
        for (int t = 0; t < NT; ++t) wf1[t] = *(const bf16x8*)(cur + fw1 + t * 2048);
#pragma unroll
        for (int t = 0; t < MT; ++t) af1[t] = *(const bf16x8*)(cur + fa1 + t * 2048);
#endif
#pragma unroll
        for (int q = 0; q < NP; ++q) dma_piece(q, dst);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (FIRST) X[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[nt], af0[mt], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                else X[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[nt], af0[mt], X[mt][nt], 0, 0, 0);
            }
        {   // pin the interleave (masks: 0x8 MFMA, 0x100 DS read, 0x20 VMEM read, 0x10 VMEM): epilogue inputs up front, then one fragment read per
            // MFMA, then one LDS-DMA piece per MFMA, the rest of the MFMAs
            constexpr int NB = (E >= 0 && HAS_BIAS && BIAS_PRE) ? 2 * PPS : 0, NR = (E >= 0 && HAS_RES) ? 2 * PPS : 0;
            static_assert(NF + NP <= MT * NT, "one MFMA per staging instruction");
            if (NB) __builtin_amdgcn_sched_group_barrier(0x100, NB, 0);
            if (NR) __builtin_amdgcn_sched_group_barrier(0x020, NR, 0);
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, MT * NT - NF - NP, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- middle: stage s+1 landed (all but the NP newest VMEM operations — those of stage s+2 — are complete), kk = 1 fragments here,
        // every wave done with buffer s-1... and with the bias slot
#if defined(MQ_PP_DIAG) && (MQ_PP_DIAG & 8)   // ablation: no counted wait for the staged data (fragment reads still waited for)
        __builtin_amdgcn_s_waitcnt(0xC07F);   // vmcnt(63) lgkmcnt(0)
#else
        wait_vm_lgkm0<NP>();
#endif
        if (HAS_BIAS && park_bias) { if (tid < BN) bias_lds[c_slot * BN + tid] = bias_val; }
        wg_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- second half: kk = 1 MFMAs | read stage s+1's kk = 0 fragments | epilogue units of the previous tile
#if defined(MQ_PP_DIAG) && (MQ_PP_DIAG & 2)
#pragma unroll
        for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(wf0[t]));
#pragma unroll
        for (int t = 0; t < MT; ++t) asm volatile("" : "+v"(af0[t]));
#else
#pragma unroll
        for (int t = 0; t < NT; ++t) wf0[t] = *(const bf16x8*)(nxt + fw0 + t * 2048);
#pragma unroll
        for (int t = 0; t < MT; ++t) af0[t] = *(const bf16x8*)(nxt + fa0 + t * 2048);
#endif
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (LAST) Y[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[nt], af1[mt], X[mt][nt], 0, 0, 0);
                else X[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[nt], af1[mt], X[mt][nt], 0, 0, 0);
            }
#if defined(MQ_PP_DIAG) && (MQ_PP_DIAG & 16)   // ablation: no interleaved epilogue (the tail's units keep the accumulators live)
        if constexpr (false) {
#else
        if constexpr (E >= 0) {
#endif
            if constexpr (HAS_BIAS && !BIAS_PRE) {
                static_for<PPS>([&](auto i) { bias_load(ic<E * PPS + decltype(i)::value>{}, bpre[decltype(i)::value]); });
            }
            static_for<PPS>([&](auto i) { epi_unit(ic<E * PPS + decltype(i)::value>{}, rpre[decltype(i)::value], bpre[decltype(i)::value]); });
        }
        {
            constexpr int G = (MT * NT) / NF;
            constexpr int VPG = (E >= 0) ? ((FLAGS & (MQ_EPI_GELU | MQ_EPI_QUICKGELU)) ? 10 * PPS / 2 + 2 : 2 * PPS + 1) : 0;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, G, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (VPG) __builtin_amdgcn_sched_group_barrier(0x002, VPG, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, MT * NT - G * NF, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        sb = sb1;
    };
    // bookkeeping in front of a k-step whose first half stages the NEXT source stage: switch the sources at the tile seam
    auto advance_sources = [&]() {
        if (staged == nk) {   // the current source tile is fully staged: from here on the LDS-DMA fetches the workgroup's next tile
            if (more) set_sources(nm0, nn0);
            else { a_base -= (int64_t)nk * 128; w_base -= (int64_t)nk * 128; }   // nothing left: re-fetch this tile's first stages (never read)
            staged = 0;
        }
    };

    for (;;) {
        {   // the workgroup's next tile (its first two stages are fetched by this tile's last two k-steps)
            const int nvb = vb + gridDim.x;
            more = nvb < num_tiles;
            if (more) tile_origin(nvb, nm0, nn0);
        }
        const int c_slot = p_slot ^ 1;
        float bias_val = 0.f;
        if (HAS_BIAS && tid < BN && n0 + tid < N) bias_val = bias[n0 + tid];
        // ---- k-steps 0 .. EC-1 carry the previous tile's epilogue
        static_for<EC>([&](auto e) {
            constexpr int Ei = decltype(e)::value;
            advance_sources();
            kstep(ic<Ei>{}, std::bool_constant<Ei == 0>{}, std::false_type{}, bias_val, Ei == 1, c_slot);
            a_base += 128; w_base += 128; ++staged;
        });
        for (int kt = EC; kt < nk - 1; ++kt) {
            advance_sources();
            kstep(ic<-1>{}, std::false_type{}, std::false_type{}, 0.f, false, 0);
            a_base += 128; w_base += 128; ++staged;
        }
        advance_sources();
        kstep(ic<-1>{}, std::false_type{}, std::true_type{}, 0.f, false, 0);
        a_base += 128; w_base += 128; ++staged;
        // Y now holds this tile
        describe_prev(m0, n0);
        p_slot = c_slot;
        if (!more) break;
        vb += gridDim.x;
        m0 = nm0; n0 = nn0;
    }
    // ---- tail: the last tile's epilogue, nothing left to hide it under
    static_for<UNITS>([&](auto u) {
        res_raw_t r[2];
        f32x4 b[2];
        if (HAS_BIAS) bias_load(u, b);
        if (HAS_RES) res_load(u, r);
        epi_unit(u, r, b);
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

struct PpTune {
    int on, mt, pps, skew, waves;
    static int env(const char* k, int d) { const char* v = getenv(k); return v ? atoi(v) : d; }
    PpTune() : on(env("MQ_GEMM_PP", 0)), mt(env("MQ_GEMM_PP_MT", 0)), pps(env("MQ_GEMM_PP_PPS", 0)), skew(env("MQ_GEMM_PP_SKEW", 0)), waves(env("MQ_GEMM_PP_WAVES", 8)) {}
};
PpTune g_pp;

template <int FLAGS, int MT, int PPS, int WN>
int launch_pp(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const void* residual, void* out, int64_t ldc,
              int M, int N, int K, int cgroup_knob, hipStream_t s) {
    constexpr int BM = 32 * MT;
    constexpr int LDS = 3 * (BM * BK * 2 + W_BYTES) + 2 * BN * 4;
    static_assert(LDS <= 160 * 1024, "three stages must fit the CU's LDS");
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = mq_ensure_dyn_lds((const void*)gemm_pp_kernel<FLAGS, MT, PPS, WN>, LDS, attr_done); e != hipSuccess) {
        mq_set_error("mq_gemm_bf16(pp): hipFuncSetAttribute: %s", hipGetErrorString(e));
        return MQ_ERR_HIP;
    }
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int num_tiles = tiles_m * tiles_n;
    const int cgroup = (cgroup_knob > 0 && tiles_n > cgroup_knob && tiles_m >= 16) ? cgroup_knob : 0;
    const int band_rows = (tiles_m + 7) / 8;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        static int cached_cus[64] = {0};
        if (!cached_cus[dev & 63]) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cached_cus[dev & 63] = v;
            else cached_cus[dev & 63] = 256;
        }
        cus = cached_cus[dev & 63];
    }
    const int grid = num_tiles < cus ? num_tiles : cus;
    hipLaunchKernelGGL((gemm_pp_kernel<FLAGS, MT, PPS, WN>), dim3(grid), dim3(128 * WN), LDS, s, (const bf16_t*)A, lda, (const bf16_t*)W, ldw, bias,
                       residual, out, ldc, M, N, K, tiles_n, num_tiles, cgroup, band_rows, g_pp.skew);
    MQ_CHECK_LAUNCH("mq_gemm_bf16(pp)");
    return MQ_OK;
}

}  // namespace

// knobs (mq_tune): "gemm_pp" 0 = off, 1 = on where the cost model prefers it, 2 = wherever the shape is legal;
// "gemm_pp_pps" 2 (default) = the epilogue rides on 8 k-steps of the next tile, 4 = on 4; "gemm_pp_waves" 8 (default) / 4; "gemm_pp_skew"
void mq_gemm_pp_tune(const char* key, int value) {
    const std::string_view k(key);
    if (k == "gemm_pp") g_pp.on = value;
    else if (k == "gemm_pp_mt") g_pp.mt = value;
    else if (k == "gemm_pp_pps") g_pp.pps = value;
    else if (k == "gemm_pp_skew") g_pp.skew = value;
    else if (k == "gemm_pp_waves") g_pp.waves = value == 4 ? 4 : 8;
}
int mq_gemm_pp_mode() { return g_pp.on; }

// -> 0: not applicable (the caller falls back to the (32*MT) x 128 kernel), else the tile height in 32-row units.
// Only MT = 4 is instantiated: at MT = 5 the two 160-register accumulator sets + two fragment sets do not fit the 512 registers of a
// one-wave-per-SIMD kernel without scratch spills (hipcc: 256 VGPRs + 256 AGPRs + 21..109 spilled), and _lib.build() refuses scratch.
int mq_gemm_pp_plan(int M, int N, int K, int flags, bool inplace) {
    if (!g_pp.on) return 0;
    if ((flags & MQ_EPI_RESIDUAL) && !inplace) return 0;   // the kernel reads the residual through the output's buffer descriptor
    const int nk = K / BK;
    if (nk < 5 || N < 256 || M < 64) return 0;             // MT*4/PPS epilogue-carrying k-steps + the last one must exist: nk >= 4 + 1
    // QUICKGELU: the interleaved epilogue returns wrong swapped halves next to the division sequences (profiles/r03b_pp_diag.txt) — not routed here
    if (flags & MQ_EPI_QUICKGELU) return 0;
    // the fp32-residual epilogue at 4 units per k-step (needed below 9 k-steps) is the one instantiation that spills: not built
    const bool f32res = flags == (MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_OUT_F32), f32bias = flags == (MQ_EPI_BIAS | MQ_EPI_OUT_F32);
    if ((f32res || (f32bias && g_pp.waves != 4)) && (nk < 9 || g_pp.pps == 4)) return 0;
    if (g_pp.on == 1) {
        // cost model against the two-workgroups-per-CU kernel: one-per-CU tiles of 128 x 256 in rounds of 256; a ragged last round
        // costs a whole round.  Take the big tile when its rounds are at least 70 % full.
        const int64_t tiles = (int64_t)((M + 127) / 128) * ((N + BN - 1) / BN);
        const int64_t rounds = (tiles + 255) / 256;
        if ((double)tiles / (double)(rounds * 256) < 0.70) return 0;
    }
    return 4;
}

template <int FLAGS>
int mq_launch_gemm_pp(int mt, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const void* residual, void* out,
                      int64_t ldc, int M, int N, int K, int cgroup_knob, hipStream_t s) {
    const int nk = K / BK;
    // k-steps of the next tile that carry the epilogue: 8 where the k-loop is long enough (lighter steps), else 4  ("gemm_pp_pps" 2 / 4)
    bool light = g_pp.pps != 4;
    if (8 + 1 > nk) light = false;
    constexpr bool F32RES = FLAGS == (MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_OUT_F32);
    if (g_pp.waves == 4) {
        if (light) return launch_pp<FLAGS, 4, 2, 2>(A, lda, W, ldw, bias, residual, out, ldc, M, N, K, cgroup_knob, s);
        if constexpr (F32RES) {
            mq_set_error("mq_gemm_bf16(pp): fp32-residual epilogue needs K >= 576");   // (mq_gemm_pp_plan never sends it here)
            return MQ_ERR_INVALID;
        } else {
            return launch_pp<FLAGS, 4, 4, 2>(A, lda, W, ldw, bias, residual, out, ldc, M, N, K, cgroup_knob, s);
        }
    }
    if (light) return launch_pp<FLAGS, 4, 1, 4>(A, lda, W, ldw, bias, residual, out, ldc, M, N, K, cgroup_knob, s);
    if constexpr (F32RES || FLAGS == (MQ_EPI_BIAS | MQ_EPI_OUT_F32)) {   // (the two 8-wave instantiations that spill: not built)
        mq_set_error("mq_gemm_bf16(pp): this fp32 epilogue needs K >= 576");   // (mq_gemm_pp_plan never sends it here)
        return MQ_ERR_INVALID;
    } else {
        return launch_pp<FLAGS, 4, 2, 4>(A, lda, W, ldw, bias, residual, out, ldc, M, N, K, cgroup_knob, s);
    }
}

#define MQ_PP_INST(F)                                                                                                              \
    template int mq_launch_gemm_pp<(F)>(int, const void*, int64_t, const void*, int64_t, const float*, const void*, void*, int64_t, \
                                        int, int, int, int, hipStream_t)
MQ_PP_INST(0);
MQ_PP_INST(MQ_EPI_OUT_F32);
MQ_PP_INST(MQ_EPI_BIAS | MQ_EPI_OUT_F32);
MQ_PP_INST(MQ_EPI_BIAS);
MQ_PP_INST(MQ_EPI_BIAS | MQ_EPI_GELU);
MQ_PP_INST(MQ_EPI_BIAS | MQ_EPI_QUICKGELU);
MQ_PP_INST(MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_OUT_F32);
MQ_PP_INST(MQ_EPI_BIAS | MQ_EPI_RESIDUAL);
