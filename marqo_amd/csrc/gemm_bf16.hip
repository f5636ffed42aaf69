// bf16 MFMA GEMM with fused epilogues for gfx950 — K3 / K5 / K1-GEMM / K6 / K8 of SURVEY.md §8a; the one bf16 main loop of the library
// (round 4: the software-pipelined loop replaced the round 1-3 kernel and its experimental siblings; history in DESIGN.md appendix).
// Reference call site: /root/reference/src/marqo/core/inference/embedding_models/open_clip_model.py:249-266 (encode_image / encode_text).
//
//   out[M,N] = epi( A[M,K] @ W[N,K]^T )        A, W bf16, K-contiguous ("NT": W is the PyTorch nn.Linear weight as stored)
//
// Design (MI355X-first, not a CUDA tiling):
//   * (32*MT)x128x64 block tile (MT = 2/4/5/6 -> 64..192 rows), 4 wave64s as 2x2, each wave a (16*MT)x64 sub-tile = MT x 4
//     v_mfma_f32_16x16x32_bf16 accumulators; two workgroups per CU; PERSISTENT: the grid is the 512 resident slots, every workgroup
//     walks tiles bid, bid + 512, ... of an XCD-aware, L2-blocked order (block b runs on XCD b % 8; each XCD takes a contiguous band of
//     tiles whose A panels + W column tiles fit its 4 MiB L2);
//   * global -> LDS by LDS-DMA through a buffer descriptor (buffer_load_dwordx4 ... lds: 32-bit per-lane voffset + scalar k offset; 1 KiB per
//     wave instruction), 2 stages.  LDS tiles are [rows][128 B]; the 16-B chunk index is XOR-swizzled with (row & 7): LDS-DMA writes
//     lane-linear, so the swizzle sits on each lane's SOURCE address and again on the ds_read_b128 side (same involution) -> conflict-free;
//   * the k-loop is software-pipelined INSIDE the wave: fragments are double-buffered in registers by k-half — while the 4*MT MFMAs of half h
//     run, the ds_read_b128s of the next half are issued between them, so no MFMA waits on LDS latency — with ONE workgroup barrier per
//     k-step, placed MID-step: [MFMA(kk=0) || read kk=1] -> vmcnt(0) + barrier -> [MFMA(kk=1) || read kk=0 of the next stage || LDS-DMA of the
//     stage after next].  After the mid-step barrier the current LDS buffer is dead (both halves are in registers), so the DMA that refills it
//     is issued a full k-step before its data is needed;
//   * the fragment reads are inline-asm ds_read_b128: hipcc's waitcnt pass cannot tell an LDS read from the LDS-DMA writes in flight (no alias
//     scopes on a dynamic __shared__ array) and would put s_waitcnt vmcnt(0) in front of every read that follows a DMA, serialising the
//     pipeline; the waits are placed by hand (lgkmcnt(0) before the first consumer, vmcnt(0) only at the mid-step barrier) — tests/test_gemm_isa.py
//     checks on the compiled ISA that no asm-read register is touched before its wait;
//   * the DMA runs on its own cursor (tile, k) two stages ahead of the MFMAs and walks on into the workgroup's next tile (the next tile's
//     first fragments are in registers before the epilogue starts), for any K / 64 >= 1; past the last tile the descriptors' sizes drop to 0,
//     which keeps the k-step ONE straight-line body;
//   * operands are fed swapped (mfma(Wfrag, Afrag)) so each lane ends up owning 4 CONSECUTIVE n of one output row: bias / residual / out are
//     8/16-byte vector accesses (gemm_epilogue.h).
// Measured against the round 1-3 loop (bit-identical results, profiles/r04a_gemm_pl_check.txt, r04c_fold_pl_ab.txt): stand-alone -7 % at 8192^3,
// -5.7 % ViT-L/14 fc1, +2 % at the ViT-B/32 shapes; inside the towers (cold weights, real launch sequence) +2.9 % ViT-B/32, +2.6 % ViT-L/14,
// +1.9 % CLIP text embeddings/s.
#include <stdlib.h>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include "common.h"
#include "gemm_epilogue.h"
#include "gemm_loop.h"

extern mq_knob mq_gemm_fp8_force_mt;  // gemm_fp8.hip
extern mq_knob mq_gemm_fp8_big;       // gemm_fp8.hip: 0 = plan, 1 = never the big tile, 3 = always
extern mq_knob mq_tower_row_select;   // towers.hip
extern mq_knob mq_tower_ln_fold;      // towers.hip
extern mq_knob mq_tower_subln_fold;
extern mq_knob mq_tower_attn_proj;
extern mq_knob mq_tower_panel_gemm;
extern mq_knob mq_attention_waves;    // attention.hip
extern mq_knob mq_tower_residual_bf16;  // towers.hip
extern mq_knob mq_gemm_small_max_rows;  // gemm_small.hip
bool mq_gemm_small_ok(int64_t M, int64_t N, int64_t K, bool ln);
bool mq_gemm_small_grouped_ok(int64_t M, int64_t N, int64_t K);
extern mq_knob mq_gemm_small_group_rows;
extern mq_knob mq_ln_prefetch;        // rowops.hip
int mq_gemm_small(const void* d_A, int64_t lda, const void* d_W, int64_t ldw, const float* d_bias, const void* d_residual, void* d_out,
                  int64_t ldc, int64_t M, int64_t N, int64_t K, int flags, hipStream_t s);
// gemm_wd.hip: the W-direct main loop on the same tile plan (-1: combination not instantiated, the caller launches its own kernel)
int mq_gemm_wd_launch(int flags, int mt, int ns, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* residual, void* out,
                      int64_t ldc, int M, int N, int K, int tiles_n, int num_tiles, int cgroup, int band_rows, int grid, int wide, unsigned a_bytes,
                      unsigned w_bytes, const GemmLn& ln, hipStream_t s);

int mq_device_ok();   // runtime.hip

namespace {

// The rows that do not fill a 256-row tile (257 tokens per ViT-L/14 image never make a multiple of 256) — the TAIL of the big-tile kernel (WM = 4).
// A second launch for them is pure latency (tiles_n workgroups walking K / 64 dependent k-steps: 13-33 us, what the big tile saves per GEMM:
// profiles/r05g) and extra launches for a split-K of them cost more still (r05h).  Inside the SAME launch they are almost free: when the workgroups have
// finished their full tiles, the one ragged row of tiles is cut along K over ALL of them — workgroup b takes tile b % tiles_n, k-range b / tiles_n
// of `tail_splits` (a power of two <= 8) — a few k-steps each.  Every range leaves its fp32 accumulators in its slot of `partials`, raises its flag and
// waits for the flags of the tile's other ranges (every workgroup of the grid is resident and in this phase: nothing can be waited for that has
// not started); then the ranges share out the tile's sum and epilogue by regions (below): partials added in range order (deterministic).  Flags carry a per-launch epoch (never reset);
// a wait of more than ~1 s gives up and raises flags[gridDim.x] instead of hanging the device.  Rows of the tail differ from the tile kernels'
// results by the fp32 association of the k-sum only.
struct GemmSk {
    float* partials;       // [gridDim.x][BM * BN] fp32, one slot per workgroup
    unsigned* flags;       // [gridDim.x] epoch of the partial that workgroup posted last; [gridDim.x] = error word
    unsigned epoch;
    int tail_splits;       // 0 = no tail (M is a multiple of BM, or not the big tile)
};

// NH: 64-column halves of a wave's sub-tile.  NH = 1: the (32*MT)x128 block tile, two workgroups per CU (the towers' short-K shapes).  NH = 2 (round 5):
// the WIDE tile, (32*MT)x256 — at MT = 8 the 256x256x64 macro-tile with 128x128 per wave: 64 MFMAs per 16 fragment reads and per 16 LDS-DMA
// pieces where the 160x128 tile has 20 per 9 and 9 — one workgroup per CU (128 KiB of LDS, 256 accumulator registers per lane: the allocator
// puts them in AGPRs), everything else — staging, swizzle, the register double-buffering, the hand-placed waits — is the same code.
// ORD: order of the second half's side work — 0: LDS-DMA pieces first, then the next stage's fragment reads; 1: reads first; 2: alternating
// WM: waves along M (the workgroup is WM x 2 wave64s).  WM = 2: the 4-wave kernels above.  WM = 4 (round 5, second attempt at the big tile): 8 waves,
// (64*MT) x (128*NH) — at MT = 4, NH = 2 the 256x256x64 macro-tile as 64x128 per wave, ONE workgroup per CU but TWO waves per SIMD again: what
// sank the 4-wave wide tile (profiles/r05a) is that a wave alone on its SIMD has nobody to cover its LDS-DMA issue stalls; here a wave issues
// 8 pieces per 64 MFMAs (160x128: 9 per 40) and its partner on the SIMD computes meanwhile.
template <int FLAGS, int MT, int NH, int WM, int ORD>
__global__ __launch_bounds__(128 * WM, (NH == 1 && WM == 2) ? 2 : 1) void gemm_nt_kernel(
    const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ Wt, int64_t ldw,
    const float* __restrict__ bias, const float* residual, void* out, int64_t ldc,
    int M, int N, int K, int tiles_n, int num_tiles, int cgroup, int band_rows, int wide_store,
    unsigned a_bytes, unsigned w_bytes, GemmLn ln, GemmSk sk) {
    static_assert(!(NH == 2 && WM == 2), "the 4-wave 224 x 256 tile (one wave per SIMD) lost on every shape (profiles/r05a, r05b) and was removed in round 6");
    constexpr int BM = 16 * MT * WM, BN = 128 * NH;
    constexpr bool TAIL = WM == 4;   // the big tile handles a ragged last row of tiles in-kernel (GemmSk)
    constexpr int NTW = 4 * NH;     // 16-column W sub-tiles per wave (a wave spans half of BN)
    constexpr int A_TILE_BYTES = BM * BK * 2, W_TILE_BYTES = BN * BK * 2;
    constexpr int STAGE_BYTES = A_TILE_BYTES + W_TILE_BYTES;
    constexpr int NPA = MT, NPW = 8 * NH / WM;   // LDS-DMA pieces (8 rows = 1 KiB each) per wave per stage: A rows BM / (2 WM), W rows BN / (2 WM)
    constexpr int NLD = NPA + NPW;  // ... in all
    constexpr int NLR = MT + NTW;   // fragment reads per wave per k-half (= NLD in the 4-wave kernels)
    constexpr int NM = NTW * MT;    // MFMAs per wave per k-half
    // residual rows prefetched together by the epilogue (16-row units): the next tile's first fragments are live across it
    // (LN_APPLY on top of the residual epilogue — mq_gemm_bf16_lnrs — holds 16 colsum + 2 MT statistics registers more: smaller residual groups)
    constexpr bool LN_RES = (FLAGS & MQ_EPI_LN_APPLY) && (FLAGS & MQ_EPI_RESIDUAL);
    constexpr int ERG = !(FLAGS & MQ_EPI_RESIDUAL) ? MT : LN_RES ? ((NH == 2 || WM == 4) ? 1 : MT <= 4 ? MT : 2) : (NH == 2 || WM == 4) ? MT
                        : (FLAGS & MQ_EPI_OUT_F32) ? (MT <= 3 ? MT : (MT + 1) / 2) : (MT <= 5 ? MT : 3);
    extern __shared__ __attribute__((aligned(16))) char smem[];

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

    // ---- staging: wave w owns A rows [8*MT*w, 8*MT*(w+1)) and W rows [8*NPW*w, 8*NPW*(w+1)) of a stage, 8 rows per LDS-DMA piece.
    // lane -> (row = base + lane/8, physical 16-B chunk = lane%8); it fetches logical chunk (lane%8) ^ (row&7) of that row, so physical
    // chunk p of row r holds logical chunk p ^ (r&7) (the swizzle lives on the SOURCE address; the LDS image is lane-linear).
    const int srow = lane >> 3;
    const unsigned chunk_off = (unsigned)(((lane & 7) ^ (srow & 7)) * 16);   // (row & 7) == (srow & 7): piece bases are multiples of 8 rows
    unsigned a_vo[NPA], w_vo[NPW];
    auto set_sources = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            int gm = m0 + wave * (8 * MT) + i * 8 + srow; gm = gm < M ? gm : M - 1;
            a_vo[i] = (unsigned)gm * (unsigned)lda * 2u + chunk_off;
        }
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            int gn = n0 + wave * (8 * NPW) + i * 8 + srow; gn = gn < N ? gn : N - 1;
            w_vo[i] = (unsigned)gn * (unsigned)ldw * 2u + chunk_off;
        }
    };
    const int nk = K / BK;
    // DMA cursor: (tile d_vbid, k-step d_k) of the next stage to request; runs two stages ahead of the MFMAs.  Once it has walked past
    // the workgroup's last tile the descriptors' sizes drop to 0: the (two) trailing requests are then out of range for every lane — no
    // memory traffic — which keeps the k-step a single straight-line body without a "nothing left to prefetch" variant.
    int d_vbid = blockIdx.x, d_k = 0;
    unsigned a_rec = a_bytes, w_rec = w_bytes;
    bool tail_mode = false;   // TAIL: the cursor walks ONE k-range of a tail tile (d_left steps) instead of whole tiles
    int d_left = 0;
    const bool has_main = !TAIL || (int)blockIdx.x < num_tiles;   // (TAIL: the grid is always the 256 workgroups; few full tiles may leave some without one)
    if (has_main) {
        int m0, n0;
        tile_origin(d_vbid, m0, n0);
        set_sources(m0, n0);
    } else {
        a_rec = 0; w_rec = 0;
#pragma unroll
        for (int i = 0; i < NPA; ++i) a_vo[i] = 0;
#pragma unroll
        for (int i = 0; i < NPW; ++i) w_vo[i] = 0;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned dma_a0 = lds0 + (unsigned)wave * (8 * MT * 128), dma_w0 = lds0 + A_TILE_BYTES + (unsigned)wave * (8 * NPW * 128);   // scalars
    auto issue_piece = [&](int i, unsigned bufoff) {
        const unsigned soff = (unsigned)d_k * (BK * 2);
        if (i < NPA) dma16(__builtin_amdgcn_make_buffer_rsrc((void*)A, 0, a_rec, 0x00020000), a_vo[i < NPA ? i : 0], soff, dma_a0 + bufoff + (unsigned)i * 1024u);
        else dma16(__builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, w_rec, 0x00020000), w_vo[i >= NPA ? i - NPA : 0], soff, dma_w0 + bufoff + (unsigned)(i - NPA) * 1024u);
    };
    auto advance_cursor = [&]() {
        if (TAIL && tail_mode) {
            ++d_k;
            if (--d_left <= 0) { a_rec = 0; w_rec = 0; }   // past the k-range: out-of-range requests, no traffic
            return;
        }
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
    const unsigned w_row = lds0 + A_TILE_BYTES + (unsigned)((wn * (16 * NTW) + l15) * 128);
    const unsigned aB0 = a_row + sw0, aB1 = a_row + sw1, wB0 = w_row + sw0, wB1 = w_row + sw1;

    f32x4 acc[NH][MT][4];
    bf16x8 wf0[NTW], af0[MT], wf1[NTW], af1[MT];
    // one fragment read of a k-half: piece p < NTW -> W sub-tile p, else A sub-tile p - NTW (offsets t * 16 rows * 128 B)
    auto read_piece = [&](auto p_tag, unsigned wbase, unsigned abase, bf16x8 (&wf)[NTW], bf16x8 (&af)[MT]) {
        constexpr int P = decltype(p_tag)::value;
        if constexpr (P < NTW) wf[P] = lds_read16<P * 2048>(wbase);
        else af[P - NTW] = lds_read16<(P - NTW) * 2048>(abase);
    };

    // ---- prologue: the workgroup's first two stages, then the kk = 0 fragments of the first (a lambda: the tail phase primes the ring again) ---------
    unsigned bufoff = 0;   // LDS byte offset of the stage the next k-step consumes
    auto prime = [&]() {
        bufoff = 0;
#pragma unroll
        for (int i = 0; i < NLD; ++i) issue_piece(i, 0);
        advance_cursor();
#pragma unroll
        for (int i = 0; i < NLD; ++i) issue_piece(i, STAGE_BYTES);
        advance_cursor();
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");   // stage 0 landed (loads retire in issue order)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        static_for<NLR>([&](auto p_tag) { read_piece(p_tag, wB0, aB0, wf0, af0); });
    };
    if (has_main) prime();

    int c_vbid = blockIdx.x;

    // one k-step on the stage at `bufoff`; on entry (wf0, af0) hold (or are about to receive) its kk = 0 fragments
    auto kstep = [&]() {
        // -------- first half: MFMAs on (wf0, af0); reads of (wf1, af1) between them
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (wf0, af0) landed
#pragma unroll
        for (int t = 0; t < NTW; ++t) landed(wf0[t]);
#pragma unroll
        for (int t = 0; t < MT; ++t) landed(af0[t]);
        __builtin_amdgcn_sched_barrier(0);
        {
            const unsigned wb = wB1 + bufoff, ab = aB1 + bufoff;
            constexpr int NS1 = NLR;  // side work of the first half: the NLR fragment reads of kk = 1
            static_for<NM>([&](auto idx_tag) {
                constexpr int idx = decltype(idx_tag)::value, mt = idx / NTW, nt = idx % NTW;
                acc[nt / 4][mt][nt % 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[nt], af0[mt], acc[nt / 4][mt][nt % 4], 0, 0, 0);
                constexpr int RG = NM / NLR > 0 ? NM / NLR : 1;   // one read per RG MFMAs; the last MFMA flushes whatever is left
                constexpr int lo = idx == 0 ? 0 : (idx / RG < NS1 ? idx / RG : NS1);
                constexpr int hi = idx == NM - 1 ? NS1 : ((idx + 1) / RG < NS1 ? (idx + 1) / RG : NS1);
                static_for<hi - lo>([&](auto p_tag) {
                    constexpr int it = lo + decltype(p_tag)::value;
                    __builtin_amdgcn_sched_barrier(0);
                    read_piece(std::integral_constant<int, it>{}, wb, ab, wf1, af1);
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        }
        // -------- mid-step: the stage after this one has landed (my pieces), my reads of this buffer are done; after the barrier both
        // hold for every wave: the next stage may be read, this buffer may be refilled
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) — as a builtin: the compiler's own scoreboard must see that nothing is pending
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < NTW; ++t) landed(wf1[t]);
#pragma unroll
        for (int t = 0; t < MT; ++t) landed(af1[t]);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // -------- second half: MFMAs on (wf1, af1); side work: the NLD DMA pieces of the stage after next (into the buffer this step
        // just finished with) and the NLR reads of the next stage's kk = 0 fragments (the next tile's first stage at a tile's last step;
        // stale bytes nobody uses at the workgroup's very last step)
        {
            const unsigned nb = bufoff ^ (unsigned)STAGE_BYTES;
            const unsigned wb = wB0 + nb, ab = aB0 + nb;
            constexpr int NSIDE = NLD + NLR;
            static_for<NM>([&](auto idx_tag) {
                constexpr int idx = decltype(idx_tag)::value, mt = idx / NTW, nt = idx % NTW;
                acc[nt / 4][mt][nt % 4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[nt], af1[mt], acc[nt / 4][mt][nt % 4], 0, 0, 0);
                constexpr int lo = idx == 0 ? 0 : (idx * NSIDE) / NM;                         // side items due after this MFMA: [lo, hi)
                constexpr int hi = idx == NM - 1 ? NSIDE : ((idx + 1) * NSIDE) / NM;
                static_for<hi - lo>([&](auto it_tag) {
                    constexpr int it = lo + decltype(it_tag)::value;
                    __builtin_amdgcn_sched_barrier(0);
                    // its kind and its number among the items of that kind; ORD 2 with unequal counts (8 waves): evenly spread (Bresenham)
                    constexpr int dma_before = (it * NLD) / NSIDE;
                    constexpr bool is_dma = ORD == 0 ? it < NLD : ORD == 1 ? it >= NLR : NLD == NLR ? (it & 1) == 0 : ((it + 1) * NLD) / NSIDE > dma_before;
                    constexpr int ord = ORD == 0 ? (is_dma ? it : it - NLD) : ORD == 1 ? (is_dma ? it - NLR : it) : NLD == NLR ? it / 2 : (is_dma ? dma_before : it - dma_before);
                    if constexpr (is_dma) issue_piece(ord, bufoff);
                    else read_piece(std::integral_constant<int, ord>{}, wb, ab, wf0, af0);
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        }
        advance_cursor();
        bufoff ^= (unsigned)STAGE_BYTES;
    };

    if (has_main) for (;;) {
        int cm0, cn0;
        tile_origin(c_vbid, cm0, cn0);
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt) kstep();
        // the compiler takes an asm's outputs as valid once the statement has executed: retire the last fragment reads before any code it
        // may place behind the loop (register copies at the tile boundary) can touch them
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < NTW; ++t) landed(wf0[t]);
#pragma unroll
        for (int t = 0; t < MT; ++t) landed(af0[t]);

        // (wide tile: the lane's row / column ids pass through an empty asm, so that the epilogue's address arithmetic is redone per tile instead of
        // being hoisted out of the tile loop — kept live across the k-loop next to 128 fragment registers it spilled)
        int l15e = l15, ge = g;
        if (NH == 2 || WM == 4) asm volatile("" : "+v"(l15e), "+v"(ge));
        static_for<NH>([&](auto h_tag) {
            constexpr int h = decltype(h_tag)::value;
            gemm_epilogue<FLAGS, MT, ERG, true>(acc[h], bias, residual, out, ldc, M, N, cm0 + wm * (16 * MT), cn0 + wn * (16 * NTW) + h * 64, l15e, ge, wide_store != 0, &ln, nullptr);
        });
        if constexpr ((FLAGS & MQ_EPI_ROW_STATS) != 0) {
            // in-launch finalise of the row statistics (GemmLn::band_ctr): behind a workgroup's FIRST tile the carried weight prefetch goes out
            // (its cold loads drain with the tile's stores), then every wave arrives at its row band's counter
            if (ln.band_ctr) {
                unsigned pf_regs[2] = {0u, 0u};
                if (c_vbid == (int)blockIdx.x) gemm_pf_issue(ln, pf_regs);
                gemm_band_arrive(ln, (cm0 / BM) * WM + wm, cm0 + wm * (16 * MT), 16 * MT, M, lane);
                landed(pf_regs[0]); landed(pf_regs[1]);   // (retired by the arrival's vmcnt(0); nothing may reuse the registers before)
            }
        }

        c_vbid += gridDim.x;
        if (c_vbid >= num_tiles) break;
        if (WM == 4) {
            // 8-wave big tile: 128 accumulator + 96 fragment registers leave the epilogue no room when the next tile's first fragments are held
            // across it (the 4-wave kernels do that).  Here the copy the last k-step fetched is dropped (retired and dead behind `landed` above) and
            // the 12 reads are issued again now — the next tile's first stage sits untouched in the buffer at `bufoff` until its mid-step barrier;
            // one exposed LDS latency per tile.
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            static_for<NLR>([&](auto p_tag) { read_piece(p_tag, wB0 + bufoff, aB0 + bufoff, wf0, af0); });
        }
    }
    // the trailing (out-of-range) LDS-DMA requests must have retired before the workgroup's LDS can be handed to another one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    if constexpr (TAIL) {
        // (the compiler routes the tile loop's back edge through its loop-invariant guard, so STATICALLY the re-read block above can fall into this
        // phase with its ds_reads in flight; it never does, and this wait makes that visible to tests/test_gemm_isa.py's control-flow walk)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ---- the tail: the ragged last row of tiles, cut along K over the whole grid (see GemmSk) ---------------------------------------------------
        const int S = sk.tail_splits;
        const int t_tile = (int)blockIdx.x % tiles_n, t_split = (int)blockIdx.x / tiles_n;
        if (S <= 0 || t_split >= S) return;
        const int k0 = (int)((int64_t)t_split * nk / S), k1 = (int)((int64_t)(t_split + 1) * nk / S);
        const int cm0 = (M / BM) * BM, cn0 = t_tile * BN;
        __builtin_amdgcn_s_barrier();      // every wave is done with the main phase's LDS ring
        tail_mode = true;
        d_k = k0;
        d_left = k1 - k0;
        a_rec = a_bytes; w_rec = w_bytes;
        set_sources(cm0, cn0);
        prime();
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // (do-while: the host gives every range at least one k-step; a zero-trip path would merge an all-zero accumulator file with the loop's and the
        // allocator answers that with 128 register copies and spills)
        {
            int kt = k0;
            do { kstep(); } while (++kt < k1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < NTW; ++t) landed(wf0[t]);
#pragma unroll
        for (int t = 0; t < MT; ++t) landed(af0[t]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing out-of-range requests
        // a lane's 32 accumulator quads of a slot: [wave][quad][lane] x 16 B — every store / load instruction moves 1 KiB contiguous per wave
        constexpr int QUADS = NH * MT * 4;
        constexpr size_t SLOT_QUADS = (size_t)(BM * BN / 4);
        {
            // every range leaves its accumulators in its slot and raises its flag ...
            f32x4* dst = (f32x4*)sk.partials + (size_t)blockIdx.x * SLOT_QUADS + (size_t)wave * (QUADS * 64) + lane;
            static_for<NH * MT>([&](auto hi_tag) {   // 4 stores (immediate offsets 0 .. 3 KiB) per pointer, then the pointer moves on
                constexpr int h = decltype(hi_tag)::value / MT, i = decltype(hi_tag)::value % MT;
                asm volatile("" : "+v"(dst));
                // (agent-scope relaxed atomics = write-through stores: the partials reach the memory side without the L2 write-back of a release
                // fence — which at this point would flush the whole main phase's output — and the readers below bypass their L2 the same way)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) __hip_atomic_store((float*)(dst + j * 64) + e, acc[h][i][j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                dst += 256;
            });
            // my (write-through, agent-scope) stores have been acknowledged before the flag goes out — as inline asm: the compiler may drop the wait
            // of a fence whose scoreboard it believes empty (MI355X_MICROARCH.md, "Compiler hazard"); no cache maintenance needed for sc1 stores
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (tid == 0) {
                __hip_atomic_store(sk.flags + blockIdx.x, sk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // ... and waits for the flags of the tile's other ranges (all of them are running this very phase)
                for (int r2 = 0; r2 < S; ++r2) {
                    const unsigned* f = sk.flags + (r2 * tiles_n + t_tile);
                    int spins = 0;
                    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sk.epoch) {
                        __builtin_amdgcn_s_sleep(4);
                        if (++spins > (1 << 22)) {   // ~1 s: never hang the device; the launch's result is then wrong and says so
                            __hip_atomic_store(sk.flags + gridDim.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                }
            }
            __builtin_amdgcn_s_barrier();
        }
        // The sum and the epilogue are shared out too (one owner per tile adding S partials of 256 KB through one CU took longer than the tile): the
        // tile is the 8 regions its 8 waves computed (64 x 128 each); range s finishes regions s, s + S, ..., and inside a workgroup wave w takes the
        // 16 x 64 piece (h, i) = (w / MT, w % MT) of the region — 4 quads per partial and lane, added in range order (deterministic) into the layout of
        // a one-row-group accumulator, which the very same gemm_epilogue finishes.
        static_assert(MT * NH == 8 && WM == 4, "the tail shares a tile out as 8 regions x 8 pieces");
        const int ph = wave / MT, pi = wave % MT;
        int l15e = l15, ge = g;
        asm volatile("" : "+v"(l15e), "+v"(ge));
        for (int region = t_split; region < 8; region += S) {
            f32x4 piece[1][4] = {{f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}};
            for (int r2 = 0; r2 < S; ++r2) {
                const f32x4* src = (const f32x4*)sk.partials + (size_t)(r2 * tiles_n + t_tile) * SLOT_QUADS + (size_t)region * (QUADS * 64)
                                   + (size_t)((ph * MT + pi) * 4) * 64 + lane;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) piece[0][j][e] += __hip_atomic_load((const float*)(src + j * 64) + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            gemm_epilogue<FLAGS, 1, 1, false>(piece, bias, residual, out, ldc, M, N, cm0 + (region >> 1) * (16 * MT) + pi * 16,
                                                     cn0 + (region & 1) * (16 * NTW) + ph * 64, l15e, ge, wide_store != 0, &ln, nullptr);
        }
    }
}

#ifdef MQ_GEMM_PROBE   // compile-and-inspect builds (tests/test_gemm_isa.py): ONE instantiation, hipcc -DMQ_GEMM_PROBE=<flags> -DMQ_GEMM_PROBE_MT=<mt> -S
#ifndef MQ_GEMM_PROBE_NH
#define MQ_GEMM_PROBE_NH 1
#endif
#ifndef MQ_GEMM_PROBE_WM
#define MQ_GEMM_PROBE_WM 2
#endif
__attribute__((used)) void* mq_gemm_probe() { return (void*)gemm_nt_kernel<MQ_GEMM_PROBE, MQ_GEMM_PROBE_MT, MQ_GEMM_PROBE_NH, MQ_GEMM_PROBE_WM, 2>; }
}  // namespace
#else
constexpr int RESIDENT_SLOTS = 512;       // 256 CUs x 2 workgroups (NH = 1)
constexpr int RESIDENT_SLOTS_WIDE = 256;  // 256 CUs x 1 workgroup (NH = 2: 120 KiB of LDS)

// tuning knobs: initialised from the environment (MQ_GEMM_MT / _CGROUP / _NH), overridable through mq_tune()
struct GemmTune {
    // tail: the big tile's in-kernel tail (GemmSk).  OFF by default: its rows carry a differently associated k-sum, so an embedding's bits would
    // depend on whether its tokens sit in the last partial row tile of a batch — the towers promise the same bits wherever an item stands
    // (tests/test_towers_gpu.py permutation equivariance; the coalescer and the ingest merging lean on it) — for +1.6 % / +3.9 % on the ViT-L/14 rows
    // (profiles/r05p).  mq_tune("gemm_tail", 1) / MQ_GEMM_TAIL=1 turns it on; without it a ragged last row tile is a tile like any other.
    // wd: the W-direct main loop (gemm_wd.hip) on the narrow tiles: 0 = off, 2 / 3 = on with that many LDS stages of A
    // rs_fin: the residual GEMMs of the bf16 stream finalise the row statistics inside their own launch (mq_gemm_bf16_rsf; 0 = a row_stats_finalize_kernel
    // launch behind them, the round 4-5 form)
    mq_knob mt, cgroup, nh, tail, wd, rs_fin;
    static int env(const char* k, int d) { const char* v = getenv(k); return v ? atoi(v) : d; }
    GemmTune() : mt(env("MQ_GEMM_MT", 0)), cgroup(env("MQ_GEMM_CGROUP", 8)), nh(env("MQ_GEMM_NH", 0)), tail(env("MQ_GEMM_TAIL", 0)), wd(env("MQ_GEMM_WD", 0)), rs_fin(env("MQ_GEMM_RS_FIN", 0)) {}
};
GemmTune g_tune;
}  // namespace
// mirrors of the knobs for gemm_fp8.hip
mq_knob mq_gemm_knob_persist{1}, mq_gemm_knob_cgroup{(int)g_tune.cgroup}, mq_gemm_knob_wide{2};
// operands are addressed through 32-bit buffer offsets: bytes below 4 GiB per launch and operand; a taller A goes in row chunks.
// mq_tune("gemm_addr_limit_mb", v) lowers it so that the chunking can be tested at small sizes (0 = back to 4 GiB).
std::atomic<uint64_t> mq_gemm_addr_limit{0xffffffffull};
namespace {

// pick the tile height: minimise rounds x (MT + fixed per-tile overhead in 16-row units).  The tile HEIGHT is a free parameter because rows
// are guarded anyway; this removes most of the tile-quantisation loss at the towers' shapes (M = 12 800, N = 768: 600 128-row tiles = 2
// rounds on 512 slots, 480 160-row tiles = 1 round).
int choose_mt(int M, int N) {
    constexpr int BN = 128;
    const int tiles_n = (N + BN - 1) / BN;
    const int cands[4] = {2, 4, 5, 6};
    int best = 4;
    double best_cost = 1e30;
    for (int c = 0; c < 4; ++c) {
        const int mt = cands[c];
        const int bm = 32 * mt;
        const int64_t tiles = (int64_t)((M + bm - 1) / bm) * tiles_n;
        const int64_t rounds = (tiles + RESIDENT_SLOTS - 1) / RESIDENT_SLOTS;
        const double cost = (double)rounds * (mt + 1.25);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = mt; }
    }
    return best;
}

// ---- workspace of the big tile's in-kernel tail (GemmSk): partial-sum slots + flags, one block per HIP stream (launches on a stream are ordered, so
// its slots are free again when the next launch starts; request threads own their streams).  Allocated on a stream's first big-tile launch with a
// tail and kept for the process — the ONE place where this library allocates device memory itself (the C ABI's GEMM entry points take no workspace).
struct SkWorkspace { float* partials = nullptr; unsigned* flags = nullptr; unsigned epoch = 0; };
std::mutex g_sk_mu;
std::map<std::pair<int, hipStream_t>, SkWorkspace> g_sk_ws;
constexpr size_t SK_SLOT_BYTES = 256 * 256 * 4;

int sk_workspace(hipStream_t s, GemmSk& out) {
    int dev = 0;
    hipGetDevice(&dev);
    std::lock_guard<std::mutex> g(g_sk_mu);
    // The tail's workgroups WAIT for each other: all 256 of a launch must be resident.  One stream's launches are ordered, so that holds; two tail
    // launches on two streams could each hold part of the CUs and starve the other's unscheduled workgroups until the bounded spin gives up (the rows
    // would then be wrong: ADVICE r5).  So the tail is refused — the ragged last row of tiles runs as ordinary tiles — from the moment a SECOND stream
    // of this device asks for it.
    for (const auto& kv : g_sk_ws)
        if (kv.first.first == dev && kv.first.second != s && kv.second.partials) { out = GemmSk{}; return MQ_OK; }
    SkWorkspace& w = g_sk_ws[{dev, s}];
    if (!w.partials) {
        void* p = nullptr;
        const size_t bytes = (size_t)RESIDENT_SLOTS_WIDE * SK_SLOT_BYTES + 4096;
        if (hipError_t e = hipMalloc(&p, bytes); e != hipSuccess) {
            mq_set_error("mq_gemm_bf16: tail workspace (%zu bytes): %s", bytes, hipGetErrorString(e));
            return MQ_ERR_HIP;
        }
        w.partials = (float*)p;
        w.flags = (unsigned*)((char*)p + (size_t)RESIDENT_SLOTS_WIDE * SK_SLOT_BYTES);
        if (hipError_t e = hipMemsetAsync(w.flags, 0, 4096, s); e != hipSuccess) {
            mq_set_error("mq_gemm_bf16: tail workspace memset: %s", hipGetErrorString(e));
            return MQ_ERR_HIP;
        }
    }
    if (++w.epoch == 0) w.epoch = 1;     // (0 is the cleared state)
    out.partials = w.partials;
    out.flags = w.flags;
    out.epoch = w.epoch;
    return MQ_OK;
}

template <int FLAGS, int MT, int NH = 1, int ORD = 2, int WM = 2>
int launch_gemm_mt(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* residual, void* out, int64_t ldc,
                   int M, int N, int K, hipStream_t s, const GemmLn& ln) {
    constexpr int BM = 16 * MT * WM, BN = 128 * NH;
    constexpr int LDS = 2 * (BM + BN) * BK * 2;
    constexpr int SLOTS = (NH == 1 && WM == 2) ? RESIDENT_SLOTS : RESIDENT_SLOTS_WIDE;
    static std::atomic<uint64_t> attr_done{0};
    auto kern = gemm_nt_kernel<FLAGS, MT, NH, WM, ORD>;
    if (hipError_t e = mq_ensure_dyn_lds((const void*)kern, LDS, attr_done); e != hipSuccess) {
        mq_set_error("mq_gemm_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return MQ_ERR_HIP;
    }
    // 16-byte bf16 epilogue stores need 16-B aligned rows
    const int wide = (!(FLAGS & MQ_EPI_OUT_F32) && ldc % 8 == 0 && ((uintptr_t)out & 15) == 0) ? 1 : 0;
    const uint64_t lim = mq_gemm_addr_limit;   // rows x leading dimension x 2 B per launch
    const uint64_t w_bytes = ((uint64_t)(N - 1) * (uint64_t)ldw + (uint64_t)K) * 2;
    if (w_bytes > lim) {
        mq_set_error("mq_gemm_bf16: weight matrix of %llu bytes exceeds the %llu bytes a launch can address", (unsigned long long)w_bytes, (unsigned long long)lim);
        return MQ_ERR_INVALID;
    }
    // ... and a taller A goes in row chunks (rows are independent; whole tiles per chunk)
    int64_t max_rows = (uint64_t)K * 2 > lim ? 0 : (int64_t)((lim - (uint64_t)K * 2) / ((uint64_t)lda * 2)) + 1;
    max_rows = max_rows / BM * BM;
    if (max_rows < BM) {
        mq_set_error("mq_gemm_bf16: lda=%ld too large", (long)lda);
        return MQ_ERR_INVALID;
    }
    const int tiles_n = (N + BN - 1) / BN;
    for (int64_t r0 = 0; r0 < M; r0 += max_rows) {
        const int m = (int)((M - r0) < max_rows ? (M - r0) : max_rows);
        // big tile: a ragged last row of tiles is not a round of its own but the in-kernel tail (GemmSk), cut along K over the whole grid
        GemmSk sk{};
        int tiles_m = (m + BM - 1) / BM;
        if constexpr (WM == 4) {
            const int nk = K / BK;
            if (g_tune.tail && m % BM != 0 && m >= BM && tiles_n <= SLOTS && MT * NH == 8) {
                if (int rc = sk_workspace(s, sk); rc != MQ_OK) return rc;
                if (sk.partials) {                                 // (nullptr: refused — a second stream of this device uses the tail)
                    int S = 8;                                     // ranges per tail tile: a power of two <= 8 (the tile's 8 regions are shared out over them)
                    while (S > 1 && (S * tiles_n > SLOTS || S > nk)) S >>= 1;
                    sk.tail_splits = S;
                    tiles_m = m / BM;
                }
            }
        }
        const int num_tiles = tiles_m * tiles_n;
        // L2 blocking only when there is something to block: more column tiles than one group and at least two row panels per XCD
        const int knob_cgroup = g_tune.cgroup;
        const int cgroup = (knob_cgroup > 0 && tiles_n > knob_cgroup && tiles_m >= 16) ? knob_cgroup : 0;
        const int band_rows = (tiles_m + 7) / 8;
        const int grid = sk.tail_splits > 0 ? SLOTS : (num_tiles > SLOTS ? SLOTS : num_tiles);
        const uint64_t a_bytes = ((uint64_t)(m - 1) * (uint64_t)lda + (uint64_t)K) * 2;
        const size_t out_row = (size_t)ldc * ((FLAGS & MQ_EPI_OUT_F32) ? 4 : 2);
        const size_t res_row = (size_t)ldc * (((FLAGS & MQ_EPI_RESIDUAL) && !(FLAGS & MQ_EPI_OUT_F32)) ? 2 : 4);
        GemmLn ln_chunk = ln;
        if (ln_chunk.rowstats) ln_chunk.rowstats += r0;
        if (ln_chunk.partials) ln_chunk.partials += r0;   // (slot-major: a row offset is a row offset)
        if (ln_chunk.band_ctr) {   // in-launch finalise: this launch's row bands (chunks are whole tiles), every wave of a band's column tiles arrives once
            ln_chunk.band_ctr += (r0 / BM) * WM;
            ln_chunk.stats_out += r0;
            ln_chunk.band_target = tiles_n * 2;
            if (sk.tail_splits > 0) {
                mq_set_error("mq_gemm_bf16: the in-launch row-statistics finalise and the in-kernel tail exclude each other");
                return MQ_ERR_INVALID;
            }
        }
        if constexpr (NH == 1 && WM == 2) {
            if (const int wd = g_tune.wd; (wd == 2 || wd == 3 || wd == 6 || wd == 7) && !ln_chunk.band_ctr) {
                const int rc = mq_gemm_wd_launch(FLAGS, MT, wd, (const bf16_t*)A + r0 * lda, lda, W, ldw, bias,
                                                 residual ? (const float*)((const char*)residual + (size_t)r0 * res_row) : nullptr, (void*)((char*)out + (size_t)r0 * out_row),
                                                 ldc, m, N, K, tiles_n, num_tiles, cgroup, band_rows, grid, wide, (unsigned)a_bytes, (unsigned)w_bytes, ln_chunk, s);
                if (rc >= 0) { if (rc != MQ_OK) return rc; continue; }
            }
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(128 * WM), LDS, s, (const bf16_t*)A + r0 * lda, lda, (const bf16_t*)W, ldw, bias,
                           residual ? (const float*)((const char*)residual + (size_t)r0 * res_row) : nullptr, (void*)((char*)out + (size_t)r0 * out_row),
                           ldc, m, N, K, tiles_n, num_tiles, cgroup, band_rows, wide, (unsigned)a_bytes, (unsigned)w_bytes, ln_chunk, sk);
        MQ_CHECK_LAUNCH("mq_gemm_bf16");
    }
    return MQ_OK;
}

// ---- the BIG tile: 256 x 256 x 64 as 8 waves (gemm_nt_kernel WM = 4, MT = 4, NH = 2), one workgroup per CU, two waves per SIMD --------------------
// Its k-loop is 14-18 % faster than the narrow tiles' (8192^3: 1 354-1 440 vs 1 162-1 223 TF/s; a wave issues 8 LDS-DMA pieces per 64 MFMAs instead of
// 9 per 40 — profiles/r05c_gemm_big_tile_ab.txt), but 256 workgroups of 256 x 256 quantise badly: the towers' ViT-B/32 shapes (QKV 450 tiles = 1.76
// rounds, out-proj 150 = 0.59) lose on it, and stream-K does not rescue them — with ~1 tile per workgroup nearly every tile is split, and 256 fp32
// partials of 256 KB are as many bytes as the GEMM itself moves (built and dropped this round).  What is left is a ROW SPLIT: the
// big tile takes the leading rows as long as its tiles fill whole rounds of the 256 workgroups, the narrow kernel takes the few rows left (ViT-L/14
// at 128 images: 32 896 rows = 128 row tiles + 128 rows; 128 x {12, 16, 4} column tiles = exactly 6 / 8 / 2 rounds).  Same bits either way.
// Measured (profiles/r05e_gemm_row_split_ab.txt): stand-alone -13...-15 % at 4096^3 / 8192^3, -4...-9 % at the ViT-L/14 x 128 shapes, -14 % on its fc2 at
// 240 crops — but +16 % at K = 512, +5 % on some K = 1024 shapes, and INSIDE the towers nothing: ViT-L/14 dual 10 390 embeddings/s with and without.
// The default plan therefore only takes long-K problems with a predicted gain of 8 % or more (square-ish GEMMs of the C ABI's mq_gemm_bf16, the
// ViT-L/14 fc2 at large batches); the towers' other GEMMs stay on the narrow tile.
// mq_tune("gemm_nh", 3) forces the big tile on every row (N >= 256), 4 = the row-split plan without the long-K / 8 % restriction (the A/B above),
// 1 forbids it, 0 = the default plan.
// (The 4-wave 224 x 256 tile, NH = 2 / WM = 2 — one wave per SIMD — lost on every shape, profiles/r05a, r05b: removed in round 6, the template refuses it.)
constexpr double BIG_TILE_SPEEDUP = 1.12;   // k-loop advantage priced into the plan (measured 1.14-1.18 at full rounds)

// rows (a multiple of 256, 0 = none) the big tile should take of an M x N x K problem
int plan_big_rows(int M, int N, int K) {
    if (g_tune.nh == 1 || N < 256 || K < 512) return 0;
    if (g_tune.nh == 3) return M;                                   // forced: every row (a ragged last row tile is guarded)
    const bool eager = g_tune.nh == 4;
    if (!eager && K < 2048) return 0;
    const int tiles_n = (N + 255) / 256;
    const double fill_n = (double)N / (tiles_n * 256.0);           // columns of the last tile column that exist
    const int rt_max = M / 256;
    if (rt_max < 16 || fill_n < 0.9) return 0;
    // what the narrow kernel costs for m rows, in (32 rows x 128 columns x K) units per resident slot: rounds x (mt + per-tile overhead), as choose_mt prices it
    auto narrow_cost = [&](int m) {
        if (m <= 0) return 0.0;
        const int mt = choose_mt(m, N);
        const int64_t tiles = (int64_t)((m + 32 * mt - 1) / (32 * mt)) * ((N + 127) / 128);
        return (double)((tiles + RESIDENT_SLOTS - 1) / RESIDENT_SLOTS) * (mt + 1.25);
    };
    // ... and the big tile for rt row tiles: a 256 x 256 tile is 16 such units on ONE slot per CU where the narrow kernel has two -> 8 per round and
    // slot pair, divided by its speed-up; + the same per-tile overhead
    auto big_cost = [&](int rt) {
        const int64_t tiles = (int64_t)rt * tiles_n;
        return (double)((tiles + RESIDENT_SLOTS_WIDE - 1) / RESIDENT_SLOTS_WIDE) * (8.0 / BIG_TILE_SPEEDUP + 1.25);
    };
    const double all_narrow = narrow_cost(M);
    int best_rows = 0;
    double best = all_narrow * (eager ? 0.97 : 0.92);               // what the big tile has to win (see the measurements above)
    // (a) every row on the big tile: the full row tiles in rounds, the ragged rest as the in-kernel tail (a few k-steps per workgroup + the partial
    // sums' round trip: ~9 us whatever K is; a cost unit is ~3.1 ns x K)
    if (M % 256 == 0 || (g_tune.tail && tiles_n <= RESIDENT_SLOTS_WIDE)) {
        const double c = big_cost(rt_max) + (M % 256 ? 2900.0 / K : 0.0);
        if (c < best) { best = c; best_rows = M; }
    }
    // (b) the leading rows that fill whole rounds on the big tile, the rest in a second launch of the narrow kernel
    for (int rt = rt_max; rt >= rt_max - 32 && rt >= 16; --rt) {
        const double c = big_cost(rt) + narrow_cost(M - rt * 256);
        if (c < best) { best = c; best_rows = rt * 256; }
    }
    return best_rows;
}

// the narrow tile at height mt (32-row units).  LN_APPLY on top of the residual epilogue (mq_gemm_bf16_lnrs) has no registers for the tallest tile: 5 instead
template <int FLAGS>
int launch_narrow(int mt, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* residual, void* out, int64_t ldc,
                  int M, int N, int K, hipStream_t s, const GemmLn& ln) {
    constexpr bool LN_RES = (FLAGS & MQ_EPI_LN_APPLY) && (FLAGS & MQ_EPI_RESIDUAL);
    switch (mt) {
        case 2: return launch_gemm_mt<FLAGS, 2>(A, lda, W, ldw, bias, residual, out, ldc, M, N, K, s, ln);
        case 5: return launch_gemm_mt<FLAGS, 5>(A, lda, W, ldw, bias, residual, out, ldc, M, N, K, s, ln);
        case 6:
            if constexpr (LN_RES) return launch_gemm_mt<FLAGS, 5>(A, lda, W, ldw, bias, residual, out, ldc, M, N, K, s, ln);
            else return launch_gemm_mt<FLAGS, 6>(A, lda, W, ldw, bias, residual, out, ldc, M, N, K, s, ln);
        default: return launch_gemm_mt<FLAGS, 4>(A, lda, W, ldw, bias, residual, out, ldc, M, N, K, s, ln);
    }
}

template <int FLAGS>
int launch_gemm(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* residual, void* out, int64_t ldc,
                int M, int N, int K, hipStream_t s, const GemmLn& ln = GemmLn{}) {
    MQ_TRY(mq_device_ok());   // 256 CUs in 8 XCDs or nothing (runtime.hip)
    const int big_rows = plan_big_rows(M, N, K);
    if (big_rows >= M) return launch_gemm_mt<FLAGS, 4, 2, 2, 4>(A, lda, W, ldw, bias, residual, out, ldc, M, N, K, s, ln);
    if (big_rows > 0) {
        // leading rows on the big tile, the rest on the narrow one (rows are independent: row-offset views of every row-indexed operand)
        if (int rc = launch_gemm_mt<FLAGS, 4, 2, 2, 4>(A, lda, W, ldw, bias, residual, out, ldc, big_rows, N, K, s, ln); rc != MQ_OK) return rc;
        const size_t out_row = (size_t)ldc * ((FLAGS & MQ_EPI_OUT_F32) ? 4 : 2);
        const size_t res_row = (size_t)ldc * (((FLAGS & MQ_EPI_RESIDUAL) && !(FLAGS & MQ_EPI_OUT_F32)) ? 2 : 4);
        GemmLn ln2 = ln;
        if (ln2.rowstats) ln2.rowstats += big_rows;
        if (ln2.partials) ln2.partials += big_rows;
        if (ln2.band_ctr) { ln2.band_ctr += (big_rows / 256) * 4; ln2.stats_out += big_rows; ln2.pf_na = ln2.pf_nb = 0; }   // (the first launch carried the prefetch)
        A = (const bf16_t*)A + (int64_t)big_rows * lda;
        if (residual) residual = (const float*)((const char*)residual + (size_t)big_rows * res_row);
        out = (char*)out + (size_t)big_rows * out_row;
        M -= big_rows;
        const int knob_mt2 = g_tune.mt;
        const int mt2 = knob_mt2 ? knob_mt2 : choose_mt(M, N);
        return launch_narrow<FLAGS>(mt2, A, lda, W, ldw, bias, residual, out, ldc, M, N, K, s, ln2);
    }
    const int knob_mt = g_tune.mt;
    const int mt = knob_mt ? knob_mt : choose_mt(M, N);
    return launch_narrow<FLAGS>(mt, A, lda, W, ldw, bias, residual, out, ldc, M, N, K, s, ln);
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
    MQ_CHECK_ARG(!(flags & MQ_EPI_GLU) || N % 32 == 0, "mq_gemm_bf16: MQ_EPI_GLU needs N %% 32 == 0 (16 up + 16 gate rows per group)");
    hipStream_t s = (hipStream_t)stream;
    if (flags & MQ_EPI_GLU) {   // (the skinny kernels have no gated epilogue: the tiled family for any row count)
        MqProfScope prof(0, s, 2.0 * (double)M * (double)N * (double)K);
        return launch_gemm<MQ_EPI_BIAS | MQ_EPI_GLU>(d_A, lda, d_W, ldw, d_bias, d_residual ? (const float*)d_residual : nullptr, d_out, ldc, (int)M, (int)N, (int)K, s);
    }
    // a handful of rows (single queries, pooled rows of a small batch): the column-sliced skinny kernel spreads the weight stream over the
    // whole chip instead of N/128 workgroups (gemm_small.hip)
    if (mq_gemm_small_ok(M, N, K, false) || mq_gemm_small_grouped_ok(M, N, K)) return mq_gemm_small(d_A, lda, d_W, ldw, d_bias, d_residual, d_out, ldc, M, N, K, flags, s);
    MqProfScope prof(0, s, 2.0 * (double)M * (double)N * (double)K);
    const int m = (int)M, n = (int)N, k = (int)K;
#define MQ_GEMM_CASE(F) \
    case (F): return launch_gemm<(F)>(d_A, lda, d_W, ldw, d_bias, d_residual, d_out, ldc, m, n, k, s)
    switch (flags) {
        MQ_GEMM_CASE(0);
        MQ_GEMM_CASE(MQ_EPI_OUT_F32);
        MQ_GEMM_CASE(MQ_EPI_BIAS | MQ_EPI_OUT_F32);   // biased fp32 heads (M-CLIP LinearTransformation)
        MQ_GEMM_CASE(MQ_EPI_BIAS);
        MQ_GEMM_CASE(MQ_EPI_BIAS | MQ_EPI_GELU);
        MQ_GEMM_CASE(MQ_EPI_BIAS | MQ_EPI_QUICKGELU);
        MQ_GEMM_CASE(MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_OUT_F32);
        MQ_GEMM_CASE(MQ_EPI_BIAS | MQ_EPI_RESIDUAL);   // bf16 residual in (d_residual is bf16), bf16 out
        MQ_GEMM_CASE(MQ_EPI_BIAS | MQ_EPI_GLU);        // gated MLP: out [M, N / 2] = up * silu(gate), W rows interleaved 16 by 16
        default:
            mq_set_error("mq_gemm_bf16: unsupported epilogue flag combination 0x%x", flags);
            return MQ_ERR_INVALID;
    }
#undef MQ_GEMM_CASE
}

// GEMM over the UN-normalised bf16 rows with the LayerNorm folded in (gemm_epilogue.h): out = act( LN(A) @ W0^T + b0 ) where d_W = bf16(gamma * W0)
// (the LayerNorm's scale folded into the weight's columns), d_bias = b0 + W0 @ beta, d_colsum[n] = sum_k d_W[n, k] (of the ROUNDED folded weight),
// d_rowstats = (mean, rstd) per row of A (mq_row_stats) and K = the normalised width.  flags: MQ_EPI_BIAS [| MQ_EPI_GELU | MQ_EPI_QUICKGELU]; bf16 out.
extern "C" int mq_gemm_bf16_ln(const void* d_A, int64_t lda, const void* d_W, int64_t ldw, const float* d_bias, const float* d_colsum,
                               const float* d_rowstats, void* d_out, int64_t ldc, int64_t M, int64_t N, int64_t K, int flags, void* stream) {
    MQ_CHECK_ARG(d_A && d_W && d_out && d_bias && d_colsum && d_rowstats, "mq_gemm_bf16_ln: null operand");
    MQ_CHECK_ARG(M >= 1 && N >= 4 && K >= BK && K % BK == 0 && N % 4 == 0, "mq_gemm_bf16_ln: bad shape M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
    MQ_CHECK_ARG(!(flags & MQ_EPI_GLU) || N % 32 == 0, "mq_gemm_bf16_ln: MQ_EPI_GLU needs N %% 32 == 0");
    MQ_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0, "mq_gemm_bf16_ln: leading dims must keep 16-byte rows");
    MQ_CHECK_ARG(M < (1 << 30) && N < (1 << 30) && K < (1 << 30), "mq_gemm_bf16_ln: shape too large");
    hipStream_t s = (hipStream_t)stream;
    MqProfScope prof(0, s, 2.0 * (double)M * (double)N * (double)K);
    const int m = (int)M, n = (int)N, k = (int)K;
    GemmLn ln{};
    ln.colsum = d_colsum;
    ln.rowstats = (const float2*)d_rowstats;
#define MQ_GEMM_LN_CASE(F) \
    case (F): return launch_gemm<(F)>(d_A, lda, d_W, ldw, d_bias, nullptr, d_out, ldc, m, n, k, s, ln)
    switch (flags | MQ_EPI_LN_APPLY) {
        MQ_GEMM_LN_CASE(MQ_EPI_BIAS | MQ_EPI_LN_APPLY);
        MQ_GEMM_LN_CASE(MQ_EPI_BIAS | MQ_EPI_GELU | MQ_EPI_LN_APPLY);
        MQ_GEMM_LN_CASE(MQ_EPI_BIAS | MQ_EPI_QUICKGELU | MQ_EPI_LN_APPLY);
        MQ_GEMM_LN_CASE(MQ_EPI_BIAS | MQ_EPI_GLU | MQ_EPI_LN_APPLY);
        default:
            mq_set_error("mq_gemm_bf16_ln: unsupported epilogue flag combination 0x%x", flags);
            return MQ_ERR_INVALID;
    }
#undef MQ_GEMM_LN_CASE
}

// mq_gemm_bf16 (bias + bf16 residual read-modify-write) that also leaves the rows' partial statistics behind (gemm_epilogue.h, MQ_EPI_ROW_STATS):
// d_partials fp32 [ceil(N/64)][M][2] (slot-major)
extern "C" int mq_gemm_bf16_rs(const void* d_A, int64_t lda, const void* d_W, int64_t ldw, const float* d_bias, const void* d_residual, void* d_out,
                               int64_t ldc, int64_t M, int64_t N, int64_t K, int flags, float* d_partials, void* stream) {
    MQ_CHECK_ARG(d_A && d_W && d_out && d_bias && d_residual && d_partials, "mq_gemm_bf16_rs: null operand");
    MQ_CHECK_ARG((flags | MQ_EPI_ROW_STATS) == (MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_ROW_STATS), "mq_gemm_bf16_rs: flags must be MQ_EPI_BIAS | MQ_EPI_RESIDUAL");
    MQ_CHECK_ARG(M >= 1 && N >= 4 && K >= BK && K % BK == 0 && N % 4 == 0, "mq_gemm_bf16_rs: bad shape M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
    MQ_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0, "mq_gemm_bf16_rs: leading dims must keep 16-byte rows");
    MQ_CHECK_ARG(M < (1 << 30) && N < (1 << 30) && K < (1 << 30), "mq_gemm_bf16_rs: shape too large");
    hipStream_t s = (hipStream_t)stream;
    MqProfScope prof(0, s, 2.0 * (double)M * (double)N * (double)K);
    GemmLn ln{};
    ln.partials = (float2*)d_partials;
    ln.nslots = (int)((N + 63) / 64);
    ln.part_ld = M;
    return launch_gemm<MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_ROW_STATS>(d_A, lda, d_W, ldw, d_bias, (const float*)d_residual, d_out, ldc, (int)M, (int)N, (int)K, s, ln);
}

// LN_APPLY and ROW_STATS together (ABI 12; the EVA02 sub-LayerNorms folded into the GEMMs around them): out = [residual +] LN(A) @ W0^T + b0 with
// d_W / d_bias / d_colsum folded as for mq_gemm_bf16_ln and d_rowstats = (mean, rstd) of A's rows — which come from the launch that WROTE A (the
// attention kernel's per-head sums, mq_attention_stats; the gated epilogue's, below) through mq_row_stats_finalize — and, in d_partials, the (sum, sum of
// squares) per row and 64-column slot of what THIS launch stores, for the LayerNorm behind it.  flags:
//   MQ_EPI_BIAS | MQ_EPI_RESIDUAL  the out-projection / fc2 form: bf16 residual read-modify-write, d_partials [ceil(N / 64)][M][2] (slot-major)
//   MQ_EPI_BIAS | MQ_EPI_GLU       the (up | gate) form: out [M, N / 2] = up * silu(gate); a slot = a wave's 64 GEMM columns = 32 hidden units:
//                                  d_partials [ceil(N / 64)][M][2] of the rounded products
extern "C" int mq_gemm_bf16_lnrs(const void* d_A, int64_t lda, const void* d_W, int64_t ldw, const float* d_bias, const float* d_colsum, const float* d_rowstats,
                                 const void* d_residual, void* d_out, int64_t ldc, int64_t M, int64_t N, int64_t K, int flags, float* d_partials, void* stream) {
    MQ_CHECK_ARG(d_A && d_W && d_out && d_bias && d_colsum && d_rowstats && d_partials, "mq_gemm_bf16_lnrs: null operand");
    MQ_CHECK_ARG(flags == (MQ_EPI_BIAS | MQ_EPI_RESIDUAL) || flags == (MQ_EPI_BIAS | MQ_EPI_GLU), "mq_gemm_bf16_lnrs: flags must be MQ_EPI_BIAS | MQ_EPI_RESIDUAL or MQ_EPI_BIAS | MQ_EPI_GLU");
    MQ_CHECK_ARG(!(flags & MQ_EPI_RESIDUAL) || d_residual, "mq_gemm_bf16_lnrs: MQ_EPI_RESIDUAL without residual");
    MQ_CHECK_ARG(M >= 1 && N >= 4 && K >= BK && K % BK == 0 && N % 4 == 0, "mq_gemm_bf16_lnrs: bad shape M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
    MQ_CHECK_ARG(!(flags & MQ_EPI_GLU) || N % 32 == 0, "mq_gemm_bf16_lnrs: MQ_EPI_GLU needs N %% 32 == 0");
    MQ_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0, "mq_gemm_bf16_lnrs: leading dims must keep 16-byte rows");
    MQ_CHECK_ARG(M < (1 << 30) && N < (1 << 30) && K < (1 << 30), "mq_gemm_bf16_lnrs: shape too large");
    hipStream_t s = (hipStream_t)stream;
    MqProfScope prof(0, s, 2.0 * (double)M * (double)N * (double)K);
    GemmLn ln{};
    ln.colsum = d_colsum;
    ln.rowstats = (const float2*)d_rowstats;
    ln.partials = (float2*)d_partials;
    ln.nslots = (int)((N + 63) / 64);
    ln.part_ld = M;
    if (flags & MQ_EPI_GLU)
        return launch_gemm<MQ_EPI_BIAS | MQ_EPI_GLU | MQ_EPI_LN_APPLY | MQ_EPI_ROW_STATS>(d_A, lda, d_W, ldw, d_bias, nullptr, d_out, ldc, (int)M, (int)N, (int)K, s, ln);
    return launch_gemm<MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_LN_APPLY | MQ_EPI_ROW_STATS>(d_A, lda, d_W, ldw, d_bias, (const float*)d_residual, d_out, ldc, (int)M, (int)N, (int)K, s, ln);
}

// mq_gemm_bf16_rs + the finalise of the row statistics: on return (stream order) d_stats holds (mean, rstd) of every row of d_out.  The finalise runs
// INSIDE the GEMM's launch — the last wave to arrive at a row band's counter sums the band's partials (GemmLn::band_ctr, gemm_epilogue.h) — and the
// launch carries the weight prefetch of the GEMMs behind it; d_band_ctr: mq_gemm_band_counters(M) zeroed 32-bit counters (left zeroed).  With
// mq_tune("rs_finalize", 0), the in-kernel tail (MQ_GEMM_TAIL) or no counters: the round 4-5 form, a row_stats_finalize_kernel launch behind the GEMM.
// Same bits either way (mq_finalize_stats, slot order).
int mq_row_stats_finalize_pf(const float* d_partials, int32_t nslots, float* d_stats, int64_t rows, int32_t W, float eps, const void* pf_a, size_t bytes_a,
                             const void* pf_b, size_t bytes_b, hipStream_t s);
extern "C" int64_t mq_gemm_band_counters(int64_t M) { return M / 32 + 16; }
bool mq_gemm_rs_in_launch() { return g_tune.rs_fin && !g_tune.tail; }   // towers.hip: whether a pass needs (zeroed) band counters at all
extern "C" int mq_gemm_bf16_rsf(const void* d_A, int64_t lda, const void* d_W, int64_t ldw, const float* d_bias, const void* d_residual, void* d_out,
                                int64_t ldc, int64_t M, int64_t N, int64_t K, int flags, float* d_partials, float* d_stats, float eps,
                                uint32_t* d_band_ctr, const void* d_pf_a, size_t pf_a_bytes, const void* d_pf_b, size_t pf_b_bytes, void* stream) {
    MQ_CHECK_ARG(d_A && d_W && d_out && d_bias && d_residual && d_partials && d_stats, "mq_gemm_bf16_rsf: null operand");
    MQ_CHECK_ARG((flags | MQ_EPI_ROW_STATS) == (MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_ROW_STATS), "mq_gemm_bf16_rsf: flags must be MQ_EPI_BIAS | MQ_EPI_RESIDUAL");
    MQ_CHECK_ARG(M >= 1 && N >= 4 && K >= BK && K % BK == 0 && N % 4 == 0, "mq_gemm_bf16_rsf: bad shape M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
    MQ_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0, "mq_gemm_bf16_rsf: leading dims must keep 16-byte rows");
    MQ_CHECK_ARG(M < (1 << 30) && N < (1 << 30) && K < (1 << 30), "mq_gemm_bf16_rsf: shape too large");
    hipStream_t s = (hipStream_t)stream;
    const int nslots = (int)((N + 63) / 64);
    const bool in_launch = d_band_ctr && g_tune.rs_fin && !g_tune.tail;
    {
        MqProfScope prof(0, s, 2.0 * (double)M * (double)N * (double)K);
        GemmLn ln{};
        ln.partials = (float2*)d_partials;
        ln.nslots = nslots;
        ln.part_ld = M;
        if (in_launch) {
            ln.band_ctr = d_band_ctr;
            ln.stats_out = (float2*)d_stats;
            ln.inv_w = 1.0f / (float)N;
            ln.eps = eps;
            ln.pf_a = (const unsigned*)d_pf_a;
            ln.pf_b = (const unsigned*)d_pf_b;
            ln.pf_na = d_pf_a ? (unsigned)(pf_a_bytes / 128) : 0u;
            ln.pf_nb = d_pf_b ? (unsigned)(pf_b_bytes / 128) : 0u;
        }
        MQ_TRY(launch_gemm<MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_ROW_STATS>(d_A, lda, d_W, ldw, d_bias, (const float*)d_residual, d_out, ldc, (int)M, (int)N, (int)K, s, ln));
    }
    if (in_launch) return MQ_OK;
    return mq_row_stats_finalize_pf(d_partials, nslots, d_stats, M, (int32_t)N, eps, d_pf_a, pf_a_bytes, d_pf_b, pf_b_bytes, s);
}

// Run-time knobs (A/B benchmarking and the parity tests of every code path in one process).  TEST / BENCH ONLY (the loaders never touch them):
// atomics (common.h, mq_knob) that the launch code of every request thread reads — a change takes effect from the next launch that reads it,
// so set them while no request is in flight if one call must run under one setting.
// keys: "gemm_mt" (0 = auto, else tile height in 32-row units), "gemm_cgroup", "row_select", "ln_fold", "residual_bf16", "small_m", "small_m_grouped",
// "ln_prefetch", "xcd_band", "attn_waves", "gemm_addr_limit_mb", "gemm_nh" (0 = default plan, 1 = (32*MT) x 128 tiles only, 3 = the big 256 x 256 tile on every row wherever N >= 256, 4 = the eager row-split plan),
// "gemm_tail", "gemm_wd" (gemm_wd.hip: 0 = off, 2 / 3 / 6 / 7), "rs_finalize" (1 = the row statistics are finalised inside the residual GEMM's launch),
// "subln_fold" (0 = the EVA02 sub-LayerNorms run as LayerNorm passes instead of inside the out-projection / fc2 GEMMs),
// "attn_proj" (fewest fixed-length sequences from which a ViT-B/32-shaped block runs attention + out-projection + residual + statistics as ONE launch,
// attn_proj.hip; 0 = never), "panel_gemm" (fewest fixed-length sequences from which the folded QKV / fc1 GEMMs of a 768-wide tower run one workgroup per
// sequence, panel_gemm.hip; 0 = never).
extern "C" int mq_tune(const char* key, int value) {
    MQ_CHECK_ARG(key, "mq_tune: null key");
    const std::string k(key);
    if (k == "gemm_mt") { g_tune.mt = value; mq_gemm_fp8_force_mt = value; }
    else if (k == "gemm_cgroup") { g_tune.cgroup = value; mq_gemm_knob_cgroup = value; }
    else if (k == "gemm_nh") { g_tune.nh = value; mq_gemm_fp8_big = value == 4 ? 0 : value; }
    else if (k == "gemm_tail") g_tune.tail = value;
    else if (k == "gemm_wd") g_tune.wd = value;
    else if (k == "rs_finalize") g_tune.rs_fin = value;
    else if (k == "row_select") mq_tower_row_select = value;
    else if (k == "ln_fold") mq_tower_ln_fold = value;
    else if (k == "subln_fold") mq_tower_subln_fold = value;
    else if (k == "attn_proj") mq_tower_attn_proj = value;
    else if (k == "panel_gemm") mq_tower_panel_gemm = value;
    else if (k == "xcd_band") mq_xcd_band = value;
    else if (k == "attn_waves") mq_attention_waves = value;
    else if (k == "residual_bf16") mq_tower_residual_bf16 = value;
    else if (k == "small_m") mq_gemm_small_max_rows = value;
    else if (k == "small_m_grouped") mq_gemm_small_group_rows = value;
    else if (k == "ln_prefetch") mq_ln_prefetch = value;
    else if (k == "gemm_addr_limit_mb") mq_gemm_addr_limit = value > 0 ? ((uint64_t)value << 20) - 1 : 0xffffffffull;
    else { mq_set_error("mq_tune: unknown key %s", key); return MQ_ERR_INVALID; }
    return MQ_OK;
}
#endif  // MQ_GEMM_PROBE
