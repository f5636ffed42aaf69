// bf16 MFMA GEMM with fused epilogues for gfx950 (K3 / K5 / K1-GEMM / K6 / K8 of SURVEY.md §8a).
//
//   out[M,N] = epi( A[M,K] @ W[N,K]^T )         A, W bf16, K-contiguous ("NT": W is the
//                                                PyTorch nn.Linear weight as stored)
//
// Design (MI355X-first, not a CUDA tiling):
//   * (32*MT)x128x64 block tile (MT = 2/4/5/6 -> 64..192 rows), 4 wave64s as 2x2, each wave a
//     (16*MT)x64 sub-tile = MT x 4 v_mfma_f32_16x16x32_bf16 accumulators.
//   * global -> LDS by direct LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction),
//     2 stages x (BM*128 B of A + 16 KiB of W) = 48..80 KiB LDS -> 2 workgroups per CU.
//   * LDS tiles are [rows][128 B]; the 16-B chunk index is XOR-swizzled with (row & 7).
//     LDS-DMA writes lane-linear, so the swizzle is applied to each lane's GLOBAL source
//     address and again on the ds_read_b128 side (same involution) -> conflict-free reads.
//   * operands are fed swapped (mfma(Wfrag, Afrag)) so each lane ends up owning 4
//     CONSECUTIVE n of one output row: bias/residual/out are 8/16-byte vector accesses.
//   * workgroup -> tile map is XCD-aware: the 8 XCDs (block b runs on XCD b % 8) each take a
//     contiguous range of tiles, so tiles sharing an A row-panel hit the same private L2.
#include <stdlib.h>
#include <string>
#include <type_traits>
#include "common.h"
#include "gemm_epilogue.h"

extern int mq_gemm_fp8_force_mt;  // gemm_fp8.hip
extern int mq_tower_row_select;   // towers.hip
extern int mq_tower_ln_fold;      // towers.hip
extern int mq_ln_rows_per_wave;   // rowops.hip
extern int mq_ln_bf16_wide;       // rowops.hip
extern int mq_attention_waves;    // attention.hip
extern int mq_tower_residual_bf16;  // towers.hip
extern int mq_gemm_small_max_rows;  // gemm_small.hip
bool mq_gemm_small_ok(int64_t M, int64_t N, int64_t K, bool ln);
bool mq_gemm_small_grouped_ok(int64_t M, int64_t N, int64_t K);
extern int mq_gemm_small_group_rows;
extern int mq_ln_prefetch;        // rowops.hip
int mq_gemm_small(const void* d_A, int64_t lda, const void* d_W, int64_t ldw, const float* d_bias, const void* d_residual, void* d_out,
                  int64_t ldc, int64_t M, int64_t N, int64_t K, int flags, hipStream_t s);

// CU-sized-tile main loop (gemm_big.hip)
template <int FLAGS>
int mq_launch_gemm_big(int mt, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* residual,
                       void* out, int64_t ldc, int M, int N, int K, hipStream_t s);

#ifdef MQ_GEMM_TRACE
// Diagnostic build only (tools/probes/gemm_trace.py): per-wave cycle sums of the main loop's phases, written once at kernel
// end by the first MQ_TRACE_BLOCKS workgroups: [block][wave][0..5] = k-steps, cycles parked at s_waitcnt vmcnt(0), cycles at the
// barrier, cycles in the k-step body (ds_reads + MFMAs + next-stage LDS-DMA issues), cycles in the epilogue, tiles.
#define MQ_TRACE_BLOCKS 64
__device__ unsigned long long mq_gemm_trace_buf[MQ_TRACE_BLOCKS * 4 * 6];
extern "C" int mq_gemm_trace_read(unsigned long long* h_out) {
    return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(mq_gemm_trace_buf), sizeof(mq_gemm_trace_buf)) == hipSuccess ? 0 : -2;
}
#define MQ_TR_NOW() __builtin_amdgcn_s_memtime()
// per-WORKGROUP spans on the device-wide constant 100 MHz counter (s_memrealtime): [block][0..3] = start, end, HW_ID | XCC_ID << 32, tiles —
// launch ramp, residency (workgroups per CU) and tail of a launch
#define MQ_SPAN_BLOCKS 2048
__device__ unsigned long long mq_gemm_span_buf[MQ_SPAN_BLOCKS * 4];
extern "C" int mq_gemm_span_read(unsigned long long* h_out) {
    return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(mq_gemm_span_buf), sizeof(mq_gemm_span_buf)) == hipSuccess ? 0 : -2;
}
#endif

// one workgroup per CU, two accumulator sets: the epilogue of tile i under the k-loop of tile i+1 (gemm_pp.hip)
int mq_gemm_pp_plan(int M, int N, int K, int flags, bool inplace);
void mq_gemm_pp_tune(const char* key, int value);
template <int FLAGS>
int mq_launch_gemm_pp(int mt, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const void* residual, void* out,
                      int64_t ldc, int M, int N, int K, int cgroup_knob, hipStream_t s);

// short-k-step / 3-4 workgroups per CU form (gemm_k32.hip)
template <int FLAGS>
int mq_launch_gemm_k32(int wgs, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* residual,
                       void* out, int64_t ldc, int M, int N, int K, int cgroup_knob, int wide_knob, hipStream_t s);

// software-pipelined k-loop (gemm_pl.hip)
bool mq_gemm_pl_fits(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldw);
int mq_gemm_pl_mode();
void mq_gemm_pl_tune(const char* key, int value);
template <int FLAGS>
int mq_launch_gemm_pl(int mt, const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias, const float* residual, void* out,
                      int64_t ldc, int M, int N, int K, int cgroup_knob, int wide_knob, hipStream_t s, const GemmLn& ln);

namespace {

constexpr int BN = 128, BK = 64;
constexpr int W_TILE_BYTES = BN * BK * 2;  // 16 KiB

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    // 16 B per lane, LDS destination = wave-uniform base + lane * 16.
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// MT = 16-row MFMA sub-tiles per wave along M  ->  block tile BM = 32*MT rows (64 .. 192).
// The tile HEIGHT is a free parameter because rows are guarded anyway; the launcher picks the MT
// that minimises (rounds of resident workgroups) x (tile cost), which removes most of the tile
// quantisation loss at the towers' shapes (e.g. M=12800,N=768: 600 128-row tiles = 2 rounds on
// 512 slots, 480 160-row tiles = 1 round).
// PERSIST: the grid is the number of resident slots and every workgroup walks tiles bid, bid + grid, ...; the LAST
// k-step of a tile prefetches stage 0 of the workgroup's next tile, so only the first tile pays the cold first-stage
// fetch (at K = 768 a tile is 12 k-steps: the cold fetch is ~10 % of it).
template <int FLAGS, int MT, bool PERSIST>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(
    const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ Wt, int64_t ldw,
    const float* __restrict__ bias, const float* residual, void* out, int64_t ldc,
    int M, int N, int K, int tiles_n, int num_tiles, int cgroup, int band_rows, int wide_store, GemmLn ln) {
    constexpr int BM = 32 * MT;
    constexpr int A_TILE_BYTES = BM * BK * 2;
    constexpr int STAGE_BYTES = A_TILE_BYTES + W_TILE_BYTES;
    // the tile's 128 bias values ride through LDS (two 512-B slots behind the stages, alternating per tile): fetched at the top of the
    // tile, parked in LDS after the first landed stage, read by the epilogue.  Not at MT = 6, whose stages fill the CU's LDS budget.
    constexpr bool LDS_BIAS = (FLAGS & MQ_EPI_BIAS) && MT <= 5;
    // MQ_EPI_LN_APPLY (gemm_epilogue.h): the rows' (sum x, sum x^2) are accumulated from the A tiles as they pass through LDS — thread t owns
    // 16-byte chunk t % 8 of tile rows t / 8 + 32 i (i < MT: BM rows x 8 chunks = 256 * MT chunks) — and (mean, rstd) per tile row is left
    // in LDS behind the bias slots for the epilogue
    constexpr bool LN_APPLY = (FLAGS & MQ_EPI_LN_APPLY) != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const bias_lds = (float*)(smem + 2 * STAGE_BYTES);
    float2* const rowstats_lds = (float2*)(smem + 2 * STAGE_BYTES + (LDS_BIAS ? 2 * BN * 4 : 0));
    int bias_slot = 0;

    // ---- XCD-aware, bijective (virtual) block -> tile map ----------------------------------
    const int q = num_tiles >> 3, r = num_tiles & 7;
    const int tiles_m = (M + BM - 1) / BM;
    auto tile_origin = [&](int vbid, int& m0, int& n0) {
        const int xcd = vbid & 7, idx = vbid >> 3;
        const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        int tm, tn;
        if (cgroup > 0) {
            // L2-blocked order inside an XCD's share: the linear order is (row band, column group, row panel, column) with a
            // band = band_rows row panels (about one XCD's share) and a group = cgroup column tiles, so the ~64 tiles resident
            // on an XCD at any time are ~64/cgroup row panels x cgroup column tiles: their A panels + W column tiles fit the
            // XCD's 4 MiB L2 and the W group stays put while the A panels stream past.  (The plain row-major order swept ALL
            // column tiles per panel: W alone — 4.7 MB at N = 3072, K = 768 — overflowed L2 and the fabric-side read traffic
            // measured 7x the operand bytes, profiles/r01_traffic_*.)
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

    // ---- staging: wave w owns A rows [8*MT*w, 8*MT*(w+1)) and W rows [32w, 32w+32), 8 rows per
    // LDS-DMA.  lane -> (row = base + lane/8, physical chunk = lane%8); it fetches logical chunk
    // (lane%8) ^ (row&7) of that row, so physical chunk p of row r holds logical chunk p^(r&7).
    const int srow = lane >> 3;
    const bf16_t* a_src[MT];
    const bf16_t* w_src[4];
    auto set_sources = [&](int m0, int n0) {
#ifdef MQ_GEMM_ALIAS
        // diagnostic build (tools/probes/build_gemm_alias.sh): operands are fetched from an aliased origin while results still go to
        // the tile's own place.  1: every tile reads the SAME A / W tile (L1- and L2-resident); 2: origins folded into 4 row
        // panels x 4 column tiles (L2-resident on every XCD, far larger than a CU's L1).  Wrong results by construction.
        if (MQ_GEMM_ALIAS == 1) { m0 = 0; n0 = 0; }
        else { m0 %= 4 * BM; n0 %= 4 * BN; }
#endif
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = wave * (8 * MT) + i * 8 + srow;
            int gm = m0 + row; gm = gm < M ? gm : M - 1;
            a_src[i] = A + (int64_t)gm * lda + ((lane & 7) ^ (row & 7)) * 8;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wave * 32 + i * 8 + srow;
            int gn = n0 + row; gn = gn < N ? gn : N - 1;
            w_src[i] = Wt + (int64_t)gn * ldw + ((lane & 7) ^ (row & 7)) * 8;
        }
    };
#ifdef MQ_GEMM_DIAG
    // diagnostic builds (tools/probes/build_gemm_diag.sh; timing only, results wrong by construction): which resource bounds the k-loop?
    //   1 = no global -> LDS traffic after the first stage (MFMA + ds_read + epilogue ceiling)
    //   2 = no MFMAs / fragment reads (global -> LDS fill + barriers + epilogue ceiling)
    //   3 = no epilogue (k-loop only)
#endif
    auto stage = [&](int buf, int kt) {
        char* sa = smem + buf * STAGE_BYTES + wave * (8 * MT * 128);
        char* sw = smem + buf * STAGE_BYTES + A_TILE_BYTES + wave * (32 * 128);
#pragma unroll
        for (int i = 0; i < MT; ++i) glds16(a_src[i] + (int64_t)kt * BK, sa + i * (8 * 128));
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(w_src[i] + (int64_t)kt * BK, sw + i * (8 * 128));
    };

    f32x4 acc[MT][4];

    const int nk = K / BK;
    int vbid = blockIdx.x;
    int m0, n0;
    tile_origin(vbid, m0, n0);
    set_sources(m0, n0);
    stage(0, 0);
    // experiment knob (mq_tune("gemm_stagger", n)): the second resident workgroup of every CU starts n x 1024 cycles late, so that the two
    // workgroups' epilogues (matrix pipe idle) do not coincide on multi-tile persistent launches
    if (PERSIST && (wide_store >> 8) && blockIdx.x >= (gridDim.x >> 1))
        for (int i = 0; i < (wide_store >> 8); ++i) __builtin_amdgcn_s_sleep(16);
    const bool lds_bias_on = LDS_BIAS && (wide_store & 2);   // knob mq_tune("gemm_lds_bias", 0 / 1)
    // experiment knobs (profiles/r02u_gemm_vmcnt_prio_ab.txt):
    //  * mq_tune("gemm_vmcnt", 1): after an epilogue the first k-step of the next tile waits only for the stage-0 LDS-DMA that was issued
    //    BEFORE the epilogue's stores (vmcnt retires in issue order on gfx9: the tile's EPI_STORES stores may stay in flight) instead of
    //    draining them with vmcnt(0); only for waves whose tile was fully inside the matrix (every store instruction issued)
    //  * mq_tune("gemm_prio", 1): static s_setprio 1 for the second-dispatched workgroup of every CU (MI355X_MICROARCH.md item 4)
    const bool counted_vmcnt = PERSIST && (wide_store & 4) && !(FLAGS & MQ_EPI_LN_APPLY);
    if ((wide_store & 8) && blockIdx.x >= (gridDim.x >> 1)) __builtin_amdgcn_s_setprio(1);
    wide_store &= 1;
    bool stores_pending = false;   // the previous tile's epilogue stores are still in flight and may be skipped by the next wait
    int buf = 0;  // LDS buffer of the next k-step (runs on across tiles in the persistent form)
#ifdef MQ_GEMM_TRACE
    unsigned long long tr_steps = 0, tr_vm = 0, tr_bar = 0, tr_body = 0, tr_epi = 0, tr_tiles = 0;
    const unsigned long long span_t0 = wall_clock64();
#endif
    for (;;) {
        float bias_reg = 0.f;
        if (lds_bias_on && tid < BN && n0 + tid < N) bias_reg = bias[n0 + tid];
        bool bias_parked = !lds_bias_on;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        float st1[LN_APPLY ? MT : 1], st2[LN_APPLY ? MT : 1];
#pragma unroll
        for (int i = 0; i < (LN_APPLY ? MT : 1); ++i) st1[i] = st2[i] = 0.f;
        // ---- fragment read offsets (bytes inside a tile), fixed per lane ----------------------
        // logical chunk for k-half kk is g + 4*kk; (row & 7) == (l15 & 7) because sub-tile bases are multiples of 16.
        // Persistent form: recomputed per tile from a laundered lane id, so that these 11 registers are NOT live across the
        // epilogue (kept live they pushed the 160-row persistent kernels to the 256-VGPR cap and into scratch spills).
        int l15f = l15, gf = g;
        if (PERSIST) asm volatile("" : "+v"(l15f), "+v"(gf));
        int a_off[MT], w_off[4];
#pragma unroll
        for (int t = 0; t < MT; ++t) a_off[t] = (wm * (16 * MT) + t * 16 + l15f) * 128;
#pragma unroll
        for (int t = 0; t < 4; ++t) w_off[t] = (wn * 64 + t * 16 + l15f) * 128;
        const int sw0 = ((gf) ^ (l15f & 7)) << 4;      // kk = 0
        const int sw1 = ((gf + 4) ^ (l15f & 7)) << 4;  // kk = 1
        // one k-step on LDS buffer `buf`: fragments for both 32-deep halves are read up front, then the MT*8 MFMAs run with
        // the NEXT stage's LDS-DMA issues (k offset `koff` of the current a_src / w_src, into the other buffer) sprinkled
        // between them (an LDS-DMA issue costs the wave ~60-180 cycles; bunched at the top of the step they serialised in
        // front of the MFMAs and held the matrix pipe at 25-40 %).  PREFETCH is a template flag so the steady-state loop has
        // no branch in it.
        auto kstep = [&](int buf, int64_t koff, auto prefetch_tag) {
            constexpr bool PREFETCH = decltype(prefetch_tag)::value;
            const char* sa = smem + buf * STAGE_BYTES;
            const char* sw = sa + A_TILE_BYTES;
            char* na = smem + (buf ^ 1) * STAGE_BYTES + wave * (8 * MT * 128);
            char* nw = smem + (buf ^ 1) * STAGE_BYTES + A_TILE_BYTES + wave * (32 * 128);
            bf16x8 af[2][MT], wf[2][4];
    #pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int swz = kk ? sw1 : sw0;
    #pragma unroll
                for (int t = 0; t < 4; ++t) wf[kk][t] = *(const bf16x8*)(sw + w_off[t] + swz);
    #pragma unroll
                for (int t = 0; t < MT; ++t) af[kk][t] = *(const bf16x8*)(sa + a_off[t] + swz);
            }
            if constexpr (LN_APPLY) {
                // (these LDS reads sit with the fragment reads, in front of the step's first LDS-DMA issue: behind one, hipcc would wait for it)
                const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3f803f80u);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const uint4 c = *(const uint4*)(sa + (tid + 256 * i) * 16);
                    const bf16x2_t c0 = __builtin_bit_cast(bf16x2_t, c.x), c1 = __builtin_bit_cast(bf16x2_t, c.y),
                                   c2 = __builtin_bit_cast(bf16x2_t, c.z), c3 = __builtin_bit_cast(bf16x2_t, c.w);
                    // v_dot2c_f32_bf16: two elements per VALU operation, fp32 accumulation
                    st1[i] = __builtin_amdgcn_fdot2_f32_bf16(c0, ones, st1[i], false);
                    st2[i] = __builtin_amdgcn_fdot2_f32_bf16(c0, c0, st2[i], false);
                    st1[i] = __builtin_amdgcn_fdot2_f32_bf16(c1, ones, st1[i], false);
                    st2[i] = __builtin_amdgcn_fdot2_f32_bf16(c1, c1, st2[i], false);
                    st1[i] = __builtin_amdgcn_fdot2_f32_bf16(c2, ones, st1[i], false);
                    st2[i] = __builtin_amdgcn_fdot2_f32_bf16(c2, c2, st2[i], false);
                    st1[i] = __builtin_amdgcn_fdot2_f32_bf16(c3, ones, st1[i], false);
                    st2[i] = __builtin_amdgcn_fdot2_f32_bf16(c3, c3, st2[i], false);
                }
            }
            constexpr int NL = MT + 4;            // LDS-DMA pieces per wave per step
            constexpr int NM = 8 * MT;            // MFMAs per wave per step
            constexpr int GAP = NM / NL;          // MFMAs between two pieces
            int issued = 0;
    #pragma unroll
            for (int kk = 0; kk < 2; ++kk)
    #pragma unroll
                for (int mt = 0; mt < MT; ++mt)
    #pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
#if !defined(MQ_GEMM_DIAG) || MQ_GEMM_DIAG != 2
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][nt], af[kk][mt], acc[mt][nt], 0, 0, 0);
#endif
                        const int done = (kk * MT + mt) * 4 + nt + 1;
#if defined(MQ_GEMM_DIAG) && MQ_GEMM_DIAG == 1
                        if (false) {
#else
                        if (PREFETCH && done % GAP == 0 && issued < NL) {
#endif
                            if (issued < MT) glds16(a_src[issued] + koff, na + issued * (8 * 128));
                            else glds16(w_src[issued - MT] + koff, nw + (issued - MT) * (8 * 128));
                            ++issued;
                        }
                    }
            if (PREFETCH) {
                // pin the interleave: GAP MFMAs, one VMEM, ... (sched_group_barrier masks: 0x8 MFMA, 0x10 VMEM)
    #pragma unroll
                for (int i = 0; i < NL; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, GAP, 0);
                    __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
                }
            }
        };

        for (int kt = 0; kt < nk - 1; ++kt) {
            // stage kt has landed for every wave, and every wave is done reading the other buffer
#ifndef MQ_GEMM_TRACE
            if (stores_pending) {   // (first k-step after an epilogue: the stage's DMA was issued before the stores)
                stores_pending = false;
                constexpr int EPI_STORES = (FLAGS & MQ_EPI_OUT_F32) ? 4 * MT : 2 * MT;   // fp32: one 16-B store per sub-tile; bf16 (wide): one per pair
                if (EPI_STORES == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (EPI_STORES == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                else if (EPI_STORES == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if (EPI_STORES == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if (EPI_STORES == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // (the tile's bias values — a load issued AFTER the stores — are parked one k-step later, behind that step's full wait:
                // consuming them here would make the compiler drain the stores after all)
                __syncthreads();
                kstep(buf, (int64_t)(kt + 1) * BK, std::true_type{});
                buf ^= 1;
                continue;
            }
#endif
#ifdef MQ_GEMM_TRACE
            const unsigned long long t0 = MQ_TR_NOW();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t1 = MQ_TR_NOW();
            if (!bias_parked) {
                if (tid < BN) bias_lds[bias_slot * BN + tid] = bias_reg;
                bias_parked = true;
            }
            __syncthreads();
            const unsigned long long t2 = MQ_TR_NOW();
            kstep(buf, (int64_t)(kt + 1) * BK, std::true_type{});
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long t3 = MQ_TR_NOW();
            tr_steps += 1; tr_vm += t1 - t0; tr_bar += t2 - t1; tr_body += t3 - t2;
#else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!bias_parked) {  // (the wait above covered the bias load too; the barrier below publishes the slot)
                if (tid < BN) bias_lds[bias_slot * BN + tid] = bias_reg;
                bias_parked = true;
            }
            __syncthreads();
            kstep(buf, (int64_t)(kt + 1) * BK, std::true_type{});
#endif
            buf ^= 1;
        }
        const int cm0 = m0, cn0 = n0;
        bool more = false;
        if (PERSIST) {
            vbid += gridDim.x;
            more = vbid < num_tiles;
            if (more) {  // from here on a_src / w_src address the NEXT tile (this tile's last stage is already in flight / landed)
                tile_origin(vbid, m0, n0);
                set_sources(m0, n0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!bias_parked && tid < BN) bias_lds[bias_slot * BN + tid] = bias_reg;  // K == 64: no prefetching k-step ran
        __syncthreads();
        if (PERSIST && more) kstep(buf, 0, std::true_type{});
        else kstep(buf, 0, std::false_type{});
        buf ^= 1;
#ifdef MQ_GEMM_TRACE
        const unsigned long long te0 = MQ_TR_NOW();
#endif
#if defined(MQ_GEMM_DIAG) && MQ_GEMM_DIAG == 3
        {   // 3 = no epilogue (the accumulators are folded into one never-true store so that the MFMAs stay live)
            f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) sum += acc[i][j];
            if (sum[0] + sum[1] + sum[2] + sum[3] == 1.2345678e33f) ((float*)out)[0] = sum[0];
        }
#else
        if constexpr (LN_APPLY) {
            // the 8 lanes that share a row (lane % 8 = chunk) add up their shares in a fixed order (DPP: xor 1, xor 2, half-row mirror)
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
            __syncthreads();   // (rewritten one whole k-loop later: no second barrier needed behind the epilogue's reads)
        }
        gemm_epilogue<FLAGS, MT>(acc, bias, residual, out, ldc, M, N, cm0 + wm * (16 * MT), cn0 + wn * 64, l15, g, wide_store != 0, &ln,
                                 lds_bias_on ? bias_lds + bias_slot * BN + wn * 64 : nullptr, LN_APPLY ? rowstats_lds + wm * (16 * MT) : nullptr);
#endif
        bias_slot ^= 1;
#ifdef MQ_GEMM_TRACE
        __builtin_amdgcn_sched_barrier(0);
        tr_epi += MQ_TR_NOW() - te0;  // issue time of the epilogue (its stores retire later)
        tr_tiles += 1;
#endif
        if (!PERSIST || !more) break;
        // every store instruction of the epilogue was issued by this wave iff its sub-tile lay fully inside the matrix (wave-uniform);
        // the bf16 count holds for the widened-store form only
        stores_pending = counted_vmcnt && nk > 1 && (wide_store != 0 || (FLAGS & MQ_EPI_OUT_F32)) &&
                         cm0 + wm * (16 * MT) + 16 * MT <= M && cn0 + wn * 64 + 64 <= N;
    }
#ifdef MQ_GEMM_TRACE
    if (blockIdx.x < MQ_TRACE_BLOCKS && lane == 0) {
        unsigned long long* o = mq_gemm_trace_buf + (blockIdx.x * 4 + wave) * 6;
        o[0] = tr_steps; o[1] = tr_vm; o[2] = tr_bar; o[3] = tr_body; o[4] = tr_epi; o[5] = tr_tiles;
    }
    if (blockIdx.x < MQ_SPAN_BLOCKS && tid == 0) {
        unsigned long long* o = mq_gemm_span_buf + blockIdx.x * 4;
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        o[0] = span_t0; o[1] = wall_clock64(); o[2] = (unsigned long long)hw | ((unsigned long long)xcc << 32); o[3] = tr_tiles;
    }
#endif
}

// tuning knobs: initialised from the environment (MQ_GEMM_MT / _PERSIST / _CGROUP / _WIDE / _BIG), overridable through mq_tune()
struct GemmTune {
    int mt, persist, big, cgroup, wide, k32, stagger = 0, lds_bias = 1, vmcnt = 0, prio = 0;
    static int env(const char* k, int d) { const char* v = getenv(k); return v ? atoi(v) : d; }
    GemmTune() : mt(env("MQ_GEMM_MT", 0)), persist(env("MQ_GEMM_PERSIST", 1)), big(env("MQ_GEMM_BIG", 0)), cgroup(env("MQ_GEMM_CGROUP", 8)), wide(env("MQ_GEMM_WIDE", 2)), k32(env("MQ_GEMM_K32", 0)) {}
};
GemmTune g_tune;
}  // namespace
// mirrors of the knobs for gemm_fp8.hip
int mq_gemm_knob_persist = g_tune.persist, mq_gemm_knob_cgroup = g_tune.cgroup, mq_gemm_knob_wide = g_tune.wide;
namespace {

constexpr int RESIDENT_SLOTS = 512;  // 256 CUs x 2 workgroups (64..80 KiB LDS each)

// pick the tile height: minimise rounds x (MT + fixed per-tile overhead in 16-row units)
int choose_mt(int M, int N) {
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

template <int FLAGS, int MT, bool PERSIST>
int launch_gemm_mt(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                   const float* residual, void* out, int64_t ldc, int M, int N, int K, hipStream_t s, const GemmLn& ln) {
    constexpr int BM = 32 * MT;
    constexpr int LDS = 2 * (BM * BK * 2 + W_TILE_BYTES) + (((FLAGS & MQ_EPI_BIAS) && MT <= 5) ? 2 * BN * 4 : 0) + ((FLAGS & MQ_EPI_LN_APPLY) ? BM * 8 : 0);
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = mq_ensure_dyn_lds((const void*)gemm_nt_kernel<FLAGS, MT, PERSIST>, LDS, attr_done); e != hipSuccess) {
        mq_set_error("mq_gemm_bf16: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return MQ_ERR_HIP;
    }
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int num_tiles = tiles_m * tiles_n;
    // L2 blocking only when there is something to block: more column tiles than one group and at least two row panels per XCD
    const int cgroup = (g_tune.cgroup > 0 && tiles_n > g_tune.cgroup && tiles_m >= 16) ? g_tune.cgroup : 0;
    const int band_rows = (tiles_m + 7) / 8;
    // 16-byte bf16 epilogue stores need 16-B aligned rows
    // (gemm_wide = 2, the default: everywhere; 1: not beside the GELU epilogues; 0: off.  profiles/r01b_gemm_knobs_ab.txt)
    const bool act = (FLAGS & (MQ_EPI_GELU | MQ_EPI_QUICKGELU)) != 0;
    const int wide = (g_tune.wide && (g_tune.wide >= 2 || !act) && !(FLAGS & MQ_EPI_OUT_F32) && ldc % 8 == 0 && ((uintptr_t)out & 15) == 0) ? 1 : 0;
    const int grid = PERSIST && num_tiles > RESIDENT_SLOTS ? RESIDENT_SLOTS : num_tiles;
    const int stagger = (PERSIST && num_tiles >= 2 * RESIDENT_SLOTS) ? g_tune.stagger : 0;
    hipLaunchKernelGGL((gemm_nt_kernel<FLAGS, MT, PERSIST>), dim3(grid), dim3(256), LDS, s,
                       (const bf16_t*)A, lda, (const bf16_t*)W, ldw, bias, residual, out, ldc,
                       M, N, K, tiles_n, num_tiles, cgroup, band_rows,
                       wide | (g_tune.lds_bias ? 2 : 0) | (g_tune.vmcnt ? 4 : 0) | (g_tune.prio ? 8 : 0) | (stagger << 8), ln);
    MQ_CHECK_LAUNCH("mq_gemm_bf16");
    return MQ_OK;
}

template <int FLAGS>
int launch_gemm(const void* A, int64_t lda, const void* W, int64_t ldw, const float* bias,
                const float* residual, void* out, int64_t ldc, int M, int N, int K, hipStream_t s, const GemmLn& ln = GemmLn{}) {
    const int force_mt = g_tune.mt;
    const int persist = g_tune.persist;
    int mt = force_mt ? force_mt : choose_mt(M, N);
    if ((FLAGS & MQ_EPI_LN_APPLY) && mt == 6) mt = 5;   // the 192-row tile's stages fill the LDS of two workgroups per CU: no room for the row statistics
    if constexpr ((FLAGS & MQ_EPI_LN_APPLY) != 0) {
        // the folded-LayerNorm GEMMs run on the software-pipelined loop (persistent at every tile height; the round 1-3 loop below spills there)
        if (mq_gemm_pl_fits(M, N, K, lda, ldw))
            return mq_launch_gemm_pl<FLAGS>(mt, A, lda, W, ldw, bias, residual, out, ldc, M, N, K, g_tune.cgroup, g_tune.wide, s, ln);
    }
    if constexpr ((FLAGS & MQ_EPI_LN_APPLY) == 0) {
        if (!g_tune.big && !g_tune.k32 && mq_gemm_pl_mode() && mq_gemm_pl_fits(M, N, K, lda, ldw))
            return mq_launch_gemm_pl<FLAGS>(mt, A, lda, W, ldw, bias, residual, out, ldc, M, N, K, g_tune.cgroup, g_tune.wide, s, ln);
        if (g_tune.big) return mq_launch_gemm_big<FLAGS>(g_tune.big, A, lda, W, ldw, bias, residual, out, ldc, M, N, K, s);
        if (g_tune.k32) return mq_launch_gemm_k32<FLAGS>(g_tune.k32, A, lda, W, ldw, bias, residual, out, ldc, M, N, K, g_tune.cgroup, g_tune.wide, s);
        if (!force_mt && ldc % 8 == 0 && ((uintptr_t)out & 15) == 0 && lda < (1 << 22) && ldw < (1 << 22) && ldc < (1 << 22)) {
            if (const int pmt = mq_gemm_pp_plan(M, N, K, FLAGS, (const void*)residual == (const void*)out))
                return mq_launch_gemm_pp<FLAGS>(pmt, A, lda, W, ldw, bias, residual, out, ldc, M, N, K, g_tune.cgroup, s);
        }
    }
    // persistent form unless the epilogue does not fit its register budget (LN_STATS: a one-VGPR scratch spill)
    auto run = [&](auto mt_tag) {
        constexpr int T = decltype(mt_tag)::value;
        if constexpr (T == 6 || (T == 5 && (FLAGS & MQ_EPI_LN_APPLY))) {  // spills in the persistent form: the 192-row tile; the 160-row tile + row statistics
            return launch_gemm_mt<FLAGS, T, false>(A, lda, W, ldw, bias, residual, out, ldc, M, N, K, s, ln);
        } else {
            return persist ? launch_gemm_mt<FLAGS, T, true>(A, lda, W, ldw, bias, residual, out, ldc, M, N, K, s, ln)
                           : launch_gemm_mt<FLAGS, T, false>(A, lda, W, ldw, bias, residual, out, ldc, M, N, K, s, ln);
        }
    };
    switch (mt) {
        case 2: return run(std::integral_constant<int, 2>{});
        case 5: return run(std::integral_constant<int, 5>{});
        case 6: return run(std::integral_constant<int, 6>{});
        default: return run(std::integral_constant<int, 4>{});
    }
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
        default:
            mq_set_error("mq_gemm_bf16: unsupported epilogue flag combination 0x%x", flags);
            return MQ_ERR_INVALID;
    }
#undef MQ_GEMM_CASE
}

// GEMM over the UN-normalised bf16 rows with the LayerNorm folded in (gemm_epilogue.h): out = act( LN(A) @ W0^T + b0 ) where d_W = bf16(gamma * W0)
// (the LayerNorm's scale folded into the weight's columns), d_bias = b0 + W0 @ beta, d_colsum[n] = sum_k d_W[n, k] (of the ROUNDED folded weight)
// and K = the normalised width (a tile spans whole rows).  flags: MQ_EPI_BIAS [| MQ_EPI_GELU | MQ_EPI_QUICKGELU] (MQ_EPI_LN_APPLY implied); bf16 out.
extern "C" int mq_gemm_bf16_ln(const void* d_A, int64_t lda, const void* d_W, int64_t ldw, const float* d_bias, const float* d_colsum, void* d_out,
                               int64_t ldc, int64_t M, int64_t N, int64_t K, int flags, float eps, void* stream) {
    MQ_CHECK_ARG(d_A && d_W && d_out && d_bias && d_colsum, "mq_gemm_bf16_ln: null operand");
    MQ_CHECK_ARG(M >= 1 && N >= 4 && K >= BK && K % BK == 0 && N % 4 == 0, "mq_gemm_bf16_ln: bad shape M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
    MQ_CHECK_ARG(lda % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0, "mq_gemm_bf16_ln: leading dims must keep 16-byte rows");
    MQ_CHECK_ARG(M < (1 << 30) && N < (1 << 30) && K < (1 << 30), "mq_gemm_bf16_ln: shape too large");
    hipStream_t s = (hipStream_t)stream;
    MqProfScope prof(0, s, 2.0 * (double)M * (double)N * (double)K);
    const int m = (int)M, n = (int)N, k = (int)K;
    GemmLn ln{};
    ln.eps = eps;
    ln.colsum = d_colsum;
    ln.inv_w = 1.0f / (float)k;
#define MQ_GEMM_LN_CASE(F) \
    case (F): return launch_gemm<(F)>(d_A, lda, d_W, ldw, d_bias, nullptr, d_out, ldc, m, n, k, s, ln)
    switch (flags | MQ_EPI_LN_APPLY) {
        MQ_GEMM_LN_CASE(MQ_EPI_BIAS | MQ_EPI_LN_APPLY);
        MQ_GEMM_LN_CASE(MQ_EPI_BIAS | MQ_EPI_GELU | MQ_EPI_LN_APPLY);
        MQ_GEMM_LN_CASE(MQ_EPI_BIAS | MQ_EPI_QUICKGELU | MQ_EPI_LN_APPLY);
        default:
            mq_set_error("mq_gemm_bf16_ln: unsupported epilogue flag combination 0x%x", flags);
            return MQ_ERR_INVALID;
    }
#undef MQ_GEMM_LN_CASE
}

// Select a GEMM main-loop variant at run time (A/B benchmarking and parity tests of every variant in one process).
// key: "gemm_mt" (0 = auto, else tile height in 32-row units), "gemm_persist", "gemm_cgroup", "gemm_big" (0 / 4 / 6 / 8), "row_select".
extern "C" int mq_tune(const char* key, int value) {
    MQ_CHECK_ARG(key, "mq_tune: null key");
    const std::string k(key);
    if (k == "gemm_mt") { g_tune.mt = value; mq_gemm_fp8_force_mt = value; }
    else if (k == "gemm_persist") mq_gemm_knob_persist = g_tune.persist = value;
    else if (k == "gemm_big") g_tune.big = value;
    else if (k == "gemm_cgroup") mq_gemm_knob_cgroup = g_tune.cgroup = value;
    else if (k == "gemm_wide") mq_gemm_knob_wide = g_tune.wide = value;
    else if (k == "gemm_stagger") g_tune.stagger = value;
    else if (k == "gemm_lds_bias") g_tune.lds_bias = value;
    else if (k == "gemm_vmcnt") g_tune.vmcnt = value;
    else if (k == "gemm_prio") g_tune.prio = value;
    else if (k == "gemm_k32") g_tune.k32 = value;
    else if (k == "gemm_pl" || k == "gemm_pl_ord") mq_gemm_pl_tune(key, value);
    else if (k == "gemm_pp" || k == "gemm_pp_pps" || k == "gemm_pp_skew" || k == "gemm_pp_waves") mq_gemm_pp_tune(key, value);
    else if (k == "row_select") mq_tower_row_select = value;
    else if (k == "ln_fold") mq_tower_ln_fold = value;
    else if (k == "ln_rows") mq_ln_rows_per_wave = value;
    else if (k == "ln_bf16_wide") mq_ln_bf16_wide = value;
    else if (k == "xcd_band") mq_xcd_band = value;
    else if (k == "attn_waves") mq_attention_waves = value;
    else if (k == "residual_bf16") mq_tower_residual_bf16 = value;
    else if (k == "small_m") mq_gemm_small_max_rows = value;
    else if (k == "small_m_grouped") mq_gemm_small_group_rows = value;
    else if (k == "ln_prefetch") mq_ln_prefetch = value;
    else { mq_set_error("mq_tune: unknown key %s", key); return MQ_ERR_INVALID; }
    return MQ_OK;
}
