// fp8 (OCP e4m3) MFMA GEMM for gfx950 (K13 of SURVEY.md §8a; BASELINE config 5):
//
//   out[M,N] = epi( (A8[M,K] @ W8[N,K]^T) * a_scale[m] * w_scale[n] )
//
// A8 / W8 hold e4m3 codes; a_scale is per ROW (written by the producing LayerNorm) or one per-tensor scalar (static,
// calibrated: attention / GELU outputs), w_scale per OUTPUT CHANNEL (computed once at load).  The contraction runs on
// v_mfma_scale_f32_16x16x128_f8f6f4 with unit (e8m0 = 127) block scales: that is the only fp8 MFMA on gfx950 that runs
// at twice the bf16 rate (the non-scaled 16x16x32_fp8 runs at the bf16 rate), and fp8 operands halve the bytes moved
// through the LDS-DMA path and the LDS, which is what bounds the bf16 kernel.
//
// Structure = gemm_bf16.hip (one design, round 4): (32*MT)x128 tile, 4 wave64s as 2x2, BK = 128 elements (= the same 128-B LDS rows), a 2-stage
// LDS ring filled by buffer-descriptor LDS-DMA two stages ahead of the MFMAs (a cursor that walks over tile boundaries, so a tile's first stages
// land during the previous tile's epilogue), fragment reads as inline-asm ds_read_b128 with hand-placed waits, ONE barrier per k-step placed
// mid-step, persistent XCD-aware tile walk, operands fed swapped so a lane owns 4 consecutive n.  A lane's fragment is 32 consecutive k-bytes
// (two ds_read_b128): 16-B chunk c of row r is stored at c ^ f(r) with f below — chosen so that each 16-lane ds_read_b128 group (8 rows
// reading chunk 2g+j, 8 rows reading chunk 2g+2+j) touches 16 distinct 16-B slots of the 256-B bank row.
//
// The k-step (a 16x16x128 MFMA consumes the whole 128-B row: there is no k-half to pipeline over as in the bf16 kernel; the halves are the W side):
//   top     s_waitcnt lgkmcnt(0): A fragments + W sub-tiles 0, 1 of this stage (read during the previous step) have landed
//   half A  2 MT MFMAs (mt, nt = 0 | 1); between them the reads of W sub-tiles 2, 3 of this stage
//   mid     vmcnt(0) (my pieces of the NEXT stage have landed), lgkmcnt(0), s_barrier: the next stage may be read, this buffer may be refilled
//   half B  2 MT MFMAs (mt, nt = 2 | 3), row group by row group; behind row group mt's last MFMA the read of the NEXT stage's A fragment mt
//           into the same registers (an MFMA has read its operands long before a ds_read issued behind it returns), the next stage's W
//           sub-tiles 0, 1, and the LDS-DMA pieces of the stage after next into the buffer this step has finished with
#include <stdlib.h>
#include <type_traits>
#include "common.h"

typedef int i32x8 __attribute__((ext_vector_type(8)));

// scheduling knobs shared with gemm_bf16.hip (mq_tune "gemm_cgroup"; the widened epilogue stores are always on)
extern mq_knob mq_gemm_knob_cgroup, mq_gemm_knob_wide;
int mq_device_ok();   // runtime.hip
extern std::atomic<uint64_t> mq_gemm_addr_limit;   // gemm_bf16.hip: bytes one launch may address per operand (4 GiB - 1; tests lower it)

mq_knob mq_gemm_fp8_big{getenv("MQ_GEMM_FP8_NH") ? atoi(getenv("MQ_GEMM_FP8_NH")) : 0};   // mq_tune("gemm_nh", v) sets it too (gemm_bf16.hip)

namespace {

constexpr int BK = 128;                    // BK in fp8 elements == bytes
constexpr int UNIT_SCALE = 0x7F7F7F7F;     // e8m0 127 = 2^0 in every byte
constexpr int RESIDENT_SLOTS = 512;        // 256 CUs x 2 workgroups
constexpr int RESIDENT_SLOTS_BIG = 256;    // the 8-wave 256 x 256 tile: one workgroup per CU

typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, unsigned lds_wave_base) {
    // 16 B per lane; LDS destination = wave-uniform base (M0) + lane * 16; source = descriptor base + voff (per lane) + soff (scalar)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(uintptr_t)lds_wave_base, 16, voff, soff, 0, 0);
}
// a 32-byte fragment = two ds_read_b128 (logical chunks 2g, 2g + 1 of the lane's row) as inline asm: hipcc cannot tell a compiler-visible LDS
// read from the LDS-DMA writes in flight and would put s_waitcnt vmcnt(0) in front of every one (gemm_bf16.hip)
template <int OFF>
__device__ __forceinline__ i32x8 lds_read32(unsigned addr0, unsigned addr1) {
    i32x4_t lo, hi;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(lo) : "v"(addr0), "n"(OFF));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(hi) : "v"(addr1), "n"(OFF));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// ties the consumers of an asm-read fragment to the hand-placed s_waitcnt by data flow (gemm_bf16.hip, `landed`)
template <class T>
__device__ __forceinline__ void landed(T& v) { asm volatile("" : "+v"(v)); }

template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// chunk swizzle: u = (row >> 1) & 7;  f = u for u in {0,1,6,7}, u ^ 2 for u in {2,3,4,5}
__device__ __forceinline__ int fswz(int row) {
    const int u = (row >> 1) & 7;
    return u ^ ((((u >> 1) ^ (u >> 2)) & 1) << 1);
}

__device__ __forceinline__ float clamp448(float v) { return fminf(fmaxf(v, -448.f), 448.f); }

// FLAGS: MQ_EPI_BIAS / GELU / QUICKGELU / RESIDUAL / OUT_F32 / OUT_FP8; ROWSCALE: a_scale is per row (else scalar).
// cgroup / wide: L2-blocked tile order and widened epilogue stores as in gemm_bf16.hip (at K = 768 a tile is only SIX 128-deep k-steps here, so
// the per-tile overheads weigh even more than in the bf16 kernel).  Widened stores: bf16 out -> 16 B per lane after one v_permlane16_swap per
// pair; e4m3 out -> a lane's 4 codes per sub-tile become 16 CONSECUTIVE codes after a permlane16 stage (pairs of sub-tiles) and a permlane32
// stage (the two pairs), i.e. one 16-byte store per 16-row sub-tile instead of four 4-byte ones.
// NH / WM (round 6): the tile shapes of gemm_bf16.hip.  NH = 1, WM = 2: the (32 MT) x 128 tiles, 4 waves as 2 x 2, two workgroups per CU.  NH = 2, WM = 4, MT = 4:
// the BIG tile, 256 x 256 x 128 as 8 waves (4 x 2, 64 x 128 per wave: 8 W sub-tiles), one workgroup per CU, two waves per SIMD — half the LDS-DMA
// and vector-memory bytes per MFMA (64 KB per 2 048 MFMA cycles where two 160 x 128 workgroups move 72 KB per 1 280).  Same k-step: the halves are
// the W side (NTW / 2 sub-tiles each), same staging, swizzle, waits and epilogue (once per 64-column half of the wave's columns).
template <int FLAGS, int MT, bool ROWSCALE, int NH = 1, int WM = 2>
__global__ __launch_bounds__(128 * WM, (NH == 1 && WM == 2) ? 2 : 1) void gemm_fp8_kernel(
    const uint8_t* __restrict__ A, int64_t lda, const uint8_t* __restrict__ Wt, int64_t ldw,
    const float* __restrict__ a_scale, const float* __restrict__ w_scale,
    const float* __restrict__ bias, const float* residual, void* out, int64_t ldc,
    const float* __restrict__ out_scale, float* amax_out,
    int M, int N, int K, int tiles_n, int num_tiles, int cgroup, int band_rows, int wide, unsigned a_bytes, unsigned w_bytes) {
    constexpr int BM = 16 * MT * WM, BN = 128 * NH;
    constexpr int NTW = 4 * NH;                 // 16-column W sub-tiles per wave (a wave spans half of BN)
    constexpr int HW = NTW / 2;                 // ... per half of the k-step
    constexpr int A_TILE_BYTES = BM * BK, W_TILE_BYTES = BN * BK;
    constexpr int STAGE_BYTES = A_TILE_BYTES + W_TILE_BYTES;
    constexpr int NPW = 8 * NH / WM;            // W pieces (8 rows x 128 B) per wave per stage: BN / (2 WM) rows
    extern __shared__ __attribute__((aligned(16))) char smem[];

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

    constexpr int NL = MT + NPW;   // LDS-DMA pieces (8 rows x 128 B) per wave per stage
    constexpr int NB = HW * MT;    // MFMAs per wave per half
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;

    // ---- staging: wave w owns A rows [8*MT*w, 8*MT*(w+1)) and W rows [32w, 32w+32) of a stage; lane -> (row = base + lane/8, physical 16-B
    // chunk = lane%8), it fetches logical chunk (lane%8) ^ f(row): the swizzle lives on the SOURCE address, the LDS image is lane-linear
    const int srow = lane >> 3;
    unsigned a_vo[MT], w_vo[NPW];
    auto set_sources = [&](int m0, int n0) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = wave * (8 * MT) + i * 8 + srow;
            int gm = m0 + row; gm = gm < M ? gm : M - 1;
            a_vo[i] = (unsigned)gm * (unsigned)lda + (unsigned)(((lane & 7) ^ fswz(row)) * 16);
        }
#pragma unroll
        for (int i = 0; i < NPW; ++i) {
            const int row = wave * (8 * NPW) + i * 8 + srow;
            int gn = n0 + row; gn = gn < N ? gn : N - 1;
            w_vo[i] = (unsigned)gn * (unsigned)ldw + (unsigned)(((lane & 7) ^ fswz(row)) * 16);
        }
    };
    const int nk = K / BK;
    // DMA cursor: (tile d_vbid, k-step d_k) of the next stage to request, two stages ahead of the MFMAs; past the workgroup's last tile the
    // descriptors' sizes drop to 0 (the two trailing requests are out of range for every lane: no traffic, no second k-step variant)
    int d_vbid = blockIdx.x, d_k = 0;
    unsigned a_rec = a_bytes, w_rec = w_bytes;
    {
        int m0, n0;
        tile_origin(d_vbid, m0, n0);
        set_sources(m0, n0);
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned dma_a0 = lds0 + (unsigned)wave * (8 * MT * 128), dma_w0 = lds0 + A_TILE_BYTES + (unsigned)wave * (8 * NPW * 128);   // scalars
    auto issue_piece = [&](int i, unsigned bufoff) {
        const unsigned soff = (unsigned)d_k * BK;
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

    // ---- fragment read addresses (LDS byte offsets), fixed per lane: logical chunks 2g and 2g + 1 of row base16 + l15 (sub-tile bases are
    // multiples of 16 rows: f(row) = f(l15))
    const int fr = fswz(l15);
    const unsigned c0 = (unsigned)(((2 * g) ^ fr) << 4), c1 = (unsigned)(((2 * g + 1) ^ fr) << 4);
    const unsigned a_row = lds0 + (unsigned)((wm * (16 * MT) + l15) * 128);
    const unsigned w_row = lds0 + A_TILE_BYTES + (unsigned)((wn * (16 * NTW) + l15) * 128);
    const unsigned aB0 = a_row + c0, aB1 = a_row + c1, wB0 = w_row + c0, wB1 = w_row + c1;

    f32x4 acc[NH][MT][4];
    i32x8 af[MT], wf[NTW];
    const float a_scalar = ROWSCALE ? 1.f : a_scale[0];
    const float inv_out = (FLAGS & MQ_EPI_OUT_FP8) ? 1.0f / out_scale[0] : 1.f;
    float amax = 0.f;

    // ---- prologue: the workgroup's first two stages, then the first stage's A fragments and W sub-tiles 0, 1
#pragma unroll
    for (int i = 0; i < NL; ++i) issue_piece(i, 0);
    advance_cursor();
#pragma unroll
    for (int i = 0; i < NL; ++i) issue_piece(i, STAGE_BYTES);
    advance_cursor();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory");   // stage 0 landed (loads retire in issue order)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    unsigned bufoff = 0;   // LDS byte offset of the stage the next k-step consumes
    int c_vbid = blockIdx.x;

    // LAST: a tile's last k-step does not read the next stage's fragments — that stage belongs to the NEXT tile, whose fragments would be live
    // through the whole epilogue (56 registers at MT = 5 next to 80 accumulators and the epilogue's own: spills).  The tile loop reads them
    // after the epilogue instead: one exposed LDS latency per tile.
    auto kstep = [&](auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        // -------- half A: MFMAs (mt, nt < HW); reads of this stage's W sub-tiles HW .. NTW - 1 between them
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // af[*], wf[0 .. HW) landed
#pragma unroll
        for (int t = 0; t < HW; ++t) landed(wf[t]);
#pragma unroll
        for (int t = 0; t < MT; ++t) landed(af[t]);
        __builtin_amdgcn_sched_barrier(0);
        {
            const unsigned b0 = wB0 + bufoff, b1 = wB1 + bufoff;
            static_for<NB>([&](auto idx_tag) {
                constexpr int idx = decltype(idx_tag)::value, mt = idx / HW, nt = idx % HW;
                acc[nt / 4][mt][nt % 4] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[nt], af[mt], acc[nt / 4][mt][nt % 4], 0, 0, 0, UNIT_SCALE, 0, UNIT_SCALE);
                // read r of the second half's HW sub-tiles goes behind MFMA r * RPOS (NH = 1: behind MFMAs 0 and 2, as before)
                constexpr int RPOS = NB > HW ? 2 : 1;
                if constexpr (idx % RPOS == 0 && idx / RPOS < HW) {
                    constexpr int r = HW + idx / RPOS;
                    __builtin_amdgcn_sched_barrier(0);
                    wf[r] = lds_read32<r * 2048>(b0, b1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        }
        // -------- mid-step: the stage after this one has landed (my pieces), my reads of this buffer are done; after the barrier both hold for
        // every wave: the next stage may be read, this buffer may be refilled
        // (the two halves write DIFFERENT accumulators: nothing but these empty asms keeps half A's MFMAs — pure, results unused until the next
        // k-step — in front of the barrier; without them the optimiser sank all 4 MT MFMAs into the loop latch, behind every wait and read)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int nt = 0; nt < HW; ++nt) landed(acc[nt / 4][t][nt % 4]);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) — as a builtin: the compiler's own scoreboard must see that nothing is pending
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = HW; t < NTW; ++t) landed(wf[t]);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // -------- half B: MFMAs (mt, nt >= HW); behind MFMA idx: LDS-DMA piece idx of the stage after next (the rest behind the last one);
        // behind the first MFMAs of the first row groups the next stage's W sub-tiles 0 .. HW - 1; behind a row group's LAST MFMA the next stage's
        // A fragment of that row group (into the same registers: an MFMA has read its operands long before a ds_read issued behind it returns)
        {
            const unsigned nb = bufoff ^ (unsigned)STAGE_BYTES;
            const unsigned nw0 = wB0 + nb, nw1 = wB1 + nb, na0 = aB0 + nb, na1 = aB1 + nb;
            static_for<NB>([&](auto idx_tag) {
                constexpr int idx = decltype(idx_tag)::value, mt = idx / HW, nt = HW + idx % HW;
                acc[nt / 4][mt][nt % 4] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[nt], af[mt], acc[nt / 4][mt][nt % 4], 0, 0, 0, UNIT_SCALE, 0, UNIT_SCALE);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (idx < NL) issue_piece(idx, bufoff);
                if constexpr (idx == NB - 1) {
#pragma unroll
                    for (int i = NB; i < NL; ++i) issue_piece(i, bufoff);
                }
                // next stage's W sub-tile r (< HW) behind MFMA r * WSTEP: wf[r] was last used in half A
                constexpr int RPOS = NB > HW ? 2 : 1;
                if constexpr (!LAST && idx % RPOS == 0 && idx / RPOS < HW) wf[idx / RPOS] = lds_read32<(idx / RPOS) * 2048>(nw0, nw1);
                if constexpr (!LAST && idx % HW == HW - 1) af[mt] = lds_read32<mt * 2048>(na0, na1);
                __builtin_amdgcn_sched_barrier(0);
            });
        }
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int nt = HW; nt < NTW; ++nt) landed(acc[nt / 4][t][nt % 4]);
        advance_cursor();
        bufoff ^= (unsigned)STAGE_BYTES;
    };

    for (;;) {
        int cm0, cn0;
        tile_origin(c_vbid, cm0, cn0);
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[h][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the tile's first stage is in LDS and visible (the prologue's barrier / the previous tile's last mid-step barrier): its A fragments and W
        // sub-tiles 0, 1
        {
            const unsigned w0 = wB0 + bufoff, w1 = wB1 + bufoff, a0 = aB0 + bufoff, a1 = aB1 + bufoff;
            static_for<HW>([&](auto t_tag) { constexpr int t = decltype(t_tag)::value; wf[t] = lds_read32<t * 2048>(w0, w1); });
            static_for<MT>([&](auto t_tag) { constexpr int t = decltype(t_tag)::value; af[t] = lds_read32<t * 2048>(a0, a1); });
        }
        for (int kt = 0; kt < nk - 1; ++kt) kstep(std::false_type{});
        kstep(std::true_type{});

        // ---- epilogue: lane owns out[m][n .. n+3]; dequantise with a_scale[m] * w_scale[n] ----------------------------
        // everything the epilogue READS is fetched up front (residual tile first): left inside the (mt, nt) loop every
        // sub-tile's scale / bias / residual load was waited for on its own (gemm_epilogue.h, tools/probes/gemm_trace.py)
        static_for<NH>([&](auto h_tag) {
        constexpr int hh = decltype(h_tag)::value;
        f32x4 (&acch)[MT][4] = acc[hh];
        const int wave_n0 = cn0 + wn * (16 * NTW) + hh * 64;
        asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch below the k-loop: hoisted into it, it collides with the fragments' registers
        // (the residual tile is prefetched in 2-3 groups of the tile's rows: the whole 160-row tile next to the e4m3 fragments'
        // registers spilled; 2-3 exposed latencies instead of twenty)
        constexpr int HALF = MT >= 6 ? 1 : (MT + 1) / 2;  // rows (in 16-row units) per prefetch group (192-row tile: no register room)
        f32x4 res_v[(FLAGS & MQ_EPI_RESIDUAL) ? HALF : 1][4];
        auto prefetch_residual = [&](int mt_lo) {
            if (FLAGS & MQ_EPI_RESIDUAL) {
#pragma unroll
                for (int h = 0; h < HALF; ++h) {
                    const int m = cm0 + wm * (16 * MT) + (mt_lo + h) * 16 + l15;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const int n = wave_n0 + nt * 16 + g * 4;
                        const bool ok = mt_lo + h < MT && m < M && n < N;
                        if (FLAGS & MQ_EPI_OUT_F32) {
                            res_v[h][nt] = ok ? *(const f32x4*)(residual + (int64_t)m * ldc + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                        } else {   // bf16 residual stream: read-modify-write of the bf16 rows (BIAS | RESIDUAL, bf16 out)
                            const uint2 q = ok ? *(const uint2*)((const bf16_t*)residual + (int64_t)m * ldc + n) : make_uint2(0u, 0u);
                            res_v[h][nt] = f32x4{__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16),
                                                 __uint_as_float(q.y & 0xffff0000u)};
                        }
                    }
                }
            }
        };
        prefetch_residual(0);
        float sa_v[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = cm0 + wm * (16 * MT) + mt * 16 + l15;
            sa_v[mt] = ROWSCALE ? a_scale[m < M ? m : M - 1] : a_scalar;
        }
        f32x4 sw_v[4], bias_v[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = wave_n0 + nt * 16 + g * 4;
            sw_v[nt] = n < N ? *(const f32x4*)(w_scale + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            bias_v[nt] = ((FLAGS & MQ_EPI_BIAS) && n < N) ? *(const f32x4*)(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // one explicit, compiler-visible wait behind the up-front loads: with the next tile's LDS-DMA requests in flight hipcc cannot count
        // past them and would otherwise re-wait vmcnt(0) at the first use in every row group (gemm_epilogue.h, WAIT_LOADS)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        auto value = [&](int mt, int nt, int m, int n, bool ok, float sa) {
            f32x4 v = acch[mt][nt];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= sa * sw_v[nt][e];
            if (FLAGS & MQ_EPI_BIAS) v += bias_v[nt];
            if (FLAGS & MQ_EPI_GELU) {
                v = gelu_erf4(v);
            }
            if (FLAGS & MQ_EPI_QUICKGELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
            }
            if (FLAGS & MQ_EPI_RESIDUAL) v += res_v[(FLAGS & MQ_EPI_RESIDUAL) ? (mt % HALF) : 0][nt];
            return v;
        };
        auto to_fp8 = [&](const f32x4& v) {
            int w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[0] * inv_out), clamp448(v[1] * inv_out), 0, false);
            return __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[2] * inv_out), clamp448(v[3] * inv_out), w, true);
        };
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (mt > 0 && mt % HALF == 0) prefetch_residual(mt);  // (the previous group is finished: its registers are free)
            const int m = cm0 + wm * (16 * MT) + mt * 16 + l15;
            const bool m_ok = m < M;
            const float sa = sa_v[mt];
            if (!(FLAGS & MQ_EPI_OUT_F32) && wide) {  // `wide` is wave-uniform: every lane takes part in the swaps
                if (FLAGS & MQ_EPI_OUT_FP8) {
                    unsigned int pr[2][2];  // pr[p] = 8 consecutive codes of pair p after the first exchange
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        int cw[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int n = wave_n0 + (2 * p + h) * 16 + g * 4;
                            const bool ok = m_ok && n < N;
                            const f32x4 v = value(mt, 2 * p + h, m, n, ok, sa);
                            if (ok) amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                            cw[h] = to_fp8(v);
                        }
                        const auto r0 = __builtin_amdgcn_permlane16_swap((unsigned)cw[0], (unsigned)cw[1], false, false);
                        pr[p][0] = r0[0]; pr[p][1] = r0[1];  // n = pair block + (g&1)*16 + (g>>1)*8 .. +7
                    }
                    // lanes 32 apart (g and g^2) hold the two 8-code halves of the same 16 columns: lanes 32-63 of the first
                    // operand <-> lanes 0-31 of the second
                    const auto s0 = __builtin_amdgcn_permlane32_swap(pr[0][0], pr[1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(pr[0][1], pr[1][1], false, false);
                    // g < 2: (s0[0], s1[0]) = own pair-0 half, (s0[1], s1[1]) = lane g+2's pair-0 half (the next 8 columns)
                    // g >= 2: (s0[0], s1[0]) = lane g-2's pair-1 half, (s0[1], s1[1]) = own pair-1 half
                    const int n = wave_n0 + (g >> 1) * 32 + (g & 1) * 16;
                    uint8_t* dst = (uint8_t*)out + (int64_t)m * ldc + n;
                    if (m_ok && n + 16 <= N) *(uint4*)dst = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                    else if (m_ok && n < N) {  // ragged right edge (N % 4 == 0): dword by dword
                        const unsigned int wds[4] = {s0[0], s1[0], s0[1], s1[1]};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + 4 * e < N) *(unsigned int*)(dst + 4 * e) = wds[e];
                    }
                } else {
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        uint2 a, b;
                        {
                            const int n = wave_n0 + (2 * p) * 16 + g * 4;
                            const f32x4 v = value(mt, 2 * p, m, n, m_ok && n < N, sa);
                            a.x = pack_bf16x2(v[0], v[1]); a.y = pack_bf16x2(v[2], v[3]);
                        }
                        {
                            const int n = wave_n0 + (2 * p + 1) * 16 + g * 4;
                            const f32x4 v = value(mt, 2 * p + 1, m, n, m_ok && n < N, sa);
                            b.x = pack_bf16x2(v[0], v[1]); b.y = pack_bf16x2(v[2], v[3]);
                        }
                        const auto r0 = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
                        const auto r1 = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
                        const int n = wave_n0 + p * 32 + (g & 1) * 16 + (g >> 1) * 8;
                        bf16_t* dst = (bf16_t*)out + (int64_t)m * ldc + n;
                        if (m_ok && n + 8 <= N) *(uint4*)dst = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                        else if (m_ok && n < N) *(uint2*)dst = make_uint2(r0[0], r1[0]);
                    }
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int n = wave_n0 + nt * 16 + g * 4;
                    const bool ok = m_ok && n < N;
                    const f32x4 v = value(mt, nt, m, n, ok, sa);
                    if (!ok) continue;
                    const int64_t o = (int64_t)m * ldc + n;
                    if (FLAGS & MQ_EPI_OUT_F32) {
                        *(f32x4*)((float*)out + o) = v;
                    } else if (FLAGS & MQ_EPI_OUT_FP8) {
                        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                        *(int*)((uint8_t*)out + o) = to_fp8(v);
                    } else {
                        uint2 p;
                        p.x = pack_bf16x2(v[0], v[1]);
                        p.y = pack_bf16x2(v[2], v[3]);
                        *(uint2*)((bf16_t*)out + o) = p;
                    }
                }
            }
        }
        });
        c_vbid += gridDim.x;
        if (c_vbid >= num_tiles) break;
    }
    // the trailing (out-of-range) LDS-DMA requests must have retired before the workgroup's LDS can be handed to another one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((FLAGS & MQ_EPI_OUT_FP8) && amax_out) {
        amax = wave_max(amax);
        if (lane == 0) atomicMax((int*)amax_out, __float_as_int(amax));  // amax >= 0: int order == float order
    }
}

#ifdef MQ_GEMM_PROBE   // compile-and-inspect builds (tests/test_gemm_isa.py): ONE instantiation
#ifndef MQ_GEMM_PROBE_NH
#define MQ_GEMM_PROBE_NH 1
#endif
#ifndef MQ_GEMM_PROBE_WM
#define MQ_GEMM_PROBE_WM 2
#endif
__attribute__((used)) void* mq_gemm_fp8_probe() { return (void*)gemm_fp8_kernel<MQ_GEMM_PROBE, MQ_GEMM_PROBE_MT, (MQ_GEMM_PROBE_ROWSCALE != 0), MQ_GEMM_PROBE_NH, MQ_GEMM_PROBE_WM>; }
}  // namespace
#else
int choose_mt(int M, int N) {
    constexpr int BN = 128;
    const int tiles_n = (N + BN - 1) / BN;
    const int cands[4] = {2, 4, 5, 6};
    int best = 4;
    double best_cost = 1e30;
    for (int c = 0; c < 4; ++c) {
        const int mt = cands[c];
        const int64_t tiles = (int64_t)((M + 32 * mt - 1) / (32 * mt)) * tiles_n;
        const int64_t rounds = (tiles + RESIDENT_SLOTS - 1) / RESIDENT_SLOTS;
        const double cost = (double)rounds * (mt + 1.25);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = mt; }
    }
    return best;
}

struct Fp8Args {
    const void* A; int64_t lda; const void* W; int64_t ldw; const float* a_scale; const float* w_scale; const float* bias;
    const float* residual; void* out; int64_t ldc; const float* out_scale; float* amax; int M, N, K;
};

template <int FLAGS, int MT, bool ROWSCALE, int NH = 1, int WM = 2>
int launch_fp8_mt(const Fp8Args& a, hipStream_t s) {
    constexpr int BM = 16 * MT * WM, BN = 128 * NH;
    constexpr int LDS = 2 * (BM + BN) * BK;
    constexpr int SLOTS = (NH == 1 && WM == 2) ? RESIDENT_SLOTS : RESIDENT_SLOTS_BIG;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = mq_ensure_dyn_lds((const void*)gemm_fp8_kernel<FLAGS, MT, ROWSCALE, NH, WM>, LDS, attr_done); e != hipSuccess) {
        mq_set_error("mq_gemm_fp8: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return MQ_ERR_HIP;
    }
    // the LDS-DMA addresses both operands through 32-bit buffer offsets: the weight must fit, a taller A goes in row chunks (rows are
    // independent; whole tiles per chunk), exactly as launch_gemm_mt does in gemm_bf16.hip
    const uint64_t lim = mq_gemm_addr_limit;
    const uint64_t w_bytes = (uint64_t)(a.N - 1) * (uint64_t)a.ldw + (uint64_t)a.K;
    if (w_bytes > lim) {
        mq_set_error("mq_gemm_fp8: weight matrix of %llu bytes exceeds the %llu bytes a launch can address", (unsigned long long)w_bytes, (unsigned long long)lim);
        return MQ_ERR_INVALID;
    }
    int64_t max_rows = (uint64_t)a.K > lim ? 0 : (int64_t)((lim - (uint64_t)a.K) / (uint64_t)a.lda) + 1;
    max_rows = max_rows / BM * BM;
    if (max_rows < BM) {
        mq_set_error("mq_gemm_fp8: lda=%ld too large", (long)a.lda);
        return MQ_ERR_INVALID;
    }
    const int tiles_n = (a.N + BN - 1) / BN;
    // 16-byte epilogue stores need 16-B aligned rows (bf16: ldc % 8; e4m3: ldc % 16)
    const int row_align = (FLAGS & MQ_EPI_OUT_FP8) ? 16 : 8;
    const int wide = (mq_gemm_knob_wide && !(FLAGS & MQ_EPI_OUT_F32) && a.ldc % row_align == 0 && ((uintptr_t)a.out & 15) == 0) ? 1 : 0;
    const size_t out_row = (size_t)a.ldc * ((FLAGS & MQ_EPI_OUT_F32) ? 4 : (FLAGS & MQ_EPI_OUT_FP8) ? 1 : 2);
    const size_t res_row = (size_t)a.ldc * (((FLAGS & MQ_EPI_RESIDUAL) && !(FLAGS & MQ_EPI_OUT_F32)) ? 2 : 4);
    for (int64_t r0 = 0; r0 < a.M; r0 += max_rows) {
        const int m = (int)((a.M - r0) < max_rows ? (a.M - r0) : max_rows);
        const int tiles_m = (m + BM - 1) / BM;
        const int num_tiles = tiles_m * tiles_n;
        const int knob_cgroup = mq_gemm_knob_cgroup;
        const int cgroup = (knob_cgroup > 0 && tiles_n > knob_cgroup && tiles_m >= 16) ? knob_cgroup : 0;
        const int band_rows = (tiles_m + 7) / 8;
        const int grid = num_tiles > SLOTS ? SLOTS : num_tiles;
        const uint64_t a_bytes = (uint64_t)(m - 1) * (uint64_t)a.lda + (uint64_t)a.K;
        hipLaunchKernelGGL((gemm_fp8_kernel<FLAGS, MT, ROWSCALE, NH, WM>), dim3(grid), dim3(128 * WM), LDS, s, (const uint8_t*)a.A + r0 * a.lda, a.lda,
                           (const uint8_t*)a.W, a.ldw, ROWSCALE ? a.a_scale + r0 : a.a_scale, a.w_scale, a.bias,
                           a.residual ? (const float*)((const char*)a.residual + (size_t)r0 * res_row) : nullptr, (void*)((char*)a.out + (size_t)r0 * out_row),
                           a.ldc, a.out_scale, a.amax, m, a.N, a.K, tiles_n, num_tiles, cgroup, band_rows, wide, (unsigned)a_bytes, (unsigned)w_bytes);
        MQ_CHECK_LAUNCH("mq_gemm_fp8");
    }
    return MQ_OK;
}

// The BIG tile (192 x 256 x 128, 8 waves; MT = 3: the 256-row form needs 274 registers, this one 232-242) takes a problem whose tiles fill the chip's
// 256 workgroups for at least two rounds at >= 85 % fill, whose columns fill their last 256-wide tile and whose K is long (>= 2 048: the ViT-L/14 fc2
// at >= 128 images, square GEMMs of the C ABI); everything else stays on the (32 MT) x 128 tiles.  mq_tune("gemm_nh", 1) forbids it, 3 forces it wherever N >= 256.
bool plan_fp8_big(int M, int N, int K) {
    const int nh = mq_gemm_fp8_big;
    if (nh == 1 || N < 256) return false;
    if (nh == 3) return true;
    const int tiles_n = (N + 255) / 256;
    if ((double)N / (tiles_n * 256.0) < 0.9) return false;
    const int64_t tiles = (int64_t)((M + 191) / 192) * tiles_n;
    const int64_t rounds = (tiles + RESIDENT_SLOTS_BIG - 1) / RESIDENT_SLOTS_BIG;
    // measured (profiles/r06e_fp8_big_tile_ab.txt): -17 % at 8192^3 (2.13 -> 2.56 PF), -5 ... -9 % on the ViT-L/14 fc2 (K = 4 096) at 128 / 240 images; at
    // K = 1 024 the tile's un-overlapped prologue / epilogue (one workgroup per CU) costs more than its k-loop saves (+1 ... +20 %): long K only
    return K >= 2048 && rounds >= 2 && (double)tiles / (double)(rounds * RESIDENT_SLOTS_BIG) >= 0.85;
}

template <int FLAGS, bool ROWSCALE>
int launch_fp8(const Fp8Args& a, int force_mt, hipStream_t s) {
    if (!force_mt && plan_fp8_big(a.M, a.N, a.K)) return launch_fp8_mt<FLAGS, 3, ROWSCALE, 2, 4>(a, s);
    const int mt = force_mt ? force_mt : choose_mt(a.M, a.N);
    switch (mt) {
        case 2: return launch_fp8_mt<FLAGS, 2, ROWSCALE>(a, s);
        case 5: return launch_fp8_mt<FLAGS, 5, ROWSCALE>(a, s);
        case 6: return launch_fp8_mt<FLAGS, 6, ROWSCALE>(a, s);
        default: return launch_fp8_mt<FLAGS, 4, ROWSCALE>(a, s);
    }
}

// ---- per-output-channel weight quantisation: W bf16 [N, K] -> W8 e4m3 [N, K] + scale[N] = absmax / 448 -------------
__global__ __launch_bounds__(256) void quantize_rows_kernel(const bf16_t* __restrict__ W, int64_t ldw, uint8_t* __restrict__ W8, int64_t ld8,
                                                           float* __restrict__ scale, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const bf16_t* w = W + (int64_t)row * ldw;
    float mx = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const uint2 p = *(const uint2*)(w + k);
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(bf16_to_f32((bf16_t)(p.x & 0xffff))), fabsf(bf16_to_f32((bf16_t)(p.x >> 16)))),
                             fmaxf(fabsf(bf16_to_f32((bf16_t)(p.y & 0xffff))), fabsf(bf16_to_f32((bf16_t)(p.y >> 16))))));
    }
    mx = wave_max(mx);
    const float sc = mx > 0.f ? mx / 448.f : 1.f;
    const float inv = 1.f / sc;
    if (lane == 0) scale[row] = sc;
    uint8_t* o = W8 + (int64_t)row * ld8;
    for (int k = lane * 4; k < K; k += 256) {
        const uint2 p = *(const uint2*)(w + k);
        int wd = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16_to_f32((bf16_t)(p.x & 0xffff)) * inv), clamp448(bf16_to_f32((bf16_t)(p.x >> 16)) * inv), 0, false);
        wd = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(bf16_to_f32((bf16_t)(p.y & 0xffff)) * inv), clamp448(bf16_to_f32((bf16_t)(p.y >> 16)) * inv), wd, true);
        *(int*)(o + k) = wd;
    }
}

}  // namespace

mq_knob mq_gemm_fp8_force_mt{0};  // set through mq_tune("gemm_mt", v) (shared knob, see gemm_bf16.hip)

extern "C" int mq_gemm_fp8(const void* d_A8, int64_t lda, const void* d_W8, int64_t ldw, const float* d_a_scale, int a_scale_per_row,
                           const float* d_w_scale, const float* d_bias, const float* d_residual, void* d_out, int64_t ldc,
                           const float* d_out_scale, float* d_amax, int64_t M, int64_t N, int64_t K, int flags, void* stream) {
    MQ_CHECK_ARG(d_A8 && d_W8 && d_out && d_a_scale && d_w_scale, "mq_gemm_fp8: null operand");
    MQ_CHECK_ARG(M >= 1 && N >= 4 && K >= BK, "mq_gemm_fp8: bad shape M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
    MQ_CHECK_ARG(K % BK == 0, "mq_gemm_fp8: K=%ld must be a multiple of %d", (long)K, BK);
    MQ_CHECK_ARG(N % 4 == 0, "mq_gemm_fp8: N=%ld must be a multiple of 4", (long)N);
    MQ_CHECK_ARG(lda % 16 == 0 && ldw % 16 == 0 && ldc % 4 == 0, "mq_gemm_fp8: leading dims must keep 16-byte rows");
    MQ_CHECK_ARG(M < (1 << 30) && N < (1 << 30) && K < (1 << 30), "mq_gemm_fp8: shape too large");
    MQ_CHECK_ARG(!(flags & MQ_EPI_BIAS) || d_bias, "mq_gemm_fp8: MQ_EPI_BIAS without bias");
    MQ_CHECK_ARG(!(flags & MQ_EPI_RESIDUAL) || d_residual, "mq_gemm_fp8: MQ_EPI_RESIDUAL without residual");
    MQ_CHECK_ARG(!(flags & MQ_EPI_OUT_FP8) || d_out_scale, "mq_gemm_fp8: MQ_EPI_OUT_FP8 without out_scale");
    MQ_TRY(mq_device_ok());   // 256 CUs in 8 XCDs or nothing (runtime.hip)
    hipStream_t s = (hipStream_t)stream;
    MqProfScope prof(0, s, 2.0 * (double)M * (double)N * (double)K);
    const Fp8Args a{d_A8, lda, d_W8, ldw, d_a_scale, d_w_scale, d_bias, d_residual, d_out, ldc, d_out_scale, d_amax, (int)M, (int)N, (int)K};
    const int fm = mq_gemm_fp8_force_mt;
#define MQ_FP8_CASE(F)                                                                     \
    case (F):                                                                              \
        return a_scale_per_row ? launch_fp8<(F), true>(a, fm, s) : launch_fp8<(F), false>(a, fm, s)
    switch (flags) {
        MQ_FP8_CASE(MQ_EPI_OUT_F32);
        MQ_FP8_CASE(MQ_EPI_BIAS);
        MQ_FP8_CASE(MQ_EPI_BIAS | MQ_EPI_GELU | MQ_EPI_OUT_FP8);
        MQ_FP8_CASE(MQ_EPI_BIAS | MQ_EPI_QUICKGELU | MQ_EPI_OUT_FP8);
        MQ_FP8_CASE(MQ_EPI_BIAS | MQ_EPI_RESIDUAL | MQ_EPI_OUT_F32);
        MQ_FP8_CASE(MQ_EPI_BIAS | MQ_EPI_RESIDUAL);   // bf16 residual in / out (d_residual and d_out are bf16 [M, ldc]: the bf16 residual stream)
        default:
            mq_set_error("mq_gemm_fp8: unsupported epilogue flag combination 0x%x", flags);
            return MQ_ERR_INVALID;
    }
#undef MQ_FP8_CASE
}

extern "C" int mq_quantize_weights_fp8(const void* d_W_bf16, int64_t ldw, void* d_W8, int64_t ld8, float* d_scale, int64_t N, int64_t K,
                                       void* stream) {
    MQ_CHECK_ARG(d_W_bf16 && d_W8 && d_scale, "mq_quantize_weights_fp8: null pointer");
    MQ_CHECK_ARG(N >= 1 && K >= 4 && K % 4 == 0 && ldw % 4 == 0 && ld8 % 4 == 0, "mq_quantize_weights_fp8: bad shape N=%ld K=%ld", (long)N, (long)K);
    hipLaunchKernelGGL(quantize_rows_kernel, dim3((unsigned)cdiv64(N, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)d_W_bf16, ldw,
                       (uint8_t*)d_W8, ld8, d_scale, (int)N, (int)K);
    MQ_CHECK_LAUNCH("mq_quantize_weights_fp8");
    return MQ_OK;
}
#endif  // MQ_GEMM_PROBE
