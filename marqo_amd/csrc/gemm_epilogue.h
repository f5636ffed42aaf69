// Epilogue of the bf16 MFMA GEMM kernel (gemm_bf16.hip).
// Accumulator layout (operands are fed swapped, mfma(Wfrag, Afrag)): for sub-tile (mt, nt)
//   D[i][j] = sum_k Wfrag[i][k] * Afrag[j][k]: column j = lane & 15 -> m, row i = 4*g + reg -> n,
// so a lane owns out[m][n .. n+3]: bias / residual / fp32 out are 16-byte vector accesses.
// bf16 out: a lane's 4 values are only 8 bytes, and the 4 stores of a row (nt = 0..3) each cover 32-B row segments.  With
// WIDE the two 16-column blocks of a pair (nt, nt+1) are exchanged between the lanes 16 apart (v_permlane16_swap: odd
// 16-lane rows of the first operand <-> even rows of the second), after which a lane holds 8 CONSECUTIVE n (16 bytes) and a
// store instruction covers 64 contiguous bytes per row: half the store instructions for the same bytes (the store tail of a
// K = 768 tile is issue-bound, cdna_hip_programming.md T21).
//
// LayerNorm folding (pre-LN blocks on the bf16 residual stream, round 4): LN(x) @ W^T = rstd * (x @ (gamma*W)^T - mean * colsum(gamma*W))
// + (b + W @ beta).  MQ_EPI_LN_APPLY (the QKV / fc1 GEMM): A is the UN-normalised bf16 stream itself, W / bias / colsum are pre-folded with
// gamma / beta at load, and (mean, rstd) per row come from a one-pass statistics kernel (mq_row_stats, rowops.hip: reads the stream once,
// writes 8 bytes per row) — the LayerNorm launch that read AND wrote the whole stream, and the normalised copy, are gone.  The epilogue fetches
// its MT rows' statistics with the bias, up front.  MQ_EPI_ROW_STATS (the residual GEMM in FRONT of such a LayerNorm, bf16 read-modify-write form): the
// epilogue also leaves, per row and 64-column wave slot, (sum, sum of squares) of the bf16 values it stores — plain stores, one writer per element,
// fixed order: deterministic — so the statistics pass shrinks to a finalise over ceil(N / 64) partials per row (mq_row_stats_finalize) instead of a
// read of the whole stream.  (Tried and rejected this round: accumulating the statistics inside the GEMM from the staged
// A tiles — 40 extra VALU operations per k-step cost the k-loop 9 %, more than the LayerNorm launch they replaced; profiles/r04j_*.)
#pragma once
#include "common.h"

struct GemmLn {
    const float* colsum;     // LN_APPLY: [N]  sum_k bf16(gamma_k * W[n,k])
    const float2* rowstats;  // LN_APPLY: [M]  (mean, rstd) of row m of A
    // ROW_STATS: [nslots][part_ld]  (sum, sum of squares) of the bf16 values this launch leaves in columns 64 s .. 64 s + 63 of row m, SLOT-major (round 6: a
    // store instruction's 16 row lanes write 128 contiguous bytes, and the finalise kernel's threads — one per row — read coalesced; row-major, every lane
    // touched a line of its own: the gated GEMM's 64 slots per row cost it 12 %)
    float2* partials;
    int nslots;              // ROW_STATS: ceil(N / 64)
    int64_t part_ld;         // ROW_STATS: rows of the whole partials buffer (the slot stride)
    // ROW_STATS with the finalise INSIDE the launch (round 6, mq_gemm_bf16_rsf): every wave, when its rows' partials have left, arrives at the counter of
    // its row band (a row tile x its position along M); the LAST wave to arrive — the band's partials are then all written — sums them in slot order
    // and writes (mean, rstd), exactly as row_stats_finalize_kernel would (mq_finalize_stats), and puts the counter back to zero.  Partials travel as
    // agent-scope (write-through) stores and are read back with agent-scope loads: the band's tiles may run on different XCDs.
    unsigned* band_ctr;      // [row tiles x waves along M], zero between launches; nullptr = no in-launch finalise (the caller runs mq_row_stats_finalize)
    float2* stats_out;       // [M]
    float inv_w, eps;
    int band_target;         // arrivals that complete a band: column tiles x 2 waves along N
    // the weight prefetch the finalise launch used to carry (one dword per 128-byte line of the next GEMMs' weights), issued behind the epilogue of a
    // workgroup's first tile: the (cold) loads ride out with the tile's stores, in front of the arrival's vmcnt(0)
    const unsigned* pf_a;
    const unsigned* pf_b;
    unsigned pf_na, pf_nb;
};

// the carried weight prefetch (GemmLn::pf_*): thread t of the grid touches lines t and t + (threads of the grid).  Inline asm, so that the loads stay
// where they are issued (vector-memory operations retire in order: a cold load in FRONT of a load somebody waits for would make that wait a cold one);
// the caller passes pf_regs through an empty asm behind the next vmcnt(0) (gemm_band_arrive's), so that nothing reuses the registers before the loads return
__device__ __forceinline__ void gemm_pf_issue(const GemmLn& ln, unsigned (&pf_regs)[2]) {
    const unsigned nt = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const unsigned c = t + j * nt;
        const unsigned* src = c < ln.pf_na ? ln.pf_a + (size_t)c * 32 : (c - ln.pf_na < ln.pf_nb ? ln.pf_b + (size_t)(c - ln.pf_na) * 32 : nullptr);
        if (src) asm volatile("global_load_dword %0, %1, off" : "=v"(pf_regs[j]) : "v"(src) : "memory");
    }
}

// the last-arriver finalise of a row band (see GemmLn): called by every wave behind its epilogue; rows [row0, row0 + nrows) x all column slots
__device__ __forceinline__ void gemm_band_arrive(const GemmLn& ln, int band, int row0, int nrows, int M, int lane) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my partials (write-through stores) have been acknowledged
    unsigned old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(ln.band_ctr + band, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
    if ((int)old + 1 != ln.band_target) return;
    if (lane == 0) __hip_atomic_store(ln.band_ctr + band, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // one lane per row (a wave's band is at most 96 rows: two rows per lane), a row's partials fetched 16 at a time BEFORE any of them is added — the
    // first form (one dependent L2 round trip per slot) left the band's last wave 12-16 round trips behind the kernel's tail (profiles/r06c)
    const int m0 = row0 + lane, m1 = row0 + 64 + lane;
    const bool ok0 = lane < nrows && m0 < M, ok1 = 64 + lane < nrows && m1 < M;
    const unsigned long long* p0 = (const unsigned long long*)(ln.partials + (ok0 ? m0 : row0));   // (slot i at + i * part_ld)
    const unsigned long long* p1 = (const unsigned long long*)(ln.partials + (ok1 ? m1 : row0));
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
    for (int i0 = 0; i0 < ln.nslots; i0 += 16) {
        unsigned long long q[2][16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            q[0][j] = (ok0 && i0 + j < ln.nslots) ? __hip_atomic_load(p0 + (i0 + j) * ln.part_ld, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            q[1][j] = (ok1 && i0 + j < ln.nslots) ? __hip_atomic_load(p1 + (i0 + j) * ln.part_ld, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)   // slot order, as row_stats_finalize_kernel adds them
            if (i0 + j < ln.nslots) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    s1[h] += __uint_as_float((unsigned)q[h][j]);
                    s2[h] += __uint_as_float((unsigned)(q[h][j] >> 32));
                }
            }
    }
    if (ok0) ln.stats_out[m0] = mq_finalize_stats(s1[0], s2[0], ln.inv_w, ln.eps);
    if (ok1) ln.stats_out[m1] = mq_finalize_stats(s1[1], s2[1], ln.inv_w, ln.eps);
}

// RG = rows (16-row units) whose residual is prefetched together: the whole tile where the registers allow (the 4-wave kernel
// after its k-loop), a few rows at a time in the 8-wave kernel whose accumulators already fill the file.
// WAIT_LOADS (gemm_bf16.hip): one explicit, compiler-visible s_waitcnt vmcnt(0) behind the up-front loads.  That kernel has LDS-DMA requests of
// the next tile in flight here, so hipcc cannot count past them: without the explicit wait it re-waits vmcnt(0) at the first use of the
// bias in every row group — draining the row groups' own stores one after the other — and, seeing the fragment registers as possibly
// pending load destinations, puts another vmcnt(0) into the k-loop.
template <int FLAGS, int MT, int RG = MT, bool WAIT_LOADS = false>
__device__ __forceinline__ void gemm_epilogue(f32x4 (&acc)[MT][4], const float* __restrict__ bias, const float* residual, void* out,
                                              int64_t ldc, int M, int N, int wave_m0, int wave_n0, int l15, int g, bool wide = false,
                                              const GemmLn* lnp = nullptr, const float* lds_bias = nullptr) {
    constexpr bool BF16_OUT = !(FLAGS & MQ_EPI_OUT_F32);
    // bf16 residual stream (towers.hip): MQ_EPI_RESIDUAL without MQ_EPI_OUT_F32 = the residual is read as bf16 and the sum written
    // as bf16, in place — half the epilogue bytes of the fp32 stream, the memory-bound part of the K = 768 residual GEMMs
    constexpr bool RES_BF16 = (FLAGS & MQ_EPI_RESIDUAL) && BF16_OUT;
    constexpr bool LN_APPLY = (FLAGS & MQ_EPI_LN_APPLY) != 0, ROW_STATS = (FLAGS & MQ_EPI_ROW_STATS) != 0, GLU = (FLAGS & MQ_EPI_GLU) != 0;
    static_assert(!GLU || (BF16_OUT && !(FLAGS & (MQ_EPI_RESIDUAL | MQ_EPI_GELU | MQ_EPI_QUICKGELU))), "GLU: bias (+ LN apply, + row statistics of the product), bf16 out");
    static_assert(!ROW_STATS || RES_BF16 || GLU, "ROW_STATS rides on the bf16 read-modify-write residual epilogue or on the gated product");
    // (LN_APPLY with the bf16 residual: the EVA02 sub-LayerNorms — attn.norm in front of the out-projection, mlp.norm in front of fc2 — folded into
    // those GEMMs, whose A rows' statistics come from the attention kernel / the gated epilogue; mq_gemm_bf16_lnrs)
    static_assert(!LN_APPLY || (BF16_OUT && (!(FLAGS & MQ_EPI_RESIDUAL) || RES_BF16)), "LN_APPLY: bf16 out (QKV / fc1, or the bf16 residual epilogue)");
    float row_mean = 0.f, row_rstd = 1.f;
    // Everything the epilogue READS is fetched up front, the long-latency residual tile first.  Measured with the phase trace
    // (tools/probes/gemm_trace.py): left inside the (mt, nt) loop, each sub-tile's bias / residual load was waited for on its
    // own — the compiler cannot move a load of `residual` above the earlier stores to `out` (they alias: the update is in
    // place) and did not hoist the bias either — which cost 20 serialised L2 / HBM round trips per tile: 9 k cycles for a
    // bias, 24 k for bias + fp32 residual, against 17 k for the whole 12-step k-loop at K = 768.
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch below the k-loop (hoisted into it, it collides with the fragments' registers)
    f32x4 res_v[(FLAGS & MQ_EPI_RESIDUAL) ? RG : 1][4];
    // LN_APPLY on top of the residual epilogue: the rows' (mean, rstd) travel with their residual group instead of all MT up front (registers)
    constexpr bool LN_RES = LN_APPLY && (FLAGS & MQ_EPI_RESIDUAL) != 0;
    float2 ms_g[LN_RES ? RG : 1];
    auto prefetch_residual = [&](int mt_lo) {
        if (FLAGS & MQ_EPI_RESIDUAL) {
#pragma unroll
            for (int h = 0; h < RG; ++h) {
                const int m = wave_m0 + (mt_lo + h) * 16 + l15;
                if constexpr (LN_RES) ms_g[h] = (mt_lo + h < MT && m < M) ? lnp->rowstats[m] : make_float2(0.f, 1.f);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int n = wave_n0 + nt * 16 + g * 4;
                    if (RES_BF16) {
                        uint2 q = make_uint2(0u, 0u);
                        if (mt_lo + h < MT && m < M && n < N) q = *(const uint2*)((const bf16_t*)residual + (int64_t)m * ldc + n);
                        res_v[h][nt] = f32x4{__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16),
                                             __uint_as_float(q.y & 0xffff0000u)};
                    } else {
                        res_v[h][nt] = (mt_lo + h < MT && m < M && n < N) ? *(const f32x4*)(residual + (int64_t)m * ldc + n)
                                                                        : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
            }
        }
    };
    prefetch_residual(0);
    f32x4 bias_v[4], cs_v[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = wave_n0 + nt * 16 + g * 4;
        // lds_bias: the wave's 64 bias values staged in LDS during the k-loop (zero past N) — an L2 round trip (~1.5-3 k cycles at the
        // head of every tile's epilogue, profiles/r01d_gemm_phase_trace.txt: "qkv plain" vs "qkv bias") becomes a ds_read
        if ((FLAGS & MQ_EPI_BIAS) && lds_bias) bias_v[nt] = *(const f32x4*)(lds_bias + nt * 16 + g * 4);
        else bias_v[nt] = ((FLAGS & MQ_EPI_BIAS) && n < N) ? *(const f32x4*)(bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
        cs_v[nt] = (LN_APPLY && n < N) ? *(const f32x4*)(lnp->colsum + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float2 ms_v[(LN_APPLY && !LN_RES) ? MT : 1];
    if (LN_APPLY && !LN_RES) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = wave_m0 + mt * 16 + l15;
            ms_v[mt] = m < M ? lnp->rowstats[m] : make_float2(0.f, 1.f);
        }
    }
    if (WAIT_LOADS) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), lgkmcnt / expcnt untouched
    // value of one (mt, nt) sub-tile after (LN apply) / bias / activation / residual
    auto value = [&](int mt, int nt, int m, int n, bool ok) {
        f32x4 v = acc[mt][nt];
        if (LN_APPLY) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = row_rstd * (v[e] - row_mean * cs_v[nt][e]);
        }
        if (FLAGS & MQ_EPI_BIAS) v += bias_v[nt];
        if (FLAGS & MQ_EPI_GELU) {
            v = gelu_erf4(v);
        }
        if (FLAGS & MQ_EPI_QUICKGELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
        }
        if (FLAGS & MQ_EPI_RESIDUAL) v += res_v[(FLAGS & MQ_EPI_RESIDUAL) ? (mt % RG) : 0][nt];
        return v;
    };
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (mt > 0 && mt % RG == 0) prefetch_residual(mt);  // (the previous group is finished: its registers are free)
        const int m = wave_m0 + mt * 16 + l15;
        const bool m_ok = m < M;
        if constexpr (LN_RES) {
            row_mean = ms_g[mt % RG].x;
            row_rstd = ms_g[mt % RG].y;
        } else if (LN_APPLY) {
            row_mean = ms_v[LN_APPLY ? mt : 0].x;
            row_rstd = ms_v[LN_APPLY ? mt : 0].y;
        }
        // ROW_STATS: this lane's share of (sum, sum of squares) of row m over the wave's 64 columns, as (even, odd) element pairs: packed adds / fmas, no
        // branch (columns past N count as zeros) — the scalar, branched form cost the gated GEMM's epilogue 6 % (tools/probes/lnrs_bench.py)
        f32x2_t st1v = {0.f, 0.f}, st2v = {0.f, 0.f};
        auto stat_add = [&](uint2 pk, bool ok) {   // of the ROUNDED values: they are what the next GEMM multiplies
            if (ROW_STATS) {
                const unsigned x = ok ? pk.x : 0u, y = ok ? pk.y : 0u;
                const f32x2_t a = {__uint_as_float(x << 16), __uint_as_float(x & 0xffff0000u)}, b = {__uint_as_float(y << 16), __uint_as_float(y & 0xffff0000u)};
                st1v += a + b;
                st2v = __builtin_elementwise_fma(a, a, __builtin_elementwise_fma(b, b, st2v));
            }
        };
        if constexpr (GLU) {
            // gated MLP (marqo_hip.h, MQ_EPI_GLU): sub-tiles (0, 1) and (2, 3) are (up, gate) of the same 16 hidden units — the lane's 4 consecutive GEMM
            // columns of sub-tile 2 p are units u .. u + 3 with u = wave_n0 / 2 + 16 p + 4 g, their gates sit in the same registers of sub-tile 2 p + 1
            uint2 pk[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int n = wave_n0 + (2 * p) * 16 + g * 4;
                const bool ok = m_ok && n < N;
                const f32x4 up = value(mt, 2 * p, m, n, ok), gt = value(mt, 2 * p + 1, m, n + 16, ok);
                pk[p].x = pack_bf16x2(up[0] * silu(gt[0]), up[1] * silu(gt[1]));
                pk[p].y = pack_bf16x2(up[2] * silu(gt[2]), up[3] * silu(gt[3]));
                stat_add(pk[p], ok);   // (ROW_STATS: of the rounded product — slot = this wave's 64 GEMM columns = 32 hidden units)
            }
            const int u0 = wave_n0 >> 1, NU = N >> 1;
            if (wide) {   // the two 16-unit blocks exchanged between lanes 16 apart: a lane then owns 8 consecutive units (one 16-byte store)
                const auto r0 = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
                const int u = u0 + (g & 1) * 16 + (g >> 1) * 8;
                bf16_t* dst = (bf16_t*)out + (int64_t)m * ldc + u;
                if (m_ok && u + 8 <= NU) *(uint4*)dst = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                else if (m_ok && u < NU) *(uint2*)dst = make_uint2(r0[0], r1[0]);
            } else {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int u = u0 + 16 * p + 4 * g;
                    if (m_ok && u < NU) *(uint2*)((bf16_t*)out + (int64_t)m * ldc + u) = pk[p];
                }
            }
        } else if (BF16_OUT && wide) {  // `wide` is wave-uniform: every lane takes part in the swaps
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                uint2 a, b;
                {
                    const int n = wave_n0 + (2 * p) * 16 + g * 4;
                    const f32x4 v = value(mt, 2 * p, m, n, m_ok && n < N);
                    a.x = pack_bf16x2(v[0], v[1]); a.y = pack_bf16x2(v[2], v[3]);
                    stat_add(a, m_ok && n < N);
                }
                {
                    const int n = wave_n0 + (2 * p + 1) * 16 + g * 4;
                    const f32x4 v = value(mt, 2 * p + 1, m, n, m_ok && n < N);
                    b.x = pack_bf16x2(v[0], v[1]); b.y = pack_bf16x2(v[2], v[3]);
                    stat_add(b, m_ok && n < N);
                }
                const auto r0 = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
                // the lane now owns n = base .. base+7 with base = pair block + (g&1)*16 + (g>>1)*8, low half in (r0[0], r1[0])
                const int n = wave_n0 + p * 32 + (g & 1) * 16 + (g >> 1) * 8;
                bf16_t* dst = (bf16_t*)out + (int64_t)m * ldc + n;
                if (m_ok && n + 8 <= N) *(uint4*)dst = make_uint4(r0[0], r1[0], r0[1], r1[1]);
                else if (m_ok && n < N) *(uint2*)dst = make_uint2(r0[0], r1[0]);  // N % 4 == 0: exactly the low half is in range
            }
        } else {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int n = wave_n0 + nt * 16 + g * 4;
                const bool ok = m_ok && n < N;
                const f32x4 v = value(mt, nt, m, n, ok);
                const int64_t o = (int64_t)m * ldc + n;
                if (!ok) continue;
                if (!BF16_OUT) {
                    *(f32x4*)((float*)out + o) = v;
                } else {
                    uint2 pk;
                    pk.x = pack_bf16x2(v[0], v[1]);
                    pk.y = pack_bf16x2(v[2], v[3]);
                    *(uint2*)((bf16_t*)out + o) = pk;
                    stat_add(pk, true);
                }
            }
        }
        if (ROW_STATS) {  // the row's 4 lanes (g = 0..3) add up in a fixed order; one writer per (row, slot)
            float st1 = st1v[0] + st1v[1], st2 = st2v[0] + st2v[1];
            // across the row's 4 lanes with the swap instructions (VALU; bit for bit the sums of the __shfl_xor form they replace — same operands, same order
            // — without its ds_bpermute round trips: -5 us on the gated GEMM): rows (0,1) and (2,3) of 16 lanes, then the wave's halves
            auto fold = [](float v) {
                const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                const float w = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(w), __float_as_uint(w), false, false);
                return __uint_as_float(q[0]) + __uint_as_float(q[1]);
            };
            st1 = fold(st1);
            st2 = fold(st2);
            if (g == 0 && m_ok && wave_n0 < N) {
                float2* dst = lnp->partials + (int64_t)(wave_n0 >> 6) * lnp->part_ld + m;
                // in-launch finalise (band_ctr): write-through, the band's last wave reads them back inside this launch.  Otherwise a plain store: the next
                // LAUNCH reads them (measured neutral on the gated GEMM; it keeps memory-side acknowledgements out of a persistent tile walk's vmcnt waits)
                if (lnp->band_ctr)
                    __hip_atomic_store((unsigned long long*)dst, (unsigned long long)__float_as_uint(st1) | ((unsigned long long)__float_as_uint(st2) << 32), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                else *dst = make_float2(st1, st2);
            }
        }
    }
}
