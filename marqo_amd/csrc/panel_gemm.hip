// LayerNorm-folded GEMM over ONE image's token rows per workgroup ("row-panel GEMM") — the QKV and fc1 GEMMs of the ViT-B/32 image tower at chip-filling
// batches (T = 50 tokens, K = 768: BASELINE configs[1], the headline): K3 / K5(fc1) of SURVEY.md §8a in the form attn_proj.hip's out-projection runs.
// Reference call site: /root/reference/src/marqo/core/inference/embedding_models/open_clip_model.py:249-266 (encode_image -> the third-party model's
// residual attention block: qkv = in_proj(ln_1(x)), h = gelu(c_fc(ln_2(x)))).
//
//   out[rows, N] = act( LN(x) @ W0^T + b0 )   as mq_gemm_bf16_ln computes it: W = bf16(gamma * W0), bias = b0 + W0 @ beta, colsum[n] = sum_k W[n, k],
//                                             rowstats = (mean, rstd) per row; out = act(rstd * (x @ W^T - mean * colsum) + bias), bf16
//
// Why: the tiled GEMM runs these K = 768 shapes at 0.34-0.37 of the matrix peak — 12 k-steps per 160 x 128 tile, and what surrounds the
// k-loop (cold first stages, epilogue, the last round's tail) is as long as the loop.  One workgroup (8 wave64s) per image instead: the image's 50 x 768
// rows are staged ONCE as the MFMA token operand (12 k-step tiles of [64 tokens][128 B], XOR-swizzled: 96 KB of LDS), and the weight streams through
// per-wave private rings (four 2-KiB slots, three units in flight, counted vmcnt waits, NO workgroup barrier after the panel has landed) in passes of 768
// output columns (96 accumulator registers per lane); a pass's epilogue (LayerNorm apply, bias, GELU, 16-byte stores) runs with the next pass's first units
// already in flight.  64 FLOP per byte of weight — ON the ridge of the CU's vector-memory path (64 B/clk): the k-loop sustains ~0.67 of it, with nothing
// around it.  The price: 64-row tiles for 50 tokens (78 % useful), every CU streams the whole weight.  The k order (k-steps of 64, two 32-deep halves) and
// the epilogue's operation order are the tiled kernel's: bit-identical output (tests/test_panel_gemm_gpu.py).
//
// MEASURED, AND NOT THE DEFAULT (profiles/r06za_panel_gemm_ab.txt): correct and bit-identical, and 17-30 % SLOWER than the tiled kernel at these shapes
// (256 images: QKV 62.5 vs 53.2 us, fc1 83.0 vs 70.9 us; headline -6 %).  Where the form wins — attn_proj.hip's out-projection — the weight is 1.2 MB and
// lives in every XCD's L2; here 3.5 / 4.7 MB stream per CU with 48 KB in flight (the ring is what the 96 KB panel leaves of the LDS), 15-16 us per 768-column
// pass against 13 us there, and even with the epilogue switched off (53 us) the QKV GEMM only ties the tiled kernel.  mq_tune("panel_gemm", n) / MQ_PANEL_GEMM=n
// turns it on from n sequences; the towers never take it by themselves.
#include "common.h"
#include "gemm_loop.h"

extern mq_knob mq_xcd_band;
int mq_device_ok();   // runtime.hip

namespace {

constexpr int PG_TILE = 8192;   // one k-step of the panel: 64 tokens x 128 B
constexpr int PG_RING = 65536;  // 8 waves x 4 slots x 2 KiB of weight rows

// ACT: 0 none, 1 erf-GELU (gelu_erf4), 2 QuickGELU
template <int W, int ACT>
__global__ __launch_bounds__(512, 1) void panel_gemm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ Wt, const float* __restrict__ bias,
                                                             const float* __restrict__ colsum, const float2* __restrict__ rowstats, bf16_t* __restrict__ out, int ldc,
                                                             int len, int npass, int band) {
    static_assert(W == 768, "the wave / slot arithmetic is written for 768-wide rows (96 output columns per wave and pass)");
    constexpr int NKT = W / 64;            // k-steps (12)
    constexpr int PANEL = NKT * PG_TILE;
    constexpr int WC = 96;                 // output columns per wave and pass
    constexpr int UPK = 6, UPI = 12;       // units (16 weight rows x 64 k) per k-step and per loop iteration (two k-steps: a multiple of the 4 slots)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const unsigned img = xcd_banded_block(blockIdx.x, gridDim.x, band);
    const int64_t row0 = (int64_t)img * len;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned slot0 = lds0 + PANEL + (unsigned)wave * 8192u;

    // the rows' statistics (the same for every pass; the kernel's oldest loads: every counted wait below covers them)
    float2 ms[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = mt * 16 + l15;
        ms[mt] = rowstats[row0 + (m < len ? m : len - 1)];
    }
    // ---- the panel: 12 tiles x 8 pieces of 8 rows, 12 pieces per wave; rows past the sequence re-read its last row (finite; their outputs are not stored)
    {
        const bf16_t* x0 = x + row0 * W;
#pragma unroll
        for (int i = 0; i < NKT; ++i) {
            const int p = wave * NKT + i;            // 0..95: tile p / 8, piece p % 8
            const int kt = p >> 3, piece = p & 7;
            const int row = piece * 8 + (lane >> 3);
            const int r = row < len ? row : len - 1;
            const int lc = (lane & 7) ^ (row & 7);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(x0 + (int64_t)r * W + kt * 64 + lc * 8),
                                             (__attribute__((address_space(3))) void*)(smem + kt * PG_TILE + piece * 1024), 16, 0, 0);
        }
    }
    // ---- the weight stream: unit (pass, kt, i): rows pass * W + 96 wave + 16 i .. + 15, k = 64 kt .. + 63 -> slot (running unit number) & 3
    const unsigned w_rd0 = (unsigned)(l15 * 128 + ((g ^ (l15 & 7)) << 4)), w_rd1 = (unsigned)(l15 * 128 + (((g + 4) ^ (l15 & 7)) << 4));
    const unsigned w_vo = (unsigned)((lane >> 3) * (W * 2) + (((lane & 7) ^ ((lane >> 3) & 7)) << 4));
    const unsigned w_bytes = (unsigned)npass * W * W * 2;
    auto issue_unit = [&](int pass, int kt, int i, int slot) {
        const unsigned rec = pass < npass ? w_bytes : 0u;   // past the last pass: out of range for every lane — no traffic, the counters still tick
        const unsigned soff = (unsigned)((pass * W + wave * WC + i * 16) * (W * 2) + kt * 128);
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)Wt, 0, rec, 0x00020000);
        dma16(rs, w_vo, soff, slot0 + (unsigned)slot * 2048u);
        dma16(rs, w_vo + 8u * (W * 2), soff, slot0 + (unsigned)slot * 2048u + 1024u);
    };
    issue_unit(0, 0, 0, 0);
    issue_unit(0, 0, 1, 1);
    issue_unit(0, 0, 2, 2);
    issue_unit(0, 0, 3, 3);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // my pieces of the panel have landed (loads retire in order; the 8 weight pieces behind them may be in flight)
    __builtin_amdgcn_s_barrier();                        // ... everybody's have: the only workgroup barrier of the kernel
    asm volatile("" ::: "memory");

    bf16x8 tf[4][2], wf[2][2];
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // unit 0 landed
    wf[0][0] = lds_read16<0>(slot0 + w_rd0);
    wf[0][1] = lds_read16<0>(slot0 + w_rd1);

    for (int pass = 0; pass < npass; ++pass) {
        f32x4 acc[3][4][2];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) { acc[a][mt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[a][mt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        // the panel's first tile (not held across the previous pass's epilogue: 32 registers the epilogue needs; one exposed LDS latency per pass)
        static_for<4>([&](auto mt_tag) {
            constexpr int mt = decltype(mt_tag)::value;
            tf[mt][0] = lds_read16<mt * 2048>(lds0 + w_rd0);
            tf[mt][1] = lds_read16<mt * 2048>(lds0 + w_rd1);
        });
        for (int ktp = 0; ktp < NKT / 2; ++ktp) {
            // the first units behind a pass's epilogue have its 12 stores between their own loads and the loads they wait for (vmcnt counts both, in order)
            const bool behind_stores = pass > 0 && ktp == 0;
            static_for<UPI>([&](auto ii_tag) {
                constexpr int ii = decltype(ii_tag)::value, i = ii % UPK, cur = ii & 1, slot = ii & 3;
                const int kt = 2 * ktp + ii / UPK;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this unit's weight fragments (and, at a k-step's first unit, the panel's) are in registers
                landed(wf[cur][0]); landed(wf[cur][1]);
                if constexpr (i == 0) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) { landed(tf[mt][0]); landed(tf[mt][1]); }
                }
                __builtin_amdgcn_sched_barrier(0);
                constexpr int a = i >> 1, j = i & 1;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    acc[a][mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cur][0], tf[mt][0], acc[a][mt][j], 0, 0, 0);
                    acc[a][mt][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cur][1], tf[mt][1], acc[a][mt][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                // behind the MFMAs' issue: the slot is free (its fragments are in registers), unit + 4 goes into it ...
                {
                    constexpr int i4 = (ii + 4) % UPI;
                    int ktp4 = ktp + (ii + 4) / UPI, pass4 = pass;
                    if (ktp4 == NKT / 2) { ktp4 = 0; ++pass4; }
                    issue_unit(pass4, 2 * ktp4 + i4 / UPK, i4 % UPK, slot);
                }
                // ... units + 2, + 3, + 4 may be in flight: unit + 1 has landed
                if (ii < 3 && behind_stores) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                {
                    constexpr int nslot = (ii + 1) & 3;
                    wf[cur ^ 1][0] = lds_read16<nslot * 2048>(slot0 + w_rd0);
                    wf[cur ^ 1][1] = lds_read16<nslot * 2048>(slot0 + w_rd1);
                }
                if constexpr (i == UPK - 1) {   // the k-step's last unit: the next k-step's panel tile (behind this unit's MFMAs in program order)
                    __builtin_amdgcn_sched_barrier(0);
                    if (kt + 1 < NKT) {
                        const unsigned tb = lds0 + (unsigned)(kt + 1) * PG_TILE;
                        static_for<4>([&](auto mt_tag) {
                            constexpr int mt = decltype(mt_tag)::value;
                            tf[mt][0] = lds_read16<mt * 2048>(tb + w_rd0);
                            tf[mt][1] = lds_read16<mt * 2048>(tb + w_rd1);
                        });
                    }
                }
            });
        }
        // (the first fragments of the next pass's first unit are in flight: retire them here — the compiler takes an asm's outputs as valid once the statement
        // has executed and may move them while the epilogue runs)
        // (the wait carries the epilogue's column base as an operand: whatever the epilogue computes — its addresses first of all — depends on n0 and so cannot be
        // hoisted above the wait into registers an LDS read may still be writing; tests/test_attn_proj_isa.py walks the ISA for exactly that)
        int n0 = pass * W + wave * WC;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(n0)::"memory");
        landed(wf[0][0]); landed(wf[0][1]); landed(wf[1][0]); landed(wf[1][1]);
        // ---- the pass's epilogue (gemm_epilogue.h's LN_APPLY order: rstd * (acc - mean * colsum), + bias, activation, round, 16-byte stores); the next
        // pass's first fragments stay in flight / in registers across it
        f32x4 bias_v[3][2], cs_v[3][2];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bias_v[a][j] = *(const f32x4*)(bias + n0 + a * 32 + j * 16 + 4 * g);
                cs_v[a][j] = *(const f32x4*)(colsum + n0 + a * 32 + j * 16 + 4 * g);
            }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int m = mt * 16 + l15;
            const bool m_ok = m < len;
            const float row_mean = ms[mt].x, row_rstd = ms[mt].y;
            bf16_t* orow = out + (row0 + (m_ok ? m : 0)) * (int64_t)ldc + n0;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                uint2 pk[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x4 v = acc[a][mt][j];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = row_rstd * (v[e] - row_mean * cs_v[a][j][e]);
                    v += bias_v[a][j];
                    if (ACT == 1) v = gelu_erf4(v);
                    if (ACT == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
                    }
                    pk[j].x = pack_bf16x2(v[0], v[1]);
                    pk[j].y = pack_bf16x2(v[2], v[3]);
                }
                const auto r0 = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
                if (m_ok) *(uint4*)(orow + a * 32 + (g & 1) * 16 + (g >> 1) * 8) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (out-of-range) requests
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    landed(wf[0][0]); landed(wf[0][1]); landed(wf[1][0]); landed(wf[1][1]);
}

}  // namespace

// 1 = the shapes mq_panel_gemm_ln takes
extern "C" int mq_panel_gemm_ln_ok(int64_t nseq, int32_t fixed_len, int64_t N, int64_t K) {
    return nseq >= 1 && fixed_len >= 1 && fixed_len <= 64 && K == 768 && N >= 768 && N % 768 == 0 && N <= 768 * 16 && nseq * fixed_len < (1LL << 31);
}

// mq_gemm_bf16_ln for nseq fixed-length sequences of <= 64 rows with K = 768 and N a multiple of 768, one workgroup per sequence (rows = nseq * fixed_len);
// same operands, same output bits.  flags: MQ_EPI_BIAS [| MQ_EPI_GELU | MQ_EPI_QUICKGELU]; d_out bf16 [rows, ldc].
extern "C" int mq_panel_gemm_ln(const void* d_x, const void* d_W, const float* d_bias, const float* d_colsum, const float* d_rowstats, void* d_out, int64_t ldc,
                                int64_t nseq, int32_t fixed_len, int64_t N, int64_t K, int flags, void* stream) {
    MQ_CHECK_ARG(d_x && d_W && d_bias && d_colsum && d_rowstats && d_out, "mq_panel_gemm_ln: null operand");
    MQ_CHECK_ARG(mq_panel_gemm_ln_ok(nseq, fixed_len, N, K), "mq_panel_gemm_ln: takes 1..64-row sequences, K = 768, N a multiple of 768 (nseq=%ld len=%d N=%ld K=%ld)",
                 (long)nseq, fixed_len, (long)N, (long)K);
    MQ_CHECK_ARG(ldc >= N && ldc % 8 == 0 && (((uintptr_t)d_x | (uintptr_t)d_W | (uintptr_t)d_out | (uintptr_t)d_bias | (uintptr_t)d_colsum) & 15) == 0,
                 "mq_panel_gemm_ln: operands / rows must be 16-byte aligned");
    MQ_CHECK_ARG((flags & ~(MQ_EPI_GELU | MQ_EPI_QUICKGELU)) == MQ_EPI_BIAS && (flags & (MQ_EPI_GELU | MQ_EPI_QUICKGELU)) != (MQ_EPI_GELU | MQ_EPI_QUICKGELU),
                 "mq_panel_gemm_ln: flags must be MQ_EPI_BIAS [| MQ_EPI_GELU | MQ_EPI_QUICKGELU]");
    MQ_TRY(mq_device_ok());
    hipStream_t s = (hipStream_t)stream;
    constexpr int LDS = 12 * PG_TILE + PG_RING;   // 160 KiB
    const int npass = (int)(N / 768);
    const int band = (mq_xcd_band && nseq >= 256) ? 1 : 0;
    MqProfScope prof(0, s, 2.0 * (double)nseq * fixed_len * (double)N * (double)K);
    auto launch = [&](auto kern, std::atomic<uint64_t>& attr_done) -> int {
        if (hipError_t e = mq_ensure_dyn_lds((const void*)kern, LDS, attr_done); e != hipSuccess) {
            mq_set_error("mq_panel_gemm_ln: hipFuncSetAttribute: %s", hipGetErrorString(e));
            return MQ_ERR_HIP;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)nseq), dim3(512), LDS, s, (const bf16_t*)d_x, (const bf16_t*)d_W, d_bias, d_colsum, (const float2*)d_rowstats, (bf16_t*)d_out,
                           (int)ldc, (int)fixed_len, npass, band);
        return MQ_OK;
    };
    static std::atomic<uint64_t> done0{0}, done1{0}, done2{0};
    int rc;
    if (flags & MQ_EPI_GELU) rc = launch(panel_gemm_kernel<768, 1>, done1);
    else if (flags & MQ_EPI_QUICKGELU) rc = launch(panel_gemm_kernel<768, 2>, done2);
    else rc = launch(panel_gemm_kernel<768, 0>, done0);
    if (rc != MQ_OK) return rc;
    MQ_CHECK_LAUNCH("mq_panel_gemm_ln");
    return MQ_OK;
}
