// Multi-head attention for short sequences (T = 50 / 77 / 257 / <= 512), d_head = 64 (K4) — or up to 128: template parameter
// HD = 128 makes the LDS rows 256 B (16 chunk slots swizzled by key & 15) and HS = 96 / 112 / 128 is the head stride in global
// memory = the dims actually computed (the 80 / 88 / 104-wide heads of ViT-H / g / bigG are zero-padded to 96 / 96 / 112 at load).
//
// One workgroup (4 or 8 wave64s) per (sequence, head).  The whole K [len,64] and V [len,64] of that
// (sequence, head) are staged ONCE in LDS by LDS-DMA (sequences longer than the LDS holds — 640 keys at 128-byte rows, 320 at
// 256-byte rows — stream through it in pieces, re-staged for every round of query blocks) (global_load_lds: every 1-KiB piece is in flight at once, no
// register round trip; 160 KB/CU makes this possible up to 512 keys: 64 KB + 64 KB), both row-major with the 16-B
// chunks XOR-swizzled by (key & 7).  Each wave then owns 16-query blocks and runs a flash-style online softmax over
// 64-key tiles with both GEMMs on v_mfma_f32_16x16x32_bf16; the V^T operand of the second GEMM is produced by
// ds_read_b64_tr_b16 (the hardware 4x4 transpose read) straight from the row-major V — no transposed copy exists:
//
//   S^T = K . Q^T  (operands swapped so every lane's 16 scores belong to ONE query: the row max /
//                   row sum need only two xor-shuffles, no LDS round trip)
//   O^T = V^T . P^T  (the P^T fragment a lane needs as MFMA B-operand is exactly the set of
//                   probabilities it already holds; the k-slot <-> key permutation is applied
//                   identically to the V^T A-operand, which is legal because the contraction is
//                   permutation-invariant)
//
// Sequences are packed (cu_seqlens) or fixed-length; there is no key-padding mask: padded
// tokens are simply not rows.  MASK_CAUSAL implements the CLIP text tower's mask.
#include <type_traits>
#include "common.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));

namespace {

// OUT_FP8: the output is written as e4m3 codes = value / *out_scale (static per-tensor scale of the fp8 path, K13) and
// max|value| is folded into *amax when it is non-null (calibration).
// NW = wave64s per workgroup: 4, or 8 for sequences of more than 128 tokens (>= 9 sixteen-query blocks): their K / V image lets
// only one or two workgroups share a CU, so the extra waves per SIMD have to come from inside the workgroup.
// Register budget: the 64-wide-head kernels are asked to fit 5 waves per SIMD (92 VGPRs, no spill; left alone the allocator takes 100 ->
// 4 waves).  These short-sequence launches (T = 50 / 77: one key tile per wave) are a latency chain — K / V DMA, QK^T, softmax, PV, store —
// so a fifth resident workgroup per CU is the cover: attention -8 % on ViT-B/32, -1..2 % on the text towers
// (profiles/r02o_attention_occupancy_ab.txt).  The 128-byte-row kernels would spill at that budget and keep the default.
#ifndef MQ_ATTN_WAVES_PER_EU
#define MQ_ATTN_WAVES_PER_EU 5   // 0: no request (A/B builds, tools/probes/build_attn_occupancy.sh)
#endif
#if MQ_ATTN_WAVES_PER_EU > 0
#define MQ_ATTN_OCC __attribute__((amdgpu_waves_per_eu(HD == 64 ? MQ_ATTN_WAVES_PER_EU : 1, 8)))
#else
#define MQ_ATTN_OCC
#endif
// BIAS: an additive relative-position bias on the scores (MPNet: T5-style bucketed bias shared by all layers).  rel_bias is fp32
// [heads][2 * rel_span - 1], entry (h, d + rel_span - 1) = bias of key - query == d, ALREADY divided by the softmax scale (the kernel adds
// it to the raw q.k sums and scales everything once); rel_span >= the longest sequence.
template <int MASK, bool OUT_FP8, int HD, int HS, int NW, bool BIAS = false>
__global__ __launch_bounds__(NW * 64) MQ_ATTN_OCC void attention_kernel(
    const bf16_t* __restrict__ qkv, void* __restrict__ out_v, const int32_t* __restrict__ cu,
    int fixed_len, int W, int heads, int kpad, float scale_log2e, const float* __restrict__ out_scale, float* amax_out,
    const float* __restrict__ rel_bias = nullptr, int rel_span = 0, int band = 0, float2* __restrict__ row_part = nullptr, int64_t part_ld = 0) {
    bf16_t* out = (bf16_t*)out_v;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(HS <= HD && HS % 16 == 0 && (HD == 64 || HD == 128), "head stride: multiple of 16, at most the LDS row");
    constexpr int RB = HD * 2;       // bytes per K / V row in LDS (128 or 256)
    constexpr int NC = HD / 8;       // 16-byte chunk slots per LDS row (8 or 16)
    constexpr int NCS = HS / 8;      // chunks a head really has in global memory (slots past them are never consumed)
    constexpr int NKK = (HS + 31) / 32;  // 32-deep MFMA k-chunks of a Q.K dot product (2 .. 4)
    constexpr int NDT = HS / 16;     // 16-wide output-dim tiles (4 .. 8)
    constexpr int RPP = 1024 / RB;   // rows per 1-KiB LDS-DMA piece (8 or 4)
    char* sK = smem;                      // [kpad][RB], 16-B chunks XOR-swizzled by (key & (NC-1))
    char* sV = smem + (size_t)kpad * RB;  // [kpad][RB], same image

    const unsigned vblk = xcd_banded_block(blockIdx.x, gridDim.x, band);   // sequences (rows of qkv / out) in the GEMMs' XCD bands
    const int seq = vblk / heads, h = vblk - seq * heads;
    int row0, len;
    if (fixed_len > 0) { row0 = seq * fixed_len; len = fixed_len; }
    else { row0 = cu[seq]; len = cu[seq + 1] - row0; }
    if (len <= 0) return;
    const int ld = 3 * W;
    const bf16_t* qb = qkv + (int64_t)row0 * ld + h * HS;
    const bf16_t* kb = qb + W;
    const bf16_t* vb = qb + 2 * W;
    const int nkt = (len + 63) >> 6;
    const int kp = nkt << 6;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // ---- stage K and V: RPP rows x RB bytes per LDS-DMA; lane -> (row = RPP*piece + lane/NC, physical chunk = lane%NC) fetches
    // the logical chunk that lives there.  Rows past the sequence re-read its last row (finite values; their scores are
    // masked and their probabilities are exactly 0).  The LDS holds kpad keys: the whole sequence when it fits (nchunks == 1,
    // staged once), else the sequence goes through it in nchunks pieces of kpad keys for every round of query blocks.
    const int nchunks = (kp + kpad - 1) / kpad;
    auto stage = [&](int c) {
        const int base = c * kpad;
        const int rows_here = kp - base < kpad ? kp - base : kpad;
        const int np8 = rows_here / RPP;
        const int srow = lane / NC, pchunk = lane % NC;
        for (int p = wave; p < 2 * np8; p += NW) {
            const bool is_v = p >= np8;
            const int piece = is_v ? p - np8 : p;
            const int row = piece * RPP + srow;        // LDS row; the key is base + row (base is a multiple of 64: same swizzle)
            const int key = base + row < len ? base + row : len - 1;
            int lc = pchunk ^ (row & (NC - 1));
            if (HS < HD) lc = lc < NCS ? lc : 0;  // slots of chunks the head does not have: any finite filler (they meet zero Q dims)
            const bf16_t* src = (is_v ? vb : kb) + (int64_t)key * ld + (lc << 3);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)((is_v ? sV : sK) + piece * 1024), 16, 0, 0);
        }
    };
    if (nchunks == 1) stage(0);
    const int l15 = lane & 15, g = lane >> 4;
    const int nqb = (len + 15) >> 4;
    float amax_local = 0.f;
    // Q fragment of query row qr, k-chunk kk: dims 32kk + 8g .. +7; dims past the head (HS = 112: the last 16) are zeros, so
    // whatever sits in the matching K slots contributes nothing
    auto load_q = [&](int qr, int kk) -> bf16x8 {
        const int col = 8 * g + 32 * kk;
        if (HS % 32 != 0 && col >= HS) return bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        return *(const bf16x8*)(qb + (int64_t)qr * ld + col);
    };
    // first Q fragments are fetched while the K/V DMA is in flight
    bf16x8 qn[NKK];
    {
        const int q0 = wave * 16 + l15;
        const int qr = q0 < len ? q0 : len - 1;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) qn[kk] = load_q(qr, kk);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // per-lane constants of the V^T transpose reads: source lane (g, l15) fetches 4 consecutive d of key 4g + l15/4
    const int vkey = 4 * g + (l15 >> 2);
    const int vcol = (l15 & 3) >> 1, vhalf = (l15 & 1) << 3;

    const int nrounds = (nqb + NW - 1) / NW;
    for (int rnd = 0; rnd < nrounds; ++rnd) {
        const int qblk = rnd * NW + wave;
        const bool active = qblk < nqb;          // wave-uniform; with nchunks > 1 an idle wave still stages and keeps the barriers
        if (nchunks == 1 && !active) break;
        const int q = qblk * 16 + l15;           // this lane's query (B-operand column / output row)
        bf16x8 qf[NKK];
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) qf[kk] = qn[kk];
        if (active && qblk + NW < nqb) {         // next block's Q streams in behind this block's math
            const int q2 = q + 16 * NW;
            const int qr = q2 < len ? q2 : len - 1;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) qn[kk] = load_q(qr, kk);
        }

        float m_run = -1e30f, l_run = 0.f;
        f32x4 o[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

        int kt_end = nkt;
        if (MASK != MQ_MASK_NONE) {
            const int last = (qblk * 16 + 15) >> 6;  // last key tile any query of this block may see
            kt_end = last + 1 < nkt ? last + 1 : nkt;
        }
        for (int c = 0; c < nchunks; ++c) {
        if (nchunks > 1) {   // (re)fill the LDS with keys [c * kpad, (c + 1) * kpad)
            __syncthreads();
            stage(c);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        const int kt0 = c * (kpad >> 6);
        const int kt1 = !active ? kt0 : (kt_end < kt0 + (kpad >> 6) ? kt_end : kt0 + (kpad >> 6));
        const int cb = c * kpad;                         // LDS row of global key `key` is key - cb (cb is a multiple of 64)
        for (int kt = kt0; kt < kt1; ++kt) {
            // ---- S^T tile: keys 64kt + 16t + 4g + r for this lane's query -----------------------
            f32x4 sc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                sc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int key = kt * 64 + t * 16 + l15;  // A-operand row
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    const bf16x8 kf = *(const bf16x8*)(sK + (key - cb) * RB + (((g + 4 * kk) ^ (key & (NC - 1))) << 4));
                    sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], sc[t], 0, 0, 0);
                }
            }
            if (BIAS) {   // (rows of padded query lanes / key columns past the sequence are clamped: their scores are masked or never stored)
                const float* rb = rel_bias + (int64_t)h * (2 * rel_span - 1) + (rel_span - 1) - (q < len ? q : len - 1);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt * 64 + t * 16 + g * 4 + r;
                        sc[t][r] += rb[key < len ? key : len - 1];
                    }
            }
            // masking is needed only on the ragged last tile and (causal) on tiles that reach past the block's first query:
            // a wave-uniform test, so interior tiles skip the per-element compares / selects
            bool need_mask = (kt == nkt - 1) && (len & 63);
            if (MASK != MQ_MASK_NONE) need_mask = need_mask || (kt * 64 + 63 > qblk * 16);
            if (need_mask) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt * 64 + t * 16 + g * 4 + r;
                        bool valid = key < len;
                        if (MASK != MQ_MASK_NONE) valid = valid && (key <= q);
                        // CoCa's class-token row (the sequence's LAST row) does not see its own key (open_clip build_cls_mask, see marqo_hip.h)
                        if (MASK == MQ_MASK_CAUSAL_CLS) valid = valid && !(q == len - 1 && key == q && len > 1);
                        sc[t][r] = valid ? sc[t][r] : -INFINITY;
                    }
            }
            // running max on the RAW scores (the softmax scale is positive); scale and shift fold into one FMA per element
            float mx = fmaxf(fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3])), fmaxf(fmaxf(sc[1][0], sc[1][1]), fmaxf(sc[1][2], sc[1][3])));
            mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(sc[2][0], sc[2][1]), fmaxf(sc[2][2], sc[2][3])), fmaxf(fmaxf(sc[3][0], sc[3][1]), fmaxf(sc[3][2], sc[3][3]))));
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);   // finite: every query sees at least key 0
            const float neg_mc = -m_new * scale_log2e;
            float psum = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sc[t][r] = __builtin_amdgcn_exp2f(fmaf(sc[t][r], scale_log2e, neg_mc));
                    psum += sc[t][r];
                }
            if (__any(m_new != m_run)) {  // rescale only when some query's running max moved (rare after the first tiles)
                const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) o[dt] *= alpha;
            }
            l_run += psum;
            m_run = m_new;

            // ---- O^T += V^T . P^T over the tile's two 32-key halves --------------------------------
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                union { uint32_t w[4]; bf16x8 v; } pf;
                pf.w[0] = pack_bf16x2(sc[2 * u][0], sc[2 * u][1]);
                pf.w[1] = pack_bf16x2(sc[2 * u][2], sc[2 * u][3]);
                pf.w[2] = pack_bf16x2(sc[2 * u + 1][0], sc[2 * u + 1][1]);
                pf.w[3] = pack_bf16x2(sc[2 * u + 1][2], sc[2 * u + 1][3]);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    // lane (d = dt*16 + l15, g) needs V[keys 16*(2u) + 4g .. +3][d] and V[keys 16*(2u+1) + 4g .. +3][d]:
                    // ds_read_b64_tr_b16 hands output lane 4r+c of a 16-lane block element c of the 8 bytes fetched by
                    // lanes r, r+4, r+8, r+12 of that block, so source lane (g, l15) fetches V[key0 + 4g + l15/4][4*(l15%4) .. +3]
                    union { s16x4 t[2]; bf16x8 v; } vf;
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int key = kt * 64 + (2 * u + tt) * 16 + vkey;
                        const char* vp = sV + (key - cb) * RB + ((((dt << 1) | vcol) ^ (key & (NC - 1))) << 4) + vhalf;
                        vf.t[tt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)vp);
                    }
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf.v, o[dt], 0, 0, 0);
                }
            }
        }
        }  // chunks
        if (!active) continue;
        float l_tot = l_run + __shfl_xor(l_run, 16, 64);
        l_tot += __shfl_xor(l_tot, 32, 64);
        const float inv = 1.0f / l_tot;
        if (q < len) {
            if (OUT_FP8) {
                const float qs = 1.0f / out_scale[0];
                uint8_t* orow8 = (uint8_t*)out_v + (int64_t)(row0 + q) * W + h * HS + 4 * g;
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = o[dt][e] * inv;
                        amax_local = fmaxf(amax_local, fabsf(v[e]));
                        v[e] = fminf(fmaxf(v[e] * qs, -448.f), 448.f);
                    }
                    int w = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
                    w = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], w, true);
                    *(int*)(orow8 + dt * 16) = w;
                }
            } else {
                bf16_t* orow = out + (int64_t)(row0 + q) * W + h * HS + 4 * g;
                float st1 = 0.f, st2 = 0.f;   // row_part: (sum, sum of squares) of this row's ROUNDED values over the head's columns
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    uint2 p;
                    p.x = pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv);
                    p.y = pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv);
                    *(uint2*)(orow + dt * 16) = p;
                    if (row_part) {
                        const float e0 = __uint_as_float(p.x << 16), e1 = __uint_as_float(p.x & 0xffff0000u), e2 = __uint_as_float(p.y << 16),
                                    e3 = __uint_as_float(p.y & 0xffff0000u);
                        st1 += (e0 + e1) + (e2 + e3);
                        st2 += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
                    }
                }
                // mq_attention_stats: one (sum, sum of squares) per (row, head) — the slot layout of the GEMMs' MQ_EPI_ROW_STATS, nslots = heads — for the
                // LayerNorm that follows the attention (EVA02 attn.norm), folded into the out-projection.  The row's 4 lanes (g = 0..3) share q: all four
                // are inside this branch together; fixed order, one writer
                if (row_part) {
                    auto fold = [](float v) {   // (VALU swaps instead of ds_bpermute, as gemm_epilogue.h)
                        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
                        const float w = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                        const auto u = __builtin_amdgcn_permlane32_swap(__float_as_uint(w), __float_as_uint(w), false, false);
                        return __uint_as_float(u[0]) + __uint_as_float(u[1]);
                    };
                    st1 = fold(st1);
                    st2 = fold(st2);
                    if (g == 0) row_part[(int64_t)h * part_ld + (row0 + q)] = make_float2(st1, st2);   // slot-major [heads][rows], as GemmLn::partials
                }
            }
        }
    }
    if (OUT_FP8 && amax_out) {
        amax_local = wave_max(amax_local);
        if (lane == 0) atomicMax((int*)amax_out, __float_as_int(amax_local));
    }
}

// (round 3's attention_short_kernel — several 50-token items per workgroup — measured slower, profiles/r03ad_attn_items_ab.txt, and left the library in round 4)

}  // namespace

mq_knob mq_attention_waves{0};  // mq_tune("attn_waves", 0 = auto / 4 / 8 / 5 = five waves for 65..80-token sequences, else auto): A/B knob

static int attention_impl(const void* d_qkv, void* d_out, const int32_t* d_cu_seqlens, int64_t nseq,
                          int32_t fixed_len, int32_t max_len, int32_t W, int32_t heads, int32_t mask,
                          int32_t out_fp8, const float* d_out_scale, float* d_amax, const float* d_rel_bias, int32_t rel_span, void* stream,
                          float* d_row_part = nullptr, int64_t part_ld = 0) {
    MQ_CHECK_ARG(d_qkv && d_out, "mq_attention: null pointer");
    MQ_CHECK_ARG(!d_row_part || (!out_fp8 && !d_rel_bias), "mq_attention_stats: bf16 output, no relative-position bias");
    MQ_CHECK_ARG(heads >= 1 && W % heads == 0, "mq_attention: W=%d is not a multiple of heads=%d", W, heads);
    const int hs = W / heads;             // head stride in memory = dims computed
    MQ_CHECK_ARG(hs == 64 || hs == 96 || hs == 112 || hs == 128, "mq_attention: head dim must be 64, 96, 112 or 128 (W=%d heads=%d)", W, heads);
    const int hd = hs == 64 ? 64 : 128;   // LDS row width
    MQ_CHECK_ARG(fixed_len > 0 || d_cu_seqlens, "mq_attention: need fixed_len or cu_seqlens");
    MQ_CHECK_ARG(mask == MQ_MASK_NONE || mask == MQ_MASK_CAUSAL || mask == MQ_MASK_CAUSAL_CLS, "mq_attention: bad mask %d", mask);
    MQ_CHECK_ARG(mask != MQ_MASK_CAUSAL_CLS || !out_fp8, "mq_attention: MQ_MASK_CAUSAL_CLS runs with bf16 output only");
    if (nseq <= 0) return MQ_OK;
    const int maxl = fixed_len > 0 ? fixed_len : max_len;
    MQ_CHECK_ARG(maxl >= 1 && maxl <= 8192, "mq_attention: max sequence length %d unsupported (1..8192)", maxl);
    MQ_CHECK_ARG(!d_rel_bias || (hs == 64 && mask == MQ_MASK_NONE && !out_fp8 && rel_span >= maxl),
                 "mq_attention_bias: the relative-position bias runs with 64-wide heads, no mask, bf16 output and rel_span (%d) >= the longest "
                 "sequence (%d)", rel_span, maxl);
    MQ_CHECK_ARG(nseq * heads < (1LL << 31), "mq_attention: grid too large");
    // K + V rows of hd bf16 each live in the CU's 160 KiB of LDS: whole sequences up to 640 keys (hd 64) / 320 keys (hd 128),
    // longer ones stream through it in pieces of that many keys (ViT-H-14-378: 730 tokens, ViT-B-16-SigLIP-512: 1024)
    const int cap = hd == 64 ? 640 : 320;
    const int need = ((maxl + 63) / 64) * 64;
    const int kpad = need < cap ? need : cap;
    const size_t lds = (size_t)kpad * hd * 4;
    hipStream_t s = (hipStream_t)stream;
    // 8 waves from 9 query blocks up (measured, profiles/r01f_attention_waves_ab.txt: -25..-33 % at 257 / 512 / 577 tokens, where a
    // workgroup's K / V image leaves room for one or two workgroups per CU; +6..+25 % at 77 / 50 tokens, whose 5 / 4 blocks leave
    // the extra waves idle)
    // (65..80 tokens are FIVE query blocks — the 77-token CLIP / BERT rows; five-wave workgroups, one block per wave instead of a second round
    // with three waves idle, measured neutral to slower: CLIP text -0.7 %, BERT-base +3.6 % attention time, profiles/r02p_attention_five_waves_ab.txt;
    // kept behind mq_tune("attn_waves", 5) with its bit-identity test)
    // (round 6: NINE waves where they save a whole round of query blocks — 257 tokens = 17 blocks: 3 rounds of 8 waves, 2 of 9; 576 tokens: 5 -> 4 — were
    // built and measured: +10-11 % attention time on ViT-L/14, EVA02-L-14 and SigLIP-L-16-384, profiles/r06l_attention_nine_waves_ab.txt — an odd wave
    // count sits 3 + 2 + 2 + 2 on the SIMDs; not kept)
    const bool five = mq_attention_waves == 5 && maxl > 64 && maxl <= 80 && hs == 64 && !d_rel_bias;
    const int knob_waves = mq_attention_waves;
    const int nw = five ? 5 : (knob_waves == 4 || knob_waves == 8 ? knob_waves : (maxl > 128 ? 8 : 4));
    const float scale_log2e = 1.44269504088896340736f / sqrtf((float)hs);  // 1/sqrt(head dim) * log2(e)
    MqProfScope prof(2, s);
    auto launch = [&](auto kern) -> int {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { mq_set_error("mq_attention: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MQ_ERR_HIP; }
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)(nseq * heads)), dim3(nw * 64), lds, s, (const bf16_t*)d_qkv,
                           d_out, d_cu_seqlens, (int)fixed_len, (int)W, (int)heads, kpad, scale_log2e, d_out_scale, d_amax, d_rel_bias, (int)rel_span,
                           (mq_xcd_band && nseq * heads >= 2048) ? 1 : 0, (float2*)d_row_part, part_ld);
        return MQ_OK;
    };
    if (d_rel_bias) {
        const int rcb = nw == 8 ? launch(attention_kernel<MQ_MASK_NONE, false, 64, 64, 8, true>) : launch(attention_kernel<MQ_MASK_NONE, false, 64, 64, 4, true>);
        if (rcb != MQ_OK) return rcb;
        MQ_CHECK_LAUNCH("mq_attention_bias");
        return MQ_OK;
    }
    MQ_CHECK_ARG(!out_fp8 || d_out_scale, "mq_attention: fp8 output needs an out_scale");
    int rc;
    auto pick_nw = [&](auto hd_tag, auto hs_tag, auto nw_tag) -> int {
        constexpr int HD_ = decltype(hd_tag)::value, HS_ = decltype(hs_tag)::value, NW_ = decltype(nw_tag)::value;
        if (out_fp8) return (mask == MQ_MASK_CAUSAL) ? launch(attention_kernel<MQ_MASK_CAUSAL, true, HD_, HS_, NW_>) : launch(attention_kernel<MQ_MASK_NONE, true, HD_, HS_, NW_>);
        if (mask == MQ_MASK_CAUSAL_CLS) return launch(attention_kernel<MQ_MASK_CAUSAL_CLS, false, HD_, HS_, NW_>);
        return (mask == MQ_MASK_CAUSAL) ? launch(attention_kernel<MQ_MASK_CAUSAL, false, HD_, HS_, NW_>) : launch(attention_kernel<MQ_MASK_NONE, false, HD_, HS_, NW_>);
    };
    auto pick = [&](auto hd_tag, auto hs_tag) -> int {
        if constexpr (decltype(hd_tag)::value == 64)
            if (nw == 5) return pick_nw(hd_tag, hs_tag, std::integral_constant<int, 5>{});
        return nw == 8 ? pick_nw(hd_tag, hs_tag, std::integral_constant<int, 8>{}) : pick_nw(hd_tag, hs_tag, std::integral_constant<int, 4>{});
    };
    using std::integral_constant;
    switch (hs) {
        case 64: rc = pick(integral_constant<int, 64>{}, integral_constant<int, 64>{}); break;
        case 96: rc = pick(integral_constant<int, 128>{}, integral_constant<int, 96>{}); break;
        case 112: rc = pick(integral_constant<int, 128>{}, integral_constant<int, 112>{}); break;
        default: rc = pick(integral_constant<int, 128>{}, integral_constant<int, 128>{}); break;
    }
    if (rc != MQ_OK) return rc;
    MQ_CHECK_LAUNCH("mq_attention");
    return MQ_OK;
}

extern "C" int mq_attention_ex(const void* d_qkv, void* d_out, const int32_t* d_cu_seqlens, int64_t nseq,
                               int32_t fixed_len, int32_t max_len, int32_t W, int32_t heads, int32_t mask,
                               int32_t out_fp8, const float* d_out_scale, float* d_amax, void* stream) {
    return attention_impl(d_qkv, d_out, d_cu_seqlens, nseq, fixed_len, max_len, W, heads, mask, out_fp8, d_out_scale, d_amax, nullptr, 0, stream);
}

extern "C" int mq_attention_bias(const void* d_qkv, void* d_out, const int32_t* d_cu_seqlens, int64_t nseq, int32_t fixed_len,
                                 int32_t max_len, int32_t W, int32_t heads, const float* d_rel_bias, int32_t rel_span, void* stream) {
    MQ_CHECK_ARG(d_rel_bias, "mq_attention_bias: null bias table");
    return attention_impl(d_qkv, d_out, d_cu_seqlens, nseq, fixed_len, max_len, W, heads, MQ_MASK_NONE, 0, nullptr, nullptr, d_rel_bias, rel_span, stream);
}

extern "C" int mq_attention(const void* d_qkv, void* d_out, const int32_t* d_cu_seqlens, int64_t nseq, int32_t fixed_len,
                            int32_t max_len, int32_t W, int32_t heads, int32_t mask, void* stream) {
    return mq_attention_ex(d_qkv, d_out, d_cu_seqlens, nseq, fixed_len, max_len, W, heads, mask, 0, nullptr, nullptr, stream);
}

// mq_attention that also leaves (sum, sum of squares) of every output row's ROUNDED values per head behind: d_row_part fp32 [heads][rows][2] (slot-major, `rows` =
// the token rows of the whole call) — the partials mq_row_stats_finalize(nslots = heads, rows, W) turns into the (mean, rstd) of a LayerNorm over the attention
// output (EVA02 attn.norm, folded into the out-projection: mq_gemm_bf16_lnrs)
extern "C" int mq_attention_stats(const void* d_qkv, void* d_out, const int32_t* d_cu_seqlens, int64_t nseq, int32_t fixed_len, int32_t max_len, int32_t W,
                                  int32_t heads, int32_t mask, float* d_row_part, int64_t rows, void* stream) {
    MQ_CHECK_ARG(d_row_part && rows >= 1 && (fixed_len <= 0 || rows == nseq * fixed_len), "mq_attention_stats: null partials / rows does not match the sequences");
    return attention_impl(d_qkv, d_out, d_cu_seqlens, nseq, fixed_len, max_len, W, heads, mask, 0, nullptr, nullptr, nullptr, 0, stream, d_row_part, rows);
}
