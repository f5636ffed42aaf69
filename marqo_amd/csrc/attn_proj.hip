// Attention + out-projection + residual + LayerNorm statistics in ONE launch, for the short fixed-length sequences of the ViT-B/32 image tower
// (T = 50 tokens, 12 heads of 64, W = 768: BASELINE configs[1], the headline) — K4 + K5(out) + the statistics finalise of SURVEY.md §8a.
// Reference call site: /root/reference/src/marqo/core/inference/embedding_models/open_clip_model.py:249-266 (encode_image -> the third-party
// model's residual attention block: x = x + out_proj(attention(qkv))).
//
// Why a kernel of its own: at 256 images the three launches it replaces are each bound by something that is not their arithmetic —
// the attention kernel by the 59 MB of QKV it reads and the 20 MB it writes (17 us), the out-projection (15 GF, one round of 480 tiles whose
// workgroups all run prologue, k-loop and epilogue in lock-step: 25-27 us, 0.25 of the matrix peak) and the 4.6 us finalise of its row statistics by
// launch latency.  Here ONE workgroup (8 wave64s) owns ONE image:
//   phase 1  the image's 12 heads, two at a time (8 waves = 2 heads x 4 sixteen-query blocks): K / V of the NEXT pair stream into LDS by LDS-DMA while
//            this pair computes (S^T = K Q^T and O^T = V^T P^T on v_mfma_f32_16x16x32_bf16, V^T by ds_read_b64_tr_b16: the arithmetic of attention.hip,
//            same operations in the same order -> the same bits); O never leaves the CU: it is written as bf16 into an LDS panel laid out as the GEMM's
//            A operand — one [64 tokens][128 B] tile per head = per 64-deep k-step, 16-byte chunks XOR-swizzled by (token & 7);
//   phase 2  y[64, 768] = O[64, 768] @ Wo^T with the panel resident and Wo streamed: every wave owns 96 output columns (32 in each 256-column third) and
//            streams ITS weight rows through a PRIVATE ring of four 2-KiB LDS slots (16 rows x 64 k, by LDS-DMA through a buffer descriptor), three units in
//            flight — no workgroup barrier anywhere in the GEMM, only counted vmcnt waits; the k order (head by head, two 32-deep halves) is the tiled
//            GEMM's, so the accumulators carry the same bits as gemm_nt_kernel's;
//   epilogue x = bf16((acc + bias) + x) in place (the tiled GEMM's order of operations: bit-identical rows), and because the workgroup holds COMPLETE rows
//            the (mean, rstd) of the LayerNorm behind it are finished here instead of 12 partial slots per row in memory + a finalise launch — summed in
//            the partial-sum path's own order (per 64-column slot, lane chains handed between wave pairs through LDS), so the statistics carry its bits too.
// The price: 64-row MFMA tiles for 50 tokens (78 % useful) and the whole of Wo (1.2 MB) through every CU's vector-memory path once per image.
#include "common.h"
#include "gemm_loop.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

extern mq_knob mq_xcd_band;
extern mq_knob mq_ln_prefetch;   // rowops.hip
int mq_device_ok();   // runtime.hip

namespace {

template <int OFF>
__device__ __forceinline__ s16x4 lds_read_tr16(unsigned addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
// a 16-byte global load the compiler does not track (it would answer a tracked one with vmcnt(0) at the first use — draining the LDS-DMA of the next head
// pair with it); the caller waits by hand and passes the registers through landed()
template <int OFF>
__device__ __forceinline__ bf16x8 gload16(const void* p) {
    i32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(v) : "v"(p), "n"(OFF));
    return __builtin_bit_cast(bf16x8, v);
}
// weight lines the launch touches for the GEMMs behind it (one dword per 128-byte line, values unused): a, b = two ranges, na / nb lines
struct ApPrefetch { const unsigned* a; const unsigned* b; unsigned na, nb; };
__device__ __forceinline__ void lds_write8(unsigned addr, u32x2 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }

// sum / max over the 4 lanes (g = 0..3) that share l15, with the VALU swap instructions (as gemm_epilogue.h): rows (0,1) and (2,3) of 16 lanes, then the halves
__device__ __forceinline__ float fold_sum(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float w = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(w), __float_as_uint(w), false, false);
    return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
__device__ __forceinline__ float fold_max(float v) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float w = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(w), __float_as_uint(w), false, false);
    return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}

constexpr int AP_TILE = 8192;            // one head's O tile in the panel: 64 tokens x 128 B
constexpr int AP_KV = 32768;             // K / V image of a head pair: 2 x (64 keys x 128 B K + the same of V)
constexpr int AP_RING = 65536;           // phase 1: two K / V images; phase 2: 8 waves x 4 slots x 2 KiB of weight rows

template <int W>
__global__ __launch_bounds__(512, 1) void attn_proj_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ Wo, const float* __restrict__ bias,
                                                            bf16_t* x, float2* __restrict__ rowstats, int len, float scale_log2e, float inv_w, float eps, int band, ApPrefetch pf, int dbg) {
    static_assert(W % 256 == 0 && (W / 64) % 2 == 0, "whole 256-column thirds, heads in pairs");
    constexpr int NHEAD = W / 64, NG = NHEAD / 2, NT3 = W / 256;
    constexpr int PANEL = NHEAD * AP_TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const unsigned img = xcd_banded_block(blockIdx.x, gridDim.x, band);
    const int64_t row0 = (int64_t)img * len;
    constexpr int ld = 3 * W;
    const bf16_t* qkv0 = qkv + row0 * ld;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned ring = lds0 + PANEL;

    // =============================== phase 1: attention, a pair of heads per round =============================================================
    const int hl = wave >> 2, qblk = wave & 3;      // this wave: head 2 * grp + hl, queries 16 * qblk .. + 15
    const bool active = qblk * 16 < len;            // wave-uniform
    auto stage = [&](int grp) {                     // K / V of heads 2 grp, 2 grp + 1 -> image grp & 1: 32 pieces of 8 rows, 4 per wave
        char* buf = smem + PANEL + (grp & 1) * AP_KV;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = wave * 4 + i;
            const int ph = p >> 4, is_v = (p >> 3) & 1, piece = p & 7;
            const int row = piece * 8 + (lane >> 3);
            const int key = row < len ? row : len - 1;      // rows past the sequence re-read its last row (finite; masked / zero probability)
            const int lc = (lane & 7) ^ (row & 7);
            const bf16_t* src = qkv0 + (int64_t)key * ld + (is_v ? 2 * W : W) + (2 * grp + ph) * 64 + lc * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(buf + ph * 16384 + is_v * 8192 + piece * 1024), 16, 0, 0);
        }
    };
    const int qi = qblk * 16 + l15;
    const int64_t q_off = (int64_t)(qi < len ? qi : len - 1) * ld + hl * 64 + 8 * g;
    auto load_q = [&](int grp, bf16x8 (&q)[2]) {
        const bf16_t* qb = qkv0 + q_off + grp * 128;
        q[0] = gload16<0>(qb);
        q[1] = gload16<64>(qb);
    };
    // fixed per-lane LDS offsets inside a head's K / V image
    const unsigned k_sw0 = (unsigned)(l15 * 128 + ((g ^ (l15 & 7)) << 4)), k_sw1 = (unsigned)(l15 * 128 + (((g + 4) ^ (l15 & 7)) << 4));
    const int vkey = 4 * g + (l15 >> 2), vcol = (l15 & 3) >> 1, vhalf = (l15 & 1) << 3;
    unsigned v_off[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) v_off[dt] = (unsigned)(8192 + vkey * 128 + ((((dt << 1) | vcol) ^ (vkey & 7)) << 4) + vhalf);
    // where this lane's 4 consecutive output dims of dim tile dt land in the head's O tile (the GEMM's A image: chunk ^ (token & 7))
    unsigned o_off[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o_off[dt] = (unsigned)(qi * 128 + (((dt * 2 + (g >> 1)) ^ (qi & 7)) << 4) + (g & 1) * 8);

    // the epilogue's residual rows travel with the rounds too: round r (< 4) fetches the lane's 6 x 8 bytes of token tile r (asm loads the compiler does
    // not track; 48 registers held across the GEMM) — at the epilogue's own time all 256 workgroups would ask for their 77 KB at once and wait for HBM
    // (in the STORE layout: 16 bytes = the 8 consecutive columns the lane writes after the epilogue's lane exchange — 64-byte runs per row instead of 32)
    i32x4 res[4][NT3];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int a = 0; a < NT3; ++a) res[mt][a] = i32x4{0, 0, 0, 0};
    // ... and the weight prefetch the finalise launch used to carry (rowops.hip, LnExtra): thread t of the grid touches lines t and t + (threads of the grid)
    unsigned pf_reg[2] = {0u, 0u};
    const unsigned pf_nt = gridDim.x * 512u, pf_t = blockIdx.x * 512u + (unsigned)tid;

    bf16x8 qn[2] = {bf16x8{0, 0, 0, 0, 0, 0, 0, 0}, bf16x8{0, 0, 0, 0, 0, 0, 0, 0}};
    if (!(dbg & 1)) {
    stage(0);
    load_q(0, qn);
    // (the compiler takes an asm load's registers as valid the moment the statement has executed: it may copy or reuse them as soon as it has; so every
    // round ENDS with the wait and landed() for what it issued, in front of anything the compiler may place between two rounds)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    landed(qn[0]); landed(qn[1]);
    static_for<NG>([&](auto grp_tag) {
        constexpr int grp = decltype(grp_tag)::value;
        __builtin_amdgcn_s_barrier();   // everybody's pieces of this pair's image have landed (each wave waited for its own), and everybody is done with the image the next pair overwrites
        asm volatile("" ::: "memory");
        bf16x8 qf[2] = {qn[0], qn[1]};
        if constexpr (grp + 1 < NG) {
            stage(grp + 1);
            load_q(grp + 1, qn);
        }
        if constexpr (grp < 4) {
            const int m = grp * 16 + l15;
            const bf16_t* xr = x + (row0 + (m < len ? m : 0)) * W + wave * (W / 8) + (g & 1) * 16 + (g >> 1) * 8;
            static_for<NT3>([&](auto a_tag) {
                constexpr int a = decltype(a_tag)::value;
                res[grp][a] = __builtin_bit_cast(i32x4, gload16<a * 64>(xr));
            });
        }
        if constexpr (grp == 1 || grp == 3) {
            constexpr int j = grp >> 1;
            const unsigned c = pf_t + j * pf_nt;
            const unsigned* src = c < pf.na ? pf.a + (size_t)c * 32 : (c - pf.na < pf.nb ? pf.b + (size_t)(c - pf.na) * 32 : nullptr);
            if (src) asm volatile("global_load_dword %0, %1, off" : "=v"(pf_reg[j]) : "v"(src));
        }
        if (active) {
        const unsigned kvb = ring + (unsigned)(grp & 1) * AP_KV + (unsigned)hl * 16384u;
        // ---- every LDS read of the round up front (inline asm: hipcc would drain the LDS-DMA of the next pair in front of each) ----------------
        bf16x8 kf[4][2];
        static_for<4>([&](auto t_tag) {
            constexpr int t = decltype(t_tag)::value;
            kf[t][0] = lds_read16<t * 2048>(kvb + k_sw0);
            kf[t][1] = lds_read16<t * 2048>(kvb + k_sw1);
        });
        s16x4 vt[4][4];   // [key tile of 16][dim tile]
        static_for<4>([&](auto dt_tag) {
            constexpr int dt = decltype(dt_tag)::value;
            vt[0][dt] = lds_read_tr16<0>(kvb + v_off[dt]);
            vt[1][dt] = lds_read_tr16<2048>(kvb + v_off[dt]);
            vt[2][dt] = lds_read_tr16<4096>(kvb + v_off[dt]);
            vt[3][dt] = lds_read_tr16<6144>(kvb + v_off[dt]);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < 4; ++t) { landed(kf[t][0]); landed(kf[t][1]); }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) landed(vt[t][dt]);
        // ---- S^T = K Q^T: keys 16 t + 4 g + r of this lane's query ---------------------------------------------------------------------------
        f32x4 sc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            sc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t][kk], qf[kk], sc[t], 0, 0, 0);
        }
        if (len < 64) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[t][r] = (t * 16 + g * 4 + r < len) ? sc[t][r] : -INFINITY;
        }
        float mx = fmaxf(fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3])), fmaxf(fmaxf(sc[1][0], sc[1][1]), fmaxf(sc[1][2], sc[1][3])));
        mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(sc[2][0], sc[2][1]), fmaxf(sc[2][2], sc[2][3])), fmaxf(fmaxf(sc[3][0], sc[3][1]), fmaxf(sc[3][2], sc[3][3]))));
        mx = fold_max(mx);
        const float neg_mc = -mx * scale_log2e;
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sc[t][r] = __builtin_amdgcn_exp2f(fmaf(sc[t][r], scale_log2e, neg_mc));
                psum += sc[t][r];
            }
        // ---- O^T = V^T P^T over the two 32-key halves ---------------------------------------------------------------------------------------------
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            union { uint32_t w[4]; bf16x8 v; } pf;
            pf.w[0] = pack_bf16x2(sc[2 * u][0], sc[2 * u][1]);
            pf.w[1] = pack_bf16x2(sc[2 * u][2], sc[2 * u][3]);
            pf.w[2] = pack_bf16x2(sc[2 * u + 1][0], sc[2 * u + 1][1]);
            pf.w[3] = pack_bf16x2(sc[2 * u + 1][2], sc[2 * u + 1][3]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                union { s16x4 t[2]; bf16x8 v; } vf;
                vf.t[0] = vt[2 * u][dt];
                vf.t[1] = vt[2 * u + 1][dt];
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf.v, o[dt], 0, 0, 0);
            }
        }
        const float inv = 1.0f / fold_sum(psum);
        const unsigned ot = lds0 + (unsigned)(2 * grp + hl) * AP_TILE;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            u32x2 p;
            p[0] = pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv);
            p[1] = pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv);
            lds_write8(ot + o_off[dt], p);
        }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of the next pair's image, my next Q rows, this round's residual rows / weight lines
        landed(qn[0]); landed(qn[1]);
        if constexpr (grp < 4) {
#pragma unroll
            for (int a = 0; a < NT3; ++a) landed(res[grp][a]);
        }
        if constexpr (grp == 1 || grp == 3) landed(pf_reg[grp >> 1]);
    });
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my O rows are in the panel ...
    __builtin_amdgcn_s_barrier();                          // ... everybody's are, and the K / V images are dead: the ring is the weights' now
    asm volatile("" ::: "memory");

    // =============================== phase 2: y = O @ Wo^T, the panel resident, Wo streamed through private rings ======================================
    // unit u = (k-step kt = u / 6, third nt3 = (u % 6) / 2, half j = u % 2): weight rows n = 256 nt3 + 32 wave + 16 j .. + 15, k = 64 kt .. + 63 -> 2 pieces
    const unsigned slot0 = ring + (unsigned)wave * 8192u;
    const unsigned w_rd0 = (unsigned)(l15 * 128 + ((g ^ (l15 & 7)) << 4)), w_rd1 = (unsigned)(l15 * 128 + (((g + 4) ^ (l15 & 7)) << 4));
    const unsigned w_vo = (unsigned)((lane >> 3) * (W * 2) + (((lane & 7) ^ ((lane >> 3) & 7)) << 4));   // piece row lane / 8, logical chunk = physical ^ (row & 7)
    constexpr unsigned W_BYTES = (unsigned)W * W * 2;
    auto issue_unit = [&](int kt, int i, int slot, bool live) {   // i = 2 nt3 + j
        const unsigned rec = live ? W_BYTES : 0u;                // past the last unit: out of range for every lane — no traffic, the counters still tick
        const unsigned soff = (unsigned)((wave * (W / 8) + i * 16) * (W * 2) + kt * 128);
        const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)Wo, 0, rec, 0x00020000);
        dma16(rs, w_vo, soff, slot0 + (unsigned)slot * 2048u);
        dma16(rs, w_vo + 8u * (W * 2), soff, slot0 + (unsigned)slot * 2048u + 1024u);
    };
    f32x4 acc[NT3][4][2];
#pragma unroll
    for (int a = 0; a < NT3; ++a)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) { acc[a][mt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[a][mt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    constexpr int UPK = 2 * NT3;            // units per k-step (6)
    constexpr int UPI = 2 * UPK;            // units per loop iteration (two k-steps: 12, a multiple of the 4 slots)
    static_assert(UPI % 4 == 0, "slot indices must be static inside an iteration");
    bf16x8 tf[4][2], wf[2][2];
    issue_unit(0, 0, 0, true);
    issue_unit(0, 1, 1, true);
    issue_unit(0, 2, 2, true);
    issue_unit(0, 3, 3, true);
    static_for<4>([&](auto mt_tag) {
        constexpr int mt = decltype(mt_tag)::value;
        tf[mt][0] = lds_read16<mt * 2048>(lds0 + w_rd0);
        tf[mt][1] = lds_read16<mt * 2048>(lds0 + w_rd1);
    });
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // unit 0 landed (loads retire in order)
    wf[0][0] = lds_read16<0>(slot0 + w_rd0);
    wf[0][1] = lds_read16<0>(slot0 + w_rd1);
    for (int ktp = 0; ktp < ((dbg & 2) ? 0 : NHEAD / 2); ++ktp) {
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
                const int ktp4 = ktp + (ii + 4) / UPI;
                issue_unit(2 * ktp4 + i4 / UPK, i4 % UPK, slot, ktp4 < NHEAD / 2);
            }
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // ... units + 2, + 3, + 4 may be in flight: unit + 1 has landed
            {
                constexpr int nslot = (ii + 1) & 3;
                wf[cur ^ 1][0] = lds_read16<nslot * 2048>(slot0 + w_rd0);
                wf[cur ^ 1][1] = lds_read16<nslot * 2048>(slot0 + w_rd1);
            }
            if constexpr (i == UPK - 1) {   // the k-step's last unit: the next head's O tile (behind this unit's MFMAs in program order; stale past the last)
                __builtin_amdgcn_sched_barrier(0);
                const unsigned tb = lds0 + (unsigned)(kt + 1 < NHEAD ? kt + 1 : kt) * AP_TILE;
                static_for<4>([&](auto mt_tag) {
                    constexpr int mt = decltype(mt_tag)::value;
                    tf[mt][0] = lds_read16<mt * 2048>(tb + w_rd0);
                    tf[mt][1] = lds_read16<mt * 2048>(tb + w_rd1);
                });
            }
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (out-of-range) requests
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    landed(wf[0][0]); landed(wf[0][1]); landed(wf[1][0]); landed(wf[1][1]);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) { landed(tf[mt][0]); landed(tf[mt][1]); }

    // =============================== epilogue: x = bf16((acc + bias) + x) in place; (mean, rstd) of the complete rows ====================================
    if (dbg & 4) return;
    // Column map: wave w owns columns [96 w, 96 w + 96) = 6 sixteen-column tiles i = 2 a + j.  The LayerNorm statistics must carry the BITS of the partial-sum
    // path (gemm_epilogue.h MQ_EPI_ROW_STATS + row_stats_finalize_kernel): an image's embedding may not depend on whether its batch was big enough for this
    // kernel (the coalescer and the ingest merging lean on that; tests/test_towers_gpu.py batch-split invariance).  That path sums, per row and 64-column slot,
    // a lane's rounded values tile by tile (chain_add below, tiles in ascending order), adds the lane's two halves, then the row's 4 lanes (fold_sum), and
    // the finalise adds the 12 slots in order.  Here an even wave holds slot 3 k whole (tiles 0-3) and the first half of slot 3 k + 1 (tiles 4, 5), its odd
    // partner the second half of that slot (tiles 0, 1) and slot 3 k + 2 whole (tiles 2-5): the shared slot's chain is handed over through LDS.
    constexpr int WC = W / 8;               // columns per wave (96)
    f32x4 bias_v[NT3][2];
#pragma unroll
    for (int a = 0; a < NT3; ++a)
#pragma unroll
        for (int j = 0; j < 2; ++j) bias_v[a][j] = *(const f32x4*)(bias + wave * WC + a * 32 + j * 16 + 4 * g);
    char* mine = smem + PANEL + wave * 8192;   // this wave's own ring region: [0, 512) whole-slot sums [64 rows], [512, 1024) shared-slot sums, [1024, 5120) chain state [4][64 lanes]
    const bool odd = (wave & 1) != 0;          // wave-uniform
    auto chain_add = [](f32x2_t& s1, f32x2_t& s2, unsigned px, unsigned py) {
        const f32x2_t e0 = {__uint_as_float(px << 16), __uint_as_float(px & 0xffff0000u)}, e1 = {__uint_as_float(py << 16), __uint_as_float(py & 0xffff0000u)};
        s1 += e0 + e1;
        s2 = __builtin_elementwise_fma(e0, e0, __builtin_elementwise_fma(e1, e1, s2));
    };
    unsigned held[4][4];   // odd waves: the rounded values of tiles 0, 1 (the shared slot's second half) until the partner's chain state is there
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = mt * 16 + l15;
        const bool m_ok = m < len;
        bf16_t* xr = x + (row0 + (m_ok ? m : 0)) * W + wave * WC;
        f32x2_t w1 = {0.f, 0.f}, w2 = {0.f, 0.f};   // the whole slot's chain
        f32x2_t h1 = {0.f, 0.f}, h2 = {0.f, 0.f};   // even waves: the shared slot's first half
#pragma unroll
        for (int a = 0; a < NT3; ++a) {
            uint2 pk[2];
            // the residual quad back into the accumulators' layout: (low, high) halves exchanged between lanes 16 apart — the inverse of the store's exchange
            const auto q0 = __builtin_amdgcn_permlane16_swap((unsigned)res[mt][a][0], (unsigned)res[mt][a][2], false, false);
            const auto q1 = __builtin_amdgcn_permlane16_swap((unsigned)res[mt][a][1], (unsigned)res[mt][a][3], false, false);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x4 v = acc[a][mt][j];
                v += bias_v[a][j];
                const unsigned rq[2] = {q0[j], q1[j]};
                v += f32x4{__uint_as_float(rq[0] << 16), __uint_as_float(rq[0] & 0xffff0000u), __uint_as_float(rq[1] << 16), __uint_as_float(rq[1] & 0xffff0000u)};
                pk[j].x = pack_bf16x2(v[0], v[1]);
                pk[j].y = pack_bf16x2(v[2], v[3]);
                // statistics of the ROUNDED values (what the next GEMM multiplies), rows past the sequence count as zeros
                const unsigned px = m_ok ? pk[j].x : 0u, py = m_ok ? pk[j].y : 0u;
                if constexpr (NT3 == 3) {
                    if (a == 1) chain_add(w1, w2, px, py);                       // tiles 2, 3: the whole slot of either parity
                    else if (a == 0) {
                        if (odd) { held[mt][2 * j] = px; held[mt][2 * j + 1] = py; }
                        else chain_add(w1, w2, px, py);
                    } else {
                        if (odd) chain_add(w1, w2, px, py);
                        else chain_add(h1, h2, px, py);
                    }
                }
            }
            // the two 16-column blocks exchanged between lanes 16 apart: a lane then owns 8 consecutive columns (one 16-byte store), as gemm_epilogue.h
            const auto r0 = __builtin_amdgcn_permlane16_swap(pk[0].x, pk[1].x, false, false);
            const auto r1 = __builtin_amdgcn_permlane16_swap(pk[0].y, pk[1].y, false, false);
            if (m_ok) *(uint4*)(xr + a * 32 + (g & 1) * 16 + (g >> 1) * 8) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
        }
        if (rowstats) {
            const float st1 = fold_sum(w1[0] + w1[1]), st2 = fold_sum(w2[0] + w2[1]);
            if (g == 0) ((float2*)mine)[m] = make_float2(st1, st2);
            if (!odd) ((f32x4*)(mine + 1024))[mt * 64 + lane] = f32x4{h1[0], h1[1], h2[0], h2[1]};
        }
    }
    if (rowstats) {
        static_assert(NT3 == 3, "the slot hand-over is written for 96 columns per wave");
        __syncthreads();
        if (odd) {
            const f32x4* theirs = (const f32x4*)(smem + PANEL + (wave - 1) * 8192 + 1024);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const f32x4 st = theirs[mt * 64 + lane];
                f32x2_t h1 = {st[0], st[1]}, h2 = {st[2], st[3]};
                chain_add(h1, h2, held[mt][0], held[mt][1]);
                chain_add(h1, h2, held[mt][2], held[mt][3]);
                const float st1 = fold_sum(h1[0] + h1[1]), st2 = fold_sum(h2[0] + h2[1]);
                if (g == 0) ((float2*)(mine + 512))[mt * 16 + l15] = make_float2(st1, st2);
            }
        }
        __syncthreads();
        if (tid < len) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int slot = 0; slot < W / 64; ++slot) {   // slot order, as row_stats_finalize_kernel
                const int k = slot / 3, r = slot % 3;
                const char* src = smem + PANEL + (2 * k + (r ? 1 : 0)) * 8192 + (r == 1 ? 512 : 0);
                const float2 p = ((const float2*)src)[tid];
                s1 += p.x;
                s2 += p.y;
            }
            rowstats[row0 + tid] = mq_finalize_stats(s1, s2, inv_w, eps);
        }
    }
}

}  // namespace

// 1 = the shapes mq_attention_proj takes (towers.hip asks before it plans a block around it)
extern "C" int mq_attention_proj_ok(int64_t nseq, int32_t fixed_len, int32_t W, int32_t heads) {
    return nseq >= 1 && fixed_len >= 1 && fixed_len <= 64 && W == 768 && heads == 12 && nseq * fixed_len < (1LL << 31);
}

// x[rows, W] (bf16, in place) += out_proj(attention(qkv)) + bias for nseq sequences of fixed_len <= 64 tokens (rows = nseq * fixed_len, no mask);
// d_rowstats (optional) receives (mean, rstd) of every written row, as mq_row_stats_finalize leaves them (eps: the LayerNorm's).
// d_qkv bf16 [rows, 3 W] (q | k | v, head-major columns), d_w bf16 [W, W] row-major (the nn.Linear weight as stored), d_bias fp32 [W].
// d_pf_a / d_pf_b (optional): weight ranges of the GEMMs behind this launch, touched one dword per 128-byte line (mq_tune("ln_prefetch", 0): not).
extern "C" int mq_attention_proj(const void* d_qkv, const void* d_w, const float* d_bias, void* d_x, float* d_rowstats, int64_t nseq, int32_t fixed_len, int32_t W,
                                 int32_t heads, float eps, const void* d_pf_a, size_t pf_a_bytes, const void* d_pf_b, size_t pf_b_bytes, void* stream) {
    MQ_CHECK_ARG(d_qkv && d_w && d_bias && d_x, "mq_attention_proj: null operand");
    MQ_CHECK_ARG(mq_attention_proj_ok(nseq, fixed_len, W, heads), "mq_attention_proj: takes 1..64-token sequences of a 768-wide tower with 12 heads (nseq=%ld len=%d W=%d heads=%d)",
                 (long)nseq, fixed_len, W, heads);
    MQ_CHECK_ARG((((uintptr_t)d_qkv | (uintptr_t)d_w | (uintptr_t)d_x | (uintptr_t)d_bias) & 15) == 0, "mq_attention_proj: operands must be 16-byte aligned");
    MQ_TRY(mq_device_ok());
    hipStream_t s = (hipStream_t)stream;
    constexpr int LDS = 12 * AP_TILE + AP_RING;   // 160 KiB: the whole CU's
    static std::atomic<uint64_t> attr_done{0};
    auto kern = attn_proj_kernel<768>;
    if (hipError_t e = mq_ensure_dyn_lds((const void*)kern, LDS, attr_done); e != hipSuccess) {
        mq_set_error("mq_attention_proj: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return MQ_ERR_HIP;
    }
    ApPrefetch pf{nullptr, nullptr, 0u, 0u};
    if (mq_ln_prefetch && nseq * fixed_len >= 1024) {   // (as the LayerNorm kernels': a small call is latency-bound, nothing to hide the touches behind)
        auto lines = [](const void* p, size_t b) { return (p && ((uintptr_t)p & 3) == 0 && b < ((size_t)1 << 30)) ? (unsigned)(b / 128) : 0u; };
        pf.a = (const unsigned*)d_pf_a; pf.na = lines(d_pf_a, pf_a_bytes);
        pf.b = (const unsigned*)d_pf_b; pf.nb = lines(d_pf_b, pf_b_bytes);
    }
    // MQ_AP_DEBUG (tools/attn_proj_bench.py --phases; looked at per launch only when it was set at the first one): bits 1 / 2 / 4 = without phase 1 / the GEMM
    // loop / the epilogue — timing builds of the phases, the results are then meaningless
    static const bool dbg_on = getenv("MQ_AP_DEBUG") != nullptr;
    const char* dbg_s = dbg_on ? getenv("MQ_AP_DEBUG") : nullptr;
    const int dbg = dbg_s ? atoi(dbg_s) : 0;
    MqProfScope prof(2, s);
    const float scale_log2e = 1.44269504088896340736f / sqrtf(64.0f);
    hipLaunchKernelGGL(kern, dim3((unsigned)nseq), dim3(512), LDS, s, (const bf16_t*)d_qkv, (const bf16_t*)d_w, d_bias, (bf16_t*)d_x, (float2*)d_rowstats, (int)fixed_len,
                       scale_log2e, 1.0f / (float)W, eps, (mq_xcd_band && nseq >= 256) ? 1 : 0, pf, dbg);
    MQ_CHECK_LAUNCH("mq_attention_proj");
    return MQ_OK;
}
