// Multi-head attention for short sequences (T = 50 / 77 / 257 / <= 512), d_head = 64 (K4).
//
// One workgroup (4 wave64s) per (sequence, head).  The whole K [len,64] and V^T [64,len] of that
// (sequence, head) are staged ONCE in LDS (160 KB/CU makes this possible up to 512 keys: 64 KB +
// 64 KB), then each wave owns 16-query blocks and runs a flash-style online softmax over 64-key
// tiles with both GEMMs on v_mfma_f32_16x16x32_bf16:
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
#include "common.h"

namespace {

// OUT_FP8: the output is written as e4m3 codes = value / *out_scale (static per-tensor scale of the fp8 path, K13) and
// max|value| is folded into *amax when it is non-null (calibration).
template <int MASK, bool OUT_FP8>
__global__ __launch_bounds__(256) void attention_kernel(
    const bf16_t* __restrict__ qkv, void* __restrict__ out_v, const int32_t* __restrict__ cu,
    int fixed_len, int W, int heads, int kpad, float scale_log2e, const float* __restrict__ out_scale, float* amax_out) {
    bf16_t* out = (bf16_t*)out_v;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;                                   // [kpad][128 B], 16-B chunks XOR-swizzled by (key & 7)
    bf16_t* sVt = (bf16_t*)(smem + (size_t)kpad * 128);  // [64][kpad + 4]
    const int vstride = kpad + 4;

    const int seq = blockIdx.x / heads, h = blockIdx.x - seq * heads;
    int row0, len;
    if (fixed_len > 0) { row0 = seq * fixed_len; len = fixed_len; }
    else { row0 = cu[seq]; len = cu[seq + 1] - row0; }
    if (len <= 0) return;
    const int ld = 3 * W;
    const bf16_t* qb = qkv + (int64_t)row0 * ld + h * 64;
    const bf16_t* kb = qb + W;
    const bf16_t* vb = qb + 2 * W;
    const int nkt = (len + 63) >> 6;
    const int kp = nkt << 6;

    // ---- stage K (row-major, swizzled) and V^T (transposed scatter) ----------------------------
    for (int idx = threadIdx.x; idx < kp * 8; idx += 256) {
        const int key = idx >> 3, chunk = idx & 7;
        uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
        if (key < len) {
            kv = *(const uint4*)(kb + (int64_t)key * ld + chunk * 8);
            vv = *(const uint4*)(vb + (int64_t)key * ld + chunk * 8);
        }
        *(uint4*)(sK + key * 128 + ((chunk ^ (key & 7)) << 4)) = kv;
        const uint32_t w[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sVt[(chunk * 8 + 2 * j) * vstride + key] = (bf16_t)(w[j] & 0xffffu);
            sVt[(chunk * 8 + 2 * j + 1) * vstride + key] = (bf16_t)(w[j] >> 16);
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int nqb = (len + 15) >> 4;
    float amax_local = 0.f;

    for (int qblk = wave; qblk < nqb; qblk += 4) {
        const int q = qblk * 16 + l15;           // this lane's query (B-operand column / output row)
        const int qr = q < len ? q : len - 1;    // clamp loads of the ragged tail
        bf16x8 qf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) qf[kk] = *(const bf16x8*)(qb + (int64_t)qr * ld + 8 * g + 32 * kk);

        float m_run = -1e30f, l_run = 0.f;
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

        int kt_end = nkt;
        if (MASK == MQ_MASK_CAUSAL) {
            const int last = (qblk * 16 + 15) >> 6;  // last key tile any query of this block may see
            kt_end = last + 1 < nkt ? last + 1 : nkt;
        }
        for (int kt = 0; kt < kt_end; ++kt) {
            // ---- S^T tile: keys 64kt + 16t + 4g + r for this lane's query -----------------------
            f32x4 sc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                sc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int key = kt * 64 + t * 16 + l15;  // A-operand row
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const bf16x8 kf = *(const bf16x8*)(sK + key * 128 + (((g + 4 * kk) ^ (key & 7)) << 4));
                    sc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], sc[t], 0, 0, 0);
                }
            }
            float mx = -1e30f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 64 + t * 16 + g * 4 + r;
                    bool valid = key < len;
                    if (MASK == MQ_MASK_CAUSAL) valid = valid && (key <= q);
                    sc[t][r] = valid ? sc[t][r] * scale_log2e : -INFINITY;
                    mx = fmaxf(mx, sc[t][r]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = exp2f(m_run - m_new);
            float psum = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sc[t][r] = exp2f(sc[t][r] - m_new);
                    psum += sc[t][r];
                }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;

            // ---- O^T += V^T . P^T over the tile's two 32-key halves --------------------------------
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                union { uint32_t w[4]; bf16x8 v; } pf;
                pf.w[0] = pack_bf16x2(sc[2 * u][0], sc[2 * u][1]);
                pf.w[1] = pack_bf16x2(sc[2 * u][2], sc[2 * u][3]);
                pf.w[2] = pack_bf16x2(sc[2 * u + 1][0], sc[2 * u + 1][1]);
                pf.w[3] = pack_bf16x2(sc[2 * u + 1][2], sc[2 * u + 1][3]);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bf16_t* vrow = sVt + (dt * 16 + l15) * vstride + kt * 64 + 32 * u + 4 * g;
                    union { uint2 h[2]; bf16x8 v; } vf;
                    vf.h[0] = *(const uint2*)(vrow);        // keys 16*(2u)   + 4g .. +3
                    vf.h[1] = *(const uint2*)(vrow + 16);   // keys 16*(2u+1) + 4g .. +3
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf.v, o[dt], 0, 0, 0);
                }
            }
        }
        float l_tot = l_run + __shfl_xor(l_run, 16, 64);
        l_tot += __shfl_xor(l_tot, 32, 64);
        const float inv = 1.0f / l_tot;
        if (q < len) {
            if (OUT_FP8) {
                const float qs = 1.0f / out_scale[0];
                uint8_t* orow8 = (uint8_t*)out_v + (int64_t)(row0 + q) * W + h * 64 + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
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
                bf16_t* orow = out + (int64_t)(row0 + q) * W + h * 64 + 4 * g;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    uint2 p;
                    p.x = pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv);
                    p.y = pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv);
                    *(uint2*)(orow + dt * 16) = p;
                }
            }
        }
    }
    if (OUT_FP8 && amax_out) {
        amax_local = wave_max(amax_local);
        if (lane == 0) atomicMax((int*)amax_out, __float_as_int(amax_local));
    }
}

}  // namespace

extern "C" int mq_attention_ex(const void* d_qkv, void* d_out, const int32_t* d_cu_seqlens, int64_t nseq,
                               int32_t fixed_len, int32_t max_len, int32_t W, int32_t heads, int32_t mask,
                               int32_t out_fp8, const float* d_out_scale, float* d_amax, void* stream) {
    MQ_CHECK_ARG(d_qkv && d_out, "mq_attention: null pointer");
    MQ_CHECK_ARG(heads >= 1 && W == heads * 64, "mq_attention: head dim must be 64 (W=%d heads=%d)", W, heads);
    MQ_CHECK_ARG(fixed_len > 0 || d_cu_seqlens, "mq_attention: need fixed_len or cu_seqlens");
    MQ_CHECK_ARG(mask == MQ_MASK_NONE || mask == MQ_MASK_CAUSAL, "mq_attention: bad mask %d", mask);
    if (nseq <= 0) return MQ_OK;
    const int maxl = fixed_len > 0 ? fixed_len : max_len;
    MQ_CHECK_ARG(maxl >= 1 && maxl <= 1024, "mq_attention: max sequence length %d unsupported (1..1024)", maxl);
    MQ_CHECK_ARG(nseq * heads < (1LL << 31), "mq_attention: grid too large");
    const int kpad = ((maxl + 63) / 64) * 64;
    const size_t lds = (size_t)kpad * 128 + (size_t)64 * (kpad + 4) * 2;
    MQ_CHECK_ARG(lds <= 160 * 1024, "mq_attention: sequence length %d needs %zu B of LDS (> 160 KiB)", maxl, lds);
    hipStream_t s = (hipStream_t)stream;
    const float scale_log2e = 0.125f * 1.44269504088896340736f;  // 1/sqrt(64) * log2(e)
    MqProfScope prof(2, s);
    auto launch = [&](auto kern) -> int {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) { mq_set_error("mq_attention: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MQ_ERR_HIP; }
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)(nseq * heads)), dim3(256), lds, s, (const bf16_t*)d_qkv,
                           d_out, d_cu_seqlens, (int)fixed_len, (int)W, (int)heads, kpad, scale_log2e, d_out_scale, d_amax);
        return MQ_OK;
    };
    MQ_CHECK_ARG(!out_fp8 || d_out_scale, "mq_attention: fp8 output needs an out_scale");
    int rc;
    if (out_fp8) rc = (mask == MQ_MASK_CAUSAL) ? launch(attention_kernel<MQ_MASK_CAUSAL, true>) : launch(attention_kernel<MQ_MASK_NONE, true>);
    else rc = (mask == MQ_MASK_CAUSAL) ? launch(attention_kernel<MQ_MASK_CAUSAL, false>) : launch(attention_kernel<MQ_MASK_NONE, false>);
    if (rc != MQ_OK) return rc;
    MQ_CHECK_LAUNCH("mq_attention");
    return MQ_OK;
}

extern "C" int mq_attention(const void* d_qkv, void* d_out, const int32_t* d_cu_seqlens, int64_t nseq, int32_t fixed_len,
                            int32_t max_len, int32_t W, int32_t heads, int32_t mask, void* stream) {
    return mq_attention_ex(d_qkv, d_out, d_cu_seqlens, nseq, fixed_len, max_len, W, heads, mask, 0, nullptr, nullptr, stream);
}
