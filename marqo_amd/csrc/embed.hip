// Front/back ends of the towers (K1 gather, K6/K8 gathers, K7, K9, K10-normalise):
//   * patchify: uint8 HWC (or fp32 CHW) image -> bf16 im2col matrix [n*np, Kp]; since the conv
//     has stride == kernel this is a pure index remap; ToTensor (/255) + Normalize are fused in.
//   * vit_assemble: class token + patch embeddings + positional embedding + ln_pre -> fp32 stream.
//   * embed_tokens: token (+position, +type) embedding gather (+ LayerNorm for BERT).
//   * pool: masked mean / CLS pooling (+ L2) over packed sequences.
#include "common.h"

namespace {

// ---- patchify ---------------------------------------------------------------------------------
// out[(img*np + py*G + px), c*P*P + ky*P + kx] = norm(in[img, py*P+ky, px*P+kx, c]); cols >= 3*P*P are 0.
// One thread produces 8 consecutive output columns (one 16-byte store).
template <bool U8>
__global__ __launch_bounds__(256) void patchify_kernel(const void* __restrict__ in, bf16_t* __restrict__ out,
                                                       int64_t total_groups, int S, int P, int G, int Kp,
                                                       float sc0, float sc1, float sc2, float of0, float of1, float of2) {
    const int groups_per_row = Kp >> 3;
    const int PP = P * P;
    // uint8 pixels: (b / 255 - mean) / std evaluated once per (channel, byte value) into an LDS table (see patchify_u8x8_kernel)
    __shared__ float lutf[U8 ? 3 * 256 : 1];
    if (U8) {
        const float t = (float)threadIdx.x;
        lutf[threadIdx.x] = (t / 255.0f - of0) / sc0;
        lutf[(U8 ? 256 : 0) + (U8 ? threadIdx.x : 0)] = (t / 255.0f - of1) / sc1;
        lutf[(U8 ? 512 : 0) + (U8 ? threadIdx.x : 0)] = (t / 255.0f - of2) / sc2;
        __syncthreads();
    }
    for (int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x; gi < total_groups; gi += (int64_t)gridDim.x * 256) {
        const int64_t prow = gi / groups_per_row;
        const int col0 = (int)(gi - prow * groups_per_row) * 8;
        const int64_t img = prow / (G * G);
        const int pidx = (int)(prow - img * (G * G));
        const int py = pidx / G, px = pidx - py * G;
        float v[8];
        if (U8 && (P & 7) == 0 && col0 < 3 * PP) {
            // fast path: the 8 columns are 8 consecutive kx of one (c, ky) row -> one div chain per thread
            const int c = col0 / PP;
            const int rem = col0 - c * PP;
            const int ky = rem / P, kx = rem - ky * P;
            const uint8_t* src = (const uint8_t*)in + ((img * S + (py * P + ky)) * S + (px * P + kx)) * 3 + c;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = lutf[U8 ? c * 256 + src[e * 3] : 0];
        } else
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int col = col0 + e;
            float val = 0.f;
            if (col < 3 * PP) {
                const int c = col / PP;
                const int rem = col - c * PP;
                const int ky = rem / P, kx = rem - ky * P;
                const int y = py * P + ky, x = px * P + kx;
                if (U8) {
                    const uint8_t b = ((const uint8_t*)in)[((img * S + y) * S + x) * 3 + c];
                    // ToTensor: b / 255 ; Normalize: (t - mean) / std   (clip_utils.py:61-66), from the table
                    val = lutf[U8 ? c * 256 + b : 0];
                } else {
                    val = ((const float*)in)[((img * 3 + c) * S + y) * (int64_t)S + x];
                }
            }
            v[e] = val;
        }
        uint4 p;
        p.x = pack_bf16x2(v[0], v[1]);
        p.y = pack_bf16x2(v[2], v[3]);
        p.z = pack_bf16x2(v[4], v[5]);
        p.w = pack_bf16x2(v[6], v[7]);
        *(uint4*)(out + prow * Kp + col0) = p;
    }
}

// uint8 fast path for P % 8 == 0 (ViT-B/32, B/16): a thread takes 8 consecutive pixels of one patch row = 24 CONTIGUOUS
// bytes (three 8-byte loads) and writes the three channels' 8 columns (three 16-byte stores).  The generic kernel above
// reads one channel at a time, i.e. every third byte: PMC showed 3.4x the algorithmic read traffic (profiles/r01_traffic*).
__global__ __launch_bounds__(256) void patchify_u8x8_kernel(const uint8_t* __restrict__ in, bf16_t* __restrict__ out, int64_t total,
                                                            int S, int P, int G, int Kp, float sc0, float sc1, float sc2,
                                                            float of0, float of1, float of2) {
    // ToTensor: b / 255 ; Normalize: (t - mean) / std   (clip_utils.py:61-66) — the same expression as the generic path, but evaluated ONCE per
    // (channel, byte value): two fp32 divisions per element made this kernel VALU-bound (36 us for 115 MB at the headline); a pixel is a byte,
    // so the 3 x 256 possible results sit in an LDS table and an element costs one ds_read_u16.  Bit-identical by construction.
    __shared__ bf16_t lut[3 * 256];
    {
        const float t = (float)threadIdx.x;
        lut[threadIdx.x] = f32_to_bf16((t / 255.0f - of0) / sc0);
        lut[256 + threadIdx.x] = f32_to_bf16((t / 255.0f - of1) / sc1);
        lut[512 + threadIdx.x] = f32_to_bf16((t / 255.0f - of2) / sc2);
    }
    __syncthreads();
    const int per_row = P >> 3, per_patch = P * per_row, PP = P * P;
    for (int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x; gi < total; gi += (int64_t)gridDim.x * 256) {
        const int64_t prow = gi / per_patch;
        const int rem = (int)(gi - prow * per_patch);
        const int ky = rem / per_row, kx = (rem - ky * per_row) * 8;
        const int64_t img = prow / (G * G);
        const int pidx = (int)(prow - img * (G * G));
        const int py = pidx / G, px = pidx - py * G;
        const uint2* src = (const uint2*)(in + ((img * S + (py * P + ky)) * S + (px * P + kx)) * 3);
        const uint2 w0 = src[0], w1 = src[1], w2 = src[2];
        const uint32_t words[6] = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y};
        uint32_t v[3][8];
#pragma unroll
        for (int b = 0; b < 24; ++b) {
            const uint32_t byte = (words[b >> 2] >> ((b & 3) * 8)) & 0xffu;
            const int c = b % 3, e = b / 3;
            v[c][e] = (uint32_t)lut[c * 256 + byte];
        }
        bf16_t* dst = out + prow * Kp + ky * P + kx;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            uint4 p;
            p.x = v[c][0] | (v[c][1] << 16);
            p.y = v[c][2] | (v[c][3] << 16);
            p.z = v[c][4] | (v[c][5] << 16);
            p.w = v[c][6] | (v[c][7] << 16);
            *(uint4*)(dst + c * PP) = p;
        }
    }
}

// ---- ViT token assembly + ln_pre ------------------------------------------------------------------
// row (b, t): t == 0 ? cls : patch_out[b*np + t-1];  + pos[t];  LayerNorm(ln_pre) -> x fp32
// cls == nullptr: no class token (T = np; timm / SigLIP ViTs), gam == nullptr: no ln_pre (their patch bias is folded into pos)
constexpr int MAXC = 8;
template <int CH>
__global__ __launch_bounds__(256, (CH <= 4 ? 8 : 4)) void vit_assemble_kernel(
    const float* __restrict__ patch_out, const float* __restrict__ cls, const float* __restrict__ pos,
    const float* __restrict__ gam, const float* __restrict__ bet, float* __restrict__ x, int64_t rows, int T, int W, float eps, int x_bf16) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    auto put = [&](int c, const f32x4& y) {   // the residual stream is fp32, or bf16 when the tower runs its bf16-stream form
        if (x_bf16) {
            uint2 q;
            q.x = pack_bf16x2(y[0], y[1]);
            q.y = pack_bf16x2(y[2], y[3]);
            *(uint2*)((bf16_t*)x + row * W + c * 4) = q;
        } else {
            *(f32x4*)(x + row * W + c * 4) = y;
        }
    };
    const int64_t b = row / T;
    const int t = (int)(row - b * T);
    const float* src = !cls ? patch_out + row * W : (t == 0 ? cls : patch_out + (b * (T - 1) + (t - 1)) * W);
    const float* pr = pos + (int64_t)t * W;
    const int nch = W >> 2;
    f32x4 v[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = lane + i * 64;
        v[i] = c < nch ? *(const f32x4*)(src + c * 4) + *(const f32x4*)(pr + c * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (!gam) {  // wave-uniform
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = lane + i * 64;
            if (c < nch) put(c, v[i]);
        }
        return;
    }
    ln_normalize_row<CH>(v, lane, nch, W, eps);
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            const f32x4 gg = *(const f32x4*)(gam + c * 4);
            const f32x4 bb = *(const f32x4*)(bet + c * 4);
            f32x4 y;
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = v[i][e] * gg[e] + bb[e];
            put(c, y);
        }
    }
}

// ---- attention pooling with ONE learned query (timm AttentionPoolLatent, SigLIP's 'map' head) --------------------------------
// kv: bf16 [n*T, 2W] = (K | V) rows of every token; q: fp32 [W], already scaled by 1/sqrt(hd); out: bf16 [n, W].
// One wave per (image, head): lanes take keys t = lane, lane + 64, ... for the scores (each reads its key's hd-wide row), a wave
// softmax, then lanes take output dims (hd <= 128: dims lane, lane + 64) and walk the keys with the probabilities in LDS.
// HBM-bound on the kv read (n*T*2W*2 bytes, once).
__global__ __launch_bounds__(256) void map_pool_kernel(const bf16_t* __restrict__ kv, const float* __restrict__ q, bf16_t* __restrict__ out,
                                                       int64_t nh, int T, int W, int heads, int hd) {
    extern __shared__ float sp[];  // [4 waves][T]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;
    if (item >= nh) return;
    const int64_t img = item / heads;
    const int h = (int)(item - img * heads);
    float* p = sp + wave * T;
    const bf16_t* kbase = kv + img * T * (int64_t)(2 * W) + h * hd;
    const float* qh = q + h * hd;
    float mx = -INFINITY;
    for (int t = lane; t < T; t += 64) {
        const bf16_t* kr = kbase + (int64_t)t * (2 * W);
        float s = 0.f;
        for (int d = 0; d < hd; d += 8) {
            const uint4 k8 = *(const uint4*)(kr + d);
            const uint32_t kw[4] = {k8.x, k8.y, k8.z, k8.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                s += __uint_as_float(kw[e] << 16) * qh[d + 2 * e] + __uint_as_float(kw[e] & 0xffff0000u) * qh[d + 2 * e + 1];
        }
        p[t] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int t = lane; t < T; t += 64) {
        const float e = __expf(p[t] - mx);
        p[t] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // this wave's LDS writes above are read by its other lanes below
    __builtin_amdgcn_wave_barrier();
    const bf16_t* vbase = kbase + W;
    float o0 = 0.f, o1 = 0.f;
    const bool two = lane + 64 < hd;
    if (lane < hd) {
        for (int t = 0; t < T; ++t) {
            const bf16_t* vr = vbase + (int64_t)t * (2 * W);
            const float pt = p[t];
            o0 += pt * bf16_to_f32(vr[lane]);
            if (two) o1 += pt * bf16_to_f32(vr[lane + 64]);
        }
        bf16_t* orow = out + img * W + h * hd;
        orow[lane] = f32_to_bf16(o0 * inv);
        if (two) orow[lane + 64] = f32_to_bf16(o1 * inv);
    }
}

// ---- token embedding (+ pos, + type) (+ LayerNorm) ---------------------------------------------------
// grid = nseq blocks; the 4 waves of a block walk the sequence's rows.
template <bool LN, int CH>
__global__ __launch_bounds__(256, (CH <= 2 ? 8 : 4)) void embed_tokens_kernel(
    const int32_t* __restrict__ ids, const int32_t* __restrict__ cu, const float* __restrict__ tok,
    const float* __restrict__ pos, const float* __restrict__ type0, const float* __restrict__ gam,
    const float* __restrict__ bet, float* __restrict__ x, bf16_t* __restrict__ xb, int W, int vocab, float eps, int last_pos) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = cu[blockIdx.x], len = cu[blockIdx.x + 1] - row0;
    const int nch = W >> 2;
    for (int t = wave; t < len; t += 4) {
        const int64_t row = row0 + t;
        int id = ids[row];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const float* tr = tok + (int64_t)id * W;
        // (rotary models carry no absolute position table; last_pos > 0: the sequence's last row — CoCa's appended class embedding — sits at that
        // position whatever the sequence's length, the padding between it and the text is never a row)
        const float* pr = pos ? pos + (int64_t)((last_pos > 0 && t == len - 1) ? last_pos : t) * W : nullptr;
        f32x4 v[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = lane + i * 64;
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < nch) {
                v[i] = *(const f32x4*)(tr + c * 4);
                if (pr) v[i] += *(const f32x4*)(pr + c * 4);
                if (type0) v[i] += *(const f32x4*)(type0 + c * 4);
            }
        }
        if (LN) ln_normalize_row<CH>(v, lane, nch, W, eps);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = lane + i * 64;
            if (c < nch) {
                f32x4 y = v[i];
                if (LN) {
                    const f32x4 gg = *(const f32x4*)(gam + c * 4);
                    const f32x4 bb = *(const f32x4*)(bet + c * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = v[i][e] * gg[e] + bb[e];
                }
                if (x) *(f32x4*)(x + row * W + c * 4) = y;
                if (xb) {
                    uint2 p;
                    p.x = pack_bf16x2(y[0], y[1]);
                    p.y = pack_bf16x2(y[2], y[3]);
                    *(uint2*)(xb + row * W + c * 4) = p;
                }
            }
        }
    }
}

// ---- pooling over packed sequences (+ L2) ---------------------------------------------------------------
// grid = nseq; thread owns columns tid, tid+256, ...  (W <= 2048)
__global__ __launch_bounds__(256) void pool_kernel(const float* __restrict__ x, const int32_t* __restrict__ cu,
                                                   float* __restrict__ out, int W, int pool, int normalize) {
    __shared__ float red[4];
    const int row0 = cu[blockIdx.x], len = cu[blockIdx.x + 1] - row0;
    float acc[8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        acc[i] = 0.f;
        const int c = threadIdx.x + i * 256;
        if (c < W) {
            if (pool == MQ_POOL_CLS) {
                acc[i] = x[(int64_t)row0 * W + c];
            } else {
                float s = 0.f;
                for (int t = 0; t < len; ++t) s += x[(int64_t)(row0 + t) * W + c];
                // reference divides by attention_mask.sum() with no epsilon (hugging_face_model.py:208-209)
                acc[i] = s / (float)len;
            }
            ss += acc[i] * acc[i];
        }
    }
    float inv = 1.f;
    if (normalize) {
        ss = wave_sum(ss);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
        __syncthreads();
        const float tot = red[0] + red[1] + red[2] + red[3];
        // F.normalize(p=2, dim=1): x / max(||x||, 1e-12)   (hugging_face_model.py:194-195)
        inv = 1.0f / fmaxf(sqrtf(tot), 1e-12f);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = threadIdx.x + i * 256;
        if (c < W) out[(int64_t)blockIdx.x * W + c] = acc[i] * inv;
    }
}

// last-row index of every packed sequence (CLIP text EOT pooling when the caller passes no rows)
__global__ void last_rows_kernel(const int32_t* __restrict__ cu, int32_t* __restrict__ rows, int nseq) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nseq) rows[i] = cu[i + 1] - 1;
}
// class-token row of every image
__global__ void cls_rows_kernel(int32_t* __restrict__ rows, int n, int T) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rows[i] = i * T;
}

// dense[i,:] <- sparse[idx[i],:] (SCATTER = false) or sparse[idx[i],:] <- dense[i,:] (true); rows are `chunks` 16-B pieces
template <bool SCATTER>
__global__ __launch_bounds__(256) void move_rows_kernel(uint4* __restrict__ sparse, const int32_t* __restrict__ idx,
                                                         uint4* __restrict__ dense, int64_t n, int chunks) {
    const int64_t total = n * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / chunks;
        const int c = (int)(i - r * chunks);
        uint4* sp = sparse + (int64_t)idx[r] * chunks + c;
        if (SCATTER) *sp = dense[i];
        else dense[i] = *sp;
    }
}

}  // namespace

// ---- internal host entry points (declared in towers.hip) ---------------------------------------------------
int mq_patchify(const void* d_in, bool is_u8, void* d_out, int64_t n, int S, int P, int Kp,
                const float* mean, const float* std, hipStream_t s) {
    const int G = S / P;
    const int64_t total = n * G * G * (Kp >> 3);
    if (total <= 0) return MQ_OK;
    MqProfScope prof(5, s);
    const unsigned grid = (unsigned)(cdiv64(total, 256) < 16384 ? cdiv64(total, 256) : 16384);
    if (is_u8 && (P & 7) == 0 && Kp == 3 * P * P && (S * 3) % 8 == 0 && ((uintptr_t)d_in & 7) == 0) {
        const int64_t items = n * G * G * P * (P >> 3);
        const unsigned g8 = (unsigned)(cdiv64(items, 256) < 16384 ? cdiv64(items, 256) : 16384);
        hipLaunchKernelGGL(patchify_u8x8_kernel, dim3(g8), dim3(256), 0, s, (const uint8_t*)d_in, (bf16_t*)d_out, items, S, P, G, Kp,
                           std[0], std[1], std[2], mean[0], mean[1], mean[2]);
    } else if (is_u8)
        hipLaunchKernelGGL(patchify_kernel<true>, dim3(grid), dim3(256), 0, s, d_in, (bf16_t*)d_out, total, S, P, G, Kp,
                           std[0], std[1], std[2], mean[0], mean[1], mean[2]);
    else
        hipLaunchKernelGGL(patchify_kernel<false>, dim3(grid), dim3(256), 0, s, d_in, (bf16_t*)d_out, total, S, P, G, Kp,
                           1.f, 1.f, 1.f, 0.f, 0.f, 0.f);
    MQ_CHECK_LAUNCH("mq_patchify");
    return MQ_OK;
}

int mq_vit_assemble(const float* d_patch_out, const float* cls, const float* pos, const float* g, const float* b,
                    float* d_x, int64_t n, int T, int W, float eps, hipStream_t s, int x_bf16) {
    MQ_CHECK_ARG(W % 4 == 0 && W <= 64 * 4 * MAXC, "vit_assemble: W=%d unsupported", W);
    const int64_t rows = n * T;
    if (rows <= 0) return MQ_OK;
    MqProfScope prof(3, s);
    MQ_DISPATCH_CH(W, hipLaunchKernelGGL(vit_assemble_kernel<CH>, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, s, d_patch_out,
                                         cls, pos, g, b, d_x, rows, T, W, eps, x_bf16));
    MQ_CHECK_LAUNCH("vit_assemble");
    return MQ_OK;
}

// mean over the tokens [first, T) of every image (open_clip VisionTransformer pool_type 'avg': the patch tokens, without the class token)
__global__ __launch_bounds__(256) void avg_tokens_kernel(const void* __restrict__ xv, int x_bf16, float* __restrict__ out, int T, int first, int W) {
    const int64_t img = blockIdx.x;
    const float inv = 1.0f / (float)(T - first);
    for (int c = threadIdx.x * 4; c < W; c += 256 * 4) {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t = first; t < T; ++t) {
            const int64_t o = (img * T + t) * (int64_t)W + c;
            if (x_bf16) {
                const uint2 q = *(const uint2*)((const bf16_t*)xv + o);
                acc += f32x4{__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u)};
            } else {
                acc += *(const f32x4*)((const float*)xv + o);
            }
        }
        *(f32x4*)(out + img * W + c) = acc * inv;
    }
}

int mq_avg_tokens(const void* d_x, int x_bf16, float* d_out, int64_t n, int T, int first, int W, hipStream_t s) {
    MQ_CHECK_ARG(W % 4 == 0 && first >= 0 && first < T, "avg_tokens: bad shape W=%d T=%d first=%d", W, T, first);
    if (n <= 0) return MQ_OK;
    MqProfScope prof(4, s);
    hipLaunchKernelGGL(avg_tokens_kernel, dim3((unsigned)n), dim3(256), 0, s, d_x, x_bf16, d_out, T, first, W);
    MQ_CHECK_LAUNCH("avg_tokens");
    return MQ_OK;
}

int mq_map_pool(const void* d_kv, const float* d_q, void* d_out, int64_t n, int T, int W, int heads, hipStream_t s) {
    MQ_CHECK_ARG(heads >= 1 && W % heads == 0, "map_pool: W=%d heads=%d", W, heads);
    const int hd = W / heads;
    MQ_CHECK_ARG(hd % 8 == 0 && hd <= 128, "map_pool: head dim %d unsupported (multiple of 8, <= 128)", hd);
    MQ_CHECK_ARG(T >= 1 && T <= 4096, "map_pool: %d tokens unsupported", T);
    if (n <= 0) return MQ_OK;
    MqProfScope prof(4, s);
    const int64_t nh = n * heads;
    hipLaunchKernelGGL(map_pool_kernel, dim3((unsigned)cdiv64(nh, 4)), dim3(256), (size_t)4 * T * sizeof(float), s, (const bf16_t*)d_kv, d_q,
                       (bf16_t*)d_out, nh, T, W, heads, hd);
    MQ_CHECK_LAUNCH("map_pool");
    return MQ_OK;
}

int mq_embed_tokens(const int32_t* d_ids, const int32_t* d_cu, int64_t nseq, const float* tok, const float* pos,
                    const float* type0, const float* g, const float* b, float* d_x, void* d_xb, int W, int vocab,
                    float eps, hipStream_t s, int last_pos) {
    MQ_CHECK_ARG(W % 4 == 0 && W <= 64 * 4 * MAXC, "embed_tokens: W=%d unsupported", W);
    if (nseq <= 0) return MQ_OK;
    MqProfScope prof(3, s);
    if (g)
        MQ_DISPATCH_CH(W, hipLaunchKernelGGL((embed_tokens_kernel<true, CH>), dim3((unsigned)nseq), dim3(256), 0, s, d_ids, d_cu,
                                             tok, pos, type0, g, b, d_x, (bf16_t*)d_xb, W, vocab, eps, last_pos));
    else
        MQ_DISPATCH_CH(W, hipLaunchKernelGGL((embed_tokens_kernel<false, CH>), dim3((unsigned)nseq), dim3(256), 0, s, d_ids, d_cu,
                                             tok, pos, type0, g, b, d_x, (bf16_t*)d_xb, W, vocab, eps, last_pos));
    MQ_CHECK_LAUNCH("embed_tokens");
    return MQ_OK;
}

// ---- rotary position embedding on the Q and K columns of the QKV buffer, in place (NewModel encoders: stella / gte-*-en-v1.5) -------
// grid = sequences; a thread owns 8 consecutive dims d .. d+7 (< hd/2) of one (row, q|k, head) and their partners d + hd/2 ..:
// two 16-byte loads, fp32 rotation with accurate sincosf (angles reach hundreds of radians), two 16-byte stores.
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* __restrict__ qkv, const int32_t* __restrict__ cu, int fixed_len, int Wa, int heads,
                                                   int hs, int hd, const float* __restrict__ inv_freq) {
    const int row0 = cu ? cu[blockIdx.x] : blockIdx.x * fixed_len;
    const int len = cu ? cu[blockIdx.x + 1] - row0 : fixed_len;
    const int half = hd >> 1, chunks = half >> 3;           // 16-byte chunks per half head
    const int per_row = 2 * heads * chunks;
    for (int it = threadIdx.x; it < len * per_row; it += 256) {
        const int t = it / per_row, r = it - t * per_row;
        const int qk = r / (heads * chunks), r2 = r - qk * heads * chunks;
        const int h = r2 / chunks, c = r2 - h * chunks;
        bf16_t* p = qkv + (int64_t)(row0 + t) * (3 * Wa) + qk * Wa + h * hs + c * 8;
        uint4 lo = *(const uint4*)p, hi = *(const uint4*)(p + half);
        uint32_t* l = (uint32_t*)&lo;
        uint32_t* u = (uint32_t*)&hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x1[2] = {bf16_to_f32((bf16_t)(l[e] & 0xffff)), bf16_to_f32((bf16_t)(l[e] >> 16))};
            float x2[2] = {bf16_to_f32((bf16_t)(u[e] & 0xffff)), bf16_to_f32((bf16_t)(u[e] >> 16))};
            float y1[2], y2[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float sn, cs;
                sincosf((float)t * inv_freq[c * 8 + e * 2 + k], &sn, &cs);
                y1[k] = x1[k] * cs - x2[k] * sn;
                y2[k] = x2[k] * cs + x1[k] * sn;
            }
            l[e] = pack_bf16x2(y1[0], y1[1]);
            u[e] = pack_bf16x2(y2[0], y2[1]);
        }
        *(uint4*)p = lo;
        *(uint4*)(p + half) = hi;
    }
}

int mq_rope(void* d_qkv, const int32_t* d_cu, int64_t nseq, int fixed_len, int Wa, int heads, const float* d_inv_freq, hipStream_t s) {
    const int hs = Wa / heads;
    MQ_CHECK_ARG(hs * heads == Wa && hs % 16 == 0, "rope: head width %d must be a multiple of 16", hs);
    if (nseq <= 0) return MQ_OK;
    MqProfScope prof(3, s);
    hipLaunchKernelGGL(rope_kernel, dim3((unsigned)nseq), dim3(256), 0, s, (bf16_t*)d_qkv, d_cu, fixed_len, Wa, heads, hs, hs, d_inv_freq);
    MQ_CHECK_LAUNCH("rope");
    return MQ_OK;
}

// ---- gated MLP: buf [rows, 2F] = (up | gate) from the fc1 GEMM -> buf[:, :F] = up * act(gate), in place (row stride stays 2F) -----
__device__ __forceinline__ float glu_act(float g, int act) { return act == MQ_ACT_SILU ? silu(g) : act == MQ_ACT_QUICKGELU ? quick_gelu(g) : gelu_erf(g); }

__global__ __launch_bounds__(256) void glu_kernel(bf16_t* __restrict__ buf, int64_t rows, int F, int act) {
    const int chunks = F >> 3;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * chunks; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / chunks;
        const int c = (int)(i - r * chunks);
        bf16_t* p = buf + r * (2 * (int64_t)F) + c * 8;
        uint4 up = *(const uint4*)p;
        const uint4 gt = *(const uint4*)(p + F);
        uint32_t* a = (uint32_t*)&up;
        const uint32_t* g = (const uint32_t*)&gt;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float u0 = bf16_to_f32((bf16_t)(a[e] & 0xffff)), u1 = bf16_to_f32((bf16_t)(a[e] >> 16));
            const float g0 = bf16_to_f32((bf16_t)(g[e] & 0xffff)), g1 = bf16_to_f32((bf16_t)(g[e] >> 16));
            a[e] = pack_bf16x2(u0 * glu_act(g0, act), u1 * glu_act(g1, act));
        }
        *(uint4*)p = up;
    }
}

int mq_glu_il_rows(void* d_buf, int64_t rows, int F, int act, hipStream_t s);
int mq_glu(void* d_buf, int64_t rows, int F, int act, hipStream_t s, int interleaved) {
    MQ_CHECK_ARG(F % 8 == 0 && (!interleaved || F % 16 == 0), "glu: F=%d must be a multiple of 8 (16 interleaved)", F);
    if (rows <= 0) return MQ_OK;
    MqProfScope prof(3, s);
    const int64_t blocks = cdiv64(rows * (F >> 3), 256);
    if (interleaved) {
        // (up, gate) interleaved 16 by 16 — MQ_EPI_GLU's weight layout behind a GEMM without the gated epilogue (the skinny kernels): the product is
        // written compactly, in place; one wave per row with the whole row in registers before anything is written (glu_ln_kernel, MODE 3)
        MQ_CHECK_ARG(F <= 4096, "glu (interleaved): F=%d > 4096", F);
        return mq_glu_il_rows(d_buf, rows, F, act, s);
    }
    hipLaunchKernelGGL(glu_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, s, (bf16_t*)d_buf, rows, F, act);
    MQ_CHECK_LAUNCH("glu");
    return MQ_OK;
}

// ---- gated MLP with the LayerNorm of the EVA02 blocks behind the gate (timm SwiGLU with norm_layer): buf [rows, 2F] = (up | gate) ->
// buf[:, :F] = LN(up * act(gate)) * g + b, in place (row stride stays 2F).  One wave per row, the row's products stay in registers (fp32) between the
// two passes of the statistics; mean / variance over the first Ft columns (F = Ft zero-padded to a multiple of 64: the padded products are exactly 0,
// g = b = 0 there, so the padding stays 0 for fc2).  CH = 16-byte chunks per lane: F <= 512 * CH.
// MODE 0: (up | gate) halves -> LN(product).  MODE 1: (up, gate) interleaved 16 by 16 (MQ_EPI_GLU's weight layout behind a GEMM WITHOUT the gated
// epilogue: the skinny kernels) -> LN(product), written compactly.  MODE 2: the product is already there (the GEMM's MQ_EPI_GLU epilogue formed it) ->
// LN only.  MODE 3: interleaved -> product, NO LayerNorm (towers without mlp.norm).  Row stride 2F in every mode.
template <int CH, int MODE = 0>
__global__ __launch_bounds__(256) void glu_ln_kernel(bf16_t* __restrict__ buf, int64_t rows, int F, int Ft, int act, const float* __restrict__ g,
                                                     const float* __restrict__ b, float eps) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    bf16_t* p = buf + row * (2 * (int64_t)F);
    float v[CH][8];
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int col = (lane + j * 64) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
        if (col < F) {
            const int src = (MODE == 1 || MODE == 3) ? 32 * (col >> 4) + (col & 15) : col;     // (col & 15) is 0 or 8
            const uint4 up = *(const uint4*)(p + src);
            const uint32_t* a = (const uint32_t*)&up;
            if (MODE == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[j][2 * e] = bf16_to_f32((bf16_t)(a[e] & 0xffff));
                    v[j][2 * e + 1] = bf16_to_f32((bf16_t)(a[e] >> 16));
                }
            } else {
                const uint4 gt = *(const uint4*)(p + src + (MODE == 0 ? F : 16));
                const uint32_t* q = (const uint32_t*)&gt;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[j][2 * e] = bf16_to_f32((bf16_t)(a[e] & 0xffff)) * glu_act(bf16_to_f32((bf16_t)(q[e] & 0xffff)), act);
                    v[j][2 * e + 1] = bf16_to_f32((bf16_t)(a[e] >> 16)) * glu_act(bf16_to_f32((bf16_t)(q[e] >> 16)), act);
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (col + e < Ft) s1 += v[j][e];
        }
    }
    if (MODE == 3) {   // (every source of the row is in this wave's registers: the compaction may overwrite them now)
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int col = (lane + j * 64) * 8;
            if (col < F) {
                uint4 o;
                o.x = pack_bf16x2(v[j][0], v[j][1]); o.y = pack_bf16x2(v[j][2], v[j][3]); o.z = pack_bf16x2(v[j][4], v[j][5]); o.w = pack_bf16x2(v[j][6], v[j][7]);
                *(uint4*)(p + col) = o;
            }
        }
        return;
    }
    const float mean = wave_sum(s1) / (float)Ft;
    float s2 = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int col = (lane + j * 64) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (col + e < Ft) { const float d = v[j][e] - mean; s2 += d * d; }
    }
    const float rstd = rsqrtf(wave_sum(s2) / (float)Ft + eps);
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int col = (lane + j * 64) * 8;
        if (col < F) {
            const f32x4 g0 = *(const f32x4*)(g + col), g1 = *(const f32x4*)(g + col + 4), b0 = *(const f32x4*)(b + col), b1 = *(const f32x4*)(b + col + 4);
            uint4 o;
            o.x = pack_bf16x2((v[j][0] - mean) * rstd * g0[0] + b0[0], (v[j][1] - mean) * rstd * g0[1] + b0[1]);
            o.y = pack_bf16x2((v[j][2] - mean) * rstd * g0[2] + b0[2], (v[j][3] - mean) * rstd * g0[3] + b0[3]);
            o.z = pack_bf16x2((v[j][4] - mean) * rstd * g1[0] + b1[0], (v[j][5] - mean) * rstd * g1[1] + b1[1]);
            o.w = pack_bf16x2((v[j][6] - mean) * rstd * g1[2] + b1[2], (v[j][7] - mean) * rstd * g1[3] + b1[3]);
            *(uint4*)(p + col) = o;
        }
    }
}

int mq_glu_ln(void* d_buf, int64_t rows, int F, int Ft, int act, const float* g, const float* b, float eps, hipStream_t s, int mode = 0) {
    MQ_CHECK_ARG(F % 8 == 0 && F >= 8 && F <= 4096 && Ft >= 1 && Ft <= F && ((g && b) || mode == 3), "glu_ln: F=%d (a multiple of 8, <= 4096) / Ft=%d unsupported", F, Ft);
    MQ_CHECK_ARG(mode >= 0 && mode <= 3 && (mode == 0 || mode == 2 || F % 16 == 0), "glu_ln: mode %d / F=%d", mode, F);
    if (rows <= 0) return MQ_OK;
    MqProfScope prof(1, s);
    const unsigned grid = (unsigned)cdiv64(rows, 4);
#define MQ_GL(C, MD) hipLaunchKernelGGL((glu_ln_kernel<C, MD>), dim3(grid), dim3(256), 0, s, (bf16_t*)d_buf, rows, F, Ft, act, g, b, eps)
#define MQ_GLC(MD) do { if (F <= 1024) MQ_GL(2, MD); else if (F <= 2048) MQ_GL(4, MD); else if (F <= 3072) MQ_GL(6, MD); else MQ_GL(8, MD); } while (0)
    if (mode == 0) MQ_GLC(0); else if (mode == 1) MQ_GLC(1); else if (mode == 2) MQ_GLC(2); else MQ_GLC(3);
#undef MQ_GLC
#undef MQ_GL
    MQ_CHECK_LAUNCH("glu_ln");
    return MQ_OK;
}
// the un-normalised interleaved product (mq_glu's il form): one wave per row
int mq_glu_il_rows(void* d_buf, int64_t rows, int F, int act, hipStream_t s) { return mq_glu_ln(d_buf, rows, F, F, act, nullptr, nullptr, 0.f, s, 3); }

// ---- 2-D rotary position embedding of the EVA02 vision towers (timm RotaryEmbeddingCat + apply_rot_embed_cat) on the Q and K columns of the QKV
// buffer, in place.  table: fp32 [T - prefix][2][hs] = (cos | sin) per rotated position, the same for every head; the first `prefix` rows of every
// T-row sequence (the class token) are left alone.  A thread owns 8 consecutive dims = 4 interleaved pairs of one (row, q | k, head):
// (y[2i], y[2i+1]) = (x[2i] cos[2i] - x[2i+1] sin[2i], x[2i+1] cos[2i+1] + x[2i] sin[2i+1])   — x * cos + rot(x) * sin with rot = (-x_odd, x_even).
__global__ __launch_bounds__(256) void rope_table_kernel(bf16_t* __restrict__ qkv, int64_t rows, int T, int prefix, int Wa, int heads, int hs,
                                                         const float* __restrict__ table) {
    const int chunks = hs >> 3;
    const int per_row = 2 * heads * chunks;
    for (int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x; it < rows * per_row; it += (int64_t)gridDim.x * 256) {
        const int64_t row = it / per_row;
        const int r = (int)(it - row * per_row);
        const int t = (int)(row % T);
        if (t < prefix) continue;
        const int qk = r / (heads * chunks), r2 = r - qk * heads * chunks;
        const int h = r2 / chunks, c = r2 - h * chunks;
        bf16_t* p = qkv + row * (3 * (int64_t)Wa) + qk * Wa + h * hs + c * 8;
        const float* cs = table + (int64_t)(t - prefix) * (2 * hs) + c * 8;
        const f32x4 c0 = *(const f32x4*)cs, c1 = *(const f32x4*)(cs + 4), s0 = *(const f32x4*)(cs + hs), s1 = *(const f32x4*)(cs + hs + 4);
        uint4 v = *(const uint4*)p;
        uint32_t* w = (uint32_t*)&v;
        const float cosv[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
        const float sinv[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xe = bf16_to_f32((bf16_t)(w[e] & 0xffff)), xo = bf16_to_f32((bf16_t)(w[e] >> 16));
            w[e] = pack_bf16x2(xe * cosv[2 * e] - xo * sinv[2 * e], xo * cosv[2 * e + 1] + xe * sinv[2 * e + 1]);
        }
        *(uint4*)p = v;
    }
}

int mq_rope_table(void* d_qkv, int64_t rows, int T, int prefix, int Wa, int heads, const float* d_table, hipStream_t s) {
    const int hs = Wa / heads;
    MQ_CHECK_ARG(hs * heads == Wa && hs % 8 == 0 && T >= 1 && prefix >= 0 && prefix <= T && rows % T == 0 && d_table,
                 "rope_table: head width %d must be a multiple of 8, rows (%ld) a multiple of the sequence length %d", hs, (long)rows, T);
    if (rows <= 0 || prefix == T) return MQ_OK;
    MqProfScope prof(3, s);
    const int64_t blocks = cdiv64(rows * 2 * heads * (hs >> 3), 256);
    hipLaunchKernelGGL(rope_table_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, s, (bf16_t*)d_qkv, rows, T, prefix, Wa, heads, hs,
                       d_table);
    MQ_CHECK_LAUNCH("rope_table");
    return MQ_OK;
}

int mq_pool(const float* d_x, const int32_t* d_cu, int64_t nseq, float* d_out, int W, int pool, int normalize,
            hipStream_t s) {
    MQ_CHECK_ARG(W <= 2048, "pool: W=%d unsupported", W);
    if (nseq <= 0) return MQ_OK;
    MqProfScope prof(4, s);
    hipLaunchKernelGGL(pool_kernel, dim3((unsigned)nseq), dim3(256), 0, s, d_x, d_cu, d_out, W, pool, normalize);
    MQ_CHECK_LAUNCH("pool");
    return MQ_OK;
}

int mq_last_rows(const int32_t* d_cu, int32_t* d_rows, int64_t nseq, hipStream_t s) {
    if (nseq <= 0) return MQ_OK;
    hipLaunchKernelGGL(last_rows_kernel, dim3((unsigned)cdiv64(nseq, 256)), dim3(256), 0, s, d_cu, d_rows, (int)nseq);
    MQ_CHECK_LAUNCH("last_rows");
    return MQ_OK;
}

// row gather / scatter through an int32 row index (the pooled-rows-only last encoder block, towers.hip)
int mq_move_rows(void* d_sparse, const int32_t* d_idx, void* d_dense, int64_t n, int64_t row_bytes, bool scatter, hipStream_t s) {
    MQ_CHECK_ARG(row_bytes % 16 == 0 && row_bytes / 16 < (1 << 30), "move_rows: row_bytes=%ld must be a multiple of 16", (long)row_bytes);
    if (n <= 0) return MQ_OK;
    MqProfScope prof(3, s);
    const int chunks = (int)(row_bytes / 16);
    const int64_t blocks = cdiv64(n * chunks, 256);
    const unsigned grid = (unsigned)(blocks < 8192 ? blocks : 8192);
    if (scatter) hipLaunchKernelGGL(move_rows_kernel<true>, dim3(grid), dim3(256), 0, s, (uint4*)d_sparse, d_idx, (uint4*)d_dense, n, chunks);
    else hipLaunchKernelGGL(move_rows_kernel<false>, dim3(grid), dim3(256), 0, s, (uint4*)d_sparse, d_idx, (uint4*)d_dense, n, chunks);
    MQ_CHECK_LAUNCH("move_rows");
    return MQ_OK;
}

int mq_cls_rows(int32_t* d_rows, int64_t n, int T, hipStream_t s) {
    if (n <= 0) return MQ_OK;
    hipLaunchKernelGGL(cls_rows_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, s, d_rows, (int)n, T);
    MQ_CHECK_LAUNCH("cls_rows");
    return MQ_OK;
}
