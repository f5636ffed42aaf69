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
// Structure = gemm_bf16.hip: (32*MT)x128 tile, 4 wave64s as 2x2, BK = 128 elements (= the same 128-B LDS rows),
// 2 stages, LDS-DMA issues interleaved with the MFMAs, XCD-aware tile map, operands fed swapped so a lane owns 4
// consecutive n.  A lane's fragment is 32 consecutive k-bytes (two ds_read_b128): 16-B chunk c of row r is stored at
// c ^ f(r) with f below — chosen so that each 16-lane ds_read_b128 group (8 rows reading chunk 2g+j, 8 rows reading
// chunk 2g+2+j) touches 16 distinct 16-B slots of the 256-B bank row.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

typedef int i32x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BN = 128, BK = 128;          // BK in fp8 elements == bytes
constexpr int W_TILE_BYTES = BN * BK;      // 16 KiB
constexpr int UNIT_SCALE = 0x7F7F7F7F;     // e8m0 127 = 2^0 in every byte

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// chunk swizzle: u = (row >> 1) & 7;  f = u for u in {0,1,6,7}, u ^ 2 for u in {2,3,4,5}
__device__ __forceinline__ int fswz(int row) {
    const int u = (row >> 1) & 7;
    return u ^ ((((u >> 1) ^ (u >> 2)) & 1) << 1);
}

__device__ __forceinline__ float clamp448(float v) { return fminf(fmaxf(v, -448.f), 448.f); }

// FLAGS: MQ_EPI_BIAS / GELU / QUICKGELU / RESIDUAL / OUT_F32 / OUT_FP8; ROWSCALE: a_scale is per row (else scalar)
template <int FLAGS, int MT, bool ROWSCALE>
__global__ __launch_bounds__(256, 2) void gemm_fp8_kernel(
    const uint8_t* __restrict__ A, int64_t lda, const uint8_t* __restrict__ Wt, int64_t ldw,
    const float* __restrict__ a_scale, const float* __restrict__ w_scale,
    const float* __restrict__ bias, const float* residual, void* out, int64_t ldc,
    const float* __restrict__ out_scale, float* amax_out,
    int M, int N, int K, int tiles_n, int num_tiles) {
    constexpr int BM = 32 * MT;
    constexpr int A_TILE_BYTES = BM * BK;
    constexpr int STAGE_BYTES = A_TILE_BYTES + W_TILE_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int bid = blockIdx.x;
    const int q = num_tiles >> 3, r = num_tiles & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    const int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g = lane >> 4;

    // ---- staging (8 rows x 128 B per LDS-DMA; lane -> row = base + lane/8, physical chunk = lane%8) ------------
    const int srow = lane >> 3;
    const uint8_t* a_src[MT];
    const uint8_t* w_src[4];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int row = wave * (8 * MT) + i * 8 + srow;
        int gm = m0 + row; gm = gm < M ? gm : M - 1;
        a_src[i] = A + (int64_t)gm * lda + ((lane & 7) ^ fswz(row)) * 16;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 32 + i * 8 + srow;
        int gn = n0 + row; gn = gn < N ? gn : N - 1;
        w_src[i] = Wt + (int64_t)gn * ldw + ((lane & 7) ^ fswz(row)) * 16;
    }
    auto stage = [&](int buf, int kt) {
        char* sa = smem + buf * STAGE_BYTES + wave * (8 * MT * 128);
        char* sw = smem + buf * STAGE_BYTES + A_TILE_BYTES + wave * (32 * 128);
#pragma unroll
        for (int i = 0; i < MT; ++i) glds16(a_src[i] + (int64_t)kt * BK, sa + i * (8 * 128));
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(w_src[i] + (int64_t)kt * BK, sw + i * (8 * 128));
    };

    // ---- fragment offsets: lane (l15, g) reads logical chunks 2g and 2g+1 of row base16 + l15 ---------------------
    const int fr = fswz(l15);  // sub-tile bases are multiples of 16 rows
    const int c0 = ((2 * g) ^ fr) << 4, c1 = ((2 * g + 1) ^ fr) << 4;
    int a_off[MT], w_off[4];
#pragma unroll
    for (int t = 0; t < MT; ++t) a_off[t] = (wm * (16 * MT) + t * 16 + l15) * 128;
#pragma unroll
    for (int t = 0; t < 4; ++t) w_off[t] = (wn * 64 + t * 16 + l15) * 128;

    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto load_frag = [&](const char* base) -> i32x8 {
        const uint4 lo = *(const uint4*)(base + c0);
        const uint4 hi = *(const uint4*)(base + c1);
        return i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    };

    auto kstep = [&](int kt, auto prefetch_tag) {
        constexpr bool PREFETCH = decltype(prefetch_tag)::value;
        const char* sa = smem + (kt & 1) * STAGE_BYTES;
        const char* sw = sa + A_TILE_BYTES;
        char* na = smem + ((kt + 1) & 1) * STAGE_BYTES + wave * (8 * MT * 128);
        char* nw = smem + ((kt + 1) & 1) * STAGE_BYTES + A_TILE_BYTES + wave * (32 * 128);
        const int64_t koff = (int64_t)(kt + 1) * BK;
        i32x8 af[MT], wf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) wf[t] = load_frag(sw + w_off[t]);
#pragma unroll
        for (int t = 0; t < MT; ++t) af[t] = load_frag(sa + a_off[t]);
        constexpr int NL = MT + 4;
        constexpr int NM = 4 * MT;
        constexpr int GAP = NM / NL > 0 ? NM / NL : 1;
        int issued = 0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[nt], af[mt], acc[mt][nt], 0, 0, 0, UNIT_SCALE, 0,
                                                                              UNIT_SCALE);
                const int done = mt * 4 + nt + 1;
                if (PREFETCH && done % GAP == 0 && issued < NL) {
                    if (issued < MT) glds16(a_src[issued] + koff, na + issued * (8 * 128));
                    else glds16(w_src[issued - MT] + koff, nw + (issued - MT) * (8 * 128));
                    ++issued;
                }
            }
        if (PREFETCH) {
#pragma unroll
            for (; issued < NL; ++issued) {
                if (issued < MT) glds16(a_src[issued] + koff, na + issued * (8 * 128));
                else glds16(w_src[issued - MT] + koff, nw + (issued - MT) * (8 * 128));
            }
        }
    };

    const int nk = K / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk - 1; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        kstep(kt, std::true_type{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    kstep(nk - 1, std::false_type{});

    // ---- epilogue: lane owns out[m][n .. n+3]; dequantise with a_scale[m] * w_scale[n] -------------------------------
    const float a_scalar = ROWSCALE ? 1.f : a_scale[0];
    const float inv_out = (FLAGS & MQ_EPI_OUT_FP8) ? 1.0f / out_scale[0] : 1.f;
    float amax = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + wm * (16 * MT) + mt * 16 + l15;
        if (m >= M) continue;
        const float sa = ROWSCALE ? a_scale[m] : a_scalar;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + wn * 64 + nt * 16 + g * 4;
            if (n >= N) continue;
            const f32x4 sw4 = *(const f32x4*)(w_scale + n);
            f32x4 v = acc[mt][nt];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= sa * sw4[e];
            if (FLAGS & MQ_EPI_BIAS) v += *(const f32x4*)(bias + n);
            if (FLAGS & MQ_EPI_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            }
            if (FLAGS & MQ_EPI_QUICKGELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
            }
            const int64_t o = (int64_t)m * ldc + n;
            if (FLAGS & MQ_EPI_RESIDUAL) v += *(const f32x4*)(residual + o);
            if (FLAGS & MQ_EPI_OUT_F32) {
                *(f32x4*)((float*)out + o) = v;
            } else if (FLAGS & MQ_EPI_OUT_FP8) {
                amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                int w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[0] * inv_out), clamp448(v[1] * inv_out), 0, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32(clamp448(v[2] * inv_out), clamp448(v[3] * inv_out), w, true);
                *(int*)((uint8_t*)out + o) = w;
            } else {
                uint2 p;
                p.x = pack_bf16x2(v[0], v[1]);
                p.y = pack_bf16x2(v[2], v[3]);
                *(uint2*)((bf16_t*)out + o) = p;
            }
        }
    }
    if ((FLAGS & MQ_EPI_OUT_FP8) && amax_out) {
        amax = wave_max(amax);
        if (lane == 0) atomicMax((int*)amax_out, __float_as_int(amax));  // amax >= 0: int order == float order
    }
}

constexpr int RESIDENT_SLOTS = 512;
int choose_mt(int M, int N) {
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

template <int FLAGS, int MT, bool ROWSCALE>
int launch_fp8_mt(const Fp8Args& a, hipStream_t s) {
    constexpr int BM = 32 * MT;
    constexpr int LDS = 2 * (BM * BK + W_TILE_BYTES);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_fp8_kernel<FLAGS, MT, ROWSCALE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (e != hipSuccess) { mq_set_error("mq_gemm_fp8: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MQ_ERR_HIP; }
        attr_set = true;
    }
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const int num_tiles = tiles_m * tiles_n;
    hipLaunchKernelGGL((gemm_fp8_kernel<FLAGS, MT, ROWSCALE>), dim3(num_tiles), dim3(256), LDS, s, (const uint8_t*)a.A, a.lda,
                       (const uint8_t*)a.W, a.ldw, a.a_scale, a.w_scale, a.bias, a.residual, a.out, a.ldc, a.out_scale, a.amax, a.M, a.N,
                       a.K, tiles_n, num_tiles);
    MQ_CHECK_LAUNCH("mq_gemm_fp8");
    return MQ_OK;
}

template <int FLAGS, bool ROWSCALE>
int launch_fp8(const Fp8Args& a, int force_mt, hipStream_t s) {
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

int mq_gemm_fp8_force_mt = 0;  // set through mq_tune("gemm_mt", v) (shared knob, see gemm_bf16.hip)

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
