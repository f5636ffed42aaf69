// Weighted combination of sub-embeddings into one vector per group:
//   out[g,:] = mean_i( w_i * E[row_i,:] )  over the group's terms,  then / ||.||_2 when asked.
// Reference (numpy float64 on the host, one group at a time):
//   add_documents  src/marqo/core/inference/tensor_fields_container.py:346-365  (normalise unconditionally)
//   search         src/marqo/tensor_search/tensor_search.py:1913-1984           (normalise only when the norm is > 0)
// HBM-bound row pass: one workgroup per group, a thread owns columns tid, tid+256, ...; products, the mean and the norm are
// accumulated in fp64 like the reference (np.asarray(list_of_python_floats) is float64); the result is stored as fp32.
#include "common.h"

namespace {

constexpr int MAXC = 8;  // D <= 2048

__global__ __launch_bounds__(256) void weighted_combine_kernel(const float* __restrict__ emb, int64_t ld, const int32_t* __restrict__ rows,
                                                               const float* __restrict__ weights, const int32_t* __restrict__ cu,
                                                               float* __restrict__ out, int D, int mode) {
    __shared__ double red[4];
    const int g = blockIdx.x;
    const int t0 = cu[g], nt = cu[g + 1] - t0;
    double acc[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) acc[i] = 0.0;
    for (int t = 0; t < nt; ++t) {
        const int64_t r = rows ? (int64_t)rows[t0 + t] : (int64_t)(t0 + t);
        const double w = (double)weights[t0 + t];
        const float* e = emb + r * ld;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = threadIdx.x + i * 256;
            if (c < D) acc[i] += (double)e[c] * w;
        }
    }
    double ss = 0.0;
    const double inv_n = nt > 0 ? 1.0 / (double)nt : 0.0;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        acc[i] = nt > 0 ? acc[i] / (double)nt : 0.0;  // np.mean: sum / count
        ss += acc[i] * acc[i];
    }
    (void)inv_n;
    double scale = 1.0;
    if (mode != MQ_COMBINE_RAW) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
        __syncthreads();
        const double norm = sqrt(red[0] + red[1] + red[2] + red[3]);
        // MQ_COMBINE_NORMALIZE: vector / norm whatever the norm (0/0 = NaN exactly like numpy);  _IF_NONZERO: the search-side guard
        if (mode == MQ_COMBINE_NORMALIZE || norm > 0.0) scale = 1.0 / norm;
        if (mode == MQ_COMBINE_NORMALIZE && norm == 0.0) scale = __longlong_as_double(0x7ff8000000000000LL);
    }
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = threadIdx.x + i * 256;
        if (c < D) out[(int64_t)g * D + c] = (float)(acc[i] * scale);
    }
}

}  // namespace

extern "C" int mq_weighted_combine(const float* d_emb, int64_t ld, const int32_t* d_rows, const float* d_weights,
                                   const int32_t* d_cu_terms, int64_t n_groups, int32_t D, int32_t mode, float* d_out,
                                   void* stream) {
    MQ_CHECK_ARG(D >= 1 && D <= 256 * MAXC, "mq_weighted_combine: D=%d unsupported (max %d)", D, 256 * MAXC);
    MQ_CHECK_ARG(ld >= D, "mq_weighted_combine: ld=%ld < D=%d", (long)ld, D);
    MQ_CHECK_ARG(mode == MQ_COMBINE_RAW || mode == MQ_COMBINE_NORMALIZE || mode == MQ_COMBINE_NORMALIZE_IF_NONZERO,
                 "mq_weighted_combine: bad mode %d", mode);
    if (n_groups <= 0) return MQ_OK;
    MQ_CHECK_ARG(d_emb && d_weights && d_cu_terms && d_out, "mq_weighted_combine: null pointer");
    MQ_CHECK_ARG(n_groups < (1LL << 31), "mq_weighted_combine: too many groups");
    hipStream_t s = (hipStream_t)stream;
    MqProfScope prof(4, s);
    hipLaunchKernelGGL(weighted_combine_kernel, dim3((unsigned)n_groups), dim3(256), 0, s, d_emb, ld, d_rows, d_weights, d_cu_terms,
                       d_out, D, mode);
    MQ_CHECK_LAUNCH("mq_weighted_combine");
    return MQ_OK;
}
