// Measurement support (bench.py `roofline.peak_sustained_measured`): what the matrix pipes of THIS chip sustain, at the clock it holds under
// full MFMA load, with register-resident operands — no LDS, no memory.  The GEMM kernels' roofline fraction is quoted against the guide's
// dense peak (2.5 PF bf16); boxes of the pool differ by ~5-9 % in the clock they hold under load (1.86-2.03 GHz measured), so a second
// figure, the fraction of this burn's rate on the same box in the same run, is what is comparable across boxes.
// Not on the product path: nothing in the towers calls it.
#include <mutex>
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 probe_bf16x8;
union ProbeFrag { probe_bf16x8 v; unsigned u[4]; };

// 16 independent 16x16x32 accumulators per wave (the tile GEMMs' instruction), operand bits pseudo-random in +-[0.5, 1) (toggle rate sets the
// power the chip clocks against); runs until `ticks` shader cycles (s_memtime) have passed.  out[2w] = ticks seen, out[2w + 1] = MFMAs issued.
__global__ __launch_bounds__(256) void mfma_burn_kernel(unsigned long long ticks, unsigned long long* out, unsigned seed) {
    ProbeFrag a[4], b[4];
    unsigned s = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            s = s * 1664525u + 1013904223u; a[i].u[j] = (s & 0x807f807fu) | 0x3f003f00u;
            s = s * 1664525u + 1013904223u; b[i].u[j] = (s & 0x807f807fu) | 0x3f003f00u;
        }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), t1 = t0, n = 0;
    while (t1 - t0 < ticks) {
        for (int rep = 0; rep < 16; ++rep)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i].v, b[j].v, acc[i * 4 + j], 0, 0, 0);
        n += 256;
        t1 = __builtin_amdgcn_s_memtime();
    }
    float sum = 0.f;
    for (int i = 0; i < 16; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = n + (sum == 12345.678f);
    }
}

}  // namespace

// Runs the burn for about `target_ms` on 256 CUs x 4 SIMDs x 2 waves (the GEMMs' occupancy) on `stream` and waits for it.
// *tflops = bf16 MFMA rate sustained (dense FLOPs / wall time between two events), *shader_mhz = s_memtime ticks / wall time = the shader
// clock held during the burn.  d_scratch: >= 32 KiB of device memory.
extern "C" int mq_probe_mfma_peak(double target_ms, void* d_scratch, int64_t scratch_bytes, double* tflops, double* shader_mhz, void* stream) {
    constexpr int BLOCKS = 512, WAVES = BLOCKS * 4;
    MQ_CHECK_ARG(d_scratch && scratch_bytes >= (int64_t)WAVES * 16 && tflops && shader_mhz, "mq_probe_mfma_peak: needs %d bytes of scratch", WAVES * 16);
    MQ_CHECK_ARG(target_ms > 0.0 && target_ms <= 2000.0, "mq_probe_mfma_peak: target_ms out of range");
    hipStream_t s = (hipStream_t)stream;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { mq_set_error("mq_probe_mfma_peak: hipEventCreate failed"); return MQ_ERR_HIP; }
    unsigned long long* d = (unsigned long long*)d_scratch;
    hipLaunchKernelGGL(mfma_burn_kernel, dim3(BLOCKS), dim3(256), 0, s, 200000ull, d, 1u);          // warm: clocks up, code resident
    hipEventRecord(e0, s);
    hipLaunchKernelGGL(mfma_burn_kernel, dim3(BLOCKS), dim3(256), 0, s, (unsigned long long)(target_ms * 2.4e6), d, 7u);
    hipEventRecord(e1, s);
    hipError_t err = hipEventSynchronize(e1);
    float ms = 0.f;
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[WAVES * 2];
    static std::mutex mu;
    std::lock_guard<std::mutex> g(mu);
    if (err == hipSuccess) err = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    if (err != hipSuccess || ms <= 0.f) { mq_set_error("mq_probe_mfma_peak: %s", hipGetErrorString(err)); return MQ_ERR_HIP; }
    double n = 0.0, t = 0.0;
    for (int w = 0; w < WAVES; ++w) { t += (double)h[2 * w]; n += (double)h[2 * w + 1]; }
    *tflops = n * 16384.0 / (ms * 1e-3) / 1e12;
    *shader_mhz = t / WAVES / (ms * 1e3);
    return MQ_OK;
}
