// Sustained matrix-pipe rate of v_mfma_f32_16x16x32_bf16 with register-resident operands (no LDS, no memory): what the chip
// delivers at the clock it actually holds under full MFMA load, and the s_memtime tick rate during it (s_memtime = shader clock,
// so ticks / wall time = that clock).  Operand bits are pseudo-random (toggle rate matters for power).
//   hipcc --offload-arch=gfx950 -O2 mfma_peak.hip -o mfma_peak ; ./mfma_peak [waves_per_simd=2] [ms=200]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
union frag { bf16x8_t v; unsigned u[4]; };
__global__ __launch_bounds__(256) void burn(unsigned long long ticks, unsigned long long* out, unsigned seed, int zero) {
    frag a[4], b[4];
    unsigned s = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            s = s * 1664525u + 1013904223u; a[i].u[j] = zero ? 0u : ((s & 0x807f807fu) | 0x3f003f00u);   // |x| in [0.5, 1)
            s = s * 1664525u + 1013904223u; b[i].u[j] = zero ? 0u : ((s & 0x807f807fu) | 0x3f003f00u);
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
        out[2 * w] = t1 - t0; out[2 * w + 1] = n + (sum == 12345.678f);
    }
}
// the same burn with v_mfma_f32_32x32x16_bf16 (half the operand register reads per FLOP: 32 K FLOP per 2 KiB of A + B fragments against
// 16 K FLOP for the 16x16x32 form): does the chip hold a higher clock / rate with it?  4 x 2 independent 16-register accumulators.
__global__ __launch_bounds__(256) void burn32(unsigned long long ticks, unsigned long long* out, unsigned seed, int zero) {
    frag a[4], b[2];
    unsigned s = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            s = s * 1664525u + 1013904223u; a[i].u[j] = zero ? 0u : ((s & 0x807f807fu) | 0x3f003f00u);
            s = s * 1664525u + 1013904223u; b[i & 1].u[j] = zero ? 0u : ((s & 0x807f807fu) | 0x3f003f00u);
        }
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), t1 = t0, n = 0;
    while (t1 - t0 < ticks) {
        for (int rep = 0; rep < 16; ++rep)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i].v, b[j].v, acc[i * 2 + j], 0, 0, 0);
        n += 128;
        t1 = __builtin_amdgcn_s_memtime();
    }
    float sum = 0.f;
    for (int i = 0; i < 8; ++i) sum += acc[i][0] + acc[i][5] + acc[i][10] + acc[i][15];
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0; out[2 * w + 1] = n + (sum == 12345.678f);
    }
}
int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2;
    const double ms_target = argc > 2 ? atof(argv[2]) : 200.0;
    const int blocks = 256 * wps, waves = blocks * 4;
    unsigned long long* d; hipMalloc(&d, waves * 16);
    unsigned long long* h = (unsigned long long*)malloc(waves * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int zero = 0; zero < 2; ++zero) {   // 32x32x16 form: 32768 FLOP per MFMA per wave
        burn32<<<blocks, 256>>>(100000, d, 1u, zero); hipDeviceSynchronize();
        const unsigned long long ticks = (unsigned long long)(ms_target * 2.4e6);
        hipEventRecord(e0); burn32<<<blocks, 256>>>(ticks, d, 7u, zero); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d, waves * 16, hipMemcpyDeviceToHost);
        double n = 0, t = 0; for (int w = 0; w < waves; ++w) { t += h[2 * w]; n += h[2 * w + 1]; }
        printf("32x32x16 %s operands, %d waves/SIMD: %.1f ms wall, s_memtime %.0f MHz, %.0f TFLOP/s, %.2f ticks per MFMA per SIMD\n",
               zero ? "zero  " : "random", wps, ms, t / waves / (ms * 1e3), n * 32768.0 / (ms * 1e-3) / 1e12, (t / waves) / (n / waves) / wps);
    }
    for (int zero = 0; zero < 2; ++zero) {
        burn<<<blocks, 256>>>(100000, d, 1u, zero); hipDeviceSynchronize();
        const unsigned long long ticks = (unsigned long long)(ms_target * 2.4e6);
        hipEventRecord(e0); burn<<<blocks, 256>>>(ticks, d, 7u, zero); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d, waves * 16, hipMemcpyDeviceToHost);
        double n = 0, t = 0; for (int w = 0; w < waves; ++w) { t += h[2 * w]; n += h[2 * w + 1]; }
        printf("%s operands, %d waves/SIMD: %.1f ms wall, s_memtime %.0f MHz, %.0f TFLOP/s, %.2f ticks per MFMA per SIMD\n",
               zero ? "zero  " : "random", wps, ms, t / waves / (ms * 1e3), n * 16384.0 / (ms * 1e-3) / 1e12, (t / waves) / (n / waves) / wps);
    }
    return 0;
}
