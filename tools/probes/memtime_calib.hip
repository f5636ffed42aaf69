// s_memtime calibration: one wave spins until s_memtime has advanced by N ticks; hipEvents give the wall time.
// ticks / second = the s_memtime rate (compare with the shader clock the SQ counters report) — tools/probes/gemm_trace.py
// turns the phase-trace tick sums into cycles with it.     hipcc --offload-arch=gfx950 -O2 memtime_calib.hip -o memtime_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(unsigned long long ticks, unsigned long long* out, int mfma_load) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long t1 = t0;
    float acc = threadIdx.x;
    while (t1 - t0 < ticks) {
        if (mfma_load) for (int i = 0; i < 64; ++i) acc = acc * 1.0001f + 0.5f;
        t1 = __builtin_amdgcn_s_memtime();
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)acc; }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int load = 0; load < 2; ++load)
        for (unsigned long long n : {1000000ULL, 10000000ULL, 100000000ULL}) {
            const int grid = load ? 2048 : 1;
            spin<<<grid, 256>>>(1000, d, load); hipDeviceSynchronize();
            hipEventRecord(a); spin<<<grid, 256>>>(n, d, load); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("grid %4d valu_load %d: %llu ticks in %.3f ms -> %.1f MHz\n", grid, load, h[0], ms, h[0] / (ms * 1e3));
        }
    return 0;
}
