// Probe: what does ds_read_b64_tr_b16 deliver to each lane?  LDS holds u16 values = their own element index.
// Each lane passes the address of 4 consecutive u16 (row-major [16][..] tile, lane -> row lane%16, 4-col group lane/16).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(uint16_t* out, int row_stride_elems) {
    __shared__ uint16_t lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int lane = threadIdx.x;
    const int row = lane & 15, cg = lane >> 4;
    const uint16_t* p = &lds[row * row_stride_elems + cg * 4];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = (uint16_t)v[e];
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int stride : {64, 16}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, stride);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("row_stride=%d  (lane: 4 values as (row,col) of the [16][stride] tile)\n", stride);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int e = 0; e < 4; ++e) printf(" (%2d,%2d)", h[l * 4 + e] / stride, h[l * 4 + e] % stride);
            printf("\n");
        }
    }
    return 0;
}
