#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short short4_ __attribute__((ext_vector_type(4)));
__global__ void k(uint16_t *out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    // lane l reads address 8*l bytes (4 halves): element index 4*l
    short4_ v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_ *)(lds + 4 * threadIdx.x));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}
int main() {
    uint16_t *d; hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    uint16_t h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", h[l * 4 + j]); printf("\n"); }
    return 0;
}
