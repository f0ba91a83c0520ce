// What would a fused data-gradient + weight-gradient kernel pay to flush its per-tile weight-gradient accumulators?  Every workgroup
// (256 points) would add a 256 x 256 fp32 partial per layer into the layer's dW: 4096 workgroups x 10 layers x 256 KB = 10.7 GB of
// fp32 atomic adds per 4096-ray step, all landing on 2.6 MB of addresses.  This probe issues exactly that pattern with
// global_atomic_add_f32 (no return) and times it:   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_flush_probe.hip -o tools/bin/atomic_flush_probe
//   mode 0: one shared 10 x 256 KB target;  mode 1: one target per XCD (blockIdx % 8);  mode 2: plain stores of the same bytes to private
//   slots (the "write partials" alternative, upper bound of the store path)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(512) void k_flush(float *dst, int layers, int mode, size_t slot_stride) {
    const int tid = threadIdx.x;
    float *base = dst + (mode == 1 ? (size_t)(blockIdx.x & 7) * slot_stride : (mode == 2 ? (size_t)(blockIdx.x % 1024) * slot_stride : 0));
    for (int l = 0; l < layers; ++l) {
        float *p = base + (size_t)l * 65536;
#pragma unroll 8
        for (int i = 0; i < 128; ++i) {
            const float v = 1e-6f * (float)(tid + i);
            float *a = p + (size_t)i * 512 + tid;
            if (mode == 2) __builtin_nontemporal_store(v, a);
            else unsafeAtomicAdd(a, v);
        }
    }
}
int main() {
    const int layers = 10, wgs = 4096;
    const size_t slot = (size_t)layers * 65536;
    float *d;
    hipMalloc(&d, slot * 1024 * sizeof(float));
    hipMemset(d, 0, slot * 1024 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_flush, dim3(wgs), dim3(512), 0, 0, d, layers, mode, slot);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double gb = (double)wgs * layers * 65536 * 4 / 1e9;
            printf("mode %d (%s): %.3f ms for %.2f GB = %.2f TB/s\n", mode, mode == 0 ? "atomics, one target" : (mode == 1 ? "atomics, target per XCD" : "plain stores"), ms, gb, gb / ms);
        }
    }
    return 0;
}
