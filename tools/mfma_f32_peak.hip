// What v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 sustain on an MI355X when NOTHING else happens: W waves per SIMD, each
// issuing NACC independent accumulator chains back to back, no memory traffic.  Prints TFLOP/s against the 157.3 nominal peak.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f32_peak.hip -o tools/bin/mfma_f32_peak && tools/bin/mfma_f32_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float *out, int iters, float a, float b) {
    floatx16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (floatx16)(0.f);
    for (int it = 0; it < iters; it += 16 / NACC) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int WG>
__global__ __launch_bounds__(WG) void k16(float *out, int iters, float a, float b) {
    floatx4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (floatx4)(0.f);
    for (int it = 0; it < iters; it += 16 / NACC) {          // 16 MFMAs per trip whatever NACC (loop overhead out of the picture)
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
static void run(const char *name, F launch, double flop_per_launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double tf = 5 * flop_per_launch / (ms * 1e-3) / 1e12;
    printf("%-44s %8.1f TFLOP/s  = %.3f of 157.3\n", name, tf, tf / 157.3);
}
int main() {
    float *out;
    hipMalloc(&out, 256 * 8 * 512 * sizeof(float));
    const int iters = 20000;
    const int grid = 256 * 8;                 // 8 workgroups per CU's worth of work queued
    run("32x32x2, 1 wave/SIMD, 16 accumulators", [&] { hipLaunchKernelGGL(k32<16>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); },
        (double)grid * 4 * iters * 16 * 4096.0);
    run("32x32x2, 1 wave/SIMD, 4 accumulators", [&] { hipLaunchKernelGGL(k32<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); },
        (double)grid * 4 * iters * 4 * 4096.0);
    run("32x32x2, 1 wave/SIMD, 2 accumulators", [&] { hipLaunchKernelGGL(k32<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); },
        (double)grid * 4 * iters * 2 * 4096.0);
    run("32x32x2, 1 wave/SIMD, 1 accumulator", [&] { hipLaunchKernelGGL(k32<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); },
        (double)grid * 4 * iters * 1 * 4096.0);
    run("16x16x4, 1 wave/SIMD, 16 accumulators", [&] { hipLaunchKernelGGL((k16<16, 256>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); },
        (double)grid * 4 * iters * 16 * 2048.0);
    run("16x16x4, 2 waves/SIMD, 16 accumulators", [&] { hipLaunchKernelGGL((k16<16, 512>), dim3(grid), dim3(512), 0, 0, out, iters, 1.f, 2.f); },
        (double)grid * 8 * iters * 16 * 2048.0);
    run("16x16x4, 2 waves/SIMD, 4 accumulators", [&] { hipLaunchKernelGGL((k16<4, 512>), dim3(grid), dim3(512), 0, 0, out, iters, 1.f, 2.f); },
        (double)grid * 8 * iters * 4 * 2048.0);
    run("16x16x4, 1 wave/SIMD, 2 accumulators", [&] { hipLaunchKernelGGL((k16<2, 256>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); },
        (double)grid * 4 * iters * 2 * 2048.0);
    run("16x16x4, 1 wave/SIMD, 1 accumulator", [&] { hipLaunchKernelGGL((k16<1, 256>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f); },
        (double)grid * 4 * iters * 1 * 2048.0);
    run("16x16x4, 2 waves/SIMD, 2 accumulators", [&] { hipLaunchKernelGGL((k16<2, 512>), dim3(grid), dim3(512), 0, 0, out, iters, 1.f, 2.f); },
        (double)grid * 8 * iters * 2 * 2048.0);
    run("16x16x4, 2 waves/SIMD, 1 accumulator", [&] { hipLaunchKernelGGL((k16<1, 512>), dim3(grid), dim3(512), 0, 0, out, iters, 1.f, 2.f); },
        (double)grid * 8 * iters * 1 * 2048.0);
    return 0;
}
