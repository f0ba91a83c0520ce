// Micro-benchmark: how many independent VALU instructions hide in the shadow of one MFMA issued by the SAME wave
// (one wave per SIMD, 4 waves per workgroup, one workgroup per CU)?  Prints cycles per MFMA for N = 0..8 VALU per MFMA
// and a few VALU kinds.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_coissue.hip -o tools/bin/mfma_valu_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int N, int KIND, int SHAPE>
__global__ __launch_bounds__(256) void k(float *out, long long *cyc, int iters) {
    floatx16 acc[4];
    floatx4 acc4[8];
    for (int i = 0; i < 4; ++i) acc[i] = (floatx16)(0.f);
    for (int i = 0; i < 8; ++i) acc4[i] = (floatx4)(0.f);
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)(float)(e); }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = threadIdx.x * 0.001f + e;
    unsigned u[8];
    for (int e = 0; e < 8; ++e) u[e] = threadIdx.x + e;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (SHAPE == 0) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
            else acc4[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[m], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const int r = (m * N + n) & 7;
                if (KIND == 0) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(u[r]));
                else if (KIND == 1) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[r]) : "v"(v[(r + 1) & 7]));
                else if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[r]) : "v"(v[r]), "v"(v[(r + 1) & 7]));
                else if (KIND == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[r]) : "v"(u[(r + 1) & 7]), "s"(0x07060302));
                else if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[r]) : "v"(v[(r + 1) & 7]));
                else if (KIND == 5) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[r]) : "v"(v[(r + 1) & 7]));
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc4[i][r];
    for (int e = 0; e < 8; ++e) s += v[e] + (float)u[e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int N, int KIND, int SHAPE>
void run(float *out, long long *cyc, const char *name) {
    const int iters = 20000, grid = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<N, KIND, SHAPE>), dim3(grid), dim3(256), 0, 0, out, cyc, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<N, KIND, SHAPE>), dim3(grid), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-10s shape %s N=%d: %.1f ns/MFMA, %.1f refclk ticks/MFMA\n", name, SHAPE ? "16x16x32" : "32x32x16", N, ms * 1e6 / (iters * 8.0), (double)c / (iters * 8.0));
}
template <int KIND, int SHAPE>
void sweep(float *out, long long *cyc, const char *name) {
    run<0, KIND, SHAPE>(out, cyc, name); run<1, KIND, SHAPE>(out, cyc, name); run<2, KIND, SHAPE>(out, cyc, name); run<3, KIND, SHAPE>(out, cyc, name);
    run<4, KIND, SHAPE>(out, cyc, name); run<5, KIND, SHAPE>(out, cyc, name); run<6, KIND, SHAPE>(out, cyc, name); run<8, KIND, SHAPE>(out, cyc, name);
}
int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    sweep<0, 0>(out, cyc, "v_and");
    sweep<1, 0>(out, cyc, "v_sub");
    sweep<2, 0>(out, cyc, "cvt_pk");
    sweep<3, 0>(out, cyc, "v_perm");
    sweep<4, 0>(out, cyc, "v_fma");
    sweep<1, 1>(out, cyc, "v_sub");
    sweep<2, 1>(out, cyc, "cvt_pk");
    return 0;
}
