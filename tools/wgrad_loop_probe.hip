// What the exact-fp32 256 x 256 weight-gradient kernel's block loop (k_wgrad256_w8: 8 waves, wave (wm, wk) owns 2 x 4 tiles of 32 x 32, per k-step 2 A + 4 B values
// from a row-major LDS tile and 8 v_mfma_f32_32x32x2_f32) sustains with NOTHING around it: LDS tile resident, no global loads, no LDS stores;
// with and without a workgroup barrier per 32-point block.
//   hipcc --offload-arch=gfx950 -O3 tools/wgrad_loop_probe.hip -o tools/bin/wgrad_loop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <bool BARRIER, bool BSUM>
__global__ __launch_bounds__(512) void k_loop(float *out, int blocks) {
    extern __shared__ float lds[];
    constexpr int Mp = 256, Kp = 256, MTW = 2, KTW = 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31, wm = wave & 3, wk = wave >> 2;
    for (int i = tid; i < 2 * 32 * 512; i += 512) lds[i] = 1e-3f * (float)(i & 1023);
    __syncthreads();
    floatx16 acc[MTW][KTW];
    float bsum[MTW] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < KTW; ++j) acc[i][j] = (floatx16)(0.f);
    for (int blk = 0; blk < blocks; ++blk) {
        const float *la = lds + (blk & 1) * 32 * 512, *lb = la + 32 * Mp;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float af[MTW], bf[KTW];
#pragma unroll
            for (int i = 0; i < MTW; ++i) af[i] = la[(2 * s + h) * Mp + 32 * (wm * MTW + i) + l31];
#pragma unroll
            for (int j = 0; j < KTW; ++j) bf[j] = lb[(2 * s + h) * Kp + 32 * (wk * KTW + j) + l31];
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                if (BSUM) bsum[i] += af[i];
#pragma unroll
                for (int j = 0; j < KTW; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
            }
        }
        if (BARRIER) __syncthreads();
    }
    float s = bsum[0] + bsum[1];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < KTW; ++j) s += acc[i][j][0] + acc[i][j][15];
    out[blockIdx.x * 512 + tid] = s;
}
template <bool BARRIER, bool BSUM>
static void run(float *out, const char *what) {
    const int blocks = 1500;
    (void)hipFuncSetAttribute((const void *)k_loop<BARRIER, BSUM>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32 * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_loop<BARRIER, BSUM>), dim3(256), dim3(512), 2 * 32 * 512 * 4, 0, out, blocks);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double tf = 256.0 * 8 * blocks * 128.0 * 4096.0 / ms * 1e-9;
    printf("%-58s %8.3f ms  %.3f of 157.3\n", what, ms, tf / 157.3);
}
int main() {
    float *out; (void)hipMalloc(&out, 256 * 512 * 4);
    run<false, false>(out, "block loop, no barrier, no bias sums");
    run<false, true>(out, "block loop, no barrier, bias sums");
    run<true, true>(out, "block loop, barrier per 32-point block, bias sums");
    return 0;
}
