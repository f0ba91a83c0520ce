// What the exact-fp32 MLP kernels' stage loop (gemm_stage_bf<16, 2, 2> on f32q fragments: 256 v_mfma_f32_16x16x4_f32 + 64 ds_read_b128 per wave and
// 64 KiB stage) sustains with NOTHING around it -- no weight DMA, no barriers, no epilogues, no stores: W waves per SIMD looping over one
// resident LDS stage.  Separates the loop's own efficiency (LDS reads among the MFMAs) from the stage-boundary costs of the real kernels.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I vip-nerf_amd/csrc tools/f32_stage_probe.hip -o tools/bin/f32_stage_probe
#include "../vip-nerf_amd/csrc/vipnerf_bf16n.h"
#include <cstdio>
using namespace vn;

template <int WAVES, bool READS>
__global__ __launch_bounds__(64 * WAVES) void k_probe(float *out, int stages) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 64 * 256; i += 64 * WAVES) lds[i] = 1e-3f * (float)(i & 255);
    __syncthreads();
    f32q bin[8][2];
    floatx4 acc[16];
#pragma unroll
    for (int s = 0; s < 8; ++s) { bin[s][0].v = (floatx4)(1e-3f * s + lane); bin[s][1].v = (floatx4)(2e-3f * s + lane); }
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = (floatx4)(0.f);
    NoStream none;
    for (int st = 0; st < stages; ++st) {
        if (READS) gemm_stage_bf<16, 2, 2>(lds, lane, acc, bin, 2 * (st & 3), none);
        else {
#pragma unroll
            for (int g = 0; g < 256; ++g) acc[g & 15] = __builtin_amdgcn_mfma_f32_16x16x4f32(bin[g & 7][0].v[g & 3], bin[(g + 1) & 7][1].v[g & 3], acc[g & 15], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int WAVES, bool READS>
static void run(const char *what, float *out, int stages) {
    hipFuncSetAttribute((const void *)k_probe<WAVES, READS>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256;                       // one workgroup per CU (128 KiB of LDS)
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_probe<WAVES, READS>), dim3(grid), dim3(64 * WAVES), 128 * 1024, 0, out, stages);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)grid * WAVES * stages * 256.0 * 2048.0;
        if (rep == 2) printf("%-46s %2d waves per CU: %8.3f ms  %7.1f TFLOP/s = %.3f of 157.3\n", what, WAVES, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 157.3);
    }
}

int main() {
    float *out; hipMalloc(&out, 256 * 512 * sizeof(float));
    const int stages = 4000;
    run<4, false>("MFMAs only (16 accumulators round robin)", out, stages);
    run<8, false>("MFMAs only (16 accumulators round robin)", out, stages);
    run<4, true>("gemm_stage_bf<16,2,2>: 256 MFMA + 64 ds_read_b128", out, stages);
    run<8, true>("gemm_stage_bf<16,2,2>: 256 MFMA + 64 ds_read_b128", out, stages);
    return 0;
}
