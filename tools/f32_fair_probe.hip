// Does THROTTLING the older wave of a SIMD make the two waves share the MFMA pipe for the whole stage?  The product's exact-fp32 stage loop with a
// workgroup barrier per stage (8 waves): the arbiter favours the older wave of each SIMD, which finishes a stage's 256 MFMAs at ~60 % of the stage
// and waits, the younger then runs ALONE at the single-wave rate (0.88 of the pipe against 0.945 for two: tools/f32_loop_probe.hip).
// THR: s_nop 15 instructions the waves 0..3 execute behind each of their MFMAs (none: the product).
#include "../vip-nerf_amd/csrc/vipnerf_bf16n.h"
#include <cstdio>
using namespace vn;

template <int THR>
__device__ __forceinline__ void stage_loop(const float *lds, int lane, floatx4 (&acc)[16], const f32q (&bin)[8][2], int ks0) {
    constexpr int G = 2, D = 2, NT = 16, NG = 16, NB = 3;
    f32q fr[NB][G][2];
    const float *base = lds + lane * 4;
#pragma unroll
    for (int g = 0; g < D; ++g)
#pragma unroll
        for (int tt = 0; tt < G; ++tt)
#pragma unroll
            for (int i = 0; i < 2; ++i) fr[g][tt][i] = *(const f32q *)(base + ((g * G + tt) * 2 + i) * CHUNK_F);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + D < NG) {
#pragma unroll
            for (int tt = 0; tt < G; ++tt)
#pragma unroll
                for (int i = 0; i < 2; ++i) fr[(g + D) % NB][tt][i] = *(const f32q *)(base + (((g + D) * G + tt) * 2 + i) * CHUNK_F);
        }
#pragma unroll
        for (int tt = 0; tt < G; ++tt) {
            const int lin = g * G + tt, ks = lin / NT, t = lin % NT;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[g % NB][tt][i].v[e], bin[ks0 + ks][i].v[e], acc[t], 0, 0, 0);
#pragma unroll
                    for (int n = 0; n < THR; ++n) asm volatile("s_nop 15");
                }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int THR, bool BARRIER>
__global__ __launch_bounds__(512) void k_fair(float *out, int stages) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 64 * 256; i += 512) lds[i] = 1e-3f * (float)(i & 255);
    __syncthreads();
    f32q bin[8][2];
    floatx4 acc[16];
#pragma unroll
    for (int s = 0; s < 8; ++s) { bin[s][0].v = (floatx4)(1e-3f * s + lane); bin[s][1].v = (floatx4)(2e-3f * s + lane); }
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = (floatx4)(0.f);
    if (wave < 4) {
        for (int st = 0; st < stages; ++st) { stage_loop<THR>(lds, lane, acc, bin, 2 * (st & 3)); if (BARRIER) __builtin_amdgcn_s_barrier(); }
    } else {
        for (int st = 0; st < stages; ++st) { stage_loop<0>(lds, lane, acc, bin, 2 * (st & 3)); if (BARRIER) __builtin_amdgcn_s_barrier(); }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int THR, bool BARRIER>
static void run(float *out) {
    const int stages = 2000;
    (void)hipFuncSetAttribute((const void *)k_fair<THR, BARRIER>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_fair<THR, BARRIER>), dim3(256), dim3(512), 128 * 1024, 0, out, stages);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double tf = 256.0 * 8 * stages * 256.0 * 2048.0 / ms * 1e-9;
    printf("older waves: %d x s_nop 15 per MFMA, %s: %8.3f ms  %.3f of 157.3\n", THR, BARRIER ? "barrier per stage" : "no barriers      ", ms, tf / 157.3);
}

int main() {
    float *out; (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    run<0, false>(out); run<0, true>(out); run<1, true>(out); run<2, true>(out); run<3, true>(out); run<4, true>(out);
    return 0;
}
