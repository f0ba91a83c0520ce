// What does ONE wave per SIMD lose in the exact-fp32 stage loop against bare MFMAs (a first probe said 11 %: an artifact of run-time operand indices)?  Three variants of the same loop
// (G = 2 tiles per group, 16 groups per stage, 256 MFMAs per stage, 4 waves):
//   K1: NO LDS reads; the MFMAs take their A operands from a register array loaded once (varying registers per MFMA, like the real loop)
//   K2: the product's LDS reads are issued (into a ring nobody multiplies: kept alive by a final sum), the MFMAs use the K1 array
//   K3: the product: MFMAs multiply what the reads return
#include "../vip-nerf_amd/csrc/vipnerf_bf16n.h"
#include <cstdio>
using namespace vn;

template <int MODE>
__global__ __launch_bounds__(256) void k_why(float *out, int stages) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 64 * 256; i += 256) lds[i] = 1e-3f * (float)(i & 255);
    __syncthreads();
    constexpr int G = 2, D = 2, NT = 16, NG = 16, NB = 3;
    f32q bin[8][2], fixed[NB][G][2], fr[NB][G][2];
    floatx4 acc[16];
    const float *base = lds + lane * 4;
#pragma unroll
    for (int s = 0; s < 8; ++s) { bin[s][0].v = *(const floatx4 *)(base + s * 256); bin[s][1].v = *(const floatx4 *)(base + (s + 8) * 256); }
#pragma unroll
    for (int g = 0; g < NB; ++g)
#pragma unroll
        for (int tt = 0; tt < G; ++tt)
#pragma unroll
            for (int i = 0; i < 2; ++i) { fixed[g][tt][i].v = *(const floatx4 *)(base + (16 + g * 4 + tt * 2 + i) * 256); fr[g][tt][i].v = (floatx4)(0.f); }
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = (floatx4)(0.f);
    for (int st4 = 0; st4 < stages; st4 += 4) {
#pragma unroll
    for (int sj = 0; sj < 4; ++sj) {
        constexpr int dummy = 0; (void)dummy;
        const int ks0 = 2 * sj;                      // compile-time after unrolling (a run-time index into the operand registers would cost selects)
        if (MODE >= 2) {
#pragma unroll
            for (int g = 0; g < D; ++g)
#pragma unroll
                for (int tt = 0; tt < G; ++tt)
#pragma unroll
                    for (int i = 0; i < 2; ++i) fr[g][tt][i] = *(const f32q *)(base + ((g * G + tt) * 2 + i) * CHUNK_F);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const bool reads = MODE >= 2 && g + D < NG;
            if (reads) {
#pragma unroll
                for (int tt = 0; tt < G; ++tt)
#pragma unroll
                    for (int i = 0; i < 2; ++i) fr[(g + D) % NB][tt][i] = *(const f32q *)(base + (((g + D) * G + tt) * 2 + i) * CHUNK_F);
            }
#pragma unroll
            for (int tt = 0; tt < G; ++tt) {
                const int lin = g * G + tt, ks = lin / NT, t = lin % NT;
                if (MODE == 3) acc[t] = mfma_split<2>(fr[g % NB][tt], bin[ks0 + ks], acc[t]);
                else acc[t] = mfma_split<2>(fixed[g % NB][tt], bin[ks0 + ks], acc[t]);
            }
            if (reads) {
#pragma unroll
                for (int i = 0; i < G * 2; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                __builtin_amdgcn_sched_group_barrier(0x8, G * 8 - G * 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][3];
#pragma unroll
    for (int g = 0; g < NB; ++g) s += fr[g][0][0].v[0] + fr[g][1][1].v[3];
    out[blockIdx.x * 256 + tid] = s;
}
template <int MODE>
static void run(float *out, const char *what) {
    const int stages = 4000;
    (void)hipFuncSetAttribute((const void *)k_why<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_why<MODE>, dim3(256), dim3(256), 128 * 1024, 0, out, stages);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double tf = 256.0 * 4 * stages * 256.0 * 2048.0 / ms * 1e-9;
    printf("%-86s %8.3f ms  %.3f of 157.3\n", what, ms, tf / 157.3);
}
int main() {
    float *out; (void)hipMalloc(&out, 256 * 256 * 4);
    run<1>(out, "K1: no LDS reads, operands from a register array loaded once");
    run<2>(out, "K2: the product's LDS reads issued, MFMAs multiply the K1 array");
    run<3>(out, "K3: the product (MFMAs multiply what the reads return)");
    return 0;
}
