// Variants of the exact-fp32 stage loop (256 v_mfma_f32_16x16x4_f32 + 64 ds_read_b128 per wave and 64 KiB stage) with nothing around them:
// which placement of the LDS reads among the MFMAs keeps the pipe full with ONE wave per SIMD (the state the younger wave of a SIMD is in for
// ~40 % of a stage: its partner finishes first and waits at the barrier) and with two.
//   G groups of tiles, D groups of read lead, MODE: 0 = one read behind each of the group's first MFMAs (the product's placement), 1 = all reads
//   ahead of the group's MFMAs, 2 = one read per two MFMAs, 3 = reads behind the group's last MFMA; CONT: the read pipeline runs across stage
//   boundaries (no per-stage prologue with the pipe empty)
#include "../vip-nerf_amd/csrc/vipnerf_bf16n.h"
#include <cstdio>
using namespace vn;

template <int G, int D, int MODE, bool CONT, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_loop(float *out, int stages) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 64 * 256; i += 64 * WAVES) lds[i] = 1e-3f * (float)(i & 255);
    __syncthreads();
    constexpr int NT = 16, NKS = 2, NG = NKS * NT / G, NB = 4;       // ring of 4 group buffers (divides NG)
    static_assert(D < NB && NG % NB == 0, "ring");
    f32q bin[8][2];
    floatx4 acc[16];
#pragma unroll
    for (int s = 0; s < 8; ++s) { bin[s][0].v = (floatx4)(1e-3f * s + lane); bin[s][1].v = (floatx4)(2e-3f * s + lane); }
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = (floatx4)(0.f);
    f32q fr[NB][G][2];
    const float *base = lds + lane * 4;
    auto rd = [&](int g) {       // group g (mod NG) of the resident stage -> ring slot g % NB
#pragma unroll
        for (int tt = 0; tt < G; ++tt)
#pragma unroll
            for (int i = 0; i < 2; ++i) fr[g % NB][tt][i] = *(const f32q *)(base + (((g % NG) * G + tt) * 2 + i) * CHUNK_F);
    };
    if (CONT) {
#pragma unroll
        for (int g = 0; g < D; ++g) rd(g);
    }
    for (int st = 0; st < stages; ++st) {
        if (!CONT) {
#pragma unroll
            for (int g = 0; g < D; ++g) rd(g);
            __builtin_amdgcn_sched_barrier(0);
        }
        const int ks0 = 2 * (st & 3);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            constexpr int R = G * 2, M = G * 8;
            const bool reads = CONT || g + D < NG;
            if (reads && MODE != 3) rd(g + D);
#pragma unroll
            for (int tt = 0; tt < G; ++tt) {
                const int lin = g * G + tt, ks = lin / NT, t = lin % NT;
                acc[t] = mfma_split<2>(fr[g % NB][tt], bin[ks0 + ks], acc[t]);
            }
            if (reads && MODE == 3) rd(g + D);
            if (reads) {
                if (MODE == 0) {
#pragma unroll
                    for (int i = 0; i < R; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                    __builtin_amdgcn_sched_group_barrier(0x8, M - R, 0);
                } else if (MODE == 1) {
                    __builtin_amdgcn_sched_group_barrier(0x100, R, 0); __builtin_amdgcn_sched_group_barrier(0x8, M, 0);
                } else if (MODE == 2) {
#pragma unroll
                    for (int i = 0; i < R; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                    __builtin_amdgcn_sched_group_barrier(0x8, M - 2 * R, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x8, M, 0); __builtin_amdgcn_sched_group_barrier(0x100, R, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][3];
#pragma unroll
    for (int g = 0; g < NB; ++g) s += fr[g][0][0].v[0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int G, int D, int MODE, bool CONT, int WAVES>
static void run1(float *out, int stages) {
    (void)hipFuncSetAttribute((const void *)k_loop<G, D, MODE, CONT, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_loop<G, D, MODE, CONT, WAVES>), dim3(256), dim3(64 * WAVES), 128 * 1024, 0, out, stages);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double tf = 256.0 * WAVES * stages * 256.0 * 2048.0 / ms * 1e-9;
    printf("G %d  D %d  mode %d  %s  %d waves/SIMD: %8.3f ms  %.3f of 157.3\n", G, D, MODE, CONT ? "continuous" : "per-stage  ", WAVES / 4, ms, tf / 157.3);
}
template <int G, int D, int MODE, bool CONT>
static void run(float *out) { run1<G, D, MODE, CONT, 4>(out, 2000); run1<G, D, MODE, CONT, 8>(out, 2000); }

int main() {
    float *out; (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    run<2, 2, 0, false>(out);     // the product
    run<2, 2, 0, true>(out);
    run<2, 2, 1, true>(out);
    run<2, 2, 2, true>(out);
    run<2, 2, 3, true>(out);
    run<2, 1, 0, true>(out);
    run<2, 1, 3, true>(out);
    run<2, 3, 0, true>(out);
    run<4, 1, 0, true>(out);
    run<4, 1, 2, true>(out);
    run<4, 1, 3, true>(out);
    run<1, 3, 0, true>(out);
    run<1, 2, 3, true>(out);
    return 0;
}
