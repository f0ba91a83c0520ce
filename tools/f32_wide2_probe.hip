// PROBE: the alternative tiling of DESIGN 7 (1) -- ONE wave per SIMD owning TWO 16-point tiles (32 points), so that every ds_read_b128 of a weight
// fragment feeds 8 v_mfma_f32_16x16x4_f32 instead of 4 (half the LDS bytes per MFMA), 128 accumulators + 128 operand registers per wave (needs the
// 512-register budget of one wave per SIMD).  What would its stage loop sustain?  4 waves, 64 KiB stage resident in LDS, per stage and wave 512 MFMAs +
// 64 ds_read_b128; optionally a 16-piece global_load_lds burst per wave and stage (the weight DMA a real kernel would issue: nobody to cover it) and a
// workgroup barrier per stage.
#include "../vip-nerf_amd/csrc/vipnerf_bf16n.h"
#include <cstdio>
using namespace vn;

template <bool DMA, bool BARRIER>
__global__ __launch_bounds__(256) void k_w2(float *out, const float *wsrc, int stages) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 64 * 256; i += 256) lds[i] = 1e-3f * (float)(i & 255);
    __syncthreads();
    f32q bin[2][8][2];
    floatx4 acc[2][16];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
#pragma unroll
        for (int s = 0; s < 8; ++s) { bin[pt][s][0].v = (floatx4)(1e-3f * s + lane + pt); bin[pt][s][1].v = (floatx4)(2e-3f * s + lane - pt); }
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[pt][t] = (floatx4)(0.f);
    }
    constexpr int G = 2, D = 2, NT = 16, NG = 16, NB = 3;
    for (int st4 = 0; st4 < stages; st4 += 4) {
#pragma unroll
    for (int sj = 0; sj < 4; ++sj) {
        const int st = st4 + sj;
        const float *base = lds + (sj & 1) * 64 * 256 + lane * 4;
        const int ks0 = 2 * sj;                      // compile-time after unrolling, as in the kernels
        f32q fr[NB][G][2];
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
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int pt = 0; pt < 2; ++pt)
                            acc[pt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fr[g % NB][tt][i].v[e], bin[pt][ks0 + ks][i].v[e], acc[pt][t], 0, 0, 0);
            }
            if (g + D < NG) {
#pragma unroll
                for (int i = 0; i < G * 2; ++i) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                __builtin_amdgcn_sched_group_barrier(0x8, G * 16 - G * 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (DMA && g == 0) {      // this wave's quarter of the NEXT stage's weights (the other LDS buffer)
                glds_run<16>(wsrc + (size_t)((st + 1) & 63) * 16384 + wave * 16 * 256 + lane * 4, lds + ((st + 1) & 1) * 64 * 256 + wave * 16 * 256);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (BARRIER) __builtin_amdgcn_s_barrier();
    }
    }
    float s = 0.f;
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int t = 0; t < 16; ++t) s += acc[pt][t][0] + acc[pt][t][3];
    out[blockIdx.x * 256 + tid] = s;
}
template <bool DMA, bool BARRIER>
static void run(float *out, const float *w, const char *what) {
    const int stages = 2000;
    (void)hipFuncSetAttribute((const void *)k_w2<DMA, BARRIER>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_w2<DMA, BARRIER>), dim3(256), dim3(256), 128 * 1024, 0, out, w, stages);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double tf = 256.0 * 4 * stages * 512.0 * 2048.0 / ms * 1e-9;
    printf("%-70s %8.3f ms  %.3f of 157.3\n", what, ms, tf / 157.3);
}
int main() {
    float *out, *w; (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&w, 64 * 16384 * 4); (void)hipMemset(w, 0, 64 * 16384 * 4);
    run<false, false>(out, w, "one wave per SIMD x two point tiles: loop only");
    run<false, true>(out, w, "  + barrier per stage");
    run<true, true>(out, w, "  + 16-piece weight DMA per wave and stage + barrier");
    return 0;
}
