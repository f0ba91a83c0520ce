// The exact-fp32 MLP kernels' stage loop rebuilt from the PRODUCT'S OWN pieces (gemm_stage_bf<16, 2, 2> on f32q fragments, WStreamT / StreamOlder, 64 KiB
// stages), adding one cost at a time -- every operand index compile-time, as in the kernels (a first version of these probes indexed the operand
// registers with a run-time k-step offset and measured the compiler's selects, not the loop: 0.885 / 0.945 where the loop gives 0.97 / 0.98):
//   L: the loop over a resident LDS stage, 8 waves (two per SIMD)            B: + one workgroup barrier per stage
//   S: the real weight stream instead (DMA of every stage by the four older waves, counted wait, barrier)
//   H: + what a stage carries besides MFMAs in the data-gradient kernel: 4 fp32 tile stores per wave and the ReLU-bit conversion of 2 operand k-steps
#include "../vip-nerf_amd/csrc/vipnerf_bf16n.h"
#include <cstdio>
using namespace vn;
typedef BnPlan<2> PL;

__device__ __forceinline__ float keep_if_bit(float x, unsigned word, int bit) {
    int sel = __builtin_amdgcn_sbfe((int)word, (unsigned)bit, 1u);
    asm("" : "+v"(sel));
    return __uint_as_float(__float_as_uint(x) & (unsigned)sel);
}
template <bool ON, bool CONV = true, bool STORES = true, bool PLAIN = false>
struct Hook {
    float *dst; int64_t p; int q, wave, s0; f32q (*bin)[2]; const floatx4 *xr; unsigned m0, m1;
    template <int g, int NG> static constexpr bool active() { return ON && ((CONV && g == 2) || (STORES && (g == 6 || g == 12))); }
    template <int g, int NG> __device__ __forceinline__ void at() const {
        if (CONV && g == 2) {
#pragma unroll
            for (int s = s0 + 2; s < s0 + 4; ++s)
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) bin[s & 7][u].v[r] = keep_if_bit(xr[(2 * s + u) & 15][r], (2 * s + u) & 8 ? m1 : m0, 4 * ((2 * s + u) & 7) + r);
        }
        if (STORES && (g == 6 || g == 12) && (g == 6) == (wave < 4)) {
#pragma unroll
            for (int s = s0; s < s0 + 2; ++s) {
                if (PLAIN) { *(floatx4 *)(dst + (size_t)p * 256 + 32 * s + 4 * q) = bin[s][0].v; *(floatx4 *)(dst + (size_t)p * 256 + 32 * s + 16 + 4 * q) = bin[s][1].v; }
                else { store_tile16(dst, p, 256, q, 2 * s, bin[s][0].v); store_tile16(dst, p, 256, q, 2 * s + 1, bin[s][1].v); }
            }
        }
    }
};

// the same work SPREAD over the stage: one tile store behind each of four groups (waves 0..3: groups 3 5 7 9, waves 4..7: 9 11 13 15 / or all early),
// the conversion two values (4 VALU) behind each of groups 1..8
template <int SMODE>       // 1: stores spread, conversion in one burst; 2: both spread; 3: conversion spread, stores in bursts
struct HookSpread {
    float *dst; int64_t p; int q, wave, s0; f32q (*bin)[2]; const floatx4 *xr; unsigned m0, m1;
    template <int g, int NG> static constexpr bool active() { return g >= 1; }
    template <int g, int NG> __device__ __forceinline__ void at() const {
        if (SMODE == 1 ? g == 2 : (g >= 1 && g <= 8)) {
            const int lo = SMODE == 1 ? 0 : 2 * (g - 1), hi = SMODE == 1 ? 16 : 2 * g;       // elements of the 16 values of the two k-steps
#pragma unroll
            for (int e = lo; e < hi; ++e) {
                const int s = s0 + 2 + (e >> 3), u = (e >> 2) & 1, r = e & 3, t = 2 * s + u;
                bin[s & 7][u].v[r] = keep_if_bit(xr[t & 15][r], t & 8 ? m1 : m0, 4 * (t & 7) + r);
            }
        }
        if (SMODE == 3) {
            if ((g == 6 || g == 12) && (g == 6) == (wave < 4)) {
#pragma unroll
                for (int s = s0; s < s0 + 2; ++s) { store_tile16(dst, p, 256, q, 2 * s, bin[s][0].v); store_tile16(dst, p, 256, q, 2 * s + 1, bin[s][1].v); }
            }
        } else {
            const int first = wave < 4 ? 3 : 9;
            if (g >= first && g < first + 8 && ((g - first) & 1) == 0) {
                const int k = (g - first) >> 1;         // 0..3: tile 2 s0 + k
                store_tile16(dst, p, 256, q, 2 * s0 + k, bin[s0 + (k >> 1)][k & 1].v);
            }
        }
    }
};

template <int MODE>        // 0: L, 1: B, 2: S, 6 / 7 / 8: S + the spread hooks, 3: S + H, 4: S + conversion only, 5: S + stores only
__global__ __launch_bounds__(512) void k_stack(float *out, const float *packed, float *sink, int tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, j = lane & 15;
    for (int i = tid; i < 2 * PL::STAGE_F; i += 512) lds[i] = 1e-3f * (float)(i & 255);
    __syncthreads();
    f32q bin[8][2];
    floatx4 acc[16], xr[16];
#pragma unroll
    for (int s = 0; s < 8; ++s) { bin[s][0].v = *(const floatx4 *)(lds + (s * 64 + lane) * 4); bin[s][1].v = *(const floatx4 *)(lds + ((s + 8) * 64 + lane) * 4); }
#pragma unroll
    for (int t = 0; t < 16; ++t) { acc[t] = (floatx4)(0.f); xr[t] = *(const floatx4 *)(lds + ((t + 16) * 64 + lane) * 4); }
    const unsigned m0 = 0xf0f0f0ffu ^ lane, m1 = 0xffff0f0fu ^ (lane << 3);
    const int64_t p = (int64_t)blockIdx.x * 128 + wave * 16 + j;
    typename StreamOlder<PL>::type ws;
    NoStream none;
    if (MODE >= 2) ws.start(packed, 32, lds, lane, wave, tiles);
    for (int tile = 0; tile < tiles; ++tile) {
#pragma unroll 1
        for (int it = 0; it < 8; ++it) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float *st = lds + (jj & 1) * PL::STAGE_F;
                if (MODE >= 2) st = (MODE == 3 || MODE >= 5) ? ws.template wait<4>() : ws.wait();
                if (MODE >= 6 && MODE <= 8) {
                    HookSpread<MODE - 5> h{sink + (size_t)(it & 1) * 262144 * 256, p, q, wave, 2 * jj, bin, xr, m0, m1};
                    gemm_stage_bf<16, 2, 2>(st, lane, acc, bin, 2 * jj, ws, h);
                } else if (MODE >= 3) {
                    Hook<true, MODE != 5 && MODE != 9, MODE != 4, MODE == 9> h{sink + (size_t)(it & 1) * 262144 * 256, p, q, wave, 2 * jj, bin, xr, m0, m1};
                    gemm_stage_bf<16, 2, 2>(st, lane, acc, bin, 2 * jj, ws, h);
                } else if (MODE == 2) gemm_stage_bf<16, 2, 2>(st, lane, acc, bin, 2 * jj, ws);
                else gemm_stage_bf<16, 2, 2>(st, lane, acc, bin, 2 * jj, none);
                if (MODE == 1) __builtin_amdgcn_s_barrier();
            }
            if (MODE >= 3) {
#pragma unroll
                for (int t = 0; t < 16; ++t) { xr[t] = acc[t]; acc[t] = (floatx4)(0.f); }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][3] + xr[t][1];
    out[blockIdx.x * 512 + tid] = s;
}
template <int MODE>
static void run(float *out, const float *packed, float *sink, const char *what) {
    const int tiles = 24;
    (void)hipFuncSetAttribute((const void *)k_stack<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PL::STAGE_F * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_stack<MODE>, dim3(256), dim3(512), 2 * PL::STAGE_F * 4, 0, out, packed, sink, tiles);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double tf = 256.0 * 8 * tiles * 32 * 256.0 * 2048.0 / ms * 1e-9;
    printf("%-96s %8.3f ms  %.3f of 157.3\n", what, ms, tf / 157.3);
}
int main() {
    float *out, *packed, *sink;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&packed, (size_t)32 * PL::STAGE_F * 4); (void)hipMemset(packed, 0, (size_t)32 * PL::STAGE_F * 4);
    (void)hipMalloc(&sink, (size_t)2 * 262144 * 256 * 4);
    run<0>(out, packed, sink, "L: gemm_stage_bf<16,2,2> over a resident stage, two waves per SIMD");
    run<1>(out, packed, sink, "B: + a workgroup barrier per stage");
    run<2>(out, packed, sink, "S: the weight stream (DMA by the four older waves, counted wait, barrier per stage)");
    run<3>(out, packed, sink, "H: S + per stage 4 fp32 tile stores per wave and the ReLU-bit conversion of two operand k-steps");
    run<4>(out, packed, sink, "   S + the conversion only");
    run<5>(out, packed, sink, "   S + the stores only");
    run<9>(out, packed, sink, "   S + the stores only, PLAIN (temporal) stores");
    run<6>(out, packed, sink, "   H with the stores spread (one behind each of four groups), conversion in one burst");
    run<7>(out, packed, sink, "   H with both spread (conversion: 4 VALU behind each of groups 1..8)");
    run<8>(out, packed, sink, "   H with the conversion spread, stores in two-tile bursts");
    return 0;
}
