// Where do the 4.6 % go that four fp32 tile stores per wave and stage cost the exact-fp32 stage loop (f32_stack_probe.hip: S 0.965 -> 0.916)?
// The same loop (gemm_stage_bf<16, 2, 2>, StreamOlder, 64 KiB stages, 8 waves), stores only, one variant at a time:
//   0: as the kernel (older waves behind group 6, younger behind 12, non-temporal 16 x 16 tiles of a row-major [P][256] array: 16 segments of 64 B per instruction)
//   1: only the OLDER wave of each SIMD stores          2: only the YOUNGER one
//   3: dense tiles (every instruction writes 1 KiB contiguous)
//   4: eight 8-byte stores instead of four 16-byte ones (same bytes)
//   5: the same bytes to LDS (ds_write_b128) instead of global memory
//   6: younger waves at s_setprio 1          7: younger waves at s_setprio 1 in the first half of every stage only
//   8: all eight waves store behind group 1          9: all behind group 14
//  10: both row halves of a k-step pair in adjacent instructions (0 already does)  -- no stores at all, for the box's S figure
//  11: stores through a scalar base (saddr form, one address VGPR)
#include "../vip-nerf_amd/csrc/vipnerf_bf16n.h"
#include <cstdio>
using namespace vn;
typedef BnPlan<2> PL;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int VAR>
struct Hook {
    float *dst; unsigned voff; int q, wave, s0, lane; f32q (*bin)[2]; float *lds_extra;      // dst: wave-uniform base of the workgroup's 128 rows
    template <int g, int NG> static constexpr bool active() {
        return (VAR == 10 || VAR >= 12) ? false : VAR == 8 ? g == 1 : VAR == 9 ? g == 14 : VAR == 7 ? (g == 0 || g == 8 || g == 6 || g == 12) : (g == 6 || g == 12);
    }
    template <int g, int NG> __device__ __forceinline__ void at() const {
        if (VAR == 7) {
            if (g == 0 && wave >= 4) asm volatile("s_setprio 1");
            if (g == 8 && wave >= 4) asm volatile("s_setprio 0");
            if (g == 0 || g == 8) return;
        }
        const bool mine = (VAR == 8 || VAR == 9) ? true : (g == 6) == (wave < 4);
        if (!mine) return;
        if (VAR == 1 && wave >= 4) return;
        if (VAR == 2 && wave < 4) return;
#pragma unroll
        for (int s = s0; s < s0 + 2; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const floatx4 &v = bin[s][u].v;
                f4 val = {v[0], v[1], v[2], v[3]};
                const int T = 2 * s + u;
                if (VAR == 3) __builtin_nontemporal_store(val, (f4 *)((char *)dst + (size_t)((wave * 16 + T) * 1024 + 16 * lane)));
                else if (VAR == 4) {
                    f2 a = {v[0], v[1]}, b = {v[2], v[3]};
                    const unsigned off = voff + 64 * T;
                    asm volatile("global_store_dwordx2 %0, %1, %2 nt" :: "v"(off), "v"(a), "s"(dst) : "memory");
                    asm volatile("global_store_dwordx2 %0, %1, %2 offset:8 nt" :: "v"(off), "v"(b), "s"(dst) : "memory");
                } else if (VAR == 5) *(f4 *)(lds_extra + (wave * 4 + (T & 3)) * 256 + 4 * lane) = val;
                else if (VAR == 11) {
                    const unsigned off = voff + 64 * T;
                    asm volatile("global_store_dwordx4 %0, %1, %2 nt" :: "v"(off), "v"(val), "s"(dst) : "memory");
                } else __builtin_nontemporal_store(val, (f4 *)((char *)dst + (size_t)(voff + 64 * T)));
            }
    }
};

__device__ __forceinline__ void store4(float *dst, unsigned voff, const f32q (*bin)[2], int s0) {
#pragma unroll
    for (int s = s0; s < s0 + 2; ++s)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const floatx4 &v = bin[s][u].v;
            f4 val = {v[0], v[1], v[2], v[3]};
            __builtin_nontemporal_store(val, (f4 *)((char *)dst + (size_t)(voff + 64 * (2 * s + u))));
        }
}
template <int VAR>
__global__ __launch_bounds__(512) void k_store(float *out, const float *packed, float *sink, int tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
    for (int i = tid; i < 2 * PL::STAGE_F; i += 512) lds[i] = 1e-3f * (float)(i & 255);
    __syncthreads();
    f32q bin[8][2];
    floatx4 acc[16];
#pragma unroll
    for (int s = 0; s < 8; ++s) { bin[s][0].v = *(const floatx4 *)(lds + (s * 64 + lane) * 4); bin[s][1].v = *(const floatx4 *)(lds + ((s + 8) * 64 + lane) * 4); }
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = (floatx4)(0.f);
    const unsigned voff = (unsigned)(((wave * 16 + j) * 256 + 4 * q) * 4);
    typename StreamOlder<PL>::type ws;
    ws.start(packed, 32, lds, lane, wave, tiles);
    if (VAR == 6 && wave >= 4) asm volatile("s_setprio 1");
    for (int tile = 0; tile < tiles; ++tile) {
#pragma unroll 1
        for (int it = 0; it < 8; ++it) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float *st = VAR == 10 ? ws.wait() : ws.template wait<4>();
                Hook<VAR> h{sink + ((size_t)((tile * 8 + it) & 15) * 256 + blockIdx.x) * (128 * 256), voff, q, wave, 2 * jj, lane, bin, lds + 2 * PL::STAGE_F};
                // 12: younger waves store BEFORE their first MFMA of the stage (while the older wave has the pipe), older behind group 6 (Hook<0>)
                // 13: younger before the first MFMA, older AFTER their last one (when they would wait at the barrier)      14: older after the last, younger behind 12
                if ((VAR == 12 || VAR == 13) && wave >= 4) { store4(h.dst, voff, bin, 2 * jj); __builtin_amdgcn_sched_barrier(0); }
                if (VAR == 12) { Hook<1> h1{h.dst, voff, q, wave, 2 * jj, lane, bin, h.lds_extra}; gemm_stage_bf<16, 2, 2>(st, lane, acc, bin, 2 * jj, ws, h1); }
                else if (VAR == 14) { Hook<2> h2{h.dst, voff, q, wave, 2 * jj, lane, bin, h.lds_extra}; gemm_stage_bf<16, 2, 2>(st, lane, acc, bin, 2 * jj, ws, h2); }
                else gemm_stage_bf<16, 2, 2>(st, lane, acc, bin, 2 * jj, ws, h);
                if ((VAR == 13 || VAR == 14) && wave < 4) { __builtin_amdgcn_sched_barrier(0); store4(h.dst, voff, bin, 2 * jj); __builtin_amdgcn_sched_barrier(0); }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][3];
    if (VAR == 5) s += lds[2 * PL::STAGE_F + tid];
    out[blockIdx.x * 512 + tid] = s;
}
template <int VAR>
static void run(float *out, const float *packed, float *sink, const char *what) {
    const int tiles = 24;
    const int shm = 2 * PL::STAGE_F * 4 + (VAR == 5 ? 8 * 4 * 256 * 4 : 0);
    (void)hipFuncSetAttribute((const void *)k_store<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_store<VAR>, dim3(256), dim3(512), shm, 0, out, packed, sink, tiles);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double tf = 256.0 * 8 * tiles * 32 * 256.0 * 2048.0 / best * 1e-9;
    printf("%-100s %8.3f ms  %.3f of 157.3\n", what, best, tf / 157.3);
}
int main() {
    float *out, *packed, *sink;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&packed, (size_t)32 * PL::STAGE_F * 4); (void)hipMemset(packed, 0, (size_t)32 * PL::STAGE_F * 4);
    (void)hipMalloc(&sink, (size_t)16 * 256 * 128 * 256 * 4);
    run<10>(out, packed, sink, "no stores (the weight stream, counted wait, barrier per stage)");
    run<0>(out, packed, sink, "0: four 16-byte tile stores per wave and stage, as the kernel (older behind group 6, younger behind 12)");
    run<1>(out, packed, sink, "1: only the older wave of each SIMD stores");
    run<2>(out, packed, sink, "2: only the younger wave stores");
    run<3>(out, packed, sink, "3: dense tiles (1 KiB contiguous per instruction)");
    run<4>(out, packed, sink, "4: eight 8-byte stores instead");
    run<5>(out, packed, sink, "5: the same bytes to LDS (ds_write_b128)");
    run<6>(out, packed, sink, "6: younger waves at s_setprio 1");
    run<7>(out, packed, sink, "7: younger waves at s_setprio 1 for groups 0..7 of every stage");
    run<8>(out, packed, sink, "8: all waves store behind group 1");
    run<9>(out, packed, sink, "9: all waves store behind group 14");
    run<11>(out, packed, sink, "11: scalar base + 32-bit offset (one address VGPR)");
    run<12>(out, packed, sink, "12: younger waves store BEFORE their first MFMA of the stage, older behind group 6");
    run<13>(out, packed, sink, "13: younger before their first MFMA, older AFTER their last");
    run<14>(out, packed, sink, "14: older after their last MFMA, younger behind group 12");
    return 0;
}
