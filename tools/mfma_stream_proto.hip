// Prototype: is a 2-waves-per-SIMD layout (16-point waves on v_mfma_f32_16x16x32_bf16) better at hiding the weight
// stream than the shipped 1-wave-per-SIMD layout (32-point waves on 32x32x16)?  Both kernels stream the same 64 KiB
// stages (bf16x3: hi+lo A fragments) through a double-buffered LDS ring and do 3 MFMAs per (tile, k-step) cell; no
// epilogue.  Reported: time per 36-stage "task" and MFMA utilisation.
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_stream_proto.hip -o tools/bin/mfma_stream_proto
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int STAGE_BYTES = 64 * 1024, N_STAGES = 36;

template <int WAVES>
__device__ __forceinline__ void glds_stage(const char *g, unsigned lds_base, int wave, int lane) {
    constexpr int PER_WAVE = 64 / WAVES;
    const char *src = g + (wave * PER_WAVE) * 1024 + lane * 16;
    unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (wave * PER_WAVE) * 1024);
#pragma unroll
    for (int i = 0; i < PER_WAVE; i += 4) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(src + i * 1024), "s"(dst + i * 1024) : "memory");
    }
}

// ---- 4 waves x 32 points, 32x32x16: stage = 4 k-steps x 8 tiles x 2 parts
__global__ __launch_bounds__(256) void k32(const char *w, int reps, float *out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    floatx16 acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = (floatx16)(0.f);
    bf16x8 b[4][2];
    for (int s = 0; s < 4; ++s) for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) b[s][i][e] = (__bf16)(0.001f * (lane + e + s + i));
    for (int r = 0; r < reps; ++r) {
        glds_stage<4>(w, (unsigned)(size_t)lds, wave, lane);
        for (int s = 0; s < N_STAGES; ++s) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const char *st = lds + (s & 1) * STAGE_BYTES + lane * 16;
            bf16x8 fr[3][2][2];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int i = 0; i < 2; ++i) fr[g][tt][i] = *(const bf16x8 *)(st + ((g * 2 + tt) * 2 + i) * 1024);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < N_STAGES) glds_stage<4>(w + (size_t)(s + 1) * STAGE_BYTES, (unsigned)(size_t)lds + ((s + 1) & 1) * STAGE_BYTES, wave, lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                if (g + 2 < 16) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int i = 0; i < 2; ++i) fr[(g + 2) % 3][tt][i] = *(const bf16x8 *)(st + (((g + 2) * 2 + tt) * 2 + i) * 1024);
                }
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int lin = g * 2 + tt, ks = lin / 8, t = lin % 8;
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[g % 3][tt][1], b[ks][0], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[g % 3][tt][0], b[ks][1], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[g % 3][tt][0], b[ks][0], acc[t], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + tid] = s;
}

// ---- 8 waves x 16 points, 16x16x32: stage = 2 k-steps x 16 tiles x 2 parts
template <int ORDER>
__global__ __launch_bounds__(512) void k16(const char *w, int reps, float *out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    floatx4 acc[16];
    for (int t = 0; t < 16; ++t) acc[t] = (floatx4)(0.f);
    bf16x8 b[2][2];
    for (int s = 0; s < 2; ++s) for (int i = 0; i < 2; ++i) for (int e = 0; e < 8; ++e) b[s][i][e] = (__bf16)(0.001f * (lane + e + s + i));
    for (int r = 0; r < reps; ++r) {
        glds_stage<8>(w, (unsigned)(size_t)lds, wave, lane);
        for (int s = 0; s < N_STAGES; ++s) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const char *st = lds + (s & 1) * STAGE_BYTES + lane * 16;
            bf16x8 fr[3][2][2];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int i = 0; i < 2; ++i) fr[g][tt][i] = *(const bf16x8 *)(st + ((g * 2 + tt) * 2 + i) * 1024);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < N_STAGES) glds_stage<8>(w + (size_t)(s + 1) * STAGE_BYTES, (unsigned)(size_t)lds + ((s + 1) & 1) * STAGE_BYTES, wave, lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                if (g + 2 < 16) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                        for (int i = 0; i < 2; ++i) fr[(g + 2) % 3][tt][i] = *(const bf16x8 *)(st + (((g + 2) * 2 + tt) * 2 + i) * 1024);
                }
                {
                    const int lin = g * 2, ks = lin / 16, t = lin % 16;
                    if (ORDER == 0) {
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) {
                            acc[t + tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[g % 3][tt][1], b[ks][0], acc[t + tt], 0, 0, 0);
                            acc[t + tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[g % 3][tt][0], b[ks][1], acc[t + tt], 0, 0, 0);
                            acc[t + tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[g % 3][tt][0], b[ks][0], acc[t + tt], 0, 0, 0);
                        }
                    } else {
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[g % 3][0][1], b[ks][0], acc[t], 0, 0, 0);
                        acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[g % 3][1][1], b[ks][0], acc[t + 1], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[g % 3][0][0], b[ks][1], acc[t], 0, 0, 0);
                        acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[g % 3][1][0], b[ks][1], acc[t + 1], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[g % 3][0][0], b[ks][0], acc[t], 0, 0, 0);
                        acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fr[g % 3][1][0], b[ks][0], acc[t + 1], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
    float s = 0.f;
    for (int t = 0; t < 16; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    out[blockIdx.x * 512 + tid] = s;
}

int main() {
    const size_t bytes = (size_t)N_STAGES * STAGE_BYTES;
    char *w; float *out;
    CK(hipMalloc(&w, bytes)); CK(hipMemset(w, 0, bytes));
    CK(hipMalloc(&out, 4096 * 512 * 4));
    const size_t lds = 2 * STAGE_BYTES;
    CK(hipFuncSetAttribute((const void *)k32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)k16<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void *)k16<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 20;
    // MFMA cycles per task per SIMD: 36 stages x 96 MFMAs x 32 cycles (k32: one wave) = 36 x 2 waves x 96 x 16 (k16)
    const double mfma_cycles = 36.0 * 96 * 32;
    for (int grid : {8, 256, 1024}) {
        for (int which = 0; which < 3; ++which) {
            for (int pass = 0; pass < 2; ++pass) {
                CK(hipEventRecord(e0));
                if (which == 0) hipLaunchKernelGGL(k32, dim3(grid), dim3(256), lds, 0, w, pass ? reps : 2, out);
                else if (which == 1) hipLaunchKernelGGL(k16<0>, dim3(grid), dim3(512), lds, 0, w, pass ? reps : 2, out);
                else hipLaunchKernelGGL(k16<1>, dim3(grid), dim3(512), lds, 0, w, pass ? reps : 2, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double rounds = grid <= 256 ? 1.0 : grid / 256.0;
            const double us_task = ms * 1e3 / reps / rounds;
            printf("%s grid=%4d  %7.1f us per task   MFMA utilisation at 2.4 GHz: %.2f\n", which == 2 ? "16-pt x 8 waves, term-major MFMA order" : (which ? "16-pt x 8 waves (16x16x32)            " : "32-pt x 4 waves (32x32x16)            "),
                   grid, us_task, mfma_cycles / 2400.0 / us_task);
        }
    }
    return 0;
}
