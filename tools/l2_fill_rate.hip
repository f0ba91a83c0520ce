// Microbenchmark: how fast can ONE workgroup (4 waves, one per SIMD) pull an L2-resident buffer into its CU?
//   mode 0: LDS-DMA (global_load_lds_dwordx4), 64 KiB stages, vmcnt(0) + barrier per stage
//   mode 1: global_load_dwordx4 into registers, ds_write_b128 into LDS, barrier per stage
//   mode 2: global_load_dwordx4 into registers only (xor-reduced), no LDS
//   mode 3: LDS-DMA issued by 8 waves (2 per SIMD)
// build: hipcc --offload-arch=gfx950 -O3 tools/l2_fill_rate.hip -o tools/bin/l2_fill_rate ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int STAGE_BYTES = 64 * 1024;

template <int WAVES>
__device__ __forceinline__ void glds_stage(const char *g, unsigned lds_base, int wave, int lane) {
    constexpr int PER_WAVE = 64 / WAVES;          // 1 KiB chunks per wave
    const char *src = g + (wave * PER_WAVE) * 1024 + lane * 16;
    unsigned dst = lds_base + (wave * PER_WAVE) * 1024;
    dst = __builtin_amdgcn_readfirstlane(dst);
#pragma unroll
    for (int i = 0; i < PER_WAVE; i += 4) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(src + i * 1024), "s"(dst + i * 1024) : "memory");
    }
}

template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_fill(const char *buf, size_t bytes, int reps, unsigned *out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_stages = (int)(bytes / STAGE_BYTES);
    unsigned acc = 0;
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0 || MODE == 3) {
            glds_stage<WAVES>(buf, (unsigned)(size_t)lds, wave, lane);
            for (int s = 0; s < n_stages; ++s) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (s + 1 < n_stages) glds_stage<WAVES>(buf + (size_t)(s + 1) * STAGE_BYTES, (unsigned)(size_t)lds + ((s + 1) & 1) * STAGE_BYTES, wave, lane);
                acc ^= *(const unsigned *)(lds + (s & 1) * STAGE_BYTES + tid * 4);
            }
        } else {
            for (int s = 0; s < n_stages; ++s) {
                const uint4 *g4 = (const uint4 *)(buf + (size_t)s * STAGE_BYTES);
                uint4 v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = g4[i * 256 + tid];
                if (MODE == 1) {
                    uint4 *l4 = (uint4 *)(lds + (s & 1) * STAGE_BYTES);
#pragma unroll
                    for (int i = 0; i < 16; ++i) l4[i * 256 + tid] = v[i];
                    __syncthreads();
                    acc ^= *(const unsigned *)(lds + (s & 1) * STAGE_BYTES + ((tid * 4 + 64) & (STAGE_BYTES - 1)));
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
                }
            }
        }
    }
    out[blockIdx.x * blockDim.x + tid] = acc;
}

template <int MODE, int WAVES>
int run(const char *name, const char *buf, size_t bytes, unsigned *out, int grid) {
    const int reps = 20;
    const size_t lds = 2 * STAGE_BYTES;
    CK(hipFuncSetAttribute((const void *)k_fill<MODE, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_fill<MODE, WAVES>), dim3(grid), dim3(WAVES * 64), lds, 0, buf, bytes, 2, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_fill<MODE, WAVES>), dim3(grid), dim3(WAVES * 64), lds, 0, buf, bytes, reps, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double per_wg = (double)bytes * reps / (ms * 1e-3) / 1e9;
    printf("%-34s grid=%4d  %8.3f ms  %7.1f GB/s per workgroup  %7.2f TB/s total\n", name, grid, ms, per_wg, per_wg * grid / 1e3);
    return 0;
}

int main() {
    const size_t bytes = 36 * STAGE_BYTES;      // 2.25 MiB: about one forward weight image, L2-resident
    char *buf; unsigned *out;
    CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes));
    CK(hipMalloc(&out, 4096 * 512 * 4));
    for (int grid : {8, 256, 1024}) {
        if (run<0, 4>("LDS-DMA, 4 waves", buf, bytes, out, grid)) return 1;
        if (run<3, 8>("LDS-DMA, 8 waves", buf, bytes, out, grid)) return 1;
        if (run<1, 4>("global_load -> ds_write, 4 waves", buf, bytes, out, grid)) return 1;
        if (run<2, 4>("global_load -> registers, 4 waves", buf, bytes, out, grid)) return 1;
        if (run<2, 8>("global_load -> registers, 8 waves", buf, bytes, out, grid)) return 1;
    }
    return 0;
}
