// Does halving the LDS bytes per MFMA buy anything on the 16-bit MLP loop?  The product's 16-bit kernels read one 1 KiB A fragment
// (ds_read_b128) per TWO v_mfma_f32_16x16x32_bf16 (two 16-point tiles per wave, two waves per SIMD).  This probe runs that loop from a resident 64 KiB LDS
// stage with data-like operands -- the socket sits at its power cap with them, so the question is what the CLOCK does when the LDS traffic per FLOP changes --
// and the candidate geometries next to it:
//   regs        : no LDS at all (fragments in registers): the MFMA stream's own sustained rate
//   2w x 2t     : two waves per SIMD, a fragment feeds 2 MFMAs 16x16x32            (the product: 1 KiB per 32 768 FLOP)
//   1w x 4t     : one wave per SIMD, a fragment feeds 4 MFMAs 16x16x32             (0.5 KiB per 32 768 FLOP)
//   1w x 2T wide: one wave per SIMD, a fragment feeds 2 MFMAs 32x32x16 (2 x 32 pts)  (0.5 KiB per 32 768 FLOP)
//   2w x 1T wide: two waves per SIMD, a fragment feeds 1 MFMA 32x32x16              (1 KiB per 32 768 FLOP: the product's ratio with the wide instruction)
//   2w x 4t     : two waves per SIMD, a fragment feeds 4 MFMAs 16x16x32 (M = 64 accumulator registers per tile are what a real kernel could not hold: probe only)
// Prints TFLOP/s per one-second window (sustained) and for 5 ms bursts after idle; run tools/power_clock-style rocm-smi sampling next to it for power / clock.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_ratio_probe.hip -o tools/bin/lds_ratio_probe && tools/bin/lds_ratio_probe [seconds]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 make_operand(unsigned seed) {
    bf16x8 h;
    for (int e = 0; e < 8; ++e) {
        seed = seed * 1664525u + 1013904223u;
        h[e] = (__bf16)(((int)(seed >> 16) % 2001 - 1000) * 1e-3f);
    }
    return h;
}

constexpr int FRAGS = 64;     // 64 KiB stage = 64 fragments of 1 KiB

// T point tiles per wave; WIDE: 32x32x16 (a tile is 32 points), else 16x16x32 (16 points); LDS: fragments come from LDS, else from 4 registers
template <int WG, int T, bool WIDE, bool LDS>
__global__ __launch_bounds__(WG) void k(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) bf16x8 stage[96 * 64];      // 96 KiB declared: ONE workgroup per CU, so that WG / 256 IS the waves per SIMD
    for (int i = threadIdx.x; i < FRAGS * 64; i += WG) stage[i] = make_operand(i * 31 + blockIdx.x * 977);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    bf16x8 b[T][4], areg[4];
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < 4; ++i) b[t][i] = make_operand(threadIdx.x * 8 + 4 + i + t * 1237);
    for (int i = 0; i < 4; ++i) areg[i] = make_operand(threadIdx.x * 8 + i + blockIdx.x * 977);
    float s = 0.f;
    if constexpr (WIDE) {
        floatx16 acc[T][2];
        for (int t = 0; t < T; ++t) for (int m = 0; m < 2; ++m) acc[t][m] = (floatx16)(0.f);
        for (int it = 0; it < iters; ++it) {
#pragma unroll 16
            for (int f = 0; f < FRAGS; ++f) {
                const bf16x8 a = LDS ? stage[f * 64 + lane] : areg[f & 3];
#pragma unroll
                for (int t = 0; t < T; ++t) acc[t][f & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[t][f & 3], acc[t][f & 1], 0, 0, 0);
            }
            asm volatile("" ::: "memory");
        }
        for (int t = 0; t < T; ++t) for (int m = 0; m < 2; ++m) s += acc[t][m][0] + acc[t][m][15];
    } else {
        floatx4 acc[T][4];
        for (int t = 0; t < T; ++t) for (int m = 0; m < 4; ++m) acc[t][m] = (floatx4)(0.f);
        for (int it = 0; it < iters; ++it) {
#pragma unroll 16
            for (int f = 0; f < FRAGS; ++f) {
                const bf16x8 a = LDS ? stage[f * 64 + lane] : areg[f & 3];
#pragma unroll
                for (int t = 0; t < T; ++t) acc[t][f & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[t][(f >> 2) & 3], acc[t][f & 3], 0, 0, 0);
            }
            asm volatile("" ::: "memory");
        }
        for (int t = 0; t < T; ++t) for (int m = 0; m < 4; ++m) s += acc[t][m][0] + acc[t][m][3];
    }
    out[blockIdx.x * WG + threadIdx.x] = s;
}

template <typename F>
static void run(const char *name, F launch, double flop_per_launch, int seconds) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    printf("%-58s", name);
    // bursts after idle first (what a 1-5 ms kernel inside a step sees), then the sustained windows
    double burst = 0;
    for (int r = 0; r < 3; ++r) {
        std::this_thread::sleep_for(std::chrono::milliseconds(200));
        float ms = 0.f;
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        burst = flop_per_launch / (ms * 1e-3) / 1e12;
        printf(" b%5.0f(%.1fms)", burst, ms);
    }
    printf(" |");
    for (int w = 0; w < seconds; ++w) {
        int n = 0; float ms = 0.f;
        hipEventRecord(e0);
        do { for (int i = 0; i < 4; ++i) launch(); n += 4; hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); } while (ms < 1000.f);
        printf(" %6.0f", n * flop_per_launch / (ms * 1e-3) / 1e12);
        fflush(stdout);
    }
    printf("\n");
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int seconds = argc > 1 ? atoi(argv[1]) : 4;
    float *out;
    hipMalloc(&out, 256 * 4 * 512 * sizeof(float));
    const int grid = 256 * 4;            // four rounds of one workgroup per CU
    printf("columns: three 1-launch bursts after 200 ms idle, then TFLOP/s per 1-s window (nominal peak 2500)\n");
#define RUN(NAME, WG, T, WIDE, LDS, ITERS)                                                                                                    \
    run(NAME, [&] { hipLaunchKernelGGL((k<WG, T, WIDE, LDS>), dim3(grid), dim3(WG), 0, 0, out, ITERS); },                                 \
        (double)grid * (WG / 64) * (ITERS) * FRAGS * T * (WIDE ? 32768.0 : 16384.0), seconds)
    RUN("regs  2 waves/SIMD x 2 tiles 16x16x32 (no LDS)", 512, 2, false, false, 400);
    RUN("LDS   2 waves/SIMD x 2 tiles 16x16x32 (product: 1 KiB/2 MFMA)", 512, 2, false, true, 400);
    RUN("LDS   1 wave/SIMD  x 4 tiles 16x16x32 (1 KiB/4 MFMA)", 256, 4, false, true, 400);
    RUN("LDS   1 wave/SIMD  x 2 tiles 32x32x16 (1 KiB/2 wide MFMA)", 256, 2, true, true, 400);
    RUN("LDS   2 waves/SIMD x 1 tile  32x32x16 (1 KiB/1 wide MFMA)", 512, 1, true, true, 400);
    RUN("LDS   2 waves/SIMD x 4 tiles 16x16x32 (1 KiB/4 MFMA)", 512, 4, false, true, 200);
    RUN("LDS   2 waves/SIMD x 2 tiles 32x32x16 (1 KiB/2 wide MFMA)", 512, 2, true, true, 200);
    RUN("regs  1 wave/SIMD  x 2 tiles 32x32x16 (no LDS)", 256, 2, true, false, 400);
    return 0;
}
