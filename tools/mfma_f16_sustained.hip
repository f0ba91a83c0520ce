// What v_mfma_f32_16x16x32_f16 / v_mfma_f32_32x32x16_f16 SUSTAIN on an MI355X over seconds, from registers only (no LDS, no memory
// traffic), with operands that look like data (pseudo-random fp16 values, different per lane) rather than constants: the nominal
// 2.5 PFLOP/s assumes 2.4 GHz, and these kernels run into the socket power limit long before that.  Prints TFLOP/s per one-second
// window; run rocm-smi next to it for the clock and the power (tools/power_clock.sh).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f16_sustained.hip -o tools/bin/mfma_f16_sustained && tools/bin/mfma_f16_sustained
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ half8 make_operand(unsigned seed) {
    half8 h;
    for (int e = 0; e < 8; ++e) {
        seed = seed * 1664525u + 1013904223u;
        h[e] = (_Float16)(((int)(seed >> 16) % 2001 - 1000) * 1e-3f);
    }
    return h;
}
// the same loop on v_mfma_f32_16x16x32_bf16
__global__ __launch_bounds__(512) void kbf(float *out, int iters) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        const half8 x = make_operand(threadIdx.x * 8 + i + blockIdx.x * 977), y = make_operand(threadIdx.x * 8 + 4 + i);
        for (int e = 0; e < 8; ++e) { a[i][e] = (__bf16)(float)x[e]; b[i][e] = (__bf16)(float)y[e]; }
    }
    floatx4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (floatx4)(0.f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + u) & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
// WG waves per SIMD = WG / 256; independent accumulator tiles per wave; operand sets cycled so consecutive MFMAs differ
template <int WG, bool WIDE>
__global__ __launch_bounds__(WG) void k(float *out, int iters) {
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = make_operand(threadIdx.x * 8 + i + blockIdx.x * 977); b[i] = make_operand(threadIdx.x * 8 + 4 + i); }
    float s = 0.f;
    if (WIDE) {
        floatx16 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = (floatx16)(0.f);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + u) & 3], b[u], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    } else {
        floatx4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (floatx4)(0.f);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + u) & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    }
    out[blockIdx.x * WG + threadIdx.x] = s;
}
template <typename F>
static void run(const char *name, F launch, double flop_per_launch, int seconds) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    printf("%-52s", name);
    for (int w = 0; w < seconds; ++w) {
        int n = 0; float ms = 0.f;
        hipEventRecord(e0);
        do { for (int i = 0; i < 4; ++i) launch(); n += 4; hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); } while (ms < 1000.f);
        printf(" %7.0f", n * flop_per_launch / (ms * 1e-3) / 1e12);
        fflush(stdout);
    }
    printf("  TFLOP/s per 1-s window (nominal 2500)\n");
}
// ---- burst mode (round 3: which ceiling applies to a 1-10 ms kernel?) --------------------------------------------------------------
// `mfma_f16_sustained burst`: ONE launch of the 16x16x32 f16 loop per measurement, of 0.05 ... 1000 ms, after 200 ms of idle (the chip
// starts cool, at full clock), with operands that are all ZERO (what a peak micro-benchmark typically multiplies: no operand bit toggles)
// or data-like; then the same launch repeated back to back for 2 s with a duty cycle (burst, idle gap) like a kernel inside a training step.
#include <chrono>
#include <thread>
template <bool ZERO>
__global__ __launch_bounds__(512) void kburst(float *out, int iters) {
    half8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = make_operand(threadIdx.x * 8 + i + blockIdx.x * 977); b[i] = make_operand(threadIdx.x * 8 + 4 + i);
        if (ZERO) for (int e = 0; e < 8; ++e) { a[i][e] = (_Float16)(out[0] * 0.f); b[i][e] = a[i][e]; }     // (run-time zero: not folded away)
    }
    floatx4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (floatx4)(0.f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + u) & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <bool ZERO>
static void bursts(float *out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 8;
    hipMemset(out, 0, 256 * 8 * 512 * sizeof(float));
    hipLaunchKernelGGL(kburst<ZERO>, dim3(grid), dim3(512), 0, 0, out, 10); hipDeviceSynchronize();
    printf("%s operands, one launch after 200 ms idle: ms -> TFLOP/s (3 repeats)\n", ZERO ? "ZERO" : "data-like");
    for (int iters : {30, 100, 300, 1000, 3000, 10000, 30000, 100000, 600000}) {
        const double flop = (double)grid * 8 * iters * 16 * 16384.0;
        printf("  iters %6d:", iters);
        for (int r = 0; r < 3; ++r) {
            std::this_thread::sleep_for(std::chrono::milliseconds(200));
            float ms = 0.f;
            hipEventRecord(e0);
            hipLaunchKernelGGL(kburst<ZERO>, dim3(grid), dim3(512), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            printf("  %8.3f ms %5.0f", ms, flop / (ms * 1e-3) / 1e12);
        }
        printf("\n");
    }
    // duty cycle: a ~3 ms burst every ~10 ms for 2 s (the MLP kernel of a 16-bit training step), rate of the bursts themselves
    for (int gap_ms : {0, 3, 7}) {
        const int iters = 5000;
        const double flop = (double)grid * 8 * iters * 16 * 16384.0;
        double tot_ms = 0; int n = 0;
        const auto t0 = std::chrono::steady_clock::now();
        double last = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.0) {
            float ms = 0.f;
            hipEventRecord(e0);
            hipLaunchKernelGGL(kburst<ZERO>, dim3(grid), dim3(512), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            tot_ms += ms; ++n; last = ms;
            if (gap_ms) std::this_thread::sleep_for(std::chrono::milliseconds(gap_ms));
        }
        printf("  duty cycle: %.2f ms bursts with %d ms gaps for 2 s: mean %5.0f TFLOP/s over %d bursts (last burst %5.0f)\n", tot_ms / n, gap_ms,
               flop * n / (tot_ms * 1e-3) / 1e12, n, flop / (last * 1e-3) / 1e12);
    }
}
int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "burst")) {
        float *o;
        hipMalloc(&o, 256 * 8 * 512 * sizeof(float));
        bursts<true>(o);
        bursts<false>(o);
        return 0;
    }
    const int seconds = argc > 1 ? atoi(argv[1]) : 6;
    float *out;
    hipMalloc(&out, 256 * 8 * 512 * sizeof(float));
    const int iters = 4000, grid = 256 * 8;
    run("16x16x32 f16, 2 waves/SIMD, 8 accumulator tiles", [&] { hipLaunchKernelGGL((k<512, false>), dim3(grid), dim3(512), 0, 0, out, iters); },
        (double)grid * 8 * iters * 16 * 16384.0, seconds);
    run("32x32x16 f16, 1 wave/SIMD, 4 accumulator tiles", [&] { hipLaunchKernelGGL((k<256, true>), dim3(grid), dim3(256), 0, 0, out, iters); },
        (double)grid * 4 * iters * 16 * 32768.0, seconds);
    run("32x32x16 f16, 2 waves/SIMD, 4 accumulator tiles", [&] { hipLaunchKernelGGL((k<512, true>), dim3(grid), dim3(512), 0, 0, out, iters); },
        (double)grid * 8 * iters * 16 * 32768.0, seconds);
    run("16x16x32 bf16, 2 waves/SIMD, 8 accumulator tiles", [&] { hipLaunchKernelGGL(kbf, dim3(grid), dim3(512), 0, 0, out, iters); },
        (double)grid * 8 * iters * 16 * 16384.0, seconds);
    return 0;
}
