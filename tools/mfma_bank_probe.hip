// Does v_mfma_f32_16x16x4_f32 lose cycles when its A and B source registers sit in the same VGPR bank (register index mod 4)?  The exact-fp32
// kernels multiply a.v[e] * b.v[e], e = 0..3, of two 4-register operands: if the two quads start at congruent registers every MFMA of the four pairs
// same-bank sources.  One wave per SIMD, 16 accumulators round robin, operands in fixed registers chosen through inline-asm constraints.
//   SAME: A in v[.. +e], B in v[.. +e] with congruent bases;  ROT: B element (e + 3) & 3 (bank differs by one)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int ROT>
__global__ __launch_bounds__(256) void k_bank(float *out, int iters, float a0, float b0) {
    // two operand quads pinned to registers whose indices are congruent mod 4: v[40:43] and v[48:51]
    float av0, av1, av2, av3, bv0, bv1, bv2, bv3;
    asm volatile("v_mov_b32 v40, %0\n v_mov_b32 v41, %0\n v_mov_b32 v42, %0\n v_mov_b32 v43, %0\n"
                 "v_mov_b32 v48, %1\n v_mov_b32 v49, %1\n v_mov_b32 v50, %1\n v_mov_b32 v51, %1" :: "v"(a0), "v"(b0)
                 : "v40", "v41", "v42", "v43", "v48", "v49", "v50", "v51");
    floatx4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (floatx4)(0.f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            // four MFMAs of one (tile, part): A register 40 + e, B register 48 + ((e + ROT) & 3)
            if (ROT == 0)
                asm volatile("v_mfma_f32_16x16x4_f32 %0, v40, v48, %0\n v_mfma_f32_16x16x4_f32 %0, v41, v49, %0\n"
                             "v_mfma_f32_16x16x4_f32 %0, v42, v50, %0\n v_mfma_f32_16x16x4_f32 %0, v43, v51, %0" : "+v"(acc[t]) :: "v40", "v41", "v42", "v43", "v48", "v49", "v50", "v51");
            else
                asm volatile("v_mfma_f32_16x16x4_f32 %0, v40, v51, %0\n v_mfma_f32_16x16x4_f32 %0, v41, v48, %0\n"
                             "v_mfma_f32_16x16x4_f32 %0, v42, v49, %0\n v_mfma_f32_16x16x4_f32 %0, v43, v50, %0" : "+v"(acc[t]) :: "v40", "v41", "v42", "v43", "v48", "v49", "v50", "v51");
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int ROT>
static void run(float *out, const char *what) {
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_bank<ROT>, dim3(256), dim3(256), 0, 0, out, iters, 1e-3f, 2e-3f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double tf = 256.0 * 4 * iters * 64.0 * 2048.0 / ms * 1e-9;
    printf("%-64s %8.3f ms  %.3f of 157.3\n", what, ms, tf / 157.3);
}
int main() {
    float *out; (void)hipMalloc(&out, 256 * 256 * 4);
    run<0>(out, "A = v[40 + e], B = v[48 + e]: same bank for every MFMA");
    run<1>(out, "A = v[40 + e], B = v[48 + (e + 3) % 4]: banks differ by one");
    return 0;
}
