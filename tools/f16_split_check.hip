// Standalone numeric check: one 16x16x32 product with the fp16x3 split (3 cross terms), with and without power-of-two
// operand scaling, against double.  hipcc --offload-arch=gfx950 -O3 tools/f16_split_check.hip -o tools/bin/f16_split_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
// D[16x16] = A[16x32] * B[32x16] with fp16x3 split, compare against double
__global__ void k(const float *A, const float *B, float *D, float ws, float xs) {
    const int l = threadIdx.x, i = l & 15, q = l >> 4;
    half8 a0, a1, b0, b1;
    for (int e = 0; e < 8; ++e) {
        float av = A[i * 32 + 8 * q + e] * ws, bv = B[(8 * q + e) * 16 + i] * xs;
        _Float16 p = (_Float16)av; a0[e] = p; a1[e] = (_Float16)(av - (float)p);
        _Float16 r = (_Float16)bv; b0[e] = r; b1[e] = (_Float16)(bv - (float)r);
    }
    floatx4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b0, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + i] = c[r] / (ws * xs);
}
int main() {
    std::vector<float> A(512), B(512), D(256);
    srand(1);
    for (auto &v : A) v = 0.06f * ((rand() / (float)RAND_MAX) * 2 - 1);
    for (auto &v : B) v = 3.0f * (rand() / (float)RAND_MAX);
    float *dA, *dB, *dD;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    for (int cfg = 0; cfg < 2; ++cfg) {
        float ws = cfg ? 256.f : 1.f, xs = cfg ? 16.f : 1.f;
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, ws, xs);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        double maxe = 0, maxv = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double s = 0; for (int kk = 0; kk < 32; ++kk) s += (double)A[i * 32 + kk] * B[kk * 16 + j];
            maxe = fmax(maxe, fabs(s - D[i * 16 + j])); maxv = fmax(maxv, fabs(s));
        }
        printf("scales %g %g: max abs err %.3e (max |d| %.3f) -> rel %.2e\n", ws, xs, maxe, maxv, maxe / maxv);
    }
    return 0;
}
