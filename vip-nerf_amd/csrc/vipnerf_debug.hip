// Stage exports for the parity tests of pieces that otherwise only run inside the fused kernels: the secondary
// viewing directions (VipNeRF.compute_other_view_dirs, reference src/models/VipNeRF01.py:218-226) and the on-device
// Philox4x32-10 generator that replaces the reference's torch.rand / torch.randn draws (VipNeRF01.py:200,242,551).
// They call the very device functions the production kernels use (secondary_dir, philox4x32, rng_uniform, rng_normal).
#include "vipnerf_mlp.h"

namespace vn {

__global__ void k_secondary_dirs(PointSrc s, float *out) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= s.P) return;
    PointCtx c;
    load_point(s, p, c);
    for (int v = 0; v < s.V; ++v) {
        float d[3];
        secondary_dir(s, c, v, d);
#pragma unroll
        for (int i = 0; i < 3; ++i) out[(p * s.V + v) * 3 + i] = d[i];
    }
}

__global__ void k_philox(int64_t n, const uint32_t *ctr, const uint32_t *key, uint32_t *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4 r = philox4x32(make_uint4(ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3]),
                               make_uint2(key[2 * i], key[2 * i + 1]));
    out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
}

__global__ void k_rng_draw(int kind, uint64_t seed, uint64_t offset, uint32_t stream, uint64_t first, int64_t n, float *out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = kind ? rng_normal(seed, offset, stream, first + (uint64_t)i) : rng_uniform(seed, offset, stream, first + (uint64_t)i);
}

int launch_secondary_dirs(const PointSrc &s, float *out, hipStream_t st) {
    if (s.P <= 0 || s.V <= 0) return VIPNERF_OK;
    hipLaunchKernelGGL(k_secondary_dirs, dim3((unsigned)((s.P + 255) / 256)), dim3(256), 0, st, s, out);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}
int launch_philox(int64_t n, const uint32_t *ctr, const uint32_t *key, uint32_t *out, hipStream_t st) {
    if (n <= 0) return VIPNERF_OK;
    hipLaunchKernelGGL(k_philox, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, ctr, key, out);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}
int launch_rng_draw(int kind, uint64_t seed, uint64_t offset, uint32_t stream, uint64_t first, int64_t n, float *out, hipStream_t st) {
    if (n <= 0) return VIPNERF_OK;
    hipLaunchKernelGGL(k_rng_draw, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, kind, seed, offset, stream, first, n, out);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

}  // namespace vn
