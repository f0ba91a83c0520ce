// The two callers' sides of the path, HBM-bound and trivially parallel:
//   k_gen_rays     : ray generation + batch gather on device -- get_rays, get_ndc_rays, get_view_dirs
//                    (reference src/data_preprocessors/DataPreprocessor01.py:335-378), the per-iteration index
//                    gather of load_nerf_cached_batch / load_visibility_prior_cached_batch (:566-615, :702-724)
//                    and the secondary camera centres of VipNeRF01.py:88-98.  Instead of caching every ray of
//                    every training frame (n*h*w*~30 floats) and gathering rows, the rays of the selected pixels
//                    are recomputed from the camera (13 floats per frame): 8 B in, ~110 B out per ray.
//   k_postprocess  : retrieve_inference_outputs (:866-894, :1074-1090): uint8 image (clip, round-half-even),
//                    non-negative depths.
// Arithmetic follows the reference's float32 operation order (numpy evaluates the expressions left to right in
// float32), with explicit round-to-nearest intrinsics so that nothing is contracted into FMAs.
#include "vipnerf_camera.h"

namespace vn {

__global__ void k_gen_rays(RayGenArgs a) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= a.N) return;
    const int64_t idx = a.g.indices ? a.g.indices[n] : a.g.first_index + n;
    const int64_t hw = (int64_t)a.g.height * a.g.width;
    const int f = (int)(idx / hw);
    const int rem = (int)(idx % hw);
    const int yi = rem / a.g.width, xi = rem % a.g.width;
    const vipnerf_camera &c = a.g.cameras[f];
    const float x = (float)xi, y = (float)yi;
    // dirs = Kinv @ [x, y, 1]; dirs[1:] *= -1
    float dir[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        dir[i] = __fadd_rn(__fadd_rn(__fmul_rn(c.kinv[3 * i], x), __fmul_rn(c.kinv[3 * i + 1], y)), c.kinv[3 * i + 2]);
    dir[1] = -dir[1];
    dir[2] = -dir[2];
    // rays_d = sum_j dirs_j * R[i][j]; rays_o = t
    float d[3], o[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        d[i] = __fadd_rn(__fadd_rn(__fmul_rn(dir[0], c.pose[4 * i]), __fmul_rn(dir[1], c.pose[4 * i + 1])), __fmul_rn(dir[2], c.pose[4 * i + 2]));
        o[i] = c.pose[4 * i + 3];
    }
    const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(d[0], d[0]), __fmul_rn(d[1], d[1])), __fmul_rn(d[2], d[2])));
    const vipnerf_ray_batch &b = a.out;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        b.rays_o[3 * n + i] = o[i];
        b.rays_d[3 * n + i] = d[i];
        if (b.view_dirs) b.view_dirs[3 * n + i] = __fdiv_rn(d[i], nrm);
    }
    if (b.near) b.near[n] = a.g.near;
    if (b.far) b.far[n] = a.g.far;
    if (a.g.ndc && b.rays_o_ndc && b.rays_d_ndc) {
        const float t = __fdiv_rn(-__fadd_rn(a.g.near, o[2]), d[2]);
        float os[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) os[i] = __fadd_rn(o[i], __fmul_rn(t, d[i]));
        const float two_near = __fmul_rn(2.f, a.g.near);
        b.rays_o_ndc[3 * n + 0] = __fdiv_rn(__fmul_rn(c.ndc_cx, os[0]), os[2]);
        b.rays_o_ndc[3 * n + 1] = __fdiv_rn(__fmul_rn(c.ndc_cy, os[1]), os[2]);
        b.rays_o_ndc[3 * n + 2] = __fadd_rn(1.f, __fdiv_rn(two_near, os[2]));
        b.rays_d_ndc[3 * n + 0] = __fmul_rn(c.ndc_cx, __fsub_rn(__fdiv_rn(d[0], d[2]), __fdiv_rn(os[0], os[2])));
        b.rays_d_ndc[3 * n + 1] = __fmul_rn(c.ndc_cy, __fsub_rn(__fdiv_rn(d[1], d[2]), __fdiv_rn(os[1], os[2])));
        b.rays_d_ndc[3 * n + 2] = __fdiv_rn(-two_near, os[2]);
        if (b.near_ndc) b.near_ndc[n] = a.g.near_ndc;
        if (b.far_ndc) b.far_ndc[n] = a.g.far_ndc;
    }
    if (b.pixel_id) { b.pixel_id[3 * n] = f; b.pixel_id[3 * n + 1] = xi; b.pixel_id[3 * n + 2] = yi; }
    // sparse-depth rows (DataPreprocessor01.py:544-563, :635-681): the same rays, but -1 where the nerf rows carry their
    // colour / visibility prior, and the per-pixel sparse-depth tables where the nerf rows carry -1
    const bool sd_row = a.g.row_is_sparse && a.g.row_is_sparse[n];
    if (b.target_rgb && a.g.images) {
#pragma unroll
        for (int i = 0; i < 3; ++i) b.target_rgb[3 * n + i] = sd_row ? -1.f : a.g.images[3 * idx + i];
    }
    const int V = a.g.n_frames - 1;
    if (b.prior && a.g.prior)                      // masks stored (n, n-1, h, w)
        for (int v = 0; v < V; ++v) b.prior[n * V + v] = sd_row ? -1.f : a.g.prior[((int64_t)f * V + v) * hw + rem];
    if (b.sparse_depth_values) b.sparse_depth_values[n] = (sd_row && a.g.sparse_depths) ? a.g.sparse_depths[idx] : -1.f;
    if (b.sparse_depth_errors) b.sparse_depth_errors[n] = (sd_row && a.g.sparse_errors) ? a.g.sparse_errors[idx] : -1.f;
    if (b.sparse_depth_values_ndc) b.sparse_depth_values_ndc[n] = (sd_row && a.g.sparse_depths_ndc) ? a.g.sparse_depths_ndc[idx] : -1.f;
    if (b.rays_o2)                                 // centre of camera v + (v >= f)   (VipNeRF01.py:93-97)
        for (int v = 0; v < V; ++v) {
            const vipnerf_camera &c2 = a.g.cameras[v + (v >= f ? 1 : 0)];
#pragma unroll
            for (int i = 0; i < 3; ++i) b.rays_o2[(n * V + v) * 3 + i] = c2.pose[4 * i + 3];
        }
}

__global__ void k_postprocess(int64_t n, const float *rgb, const float *depth, const float *depth_var, const float *depth_ndc,
                              const float *depth_var_ndc, uint8_t *image, float *o_depth, float *o_depth_var,
                              float *o_depth_ndc, float *o_depth_var_ndc) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (rgb && image) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float v = fminf(fmaxf(rgb[3 * i + c], 0.f), 1.f);
            image[3 * i + c] = (uint8_t)rintf(__fmul_rn(v, 255.f));      // numpy.round: half to even
        }
    }
    if (depth && o_depth) o_depth[i] = fmaxf(depth[i], 0.f);
    if (depth_var && o_depth_var) o_depth_var[i] = fmaxf(depth_var[i], 0.f);
    if (depth_ndc && o_depth_ndc) o_depth_ndc[i] = fmaxf(depth_ndc[i], 0.f);
    if (depth_var_ndc && o_depth_var_ndc) o_depth_var_ndc[i] = fmaxf(depth_var_ndc[i], 0.f);
}

// rays_o2[n][v] = translation of pose[v + (v >= frame(n))]  (VipNeRF01.py:88-98): the centres of the other cameras, per row
template <typename IDX>
__global__ void k_secondary_origins(int64_t N, int nf, const float *poses, const IDX *pixel_id, float *rays_o2) {
    const int V = nf - 1;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * V) return;
    const int64_t n = i / V;
    const int v = (int)(i % V);
    const int f = (int)pixel_id[3 * n];
    const int other = v + (v >= f ? 1 : 0);
    const float *T = poses + (size_t)other * 16;
    rays_o2[3 * i + 0] = T[3]; rays_o2[3 * i + 1] = T[7]; rays_o2[3 * i + 2] = T[11];
}
int launch_secondary_origins(int64_t N, int nf, const float *poses, const void *pixel_id, int idx64, float *rays_o2, hipStream_t st) {
    if (N <= 0 || nf <= 1) return VIPNERF_OK;
    const unsigned grid = (unsigned)((N * (nf - 1) + 255) / 256);
    if (idx64) hipLaunchKernelGGL(k_secondary_origins<int64_t>, dim3(grid), dim3(256), 0, st, N, nf, poses, (const int64_t *)pixel_id, rays_o2);
    else hipLaunchKernelGGL(k_secondary_origins<int32_t>, dim3(grid), dim3(256), 0, st, N, nf, poses, (const int32_t *)pixel_id, rays_o2);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

int launch_gen_rays(const RayGenArgs &a, hipStream_t st) {
    if (a.N <= 0) return VIPNERF_OK;
    hipLaunchKernelGGL(k_gen_rays, dim3((unsigned)((a.N + 255) / 256)), dim3(256), 0, st, a);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

int launch_postprocess(int64_t n, const float *rgb, const float *depth, const float *depth_var, const float *depth_ndc,
                       const float *depth_var_ndc, uint8_t *image, float *o_depth, float *o_depth_var, float *o_depth_ndc,
                       float *o_depth_var_ndc, hipStream_t st) {
    if (n <= 0) return VIPNERF_OK;
    hipLaunchKernelGGL(k_postprocess, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, rgb, depth, depth_var, depth_ndc,
                       depth_var_ndc, image, o_depth, o_depth_var, o_depth_ndc, o_depth_var_ndc);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

}  // namespace vn
