// Argument blocks of the ray-generation / post-processing kernels (vipnerf_camera.hip).
#pragma once
#include "vipnerf_common.h"

namespace vn {
struct RayGenArgs {
    vipnerf_raygen g;
    vipnerf_ray_batch out;
    int64_t N;
};
int launch_gen_rays(const RayGenArgs &a, hipStream_t st);
int launch_postprocess(int64_t n, const float *rgb, const float *depth, const float *depth_var, const float *depth_ndc,
                       const float *depth_var_ndc, uint8_t *image, float *o_depth, float *o_depth_var, float *o_depth_ndc,
                       float *o_depth_var_ndc, hipStream_t st);
int launch_secondary_origins(int64_t N, int nf, const float *poses, const void *pixel_id, int idx64, float *rays_o2, hipStream_t st);
int launch_psv(const vipnerf_psv *p, double *weights64, float *weights32, uint8_t *mask, hipStream_t st);
}  // namespace vn
