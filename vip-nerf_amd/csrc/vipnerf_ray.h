// Argument blocks of the per-ray kernels (vipnerf_ray.hip).
#pragma once
#include "vipnerf_common.h"

namespace vn {

struct CompositeArgs {
    int64_t N;
    int32_t S, V, ndc, white_bkgd;
    const float *rays_o, *rays_d, *rays_d_s;
    vipnerf_level_out lvl;            // reads z_vals, raw_*; writes the rest
};

struct CompositeBwdArgs {
    int64_t N;
    int32_t S, V, ndc, white_bkgd;
    const float *rays_o, *rays_d, *rays_d_s;
    vipnerf_level_out lvl;            // forward outputs (read only)
    vipnerf_level_grads g;            // upstream gradients (nullable members)
    float *dsig, *drgb, *dvis, *dvis2;   // dLoss/d(raw network outputs): (P), (P,3), (P), (P,V)
};

struct SampleArgs {
    int64_t N;
    int32_t Sc, Sf;
    const float *z_coarse, *w_coarse, *u;
    int32_t device_rng;
    uint64_t seed, offset, ray_base;
    const int64_t *ray_ids;
    float *z_fine;
    int32_t *inds;
    float *z_samples;
};

struct LossArgs {
    int64_t N;
    int32_t V, n_levels, S_coarse, S_fine;
    vipnerf_loss_in in;
    vipnerf_level_out coarse, fine;
    vipnerf_loss_level_seeds seeds_coarse, seeds_fine;
    float *partial;                   // (7*N) + 2 counts appended by the caller
    float *counts;                    // (2)
    float *loss_values;               // (8)
    float w[8];                       // with total: total[0] = sum_k w[k] * loss_values[k], named[j] = loss_values[2 j] + loss_values[2 j + 1] (k_loss_final)
    float *total, *named;
};

struct SecOriginArgs {                // k_coarse_z's side job in vipnerf_train_step: the other cameras' centres per row (rays_o2 == NULL: none)
    const float *poses; const void *pixel_id; int idx64, nf; float *rays_o2;
};
int launch_coarse_z(int64_t N, int S, int lindisp, const float *near, const float *far, const float *t_rand,
                    int device_rng, uint64_t seed, uint64_t offset, uint64_t ray_base, const int64_t *ray_ids, float *z_out,
                    hipStream_t st, const SecOriginArgs *so = nullptr);
int launch_composite(const CompositeArgs &a, hipStream_t st);
int launch_composite_bwd(const CompositeBwdArgs &a, hipStream_t st);
int launch_sample_fine(const SampleArgs &a, hipStream_t st);
int launch_composite_sample(const CompositeArgs &c, const SampleArgs &a, hipStream_t st);      // the coarse level's compositing + the sampling it feeds, one launch
int launch_losses(const LossArgs &a, hipStream_t st, bool defer_final = false);
struct LossFinalTail { LossArgs a; };
struct ScaleArgs {
    vipnerf_scale_seg s[VIPNERF_MAX_SCALE_SEGS];
    int n;
    const float *g;                   // (8) device: the factors by slot; NULL = w below
    const float *g1;                  // (1) device, with g == NULL: the factors are g1[0] * w[slot] (the upstream gradient of a weighted TotalLoss)
    float w[8];                       // host-side factors (vipnerf_train_step: the loss weights)
    const float *loss_values;         // with total: total[0] = sum_k w[k] * loss_values[k] (fixed order, block (0, 0) thread 0)
    float *total;
};
int launch_scale_segments(const ScaleArgs &a, hipStream_t st, const LossArgs *fin = nullptr);
int launch_adam_step(int64_t n, float *p, float *m, float *v, const float *g, float lerp_w, float beta2, float sq_w, float inv_s, float eps,
                     float neg_step, int mask, hipStream_t st);

}  // namespace vn
