// Argument blocks and device building blocks shared by the forward and backward MLP kernels.
#pragma once
#include "vipnerf_common.h"

namespace vn {

// Where a workgroup's points come from.
//  rays mode   (the render path): point p = ray n*S + k;  x = o_s[n] + d_s[n]*z[p];  dir = view_dirs[n];
//              secondary directions computed from (z, rays_o, rays_d, rays_o2) as VipNeRF01.py:218-226.
//  points mode (vipnerf_mlp_forward): pts[p], dirs[p], dirs2[p][v] given explicitly.
struct PointSrc {
    int64_t P;            // points
    int32_t S;            // samples per ray (rays mode)
    int32_t V;            // secondary directions
    int32_t rays_mode;
    int32_t ndc;
    const float *z;       // (P)
    const float *rays_o, *rays_d, *rays_o_s, *rays_d_s, *view_dirs, *rays_o2;
    const float *pts, *dirs, *dirs2;
};

struct NoiseSrc {
    const float *noise;   // (P) or NULL
    float std;
    int32_t device_rng;   // 1: draw N(0,1) from Philox when noise == NULL
    uint32_t stream;
    uint64_t seed, offset;
    uint64_t idx_base;    // Philox index of point 0 (= vipnerf_rng::ray_base * S)
    const int64_t *ray_ids;   // rays mode: explicit global row index per ray (vipnerf_rng::ray_ids) or NULL
};

#if defined(__HIPCC__)
// Philox index of point p's sigma-noise draw: (global ray index) * S + sample
__device__ __forceinline__ uint64_t noise_index(const NoiseSrc &ns, const PointSrc &s, int64_t p) {
    if (ns.ray_ids && s.rays_mode) return (uint64_t)ns.ray_ids[p / s.S] * (uint64_t)s.S + (uint64_t)(p % s.S);
    return ns.idx_base + (uint64_t)p;
}
#endif

struct MlpFwdArgs {
    PointSrc src;
    NoiseSrc ns;
    const float *packed;
    float *sigma, *rgb, *vis, *vis2;   // (P), (P,3), (P), (P,V)
    float *acts;                       // activation store base (NULL = do not save)
    ActLayout al;
};

struct MlpBwdArgs {
    PointSrc src;
    const float *packed;
    const float *sigma, *rgb, *vis, *vis2;       // forward outputs (for the activation derivatives)
    const float *acts;
    ActLayout al;
    float *bwd;                                  // backward scratch base
    BwdLayout bl;
    const unsigned *gmax;                        // FP16X3: max |seed| slot (grad_scale_from_max); else nullptr
};

constexpr int MLP_PTS_PER_WG = 128;         // 8 waves x 16 points (narrow layout, vipnerf_bf16n.h)

#if defined(__HIPCC__)
// Loads / computes the wave's point: position x, primary direction, and what is needed for the secondary ones.
struct PointCtx {
    float x[3], dir[3];
    float o[3], d[3], z;      // world-space ray + depth (rays mode)
    int64_t p, n;             // point, ray
};
__device__ __forceinline__ void load_point(const PointSrc &s, int64_t p, PointCtx &c) {
    c.p = p;
    if (s.rays_mode) {
        const int64_t n = p / s.S;
        c.n = n;
        c.z = s.z[p];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            c.x[i] = __fadd_rn(s.rays_o_s[3 * n + i], __fmul_rn(s.rays_d_s[3 * n + i], c.z));
            c.dir[i] = s.view_dirs[3 * n + i];
            c.o[i] = s.rays_o[3 * n + i];
            c.d[i] = s.rays_d[3 * n + i];
        }
    } else {
        c.n = p;
        c.z = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            c.x[i] = s.pts[3 * p + i];
            c.dir[i] = s.dirs[3 * p + i];
            c.o[i] = c.d[i] = 0.f;
        }
    }
}
// the rays-mode arithmetic of secondary_dir with the secondary camera's origin already in registers (callers that load it ahead of use)
__device__ __forceinline__ void secondary_dir_from(const PointSrc &s, const PointCtx &c, const float (&o2)[3], float out[3]) {
    float t = c.z;
    if (s.ndc) {
        const float tn = __fdiv_rn(-(1.f + c.o[2]), c.d[2]);
        const float num = __fadd_rn(c.o[2], __fmul_rn(tn, c.d[2]));
        const float den = __fadd_rn(__fsub_rn(1.f, c.z), 1e-6f);
        t = __fdiv_rn(__fsub_rn(__fdiv_rn(num, den), c.o[2]), c.d[2]);
    }
    float w[3], n2 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float pt = __fadd_rn(c.o[i], __fmul_rn(t, c.d[i]));
        w[i] = __fsub_rn(pt, o2[i]);
        n2 = __fadd_rn(n2, __fmul_rn(w[i], w[i]));
    }
    const float nrm = sqrtf(n2);
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = __fdiv_rn(w[i], nrm);
}
// direction from secondary camera v to the point (VipNeRF01.py:218-226)
__device__ __forceinline__ void secondary_dir(const PointSrc &s, const PointCtx &c, int v, float out[3]) {
    if (!s.rays_mode) {
#pragma unroll
        for (int i = 0; i < 3; ++i) out[i] = s.dirs2[(c.p * s.V + v) * 3 + i];
        return;
    }
    float o2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) o2[i] = s.rays_o2[(c.n * s.V + v) * 3 + i];
    secondary_dir_from(s, c, o2, out);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
#endif


}  // namespace vn
