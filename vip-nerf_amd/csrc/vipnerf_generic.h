// Generic-topology MLP path (vipnerf_generic.hip): topology descriptor, flat parameter buffer, workspace layouts.
#pragma once
#include <cstring>
#include "vipnerf_mlp.h"

namespace vn {

// One MLP of the reference (src/models/VipNeRF01.py:451-492): D trunk layers of width W, positional-encoding degrees lp / lv,
// gamma(x) concatenated to the input of layer `skip` (= 5: `self.skips = [4]` re-injects after layer 4) when D > 5, else -1;
// W/2-wide view layer.  Heads (VipNeRF01.py:467-491), selected by `heads` (vipnerf_config.head_variant):
//   0                              sigma from the trunk; rgb(3) + visibility(1) from the view branch -- every shipped / BASELINE config
//   VIPNERF_HEAD_RGB_TRUNK         view_dependent_rgb = False: the trunk head is [sigma, rgb(3)], the view branch predicts visibility only
//   VIPNERF_HEAD_NO_VISIBILITY     predict_visibility = False: the view branch predicts rgb only; no secondary views
//   both                           no view-dependent output at all: no feature layer, no view layer, no view head
struct GenTopo {
    int D, W, lp, lv, dp, dv, skip, heads;
};
__host__ __device__ inline GenTopo gen_topo(int depth, int width, int lp, int lv, int heads = 0) {
    GenTopo t;
    t.D = depth; t.W = width; t.lp = lp; t.lv = lv; t.dp = 3 + 6 * lp; t.dv = 3 + 6 * lv; t.skip = depth > 5 ? 5 : -1; t.heads = heads;
    return t;
}
__host__ __device__ inline bool gen_rgb_trunk(const GenTopo &t) { return (t.heads & VIPNERF_HEAD_RGB_TRUNK) != 0; }
__host__ __device__ inline bool gen_pred_vis(const GenTopo &t) { return (t.heads & VIPNERF_HEAD_NO_VISIBILITY) == 0; }
__host__ __device__ inline int gen_trunk_outs(const GenTopo &t) { return gen_rgb_trunk(t) ? 4 : 1; }                  // rows of pts_output_linear
__host__ __device__ inline int gen_view_outs(const GenTopo &t) { return (gen_rgb_trunk(t) ? 0 : 3) + (gen_pred_vis(t) ? 1 : 0); }   // rows of views_output_linear
// the specialised MFMA kernels cover exactly this one
__host__ __device__ inline bool gen_is_fused_topology(const GenTopo &t) { return t.D == 8 && t.W == 256 && t.lp == LP && t.lv == LV && t.heads == 0; }

// elements of parameter slot i (vipnerf_mlp_params order); 0 = slot unused by this topology
__host__ __device__ inline size_t gen_param_numel(const GenTopo &t, int i) {
    if (i < 16) {
        const int l = i >> 1;
        if (l >= t.D) return 0;
        const int in = l == 0 ? t.dp : (l == t.skip ? t.W + t.dp : t.W);
        return (i & 1) ? (size_t)t.W : (size_t)t.W * in;
    }
    const size_t vo = (size_t)gen_view_outs(t), to = (size_t)gen_trunk_outs(t);
    switch (i) {
        case P_VW: return vo ? (size_t)(t.W / 2) * (t.W + t.dv) : 0;
        case P_VB: return vo ? (size_t)t.W / 2 : 0;
        case P_SW: return to * t.W;
        case P_SB: return to;
        case P_FW: return vo ? (size_t)t.W * t.W : 0;
        case P_FB: return vo ? (size_t)t.W : 0;
        case P_OW: return vo * (t.W / 2);
        case P_OB: return vo;
    }
    return 0;
}
// float offsets inside the flat parameter buffer (tensors back to back in slot order)
struct GenParams {
    size_t w[D], b[D], wv, bv, ws, bs, wf, bf, wo, bo, total;
};
__host__ __device__ inline GenParams gen_params(const GenTopo &t) {
    GenParams g;
    size_t off[VIPNERF_N_PARAMS], o = 0;
    for (int i = 0; i < VIPNERF_N_PARAMS; ++i) { off[i] = o; o += gen_param_numel(t, i); }
    for (int l = 0; l < D; ++l) { g.w[l] = off[2 * l]; g.b[l] = off[2 * l + 1]; }
    g.wv = off[P_VW]; g.bv = off[P_VB]; g.ws = off[P_SW]; g.bs = off[P_SB]; g.wf = off[P_FW]; g.bf = off[P_FB];
    g.wo = off[P_OW]; g.bo = off[P_OB]; g.total = o;
    return g;
}

// activation store of one level (floats): natural row-major arrays, one per layer
struct GenActs {
    size_t pex, h[D], feat, ped[1 + VIPNERF_MAX_SEC], g[1 + VIPNERF_MAX_SEC], q[1 + VIPNERF_MAX_SEC], total;    // q: [P][4] head outputs per direction
};
__host__ __device__ inline GenActs gen_acts(size_t P, int V, const GenTopo &t) {
    GenActs a;
    size_t o = 0;
    a.pex = o; o += P * t.dp;
    for (int i = 0; i < D; ++i) { a.h[i] = o; if (i < t.D) o += P * t.W; }
    a.feat = o; o += P * t.W;
    // the 1 + V direction encodings are contiguous ([a][P][dv]: k_gen_encode strides by P * dv)
    for (int k = 0; k <= VIPNERF_MAX_SEC; ++k) { a.ped[k] = o; if (k <= V) o += P * t.dv; }
    for (int k = 0; k <= VIPNERF_MAX_SEC; ++k) { a.g[k] = o; if (k <= V) o += P * (t.W / 2); }
    for (int k = 0; k <= VIPNERF_MAX_SEC; ++k) { a.q[k] = o; if (k <= V) o += P * 4; }
    a.total = (o + 63) & ~(size_t)63;
    return a;
}
// backward scratch of one level (floats)
struct GenBwd {
    size_t dh[2], dfeat, dg, dq[1 + VIPNERF_MAX_SEC], dsraw, dsig, drgb, dvis, dvis2, dqt, total;   // dqt: [P][4] d(trunk head pre-activations)
};
__host__ __device__ inline GenBwd gen_bwd(size_t P, int V, const GenTopo &t) {
    GenBwd b;
    size_t o = 0;
    b.dh[0] = o; o += P * t.W;
    b.dh[1] = o; o += P * t.W;
    b.dfeat = o; o += P * t.W;
    b.dg = o; o += P * (t.W / 2);
    for (int k = 0; k <= VIPNERF_MAX_SEC; ++k) { b.dq[k] = o; if (k <= V) o += P * 4; }
    b.dsraw = o; o += P;
    b.dsig = o; o += P;
    b.drgb = o; o += 3 * P;
    b.dvis = o; o += P;
    b.dvis2 = o; o += P * (V > 0 ? V : 1);
    b.dqt = o; o += P * 4;
    b.total = (o + 63) & ~(size_t)63;
    return b;
}

int launch_gen_fwd(const GenTopo &t, const PointSrc &s, const NoiseSrc &ns, const float *flat_params, float *sigma, float *rgb,
                   float *vis, float *vis2, float *acts, hipStream_t st);
int launch_gen_bwd(const GenTopo &t, const PointSrc &s, const float *flat_params, const float *sigma, const float *rgb, const float *acts, float *bwd,
                   const GenBwd &bl, const vipnerf_mlp_grads *G, hipStream_t st);
int launch_gen_pack(const GenTopo &t, const vipnerf_mlp_params *p, float *flat, hipStream_t st);

}  // namespace vn
