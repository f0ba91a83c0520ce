// Per-ray kernels around the MLP: stratified depths, alpha compositing (forward and backward), inverse-CDF
// importance sampling with the sorted merge, and the fused losses.  All are one-wavefront-per-ray: the S
// samples of a ray are spread over the 64 lanes (S/64 consecutive samples per lane, so global accesses are
// contiguous per wave) and the transmittance product / suffix sums are wavefront scans.
// These stages are < 1.5 % of the path's work and HBM/latency bound.
#include "vipnerf_ray.h"
#include <cstring>

namespace vn {

constexpr int RAY_WG = 256;          // 4 rays per workgroup
constexpr int MAX_IPL = 4;           // samples per lane: S <= 256

// torch.linspace(0, 1, n)[i] as the CPU kernel rounds it (verified bit-for-bit against torch 2.10 for n = 64,
// 128): start + step*i for the lower half, end - step*(n-1-i) for the upper, each as ONE fused multiply-add.
__device__ __forceinline__ float linspace01(int i, int n) {
    const float step = __fdiv_rn(1.0f, (float)(n - 1));
    return (i < n / 2) ? __fmaf_rn(step, (float)i, 0.0f) : __fmaf_rn(-step, (float)(n - 1 - i), 1.0f);
}

// ------------------------------------------------------------------------------------------- coarse depths
// VipNeRF.get_z_vals_coarse (VipNeRF01.py:173-203)
// so (vipnerf_train_step): the ray's first 3 (nf - 1) sample threads also write the centres of the OTHER cameras of its row (k_secondary_origins' values:
// VipNeRF01.py:88-98) -- the step's first launch does both jobs.
__global__ void k_coarse_z(int64_t N, int S, int lindisp, const float *near, const float *far,
                           const float *t_rand, int device_rng, uint64_t seed, uint64_t offset, uint64_t ray_base,
                           const int64_t *ray_ids, float *z_out, SecOriginArgs so) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * S) return;
    const int64_t n = idx / S;
    const int k = (int)(idx % S);
    if (so.rays_o2 && k < 3 * (so.nf - 1)) {
        const int v = k / 3, c = k % 3;
        const int f = so.idx64 ? (int)((const int64_t *)so.pixel_id)[3 * n] : (int)((const int32_t *)so.pixel_id)[3 * n];
        so.rays_o2[3 * (n * (so.nf - 1) + v) + c] = so.poses[(size_t)(v + (v >= f ? 1 : 0)) * 16 + 3 + 4 * c];
    }
    const float nr = near[n], fr = far[n];
    auto zk = [&](int i) {
        const float t = linspace01(i, S);
        const float omt = __fsub_rn(1.0f, t);
        if (!lindisp) return __fadd_rn(__fmul_rn(nr, omt), __fmul_rn(fr, t));
        return __fdiv_rn(1.0f, __fadd_rn(__fmul_rn(__fdiv_rn(1.0f, nr), omt), __fmul_rn(__fdiv_rn(1.0f, fr), t)));
    };
    float z = zk(k);
    if (t_rand || device_rng) {
        const float zl = k > 0 ? zk(k - 1) : z, zu = k < S - 1 ? zk(k + 1) : z;
        const float lo = k > 0 ? __fmul_rn(0.5f, __fadd_rn(z, zl)) : z;
        const float hi = k < S - 1 ? __fmul_rn(0.5f, __fadd_rn(zu, z)) : z;
        const float t = t_rand ? t_rand[idx] : rng_uniform(seed, offset, RS_TRAND, (ray_ids ? (uint64_t)ray_ids[n] : ray_base + (uint64_t)n) * (uint64_t)S + (uint64_t)k);
        z = __fadd_rn(lo, __fmul_rn(__fsub_rn(hi, lo), t));
    }
    z_out[idx] = z;
}

// ------------------------------------------------------------------------------------------- compositing
// VipNeRF.volume_rendering + convert_depth_from_ndc (VipNeRF01.py:331-403).
__device__ __forceinline__ float metric_depth(float z_ndc, float oz, float dz) {
    const float tn = __fdiv_rn(-(1.f + oz), dz);
    const float c = (z_ndc == 1.f) ? 1e-3f : 0.f;
    const float a = __fdiv_rn(__fadd_rn(oz, __fmul_rn(tn, dz)), dz);
    const float b = __fsub_rn(__fdiv_rn(1.f, __fadd_rn(__fsub_rn(1.f, z_ndc), c)), 1.f);
    return __fadd_rn(__fmul_rn(a, b), tn);
}

template <int IPL>
__device__ __forceinline__ void composite_body(const CompositeArgs &a, int64_t n, int lane) {
    const int S = a.S, V = a.V;
    const int k0 = lane * IPL;
    const float *zr = a.lvl.z_vals + n * S, *sg = a.lvl.raw_sigma + n * S;
    const float *dn = a.rays_d_s + 3 * n;
    const float dnorm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dn[0], dn[0]), __fmul_rn(dn[1], dn[1])), __fmul_rn(dn[2], dn[2])));
    const float oz = a.rays_o[3 * n + 2], dz = a.rays_d[3 * n + 2];

    float z[IPL + 1], al[IPL], T[IPL], w[IPL];
    bool act[IPL];
#pragma unroll
    for (int i = 0; i < IPL; ++i) { act[i] = k0 + i < S; z[i] = act[i] ? zr[k0 + i] : 0.f; }
    {
        const float znext = __shfl_down(z[0], 1, 64);         // first sample of the next lane
        z[IPL] = znext;
    }
    const float zlast = a.ndc ? 1.f : 1e10f;
    float lp = 1.f;                                           // product of this lane's (1 - alpha + 1e-10)
    float av[IPL];
#pragma unroll
    for (int i = 0; i < IPL; ++i) {
        const float zn = (k0 + i == S - 1) ? zlast : z[i + 1];
        const float delta = __fmul_rn(__fsub_rn(zn, z[i]), dnorm);
        al[i] = act[i] ? __fsub_rn(1.f, expf(-__fmul_rn(sg[act[i] ? k0 + i : 0], delta))) : 0.f;
        av[i] = act[i] ? __fadd_rn(__fsub_rn(1.f, al[i]), 1e-10f) : 1.f;
        lp = __fmul_rn(lp, av[i]);
    }
    const float incl = wave_scan_mul(lp, lane);
    float run = __shfl_up(incl, 1, 64);                       // exclusive prefix across lanes
    if (lane == 0) run = 1.f;
    float s_rgb[3] = {0.f, 0.f, 0.f}, s_acc = 0.f, s_z = 0.f, s_zm = 0.f, s_v2[VIPNERF_MAX_SEC] = {0.f, 0.f, 0.f};
    float zm[IPL];
#pragma unroll
    for (int i = 0; i < IPL; ++i) {
        T[i] = run;
        w[i] = __fmul_rn(al[i], T[i]);
        run = __fmul_rn(run, av[i]);
        zm[i] = a.ndc ? metric_depth(z[i], oz, dz) : z[i];
        if (act[i]) {
            const int64_t ps = n * S + k0 + i;
            a.lvl.alpha[ps] = al[i]; a.lvl.visibility[ps] = T[i]; a.lvl.weights[ps] = w[i];
            s_acc += w[i];
            s_z += w[i] * z[i];
            s_zm += w[i] * zm[i];
#pragma unroll
            for (int c = 0; c < 3; ++c) s_rgb[c] += w[i] * a.lvl.raw_rgb[3 * ps + c];
            for (int v = 0; v < V; ++v) s_v2[v] += w[i] * a.lvl.raw_vis2[ps * V + v];
        }
    }
    s_acc = wave_sum(s_acc); s_z = wave_sum(s_z); s_zm = wave_sum(s_zm);
#pragma unroll
    for (int c = 0; c < 3; ++c) s_rgb[c] = wave_sum(s_rgb[c]);
    for (int v = 0; v < V; ++v) s_v2[v] = wave_sum(s_v2[v]);
    const float den = __fadd_rn(s_acc, 1e-6f);
    const float d_s = __fdiv_rn(s_z, den), d_m = __fdiv_rn(s_zm, den);
    float var_s = 0.f, var_m = 0.f;
#pragma unroll
    for (int i = 0; i < IPL; ++i)
        if (act[i]) {
            const float e1 = z[i] - d_s, e2 = zm[i] - d_m;
            var_s += w[i] * (e1 * e1);
            var_m += w[i] * (e2 * e2);
        }
    var_s = wave_sum(var_s); var_m = wave_sum(var_m);
    if (lane == 0) {
        const float bg = a.white_bkgd ? 1.f - s_acc : 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) a.lvl.rgb[3 * n + c] = s_rgb[c] + bg;
        a.lvl.acc[n] = s_acc;
        a.lvl.depth[n] = d_m;
        a.lvl.depth_var[n] = var_m;
        if (a.ndc) {
            if (a.lvl.depth_ndc) a.lvl.depth_ndc[n] = d_s;
            if (a.lvl.depth_var_ndc) a.lvl.depth_var_ndc[n] = var_s;
        }
        for (int v = 0; v < V; ++v) a.lvl.vis2[n * V + v] = __fdiv_rn(s_v2[v], den);
    }
}
template <int IPL>
__global__ __launch_bounds__(RAY_WG) void k_composite(CompositeArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * (RAY_WG / 64) + (threadIdx.x >> 6);
    if (n >= a.N) return;
    composite_body<IPL>(a, n, lane);
}

int launch_composite(const CompositeArgs &a, hipStream_t st) {
    if (a.N <= 0) return VIPNERF_OK;
    const unsigned grid = (unsigned)((a.N + 3) / 4);
    const int ipl = (a.S + 63) / 64;
    switch (ipl) {
        case 1: hipLaunchKernelGGL(k_composite<1>, dim3(grid), dim3(RAY_WG), 0, st, a); break;
        case 2: hipLaunchKernelGGL(k_composite<2>, dim3(grid), dim3(RAY_WG), 0, st, a); break;
        case 3: hipLaunchKernelGGL(k_composite<3>, dim3(grid), dim3(RAY_WG), 0, st, a); break;
        case 4: hipLaunchKernelGGL(k_composite<4>, dim3(grid), dim3(RAY_WG), 0, st, a); break;
        default: set_error("composite: n_samples %d > 256", a.S); return VIPNERF_E_UNSUPPORTED;
    }
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

// Backward of the compositing (SURVEY.md §9 "Compositing backward", extended to every differentiable output):
//   G_k = dL/dw_k = gw_k + g_rgb.c_k (- sum g_rgb if white_bkgd) + g_acc + g_depth (zm_k - D)/(A+eps)
//         + g_depth_ndc (z_k - D')/(A+eps) + sum_v g_vis2_v (v2_kv - V2_v)/(A+eps)
//         + g_depth_var [(zm_k - D)^2 - 2 (zm_k - D)/(A+eps) sum_j w_j (zm_j - D)]  (same for depth_var_ndc on z)
//   R_k = sum_{j>k} (G_j w_j + gT_j T_j)
//   dL/dalpha_k = G_k T_k + galpha_k - R_k / (1 - alpha_k + 1e-10)
//   dL/dsigma_k = dL/dalpha_k * delta_k * (1 - alpha_k)  (+ direct)
template <int IPL>
__global__ __launch_bounds__(RAY_WG) void k_composite_bwd(CompositeBwdArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * (RAY_WG / 64) + (threadIdx.x >> 6);
    if (n >= a.N) return;
    const int S = a.S, V = a.V;
    const int k0 = lane * IPL;
    const vipnerf_level_grads &g = a.g;
    const float *dn = a.rays_d_s + 3 * n;
    const float dnorm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(dn[0], dn[0]), __fmul_rn(dn[1], dn[1])), __fmul_rn(dn[2], dn[2])));
    const float oz = a.rays_o[3 * n + 2], dz = a.rays_d[3 * n + 2];
    const float A = a.lvl.acc[n], den = __fadd_rn(A, 1e-6f);
    float g_rgb[3] = {0.f, 0.f, 0.f}, g_v2[VIPNERF_MAX_SEC] = {0.f, 0.f, 0.f}, V2[VIPNERF_MAX_SEC] = {0.f, 0.f, 0.f};
    if (g.rgb) { g_rgb[0] = g.rgb[3 * n]; g_rgb[1] = g.rgb[3 * n + 1]; g_rgb[2] = g.rgb[3 * n + 2]; }
    const float g_acc = (g.acc ? g.acc[n] : 0.f) - (a.white_bkgd ? (g_rgb[0] + g_rgb[1] + g_rgb[2]) : 0.f);
    const float g_dep = g.depth ? g.depth[n] : 0.f;
    const float g_dnd = (g.depth_ndc && a.ndc) ? g.depth_ndc[n] : 0.f;
    const float g_var = g.depth_var ? g.depth_var[n] : 0.f;
    const float g_vnd = (g.depth_var_ndc && a.ndc) ? g.depth_var_ndc[n] : 0.f;
    const float Dm = a.lvl.depth[n];
    const float Ds = (a.ndc && a.lvl.depth_ndc) ? a.lvl.depth_ndc[n] : 0.f;
    for (int v = 0; v < V; ++v) { g_v2[v] = g.vis2 ? g.vis2[n * V + v] : 0.f; V2[v] = a.lvl.vis2[n * V + v]; }

    float z[IPL + 1], Gw[IPL], Tt[IPL], ww[IPL], term[IPL];
    bool act[IPL];
#pragma unroll
    for (int i = 0; i < IPL; ++i) { act[i] = k0 + i < S; z[i] = act[i] ? a.lvl.z_vals[n * S + k0 + i] : 0.f; }
    z[IPL] = __shfl_down(z[0], 1, 64);
    const float zlast = a.ndc ? 1.f : 1e10f;
    float lsum = 0.f;
    // depth_var = sum_j w_j (zeta_j - D)^2 also depends on w_k through D = sum_j w_j zeta_j / (A + eps):
    // d var / d w_k = (zeta_k - D)^2 - 2 (zeta_k - D)/(A + eps) * sum_j w_j (zeta_j - D)
    float sm = 0.f, ss = 0.f;
    if (g.depth_var || g.depth_var_ndc) {
#pragma unroll
        for (int i = 0; i < IPL; ++i)
            if (act[i]) {
                const float wv_ = a.lvl.weights[n * S + k0 + i];
                sm += wv_ * ((a.ndc ? metric_depth(z[i], oz, dz) : z[i]) - Dm);
                ss += wv_ * (z[i] - Ds);
            }
        sm = wave_sum(sm); ss = wave_sum(ss);
    }
#pragma unroll
    for (int i = 0; i < IPL; ++i) {
        Gw[i] = 0.f; Tt[i] = 0.f; ww[i] = 0.f; term[i] = 0.f;
        if (act[i]) {
            const int64_t ps = n * S + k0 + i;
            Tt[i] = a.lvl.visibility[ps]; ww[i] = a.lvl.weights[ps];
            const float zmv = a.ndc ? metric_depth(z[i], oz, dz) : z[i];
            float G = g_acc + (g.weights ? g.weights[ps] : 0.f);
#pragma unroll
            for (int c = 0; c < 3; ++c) G += g_rgb[c] * a.lvl.raw_rgb[3 * ps + c];
            G += g_dep * (zmv - Dm) / den;
            G += g_dnd * (z[i] - Ds) / den;
            G += g_var * ((zmv - Dm) * (zmv - Dm) - 2.f * sm * (zmv - Dm) / den);
            G += g_vnd * ((z[i] - Ds) * (z[i] - Ds) - 2.f * ss * (z[i] - Ds) / den);
            for (int v = 0; v < V; ++v) G += g_v2[v] * (a.lvl.raw_vis2[ps * V + v] - V2[v]) / den;
            Gw[i] = G;
            term[i] = G * ww[i] + (g.visibility ? g.visibility[ps] * Tt[i] : 0.f);
            lsum += term[i];
        }
    }
    const float incl = wave_rscan_add(lsum, lane);            // this lane's + all later lanes' terms
    float after = incl - lsum;                                // strictly later lanes
#pragma unroll
    for (int i = IPL - 1; i >= 0; --i) {
        if (act[i]) {
            const int64_t ps = n * S + k0 + i;
            const float R = after;                            // sum over j > k
            after += term[i];
            const float alpha = a.lvl.alpha[ps];
            const float zn = (k0 + i == S - 1) ? zlast : z[i + 1];
            const float delta = __fmul_rn(__fsub_rn(zn, z[i]), dnorm);
            const float av = __fadd_rn(__fsub_rn(1.f, alpha), 1e-10f);
            const float dalpha = Gw[i] * Tt[i] + (g.alpha ? g.alpha[ps] : 0.f) - R / av;
            float dsig = dalpha * delta * (1.f - alpha);
            if (g.raw_sigma) dsig += g.raw_sigma[ps];
            a.dsig[ps] = dsig;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                a.drgb[3 * ps + c] = g_rgb[c] * ww[i] + (g.raw_rgb ? g.raw_rgb[3 * ps + c] : 0.f);
            a.dvis[ps] = g.raw_vis ? g.raw_vis[ps] : 0.f;
            for (int v = 0; v < V; ++v)
                a.dvis2[ps * V + v] = g_v2[v] * ww[i] / den + (g.raw_vis2 ? g.raw_vis2[ps * V + v] : 0.f);
        }
    }
}

int launch_composite_bwd(const CompositeBwdArgs &a, hipStream_t st) {
    if (a.N <= 0) return VIPNERF_OK;
    const unsigned grid = (unsigned)((a.N + 3) / 4);
    const int ipl = (a.S + 63) / 64;
    switch (ipl) {
        case 1: hipLaunchKernelGGL(k_composite_bwd<1>, dim3(grid), dim3(RAY_WG), 0, st, a); break;
        case 2: hipLaunchKernelGGL(k_composite_bwd<2>, dim3(grid), dim3(RAY_WG), 0, st, a); break;
        case 3: hipLaunchKernelGGL(k_composite_bwd<3>, dim3(grid), dim3(RAY_WG), 0, st, a); break;
        case 4: hipLaunchKernelGGL(k_composite_bwd<4>, dim3(grid), dim3(RAY_WG), 0, st, a); break;
        default: set_error("composite_bwd: n_samples %d > 256", a.S); return VIPNERF_E_UNSUPPORTED;
    }
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

// ------------------------------------------------------------------------------------------- importance sampling
// VipNeRF.get_z_vals_fine + sample_pdf (VipNeRF01.py:205-262).  One wave per ray; LDS per wave:
// cdf[Sc-1], bins[Sc-1], merged values [Sc+Sf].
__device__ __forceinline__ void sample_fine_body(const SampleArgs &a, float *sl, int lane, int wv, int64_t n) {
    const int Sc = a.Sc, Sf = a.Sf, NB = Sc - 1, NW = Sc - 2, ST = Sc + Sf;
    float *cdf = sl + wv * (2 * NB + ST), *bins = cdf + NB, *vals = bins + NB;
    const bool live = n < a.N;
    const int64_t nn = live ? n : a.N - 1;
    const float *zc = a.z_coarse + nn * Sc, *wc = a.w_coarse + nn * Sc;

    // pdf weights w[1..Sc-2] + 1e-5, total, serial cumsum (fp32, left to right like torch.cumsum)
    for (int k = lane; k < NB; k += 64) bins[k] = __fmul_rn(0.5f, __fadd_rn(zc[k + 1], zc[k]));
    for (int k = lane; k < Sc; k += 64) vals[k] = zc[k];
    for (int k = lane; k < NW; k += 64) cdf[1 + k] = __fadd_rn(wc[1 + k], 1e-5f);   // stash w in cdf[1..]
    __builtin_amdgcn_s_waitcnt(0);                // LDS is per wave: order only within the wave
    __builtin_amdgcn_wave_barrier();
    // total = torch.sum(w, -1) in the association order of ATen's CPU kernel (8 vector lanes x 4-way ILP,
    // scalar tail added first; verified bit-for-bit), so that pdf, cdf and the searchsorted indices are
    // bit-identical to the reference's on the same weights.
    float tot;
    {
        const int nv = NW / 8, nblk = nv / 4;
        float part = 0.f;
        if (lane < 8) {
            float p4[4] = {0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < nblk; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) p4[k] = __fadd_rn(p4[k], cdf[1 + (i * 4 + k) * 8 + lane]);
            for (int i = nblk * 4; i < nv; ++i) p4[0] = __fadd_rn(p4[0], cdf[1 + i * 8 + lane]);
            part = __fadd_rn(__fadd_rn(__fadd_rn(p4[0], p4[1]), p4[2]), p4[3]);
        }
        float f = 0.f;
        for (int k = nv * 8; k < NW; ++k) f = __fadd_rn(f, cdf[1 + k]);
        for (int i = 0; i < 8; ++i) f = __fadd_rn(f, __shfl(part, i, 64));
        tot = f;
    }
    // cdf = torch.cumsum(pdf): the CPU kernel accumulates float inputs in double and rounds each output.  The divisions in parallel
    // (lane k + 64 i holds pdf[k + 64 i]), then the serial chain on registers alone: v_readlane of the next pdf, one double add, the
    // rounded sum kept by the lane that owns the element -- no LDS round trip and no division inside the chain (it was 62 x ~400 cycles).
    {
        constexpr int MAXR = 16;                   // NW <= 1021 (vipnerf_sample_fine: n_coarse + n_fine <= 1024)
        float pdfr[MAXR], outr[MAXR];
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int k = lane + 64 * i;
            pdfr[i] = k < NW ? __fdiv_rn(cdf[1 + k], tot) : 0.f;
            outr[i] = 0.f;
        }
        double run = 0.0;
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int cntk = min(64, NW - 64 * i);
            for (int l = 0; l < cntk; ++l) {
                run += (double)__int_as_float(__builtin_amdgcn_readlane(__float_as_int(pdfr[i]), l));
                if (lane == l) outr[i] = (float)run;
            }
        }
        __builtin_amdgcn_wave_barrier();          // every lane has read its w before anyone overwrites the slots
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int k = lane + 64 * i;
            if (k < NW) cdf[1 + k] = outr[i];
        }
    }
    if (lane == 0) cdf[0] = 0.f;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();

    // z_coarse ascending (it is, for near <= far: stratified samples of increasing bins)?  Then a value's rank among the coarse depths is a
    // binary search instead of Sc compares; any other input takes the general count below.
    bool asc = true;
    for (int k = lane; k + 1 < Sc; k += 64) asc = asc && (vals[k] <= vals[k + 1]);
    asc = __all(asc);

    for (int jj = lane; jj < Sf; jj += 64) {
        float u;
        if (a.u) u = a.u[nn * Sf + jj];
        else if (a.device_rng) u = rng_uniform(a.seed, a.offset, RS_U, (a.ray_ids ? (uint64_t)a.ray_ids[nn] : a.ray_base + (uint64_t)nn) * (uint64_t)Sf + (uint64_t)jj);
        else u = linspace01(jj, Sf);
        // searchsorted(cdf, u, right=True) = #{cdf <= u}: the cdf is a rounded running sum of positive terms, i.e. non-decreasing -- upper bound
        // by bisection (8 steps for 255 bins instead of 255 compares); a NaN cdf (NaN weights) compares false everywhere in both forms' first step
        int lo_i = 0, hi_i = NB;
        while (lo_i < hi_i) {
            const int mid = (lo_i + hi_i) >> 1;
            if (cdf[mid] <= u) lo_i = mid + 1; else hi_i = mid;
        }
        const int cnt = lo_i;
        const int lo = max(cnt - 1, 0), hi = min(cnt, NB - 1);
        const float cl = cdf[lo], ch = cdf[hi], bl = bins[lo], bh = bins[hi];
        float dnm = __fsub_rn(ch, cl);
        if (dnm < 1e-5f) dnm = 1.f;
        const float t = __fdiv_rn(__fsub_rn(u, cl), dnm);
        const float smp = __fadd_rn(bl, __fmul_rn(t, __fsub_rn(bh, bl)));
        vals[Sc + jj] = smp;
        if (live) {
            if (a.inds) a.inds[n * Sf + jj] = cnt;
            if (a.z_samples) a.z_samples[n * Sf + jj] = smp;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    // stable rank sort of the Sc+Sf values (torch.sort of the concatenation, values only): rank = #{o < v} + #{o == v, earlier index}
    for (int i = lane; i < ST; i += 64) {
        const float v = vals[i];
        int rank = 0;
        int k0 = 0;
        if (asc) {
            // among the ascending coarse depths: a coarse element's rank is its own index (ties are contiguous and the earlier ones are exactly
            // those before it); a sample's is #{o <= v} (every coarse index is earlier)
            if (i < Sc) rank = i;
            else {
                int lo_i = 0, hi_i = Sc;
                while (lo_i < hi_i) {
                    const int mid = (lo_i + hi_i) >> 1;
                    if (vals[mid] <= v) lo_i = mid + 1; else hi_i = mid;
                }
                rank = lo_i;
            }
            k0 = Sc;
        }
#pragma unroll 8
        for (int k = k0; k < ST; ++k) {           // (unrolled: eight LDS reads in flight instead of one)
            const float o = vals[k];
            rank += (o < v || (o == v && k < i)) ? 1 : 0;
        }
        if (live) a.z_fine[n * ST + rank] = v;
    }
}
__global__ __launch_bounds__(RAY_WG) void k_sample_fine(SampleArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sl[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    sample_fine_body(a, sl, lane, wv, (int64_t)blockIdx.x * (RAY_WG / 64) + wv);
}
// The coarse level's compositing and the importance sampling it feeds in ONE launch: a wave composites its ray (k_composite's body), then samples from the
// weights it has just written (k_sample_fine's body; both are one wave per ray with no cross-wave step).  The same values as the two launches.
template <int IPL>
__global__ __launch_bounds__(RAY_WG) void k_composite_sample(CompositeArgs c, SampleArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sl[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * (RAY_WG / 64) + wv;
    if (n < c.N) composite_body<IPL>(c, n, lane);
    __threadfence_block();                        // the ray's weights: stored above by the lanes of this wave, read below by other lanes of it
    __builtin_amdgcn_wave_barrier();
    sample_fine_body(a, sl, lane, wv, n);
}

int launch_sample_fine(const SampleArgs &a, hipStream_t st) {
    if (a.N <= 0) return VIPNERF_OK;
    const unsigned grid = (unsigned)((a.N + 3) / 4);
    const size_t lds = (size_t)4 * (2 * (a.Sc - 1) + a.Sc + a.Sf) * sizeof(float);
    hipLaunchKernelGGL(k_sample_fine, dim3(grid), dim3(RAY_WG), lds, st, a);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}
int launch_composite_sample(const CompositeArgs &c, const SampleArgs &a, hipStream_t st) {
    if (a.N <= 0) return VIPNERF_OK;
    if (c.N != a.N || c.S != a.Sc || c.lvl.weights != a.w_coarse || c.lvl.z_vals != a.z_coarse) { set_error("composite_sample: the two stages disagree"); return VIPNERF_E_ARG; }
    const unsigned grid = (unsigned)((a.N + 3) / 4);
    const size_t lds = (size_t)4 * (2 * (a.Sc - 1) + a.Sc + a.Sf) * sizeof(float);
    switch ((c.S + 63) / 64) {
        case 1: hipLaunchKernelGGL(k_composite_sample<1>, dim3(grid), dim3(RAY_WG), lds, st, c, a); break;
        case 2: hipLaunchKernelGGL(k_composite_sample<2>, dim3(grid), dim3(RAY_WG), lds, st, c, a); break;
        case 3: hipLaunchKernelGGL(k_composite_sample<3>, dim3(grid), dim3(RAY_WG), lds, st, c, a); break;
        case 4: hipLaunchKernelGGL(k_composite_sample<4>, dim3(grid), dim3(RAY_WG), lds, st, c, a); break;
        default: set_error("composite: n_samples %d > 256", c.S); return VIPNERF_E_UNSUPPORTED;
    }
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

int launch_coarse_z(int64_t N, int S, int lindisp, const float *near, const float *far, const float *t_rand,
                    int device_rng, uint64_t seed, uint64_t offset, uint64_t ray_base, const int64_t *ray_ids, float *z_out,
                    hipStream_t st, const SecOriginArgs *so) {
    if (N <= 0) return VIPNERF_OK;
    if (so && so->rays_o2 && 3 * (so->nf - 1) > S) { set_error("coarse_z: %d secondary views need %d sample threads per ray, S = %d", so->nf - 1, 3 * (so->nf - 1), S); return VIPNERF_E_ARG; }
    const int64_t tot = N * S;
    const SecOriginArgs s0 = {nullptr, nullptr, 0, 0, nullptr};
    hipLaunchKernelGGL(k_coarse_z, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, N, S, lindisp, near, far,
                       t_rand, device_rng, seed, offset, ray_base, ray_ids, z_out, so ? *so : s0);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

// ------------------------------------------------------------------------------------------- fused losses
// MSE01 / VisibilityLoss01 / VisibilityPriorLoss01 / SparseDepthMSE01 (src/loss_functions/*.py): values and
// unweighted gradient seeds.  Two launches up to LOSS_INLINE_COUNT_MAX rows (per-ray partials + seeds with the mask counts taken by every
// workgroup itself -> ordered final sum), three beyond (a mask-count launch in front).
constexpr int64_t LOSS_INLINE_COUNT_MAX = 8192;
__global__ void k_loss_counts(int64_t N, const uint8_t *m_nerf, const uint8_t *m_sd, float *counts) {
    __shared__ float sh[2][16];
    float c0 = 0.f, c1 = 0.f;
    for (int64_t i = threadIdx.x; i < N; i += blockDim.x) {
        c0 += m_nerf ? (m_nerf[i] ? 1.f : 0.f) : 1.f;
        c1 += m_sd ? (m_sd[i] ? 1.f : 0.f) : 0.f;
    }
    c0 = wave_sum(c0); c1 = wave_sum(c1);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = c0; sh[1][threadIdx.x >> 6] = c1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a0 = 0.f, a1 = 0.f;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { a0 += sh[0][i]; a1 += sh[1][i]; }
        counts[0] = a0; counts[1] = a1;
    }
}

// one wave per ray; partial[k*N + n], k: 0 mse_c 1 mse_f 2 vis_c 3 vis_f 4 prior_c 5 prior_f 6 sd
// COUNT: every workgroup counts the mask rows itself (N <= LOSS_INLINE_COUNT_MAX: a few KB from L2 per workgroup) instead of a k_loss_counts
// launch in front; the counts are integers below 2^24, so any summation order gives k_loss_counts' floats exactly.  Workgroup 0 leaves them in
// a.counts for k_loss_final.
template <bool COUNT>
__global__ __launch_bounds__(RAY_WG) void k_loss_rays(LossArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * (RAY_WG / 64) + (threadIdx.x >> 6);
    float n_m, n_q;
    if (COUNT) {
        __shared__ float shc[2][RAY_WG / 64];
        float c0 = 0.f, c1 = 0.f;
        for (int64_t i = threadIdx.x; i < a.N; i += RAY_WG) {
            c0 += a.in.mask_nerf ? (a.in.mask_nerf[i] ? 1.f : 0.f) : 1.f;
            c1 += a.in.mask_sparse ? (a.in.mask_sparse[i] ? 1.f : 0.f) : 0.f;
        }
        c0 = wave_sum(c0); c1 = wave_sum(c1);
        if (lane == 0) { shc[0][threadIdx.x >> 6] = c0; shc[1][threadIdx.x >> 6] = c1; }
        __syncthreads();
        n_m = 0.f; n_q = 0.f;
        for (int i = 0; i < RAY_WG / 64; ++i) { n_m += shc[0][i]; n_q += shc[1][i]; }
        if (blockIdx.x == 0 && threadIdx.x == 0) { a.counts[0] = n_m; a.counts[1] = n_q; }
        if (n >= a.N) return;
    } else {
        if (n >= a.N) return;
        n_m = a.counts[0]; n_q = a.counts[1];
    }
    const bool in_m = a.in.mask_nerf ? a.in.mask_nerf[n] != 0 : true;
    const bool in_q = a.in.mask_sparse ? a.in.mask_sparse[n] != 0 : false;
    const int V = a.V;
    for (int lv = 0; lv < a.n_levels; ++lv) {
        const vipnerf_level_out &o = lv ? a.fine : a.coarse;
        const vipnerf_loss_level_seeds &sd = lv ? a.seeds_fine : a.seeds_coarse;
        const int S = lv ? a.S_fine : a.S_coarse;
        // photometric MSE
        float mse = 0.f;
        if (lane < 3) {
            const float e = o.rgb[3 * n + lane] - a.in.target_rgb[3 * n + lane];
            mse = in_m ? e * e : 0.f;
            sd.rgb[3 * n + lane] = (in_m && n_m > 0.f) ? 2.f * e / (3.f * n_m) : 0.f;
        }
        mse = wave_sum(mse) / 3.f;
        // visibility: |T^ - sg(T)| + |sg(T^) - T|, mean over samples then rays (all rays)
        float vl = 0.f;
        const float inv = 1.f / ((float)a.N * (float)S);
        for (int k = lane; k < S; k += 64) {
            const float d = o.raw_vis[n * S + k] - o.visibility[n * S + k];
            vl += fabsf(d);
            const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            sd.raw_vis[n * S + k] = sgn * inv;
            sd.visibility[n * S + k] = -sgn * inv;
        }
        vl = 2.f * wave_sum(vl) / (float)S;
        // visibility prior
        float pr = 0.f;
        if (V > 0) {
            if (lane < V) {
                const float m = a.in.prior ? a.in.prior[n * V + lane] : 1.f;
                pr = in_m ? m * (1.f - o.vis2[n * V + lane]) : 0.f;
                sd.vis2[n * V + lane] = (in_m && n_m > 0.f) ? -m / n_m : 0.f;
            }
            pr = wave_sum(pr);
        }
        if (lane == 0) {
            a.partial[(0 + lv) * a.N + n] = mse;
            a.partial[(2 + lv) * a.N + n] = vl;
            a.partial[(4 + lv) * a.N + n] = pr;
        }
        // sparse depth (on the last level only: fine if present, else coarse)
        if (lv == a.n_levels - 1 && lane == 0) {
            float sdl = 0.f, seed = 0.f;
            if (a.in.mask_sparse && in_q) {
                const float e = o.depth[n] - a.in.sparse_depth[n];
                sdl = e * e;
                seed = n_q > 0.f ? 2.f * e / n_q : 0.f;
            }
            a.partial[6 * a.N + n] = sdl;
            if (sd.depth) sd.depth[n] = seed;
        }
    }
}

// ordered reduction of the 7 partial arrays by ONE workgroup of 1024 / NV threads, each standing for NV of k_loss_final's 1024 threads (thread t + r * blockDim.x
// is its r-th): the same partial sums, folded in the same order, whatever the launch that carries it -- the values are k_loss_final's bit for bit
template <int NV>
__device__ __forceinline__ void loss_final_body(const LossArgs &a, float *sh) {
    const int nthr = 1024 / NV;                                     // == blockDim.x
    for (int k = 0; k < 7; ++k) {
        const bool used = (k == 6) || ((k & 1) < a.n_levels);
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            float s = 0.f;
            if (used)
                for (int64_t i = threadIdx.x + r * nthr; i < a.N; i += 1024) s += a.partial[(size_t)k * a.N + i];
            s = wave_sum(s);
            if ((threadIdx.x & 63) == 0) sh[((threadIdx.x + r * nthr) >> 6)] = s;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float t = 0.f;
            for (int i = 0; i < 16; ++i) t += sh[i];
            float dnm;
            if (k < 2 || k == 4 || k == 5) dnm = a.counts[0];       // mean over nerf rays
            else if (k < 4) dnm = (float)a.N;                        // mean over all rays
            else dnm = a.counts[1];                                  // mean over sparse-depth rays
            a.loss_values[k] = dnm > 0.f ? t / dnm : 0.f;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.loss_values[7] = 0.f;
        if (a.total) {               // TotalLoss in k_scale_segments' order and roundings (thread 0 wrote every value above: its own stores are visible to it)
            float t = 0.f;
            for (int k = 0; k < 8; ++k) t = __fadd_rn(t, __fmul_rn(a.w[k], a.loss_values[k]));
            a.total[0] = t;
        }
        if (a.named)
            for (int j = 0; j < 4; ++j) a.named[j] = __fadd_rn(a.loss_values[2 * j], a.loss_values[2 * j + 1]);
    }
}
__global__ void k_loss_final(LossArgs a) {
    __shared__ float sh[16];
    loss_final_body<1>(a, sh);               // launched with 1024 threads
}

// out_k = g[slot_k] * in_k for all segments in one launch (blockIdx.y = segment): the fused losses' backward
// fin (vipnerf_train_step): the workgroups of row blockIdx.y == a.n are not a segment -- the first of them runs the loss values' final sums
// (loss_final_body: k_loss_final's job, which the seeds x weights do not depend on) in THIS launch
__global__ void k_scale_segments(ScaleArgs a, LossFinalTail fin) {
    if (blockIdx.y == (unsigned)a.n) {
        __shared__ float sh[16];
        if (blockIdx.x == 0) loss_final_body<4>(fin.a, sh);      // (this launch has 256 threads per workgroup)
        return;
    }
    const vipnerf_scale_seg sg = a.s[blockIdx.y];
    const float w = a.g ? a.g[sg.slot] : (a.g1 ? __fmul_rn(a.g1[0], a.w[sg.slot]) : a.w[sg.slot]);
    if (a.total && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        float t = 0.f;
        for (int k = 0; k < 8; ++k) t = __fadd_rn(t, __fmul_rn(a.w[k], a.loss_values[k]));
        a.total[0] = t;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < sg.numel; i += (int64_t)gridDim.x * blockDim.x) sg.out[i] = w * sg.in[i];
}
int launch_scale_segments(const ScaleArgs &a, hipStream_t st, const LossArgs *fin) {
    int64_t most = 0;
    for (int k = 0; k < a.n; ++k) most = a.s[k].numel > most ? a.s[k].numel : most;
    if (a.n <= 0 || most <= 0) {
        if (fin) { set_error("scale_segments: the loss values' final sums ride in a launch that has nothing to scale"); return VIPNERF_E_ARG; }
        return VIPNERF_OK;
    }
    int64_t bx = (most + 1023) / 1024;              // 256 threads x 4 elements each where the segment is that long
    if (bx > 1024) bx = 1024;
    LossFinalTail tail;
    memset((void *)&tail, 0, sizeof(tail));
    if (fin) tail.a = *fin;
    hipLaunchKernelGGL(k_scale_segments, dim3((unsigned)bx, (unsigned)(a.n + (fin ? 1 : 0))), dim3(256), 0, st, a, tail);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

// One Adam step on flat buffers; every operation rounded where torch's separate elementwise kernels round (see include/vipnerf_hip.h)
template <int MASK>
__global__ void k_adam_step(int64_t n, float *p, float *m, float *v, const float *g, float lerp_w, float beta2, float sq_w, float inv_s, float eps,
                            float neg_step) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i], m0 = m[i];
        const float diff = __fsub_rn(gi, m0);
        const float m1 = (MASK & 1) ? fmaf(lerp_w, diff, m0) : __fadd_rn(m0, __fmul_rn(lerp_w, diff));
        const float v0 = __fmul_rn(v[i], beta2), gg = __fmul_rn(gi, gi);              // addcmul is a + alpha * (b * c)
        const float v1 = (MASK & 2) ? fmaf(sq_w, gg, v0) : __fadd_rn(v0, __fmul_rn(sq_w, gg));
        // sqrtf and / are the correctly rounded ones (hipcc's default; HIP's __fsqrt_rn / __fdiv_rn map to the NATIVE instructions unless
        // OCML_BASIC_ROUNDED_OPERATIONS is defined); the library is built with -ffp-contract=off, so nothing below fuses by itself
        const float d = __fadd_rn(__fmul_rn(sqrtf(v1), inv_s), eps);
        const float q = m1 / d;
        p[i] = (MASK & 4) ? fmaf(neg_step, q, p[i]) : __fadd_rn(p[i], __fmul_rn(neg_step, q));
        m[i] = m1;
        v[i] = v1;
    }
}
int launch_adam_step(int64_t n, float *p, float *m, float *v, const float *g, float lerp_w, float beta2, float sq_w, float inv_s, float eps,
                     float neg_step, int mask, hipStream_t st) {
    if (n <= 0) return VIPNERF_OK;
    int64_t nb = (n + 255) / 256;
    if (nb > 2048) nb = 2048;
    const dim3 grid((unsigned)nb), blk(256);
    switch (mask & 7) {
#define ADAM_CASE(M) case M: hipLaunchKernelGGL(k_adam_step<M>, grid, blk, 0, st, n, p, m, v, g, lerp_w, beta2, sq_w, inv_s, eps, neg_step); break;
        ADAM_CASE(0) ADAM_CASE(1) ADAM_CASE(2) ADAM_CASE(3) ADAM_CASE(4) ADAM_CASE(5) ADAM_CASE(6) ADAM_CASE(7)
#undef ADAM_CASE
    }
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

// defer_final: the caller runs the final sums itself (vipnerf_train_step: inside its seeds x weights launch)
int launch_losses(const LossArgs &a, hipStream_t st, bool defer_final) {
    if (a.N <= 0) return VIPNERF_OK;
    if (a.N <= LOSS_INLINE_COUNT_MAX) {
        hipLaunchKernelGGL(k_loss_rays<true>, dim3((unsigned)((a.N + 3) / 4)), dim3(RAY_WG), 0, st, a);
    } else {
        hipLaunchKernelGGL(k_loss_counts, dim3(1), dim3(1024), 0, st, a.N, a.in.mask_nerf, a.in.mask_sparse, a.counts);
        hipLaunchKernelGGL(k_loss_rays<false>, dim3((unsigned)((a.N + 3) / 4)), dim3(RAY_WG), 0, st, a);
    }
    if (!defer_final) hipLaunchKernelGGL(k_loss_final, dim3(1), dim3(1024), 0, st, a);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

}  // namespace vn
