// Visibility-prior generator: the plane-sweep-volume visibility weights of
// reference src/prior_generators/visibility/VisibilityMask02_NeRF_LLFF.py:27-162 -- for every pixel of frame 1,
// warp frame 2 to it at D inverse-depth planes (bilinear, zero padding, validity-normalised), take the minimum
// over planes of the channel-mean absolute error, w = exp(-e_min / T); mask = w > 0.5 (:275-279).
//
// One thread per pixel walks the D planes with a running minimum: 3 B in + 4 taps x 3 B gathered per plane from
// an image that stays L2-resident (a 756x1008 frame is 2.3 MB), 5 B out per pixel -> latency/L2-gather bound, not
// HBM bound.  All geometry is float64 like the reference (numpy promotes everything to float64 there), so that
// weights agree to ~1e-12 and masks are identical away from exact ties.
#include "vipnerf_common.h"

namespace vn {

struct PsvArgs {
    vipnerf_psv p;
    double *weights64;
    float *weights32;
    uint8_t *mask;
};

__global__ void k_psv(PsvArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int h = a.p.height, w = a.p.width;
    if (i >= (int64_t)h * w) return;
    const int yi = (int)(i / w), xi = (int)(i % w);
    const double gx = (double)xi, gy = (double)yi;
    const double *K1 = a.p.k1_inv, *T = a.p.transform, *K2 = a.p.k2;
    // unnormalised ray of the pixel: K1^-1 [x, y, 1]
    double ray[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) ray[r] = (K1[3 * r] * gx + K1[3 * r + 1] * gy) + K1[3 * r + 2] * 1.0;
    float f1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) f1[c] = (float)a.p.frame1[3 * i + c];
    double best = 1e300;
    for (int d = 0; d < a.p.n_planes; ++d) {
        const double z = a.p.planes[d];
        const double wp[3] = {z * ray[0], z * ray[1], z * ray[2]};
        double tw[3], pr[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) tw[r] = ((T[4 * r] * wp[0] + T[4 * r + 1] * wp[1]) + T[4 * r + 2] * wp[2]) + T[4 * r + 3] * 1.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) pr[r] = (K2[3 * r] * tw[0] + K2[3 * r + 1] * tw[1]) + K2[3 * r + 2] * tw[2];
        const double cx = pr[0] / pr[2], cy = pr[1] / pr[2];
        // flow12 = coords - grid ; trans_pos = flow12 + grid   (kept: the round trip is not exact in float64)
        const double tx = (cx - gx) + gx, ty = (cy - gy) + gy;
        double ox = tx + 1.0, oy = ty + 1.0;
        // numpy: floor/ceil -> astype(int) BEFORE the clip of the float offsets; then all three are clipped
        double fxd = floor(ox), cxd = ceil(ox), fyd = floor(oy), cyd = ceil(oy);
        const double wmax = (double)(w + 1), hmax = (double)(h + 1);
        // guard the int conversion of wild coordinates (the reference's astype(int) of huge values is clipped anyway)
        fxd = fmin(fmax(fxd, -4.0), wmax + 4.0); cxd = fmin(fmax(cxd, -4.0), wmax + 4.0);
        fyd = fmin(fmax(fyd, -4.0), hmax + 4.0); cyd = fmin(fmax(cyd, -4.0), hmax + 4.0);
        ox = fmin(fmax(ox, 0.0), wmax); oy = fmin(fmax(oy, 0.0), hmax);
        const int fx = (int)fmin(fmax(fxd, 0.0), wmax), cxi = (int)fmin(fmax(cxd, 0.0), wmax);
        const int fy = (int)fmin(fmax(fyd, 0.0), hmax), cyi = (int)fmin(fmax(cyd, 0.0), hmax);
        const double w_nw = (1.0 - (oy - (double)fy)) * (1.0 - (ox - (double)fx));
        const double w_sw = (1.0 - ((double)cyi - oy)) * (1.0 - (ox - (double)fx));
        const double w_ne = (1.0 - (oy - (double)fy)) * (1.0 - ((double)cxi - ox));
        const double w_se = (1.0 - ((double)cyi - oy)) * (1.0 - ((double)cxi - ox));
        // frame2 padded by one zero pixel on every side: padded (r, c) -> frame2[r-1][c-1]
        auto tap = [&](int r, int c, float out[3]) -> double {
            const bool in = r >= 1 && r <= h && c >= 1 && c <= w;
            if (in) {
                const uint8_t *px = a.p.frame2 + 3 * ((int64_t)(r - 1) * w + (c - 1));
                out[0] = (float)px[0]; out[1] = (float)px[1]; out[2] = (float)px[2];
            } else { out[0] = out[1] = out[2] = 0.f; }
            return in ? 1.0 : 0.0;
        };
        float p_nw[3], p_sw[3], p_ne[3], p_se[3];
        const double m_nw = tap(fy, fx, p_nw), m_sw = tap(cyi, fx, p_sw), m_ne = tap(fy, cxi, p_ne), m_se = tap(cyi, cxi, p_se);
        const double dr = ((w_nw * m_nw + w_sw * m_sw) + w_ne * m_ne) + w_se * m_se;
        double err = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double nr = ((w_nw * (double)p_nw[c] * m_nw + w_sw * (double)p_sw[c] * m_sw) + w_ne * (double)p_ne[c] * m_ne) + w_se * (double)p_se[c] * m_se;
            const double warped = dr > 0.0 ? nr / dr : 0.0;
            err += fabs(warped - (double)f1[c]);
        }
        err = err / 3.0;
        best = fmin(best, err);
    }
    const double wv = exp(-best / a.p.temperature);
    if (a.weights64) a.weights64[i] = wv;
    if (a.weights32) a.weights32[i] = (float)wv;
    if (a.mask) a.mask[i] = wv > 0.5 ? 1 : 0;
}

int launch_psv(const vipnerf_psv *p, double *weights64, float *weights32, uint8_t *mask, hipStream_t st) {
    const int64_t n = (int64_t)p->height * p->width;
    if (n <= 0) return VIPNERF_OK;
    PsvArgs a;
    a.p = *p; a.weights64 = weights64; a.weights32 = weights32; a.mask = mask;
    hipLaunchKernelGGL(k_psv, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, a);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

}  // namespace vn
