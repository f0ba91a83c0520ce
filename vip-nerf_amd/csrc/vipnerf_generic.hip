// Generic-topology MLP path: any netdepth <= 8, netwidth <= 256 (multiple of 4), positional-encoding degrees, with the
// reference's skip rule (gamma(x) re-enters after layer 4 when that layer is not the last) -- reference
// src/models/VipNeRF01.py:451-596 (MLP.__init__ / forward) for topologies OTHER than the one every shipped config uses,
// e.g. BASELINE configs[0]'s 4x64 coarse-only "plumbing" network.  The fused MFMA kernels (vipnerf_mlp_*.hip) are
// specialised on 8x256 / 10 / 4; this file is the product path for everything else: one launch per layer, activations in HBM
// (they are the backward's inputs anyway), LDS-tiled fp32 FMA GEMMs.  It is correct-first -- these networks are small and no
// throughput figure is quoted on them -- but it is the same C ABI, the same ray kernels around it, and no CPU fallback.
//
//   forward  : encode -> D x linear(+ReLU) -> sigma head (+noise, ReLU) -> feature -> per direction: view layer, output head
//   backward : output head / view layer per direction -> feature + sigma head -> trunk, each layer's weight gradient as an
//              atomically accumulated tile product (summation order not fixed: fp32 rounding-level run-to-run differences)
#include "vipnerf_generic.h"

namespace vn {

constexpr int GT = 64;      // tile edge (points / outputs / weights)
constexpr int GK = 16;      // contraction chunk

// ------------------------------------------------------------------------------------------------ encodings
__global__ void k_gen_encode(PointSrc s, GenTopo t, float *pex, float *ped, size_t ped_stride) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= s.P) return;
    PointCtx c;
    load_point(s, p, c);
    float *row = pex + (size_t)p * t.dp;
    for (int d = 0; d < 3; ++d) row[d] = c.x[d];
    for (int l = 0; l < t.lp; ++l)
        for (int d = 0; d < 3; ++d) {
            float sn, cs;
            sincosf(c.x[d] * __uint_as_float((unsigned)(127 + l) << 23), &sn, &cs);
            row[3 + 6 * l + d] = sn;
            row[3 + 6 * l + 3 + d] = cs;
        }
    for (int a = 0; a <= s.V; ++a) {
        float dir[3];
        if (a == 0) { dir[0] = c.dir[0]; dir[1] = c.dir[1]; dir[2] = c.dir[2]; }
        else secondary_dir(s, c, a - 1, dir);
        float *rd = ped + a * ped_stride + (size_t)p * t.dv;
        for (int d = 0; d < 3; ++d) rd[d] = dir[d];
        for (int l = 0; l < t.lv; ++l)
            for (int d = 0; d < 3; ++d) {
                float sn, cs;
                sincosf(dir[d] * __uint_as_float((unsigned)(127 + l) << 23), &sn, &cs);
                rd[3 + 6 * l + d] = sn;
                rd[3 + 6 * l + 3 + d] = cs;
            }
    }
}

// ------------------------------------------------------------------------------------------------ tiled products
// C[p][o] = epilogue( sum_k A[p][k] * B(k, o) ),  p < P, o < M, k < K0 + K1
//   A[p][k] = k < K0 ? A0[p * lda0 + k] : A1[p * lda1 + k - K0]
//   B(k, o) = TRANSW ? W[k * ldw + woff + o] : W[o * ldw + woff + k]
// epilogue: v = acc (+ bias[o]) (+ rank1s[p] * rank1v[o]); if mask: v = mask[p * ldm + o] > 0 ? v : 0; act 1 = ReLU, 2 = sigmoid;
//           accumulate ? C += v : C = v
struct GenGemm {
    int64_t P; int M;
    const float *A0; int lda0, K0; const float *A1; int lda1, K1;
    const float *W; int ldw, woff;
    const float *bias; const float *rank1s; const float *rank1v; const float *mask; int ldm;
    int act, accumulate;
    float *C; int ldc;
};
template <bool TRANSW>
__global__ __launch_bounds__(256) void k_gen_gemm(GenGemm g) {
    __shared__ float As[GK][GT + 1], Ws[GK][GT + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int64_t p0 = (int64_t)blockIdx.x * GT;
    const int o0 = blockIdx.y * GT, K = g.K0 + g.K1;
    float acc[4][4] = {};
    for (int kk = 0; kk < K; kk += GK) {
        for (int l = 0; l < 4; ++l) {
            const int idx = tid + 256 * l, r = idx >> 4, k = idx & 15, kg = kk + k;
            const int64_t p = p0 + r;
            float a = 0.f;
            if (p < g.P && kg < K) a = kg < g.K0 ? g.A0[(size_t)p * g.lda0 + kg] : g.A1[(size_t)p * g.lda1 + kg - g.K0];
            As[k][r] = a;
            float w = 0.f;
            if (TRANSW) {
                const int kq = idx >> 6, oq = idx & 63;          // consecutive threads along o (contiguous in W^T access)
                if (kk + kq < K && o0 + oq < g.M) w = g.W[(size_t)(kk + kq) * g.ldw + g.woff + o0 + oq];
                Ws[kq][oq] = w;
            } else {
                if (o0 + r < g.M && kg < K) w = g.W[(size_t)(o0 + r) * g.ldw + g.woff + kg];
                Ws[k][r] = w;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < GK; ++k) {
            float a[4], w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[k][ty + 16 * i]; w[i] = Ws[k][tx + 16 * i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], w[j], acc[i][j]);
        }
        __syncthreads();
    }
    for (int i = 0; i < 4; ++i) {
        const int64_t p = p0 + ty + 16 * i;
        if (p >= g.P) continue;
        for (int j = 0; j < 4; ++j) {
            const int o = o0 + tx + 16 * j;
            if (o >= g.M) continue;
            float v = acc[i][j];
            if (g.bias) v += g.bias[o];
            if (g.rank1s) v = fmaf(g.rank1s[p], g.rank1v[o], v);
            if (g.mask && !(g.mask[(size_t)p * g.ldm + o] > 0.f)) v = 0.f;
            if (g.act == 1) v = fmaxf(v, 0.f);
            else if (g.act == 2) v = 1.f / (1.f + expf(-v));
            float *c = g.C + (size_t)p * g.ldc + o;
            *c = g.accumulate ? *c + v : v;
        }
    }
}

// dW[o][woff + k] += sum_{p in chunk} dY[p][o] * X[p][k]  (X two-source like A above);  db[o] += sum_p dY[p][o] (k-tile 0 only)
struct GenWgrad {
    int64_t P; int M, chunk;
    const float *dY; int ldy;
    const float *X0; int ldx0, K0; const float *X1; int ldx1, K1;
    float *dW; int ldw, woff; float *db;
};
__global__ __launch_bounds__(256) void k_gen_wgrad(GenWgrad g) {
    __shared__ float Ys[GK][GT + 1], Xs[GK][GT + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int o0 = blockIdx.x * GT, k0 = blockIdx.y * GT, K = g.K0 + g.K1;
    const int64_t pb = (int64_t)blockIdx.z * g.chunk, pe = pb + g.chunk < g.P ? pb + g.chunk : g.P;
    float acc[4][4] = {}, bs[4] = {};
    for (int64_t pp = pb; pp < pe; pp += GK) {
        for (int l = 0; l < 4; ++l) {
            const int idx = tid + 256 * l, r = idx >> 6, c = idx & 63;       // point r of the chunk, column c: contiguous rows
            const int64_t p = pp + r;
            float y = 0.f, x = 0.f;
            if (p < pe) {
                if (o0 + c < g.M) y = g.dY[(size_t)p * g.ldy + o0 + c];
                const int kg = k0 + c;
                if (kg < K) x = kg < g.K0 ? g.X0[(size_t)p * g.ldx0 + kg] : g.X1[(size_t)p * g.ldx1 + kg - g.K0];
            }
            Ys[r][c] = y; Xs[r][c] = x;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < GK; ++r) {
            float y[4], x[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { y[i] = Ys[r][ty + 16 * i]; x[i] = Xs[r][tx + 16 * i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bs[i] += y[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(y[i], x[j], acc[i][j]);
            }
        }
        __syncthreads();
    }
    for (int i = 0; i < 4; ++i) {
        const int o = o0 + ty + 16 * i;
        if (o >= g.M) continue;
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + tx + 16 * j;
            if (k < K) atomicAdd(g.dW + (size_t)o * g.ldw + g.woff + k, acc[i][j]);
        }
        if (g.db && blockIdx.y == 0 && tx == 0) atomicAdd(g.db + o, bs[i]);
    }
}

// ------------------------------------------------------------------------------------------------ small per-point kernels
// trunk head (pts_output_linear, VipNeRF01.py:544-560): sigma = ReLU(w_0 . h + b_0 + noise * std); with NT = 4 (view_dependent_rgb = False)
// also rgb = sigmoid(w_{1..3} . h + b_{1..3})
template <int NT>
__global__ void k_gen_trunk_head(int64_t P, int W, const float *h, const float *w, const float *b, NoiseSrc ns, PointSrc s, float *sigma, float *rgb) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    float v[NT];
    for (int c = 0; c < NT; ++c) v[c] = 0.f;
    for (int k = 0; k < W; ++k) {
        const float x = h[(size_t)p * W + k];
        for (int c = 0; c < NT; ++c) v[c] = fmaf(x, w[c * W + k], v[c]);
    }
    float nz = 0.f;
    if (ns.noise) nz = ns.noise[p];
    else if (ns.device_rng) nz = rng_normal(ns.seed, ns.offset, ns.stream, noise_index(ns, s, p));
    sigma[p] = fmaxf(__fadd_rn(v[0] + b[0], __fmul_rn(nz, ns.std)), 0.f);
    for (int c = 1; c < NT; ++c) rgb[3 * p + c - 1] = 1.f / (1.f + expf(-(v[c] + b[c])));
}
// scatter the sigmoid outputs q[P][4] of direction a (columns: rgb when the view branch predicts it, then visibility when it does):
// a = 0 -> rgb, vis; a >= 1 -> vis2[:, a-1]
__global__ void k_gen_scatter(int64_t P, int V, int a, int rgb_cols, int vis_col, const float *q, float *rgb, float *vis, float *vis2) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    if (a == 0) {
        for (int c = 0; c < rgb_cols; ++c) rgb[3 * p + c] = q[4 * p + c];
        if (vis_col >= 0) vis[p] = q[4 * p + vis_col];
    } else if (vis_col >= 0) vis2[p * V + a - 1] = q[4 * p + vis_col];
}
// d(pre-sigmoid outputs) of direction a's view head
__global__ void k_gen_seeds(int64_t P, int V, int a, int rgb_cols, int vis_col, const float *q, const float *drgb, const float *dvis,
                            const float *dvis2, float *dq) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    float d[4] = {0.f, 0.f, 0.f, 0.f};
    if (a == 0) {
        for (int c = 0; c < rgb_cols; ++c) d[c] = drgb[3 * p + c];
        if (vis_col >= 0) d[vis_col] = dvis[p];
    } else if (vis_col >= 0) d[vis_col] = dvis2[p * V + a - 1];
    for (int c = 0; c < 4; ++c) { const float y = q[4 * p + c]; dq[4 * p + c] = d[c] * ((1.f - y) * y); }
}
// d(trunk head pre-activations): column 0 = d(sigma_raw) through the ReLU (also written to dsraw), columns 1..3 = d(rgb) through the sigmoid
// when the trunk predicts rgb
__global__ void k_gen_trunk_seeds(int64_t P, int nt, const float *sigma, const float *rgb, const float *dsig, const float *drgb, float *dsraw,
                                  float *dqt) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float ds = sigma[p] > 0.f ? dsig[p] : 0.f;
    dsraw[p] = ds;
    dqt[4 * p] = ds;
    for (int c = 1; c < 4; ++c) {
        float v = 0.f;
        if (c < nt) { const float y = rgb[3 * p + c - 1]; v = drgb[3 * p + c - 1] * ((1.f - y) * y); }
        dqt[4 * p + c] = v;
    }
}

// ------------------------------------------------------------------------------------------------ launch helpers
static int gemm(const GenGemm &g, bool transw, hipStream_t st) {
    if (g.P <= 0 || g.M <= 0) return VIPNERF_OK;
    const dim3 grid((unsigned)((g.P + GT - 1) / GT), (unsigned)((g.M + GT - 1) / GT));
    if (transw) hipLaunchKernelGGL(k_gen_gemm<true>, grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL(k_gen_gemm<false>, grid, dim3(256), 0, st, g);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}
static GenGemm gg(int64_t P, int M, const float *A0, int lda0, int K0, const float *W, int ldw, float *C, int ldc) {
    GenGemm g;
    memset(&g, 0, sizeof(g));
    g.P = P; g.M = M; g.A0 = A0; g.lda0 = lda0; g.K0 = K0; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc;
    return g;
}
static int wgrad(int64_t P, int M, const float *dY, int ldy, const float *X0, int ldx0, int K0, const float *X1, int ldx1, int K1,
                 float *dW, int ldw, int woff, float *db, hipStream_t st) {
    if (P <= 0) return VIPNERF_OK;
    GenWgrad g;
    g.P = P; g.M = M; g.dY = dY; g.ldy = ldy; g.X0 = X0; g.ldx0 = ldx0; g.K0 = K0; g.X1 = X1; g.ldx1 = ldx1; g.K1 = K1;
    g.dW = dW; g.ldw = ldw; g.woff = woff; g.db = db;
    // point chunks: enough workgroups to fill the chip whatever the layer's size (a 64 x 64 layer is ONE output tile -- with 4096-point
    // chunks the toy network's 65,536 points ran on 16 of 256 CUs, 0.54 ms per GEMM), at least 128 points each so that the atomics
    // that merge the chunks stay a small part of the work
    const int tiles = ((M + GT - 1) / GT) * ((K0 + K1 + GT - 1) / GT);
    int chunks = (1024 + tiles - 1) / tiles;
    const int64_t most = (P + 127) / 128;
    if (chunks > most) chunks = (int)most;
    if (chunks < 1) chunks = 1;
    g.chunk = (int)(((P + chunks - 1) / chunks + GK - 1) / GK * GK);
    chunks = (int)((P + g.chunk - 1) / g.chunk);
    hipLaunchKernelGGL(k_gen_wgrad, dim3((unsigned)((M + GT - 1) / GT), (unsigned)((K0 + K1 + GT - 1) / GT), (unsigned)chunks), dim3(256), 0, st, g);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}
#define GCHK(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)

int launch_gen_fwd(const GenTopo &t, const PointSrc &s, const NoiseSrc &ns, const float *prm, float *sigma, float *rgb, float *vis,
                   float *vis2, float *acts, hipStream_t st) {
    const int64_t P = s.P;
    if (P <= 0) return VIPNERF_OK;
    const GenParams gp = gen_params(t);
    const GenActs al = gen_acts((size_t)P, s.V, t);
    const unsigned nb = (unsigned)((P + 255) / 256);
    hipLaunchKernelGGL(k_gen_encode, dim3(nb), dim3(256), 0, st, s, t, acts + al.pex, acts + al.ped[0], (size_t)P * t.dv);
    VN_HIP(hipGetLastError());
    for (int i = 0; i < t.D; ++i) {
        const bool sk = i == t.skip;
        GenGemm g = i == 0 ? gg(P, t.W, acts + al.pex, t.dp, t.dp, prm + gp.w[i], t.dp, acts + al.h[i], t.W)
                           : gg(P, t.W, sk ? acts + al.pex : acts + al.h[i - 1], sk ? t.dp : t.W, sk ? t.dp : t.W, prm + gp.w[i],
                                sk ? t.dp + t.W : t.W, acts + al.h[i], t.W);
        if (sk) { g.A1 = acts + al.h[i - 1]; g.lda1 = t.W; g.K1 = t.W; }
        g.bias = prm + gp.b[i]; g.act = 1;
        GCHK(gemm(g, false, st));
    }
    const float *hl = acts + al.h[t.D - 1];
    const int no = gen_view_outs(t), rgb_cols = gen_rgb_trunk(t) ? 0 : 3, vis_col = gen_pred_vis(t) ? rgb_cols : -1;
    if (gen_rgb_trunk(t)) hipLaunchKernelGGL(k_gen_trunk_head<4>, dim3(nb), dim3(256), 0, st, P, t.W, hl, prm + gp.ws, prm + gp.bs, ns, s, sigma, rgb);
    else hipLaunchKernelGGL(k_gen_trunk_head<1>, dim3(nb), dim3(256), 0, st, P, t.W, hl, prm + gp.ws, prm + gp.bs, ns, s, sigma, rgb);
    VN_HIP(hipGetLastError());
    if (!gen_pred_vis(t)) VN_HIP(hipMemsetAsync(vis, 0, (size_t)P * sizeof(float), st));      // no visibility prediction: the output reads 0
    if (no == 0) return VIPNERF_OK;                                                           // no view-dependent output: no feature / view layers
    {
        GenGemm g = gg(P, t.W, hl, t.W, t.W, prm + gp.wf, t.W, acts + al.feat, t.W);
        g.bias = prm + gp.bf;
        GCHK(gemm(g, false, st));
    }
    for (int a = 0; a <= s.V; ++a) {
        GenGemm g = gg(P, t.W / 2, acts + al.feat, t.W, t.W, prm + gp.wv, t.W + t.dv, acts + al.g[a], t.W / 2);
        g.A1 = acts + al.ped[a]; g.lda1 = t.dv; g.K1 = t.dv; g.bias = prm + gp.bv; g.act = 1;
        GCHK(gemm(g, false, st));
        if (no < 4) VN_HIP(hipMemsetAsync(acts + al.q[a], 0, (size_t)P * 4 * sizeof(float), st));   // unused columns stay 0 (their seeds too)
        GenGemm o = gg(P, no, acts + al.g[a], t.W / 2, t.W / 2, prm + gp.wo, t.W / 2, acts + al.q[a], 4);
        o.bias = prm + gp.bo; o.act = 2;
        GCHK(gemm(o, false, st));
        hipLaunchKernelGGL(k_gen_scatter, dim3(nb), dim3(256), 0, st, P, s.V, a, rgb_cols, vis_col, acts + al.q[a], rgb, vis, vis2);
        VN_HIP(hipGetLastError());
    }
    return VIPNERF_OK;
}

int launch_gen_bwd(const GenTopo &t, const PointSrc &s, const float *prm, const float *sigma, const float *rgb, const float *acts, float *bwd,
                   const GenBwd &bl, const vipnerf_mlp_grads *G, hipStream_t st) {
    const int64_t P = s.P;
    const GenParams gp = gen_params(t);
    // gradients are accumulated atomically: start from zero
    for (int i = 0; i < VIPNERF_N_PARAMS; ++i) {
        const size_t n = gen_param_numel(t, i);
        if (n && G->g[i]) VN_HIP(hipMemsetAsync(G->g[i], 0, n * sizeof(float), st));
    }
    if (P <= 0) return VIPNERF_OK;
    const GenActs al = gen_acts((size_t)P, s.V, t);
    const unsigned nb = (unsigned)((P + 255) / 256);
    const int W = t.W, H = t.W / 2;
    const int no = gen_view_outs(t), nt = gen_trunk_outs(t), rgb_cols = gen_rgb_trunk(t) ? 0 : 3, vis_col = gen_pred_vis(t) ? rgb_cols : -1;
    float *dfeat = bwd + bl.dfeat, *dg = bwd + bl.dg, *dsraw = bwd + bl.dsraw, *dqt = bwd + bl.dqt;
    // view branch, per direction
    for (int a = 0; no > 0 && a <= s.V; ++a) {
        float *dq = bwd + bl.dq[a];
        hipLaunchKernelGGL(k_gen_seeds, dim3(nb), dim3(256), 0, st, P, s.V, a, rgb_cols, vis_col, acts + al.q[a], bwd + bl.drgb, bwd + bl.dvis,
                           bwd + bl.dvis2, dq);
        VN_HIP(hipGetLastError());
        GCHK(wgrad(P, no, dq, 4, acts + al.g[a], H, H, nullptr, 0, 0, G->g[P_OW], H, 0, G->g[P_OB], st));
        GenGemm g = gg(P, H, dq, 4, no, prm + gp.wo, H, dg, H);             // dg = (dq W_o) masked by g > 0
        g.mask = acts + al.g[a]; g.ldm = H;
        GCHK(gemm(g, true, st));
        GCHK(wgrad(P, H, dg, H, acts + al.feat, W, W, acts + al.ped[a], t.dv, t.dv, G->g[P_VW], W + t.dv, 0, G->g[P_VB], st));
        GenGemm f = gg(P, W, dg, H, H, prm + gp.wv, W + t.dv, dfeat, W);    // d feature (+)= dg W_v[:, :W]
        f.accumulate = a > 0;
        GCHK(gemm(f, true, st));
    }
    // trunk head seeds, feature layer + trunk head -> d h_D (masked by its ReLU)
    hipLaunchKernelGGL(k_gen_trunk_seeds, dim3(nb), dim3(256), 0, st, P, nt, sigma, rgb, bwd + bl.dsig, bwd + bl.drgb, dsraw, dqt);
    VN_HIP(hipGetLastError());
    const float *hl = acts + al.h[t.D - 1];
    if (no > 0) GCHK(wgrad(P, W, dfeat, W, hl, W, W, nullptr, 0, 0, G->g[P_FW], W, 0, G->g[P_FB], st));
    GCHK(wgrad(P, nt, dqt, 4, hl, W, W, nullptr, 0, 0, G->g[P_SW], W, 0, G->g[P_SB], st));
    float *dcur = bwd + bl.dh[0], *dnext = bwd + bl.dh[1];
    if (no > 0 && nt == 1) {                                                // the default heads: one GEMM with sigma's rank-1 term in its epilogue
        GenGemm g = gg(P, W, dfeat, W, W, prm + gp.wf, W, dcur, W);
        g.rank1s = dsraw; g.rank1v = prm + gp.ws; g.mask = hl; g.ldm = W;
        GCHK(gemm(g, true, st));
    } else {                                                                // masked sums add: mask(a) + mask(b) = mask(a + b)
        if (no > 0) {
            GenGemm g = gg(P, W, dfeat, W, W, prm + gp.wf, W, dcur, W);
            g.mask = hl; g.ldm = W;
            GCHK(gemm(g, true, st));
        }
        GenGemm g = gg(P, W, dqt, 4, nt, prm + gp.ws, W, dcur, W);           // dqt W_s: K = the trunk head's rows
        g.mask = hl; g.ldm = W; g.accumulate = no > 0;
        GCHK(gemm(g, true, st));
    }
    // trunk, last layer first: dcur = dLoss/d(pre-activation of layer i)
    for (int i = t.D - 1; i >= 0; --i) {
        const bool sk = i == t.skip;
        const float *x0 = i == 0 || sk ? acts + al.pex : acts + al.h[i - 1];
        const int k0 = i == 0 || sk ? t.dp : W;
        GCHK(wgrad(P, W, dcur, W, x0, k0, k0, sk ? acts + al.h[i - 1] : nullptr, W, sk ? W : 0, G->g[2 * i], sk ? t.dp + W : k0, 0,
                   G->g[2 * i + 1], st));
        if (i == 0) break;
        GenGemm g = gg(P, W, dcur, W, W, prm + gp.w[i], sk ? t.dp + W : W, dnext, W);
        g.woff = sk ? t.dp : 0; g.mask = acts + al.h[i - 1]; g.ldm = W;
        GCHK(gemm(g, true, st));
        float *tmp = dcur; dcur = dnext; dnext = tmp;
    }
    return VIPNERF_OK;
}

// flat parameter buffer of the generic path: the tensors back to back in parameter-slot order
int launch_gen_pack(const GenTopo &t, const vipnerf_mlp_params *p, float *flat, hipStream_t st) {
    size_t off = 0;
    for (int i = 0; i < VIPNERF_N_PARAMS; ++i) {
        const size_t n = gen_param_numel(t, i);
        if (!n) continue;
        if (!p->p[i]) { set_error("pack_weights: parameter %d is NULL (netdepth %d, netwidth %d)", i, t.D, t.W); return VIPNERF_E_ARG; }
        VN_HIP(hipMemcpyAsync(flat + off, p->p[i], n * sizeof(float), hipMemcpyDeviceToDevice, st));
        off += n;
    }
    return VIPNERF_OK;
}

}  // namespace vn
