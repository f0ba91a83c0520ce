// Data-gradient pass on split-precision bf16 MFMA: vipnerf_mlp_bwd.hip with the fragments of vipnerf_bf16.h
// (A = W^T from the packed bf16 image, B = the NS-part split of the current dY).  Everything outside the GEMMs --
// activation derivatives, ReLU masks, the sigma head's rank-1 term, the stored dY (fp32) -- is unchanged.
#include "vipnerf_bf16.h"
#include "vipnerf_mlp.h"

namespace vn {

template <int NS>
__global__ __launch_bounds__(MLP_WG) void k_mlp_bwd_bf16(MlpBwdArgs a) {
    typedef BfPlan<NS> PL;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *res = lds;
    float *stage_buf = lds + PL::R_TOTAL_PAD;
    const float *rf = res + PL::R_F32 - R_BIAS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int64_t p_raw = (int64_t)blockIdx.x * MLP_PTS_PER_WG + wave * 32 + j;
    const bool valid = p_raw < a.src.P;
    const int64_t p = valid ? p_raw : a.src.P - 1;
    const int V = a.src.V;

    WStreamT<PL::CH, PL::NBUF> ws;
    ws.start(a.packed + PL::PK_BWD, PL::B_STAGES, stage_buf, lane, wave);
    {
        const float4 *g4 = (const float4 *)(a.packed + PL::PK_RES);
        float4 *l4 = (float4 *)res;
        for (int i = tid; i < PL::R_TOTAL_PAD / 4; i += MLP_WG) l4[i] = g4[i];
    }

    const float *gb = a.bwd;
    float dq0[4];
    {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float y = a.rgb[3 * p + c];
            dq0[c] = gb[a.bl.drgb + 3 * p + c] * ((1.f - y) * y);
        }
        const float y = a.vis[p];
        dq0[3] = gb[a.bl.dvis + p] * ((1.f - y) * y);
    }
    const float dsig_raw = a.sigma[p] > 0.f ? gb[a.bl.dsig + p] : 0.f;
    __syncthreads();

    // ---------------------------------------------------------------- view branch, per direction
    floatx16 vsum[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) vsum[t] = (floatx16)(0.f);
#pragma unroll 1
    for (int dsel = 0; dsel <= V; ++dsel) {
        float dq[4];
        if (dsel == 0) { dq[0] = dq0[0]; dq[1] = dq0[1]; dq[2] = dq0[2]; dq[3] = dq0[3]; }
        else {
            const float y = a.vis2[p * V + (dsel - 1)];
            dq[0] = dq[1] = dq[2] = 0.f;
            dq[3] = gb[a.bl.dvis2 + p * V + (dsel - 1)] * ((1.f - y) * y);
        }
        if (valid && h == 0) {
            float *row = a.bwd + a.bl.dq[dsel] + (size_t)p * 8;
            *(float4 *)row = make_float4(dq[0], dq[1], dq[2], dq[3]);
            *(float4 *)(row + 4) = make_float4(dsel == 0 ? dsig_raw : 0.f, 0.f, 0.f, 0.f);
        }
        floatx16 g[4];
        load_frag<4>(a.acts + a.al.g[dsel], p, WV, h, g);
        const float *wo = rf + R_WOUT + h * 256;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 w4 = *(const float4 *)(wo + c * 64 + 16 * t + 4 * q);
                    dg.x = fmaf(w4.x, dq[c], dg.x); dg.y = fmaf(w4.y, dq[c], dg.y);
                    dg.z = fmaf(w4.z, dq[c], dg.z); dg.w = fmaf(w4.w, dq[c], dg.w);
                }
                g[t][4 * q] = g[t][4 * q] > 0.f ? dg.x : 0.f;
                g[t][4 * q + 1] = g[t][4 * q + 1] > 0.f ? dg.y : 0.f;
                g[t][4 * q + 2] = g[t][4 * q + 2] > 0.f ? dg.z : 0.f;
                g[t][4 * q + 3] = g[t][4 * q + 3] > 0.f ? dg.w : 0.f;
            }
        store_frag<4>(a.bwd + a.bl.dyv[dsel], p, WV, h, g, valid);
#pragma unroll
        for (int t = 0; t < 4; ++t) vsum[t] += g[t];
    }
    store_frag<4>(a.bwd + a.bl.dyvsum, p, WV, h, vsum, valid);

    // ---------------------------------------------------------------- d(feature) = W_vf^T sum_a dYv_a   (K = 128: 8 k-steps)
    bf16x8 bin[16][NS];
    floatx16 acc[8];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float xs[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) xs[e] = vsum[t][8 * u + e];
            split8<NS>(xs, bin[2 * t + u]);
        }
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = (floatx16)(0.f);
#pragma unroll
    for (int jj = 0; jj < PL::ST_VIEW_B; ++jj) {
        const float *st = ws.wait();
        gemm_stage_bf<8, PL::KSB, NS>(st, lane, acc, bin, PL::KSB * jj, ws);
    }
    // dY of the feature layer: store (fp32, for wgrad) and split
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        store_tile(a.bwd + a.bl.dyf, p, W, h, t, acc[t], valid);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float xs[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) xs[e] = acc[t][8 * u + e];
            split8<NS>(xs, bin[2 * t + u]);
        }
    }

    // ---------------------------------------------------------------- feature layer, then layers 7..1
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
        const int layer = 7 - it;
        const uint4 mk = *(const uint4 *)(a.acts + a.al.hm[layer] + ((size_t)p * 2 + h) * 4);
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = (floatx16)(0.f);
#pragma unroll
        for (int jj = 0; jj < PL::ST_256; ++jj) {
            const float *st = ws.wait();
            gemm_stage_bf<8, PL::KSB, NS>(st, lane, acc, bin, PL::KSB * jj, ws);
        }
        float *dst = a.bwd + a.bl.dy[layer];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            floatx16 x = acc[t];
            if (it == 0) {                               // h_8 also feeds the sigma head
                const float *wsg = rf + R_WSIG + h * 128 + 16 * t;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w4 = *(const float4 *)(wsg + 4 * q);
                    x[4 * q] = fmaf(w4.x, dsig_raw, x[4 * q]);
                    x[4 * q + 1] = fmaf(w4.y, dsig_raw, x[4 * q + 1]);
                    x[4 * q + 2] = fmaf(w4.z, dsig_raw, x[4 * q + 2]);
                    x[4 * q + 3] = fmaf(w4.w, dsig_raw, x[4 * q + 3]);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = mask_bit(mk, t, r) ? x[r] : 0.f;
            store_tile(dst, p, W, h, t, x, valid);
            if (it < 7) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float xs[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) xs[e] = x[8 * u + e];
                    split8<NS>(xs, bin[2 * t + u]);
                }
            }
        }
    }
}

template <int NS>
static int launch_one_bwd(const MlpBwdArgs &a, unsigned grid, hipStream_t st) {
    const size_t lds = (size_t)BfPlan<NS>::LDS_F * sizeof(float);
    VN_HIP(hipFuncSetAttribute((const void *)k_mlp_bwd_bf16<NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_mlp_bwd_bf16<NS>), dim3(grid), dim3(MLP_WG), lds, st, a);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

int launch_mlp_bwd_bf16(const MlpBwdArgs &a, int precision, hipStream_t st) {
    if (a.src.P <= 0) return VIPNERF_OK;
    const unsigned grid = (unsigned)((a.src.P + MLP_PTS_PER_WG - 1) / MLP_PTS_PER_WG);
    if (precision == 1) return launch_one_bwd<2>(a, grid, st);
    if (precision == 2) return launch_one_bwd<3>(a, grid, st);
    set_error("mlp_bwd_bf16: precision %d", precision);
    return VIPNERF_E_ARG;
}

}  // namespace vn
