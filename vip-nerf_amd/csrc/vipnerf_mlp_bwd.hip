// Data-gradient pass of the MLP (autograd of reference src/models/VipNeRF01.py:509-596 w.r.t. activations).
//
// Same register-chained, transposed formulation as the forward kernel, with A = W^T streamed through LDS:
// dH_in^T[K x 32] = W^T[K x 256] * dY^T[256 x 32].  The C/D fragment of one dgrad GEMM, masked by the stored
// ReLU output, is the B operand of the next.  Every pre-activation gradient dY_i is written to the backward
// scratch ([P][256] row-major) for the weight-gradient GEMMs (vipnerf_wgrad.hip), which contract over points
// and therefore need the point axis along MFMA's k, i.e. a different kernel.
//
// The view layer's feature columns see the SUM over directions of the per-direction gradients
// (W_vf^T sum_a dYv_a), so one GEMM serves all 1+V directions.
#include "vipnerf_mlp.h"

namespace vn {

__global__ __launch_bounds__(MLP_WG) void k_mlp_bwd(MlpBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *res = lds;
    float *stage_buf = lds + R_TOTAL_PAD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int64_t p_raw = (int64_t)blockIdx.x * MLP_PTS_PER_WG + wave * 32 + j;
    const bool valid = p_raw < a.src.P;
    const int64_t p = valid ? p_raw : a.src.P - 1;
    const int V = a.src.V;

    WStream ws;
    ws.start(a.packed + PK_BWD, B_STAGES, stage_buf, lane, wave);
    {
        const float4 *g4 = (const float4 *)(a.packed + PK_RES);
        float4 *l4 = (float4 *)res;
        for (int i = tid; i < R_TOTAL_PAD / 4; i += MLP_WG) l4[i] = g4[i];
    }

    // upstream gradients w.r.t. the raw network outputs, through the output non-linearities
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
    __syncthreads();                                   // resident block visible

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
        const float *wo = res + R_WOUT + h * 256;
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

    // ---------------------------------------------------------------- d(feature) = W_vf^T sum_a dYv_a
    floatx16 in[8], acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = (floatx16)(0.f);
#pragma unroll
    for (int jj = 0; jj < ST_VIEW_B; ++jj) {
        const float *st = ws.next();
        gemm_stage_f32<8, KGS8>(st, lane, acc, KGS8 * jj, [&](int r) { return vsum[r >> 4][r & 15]; });
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) in[t] = acc[t];

    // ---------------------------------------------------------------- feature layer, then layers 7..1
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = (floatx16)(0.f);
        // ReLU mask of the layer this GEMM lands on: 16 B per lane, fetched under the GEMM
        const int layer = 7 - it;
        const uint4 mk = *(const uint4 *)(a.acts + a.al.hm[layer] + ((size_t)p * 2 + h) * 4);
        float *dy_dst = a.bwd + (it == 0 ? a.bl.dyf : a.bl.dy[8 - it]);
#pragma unroll
        for (int jj = 0; jj < ST_256; ++jj) {
            const float *st = ws.next();
            // this GEMM's B operand (a dY) is also what wgrad needs: stored one tile per stage, ahead of the MFMAs
#pragma unroll
            for (int tt = 0; tt < 8 / ST_256; ++tt) store_tile(dy_dst, p, W, h, jj * (8 / ST_256) + tt, in[jj * (8 / ST_256) + tt], valid);
            __builtin_amdgcn_sched_barrier(0);
            gemm_stage_f32<8, KGS8>(st, lane, acc, KGS8 * jj, [&](int r) { return in[r >> 4][r & 15]; });
        }
        if (it == 0) {                                   // h_8 also feeds the sigma head
            const float *wsg = res + R_WSIG + h * 128;
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w4 = *(const float4 *)(wsg + 16 * t + 4 * q);
                    acc[t][4 * q] = fmaf(w4.x, dsig_raw, acc[t][4 * q]);
                    acc[t][4 * q + 1] = fmaf(w4.y, dsig_raw, acc[t][4 * q + 1]);
                    acc[t][4 * q + 2] = fmaf(w4.z, dsig_raw, acc[t][4 * q + 2]);
                    acc[t][4 * q + 3] = fmaf(w4.w, dsig_raw, acc[t][4 * q + 3]);
                }
        }
        // acc = dLoss/d(output of layer 7-it); through its ReLU -> dY of that layer
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) in[t][r] = mask_bit(mk, t, r) ? acc[t][r] : 0.f;
    }
    store_frag<8>(a.bwd + a.bl.dy[0], p, W, h, in, valid);     // dY of layer 0 has no further GEMM to hide under
}

int launch_mlp_bwd(const MlpBwdArgs &a, hipStream_t st) {
    if (a.src.P <= 0) return VIPNERF_OK;
    const unsigned grid = (unsigned)((a.src.P + MLP_PTS_PER_WG - 1) / MLP_PTS_PER_WG);
    VN_HIP(hipFuncSetAttribute((const void *)k_mlp_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MLP_LDS_BYTES));
    hipLaunchKernelGGL(k_mlp_bwd, dim3(grid), dim3(MLP_WG), MLP_LDS_BYTES, st, a);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

}  // namespace vn
