// MLP.forward (reference src/models/VipNeRF01.py:509-596) as ONE kernel: positional encoding, the 8x256 trunk
// with the skip concat, sigma head (+noise, ReLU), feature layer, and the 128-wide view branch evaluated for
// the primary direction (rgb + visibility) and each secondary direction (visibility2).
//
// Mapping: workgroup = 4 waves (one per SIMD, whole 512-register budget each); wave = 32 points; activations
// live in the C/D fragments of v_mfma_f32_32x32x2_f32 and are re-used in place as the next layer's B operand
// (see vipnerf_common.h); weights stream through LDS.  fp32 in, fp32 accumulate: bit-compatible with an fmaf
// chain, so parity with the reference's fp32 GEMMs is at rounding level.
//
// The view layer is split algebraically: W_v [f ; gamma(a)] = W_vf f + W_vd gamma(a).  W_vf f (the expensive
// 256-wide part) is computed once per point and shared by the 1+V directions, which the reference recomputes.
#include "vipnerf_mlp.h"

namespace vn {

template <bool SAVE>
__global__ __launch_bounds__(MLP_WG) void k_mlp_fwd(MlpFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *res = lds;                          // LDS-resident heads / biases / direction weights
    float *stage_buf = lds + R_TOTAL_PAD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int64_t p_raw = (int64_t)blockIdx.x * MLP_PTS_PER_WG + wave * 32 + j;
    const bool valid = p_raw < a.src.P;
    const int64_t p = valid ? p_raw : a.src.P - 1;

    WStream ws;
    ws.start(a.packed + PK_FWD, F_STAGES, stage_buf, lane, wave);
    {   // resident block: plain loads (29 KiB once per workgroup)
        const float4 *g4 = (const float4 *)(a.packed + PK_RES);
        float4 *l4 = (float4 *)res;
        for (int i = tid; i < R_TOTAL_PAD / 4; i += MLP_WG) l4[i] = g4[i];
    }

    PointCtx pc;
    load_point(a.src, p, pc);
    float pe[32];
    encode_half<LP, 32>(pc.x, h, pe);
    if (SAVE && valid) {
        float *row = a.acts + a.al.pex + (size_t)p * DPE_PAD + h;
#pragma unroll
        for (int s = 0; s < 32; ++s) row[2 * s] = pe[s];
    }

    floatx16 in[8], acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) in[t] = (floatx16)(0.f);
    float sigma_raw = 0.f;

    // ---------------------------------------------------------------- trunk (layers 0..7) + feature layer (8)
    for (int layer = 0; layer < 9; ++layer) {
        const float *bias;
        if (layer == 0) __syncthreads();        // resident block visible (also the first stage, harmlessly)
        bias = res + (layer < 8 ? R_BIAS + layer * W : R_BFEAT) + h * 16;
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = *(const floatx16 *)(bias + t * 32);

        if (layer == 0 || layer == SKIP_LAYER) {
#pragma unroll
            for (int jj = 0; jj < ST_PE; ++jj) {
                const float *st = ws.next();
                gemm_stage_f32<8, KGS8>(st, lane, acc, KGS8 * jj, [&](int r) { return pe[r]; });
            }
        }
        if (layer != 0) {
#pragma unroll
            for (int jj = 0; jj < ST_256; ++jj) {
                const float *st = ws.next();
                if (SAVE) {
                    // the previous layer's output (this layer's B operand) goes to the activation store one tile per
                    // stage, issued BEFORE the stage's MFMAs: the stores then have a whole stage (>= 8192 cycles) to
                    // retire before the next barrier's vmcnt(0), instead of a 32 KB burst draining in front of it
#pragma unroll
                    for (int tt = 0; tt < 8 / ST_256; ++tt)
                        store_tile(a.acts + a.al.h[layer - 1], p, W, h, jj * (8 / ST_256) + tt, in[jj * (8 / ST_256) + tt], valid);
                    __builtin_amdgcn_sched_barrier(0);
                }
                gemm_stage_f32<8, KGS8>(st, lane, acc, KGS8 * jj, [&](int r) { return in[r >> 4][r & 15]; });
            }
        }
        if (layer < 8) {
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) in[t][r] = fmaxf(acc[t][r], 0.f);
            if (SAVE && valid) *(uint4 *)(a.acts + a.al.hm[layer] + ((size_t)p * 2 + h) * 4) = frag_mask(in);
            if (layer == 7) {   // sigma head on h_8: per-lane partial dot over its 128 features, then fold halves
                const float *wsg = res + R_WSIG + h * 128;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 w4 = *(const float4 *)(wsg + 16 * t + 4 * q);
                        s0 = fmaf(w4.x, in[t][4 * q], s0);
                        s1 = fmaf(w4.y, in[t][4 * q + 1], s1);
                        s2 = fmaf(w4.z, in[t][4 * q + 2], s2);
                        s3 = fmaf(w4.w, in[t][4 * q + 3], s3);
                    }
                float s = (s0 + s1) + (s2 + s3);
                s += __shfl_xor(s, 32, 64);
                sigma_raw = s + res[R_BHEAD];
            }
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) in[t] = acc[t];      // feature: no activation
        }
    }

    // ---------------------------------------------------------------- sigma: noise + ReLU
    {
        float nz = 0.f;
        if (a.ns.noise) nz = a.ns.noise[p];
        else if (a.ns.device_rng) nz = rng_normal(a.ns.seed, a.ns.offset, a.ns.stream, noise_index(a.ns, a.src, p));
        const float sg = fmaxf(__fadd_rn(sigma_raw, __fmul_rn(nz, a.ns.std)), 0.f);
        if (valid && h == 0) a.sigma[p] = sg;
    }

    // ---------------------------------------------------------------- view branch
    floatx16 vb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) vb[t] = *(const floatx16 *)(res + R_BVIEW + t * 32 + h * 16);
#pragma unroll
    for (int jj = 0; jj < ST_VIEW_F; ++jj) {
        const float *st = ws.next();
        if (SAVE) {
#pragma unroll
            for (int tt = 0; tt < 8 / ST_VIEW_F; ++tt)
                store_tile(a.acts + a.al.feat, p, W, h, jj * (8 / ST_VIEW_F) + tt, in[jj * (8 / ST_VIEW_F) + tt], valid);
            __builtin_amdgcn_sched_barrier(0);
        }
        gemm_stage_f32<4, KGS4>(st, lane, vb, KGS4 * jj, [&](int r) { return in[r >> 4][r & 15]; });
    }

    for (int dsel = 0; dsel <= a.src.V; ++dsel) {
        float dir[3];
        if (dsel == 0) { dir[0] = pc.dir[0]; dir[1] = pc.dir[1]; dir[2] = pc.dir[2]; }
        else secondary_dir(a.src, pc, dsel - 1, dir);
        float ped[16];
        encode_half<LV, 16>(dir, h, ped);
        floatx16 g[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) g[t] = vb[t];
        gemm_stage_f32<4, 4>(res + R_DIRW, lane, g, 0, [&](int r) { return ped[r]; });
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) g[t][r] = fmaxf(g[t][r], 0.f);
        if (SAVE) {
            store_frag<4>(a.acts + a.al.g[dsel], p, WV, h, g, valid);
            if (valid) {
                float *row = a.acts + a.al.ped[dsel] + (size_t)p * DVE_PAD + h;
#pragma unroll
                for (int s = 0; s < 16; ++s) row[2 * s] = ped[s];
            }
        }
        // output head: q[c] = b_o[c] + sum_r W_o[c][feat(r,h)] g_r
        float qv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float *wo = res + R_WOUT + h * 256 + c * 64;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w4 = *(const float4 *)(wo + 16 * t + 4 * q);
                    s0 = fmaf(w4.x, g[t][4 * q], s0);
                    s1 = fmaf(w4.y, g[t][4 * q + 1], s1);
                    s0 = fmaf(w4.z, g[t][4 * q + 2], s0);
                    s1 = fmaf(w4.w, g[t][4 * q + 3], s1);
                }
            float s = s0 + s1;
            s += __shfl_xor(s, 32, 64);
            qv[c] = sigmoidf_(s + res[R_BHEAD + 1 + c]);
        }
        if (valid && h == 0) {
            if (dsel == 0) {
                a.rgb[3 * p + 0] = qv[0]; a.rgb[3 * p + 1] = qv[1]; a.rgb[3 * p + 2] = qv[2];
                a.vis[p] = qv[3];
            } else {
                a.vis2[p * a.src.V + (dsel - 1)] = qv[3];
            }
        }
    }
}

int launch_mlp_fwd(const MlpFwdArgs &a, hipStream_t st) {
    if (a.src.P <= 0) return VIPNERF_OK;
    const unsigned grid = (unsigned)((a.src.P + MLP_PTS_PER_WG - 1) / MLP_PTS_PER_WG);
    if (a.acts) {
        VN_HIP(hipFuncSetAttribute((const void *)k_mlp_fwd<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MLP_LDS_BYTES));
        hipLaunchKernelGGL(k_mlp_fwd<true>, dim3(grid), dim3(MLP_WG), MLP_LDS_BYTES, st, a);
    } else {
        VN_HIP(hipFuncSetAttribute((const void *)k_mlp_fwd<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MLP_LDS_BYTES));
        hipLaunchKernelGGL(k_mlp_fwd<false>, dim3(grid), dim3(MLP_WG), MLP_LDS_BYTES, st, a);
    }
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

}  // namespace vn
