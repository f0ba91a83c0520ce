// MLP.forward on split-precision bf16 MFMA, narrow-wave layout (vipnerf_bf16n.h): 16 points per wave on
// v_mfma_f32_16x16x32_bf16, 8 waves (two per SIMD) and 128 points per workgroup.  Same algorithm, stage order and
// stored activations as vipnerf_mlp_fwd_bf16.hip; only the lane <-> (point, feature) map differs: lane (j, q) holds
// features 16T + 4q .. +3 of tile T for point j.
#include "vipnerf_bf16n.h"
#include "vipnerf_mlp.h"

namespace vn {

// gamma_L(v) for the narrow fragment: k-step s, element e of lane group q is feature 32 s + 8 q + e (natural order)
template <int L, int NKS>
__device__ __forceinline__ void encode_bn(const float v[3], int q, float (&out)[NKS][8]) {
    float val[32 * NKS];
#pragma unroll
    for (int f = 0; f < 32 * NKS; ++f) val[f] = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) val[d] = v[d];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float s, c;
            sincosf(v[d] * (float)(1 << l), &s, &c);
            val[3 + 6 * l + d] = s;
            val[3 + 6 * l + 3 + d] = c;
        }
#pragma unroll
    for (int s = 0; s < NKS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float a0 = val[32 * s + e], a1 = val[32 * s + 8 + e], a2 = val[32 * s + 16 + e], a3 = val[32 * s + 24 + e];
            out[s][e] = q == 0 ? a0 : (q == 1 ? a1 : (q == 2 ? a2 : a3));
        }
}

template <bool SAVE, int NS>
__global__ __launch_bounds__(BnPlan<NS>::WG) void k_mlp_fwd_bf16n(MlpFwdArgs a) {
    typedef BnPlan<NS> PL;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *res = lds;
    float *stage_buf = lds + PL::R_TOTAL_PAD;
    const float *rf = res + PL::R_F32;                   // fp32 block, natural feature order

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, j = lane & 15;
    const int64_t p_raw = (int64_t)blockIdx.x * MLP_PTS_PER_WG + wave * 16 + j;
    const bool valid = p_raw < a.src.P;
    const int64_t p = valid ? p_raw : a.src.P - 1;

    WStreamT<PL::CH, PL::NBUF, PL::WAVES> ws;
    ws.start(a.packed + PL::PK_FWD, PL::F_STAGES, stage_buf, lane, wave);
    {
        const float4 *g4 = (const float4 *)(a.packed + PL::PK_RES);
        float4 *l4 = (float4 *)res;
        for (int i = tid; i < PL::R_TOTAL_PAD / 4; i += PL::WG) l4[i] = g4[i];
    }

    PointCtx pc;
    load_point(a.src, p, pc);
    float pe[2][8];
    encode_bn<LP, 2>(pc.x, q, pe);
    if (SAVE && valid) {
        float *row = a.acts + a.al.pex + (size_t)p * DPE_PAD + 8 * q;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            *(float4 *)(row + 32 * s) = make_float4(pe[s][0], pe[s][1], pe[s][2], pe[s][3]);
            *(float4 *)(row + 32 * s + 4) = make_float4(pe[s][4], pe[s][5], pe[s][6], pe[s][7]);
        }
    }

    bf16x8 bin[8][NS];                       // the layer input as B fragments: k-step s <- C/D tiles 2s, 2s+1
    floatx4 acc[16];
    float sigma_raw = 0.f;

    // ---------------------------------------------------------------- trunk (layers 0..7) + feature layer (8)
#pragma unroll 1
    for (int layer = 0; layer < 9; ++layer) {
        if (layer == 0) __syncthreads();
        const float *bias = rf + (layer < 8 ? PL::N_BIAS + layer * W : PL::N_BFEAT) + 4 * q;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float4 b4 = *(const float4 *)(bias + 16 * t);
            acc[t][0] = b4.x; acc[t][1] = b4.y; acc[t][2] = b4.z; acc[t][3] = b4.w;
        }

        if (layer == 0 || layer == SKIP_LAYER) {
            bf16x8 bpe[2][NS];
#pragma unroll
            for (int s = 0; s < 2; ++s) split8<NS>(pe[s], bpe[s]);
#pragma unroll
            for (int jj = 0; jj < PL::ST_PE; ++jj) {
                const float *st = ws.wait();
                gemm_stage_bf<16, PL::KSB, NS>(st, lane, acc, bpe, PL::KSB * jj, ws);
            }
        }
        if (layer != 0) {
#pragma unroll
            for (int jj = 0; jj < PL::ST_256; ++jj) {
                const float *st = ws.wait();
                gemm_stage_bf<16, PL::KSB, NS>(st, lane, acc, bin, PL::KSB * jj, ws);
            }
        }
        // epilogue: ReLU (trunk), activation store, mask, sigma head, split into the next layer's B fragments
        unsigned mk0 = 0u, mk1 = 0u;
        float sg[4] = {0.f, 0.f, 0.f, 0.f};
        float *dst = SAVE ? a.acts + (layer < 8 ? a.al.h[layer] : a.al.feat) : nullptr;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            floatx4 x[2] = {acc[2 * s], acc[2 * s + 1]};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = 2 * s + u;
                if (layer < 8) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[u][r] = fmaxf(x[u][r], 0.f);
                }
                if (SAVE) store_tile16(dst, p, W, q, t, x[u], valid);
                if (layer < 8) {
                    unsigned m = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) m |= (x[u][r] > 0.f ? 1u : 0u) << r;
                    if (t < 8) mk0 |= m << (4 * t); else mk1 |= m << (4 * (t - 8));
                }
                if (layer == 7) {
                    const float4 w4 = *(const float4 *)(rf + PL::N_WSIG + 16 * t + 4 * q);
                    sg[0] = fmaf(w4.x, x[u][0], sg[0]); sg[1] = fmaf(w4.y, x[u][1], sg[1]);
                    sg[2] = fmaf(w4.z, x[u][2], sg[2]); sg[3] = fmaf(w4.w, x[u][3], sg[3]);
                }
            }
            split_pair<NS>(x[0], x[1], bin[s]);
        }
        if (SAVE && valid && layer < 8) *(uint2 *)(a.acts + a.al.hm[layer] + ((size_t)p * 4 + q) * 2) = make_uint2(mk0, mk1);
        if (layer == 7) {
            float s = (sg[0] + sg[1]) + (sg[2] + sg[3]);
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            sigma_raw = s + rf[PL::N_BHEAD];
        }
    }

    {
        float nz = 0.f;
        if (a.ns.noise) nz = a.ns.noise[p];
        else if (a.ns.device_rng) nz = rng_normal(a.ns.seed, a.ns.offset, a.ns.stream, (uint64_t)p);
        const float sgm = fmaxf(__fadd_rn(sigma_raw, __fmul_rn(nz, a.ns.std)), 0.f);
        if (valid && q == 0) a.sigma[p] = sgm;
    }

    // ---------------------------------------------------------------- view branch
    floatx4 vb[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const float4 b4 = *(const float4 *)(rf + PL::N_BVIEW + 16 * t + 4 * q);
        vb[t][0] = b4.x; vb[t][1] = b4.y; vb[t][2] = b4.z; vb[t][3] = b4.w;
    }
#pragma unroll
    for (int jj = 0; jj < PL::ST_VIEW_F; ++jj) {
        const float *st = ws.wait();
        gemm_stage_bf<8, PL::KSV, NS>(st, lane, vb, bin, PL::KSV * jj, ws);
    }

#pragma unroll 1
    for (int dsel = 0; dsel <= a.src.V; ++dsel) {
        float dir[3];
        if (dsel == 0) { dir[0] = pc.dir[0]; dir[1] = pc.dir[1]; dir[2] = pc.dir[2]; }
        else secondary_dir(a.src, pc, dsel - 1, dir);
        float ped[1][8];
        encode_bn<LV, 1>(dir, q, ped);
        bf16x8 bpd[1][NS];
        split8<NS>(ped[0], bpd[0]);
        floatx4 g[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) g[t] = vb[t];
        { NoStream none; gemm_stage_bf<8, 1, NS>(res + PL::R_DIRW, lane, g, bpd, 0, none); }
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) g[t][r] = fmaxf(g[t][r], 0.f);
        if (SAVE) {
#pragma unroll
            for (int t = 0; t < 8; ++t) store_tile16(a.acts + a.al.g[dsel], p, WV, q, t, g[t], valid);
            if (valid) {
                float *row = a.acts + a.al.ped[dsel] + (size_t)p * DVE_PAD + 8 * q;
                *(float4 *)(row) = make_float4(ped[0][0], ped[0][1], ped[0][2], ped[0][3]);
                *(float4 *)(row + 4) = make_float4(ped[0][4], ped[0][5], ped[0][6], ped[0][7]);
            }
        }
        float qv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float *wo = rf + PL::N_WOUT + c * WV + 4 * q;
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float4 w4 = *(const float4 *)(wo + 16 * t);
                s0 = fmaf(w4.x, g[t][0], s0);
                s1 = fmaf(w4.y, g[t][1], s1);
                s0 = fmaf(w4.z, g[t][2], s0);
                s1 = fmaf(w4.w, g[t][3], s1);
            }
            float s = s0 + s1;
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            qv[c] = sigmoidf_(s + rf[PL::N_BHEAD + 1 + c]);
        }
        if (valid && q == 0) {
            if (dsel == 0) {
                a.rgb[3 * p + 0] = qv[0]; a.rgb[3 * p + 1] = qv[1]; a.rgb[3 * p + 2] = qv[2];
                a.vis[p] = qv[3];
            } else {
                a.vis2[p * a.src.V + (dsel - 1)] = qv[3];
            }
        }
    }
}

template <bool SAVE, int NS>
static int launch_one_n(const MlpFwdArgs &a, unsigned grid, hipStream_t st) {
    const size_t lds = (size_t)BnPlan<NS>::LDS_F * sizeof(float);
    VN_HIP(hipFuncSetAttribute((const void *)k_mlp_fwd_bf16n<SAVE, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_mlp_fwd_bf16n<SAVE, NS>), dim3(grid), dim3(BnPlan<NS>::WG), lds, st, a);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

// a.packed must point at the narrow bf16 image of the requested precision
int launch_mlp_fwd_bf16n(const MlpFwdArgs &a, int precision, hipStream_t st) {
    if (a.src.P <= 0) return VIPNERF_OK;
    const unsigned grid = (unsigned)((a.src.P + MLP_PTS_PER_WG - 1) / MLP_PTS_PER_WG);
    if (precision == 1) return a.acts ? launch_one_n<true, 2>(a, grid, st) : launch_one_n<false, 2>(a, grid, st);
    if (precision == 2) return a.acts ? launch_one_n<true, 3>(a, grid, st) : launch_one_n<false, 3>(a, grid, st);
    set_error("mlp_fwd_bf16n: precision %d", precision);
    return VIPNERF_E_ARG;
}

}  // namespace vn
