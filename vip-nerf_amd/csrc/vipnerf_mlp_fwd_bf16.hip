// MLP.forward on split-precision bf16 MFMA (vipnerf_bf16.h): same structure as vipnerf_mlp_fwd.hip -- transposed,
// register-chained layers; weights streamed through LDS in fragment order -- with v_mfma_f32_32x32x16_bf16 and
// operands split into NS bf16 parts (NS = 2: "bf16x3", ~1e-5 relative; NS = 3: "bf16x6", fp32 grade).  Positional
// encodings, biases, ReLU, the sigma / output heads, sigmoids and all ray arithmetic stay fp32.
#include "vipnerf_bf16.h"
#include "vipnerf_mlp.h"

namespace vn {

// gamma_L(v) for the bf16 fragment: k-step s, element e of half h is feature 16 s + 8 h + e
template <int L, int NKS>
__device__ __forceinline__ void encode_bf(const float v[3], int h, float (&out)[NKS][8]) {
    float val[16 * NKS];
#pragma unroll
    for (int f = 0; f < 16 * NKS; ++f) val[f] = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) val[d] = v[d];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float s, c;
#if defined(VN_EXP) && VN_EXP == 7
            s = v[d] * (float)(1 << l); c = s + 1.f;   // timing experiment only: no sincos
#else
            sincosf(v[d] * (float)(1 << l), &s, &c);
#endif
            val[3 + 6 * l + d] = s;
            val[3 + 6 * l + 3 + d] = c;
        }
#pragma unroll
    for (int s = 0; s < NKS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) out[s][e] = h ? val[16 * s + 8 + e] : val[16 * s + e];
}

template <bool SAVE, int NS>
__global__ __launch_bounds__(MLP_WG) void k_mlp_fwd_bf16(MlpFwdArgs a) {
    typedef BfPlan<NS> PL;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *res = lds;
    float *stage_buf = lds + PL::R_TOTAL_PAD;
    const float *rf = res + PL::R_F32 - R_BIAS;          // rf[R_x] addresses the fp32 entries like the fp32 image

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, j = lane & 31;
    const int64_t p_raw = (int64_t)blockIdx.x * MLP_PTS_PER_WG + wave * 32 + j;
    const bool valid = p_raw < a.src.P;
    const int64_t p = valid ? p_raw : a.src.P - 1;

    WStreamT<PL::CH, PL::NBUF> ws;
    ws.start(a.packed + PL::PK_FWD, PL::F_STAGES, stage_buf, lane, wave);
    {
        const float4 *g4 = (const float4 *)(a.packed + PL::PK_RES);
        float4 *l4 = (float4 *)res;
        for (int i = tid; i < PL::R_TOTAL_PAD / 4; i += MLP_WG) l4[i] = g4[i];
    }

    PointCtx pc;
    load_point(a.src, p, pc);
    float pe[4][8];
    encode_bf<LP, 4>(pc.x, h, pe);
    if (SAVE && valid) {
        float *row = a.acts + a.al.pex + (size_t)p * DPE_PAD + 8 * h;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            *(float4 *)(row + 16 * s) = make_float4(pe[s][0], pe[s][1], pe[s][2], pe[s][3]);
            *(float4 *)(row + 16 * s + 4) = make_float4(pe[s][4], pe[s][5], pe[s][6], pe[s][7]);
        }
    }

    bf16x8 bin[16][NS];                      // the layer input as B fragments: k-step s = 2*tile + u
    floatx16 acc[8];
    float sigma_raw = 0.f;

    // ---------------------------------------------------------------- trunk (layers 0..7) + feature layer (8)
#pragma unroll 1
    for (int layer = 0; layer < 9; ++layer) {
        if (layer == 0) __syncthreads();
        const float *bias = rf + (layer < 8 ? R_BIAS + layer * W : R_BFEAT) + h * 16;
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = *(const floatx16 *)(bias + t * 32);

        if (layer == 0 || layer == SKIP_LAYER) {
            bf16x8 bpe[4][NS];
#pragma unroll
            for (int s = 0; s < 4; ++s) split8<NS>(pe[s], bpe[s]);
#pragma unroll
            for (int jj = 0; jj < PL::ST_PE; ++jj) {
                const float *st = ws.wait();
                gemm_stage_bf<8, PL::KSB, NS>(st, lane, acc, bpe, PL::KSB * jj, ws);
            }
        }
        if (layer != 0) {
#pragma unroll
            for (int jj = 0; jj < PL::ST_256; ++jj) {
                const float *st = ws.wait();
                gemm_stage_bf<8, PL::KSB, NS>(st, lane, acc, bin, PL::KSB * jj, ws);
            }
        }
        // epilogue: ReLU (trunk), activation store, mask, sigma head, split into the next layer's B fragments
        uint4 mk = make_uint4(0u, 0u, 0u, 0u);
        float sg[4] = {0.f, 0.f, 0.f, 0.f};
        float *dst = SAVE ? a.acts + (layer < 8 ? a.al.h[layer] : a.al.feat) : nullptr;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            floatx16 x = acc[t];
            if (layer < 8) {
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = fmaxf(x[r], 0.f);
            }
            if (SAVE) store_tile(dst, p, W, h, t, x, valid);
            if (layer < 8) {
                unsigned m = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) m |= (x[r] > 0.f ? 1u : 0u) << r;
                const unsigned sh = m << (16 * (t & 1));
                if ((t >> 1) == 0) mk.x |= sh; else if ((t >> 1) == 1) mk.y |= sh; else if ((t >> 1) == 2) mk.z |= sh; else mk.w |= sh;
            }
            if (layer == 7) {
                const float *wsg = rf + R_WSIG + h * 128 + 16 * t;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w4 = *(const float4 *)(wsg + 4 * q);
                    sg[0] = fmaf(w4.x, x[4 * q], sg[0]); sg[1] = fmaf(w4.y, x[4 * q + 1], sg[1]);
                    sg[2] = fmaf(w4.z, x[4 * q + 2], sg[2]); sg[3] = fmaf(w4.w, x[4 * q + 3], sg[3]);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float xs[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) xs[e] = x[8 * u + e];
                split8<NS>(xs, bin[2 * t + u]);
            }
        }
        if (SAVE && valid && layer < 8) *(uint4 *)(a.acts + a.al.hm[layer] + ((size_t)p * 2 + h) * 4) = mk;
        if (layer == 7) {
            float s = (sg[0] + sg[1]) + (sg[2] + sg[3]);
            s += __shfl_xor(s, 32, 64);
            sigma_raw = s + rf[R_BHEAD];
        }
    }

    {
        float nz = 0.f;
        if (a.ns.noise) nz = a.ns.noise[p];
        else if (a.ns.device_rng) nz = rng_normal(a.ns.seed, a.ns.offset, a.ns.stream, noise_index(a.ns, a.src, p));
        const float sgm = fmaxf(__fadd_rn(sigma_raw, __fmul_rn(nz, a.ns.std)), 0.f);
        if (valid && h == 0) a.sigma[p] = sgm;
    }

    // ---------------------------------------------------------------- view branch
    floatx16 vb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) vb[t] = *(const floatx16 *)(rf + R_BVIEW + t * 32 + h * 16);
#pragma unroll
    for (int jj = 0; jj < PL::ST_VIEW_F; ++jj) {
        const float *st = ws.wait();
        gemm_stage_bf<4, PL::KSV, NS>(st, lane, vb, bin, PL::KSV * jj, ws);
    }

#pragma unroll 1
    for (int dsel = 0; dsel <= a.src.V; ++dsel) {
        float dir[3];
        if (dsel == 0) { dir[0] = pc.dir[0]; dir[1] = pc.dir[1]; dir[2] = pc.dir[2]; }
        else secondary_dir(a.src, pc, dsel - 1, dir);
        float ped[2][8];
        encode_bf<LV, 2>(dir, h, ped);
        bf16x8 bpd[2][NS];
#pragma unroll
        for (int s = 0; s < 2; ++s) split8<NS>(ped[s], bpd[s]);
        floatx16 g[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) g[t] = vb[t];
        { NoStream none; gemm_stage_bf<4, 2, NS>(res + PL::R_DIRW, lane, g, bpd, 0, none); }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) g[t][r] = fmaxf(g[t][r], 0.f);
        if (SAVE) {
            store_frag<4>(a.acts + a.al.g[dsel], p, WV, h, g, valid);
            if (valid) {
                float *row = a.acts + a.al.ped[dsel] + (size_t)p * DVE_PAD + 8 * h;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    *(float4 *)(row + 16 * s) = make_float4(ped[s][0], ped[s][1], ped[s][2], ped[s][3]);
                    *(float4 *)(row + 16 * s + 4) = make_float4(ped[s][4], ped[s][5], ped[s][6], ped[s][7]);
                }
            }
        }
        float qv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float *wo = rf + R_WOUT + h * 256 + c * 64;
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
            qv[c] = sigmoidf_(s + rf[R_BHEAD + 1 + c]);
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

template <bool SAVE, int NS>
static int launch_one(const MlpFwdArgs &a, unsigned grid, hipStream_t st) {
    const size_t lds = (size_t)BfPlan<NS>::LDS_F * sizeof(float);
    VN_HIP(hipFuncSetAttribute((const void *)k_mlp_fwd_bf16<SAVE, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_mlp_fwd_bf16<SAVE, NS>), dim3(grid), dim3(MLP_WG), lds, st, a);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

// a.packed must point at the bf16 image of the requested precision
int launch_mlp_fwd_bf16(const MlpFwdArgs &a, int precision, hipStream_t st) {
    if (a.src.P <= 0) return VIPNERF_OK;
    const unsigned grid = (unsigned)((a.src.P + MLP_PTS_PER_WG - 1) / MLP_PTS_PER_WG);
    if (precision == 1) return a.acts ? launch_one<true, 2>(a, grid, st) : launch_one<false, 2>(a, grid, st);
    if (precision == 2) return a.acts ? launch_one<true, 3>(a, grid, st) : launch_one<false, 3>(a, grid, st);
    set_error("mlp_fwd_bf16: precision %d", precision);
    return VIPNERF_E_ARG;
}

}  // namespace vn
