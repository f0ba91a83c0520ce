// MLP.forward in EXACT fp32 (the headline arithmetic; reference src/models/VipNeRF01.py:509-596): v_mfma_f32_16x16x4_f32 on the narrow layout
// (vipnerf_bf16n.h: 16 points per wave, two waves per SIMD, BnPlan<2>'s 64 KiB weight stages, f32q fragments) -- the forward counterpart of
// vipnerf_mlp_bwd_f32.hip, written for this one arithmetic instead of as an instantiation of k_mlp_fwd_bf16n's template.
//
// Same arithmetic, bit for bit, as k_mlp_fwd_bf16n<SAVE, 2, false, 3 | 0, true>: accumulators start at the bias, k ascending (layer 5: the 256
// h columns, then gamma(x)), ReLU as one integer max in training / v_max in eval, the sigma head summed over the tiles in ascending order,
// the view tail unchanged.  What differs is WHEN a layer's epilogue runs: the ReLU, the ReLU bits and (layer 7) the sigma head's products are
// applied to the 64 raw accumulators INSIDE the next layer's GEMM -- the two operand k-steps the first weight stage consumes ahead of it, the
// other six behind MFMA group VN_F32B_CONV_GROUP of stages 0..2, where a wave64 VALU instruction issues in the 28 cycles a 32-cycle fp32 MFMA
// leaves free.  Before, the younger wave of every SIMD ran each layer's whole epilogue with the MFMA pipe idle (4.4k cycles per layer in
// training, 2.3k in eval: profiles/r04_ablation_pt2.md section 6).
#include "vipnerf_bf16n.h"
#include "vipnerf_mlp.h"
#include "vipnerf_mlp_pt2.h"

namespace vn {

TS_DECL(g_f32f_timeline);
#define TSFF(tag) TS_AT(g_f32f_timeline, tag)

typedef BnPlan<2> PLFF;

// operand k-step s of the next GEMM <- ReLU of raw accumulator tiles 2s, 2s + 1; training: the tiles' ReLU bits appended to mk[t >> 3]
// (push_nibble: after eight tiles the first sits in bits 0..3); SIG: the sigma head's products w_sigma . h_8 accumulated tile by tile
template <bool SAVE>
__device__ __forceinline__ void fwd_conv_kstep(const floatx4 *xr, f32q (*bin)[2], int s, unsigned *mk, bool sig, float *sg, const float *wsig) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = 2 * s + u;
        floatx4 x;
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = SAVE ? relu_bits(xr[t][r], 0) : fmaxf(xr[t][r], 0.f);
        if (SAVE) mk[t >> 3] = push_nibble(mk[t >> 3], positive_nibble(x));
        if (sig) {
            const float4 w4 = *(const float4 *)(wsig + 16 * t);
            sg[0] = fmaf(w4.x, x[0], sg[0]); sg[1] = fmaf(w4.y, x[1], sg[1]);
            sg[2] = fmaf(w4.z, x[2], sg[2]); sg[3] = fmaf(w4.w, x[3], sg[3]);
        }
        bin[s][u].v = x;
    }
}

// besides its MFMAs a weight stage (operand k-steps s0, s0 + 1) sends the fp32 stores of those k-steps (training: h_layer = the operand) and
// converts the k-steps the NEXT stage consumes (s0 + 2, s0 + 3); the stage that completes the second mask word stores the layer's ReLU bits
template <bool SAVE, bool CONV>
struct F32FwdMid {
    float *dst; int64_t p; int q, wave, s0;
    f32q (*bin)[2];
    const floatx4 *xr;
    unsigned *mk;
    bool sig; float *sg; const float *wsig;
    uint2 *mask_dst;
    template <int g, int NG> static constexpr bool active() {
        return (SAVE && (g == VN_STORE_GROUP_A || g == VN_STORE_GROUP_B)) || (CONV && g == VN_F32B_CONV_GROUP);
    }
    template <int g, int NG>
    __device__ __forceinline__ void at() const {
        if (CONV && g == VN_F32B_CONV_GROUP) {
            fwd_conv_kstep<SAVE>(xr, bin, s0 + 2, mk, sig, sg, wsig);
            fwd_conv_kstep<SAVE>(xr, bin, s0 + 3, mk, sig, sg, wsig);
            if (SAVE && s0 == 4 && !EXP_NO_STORES) *mask_dst = make_uint2(mk[0], mk[1]);
        }
        if (SAVE && (g == VN_STORE_GROUP_A || g == VN_STORE_GROUP_B) && (g == VN_STORE_GROUP_A) == (wave < 4) && !EXP_NO_STORES) {
#pragma unroll
            for (int s = s0; s < s0 + 2; ++s) {
                store_tile16(dst, p, W, q, 2 * s, bin[s][0].v);
                store_tile16(dst, p, W, q, 2 * s + 1, bin[s][1].v);
            }
        }
    }
};

template <bool SAVE>
__global__ __launch_bounds__(PLFF::WG) void k_mlp_fwd_f32(MlpFwdArgs a) {
    typedef PLFF PL;
    typedef f32q FR;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *res = lds;
    float *stage_buf = lds + PL::R_TOTAL_PAD;
    const float *rf = res + PL::R_F32;                   // fp32 block, natural feature order

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, j = lane & 15;
    const int64_t p_raw = (int64_t)blockIdx.x * MLP_PTS_PER_WG + wave * 16 + j;
    const bool valid = p_raw < a.src.P;
    const int64_t p = valid ? p_raw : a.src.P - 1;

    TS_INIT();
    TSFF(TS_ENTRY);
    // training: the four older waves issue a quarter of every stage's DMA each and wait for it with a counted vmcnt (their activation stores stay in flight);
    // eval: the same stream (0.908 -> 0.918 of the peak against every wave issuing its eighth)
    // build switch VN_F32_EVAL_ROTATE (default 1, vipnerf_knobs.h)
    typename std::conditional<SAVE || VN_F32_EVAL_ROTATE, typename StreamOlder<PL>::type, typename StreamShared<PL>::type>::type ws;
    ws.start(a.packed + PL::PK_FWD, PL::F_STAGES, stage_buf, lane, wave);
    {
        const float4 *g4 = (const float4 *)(a.packed + PL::PK_RES);
        float4 *l4 = (float4 *)res;
        for (int i = tid; i < PL::R_TOTAL_PAD / 4; i += PL::WG) l4[i] = g4[i];
    }
    float pe[2][8];
    {
        PointCtx pc0;                        // scoped: the ray / direction data is re-read for the view branch rather
        load_point(a.src, p, pc0);           // than kept in 13 registers across the trunk
        encode_x16(pc0.x, q, pe);
    }
    if (SAVE && valid && !EXP_NO_PE) store_x16(a.acts + a.al.pex + (size_t)p * DPE_PAD, q, pe);
    FR bpe[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) split8<2>(pe[s], bpe[s]);

    FR bin[8][2];                            // the layer input as B fragments: k-step s <- C/D tiles 2s, 2s+1
    floatx4 acc[16], xr[16];
    unsigned mk[2] = {0u, 0u};
    float sg[4] = {0.f, 0.f, 0.f, 0.f};
    const float *wsig = rf + PL::N_WSIG + 4 * q;

    __syncthreads();                         // resident block visible
    TSFF(TS_RESIDENT);

    // ---------------------------------------------------------------- layer 0: gamma(x) only (one stage)
    {
        const float *bias = rf + PL::N_BIAS + 4 * q;
#pragma unroll
        for (int t = 0; t < 16; ++t) { const float4 b4 = *(const float4 *)(bias + 16 * t); acc[t][0] = b4.x; acc[t][1] = b4.y; acc[t][2] = b4.z; acc[t][3] = b4.w; }
        TSFF(TS_PRE);
        const float *st = ws.wait();
        TSFF(TS_POST);
        gemm_stage_bf<16, PL::KSB, 2>(st, lane, acc, bpe, 0, ws);
        TSFF(TS_END);
#pragma unroll
        for (int t = 0; t < 16; ++t) xr[t] = acc[t];
    }

    // ---------------------------------------------------------------- GEMMs 1..7 (trunk) and 8 (feature layer): operand = ReLU(output of GEMM - 1) = h_layer
#pragma unroll 1
    for (int layer = 1; layer <= 8; ++layer) {
        const float *bias = rf + (layer < 8 ? PL::N_BIAS + layer * W : PL::N_BFEAT) + 4 * q;
#pragma unroll
        for (int t = 0; t < 16; ++t) { const float4 b4 = *(const float4 *)(bias + 16 * t); acc[t][0] = b4.x; acc[t][1] = b4.y; acc[t][2] = b4.z; acc[t][3] = b4.w; }
        TSFF(TS_EPI_A);
        const bool sig = layer == 8;                      // h_8 also feeds the sigma head
        float *hdst = SAVE ? a.acts + a.al.h[layer - 1] : nullptr;
        uint2 *mdst = SAVE ? (uint2 *)(a.acts + a.al.hm[layer - 1] + ((size_t)p * 4 + q) * 2) : nullptr;
        mk[0] = 0u; mk[1] = 0u;
        fwd_conv_kstep<SAVE>(xr, bin, 0, mk, sig, sg, wsig);
        fwd_conv_kstep<SAVE>(xr, bin, 1, mk, sig, sg, wsig);
        TSFF(TS_EPI_B);
#pragma unroll
        for (int jj = 0; jj < PL::ST_256; ++jj) {
            // counted wait (training): since it issued this stage's DMA (behind group 0 of the stage before) the issuing wave has executed that
            // stage's 4 tile stores -- none if that stage was a gamma(x) stage (GEMMs 1 and 6 follow one)
            TSFF(TS_PRE);
            const float *st = SAVE ? (jj == 0 ? ws.template wait<4, 0>(layer == 1 || layer == 6) : ws.template wait<4>()) : ws.wait();
            TSFF(TS_POST);
            if (jj + 1 < PL::ST_256) {
                F32FwdMid<SAVE, true> mid{hdst, p, q, wave, 2 * jj, bin, xr, mk, sig, sg, wsig, mdst};
                gemm_stage_bf<16, PL::KSB, 2>(st, lane, acc, bin, PL::KSB * jj, ws, mid);
            } else {
                F32FwdMid<SAVE, false> mid{hdst, p, q, wave, 2 * jj, bin, xr, mk, sig, sg, wsig, mdst};
                gemm_stage_bf<16, PL::KSB, 2>(st, lane, acc, bin, PL::KSB * jj, ws, mid);
            }
            TSFF(TS_END);
        }
        if (layer == SKIP_LAYER) {                        // gamma(x) columns last (the packed image's order)
            TSFF(TS_PRE);
            const float *st = SAVE ? ws.template wait<4>() : ws.wait();
            TSFF(TS_POST);
            gemm_stage_bf<16, PL::KSB, 2>(st, lane, acc, bpe, 0, ws);
            TSFF(TS_END);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) xr[t] = acc[t];
        TSFF(15);
    }

    // ---------------------------------------------------------------- sigma (from the products gathered during GEMM 8's conversions)
    {
        float s = (sg[0] + sg[1]) + (sg[2] + sg[3]);
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const float sigma_raw = s + rf[PL::N_BHEAD];
        float nz = 0.f;
        if (a.ns.noise) nz = a.ns.noise[p];
        else if (a.ns.device_rng) nz = rng_normal(a.ns.seed, a.ns.offset, a.ns.stream, noise_index(a.ns, a.src, p));
        const float sgm = fmaxf(__fadd_rn(sigma_raw, __fmul_rn(nz, a.ns.std)), 0.f);
        if (valid && q == 0) a.sigma[p] = sgm;
    }

    // ---------------------------------------------------------------- the feature (no ReLU): stored, and the view layer's operand
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        bin[s][0].v = xr[2 * s]; bin[s][1].v = xr[2 * s + 1];
        if (SAVE && !EXP_NO_EXTRAS) {
            store_tile16(a.acts + a.al.feat, p, W, q, 2 * s, xr[2 * s]);
            store_tile16(a.acts + a.al.feat, p, W, q, 2 * s + 1, xr[2 * s + 1]);
        }
    }
    PointCtx pc;
    load_point(a.src, p, pc);
    floatx4 vb[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const float4 b4 = *(const float4 *)(rf + PL::N_BVIEW + 16 * t + 4 * q);
        vb[t][0] = b4.x; vb[t][1] = b4.y; vb[t][2] = b4.z; vb[t][3] = b4.w;
    }
#pragma unroll
    for (int jj = 0; jj < PL::ST_VIEW_F; ++jj) {
        TSFF(TS_PRE);
        // (behind GEMM 8's last stage -- 4 tile stores -- and the feature's 16)
        const float *st = SAVE ? (jj == 0 ? ws.template wait<20>() : ws.template wait<0>()) : ws.wait();
        TSFF(TS_POST);
        gemm_stage_bf<8, PL::KSV, 2>(st, lane, vb, bin, PL::KSV * jj, ws);
        TSFF(TS_VIEW);
    }
    stream_end(ws);
    TSFF(16);

#pragma unroll 1
    for (int dsel = 0; dsel <= a.src.V; ++dsel) {
        float dir[3];
        if (dsel == 0) { dir[0] = pc.dir[0]; dir[1] = pc.dir[1]; dir[2] = pc.dir[2]; }
        else secondary_dir(a.src, pc, dsel - 1, dir);
        TSFF(10);
        float ped[1][8];
        encode_d16(dir, q, ped);
        FR bpd[1][2];
        split8<2>(ped[0], bpd[0]);
        floatx4 g[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) g[t] = vb[t];
        { NoStream none; gemm_stage_bf<8, 1, 2>(res + PL::R_DIRW, lane, g, bpd, 0, none); }
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) g[t][r] = relu_lo<SAVE>(g[t][r], 0.f);     // (training: +0 | positive | NaN for the ReLU bits)
        TSFF(12);
        if (SAVE) {
            if (!EXP_NO_EXTRAS) {
#pragma unroll
                for (int t = 0; t < 8; ++t) store_tile16(a.acts + a.al.g[dsel], p, WV, q, t, g[t]);
                unsigned gmb = 0u;       // the view hidden's 32 ReLU bits per lane (bit 4 t + r): all k_mlp_bwd_f32 reads of it
#pragma unroll
                for (int t = 0; t < 8; ++t) gmb = push_nibble(gmb, positive_nibble(g[t]));
                ((unsigned *)(a.acts + a.al.gm[dsel]))[(size_t)p * 4 + q] = gmb;
            }
            if (valid && !EXP_NO_PE) store_d16(a.acts + a.al.ped[dsel] + (size_t)p * DVE_PAD, q, ped);
        }
        TSFF(13);
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
        TSFF(14);
    }
    TSFF(TS_LAST);
}

#if defined(VN_EXP) && VN_EXP == 50
extern "C" int vipnerf_exp_timeline_f32f(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_f32f_timeline), sizeof(unsigned long long) * (n < 2048 ? n : 2048));
}
#endif

template <bool SAVE>
static int launch_one_f32(const MlpFwdArgs &a, unsigned grid, hipStream_t st) {
    const size_t lds = (size_t)PLFF::LDS_F * sizeof(float);
    VN_HIP(hipFuncSetAttribute((const void *)k_mlp_fwd_f32<SAVE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_mlp_fwd_f32<SAVE>, dim3(grid), dim3(PLFF::WG), lds, st, a);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

int launch_mlp_fwd_f32(const MlpFwdArgs &a, hipStream_t st) {
    if (a.src.P <= 0) return VIPNERF_OK;
    const unsigned grid = (unsigned)((a.src.P + MLP_PTS_PER_WG - 1) / MLP_PTS_PER_WG);
    return a.acts ? launch_one_f32<true>(a, grid, st) : launch_one_f32<false>(a, grid, st);
}

}  // namespace vn
