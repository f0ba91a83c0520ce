// MLP.forward of the fp16x3 arithmetics (operands split into two fp16 parts, three cross terms per product), narrow-wave layout
// (vipnerf_bf16n.h): 16 points per wave on v_mfma_f32_16x16x32_f16, 8 waves (two per SIMD) and 128 points per workgroup; lane (j, q) holds
// features 16T + 4q .. +3 of tile T for point j.  (The exact-fp32 arithmetic: vipnerf_mlp_fwd_f32.hip; the single-MFMA 16-bit modes:
// vipnerf_mlp_fwd_pt2.hip.)
#include <type_traits>
#include "vipnerf_bf16n.h"
#include "vipnerf_mlp.h"
#include "vipnerf_mlp_pt2.h"


namespace vn {

// F16: fp16 fragments with power-of-two operand scaling (vipnerf_bf16n.h); otherwise bf16 fragments.
// H16 (with F16 and SAVE): how the trunk activations h_1..h_8 are stored for the weight-gradient GEMMs: 0 = fp32 [P][256];
// 1 = fp16 high parts only (FP16X3H); 2 = both fp16 parts in the 16 bytes a lane owns per tile (FP16X3, store_pair_split:
// the same bytes as fp32, but already split).  With H16 != 0 the stores of a layer's output leave from the NEXT layer's
// weight stages (DEFER; the stored form is that layer's B operand).  Everything else stays fp32.
// H16 == 4 (single-MFMA modes, VN_T16): every stored operand is 16-bit in the tile-blocked layout T16 (vipnerf_bf16n.h: store_t16) --
// h_1..h_8, the feature, the view hidden and its ReLU bits per direction, gamma(x) / gamma(dir) in their slot order -- written from the
// B fragments the GEMMs consume anyway.
// (The exact-fp32 arithmetic has kernels of its own: vipnerf_mlp_fwd_f32.hip.  This template serves the fp16x3 modes.)
int launch_mlp_fwd_f32(const MlpFwdArgs &a, hipStream_t st);
TS_DECL(g_n_timeline);
#define TSN(tag) TS_AT(g_n_timeline, tag)

template <bool SAVE, int NS, bool F16, int H16 = 0>
__global__ __launch_bounds__(BnPlan<NS>::WG) void k_mlp_fwd_bf16n(MlpFwdArgs a) {
    typedef BnPlan<NS> PL;
    typedef typename FragOf<F16, false>::type FR;
    static_assert(H16 != 3 && (H16 != 4 || NS == 1 || (NS == 2 && F16)), "arithmetic");
    constexpr bool T16 = SAVE && H16 == 4;
    constexpr float XS = F16 ? F16_XSCALE : 1.f;           // B operands are split as XS * x
    constexpr float AU = F16 ? F16_ACC_UNSCALE : 1.f;
    constexpr bool DEFER = SAVE && H16 != 0 && VN_DEFER_STORES;   // h_1..h_8 leave from the next layer's stages (vipnerf_bf16n.h)
    constexpr int EPI_STORES = SAVE ? (H16 == 4 ? 8 * T16_SPK : 16) : 0;   // vector-memory instructions every wave issues per layer epilogue (lower bound; T16: the feature's)
    constexpr int DEF_SPK = H16 == 4 ? T16_SPK : 2;        // deferred store instructions per operand k-step
    constexpr int S_PER_STAGE = 8 / PL::ST_256;            // operand k-steps a stage's deferred stores cover (2 stores each)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *res = lds;
    float *stage_buf = lds + PL::R_TOTAL_PAD;
    const float *rf = res + PL::R_F32;                   // fp32 block, natural feature order

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, j = lane & 15;
    const int64_t p_raw = (int64_t)blockIdx.x * MLP_PTS_PER_WG + wave * 16 + j;
    const bool valid = p_raw < a.src.P;
    const int64_t p = valid ? p_raw : a.src.P - 1;
    const int64_t grp = (int64_t)blockIdx.x * (MLP_PTS_PER_WG / 16) + wave;   // T16: this wave's 16-point group (valid is wave-uniform: P % 16 == 0)

    TS_INIT();
    TSN(TS_ENTRY);
    typename StreamOf<PL, PL::SKEW>::type ws;
    ws.start(a.packed + PL::PK_FWD, PL::F_STAGES, stage_buf, lane, wave);
    stream_counted(ws, !T16 || valid);      // a wave beyond P skips its (predicated) T16 stores: its counted waits would not hold
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
    if (SAVE && !T16 && valid && !EXP_NO_PE) store_x16(a.acts + a.al.pex + (size_t)p * DPE_PAD, q, pe);
    if (F16) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) pe[s][e] *= XS;
    }

    FR bin[8][NS];                           // the layer input as B fragments: k-step s <- C/D tiles 2s, 2s+1
    floatx4 acc[16];
    float sigma_raw = 0.f;

    // ---------------------------------------------------------------- trunk (layers 0..7) + feature layer (8)
    __syncthreads();                         // resident block visible
    TSN(TS_RESIDENT);
    stream_begin(ws);
#pragma unroll 1
    for (int layer = 0; layer < 9; ++layer) {
        const float *bias = rf + (layer < 8 ? PL::N_BIAS + layer * W : PL::N_BFEAT) + 4 * q;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float4 b4 = *(const float4 *)(bias + 16 * t);
            acc[t][0] = b4.x; acc[t][1] = b4.y; acc[t][2] = b4.z; acc[t][3] = b4.w;     // packed as AS * bias
        }

        if (layer != 0) {
#pragma unroll
            for (int jj = 0; jj < PL::ST_256; ++jj) {
                // younger than the stage's DMA -- first stage: the previous layer's epilogue (>= 16 tile stores, or just
                // the mask store when the tiles are deferred); later stages: the deferred stores behind the stage before
                TSN(TS_PRE);
                const float *st = jj == 0 ? ws.template wait<DEFER ? 1 : EPI_STORES>() : ws.template wait<DEFER ? DEF_SPK * S_PER_STAGE : 0>();
                TSN(TS_POST);
                if (DEFER) {                             // bin = the fp16 parts of h_layer, the output of layer - 1
                    DeferredStores<H16, NS, FR, S_PER_STAGE> ds{a.acts + a.al.h[layer - 1], p, q, wave, S_PER_STAGE * jj, bin, grp, j, valid};
                    gemm_stage_bf<16, PL::KSB, NS>(st, lane, acc, bin, PL::KSB * jj, ws, ds);
                } else {
                    gemm_stage_bf<16, PL::KSB, NS>(st, lane, acc, bin, PL::KSB * jj, ws);
                }
                TSN(TS_END);
            }
        }
        if (layer == 0 || layer == SKIP_LAYER) {         // gamma(x) columns last: bin is dead, its registers hold bpe
            FR bpe[PL::PE_KS][NS];
#pragma unroll
            for (int s = 0; s < 2; ++s) split8<NS>(pe[s], bpe[s]);
            if (T16 && layer == 0 && valid && !EXP_NO_PE) {   // gamma(x), slot order: column 16 q + u of a 64-wide T16 array = tile q
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                char *row = (char *)(a.acts + a.al.pex) + ((size_t)grp * 4 + q) * 512 + j * 32;
                __builtin_nontemporal_store(__builtin_bit_cast(u4, bpe[0][0]), (u4 *)row);
                __builtin_nontemporal_store(__builtin_bit_cast(u4, bpe[1][0]), (u4 *)(row + 16));
            }
#pragma unroll
            for (int s = 2; s < PL::PE_KS; ++s) {            // padding k-steps of the single-MFMA plan (zero weight columns)
                const float zero[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                split8<NS>(zero, bpe[s]);
            }
#pragma unroll
            for (int jj = 0; jj < PL::ST_PE; ++jj) {
                TSN(TS_PRE);
                const float *st = ws.wait();
                TSN(TS_POST);
                gemm_stage_bf<16, PL::KSB, NS>(st, lane, acc, bpe, PL::KSB * jj, ws);
                TSN(TS_END);
            }
        }
        // epilogue: ReLU (trunk), activation store, mask, sigma head, split into the next layer's B fragments.
        // Kept free of per-tile branches on `layer`: ReLU is a max with a per-layer bound (0, or -inf for the feature
        // layer), the sigma head is one block for layer 7.
        const float lo = relu_bound<F16>(layer < 8);
        if (layer == 7) {
            float sg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float4 w4 = *(const float4 *)(rf + PL::N_WSIG + 16 * t + 4 * q);
                sg[0] = fmaf(w4.x, relu_lo<F16>(acc[t][0] * AU, 0.f), sg[0]); sg[1] = fmaf(w4.y, relu_lo<F16>(acc[t][1] * AU, 0.f), sg[1]);
                sg[2] = fmaf(w4.z, relu_lo<F16>(acc[t][2] * AU, 0.f), sg[2]); sg[3] = fmaf(w4.w, relu_lo<F16>(acc[t][3] * AU, 0.f), sg[3]);
            }
            float s = (sg[0] + sg[1]) + (sg[2] + sg[3]);
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            sigma_raw = s + rf[PL::N_BHEAD];
        }
        unsigned mk0 = 0u, mk1 = 0u;
        float *dst = SAVE ? a.acts + (layer < 8 ? a.al.h[layer] : a.al.feat) : nullptr;
#if defined(VN_EXP) && VN_EXP == 22
        if (acc[0][0] == 12345.f)                    // timing experiment only: no ReLU / split epilogue
#endif
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            floatx4 x[2] = {acc[2 * s], acc[2 * s + 1]};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = 2 * s + u;
#pragma unroll
                for (int r = 0; r < 4; ++r) x[u][r] = relu_lo<F16>(x[u][r] * AU, lo);   // (+0 | positive | NaN for the bit masks)
                if (SAVE) {
                    if (!(H16 && layer < 8) && !T16 && !(layer < 8 ? EXP_NO_STORES : EXP_NO_EXTRAS)) store_tile16(dst, p, W, q, t, x[u]);
                    if (F16) {
                        if (t < 8) mk0 = push_nibble(mk0, positive_nibble(x[u])); else mk1 = push_nibble(mk1, positive_nibble(x[u]));
                    } else {
                        unsigned m = 0;
#pragma unroll
                        for (int r = 0; r < 4; ++r) m |= (x[u][r] > 0.f ? 1u : 0u) << r;
                        if (t < 8) mk0 |= m << (4 * t); else mk1 |= m << (4 * (t - 8));
                    }
                }
            }
            if (F16) { x[0] *= XS; x[1] *= XS; }
#if defined(VN_EXP) && VN_EXP == 21
            if (x[0][0] == 12345.f)                  // timing experiment only: no operand split
#endif
            split_pair<NS>(x[0], x[1], bin[s]);
            if (SAVE && !DEFER && H16 == 1 && layer < 8) store_pair16h(dst, p, W, q, s, bin[s][0]);
            if (SAVE && !DEFER && H16 == 2 && layer < 8) store_pair_split(dst, p, W, q, s, bin[s][0], bin[s][NS > 1 ? 1 : 0]);
            if (SAVE && !DEFER && H16 == 3 && layer < 8) store_pair_f32(dst, p, W, q, s, bin[s][0], bin[s][NS > 1 ? 1 : 0]);
            if (T16 && (layer == 8 || !DEFER) && valid && !EXP_NO_EXTRAS) store_t16(dst, grp, 16, s, j, q, bin[s][0]);   // the feature (h_1..h_8: deferred)
        }
        if (SAVE && layer < 8 && !EXP_NO_STORES) *(uint2 *)(a.acts + a.al.hm[layer] + ((size_t)p * 4 + q) * 2) = make_uint2(mk0, mk1);
    }

    {
        float nz = 0.f;
        if (a.ns.noise) nz = a.ns.noise[p];
        else if (a.ns.device_rng) nz = rng_normal(a.ns.seed, a.ns.offset, a.ns.stream, noise_index(a.ns, a.src, p));
        const float sgm = relu_lo<F16>(__fadd_rn(sigma_raw, __fmul_rn(nz, a.ns.std)), 0.f);
        if (valid && q == 0) a.sigma[p] = sgm;
    }

    // ---------------------------------------------------------------- view branch
    PointCtx pc;
    load_point(a.src, p, pc);
    floatx4 vb[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const float4 b4 = *(const float4 *)(rf + PL::N_BVIEW + 16 * t + 4 * q);
        vb[t][0] = b4.x; vb[t][1] = b4.y; vb[t][2] = b4.z; vb[t][3] = b4.w;                 // packed as AS * bias
    }
#pragma unroll
    for (int jj = 0; jj < PL::ST_VIEW_F; ++jj) {
        TSN(TS_PRE);
        const float *st = jj == 0 ? ws.template wait<EPI_STORES>() : ws.template wait<0>();   // behind the feature layer's epilogue
        TSN(TS_POST);
        gemm_stage_bf<8, PL::KSV, NS>(st, lane, vb, bin, PL::KSV * jj, ws);
        TSN(TS_VIEW);
    }
    stream_end(ws);

#if defined(VN_EXP) && VN_EXP == 23
    if (sigma_raw == 12345.f)                        // timing experiment only: no view-branch tail
#endif
#pragma unroll 1
    for (int dsel = 0; dsel <= a.src.V; ++dsel) {
        float dir[3];
        if (dsel == 0) { dir[0] = pc.dir[0]; dir[1] = pc.dir[1]; dir[2] = pc.dir[2]; }
        else secondary_dir(a.src, pc, dsel - 1, dir);
        float ped[1][8];
        encode_d16(dir, q, ped);
        FR bpd[1][NS];
        {
            float sc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) sc[e] = ped[0][e] * XS;
            split8<NS>(sc, bpd[0]);
        }
        floatx4 g[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) g[t] = vb[t];
        { NoStream none; gemm_stage_bf<8, 1, NS>(res + PL::R_DIRW, lane, g, bpd, 0, none); }
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) g[t][r] = relu_lo<F16 || T16>(g[t][r] * AU, 0.f);     // (+0 | positive | NaN for the ReLU bits)
        if (T16) {
            if (valid && !EXP_NO_EXTRAS) {
                // view hidden as 16-bit T16 (8 tiles) + its 32 ReLU bits per lane (bit 4 t + r), which is all the data-gradient
                // kernel needs of it: [P][4] words behind the T16 array in the same slot
                unsigned gm = 0u;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    FR gh[NS];
                    split_pair<NS>(g[2 * s], g[2 * s + 1], gh);
                    store_t16(a.acts + a.al.g[dsel], grp, 8, s, j, q, gh[0]);
                    gm = push_nibble(gm, positive_nibble(g[2 * s]));
                    gm = push_nibble(gm, positive_nibble(g[2 * s + 1]));
                }
                ((unsigned *)(a.acts + a.al.g[dsel] + (size_t)a.src.P * (WV / 2)))[(size_t)p * 4 + q] = gm;
            }
            if (valid && !EXP_NO_PE) {     // gamma(dir), slot order: column 8 q + e of a 32-wide T16 array
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                char *row = (char *)(a.acts + a.al.ped[dsel]) + ((size_t)grp * 2 + (q >> 1)) * 512 + j * 32 + (q & 1) * 16;
                __builtin_nontemporal_store(__builtin_bit_cast(u4, bpd[0][0]), (u4 *)row);
            }
        } else if (SAVE) {
            if (!EXP_NO_EXTRAS) {
#pragma unroll
                for (int t = 0; t < 8; ++t) store_tile16(a.acts + a.al.g[dsel], p, WV, q, t, g[t]);
            }
            if (valid && !EXP_NO_PE) store_d16(a.acts + a.al.ped[dsel] + (size_t)p * DVE_PAD, q, ped);
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
    TSN(TS_LAST);
}

#if defined(VN_EXP) && VN_EXP == 50
extern "C" int vipnerf_exp_timeline_n(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_n_timeline), sizeof(unsigned long long) * (n < 2048 ? n : 2048));
}
#endif

template <bool SAVE, int NS, bool F16 = false, int H16 = 0>
static int launch_one_n(const MlpFwdArgs &a, unsigned grid, hipStream_t st) {
    const size_t lds = (size_t)BnPlan<NS>::LDS_F * sizeof(float);
    VN_HIP(hipFuncSetAttribute((const void *)k_mlp_fwd_bf16n<SAVE, NS, F16, H16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_mlp_fwd_bf16n<SAVE, NS, F16, H16>), dim3(grid), dim3(BnPlan<NS>::WG), lds, st, a);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

// a.packed must point at the narrow bf16 image of the requested precision
int launch_mlp_fwd_bf16n(const MlpFwdArgs &a, int precision, hipStream_t st) {
    if (a.src.P <= 0) return VIPNERF_OK;
    const unsigned grid = (unsigned)((a.src.P + MLP_PTS_PER_WG - 1) / MLP_PTS_PER_WG);
    if (precision == 0) return launch_mlp_fwd_f32(a, st);      // the exact-fp32 kernels of vipnerf_mlp_fwd_f32.hip
    // (precisions 1 / 2, the split-bf16 arithmetics bf16x3 / bf16x6, were retired with ABI 5)
    if (precision == 3) return a.acts ? launch_one_n<true, 2, true, VN_F16_PRESPLIT ? 2 : 0>(a, grid, st) : launch_one_n<false, 2, true>(a, grid, st);
    if (precision == 4) return a.acts ? launch_one_n<true, 2, true, VN_T16 ? 4 : 1>(a, grid, st) : launch_one_n<false, 2, true>(a, grid, st);
    // (VIPNERF_PREC_FP16 / BF16: the two-point-tile kernels of vipnerf_mlp_fwd_pt2.hip; their 16-point NS = 1 forms are retired)
    set_error("mlp_fwd_bf16n: precision %d", precision);
    return VIPNERF_E_ARG;
}

}  // namespace vn
