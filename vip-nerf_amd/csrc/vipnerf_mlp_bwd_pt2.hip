// Data-gradient pass of the single-MFMA 16-bit modes (VIPNERF_PREC_FP16 / BF16) with TWO point tiles per wave -- the counterpart of
// vipnerf_mlp_fwd_pt2.hip: 8 waves x 32 points per workgroup, every A fragment (W^T from the narrow single-part image) read from LDS once
// for two 16x16x32 MFMAs.  Same algorithm, stage order, T16 gradient storage and ReLU bits as k_mlp_bwd_bf16n<1, ., 4> (which stays behind
// -DVN_PT2=0): autograd of reference src/models/VipNeRF01.py:509-596 w.r.t. the layer inputs.
#include "vipnerf_bf16n.h"
#include "vipnerf_mlp.h"
#include "vipnerf_mlp_pt2.h"

namespace vn {

// x where bit (4 t + r) of the 64-bit ReLU mask (m0: tiles 0..7, m1: tiles 8..15) is set, +0 elsewhere
__device__ __forceinline__ float mask_apply_pt2(float x, unsigned m0, unsigned m1, int t, int r) {
    // inline asm: the compiler turns sbfe + and back into v_and (one bit) + v_cmp + v_cndmask, three instructions and a VCC hazard
    int sel;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(sel) : "v"(t < 8 ? m0 : m1), "s"(4 * (t & 7) + r));
    return __uint_as_float(__float_as_uint(x) & (unsigned)sel);
}

TS_DECL(g_pt2_timeline_bwd);
#define TSB(tag) TS_AT(g_pt2_timeline_bwd, tag)

template <bool F16>
__global__ __launch_bounds__(BnPlan<1>::WG) void k_mlp_bwd_pt2(MlpBwdArgs a) {
    typedef BnPlan<1> PL;
    typedef typename FragOf<F16>::type FR;
    typedef BOp<FR, 2> BT;
    typedef AccN<2> AT;
    constexpr int NS = 1;
    constexpr float AU = F16 ? 1.f / F16_WSCALE : 1.f;
    constexpr int S_PER_STAGE = 8 / PL::ST_256;
    const float gs = (F16 && a.gmax) ? grad_scale_from_max(*a.gmax) : 1.f;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *res = lds;
    float *stage_buf = lds + PL::R_TOTAL_PAD;
    const float *rf = res + PL::R_F32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, j = lane & 15;
    const int V = a.src.V;
    int64_t p[2], grp[2];
    bool valid[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int64_t p_raw = (int64_t)blockIdx.x * PT2_PTS_PER_WG + wave * 32 + pt * 16 + j;
        valid[pt] = p_raw < a.src.P;
        p[pt] = valid[pt] ? p_raw : a.src.P - 1;
        grp[pt] = (int64_t)blockIdx.x * (PT2_PTS_PER_WG / 16) + wave * 2 + pt;
    }

    TS_INIT();
    TSB(TS_ENTRY);
    // Every scalar input of the view-branch head below -- outputs, their upstream gradients, the view layer's ReLU bits; both point tiles,
    // directions 0 and 1 -- loaded FIRST: under the step's store traffic a global load takes thousands of cycles to come back, and the head
    // used to pay that once per point tile and direction with the MFMA pipe idle (22 % of the workgroup's time, profiles/r04_ablation_pt2.md
    // 5).  Here they land behind the weight image's resident block.
    const float *gb = a.bwd;
    float h_y[2][4], h_dy[2][4], h_sig[2], h_dsg[2], h_y2[2] = {0.f, 0.f}, h_dy2[2] = {0.f, 0.f};
    unsigned h_gm[2][2] = {{0u, 0u}, {0u, 0u}};
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int64_t pp = p[pt];
#pragma unroll
        for (int c = 0; c < 3; ++c) { h_y[pt][c] = a.rgb[3 * pp + c]; h_dy[pt][c] = gb[a.bl.drgb + 3 * pp + c]; }
        h_y[pt][3] = a.vis[pp]; h_dy[pt][3] = gb[a.bl.dvis + pp];
        h_sig[pt] = a.sigma[pp]; h_dsg[pt] = gb[a.bl.dsig + pp];
        h_gm[pt][0] = ((const unsigned *)(a.acts + a.al.g[0] + (size_t)a.src.P * (WV / 2)))[(size_t)pp * 4 + q];
        if (V >= 1) {
            h_y2[pt] = a.vis2[pp * V]; h_dy2[pt] = gb[a.bl.dvis2 + pp * V];
            h_gm[pt][1] = ((const unsigned *)(a.acts + a.al.g[1] + (size_t)a.src.P * (WV / 2)))[(size_t)pp * 4 + q];
        }
    }
    const int store_phase = pt2_store_phase(wave);
    typename StreamOfAll<PL>::type ws;
    ws.start(a.packed + PL::PK_BWD, PL::B_STAGES, stage_buf, lane, wave);
    ws.counted = valid[0] && valid[1];       // the counted waits assume the stores of BOTH point tiles
    {
        const float4 *g4 = (const float4 *)(a.packed + PL::PK_RES);
        float4 *l4 = (float4 *)res;
        for (int i = tid; i < PL::R_TOTAL_PAD / 4; i += PL::WG) l4[i] = g4[i];
    }
    __syncthreads();
    TSB(TS_RESIDENT);

    BT bin[8][NS];
    float dsig_raw[2];

    // ---------------------------------------------------------------- view branch, per point tile and direction
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int64_t pp = p[pt];
        float dq0[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) dq0[c] = h_dy[pt][c] * gs * ((1.f - h_y[pt][c]) * h_y[pt][c]);
        dsig_raw[pt] = h_sig[pt] > 0.f ? h_dsg[pt] * gs : 0.f;
        floatx4 vsum[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) vsum[t] = (floatx4)(0.f);
#pragma unroll 1
        for (int dsel = 0; dsel <= V; ++dsel) {
            float dq[4];
            unsigned gmask;
            if (dsel == 0) { dq[0] = dq0[0]; dq[1] = dq0[1]; dq[2] = dq0[2]; dq[3] = dq0[3]; gmask = h_gm[pt][0]; }
            else {
                float y = h_y2[pt], dy = h_dy2[pt];
                gmask = h_gm[pt][1];
                if (dsel >= 2) {             // directions beyond the first secondary one: loaded here
                    y = a.vis2[pp * V + (dsel - 1)]; dy = gb[a.bl.dvis2 + pp * V + (dsel - 1)];
                    gmask = ((const unsigned *)(a.acts + a.al.g[dsel] + (size_t)a.src.P * (WV / 2)))[(size_t)pp * 4 + q];
                }
                dq[0] = dq[1] = dq[2] = 0.f;
                dq[3] = dy * gs * ((1.f - y) * y);
            }
            if (valid[pt] && !EXP_NO_EXTRAS) {     // head seeds as a 16-column T16 tile: columns 0..3 d(pre-sigmoid rgb, vis), column 4 d(sigma_raw) (direction 0), zeros
                const float x8[8] = {dq[0], dq[1], dq[2], dq[3], dsel == 0 ? dsig_raw[pt] : 0.f, 0.f, 0.f, 0.f};
                FR t8[NS];
                split8<NS>(x8, t8);
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                typedef unsigned u2 __attribute__((ext_vector_type(2)));
                const u4 w = __builtin_bit_cast(u4, t8[0]);
                const u2 mine = {q == 0 ? w[0] : (q == 1 ? w[2] : 0u), q == 0 ? w[1] : (q == 1 ? w[3] : 0u)};
                __builtin_nontemporal_store(mine, (u2 *)((char *)(a.bwd + a.bl.dq[dsel]) + (size_t)grp[pt] * 512 + j * 32 + q * 8));
            }
            floatx4 dprev = (floatx4)(0.f);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c < 3 && dsel != 0) continue;      // a secondary direction seeds the visibility output only (its other dq are +0: the same bits)
                    const float4 w4 = *(const float4 *)(rf + PL::N_WOUT + c * WV + 16 * t + 4 * q);
                    dg.x = fmaf(w4.x, dq[c], dg.x); dg.y = fmaf(w4.y, dq[c], dg.y);
                    dg.z = fmaf(w4.z, dq[c], dg.z); dg.w = fmaf(w4.w, dq[c], dg.w);
                }
                floatx4 d;
                d[0] = mask_apply_pt2(dg.x, gmask, 0u, t, 0); d[1] = mask_apply_pt2(dg.y, gmask, 0u, t, 1);
                d[2] = mask_apply_pt2(dg.z, gmask, 0u, t, 2); d[3] = mask_apply_pt2(dg.w, gmask, 0u, t, 3);
                if (t & 1) {
                    FR dh[NS];
                    split_pair<NS>(dprev, d, dh);
                    if (valid[pt] && !EXP_NO_EXTRAS) store_t16(a.bwd + a.bl.dyv[dsel], grp[pt], 8, t >> 1, j, q, dh[0]);
                }
                dprev = d;
                vsum[t] += d;
            }
        }
        // sum over directions: the A operand of the view layer's feature-column weight gradient, and this pass's first B operand
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            FR t1[NS];
            split_pair<NS>(vsum[2 * s], vsum[2 * s + 1], t1);
            bin[s][0].v[pt] = t1[0];
            // (sent from inside the feature GEMM's stage instead -- DeferredT16 -- the data-gradient kernel measured 1.93 instead of 1.61 ms per step)
            if (valid[pt] && !EXP_NO_EXTRAS && !VN_WG16_VIEW_FUSED) store_t16(a.bwd + a.bl.dyvsum, grp[pt], 8, s, j, q, t1[0]);
        }
    }

    TSB(TS_HEAD);                            // heads and view hidden layer of both point tiles, every direction
    // ---------------------------------------------------------------- d(feature) = W_vf^T sum_a dYv_a   (K = 128: 4 k-steps)
    AT acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) { acc[t].v[0] = (floatx4)(0.f); acc[t].v[1] = (floatx4)(0.f); }
#pragma unroll
    for (int jj = 0; jj < PL::ST_VIEW_B; ++jj) {
        TSB(TS_PRE);
        const float *st = ws.wait();
        TSB(TS_POST);
        gemm_stage_bf<16, PL::KSB, NS>(st, lane, acc, bin, PL::KSB * jj, ws);
        TSB(TS_END);
    }
#pragma unroll
    for (int pt = 0; pt < 2; ++pt)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            FR t1[NS];
            split_pair<NS>(acc[2 * s].v[pt] * AU, acc[2 * s + 1].v[pt] * AU, t1);
            bin[s][0].v[pt] = t1[0];             // dY of the feature layer: stored from the next GEMM's stages
        }

    // ---------------------------------------------------------------- feature layer, then layers 7..1
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
        const int layer = 7 - it;
        uint2 mk[2];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) mk[pt] = *(const uint2 *)(a.acts + a.al.hm[layer] + ((size_t)p[pt] * 4 + q) * 2);
#pragma unroll
        for (int t = 0; t < 16; ++t) { acc[t].v[0] = (floatx4)(0.f); acc[t].v[1] = (floatx4)(0.f); }
#pragma unroll
        for (int jj = 0; jj < PL::ST_256; ++jj) {
            // younger than the stage's DMA: the deferred stores behind the stage before (nothing reliable before the first layer)
            TSB(TS_PRE);
            const float *st = jj == 0 ? ws.template wait<2 * T16_SPK * S_PER_STAGE, 0>(it == 0) : ws.template wait<2 * T16_SPK * S_PER_STAGE>();
            TSB(TS_POST);
            DeferredT16<FR, S_PER_STAGE> ds{a.bwd + (it == 0 ? a.bl.dyf : a.bl.dy[layer + 1]), {grp[0], grp[1]}, {valid[0], valid[1]}, j, q, S_PER_STAGE * jj, bin, store_phase};
            gemm_stage_bf<16, PL::KSB, NS>(st, lane, acc, bin, PL::KSB * jj, ws, ds);
            TSB(TS_END);
        }
        float *dst = a.bwd + a.bl.dy[layer];
        // ReLU bits applied AFTER the conversion, two gradients per instruction (vipnerf_mlp_pt2.h mask_pk16): 2.25 VALU instructions
        // per value with the conversion where the select on fp32 values was 2.5 .. 3 (the epilogue of a SIMD's later wave runs with the
        // MFMA pipe idle: its instruction count is kernel time, profiles/r04_ablation_pt2.md 5)
        unsigned one2 = 0x00010001u;
        asm volatile("" : "+s"(one2));
#if defined(VN_EXP) && VN_EXP == 50
        asm volatile("" : "+v"(mk[0].x), "+v"(mk[1].x), "+v"(acc[0].v[0]), "+v"(acc[15].v[1]));     // the ReLU bits and the first / last accumulators are there
#endif
        TSB(TS_EPI_A);
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const unsigned ab[4] = {unfold_pk_bits(mk[pt].x & 0xffffu), unfold_pk_bits(mk[pt].x >> 16),
                                    unfold_pk_bits(mk[pt].y & 0xffffu), unfold_pk_bits(mk[pt].y >> 16)};
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                floatx4 x[2] = {acc[2 * s].v[pt] * AU, acc[2 * s + 1].v[pt] * AU};
                if (it == 0) {                                   // h_8 also feeds the sigma head
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const float4 w4 = *(const float4 *)(rf + PL::N_WSIG + 16 * (2 * s + u) + 4 * q);
                        x[u][0] = fmaf(w4.x, dsig_raw[pt], x[u][0]); x[u][1] = fmaf(w4.y, dsig_raw[pt], x[u][1]);
                        x[u][2] = fmaf(w4.z, dsig_raw[pt], x[u][2]); x[u][3] = fmaf(w4.w, dsig_raw[pt], x[u][3]);
                    }
                }
                FR t1[NS];
                split_pair<NS>(x[0], x[1], t1);
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                const u4 w4 = __builtin_bit_cast(u4, t1[0]);
                u4 r4;
#pragma unroll
                for (int i = 0; i < 4; ++i) r4[i] = mask_pk16(w4[i], ab[s >> 1], 4 * (s & 1) + i, one2);
                t1[0] = __builtin_bit_cast(FR, r4);
                bin[s][0].v[pt] = t1[0];
                if (it == 7 && valid[pt] && !EXP_NO_EXTRAS) store_t16(dst, grp[pt], 16, s, j, q, t1[0]);     // dY_0 (the others leave from the next GEMM's stages)
            }
        }
#if defined(VN_EXP) && VN_EXP == 50
        asm volatile("" : "+v"(bin[7][0].v[1]));
#endif
        TSB(TS_EPI_B);
    }
    TSB(TS_LAST);
    if (EXP_NO_EXTRAS) {          // timing-only builds without stores: keep the whole chain alive (a store that never happens)
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        unsigned x = 0u;
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) { const u4 w = __builtin_bit_cast(u4, bin[s][0].v[pt]); x ^= w[0] ^ w[1] ^ w[2] ^ w[3]; }
        if (x == 0x7fc12345u) a.bwd[a.bl.dy[0]] = 1.f;
    }
}

#if defined(VN_EXP) && VN_EXP == 50
extern "C" int vipnerf_exp_timeline_bwd(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pt2_timeline_bwd), sizeof(unsigned long long) * (n < 2048 ? n : 2048));
}
#endif

template <bool F16>
static int launch_bwd_pt2(const MlpBwdArgs &a, hipStream_t st) {
    const unsigned grid = (unsigned)((a.src.P + PT2_PTS_PER_WG - 1) / PT2_PTS_PER_WG);
    const size_t lds = (size_t)BnPlan<1>::LDS_F * sizeof(float);
    VN_HIP(hipFuncSetAttribute((const void *)k_mlp_bwd_pt2<F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_mlp_bwd_pt2<F16>), dim3(grid), dim3(BnPlan<1>::WG), lds, st, a);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

// a.packed: the narrow single-part image of the precision; FP16: a.gmax = the level's largest seed (k_seed_absmax ran before)
int launch_mlp_bwd_pt2(const MlpBwdArgs &a, int precision, hipStream_t st) {
    if (a.src.P <= 0) return VIPNERF_OK;
    if (a.src.P % 16) { set_error("mlp_bwd: the 16-bit training kernels need a multiple of 16 points (got %lld)", (long long)a.src.P); return VIPNERF_E_UNSUPPORTED; }
    if (precision == VIPNERF_PREC_FP16) return launch_bwd_pt2<true>(a, st);
    if (precision == VIPNERF_PREC_BF16) return launch_bwd_pt2<false>(a, st);
    set_error("mlp_bwd_pt2: precision %d", precision);
    return VIPNERF_E_ARG;
}

}  // namespace vn
