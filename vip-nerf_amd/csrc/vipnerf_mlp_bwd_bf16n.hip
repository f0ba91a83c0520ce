// Data-gradient pass of the fp16x3 arithmetics, narrow-wave layout (vipnerf_bf16n.h): 16-point waves on v_mfma_f32_16x16x32_f16, two waves per
// SIMD (the exact-fp32 arithmetic: vipnerf_mlp_bwd_f32.hip; the single-MFMA 16-bit modes: vipnerf_mlp_bwd_pt2.hip).  A = W^T from the narrow packed image, B = the NS-part split of the
// current dY; ReLU masks are the 64 bits per lane and layer the narrow forward wrote.
#include "vipnerf_bf16n.h"
#include "vipnerf_mlp.h"
#include "vipnerf_mlp_pt2.h"


namespace vn {

// x where bit (4t + r) of the lane's 64-bit ReLU mask is set, +0 elsewhere: a signed 1-bit field extract gives 0 / -1,
// which ANDs the value (2 VALU; compare + select would be 3)
__device__ __forceinline__ float mask_apply16(float x, unsigned m0, unsigned m1, int t, int r) {
    const int sel = __builtin_amdgcn_sbfe((int)(t < 8 ? m0 : m1), (unsigned)(4 * (t & 7) + r), 1u);
    return __uint_as_float(__float_as_uint(x) & (unsigned)sel);
}

// max over the level of the seeds the data-gradient pass actually starts from -- d(pre-sigmoid rgb, vis, vis2) and
// d(sigma) where sigma > 0, exactly as k_mlp_bwd_bf16n forms them; non-finite entries are skipped -- written to *slot
// as the bit pattern of a non-negative float (an unsigned atomicMax orders those; order-independent, deterministic)
__global__ void k_seed_absmax(MlpBwdArgs a, unsigned *slot) {
    const float *gb = a.bwd;
    const int V = a.src.V;
    float m = 0.f;
    auto take = [&](float v) { v = fabsf(v); if (v < 3.0e38f) m = fmaxf(m, v); };
#pragma unroll 4
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < a.src.P; p += (int64_t)gridDim.x * blockDim.x) {
        for (int c = 0; c < 3; ++c) { const float y = a.rgb[3 * p + c]; take(gb[a.bl.drgb + 3 * p + c] * ((1.f - y) * y)); }
        { const float y = a.vis[p]; take(gb[a.bl.dvis + p] * ((1.f - y) * y)); }
        for (int v = 0; v < V; ++v) { const float y = a.vis2[p * V + v]; take(gb[a.bl.dvis2 + p * V + v] * ((1.f - y) * y)); }
        if (a.sigma[p] > 0.f) take(gb[a.bl.dsig + p]);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float wm[4];                                // one atomic per workgroup: thousands on one address serialise
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (m > 0.f) atomicMax(slot, __float_as_uint(m));
    }
}

// F16: fp16 fragments (VIPNERF_PREC_FP16X3): W^T is packed as 2^8 W^T, every gradient in the workspace is 2^S times its
// true value (grad_scale_from_max), accumulators are taken back by 2^-8
// H16 (with F16): dY of the feature layer and of layers 1..7 -- read back only by the 256x256 weight-gradient GEMMs -- are
// stored as the fp16 parts of the split that is made for the next GEMM anyway: 1 = high parts only (FP16X3H), 2 = high
// and low parts in the fp32 slot (FP16X3, store_pair_split), sent from the next GEMM's weight stages (DEFER); with high
// parts only, dY_5 additionally in fp32 for layer 5's gamma(x) GEMM; dY_0 stays fp32.
int launch_mlp_bwd_f32(const MlpBwdArgs &a, hipStream_t st);
TS_DECL(g_nb_timeline);
#define TSNB(tag) TS_AT(g_nb_timeline, tag)

template <int NS, bool F16, int H16 = 0>
__global__ __launch_bounds__(BnPlan<NS>::WG) void k_mlp_bwd_bf16n(MlpBwdArgs a) {
    typedef BnPlan<NS> PL;
    typedef typename FragOf<F16, false>::type FR;
    constexpr float AU = F16 ? 1.f / F16_WSCALE : 1.f;
    constexpr bool DEFER = H16 != 0 && VN_DEFER_STORES;    // fp16-stored gradients leave from the next GEMM's stages (vipnerf_bf16n.h)
    // H16 == 4 (VN_T16): every gradient the weight-gradient GEMMs read is stored as 16-bit T16 (store_t16) -- dY_0..dY_7, dY_feature, dYv per
    // direction and their sum, the head seeds as one 16-column tile per direction -- and the view hidden's ReLU comes from the 32 bits per
    // lane the forward left (no fp32 copy of dY_5, no read of the view hidden)
    constexpr bool T16 = H16 == 4;
    static_assert(!T16 || NS == 1 || (NS == 2 && F16), "T16 storage: the single-MFMA modes, and FP16X3H (high parts only)");
    constexpr int S_PER_STAGE = 8 / PL::ST_256;
    const float gs = (F16 && a.gmax) ? grad_scale_from_max(*a.gmax) : 1.f;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *res = lds;
    float *stage_buf = lds + PL::R_TOTAL_PAD;
    const float *rf = res + PL::R_F32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, j = lane & 15;
    const int64_t p_raw = (int64_t)blockIdx.x * MLP_PTS_PER_WG + wave * 16 + j;
    const bool valid = p_raw < a.src.P;
    const int64_t p = valid ? p_raw : a.src.P - 1;
    const int64_t grp = (int64_t)blockIdx.x * (MLP_PTS_PER_WG / 16) + wave;   // T16: the wave's 16-point group (valid is wave-uniform)
    const int V = a.src.V;

    TS_INIT();
    TSNB(TS_ENTRY);
    typename StreamOf<PL, PL::SKEW>::type ws;
    ws.start(a.packed + PL::PK_BWD, PL::B_STAGES, stage_buf, lane, wave);
    stream_counted(ws, !T16 || valid);      // a wave beyond P skips its (predicated) T16 stores: its counted waits would not hold
    {
        const float4 *g4 = (const float4 *)(a.packed + PL::PK_RES);
        float4 *l4 = (float4 *)res;
        for (int i = tid; i < PL::R_TOTAL_PAD / 4; i += PL::WG) l4[i] = g4[i];
    }

    const float *gb = a.bwd;
    float dq0[4];
    {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float y = a.rgb[3 * p + c];
            dq0[c] = gb[a.bl.drgb + 3 * p + c] * gs * ((1.f - y) * y);
        }
        const float y = a.vis[p];
        dq0[3] = gb[a.bl.dvis + p] * gs * ((1.f - y) * y);
    }
    const float dsig_raw = a.sigma[p] > 0.f ? gb[a.bl.dsig + p] * gs : 0.f;
    __syncthreads();
    TSNB(TS_RESIDENT);

    // ---------------------------------------------------------------- view branch, per direction
    floatx4 vsum[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) vsum[t] = (floatx4)(0.f);
#pragma unroll 1
    for (int dsel = 0; dsel <= V; ++dsel) {
        float dq[4];
        if (dsel == 0) { dq[0] = dq0[0]; dq[1] = dq0[1]; dq[2] = dq0[2]; dq[3] = dq0[3]; }
        else {
            const float y = a.vis2[p * V + (dsel - 1)];
            dq[0] = dq[1] = dq[2] = 0.f;
            dq[3] = gb[a.bl.dvis2 + p * V + (dsel - 1)] * gs * ((1.f - y) * y);
        }
        if (T16) {
            if (valid) {     // head seeds as a 16-column T16 tile: columns 0..3 d(pre-sigmoid rgb, vis), column 4 d(sigma_raw) (direction 0), zeros
                const float x8[8] = {dq[0], dq[1], dq[2], dq[3], dsel == 0 ? dsig_raw : 0.f, 0.f, 0.f, 0.f};
                FR t8[NS];
                split8<NS>(x8, t8);
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                typedef unsigned u2 __attribute__((ext_vector_type(2)));
                const u4 w = __builtin_bit_cast(u4, t8[0]);
                const u2 mine = {q == 0 ? w[0] : (q == 1 ? w[2] : 0u), q == 0 ? w[1] : (q == 1 ? w[3] : 0u)};
                __builtin_nontemporal_store(mine, (u2 *)((char *)(a.bwd + a.bl.dq[dsel]) + (size_t)grp * 512 + j * 32 + q * 8));
            }
        } else if (valid && q == 0) {
            float *row = a.bwd + a.bl.dq[dsel] + (size_t)p * 8;
            *(float4 *)row = make_float4(dq[0], dq[1], dq[2], dq[3]);
            *(float4 *)(row + 4) = make_float4(dsel == 0 ? dsig_raw : 0.f, 0.f, 0.f, 0.f);
        }
        const unsigned gmask = T16 ? ((const unsigned *)(a.acts + a.al.g[dsel] + (size_t)a.src.P * (WV / 2)))[(size_t)p * 4 + q] : 0u;
        floatx4 dprev = (floatx4)(0.f);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            floatx4 g = (floatx4)(0.f);
            if (!T16) g = load_tile16(a.acts + a.al.g[dsel], p, WV, q, t);
            float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 w4 = *(const float4 *)(rf + PL::N_WOUT + c * WV + 16 * t + 4 * q);
                dg.x = fmaf(w4.x, dq[c], dg.x); dg.y = fmaf(w4.y, dq[c], dg.y);
                dg.z = fmaf(w4.z, dq[c], dg.z); dg.w = fmaf(w4.w, dq[c], dg.w);
            }
            floatx4 d;
            if (T16) {
                d[0] = mask_apply16(dg.x, gmask, 0u, t, 0); d[1] = mask_apply16(dg.y, gmask, 0u, t, 1);
                d[2] = mask_apply16(dg.z, gmask, 0u, t, 2); d[3] = mask_apply16(dg.w, gmask, 0u, t, 3);
                if (t & 1) {
                    FR dh[NS];
                    split_pair<NS>(dprev, d, dh);
                    if (valid && !EXP_NO_EXTRAS) store_t16(a.bwd + a.bl.dyv[dsel], grp, 8, t >> 1, j, q, dh[0]);
                }
                dprev = d;
            } else {
            d[0] = g[0] > 0.f ? dg.x : 0.f;
            d[1] = g[1] > 0.f ? dg.y : 0.f;
            d[2] = g[2] > 0.f ? dg.z : 0.f;
            d[3] = g[3] > 0.f ? dg.w : 0.f;
            if (!EXP_NO_EXTRAS) store_tile16(a.bwd + a.bl.dyv[dsel], p, WV, q, t, d);
            }
            vsum[t] += d;
        }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) if (!EXP_NO_EXTRAS && !T16) store_tile16(a.bwd + a.bl.dyvsum, p, WV, q, t, vsum[t]);

    TSNB(TS_HEAD);
    // ---------------------------------------------------------------- d(feature) = W_vf^T sum_a dYv_a   (K = 128: 4 k-steps)
    FR bin[8][NS];
    floatx4 acc[16];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        split_pair<NS>(vsum[2 * s], vsum[2 * s + 1], bin[s]);
        if (T16 && valid && !EXP_NO_EXTRAS && !VN_WG16_VIEW_FUSED) store_t16(a.bwd + a.bl.dyvsum, grp, 8, s, j, q, bin[s][0]);
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = (floatx4)(0.f);
    stream_begin(ws);
#pragma unroll
    for (int jj = 0; jj < PL::ST_VIEW_B; ++jj) {
        TSNB(TS_PRE);
        const float *st = ws.wait();
        TSNB(TS_POST);
        gemm_stage_bf<16, PL::KSB, NS>(st, lane, acc, bin, PL::KSB * jj, ws);
        TSNB(TS_END);
    }
    // dY of the feature layer: store (fp32, for wgrad) and split
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const floatx4 x0 = acc[2 * s] * AU, x1 = acc[2 * s + 1] * AU;
        if (!H16) {
            store_tile16(a.bwd + a.bl.dyf, p, W, q, 2 * s, x0);
            store_tile16(a.bwd + a.bl.dyf, p, W, q, 2 * s + 1, x1);
        }
        split_pair<NS>(x0, x1, bin[s]);
        if (!DEFER && H16 == 1) store_pair16h(a.bwd + a.bl.dyf, p, W, q, s, bin[s][0]);
        if (!DEFER && H16 == 2) store_pair_split(a.bwd + a.bl.dyf, p, W, q, s, bin[s][0], bin[s][NS > 1 ? 1 : 0]);
        if (!DEFER && H16 == 3) store_pair_f32(a.bwd + a.bl.dyf, p, W, q, s, bin[s][0], bin[s][NS > 1 ? 1 : 0]);
        if (!DEFER && T16 && valid) store_t16(a.bwd + a.bl.dyf, grp, 16, s, j, q, bin[s][0]);
    }

    // ---------------------------------------------------------------- feature layer, then layers 7..1
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
        const int layer = 7 - it;
#if defined(VN_EXP) && VN_EXP == 31
        const uint2 mk = make_uint2(0xffffffffu, 0xfffffffeu + (unsigned)(layer & 1));   // timing experiment only: no ReLU-mask loads
#else
        const uint2 mk = *(const uint2 *)(a.acts + a.al.hm[layer] + ((size_t)p * 4 + q) * 2);
#endif
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = (floatx4)(0.f);
#pragma unroll
        for (int jj = 0; jj < PL::ST_256; ++jj) {
            // younger than the stage's DMA -- first stage of a layer: the epilogue before it (>= 16 gradient tile stores;
            // with deferred stores only those behind the previous GEMM's last stage, and nothing before the first layer);
            // later stages: the deferred stores behind the stage before
            constexpr int DEF_SPK = H16 == 4 ? T16_SPK : 2;          // deferred store instructions per operand k-step
            TSNB(TS_PRE);
            const float *st = jj == 0 ? ws.template wait<DEFER ? DEF_SPK * S_PER_STAGE : 16, DEFER ? 0 : 16>(it == 0)
                                      : ws.template wait<DEFER ? DEF_SPK * S_PER_STAGE : 0>();
            TSNB(TS_POST);
            if (DEFER) {                                 // bin = the fp16 parts of the gradient this GEMM consumes
                DeferredStores<H16, NS, FR, S_PER_STAGE> ds{a.bwd + (it == 0 ? a.bl.dyf : a.bl.dy[layer + 1]), p, q, wave, S_PER_STAGE * jj, bin, grp, j, valid};
                gemm_stage_bf<16, PL::KSB, NS>(st, lane, acc, bin, PL::KSB * jj, ws, ds);
            } else {
                gemm_stage_bf<16, PL::KSB, NS>(st, lane, acc, bin, PL::KSB * jj, ws);
            }
            TSNB(TS_END);
        }
        float *dst = a.bwd + a.bl.dy[layer];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            floatx4 x[2] = {acc[2 * s] * AU, acc[2 * s + 1] * AU};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = 2 * s + u;
                if (it == 0) {                               // h_8 also feeds the sigma head
                    const float4 w4 = *(const float4 *)(rf + PL::N_WSIG + 16 * t + 4 * q);
                    x[u][0] = fmaf(w4.x, dsig_raw, x[u][0]); x[u][1] = fmaf(w4.y, dsig_raw, x[u][1]);
                    x[u][2] = fmaf(w4.z, dsig_raw, x[u][2]); x[u][3] = fmaf(w4.w, dsig_raw, x[u][3]);
                }
#if !(defined(VN_EXP) && VN_EXP == 32)                                   // timing experiment 32: no mask arithmetic
#pragma unroll
                for (int r = 0; r < 4; ++r) x[u][r] = mask_apply16(x[u][r], mk.x, mk.y, t, r);
#endif
                if ((!H16 || it == 7) && !T16 && !((H16 && it == 7) ? EXP_NO_EXTRAS : EXP_NO_STORES)) store_tile16(dst, p, W, q, t, x[u]);
                if (H16 == 1 && layer == SKIP_LAYER && !EXP_NO_EXTRAS) store_tile16(a.bwd + a.bl.dy5f, p, W, q, t, x[u]);   // pre-split storage keeps both parts: no copy
            }
            if (it < 7 || T16) {
                split_pair<NS>(x[0], x[1], bin[s]);
                if (T16 && (it == 7 || !DEFER) && valid && !EXP_NO_EXTRAS) store_t16(dst, grp, 16, s, j, q, bin[s][0]);   // dY_0 (the others: deferred)
                if (!DEFER && H16 == 1) store_pair16h(dst, p, W, q, s, bin[s][0]);
                if (!DEFER && H16 == 2) store_pair_split(dst, p, W, q, s, bin[s][0], bin[s][NS > 1 ? 1 : 0]);
                if (!DEFER && H16 == 3) store_pair_f32(dst, p, W, q, s, bin[s][0], bin[s][NS > 1 ? 1 : 0]);
            }
        }
    }
    stream_end(ws);
    TSNB(TS_LAST);
}

#if defined(VN_EXP) && VN_EXP == 50
extern "C" int vipnerf_exp_timeline_nb(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nb_timeline), sizeof(unsigned long long) * (n < 2048 ? n : 2048));
}
#endif

template <int NS, bool F16 = false, int H16 = 0>
static int launch_one_bwd_n(const MlpBwdArgs &a, unsigned grid, hipStream_t st) {
    const size_t lds = (size_t)BnPlan<NS>::LDS_F * sizeof(float);
    VN_HIP(hipFuncSetAttribute((const void *)k_mlp_bwd_bf16n<NS, F16, H16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_mlp_bwd_bf16n<NS, F16, H16>), dim3(grid), dim3(BnPlan<NS>::WG), lds, st, a);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

int launch_mlp_bwd_bf16n(const MlpBwdArgs &a, int precision, hipStream_t st) {
    if (a.src.P <= 0) return VIPNERF_OK;
    const unsigned grid = (unsigned)((a.src.P + MLP_PTS_PER_WG - 1) / MLP_PTS_PER_WG);
    if (precision == 0) return launch_mlp_bwd_f32(a, st);      // the exact-fp32 kernel of vipnerf_mlp_bwd_f32.hip
    // (precisions 1 / 2, the split-bf16 arithmetics bf16x3 / bf16x6, were retired with ABI 5)
    if ((precision == 5 || precision == 6) && !single_mfma_t16(precision)) {
        set_error("this library was built without T16 storage (VN_T16 / VN_BF16_H16 = 0): no single-MFMA 16-bit kernels"); return VIPNERF_E_UNSUPPORTED; }
    if (precision == 6) return launch_mlp_bwd_pt2(a, precision, st);     // two point tiles per wave (the 16-point NS = 1 form is retired)
    if (precision == 3 || precision == 4 || precision == 5) {
        // the level's largest seed first (one pass over 5+V floats per point)
        unsigned *slot = (unsigned *)(a.bwd + a.bl.gmax);
        VN_HIP(hipMemsetAsync(slot, 0, sizeof(unsigned), st));
        // four points per thread (independent loads in flight), a quarter of the atomics: 28 -> ~15 us per level
        hipLaunchKernelGGL(k_seed_absmax, dim3((unsigned)((a.src.P + 1023) / 1024)), dim3(256), 0, st, a, slot);
        VN_HIP(hipGetLastError());
        MlpBwdArgs b = a;
        b.gmax = slot;
        if (precision == 5) return launch_mlp_bwd_pt2(b, precision, st);
        return precision == 4 ? launch_one_bwd_n<2, true, VN_T16 ? 4 : 1>(b, grid, st) : launch_one_bwd_n<2, true, VN_F16_PRESPLIT ? 2 : 0>(b, grid, st);
    }
    set_error("mlp_bwd_bf16n: precision %d", precision);
    return VIPNERF_E_ARG;
}

}  // namespace vn
