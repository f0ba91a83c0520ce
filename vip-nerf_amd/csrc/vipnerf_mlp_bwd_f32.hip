// Data-gradient pass of the MLP in EXACT fp32 (the headline arithmetic; autograd of reference src/models/VipNeRF01.py:537-596):
// v_mfma_f32_16x16x4_f32 on the narrow layout (vipnerf_bf16n.h: 16 points per wave, two waves per SIMD, BnPlan<2>'s 64 KiB weight stages,
// f32q fragments), written for this one arithmetic instead of as an instantiation of k_mlp_bwd_bf16n's seven-way template.
//
// What is different from that instantiation (round 4's per-wave timeline, profiles/r04_ablation_pt2.md section 9: 702.7k cycles per
// workgroup for 557k of MFMA pipe time -- head 38k, resident block 13k, layer epilogues 34k on the critical wave, waits 40k):
//   * the view-branch head reads the view hidden layer's ReLU as 32 BITS per lane and direction (written by the fp32 training forward
//     next to the activations) instead of its 128 fp32 activations, and every input of the head -- the seeds of all directions, the
//     bits -- is requested before the resident block is copied: ONE exposed load latency instead of one per direction;
//   * a layer's epilogue (ReLU bits applied to the 64 accumulators that become the next GEMM's B operand) is PIPELINED into the next
//     GEMM: only the two operand k-steps the first weight stage consumes are converted ahead of it, the other six inside stages 0..2 (behind
//     MFMA group CONV_GROUP of each), where a wave64 VALU instruction issues in the 28 cycles a 32-cycle fp32 MFMA leaves free -- the younger
//     wave of a SIMD no longer runs a whole epilogue with the pipe idle;
//   * no per-tile branches on the layer index: the sigma head's column enters as the feature GEMM's accumulator INIT (one wave-uniform
//     branch per layer), dY_0 is stored after the loop, and the operand of the first trunk GEMM (dY_feature: no ReLU) goes through the
//     same code with an all-ones mask.
// Stores: as before -- dY_feature, dY_7..dY_1 leave as plain fp32 [P][256] tiles from inside the GEMM that consumes them (DeferredStores'
// placement: waves 0..3 behind group VN_STORE_GROUP_A, waves 4..7 behind VN_STORE_GROUP_B), dY_0 from the tail.
#include "vipnerf_bf16n.h"
#include "vipnerf_mlp.h"
#include "vipnerf_mlp_pt2.h"

namespace vn {

// build switch VN_F32B_CONV_GROUP (default 2, vipnerf_knobs.h): the MFMA group of a stage behind which the next stage's operand k-steps are converted
TS_DECL(g_f32b_timeline);
#define TSF(tag) TS_AT(g_f32b_timeline, tag)

typedef BnPlan<2> PLF;

// x where bit `bit` of `word` is set, +0 elsewhere: v_bfe_i32 (0 / -1) + v_and.  The empty asm keeps the compiler from turning the pair
// back into v_and + v_cmp + v_cndmask (3 VALU and two VCC wait states per value: what the template instantiation compiled to).
__device__ __forceinline__ float keep_if_bit(float x, unsigned word, int bit) {
    int sel = __builtin_amdgcn_sbfe((int)word, (unsigned)bit, 1u);
    asm("" : "+v"(sel));
    return __uint_as_float(__float_as_uint(x) & (unsigned)sel);
}

// operand k-step s of the next GEMM <- raw accumulator tiles 2s, 2s + 1 with the producing layer's ReLU bits (bit 4 (t & 7) + r of word t >> 3)
__device__ __forceinline__ void conv_kstep(const floatx4 *xr, f32q (*bin)[2], int s, unsigned m0, unsigned m1) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int t = 2 * s + u;
#pragma unroll
        for (int r = 0; r < 4; ++r) bin[s][u].v[r] = keep_if_bit(xr[t][r], t < 8 ? m0 : m1, 4 * (t & 7) + r);
    }
}

// What a weight stage does besides its MFMAs: the fp32 stores of the operand k-steps it consumes (s0, s0 + 1: tiles 2 s0 .. 2 s0 + 3), and the
// conversion of the k-steps the NEXT stage consumes (s0 + 2, s0 + 3) from the previous GEMM's raw accumulators.
template <bool CONV>
struct F32BwdMid {
    float *dst; int64_t p; int q, wave, s0;
    f32q (*bin)[2];
    const floatx4 *xr;
    unsigned m0, m1;
    template <int g, int NG> static constexpr bool active() { return g == VN_STORE_GROUP_A || g == VN_STORE_GROUP_B || (CONV && g == VN_F32B_CONV_GROUP); }
    template <int g, int NG>
    __device__ __forceinline__ void at() const {
        if (CONV && g == VN_F32B_CONV_GROUP) {
            conv_kstep(xr, bin, s0 + 2, m0, m1);
            conv_kstep(xr, bin, s0 + 3, m0, m1);
        }
        if ((g == VN_STORE_GROUP_A || g == VN_STORE_GROUP_B) && (g == VN_STORE_GROUP_A) == (wave < 4) && !EXP_NO_STORES) {
#pragma unroll
            for (int s = s0; s < s0 + 2; ++s) {
                store_tile16(dst, p, W, q, 2 * s, bin[s][0].v);
                store_tile16(dst, p, W, q, 2 * s + 1, bin[s][1].v);
            }
        }
    }
};
static_assert(VN_F32B_CONV_GROUP != VN_STORE_GROUP_A && VN_F32B_CONV_GROUP != VN_STORE_GROUP_B && VN_F32B_CONV_GROUP >= 1 && VN_F32B_CONV_GROUP < 16, "one hook per group");

__global__ __launch_bounds__(PLF::WG) void k_mlp_bwd_f32(MlpBwdArgs a) {
    typedef PLF PL;
    typedef f32q FR;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *res = lds;
    float *stage_buf = lds + PL::R_TOTAL_PAD;
    const float *rf = res + PL::R_F32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, j = lane & 15;
    const int V = a.src.V;
    // PERSISTENT workgroups: the grid is one workgroup per CU (or fewer), workgroup b takes the 128-point tiles b, b + grid, b + 2 grid, ...: the
    // LDS-resident block is copied once, and the weight stream runs on across tiles -- the next tile's first stages are requested during the
    // current tile's last ones instead of behind a kernel-start latency
    const int64_t n_tiles = (a.src.P + MLP_PTS_PER_WG - 1) / MLP_PTS_PER_WG;
    const int my_tiles = (int)((n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x);

    TS_INIT();
    TSF(TS_ENTRY);
    typename StreamOlder<PL>::type ws;
    ws.start(a.packed + PL::PK_BWD, PL::B_STAGES, stage_buf, lane, wave, my_tiles);
    {
        const float4 *g4 = (const float4 *)(a.packed + PL::PK_RES);
        float4 *l4 = (float4 *)res;
        for (int i = tid; i < PL::R_TOTAL_PAD / 4; i += PL::WG) l4[i] = g4[i];
    }
#pragma unroll 1
    for (int tile_k = 0; tile_k < my_tiles; ++tile_k) {
    const int64_t p_raw = ((int64_t)blockIdx.x + (int64_t)tile_k * gridDim.x) * MLP_PTS_PER_WG + wave * 16 + j;
    const bool valid = p_raw < a.src.P;
    const int64_t p = valid ? p_raw : a.src.P - 1;       // a lane beyond P works on point P - 1 and writes the bytes its owner writes (vipnerf_bf16n.h)

    // ---------------------------------------------------------------- every global input of the head, requested at once
    const float *gb = a.bwd;
    float y_rgb[3], g_rgb[3], y2[VIPNERF_MAX_SEC], g2[VIPNERF_MAX_SEC];
    unsigned gm[1 + VIPNERF_MAX_SEC];
#pragma unroll
    for (int c = 0; c < 3; ++c) { y_rgb[c] = a.rgb[3 * p + c]; g_rgb[c] = gb[a.bl.drgb + 3 * p + c]; }
    const float y_vis = a.vis[p], g_vis = gb[a.bl.dvis + p];
    const float sig = a.sigma[p], g_sig = gb[a.bl.dsig + p];
#pragma unroll
    for (int v = 0; v < VIPNERF_MAX_SEC; ++v) {
        y2[v] = 0.f; g2[v] = 0.f;
        if (v < V) { y2[v] = a.vis2[p * V + v]; g2[v] = gb[a.bl.dvis2 + p * V + v]; }
    }
#pragma unroll
    for (int d = 0; d <= VIPNERF_MAX_SEC; ++d) {
        gm[d] = 0u;
        if (d <= V) gm[d] = ((const unsigned *)(a.acts + a.al.gm[d]))[(size_t)p * 4 + q];
    }
    float dq0[4];
#pragma unroll
    for (int c = 0; c < 3; ++c) dq0[c] = g_rgb[c] * ((1.f - y_rgb[c]) * y_rgb[c]);
    dq0[3] = g_vis * ((1.f - y_vis) * y_vis);
    const float dsig_raw = sig > 0.f ? g_sig : 0.f;
    if (tile_k == 0) __syncthreads();                   // (the resident block; later tiles: the stage barriers order everything)
    TSF(TS_RESIDENT);

    // ---------------------------------------------------------------- view branch, per direction: dYv_a = (W_o^T dq_a) . relu'(view hidden_a)
    floatx4 vsum[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) vsum[t] = (floatx4)(0.f);
#pragma unroll
    for (int dsel = 0; dsel <= VIPNERF_MAX_SEC; ++dsel) {
        if (dsel <= V) {
            const float dq3 = dsel == 0 ? dq0[3] : g2[dsel > 0 ? dsel - 1 : 0] * ((1.f - y2[dsel > 0 ? dsel - 1 : 0]) * y2[dsel > 0 ? dsel - 1 : 0]);
            if (valid && q == 0) {       // the head's seeds as an 8-column row for its weight-gradient GEMM
                float *row = a.bwd + a.bl.dq[dsel] + (size_t)p * 8;
                *(float4 *)row = dsel == 0 ? make_float4(dq0[0], dq0[1], dq0[2], dq3) : make_float4(0.f, 0.f, 0.f, dq3);
                *(float4 *)(row + 4) = make_float4(dsel == 0 ? dsig_raw : 0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                float4 dg = make_float4(0.f, 0.f, 0.f, 0.f);
                if (dsel == 0) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float4 w4 = *(const float4 *)(rf + PL::N_WOUT + c * WV + 16 * t + 4 * q);
                        dg.x = fmaf(w4.x, dq0[c], dg.x); dg.y = fmaf(w4.y, dq0[c], dg.y);
                        dg.z = fmaf(w4.z, dq0[c], dg.z); dg.w = fmaf(w4.w, dq0[c], dg.w);
                    }
                }
                {   // (secondary directions carry the visibility column only: rgb seeds are zero)
                    const float4 w4 = *(const float4 *)(rf + PL::N_WOUT + 3 * WV + 16 * t + 4 * q);
                    dg.x = fmaf(w4.x, dq3, dg.x); dg.y = fmaf(w4.y, dq3, dg.y);
                    dg.z = fmaf(w4.z, dq3, dg.z); dg.w = fmaf(w4.w, dq3, dg.w);
                }
                floatx4 d;
                d[0] = keep_if_bit(dg.x, gm[dsel], 4 * t + 0); d[1] = keep_if_bit(dg.y, gm[dsel], 4 * t + 1);
                d[2] = keep_if_bit(dg.z, gm[dsel], 4 * t + 2); d[3] = keep_if_bit(dg.w, gm[dsel], 4 * t + 3);
                if (!EXP_NO_EXTRAS) store_tile16(a.bwd + a.bl.dyv[dsel], p, WV, q, t, d);
                vsum[t] += d;
            }
        }
    }
    // (sum_a dYv_a is formed again by the view layer's weight-gradient kernel while it stages dYv_0..V: k_wgrad_view; stored only without it)
#pragma unroll
    for (int t = 0; t < 8; ++t) if (!EXP_NO_EXTRAS && !VN_WGRAD_VIEW_FUSED) store_tile16(a.bwd + a.bl.dyvsum, p, WV, q, t, vsum[t]);
    TSF(TS_HEAD);

    // ---------------------------------------------------------------- d(feature) = W_vf^T sum_a dYv_a   (K = 128: 4 k-steps, 2 stages)
    FR bin[8][2];
    floatx4 acc[16], xr[16];
#pragma unroll
    for (int s = 0; s < 4; ++s) { bin[s][0].v = vsum[2 * s]; bin[s][1].v = vsum[2 * s + 1]; }
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = (floatx4)(0.f);
#pragma unroll
    for (int jj = 0; jj < PL::ST_VIEW_B; ++jj) {
        TSF(TS_PRE);
        const float *st = ws.wait();
        TSF(TS_POST);
        gemm_stage_bf<16, PL::KSB, 2>(st, lane, acc, bin, PL::KSB * jj, ws);
        TSF(TS_END);
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) xr[t] = acc[t];

    // ---------------------------------------------------------------- the feature layer (it = 0), then layers 7..1: 8 GEMMs of 4 stages
    unsigned mi0 = 0xffffffffu, mi1 = 0xffffffffu;        // ReLU bits of the operand entering the GEMM (dY_feature: none)
    float *dst_in = a.bwd + a.bl.dyf;                     // where that operand is stored (from inside the GEMM)
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
        const int layer = 7 - it;
        // this GEMM's output is masked with layer `layer`'s ReLU bits: requested now, used from the next GEMM's first conversion on
        const uint2 mk = *(const uint2 *)(a.acts + a.al.hm[layer] + ((size_t)p * 4 + q) * 2);
        conv_kstep(xr, bin, 0, mi0, mi1);
        conv_kstep(xr, bin, 1, mi0, mi1);
        if (it == 0) {                                    // h_8 also feeds the sigma head: its column is where the accumulators start
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const float4 w4 = *(const float4 *)(rf + PL::N_WSIG + 16 * t + 4 * q);
                acc[t][0] = w4.x * dsig_raw; acc[t][1] = w4.y * dsig_raw; acc[t][2] = w4.z * dsig_raw; acc[t][3] = w4.w * dsig_raw;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[t] = (floatx4)(0.f);
        }
#pragma unroll
        for (int jj = 0; jj < PL::ST_256; ++jj) {
            // counted wait: since it issued this stage's DMA (behind group 0 of the stage before) the issuing wave has executed that stage's 4
            // tile stores -- and, for the first stage of a GEMM, the ReLU-bit load above (G0's stages store nothing: 1 for it == 0)
            TSF(TS_PRE);
            const float *st = jj == 0 ? ws.template wait<5, 1>(it == 0) : ws.template wait<4>();
            TSF(TS_POST);
            if (jj + 1 < PL::ST_256) {
                F32BwdMid<true> mid{dst_in, p, q, wave, 2 * jj, bin, xr, mi0, mi1};
                gemm_stage_bf<16, PL::KSB, 2>(st, lane, acc, bin, PL::KSB * jj, ws, mid);
            } else {
                F32BwdMid<false> mid{dst_in, p, q, wave, 2 * jj, bin, xr, mi0, mi1};
                gemm_stage_bf<16, PL::KSB, 2>(st, lane, acc, bin, PL::KSB * jj, ws, mid);
            }
            TSF(TS_END);
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) xr[t] = acc[t];
        mi0 = mk.x; mi1 = mk.y;
        dst_in = a.bwd + a.bl.dy[layer];
    }
    // ---------------------------------------------------------------- dY_0: layer 0's ReLU bits, stored from here (no GEMM consumes it)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        conv_kstep(xr, bin, s, mi0, mi1);
        if (!EXP_NO_EXTRAS) {
            store_tile16(dst_in, p, W, q, 2 * s, bin[s][0].v);
            store_tile16(dst_in, p, W, q, 2 * s + 1, bin[s][1].v);
        }
    }
    TSF(TS_LAST);
    }   // tiles
    stream_end(ws);
}

#if defined(VN_EXP) && VN_EXP == 50
extern "C" int vipnerf_exp_timeline_f32b(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_f32b_timeline), sizeof(unsigned long long) * (n < 2048 ? n : 2048));
}
#endif

int launch_mlp_bwd_f32(const MlpBwdArgs &a, hipStream_t st) {
    if (a.src.P <= 0) return VIPNERF_OK;
    const int64_t tiles = (a.src.P + MLP_PTS_PER_WG - 1) / MLP_PTS_PER_WG;
    const unsigned grid = (unsigned)(VN_F32_PERSISTENT && tiles > persistent_grid() ? persistent_grid() : tiles);   // one workgroup per CU (160 KB of LDS each)
    const size_t lds = (size_t)PLF::LDS_F * sizeof(float);
    VN_HIP(hipFuncSetAttribute((const void *)k_mlp_bwd_f32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_mlp_bwd_f32, dim3(grid), dim3(PLF::WG), lds, st, a);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

}  // namespace vn
