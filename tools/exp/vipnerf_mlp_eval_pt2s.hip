// MLP.forward (reference src/models/VipNeRF01.py:509-596), EVAL form of the single-MFMA 16-bit modes, with the two waves of every SIMD one
// weight interval APART ("staggered"): the same arithmetic, weight image, fragments and outputs as k_mlp_fwd_pt2<false, .>, bit for bit.
//
// Why (profiles/r06_lds_ratio_probe.txt, profiles/r05_timeline_pt2_final.log): in k_mlp_fwd_pt2 all eight waves cross every stage barrier together, so
// the two waves of a SIMD finish a layer together and run its epilogue -- ~130 VALU instructions per wave: conversion, ReLU -- TOGETHER, with the SIMD's
// MFMA pipe idle; likewise the per-direction view tail.  The eval kernel's pipe is busy 62 % of a workgroup's life; 13 % of it are those epilogues, 8 % the
// tail.  Halving the LDS bytes per MFMA (the geometry VERDICT r05 names) measures +3 % at best; what is left to take is THIS: make one wave's epilogue
// fall into its SIMD partner's MFMAs.
//
// How: the 64 KiB weight stages of the image are consumed as two 32 KiB INTERVALS each (a stage is k-step-major, so its halves are k-steps 0-1 and
// 2-3: no other image), through a ring of four 32 KiB buffers (the same LDS as two 64 KiB ones).  Waves 0-3 ("group 0": one per SIMD) take weight
// interval s in workgroup interval s, waves 4-7 (their SIMD partners) in workgroup interval s + 1.  Every workgroup interval starts with ONE barrier
// (all eight waves: counts stay equal through an idle interval at group 1's start and group 0's end); the 32 DMA pieces of weight interval i + 2 go into
// the buffer that barrier freed (last read by group 1, one interval ago), issued by ONE group's four waves BEHIND their MFMAs of the interval (StagStream::late).  A layer is four intervals; group 0's epilogue
// of layer L opens the interval in which group 1 still multiplies layer L's last quarter, and vice versa one interval later.  gamma(x)'s two zero-padded
// half stages of the image (k-steps 2-3 of the PE stages: the 64 KiB plan pads K = 64 to a whole stage) are skipped by the stream, not multiplied.
#include "vipnerf_bf16n.h"
#include "vipnerf_mlp.h"
#include "vipnerf_mlp_pt2.h"

namespace vn {

namespace {
typedef BnPlan<1> PL;
constexpr int IV_CH = 32;                             // 1 KiB chunks per interval (half of a 64-chunk stage)
constexpr int IV_F = IV_CH * CHUNK_F;                 // floats per interval
constexpr int IV_RING = 4;
constexpr int IV_PER_WAVE = IV_CH / PL::WAVES;        // DMA pieces per wave and interval
// weight intervals of one tile, in consumption order: gamma(x) (1), layers 1..4 (4 each), layer 5 (4) + its gamma(x) columns (1), layers 6, 7 (4 each),
// feature layer (4), view layer's 256 feature columns (2)
constexpr int IV_L1 = 1, IV_L5PE = IV_L1 + 5 * 4, IV_L6 = IV_L5PE + 1, IV_VIEW = IV_L6 + 3 * 4, IV_TOTAL = IV_VIEW + 2;
static_assert(PL::KSB == 4 && PL::ST_256 == 2 && PL::ST_PE == 1 && PL::ST_VIEW_F == 1 && PL::CH == 2 * IV_CH, "the 64 KiB-stage image this kernel halves");
static_assert(IV_TOTAL == 36 && 2 * PL::F_STAGES == IV_TOTAL + 2, "36 intervals + the two skipped half stages of gamma(x)'s padding");
static_assert(PL::R_TOTAL_PAD + IV_RING * IV_F == PL::LDS_F, "the ring of four 32 KiB buffers is the LDS of two 64 KiB stages");
// half stage of the image that weight interval s reads (the second halves of the two gamma(x) stages are zero padding: skipped)
__device__ __forceinline__ int iv_source(int s) { return s + (s >= IV_L1 ? 1 : 0) + (s >= IV_L6 ? 1 : 0); }

// weight interval s opens a layer (its wave starts the interval with the previous layer's epilogue)
__device__ __forceinline__ bool iv_opens_layer(int s) {
    return s >= IV_L1 && (s < IV_L5PE ? ((s - IV_L1) & 3) == 0 : (s >= IV_L6 && ((s - IV_L6) & 3) == 0));
}

struct StagStream {
    const float *img;       // the forward image (PK_FWD), + lane * 4
    float *ring;
    int iv;                 // workgroup interval about to start (the same number in every wave)
    int lag;                // 0: waves 0-3, 1: waves 4-7 (scalar)
    int wave;
    int younger;            // DMA pieces this wave has issued AFTER its pieces of the weight interval the next open() needs (0, 4 or 8)
    // pieces [first, first + N) of weight interval s
    template <int N>
    __device__ __forceinline__ void issue(int s, int first) {
        glds_run<N>(img + (size_t)iv_source(s) * IV_F + first * CHUNK_F, ring + (s & (IV_RING - 1)) * IV_F + first * CHUNK_F);
    }
    __device__ __forceinline__ void start(const float *image, float *lds_ring, int lane, int wave_) {
        img = image + lane * 4; ring = lds_ring; iv = 0; wave = wave_;
        lag = __builtin_amdgcn_readfirstlane(wave_ >> 2);
        issue<IV_PER_WAVE>(0, wave * IV_PER_WAVE);            // the first two intervals: every wave its eighth
        issue<IV_PER_WAVE>(1, wave * IV_PER_WAVE);
        younger = IV_PER_WAVE;
    }
    // Opens workgroup interval iv.  (1) This wave's pieces of weight interval iv have landed: VM_CNT retires in order and the eval kernel has no stores,
    // so vmcnt(younger) -- the pieces it issued later, for weight interval iv + 1 -- proves it.  (2) Barrier: everybody's pieces have landed, and
    // everybody is done with the buffer of weight interval iv - 2 (which late() refills).
    // -> the buffer of THIS wave's weight interval (iv - lag); meaningless in the wave's idle interval.
    __device__ __forceinline__ const float *open() {
        __builtin_amdgcn_sched_barrier(0);
        if (younger == 2 * IV_PER_WAVE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IV_PER_WAVE) : "memory");
        else if (younger == IV_PER_WAVE) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IV_PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        younger = 0;
        const float *mine = ring + ((iv - lag) & (IV_RING - 1)) * IV_F;
        ++iv;
        return mine;
    }
    // Behind this wave's MFMAs of the interval open() opened (at once in its idle interval): the DMA of weight interval (that interval + 2) into the
    // buffer open()'s barrier freed, by the four waves of ONE group, 8 pieces each (~60 cycles of issue per piece).  Which group: the one whose SIMD
    // partners START the interval with a layer epilogue -- they multiply late, exactly while these waves issue --, group 0 (the older waves: the arbiter
    // lets them multiply first) where neither does, group 1 in interval 0 (its idle one).
    __device__ __forceinline__ void late() {
        const int i = iv - 1;
        if (i + 2 >= IV_TOTAL) return;
        const int issuer = i == 0 ? 1 : (iv_opens_layer(i) ? 1 : 0);          // (group 0 opens a layer in interval i: group 1 issues; group 1 does, or nobody: group 0)
        if (issuer == lag) {
            __builtin_amdgcn_sched_barrier(0);
            issue<2 * IV_PER_WAVE>(i + 2, (wave & 3) * 2 * IV_PER_WAVE);
            younger = 2 * IV_PER_WAVE;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
};
}  // namespace

template <bool F16>
__global__ __launch_bounds__(PL::WG) void k_mlp_eval_pt2s(MlpFwdArgs a) {
    typedef typename FragOf<F16>::type FR;
    typedef BOp<FR, 2> BT;
    typedef AccN<2> AT;
    constexpr int NS = 1;
    constexpr float XS = F16 ? F16_XSCALE : 1.f;
    constexpr float AU = F16 ? F16_ACC_UNSCALE : 1.f;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *res = lds;
    const float *rf = res + PL::R_F32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, j = lane & 15;
    int64_t p[2];
    bool valid[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int64_t p_raw = (int64_t)blockIdx.x * PT2_PTS_PER_WG + wave * 32 + pt * 16 + j;
        valid[pt] = p_raw < a.src.P;
        p[pt] = valid[pt] ? p_raw : a.src.P - 1;
    }
    StagStream ws;
    ws.start(a.packed + PL::PK_FWD, lds + PL::R_TOTAL_PAD, lane, wave);
    {
        const float4 *g4 = (const float4 *)(a.packed + PL::PK_RES);
        float4 *l4 = (float4 *)res;
        for (int i = tid; i < PL::R_TOTAL_PAD / 4; i += PL::WG) l4[i] = g4[i];
    }

    BT bin[8][NS];
    AT acc[16];
    float sigma_raw[2] = {0.f, 0.f};
    NoStream none;

    auto encode_pe = [&](BT (&bpe)[2][NS]) {
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            float pe[2][8];
            {
                PointCtx pc0;
                load_point(a.src, p[pt], pc0);
                encode_x16<VN_PT2_FAST_PE != 0>(pc0.x, q, pe);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                if (F16) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) pe[s][e] *= XS;
                }
                FR t[NS];
                split8<NS>(pe[s], t);
                bpe[s][0].v[pt] = t[0];
            }
        }
    };
    auto init_acc = [&](int layer) {
        const float *bias = rf + (layer < 8 ? PL::N_BIAS + layer * W : PL::N_BFEAT) + 4 * q;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float4 b4 = *(const float4 *)(bias + 16 * t);
            const floatx4 b = {b4.x, b4.y, b4.z, b4.w};
            acc[t].v[0] = b; acc[t].v[1] = b;
        }
    };
    // ReLU (trunk), sigma head (layer 7), conversion into the next layer's B fragments: k_mlp_fwd_pt2<false, F16>'s epilogue, value for value
    auto epilogue = [&](int layer) {
        const float lo = relu_bound<true>(layer < 8);
        const int lo_i = layer < 8 ? 0 : (int)0x80000000;
        if (layer == 7) {
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                float sg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const float4 w4 = *(const float4 *)(rf + PL::N_WSIG + 16 * t + 4 * q);
                    sg[0] = fmaf(w4.x, relu_pt2<F16>(acc[t].v[pt][0] * AU, 0.f, 0), sg[0]); sg[1] = fmaf(w4.y, relu_pt2<F16>(acc[t].v[pt][1] * AU, 0.f, 0), sg[1]);
                    sg[2] = fmaf(w4.z, relu_pt2<F16>(acc[t].v[pt][2] * AU, 0.f, 0), sg[2]); sg[3] = fmaf(w4.w, relu_pt2<F16>(acc[t].v[pt][3] * AU, 0.f, 0), sg[3]);
                }
                float s = (sg[0] + sg[1]) + (sg[2] + sg[3]);
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                sigma_raw[pt] = s + rf[PL::N_BHEAD];
            }
        }
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            if constexpr (!F16) {
                unsigned lo16 = layer < 8 ? 0u : 0x80008000u;
                asm volatile("" : "+s"(lo16));
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    FR t1[NS];
                    split_pair<NS>(acc[2 * s].v[pt], acc[2 * s + 1].v[pt], t1);
                    typedef unsigned u4 __attribute__((ext_vector_type(4)));
                    const u4 w4 = __builtin_bit_cast(u4, t1[0]);
                    unsigned w[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[i] = relu_pk16(w4[i], lo16);
                    const u4 r4 = {w[0], w[1], w[2], w[3]};
                    bin[s][0].v[pt] = __builtin_bit_cast(FR, r4);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    floatx4 x[2] = {acc[2 * s].v[pt], acc[2 * s + 1].v[pt]};
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int r = 0; r < 4; ++r) x[u][r] = relu_pt2<F16>(x[u][r] * AU, lo, lo_i);
                    x[0] *= XS; x[1] *= XS;
                    FR t1[NS];
                    split_pair<NS>(x[0], x[1], t1);
                    bin[s][0].v[pt] = t1[0];
                }
            }
        }
    };

    __syncthreads();                         // resident block visible
    if (ws.lag) { ws.open(); ws.late(); }    // group 1's idle interval: group 0 multiplies gamma(x) meanwhile
    // ---------------------------------------------------------------- layer 0: gamma(x) only
    BT bpe_keep[2][NS];
    {
        BT bpe[2][NS];
        encode_pe(bpe);
        bpe_keep[0][0] = bpe[0][0]; bpe_keep[1][0] = bpe[1][0];
        init_acc(0);
        const float *st = ws.open();
        gemm_stage_bf<16, 2, NS>(st, lane, acc, bpe, 0, none);
        ws.late();
    }
    // ---------------------------------------------------------------- layers 1..7 + feature layer (8): four intervals of two k-steps each.  A layer's
    // epilogue runs BEHIND the barrier that opens the next layer's first interval: the SIMD partner -- one interval behind or ahead -- multiplies meanwhile
#pragma unroll 1
    for (int layer = 1; layer < 9; ++layer) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const float *st = ws.open();
            if (jj == 0) { epilogue(layer - 1); init_acc(layer); }
            gemm_stage_bf<16, 2, NS>(st, lane, acc, bin, 2 * jj, none);
            ws.late();
        }
        if (layer == SKIP_LAYER) {           // gamma(x) columns last
            const float *st = ws.open();
            gemm_stage_bf<16, 2, NS>(st, lane, acc, bpe_keep, 0, none);
            ws.late();
        }
    }
    // ---------------------------------------------------------------- view branch: the 256 feature columns once per point (two intervals of four k-steps) ...
    AT vb[8];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const float *st = ws.open();
        if (jj == 0) {
            epilogue(8);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float4 b4 = *(const float4 *)(rf + PL::N_BVIEW + 16 * t + 4 * q);
                const floatx4 b = {b4.x, b4.y, b4.z, b4.w};
                vb[t].v[0] = b; vb[t].v[1] = b;
            }
        }
        gemm_stage_bf<8, 4, NS>(st, lane, vb, bin, 4 * jj, none);
        ws.late();
    }
    // group 0's idle interval: its view tail below runs under group 1's last interval.  (Nothing reads the ring after this barrier but group 1.)
    if (!ws.lag) { ws.open(); ws.late(); }

#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        float nz = 0.f;
        if (a.ns.noise) nz = a.ns.noise[p[pt]];
        else if (a.ns.device_rng) nz = rng_normal(a.ns.seed, a.ns.offset, a.ns.stream, noise_index(a.ns, a.src, p[pt]));
        const float sgm = relu_lo<true>(__fadd_rn(sigma_raw[pt], __fmul_rn(nz, a.ns.std)), 0.f);
        if (valid[pt] && q == 0) a.sigma[p[pt]] = sgm;
    }

    // ... then per point tile and direction a K = 32 GEMM from the LDS-resident direction columns, ReLU, the 128 -> 4 head
#pragma unroll 1
    for (int pt = 0; pt < 2; ++pt) {
        PointCtx pc;
        load_point(a.src, p[pt], pc);
#pragma unroll 1
        for (int dsel = 0; dsel <= a.src.V; ++dsel) {
            float dir[3];
            if (dsel == 0) { dir[0] = pc.dir[0]; dir[1] = pc.dir[1]; dir[2] = pc.dir[2]; }
            else secondary_dir(a.src, pc, dsel - 1, dir);
            float ped[1][8];
            encode_d16<VN_PT2_FAST_PE != 0>(dir, q, ped);
            FR bpd[1][NS];
            {
                float sc[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) sc[e] = ped[0][e] * XS;
                split8<NS>(sc, bpd[0]);
            }
            floatx4 g[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) g[t] = pt == 0 ? vb[t].v[0] : vb[t].v[1];
            gemm_stage_bf<8, 1, NS>(res + PL::R_DIRW, lane, g, bpd, 0, none);
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) g[t][r] = relu_pt2<F16>(g[t][r] * AU, 0.f, 0);
            float qv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (c < 3 && dsel != 0) continue;          // a secondary direction: the visibility only
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
            if (valid[pt] && q == 0) {
                const int64_t pp = p[pt];
                if (dsel == 0) {
                    a.rgb[3 * pp + 0] = qv[0]; a.rgb[3 * pp + 1] = qv[1]; a.rgb[3 * pp + 2] = qv[2];
                    a.vis[pp] = qv[3];
                } else {
                    a.vis2[pp * a.src.V + (dsel - 1)] = qv[3];
                }
            }
        }
    }
}

template <bool F16>
static int launch_eval(const MlpFwdArgs &a, hipStream_t st) {
    const unsigned grid = (unsigned)((a.src.P + PT2_PTS_PER_WG - 1) / PT2_PTS_PER_WG);
    const size_t lds = (size_t)PL::LDS_F * sizeof(float);
    VN_HIP(hipFuncSetAttribute((const void *)k_mlp_eval_pt2s<F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_mlp_eval_pt2s<F16>), dim3(grid), dim3(PL::WG), lds, st, a);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

// eval only (a.acts == NULL), bf16 only (see launch_mlp_fwd_pt2); a.packed: the narrow single-part image of VIPNERF_PREC_BF16
int launch_mlp_eval_pt2s(const MlpFwdArgs &a, int precision, hipStream_t st) {
    if (a.src.P <= 0) return VIPNERF_OK;
    if (a.acts) { set_error("mlp_eval_pt2s: the staggered kernel is the eval form (no activation store)"); return VIPNERF_E_ARG; }
    if (precision == VIPNERF_PREC_BF16) return launch_eval<false>(a, st);
    set_error("mlp_eval_pt2s: precision %d", precision);
    return VIPNERF_E_ARG;
}

}  // namespace vn
