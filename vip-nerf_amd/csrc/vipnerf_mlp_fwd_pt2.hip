// MLP.forward (reference src/models/VipNeRF01.py:509-596) in the single-MFMA 16-bit modes (VIPNERF_PREC_FP16 / BF16) with TWO point
// tiles per wave: 8 waves x 32 points = 256 points per workgroup, two waves per SIMD.
//
// Why: with one MFMA per product a 16-point wave consumes a 1 KiB A fragment from LDS every 16 cycles -- eight waves at full MFMA rate
// would read 256 B/clk, all the LDS has -- and every 64 KiB weight stage carries only 64 MFMAs per wave against ~800 cycles of
// per-stage cost (barrier, first-read latency, DMA issue).  Two point tiles that share every A fragment halve the LDS bytes and the
// weight DMA per point and double the MFMAs under each stage's fixed costs.  A wave's registers: 2 x 64 accumulators + 2 x 32 operand
// registers + 24 of A-fragment ring -- what the single-part (NS = 1) fragments leave room for at two waves per SIMD, and the split
// arithmetics (2 or 3 operand parts) do not.
//
// Same algorithm, weight image (BnPlan<1>), stage order, T16 operand storage and ReLU masks as k_mlp_fwd_bf16n<., 1, ., 4> (the 16-point
// kernel is retired); lane (j, q) holds features 16 T + 4 q .. + 3 of C/D tile T for point j of EACH of its two point tiles.
#include "vipnerf_bf16n.h"
#include "vipnerf_mlp.h"
#include "vipnerf_mlp_pt2.h"

namespace vn {

TS_DECL(g_pt2_timeline);
#define TS(tag) TS_AT(g_pt2_timeline, tag)

template <bool SAVE, bool F16>
__global__ __launch_bounds__(BnPlan<1>::WG) void k_mlp_fwd_pt2(MlpFwdArgs a) {
    typedef BnPlan<1> PL;
    typedef typename FragOf<F16>::type FR;
    typedef BOp<FR, 2> BT;
    typedef AccN<2> AT;
    constexpr int NS = 1;
    constexpr float XS = F16 ? F16_XSCALE : 1.f;
    constexpr float AU = F16 ? F16_ACC_UNSCALE : 1.f;
    constexpr int S_PER_STAGE = 8 / PL::ST_256;            // operand k-steps a stage's deferred stores cover
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *res = lds;
    float *stage_buf = lds + PL::R_TOTAL_PAD;
    const float *rf = res + PL::R_F32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4, j = lane & 15;
    TS_INIT();
    TS(TS_ENTRY);                                    // 0: entry
    int64_t p[2], grp[2];
    bool valid[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int64_t p_raw = (int64_t)blockIdx.x * PT2_PTS_PER_WG + wave * 32 + pt * 16 + j;
        valid[pt] = p_raw < a.src.P;
        p[pt] = valid[pt] ? p_raw : a.src.P - 1;
        grp[pt] = (int64_t)blockIdx.x * (PT2_PTS_PER_WG / 16) + wave * 2 + pt;      // (valid is wave-uniform in training: P % 16 == 0)
    }

    const int store_phase = pt2_store_phase(wave);
    typename StreamOfAll<PL>::type ws;
    ws.start(a.packed + PL::PK_FWD, PL::F_STAGES, stage_buf, lane, wave);
    ws.counted = !SAVE || (valid[0] && valid[1]);      // the counted waits assume the stores of BOTH point tiles
    {
        const float4 *g4 = (const float4 *)(a.packed + PL::PK_RES);
        float4 *l4 = (float4 *)res;
        for (int i = tid; i < PL::R_TOTAL_PAD / 4; i += PL::WG) l4[i] = g4[i];
    }

    BT bin[8][NS];                           // the layer input as B fragments: k-step s <- C/D tiles 2s, 2s+1, both point tiles
    AT acc[16];
    float sigma_raw[2] = {0.f, 0.f};

    // gamma(x) of both point tiles as the 16-bit B fragments of its two k-steps (slot order of pe_feat16)
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
    // ReLU (trunk), ReLU bits, sigma head (layer 7), split into the next layer's B fragments, the feature's stores (layer 8)
    auto epilogue = [&](int layer) {
        const float lo = relu_bound<true>(layer < 8);
        const int lo_i = layer < 8 ? 0 : (int)0x80000000;        // the same bound for relu_pt2's integer form (bf16 fragments)
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
            unsigned mk0 = 0u, mk1 = 0u;
            if constexpr (!F16 && NS == 1) {
                // bf16: convert, then ReLU and ReLU bits on the packed halves (vipnerf_mlp_pt2.h relu_pk16): 2 (4 with the bits) VALU
                // instructions per two values instead of 3 (7)
                unsigned lo16 = layer < 8 ? 0u : 0x80008000u, one2 = 0x00010001u;
                asm volatile("" : "+s"(lo16), "+s"(one2));      // one SGPR operand: the compiler otherwise selects between two results per value
                unsigned cb[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    FR t1[NS];
                    split_pair<NS>(acc[2 * s].v[pt], acc[2 * s + 1].v[pt], t1);
                    typedef unsigned u4 __attribute__((ext_vector_type(4)));
                    const u4 w4 = __builtin_bit_cast(u4, t1[0]);
                    unsigned w[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) w[i] = relu_pk16(w4[i], lo16);
                    if (SAVE) cb[s] = positive_pk_bits(w, one2);
                    const u4 r4 = {w[0], w[1], w[2], w[3]};
                    bin[s][0].v[pt] = __builtin_bit_cast(FR, r4);
                }
                if (SAVE) {
                    mk0 = fold_pk_bits(cb[0], cb[1]) | (fold_pk_bits(cb[2], cb[3]) << 16);
                    mk1 = fold_pk_bits(cb[4], cb[5]) | (fold_pk_bits(cb[6], cb[7]) << 16);
                }
            } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                floatx4 x[2] = {acc[2 * s].v[pt], acc[2 * s + 1].v[pt]};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int t = 2 * s + u;
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[u][r] = relu_pt2<F16>(x[u][r] * AU, lo, lo_i);      // (+0 | positive | NaN: the bit masks)
                    if (SAVE) { if (t < 8) mk0 = push_nibble(mk0, positive_nibble(x[u])); else mk1 = push_nibble(mk1, positive_nibble(x[u])); }
                }
                if (F16) { x[0] *= XS; x[1] *= XS; }
                FR t1[NS];
                split_pair<NS>(x[0], x[1], t1);
                bin[s][0].v[pt] = t1[0];
            }
            }
            // valid tiles only: a tile beyond P runs on point P - 1's position but reloads gamma(x) of group 0 at layer 5 (below) -- from there on
            // its ReLU bits are not point P - 1's, and an unpredicated store would race the owner's (the 16-point kernels keep gamma(x) in
            // registers: their out-of-range lanes write point P - 1's own bits again)
            if (SAVE && layer < 8 && valid[pt] && !EXP_NO_STORES) *(uint2 *)(a.acts + a.al.hm[layer] + ((size_t)p[pt] * 4 + q) * 2) = make_uint2(mk0, mk1);
        }
    };

    __syncthreads();                         // resident block visible
    TS(TS_RESIDENT);                                    // 1: resident block in
    // ---------------------------------------------------------------- layer 0: gamma(x) only (the first two k-steps of a stage whose other two are zero padding)
    // (build switches VN_PT2_TRAIN_KEEP / VN_PT2_EVAL_KEEP, vipnerf_knobs.h: gamma(x)'s fragments kept in registers up to layer 5, or reloaded / evaluated again)
    BT bpe_keep[2][NS];
    {
        BT bpe[2][NS];
        encode_pe(bpe);
        TS(TS_HEAD);                                // 2: gamma(x) encoded
        if ((!SAVE && VN_PT2_EVAL_KEEP) || (SAVE && VN_PT2_TRAIN_KEEP)) { bpe_keep[0][0] = bpe[0][0]; bpe_keep[1][0] = bpe[1][0]; }
        if (SAVE) {
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                if (valid[pt] && !EXP_NO_PE) {           // gamma(x), slot order: column 16 q + u of a 64-wide T16 array = tile q
                    typedef unsigned u4 __attribute__((ext_vector_type(4)));
                    char *row = (char *)(a.acts + a.al.pex) + ((size_t)grp[pt] * 4 + q) * 512 + j * 32;
                    __builtin_nontemporal_store(__builtin_bit_cast(u4, bpe[0][0].v[pt]), (u4 *)row);
                    __builtin_nontemporal_store(__builtin_bit_cast(u4, bpe[1][0].v[pt]), (u4 *)(row + 16));
                }
            }
        }
        init_acc(0);
        TS(TS_PRE);
        const float *st = ws.template wait<SAVE ? 4 : 0>();      // behind gamma(x)'s own stores
        TS(TS_POST);
        gemm_stage_bf<16, 2, NS>(st, lane, acc, bpe, 0, ws);
        TS(TS_END);
        epilogue(0);
    }
    // ---------------------------------------------------------------- layers 1..7 + feature layer (8)
#pragma unroll 1
    for (int layer = 1; layer < 9; ++layer) {
        init_acc(layer);
#pragma unroll
        for (int jj = 0; jj < PL::ST_256; ++jj) {
            // younger than this stage's DMA: the mask stores of the previous epilogue (first stage), the deferred stores behind the stage before
            TS(TS_PRE);
            const float *st = jj == 0 ? ws.template wait<SAVE ? 2 : 0>() : ws.template wait<SAVE ? 2 * T16_SPK * S_PER_STAGE : 0>();
            TS(TS_POST);
            if (SAVE) {
                DeferredT16<FR, S_PER_STAGE> ds{a.acts + a.al.h[layer - 1], {grp[0], grp[1]}, {valid[0], valid[1]}, j, q, S_PER_STAGE * jj, bin, store_phase};
                gemm_stage_bf<16, PL::KSB, NS>(st, lane, acc, bin, PL::KSB * jj, ws, ds);
            } else {
                gemm_stage_bf<16, PL::KSB, NS>(st, lane, acc, bin, PL::KSB * jj, ws);
            }
            TS(TS_END);
        }
        if (layer == SKIP_LAYER) {           // gamma(x) columns last: h's operand registers are dead by then
            BT bpe[2][NS];
            if (SAVE && VN_PT2_TRAIN_KEEP) {
                bpe[0][0] = bpe_keep[0][0]; bpe[1][0] = bpe_keep[1][0];
            } else if (SAVE) {               // training: the fragments come back from the activation store (written in exactly this form)
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    const char *row = (const char *)(a.acts + a.al.pex) + ((size_t)(valid[pt] ? grp[pt] : 0) * 4 + q) * 512 + j * 32;
                    bpe[0][0].v[pt] = __builtin_bit_cast(FR, *(const u4 *)row);
                    bpe[1][0].v[pt] = __builtin_bit_cast(FR, *(const u4 *)(row + 16));
                }
            } else if (VN_PT2_EVAL_KEEP) {
                bpe[0][0] = bpe_keep[0][0]; bpe[1][0] = bpe_keep[1][0];
            } else {
                encode_pe(bpe);              // eval: evaluated again rather than held in 16 registers across layers 1..4
            }
            TS(TS_PRE);
            const float *st = ws.template wait<SAVE ? 2 * T16_SPK * S_PER_STAGE : 0>();      // behind the deferred stores of the stage before
            TS(TS_POST);
            gemm_stage_bf<16, 2, NS>(st, lane, acc, bpe, 0, ws);
            TS(TS_END);
        }
        epilogue(layer);
#if defined(VN_EXP) && VN_EXP == 47
        {   // timing experiment only: the layer epilogue's VALU work twice (profiles/r04_ablation_pt2.md: what a second copy costs)
#pragma unroll
            for (int t = 0; t < 16; ++t)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) asm volatile("" : "+v"(acc[t].v[pt]));
            epilogue(layer);
        }
#endif
    }

#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        float nz = 0.f;
        if (a.ns.noise) nz = a.ns.noise[p[pt]];
        else if (a.ns.device_rng) nz = rng_normal(a.ns.seed, a.ns.offset, a.ns.stream, noise_index(a.ns, a.src, p[pt]));
        const float sgm = relu_lo<true>(__fadd_rn(sigma_raw[pt], __fmul_rn(nz, a.ns.std)), 0.f);
        if (valid[pt] && q == 0) a.sigma[p[pt]] = sgm;
    }

    // ---------------------------------------------------------------- view branch: the 256-wide part once per point ...
    AT vb[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const float4 b4 = *(const float4 *)(rf + PL::N_BVIEW + 16 * t + 4 * q);
        const floatx4 b = {b4.x, b4.y, b4.z, b4.w};
        vb[t].v[0] = b; vb[t].v[1] = b;
    }
    static_assert(PL::ST_VIEW_F == 1, "the feature's deferred stores assume one view stage");
    PointCtx pc0, pc1;
    float sec0[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    {
        TS(TS_PRE);
        const float *st = ws.template wait<SAVE ? 2 * T16_SPK * S_PER_STAGE : 0>();      // behind the deferred stores of the feature layer's last stage
        TS(TS_POST);
        // the per-direction tail's global loads (the ray behind each point tile, the first secondary view's origin / direction) issued
        // HERE, behind the last counted wait: they come back during the stage instead of stalling the tail, which runs with the MFMA pipe
        // idle (and under the training step's store traffic a load is thousands of cycles).  Training only: the eval kernel's loads come
        // back fast and it measured 3 % slower with them up here (profiles/r04_ablation_pt2.md 5).
        if (SAVE) { load_point(a.src, p[0], pc0); load_point(a.src, p[1], pc1); }
        if (SAVE && a.src.V >= 1) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                sec0[0][i] = a.src.rays_mode ? a.src.rays_o2[(pc0.n * a.src.V) * 3 + i] : a.src.dirs2[(pc0.p * a.src.V) * 3 + i];
                sec0[1][i] = a.src.rays_mode ? a.src.rays_o2[(pc1.n * a.src.V) * 3 + i] : a.src.dirs2[(pc1.p * a.src.V) * 3 + i];
            }
        }
        if (SAVE) {          // the feature (= this GEMM's B operand) leaves from inside the stage like h_1..h_8
            DeferredT16<FR, 8> ds{a.acts + a.al.feat, {grp[0], grp[1]}, {valid[0], valid[1]}, j, q, 0, bin, store_phase};
            gemm_stage_bf<8, PL::KSV, NS>(st, lane, vb, bin, 0, ws, ds);
        } else {
            gemm_stage_bf<8, PL::KSV, NS>(st, lane, vb, bin, 0, ws);
        }
    }

    TS(TS_VIEW);                                    // view stage done
    // ... then per point tile and direction a K = 32 GEMM from the LDS-resident direction columns, ReLU, the 128 -> 4 head
#pragma unroll 1
    for (int pt = 0; pt < 2; ++pt) {
        PointCtx pc;                         // field by field: a struct select goes through scratch memory
        if (!SAVE) load_point(a.src, p[pt], pc);
        else
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            pc.x[i] = pt == 0 ? pc0.x[i] : pc1.x[i]; pc.dir[i] = pt == 0 ? pc0.dir[i] : pc1.dir[i];
            pc.o[i] = pt == 0 ? pc0.o[i] : pc1.o[i]; pc.d[i] = pt == 0 ? pc0.d[i] : pc1.d[i];
        }
        if (SAVE) { pc.z = pt == 0 ? pc0.z : pc1.z; pc.p = pt == 0 ? pc0.p : pc1.p; pc.n = pt == 0 ? pc0.n : pc1.n; }
        const float s0[3] = {pt == 0 ? sec0[0][0] : sec0[1][0], pt == 0 ? sec0[0][1] : sec0[1][1], pt == 0 ? sec0[0][2] : sec0[1][2]};
#pragma unroll 1
        for (int dsel = 0; dsel <= a.src.V; ++dsel) {
            float dir[3];
            if (dsel == 0) { dir[0] = pc.dir[0]; dir[1] = pc.dir[1]; dir[2] = pc.dir[2]; }
            else if (SAVE && dsel == 1) {
                if (a.src.rays_mode) secondary_dir_from(a.src, pc, s0, dir);
                else { dir[0] = s0[0]; dir[1] = s0[1]; dir[2] = s0[2]; }
            }
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
            { NoStream none; gemm_stage_bf<8, 1, NS>(res + PL::R_DIRW, lane, g, bpd, 0, none); }
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) g[t][r] = relu_pt2<F16>(g[t][r] * AU, 0.f, 0);
            if (SAVE && valid[pt] && !EXP_NO_EXTRAS) {
                unsigned gm = 0u, cb[4];
                unsigned one2 = 0x00010001u;
                asm volatile("" : "+s"(one2));
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    FR gh[NS];
                    split_pair<NS>(g[2 * s], g[2 * s + 1], gh);
                    store_t16(a.acts + a.al.g[dsel], grp[pt], 8, s, j, q, gh[0]);
                    if constexpr (!F16 && NS == 1) {       // bf16: the ReLU bits from the packed halves, two values per instruction (like the trunk's)
                        typedef unsigned u4 __attribute__((ext_vector_type(4)));
                        const u4 w4 = __builtin_bit_cast(u4, gh[0]);
                        const unsigned w[4] = {w4[0], w4[1], w4[2], w4[3]};
                        cb[s] = positive_pk_bits(w, one2);
                    } else {
                        gm = push_nibble(gm, positive_nibble(g[2 * s]));
                        gm = push_nibble(gm, positive_nibble(g[2 * s + 1]));
                    }
                }
                if constexpr (!F16 && NS == 1) gm = fold_pk_bits(cb[0], cb[1]) | (fold_pk_bits(cb[2], cb[3]) << 16);
                ((unsigned *)(a.acts + a.al.g[dsel] + (size_t)a.src.P * (WV / 2)))[(size_t)p[pt] * 4 + q] = gm;
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
                char *row = (char *)(a.acts + a.al.ped[dsel]) + ((size_t)grp[pt] * 2 + (q >> 1)) * 512 + j * 32 + (q & 1) * 16;
                __builtin_nontemporal_store(__builtin_bit_cast(u4, bpd[0][0]), (u4 *)row);
            }
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
    TS(TS_LAST);                                    // last: view tail done
}

#if defined(VN_EXP) && VN_EXP == 50
extern "C" int vipnerf_exp_timeline(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pt2_timeline), sizeof(unsigned long long) * (n < 2048 ? n : 2048));
}
#endif

template <bool SAVE, bool F16>
static int launch_pt2(const MlpFwdArgs &a, hipStream_t st) {
    const unsigned grid = (unsigned)((a.src.P + PT2_PTS_PER_WG - 1) / PT2_PTS_PER_WG);
    const size_t lds = (size_t)BnPlan<1>::LDS_F * sizeof(float);
    VN_HIP(hipFuncSetAttribute((const void *)k_mlp_fwd_pt2<SAVE, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_mlp_fwd_pt2<SAVE, F16>), dim3(grid), dim3(BnPlan<1>::WG), lds, st, a);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

// a.packed: the narrow single-part image of the precision (VIPNERF_PREC_FP16 = 5 / BF16 = 6)
int launch_mlp_fwd_pt2(const MlpFwdArgs &a, int precision, hipStream_t st) {
    if (a.src.P <= 0) return VIPNERF_OK;
    if (a.acts && a.src.P % 16) { set_error("mlp_fwd: the 16-bit training kernels need a multiple of 16 points (got %lld)", (long long)a.src.P); return VIPNERF_E_UNSUPPORTED; }
    if (precision == VIPNERF_PREC_FP16) return a.acts ? launch_pt2<true, true>(a, st) : launch_pt2<false, true>(a, st);
    if (precision == VIPNERF_PREC_BF16) return a.acts ? launch_pt2<true, false>(a, st) : launch_pt2<false, false>(a, st);
    set_error("mlp_fwd_pt2: precision %d", precision);
    return VIPNERF_E_ARG;
}

}  // namespace vn
