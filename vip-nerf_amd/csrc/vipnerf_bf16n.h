// "Narrow-wave" variant of the split-precision bf16 path (vipnerf_bf16.h): 16 points per wave on
// v_mfma_f32_16x16x32_bf16, 8 waves (two per SIMD) and still 128 points per workgroup.
//
// Why: with one 32-point wave per SIMD (the wide layout) nothing feeds a SIMD's MFMA pipe while its only wave issues
// the weight DMA (~60 cycles per 1 KiB piece), runs a layer's epilogue (bias/ReLU/mask/operand split: VALU) or sits
// at the stage barrier -- 50-55 % of the bf16 kernels' time (docs/HISTORY.md 4.1b).  A 16-point wave needs half the
// registers (64 accumulators + 64/96 operand registers), so two fit on a SIMD and cover for each other.  The price:
// a 16x16x32 MFMA consumes a 1 KiB A fragment every 16 cycles instead of every 32, i.e. twice the LDS read traffic
// (171 B/clk of 256 in bf16x3, 128 in bf16x6).
//
// Fragments.  lane l = (j = l & 15: point, q = l >> 4).  A: row i = l & 15 of a 16-feature tile, k = 8q + e (e = 0..7)
// of a 32-deep k-step.  C/D of tile T: features 16T + 4q + r (r = 0..3) of point j.  A k-step therefore takes its B
// operand from two consecutive C/D tiles of the previous layer: contraction index (s, q, e) of k-step s is feature
//     feat16(s, q, e) = 16 (2s + (e >> 2)) + 4q + (e & 3),
// so layers chain in registers exactly as in the wide layout, and biases / heads are read in natural feature order.
#pragma once
#include "vipnerf_bf16.h"

namespace vn {

__host__ __device__ constexpr int feat16(int s, int q, int e) { return 16 * (2 * s + (e >> 2)) + 4 * q + (e & 3); }

// gamma(x) (63 features, 2 k-steps) and gamma(dir) (27 features, 1 k-step) enter as B operands whose contraction order
// is free (the weights are packed to match), so each lane group q is given features it can compute from few sincos:
//   gamma(x):   slot u = 8s + e of lane group q: u < 15 -> feature 3 + 15q + u  (5 consecutive (level, sin|cos) triples,
//               levels (5q)>>1 .. +2: 9 sincosf per lane instead of 30);  u = 15 -> x[q] (q < 3), unused for q = 3
//   gamma(dir): slot e of lane group q: e < 6 -> feature 3 + 6q + e (level q: 3 sincosf instead of 12);
//               (q,e) = (0,6) -> d[0], (0,7) -> d[1], (1,6) -> d[2]; other slots unused
// -1 = unused slot (zero weight column).
__host__ __device__ constexpr int pe_feat16(int s, int q, int e) {
    return s >= 2 ? -1 : (8 * s + e < 15 ? 3 + 15 * q + 8 * s + e : (q < 3 ? q : -1));   // k-steps >= 2: padding (single-MFMA plan)
}
__host__ __device__ constexpr int dir_feat16(int q, int e) {
    return e < 6 ? 3 + 6 * q + e : (q == 0 ? e - 6 : (q == 1 && e == 6 ? 2 : -1));
}

// timing-only builds (tools/build_exp.sh N; results are garbage): which stores of the narrow training kernels are left out.
//   40: every activation / gradient / mask / encoding store (the kernels' compute floor)
//   41: the fp32 "extras" only -- feature, view hidden, gamma(x), gamma(dir); dY_0, the fp32 dY_5 copy, dYv, sum dYv
//   42: gamma(x) / gamma(dir) only
//   43: (vipnerf_bf16.h) no MFMAs in the two-point-tile kernels: what their stores, weight stream and epilogues cost alone
//   44: 40 and 43 together: neither stores nor MFMAs (weight stream, barriers, encodings, epilogues)
//    0: nothing left out -- the product kernels, in a library that honours the experiment-only environment switch
//       VIPNERF_EXP_SKIP_WGRAD (render_backward without its weight-gradient launches: the data-gradient kernels alone under rocm-smi)
#if defined(VN_EXP)
constexpr bool EXP_NO_STORES = VN_EXP == 40 || VN_EXP == 44, EXP_NO_EXTRAS = EXP_NO_STORES || VN_EXP == 41, EXP_NO_PE = EXP_NO_EXTRAS || VN_EXP == 42;
#else
constexpr bool EXP_NO_STORES = false, EXP_NO_EXTRAS = false, EXP_NO_PE = false;
#endif

// VN_SKEW (two-part modes only): waves 4..7 -- the second wave of every SIMD -- run one weight stage behind waves 0..3,
// so that one wave's layer epilogue (VALU, stores) falls under the other's MFMAs instead of both doing it at once.
// Needs three resident stages, hence half-size ones (32 KiB); WStreamSkew below.  Built, correct (all tests pass with
// -DVN_SKEW=1) and measured on the same box: forward 4.63 vs 4.35 ms per step, data gradients 5.05 vs 5.01 -- what the
// overlap gains, twice as many workgroup barriers take back.  Off by default.
// build switch VN_SKEW (default 0, vipnerf_knobs.h)
template <int NS>
struct BnPlan {
    static constexpr int WAVES = 8;
    static constexpr int WG = 64 * WAVES;
    static constexpr int NT = 16;                            // 16-feature tiles of a 256-wide layer
    static constexpr bool SKEW = NS == 2 && VN_SKEW;
    // k-steps (of 32) per stage for a 16-tile layer.  NS == 1 (single-MFMA fp16 / bf16 modes): 4, so that a stage is still
    // 64 KiB; gamma(x) (2 k-steps) is then padded to one whole stage with zero weight columns (pe_feat16 -> -1).
    static constexpr int KSB = SKEW ? 1 : (NS == 2 ? 2 : (NS == 1 ? 4 : 1));
    static constexpr int PE_KS = KSB > 2 ? KSB : 2;          // k-steps the gamma(x) operand occupies (>= 2, whole stages)
    static constexpr int CH = KSB * NT * NS;                 // chunks (1 KiB) per stage: 64 (NS=2) / 48 (NS=3) / 32 (skew)
    static constexpr int STAGE_F = CH * CHUNK_F;
    static constexpr int NBUF = SKEW ? 3 : 2;
    static constexpr int ST_256 = 8 / KSB;                   // 256-deep contraction = 8 k-steps
    static constexpr int ST_PE = PE_KS / KSB;                // gamma(x): K = 64 -> 2 k-steps (padded to a stage when KSB = 4)
    static constexpr int KSV = 2 * KSB;                      // k-steps per stage when a stage spans 8 tiles
    static constexpr int ST_VIEW_F = 8 / KSV;                // view layer forward: 8 tiles x 8 k-steps
    static constexpr int ST_VIEW_B = 4 / KSB;                // view layer dgrad: 16 tiles x 4 k-steps (K = 128)
    static constexpr int FS_L0PE = 0;
    static constexpr int FS_L1 = FS_L0PE + ST_PE;
    static constexpr int FS_L5 = FS_L1 + 4 * ST_256;         // layer 5: the 256 h-columns first, then gamma(x) -- the
    static constexpr int FS_L5PE = FS_L5 + ST_256;           // operand registers of h are dead by then and hold gamma(x)'s parts
    static constexpr int FS_L6 = FS_L5PE + ST_PE;
    static constexpr int FS_L7 = FS_L6 + ST_256;
    static constexpr int FS_FEAT = FS_L7 + ST_256;
    static constexpr int FS_VIEW = FS_FEAT + ST_256;
    static constexpr int F_STAGES = FS_VIEW + ST_VIEW_F;
    static constexpr int BS_VIEW = 0;
    static constexpr int BS_FEAT = BS_VIEW + ST_VIEW_B;
    static constexpr int BS_L7 = BS_FEAT + ST_256;
    static constexpr int B_STAGES = BS_L7 + 7 * ST_256;
    // LDS-resident block: direction columns of the view layer as A fragments (1 k-step x 8 tiles x NS chunks), then
    // the fp32 biases / heads in natural feature order
    static constexpr int R_DIRW = 0;
    static constexpr int R_DIRW_F = 8 * NS * CHUNK_F;
    static constexpr int R_F32 = R_DIRW_F;
    static constexpr int N_BIAS = 0;                         // [8 layers][256]
    static constexpr int N_BFEAT = N_BIAS + 8 * W;
    static constexpr int N_BVIEW = N_BFEAT + W;
    static constexpr int N_WSIG = N_BVIEW + WV;
    static constexpr int N_WOUT = N_WSIG + W;                // [4][128]
    static constexpr int N_BHEAD = N_WOUT + 4 * WV;          // sigma bias, 4 output biases, pad
    static constexpr int N_TOTAL = N_BHEAD + 8;
    static constexpr int R_TOTAL = R_F32 + N_TOTAL;
    static constexpr int R_TOTAL_PAD = (R_TOTAL + 255) / 256 * 256;
    static constexpr size_t PK_FWD = 0;
    static constexpr size_t PK_BWD = PK_FWD + (size_t)F_STAGES * STAGE_F;
    static constexpr size_t PK_RES = PK_BWD + (size_t)B_STAGES * STAGE_F;
    static constexpr size_t PK_TOTAL_F = PK_RES + R_TOTAL_PAD;
    static constexpr int LDS_F = R_TOTAL_PAD + NBUF * STAGE_F;
    static_assert(LDS_F * 4 <= 160 * 1024, "LDS budget");
};

// precision 0 (exact fp32): the fp32-narrow image (two 4-float parts per k-step: BnPlan<2>'s geometry) follows the wide fp32 one
__host__ __device__ inline size_t packed_narrow_floats(int precision) {
    return precision == 2 ? BnPlan<3>::PK_TOTAL_F : (precision >= 5 ? BnPlan<1>::PK_TOTAL_F : BnPlan<2>::PK_TOTAL_F);
}

// "fp16x3" (VIPNERF_PREC_FP16X3): the narrow kernels with fp16 fragments, x = x0 + x1 with 11-bit parts, three cross
// terms a0b0 + a0b1 + a1b0 on v_mfma_f32_16x16x32_f16: per-product error <= ~3 * 2^-22 (the dropped a1b1 and the two
// residuals), i.e. close to fp32, at the MFMA count and weight-stream size of bf16x3.  fp16 has 5 exponent bits:
//   * weights are packed as 2^8 w (exact): |w| ~ 0.06 -> ~16, so the low part 2^-11 * 16 is a normal fp16 number;
//     2^8 |w| < 65504 up to |w| = 255.  Accumulators start at 2^8 bias and the epilogue takes 2^-8 off again.
//   * the B operands (activations, encodings) are split as they are: |x| < 65504, all of fp16's range.  (A scale of
//     2^4 was measured: no accuracy difference on the goldens -- the MFMA does not flush fp16 subnormals, and what sits
//     below the normal range is <= 2^-14 of an operand whose partner is O(10) -- but it cost a factor 16 of range.)
//   * beyond that range the conversion gives inf, inf - inf in the low part gives NaN, and the F16 kernels' ReLU
//     passes NaN on (x < 0 ? 0 : x instead of max), so the outputs and the loss are NaN: out-of-range inputs fail
//     loudly instead of being squashed to finite garbage (tools/range_diag.py).
#if defined(VN_F16_WS)
constexpr float F16_WSCALE = VN_F16_WS, F16_XSCALE = VN_F16_XS;    // diagnostic build
#else
constexpr float F16_WSCALE = 256.f;           // weights
constexpr float F16_XSCALE = 1.f;             // B operands of the forward pass
#endif
constexpr float F16_ACC_SCALE = F16_WSCALE * F16_XSCALE;
constexpr float F16_ACC_UNSCALE = 1.f / F16_ACC_SCALE;

#if defined(__HIPCC__)
// Skewed weight stream (VN_SKEW): a ring of three stages.  Time is cut into intervals by one workgroup barrier each
// (tick); in interval k waves 0..3 consume stage k, waves 4..7 stage k-1, and wave (k+1) mod 8 issues the DMA of stage
// k+1 into the slot of stage k-2, which the lagging group finished reading before the barrier.  Every wave executes
// the same number of ticks: the lagging group one idle tick first (begin), the leading group one after its last stage
// (end).
template <int CH>
struct WStreamSkew {
    const float *g0;
    float *buf;
    int n_total, tickno, lag, slot, lane, wave;
    static constexpr int SF = CH * CHUNK_F;
    __device__ __forceinline__ void issue(int s) {
        if (s < n_total && (s & 7) == wave) {
            const int sl = s % 3;
            glds_run<CH>(g0 + (size_t)s * SF + lane * 4, buf + sl * SF);
        }
    }
    __device__ __forceinline__ void start(const float *stream, int n_stages, float *lds_buf, int lane_, int wave_) {
        g0 = stream; buf = lds_buf; n_total = n_stages; tickno = 0; lane = lane_; wave = wave_; lag = wave_ >> 2; slot = 0;
        issue(0);
    }
    __device__ __forceinline__ void tick() {
        glds_drain();
        __syncthreads();
        issue(tickno + 1);
        ++tickno;
    }
    __device__ __forceinline__ void begin() { if (lag) tick(); }
    __device__ __forceinline__ void end() { if (!lag) tick(); }
    template <int YOUNGER = 0, int YOUNGER_FIRST = YOUNGER>
    __device__ __forceinline__ const float *wait(bool = false) {
        tick();
        const float *ret = buf + slot * SF;
        slot = slot == 2 ? 0 : slot + 1;
        return ret;
    }
    __device__ __forceinline__ void prefetch() {}
    template <int g, int NG> __device__ __forceinline__ void prefetch_at() {}
};
// build switch VN_DMA_MODE (default 1, vipnerf_knobs.h): 1: one wave issues a whole stage (ROTATE); 2: every wave its share, staggered over the stage
// build switch VN_DMA_ROT_WAVES (default 4, vipnerf_knobs.h): the waves a stage's issuer rotates over (4: the older wave of every SIMD only; 8: all)
template <typename PL, bool SK> struct StreamOf { typedef WStreamT<PL::CH, PL::NBUF, PL::WAVES, VN_DMA_MODE == 1, VN_DMA_MODE == 2, VN_DMA_MODE == 1 ? VN_DMA_ISSUERS : 1, VN_DMA_MODE == 1 ? VN_DMA_ROT_WAVES : PL::WAVES> type; };
template <typename PL> struct StreamOf<PL, true> { typedef WStreamSkew<PL::CH> type; };
// the exact-fp32 kernels (vipnerf_mlp_{fwd,bwd}_f32.hip): the four OLDER waves share every stage's DMA, a quarter each (build switch VN_F32_DMA_ISSUERS,
// default 4, vipnerf_knobs.h).  With one issuer per stage that wave's loop took 18.1k cycles against 10.4k (64 pieces x ~120 cycles of issue among its
// MFMAs) and its SIMD was the stage's last; a quarter of the burst fits the ~7k cycles of slack every older wave has: fp32 step -0.15 ms
// (profiles/r05_ab_dma_issuers.log)
template <typename PL> struct StreamOlder { typedef WStreamT<PL::CH, PL::NBUF, PL::WAVES, true, false, VN_F32_DMA_ISSUERS, 4> type; };
// the two-point-tile 16-bit kernels: every wave takes its turn (issuing from the older waves only -- one, two or four of them per stage -- measured
// neutral there: profiles/r05_ab_rot_waves.log, r05_ab_dma_issuers.log)
template <typename PL> struct StreamOfAll { typedef WStreamT<PL::CH, PL::NBUF, PL::WAVES, VN_DMA_MODE == 1, VN_DMA_MODE == 2, VN_DMA_MODE == 1 ? VN_DMA_ISSUERS : 1> type; };
// every wave issues its eighth of a stage's DMA and drains it itself (plain vmcnt(0) + barrier): measured +1.8 % for the exact-fp32
// EVAL kernel (0.887 -> 0.903 of the fp32 peak: no stores whose latency that vmcnt(0) would sit out, and no single wave 64 pieces
// behind at the barrier); neutral for the 16-bit eval kernels, and a loss for every training kernel (their stores)
template <typename PL> struct StreamShared { typedef WStreamT<PL::CH, PL::NBUF, PL::WAVES, false, false> type; };
template <typename WS> __device__ __forceinline__ void stream_counted(WS &w, bool on) { w.counted = on; }
template <int CH> __device__ __forceinline__ void stream_counted(WStreamSkew<CH> &, bool) {}
__device__ __forceinline__ void stream_begin(...) {}
__device__ __forceinline__ void stream_end(...) {}
template <int CH> __device__ __forceinline__ void stream_begin(WStreamSkew<CH> &w) { w.begin(); }
template <int CH> __device__ __forceinline__ void stream_end(WStreamSkew<CH> &w) { w.end(); }

// max(x, lo) with lo = 0 (ReLU) or -inf (none).  The fp16 kernels use the form that passes NaN on (v_max_f32 returns
// the non-NaN operand) and maps -0.0 to +0.0, so that the result is +0, a positive number or NaN -- nothing else.
// (The one-instruction alternative, a signed-integer max of the bit patterns against 0, was measured: the NaN an MFMA
// produces for inf - inf has its sign bit SET and would be squashed to 0 -- tests/test_hip_bf16.py::test_fp16x3_range.)
template <bool NAN_THROUGH>
__device__ __forceinline__ float relu_bound(bool relu) { return relu ? 0.f : -INFINITY; }
template <bool NAN_THROUGH>
__device__ __forceinline__ float relu_lo(float x, float lo) { return NAN_THROUGH ? (x <= lo ? lo : x) : fmaxf(x, lo); }
// The same bound as a signed-integer max of the bit pattern (lo_i = 0: ReLU, INT_MIN: none): ONE instruction, +0 | positive | NaN like
// relu_lo<true> for every input but a NaN whose sign bit is set (-> +0).  For the arithmetics with fp32's exponent range (exact fp32,
// bf16), where no intermediate overflows into inf - inf; the fp16 fragments keep relu_lo<true> (their range test relies on it).
__device__ __forceinline__ float relu_bits(float x, int lo_i) {
    const int b = __float_as_int(x);
    return __int_as_float(b > lo_i ? b : lo_i);
}
// 1 if y > 0, for a y that went through relu_lo<true>(., 0) (+0, positive or NaN: "bits != 0").  v_min_u32 + shifts
// instead of v_cmp + v_cndmask + or: on gfx950 a VALU that reads VCC needs two wait states behind the v_cmp that wrote
// it, ~3.5 issue slots per element against 2.  Inline asm: the compiler turns umin(b, 1) back into the compare form.
__device__ __forceinline__ unsigned positive_bit(float y) {
    unsigned r;
    asm("v_min_u32 %0, 1, %1" : "=v"(r) : "v"(__float_as_uint(y)));
    return r;
}
// the four ReLU bits of a C/D tile (bit r: element r > 0), Horner form on v_lshl_or_b32 (the compiler expands
// (m << 1) | b into a shift and an or)
__device__ __forceinline__ unsigned positive_nibble(const floatx4 &y) {
    unsigned m = positive_bit(y[3]);
#pragma unroll
    for (int r = 2; r >= 0; --r) {
        const unsigned b = positive_bit(y[r]);
        asm("v_lshl_or_b32 %0, %1, 1, %2" : "=v"(m) : "v"(m), "v"(b));
    }
    return m;
}
// append a tile's nibble to a mask word from the top: after eight tiles the first one sits in bits 0..3
__device__ __forceinline__ unsigned push_nibble(unsigned word, unsigned nib) { return __builtin_amdgcn_alignbit(nib, word, 4); }

// C/D tile T of a narrow fragment <-> row-major [P][ld]: lane (j, q) owns features 16T + 4q .. +3.
// The per-point stores of the narrow kernels are NOT predicated on the point being in range: a lane beyond P works on
// point P - 1 (clamped index, same inputs, same arithmetic) and so writes the very bytes the lane that owns P - 1 writes
// -- a benign duplicate in place of a branch around each of the 16 tile stores of a layer.
__device__ __forceinline__ void store_tile16(float *base, int64_t p, int ld, int q, int T, const floatx4 &v) {
#if defined(VN_EXP) && VN_EXP == 1
    return;                                   // timing experiment only: no activation / gradient tile stores
#endif
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 val = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(val, (f4 *)(base + (size_t)p * ld + 16 * T + 4 * q));
}
__device__ __forceinline__ floatx4 load_tile16(const float *base, int64_t p, int ld, int q, int T) {
    const float4 f = *(const float4 *)(base + (size_t)p * ld + 16 * T + 4 * q);
    floatx4 v = {f.x, f.y, f.z, f.w};
    return v;
}
// VN_F16_PRESPLIT (default; 0 = plain fp32 storage): FP16X3 stores the trunk activations / gradients that only the
// 256x256 weight-gradient GEMMs read back already split -- hi and lo fp16 parts in the bytes of the fp32 values they
// replace (store_pair_split) -- and those GEMMs stage with a v_perm gather instead of ~320 VALU slots of conversion per
// 32-point block (k_wgrad_split16_256).  On its own this measured within noise (weight gradients 5.37 vs 5.46 ms per
// step); it pays together with two things it makes possible: the stores leave from the next layer's stages
// (VN_DEFER_STORES below) and the weight-gradient kernel hides its now cheap staging under its MFMAs (docs/HISTORY.md 4.3):
// 15.2 -> 14.3 ms per step.  (A first version with two separate [P][256]-half planes was clearly slower: 8-byte stores.)
// build switch VN_F16_PRESPLIT (default 1, vipnerf_knobs.h)
// When the stored form of a layer's output IS the next GEMM's B operand (the fp16 parts: FP16X3H, VN_F16_PRESPLIT), the
// stores need not leave in a burst at the layer's epilogue: the operand registers stay live through the whole next layer,
// so each of its weight stages sends a quarter of them -- in the shadow of the other wave's MFMAs.
// build switch VN_DEFER_STORES (default 1, vipnerf_knobs.h)
// VIPNERF_PREC_BF16 (single bf16 MFMA per product): 1 = the trunk activations h_1..h_8 and the gradients dY_1..dY_7, dY_feature are
// stored as the bf16 operands the next GEMM consumes anyway (2 bytes per value, like VIPNERF_PREC_FP16; the 256x256 weight-gradient
// GEMMs then run ONE bf16 MFMA per product on them); 0 = round 2's fp32 storage + bf16 hi/lo weight gradients.
// build switch VN_BF16_H16 (default 1, vipnerf_knobs.h)
// precisions whose 256-wide trunk activations / gradients are stored as 16-bit high parts only ([P][256] halves in the fp32 slot)
// VN_T16 (default): in the single-MFMA modes EVERY operand of the weight-gradient GEMMs -- h_1..h_8, the feature, the view hidden per
// direction, gamma(x), gamma(dir), dY_0..dY_7, dY_feature, dYv per direction, their sum, the head seeds -- is stored as 16-bit values in
// the tile-blocked layout T16 (store_t16 below), which the weight-gradient kernels DMA straight into LDS and read with the hardware
// transpose (vipnerf_wgrad16.hip); 0 = round 2's storage (row-major [P][256] halves for the 256x256 GEMMs, fp32 for everything else).
// build switch VN_T16 (default 1, vipnerf_knobs.h)
__host__ __device__ inline bool stores_t16(int precision) {
    return VN_T16 && (precision == VIPNERF_PREC_FP16 || precision == VIPNERF_PREC_FP16X3H || (precision == VIPNERF_PREC_BF16 && VN_BF16_H16));
}
// ... of which the single-MFMA ones run the two-point-tile MLP kernels (vipnerf_mlp_pt2.h); FP16X3H keeps the 16-point fp16x3 kernels
__host__ __device__ inline bool single_mfma_t16(int precision) { return stores_t16(precision) && precision != VIPNERF_PREC_FP16X3H; }
__host__ __device__ inline bool stores_high16(int precision) {
    if (stores_t16(precision)) return false;
    return precision == VIPNERF_PREC_FP16X3H || precision == VIPNERF_PREC_FP16 || (precision == VIPNERF_PREC_BF16 && VN_BF16_H16);
}
// FP16X3H: the fp16 high parts of a B fragment (k-step s <- tiles 2s, 2s+1) ARE the fp16 image of those two tiles:
// elements 4u .. 4u+3 of part 0 are features 16 (2s+u) + 4q .. +3.  Stored as [P][ld] halves (8 bytes per lane, tile).
__device__ __forceinline__ void store_pair16h(float *base, int64_t p, int ld, int q, int s, const half8 &hi) {
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    _Float16 *row = (_Float16 *)base + (size_t)p * ld + 4 * q;
    const half4 a = {hi[0], hi[1], hi[2], hi[3]}, b = {hi[4], hi[5], hi[6], hi[7]};
    __builtin_nontemporal_store(a, (half4 *)(row + 16 * (2 * s)));
    __builtin_nontemporal_store(b, (half4 *)(row + 16 * (2 * s + 1)));
}
// the same for bf16 high parts (VIPNERF_PREC_BF16: the rounded operand of the next GEMM is what is stored)
__device__ __forceinline__ void store_pair16h(float *base, int64_t p, int ld, int q, int s, const bf16x8 &hi) {
    typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
    __bf16 *row = (__bf16 *)base + (size_t)p * ld + 4 * q;
    const bf4 a = {hi[0], hi[1], hi[2], hi[3]}, b = {hi[4], hi[5], hi[6], hi[7]};
    __builtin_nontemporal_store(a, (bf4 *)(row + 16 * (2 * s)));
    __builtin_nontemporal_store(b, (bf4 *)(row + 16 * (2 * s + 1)));
}
__device__ __forceinline__ void store_pair16h(float *, int64_t, int, int, int, const f32q &) {}
// VN_F16_PRESPLIT: both fp16 parts of the two tiles, in the fp32 array's own geometry: the 16 bytes a lane owns per tile
// (4 features) hold [hi(f0,f1)] [hi(f2,f3)] [lo(f0,f1)] [lo(f2,f3)] -- the registers of the split as they are, one 16-byte
// store per tile exactly like the fp32 store.
__device__ __forceinline__ void store_pair_split(float *base, int64_t p, int ld, int q, int s, const half8 &hi, const half8 &lo) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 h = __builtin_bit_cast(u4, hi), l = __builtin_bit_cast(u4, lo);
    float *row = base + (size_t)p * ld + 4 * q;
    const u4 t0 = {h[0], h[1], l[0], l[1]}, t1 = {h[2], h[3], l[2], l[3]};
    __builtin_nontemporal_store(t0, (u4 *)(row + 16 * (2 * s)));
    __builtin_nontemporal_store(t1, (u4 *)(row + 16 * (2 * s + 1)));
}
__device__ __forceinline__ void store_pair_split(float *, int64_t, int, int, int, const bf16x8 &, const bf16x8 &) {}
__device__ __forceinline__ void store_pair_split(float *, int64_t, int, int, int, const f32q &, const f32q &) {}

// H16 == 3 (exact fp32, f32q operands): the k-step operand of the next GEMM IS the fp32 activation (part 0 = tile 2s, part 1 =
// tile 2s + 1), so its plain fp32 stores can leave from the next layer's stages exactly like the pre-split fp16 ones
__device__ __forceinline__ void store_pair_f32(float *base, int64_t p, int ld, int q, int s, const f32q &t0, const f32q &t1) {
    store_tile16(base, p, ld, q, 2 * s, t0.v);
    store_tile16(base, p, ld, q, 2 * s + 1, t1.v);
}
__device__ __forceinline__ void store_pair_f32(float *, int64_t, int, int, int, const bf16x8 &, const bf16x8 &) {}
__device__ __forceinline__ void store_pair_f32(float *, int64_t, int, int, int, const half8 &, const half8 &) {}

// T16 (H16 == 4): an array of `tiles` 16-feature tiles is [P / 16 groups][tiles][16 points][16 features] of 16-bit values -- every
// (group, tile) a dense row-major 16 x 16 matrix of 512 bytes.  A wave's 16 points are one group; the part-0 B fragment of k-step s
// (elements 0..3 = features 4q..4q+3 of tile 2s, 4..7 = the same of tile 2s + 1, for point j) is two 8-byte stores per lane, and each
// store instruction of the wave writes one whole tile = 512 contiguous bytes (no partially written lines).  The weight-gradient
// kernels copy 32-point blocks of these arrays into LDS by DMA and read their MFMA fragments (lane = feature, 4 consecutive points)
// with ds_read_b64_tr_b16.  Wave-uniformly predicated by the caller (P is a multiple of 16 in the render path).
// VN_T16_X4 (default): ONE 16-byte store per lane and k-step instead of two 8-byte ones.  The 32 features of a k-step are then laid out
// as two 16 x 16 tiles BY LANE GROUP -- stored tile 2s + (q >> 1), row j, columns 8 (q & 1) + e -- rather than by C/D tile: a fixed
// permutation of the feature index inside every 32-feature block (t16_feature below), which the weight-gradient kernels never see (any
// consistent order of an operand's features is a valid GEMM) and the chunk reduction undoes on the way into the nn.Linear layout.  Each
// store instruction of the wave then writes two whole tiles = 1 KiB contiguous, and a layer costs 8 store instructions per wave and
// point tile instead of 16.
// build switch VN_T16_X4 (default 1, vipnerf_knobs.h)
// build switch VN_T16_NT (default 1, vipnerf_knobs.h): nontemporal tile stores (the data is next read by another kernel, GBs later); 0: plain stores
constexpr int T16_SPK = VN_T16_X4 ? 1 : 2;       // store instructions store_t16 issues per k-step (what the counted stream waits assume)
// stored feature index i (tile i >> 4, column i & 15) of a T16 array written from C/D fragments -> the feature it holds
__host__ __device__ inline int t16_feature(int i) {
    if (!VN_T16_X4) return i;
    const int T = i >> 4, c = i & 15, s = T >> 1, q = 2 * (T & 1) + (c >> 3), e = c & 7;
    return 32 * s + 16 * (e >> 2) + 4 * q + (e & 3);
}
template <typename FR>
__device__ __forceinline__ void store_t16(float *base, int64_t grp, int tiles, int s, int j, int q, const FR &v) {
    static_assert(sizeof(FR) == 16, "a 16-byte fragment part");
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const u4 w = __builtin_bit_cast(u4, v);
    if (VN_T16_X4) {
        u4 *dst = (u4 *)((char *)base + ((size_t)grp * tiles + 2 * s + (q >> 1)) * 512 + j * 32 + (q & 1) * 16);
        if (VN_T16_NT) __builtin_nontemporal_store(w, dst);
        else *dst = w;
        return;
    }
    char *t0 = (char *)base + ((size_t)grp * tiles + 2 * s) * 512 + j * 32 + q * 8;
    const u2 lo = {w[0], w[1]}, hi = {w[2], w[3]};
    __builtin_nontemporal_store(lo, (u2 *)t0);
    __builtin_nontemporal_store(hi, (u2 *)(t0 + 512));
}
// two C/D tiles (2s, 2s+1) -> the NS-part B fragment of k-step s
template <int NS, typename FR>
__device__ __forceinline__ void split_pair(const floatx4 &lo, const floatx4 &hi, FR (&out)[NS]) {
    const float xs[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    split8<NS>(xs, out);
}
// The deferred stores of one weight stage (VN_DEFER_STORES), sent from INSIDE the stage: the four stores of the waves
// 0..3 behind MFMA group GA, those of the waves 4..7 behind group GB.  Wave w and wave w + 4 share a SIMD: while one of
// them queues at the vector-memory port the other keeps the SIMD's MFMA pipe busy -- at the stage's end all eight queue
// there together with nothing left to compute (and the first groups belong to the DMA burst).
// build switch VN_STORE_GROUP_A (default 6, vipnerf_knobs.h)
// build switch VN_STORE_GROUP_B (default 12, vipnerf_knobs.h)
template <int H16, int NS, typename FR, int NSTEP = 2>
struct DeferredStores {
    float *dst; int64_t p; int q, wave, s0;
    const FR (*bin)[NS];
    int64_t grp; int j; bool valid;         // H16 == 4 (T16): the wave's 16-point group, the lane's point in it, group in range
    // the 8-byte stores of the high-parts-only storage (H16 = 1, 4) measured better all together behind the last group
    template <int g, int NG> static constexpr bool active() { return (H16 == 1 || H16 == 4) ? g == NG - 1 : (g == VN_STORE_GROUP_A || g == VN_STORE_GROUP_B); }
    template <int g, int NG>
    __device__ __forceinline__ void at() const {
        if (EXP_NO_STORES) return;
        if (H16 == 4) {
            if (valid) {
#pragma unroll
                for (int s = s0; s < s0 + NSTEP; ++s) store_t16(dst, grp, 16, s, j, q, bin[s][0]);
            }
            return;
        }
        if (H16 == 1 || (g == VN_STORE_GROUP_A) == (wave < 4)) {
#pragma unroll
            for (int s = s0; s < s0 + NSTEP; ++s) {
                if (H16 == 1) store_pair16h(dst, p, 256, q, s, bin[s][0]);
                if (H16 == 2) store_pair_split(dst, p, 256, q, s, bin[s][0], bin[s][NS > 1 ? 1 : 0]);
                if (H16 == 3) store_pair_f32(dst, p, 256, q, s, bin[s][0], bin[s][NS > 1 ? 1 : 0]);
            }
        }
    }
};

// ---- positional encodings in the narrow layout's slot order (forward kernels)
__device__ __forceinline__ float pow2f(int l) { return __uint_as_float((unsigned)(127 + l) << 23); }

// gamma(x) in the slot order of pe_feat16 (vipnerf_bf16n.h): lane group q evaluates levels (5q)>>1 .. +2 only.
// out[s][e], u = 8s + e: u = 3 gg + d < 15 -> triple gg of this lane group, component d; u = 15 -> x[q] / unused
// sin / cos for the encodings of the SINGLE-MFMA 16-bit kernels (build switch VN_PT2_FAST_PE): the hardware's v_sin_f32 / v_cos_f32 on the
// argument in revolutions, reduced with v_fract first (their input range is +-256 revolutions; 2^9 x of a non-NDC scene exceeds it) --
// 5 instructions and no branch against sincosf's ~40 with two range-reduction branches.  Error: the fp32 product x 2^l / (2 pi) carries
// 2^-24 relative, i.e. <= 5e-5 rad at 2^9 |x| = 768, plus the instructions' own ~1e-6: a fifth of ONE fp16 rounding of the encoded value
// (2.4e-4), a fortieth of a bf16 one -- below the accuracy class of the modes that use it.  Never used by the fp32-grade arithmetics.
__device__ __forceinline__ void sincos_rev(float x, float *s, float *c) {
    const float r = __builtin_amdgcn_fractf(x * 0.15915494309189535f);
    *s = __builtin_amdgcn_sinf(r);
    *c = __builtin_amdgcn_cosf(r);
}
template <bool FAST = false>
__device__ __forceinline__ void encode_x16(const float v[3], int q, float (&out)[2][8]) {
    const int lb = (5 * q) >> 1;
    float S[3][3], C[3][3];
#pragma unroll
    for (int li = 0; li < 3; ++li) {
        const float f = pow2f(lb + li);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
#if defined(VN_EXP) && VN_EXP == 7
            S[li][d] = v[d] * f; C[li][d] = S[li][d] + 1.f;   // timing experiment only: no sincos
#else
            if (FAST) sincos_rev(v[d] * f, &S[li][d], &C[li][d]);
            else sincosf(v[d] * f, &S[li][d], &C[li][d]);
#endif
        }
    }
    const bool odd = q & 1;           // 5q even: triples are S0 C0 S1 C1 S2;  odd: C0 S1 C1 S2 C2
    float val[16];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        val[0 + d] = odd ? C[0][d] : S[0][d];
        val[3 + d] = odd ? S[1][d] : C[0][d];
        val[6 + d] = odd ? C[1][d] : S[1][d];
        val[9 + d] = odd ? S[2][d] : C[1][d];
        val[12 + d] = odd ? C[2][d] : S[2][d];
    }
    val[15] = q == 0 ? v[0] : (q == 1 ? v[1] : (q == 2 ? v[2] : 0.f));
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) out[s][e] = val[8 * s + e];
}
// gamma(dir) in the slot order of dir_feat16: lane group q evaluates level q only
template <bool FAST = false>
__device__ __forceinline__ void encode_d16(const float v[3], int q, float (&out)[1][8]) {
    const float f = pow2f(q);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (FAST) sincos_rev(v[d] * f, &out[0][d], &out[0][3 + d]);
        else sincosf(v[d] * f, &out[0][d], &out[0][3 + d]);
    }
    out[0][6] = q == 0 ? v[0] : (q == 1 ? v[2] : 0.f);
    out[0][7] = q == 0 ? v[1] : 0.f;
}
// natural-order rows of the activation store ([P][64] / [P][32], read by the weight-gradient GEMMs)
__device__ __forceinline__ void store_x16(float *row, int q, const float (&pe)[2][8]) {
#pragma unroll
    for (int u = 0; u < 15; ++u) row[3 + 15 * q + u] = pe[u >> 3][u & 7];
    row[q < 3 ? q : DPE] = pe[1][7];                       // x[q]; lane group 3 writes the zero pad column
}
__device__ __forceinline__ void store_d16(float *row, int q, const float (&pd)[1][8]) {
#pragma unroll
    for (int e = 0; e < 6; ++e) row[3 + 6 * q + e] = pd[0][e];
    if (q == 0) { row[0] = pd[0][6]; row[1] = pd[0][7]; }
    else if (q == 1) row[2] = pd[0][6];
    else if (q == 2) { row[DVE] = 0.f; row[DVE + 1] = 0.f; }
    else { row[DVE + 2] = 0.f; row[DVE + 3] = 0.f; row[DVE + 4] = 0.f; }
}

template <bool F16, bool F32 = false> struct FragOf { typedef bf16x8 type; };
template <> struct FragOf<true, false> { typedef half8 type; };
template <> struct FragOf<false, true> { typedef f32q type; };
#endif

// workgroups of a persistent launch of the 160 KB-of-LDS MLP kernels: one per CU of the current device (queried once)
int persistent_grid();
int launch_pack_bf16n(const vipnerf_mlp_params *p, int precision, void *packed_bn, hipStream_t st, const vipnerf_mlp_params *p2 = nullptr, void *packed2 = nullptr);   // precision 0 (fp32 narrow) .. 4

}  // namespace vn
