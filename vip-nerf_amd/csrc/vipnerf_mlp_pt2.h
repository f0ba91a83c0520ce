// The single-MFMA 16-bit MLP kernels with two point tiles per wave (vipnerf_mlp_fwd_pt2.hip, vipnerf_mlp_bwd_pt2.hip): shared pieces.
#pragma once
#include "vipnerf_bf16n.h"
#include "vipnerf_mlp.h"


namespace vn {

constexpr int PT2_PTS_PER_WG = 256;       // 8 waves x 2 point tiles x 16 points

#if defined(__HIPCC__)
// The deferred T16 stores of one weight stage, both point tiles: the part-0 B fragments of NSTEP k-steps of the layer input (= the
// previous layer's output, or the gradient the running GEMM consumes), behind the stage's last MFMA group.  2 T16_SPK NSTEP store
// instructions per wave when both tiles are in range (what the counted waits of the stream assume; WStreamT::counted otherwise).
template <typename FR, int NSTEP, int TILES = 16>       // TILES: width of the stored array in 16-feature tiles
struct DeferredT16 {
    float *dst;
    int64_t grp[2];
    bool valid[2];
    int j, q, s0;
    const BOp<FR, 2> (*bin)[1];
    int phase = 0;                                       // VN_PT2_SKEW: this wave's slot among the NPH store phases (pt2_store_phase)
    // VN_PT2_SPREAD = 1: all of a stage's stores behind its last MFMA group; = NSTEP (4, the default): one k-step's stores (4 instructions)
    // behind every NG / NSTEP-th group, so that the stage's store instructions do not queue at the vector-memory port at once --
    // measured on one box (bf16, 4096 rays): forward 1.66 -> 1.60 ms, data gradients 1.73 -> 1.61; 8 parts (one point tile's k-step each):
    // forward 1.63, data gradients 2.15 (the extra scheduling barriers cost registers: spills)
    // (build switch VN_PT2_SPREAD, default 4, vipnerf_knobs.h; = 2 NSTEP: one k-step of ONE point tile -- two instructions -- behind every
    // NG / (2 NSTEP)-th group)
    static constexpr int PARTS = VN_PT2_SPREAD > 1 ? (VN_PT2_SPREAD >= 2 * NSTEP ? 2 * NSTEP : NSTEP) : 1;
    // VN_PT2_SKEW = n in {2, 4, 8}: the eight waves send a part's stores in n phases, 1 / n of the distance between two parts apart, SIMD
    // partners (waves w, w + 4) half a distance apart: while one wave of a SIMD queues at the vector-memory port the other issues MFMAs,
    // and the workgroup's stores reach the memory pipeline as a stream instead of 16 KiB bursts.  All of a stage's stores still leave
    // inside the stage and behind its first group (the next stage's DMA): the counted waits of the stream hold as they are.
    static constexpr int NPH = VN_PT2_SKEW > 1 ? VN_PT2_SKEW : 1;
    template <int g, int NG> static constexpr bool active() {
        static_assert((NG / PARTS) % NPH == 0, "store phases must divide the distance between two parts");
        return (g + 1) % (NG / PARTS / NPH) == 0;
    }
    template <int g, int NG>
    __device__ __forceinline__ void at() const {
        if (EXP_NO_STORES) return;
        constexpr int idx = (g + 1) / (NG / PARTS / NPH) - 1;       // 0 .. PARTS NPH - 1
        constexpr int part = idx / NPH, ph = idx % NPH;
        if (NPH > 1 && ph != phase) return;
        constexpr bool split_pt = PARTS == 2 * NSTEP;
        constexpr int per = PARTS == 1 ? NSTEP : 1, sfirst = PARTS == 1 ? 0 : (split_pt ? part / 2 : part);
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            if (split_pt && pt != (part & 1)) continue;
            if (valid[pt]) {
#pragma unroll
                for (int s = s0 + sfirst; s < s0 + sfirst + per; ++s) store_t16(dst, grp[pt], TILES, s, j, q, bin[s][0].v[pt]);
            }
        }
    }
};
// the store phase of wave w (scalar): SIMD partners w, w + 4 are NPH / 2 phases apart, the last phase sits where the unskewed stores do
__device__ __forceinline__ int pt2_store_phase(int wave) {
    constexpr int NPH = VN_PT2_SKEW > 1 ? VN_PT2_SKEW : 1;
    if (NPH == 1) return 0;
    const int w = __builtin_amdgcn_readfirstlane(wave);
    return (NPH - 1) - (((w >> 2) * (NPH / 2) + (w & (NPH / 2 - 1))) % NPH);
}

// VN_EXP == 50 (timing experiment: tools/pt2_timeline.py): one workgroup in the middle of the grid records s_memtime at the kernel's phase
// boundaries, per wave, with a tag in the top byte (TS_ENTRY ...), into a device array that vipnerf_exp_timeline() / _bwd() copy out.
// TS(tag) is nothing in every other build.
enum { TS_ENTRY = 0, TS_RESIDENT = 1, TS_HEAD = 2, TS_PRE = 3, TS_POST = 4, TS_END = 5, TS_VIEW = 6, TS_LAST = 7, TS_EPI_A = 8, TS_EPI_B = 9 };
#if defined(VN_EXP) && VN_EXP == 50
#define TS_DECL(buf) __device__ unsigned long long buf[8 * 256]
#define TS_INIT() const bool ts_rec = blockIdx.x == gridDim.x / 2 && lane == 0; int ts_n = 0
#define TS_AT(buf, tag) do { if (ts_rec) { buf[wave * 256 + (ts_n < 255 ? ts_n : 255)] = ((unsigned long long)(tag) << 56) | (__builtin_readcyclecounter() & 0x00ffffffffffffffull); ++ts_n; } } while (0)
#else
#define TS_DECL(buf)
#define TS_INIT() do { } while (0)
#define TS_AT(buf, tag) do { } while (0)
#endif

// ReLU with a runtime bound (0, or "none") as the two-point-tile kernels' epilogues evaluate it.  fp16 fragments: the compare-and-select
// form of relu_lo<true> -- NaN (of either sign: the MFMA's inf - inf has its sign bit set) passes through, out-of-range inputs fail loudly
// (tests/test_hip_bf16.py::test_fp16x3_range).  bf16 fragments (fp32's exponent range: nothing overflows on the way): a signed-integer
// max of the bit pattern against lo_i = 0 / INT_MIN -- ONE instruction instead of v_cmp + two wait states + v_cndmask, the same result for
// every number (negative and -0 -> +0, the rest unchanged) and for NaNs with a clear sign bit; a NaN with its sign bit set becomes +0.
template <bool F16>
__device__ __forceinline__ float relu_pt2(float x, float lo, int lo_i) {
    if (F16) return relu_lo<true>(x, lo);
    const int b = __float_as_int(x);
    return __int_as_float(b > lo_i ? b : lo_i);
}

// bf16 fragments, the trunk's epilogue: the ReLU AFTER the conversion, on the packed halves -- v_pk_max_i16 against lo16 = 0 / 0x8000 per half
// (ReLU / none), one instruction per TWO values where relu_pt2 + v_cvt_pk is three per two; bf16(relu(x)) == relu(bf16(x)) for every x
// (round-to-nearest-even keeps the sign, -0 -> +0 either way).  A wave's VALU instruction is 4 cycles of its SIMD, and the epilogue of
// the SIMD's LATER wave runs with the MFMA pipe idle (profiles/r04_ablation_pt2.md 5): its instruction count is kernel time.
__device__ __forceinline__ unsigned relu_pk16(unsigned w, unsigned lo16) {
    unsigned r;
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(w), "s"(lo16));
    return r;
}
// The ReLU bits of the eight packed values of a k-step operand that went through relu_pk16(., 0) (halves +0 | positive | NaN): bits
// 0..3 = elements 0..3 > 0 (C/D tile 2s), 4..7 = elements 4..7 (tile 2s + 1) for the even elements' bits IN PLACE (0, 2, 4, 6) and the
// odd elements' 16 bits higher -- fold_pk_bits() of two of them brings the halves together.  v_pk_min_u16 against 1 per half: two
// values per instruction (positive_bit: one).
__device__ __forceinline__ unsigned positive_pk_bits(const unsigned (&w)[4], unsigned one2) {
    unsigned m[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) asm("v_pk_min_u16 %0, %1, %2" : "=v"(m[i]) : "v"(w[i]), "s"(one2));
    unsigned ta, tb, c;
    asm("v_lshl_or_b32 %0, %1, 2, %2" : "=v"(ta) : "v"(m[1]), "v"(m[0]));
    asm("v_lshl_or_b32 %0, %1, 2, %2" : "=v"(tb) : "v"(m[3]), "v"(m[2]));
    asm("v_lshl_or_b32 %0, %1, 4, %2" : "=v"(c) : "v"(tb), "v"(ta));
    return c;
}
// k-steps s, s + 1 (s even) -> the 16 mask bits of their four tiles in push_nibble's order (tile t's nibble at 4 (t & 7))
__device__ __forceinline__ unsigned fold_pk_bits(unsigned c0, unsigned c1) {
    unsigned a;
    asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(a) : "v"(c1), "v"(c0));
    return (a | (a >> 15)) & 0xffffu;
}

// The inverse for the data-gradient pass: 16 stored ReLU bits (four tiles = two k-steps, push_nibble's order) -> even bits in place, odd
// bits 15 higher, so that (a >> 2 i) & 0x00010001 holds packed pair i's two bits (elements 2i, 2i + 1 of the first k-step; i + 4: of
// the second) one per half -- and v_pk_mul_lo_u16 by it keeps or zeroes the two converted gradients in ONE instruction.
__device__ __forceinline__ unsigned unfold_pk_bits(unsigned m16) { return (m16 & 0x5555u) | ((m16 & 0xaaaau) << 15); }
// packed pair w (two 16-bit gradients) times its two ReLU bits: a = unfold_pk_bits(.), i = the pair's index in it (0..7)
__device__ __forceinline__ unsigned mask_pk16(unsigned w, unsigned a, int i, unsigned one2) {
    unsigned b, r;
    asm("v_and_b32 %0, %1, %2" : "=v"(b) : "s"(one2), "v"(a >> (2 * i)));
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(w), "v"(b));
    return r;
}

#endif

int launch_mlp_fwd_pt2(const MlpFwdArgs &a, int precision, hipStream_t st);
int launch_mlp_bwd_pt2(const MlpBwdArgs &a, int precision, hipStream_t st);

}  // namespace vn
