// Split-precision bf16 MFMA path ("bf16xN"): every fp32 operand x is written as a short sum of bf16 numbers,
// x ~= x0 + x1 (+ x2), x0 = bf16(x), x1 = bf16(x - x0), ..., and a product a*b is accumulated in fp32 from the
// significant cross terms on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate):
//   NS = 2 ("bf16x3"): a0b0 + a0b1 + a1b0              3 MFMAs, per-product error <= ~3 * 2^-18  (~1e-5)
//   NS = 3 ("bf16x6"): a0b0 + a0b1 + a1b0 + a0b2 + a2b0 + a1b1   6 MFMAs, error ~2^-26 (fp32 grade)
// Everything else -- transposed register-chained layers, weights streamed through LDS in fragment order -- is the
// fp32 design (vipnerf_common.h) with a different fragment: lane l supplies A[i = l&31][k = 8*(l>>5) + e] and
// B[k = 8*(l>>5) + e][j = l&31], e = 0..7, 16 k-values per MFMA; C/D layout is identical.  A C/D tile of the
// previous layer (16 registers per lane) therefore feeds two k-steps: registers 8u..8u+7 of tile T are the 8
// k-values of k-step s = 2T + u for this lane's half, i.e. the contraction index is feat(16T + 8u + e, h).
#pragma once
#include "vipnerf_common.h"

namespace vn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#if defined(__HIPCC__)
// part i of the split of x (i = 0: bf16(x); 1: bf16(x - x0); 2: bf16(x - x0 - x1)), round-to-nearest-even
__device__ __forceinline__ __bf16 split_part(float x, int i) {
    __bf16 p = (__bf16)x;
    for (int k = 0; k < i; ++k) { x = x - (float)p; p = (__bf16)x; }
    return p;
}
template <int NS>
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 (&out)[NS]) {
#if defined(VN_EXP) && VN_EXP == 6
    if (x[0] != 12345.f) return;                  // timing experiment only: no operand split
#endif
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float r = x[e];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const __bf16 p = (__bf16)r;
            out[i][e] = p;
            r = r - (float)p;
        }
    }
}

typedef float floatx4 __attribute__((ext_vector_type(4)));
// fp16 fragments ("fp16x3", narrow layout only): same fragment shapes, 11-bit parts instead of 8-bit ones
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
// build switch VN_SPLIT_FMA_MIX (default 1, vipnerf_knobs.h)
template <int NS>
__device__ __forceinline__ void split8(const float (&x)[8], half8 (&out)[NS]) {
    if constexpr (NS == 2 && VN_SPLIT_FMA_MIX) {
        // hi = fp16(x) (v_cvt_pk_f16_f32, two values per instruction); lo = fp16(x - hi) as ONE v_fma_mix{lo,hi}_f16 per value
        // (fp16 source half, -1.0, fp32 addend; exact difference, one rounding, written straight into its half of the packed
        // register) instead of v_cvt_f32_f16 + v_sub_f32 + half a v_cvt_pk: the same bits, 3 instead of 5 VALU slots per pair
        typedef _Float16 half2_ __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
            const half2_ h = {(_Float16)x[e], (_Float16)x[e + 1]};
            const unsigned hb = __builtin_bit_cast(unsigned, h);
            unsigned lb;
            asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hb), "v"(x[e]));
            asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lb) : "v"(hb), "v"(x[e + 1]));
            const half2_ l = __builtin_bit_cast(half2_, lb);
            out[0][e] = h[0]; out[0][e + 1] = h[1];
            out[1][e] = l[0]; out[1][e + 1] = l[1];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float r = x[e];
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const _Float16 p = (_Float16)r;
                out[i][e] = p;
                r = r - (float)p;
            }
        }
    }
}
__device__ __forceinline__ floatx4 mfma_bf(half8 a, half8 b, floatx4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// 16x16x32: lane l supplies A[i = l&15][k = 8*(l>>4) + e] and B[k = 8*(l>>4) + e][j = l&15]; D[i = 4*(l>>4) + r][j = l&15]
__device__ __forceinline__ floatx4 mfma_bf(bf16x8 a, bf16x8 b, floatx4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// Exact-fp32 fragments for the narrow layout ("fp32 narrow": two waves per SIMD on v_mfma_f32_16x16x4_f32).  A "k-step"
// of the narrow layout is 32 contraction indices, 8 per lane group q; in fp32 they are consumed by 8 consecutive MFMAs
// (k = 4 each: lane (., q) supplies ONE float, the index (s, q, e) of MFMA e), so a lane's k-step operand is 8 floats =
// two 16-byte parts -- the same bytes per lane, chunk geometry, stage sizes and register-chained layer structure as the
// two-part fp16 / bf16 fragments, without any operand split and with one MFMA per product.
struct f32q { floatx4 v; };
static_assert(sizeof(f32q) == 16, "one 16-byte part");
template <int NS>
__device__ __forceinline__ void split8(const float (&x)[8], f32q (&out)[NS]) {
    static_assert(NS == 2, "fp32 narrow fragments have two 4-float parts");
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e >> 2].v[e & 3] = x[e];
}
// NPT point tiles of one wave that share every A fragment (the 32-point single-MFMA layout, vipnerf_mlp_pt2.h): the B operand of a
// k-step is NPT fragments, an accumulator tile NPT C/D tiles, and a cell is NPT MFMAs per A-fragment read
template <typename FR, int NPT> struct BOp { FR v[NPT]; };
template <int NPT> struct AccN { floatx4 v[NPT]; };
template <typename T> struct FragTypeOf { typedef T type; static constexpr int npt = 1; };
template <typename FR, int NPT> struct FragTypeOf<BOp<FR, NPT>> { typedef FR type; static constexpr int npt = NPT; };
// MFMAs per (k-step, tile) cell
template <typename FR, int NS> struct CellMfmas { static constexpr int value = NS * (NS + 1) / 2; };
template <int NS> struct CellMfmas<f32q, NS> { static constexpr int value = 4 * NS; };
template <int NS>
__device__ __forceinline__ floatx4 mfma_split(const f32q (&a)[NS], const f32q (&b)[NS], floatx4 c) {
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].v[e], b[i].v[e], c, 0, 0, 0);
    return c;
}

// acc[t] += A(k-step, tile) * B(k-step) over the significant cross terms; smallest terms first
template <int NS, typename ACC, typename FR>
__device__ __forceinline__ ACC mfma_split(const FR (&a)[NS], const FR (&b)[NS], ACC c) {
    if constexpr (NS == 3) {
        c = mfma_bf(a[1], b[1], c);
        c = mfma_bf(a[2], b[0], c);
        c = mfma_bf(a[0], b[2], c);
    }
    if constexpr (NS >= 2) {
        c = mfma_bf(a[1], b[0], c);
        c = mfma_bf(a[0], b[1], c);
    }
    c = mfma_bf(a[0], b[0], c);               // NS == 1: the single-MFMA 16-bit modes (VIPNERF_PREC_FP16 / BF16)
    return c;
}

template <int NS, typename FR, int NPT>
__device__ __forceinline__ AccN<NPT> mfma_split(const FR (&a)[NS], const BOp<FR, NPT> (&b)[NS], AccN<NPT> c) {
    static_assert(NS == 1, "point-tile pairs: single-MFMA modes");
#if defined(VN_EXP) && (VN_EXP == 43 || VN_EXP == 44)
    return c;                                     // timing experiment only: no MFMAs (what the stores and the weight stream cost alone)
#endif
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) c.v[pt] = mfma_bf(a[0], b[0].v[pt], c.v[pt]);
    return c;
}

// Weight stream: a ring of NBUF LDS stage buffers filled by LDS-DMA NBUF-1 stages ahead of the consumer.
//   wait():     counted s_waitcnt (this wave's DMA of the stage to consume has landed; with NBUF > 2 the younger
//               stages' stay in flight: VM_CNT retires in issue order, stores included) + raw s_barrier (every
//               wave's part has landed, and every wave is done reading the stage consumed before)
//   prefetch(): DMA of stage (current + NBUF-1) into the buffer that barrier has just freed
// ROTATE (narrow layout, two waves per SIMD): a stage's whole DMA is issued by ONE wave, a different one each stage.
// The texture path takes a 1 KiB piece every ~16 cycles whoever issues it; when every wave issues its share at the
// same time each of them is stuck ~60 cycles per piece and no MFMA issues meanwhile, whereas a single issuing wave
// leaves the other seven (including its SIMD partner) computing.
// STAGGER (narrow layout, -DVN_DMA_MODE=2): every wave issues its 1/WAVES share, wave w behind MFMA group w * NG / WAVES of the
// stage instead of all of them behind group 0 -- no single wave is ~60 cycles x CH behind the others at the stage barrier.
// ISSUERS (ROTATE only; build switch VN_DMA_ISSUERS for the narrow kernels): how many waves share a stage's DMA, CH / ISSUERS pieces each, all
// behind the first MFMA group; the issuer set advances by ISSUERS waves per stage.  Waves w and w + 4 share a SIMD, so with ISSUERS <= 4 no
// two issuers of a stage sit on one SIMD: each issuer's partner keeps that SIMD's MFMA pipe fed while it issues.  Why it matters: a
// global_load_lds piece costs its wave ~60 cycles of issue time among MFMAs (MI355X_MICROARCH.md), so ONE wave issuing a 64-piece stage
// is ~3800 cycles behind the other seven at the stage barrier -- as long as the stage's own MFMA work (4096 cycles per SIMD in the
// two-point-tile 16-bit kernels: profiles/r04_ablation_pt2.md measured stage = skeleton + MFMA, not max).
// ROTW (ROTATE only; build switch VN_DMA_ROT_WAVES for the narrow kernels): the issuer rotates over waves 0 .. ROTW - 1 only.  Waves w and w + 4 share a
// SIMD and the arbiter favours the older one (w < 4): it finishes a stage's MFMAs first and sits ~8k cycles at the stage barrier, while the
// younger wave IS the stage's critical path -- and a 64-piece DMA burst costs its issuer ~2k cycles (profiles/r05_timeline_bwd_f32_v2.log: 19.2k
// against 17.3k cycles for the stages a younger wave issued in).  With ROTW = 4 only the waves with slack issue.
template <int CH, int NBUF, int WAVES = 4, bool ROTATE = false, bool STAGGER = false, int ISSUERS = 1, int ROTW = WAVES>
struct WStreamT {
    const float *g;
    float *buf;
    int n_left;            // stages not yet requested
    int in_flight;         // requested, not yet consumed
    int cur, fill;         // ring slot to consume next / to fill next
    int lane, wave;
    int turn;              // ROTATE: the wave that issues the next stage's DMA
    int cturn;             // ROTATE: the wave that issued the stage about to be consumed
    const float *g0;       // PERIODIC streams (persistent kernels: the same n-stage image once per tile): the image's first stage ...
    int period, pos;       // ... its length in stages, and the position of the next stage to request in it (period 0: not periodic)
    bool counted;          // false: this wave's YOUNGER bounds do not hold (a wave whose points are out of range skips its predicated
                           // stores): it drains with vmcnt(0) instead
    static constexpr int SF = CH * CHUNK_F;
    static constexpr int PER_WAVE = ROTATE ? CH / ISSUERS : CH / WAVES;
    static_assert(CH % WAVES == 0 && (NBUF == 2 || PER_WAVE * (NBUF - 2) <= 63), "vmcnt is a 6-bit counter");
    static_assert(ISSUERS >= 1 && WAVES % ISSUERS == 0 && CH % ISSUERS == 0 && (ISSUERS == 1 || ROTATE), "issuer sets tile the waves");
    static_assert(ROTW >= ISSUERS && ROTW <= WAVES && (ROTW & (ROTW - 1)) == 0 && ROTW % ISSUERS == 0, "rotation set: a power of two, whole issuer sets");
    __device__ __forceinline__ void fetch() {
#if defined(VN_EXP) && (VN_EXP == 5 || VN_EXP == 18)
        if (n_left < -1000)                       // timing experiment only: no weight DMA
#endif
        if (ROTATE) {
            const int rel = (wave - turn) & (WAVES - 1);          // (WAVES is a power of two in every ROTATE instantiation; turn + rel < ROTW for an issuer)
            if (ISSUERS == 1) { if (wave == turn) glds_run<PER_WAVE>(g + lane * 4, buf + fill * SF); }
            else if (rel < ISSUERS) glds_run<PER_WAVE>(g + (rel * PER_WAVE) * CHUNK_F + lane * 4, buf + fill * SF + (rel * PER_WAVE) * CHUNK_F);
            turn = (turn + ISSUERS) & (ROTW - 1);
        } else {
            glds_run<PER_WAVE>(g + (wave * PER_WAVE) * CHUNK_F + lane * 4, buf + fill * SF + (wave * PER_WAVE) * CHUNK_F);
        }
        g += SF;
        if (period && ++pos == period) { pos = 0; g = g0; }
        --n_left;
        ++in_flight;
        fill = fill + 1 == NBUF ? 0 : fill + 1;
    }
    // repeat = r > 1: the n_stages image is streamed r times back to back (a persistent workgroup's r tiles: the first stages of the next
    // tile are requested during the last stages of the current one)
    __device__ __forceinline__ void start(const float *stream, int n_stages, float *lds_buf, int lane_, int wave_, int repeat = 1) {
        g = stream; buf = lds_buf; n_left = n_stages * repeat; in_flight = 0; cur = 0; fill = 0; lane = lane_; wave = wave_; turn = 0; cturn = 0; counted = true;
        g0 = stream; period = repeat > 1 ? n_stages : 0; pos = 0;
#pragma unroll
        for (int i = 0; i < NBUF - 1; ++i)
            if (n_left > 0) fetch();
    }
    __device__ __forceinline__ void wait_landed() {
        const int y = in_flight - 1;              // stages requested after the one about to be consumed
        if (NBUF >= 5 && y >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER_WAVE) : "memory");
        else if (NBUF >= 4 && y >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_WAVE) : "memory");
        else if (NBUF >= 3 && y >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * PER_WAVE) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // YOUNGER: a lower bound on the vector-memory instructions (the activation stores of a layer's epilogue) the
    // issuing wave has executed since it issued this stage's DMA.  VM_CNT retires in issue order, loads and stores
    // alike, so vmcnt(YOUNGER) proves the DMA has landed without waiting for those stores to reach memory.
    template <int YOUNGER = 0, int YOUNGER_FIRST = YOUNGER>
    __device__ __forceinline__ const float *wait(bool first = false) {     // `first`: use YOUNGER_FIRST (a call site shared by loop iterations)
        static_assert(YOUNGER >= 0 && YOUNGER <= 63 && YOUNGER_FIRST >= 0 && YOUNGER_FIRST <= 63, "vmcnt is a 6-bit counter");
        if (NBUF == 2 && STAGGER) {
            // every wave drains its own share (counted: its younger stores stay in flight), bare barrier
            __builtin_amdgcn_sched_barrier(0);
            if (!counted) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (YOUNGER_FIRST != YOUNGER && first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER_FIRST) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        } else if (NBUF == 2 && ROTATE) {
            // only the wave that issued this stage's DMA has to see it land -- and not its own younger stores: a
            // vmcnt(0) in every wave (what __syncthreads() also implies: its fence waits for the wave's outstanding
            // global stores and loads) made all of them sit out the HBM latency of their activation stores / mask
            // loads at every stage boundary, and a vmcnt(0) in the issuing wave alone still did so once per layer.
            // The data goes global -> LDS by DMA and LDS -> registers by ds_read: no cache to fence, a bare barrier
            // after the issuer's drain publishes it.
            __builtin_amdgcn_sched_barrier(0);
#if defined(VN_EXP) && VN_EXP == 46
            if (false) {                          // timing experiment only (RACES): the issuer does not wait for its DMA -- what the vmcnt wait behind older stores costs
#else
            if (ISSUERS == 1 ? wave == cturn : ((wave - cturn) & (WAVES - 1)) < ISSUERS) {
#endif
                if (!counted) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (YOUNGER_FIRST != YOUNGER && first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER_FIRST) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
            }
            cturn = (cturn + ISSUERS) & (ROTW - 1);
#if !(defined(VN_EXP) && (VN_EXP == 16 || VN_EXP == 18))
            __builtin_amdgcn_s_barrier();         // (timing experiments 16 / 18 race without it)
#endif
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        } else if (NBUF == 2) {
            glds_drain();
            __syncthreads();
        } else {
            __builtin_amdgcn_sched_barrier(0);
            wait_landed();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        const float *ret = buf + cur * SF;
        cur = cur + 1 == NBUF ? 0 : cur + 1;
        --in_flight;
        return ret;
    }
    __device__ __forceinline__ void prefetch() {
        if (n_left > 0) fetch();
    }
    // called behind every MFMA group g of NG of a stage (compile-time g): where this stream issues the next stage's DMA
    template <int g, int NG>
    __device__ __forceinline__ void prefetch_at() {
        if (STAGGER) {
            static_assert(!STAGGER || NG % WAVES == 0, "groups per stage must be a multiple of the waves");
            constexpr int EVERY = STAGGER ? NG / WAVES : 1;
            if (g % EVERY == 0 && wave == g / EVERY) prefetch();
        } else if (g == 0) prefetch();
    }
    __device__ __forceinline__ const float *next() {
        const float *ret = wait();
        prefetch();
        return ret;
    }
};
// group g of a stage (see gemm_stage_bf): reads of group g+D interleaved one by one behind the first MFMAs of group
// g, then this group's share of the next stage's DMA
// `mid.at<g, NG>()` runs behind every group: the hook through which the narrow kernels send their deferred activation
// stores from inside a stage (vipnerf_bf16n.h).
struct NoMid {
    template <int g, int NG> static constexpr bool active() { return false; }
    template <int g, int NG> __device__ __forceinline__ void at() const {}
};
template <int g, int NG, int NT, int NS, int G, int D, int NBUF, int NB, typename WS, typename ACC, typename BT, typename MIDF>
__device__ __forceinline__ void gemm_groups_bf(const float *base, ACC (&acc)[NT], const BT (&B)[NB][NS], int ks0,
                                               typename FragTypeOf<BT>::type (&fr)[NBUF][G][NS], WS &ws, MIDF &mid) {
    typedef typename FragTypeOf<BT>::type FR;
    if constexpr (g < NG) {
        constexpr int R = (g + D < NG) ? G * NS : 0;           // ds_read_b128 in this group
        constexpr int M = G * CellMfmas<FR, NS>::value * FragTypeOf<BT>::npt;        // MFMAs in this group
        if (g + D < NG) {
#pragma unroll
            for (int tt = 0; tt < G; ++tt)
#pragma unroll
                for (int i = 0; i < NS; ++i)
                    fr[(g + D) % NBUF][tt][i] = *(const FR *)(base + (((g + D) * G + tt) * NS + i) * CHUNK_F);
        }
#pragma unroll
        for (int tt = 0; tt < G; ++tt) {
            const int lin = g * G + tt, ks = lin / NT, t = lin % NT;
            acc[t] = mfma_split<NS>(fr[g % NBUF][tt], B[ks0 + ks], acc[t]);
        }
// build switch VN_INTERLEAVE (default 1, vipnerf_knobs.h)
#pragma unroll
        for (int i = 0; i < (VN_INTERLEAVE ? R : 0); ++i) {    // MFMA, read, MFMA, read, ...: each read issues in the
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);   // shadow of the MFMA before it (measured +3 % over
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); // reads-then-MFMAs)
        }
        if (VN_INTERLEAVE) __builtin_amdgcn_sched_group_barrier(0x8, M - R, 0);
        else { __builtin_amdgcn_sched_group_barrier(0x100, R, 0); __builtin_amdgcn_sched_group_barrier(0x8, M, 0); }
        __builtin_amdgcn_sched_barrier(0);
        // the next stage's DMA goes out as ONE burst behind the first group: a global_load_lds costs its wave ~60
        // cycles of issue time; spread over the groups (one piece per group) the same 16 pieces measured 6-10 %
        // SLOWER than the burst, and before/after the first reads makes no difference
        ws.template prefetch_at<g, NG>();
        __builtin_amdgcn_sched_barrier(0);
        if (MIDF::template active<g, NG>()) { mid.template at<g, NG>(); __builtin_amdgcn_sched_barrier(0); }
        gemm_groups_bf<g + 1, NG, NT, NS, G, D, NBUF>(base, acc, B, ks0, fr, ws, mid);
    }
}

// One stage of a layer, software-pipelined by hand: the stage's (k-step, tile) cells are walked in groups of two
// tiles; the A fragments of the group D steps ahead are read from LDS while the MFMAs of the current group issue,
// with scheduling barriers so the compiler keeps that order (left alone it sinks every read next to its MFMA and
// waits lgkmcnt(0) per tile, which leaves the MFMA pipe idle half the time at one wave per SIMD).  The lead is
// 12 MFMAs = 384 cycles for both NS against ~100-200 cycles of ds_read_b128 latency; the reads in flight stay
// <= 12 because lgkmcnt is a 4-bit counter (with 16 outstanding the compiler falls back to lgkmcnt(0)).
//   acc[t] += A(ks, t) * B[ks0 + ks],  chunk index (ks * NT + t) * NS + part
struct NoStream {
    __device__ __forceinline__ void prefetch() {}
    template <int g, int NG> __device__ __forceinline__ void prefetch_at() {}
};
template <int NT, int NKS, int NS, int NB, typename WS, typename ACC, typename BT, typename MIDF>
__device__ __forceinline__ void gemm_stage_bf(const float *stage, int lane, ACC (&acc)[NT],
                                              const BT (&B)[NB][NS], int ks0, WS &ws, MIDF &mid) {
    typedef typename FragTypeOf<BT>::type FR;
    // narrow layout (floatx4 accumulators, 256 registers per wave) in bf16x6: one tile per group, two groups ahead
    constexpr bool TIGHT = sizeof(ACC) == 16 && NS == 3;
    // single-MFMA modes (NS == 1): a cell is ONE 16-cycle MFMA, so groups of four tiles, two groups (128 cycles of this wave's
    // MFMAs, twice that with its SIMD partner's in between) ahead of the LDS latency; with two point tiles per wave a cell is two
    // MFMAs: groups of two tiles (the same 128 cycles, half the fragment registers)
// build switch VN_PT2_G (default 1, vipnerf_knobs.h)
    constexpr int G = TIGHT ? 1 : (NS == 1 ? (FragTypeOf<BT>::npt == 2 ? VN_PT2_G : 4) : 2);
// build switch VN_PT2_D (default 2, vipnerf_knobs.h)
    constexpr int D = FragTypeOf<BT>::npt == 2 ? VN_PT2_D : ((NS == 2 || TIGHT || NS == 1) ? 2 : 1);
    constexpr int NBUF = D + 1;
    constexpr int NG = NKS * NT / G;
    static_assert(NT % G == 0 && NG >= D, "group shape");
    FR fr[NBUF][G][NS];
    const float *base = stage + lane * 4;
#pragma unroll
    for (int g = 0; g < D; ++g)
#pragma unroll
        for (int tt = 0; tt < G; ++tt)
#pragma unroll
            for (int i = 0; i < NS; ++i) fr[g][tt][i] = *(const FR *)(base + ((g * G + tt) * NS + i) * CHUNK_F);
    __builtin_amdgcn_sched_barrier(0);
    gemm_groups_bf<0, NG, NT, NS, G, D, NBUF>(base, acc, B, ks0, fr, ws, mid);
}
template <int NT, int NKS, int NS, int NB, typename WS, typename ACC, typename BT>
__device__ __forceinline__ void gemm_stage_bf(const float *stage, int lane, ACC (&acc)[NT],
                                              const BT (&B)[NB][NS], int ks0, WS &ws) {
    NoMid none;
    gemm_stage_bf<NT, NKS, NS>(stage, lane, acc, B, ks0, ws, none);
}
#endif


}  // namespace vn
