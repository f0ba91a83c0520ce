// Internal definitions shared by the gfx950 kernels of libvipnerf_hip.so.
//
// Design in one paragraph (DESIGN.md has the long form, docs/HISTORY.md the experiments): the MLP runs TRANSPOSED and REGISTER-CHAINED.  A
// wave owns 32 points (MFMA columns); a layer is H_out^T[256 x 32] = W[256 x K] * H_in^T[K x 32] computed as
// 8 row tiles of v_mfma_f32_32x32x2_f32.  In that orientation the C/D fragment of one layer (lane = point,
// registers = output features) is, register for register, a valid B fragment of the next layer provided the
// contraction index of that layer is enumerated in the order feat(r, h) below -- so activations never leave
// the register file between layers and only the WEIGHTS move: they are pre-packed once per optimizer step into
// exactly the per-lane A-fragment order (vipnerf_pack.hip) and streamed L2 -> LDS in 32 KB stages shared by
// the 4 waves of a workgroup.
#pragma once
#include "vipnerf_knobs.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/vipnerf_hip.h"

namespace vn {

// ----------------------------------------------------------------------------------------------- topology
constexpr int W = 256;          // trunk width
constexpr int WV = 128;         // view-branch width
constexpr int D = 8;            // trunk depth
constexpr int LP = 10;          // positional-encoding degree, points
constexpr int LV = 4;           // positional-encoding degree, directions
constexpr int DPE = 3 + 6 * LP; // 63
constexpr int DVE = 3 + 6 * LV; // 27
constexpr int DPE_PAD = 64;
constexpr int DVE_PAD = 32;
constexpr int SKIP_LAYER = 5;   // layer whose input is [gamma(x), h]

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// ----------------------------------------------------------------------------------------------- packed image
// One chunk = the A operands one ds_read_b128 per lane fetches for one 16-feature tile: 64 lanes x float4 = 1 KiB (vipnerf_bf16n.h: BnPlan
// lays the stages of an MLP's packed image out of them).
constexpr int CHUNK_F = 256;                    // floats per chunk

// parameter slots (vipnerf_mlp_params::p)
constexpr int P_LW0 = 0;      // pts_linears[i].weight = 2*i, bias = 2*i+1
constexpr int P_VW = 16, P_VB = 17, P_SW = 18, P_SB = 19, P_FW = 20, P_FB = 21, P_OW = 22, P_OB = 23;

__host__ __device__ inline int layer_in_dim(int i) { return i == 0 ? DPE : (i == SKIP_LAYER ? W + DPE : W); }

// ----------------------------------------------------------------------------------------------- workspaces
// Activation store of one level (floats).  P = points of the level.
struct ActLayout {
    size_t h[D];      // [P][256] output of pts_linears[i] (post ReLU)
    size_t hm[D];     // [P][2][4] uint32: ReLU masks of the same, one bit per feature in C/D-fragment order
    size_t feat;      // [P][256]
    size_t g[1 + VIPNERF_MAX_SEC];    // [P][128] view-branch hidden (post ReLU), per direction
    size_t pex;       // [P][64]  gamma(x), zero padded
    size_t ped[1 + VIPNERF_MAX_SEC];  // [P][32]  gamma(dir), zero padded, per direction
    size_t gm[1 + VIPNERF_MAX_SEC];   // [P][4] uint32, fp32 storage only: ReLU bits of the view hidden (bit 4 t + r of lane group q's word = feature 16 t + 4 q + r > 0)
    size_t total;
};
__host__ __device__ inline ActLayout act_layout(size_t P, int V, bool t16 = false) {
    // t16 (single-MFMA 16-bit modes, T16 operand storage): every array holds 16-bit values -- half the floats; the view hidden's slot
    // also carries its ReLU bits ([P][4] words behind the tiles)
    ActLayout a; size_t o = 0;
    const size_t d = t16 ? 2 : 1;
    for (int i = 0; i < D; ++i) { a.h[i] = o; o += P * W / d; }
    for (int i = 0; i < D; ++i) { a.hm[i] = o; o += P * 8; }
    a.feat = o; o += P * W / d;
    for (int k = 0; k <= VIPNERF_MAX_SEC; ++k) { a.g[k] = o; if (k <= V) o += t16 ? P * (WV / 2 + 4) : P * WV; }
    a.pex = o; o += P * DPE_PAD / d;
    for (int k = 0; k <= VIPNERF_MAX_SEC; ++k) { a.ped[k] = o; if (k <= V) o += P * DVE_PAD / d; }
    for (int k = 0; k <= VIPNERF_MAX_SEC; ++k) { a.gm[k] = o; if (k <= V && !t16) o += P * 4; }     // (T16: the bits sit behind the view hidden's tiles)
    a.total = o;
    return a;
}

// Backward scratch of one level (floats).
struct BwdLayout {
    size_t dy[D];     // [P][256] dLoss/d(pre-activation of pts_linears[i])
    size_t dyf;       // [P][256] dLoss/d(feature)
    size_t dyv[1 + VIPNERF_MAX_SEC];  // [P][128] per direction
    size_t dyvsum;    // [P][128]
    size_t dq[1 + VIPNERF_MAX_SEC];   // [P][8]: a=0: d(pre-sigmoid rgb,vis), d(sigma_raw); a>=1: (0,0,0,d pre-sigmoid vis2_a)
    size_t dsig, drgb, dvis, dvis2;   // [P], [P][3], [P], [P][V]: dLoss/d(raw network outputs)
    size_t gmax;      // [64] slot; word 0 = bit pattern of max |d raw output| of the level (FP16X3 gradient scaling)
    size_t dy5f;      // FP16X3H only (gradients stored as fp16 high parts): fp32 copy of dy[5] for layer 5's gamma(x) weight-gradient GEMM ([P][256])
    size_t partial;   // wgrad partial sums
    size_t total;
};
// wgrad work split: the point axis is cut into chunks; every (GEMM, chunk) pair is one workgroup that writes
// its partial product to `partial`, then an ordered reduction sums the chunks (deterministic, no atomics).
constexpr int WGRAD_CHUNK_PTS = 8192;
constexpr int WGRAD_MAX_CHUNKS = 256;
// The chunk count is a multiple of 32 -- the eight 256x256 GEMMs of a level then come to whole rounds of the 256 CUs
// (4096 rays: 32 / 96 chunks as before; 1024 rays: 32 / 32 instead of 8 / 24, which filled a quarter of the chip) -- with
// chunks of at most WGRAD_CHUNK_PTS and, for small inputs, at least 1024 points.
__host__ __device__ inline int wgrad_chunks(size_t P) {
    const size_t round = 32 * (size_t)WGRAD_CHUNK_PTS;
    size_t c = 32 * ((P + round - 1) / round);
    const size_t cmax = (P + 1023) / 1024;
    if (c > cmax) c = cmax;
    return (int)(c < 1 ? 1 : (c > WGRAD_MAX_CHUNKS ? WGRAD_MAX_CHUNKS : c));
}
// The small GEMMs (few MFMAs per point) use shorter chunks, chosen per class so that each launch comes to whole rounds
// of the workgroups the chip holds of that class (4096 rays: 32 / 96 chunks of the 256x256 class):
//   256x64 (gamma(x) columns, 2 GEMMs, 2 workgroups per CU)                 chunks / 8  -> 512 / 1536 workgroups
//   128x32 and 32x128 (per direction, 40 KiB of LDS: 4 workgroups per CU)   chunks / 16 -> 1024 / 3072 with V = 1
//   128x256 and the fp32 mode's sigma head (alone in their launch, 1 per CU) chunks / 8 -> 256 / 768
constexpr int WGRAD_SPLIT_PE = 8, WGRAD_SPLIT_THIN = 16, WGRAD_SINGLE_SPLIT = 8;
__host__ __device__ inline int wgrad_chunk_pts(size_t P) {            // multiple of 32 * WGRAD_SPLIT_THIN
    const int n = wgrad_chunks(P);
    const size_t c = (P + n - 1) / n, q = 32 * WGRAD_SPLIT_THIN;
    return (int)(c == 0 ? q : (c + q - 1) / q * q);          // (an empty batch still gets a well-formed plan)
}
__host__ __device__ inline int wgrad_chunks_split(size_t P, int split) {
    const size_t c = wgrad_chunk_pts(P) / split;
    return (int)((P + c - 1) / c);
}
// floats of partial output: sum over GEMMs of chunks * (Mp*Kp + Mp)   (Mp for the bias column sums)
__host__ __device__ inline size_t wgrad_partial_total(size_t P, int V) {
    const size_t big = 8 * (size_t)(256 * 256 + 256) + 320;   // layers 1-4, 5(h part), 6, 7, feature (+ the sigma head's column sums)
    const size_t pe = 2 * (size_t)(256 * 64 + 256);           // layer 0, layer 5 gamma(x) part
    const size_t single = (size_t)(32 * 256 + 32) + (size_t)(128 * 256 + 128);   // sigma head; view layer, feature columns
    const size_t thin = (size_t)(1 + V) * (128 * 32 + 128) + (size_t)(1 + V) * (32 * 128 + 32);   // view-layer direction columns, output head
    return (size_t)wgrad_chunks(P) * big + (size_t)wgrad_chunks_split(P, WGRAD_SPLIT_PE) * pe +
           (size_t)wgrad_chunks_split(P, WGRAD_SINGLE_SPLIT) * single + (size_t)wgrad_chunks_split(P, WGRAD_SPLIT_THIN) * thin;
}
__host__ __device__ inline BwdLayout bwd_layout(size_t P, int V, bool h16 = false, bool t16 = false) {
    BwdLayout b; size_t o = 0;
    const size_t d = t16 ? 2 : 1;                // T16 storage: 16-bit gradients, half the floats
    for (int i = 0; i < D; ++i) { b.dy[i] = o; o += P * W / d; }
    b.dyf = o; o += P * W / d;
    for (int k = 0; k <= VIPNERF_MAX_SEC; ++k) { b.dyv[k] = o; if (k <= V) o += P * WV / d; }
    b.dyvsum = o; o += P * WV / d;
    for (int k = 0; k <= VIPNERF_MAX_SEC; ++k) { b.dq[k] = o; if (k <= V) o += P * 8; }
    b.dsig = o; o += P;
    b.drgb = o; o += 3 * P;
    b.dvis = o; o += P;
    b.dvis2 = o; o += P * (V > 0 ? V : 1);
    o = (o + 63) & ~(size_t)63;
    b.gmax = o; o += 64;
    b.dy5f = o; if (h16) o += P * W;
    b.partial = o; o += wgrad_partial_total(P, V);
    b.total = o;
    return b;
}

// ----------------------------------------------------------------------------------------------- device helpers
#if defined(__HIPCC__)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// inclusive scans across the 64 lanes of a wave
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { float t = __shfl_up(v, o, 64); if (lane >= o) v *= t; }
    return v;
}
__device__ __forceinline__ float wave_scan_add(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { float t = __shfl_up(v, o, 64); if (lane >= o) v += t; }
    return v;
}
__device__ __forceinline__ float wave_rscan_add(float v, int lane) {   // inclusive suffix sum
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { float t = __shfl_down(v, o, 64); if (lane + o < 64) v += t; }
    return v;
}

// FP16X3 gradient scaling.  fp16 operands have 5 exponent bits, and dLoss/d(pre-activation) is ~1e-5 and smaller:
// the data-gradient pass of that mode works on 2^S * dY, S chosen from the level's largest |d raw output| m so that
// 2^S m lies in [32, 64) (1000x headroom to fp16's 65504 for growth through the layers; values 2^-19 of the largest
// still have normal low parts).  Everything in the backward workspace is then 2^S times the true value and the
// weight-gradient reduction multiplies by 2^-S -- all exact.  m is the bit pattern written by k_seed_absmax.
__device__ __forceinline__ float grad_scale_from_max(unsigned m_bits) {
    const float m = __uint_as_float(m_bits);
    if (!(m > 0.f) || !(m < 3.0e38f)) return 1.f;
    int e;
    (void)frexpf(m, &e);                                   // m = f * 2^e, f in [0.5, 1)
    int S = 6 - e;
    S = S < -100 ? -100 : (S > 100 ? 100 : S);
    return ldexpf(1.f, S);
}

// LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = one 1 KiB chunk per instruction), N consecutive chunks.
// Issued from inline asm on purpose: with the builtin hipcc sees a pending FLAT access for as long as the DMA is
// in flight (= the whole stage) and degrades every LDS wait to lgkmcnt(0), which serialises the ds_read -> MFMA
// pipeline of the consumer.  The statement has no VGPR destination, so it is register-safe; its completion is
// the caller's job (s_waitcnt vmcnt(0) before the barrier that publishes the stage, see WStream::wait).
// `lds_dst` is the wave-uniform LDS byte address of chunk 0, `gsrc` this lane's 16 bytes of chunk 0; the
// instruction offset advances the global and the LDS address together (chunks are 1 KiB apart in both).
template <int N>
__device__ __forceinline__ void glds_chunks(const float *gsrc, unsigned lds_dst) {
    static_assert(N >= 1 && N <= 4, "13-bit instruction offset: at most 4 chunks per base");
    unsigned keep;
    if (N == 4)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else if (N == 3)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, off offset:2048\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else if (N == 2)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\t"
                     "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// PER_WAVE consecutive chunks from gsrc (this lane's pointer into chunk 0) to the LDS address of dst
template <int PER_WAVE>
__device__ __forceinline__ void glds_run(const float *gsrc, const float *dst) {
    const unsigned d0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)dst);
#pragma unroll
    for (int i = 0; i + 4 <= PER_WAVE; i += 4) glds_chunks<4>(gsrc + i * CHUNK_F, d0 + i * CHUNK_F * 4);
    constexpr int REM = PER_WAVE % 4, DONE = PER_WAVE - REM;
    if (REM) glds_chunks<REM ? REM : 1>(gsrc + DONE * CHUNK_F, d0 + DONE * CHUNK_F * 4);
}
// all outstanding LDS-DMA (and stores) of this wave have landed
__device__ __forceinline__ void glds_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Philox4x32-10 (Salmon et al. 2011): one 128-bit counter -> four 32-bit words.
__device__ __forceinline__ uint4 philox4x32(uint4 c, uint2 k) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
        const uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += W0; k.y += W1;
    }
    return c;
}
// stream ids for the on-device generator
enum { RS_TRAND = 1, RS_U = 2, RS_NOISE_C = 3, RS_NOISE_F = 4 };
__device__ __forceinline__ float rng_uniform(uint64_t seed, uint64_t offset, uint32_t stream, uint64_t idx) {
    uint4 c = make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), stream, (uint32_t)offset);
    uint4 r = philox4x32(c, make_uint2((uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(offset >> 32)));
    return (float)(r.x >> 8) * (1.0f / 16777216.0f);                  // [0,1), 24 bits like torch.rand
}
__device__ __forceinline__ float rng_normal(uint64_t seed, uint64_t offset, uint32_t stream, uint64_t idx) {
    uint4 c = make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), stream, (uint32_t)offset);
    uint4 r = philox4x32(c, make_uint2((uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(offset >> 32)));
    const float u1 = ((float)(r.x >> 8) + 1.0f) * (1.0f / 16777216.0f);   // (0,1]
    const float u2 = (float)(r.y >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}
#endif

// ----------------------------------------------------------------------------------------------- error plumbing
void set_error(const char *fmt, ...);
#define VN_HIP(call)                                                                                 \
    do {                                                                                             \
        hipError_t e__ = (call);                                                                     \
        if (e__ != hipSuccess) {                                                                     \
            vn::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return VIPNERF_E_HIP;                                                                    \
        }                                                                                            \
    } while (0)

}  // namespace vn
