// Argument blocks and device building blocks shared by the forward and backward MLP kernels.
#pragma once
#include "vipnerf_common.h"

namespace vn {

// Where a workgroup's points come from.
//  rays mode   (the render path): point p = ray n*S + k;  x = o_s[n] + d_s[n]*z[p];  dir = view_dirs[n];
//              secondary directions computed from (z, rays_o, rays_d, rays_o2) as VipNeRF01.py:218-226.
//  points mode (vipnerf_mlp_forward): pts[p], dirs[p], dirs2[p][v] given explicitly.
struct PointSrc {
    int64_t P;            // points
    int32_t S;            // samples per ray (rays mode)
    int32_t V;            // secondary directions
    int32_t rays_mode;
    int32_t ndc;
    const float *z;       // (P)
    const float *rays_o, *rays_d, *rays_o_s, *rays_d_s, *view_dirs, *rays_o2;
    const float *pts, *dirs, *dirs2;
};

struct NoiseSrc {
    const float *noise;   // (P) or NULL
    float std;
    int32_t device_rng;   // 1: draw N(0,1) from Philox when noise == NULL
    uint32_t stream;
    uint64_t seed, offset;
    uint64_t idx_base;    // Philox index of point 0 (= vipnerf_rng::ray_base * S)
    const int64_t *ray_ids;   // rays mode: explicit global row index per ray (vipnerf_rng::ray_ids) or NULL
};

#if defined(__HIPCC__)
// Philox index of point p's sigma-noise draw: (global ray index) * S + sample
__device__ __forceinline__ uint64_t noise_index(const NoiseSrc &ns, const PointSrc &s, int64_t p) {
    if (ns.ray_ids && s.rays_mode) return (uint64_t)ns.ray_ids[p / s.S] * (uint64_t)s.S + (uint64_t)(p % s.S);
    return ns.idx_base + (uint64_t)p;
}
#endif

struct MlpFwdArgs {
    PointSrc src;
    NoiseSrc ns;
    const float *packed;
    float *sigma, *rgb, *vis, *vis2;   // (P), (P,3), (P), (P,V)
    float *acts;                       // activation store base (NULL = do not save)
    ActLayout al;
};

struct MlpBwdArgs {
    PointSrc src;
    const float *packed;
    const float *sigma, *rgb, *vis, *vis2;       // forward outputs (for the activation derivatives)
    const float *acts;
    ActLayout al;
    float *bwd;                                  // backward scratch base
    BwdLayout bl;
    const unsigned *gmax;                        // FP16X3: max |seed| slot (grad_scale_from_max); else nullptr
};

constexpr int MLP_WG = 256;                 // 4 waves, one per SIMD
constexpr int MLP_PTS_PER_WG = 128;         // 32 points per wave
constexpr int MLP_LDS_F = R_TOTAL_PAD + 2 * STAGE_F;
constexpr size_t MLP_LDS_BYTES = (size_t)MLP_LDS_F * 4;   // 95,232 B

#if defined(__HIPCC__)

// ------------------------------------------------------------------------------------------- weight stream
// Double-buffered L2 -> LDS stream of 32 KiB stages via LDS-DMA (global_load_lds_dwordx4; the packed image is
// already in lane order, so the DMA's lane-linear destination is exactly the fragment layout).  One
// __syncthreads() per stage: it drains this wave's DMA (hipcc emits vmcnt(0) for pending LDS-DMA) and orders
// every wave's reads of the buffer about to be overwritten.  A stage is 128 MFMAs per wave (>= 8192 cycles),
// so the next stage's DMA (issued right after the barrier) has long landed by the next barrier.
struct WStream {
    const float *g;        // next stage to fetch (global)
    float *buf;            // LDS stage buffers (2 * STAGE_F)
    int n_left;            // stages not yet fetched
    int cur;               // buffer holding the stage about to be consumed
    int lane, wave;

    __device__ __forceinline__ void fetch(int b) {
        constexpr int PER_WAVE = STAGE_CHUNKS / 4;
        glds_run<PER_WAVE>(g + (wave * PER_WAVE) * CHUNK_F + lane * 4, buf + b * STAGE_F + (wave * PER_WAVE) * CHUNK_F);
        g += STAGE_F;
        --n_left;
    }
    __device__ __forceinline__ void start(const float *stream, int n_stages, float *lds_buf, int lane_, int wave_) {
        g = stream; buf = lds_buf; n_left = n_stages; cur = 0; lane = lane_; wave = wave_;
        fetch(0);
    }
    // returns the LDS address of the stage to consume now
    __device__ __forceinline__ const float *next() {
#if defined(VN_EXP) && VN_EXP == 3
        if (n_left & 1) __syncthreads();          // timing experiment only (races): half the barriers
#elif defined(VN_EXP) && VN_EXP == 4
        if (n_left == 1000) __syncthreads();      // timing experiment only (races): no barriers
#else
        glds_drain();                             // the DMA is issued from asm: hipcc does not wait for it
        __syncthreads();
#endif
        const float *ret = buf + cur * STAGE_F;
        cur ^= 1;
        if (n_left > 0) fetch(cur);
        return ret;
    }
};

__device__ __forceinline__ floatx16 mfma(float a, float b, floatx16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// acc[0..NT) += A(stage) * B where B row r (r = 4*kg + q, kg = KG0 + gl) is the register in_r(r).
// `stage` holds NKG kgroups x NT tiles, chunk index gl*NT + t.
#define VN_GEMM_STAGE(stage, NT, NKG, KG0, acc, BEXPR)                                              \
    _Pragma("unroll") for (int gl_ = 0; gl_ < (NKG); ++gl_) {                                       \
        _Pragma("unroll") for (int t_ = 0; t_ < (NT); ++t_) {                                       \
            const float4 a_ = *(const float4 *)((stage) + (gl_ * (NT) + t_) * CHUNK_F + lane * 4);  \
            { const int r_ = 4 * ((KG0) + gl_) + 0; acc[t_] = mfma(a_.x, BEXPR, acc[t_]); }         \
            { const int r_ = 4 * ((KG0) + gl_) + 1; acc[t_] = mfma(a_.y, BEXPR, acc[t_]); }         \
            { const int r_ = 4 * ((KG0) + gl_) + 2; acc[t_] = mfma(a_.z, BEXPR, acc[t_]); }         \
            { const int r_ = 4 * ((KG0) + gl_) + 3; acc[t_] = mfma(a_.w, BEXPR, acc[t_]); }         \
        }                                                                                           \
    }

// The same product, software-pipelined by hand (see gemm_stage_bf in vipnerf_bf16.h for the why): cells are walked
// in groups of two tiles, the float4 A fragments of the next group are in flight while this group's 8 MFMAs
// (512 cycles) issue; the two tiles' MFMAs alternate so no accumulator is used back to back.  Per accumulator the
// order of the additions is that of VN_GEMM_STAGE, so results are bit-identical.  bfun(r) = B row r.
template <int NT, int NKG, typename BF>
__device__ __forceinline__ void gemm_stage_f32(const float *stage, int lane, floatx16 (&acc)[NT], int kg0, BF bfun) {
    constexpr int G = 2;
    constexpr int NG = NKG * NT / G;
    static_assert(NT % G == 0, "group shape");
    float4 fr[2][G];
    const float *base = stage + lane * 4;
#pragma unroll
    for (int tt = 0; tt < G; ++tt) fr[0][tt] = *(const float4 *)(base + tt * CHUNK_F);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) {
#pragma unroll
            for (int tt = 0; tt < G; ++tt) fr[(g + 1) & 1][tt] = *(const float4 *)(base + ((g + 1) * G + tt) * CHUNK_F);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int lin = g * G, gl = lin / NT, t0 = lin % NT, r0 = 4 * (kg0 + gl);
        const float4 a0 = fr[g & 1][0], a1 = fr[g & 1][1];
        acc[t0] = mfma(a0.x, bfun(r0), acc[t0]);         acc[t0 + 1] = mfma(a1.x, bfun(r0), acc[t0 + 1]);
        acc[t0] = mfma(a0.y, bfun(r0 + 1), acc[t0]);     acc[t0 + 1] = mfma(a1.y, bfun(r0 + 1), acc[t0 + 1]);
        acc[t0] = mfma(a0.z, bfun(r0 + 2), acc[t0]);     acc[t0 + 1] = mfma(a1.z, bfun(r0 + 2), acc[t0 + 1]);
        acc[t0] = mfma(a0.w, bfun(r0 + 3), acc[t0]);     acc[t0 + 1] = mfma(a1.w, bfun(r0 + 3), acc[t0 + 1]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ------------------------------------------------------------------------------------------- encodings
// gamma_L(v): feature f < 3 -> v[f]; f = 3 + 6*l + 3*c + d -> (c ? cos : sin)(2^l v[d])   (VipNeRF01.py:424-448)
template <int L, int NS>
__device__ __forceinline__ void encode_half(const float v[3], int h, float (&out)[NS]) {
    float val[2 * NS];
#pragma unroll
    for (int f = 0; f < 2 * NS; ++f) val[f] = 0.f;
#pragma unroll
    for (int d = 0; d < 3; ++d) val[d] = v[d];
#pragma unroll
    for (int l = 0; l < L; ++l) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            float s, c;
            sincosf(v[d] * (float)(1 << l), &s, &c);
            val[3 + 6 * l + d] = s;
            val[3 + 6 * l + 3 + d] = c;
        }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) out[s] = h ? val[2 * s + 1] : val[2 * s];
}

// Loads / computes the wave's point: position x, primary direction, and what is needed for the secondary ones.
struct PointCtx {
    float x[3], dir[3];
    float o[3], d[3], z;      // world-space ray + depth (rays mode)
    int64_t p, n;             // point, ray
};
__device__ __forceinline__ void load_point(const PointSrc &s, int64_t p, PointCtx &c) {
    c.p = p;
    if (s.rays_mode) {
        const int64_t n = p / s.S;
        c.n = n;
        c.z = s.z[p];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            c.x[i] = __fadd_rn(s.rays_o_s[3 * n + i], __fmul_rn(s.rays_d_s[3 * n + i], c.z));
            c.dir[i] = s.view_dirs[3 * n + i];
            c.o[i] = s.rays_o[3 * n + i];
            c.d[i] = s.rays_d[3 * n + i];
        }
    } else {
        c.n = p;
        c.z = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            c.x[i] = s.pts[3 * p + i];
            c.dir[i] = s.dirs[3 * p + i];
            c.o[i] = c.d[i] = 0.f;
        }
    }
}
// the rays-mode arithmetic of secondary_dir with the secondary camera's origin already in registers (callers that load it ahead of use)
__device__ __forceinline__ void secondary_dir_from(const PointSrc &s, const PointCtx &c, const float (&o2)[3], float out[3]) {
    float t = c.z;
    if (s.ndc) {
        const float tn = __fdiv_rn(-(1.f + c.o[2]), c.d[2]);
        const float num = __fadd_rn(c.o[2], __fmul_rn(tn, c.d[2]));
        const float den = __fadd_rn(__fsub_rn(1.f, c.z), 1e-6f);
        t = __fdiv_rn(__fsub_rn(__fdiv_rn(num, den), c.o[2]), c.d[2]);
    }
    float w[3], n2 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float pt = __fadd_rn(c.o[i], __fmul_rn(t, c.d[i]));
        w[i] = __fsub_rn(pt, o2[i]);
        n2 = __fadd_rn(n2, __fmul_rn(w[i], w[i]));
    }
    const float nrm = sqrtf(n2);
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = __fdiv_rn(w[i], nrm);
}
// direction from secondary camera v to the point (VipNeRF01.py:218-226)
__device__ __forceinline__ void secondary_dir(const PointSrc &s, const PointCtx &c, int v, float out[3]) {
    if (!s.rays_mode) {
#pragma unroll
        for (int i = 0; i < 3; ++i) out[i] = s.dirs2[(c.p * s.V + v) * 3 + i];
        return;
    }
    float o2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) o2[i] = s.rays_o2[(c.n * s.V + v) * 3 + i];
    secondary_dir_from(s, c, o2, out);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// C/D-fragment <-> row-major [P][ld] helpers: lane (point j, half h) owns, for tile t and register group q,
// the 4 consecutive features 32t + 8q + 4h .. +3 = registers 4q..4q+3 of acc[t].
template <int NT>
__device__ __forceinline__ void store_frag(float *base, int64_t p, int ld, int h, const floatx16 (&v)[NT], bool valid) {
    if (!valid) return;
    float *row = base + (size_t)p * ld + 4 * h;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *(float4 *)(row + 32 * t + 8 * q) = make_float4(v[t][4 * q], v[t][4 * q + 1], v[t][4 * q + 2], v[t][4 * q + 3]);
}
// one 32-feature tile of a fragment (4 x 16 B per lane)
__device__ __forceinline__ void store_tile(float *base, int64_t p, int ld, int h, int t, const floatx16 &v, bool valid) {
    if (!valid) return;
#if defined(VN_EXP) && VN_EXP == 1
    return;                                   // experiment: no activation stores at all
#endif
    float *row = base + (size_t)p * ld + 4 * h + 32 * t;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#if defined(VN_EXP) && VN_EXP == 2
        *(float4 *)(row + 8 * q) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
#else
        // non-temporal: the activation / dY store is written once and next read by wgrad ~10 ms (11 GB) later,
        // so it should not displace the L2-resident weight stream (measured: -0.4 ms per step vs plain stores)
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 val = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        __builtin_nontemporal_store(val, (f4 *)(row + 8 * q));
#endif
    }
}
// ReLU mask of a fragment: bit (16*(t&1) + r) of word t>>1 <=> v[t][r] > 0
__device__ __forceinline__ uint4 frag_mask(const floatx16 (&v)[8]) {
    unsigned m[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) m[t >> 1] |= (v[t][r] > 0.f ? 1u : 0u) << (16 * (t & 1) + r);
    return make_uint4(m[0], m[1], m[2], m[3]);
}
__device__ __forceinline__ bool mask_bit(const uint4 &m, int t, int r) {
    const unsigned w = (t >> 1) == 0 ? m.x : ((t >> 1) == 1 ? m.y : ((t >> 1) == 2 ? m.z : m.w));
    return (w >> (16 * (t & 1) + r)) & 1u;
}

template <int NT>
__device__ __forceinline__ void load_frag(const float *base, int64_t p, int ld, int h, floatx16 (&v)[NT]) {
    const float *row = base + (size_t)p * ld + 4 * h;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 f = *(const float4 *)(row + 32 * t + 8 * q);
            v[t][4 * q] = f.x; v[t][4 * q + 1] = f.y; v[t][4 * q + 2] = f.z; v[t][4 * q + 3] = f.w;
        }
}
#endif

int launch_mlp_fwd(const MlpFwdArgs &a, hipStream_t st);
int launch_mlp_bwd(const MlpBwdArgs &a, hipStream_t st);
int launch_pack(const vipnerf_mlp_params *p, void *packed, hipStream_t st);

}  // namespace vn
