// Weight gradients of one level: every nn.Linear's dW = dY^T H (and db = column sums of dY) as fp32-MFMA
// GEMMs that contract over the POINT axis (autograd of reference src/models/VipNeRF01.py:537-596 w.r.t. the
// parameters).
//
// A = dY [P][M] and B = H [P][K] are the row-major arrays the forward / dgrad kernels left in HBM.  With points
// along MFMA's k the fragments are rows of those arrays: lane l supplies A[p = 2s + (l>>5)][o = 32*ot + (l&31)]
// and B[p][k = 32*kt + (l&31)] -- 128 contiguous bytes per half-wave, conflict-free from a row-major LDS tile.
// A workgroup owns one (GEMM, point-chunk) pair, keeps the whole M x K product in accumulators (up to 16
// 32x32 tiles = 256 registers per wave), streams 32-point tiles of A and B through double-buffered LDS, and
// writes its partial product; an ordered second pass sums the chunks (deterministic; no atomics) straight into
// the nn.Linear-layout gradient tensors.
#include <type_traits>
#include "vipnerf_wgrad.h"
#include "vipnerf_bf16n.h"
#include "vipnerf_prof.h"
#include "vipnerf_mlp_pt2.h"

namespace vn {

// LDS plane layout of the split-precision kernels: [row][4 slots of 8 points], 64 bytes per feature row, NF features.
// A staging thread writes features 4 g .. 4 g + 3 for consecutive g across lanes, a fragment read takes features
// 32 t + lane.  Rows are grouped by feature & 3 (row = (f & 3) * NF/4 + f >> 2: consecutive writers 64 bytes apart
// instead of 256) and the slot is XORed with (f & 3) ^ (f >> 4), so that 8 consecutive writers and each 16-lane read
// group hit 16 distinct 4-bank groups.  (SQ_LDS_BANK_CONFLICT of these kernels is exactly 8 cycles per ds_write_b128
// wave-instruction with this and with the previous layout alike: the counter charges the 8 LDS cycles a 1 KiB store
// needs, it does not indicate address conflicts here.)
template <int NF>
__device__ __forceinline__ int wg_off(int f, int slot) {
    const int row = (f & 3) * (NF / 4) + (f >> 2);
    return row * 64 + ((slot ^ (((f & 3) ^ (f >> 4)) & 3)) << 4);
}

// build switch VN_WGRAD_DMA (default 0, vipnerf_knobs.h): exact-fp32 256 x 256 weight gradients: 1 = operand blocks HBM -> LDS by DMA instead of through registers -- built, correct, and measured SLOWER (9.10 vs 8.22 ms per step: docs/HISTORY.md 5); off
// build switch VN_WGRAD_W8 (default 2, vipnerf_knobs.h): exact-fp32 256 x 256 weight gradients: 0 = the 4-wave k_wgrad<2,8,4>; 2 / 4 = k_wgrad256_w8 with 8 / 16 waves
typedef _Float16 wg_half8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ floatx16 mfma16_32(wg_half8 a, wg_half8 b, floatx16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ floatx16 mfma16_32(wg_bf8 a, wg_bf8 b, floatx16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ floatx16 mfma32(float a, float b, floatx16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// MTW x KTW tiles per wave; WAVES_M waves split the M tiles (the other 4/WAVES_M split K).
template <int MTW, int KTW, int WAVES_M>
__global__ __launch_bounds__(256) void k_wgrad(WgArgs a) {
    constexpr int WAVES_K = 4 / WAVES_M;
    constexpr int MT = MTW * WAVES_M, KT = KTW * WAVES_K;
    constexpr int Mp = 32 * MT, Kp = 32 * KT;
    constexpr int TILE_F = 32 * (Mp + Kp);
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const WgDesc &d = a.d[blockIdx.y];
    if ((int)blockIdx.x >= d.n_chunks) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int wm = wave % WAVES_M, wk = wave / WAVES_M;
    const int64_t p0 = (int64_t)blockIdx.x * a.chunk_pts;
    const int64_t p1 = p0 + a.chunk_pts < a.P ? p0 + a.chunk_pts : a.P;
    const int nblk = (int)((p1 - p0 + 31) / 32);

    floatx16 acc[MTW][KTW];
    float bsum[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        bsum[i] = 0.f;
#pragma unroll
        for (int j = 0; j < KTW; ++j) acc[i][j] = (floatx16)(0.f);
    }

    float4 ra[MT], rb[KT];
    // whole-tile-contiguous operands (row stride == padded width, all columns valid) are a linear 16 B/lane copy
    const bool linA = d.lda == Mp && d.m_load == Mp, linB = d.ldb == Kp && d.k_load == Kp;
    // the operand bases pinned in SGPRs (see wgrad256_body: re-read from the kernel arguments per block otherwise, behind an s_waitcnt lgkmcnt(0))
    typedef float __attribute__((ext_vector_type(4))) f4v;
    typedef const f4v __attribute__((address_space(1))) *gf4p;
    unsigned long long uA = (unsigned long long)d.A, uB = (unsigned long long)d.B;
    asm volatile("" : "+s"(uA), "+s"(uB));
    const gf4p gA = (gf4p)uA, gB = (gf4p)uB;
    auto gload = [&](int blk) {
        const int64_t pb = p0 + (int64_t)blk * 32;
        const bool inrange = pb + 32 <= p1;
        if (inrange && linA && linB) {
            const gf4p ta = gA + (size_t)pb * (Mp / 4) + tid, tb = gB + (size_t)pb * (Kp / 4) + tid;
#pragma unroll
            for (int i = 0; i < MT; ++i) { const f4v v = ta[256 * i]; ra[i] = make_float4(v.x, v.y, v.z, v.w); }
#pragma unroll
            for (int i = 0; i < KT; ++i) { const f4v v = tb[256 * i]; rb[i] = make_float4(v.x, v.y, v.z, v.w); }
            return;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int idx = tid + 256 * i, row = idx / (8 * MT), col = 4 * (idx % (8 * MT));
            const bool ok = (pb + row < p1) && (col < d.m_load);
            ra[i] = ok ? *(const float4 *)(d.A + (size_t)(pb + row) * d.lda + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < KT; ++i) {
            const int idx = tid + 256 * i, row = idx / (8 * KT), col = 4 * (idx % (8 * KT));
            const bool ok = (pb + row < p1) && (col < d.k_load);
            rb[i] = ok ? *(const float4 *)(d.B + (size_t)(pb + row) * d.ldb + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&](int buf) {
        float4 *la = (float4 *)(lds + buf * TILE_F), *lb = (float4 *)(lds + buf * TILE_F + 32 * Mp);
#pragma unroll
        for (int i = 0; i < MT; ++i) la[tid + 256 * i] = ra[i];
#pragma unroll
        for (int i = 0; i < KT; ++i) lb[tid + 256 * i] = rb[i];
    };

    if (nblk > 0) { gload(0); lstore(0); }
    __syncthreads();
    int cur = 0;
    for (int blk = 0; blk < nblk; ++blk) {
        if (blk + 1 < nblk) gload(blk + 1);
        const float *la = lds + cur * TILE_F, *lb = la + 32 * Mp;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float af[MTW], bf[KTW];
#pragma unroll
            for (int i = 0; i < MTW; ++i) af[i] = la[(2 * s + h) * Mp + 32 * (wm * MTW + i) + l31];
#pragma unroll
            for (int j = 0; j < KTW; ++j) bf[j] = lb[(2 * s + h) * Kp + 32 * (wk * KTW + j) + l31];
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                bsum[i] += af[i];
#pragma unroll
                for (int j = 0; j < KTW; ++j) acc[i][j] = mfma32(af[i], bf[j], acc[i][j]);
            }
        }
        if (blk + 1 < nblk) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // partial product of this chunk: [Mp][Kp] row-major, then the bias column sums [Mp]
    float *part = a.partial + d.part_off + (size_t)blockIdx.x * d.part_stride;
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int ot = wm * MTW + i;
#pragma unroll
        for (int j = 0; j < KTW; ++j) {
            const int kt = wk * KTW + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = 32 * ot + (r & 3) + 8 * (r >> 2) + 4 * h;
                part[(size_t)o * Kp + 32 * kt + l31] = acc[i][j][r];
            }
        }
        if (wk == 0) {
            const float b = bsum[i] + __shfl_xor(bsum[i], 32, 64);
            if (h == 0) part[(size_t)Mp * Kp + 32 * ot + l31] = b;
        }
    }
}



// ---------------------------------------------------------------------------------------------------------------------------------------
// The view layer's weight gradient, exact fp32, in ONE launch: dW_view[:, 0:256] = (sum_a dYv_a)^T feature and dW_view[:, 256:283] =
// sum_a dYv_a^T gamma(dir_a) (reference src/models/VipNeRF01.py:576-590: views_linears[0] over cat(feature, gamma(dir)), evaluated once per
// direction a = 0..V).  Before: a 128 x 256 launch over dYvsum = sum_a dYv_a (an array the data-gradient kernel wrote for it) and one
// 128 x 32 GEMM per direction over dYv_a -- every dYv_a streamed twice (once as part of the sum).  Here a workgroup stages each 16-point
// block of dYv_0..dYv_V, gamma(dir_0..V) and the feature ONCE, forms the sum while staging (the thread that loads a float4 of dYv_a loads
// the same float4 of every direction), and runs both products from the tiles: 2304 B per point instead of 2816 + the 512 the sum's store
// cost (V = 1), one launch instead of two, the direction products summed in the accumulator instead of by the reduction.
// 8 waves: wave (wm = wave & 3, wk = wave >> 2) owns M tile wm x K tiles 4 wk .. 4 wk + 3 of the 128 x 256 product; the waves wk == 0 also the
// 128 x 32 tile of the direction columns.  16-point blocks, double-buffered LDS; a block's global loads are issued one block AHEAD of the
// LDS stores that consume them (registers carry block b + 2 through iteration b + 1).
struct WgViewArgs {
    const float *dyv[1 + VIPNERF_MAX_SEC], *ped[1 + VIPNERF_MAX_SEC], *feat;
    const float *g[1 + VIPNERF_MAX_SEC], *dq[1 + VIPNERF_MAX_SEC];      // HEADS: the view hidden activations [P][128] and the head seeds [P][8] of every direction
    int64_t P;
    int chunk_pts, n_chunks;
    float *part_vf, *part_vd, *part_oh;       // chunk 0 of the partial products ([128][256] + 128 column sums; [128][32] + 128 unused; HEADS: [32][128] + 32 column sums)
    size_t stride_vf, stride_vd, stride_oh;
};
// one block's worth of staged operands in registers (thread tid: float4 (row tid >> 5, columns 4 (tid & 31)) of every dYv_a, float4 tid and
// tid + 512 of the feature block -- rows tid >> 6 and 8 + (tid >> 6) --, and, the first 128 threads, float4 (row tid >> 3, columns 4 (tid & 7)) of
// every gamma(dir_a)).  (HEADS: the view hidden g_a and the head seeds dq_a go HBM -> LDS by DMA instead: no registers left for them.)
template <int NV> struct ViewStage { float4 rv[NV], rf[2], rp[NV]; };
// FULL: the caller guarantees whole 16-point blocks (the heads-fused launch: P % 16 == 0, chunks of 32 points) -- the loads are then issued
// UNCONDITIONALLY, which is what the counted wait behind the head DMA relies on: `s_waitcnt vmcnt(NV + 2)` proves the DMA issued before these
// loads has landed only if every wave really issues its NV + 2 (waves 0-1: 2 NV + 2) younger loads; a load predicated on a per-lane
// validity test compiles to an execz-skipped branch, and a wave that skips one would pass the wait with a DMA piece still in flight
// (tools/isa_exec_mfma_scan.py::scan_counted_waits checks the generated code for exactly this)
template <int NV, bool FULL = false>
__device__ __forceinline__ void view_gload(ViewStage<NV> &r, const WgViewArgs &a, int64_t pb, int64_t p1, int tid) {
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool okv = FULL || pb + (tid >> 5) < p1;
#pragma unroll
    for (int k = 0; k < NV; ++k) { r.rv[k] = z4; if (okv) r.rv[k] = *(const float4 *)(a.dyv[k] + (size_t)(pb + (tid >> 5)) * WV + 4 * (tid & 31)); }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (tid >> 6) + 8 * i;
        r.rf[i] = z4;
        if (FULL || pb + row < p1) r.rf[i] = *(const float4 *)(a.feat + (size_t)(pb + row) * W + 4 * (tid & 63));
    }
    if (__builtin_amdgcn_readfirstlane(tid) < 128) {             // waves 0-1 (wave-uniform: a scalar branch, not a per-lane predicate)
        const bool okp = FULL || pb + (tid >> 3) < p1;
#pragma unroll
        for (int k = 0; k < NV; ++k) { r.rp[k] = z4; if (okp) r.rp[k] = *(const float4 *)(a.ped[k] + (size_t)(pb + (tid >> 3)) * DVE_PAD + 4 * (tid & 7)); }
    }
}
template <int NV, int O_SUM, int O_PED, int O_FEAT>
__device__ __forceinline__ void view_lstore(const ViewStage<NV> &r, float *t, int tid) {
    constexpr int BP = 16;
    float4 sum = r.rv[0];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        ((float4 *)(t + k * BP * WV))[tid] = r.rv[k];
        if (k > 0) { sum.x += r.rv[k].x; sum.y += r.rv[k].y; sum.z += r.rv[k].z; sum.w += r.rv[k].w; }
    }
    if (NV > 1) ((float4 *)(t + O_SUM))[tid] = sum;
#pragma unroll
    for (int i = 0; i < 2; ++i) ((float4 *)(t + O_FEAT))[tid + 512 * i] = r.rf[i];
    if (tid < 128) {
#pragma unroll
        for (int k = 0; k < NV; ++k) ((float4 *)(t + O_PED + k * BP * DVE_PAD))[tid] = r.rp[k];
    }
}
// HEADS (NV <= 2: what LDS holds): the output head's weight gradient dW_out[c][:] = sum_a sum_p dq_a[p][c] g_a[p][:] (4 x 128; reference
// VipNeRF01.py:591-596, views_output_linear over the view hidden g_a of every direction) rides in the same launch: the waves wk == 1 -- which
// have no direction tile -- take its 32 x 32 tiles (M = the seeds' 4 columns in a zero-padded 32, N tile wm of the 128 hidden units), so both
// waves of a SIMD run 8 (4 + NV) MFMAs per block; before, one more launch per level (k_wgrad<1, 1, 1>, 0.21 ms per step) read g_a and dq_a.
template <int NV, bool HEADS>
__global__ __launch_bounds__(512) void k_wgrad_view(WgViewArgs a) {
    constexpr int BP = 16;                                   // points per block
    constexpr int O_SUM = NV > 1 ? NV * BP * WV : 0;         // the sum tile (NV == 1: dYv_0 itself)
    constexpr int O_PED = O_SUM + (NV > 1 ? BP * WV : NV * BP * WV);
    constexpr int O_FEAT = O_PED + NV * BP * DVE_PAD;
    constexpr int O_G = O_FEAT + BP * W;
    constexpr int O_DQ = O_G + (HEADS ? NV * BP * WV : 0);
    constexpr int TILE_F = O_DQ + (HEADS ? NV * BP * 8 : 0);          // (the seeds' blocks as they lie in HBM: [16][8])
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if ((int)blockIdx.x >= a.n_chunks) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int wm = wave & 3, wk = wave >> 2;
    const int64_t p0 = (int64_t)blockIdx.x * a.chunk_pts;
    const int64_t p1 = p0 + a.chunk_pts < a.P ? p0 + a.chunk_pts : a.P;
    const int nblk = (int)((p1 - p0 + BP - 1) / BP);

    floatx16 acc[4], accd = (floatx16)(0.f);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (floatx16)(0.f);
    float bsum = 0.f, bsx = 0.f;
    // the per-direction product of this wave: wk == 0: dYv_a^T gamma(dir_a) (M tile wm x the 32 direction columns); wk == 1 (HEADS): dq_a^T g_a (the
    // seeds' padded tile x N tile wm of the hidden units) -- one code path, the operand tiles' offsets and row strides chosen once
    const bool heads_wave = HEADS && __builtin_amdgcn_readfirstlane(wk) == 1;      // (wave-uniform, scalar)
    const bool dir_wave = __builtin_amdgcn_readfirstlane(wk) == 0;
    const bool seed_lane = l31 < 4;                          // the seeds are 4 columns of a 32-row operand: the other lanes supply zeros
    // HEADS: block `pb`'s view hidden tiles (wave w: rows 2 w, 2 w + 1 of every direction = 1 KiB each) and seed blocks (wave 7's lanes 0..31: 512 B
    // each) by DMA into buffer `buf`.  Issued BEHIND the LDS stores of a staging step and AHEAD of its register loads: VM_CNT retires in order, so
    // the compiler's own counts for the register loads stay conservative, and vmcnt(younger register loads) in front of the block barrier
    // proves the DMA has landed.
    auto dma_heads = [&](int64_t pb, int buf) {
        if (HEADS) {
#pragma unroll
            for (int k = 0; k < NV; ++k)
                glds_chunks<1>(a.g[k] + (size_t)(pb + 2 * wave) * WV + lane * 4,
                               __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds + buf * TILE_F + O_G + k * BP * WV + wave * 256)));
            if (wave == 7 && lane < 32) {
#pragma unroll
                for (int k = 0; k < NV; ++k)
                    glds_chunks<1>(a.dq[k] + (size_t)pb * 8 + lane * 4, __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds + buf * TILE_F + O_DQ + k * BP * 8)));
            }
        }
    };

    if (HEADS) {
        if (nblk > 0) dma_heads(p0, 0);
    }
    // TWO register sets (block parity): a block's loads are issued two blocks ahead of the LDS stores that consume them
    ViewStage<NV> st0, st1;
    if (nblk > 0) { view_gload<NV>(st0, a, p0, p1, tid); view_lstore<NV, O_SUM, O_PED, O_FEAT>(st0, lds, tid); }
    if (nblk > 1) view_gload<NV>(st1, a, p0 + BP, p1, tid);
    if (nblk > 2) view_gload<NV>(st0, a, p0 + 2 * BP, p1, tid);
    __syncthreads();
    // block blk computes from LDS buffer blk & 1; behind k-step 1 (wk = 0) / 5 (wk = 1: the two waves of a SIMD stage at different k-steps, one's
    // LDS stores and load issue under the other's MFMAs) it stores block blk + 1 (register set (blk + 1) & 1, requested two blocks ago) into the
    // buffer the last barrier freed and requests block blk + 3 into the same registers
    for (int blk = 0; blk < nblk; blk += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int b = blk + half;
            if (b < nblk) {
                int toff = half * TILE_F;                    // (blk is even: buffer = half)
                if (HEADS) asm volatile("" : "+s"(toff));    // (opaque: folded into the reads' offsets, the second buffer's tiles -- beyond the 64 KiB an LDS instruction
                                                             // offset reaches -- each took an address register of their own: 39 spilled registers)
                const float *t = lds + toff;
#pragma unroll
                for (int s = 0; s < BP / 2; ++s) {
                    if (s == 1 + 4 * wk && b + 1 < nblk) {
                        ViewStage<NV> &nx = half == 0 ? st1 : st0;
                        view_lstore<NV, O_SUM, O_PED, O_FEAT>(nx, lds + (1 - half) * TILE_F, tid);
                        dma_heads(p0 + (int64_t)(b + 1) * BP, 1 - half);
                        if (b + 3 < nblk) view_gload<NV, HEADS>(nx, a, p0 + (int64_t)(b + 3) * BP, p1, tid);
                    }
                    const int row = 2 * s + h;
                    const float af = t[O_SUM + row * WV + 32 * wm + l31];
                    float bf[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) bf[j] = t[O_FEAT + row * W + 32 * (4 * wk + j) + l31];
                    bsum += af;
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = mfma32(af, bf[j], acc[j]);
                    if (HEADS || dir_wave) {                 // ONE accumulator and one MFMA site for both roles (in two branches the compiler kept two accumulators)
                        float x[NV], y[NV];
                        if (dir_wave) {
#pragma unroll
                            for (int k = 0; k < NV; ++k) { x[k] = t[k * BP * WV + row * WV + 32 * wm + l31]; y[k] = t[O_PED + k * BP * DVE_PAD + row * DVE_PAD + l31]; }
                        } else {
#pragma unroll
                            for (int k = 0; k < NV; ++k) {
                                const float q = t[O_DQ + k * BP * 8 + row * 8 + (l31 & 3)];
                                x[k] = seed_lane ? q : 0.f;
                                bsx += x[k];
                                y[k] = t[O_G + k * BP * WV + row * WV + 32 * wm + l31];
                            }
                        }
#pragma unroll
                        for (int k = 0; k < NV; ++k) accd = mfma32(x[k], y[k], accd);
                    }
                }
                if (HEADS) {                                 // this wave's DMA of block b + 1 has landed (younger: the NV + 2 register loads of block b + 3 every wave issues)
                    if (b + 3 < nblk) asm volatile("s_waitcnt vmcnt(%0) ; dma-landed-wait" ::"n"(NV + 2) : "memory");   // (the tag: what tests/test_isa_guards_cpu.py finds)
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
            }
        }
    }
    float *pf = a.part_vf + (size_t)blockIdx.x * a.stride_vf, *pd = a.part_vd + (size_t)blockIdx.x * a.stride_vd;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
        for (int j = 0; j < 4; ++j) pf[(size_t)o * W + 32 * (4 * wk + j) + l31] = acc[j][r];
        if (wk == 0) pd[(size_t)o * DVE_PAD + l31] = accd[r];
    }
    if (wk == 0) {
        const float b = bsum + __shfl_xor(bsum, 32, 64);
        if (h == 0) pf[(size_t)WV * W + 32 * wm + l31] = b;
    }
    if (HEADS) {
        if (wk == 1) {                                       // [32][128] (rows 0..3: the head's four outputs) + the seeds' 32 column sums
            float *po = a.part_oh + (size_t)blockIdx.x * a.stride_oh;
#pragma unroll
            for (int r = 0; r < 16; ++r) po[(size_t)((r & 3) + 8 * (r >> 2) + 4 * h) * WV + 32 * wm + l31] = accd[r];
            const float b = bsx + __shfl_xor(bsx, 32, 64);
            if (wm == 0 && h == 0) po[(size_t)32 * WV + l31] = b;
        }
    }
}
template <int NV>
static int launch_view(const WgViewArgs &va, bool heads, hipStream_t st) {
    if (heads) {
        if constexpr (NV <= 2) {
            const size_t ldsb = (size_t)2 * 16 * ((NV > 1 ? NV + 1 : 1) * WV + NV * DVE_PAD + W + NV * WV + NV * 8) * sizeof(float);
            VN_HIP(hipFuncSetAttribute((const void *)k_wgrad_view<NV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
            hipLaunchKernelGGL((k_wgrad_view<NV, true>), dim3(va.n_chunks), dim3(512), ldsb, st, va);
            VN_HIP(hipGetLastError());
            return VIPNERF_OK;
        } else { set_error("wgrad: the output head rides in the view launch for at most two directions"); return VIPNERF_E_ARG; }
    }
    const size_t ldsb = (size_t)2 * 16 * ((NV > 1 ? NV + 1 : 1) * WV + NV * DVE_PAD + W) * sizeof(float);
    VN_HIP(hipFuncSetAttribute((const void *)k_wgrad_view<NV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipLaunchKernelGGL((k_wgrad_view<NV, false>), dim3(va.n_chunks), dim3(512), ldsb, st, va);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

// The 256 x 256 fp32 GEMMs with EIGHT waves (two per SIMD): wave (wm, wk) owns 2 x 4 of the 8 x 8 tiles (128 accumulators), so a
// second wave keeps a SIMD's MFMA pipe busy while the first one issues its vector-memory instructions.  Why it matters: a
// global_load_dwordx4 holds a wave's issue port ~60 cycles whatever else is going on; the 4-wave kernel above issues 16 of them per
// wave and 32-point block -- ~960 of the block's 16384 MFMA cycles with nothing else to issue on that SIMD, wherever in the block they
// are placed (spreading them over the k-steps, or issuing the LDS stores at the block's start, measured no different: docs/HISTORY.md 5).
// Same tiles, same LDS layout, same partial-product format as k_wgrad<2, 8, 4>.
// (The 128 x 256 and 256 x 64 GEMMs were tried on this kernel too -- templated on the tile counts -- and measured the same 1.99 ms per step as
// on the 4-wave kernel above: they are not issue-bound; profiles/r04_ablation_pt2.md 8.)
TS_DECL(g_wg_timeline);            // VN_EXP == 50: per-block time stamps of one workgroup (tools/pt2_timeline.py wg)
#define TSW(tag) TS_AT(g_wg_timeline, tag)
// WC: this workgroup's GEMM carries a weighted-column-sum head (WgDesc::wcol).  Compiled as a second copy of the body that only the feature
// layer's workgroups enter: with the head's code behind a run-time flag in ONE body the other seven GEMMs' blocks measured 2 % slower.
// FAST: both operands are whole [P][256] arrays and the chunk is whole 32-point blocks (the render / training path: always) -- a block's loads are a
// scalar base (advanced on the scalar unit) plus a per-thread 32-bit offset that never changes, with no bounds logic in the loop: as one body with
// the general path behind run-time flags, every block carried ~50 vector instructions of 64-bit address arithmetic and range compares for loads it
// did not take (each one an issue slot an MFMA does not get).
template <int WK, bool WC, bool FAST>
__device__ __forceinline__ void wgrad256_body(const WgArgs &a) {
    constexpr int MTW = 2, KTW = 8 / WK, Mp = 256, Kp = 256, NTH = 256 * WK, NLD = 2048 / NTH;   // float4 of A and of B per thread and block
    constexpr int TILE_F = 32 * (Mp + Kp);
    constexpr bool VECFRAG = VN_WGRAD_VECFRAG && KTW == 4;
    typedef float __attribute__((ext_vector_type(2))) f2v;
    typedef float __attribute__((ext_vector_type(4))) f4v;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const WgDesc &d = a.d[blockIdx.y];
    if ((int)blockIdx.x >= d.n_chunks) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int wm = wave & 3, wk = wave >> 2;
    const int64_t p0 = (int64_t)blockIdx.x * a.chunk_pts;
    const int64_t p1 = p0 + a.chunk_pts < a.P ? p0 + a.chunk_pts : a.P;
    const int nblk = (int)((p1 - p0 + 31) / 32);
#if defined(VN_EXP) && VN_EXP == 50
    const bool ts_rec = blockIdx.x == gridDim.x / 2 && blockIdx.y == 3 && lane == 0 && wave < 8; int ts_n = 0;
#endif
    TSW(TS_ENTRY);

    floatx16 acc[MTW][KTW];
    float bsum[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        bsum[i] = 0.f;
#pragma unroll
        for (int j = 0; j < KTW; ++j) acc[i][j] = (floatx16)(0.f);
    }
    float4 ra[NLD], rb[NLD];
    // wcol (the sigma head riding in the feature layer's GEMM: WgDesc): w[p] for the rows this thread stages, the weighted column sums of its
    // four B columns over those rows, and sum w.  A thread's float4 i of a block is row (tid >> 6) + (NTH / 64) i, columns 4 (tid & 63) .. +3.
    constexpr bool wc = WC;
    float wr[NLD];
    float4 wsum = make_float4(0.f, 0.f, 0.f, 0.f);
    float wtot = 0.f;
    const bool linA = d.lda == Mp && d.m_load == Mp, linB = d.ldb == Kp && d.k_load == Kp;
    // the operand bases live in SGPRs for the whole kernel: left to the compiler they are re-read from the kernel arguments by every block's
    // gload (s_load + s_waitcnt lgkmcnt(0) -- which also drains the wave's LDS fragment reads in flight in the middle of its MFMA loop)
    // (as GLOBAL-address-space pointers: a pointer that went through an asm operand is generic to the compiler, and flat loads count in lgkmcnt too)
    typedef const float __attribute__((address_space(1))) *gfp;
    typedef const f4v __attribute__((address_space(1))) *gf4p;
    unsigned long long uA = (unsigned long long)d.A, uB = (unsigned long long)d.B, uW = (unsigned long long)d.wcol;
    int dwstride = d.wcol_stride;
    asm volatile("" : "+s"(uA), "+s"(uB), "+s"(uW), "+s"(dwstride));
    const gfp dA = (gfp)uA, dB = (gfp)uB, dwcol = (gfp)uW;
    // the head's weights of a block: lane i < NLD of every wave loads w[row (tid >> 6) + (NTH / 64) i] with ONE vector load BEHIND the operand
    // loads (so that the counted waits on those are what they were); lstore broadcasts the NLD values with v_readlane.  (As NLD loads ahead
    // of the operands, vector or scalar, the feature layer's workgroups ran ~15 % longer: profiles/r05_ab_thin_wgrad.log.)
    float wv = 0.f;
    auto wload = [&](int64_t pb) {
        if (wc) {
            const int64_t pr = pb + (tid >> 6) + (NTH / 64) * (lane < NLD ? lane : 0);
            wv = (lane < NLD && pr < p1) ? dwcol[(size_t)pr * dwstride] : 0.f;
        }
    };
    typedef const char __attribute__((address_space(1))) *gcp;
    const unsigned voff = (unsigned)tid * 16u;
    auto gload = [&](int blk) {
        const int64_t pb = p0 + (int64_t)blk * 32;
        if (FAST || (pb + 32 <= p1 && linA && linB)) {
            const gcp ba = (gcp)(dA + (size_t)pb * Mp), bb = (gcp)(dB + (size_t)pb * Kp);       // wave-uniform
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const f4v va = *(gf4p)(ba + (voff + (unsigned)(NTH * 16 * i))), vb = *(gf4p)(bb + (voff + (unsigned)(NTH * 16 * i)));
                ra[i] = make_float4(va.x, va.y, va.z, va.w); rb[i] = make_float4(vb.x, vb.y, vb.z, vb.w);
            }
            wload(pb);
            return;
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + NTH * i, row = idx / 64, col = 4 * (idx % 64);
            const bool oka = (pb + row < p1) && (col < d.m_load), okb = (pb + row < p1) && (col < d.k_load);
            ra[i] = oka ? *(const float4 *)(d.A + (size_t)(pb + row) * d.lda + col) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[i] = okb ? *(const float4 *)(d.B + (size_t)(pb + row) * d.ldb + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        wload(pb);
    };
    auto lstore = [&](int buf) {
        float4 *la = (float4 *)(lds + buf * TILE_F), *lb = (float4 *)(lds + buf * TILE_F + 32 * Mp);
#pragma unroll
        for (int i = 0; i < NLD; ++i) { la[tid + NTH * i] = ra[i]; lb[tid + NTH * i] = rb[i]; }
        if (wc) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) wr[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wv), i));
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                wsum.x = fmaf(wr[i], rb[i].x, wsum.x); wsum.y = fmaf(wr[i], rb[i].y, wsum.y);
                wsum.z = fmaf(wr[i], rb[i].z, wsum.z); wsum.w = fmaf(wr[i], rb[i].w, wsum.w);
                wtot += wr[i];
            }
        }
    };
    // VN_WGRAD_DMA (round 3): when both operands are whole [P][256] arrays and the chunk is whole 32-point blocks (the render path:
    // always), a block's 32 KiB + 32 KiB are contiguous in HBM and go HBM -> LDS by DMA (global_load_lds_dwordx4; the row-major LDS
    // tile IS the HBM image) instead of through registers: no ds_write_b128 phase in front of the block barrier, no staging registers.
    const bool dma = VN_WGRAD_DMA && linA && linB && (p1 - p0) % 32 == 0;
    auto issue = [&](int blk, int buf) {
        const float *ga = d.A + (size_t)(p0 + (int64_t)blk * 32) * Mp + lane * 4, *gb = d.B + (size_t)(p0 + (int64_t)blk * 32) * Kp + lane * 4;
        constexpr int PER_WAVE = 64 / (4 * WK);              // 1 KiB pieces per wave: 32 of A and 32 of B over 4 WK waves
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int pc = wave * PER_WAVE + i;              // pieces 0..31: A, 32..63: B
            const float *src = pc < 32 ? ga + pc * 256 : gb + (pc - 32) * 256;
            glds_chunks<1>(src, __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds + buf * TILE_F + pc * 256)));
        }
    };
    if (dma) {
        if (nblk > 0) issue(0, 0);
    } else {
        if (nblk > 0) { gload(0); lstore(0); }
        __syncthreads();
    }
    // Every wave fetches the next block behind k-step VN_WGRAD_LATE_LOAD (the second wave of a SIMD four steps later) and stores it to LDS
    // behind k-step VN_WGRAD_LATE_STORE (the second wave two steps later) instead of at the block's two ends: after the block barrier
    // both waves of a SIMD start their MFMAs at once, and at the block's end they go straight to the barrier.  Per-block timeline of one
    // workgroup (tools/pt2_timeline.py wg, profiles/r04_ablation_pt2.md 8): with the loads at the top and the stores at the end 1.9k of
    // a block's 19.7k cycles had no MFMA in flight on the SIMD (the block's MFMAs are 16.4k); now 18.8k per block.
    const bool late = VN_WGRAD_LATE_LOAD >= 0;
    const int second = (wave >> 2) & 1;
    f2v a2n = {0.f, 0.f}; f4v b4n = {0.f, 0.f, 0.f, 0.f};
    int cur = 0;
    // only the waves that store a bias sum (wk == 0: the OLDER wave of every SIMD) form it
    const bool bias_wave = __builtin_amdgcn_readfirstlane(wave >> 2) == 0;
    for (int blk = 0; blk < nblk; ++blk) {
        if (dma) {
            glds_drain();                                    // this wave's pieces of block blk (issued one block ago)
            __builtin_amdgcn_s_barrier();                    // every wave's pieces are in, every wave is done with the other buffer
            asm volatile("" ::: "memory");
            if (blk + 1 < nblk) issue(blk + 1, cur ^ 1);
        } else if (!late && blk + 1 < nblk) gload(blk + 1);
        TSW(TS_POST);
        const float *la = lds + cur * TILE_F, *lb = la + 32 * Mp;
        float afs[16][MTW];                                  // (VN_WGRAD_BIAS_WK0) the block's A fragments, summed behind its MFMAs by the waves that store the sums
        if (VECFRAG && VN_WGRAD_PREFETCH) {
            a2n = *(const f2v *)(la + h * Mp + 64 * wm + 2 * l31);
            b4n = *(const f4v *)(lb + h * Kp + 128 * wk + 4 * l31);
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (!dma && late && s == VN_WGRAD_LATE_LOAD + 4 * second && blk + 1 < nblk) gload(blk + 1);
            float af[MTW], bf[KTW];
            if (VECFRAG && VN_WGRAD_PREFETCH) {
                // the fragments of k-step s + 1 are requested BEFORE the MFMAs of k-step s (two registers sets): a wave alone on its SIMD -- the younger
                // one at the end of every block -- otherwise sits out the LDS latency once per k-step pair
                const f2v a2 = a2n; const f4v b4 = b4n;
                if (s + 1 < 16) {
                    a2n = *(const f2v *)(la + (2 * (s + 1) + h) * Mp + 64 * wm + 2 * l31);
                    b4n = *(const f4v *)(lb + (2 * (s + 1) + h) * Kp + 128 * wk + 4 * l31);
                }
                __builtin_amdgcn_sched_barrier(0);
                af[0] = a2.x; af[1] = a2.y;
                bf[0] = b4.x; bf[1] = b4.y; bf[2] = b4.z; bf[KTW - 1] = b4.w;
            } else if (VECFRAG) {
                // a wave's M tile i is features 64 wm + 2 l + i, its K tile j features 128 wk + 4 l + j (l = 0..31): the lane's fragments of a k-step are
                // ONE 8-byte and ONE 16-byte read of the row-major tile at immediate offsets from two per-block bases -- no address arithmetic among
                // the MFMAs (32 consecutive features per tile took three ds_read2_b32 and three v_add per k-step)
                const f2v a2 = *(const f2v *)(la + (2 * s + h) * Mp + 64 * wm + 2 * l31);
                const f4v b4 = *(const f4v *)(lb + (2 * s + h) * Kp + 128 * wk + 4 * l31);
                af[0] = a2.x; af[1] = a2.y;
                bf[0] = b4.x; bf[1] = b4.y; bf[2] = b4.z; bf[KTW - 1] = b4.w;
            } else {
#pragma unroll
            for (int i = 0; i < MTW; ++i) af[i] = la[(2 * s + h) * Mp + 32 * (wm * MTW + i) + l31];
#pragma unroll
            for (int j = 0; j < KTW; ++j) bf[j] = lb[(2 * s + h) * Kp + 32 * (wk * KTW + j) + l31];
            }
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                if (!VN_WGRAD_BIAS_WK0) bsum[i] += af[i];
                if (VN_WGRAD_BIAS_WK0) afs[s][i] = af[i];
#pragma unroll
                for (int j = 0; j < KTW; ++j) acc[i][j] = mfma32(af[i], bf[j], acc[i][j]);
            }
            if (VECFRAG && VN_WGRAD_PREFETCH) __builtin_amdgcn_sched_barrier(0);
            if (!dma && late && s == VN_WGRAD_LATE_STORE + VN_WGRAD_STORE_SKEW * second && blk + 1 < nblk) lstore(cur ^ 1);
        }
        if (VN_WGRAD_BIAS_WK0 && bias_wave) {
            asm volatile("" ::);                             // (a real scalar branch: as a select the other waves would form the sums too)
#pragma unroll
            for (int s = 0; s < 16; ++s)
#pragma unroll
                for (int i = 0; i < MTW; ++i) bsum[i] += afs[s][i];
        }
        TSW(TS_END);
        if (!dma) {
            if (!late && blk + 1 < nblk) lstore(cur ^ 1);
            TSW(TS_EPI_A);
            __syncthreads();
        }
        TSW(TS_PRE);
        cur ^= 1;
    }
    float *part = a.partial + d.part_off + (size_t)blockIdx.x * d.part_stride;
    if (VECFRAG) {
#pragma unroll
        for (int i = 0; i < MTW; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = 64 * wm + 2 * ((r & 3) + 8 * (r >> 2) + 4 * h) + i;
                *(float4 *)(part + (size_t)o * Kp + 128 * wk + 4 * l31) = make_float4(acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][KTW - 1][r]);
            }
            if (wk == 0) {
                const float b = bsum[i] + __shfl_xor(bsum[i], 32, 64);
                if (h == 0) part[(size_t)Mp * Kp + 64 * wm + 2 * l31 + i] = b;
            }
        }
    } else
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int ot = wm * MTW + i;
#pragma unroll
        for (int j = 0; j < KTW; ++j) {
            const int kt = wk * KTW + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = 32 * ot + (r & 3) + 8 * (r >> 2) + 4 * h;
                part[(size_t)o * Kp + 32 * kt + l31] = acc[i][j][r];
            }
        }
        if (wk == 0) {
            const float b = bsum[i] + __shfl_xor(bsum[i], 32, 64);
            if (h == 0) part[(size_t)Mp * Kp + 32 * ot + l31] = b;
        }
    }
    if (wc) {      // the head's partial behind the bias sums: [256 weighted column sums][sum w]; the NTH / 64 row groups (waves) summed in order
        static_assert(!VN_WGRAD_DMA, "wcol needs the operands in registers");
        __syncthreads();                                     // (every wave is done with the last block's tiles)
        float4 *ws4 = (float4 *)lds;
        ws4[tid] = wsum;
        if (lane == 0) lds[4 * NTH + wave] = wtot;
        __syncthreads();
        if (wave == 0) {
            float4 t = ws4[lane];
#pragma unroll
            for (int w = 1; w < NTH / 64; ++w) { const float4 u = ws4[64 * w + lane]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
            *(float4 *)(part + (size_t)Mp * Kp + Mp + 4 * lane) = t;
            if (lane == 0) {
                float tt = lds[4 * NTH];
#pragma unroll
                for (int w = 1; w < NTH / 64; ++w) tt += lds[4 * NTH + w];
                part[(size_t)Mp * Kp + Mp + 256] = tt;
            }
        }
    }
}
template <int WK>       // waves along K: 2 -> 8 waves (2 x 4 tiles each), 4 -> 16 waves (2 x 2 tiles each)
__global__ __launch_bounds__(256 * WK) void k_wgrad256_w8(WgArgs a) {
    const WgDesc &d = a.d[blockIdx.y];
    const int64_t p0 = (int64_t)blockIdx.x * a.chunk_pts;
    const int64_t p1 = p0 + a.chunk_pts < a.P ? p0 + a.chunk_pts : a.P;
    const bool fast = d.lda == 256 && d.m_load == 256 && d.ldb == 256 && d.k_load == 256 && (p1 - p0) % 32 == 0 && !VN_WGRAD_DMA && VN_WGRAD_FAST;
    if (d.wcol != nullptr) { if (fast) wgrad256_body<WK, true, true>(a); else wgrad256_body<WK, true, false>(a); }
    else { if (fast) wgrad256_body<WK, false, true>(a); else wgrad256_body<WK, false, false>(a); }
}

// Weight-gradient GEMMs on split-precision bf16 MFMA ("bf16x3": hi/lo parts, 3 cross terms, fp32 accumulate),
// for M = 32*MT, K = 32*KT with MT in {4, 8} (4 waves split the M tiles), KT in {2, 8}.  The fp32 rows of A and B
// are split while they are staged: a staging task is (4 consecutive features, one 8-point slot); the thread loads
// those 8 float4, converts to two bf16 planes and writes, per feature and plane, the 8 points as ONE 16-byte LDS
// store -- the transposition the fragment needs (lane = feature, 8 consecutive k = points) happens in that write.
// LDS plane layout: [feature][4 slots of 8 points], slot XOR-swizzled by (feature >> 2) & 3 so that the 16-lane
// groups of ds_read_b128 hit 16 distinct slots.  Point q of a 32-point block sits in slot q & 3, element q >> 2:
// a fixed permutation of the contraction index, identical for A and B.
template <int MT, int KT, int NSET>
__global__ __launch_bounds__(256) void k_wgrad_bf16x3(WgArgs a) {
    constexpr int MTW = MT / 4, Mp = 32 * MT, Kp = 32 * KT;
    constexpr int PA = Mp * 64, PB = Kp * 64;             // bytes of one plane (one part) of A / B
    constexpr int BUF = 2 * PA + 2 * PB;                  // A hi, A lo, B hi, B lo
    constexpr int TA = Mp / 4 * 4, TB = Kp / 4 * 4;       // staging tasks: (features / 4) x 4 slots = Mp, Kp
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char *lb = (char *)lds;
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

    const WgDesc &d = a.d[blockIdx.y];
    if ((int)blockIdx.x >= d.n_chunks) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int64_t p0 = (int64_t)blockIdx.x * a.chunk_pts;
    const int64_t p1 = p0 + a.chunk_pts < a.P ? p0 + a.chunk_pts : a.P;
    const int nblk = (int)((p1 - p0 + 31) / 32);

    floatx16 acc[MTW][KT];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j) acc[i][j] = (floatx16)(0.f);
    float bs[4] = {0.f, 0.f, 0.f, 0.f};                   // column sums of A: this thread's 4 features over its slot's points

    // task of this thread for A / B: feature group c (features 4c..4c+3) and slot sl (points sl + 4e, e = 0..7)
    const bool hasA = tid < TA, hasB = tid < TB;
    const int cA = tid % (Mp / 4), slA = tid / (Mp / 4), cB = tid % (Kp / 4), slB = tid / (Kp / 4);
    // NSET register sets: block b is loaded into set b % NSET, and the set is re-armed (block b + NSET) as soon as it has
    // been staged, so a workgroup keeps NSET 32-point blocks in flight.  These GEMMs have few MFMAs per byte: what bounds
    // them is how much they have outstanding against the HBM latency -- from a second set where the accumulators leave
    // room for it without costing a co-resident workgroup (128x256: 253 vs 276 us per launch; 256x64, which runs two
    // workgroups per CU on 160 registers, loses with 268: 369 vs 351 us).
    float4 ra[NSET][8], rb[NSET][8];
    const bool fullA = d.m_load == Mp, fullB = d.k_load == Kp;
    auto gload = [&](float4 (&qa)[8], float4 (&qb)[8], int blk) {
        const int64_t pb = p0 + (int64_t)blk * 32;
        if (pb + 32 <= p1 && fullA && fullB) {             // whole block in range, all columns valid: no predication
            if (hasA) {
                const float4 *ta = (const float4 *)(d.A + (size_t)(pb + slA) * d.lda) + cA;
#pragma unroll
                for (int e = 0; e < 8; ++e) qa[e] = ta[(size_t)e * d.lda];          // rows slA + 4e: 4*lda floats = lda float4
            }
            if (hasB) {
                const float4 *tb = (const float4 *)(d.B + (size_t)(pb + slB) * d.ldb) + cB;
#pragma unroll
                for (int e = 0; e < 8; ++e) qb[e] = tb[(size_t)e * d.ldb];
            }
            return;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int64_t rA = pb + slA + 4 * e, rB = pb + slB + 4 * e;
            qa[e] = (hasA && rA < p1 && 4 * cA < d.m_load) ? *((const float4 *)(d.A + (size_t)rA * d.lda) + cA) : make_float4(0.f, 0.f, 0.f, 0.f);
            qb[e] = (hasB && rB < p1 && 4 * cB < d.k_load) ? *((const float4 *)(d.B + (size_t)rB * d.ldb) + cB) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto put = [&](char *plane, int psize, int f, int sl, const float (&x)[8]) {   // 8 points of feature f -> hi / lo planes
        bf16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) { hi[e] = (__bf16)x[e]; lo[e] = (__bf16)(x[e] - (float)hi[e]); }
        const int off = psize == PA ? wg_off<Mp>(f, sl) : wg_off<Kp>(f, sl);
        *(bf16x8 *)(plane + off) = hi;
        *(bf16x8 *)(plane + psize + off) = lo;
    };
    const bool a_split = d.a_split16 != 0;
    auto lstore = [&](const float4 (&qa)[8], const float4 (&qb)[8], int buf) {
        char *base = lb + buf * BUF;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float xa[8], xb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (a_split) {                           // feature c = hi + lo, halves (c & 1) of words c >> 1 and 2 + (c >> 1)
                    typedef _Float16 half2_ __attribute__((ext_vector_type(2)));
                    const half2_ hh = __builtin_bit_cast(half2_, c < 2 ? qa[i].x : qa[i].y), ll = __builtin_bit_cast(half2_, c < 2 ? qa[i].z : qa[i].w);
                    xa[i] = (float)hh[c & 1] + (float)ll[c & 1];
                } else {
                    xa[i] = c == 0 ? qa[i].x : (c == 1 ? qa[i].y : (c == 2 ? qa[i].z : qa[i].w));
                }
                xb[i] = c == 0 ? qb[i].x : (c == 1 ? qb[i].y : (c == 2 ? qb[i].z : qb[i].w));
                bs[c] += xa[i];
            }
            if (hasA) put(base, PA, 4 * cA + c, slA, xa);
            if (hasB) put(base + 2 * PA, PB, 4 * cB + c, slB, xb);
        }
    };
    // the MFMAs of block blk (LDS buffer blk & 1), then stage block blk + 1 from its register set and re-arm the set
    auto iter = [&](float4 (&qa)[8], float4 (&qb)[8], int blk) {
        const char *base = lb + (blk & 1) * BUF;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int slot = 2 * ks + h;
            bf16x8 af[MTW][2], bf[KT][2];
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                const int f = 32 * (wave * MTW + i) + l31;
                const int off = wg_off<Mp>(f, slot);
                af[i][0] = *(const bf16x8 *)(base + off);
                af[i][1] = *(const bf16x8 *)(base + PA + off);
            }
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                const int f = 32 * j + l31;
                const int off = wg_off<Kp>(f, slot);
                bf[j][0] = *(const bf16x8 *)(base + 2 * PA + off);
                bf[j][1] = *(const bf16x8 *)(base + 2 * PA + PB + off);
            }
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int j = 0; j < KT; ++j) {
                    floatx16 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        }
        if (blk + 1 < nblk) {
            lstore(qa, qb, (blk + 1) & 1);
            if (blk + 1 + NSET < nblk) gload(qa, qb, blk + 1 + NSET);
        }
        __syncthreads();
    };

    if (nblk > 0) {
#pragma unroll
        for (int k = 0; k < NSET; ++k)
            if (k < nblk) gload(ra[k], rb[k], k);
        lstore(ra[0], rb[0], 0);
        if (NSET < nblk) gload(ra[0], rb[0], NSET);
    }
    __syncthreads();
    for (int blk = 0; blk < nblk; blk += NSET) {          // iteration b stages block b + 1 from set (b + 1) % NSET
#pragma unroll
        for (int k = 0; k < NSET; ++k)
            if (blk + k < nblk) iter(ra[(k + 1) % NSET], rb[(k + 1) % NSET], blk + k);
    }

    float *part = a.partial + d.part_off + (size_t)blockIdx.x * d.part_stride;
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int ot = wave * MTW + i;
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = 32 * ot + (r & 3) + 8 * (r >> 2) + 4 * h;
                part[(size_t)o * Kp + 32 * j + l31] = acc[i][j][r];
            }
    }
    // bias column sums: each A task holds the sums of ITS slot's points for 4 features -> fold the 4 slots through LDS
    float *red = (float *)lb;                              // all fragment reads are behind the loop's last barrier
    if (hasA)
#pragma unroll
        for (int c = 0; c < 4; ++c) red[slA * Mp + 4 * cA + c] = bs[c];
    __syncthreads();
    if (tid < Mp) part[(size_t)Mp * Kp + tid] = (red[tid] + red[Mp + tid]) + (red[2 * Mp + tid] + red[3 * Mp + tid]);
}

template <int MT, int KT, int NSET>
static int launch_bf16x3(const WgArgs &args, int n_desc, int n_chunks, hipStream_t st) {
    if (n_desc == 0) return VIPNERF_OK;
    const size_t ldsb = (size_t)2 * 2 * (32 * MT + 32 * KT) * 64;
    VN_HIP(hipFuncSetAttribute((const void *)k_wgrad_bf16x3<MT, KT, NSET>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipLaunchKernelGGL((k_wgrad_bf16x3<MT, KT, NSET>), dim3(n_chunks, n_desc), dim3(256), ldsb, st, args);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}


// The 256x256 class when both operands were STORED as fp16 ([P][256] halves: the trunk activations h_1..h_8 and the
// gradients dY_1..dY_7, dY_feature).  PARTS = 1 (FP16X3H): high parts only, ONE v_mfma_f32_32x32x16_f16 per product --
// half the bytes and a third of the MFMAs of the bf16 hi/lo kernel.  PARTS = 2 (FP16X3): a second plane of low parts
// follows the first in the array's slot (P * 128 floats further), three cross terms -- the bytes of the fp32 arrays, but
// no split and no conversion while staging: a thread gathers the 8 points of a feature from its 8 row registers with
// v_perm and writes them as one 16-byte LDS store per plane (same [feature][4 slots of 8 points] planes and swizzle).
// BF: the 16-bit values are bf16 (VIPNERF_PREC_BF16, v_mfma_f32_32x32x16_bf16) instead of fp16 -- same bytes, same staging.
template <bool HAS_W, int PARTS, bool BF = false>
__device__ __forceinline__ void wgrad_h16_256_body(const WgArgs &a) {
    constexpr int MTW = 2, KTW = 8, Mp = 256, Kp = 256;
    constexpr int PLANE = 256 * 64;                       // bytes: one operand, one part, 256 features x 32 points x 2 B
    constexpr int BUF = 2 * PARTS * PLANE;                // A parts, then B parts
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char *lb = (char *)lds;
    static_assert(!BF || PARTS == 1, "bf16 storage is high parts only");
    typedef typename std::conditional<BF, __bf16, _Float16>::type el16;
    typedef el16 half8_ __attribute__((ext_vector_type(8)));

    const WgDesc &d = a.d[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int64_t p0 = (int64_t)blockIdx.x * a.chunk_pts;
    const int64_t p1 = p0 + a.chunk_pts < a.P ? p0 + a.chunk_pts : a.P;
    const int nblk = (int)((p1 - p0 + 31) / 32);

    floatx16 acc[MTW][KTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < KTW; ++j) acc[i][j] = (floatx16)(0.f);
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    float ws[4] = {0.f, 0.f, 0.f, 0.f}, wsum = 0.f;
    float wv[8];

    uint2 ua[PARTS][8], ub[PARTS][8];                     // rows wave + 4 i, features 4 lane .. + 3 (4 halves = 8 bytes)
    const uint2 *A2 = (const uint2 *)d.A, *B2 = (const uint2 *)d.B;   // a row = 256 halves = 64 uint2
    const size_t plane2 = (size_t)a.P * 64;               // second (low-part) plane, in uint2
    auto gload = [&](int blk) {
        const int64_t pb = p0 + (int64_t)blk * 32;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t row = pb + wave + 4 * i;
            const bool ok = row < p1;
#pragma unroll
            for (int pt = 0; pt < PARTS; ++pt) {
                ua[pt][i] = ok ? A2[pt * plane2 + (size_t)row * 64 + lane] : make_uint2(0u, 0u);
                ub[pt][i] = ok ? B2[pt * plane2 + (size_t)row * 64 + lane] : make_uint2(0u, 0u);
            }
            if (HAS_W) wv[i] = ok ? d.wcol[(size_t)row * d.wcol_stride] : 0.f;
        }
    };
    auto half_of = [](const uint2 &u, int c) -> float {   // feature c (0..3) of a row's 8 bytes
        const unsigned w = c < 2 ? u.x : u.y;
        const unsigned short bits = (unsigned short)((c & 1) ? (w >> 16) : (w & 0xffffu));
        if (BF) return __uint_as_float((unsigned)bits << 16);
        return (float)__builtin_bit_cast(_Float16, bits);
    };
    // the 8 points of feature c: dword k packs rows 2k (low half) and 2k + 1 (high half)
    auto gather = [](const uint2 (&u)[8], int c, uint4 &o) {
        const unsigned sel = (c & 1) ? 0x07060302u : 0x05040100u;
        unsigned r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned lo = c < 2 ? u[2 * k].x : u[2 * k].y, hi = c < 2 ? u[2 * k + 1].x : u[2 * k + 1].y;
            r[k] = __builtin_amdgcn_perm(hi, lo, sel);
        }
        o = make_uint4(r[0], r[1], r[2], r[3]);
    };
    auto lstore = [&](int buf) {
        char *base = lb + buf * BUF;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int f = 4 * lane + c;
            const int off = wg_off<256>(f, wave);
#pragma unroll
            for (int pt = 0; pt < PARTS; ++pt) {
                uint4 pa, pb;
                gather(ua[pt], c, pa);
                gather(ub[pt], c, pb);
                *(uint4 *)(base + pt * PLANE + off) = pa;
                *(uint4 *)(base + (PARTS + pt) * PLANE + off) = pb;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float av = half_of(ua[0][i], c), bv = HAS_W ? half_of(ub[0][i], c) : 0.f;
                if (PARTS == 2) { av += half_of(ua[PARTS - 1][i], c); if (HAS_W) bv += half_of(ub[PARTS - 1][i], c); }
                bs[c] += av;
                if (HAS_W) ws[c] = fmaf(wv[i], bv, ws[c]);
            }
        }
        if (HAS_W) {
#pragma unroll
            for (int i = 0; i < 8; ++i) wsum += wv[i];
        }
    };

    if (nblk > 0) { gload(0); lstore(0); }
    __syncthreads();
    int cur = 0;
    for (int blk = 0; blk < nblk; ++blk) {
        if (blk + 1 < nblk) gload(blk + 1);
        const char *base = lb + cur * BUF;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int slot = 2 * ks + h;
            half8_ af[MTW][PARTS], bf[KTW][PARTS];
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                const int f = 32 * (wave * MTW + i) + l31;
#pragma unroll
                for (int pt = 0; pt < PARTS; ++pt) af[i][pt] = *(const half8_ *)(base + pt * PLANE + wg_off<256>(f, slot));
            }
#pragma unroll
            for (int j = 0; j < KTW; ++j) {
                const int f = 32 * j + l31;
#pragma unroll
                for (int pt = 0; pt < PARTS; ++pt) bf[j][pt] = *(const half8_ *)(base + (PARTS + pt) * PLANE + wg_off<256>(f, slot));
            }
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int j = 0; j < KTW; ++j) {
                    floatx16 c = acc[i][j];
                    if (PARTS == 2) {
                        c = mfma16_32(af[i][PARTS - 1], bf[j][0], c);
                        c = mfma16_32(af[i][0], bf[j][PARTS - 1], c);
                    }
                    c = mfma16_32(af[i][0], bf[j][0], c);
                    acc[i][j] = c;
                }
        }
        if (blk + 1 < nblk) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    float *part = a.partial + d.part_off + (size_t)blockIdx.x * d.part_stride;
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int ot = wave * MTW + i;
#pragma unroll
        for (int j = 0; j < KTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = 32 * ot + (r & 3) + 8 * (r >> 2) + 4 * h;
                part[(size_t)o * Kp + 32 * j + l31] = acc[i][j][r];
            }
    }
    float *red = (float *)lb;
#pragma unroll
    for (int c = 0; c < 4; ++c) red[wave * 256 + 4 * lane + c] = bs[c];
    __syncthreads();
    part[(size_t)Mp * Kp + tid] = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
    if (HAS_W) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) red[wave * 256 + 4 * lane + c] = ws[c];
        if (lane == 0) red[1024 + wave] = wsum;
        __syncthreads();
        part[(size_t)Mp * Kp + Mp + tid] = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
        if (tid == 0) part[(size_t)Mp * Kp + Mp + 256] = (red[1024] + red[1025]) + (red[1026] + red[1027]);
    }
}
// FP16X3 with VN_F16_PRESPLIT: both operands stored already split, in the fp32 arrays' own geometry -- the 16 bytes of
// (row, 4 features) hold [hi(f0,f1)] [hi(f2,f3)] [lo(f0,f1)] [lo(f2,f3)] (store_pair_split).  Same 16-byte loads as the
// fp32 layout, staging = v_perm gathers (8 per feature), three fp16 cross terms.
template <bool HAS_W>
__device__ __forceinline__ void wgrad_split16_256_body(const WgArgs &a) {
    constexpr int MTW = 2, KTW = 8, Mp = 256, Kp = 256;
    constexpr int PLANE = 256 * 64;
    constexpr int BUF = 4 * PLANE;                        // A hi, A lo, B hi, B lo
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char *lb = (char *)lds;
    typedef _Float16 half8_ __attribute__((ext_vector_type(8)));

    const WgDesc &d = a.d[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int64_t p0 = (int64_t)blockIdx.x * a.chunk_pts;
    const int64_t p1 = p0 + a.chunk_pts < a.P ? p0 + a.chunk_pts : a.P;
    const int nblk = (int)((p1 - p0 + 31) / 32);

    floatx16 acc[MTW][KTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < KTW; ++j) acc[i][j] = (floatx16)(0.f);
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    float ws[4] = {0.f, 0.f, 0.f, 0.f}, wsum = 0.f;
    float wv[8];

    uint4 ua[8], ub[8];                                   // rows wave + 4 i, features 4 lane .. + 3
    auto gload = [&](int blk) {
        const int64_t pb = p0 + (int64_t)blk * 32;
        if (HAS_W) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t row = pb + wave + 4 * i;
                wv[i] = row < p1 ? d.wcol[(size_t)row * d.wcol_stride] : 0.f;
            }
        }
        if (pb + 32 <= p1) {
            const uint4 *ta = (const uint4 *)(d.A + (size_t)pb * Mp) + tid;
            const uint4 *tb = (const uint4 *)(d.B + (size_t)pb * Kp) + tid;
#pragma unroll
            for (int i = 0; i < 8; ++i) { ua[i] = ta[256 * i]; ub[i] = tb[256 * i]; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t row = pb + wave + 4 * i;
                const bool ok = row < p1;
                ua[i] = ok ? *((const uint4 *)(d.A + (size_t)row * Mp) + lane) : make_uint4(0u, 0u, 0u, 0u);
                ub[i] = ok ? *((const uint4 *)(d.B + (size_t)row * Kp) + lane) : make_uint4(0u, 0u, 0u, 0u);
            }
        }
    };
    auto word_of = [](const uint4 &u, int c, int part) -> unsigned {   // the dword holding feature c of part (0 hi, 1 lo)
        return part == 0 ? (c < 2 ? u.x : u.y) : (c < 2 ? u.z : u.w);
    };
    auto value_of = [&](const uint4 &u, int c) -> float {  // hi + lo of feature c
        const unsigned wh = word_of(u, c, 0), wl = word_of(u, c, 1);
        const unsigned short bh = (unsigned short)((c & 1) ? (wh >> 16) : (wh & 0xffffu));
        const unsigned short bl = (unsigned short)((c & 1) ? (wl >> 16) : (wl & 0xffffu));
        return (float)__builtin_bit_cast(_Float16, bh) + (float)__builtin_bit_cast(_Float16, bl);
    };
    auto gather = [&](const uint4 (&u)[8], int c, int part, uint4 &o) {
        const unsigned sel = (c & 1) ? 0x07060302u : 0x05040100u;
        unsigned r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = __builtin_amdgcn_perm(word_of(u[2 * k + 1], c, part), word_of(u[2 * k], c, part), sel);
        o = make_uint4(r[0], r[1], r[2], r[3]);
    };
    auto lstore = [&](int buf) {
        char *base = lb + buf * BUF;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int f = 4 * lane + c;
            const int off = wg_off<256>(f, wave);
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                uint4 pa, pb;
                gather(ua, c, pt, pa);
                gather(ub, c, pt, pb);
                *(uint4 *)(base + pt * PLANE + off) = pa;
                *(uint4 *)(base + (2 + pt) * PLANE + off) = pb;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bs[c] += value_of(ua[i], c);
                if (HAS_W) ws[c] = fmaf(wv[i], value_of(ub[i], c), ws[c]);
            }
        }
        if (HAS_W) {
#pragma unroll
            for (int i = 0; i < 8; ++i) wsum += wv[i];
        }
    };

    if (nblk > 0) { gload(0); lstore(0); }
    __syncthreads();
    int cur = 0;
    for (int blk = 0; blk < nblk; ++blk) {
        if (blk + 1 < nblk) gload(blk + 1);
        const char *base = lb + cur * BUF;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int slot = 2 * ks + h;
            half8_ af[MTW][2], bf[KTW][2];
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                const int f = 32 * (wave * MTW + i) + l31;
                const int off = wg_off<256>(f, slot);
                af[i][0] = *(const half8_ *)(base + off);
                af[i][1] = *(const half8_ *)(base + PLANE + off);
            }
#pragma unroll
            for (int j = 0; j < KTW; ++j) {
                const int f = 32 * j + l31;
                const int off = wg_off<256>(f, slot);
                bf[j][0] = *(const half8_ *)(base + 2 * PLANE + off);
                bf[j][1] = *(const half8_ *)(base + 3 * PLANE + off);
            }
#pragma unroll
            for (int i = 0; i < MTW; ++i)
#pragma unroll
                for (int j = 0; j < KTW; ++j) {
                    floatx16 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][1], bf[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][0], bf[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i][0], bf[j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        }
        if (blk + 1 < nblk) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    float *part = a.partial + d.part_off + (size_t)blockIdx.x * d.part_stride;
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        const int ot = wave * MTW + i;
#pragma unroll
        for (int j = 0; j < KTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = 32 * ot + (r & 3) + 8 * (r >> 2) + 4 * h;
                part[(size_t)o * Kp + 32 * j + l31] = acc[i][j][r];
            }
    }
    float *red = (float *)lb;
#pragma unroll
    for (int c = 0; c < 4; ++c) red[wave * 256 + 4 * lane + c] = bs[c];
    __syncthreads();
    part[(size_t)Mp * Kp + tid] = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
    if (HAS_W) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) red[wave * 256 + 4 * lane + c] = ws[c];
        if (lane == 0) red[1024 + wave] = wsum;
        __syncthreads();
        part[(size_t)Mp * Kp + Mp + tid] = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
        if (tid == 0) part[(size_t)Mp * Kp + Mp + 256] = (red[1024] + red[1025]) + (red[1026] + red[1027]);
    }
}
// The pre-split GEMM with the staging of block n+1 and the reloads for block n+2 issued in the shadow of block n's MFMAs.
// With one wave per SIMD (256 accumulators) nothing else can fill that time, and measured piece by piece (this kernel's
// fp32-input sibling on the same box: MFMAs alone 1.67 ms per step, + fragment reads 1.96, + staging 2.90, loads + MFMAs
// without staging 2.95, everything 3.7-3.9) the staging VALU does not hide for free: a wave issues one instruction per
// 4 cycles, a 32x32x16 MFMA leaves ~6 slots, and conversions (v_cvt_pk_*: 8 cycles) use two (tools/
// mfma_valu_coissue.hip).  Hence: operands that arrive already split (a v_perm gather, no conversion), bias sums on
// v_dot2_f32_f16 (one instruction adds one stored half to an fp32 sum), and an explicit schedule:
//   * fragment order j-outer / i-inner: a B fragment pair lives for six MFMAs, so four pairs (32 registers) are in
//     flight instead of all eight of a k-step (64); the A fragments of both k-steps stay resident (32)
//   * eight super-steps of two (k-step, j) cells = 12 MFMAs; super-step u carries staging piece u (operand, component
//     c: the 8 points of one feature per lane -> two 16-byte LDS stores) and the fragment reads of super-step u + 1;
//     an operand's 8 row registers are reloaded (block n+2) right after its fourth piece -- no second register set
//   * the main loop is branch-free (full blocks only, reload index clamped); the last full block and a ragged tail run
//     without staging
template <bool HAS_W>
__device__ __forceinline__ void wgrad_split16_256_pipe(const WgArgs &a) {
    constexpr int Mp = 256, Kp = 256;
    constexpr int PLANE = 256 * 64;
    constexpr int BUF = 4 * PLANE;                        // A hi, A lo, B hi, B lo
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char *lb = (char *)lds;
    typedef _Float16 half8_ __attribute__((ext_vector_type(8)));
    typedef _Float16 half2_ __attribute__((ext_vector_type(2)));

    const WgDesc &d = a.d[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, l31 = lane & 31;
    const int64_t p0 = (int64_t)blockIdx.x * a.chunk_pts;
    const int64_t p1 = p0 + a.chunk_pts < a.P ? p0 + a.chunk_pts : a.P;
    const int nfull = (int)((p1 - p0) / 32);
    const bool tail = ((p1 - p0) & 31) != 0;

    floatx16 acc[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (floatx16)(0.f);
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    float ws[4] = {0.f, 0.f, 0.f, 0.f}, wsum = 0.f;
    float wv[8];
    uint4 ua[8], ub[8];                                   // rows wave + 4 i, features 4 lane .. + 3: [hi01][hi23][lo01][lo23]

    // fragment read offsets (wg_off<256>(32 tt + l31, 2 ks + h)): e0 ^ (((ks ^ tt) & 1) << 5), + 512 tt
    const int e0 = ((l31 & 3) * 64 + (l31 >> 2)) * 64 + (((h ^ (l31 & 3) ^ (l31 >> 4)) & 3) << 4);
    // staging store offset of feature 4 lane + c, slot wave: (w0 ^ (c << 4)) + 4096 c
    const int w0 = lane * 64 + (((wave ^ (lane >> 2)) & 3) << 4);

    auto load_a = [&](int blk) {                          // full blocks only
        const uint4 *ta = (const uint4 *)(d.A + (size_t)(p0 + (int64_t)blk * 32) * Mp) + tid;
#pragma unroll
        for (int i = 0; i < 8; ++i) ua[i] = ta[256 * i];
    };
    auto load_b = [&](int blk) {
        const int64_t pb = p0 + (int64_t)blk * 32;
        const uint4 *tb = (const uint4 *)(d.B + (size_t)pb * Kp) + tid;
#pragma unroll
        for (int i = 0; i < 8; ++i) ub[i] = tb[256 * i];
        if (HAS_W) {
#pragma unroll
            for (int i = 0; i < 8; ++i) wv[i] = d.wcol[(size_t)(pb + wave + 4 * i) * d.wcol_stride];
        }
    };
    auto load_tail = [&](int blk) {                       // ragged last block: rows beyond p1 read as zero
        const int64_t pb = p0 + (int64_t)blk * 32;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t row = pb + wave + 4 * i;
            const bool ok = row < p1;
            ua[i] = ok ? *((const uint4 *)(d.A + (size_t)row * Mp) + lane) : make_uint4(0u, 0u, 0u, 0u);
            ub[i] = ok ? *((const uint4 *)(d.B + (size_t)row * Kp) + lane) : make_uint4(0u, 0u, 0u, 0u);
            if (HAS_W) wv[i] = ok ? d.wcol[(size_t)row * d.wcol_stride] : 0.f;
        }
    };
    auto word_of = [](const uint4 &u, int c, int part) -> unsigned {   // the dword holding feature c of part (0 hi, 1 lo)
        return part == 0 ? (c < 2 ? u.x : u.y) : (c < 2 ? u.z : u.w);
    };
    // staging piece u of the block in the registers: u < 4: operand A, component u; else operand B, component u - 4
    auto piece = [&](char *base, int u) {
        const int c = u & 3;
        const uint4 (&r)[8] = u < 4 ? ua : ub;
        char *q = base + (u < 4 ? 0 : 2 * PLANE) + ((w0 ^ (c << 4)) + 4096 * c);
        const unsigned sel = (c & 1) ? 0x07060302u : 0x05040100u;
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            uint4 o;
            o.x = __builtin_amdgcn_perm(word_of(r[1], c, part), word_of(r[0], c, part), sel);
            o.y = __builtin_amdgcn_perm(word_of(r[3], c, part), word_of(r[2], c, part), sel);
            o.z = __builtin_amdgcn_perm(word_of(r[5], c, part), word_of(r[4], c, part), sel);
            o.w = __builtin_amdgcn_perm(word_of(r[7], c, part), word_of(r[6], c, part), sel);
            *(uint4 *)(q + part * PLANE) = o;
        }
        const half2_ one = (c & 1) ? half2_{(_Float16)0.f, (_Float16)1.f} : half2_{(_Float16)1.f, (_Float16)0.f};
        if (u < 4) {                                      // bias sums: one v_dot2_f32_f16 per stored half
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bs[c] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_, word_of(r[i], c, 0)), one, bs[c], false);
                bs[c] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_, word_of(r[i], c, 1)), one, bs[c], false);
            }
        } else if (HAS_W) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_, word_of(r[i], c, 0)), one, 0.f, false);
                v = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2_, word_of(r[i], c, 1)), one, v, false);
                ws[c] = fmaf(wv[i], v, ws[c]);
            }
            if (c == 3) {
#pragma unroll
                for (int i = 0; i < 8; ++i) wsum += wv[i];
            }
        }
    };
    auto read_a = [&](const char *base, int ks, half8_ (&af)[2][2]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const char *q = base + ((e0 ^ (((ks ^ i) & 1) << 5)) + 512 * (2 * wave + i));
            af[i][0] = *(const half8_ *)q;
            af[i][1] = *(const half8_ *)(q + PLANE);
        }
    };
    auto read_b = [&](const char *base, int ks, int j, half8_ (&bq)[2]) {
        const char *q = base + 2 * PLANE + ((e0 ^ (((ks ^ j) & 1) << 5)) + 512 * j);
        bq[0] = *(const half8_ *)q;
        bq[1] = *(const half8_ *)(q + PLANE);
    };
    auto mfma6 = [&](const half8_ (&af)[2][2], const half8_ (&bq)[2], int j) {   // per accumulator: lo*hi, hi*lo, hi*hi
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][1], bq[0], acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][1], bq[0], acc[1][j], 0, 0, 0);
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][0], bq[1], acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][0], bq[1], acc[1][j], 0, 0, 0);
        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0][0], bq[0], acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[1][0], bq[0], acc[1][j], 0, 0, 0);
    };
    // the MFMAs of the block in `base`; STAGE: stage the block in the registers into `nbase`, reloading them for block `nb`
    auto block = [&](const char *base, auto stage_tag, char *nbase, int nb) {
        constexpr bool STAGE = decltype(stage_tag)::value;
        half8_ af[2][2][2], bq[4][2];
        read_a(base, 0, af[0]);
        read_b(base, 0, 0, bq[0]);
        read_b(base, 0, 1, bq[1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t0 = 2 * u, t1 = 2 * u + 1;
            int nread = 0;
            if (u < 7) {
                read_b(base, (t0 + 2) >> 3, (t0 + 2) & 7, bq[(t0 + 2) & 3]);
                read_b(base, (t1 + 2) >> 3, (t1 + 2) & 7, bq[(t1 + 2) & 3]);
                nread += 4;
            }
            if (u == 2) { read_a(base, 1, af[1]); nread += 4; }
            mfma6(af[t0 >> 3], bq[t0 & 3], t0 & 7);
            mfma6(af[t1 >> 3], bq[t1 & 3], t1 & 7);
            if (STAGE) {
                piece(nbase, u);
                if (u == 4) load_a(nb);
                if (u == 7) load_b(nb);
            }
            // MFMA, then what may issue in its shadow: a fragment read, a reload, staging VALU, a staging store
#pragma unroll
            for (int m = 0; m < 12; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                if (m < nread) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (STAGE) {
                    if (u == 4 && m < 4) __builtin_amdgcn_sched_group_barrier(0x20, 2, 0);     // A reload first: its registers are free
                    __builtin_amdgcn_sched_group_barrier(0x2, 3, 0);
                    if (m >= 10) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    if (u == 7 && m >= 10) __builtin_amdgcn_sched_group_barrier(0x20, HAS_W ? 8 : 4, 0);   // B reload behind its last piece
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    int cur = 0;
    if (nfull > 0) {
        load_a(0);
        load_b(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) piece(lb, u);
        if (nfull > 1) { load_a(1); load_b(1); }
    }
    __syncthreads();
    for (int blk = 0; blk + 1 < nfull; ++blk) {
        const int nb = blk + 2 < nfull ? blk + 2 : nfull - 1;
        block(lb + cur * BUF, std::true_type(), lb + (cur ^ 1) * BUF, nb);
        __syncthreads();
        cur ^= 1;
    }
    if (tail) load_tail(nfull);
    if (nfull > 0) {
        block(lb + cur * BUF, std::false_type(), nullptr, 0);
        cur ^= 1;
    }
    if (tail) {
#pragma unroll
        for (int u = 0; u < 8; ++u) piece(lb + cur * BUF, u);
        __syncthreads();
        block(lb + cur * BUF, std::false_type(), nullptr, 0);
    }
    __syncthreads();                                      // fragment reads done before the LDS is reused below

    float *part = a.partial + d.part_off + (size_t)blockIdx.x * d.part_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ot = wave * 2 + i;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = 32 * ot + (r & 3) + 8 * (r >> 2) + 4 * h;
                part[(size_t)o * Kp + 32 * j + l31] = acc[i][j][r];
            }
    }
    float *red = (float *)lb;
#pragma unroll
    for (int c = 0; c < 4; ++c) red[wave * 256 + 4 * lane + c] = bs[c];
    __syncthreads();
    part[(size_t)Mp * Kp + tid] = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
    if (HAS_W) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 4; ++c) red[wave * 256 + 4 * lane + c] = ws[c];
        if (lane == 0) red[1024 + wave] = wsum;
        __syncthreads();
        part[(size_t)Mp * Kp + Mp + tid] = (red[tid] + red[256 + tid]) + (red[512 + tid] + red[768 + tid]);
        if (tid == 0) part[(size_t)Mp * Kp + Mp + 256] = (red[1024] + red[1025]) + (red[1026] + red[1027]);
    }
}
// build switch VN_WGRAD_PIPE (default 1, vipnerf_knobs.h)
__global__ __launch_bounds__(256) void k_wgrad_split16_256(WgArgs a) {
    if ((int)blockIdx.x >= a.d[blockIdx.y].n_chunks) return;
#if VN_WGRAD_PIPE
    if (a.d[blockIdx.y].wcol) wgrad_split16_256_pipe<true>(a);
    else wgrad_split16_256_pipe<false>(a);
#else
    if (a.d[blockIdx.y].wcol) wgrad_split16_256_body<true>(a);
    else wgrad_split16_256_body<false>(a);
#endif
}

template <int PARTS, bool BF = false>
__global__ __launch_bounds__(256) void k_wgrad_h16_256(WgArgs a) {
    if ((int)blockIdx.x >= a.d[blockIdx.y].n_chunks) return;
    if (a.d[blockIdx.y].wcol) wgrad_h16_256_body<true, PARTS, BF>(a);
    else wgrad_h16_256_body<false, PARTS, BF>(a);
}

// Ordered sum over chunks.  A workgroup of 256 threads handles 64 consecutive output elements: thread (e, q) sums
// every 4th chunk starting at q (4 independent load streams per element), the four partial sums are folded in a
// fixed order through LDS -> deterministic, and 4x the memory-level parallelism of one thread per element.
__global__ __launch_bounds__(256) void k_wgrad_reduce(WgReduceArgs a) {
    __shared__ float sh[4][64];
    const WgGroup &g = a.g[blockIdx.y];
    const int n_w = g.m_valid * g.k_valid;
    const int n_all = n_w + (g.dbias ? g.m_valid : 0);
    const int el = threadIdx.x & 63, q = threadIdx.x >> 6;
    const float unscale = a.gmax ? 1.f / grad_scale_from_max(*a.gmax) : 1.f;
    for (int e0 = blockIdx.x * 64; e0 < n_all; e0 += gridDim.x * 64) {
        const int e = e0 + el;
        float s = 0.f;
        size_t off = 0;
        float *dst = nullptr;
        if (e < n_all) {
            if (e < n_w) {
                const int m = e / g.k_valid, k = e % g.k_valid;
                off = (size_t)m * g.Kp + k;
                const int kn = wg_colperm(g.colperm, k);         // stored column / row order (T16 storage) -> feature
                dst = kn >= 0 ? g.dW + (size_t)wg_colperm(g.rowperm, m) * g.ldw + g.col_off + kn : nullptr;
            } else {
                const int m = e - n_w;
                off = g.bias_off + m;
                dst = g.dbias + wg_colperm(g.rowperm, m);
            }
            for (int dd = 0; dd < (dst ? g.n_desc : 0); ++dd) {
                const float *pp = a.partial + g.part_off + (size_t)dd * g.desc_stride + off;
                // eight loads in flight per thread (the sum is latency-bound otherwise); fixed order -> deterministic
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                int c = q;
                for (; c + 28 < g.n_chunks; c += 32) {
                    const float v0 = pp[(size_t)c * g.part_stride], v1 = pp[(size_t)(c + 4) * g.part_stride];
                    const float v2 = pp[(size_t)(c + 8) * g.part_stride], v3 = pp[(size_t)(c + 12) * g.part_stride];
                    const float v4 = pp[(size_t)(c + 16) * g.part_stride], v5 = pp[(size_t)(c + 20) * g.part_stride];
                    const float v6 = pp[(size_t)(c + 24) * g.part_stride], v7 = pp[(size_t)(c + 28) * g.part_stride];
                    s0 += v0; s1 += v1; s2 += v2; s3 += v3;
                    s0 += v4; s1 += v5; s2 += v6; s3 += v7;
                }
                for (; c + 12 < g.n_chunks; c += 16) {
                    s0 += pp[(size_t)c * g.part_stride];
                    s1 += pp[(size_t)(c + 4) * g.part_stride];
                    s2 += pp[(size_t)(c + 8) * g.part_stride];
                    s3 += pp[(size_t)(c + 12) * g.part_stride];
                }
                for (; c < g.n_chunks; c += 4) s0 += pp[(size_t)c * g.part_stride];
                s += (s0 + s1) + (s2 + s3);
            }
        }
        sh[q][el] = s;
        __syncthreads();
        if (q == 0 && e < n_all && dst) *dst = ((sh[0][el] + sh[1][el]) + (sh[2][el] + sh[3][el])) * unscale;
        __syncthreads();
    }
}

template <int MTW, int KTW, int WAVES_M>
static int launch_class(const WgArgs &args, int n_desc, int n_chunks, hipStream_t st) {
    if (n_desc == 0) return VIPNERF_OK;
    constexpr int MT = MTW * WAVES_M, KT = KTW * (4 / WAVES_M);
    const size_t lds = (size_t)2 * 32 * (32 * MT + 32 * KT) * sizeof(float);
    VN_HIP(hipFuncSetAttribute((const void *)k_wgrad<MTW, KTW, WAVES_M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_wgrad<MTW, KTW, WAVES_M>), dim3(n_chunks, n_desc), dim3(256), lds, st, args);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

int launch_wgrad(size_t P, int V, const float *acts, const ActLayout &al, float *bwd, const BwdLayout &bl,
                 const vipnerf_mlp_grads *G, int precision, hipStream_t st, const unsigned *gmax) {
    if (P == 0) return VIPNERF_OK;
    if (stores_t16(precision)) return launch_wgrad16(P, V, acts, al, bwd, bl, G, precision, st, gmax);
// build switch VN_WGRAD_ONE_ROUND (default 1, vipnerf_knobs.h): point chunks: one round of workgroups per launch where the level is large enough (like vipnerf_wgrad16.hip)
    int n_chunks = wgrad_chunks(P), n_pe = wgrad_chunks_split(P, WGRAD_SPLIT_PE), n_thin = wgrad_chunks_split(P, WGRAD_SPLIT_THIN),
        n_single = wgrad_chunks_split(P, WGRAD_SINGLE_SPLIT);
    int chunk_pts = wgrad_chunk_pts(P), chunk_pe = chunk_pts / WGRAD_SPLIT_PE, chunk_thin = chunk_pts / WGRAD_SPLIT_THIN,
        chunk_single = chunk_pts / WGRAD_SINGLE_SPLIT;
    int n_sigma = n_single, chunk_sigma = chunk_single;
    if (VN_WGRAD_ONE_ROUND) {
        // (workgroups a CU holds of the class) x 256 CUs over the launch's GEMMs, never more chunks than the plan the partial buffer was sized
        // for: every chunk costs a partial product written and read back by the reduction (4096 rays, fine level: 768 -> 256 chunks of the
        // 256 x 64 pair and of the 128 x 256 GEMM, 1536 -> 512 of the per-direction ones)
        auto plan = [&](int &n, int &pts, int slots, int n_desc) {
            const int target = slots / n_desc;
            if (target >= n) return;
            pts = (int)(((P + target - 1) / target + 31) / 32 * 32);
            n = (int)((P + pts - 1) / pts);
        };
        // (the 256 x 256 class keeps its chunks of <= 8192 points, three rounds at the fine level: in one round the exact-fp32 kernel measured
        // the same 8.22 ms per step and the HBM-bound split kernel of fp16x3 3.31 -> 3.61 -- one synchronised wave of workgroups streams worse)
// build switch VN_WGRAD_ROUNDS (default 1, vipnerf_knobs.h)
        plan(n_pe, chunk_pe, 512 * VN_WGRAD_ROUNDS, 2);
        plan(n_sigma, chunk_sigma, 512 * VN_WGRAD_ROUNDS, 1);        // the sigma head's launch: 72 KiB of LDS, two workgroups per CU
        plan(n_single, chunk_single, 256 * VN_WGRAD_ROUNDS, 1);
        plan(n_thin, chunk_thin, 1024 * VN_WGRAD_ROUNDS, 1 + V);
    }
    float *partial = bwd + bl.partial;
    // storage of the 256x256 class's operands: 0 = fp32, 1 = fp16 high parts (FP16X3H), 2 = fp16 hi + lo planes (FP16X3)
    const int halves = stores_high16(precision) ? 1 : (precision == VIPNERF_PREC_FP16X3 && VN_F16_PRESPLIT ? 2 : 0);

    WgArgs c88, c82, c48, c41, c18, c14;       // classes by (M tiles, K tiles)
    WgReduceArgs red;
    int n88 = 0, n82 = 0, n48 = 0, n41 = 0, n18 = 0, n14 = 0, ng = 0;
    size_t off = 0;
    auto init = [&](WgArgs &w, int cp) { w.P = (int64_t)P; w.chunk_pts = cp; w.partial = partial; };
    init(c88, chunk_pts); init(c82, chunk_pe); init(c48, chunk_single); init(c41, chunk_thin); init(c18, chunk_sigma); init(c14, chunk_thin);
    red.partial = partial;
    red.gmax = gmax;

    // adds one GEMM; returns its partial offset
    auto add = [&](WgArgs &w, int &n, int Mp, int Kp, const float *A, int lda, int m_load, const float *B, int ldb, int k_load) {
        WgDesc &d = w.d[n++];
        d.A = A; d.lda = lda; d.m_load = m_load; d.B = B; d.ldb = ldb; d.k_load = k_load;
        d.wcol = nullptr; d.wcol_stride = 0; d.a_split16 = 0;
        d.part_off = off; d.part_stride = (size_t)Mp * Kp + Mp;
        d.n_chunks = (&w == &c88) ? n_chunks : (&w == &c48 ? n_single : (&w == &c18 ? n_sigma : (&w == &c82 ? n_pe : n_thin)));
        const size_t o = off;
        off += (size_t)d.n_chunks * d.part_stride;
        return o;
    };
    auto group = [&](int chunks, size_t part_off, int n_desc, int Mp, int Kp, int m_valid, int k_valid, float *dW, int ldw, int col_off, float *dbias) {
        WgGroup &g = red.g[ng++];
        g.part_off = part_off; g.part_stride = (size_t)Mp * Kp + Mp; g.n_desc = n_desc;
        g.n_chunks = chunks;
        g.desc_stride = (size_t)g.n_chunks * g.part_stride;
        g.Mp = Mp; g.Kp = Kp; g.m_valid = m_valid; g.k_valid = k_valid; g.dW = dW; g.ldw = ldw; g.col_off = col_off; g.dbias = dbias;
        g.bias_off = (size_t)Mp * Kp; g.colperm = 0; g.rowperm = 0;
    };
    const float *pex = acts + al.pex;
    // the sigma head rides in the feature layer's 256 x 256 GEMM (weighted column sums of its B operand h_8 while it is staged): no second
    // pass over h_8 (1 KiB per point) and no launch of its own -- the split-precision kernels always, exact fp32 with k_wgrad256_w8
    const bool fuse_sigma = precision != VIPNERF_PREC_FP32 || (VN_WGRAD_W8 && !VN_WGRAD_DMA && VN_WGRAD_SIGMA_FUSED);
    {   // feature_linear -- FIRST in the launch (blockIdx.y = 0): its workgroups carry the sigma head (a few per cent longer) and should not be the last round's
        const size_t o = add(c88, n88, 256, 256, bwd + bl.dyf, W, W, acts + al.h[D - 1], W, W);
        group(n_chunks, o, 1, 256, 256, W, W, G->g[P_FW], W, 0, G->g[P_FB]);
        if (fuse_sigma) {
            // sigma head: dW = sum_p dsigma_raw[p] h_8[p][:], db = sum_p dsigma_raw[p] -- weighted column sums of this
            // GEMM's B operand, taken while it is staged (no second pass over h_8, no extra launch)
            WgDesc &d = c88.d[n88 - 1];
            d.wcol = bwd + bl.dq[0] + 4; d.wcol_stride = 8;
            d.part_stride += WCOL_EXTRA;
            off += (size_t)d.n_chunks * WCOL_EXTRA;
            group(n_chunks, o + (size_t)256 * 256 + 256, 1, 1, 256, 1, W, G->g[P_SW], W, 0, G->g[P_SB]);
            red.g[ng - 1].part_stride = d.part_stride;
            red.g[ng - 1].desc_stride = (size_t)n_chunks * d.part_stride;
            red.g[ng - 2].part_stride = d.part_stride;
            red.g[ng - 2].desc_stride = (size_t)n_chunks * d.part_stride;
        }
    }
    // trunk
    for (int i = 0; i < D; ++i) {
        const float *dy = bwd + bl.dy[i];
        float *dW = G->g[2 * i], *db = G->g[2 * i + 1];
        if (i == 0) {
            const size_t o = add(c82, n82, 256, 64, dy, W, W, pex, DPE_PAD, DPE_PAD);
            group(n_pe, o, 1, 256, 64, W, DPE, dW, DPE, 0, db);
        } else if (i == SKIP_LAYER) {
            // fp16 high parts only (FP16X3H): the fp32 copy the data-gradient kernel leaves; pre-split: dY_5 itself, decoded on the way
            const size_t o1 = add(c82, n82, 256, 64, halves == 1 ? bwd + bl.dy5f : dy, W, W, pex, DPE_PAD, DPE_PAD);
            c82.d[n82 - 1].a_split16 = halves == 2;
            group(n_pe, o1, 1, 256, 64, W, DPE, dW, W + DPE, 0, nullptr);
            const size_t o2 = add(c88, n88, 256, 256, dy, W, W, acts + al.h[i - 1], W, W);
            group(n_chunks, o2, 1, 256, 256, W, W, dW, W + DPE, DPE, db);
        } else {
            const size_t o = add(c88, n88, 256, 256, dy, W, W, acts + al.h[i - 1], W, W);
            group(n_chunks, o, 1, 256, 256, W, W, dW, W, 0, db);
        }
    }
    if (!fuse_sigma) {   // sigma head: A = column 4 of DQ[0]
        const size_t o = add(c18, n18, 32, 256, bwd + bl.dq[0] + 4, 8, 4, acts + al.h[D - 1], W, W);
        group(n_sigma, o, 1, 32, 256, 1, W, G->g[P_SW], W, 0, G->g[P_SB]);
    }
    // the view layer: exact fp32 in ONE launch over dYv_0..V (k_wgrad_view; no dYvsum); the split-precision modes as two GEMM classes
    const bool view_fused = precision == VIPNERF_PREC_FP32 && VN_WGRAD_VIEW_FUSED;
    const bool heads_fused = view_fused && VN_WGRAD_HEADS_FUSED && V <= 1 && P % 16 == 0;       // (its DMA moves whole 16-point blocks)
    WgViewArgs va;
    va.part_oh = nullptr; va.stride_oh = 0;
    for (int k = 0; k <= VIPNERF_MAX_SEC; ++k) { va.g[k] = nullptr; va.dq[k] = nullptr; }
    if (view_fused) {
        for (int k = 0; k <= VIPNERF_MAX_SEC; ++k) { va.dyv[k] = bwd + bl.dyv[k <= V ? k : 0]; va.ped[k] = acts + al.ped[k <= V ? k : 0]; }
        va.feat = acts + al.feat; va.P = (int64_t)P; va.chunk_pts = chunk_single; va.n_chunks = n_single;
        va.stride_vf = (size_t)128 * 256 + 128; va.stride_vd = (size_t)128 * 32 + 128;
        va.part_vf = partial + off;
        group(n_single, off, 1, 128, 256, WV, W, G->g[P_VW], W + DVE, 0, G->g[P_VB]);
        off += (size_t)n_single * va.stride_vf;
        va.part_vd = partial + off;
        group(n_single, off, 1, 128, 32, WV, DVE, G->g[P_VW], W + DVE, W, nullptr);
        off += (size_t)n_single * va.stride_vd;
        if (heads_fused) {     // the output head in the same launch (V <= 1): A = DQ[k][:, 0:4], B = view hidden of direction k, summed over k in the accumulator
            for (int k = 0; k <= VIPNERF_MAX_SEC; ++k) { va.g[k] = acts + al.g[k <= V ? k : 0]; va.dq[k] = bwd + bl.dq[k <= V ? k : 0]; }
            va.stride_oh = (size_t)32 * 128 + 32;
            va.part_oh = partial + off;
            group(n_single, off, 1, 32, 128, 4, WV, G->g[P_OW], WV, 0, G->g[P_OB]);
            off += (size_t)n_single * va.stride_oh;
        }
    } else {
    {   // view layer, feature columns: A = sum over directions
        const size_t o = add(c48, n48, 128, 256, bwd + bl.dyvsum, WV, WV, acts + al.feat, W, W);
        group(n_single, o, 1, 128, 256, WV, W, G->g[P_VW], W + DVE, 0, G->g[P_VB]);
    }
    {   // view layer, direction columns: one GEMM per direction, summed in order
        size_t first = 0;
        for (int k = 0; k <= V; ++k) {
            const size_t o = add(c41, n41, 128, 32, bwd + bl.dyv[k], WV, WV, acts + al.ped[k], DVE_PAD, DVE_PAD);
            if (k == 0) first = o;
        }
        group(n_thin, first, 1 + V, 128, 32, WV, DVE, G->g[P_VW], W + DVE, W, nullptr);
    }
    }
    if (!heads_fused) {   // output head: A = DQ[k][:, 0:4], B = view hidden of direction k
        size_t first = 0;
        for (int k = 0; k <= V; ++k) {
            const size_t o = add(c14, n14, 32, 128, bwd + bl.dq[k], 8, 4, acts + al.g[k], WV, WV);
            if (k == 0) first = o;
        }
        group(n_thin, first, 1 + V, 32, 128, 4, WV, G->g[P_OW], WV, 0, G->g[P_OB]);
    }
    if (off > wgrad_partial_total(P, V)) { set_error("wgrad: partial buffer plan mismatch"); return VIPNERF_E_ARG; }

    int rc;
    {
        ProfScope ps("wgrad_256x256", st);
        if (precision == VIPNERF_PREC_FP32) {
#if VN_WGRAD_W8
            const size_t ldsb = (size_t)2 * 32 * 512 * sizeof(float);
            VN_HIP(hipFuncSetAttribute((const void *)k_wgrad256_w8<VN_WGRAD_W8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
            hipLaunchKernelGGL(k_wgrad256_w8<VN_WGRAD_W8>, dim3(n_chunks, n88), dim3(256 * VN_WGRAD_W8), ldsb, st, c88);
            VN_HIP(hipGetLastError());
#else
            if ((rc = launch_class<2, 8, 4>(c88, n88, n_chunks, st))) return rc;
#endif
        } else if (halves == 1) {                  // operands stored as fp16 high parts: single-MFMA kernel, half the bytes
            const size_t ldsb = (size_t)2 * 2 * 256 * 64;
            if (precision == VIPNERF_PREC_BF16) {
                VN_HIP(hipFuncSetAttribute((const void *)k_wgrad_h16_256<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
                hipLaunchKernelGGL((k_wgrad_h16_256<1, true>), dim3(n_chunks, n88), dim3(256), ldsb, st, c88);
            } else {
                VN_HIP(hipFuncSetAttribute((const void *)k_wgrad_h16_256<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
                hipLaunchKernelGGL(k_wgrad_h16_256<1>, dim3(n_chunks, n88), dim3(256), ldsb, st, c88);
            }
            VN_HIP(hipGetLastError());
        } else if (halves == 2) {                  // operands stored pre-split (hi and lo fp16 planes): 3 fp16 cross terms
            const size_t ldsb = (size_t)2 * 4 * 256 * 64;
            VN_HIP(hipFuncSetAttribute((const void *)k_wgrad_split16_256, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
            hipLaunchKernelGGL(k_wgrad_split16_256, dim3(n_chunks, n88), dim3(256), ldsb, st, c88);
            VN_HIP(hipGetLastError());
        } else {
            set_error("wgrad: precision %d has no 256 x 256 weight-gradient kernel (the split-bf16 arithmetics were retired with ABI 5)", precision);
            return VIPNERF_E_UNSUPPORTED;
        }
    }
    ProfScope ps("wgrad_small", st);
    if (precision == VIPNERF_PREC_FP32) {
        if (view_fused) {
            rc = V == 0 ? launch_view<1>(va, heads_fused, st) : (V == 1 ? launch_view<2>(va, heads_fused, st) : (V == 2 ? launch_view<3>(va, false, st) : launch_view<4>(va, false, st)));
            if (rc) return rc;
        } else if ((rc = launch_class<1, 8, 4>(c48, n48, n_single, st))) return rc;
        if ((rc = launch_class<2, 2, 4>(c82, n82, n_pe, st))) return rc;
    } else {
        if ((rc = launch_bf16x3<4, 8, 2>(c48, n48, n_single, st))) return rc;
        if ((rc = launch_bf16x3<8, 2, 1>(c82, n82, n_pe, st))) return rc;
    }
    if (n41 && (rc = launch_class<1, 1, 4>(c41, n41, n_thin, st))) return rc;
    if (n18 && (rc = launch_class<1, 2, 1>(c18, n18, n_sigma, st))) return rc;
    if (n14 && (rc = launch_class<1, 1, 1>(c14, n14, n_thin, st))) return rc;
    return launch_wgrad_reduce(red, ng, st);
}

#if defined(VN_EXP) && VN_EXP == 50
extern "C" int vipnerf_exp_timeline_wg(unsigned long long *out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_timeline), sizeof(unsigned long long) * (n < 2048 ? n : 2048));
}
#endif

int launch_wgrad_reduce(const WgReduceArgs &red, int ng, hipStream_t st) {
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(512, ng), dim3(256), 0, st, red);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

}  // namespace vn
