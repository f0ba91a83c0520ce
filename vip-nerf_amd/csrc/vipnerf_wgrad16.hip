// Weight gradients of one level in the single-MFMA 16-bit modes (VIPNERF_PREC_FP16 / BF16) with T16 operand storage
// (vipnerf_bf16n.h: store_t16): dW[M][N] = sum_p A[p][M] * B[p][N], db[M] = sum_p A[p][M]  (autograd of reference
// src/models/VipNeRF01.py:537-596 w.r.t. the parameters).
//
// Every operand is a 16-bit array in the tile-blocked layout the forward / data-gradient kernels wrote:
//     [P / 16 groups][width / 16 tiles][16 points][16 features],   one (group, tile) = a row-major 16 x 16 matrix = 512 B,
// so the bytes of a 32-point block of an operand are contiguous in HBM.  No conversion, no register transposition anywhere:
//   * HBM -> LDS by DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, lane-linear: the LDS image IS the HBM image), a ring of
//     NB blocks, counted vmcnt + one raw workgroup barrier per block; the 256 x 256 launch (HY) moves half of every block through
//     registers instead (global_load_dwordx4 + ds_write_b128 of the same image), two blocks per barrier -- the DMA path alone tops out
//     below what HBM delivers;
//   * LDS -> MFMA fragments by the hardware transpose read: ds_read_b64_tr_b16 of lane l at (tile base + 8 l) returns, for feature
//     l & 15, the points 4 (l >> 4) .. + 3 of the tile -- one read per 16-point group, two per 32-deep v_mfma_f32_16x16x32 fragment.
//     The same read serves A and B (contraction index = point: element e < 4 of lane group q is point 4 q + e of the block's first
//     group, e >= 4 point 4 q + e - 4 of its second), and every read instruction covers 512 contiguous bytes: no bank conflicts;
//   * bias gradients = column sums of A, taken from the A fragments with v_dot2 (fp32 accumulate) under the MFMAs.
// A workgroup owns one (GEMM, point chunk) pair, keeps its share of the product in accumulators and writes a partial; the ordered
// reduction of vipnerf_wgrad.hip (k_wgrad_reduce) sums the chunks into the nn.Linear-layout gradients (deterministic, no atomics).
// These GEMMs have 0.5 ... 64 MFMAs per KiB of operand: all of them are bound by how many bytes a CU keeps in flight, which is what
// the DMA ring is for (3-4 blocks per workgroup, several workgroups per CU for the thin ones).
#include <type_traits>
#include "vipnerf_wgrad.h"
#include "vipnerf_prof.h"

namespace vn {

// lane l: for feature (l & 15), the 4 points 4 (l >> 4) .. + 3 of the 16 x 16 tile at `tile` (LDS), 16-bit each
__device__ __forceinline__ uint2 tr_read16(const char *tile, int lane) {
    typedef short s4 __attribute__((ext_vector_type(4)));
    const s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4 *)(tile + lane * 8));
    return __builtin_bit_cast(uint2, v);
}
template <typename FR>
__device__ __forceinline__ FR frag16(const char *tile_g0, const char *tile_g1, int lane) {
    const uint2 a = tr_read16(tile_g0, lane), b = tr_read16(tile_g1, lane);
    const uint4 w = make_uint4(a.x, a.y, b.x, b.y);
    return __builtin_bit_cast(FR, w);
}
__device__ __forceinline__ float dot_ones(const half8 &v, float s) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const uint4 w = __builtin_bit_cast(uint4, v);
    const h2 one = {(_Float16)1.f, (_Float16)1.f};
    s = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w.x), one, s, false);
    s = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w.y), one, s, false);
    s = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w.z), one, s, false);
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, w.w), one, s, false);
}
__device__ __forceinline__ float dot_ones(const bf16x8 &v, float s) {
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const uint4 w = __builtin_bit_cast(uint4, v);
    const b2 one = {(__bf16)1.f, (__bf16)1.f};
    s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, w.x), one, s, false);
    s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, w.y), one, s, false);
    s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, w.z), one, s, false);
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(b2, w.w), one, s, false);
}
// build switch VN_WG16_BIG_WM (default 2, vipnerf_knobs.h): wave grid and ring depth of the 256 x 256 kernel
// build switch VN_WG16_BIG_WN (default 4, vipnerf_knobs.h)
// build switch VN_WG16_BIG_NB (default 4, vipnerf_knobs.h)
// build switch VN_WG16_HYBRID (default 1, vipnerf_knobs.h): the 256 x 256 kernel streams half of each block by DMA, half through registers (k_wg16's HY)
// build switch VN_WG16_SIGMA_FUSED (default 1, vipnerf_knobs.h): the sigma head rides in the feature layer's GEMM (XA below); 0: its own 16 x 256 launch
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// M = 16 MT, N = 16 NT; WM x WN waves, wave (wm, wn) owns MT / WM x NT / WN tiles; NB 32-point blocks resident in LDS.
// XA: a GEMM descriptor may carry ONE extra 16-row A tile (WgDesc::wcol = a 16-wide T16 array) against the same B -- the sigma head
// (row 4 of direction 0's head-seed tile against h_8) rides in the feature layer's GEMM instead of reading h_8 a second time: one more
// 1 KiB piece per block, one more MFMA per column tile for the waves of the first row block, its partial [16][N] + 16 sums behind the rest.
// HY (hybrid stream, NB = 4): the DMA path alone tops out at 6.0-6.2 TB/s on this part, register loads at 7.0-7.3, half and half at 7.0
// (tools/hbm_stream_probe.hip) -- so half of every block's pieces arrive by DMA as before and half through registers (global_load_dwordx4
// one trip = two blocks ahead, ds_write_b128 of the same lane-linear image); see the trip loop in the kernel.
template <bool BF, int MT, int NT, int WM, int WN, int NB, bool XA = false, bool HY = false>
__global__ __launch_bounds__(64 * WM * WN) void k_wg16(WgArgs a) {
    typedef typename FragOf<!BF>::type FR;
    constexpr int NW = WM * WN, TM = MT / WM, TN = NT / WN;
    constexpr int PIECES = MT + NT;                  // 1 KiB DMA pieces per block: the block's A bytes, then its B bytes
    constexpr int BLK = (PIECES + (XA ? 1 : 0)) * 1024;
    constexpr int PW = (PIECES + NW - 1) / NW;       // pieces per wave and block: PW, or PW - 1 for the last waves
    constexpr int MINP = PIECES / NW;                // what the waits count (conservative for the waves that issue PW)
    static_assert(MT % WM == 0 && NT % WN == 0 && NB >= 2 && NB <= 4 && (NB - 1) * PW <= 63 && MINP >= 1, "shape");
    static_assert(!HY || NB == 4, "hybrid stream: four slots");
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    char *lds = (char *)lds_f;

    const WgDesc &d = a.d[blockIdx.y];
    if ((int)blockIdx.x >= d.n_chunks) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int64_t p0 = (int64_t)blockIdx.x * a.chunk_pts;
    const int64_t p1 = p0 + a.chunk_pts < a.P ? p0 + a.chunk_pts : a.P;
    const int nblk = (int)((p1 - p0) / 32);          // P and the chunk sizes are multiples of 32
    const char *gA = (const char *)d.A + (size_t)(p0 >> 4) * (MT * 512) + lane * 16;
    const char *gB = (const char *)d.B + (size_t)(p0 >> 4) * (NT * 512) + lane * 16;

    floatx4 acc[TM][TN];
    float bsum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        bsum[i] = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (floatx4)(0.f);
    }
    const bool has_x = XA && d.wcol != nullptr;      // (workgroup-uniform)
    const char *gX = has_x ? (const char *)d.wcol + (size_t)(p0 >> 4) * 512 + lane * 16 : nullptr;
    floatx4 accx[XA ? TN : 1];
    float bsumx = 0.f;
#pragma unroll
    for (int j = 0; j < (XA ? TN : 1); ++j) accx[j] = (floatx4)(0.f);

    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    auto piece_src = [&](int b, int pc) { return pc < MT ? gA + ((size_t)b * MT + pc) * 1024 : gB + ((size_t)b * NT + (pc - MT)) * 1024; };
    auto dma_piece = [&](int b, int slot, int pc) {
        glds_chunks<1>((const float *)piece_src(b, pc), __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds + slot * BLK + pc * 1024)));
    };
    auto dma_extra = [&](int b, int slot) {          // the extra tile's block (2 groups x 512 B); one more load than the waits count: still a lower bound
        if (XA && has_x && wave == NW - 1)
            glds_chunks<1>((const float *)(gX + (size_t)b * 1024), __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds + slot * BLK + PIECES * 1024)));
    };
    // the product of the block in LDS slot `slot` into the accumulators
    auto multiply = [&](int slot) {
        const char *A = lds + slot * BLK, *B = A + MT * 1024;
        FR af[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int t = wm * TM + i;
            af[i] = frag16<FR>(A + t * 512, A + (MT + t) * 512, lane);
        }
        // (wave-uniform, and told so: as a per-lane condition the compiler predicates the extra tile's MFMAs with EXEC, which MFMA ignores --
        // they then run in every wave on an unloaded fragment; harmless only as long as nothing else lives in accx)
        const bool do_x = XA && has_x && __builtin_amdgcn_readfirstlane(wm) == 0;
        FR ax = __builtin_bit_cast(FR, make_uint4(0u, 0u, 0u, 0u));
        if (do_x) ax = frag16<FR>(A + PIECES * 1024, A + PIECES * 1024 + 512, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int t = wn * TN + j;
            const FR bf = frag16<FR>(B + t * 512, B + (NT + t) * 512, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[i][j] = mfma_bf(af[i], bf, acc[i][j]);
            if (XA) { if (do_x) accx[j] = mfma_bf(ax, bf, accx[j]); }
        }
        if (wn == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i) bsum[i] = dot_ones(af[i], bsum[i]);
            if (XA) { if (do_x) bsumx = dot_ones(ax, bsumx); }
        }
    };
    auto publish = [&]() {                            // every wave's pieces are in; every wave is done with the slot about to be refilled
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    if constexpr (!HY) {
        auto issue = [&](int b, int slot) {
#pragma unroll
            for (int i = 0; i < PW; ++i) {
                const int pc = wave + i * NW;
                if (pc < PIECES) dma_piece(b, slot, pc);
            }
            dma_extra(b, slot);
        };
        int fill = 0;
#pragma unroll
        for (int b = 0; b < NB - 1; ++b)
            if (b < nblk) { issue(b, fill); fill = fill + 1 == NB ? 0 : fill + 1; }
        int cur = 0;
        for (int b = 0; b < nblk; ++b) {
            // this wave's pieces of block b have landed: at most the pieces of the y younger blocks it has issued stay outstanding
            const int y = nblk - 1 - b < NB - 2 ? nblk - 1 - b : NB - 2;
            __builtin_amdgcn_sched_barrier(0);
            if (y <= 0) wait_vm<0>();
            else if (y == 1) wait_vm<MINP>();
            else wait_vm<2 * MINP>();
            publish();
            if (b + NB - 1 < nblk) { issue(b + NB - 1, fill); fill = fill + 1 == NB ? 0 : fill + 1; }
            multiply(cur);
            cur = cur + 1 == NB ? 0 : cur + 1;
        }
    } else {
        // Hybrid stream, four slots (block k in slot k % 4), two blocks per trip.  Per block a wave moves PWH pieces by DMA (pieces
        // [0, PIECES / 2): the A operand's) and PWH through registers ([PIECES / 2, PIECES)).  Everything trip b + 2 needs -- DMA(b + 2),
        // DMA(b + 3), the register pieces of both -- is issued together at the start of trip b, right behind its barrier (the two slots are
        // the ones trip b - 2 used), so at the start of a trip the wave has outstanding exactly that trip's operations, issued one trip
        // (two block times) earlier: ONE s_waitcnt vmcnt(0) per trip is exact, the register pieces go to LDS in front of the trip's only
        // barrier, and the compiler's own bookkeeping for the register loads (which cannot see the inline-asm DMA) costs nothing.
// build switch VN_WG16_DMA_PIECES (default 2, vipnerf_knobs.h): of a wave's 4 pieces per block (256 x 256 launch): measured 1.36 ms per 4096-ray step; see docs/HISTORY.md 4.3a for 0 / 1 / 3
        constexpr int PWD = VN_WG16_DMA_PIECES < PW ? VN_WG16_DMA_PIECES : PW / 2, PWR = PW - PWD;     // pieces per wave and block by DMA / through registers
        static_assert(NB == 4 && PIECES % NW == 0 && PWR >= 1, "hybrid stream: four slots, every wave the same number of pieces");
        u4 r0[PWR], r1[PWR];
        auto dma_half = [&](int b, int slot) {
#pragma unroll
            for (int i = 0; i < PWD; ++i) dma_piece(b, slot, wave + i * NW);
            dma_extra(b, slot);
        };
        auto load_regs = [&](int b, u4 (&r)[PWR]) {
#pragma unroll
            for (int i = 0; i < PWR; ++i) r[i] = __builtin_nontemporal_load((const u4 *)piece_src(b, PWD * NW + wave + i * NW));
        };
        auto write_regs = [&](int slot, const u4 (&r)[PWR]) {
#pragma unroll
            for (int i = 0; i < PWR; ++i) *(u4 *)(lds + slot * BLK + (PWD * NW + wave + i * NW) * 1024 + lane * 16) = r[i];
        };
        int b = 0, sb = 0;                           // sb: slot of block b (even blocks: 0 or 2)
        bool in_flight = false;                      // blocks b, b + 1: DMA issued, register pieces in r0 / r1
        if (nblk >= 4) {
            dma_half(0, 0); dma_half(1, 1);
            load_regs(0, r0); load_regs(1, r1);
            in_flight = true;
            for (; b + 3 < nblk; b += 2) {
                __builtin_amdgcn_sched_barrier(0);
                wait_vm<0>();
                write_regs(sb, r0);
                write_regs(sb + 1, r1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                publish();                           // blocks b, b + 1 are in; every wave is done with blocks b - 2, b - 1
                dma_half(b + 2, sb ^ 2); dma_half(b + 3, (sb ^ 2) + 1);
                load_regs(b + 2, r0); load_regs(b + 3, r1);
                multiply(sb);
                multiply(sb + 1);
                sb ^= 2;
            }
        }
        // the last 2 or 3 blocks of a long chunk (b, b + 1 in flight), or all 1 .. 3 blocks of a short one: everything in, one barrier
        const int left = nblk - b;
        __builtin_amdgcn_sched_barrier(0);
        if (in_flight) {
            write_regs(sb, r0);
            write_regs(sb + 1, r1);
        }
        for (int k = in_flight ? 2 : 0; k < left; ++k) {
            const int sk = (sb + k) & 3;
            dma_half(b + k, sk);
            load_regs(b + k, r0);
            write_regs(sk, r0);
        }
        wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        publish();
        for (int k = 0; k < left; ++k) multiply((sb + k) & 3);
    }

    // partial product of this chunk: [16 MT][16 NT] row-major, then the column sums of A [16 MT]
    constexpr int Mp = 16 * MT, Np = 16 * NT;
    float *part = a.partial + d.part_off + (size_t)blockIdx.x * d.part_stride;
    const int jn = lane & 15, qr = lane >> 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int tm = wm * TM + i;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int tn = wn * TN + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(size_t)(16 * tm + 4 * qr + r) * Np + 16 * tn + jn] = acc[i][j][r];
        }
        if (wn == 0) {
            float s = bsum[i];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (lane < 16) part[(size_t)Mp * Np + 16 * tm + lane] = s;
        }
    }
    if (XA) {
        if (has_x && wm == 0) {                      // the extra tile's partial: [16][Np] and its 16 column sums, behind the bias sums
            float *px = part + (size_t)Mp * Np + Mp;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int tn = wn * TN + j;
#pragma unroll
                for (int r = 0; r < 4; ++r) px[(size_t)(4 * qr + r) * Np + 16 * tn + jn] = accx[j][r];
            }
            if (wn == 0) {
                float s = bsumx;
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                if (lane < 16) px[(size_t)16 * Np + lane] = s;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The view layer's weight gradient in ONE launch (the T16 counterpart of vipnerf_wgrad.hip's k_wgrad_view):
//     dW_view[:, 0:256]   = sum_k dYv_k^T feature       (128 x 256; the sum over directions is taken in the accumulator: 1 + V MFMAs per cell)
//     dW_view[:, 256:283] = sum_k dYv_k^T gamma(dir_k)  (128 x 32, likewise)
//     db_view             = sum_k column sums of dYv_k
// A 32-point block is NV x 8 KiB of dYv_0..V, 16 KiB of the feature and NV x 2 KiB of gamma(dir_0..V): every byte read ONCE -- the two
// launches this replaces read dYv_k twice (once as dYvsum, which the data-gradient kernels no longer write: 256 B per point less stored).
// Same stream as k_wg16's: 1 KiB DMA pieces into a ring of NB blocks (the LDS image is the HBM image), counted vmcnt, one raw barrier
// per block, ds_read_b64_tr_b16 fragments.  Wave (wm, wn) of 2 x WN owns rows 64 wm .. + 63 and feature columns (256 / WN) wn ..; the
// waves wn < 2 also own gamma(dir) column tile wn.
// build switch VN_WG16_VIEW_FUSED (default 1, vipnerf_knobs.h): 0 = the 128 x 256 GEMM over dYvsum + one 128 x 32 GEMM per direction (two launches)
struct WgView16Args {
    const float *dyv[VIPNERF_MAX_SEC + 1], *ped[VIPNERF_MAX_SEC + 1], *feat;
    int64_t P;
    int chunk_pts, n_chunks;
    float *part_vf, *part_vd;                 // per chunk: [128][256] + 128 column sums | [128][32] (+ 128 unused)
    size_t stride_vf, stride_vd;
};
template <bool BF, int NV, int WN, int NB>
__global__ __launch_bounds__(128 * WN) void k_wg16_view(WgView16Args a) {
    typedef typename FragOf<!BF>::type FR;
    constexpr int MT = 8, NT = 16, PT = 2, WM = 2, NW = WM * WN, TM = MT / WM, TN = NT / WN;
    constexpr int NPA = NV * MT / NW, NPF = NT / NW, NPP = (NV * PT + NW - 1) / NW;     // pieces per wave and block: dYv, feature, gamma(dir)
    constexpr int PIECES = NV * MT + NT + NV * PT, BLK = PIECES * 1024;
    constexpr int PW = NPA + NPF + NPP, MINP = NPA + NPF + NV * PT / NW;               // (the waits count the waves that issue fewest)
    constexpr int OFF_F = NV * MT * 1024, OFF_P = OFF_F + NT * 1024;
    static_assert((NV * MT) % NW == 0 && NT % NW == 0 && NT % WN == 0 && WN >= PT && NB >= 2 && NB <= 4 && (NB - 1) * PW <= 63, "shape");
    extern __shared__ __attribute__((aligned(16))) float lds_f[];
    char *lds = (char *)lds_f;
    if ((int)blockIdx.x >= a.n_chunks) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int64_t p0 = (int64_t)blockIdx.x * a.chunk_pts;
    const int64_t p1 = p0 + a.chunk_pts < a.P ? p0 + a.chunk_pts : a.P;
    const int nblk = (int)((p1 - p0) / 32);          // P and the chunk sizes are multiples of 32
    const size_t g0 = (size_t)(p0 >> 4);             // first 16-point group of the chunk

    // this wave's pieces of a block: source (block 0 of the chunk, this lane's 16 bytes) and LDS offset inside a block's image
    const char *srcA[NPA], *srcF[NPF], *srcP[NPP];
    int offA[NPA], offF[NPF], offP[NPP];
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
        constexpr int PER = MT / NW > 0 ? MT / NW : 1;     // pieces of one direction per wave (NW <= MT)
        const int k = i / PER, r = (i % PER) * NW + wave;
        srcA[i] = (const char *)a.dyv[k] + g0 * (MT * 512) + r * 1024 + lane * 16;
        offA[i] = (k * MT + r) * 1024;
    }
#pragma unroll
    for (int i = 0; i < NPF; ++i) {
        const int r = i * NW + wave;
        srcF[i] = (const char *)a.feat + g0 * (NT * 512) + r * 1024 + lane * 16;
        offF[i] = OFF_F + r * 1024;
    }
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
        const int e = i * NW + wave, ec = e < NV * PT ? e : 0, k = ec / PT, r = ec % PT;
        const float *pk = a.ped[0];
#pragma unroll
        for (int kk = 1; kk < NV; ++kk) pk = k == kk ? a.ped[kk] : pk;
        srcP[i] = (const char *)pk + g0 * (PT * 512) + r * 1024 + lane * 16;
        offP[i] = OFF_P + ec * 1024;
    }
    static_assert(NW <= MT, "a wave's dYv pieces belong to whole directions");

    floatx4 acc[TM][TN], accp[TM];
    float bsum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        bsum[i] = 0.f; accp[i] = (floatx4)(0.f);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (floatx4)(0.f);
    }
    auto issue = [&](int b, int slot) {
        const unsigned base = (unsigned)(size_t)(lds + slot * BLK);
#pragma unroll
        for (int i = 0; i < NPA; ++i) glds_chunks<1>((const float *)(srcA[i] + (size_t)b * (MT * 1024)), __builtin_amdgcn_readfirstlane(base + offA[i]));
#pragma unroll
        for (int i = 0; i < NPF; ++i) glds_chunks<1>((const float *)(srcF[i] + (size_t)b * (NT * 1024)), __builtin_amdgcn_readfirstlane(base + offF[i]));
#pragma unroll
        for (int i = 0; i < NPP; ++i)
            if (i * NW + wave < NV * PT) glds_chunks<1>((const float *)(srcP[i] + (size_t)b * (PT * 1024)), __builtin_amdgcn_readfirstlane(base + offP[i]));
    };
    const bool has_p = __builtin_amdgcn_readfirstlane(wn) < PT;      // (wave-uniform, and told so: an EXEC-predicated MFMA would run regardless)
    auto multiply = [&](int slot) {
        const char *S = lds + slot * BLK, *F = S + OFF_F;
        FR af[NV][TM];
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int t = wm * TM + i;
                af[k][i] = frag16<FR>(S + k * MT * 1024 + t * 512, S + k * MT * 1024 + (MT + t) * 512, lane);
            }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int t = wn * TN + j;
            const FR bf = frag16<FR>(F + t * 512, F + (NT + t) * 512, lane);
#pragma unroll
            for (int k = 0; k < NV; ++k)
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][j] = mfma_bf(af[k][i], bf, acc[i][j]);
        }
        if (has_p) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const char *Pk = S + OFF_P + k * PT * 1024;
                const FR bp = frag16<FR>(Pk + wn * 512, Pk + (PT + wn) * 512, lane);
#pragma unroll
                for (int i = 0; i < TM; ++i) accp[i] = mfma_bf(af[k][i], bp, accp[i]);
            }
        }
        if (wn == 0) {
#pragma unroll
            for (int k = 0; k < NV; ++k)
#pragma unroll
                for (int i = 0; i < TM; ++i) bsum[i] = dot_ones(af[k][i], bsum[i]);
        }
    };

    int fill = 0;
#pragma unroll
    for (int b = 0; b < NB - 1; ++b)
        if (b < nblk) { issue(b, fill); fill = fill + 1 == NB ? 0 : fill + 1; }
    int cur = 0;
    for (int b = 0; b < nblk; ++b) {
        // this wave's pieces of block b have landed: at most the pieces of the y younger blocks it has issued stay outstanding
        const int y = nblk - 1 - b < NB - 2 ? nblk - 1 - b : NB - 2;
        __builtin_amdgcn_sched_barrier(0);
        if (y <= 0) wait_vm<0>();
        else if (y == 1) wait_vm<MINP>();
        else wait_vm<2 * MINP>();
        __builtin_amdgcn_s_barrier();                 // every wave's pieces are in; every wave is done with the slot about to be refilled
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (b + NB - 1 < nblk) { issue(b + NB - 1, fill); fill = fill + 1 == NB ? 0 : fill + 1; }
        multiply(cur);
        cur = cur + 1 == NB ? 0 : cur + 1;
    }

    float *pvf = a.part_vf + (size_t)blockIdx.x * a.stride_vf, *pvd = a.part_vd + (size_t)blockIdx.x * a.stride_vd;
    const int jn = lane & 15, qr = lane >> 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int tm = wm * TM + i;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int tn = wn * TN + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) pvf[(size_t)(16 * tm + 4 * qr + r) * 256 + 16 * tn + jn] = acc[i][j][r];
        }
        if (has_p) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pvd[(size_t)(16 * tm + 4 * qr + r) * 32 + 16 * wn + jn] = accp[i][r];
        }
        if (wn == 0) {
            float s = bsum[i];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (lane < 16) pvf[(size_t)128 * 256 + 16 * tm + lane] = s;
        }
    }
}
template <bool BF, int NV, int WN, int NB>
static int launch_wg16_view(const WgView16Args &va, hipStream_t st) {
    const size_t ldsb = (size_t)NB * (NV * 8 + 16 + NV * 2) * 1024;
    VN_HIP(hipFuncSetAttribute((const void *)k_wg16_view<BF, NV, WN, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipLaunchKernelGGL((k_wg16_view<BF, NV, WN, NB>), dim3(va.n_chunks), dim3(128 * WN), ldsb, st, va);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}
// ring depth by direction count: 26 / 36 / 46 / 56 KiB per block -> 104 / 144 / 138 / 112 KiB of LDS, one workgroup per CU
template <bool BF>
static int launch_wg16_view_v(const WgView16Args &va, int V, hipStream_t st) {
    switch (V) {
    case 0: return launch_wg16_view<BF, 1, VN_WG16_VIEW_WN, 4>(va, st);
    case 1: return launch_wg16_view<BF, 2, VN_WG16_VIEW_WN, 4>(va, st);
    case 2: return launch_wg16_view<BF, 3, VN_WG16_VIEW_WN, 3>(va, st);
    default: return launch_wg16_view<BF, 4, VN_WG16_VIEW_WN, 2>(va, st);
    }
}

template <bool BF, int MT, int NT, int WM, int WN, int NB, bool XA = false, bool HY = false>
static int launch_wg16(const WgArgs &args, int n_desc, int n_chunks, hipStream_t st) {
    if (n_desc == 0) return VIPNERF_OK;
    const size_t ldsb = (size_t)NB * (MT + NT + (XA ? 1 : 0)) * 1024;
    VN_HIP(hipFuncSetAttribute((const void *)k_wg16<BF, MT, NT, WM, WN, NB, XA, HY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipLaunchKernelGGL((k_wg16<BF, MT, NT, WM, WN, NB, XA, HY>), dim3(n_chunks, n_desc), dim3(64 * WM * WN), ldsb, st, args);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

template <bool BF>
static int launch_all(const WgArgs &big, int nbig, int n_chunks, const WgArgs &pe, int npe, int n_pe, const WgArgs &vf, int nvf, const WgArgs &sg, int nsg,
                      int n_single, const WgArgs &vd, int nvd, int n_vd, const WgArgs &oh, int noh, int n_oh, hipStream_t st) {
    int rc;
    {
        ProfScope ps("wgrad_256x256", st);
        if ((rc = launch_wg16<BF, 16, 16, VN_WG16_BIG_WM, VN_WG16_BIG_WN, VN_WG16_HYBRID ? 4 : VN_WG16_BIG_NB, VN_WG16_SIGMA_FUSED != 0, VN_WG16_HYBRID != 0>(big, nbig, n_chunks, st))) return rc;
    }
    ProfScope ps("wgrad_small", st);
// build switch VN_WG16_THIN_HYBRID (default 0, vipnerf_knobs.h): the 256 x 64 and 128 x 256 launches on the hybrid trip stream too (measured: docs/HISTORY.md 4.3a)
    if ((rc = launch_wg16<BF, 16, 4, 4, 1, VN_WG16_THIN_HYBRID ? 4 : 3, false, VN_WG16_THIN_HYBRID != 0>(pe, npe, n_pe, st))) return rc;
    if ((rc = launch_wg16<BF, 8, 16, 2, 2, VN_WG16_THIN_HYBRID ? 4 : 3, false, VN_WG16_THIN_HYBRID != 0>(vf, nvf, n_single, st))) return rc;
    if ((rc = launch_wg16<BF, 1, 16, 1, 4, 4>(sg, nsg, n_single, st))) return rc;
    if ((rc = launch_wg16<BF, 8, 2, 4, 1, 4>(vd, nvd, n_vd, st))) return rc;
    return launch_wg16<BF, 1, 8, 1, 4, 4>(oh, noh, n_oh, st);
}

int launch_wgrad16(size_t P, int V, const float *acts, const ActLayout &al, float *bwd, const BwdLayout &bl,
                   const vipnerf_mlp_grads *G, int precision, hipStream_t st, const unsigned *gmax) {
    if (P == 0) return VIPNERF_OK;
    if (P % 32) { set_error("wgrad16: %zu points (a multiple of 32 is required)", P); return VIPNERF_E_UNSUPPORTED; }
    // Point chunks: ONE round of workgroups per launch where the level is large enough -- (workgroups a CU holds of the class) x 256 CUs
    // over the launch's GEMMs -- and never more chunks than the plan the partial buffer was sized for (vipnerf_common.h): every chunk
    // costs a partial product written and read back by the reduction (a 256 x 64 partial is 66 KB against 640 B of operands per point).
    struct Plan { int n, pts; };
    auto plan = [&](int n_old, int pts_old, int slots, int n_desc) {
        const int target = slots / n_desc;
        if (target >= n_old) return Plan{n_old, pts_old};
        const int pts = (int)(((P + target - 1) / target + 31) / 32 * 32);
        return Plan{(int)((P + pts - 1) / pts), pts};
    };
    const int cp0 = wgrad_chunk_pts(P);
// build switch VN_WG16_BIG_SLOTS (default 256, vipnerf_knobs.h): workgroups of the 256 x 256 launch at a large level: one round of the chip (two / three rounds measured: docs/HISTORY.md 4.3a)
    const Plan pb = plan(wgrad_chunks(P), cp0, VN_WG16_BIG_SLOTS, 8);
    const Plan pp = plan(wgrad_chunks_split(P, WGRAD_SPLIT_PE), cp0 / WGRAD_SPLIT_PE, 512, 2);
    const Plan psg = plan(wgrad_chunks_split(P, WGRAD_SINGLE_SPLIT), cp0 / WGRAD_SINGLE_SPLIT, 512, 1);
    const Plan pd = plan(wgrad_chunks_split(P, WGRAD_SPLIT_THIN), cp0 / WGRAD_SPLIT_THIN, 768, 1 + V);
    const Plan po = plan(wgrad_chunks_split(P, WGRAD_SPLIT_THIN), cp0 / WGRAD_SPLIT_THIN, 1024, 1 + V);
    const int n_chunks = pb.n, n_pe = pp.n, n_single = psg.n, n_vd = pd.n, n_oh = po.n;
    float *partial = bwd + bl.partial;

    WgArgs big, pe, vf, sg, vd, oh;            // 256x256 x8 | 256x64 (gamma(x) columns) x2 | 128x256 | sigma head 16x256 | 128x32 per direction | 16x128 per direction
    int nbig = 0, npe = 0, nvf = 0, nsg = 0, nvd = 0, noh = 0, ng = 0;
    WgReduceArgs red;
    red.partial = partial;
    red.gmax = gmax;
    size_t off = 0;
    auto init = [&](WgArgs &w, int cp) { w.P = (int64_t)P; w.chunk_pts = cp; w.partial = partial; };
    init(big, pb.pts); init(pe, pp.pts); init(vf, psg.pts); init(sg, psg.pts); init(vd, pd.pts); init(oh, po.pts);
    auto add = [&](WgArgs &w, int &n, int chunks, int Mp, int Np, const float *A, const float *B) {
        WgDesc &d = w.d[n++];
        d.A = A; d.lda = Mp; d.m_load = Mp; d.B = B; d.ldb = Np; d.k_load = Np;
        d.wcol = nullptr; d.wcol_stride = 0; d.a_split16 = 0;
        d.part_off = off; d.part_stride = (size_t)Mp * Np + Mp; d.n_chunks = chunks;
        const size_t o = off;
        off += (size_t)chunks * d.part_stride;
        return o;
    };
    auto group = [&](int chunks, size_t part_off, int n_desc, int Mp, int Np, int m_valid, int k_valid, float *dW, int ldw, int col_off, float *dbias,
                     int colperm = 0, int rowperm = 3) -> WgGroup & {
        WgGroup &g = red.g[ng++];
        g.part_off = part_off; g.part_stride = (size_t)Mp * Np + Mp; g.n_desc = n_desc; g.n_chunks = chunks;
        g.desc_stride = (size_t)chunks * g.part_stride;
        g.Mp = Mp; g.Kp = Np; g.m_valid = m_valid; g.k_valid = k_valid; g.dW = dW; g.ldw = ldw; g.col_off = col_off; g.dbias = dbias;
        g.bias_off = (size_t)Mp * Np; g.colperm = colperm; g.rowperm = rowperm;
        return g;
    };
    const float *pex = acts + al.pex;
    for (int i = 0; i < D; ++i) {
        const float *dy = bwd + bl.dy[i];
        float *dW = G->g[2 * i], *db = G->g[2 * i + 1];
        if (i == 0) {
            const size_t o = add(pe, npe, n_pe, 256, 64, dy, pex);
            group(n_pe, o, 1, 256, 64, W, 64, dW, DPE, 0, db, 1);
        } else if (i == SKIP_LAYER) {
            const size_t o1 = add(pe, npe, n_pe, 256, 64, dy, pex);
            group(n_pe, o1, 1, 256, 64, W, 64, dW, W + DPE, 0, nullptr, 1);
            const size_t o2 = add(big, nbig, n_chunks, 256, 256, dy, acts + al.h[i - 1]);
            group(n_chunks, o2, 1, 256, 256, W, W, dW, W + DPE, DPE, db, 3);
        } else {
            const size_t o = add(big, nbig, n_chunks, 256, 256, dy, acts + al.h[i - 1]);
            group(n_chunks, o, 1, 256, 256, W, W, dW, W, 0, db, 3);
        }
    }
    {   // feature_linear
        const size_t o = add(big, nbig, n_chunks, 256, 256, bwd + bl.dyf, acts + al.h[D - 1]);
        WgGroup &gf = group(n_chunks, o, 1, 256, 256, W, W, G->g[P_FW], W, 0, G->g[P_FB], 3);
        if (VN_WG16_SIGMA_FUSED) {
            // + the sigma head as an extra A tile of this GEMM (the head-seed tile of direction 0, row 4, against the same h_8): its partial
            // [16][256] + 16 sums follows the GEMM's own in every chunk
            WgDesc &d = big.d[nbig - 1];
            constexpr size_t X = (size_t)16 * 256 + 16;
            d.wcol = bwd + bl.dq[0];
            d.part_stride += X;
            off += (size_t)n_chunks * X;
            gf.part_stride = d.part_stride; gf.desc_stride = (size_t)n_chunks * d.part_stride;
            WgGroup &g = group(n_chunks, o + (size_t)256 * 256 + 256 + 4 * 256, 1, 16, 256, 1, W, G->g[P_SW], W, 0, G->g[P_SB], 3, 0);
            g.part_stride = d.part_stride; g.desc_stride = gf.desc_stride;
            g.bias_off = (size_t)16 * 256 + 4 - 4 * 256;
        }
    }
    if (!VN_WG16_SIGMA_FUSED) {   // sigma head on its own: row 4 of the head-seed tile of direction 0 against h_8 (a second pass over h_8)
        const size_t o = add(sg, nsg, n_single, 16, 256, bwd + bl.dq[0], acts + al.h[D - 1]);
        WgGroup &g = group(n_single, o + 4 * 256, 1, 16, 256, 1, W, G->g[P_SW], W, 0, G->g[P_SB], 3, 0);
        g.bias_off = (size_t)16 * 256 + 4 - 4 * 256;
    }
    WgView16Args va;
    if (VN_WG16_VIEW_FUSED) {   // view layer: both column classes in one launch over dYv_0..V (k_wg16_view; no dYvsum)
        const Plan pv = plan(wgrad_chunks_split(P, WGRAD_SINGLE_SPLIT), cp0 / WGRAD_SINGLE_SPLIT, 256, 1);
        for (int k = 0; k <= VIPNERF_MAX_SEC; ++k) { va.dyv[k] = bwd + bl.dyv[k <= V ? k : 0]; va.ped[k] = acts + al.ped[k <= V ? k : 0]; }
        va.feat = acts + al.feat; va.P = (int64_t)P; va.chunk_pts = pv.pts; va.n_chunks = pv.n;
        va.stride_vf = (size_t)128 * 256 + 128; va.stride_vd = (size_t)128 * 32 + 128;
        va.part_vf = partial + off;
        group(pv.n, off, 1, 128, 256, WV, W, G->g[P_VW], W + DVE, 0, G->g[P_VB], 3);
        off += (size_t)pv.n * va.stride_vf;
        va.part_vd = partial + off;
        group(pv.n, off, 1, 128, 32, WV, 32, G->g[P_VW], W + DVE, W, nullptr, 2);
        off += (size_t)pv.n * va.stride_vd;
    } else {   // view layer, feature columns: A = sum over directions of dYv
        const size_t o = add(vf, nvf, n_single, 128, 256, bwd + bl.dyvsum, acts + al.feat);
        group(n_single, o, 1, 128, 256, WV, W, G->g[P_VW], W + DVE, 0, G->g[P_VB], 3);
    }
    {   // view layer, direction columns (gamma(dir) in slot order) and the output head: one GEMM per direction, summed in order
        size_t first_d = 0, first_o = 0;
        for (int k = 0; k <= V && !VN_WG16_VIEW_FUSED; ++k) {
            const size_t o = add(vd, nvd, n_vd, 128, 32, bwd + bl.dyv[k], acts + al.ped[k]);
            if (k == 0) first_d = o;
        }
        if (!VN_WG16_VIEW_FUSED) group(n_vd, first_d, 1 + V, 128, 32, WV, 32, G->g[P_VW], W + DVE, W, nullptr, 2);
        for (int k = 0; k <= V; ++k) {
            const size_t o = add(oh, noh, n_oh, 16, 128, bwd + bl.dq[k], acts + al.g[k]);
            if (k == 0) first_o = o;
        }
        group(n_oh, first_o, 1 + V, 16, 128, 4, WV, G->g[P_OW], WV, 0, G->g[P_OB], 3, 0);
    }
    if (off > wgrad_partial_total(P, V)) { set_error("wgrad16: partial buffer plan mismatch"); return VIPNERF_E_ARG; }

    int rc = precision == VIPNERF_PREC_BF16 ? launch_all<true>(big, nbig, n_chunks, pe, npe, n_pe, vf, nvf, sg, nsg, n_single, vd, nvd, n_vd, oh, noh, n_oh, st)
                                            : launch_all<false>(big, nbig, n_chunks, pe, npe, n_pe, vf, nvf, sg, nsg, n_single, vd, nvd, n_vd, oh, noh, n_oh, st);
    if (rc) return rc;
    ProfScope ps("wgrad_small", st);
    if (VN_WG16_VIEW_FUSED && (rc = precision == VIPNERF_PREC_BF16 ? launch_wg16_view_v<true>(va, V, st) : launch_wg16_view_v<false>(va, V, st))) return rc;
    return launch_wgrad_reduce(red, ng, st);
}

}  // namespace vn
