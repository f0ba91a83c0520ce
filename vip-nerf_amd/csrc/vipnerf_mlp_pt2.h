// The single-MFMA 16-bit MLP kernels with two point tiles per wave (vipnerf_mlp_fwd_pt2.hip, vipnerf_mlp_bwd_pt2.hip): shared pieces.
#pragma once
#include "vipnerf_bf16n.h"
#include "vipnerf_mlp.h"

#ifndef VN_PT2
#define VN_PT2 1            // 0: the 16-point kernels (k_mlp_fwd_bf16n / k_mlp_bwd_bf16n with NS = 1, H16 = 4) for VIPNERF_PREC_FP16 / BF16
#endif

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
    // VN_PT2_SPREAD = 1: all of a stage's stores behind its last MFMA group; = NSTEP (4, the default): one k-step's stores (4 instructions)
    // behind every NG / NSTEP-th group, so that the stage's store instructions do not queue at the vector-memory port at once --
    // measured on one box (bf16, 4096 rays): forward 1.66 -> 1.60 ms, data gradients 1.73 -> 1.61; 8 parts (one point tile's k-step each):
    // forward 1.63, data gradients 2.15 (the extra scheduling barriers cost registers: spills)
#ifndef VN_PT2_SPREAD
#define VN_PT2_SPREAD 4
#endif
    // (VN_PT2_SPREAD = 2 NSTEP: one k-step of ONE point tile -- two instructions -- behind every NG / (2 NSTEP)-th group)
    static constexpr int PARTS = VN_PT2_SPREAD > 1 ? (VN_PT2_SPREAD >= 2 * NSTEP ? 2 * NSTEP : NSTEP) : 1;
    template <int g, int NG> static constexpr bool active() { return (g + 1) % (NG / PARTS) == 0; }
    template <int g, int NG>
    __device__ __forceinline__ void at() const {
        if (EXP_NO_STORES) return;
        constexpr int part = (g + 1) / (NG / PARTS) - 1;
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
#endif

int launch_mlp_fwd_pt2(const MlpFwdArgs &a, int precision, hipStream_t st);
int launch_mlp_bwd_pt2(const MlpBwdArgs &a, int precision, hipStream_t st);

}  // namespace vn
