// Weight-gradient GEMMs (vipnerf_wgrad.hip, vipnerf_wgrad16.hip): dW[M][K] = sum_p A[p][M] * B[p][K] over the points of a level.
#pragma once
#include "vipnerf_bf16n.h"

namespace vn {

struct WgDesc {
    const float *A; int lda; int m_load;      // A[p][0..m_load) is read (m_load multiple of 4), zero beyond
    const float *B; int ldb; int k_load;
    size_t part_off;                          // float offset of this GEMM's partials (chunk 0)
    int n_chunks;                             // chunks of this GEMM (workgroups beyond it exit)
    size_t part_stride;                       // floats per chunk: Mp*Kp + Mp (+ WCOL_EXTRA with wcol)
    // 256x256 split-precision kernel only: a per-point weight column w[p] = wcol[p * wcol_stride].  The workgroup also
    // accumulates sum_p w[p] * B[p][0..256) and sum_p w[p] into its partial (after the bias sums) -- the weight and
    // bias gradient of a 1-output head fed by B (the sigma head reads the same h_8 as the feature layer's GEMM)
    const float *wcol; int wcol_stride;
    // k_wgrad_bf16x3 only: A is stored pre-split (store_pair_split: the 16 bytes of 4 features hold [hi f0 f1][hi f2 f3]
    // [lo f0 f1][lo f2 f3] as fp16) instead of 4 floats -- layer 5's gradient, which the 256x256 GEMM reads in that form
    int a_split16;
};
constexpr int WCOL_EXTRA = 256 + 64;          // 256 weighted column sums, the weight sum, pad
constexpr int WG_MAX_DESC = 12;
struct WgArgs {
    WgDesc d[WG_MAX_DESC];
    int64_t P;
    int chunk_pts;
    float *partial;
};

// output groups for the ordered reduction
struct WgGroup {
    size_t part_off, part_stride;             // of the group's first GEMM
    int n_desc;                               // GEMMs summed into this output (consecutive, same shape)
    size_t desc_stride;                       // float distance between consecutive GEMMs' partial blocks
    int Mp, Kp, m_valid, k_valid, n_chunks;
    float *dW; int ldw; int col_off;
    float *dbias;                             // NULL = no bias output
    size_t bias_off;                          // float offset of the column sums inside a chunk's partial block (default Mp * Kp)
    int colperm;                              // 0: partial column k is feature k; 1 / 2: the slot order of gamma(x) / gamma(dir) in T16 storage;
                                              // 3: the stored order of a T16 array written from C/D fragments (t16_feature)
    int rowperm;                              // 0 / 3: the same for the partial's rows (the A operand's features)
};
constexpr int WG_MAX_GROUP = 24;
struct WgReduceArgs {
    WgGroup g[WG_MAX_GROUP];
    const float *partial;
    const unsigned *gmax;                     // FP16X3: the partials are 2^S times the gradients (grad_scale_from_max)
};


// column k of a partial product -> feature index of the nn.Linear weight (-1: padding column)
__host__ __device__ inline int wg_colperm(int mode, int k) {
    if (mode == 1) return pe_feat16(0, k >> 4, k & 15);      // gamma(x) slots: column 16 q + u
    if (mode == 2) return dir_feat16(k >> 3, k & 7);         // gamma(dir) slots: column 8 q + e
    if (mode == 3) return t16_feature(k);
    return k;
}

int launch_wgrad(size_t P, int V, const float *acts, const ActLayout &al, float *bwd, const BwdLayout &bl,
                 const vipnerf_mlp_grads *G, int precision, hipStream_t st, const unsigned *gmax = nullptr);
// the single-MFMA 16-bit modes with T16 operand storage (vipnerf_wgrad16.hip)
int launch_wgrad16(size_t P, int V, const float *acts, const ActLayout &al, float *bwd, const BwdLayout &bl,
                   const vipnerf_mlp_grads *G, int precision, hipStream_t st, const unsigned *gmax);
int launch_wgrad_reduce(const WgReduceArgs &red, int ng, hipStream_t st);
}
