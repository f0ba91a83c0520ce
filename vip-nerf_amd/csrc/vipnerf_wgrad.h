// Weight-gradient GEMMs (vipnerf_wgrad.hip): dW[M][K] = sum_p A[p][M] * B[p][K] over the points of a level.
#pragma once
#include "vipnerf_common.h"

namespace vn {
int launch_wgrad(size_t P, int V, const float *acts, const ActLayout &al, float *bwd, const BwdLayout &bl,
                 const vipnerf_mlp_grads *G, int precision, hipStream_t st, const unsigned *gmax = nullptr);
}
