#include "vipnerf_wgrad.h"
namespace vn {
int launch_wgrad(size_t, int, const float *, const ActLayout &, float *, const BwdLayout &, const vipnerf_mlp_grads *, hipStream_t) {
    set_error("wgrad not built yet"); return VIPNERF_E_UNSUPPORTED; }
}
