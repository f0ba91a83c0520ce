// Weight packing: nn.Linear tensors -> the streaming image the MLP kernels consume (layout in
// vipnerf_common.h).  One thread per packed float; ~1.15 M floats per MLP, runs once per optimizer step.
#include "vipnerf_common.h"

namespace vn {

struct PackArgs {
    vipnerf_mlp_params p;
    float *out;
};

__device__ __forceinline__ const float *layer_w(const vipnerf_mlp_params &p, int l) { return p.p[2 * l]; }

__global__ void k_pack(PackArgs a) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= PK_TOTAL_F) return;
    float v = 0.f;
    if (idx < PK_RES) {
        const bool bwd = idx >= PK_BWD;
        const size_t i = bwd ? idx - PK_BWD : idx;
        const int s = (int)(i / STAGE_F);
        const int w = (int)(i % STAGE_F);
        const int c = w / CHUNK_F, lane = (w % CHUNK_F) >> 2, q = w & 3;
        const int h = lane >> 5, l31 = lane & 31;
        if (!bwd) {
            if (s < FS_L1 || (s >= FS_L5PE && s < FS_L5)) {                     // gamma(x) part of layer 0 / 5
                const int layer = s < FS_L1 ? 0 : SKIP_LAYER;
                const int j = s < FS_L1 ? s - FS_L0PE : s - FS_L5PE;
                const int gl = c >> 3, t = c & 7;
                const int k = 2 * (4 * (KGS8 * j + gl) + q) + h;
                const int o = 32 * t + l31;
                v = k < DPE ? layer_w(a.p, layer)[(size_t)o * layer_in_dim(layer) + k] : 0.f;
            } else if (s < FS_VIEW) {                                           // 256-wide register-sourced layers
                int layer, j;
                const float *wp;
                int ld, koff = 0;
                if (s < FS_L5PE) { layer = 1 + (s - FS_L1) / ST_256; j = (s - FS_L1) % ST_256; }
                else if (s < FS_L6) { layer = 5; j = s - FS_L5; koff = DPE; }
                else if (s < FS_L7) { layer = 6; j = s - FS_L6; }
                else if (s < FS_FEAT) { layer = 7; j = s - FS_L7; }
                else { layer = 8; j = s - FS_FEAT; }
                if (layer < 8) { wp = layer_w(a.p, layer); ld = layer_in_dim(layer); }
                else { wp = a.p.p[P_FW]; ld = W; }
                const int gl = c >> 3, t = c & 7;
                const int r = 4 * (KGS8 * j + gl) + q;
                v = wp[(size_t)(32 * t + l31) * ld + koff + feat_of(r, h)];
            } else {                                                            // view layer, feature columns
                const int j = s - FS_VIEW;
                const int gl = c >> 2, t = c & 3;
                const int r = 4 * (KGS4 * j + gl) + q;
                v = a.p.p[P_VW][(size_t)(32 * t + l31) * (W + DVE) + feat_of(r, h)];
            }
        } else {
            const int gl = c >> 3, t = c & 7;
            const int k = 32 * t + l31;                                         // dgrad output row = input feature
            if (s < BS_FEAT) {
                const int r = 4 * (KGS8 * s + gl) + q;                          // r < 64 -> o < 128
                v = a.p.p[P_VW][(size_t)feat_of(r, h) * (W + DVE) + k];
            } else if (s < BS_L7) {
                const int r = 4 * (KGS8 * (s - BS_FEAT) + gl) + q;
                v = a.p.p[P_FW][(size_t)feat_of(r, h) * W + k];
            } else {
                const int layer = 7 - (s - BS_L7) / ST_256;
                const int j = (s - BS_L7) % ST_256;
                const int r = 4 * (KGS8 * j + gl) + q;
                const int koff = layer == SKIP_LAYER ? DPE : 0;
                v = layer_w(a.p, layer)[(size_t)feat_of(r, h) * layer_in_dim(layer) + koff + k];
            }
        }
    } else {
        const int i = (int)(idx - PK_RES);
        if (i < R_BIAS) {                                                       // direction columns of the view layer
            const int c = i / CHUNK_F, lane = (i % CHUNK_F) >> 2, q = i & 3;
            const int g = c >> 2, t = c & 3, h = lane >> 5;
            const int kk = 2 * (4 * g + q) + h;
            v = kk < DVE ? a.p.p[P_VW][(size_t)(32 * t + (lane & 31)) * (W + DVE) + W + kk] : 0.f;
        } else if (i < R_BVIEW + WV) {                                          // biases in C/D fragment order
            const float *b;
            int rem;
            if (i < R_BFEAT) { b = a.p.p[2 * ((i - R_BIAS) / W) + 1]; rem = (i - R_BIAS) % W; }
            else if (i < R_BVIEW) { b = a.p.p[P_FB]; rem = i - R_BFEAT; }
            else { b = a.p.p[P_VB]; rem = i - R_BVIEW; }
            const int t = rem >> 5, h = (rem >> 4) & 1, r = rem & 15;
            v = b[32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
        } else if (i < R_WOUT) {
            const int rem = i - R_WSIG;
            v = a.p.p[P_SW][feat_of(rem & 127, rem >> 7)];
        } else if (i < R_BHEAD) {
            const int rem = i - R_WOUT;
            const int h = rem >> 8, c = (rem >> 6) & 3, r = rem & 63;
            v = a.p.p[P_OW][c * WV + feat_of(r, h)];
        } else if (i < R_TOTAL) {
            const int rem = i - R_BHEAD;
            v = rem == 0 ? a.p.p[P_SB][0] : (rem <= 4 ? a.p.p[P_OB][rem - 1] : 0.f);
        }
    }
    a.out[idx] = v;
}

int launch_pack(const vipnerf_mlp_params *p, void *packed, hipStream_t st) {
    PackArgs a;
    a.p = *p;
    a.out = (float *)packed;
    const int bs = 256;
    const unsigned grid = (unsigned)((PK_TOTAL_F + bs - 1) / bs);
    hipLaunchKernelGGL(k_pack, dim3(grid), dim3(bs), 0, st, a);
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

}  // namespace vn
