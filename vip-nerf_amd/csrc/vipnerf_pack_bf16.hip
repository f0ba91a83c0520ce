// Packing for the split-precision bf16 path (layout in vipnerf_bf16.h): each weight is split into NS bf16 parts and
// laid out as per-lane A fragments of v_mfma_f32_32x32x16_bf16 in consumption order.  One thread per 32-bit cell
// (two bf16 of one lane's 8-element fragment).
#include "vipnerf_bf16.h"

namespace vn {

struct PackBfArgs {
    vipnerf_mlp_params p;
    uint32_t *out;
};

__device__ __forceinline__ uint32_t pack2(float w0, float w1, int part) {
    const __bf16 a = split_part(w0, part), b = split_part(w1, part);
    const uint16_t ua = __builtin_bit_cast(uint16_t, a), ub = __builtin_bit_cast(uint16_t, b);
    return (uint32_t)ua | ((uint32_t)ub << 16);
}

template <int NS>
__global__ void k_pack_bf16(PackBfArgs a) {
    typedef BfPlan<NS> PL;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= PL::PK_TOTAL_F) return;
    uint32_t cell = 0;
    if (idx < PL::PK_RES) {
        const bool bwd = idx >= PL::PK_BWD;
        const size_t i = bwd ? idx - PL::PK_BWD : idx;
        const int s = (int)(i / PL::STAGE_F);
        const int w = (int)(i % PL::STAGE_F);
        const int c = w / CHUNK_F, lane = (w % CHUNK_F) >> 2, e0 = 2 * (w & 3);
        const int h = lane >> 5, l31 = lane & 31;
        float v[2] = {0.f, 0.f};
        int part = 0;
        if (!bwd) {
            if (s < PL::FS_L1 || (s >= PL::FS_L5PE && s < PL::FS_L5)) {            // gamma(x) columns of layer 0 / 5
                const int layer = s < PL::FS_L1 ? 0 : SKIP_LAYER;
                const int j = s < PL::FS_L1 ? s - PL::FS_L0PE : s - PL::FS_L5PE;
                part = c % NS;
                const int t = (c / NS) % 8, ks = PL::KSB * j + (c / NS) / 8;
                const int o = 32 * t + l31;
                for (int q = 0; q < 2; ++q) {
                    const int k = 16 * ks + 8 * h + e0 + q;
                    v[q] = k < DPE ? a.p.p[2 * layer][(size_t)o * layer_in_dim(layer) + k] : 0.f;
                }
            } else if (s < PL::FS_VIEW) {                                          // 256-deep register-sourced layers
                int layer, j, koff = 0;
                if (s < PL::FS_L5PE) { layer = 1 + (s - PL::FS_L1) / PL::ST_256; j = (s - PL::FS_L1) % PL::ST_256; }
                else if (s < PL::FS_L6) { layer = 5; j = s - PL::FS_L5; koff = DPE; }
                else if (s < PL::FS_L7) { layer = 6; j = s - PL::FS_L6; }
                else if (s < PL::FS_FEAT) { layer = 7; j = s - PL::FS_L7; }
                else { layer = 8; j = s - PL::FS_FEAT; }
                const float *wp = layer < 8 ? a.p.p[2 * layer] : a.p.p[P_FW];
                const int ld = layer < 8 ? layer_in_dim(layer) : W;
                part = c % NS;
                const int t = (c / NS) % 8, ks = PL::KSB * j + (c / NS) / 8;       // k-step = 2*T + u
                for (int q = 0; q < 2; ++q)
                    v[q] = wp[(size_t)(32 * t + l31) * ld + koff + feat_of(8 * ks + e0 + q, h)];
            } else {                                                               // view layer, feature columns (4 tiles)
                const int j = s - PL::FS_VIEW;
                part = c % NS;
                const int t = (c / NS) % 4, ks = PL::KSV * j + (c / NS) / 4;
                for (int q = 0; q < 2; ++q)
                    v[q] = a.p.p[P_VW][(size_t)(32 * t + l31) * (W + DVE) + feat_of(8 * ks + e0 + q, h)];
            }
        } else {                                                                   // dgrad: A = W^T, 8 tiles of input features
            part = c % NS;
            const int t = (c / NS) % 8, ksl = (c / NS) / 8;
            const int k = 32 * t + l31;
            for (int q = 0; q < 2; ++q) {
                if (s < PL::BS_FEAT) {
                    const int r = 8 * (PL::KSB * s + ksl) + e0 + q;                // < 64 -> output feature < 128
                    v[q] = a.p.p[P_VW][(size_t)feat_of(r, h) * (W + DVE) + k];
                } else if (s < PL::BS_L7) {
                    const int r = 8 * (PL::KSB * (s - PL::BS_FEAT) + ksl) + e0 + q;
                    v[q] = a.p.p[P_FW][(size_t)feat_of(r, h) * W + k];
                } else {
                    const int layer = 7 - (s - PL::BS_L7) / PL::ST_256;
                    const int j = (s - PL::BS_L7) % PL::ST_256;
                    const int r = 8 * (PL::KSB * j + ksl) + e0 + q;
                    v[q] = a.p.p[2 * layer][(size_t)feat_of(r, h) * layer_in_dim(layer) + (layer == SKIP_LAYER ? DPE : 0) + k];
                }
            }
        }
        cell = pack2(v[0], v[1], part);
    } else {
        const int i = (int)(idx - PL::PK_RES);
        if (i < PL::R_DIRW_F) {                                                    // direction columns, chunk (ks*4 + t)*NS + part
            const int c = i / CHUNK_F, lane = (i % CHUNK_F) >> 2, e0 = 2 * (i & 3);
            const int part = c % NS, t = (c / NS) % 4, ks = (c / NS) / 4, h = lane >> 5;
            float v[2];
            for (int q = 0; q < 2; ++q) {
                const int kk = 16 * ks + 8 * h + e0 + q;
                v[q] = kk < DVE ? a.p.p[P_VW][(size_t)(32 * t + (lane & 31)) * (W + DVE) + W + kk] : 0.f;
            }
            cell = pack2(v[0], v[1], part);
        } else if (i < PL::R_TOTAL) {                                              // fp32 biases / heads: same as the fp32 image
            const int f = R_BIAS + (i - PL::R_F32);
            float v = 0.f;
            if (f < R_BVIEW + WV) {
                const float *b;
                int rem;
                if (f < R_BFEAT) { b = a.p.p[2 * ((f - R_BIAS) / W) + 1]; rem = (f - R_BIAS) % W; }
                else if (f < R_BVIEW) { b = a.p.p[P_FB]; rem = f - R_BFEAT; }
                else { b = a.p.p[P_VB]; rem = f - R_BVIEW; }
                const int t = rem >> 5, hh = (rem >> 4) & 1, r = rem & 15;
                v = b[32 * t + (r & 3) + 8 * (r >> 2) + 4 * hh];
            } else if (f < R_WOUT) {
                const int rem = f - R_WSIG;
                v = a.p.p[P_SW][feat_of(rem & 127, rem >> 7)];
            } else if (f < R_BHEAD) {
                const int rem = f - R_WOUT;
                v = a.p.p[P_OW][((rem >> 6) & 3) * WV + feat_of(rem & 63, rem >> 8)];
            } else {
                const int rem = f - R_BHEAD;
                v = rem == 0 ? a.p.p[P_SB][0] : (rem <= 4 ? a.p.p[P_OB][rem - 1] : 0.f);
            }
            cell = __float_as_uint(v);
        }
    }
    a.out[idx] = cell;
}

int launch_pack_bf16(const vipnerf_mlp_params *p, int precision, void *packed_bf, hipStream_t st) {
    PackBfArgs a;
    a.p = *p;
    a.out = (uint32_t *)packed_bf;
    const int bs = 256;
    if (precision == 1) {
        hipLaunchKernelGGL(k_pack_bf16<2>, dim3((unsigned)((BfPlan<2>::PK_TOTAL_F + bs - 1) / bs)), dim3(bs), 0, st, a);
    } else if (precision == 2) {
        hipLaunchKernelGGL(k_pack_bf16<3>, dim3((unsigned)((BfPlan<3>::PK_TOTAL_F + bs - 1) / bs)), dim3(bs), 0, st, a);
    } else {
        set_error("pack_bf16: precision %d", precision);
        return VIPNERF_E_ARG;
    }
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

}  // namespace vn
