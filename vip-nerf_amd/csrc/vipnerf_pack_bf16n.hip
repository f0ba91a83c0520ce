// Packing for the narrow-wave split-precision path (layout in vipnerf_bf16n.h): each weight is split into NS bf16
// parts and laid out as per-lane A fragments of v_mfma_f32_16x16x32_bf16 in consumption order.  One thread per
// 32-bit cell (two bf16 of one lane's 8-element fragment).
#include "vipnerf_bf16n.h"

namespace vn {

struct PackBnArgs {
    vipnerf_mlp_params pp[2];         // blockIdx.y selects the MLP: coarse and fine in ONE launch (vipnerf_pack_weights2_c)
    uint32_t *outs[2];
};

__device__ __forceinline__ _Float16 split_part_h(float x, int i) {
    _Float16 p = (_Float16)x;
    for (int k = 0; k < i; ++k) { x = x - (float)p; p = (_Float16)x; }
    return p;
}
template <bool F16>
__device__ __forceinline__ uint32_t pack2n(float w0, float w1, int part) {
    uint16_t ua, ub;
    if (F16) {
        ua = __builtin_bit_cast(uint16_t, split_part_h(w0 * F16_WSCALE, part));
        ub = __builtin_bit_cast(uint16_t, split_part_h(w1 * F16_WSCALE, part));
    } else {
        ua = __builtin_bit_cast(uint16_t, split_part(w0, part));
        ub = __builtin_bit_cast(uint16_t, split_part(w1, part));
    }
    return (uint32_t)ua | ((uint32_t)ub << 16);
}

// F32 (fp32-narrow image): a 32-bit cell is ONE weight -- element e = 4 part + (cell & 3) of the lane's 8-float k-step
// operand (f32q parts, vipnerf_bf16.h) -- instead of two 16-bit parts of elements e0, e0 + 1.
template <int NS, bool F16, bool F32 = false>
__global__ void k_pack_bf16n(PackBnArgs a2) {
    typedef BnPlan<NS> PL;
    constexpr int NU = F32 ? 1 : 2;              // weights per 32-bit cell
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= PL::PK_TOTAL_F) return;
    const struct { const vipnerf_mlp_params &p; uint32_t *out; } a = {a2.pp[blockIdx.y], a2.outs[blockIdx.y]};
    uint32_t cell = 0;
    if (idx < PL::PK_RES) {
        const bool bwd = idx >= PL::PK_BWD;
        const size_t i0 = bwd ? idx - PL::PK_BWD : idx;
        const int s = (int)(i0 / PL::STAGE_F);
        const int w = (int)(i0 % PL::STAGE_F);
        const int c = w / CHUNK_F, lane = (w % CHUNK_F) >> 2;
        const int q = lane >> 4, i = lane & 15;
        const int part = c % NS, cellno = c / NS;
        const int e0 = F32 ? 4 * part + (w & 3) : 2 * (w & 3);
        float v[2] = {0.f, 0.f};
        if (!bwd) {
            if (s < PL::FS_L1 || (s >= PL::FS_L5PE && s < PL::FS_L6)) {            // gamma(x) columns of layer 0 / 5
                const int layer = s < PL::FS_L1 ? 0 : SKIP_LAYER;
                const int j = s < PL::FS_L1 ? s - PL::FS_L0PE : s - PL::FS_L5PE;
                const int t = cellno % 16, ks = PL::KSB * j + cellno / 16;
                for (int u = 0; u < NU; ++u) {
                    const int k = pe_feat16(ks, q, e0 + u);
                    v[u] = k >= 0 ? a.p.p[2 * layer][(size_t)(16 * t + i) * layer_in_dim(layer) + k] : 0.f;
                }
            } else if (s < PL::FS_VIEW) {                                          // 256-deep register-sourced layers
                int layer, j, koff = 0;
                if (s < PL::FS_L5) { layer = 1 + (s - PL::FS_L1) / PL::ST_256; j = (s - PL::FS_L1) % PL::ST_256; }
                else if (s < PL::FS_L5PE) { layer = 5; j = s - PL::FS_L5; koff = DPE; }
                else if (s < PL::FS_L7) { layer = 6; j = s - PL::FS_L6; }
                else if (s < PL::FS_FEAT) { layer = 7; j = s - PL::FS_L7; }
                else { layer = 8; j = s - PL::FS_FEAT; }
                const float *wp = layer < 8 ? a.p.p[2 * layer] : a.p.p[P_FW];
                const int ld = layer < 8 ? layer_in_dim(layer) : W;
                const int t = cellno % 16, ks = PL::KSB * j + cellno / 16;
                for (int u = 0; u < NU; ++u) v[u] = wp[(size_t)(16 * t + i) * ld + koff + feat16(ks, q, e0 + u)];
            } else {                                                               // view layer, feature columns (8 tiles)
                const int j = s - PL::FS_VIEW;
                const int t = cellno % 8, ks = PL::KSV * j + cellno / 8;
                for (int u = 0; u < NU; ++u) v[u] = a.p.p[P_VW][(size_t)(16 * t + i) * (W + DVE) + feat16(ks, q, e0 + u)];
            }
        } else {                                                                   // dgrad: A = W^T, 16 tiles of input features
            const int t = cellno % 16, ksl = cellno / 16;
            const int k = 16 * t + i;
            for (int u = 0; u < NU; ++u) {
                if (s < PL::BS_FEAT) {
                    const int r = feat16(PL::KSB * s + ksl, q, e0 + u);            // < 128: output feature of the view layer
                    v[u] = a.p.p[P_VW][(size_t)r * (W + DVE) + k];
                } else if (s < PL::BS_L7) {
                    const int r = feat16(PL::KSB * (s - PL::BS_FEAT) + ksl, q, e0 + u);
                    v[u] = a.p.p[P_FW][(size_t)r * W + k];
                } else {
                    const int layer = 7 - (s - PL::BS_L7) / PL::ST_256;
                    const int j = (s - PL::BS_L7) % PL::ST_256;
                    const int r = feat16(PL::KSB * j + ksl, q, e0 + u);
                    v[u] = a.p.p[2 * layer][(size_t)r * layer_in_dim(layer) + (layer == SKIP_LAYER ? DPE : 0) + k];
                }
            }
        }
        cell = F32 ? __float_as_uint(v[0]) : pack2n<F16>(v[0], v[1], part);
    } else {
        const int i0 = (int)(idx - PL::PK_RES);
        if (i0 < PL::R_DIRW_F) {                                                   // direction columns, chunk t * NS + part
            const int c = i0 / CHUNK_F, lane = (i0 % CHUNK_F) >> 2;
            const int part = c % NS, t = c / NS, q = lane >> 4, i = lane & 15;
            const int e0 = F32 ? 4 * part + (i0 & 3) : 2 * (i0 & 3);
            float v[2] = {0.f, 0.f};
            for (int u = 0; u < NU; ++u) {
                const int kk = dir_feat16(q, e0 + u);
                v[u] = kk >= 0 ? a.p.p[P_VW][(size_t)(16 * t + i) * (W + DVE) + W + kk] : 0.f;
            }
            cell = F32 ? __float_as_uint(v[0]) : pack2n<F16>(v[0], v[1], part);
        } else if (i0 < PL::R_TOTAL) {                                             // fp32 biases / heads, natural order
            const int f = i0 - PL::R_F32;
            float v = 0.f;
            if (f < PL::N_BFEAT) v = a.p.p[2 * (f / W) + 1][f % W];
            else if (f < PL::N_BVIEW) v = a.p.p[P_FB][f - PL::N_BFEAT];
            else if (f < PL::N_WSIG) v = a.p.p[P_VB][f - PL::N_BVIEW];
            else if (f < PL::N_WOUT) v = a.p.p[P_SW][f - PL::N_WSIG];
            else if (f < PL::N_BHEAD) v = a.p.p[P_OW][f - PL::N_WOUT];             // [4][128] row-major like nn.Linear
            else {
                const int rem = f - PL::N_BHEAD;
                v = rem == 0 ? a.p.p[P_SB][0] : (rem <= 4 ? a.p.p[P_OB][rem - 1] : 0.f);
            }
            if (F16 && f < PL::N_WSIG) v *= F16_ACC_SCALE;                         // layer biases enter the accumulators scaled
            cell = __float_as_uint(v);
        }
    }
    a.out[idx] = cell;
}

int persistent_grid() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return cus;
    }();
    return n;
}

// p2 / packed2: a second MLP packed by the same launch (NULL: one)
int launch_pack_bf16n(const vipnerf_mlp_params *p, int precision, void *packed_bn, hipStream_t st, const vipnerf_mlp_params *p2, void *packed2) {
    PackBnArgs a;
    a.pp[0] = *p;
    a.outs[0] = (uint32_t *)packed_bn;
    a.pp[1] = p2 ? *p2 : *p;
    a.outs[1] = (uint32_t *)(p2 ? packed2 : packed_bn);
    const unsigned ny = p2 ? 2 : 1;
    const int bs = 256;
    if (precision == 0) {
        hipLaunchKernelGGL((k_pack_bf16n<2, false, true>), dim3((unsigned)((BnPlan<2>::PK_TOTAL_F + bs - 1) / bs), ny), dim3(bs), 0, st, a);
    } else if (precision == 3 || precision == 4) {
        hipLaunchKernelGGL((k_pack_bf16n<2, true>), dim3((unsigned)((BnPlan<2>::PK_TOTAL_F + bs - 1) / bs), ny), dim3(bs), 0, st, a);
    } else if (precision == 5) {
        hipLaunchKernelGGL((k_pack_bf16n<1, true>), dim3((unsigned)((BnPlan<1>::PK_TOTAL_F + bs - 1) / bs), ny), dim3(bs), 0, st, a);
    } else if (precision == 6) {
        hipLaunchKernelGGL((k_pack_bf16n<1, false>), dim3((unsigned)((BnPlan<1>::PK_TOTAL_F + bs - 1) / bs), ny), dim3(bs), 0, st, a);
    } else {
        set_error("pack_bf16n: precision %d", precision);
        return VIPNERF_E_ARG;
    }
    VN_HIP(hipGetLastError());
    return VIPNERF_OK;
}

}  // namespace vn
