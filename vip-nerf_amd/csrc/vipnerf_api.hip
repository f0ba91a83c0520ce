// C ABI of libvipnerf_hip.so (include/vipnerf_hip.h): argument checking and kernel sequencing.  No device
// synchronisation, no allocation; everything is queued on the caller's stream.
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include <mutex>
#include <string>
#include <vector>

#include <cstdlib>
#include <cstring>
#include "vipnerf_bf16n.h"
#include "vipnerf_camera.h"
#include "vipnerf_generic.h"
#include "vipnerf_mlp.h"
#include "vipnerf_mlp_pt2.h"
#include "vipnerf_prof.h"
#include "vipnerf_ray.h"
#include "vipnerf_wgrad.h"

namespace vn {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- per-kernel event timing -----------------------------------------------------------------------------
struct ProfRec { const char *name; hipEvent_t e0, e1; };
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;

static hipEvent_t prof_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

ProfScope::ProfScope(const char *name, hipStream_t s) : idx(-1), st(s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r{name, prof_event(), prof_event()};
    if (!r.e0 || !r.e1) return;
    if (hipEventRecord(r.e0, st) != hipSuccess) return;       // profiling only: a scope that cannot be timed is dropped, not an error
    g_recs.push_back(r);
    idx = (int)g_recs.size() - 1;
}
ProfScope::~ProfScope() {
    if (idx < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (idx < (int)g_recs.size()) (void)hipEventRecord(g_recs[idx].e1, st);   // a failed record shows up as a failed hipEventElapsedTime in profile_read, which skips the scope
}

// hipGetLastError() is sticky per thread: an error left behind by an unrelated earlier HIP call (e.g. a probe while
// the device was still waking up) must not be reported as the failure of one of our launches.
static inline void clear_stale_hip_error() { (void)hipGetLastError(); }

static bool precision_retired(int precision) { return precision == VIPNERF_PREC_BF16X3 || precision == VIPNERF_PREC_BF16X6; }

static int check_cfg(const vipnerf_config *cfg) {
    clear_stale_hip_error();
    if (!cfg) { set_error("cfg is NULL"); return VIPNERF_E_ARG; }
    // Any sample counts, as in the reference (VipNeRF01.py:173-216): the kernels index points, not 32-sample blocks (a point's ray is p / S per lane;
    // the per-ray kernels give a lane ceil(S / 64) samples and predicate the tail).  Measured against the oracle for 5 + 11 ... 100 + 156, odd counts
    // and odd ray numbers included (profiles/r06_sample_counts_probe.log).  Limits: <= 256 samples per ray and level (four per lane), >= 2 coarse
    // samples, >= 3 with importance sampling (sample_pdf's bins).  The 16-bit TRAINING kernels additionally need a level's POINT count (rays x
    // samples) to be a multiple of 32 -- their T16 tile storage -- and say so per call (launch_mlp_fwd_pt2, launch_wgrad16).
    if (cfg->n_coarse < 2 || cfg->n_coarse > 256) {
        set_error("n_coarse=%d unsupported (2..256)", cfg->n_coarse); return VIPNERF_E_UNSUPPORTED; }
    if (cfg->n_fine < 0 || (cfg->n_fine > 0 && (cfg->n_coarse < 3 || cfg->n_coarse + cfg->n_fine > 256))) {
        set_error("n_fine=%d with n_coarse=%d unsupported (n_coarse >= 3, n_coarse + n_fine <= 256)", cfg->n_fine, cfg->n_coarse); return VIPNERF_E_UNSUPPORTED; }
    if (cfg->n_sec < 0 || cfg->n_sec > VIPNERF_MAX_SEC) {
        set_error("n_sec=%d unsupported (0..%d)", cfg->n_sec, VIPNERF_MAX_SEC); return VIPNERF_E_UNSUPPORTED; }
    if (cfg->precision < 0 || cfg->precision > VIPNERF_PREC_BF16) {
        set_error("precision=%d unsupported", cfg->precision); return VIPNERF_E_UNSUPPORTED; }
    if (precision_retired(cfg->precision)) {
        set_error("precision=%d (split bf16: bf16x3 / bf16x6) was retired with ABI 5 -- no BASELINE configuration uses it; fp16x3 is the fp32-grade "
                  "fast arithmetic, bf16 / fp16 the mixed-precision ones", cfg->precision); return VIPNERF_E_UNSUPPORTED; }
    if (cfg->bf16_layout == VIPNERF_LAYOUT_WIDE) {
        set_error("bf16_layout = VIPNERF_LAYOUT_WIDE (32-point waves, one per SIMD) was retired with ABI 5: every kernel runs the narrow layout");
        return VIPNERF_E_UNSUPPORTED; }
    if (cfg->bf16_layout < VIPNERF_LAYOUT_DEFAULT || cfg->bf16_layout > VIPNERF_LAYOUT_NARROW) {
        set_error("bf16_layout=%d unsupported", cfg->bf16_layout); return VIPNERF_E_UNSUPPORTED; }
    const int dpt = cfg->netdepth ? cfg->netdepth : D, wid = cfg->netwidth ? cfg->netwidth : W;
    const int lp = cfg->pe_degrees ? (cfg->pe_degrees & 0xff) : LP, lv = cfg->pe_degrees ? ((cfg->pe_degrees >> 8) & 0xff) : LV;
    if (dpt < 1 || dpt > D || wid < 8 || wid > W || wid % 8 || lp < 0 || lp > 16 || lv < 0 || lv > 8) {
        set_error("MLP topology netdepth=%d netwidth=%d degrees %d/%d unsupported (depth 1..8, width 8..256 multiple of 8, degrees <= 16 / 8)",
                  dpt, wid, lp, lv); return VIPNERF_E_UNSUPPORTED; }
    if (cfg->head_variant < 0 || cfg->head_variant > (VIPNERF_HEAD_RGB_TRUNK | VIPNERF_HEAD_NO_VISIBILITY)) {
        set_error("head_variant=%d unsupported (VIPNERF_HEAD_* bits)", cfg->head_variant); return VIPNERF_E_UNSUPPORTED; }
    if ((cfg->head_variant & VIPNERF_HEAD_NO_VISIBILITY) && cfg->n_sec > 0) {
        set_error("n_sec=%d with VIPNERF_HEAD_NO_VISIBILITY: an MLP that predicts no visibility has no secondary views (VipNeRF01.py:84)", cfg->n_sec);
        return VIPNERF_E_UNSUPPORTED; }
    if (!gen_is_fused_topology(gen_topo(dpt, wid, lp, lv, cfg->head_variant)) && cfg->precision != VIPNERF_PREC_FP32) {
        set_error("the generic-topology kernels (netdepth=%d netwidth=%d head_variant=%d) are fp32 only; precision=%d", dpt, wid, cfg->head_variant,
                  cfg->precision);
        return VIPNERF_E_UNSUPPORTED; }
    return VIPNERF_OK;
}
static GenTopo cfg_topo(const vipnerf_config *cfg) {
    return gen_topo(cfg->netdepth ? cfg->netdepth : D, cfg->netwidth ? cfg->netwidth : W, cfg->pe_degrees ? (cfg->pe_degrees & 0xff) : LP,
                    cfg->pe_degrees ? ((cfg->pe_degrees >> 8) & 0xff) : LV, cfg->head_variant);
}
static bool cfg_generic(const vipnerf_config *cfg) { return !gen_is_fused_topology(cfg_topo(cfg)); }

int launch_mlp_fwd_bf16n(const MlpFwdArgs &a, int precision, hipStream_t st);
int launch_mlp_bwd_bf16n(const MlpBwdArgs &a, int precision, hipStream_t st);

// ONE lane layout (the "narrow" one: 16-point waves, two per SIMD; vipnerf_bf16n.h) and one packed image per precision.  The round-1
// "wide" generation (32-point waves, one per SIMD: k_mlp_fwd / k_mlp_bwd / k_mlp_*_bf16) and the split-bf16 arithmetics (bf16x3, bf16x6)
// were retired with ABI 5 (docs/HISTORY.md 4.1, 4.1b hold their measurements).
static size_t packed_floats_all(int precision) { return packed_narrow_floats(precision); }

static int launch_mlp_fwd_any(MlpFwdArgs &a, int precision, hipStream_t st) {
    if (precision_retired(precision)) { set_error("precision=%d (split bf16) was retired with ABI 5", precision); return VIPNERF_E_UNSUPPORTED; }
    if (precision == VIPNERF_PREC_FP16 || precision == VIPNERF_PREC_BF16) {                       // single-MFMA modes: two point tiles per wave
        if (!single_mfma_t16(precision)) { set_error("this library was built without T16 storage (VN_T16 / VN_BF16_H16 = 0): no single-MFMA 16-bit kernels"); return VIPNERF_E_UNSUPPORTED; }
        return launch_mlp_fwd_pt2(a, precision, st);
    }
    return launch_mlp_fwd_bf16n(a, precision, st);
}

static int check_rays(const vipnerf_config *cfg, const vipnerf_rays *r) {
    if (!r) { set_error("rays is NULL"); return VIPNERF_E_ARG; }
    if (r->n_rays < 0) { set_error("n_rays < 0"); return VIPNERF_E_ARG; }
    if (r->n_rays == 0) return VIPNERF_OK;
    if (!r->rays_o || !r->rays_d || !r->rays_o_s || !r->rays_d_s || !r->view_dirs || !r->near || !r->far) {
        set_error("a required ray pointer is NULL"); return VIPNERF_E_ARG; }
    if (cfg->n_sec > 0 && !r->rays_o2) { set_error("n_sec > 0 but rays_o2 is NULL"); return VIPNERF_E_ARG; }
    return VIPNERF_OK;
}

static int check_level(const vipnerf_config *cfg, const vipnerf_level_out *l, const char *name) {
    if (!l->z_vals || !l->raw_sigma || !l->raw_rgb || !l->raw_vis || !l->alpha || !l->visibility || !l->weights ||
        !l->rgb || !l->acc || !l->depth || !l->depth_var || (cfg->n_sec > 0 && (!l->raw_vis2 || !l->vis2))) {
        set_error("a required output pointer of level '%s' is NULL", name); return VIPNERF_E_ARG; }
    return VIPNERF_OK;
}

static PointSrc ray_points(const vipnerf_config *cfg, const vipnerf_rays *r, int S, const float *z) {
    PointSrc s;
    memset(&s, 0, sizeof(s));
    s.P = r->n_rays * S; s.S = S; s.V = cfg->n_sec; s.rays_mode = 1; s.ndc = cfg->ndc;
    s.z = z; s.rays_o = r->rays_o; s.rays_d = r->rays_d; s.rays_o_s = r->rays_o_s; s.rays_d_s = r->rays_d_s;
    s.view_dirs = r->view_dirs; s.rays_o2 = r->rays_o2;
    return s;
}

int launch_secondary_dirs(const PointSrc &s, float *out, hipStream_t st);
int launch_philox(int64_t n, const uint32_t *ctr, const uint32_t *key, uint32_t *out, hipStream_t st);
int launch_rng_draw(int kind, uint64_t seed, uint64_t offset, uint32_t stream, uint64_t first, int64_t n, float *out, hipStream_t st);

struct LevelWs { size_t acts_off, bwd_off; };   // float offsets of the fine level inside the workspaces

static size_t bwd_total(size_t P, int V, bool h16, bool t16) { return bwd_layout(P, V, h16, t16).total; }

}  // namespace vn

using namespace vn;

extern "C" {

int32_t vipnerf_abi_version(void) { return VIPNERF_ABI_VERSION; }

// How this library was built: every build-time switch with its value (vipnerf_knobs.h).  "VN_EXP=unset" for a product build.
const char *vipnerf_build_info(void) {
    return "libvipnerf_hip abi=" VN_KNOB_STR(VIPNERF_ABI_VERSION) " arch=gfx950 " VN_BUILD_INFO_KNOBS;
}
int32_t vipnerf_build_is_experiment(void) { return VN_EXP_VALUE >= 0 ? 1 : 0; }

int32_t vipnerf_last_error(char *buf, size_t n) {
    if (!buf || n == 0) return VIPNERF_E_ARG;
    strncpy(buf, g_err, n - 1);
    buf[n - 1] = 0;
    return VIPNERF_OK;
}

static int check_pack_args(const vipnerf_mlp_params *params, void *packed) {
    clear_stale_hip_error();
    if (!params || !packed) { set_error("pack_weights: NULL argument"); return VIPNERF_E_ARG; }
    for (int i = 0; i < VIPNERF_N_PARAMS; ++i)
        if (!params->p[i]) { set_error("pack_weights: parameter %d is NULL", i); return VIPNERF_E_ARG; }
    return VIPNERF_OK;
}

size_t vipnerf_packed_weights_bytes_p(int32_t precision) {
    return (precision < 0 || precision > VIPNERF_PREC_BF16 || precision_retired(precision)) ? 0 : packed_floats_all(precision) * sizeof(float);
}

// The unsuffixed pair is the FP32 case of the _p pair.
size_t vipnerf_packed_weights_bytes(void) { return vipnerf_packed_weights_bytes_p(VIPNERF_PREC_FP32); }

int32_t vipnerf_pack_weights(const vipnerf_mlp_params *params, void *packed, vipnerf_stream_t stream) {
    return vipnerf_pack_weights_p(params, VIPNERF_PREC_FP32, packed, stream);
}

int32_t vipnerf_pack_weights_p(const vipnerf_mlp_params *params, int32_t precision, void *packed, vipnerf_stream_t stream) {
    if (precision < 0 || precision > VIPNERF_PREC_BF16) { set_error("pack_weights: precision=%d unsupported", precision); return VIPNERF_E_UNSUPPORTED; }
    if (precision_retired(precision)) { set_error("pack_weights: precision=%d (split bf16) was retired with ABI 5", precision); return VIPNERF_E_UNSUPPORTED; }
    int rc = check_pack_args(params, packed);
    if (rc) return rc;
    return launch_pack_bf16n(params, precision, (float *)packed, (hipStream_t)stream);
}

size_t vipnerf_packed_weights_bytes_c(const vipnerf_config *cfg) {
    if (check_cfg(cfg)) return 0;
    return cfg_generic(cfg) ? gen_params(cfg_topo(cfg)).total * sizeof(float) : vipnerf_packed_weights_bytes_p(cfg->precision);
}

int32_t vipnerf_pack_weights_c(const vipnerf_config *cfg, const vipnerf_mlp_params *params, void *packed, vipnerf_stream_t stream) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!cfg_generic(cfg)) return vipnerf_pack_weights_p(params, cfg->precision, packed, stream);
    if (!params || !packed) { set_error("pack_weights: NULL argument"); return VIPNERF_E_ARG; }
    return launch_gen_pack(cfg_topo(cfg), params, (float *)packed, (hipStream_t)stream);
}

int32_t vipnerf_pack_weights2_c(const vipnerf_config *cfg, const vipnerf_mlp_params *params_a, void *packed_a, const vipnerf_mlp_params *params_b,
                                void *packed_b, vipnerf_stream_t stream) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (cfg_generic(cfg) || precision_retired(cfg->precision)) {          // generic topologies: one launch per MLP as before
        if ((rc = vipnerf_pack_weights_c(cfg, params_a, packed_a, stream))) return rc;
        return vipnerf_pack_weights_c(cfg, params_b, packed_b, stream);
    }
    if ((rc = check_pack_args(params_a, packed_a)) || (rc = check_pack_args(params_b, packed_b))) return rc;
    return launch_pack_bf16n(params_a, cfg->precision, packed_a, (hipStream_t)stream, params_b, packed_b);
}

int32_t vipnerf_query_workspace(const vipnerf_config *cfg, int64_t n_rays, size_t *acts_bytes, size_t *bwd_bytes) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (n_rays < 0) { set_error("n_rays < 0"); return VIPNERF_E_ARG; }
    const size_t Pc = (size_t)n_rays * cfg->n_coarse;
    const size_t Pf = cfg->n_fine > 0 ? (size_t)n_rays * (cfg->n_coarse + cfg->n_fine) : 0;
    if (cfg_generic(cfg)) {
        const GenTopo t = cfg_topo(cfg);
        if (acts_bytes) *acts_bytes = (gen_acts(Pc, cfg->n_sec, t).total + (Pf ? gen_acts(Pf, cfg->n_sec, t).total : 0)) * sizeof(float);
        if (bwd_bytes) {
            const size_t a = gen_bwd(Pc, cfg->n_sec, t).total, b = Pf ? gen_bwd(Pf, cfg->n_sec, t).total : 0;
            *bwd_bytes = (a > b ? a : b) * sizeof(float);
        }
        return VIPNERF_OK;
    }
    if (acts_bytes)
        *acts_bytes = cfg->save_acts ? (act_layout(Pc, cfg->n_sec, stores_t16(cfg->precision)).total + (Pf ? act_layout(Pf, cfg->n_sec, stores_t16(cfg->precision)).total : 0)) * sizeof(float) : 0;
    if (bwd_bytes) {
        const bool h16 = stores_high16(cfg->precision);       // the modes with an fp32 copy of dY_5 in the workspace
        const bool t16 = stores_t16(cfg->precision);
        const size_t a = bwd_total(Pc, cfg->n_sec, h16, t16), b = Pf ? bwd_total(Pf, cfg->n_sec, h16, t16) : 0;
        *bwd_bytes = (a > b ? a : b) * sizeof(float);       // levels run one after the other
    }
    return VIPNERF_OK;
}

int32_t vipnerf_coarse_depths(int64_t n_rays, int32_t n_samples, int32_t lindisp, const float *near,
                              const float *far, const float *t_rand, float *z_out, vipnerf_stream_t stream) {
    clear_stale_hip_error();
    if (!near || !far || !z_out) { set_error("coarse_depths: NULL argument"); return VIPNERF_E_ARG; }
    if (n_samples < 2) { set_error("coarse_depths: n_samples < 2"); return VIPNERF_E_UNSUPPORTED; }
    return launch_coarse_z(n_rays, n_samples, lindisp, near, far, t_rand, 0, 0, 0, 0, nullptr, z_out, (hipStream_t)stream);
}

int32_t vipnerf_sample_fine(int64_t n_rays, int32_t n_coarse, int32_t n_fine, const float *z_coarse,
                            const float *weights_coarse, const float *u, float *z_fine, int32_t *inds,
                            float *z_samples, vipnerf_stream_t stream) {
    clear_stale_hip_error();
    if (!z_coarse || !weights_coarse || !z_fine) { set_error("sample_fine: NULL argument"); return VIPNERF_E_ARG; }
    if (n_coarse < 3 || n_fine < 1 || n_coarse + n_fine > 1024) { set_error("sample_fine: unsupported sizes"); return VIPNERF_E_UNSUPPORTED; }
    SampleArgs a;
    memset(&a, 0, sizeof(a));
    a.N = n_rays; a.Sc = n_coarse; a.Sf = n_fine; a.z_coarse = z_coarse; a.w_coarse = weights_coarse; a.u = u;
    a.z_fine = z_fine; a.inds = inds; a.z_samples = z_samples;
    return launch_sample_fine(a, (hipStream_t)stream);
}

int32_t vipnerf_mlp_forward(int64_t n_points, int32_t n_sec, const float *pts, const float *view_dirs,
                            const float *view_dirs2, const float *noise, float noise_std, const void *packed,
                            float *sigma, float *rgb, float *vis, float *vis2, vipnerf_stream_t stream) {
    // the FP32 case of vipnerf_mlp_forward_p (`packed` from vipnerf_pack_weights)
    return vipnerf_mlp_forward_p(n_points, n_sec, pts, view_dirs, view_dirs2, noise, noise_std, VIPNERF_PREC_FP32, packed, sigma, rgb, vis, vis2, stream);
}

int32_t vipnerf_mlp_forward_p(int64_t n_points, int32_t n_sec, const float *pts, const float *view_dirs,
                              const float *view_dirs2, const float *noise, float noise_std, int32_t precision,
                              const void *packed, float *sigma, float *rgb, float *vis, float *vis2,
                              vipnerf_stream_t stream) {
    clear_stale_hip_error();
    if (precision < 0 || precision > VIPNERF_PREC_BF16) { set_error("mlp_forward: precision=%d unsupported", precision); return VIPNERF_E_UNSUPPORTED; }
    if (n_points == 0) return VIPNERF_OK;
    if (!pts || !view_dirs || !packed || !sigma || !rgb || !vis || (n_sec > 0 && (!view_dirs2 || !vis2))) {
        set_error("mlp_forward: NULL argument"); return VIPNERF_E_ARG; }
    if (n_sec < 0 || n_sec > VIPNERF_MAX_SEC) { set_error("mlp_forward: n_sec=%d unsupported", n_sec); return VIPNERF_E_UNSUPPORTED; }
    MlpFwdArgs a;
    memset(&a, 0, sizeof(a));
    a.src.P = n_points; a.src.S = 1; a.src.V = n_sec; a.src.rays_mode = 0;
    a.src.pts = pts; a.src.dirs = view_dirs; a.src.dirs2 = view_dirs2;
    a.ns.noise = noise; a.ns.std = noise_std;
    a.packed = (const float *)packed;
    a.sigma = sigma; a.rgb = rgb; a.vis = vis; a.vis2 = vis2;
    return launch_mlp_fwd_any(a, precision, (hipStream_t)stream);
}

int32_t vipnerf_composite(const vipnerf_config *cfg, const vipnerf_rays *rays, int32_t n_samples,
                          const vipnerf_level_out *lvl, vipnerf_stream_t stream) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!rays || !lvl) { set_error("composite: NULL argument"); return VIPNERF_E_ARG; }
    if ((rc = check_level(cfg, lvl, "composite"))) return rc;
    CompositeArgs a;
    memset(&a, 0, sizeof(a));
    a.N = rays->n_rays; a.S = n_samples; a.V = cfg->n_sec; a.ndc = cfg->ndc; a.white_bkgd = cfg->white_bkgd;
    a.rays_o = rays->rays_o; a.rays_d = rays->rays_d; a.rays_d_s = rays->rays_d_s; a.lvl = *lvl;
    return launch_composite(a, (hipStream_t)stream);
}

static int32_t render_forward_impl(const vipnerf_config *cfg, const vipnerf_rays *rays, const vipnerf_rng *rng,
                                   const void *packed_coarse, const void *packed_fine,
                                   const vipnerf_outputs *out, void *acts, vipnerf_stream_t stream, const SecOriginArgs *so);

int32_t vipnerf_render_forward(const vipnerf_config *cfg, const vipnerf_rays *rays, const vipnerf_rng *rng,
                               const void *packed_coarse, const void *packed_fine,
                               const vipnerf_outputs *out, void *acts, vipnerf_stream_t stream) {
    return render_forward_impl(cfg, rays, rng, packed_coarse, packed_fine, out, acts, stream, nullptr);
}

static int32_t render_forward_impl(const vipnerf_config *cfg, const vipnerf_rays *rays, const vipnerf_rng *rng,
                                   const void *packed_coarse, const void *packed_fine,
                                   const vipnerf_outputs *out, void *acts, vipnerf_stream_t stream, const SecOriginArgs *so) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if ((rc = check_rays(cfg, rays))) return rc;
    if (!out || !packed_coarse || (cfg->n_fine > 0 && !packed_fine)) { set_error("render_forward: NULL argument"); return VIPNERF_E_ARG; }
    if ((rc = check_level(cfg, &out->coarse, "coarse"))) return rc;
    if (cfg->n_fine > 0 && (rc = check_level(cfg, &out->fine, "fine"))) return rc;
    if (cfg->save_acts && !acts) { set_error("render_forward: save_acts set but acts is NULL"); return VIPNERF_E_ARG; }
    const bool generic = cfg_generic(cfg);
    if (generic && !acts) { set_error("render_forward: the generic-topology kernels need the `acts` workspace (vipnerf_query_workspace)"); return VIPNERF_E_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const int64_t N = rays->n_rays;
    if (N == 0) return VIPNERF_OK;
    const int Sc = cfg->n_coarse, Sf = cfg->n_fine, V = cfg->n_sec;
    const bool train = cfg->train != 0, perturb = cfg->perturb != 0;
    const uint64_t seed = rng ? rng->seed : 0, offset = rng ? rng->offset : 0, ray_base = rng ? rng->ray_base : 0;
    const int64_t *ray_ids = rng ? rng->ray_ids : nullptr;

    // 1. coarse depths
    const float *t_rand = (perturb && rng) ? rng->t_rand : nullptr;
    {
        ProfScope ps("coarse_z", st);
        rc = launch_coarse_z(N, Sc, cfg->lindisp, rays->near, rays->far, t_rand, perturb && !t_rand, seed, offset, ray_base, ray_ids,
                             out->coarse.z_vals, st, so);
    }
    if (rc) return rc;

    for (int lv = 0; lv < (Sf > 0 ? 2 : 1); ++lv) {
        const vipnerf_level_out &L = lv ? out->fine : out->coarse;
        const int S = lv ? Sc + Sf : Sc;
        // (4. importance sampling + sorted merge: in the coarse level's compositing launch, below)
        // 2./5. MLP
        MlpFwdArgs ma;
        memset(&ma, 0, sizeof(ma));
        ma.src = ray_points(cfg, rays, S, L.z_vals);
        if (train && cfg->noise_std > 0.f) {
            ma.ns.noise = rng ? (lv ? rng->noise_fine : rng->noise_coarse) : nullptr;
            ma.ns.device_rng = !ma.ns.noise;
            ma.ns.std = cfg->noise_std;
            ma.ns.stream = lv ? RS_NOISE_F : RS_NOISE_C;
            ma.ns.seed = seed; ma.ns.offset = offset; ma.ns.idx_base = ray_base * (uint64_t)S; ma.ns.ray_ids = ray_ids;
        }
        ma.packed = (const float *)(lv ? packed_fine : packed_coarse);
        ma.sigma = L.raw_sigma; ma.rgb = L.raw_rgb; ma.vis = L.raw_vis; ma.vis2 = L.raw_vis2;
        if (cfg->save_acts) {
            const size_t Pc = (size_t)N * Sc;
            ma.al = act_layout((size_t)N * S, V, stores_t16(cfg->precision));
            ma.acts = (float *)acts + (lv ? act_layout(Pc, V, stores_t16(cfg->precision)).total : 0);
        }
        if (generic) {
            const GenTopo t = cfg_topo(cfg);
            float *ga = (float *)acts + (lv ? gen_acts((size_t)N * Sc, V, t).total : 0);
            ProfScope ps(lv ? "mlp_fwd_fine" : "mlp_fwd_coarse", st);
            if ((rc = launch_gen_fwd(t, ma.src, ma.ns, ma.packed, ma.sigma, ma.rgb, ma.vis, ma.vis2, ga, st))) return rc;
        } else {
            ProfScope ps(lv ? "mlp_fwd_fine" : "mlp_fwd_coarse", st);
            if ((rc = launch_mlp_fwd_any(ma, cfg->precision, st))) return rc;
        }
        // 3./6. compositing
        CompositeArgs ca;
        memset(&ca, 0, sizeof(ca));
        ca.N = N; ca.S = S; ca.V = V; ca.ndc = cfg->ndc; ca.white_bkgd = cfg->white_bkgd;
        ca.rays_o = rays->rays_o; ca.rays_d = rays->rays_d; ca.rays_d_s = rays->rays_d_s; ca.lvl = L;
        if (lv == 0 && Sf > 0 && !cfg->given_z_fine) {
            // 3. + 4. the coarse level's compositing AND the importance sampling + sorted merge its weights feed, one launch
            SampleArgs sa;
            memset(&sa, 0, sizeof(sa));
            sa.N = N; sa.Sc = Sc; sa.Sf = Sf; sa.z_coarse = out->coarse.z_vals; sa.w_coarse = out->coarse.weights;
            sa.u = (perturb && rng) ? rng->u : nullptr;
            sa.device_rng = perturb && !sa.u; sa.seed = seed; sa.offset = offset; sa.ray_base = ray_base; sa.ray_ids = ray_ids;
            sa.z_fine = out->fine.z_vals; sa.inds = out->sample_inds; sa.z_samples = out->z_samples;
            ProfScope ps("composite", st);
            if ((rc = launch_composite_sample(ca, sa, st))) return rc;
            continue;
        }
        ProfScope ps("composite", st);
        if ((rc = launch_composite(ca, st))) return rc;
    }
    return VIPNERF_OK;
}

int32_t vipnerf_render_backward(const vipnerf_config *cfg, const vipnerf_rays *rays,
                                const void *packed_coarse, const void *packed_fine,
                                const vipnerf_outputs *out, const vipnerf_out_grads *gout,
                                const void *acts, void *bwd_ws,
                                const vipnerf_mlp_grads *grads_coarse, const vipnerf_mlp_grads *grads_fine,
                                vipnerf_stream_t stream) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if ((rc = check_rays(cfg, rays))) return rc;
    if (!out || !gout || !acts || !bwd_ws || !packed_coarse || !grads_coarse || (cfg->n_fine > 0 && (!packed_fine || !grads_fine))) {
        set_error("render_backward: NULL argument"); return VIPNERF_E_ARG; }
    hipStream_t st = (hipStream_t)stream;
    const int64_t N = rays->n_rays;
    const int Sc = cfg->n_coarse, Sf = cfg->n_fine, V = cfg->n_sec;
    for (int lv = (Sf > 0 ? 1 : 0); lv >= 0; --lv) {
        const vipnerf_level_out &L = lv ? out->fine : out->coarse;
        const vipnerf_mlp_grads *G = lv ? grads_fine : grads_coarse;
        for (int i = 0; i < VIPNERF_N_PARAMS; ++i)
            if (!G->g[i] && gen_param_numel(cfg_topo(cfg), i)) { set_error("render_backward: grad pointer %d is NULL", i); return VIPNERF_E_ARG; }
        if (N == 0) {
            continue;
        }
        const int S = lv ? Sc + Sf : Sc;
        const size_t P = (size_t)N * S;
        float *bw = (float *)bwd_ws;
        if (cfg_generic(cfg)) {
            const GenTopo t = cfg_topo(cfg);
            const GenBwd gb = gen_bwd(P, V, t);
            CompositeBwdArgs cb;
            memset(&cb, 0, sizeof(cb));
            cb.N = N; cb.S = S; cb.V = V; cb.ndc = cfg->ndc; cb.white_bkgd = cfg->white_bkgd;
            cb.rays_o = rays->rays_o; cb.rays_d = rays->rays_d; cb.rays_d_s = rays->rays_d_s; cb.lvl = L;
            cb.g = lv ? gout->fine : gout->coarse;
            cb.dsig = bw + gb.dsig; cb.drgb = bw + gb.drgb; cb.dvis = bw + gb.dvis; cb.dvis2 = bw + gb.dvis2;
            {
                ProfScope ps("composite_bwd", st);
                if ((rc = launch_composite_bwd(cb, st))) return rc;
            }
            const PointSrc src = ray_points(cfg, rays, S, L.z_vals);
            const float *ga = (const float *)acts + (lv ? gen_acts((size_t)N * Sc, V, t).total : 0);
            ProfScope ps(lv ? "mlp_bwd_generic_fine" : "mlp_bwd_generic_coarse", st);
            if ((rc = launch_gen_bwd(t, src, (const float *)(lv ? packed_fine : packed_coarse), L.raw_sigma, L.raw_rgb, ga, bw, gb, G, st))) return rc;
            continue;
        }
        const BwdLayout bl = bwd_layout(P, V, stores_high16(cfg->precision), stores_t16(cfg->precision));
        // 1. compositing backward -> dLoss/d(raw network outputs)
        CompositeBwdArgs cb;
        memset(&cb, 0, sizeof(cb));
        cb.N = N; cb.S = S; cb.V = V; cb.ndc = cfg->ndc; cb.white_bkgd = cfg->white_bkgd;
        cb.rays_o = rays->rays_o; cb.rays_d = rays->rays_d; cb.rays_d_s = rays->rays_d_s; cb.lvl = L;
        cb.g = lv ? gout->fine : gout->coarse;
        cb.dsig = bw + bl.dsig; cb.drgb = bw + bl.drgb; cb.dvis = bw + bl.dvis; cb.dvis2 = bw + bl.dvis2;
        {
            ProfScope ps("composite_bwd", st);
            if ((rc = launch_composite_bwd(cb, st))) return rc;
        }
        // 2. MLP data gradients (register-chained, transposed weights)
        MlpBwdArgs mb;
        memset(&mb, 0, sizeof(mb));
        mb.src = ray_points(cfg, rays, S, L.z_vals);
        mb.packed = (const float *)(lv ? packed_fine : packed_coarse);
        mb.sigma = L.raw_sigma; mb.rgb = L.raw_rgb; mb.vis = L.raw_vis; mb.vis2 = L.raw_vis2;
        mb.al = act_layout(P, V, stores_t16(cfg->precision));
        mb.acts = (const float *)acts + (lv ? act_layout((size_t)N * Sc, V, stores_t16(cfg->precision)).total : 0);
        mb.bwd = bw; mb.bl = bl;
        {
            ProfScope ps(lv ? "mlp_dgrad_fine" : "mlp_dgrad_coarse", st);
            if ((rc = launch_mlp_bwd_bf16n(mb, cfg->precision, st))) return rc;
        }
        // 3. weight gradients: dW = dY^T H as MFMA GEMMs over the point axis
#if defined(VN_EXP)
        {   // experiment builds only (tools/ablation_pt2.py): the data-gradient kernels alone, e.g. under rocm-smi
            static const bool skip = [] { const char *e = getenv("VIPNERF_EXP_SKIP_WGRAD"); return e && *e == '1'; }();
            if (skip) continue;
        }
#endif
        if ((rc = launch_wgrad(P, V, mb.acts, mb.al, bw, bl, G, cfg->precision, st,
                               (cfg->precision >= VIPNERF_PREC_FP16X3 && cfg->precision <= VIPNERF_PREC_FP16) ? (const unsigned *)(bw + bl.gmax) : nullptr))) return rc;
    }
    return VIPNERF_OK;
}

// defer: the final sums are NOT launched; *defer receives the arguments for whoever carries them (vipnerf_train_step: its seeds x weights launch)
static int32_t losses_forward_impl(const vipnerf_config *cfg, int64_t n_rays, const vipnerf_loss_in *in, const vipnerf_outputs *out,
                                   const vipnerf_loss_out *lout, const float *weights, float *total, float *named, vipnerf_stream_t stream,
                                   LossArgs *defer = nullptr) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!in || !out || !lout || !in->target_rgb || !lout->loss_values || !lout->scratch) {
        set_error("losses_forward: NULL argument"); return VIPNERF_E_ARG; }
    if (in->mask_sparse && !in->sparse_depth) { set_error("losses_forward: mask_sparse without sparse_depth"); return VIPNERF_E_ARG; }
    LossArgs a;
    memset(&a, 0, sizeof(a));
    a.N = n_rays; a.V = cfg->n_sec; a.n_levels = cfg->n_fine > 0 ? 2 : 1;
    a.S_coarse = cfg->n_coarse; a.S_fine = cfg->n_coarse + cfg->n_fine;
    a.in = *in; a.coarse = out->coarse; a.fine = out->fine;
    a.seeds_coarse = lout->coarse; a.seeds_fine = lout->fine;
    for (int lv = 0; lv < a.n_levels; ++lv) {
        const vipnerf_loss_level_seeds &s = lv ? a.seeds_fine : a.seeds_coarse;
        if (!s.rgb || !s.visibility || !s.raw_vis || (a.V > 0 && !s.vis2)) {
            set_error("losses_forward: a seed pointer is NULL"); return VIPNERF_E_ARG; }
    }
    a.partial = lout->scratch; a.counts = lout->scratch + 7 * (size_t)n_rays; a.loss_values = lout->loss_values;
    if (weights) {
        for (int k = 0; k < 8; ++k) a.w[k] = weights[k];
        a.total = total; a.named = named;
    }
    if (defer) *defer = a;
    ProfScope ps("losses", (hipStream_t)stream);
    return launch_losses(a, (hipStream_t)stream, defer != nullptr);
}

int32_t vipnerf_losses_forward(const vipnerf_config *cfg, int64_t n_rays, const vipnerf_loss_in *in,
                               const vipnerf_outputs *out, const vipnerf_loss_out *lout,
                               vipnerf_stream_t stream) {
    return losses_forward_impl(cfg, n_rays, in, out, lout, nullptr, nullptr, nullptr, stream);
}

int32_t vipnerf_losses_forward_w(const vipnerf_config *cfg, int64_t n_rays, const vipnerf_loss_in *in,
                                 const vipnerf_outputs *out, const vipnerf_loss_out *lout, const float *weights,
                                 float *total, float *named, vipnerf_stream_t stream) {
    if (!weights || !total) { set_error("losses_forward_w: weights (host, 8) and total (device, 1) are needed"); return VIPNERF_E_ARG; }
    if (n_rays <= 0) { set_error("losses_forward_w: no rays (TotalLoss of an empty batch is not defined)"); return VIPNERF_E_ARG; }
    return losses_forward_impl(cfg, n_rays, in, out, lout, weights, total, named, stream);
}

static int32_t scale_segments_impl(int32_t n_segs, const vipnerf_scale_seg *segs, const float *g, const float *g_total, const float *weights,
                                   vipnerf_stream_t stream) {
    clear_stale_hip_error();
    if (n_segs < 0 || n_segs > VIPNERF_MAX_SCALE_SEGS) { set_error("scale_segments: n_segs=%d (0..%d)", n_segs, VIPNERF_MAX_SCALE_SEGS); return VIPNERF_E_ARG; }
    if (n_segs == 0) return VIPNERF_OK;
    if (!segs || (!g && !(g_total && weights))) { set_error("scale_segments: NULL argument"); return VIPNERF_E_ARG; }
    ScaleArgs a;
    memset(&a, 0, sizeof(a));
    a.n = n_segs; a.g = g; a.g1 = g ? nullptr : g_total;
    if (!g) for (int k = 0; k < 8; ++k) a.w[k] = weights[k];
    for (int k = 0; k < n_segs; ++k) {
        if (segs[k].numel < 0 || segs[k].slot < 0 || segs[k].slot > 7 || (segs[k].numel > 0 && (!segs[k].in || !segs[k].out))) {
            set_error("scale_segments: segment %d: numel %lld, slot %d, NULL pointers?", k, (long long)segs[k].numel, segs[k].slot); return VIPNERF_E_ARG; }
        a.s[k] = segs[k];
    }
    ProfScope ps("losses_bwd", (hipStream_t)stream);
    return launch_scale_segments(a, (hipStream_t)stream);
}

int32_t vipnerf_scale_segments(int32_t n_segs, const vipnerf_scale_seg *segs, const float *g, vipnerf_stream_t stream) {
    return scale_segments_impl(n_segs, segs, g, nullptr, nullptr, stream);
}

int32_t vipnerf_scale_segments_w(int32_t n_segs, const vipnerf_scale_seg *segs, const float *g_total, const float *weights,
                                 vipnerf_stream_t stream) {
    return scale_segments_impl(n_segs, segs, nullptr, g_total, weights, stream);
}

// build switch VN_ADAM_FMA_MASK (default 7, vipnerf_knobs.h): which of torch's three update expressions its kernels contract into an fma on gfx950 (tests/test_hip_fullsize.py)
int32_t vipnerf_adam_step(int64_t n, float *param, float *exp_avg, float *exp_avg_sq, const float *grad, float lerp_w, float beta2, float sq_w,
                          float inv_sqrt_bc2, float eps, float neg_step, int32_t fma_mask, vipnerf_stream_t stream) {
    clear_stale_hip_error();
    if (n < 0) { set_error("adam_step: n < 0"); return VIPNERF_E_ARG; }
    if (n == 0) return VIPNERF_OK;
    if (!param || !exp_avg || !exp_avg_sq || !grad) { set_error("adam_step: NULL argument"); return VIPNERF_E_ARG; }
    if (fma_mask < -1 || fma_mask > 7) { set_error("adam_step: fma_mask=%d (-1 or 0..7)", fma_mask); return VIPNERF_E_ARG; }
    ProfScope ps("adam", (hipStream_t)stream);
    return launch_adam_step(n, param, exp_avg, exp_avg_sq, grad, lerp_w, beta2, sq_w, inv_sqrt_bc2, eps, neg_step,
                            fma_mask < 0 ? VN_ADAM_FMA_MASK : fma_mask, (hipStream_t)stream);
}

int32_t vipnerf_train_step(const vipnerf_train_step_args *t, vipnerf_stream_t stream) {
    clear_stale_hip_error();
    if (!t || !t->cfg || !t->rays || !t->loss_in || !t->params_coarse || !t->packed_coarse || !t->out || !t->lout || !t->acts || !t->bwd_ws ||
        !t->grads_coarse) { set_error("train_step: NULL argument"); return VIPNERF_E_ARG; }
    const vipnerf_config *cfg = t->cfg;
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if (!cfg->train || !cfg->save_acts) { set_error("train_step: cfg.train and cfg.save_acts must be set"); return VIPNERF_E_ARG; }
    const bool two = cfg->n_fine > 0;
    if (two && (!t->params_fine || !t->packed_fine || !t->grads_fine)) { set_error("train_step: n_fine > 0 but a fine-level argument is NULL"); return VIPNERF_E_ARG; }
    if (t->adam_n < 0 || (t->adam_n > 0 && (!t->adam_param || !t->adam_exp_avg || !t->adam_exp_avg_sq || !t->adam_grad))) {
        set_error("train_step: bad Adam arguments"); return VIPNERF_E_ARG; }
    const int64_t N = t->rays->n_rays;
    // 0. the other cameras' centres of every row (VipNeRF01.py:84-98)
    if (t->poses) {
        if (!t->pixel_id || !t->rays_o2_out || t->rays_o2_out != t->rays->rays_o2) {
            set_error("train_step: poses given, so pixel_id and rays_o2_out (== rays->rays_o2) are needed"); return VIPNERF_E_ARG; }
        if (t->n_frames < 1 || t->n_frames > 1 + VIPNERF_MAX_SEC) { set_error("train_step: n_frames=%d", t->n_frames); return VIPNERF_E_ARG; }
    }
    // (the centres are written by the step's first launch, k_coarse_z: no launch of their own)
    SecOriginArgs so;
    memset(&so, 0, sizeof(so));
    if (t->poses && t->n_frames > 1 && N > 0) { so.poses = t->poses; so.pixel_id = t->pixel_id; so.idx64 = t->pixel_id_is_int64; so.nf = t->n_frames; so.rays_o2 = t->rays_o2_out; }
    // 1. this iteration's weights in fragment order
    if (two) {
        if ((rc = vipnerf_pack_weights2_c(cfg, t->params_coarse, t->packed_coarse, t->params_fine, t->packed_fine, stream))) return rc;
    } else if ((rc = vipnerf_pack_weights_c(cfg, t->params_coarse, t->packed_coarse, stream))) return rc;
    // 2. forward, 3. losses
    if ((rc = render_forward_impl(cfg, t->rays, t->rng, t->packed_coarse, t->packed_fine, t->out, t->acts, stream, so.rays_o2 ? &so : nullptr))) return rc;
    LossArgs la;
    if ((rc = losses_forward_impl(cfg, N, t->loss_in, t->out, t->lout, t->loss_weights, t->total_loss, nullptr, stream, &la))) return rc;
    // 4. d TotalLoss / d outputs = weight of the loss x its unweighted seeds, in place (what autograd does with the five-call path's seeds:
    //    the same segments in the same order through the same kernel), and TotalLoss itself
    const int V = cfg->n_sec, Sc = cfg->n_coarse, Sf = cfg->n_coarse + cfg->n_fine;
    ScaleArgs sa;
    memset(&sa, 0, sizeof(sa));
    for (int k = 0; k < 8; ++k) sa.w[k] = t->loss_weights[k];
    // (TotalLoss and the loss values' final sums: loss_final_body, carried by the same launch -- `la` above; the seeds x weights do not depend on them)
    vipnerf_out_grads og;
    memset(&og, 0, sizeof(og));
    for (int lv = 0; lv < (two ? 2 : 1); ++lv) {
        const vipnerf_loss_level_seeds &sd = lv ? t->lout->fine : t->lout->coarse;
        vipnerf_level_grads &g = lv ? og.fine : og.coarse;
        const int S = lv ? Sf : Sc;
        auto seg = [&](float *p, int64_t numel, int slot) {
            vipnerf_scale_seg &e = sa.s[sa.n++];
            e.in = p; e.out = p; e.numel = numel; e.slot = slot; e.reserved = 0;
        };
        seg(sd.rgb, N * 3, 0 + lv); g.rgb = sd.rgb;
        seg(sd.visibility, N * S, 2 + lv); g.visibility = sd.visibility;
        seg(sd.raw_vis, N * S, 2 + lv); g.raw_vis = sd.raw_vis;
        if (V > 0) { seg(sd.vis2, N * V, 4 + lv); g.vis2 = sd.vis2; }
        if (lv == (two ? 1 : 0) && sd.depth) { seg(sd.depth, N, 6); g.depth = sd.depth; }
    }
    if (N > 0) {
        ProfScope ps("losses_bwd", (hipStream_t)stream);
        if ((rc = launch_scale_segments(sa, (hipStream_t)stream, &la))) return rc;
    }
    // 5. backward
    if ((rc = vipnerf_render_backward(cfg, t->rays, t->packed_coarse, t->packed_fine, t->out, &og, t->acts, t->bwd_ws, t->grads_coarse,
                                      t->grads_fine, stream))) return rc;
    // 6. optimizer
    if (t->adam_n > 0)
        return vipnerf_adam_step(t->adam_n, t->adam_param, t->adam_exp_avg, t->adam_exp_avg_sq, t->adam_grad, t->lerp_w, t->beta2, t->sq_w,
                                 t->inv_sqrt_bc2, t->eps, t->neg_step, t->fma_mask, stream);
    return VIPNERF_OK;
}

int32_t vipnerf_generate_rays(const vipnerf_raygen *gen, int64_t n_rays, const vipnerf_ray_batch *out,
                              vipnerf_stream_t stream) {
    clear_stale_hip_error();
    if (!gen || !out || !gen->cameras || !out->rays_o || !out->rays_d) { set_error("generate_rays: NULL argument"); return VIPNERF_E_ARG; }
    if (gen->height <= 0 || gen->width <= 0 || gen->n_frames <= 0 || n_rays < 0) { set_error("generate_rays: bad sizes"); return VIPNERF_E_ARG; }
    if (gen->ndc && (!out->rays_o_ndc || !out->rays_d_ndc)) { set_error("generate_rays: ndc set but NDC outputs are NULL"); return VIPNERF_E_ARG; }
    RayGenArgs a;
    a.g = *gen; a.out = *out; a.N = n_rays;
    ProfScope ps("gen_rays", (hipStream_t)stream);
    return launch_gen_rays(a, (hipStream_t)stream);
}

int32_t vipnerf_postprocess_frame(int64_t n_pixels, const float *rgb, const float *depth, const float *depth_var,
                                  const float *depth_ndc, const float *depth_var_ndc, uint8_t *image,
                                  float *o_depth, float *o_depth_var, float *o_depth_ndc, float *o_depth_var_ndc,
                                  vipnerf_stream_t stream) {
    clear_stale_hip_error();
    if (n_pixels < 0) { set_error("postprocess_frame: n_pixels < 0"); return VIPNERF_E_ARG; }
    ProfScope ps("postprocess", (hipStream_t)stream);
    return launch_postprocess(n_pixels, rgb, depth, depth_var, depth_ndc, depth_var_ndc, image, o_depth, o_depth_var,
                              o_depth_ndc, o_depth_var_ndc, (hipStream_t)stream);
}

int32_t vipnerf_visibility_prior(const vipnerf_psv *psv, double *weights64, float *weights32, uint8_t *mask,
                                 vipnerf_stream_t stream) {
    clear_stale_hip_error();
    if (!psv || !psv->planes || !psv->frame1 || !psv->frame2) { set_error("visibility_prior: NULL argument"); return VIPNERF_E_ARG; }
    if (psv->height <= 0 || psv->width <= 0 || psv->n_planes <= 0 || !(psv->temperature > 0)) { set_error("visibility_prior: bad sizes"); return VIPNERF_E_ARG; }
    ProfScope ps("visibility_prior", (hipStream_t)stream);
    return launch_psv(psv, weights64, weights32, mask, (hipStream_t)stream);
}

int32_t vipnerf_secondary_dirs(const vipnerf_config *cfg, const vipnerf_rays *rays, int32_t n_samples, const float *z,
                               float *dirs2, vipnerf_stream_t stream) {
    int rc = check_cfg(cfg);
    if (rc) return rc;
    if ((rc = check_rays(cfg, rays))) return rc;
    if (!z || !dirs2 || n_samples <= 0) { set_error("secondary_dirs: bad argument"); return VIPNERF_E_ARG; }
    if (cfg->n_sec <= 0) { set_error("secondary_dirs: n_sec == 0"); return VIPNERF_E_ARG; }
    const PointSrc s = ray_points(cfg, rays, n_samples, z);
    return launch_secondary_dirs(s, dirs2, (hipStream_t)stream);
}

int32_t vipnerf_secondary_origins(int64_t n_rays, int32_t n_frames, const float *poses, const void *pixel_id, int32_t pixel_id_is_int64,
                                  float *rays_o2, vipnerf_stream_t stream) {
    clear_stale_hip_error();
    if (n_rays < 0 || n_frames < 1 || (n_rays > 0 && n_frames > 1 && (!poses || !pixel_id || !rays_o2))) {
        set_error("secondary_origins: bad argument"); return VIPNERF_E_ARG; }
    ProfScope ps("secondary_origins", (hipStream_t)stream);
    return launch_secondary_origins(n_rays, n_frames, poses, pixel_id, pixel_id_is_int64, rays_o2, (hipStream_t)stream);
}

int32_t vipnerf_philox4x32_10(int64_t n, const uint32_t *counters, const uint32_t *keys, uint32_t *out, vipnerf_stream_t stream) {
    clear_stale_hip_error();
    if (n < 0 || (n > 0 && (!counters || !keys || !out))) { set_error("philox4x32_10: bad argument"); return VIPNERF_E_ARG; }
    return launch_philox(n, counters, keys, out, (hipStream_t)stream);
}

int32_t vipnerf_rng_draw(int32_t kind, uint64_t seed, uint64_t offset, uint32_t stream_id, uint64_t first_idx, int64_t n,
                         float *out, vipnerf_stream_t stream) {
    clear_stale_hip_error();
    if (n < 0 || (n > 0 && !out) || kind < 0 || kind > 1) { set_error("rng_draw: bad argument"); return VIPNERF_E_ARG; }
    return launch_rng_draw(kind, seed, offset, stream_id, first_idx, n, out, (hipStream_t)stream);
}

int32_t vipnerf_profile_enable(int32_t on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return VIPNERF_OK;
}

int32_t vipnerf_profile_read(vipnerf_profile_entry *entries, int32_t max_entries, int32_t *n_out) {
    if (!entries || !n_out || max_entries <= 0) { set_error("profile_read: bad argument"); return VIPNERF_E_ARG; }
    std::lock_guard<std::mutex> lk(g_prof_mu);
    int n = 0;
    for (auto &r : g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            int k = 0;
            for (; k < n; ++k) if (!strcmp(entries[k].name, r.name)) break;
            if (k == n && n < max_entries) { memset(&entries[n], 0, sizeof(entries[n])); strncpy(entries[n].name, r.name, sizeof(entries[n].name) - 1); ++n; }
            if (k < n) { entries[k].count += 1; entries[k].total_ms += ms; }
        }
        g_pool.push_back(r.e0); g_pool.push_back(r.e1);
    }
    g_recs.clear();
    *n_out = n;
    return VIPNERF_OK;
}

}  // extern "C"
