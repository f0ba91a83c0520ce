/*
 * vipnerf_hip.h -- C ABI of libvipnerf_hip.so: the MI355X (gfx950) implementation of ViP-NeRF's per-ray
 * volumetric-rendering hot path.
 *
 * The reference (NagabhushanSN95/ViP-NeRF v1.0) is pure Python/PyTorch and has NO FFI of its own for this
 * path; the boundary it offers is the Python module contract of src/models/VipNeRF01.py and
 * src/loss_functions/ modules.  This header is the C surface a binding for that contract calls into; each entry
 * point cites the reference function(s) it replaces (paths relative to the reference repo root).  The ctypes
 * binding that implements the reference's module contract on top of it lives in
 * vip-nerf_amd/vipnerf_hip/ and vip-nerf_amd/src/{models,loss_functions}/ (see INTEGRATION.md).
 *
 * Conventions
 *   - plain C structs of device pointers and sizes; no torch types.  All tensors are row-major contiguous
 *     fp32 unless noted (int32 sample indices, uint8 masks).
 *   - the library BORROWS every pointer for the duration of the call (asynchronously: until the work queued
 *     on `stream` has run).  It allocates nothing the caller must free; outputs and workspaces are caller
 *     allocated (sizes from vipnerf_query_workspace).
 *   - every function returns 0 on success, <0 on error (VIPNERF_E_*); vipnerf_last_error() gives the text
 *     (thread local).  Nothing throws, nothing synchronises the device.
 *   - re-entrant; all state is in the arguments.  The hand-written MFMA kernels serve ONE MLP topology: 8x256 trunk, skip into
 *     layer 5, positional-encoding degrees 10 (points) / 4 (directions), 128-wide view branch with an
 *     rgb(3)+visibility(1) head -- the only topology any shipped reference config uses.  Every other one
 *     MLP.__init__ can build (vipnerf_config.netdepth / netwidth / pe_degrees / head_variant) runs generic per-layer kernels, fp32.
 */
#ifndef VIPNERF_HIP_H
#define VIPNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIPNERF_ABI_VERSION 6

#define VIPNERF_OK             0
#define VIPNERF_E_ARG         (-1)   /* null / inconsistent argument */
#define VIPNERF_E_UNSUPPORTED (-2)   /* configuration outside the supported topology / sizes */
#define VIPNERF_E_HIP         (-3)   /* a HIP runtime call failed (text in vipnerf_last_error) */

/* MLP GEMM arithmetic.  FP32: v_mfma_f32_16x16x4_f32 / 32x32x2_f32, exact fp32 operands, products and accumulation, bit-equivalent to
 * an fmaf chain (the parity default; BASELINE configs[1], [2], [3]).
 * BF16X3 / BF16X6 (operands split into 2 / 3 bf16 parts): RETIRED with ABI 5 -- the codes stay reserved and every entry point refuses them
 * with VIPNERF_E_UNSUPPORTED (no BASELINE configuration used them; FP16X3 is the fp32-grade fast arithmetic). */
#define VIPNERF_PREC_FP32   0
#define VIPNERF_PREC_BF16X3 1
#define VIPNERF_PREC_BF16X6 2
/* FP16X3: operands split into 2 fp16 parts (11 bits each), 3 cross terms on v_mfma_f32_16x16x32_f16 with fp32
 * accumulation, power-of-two operand scaling (vip-nerf_amd/csrc/vipnerf_bf16n.h): ~2^-21 relative error per product
 * at 3 MFMAs per product. */
#define VIPNERF_PREC_FP16X3 3
/* FP16X3H ("mixed"): FP16X3 in the forward and data-gradient GEMMs (so outputs and losses are the FP16X3 ones), but the
 * 256-wide activations and gradients that only the weight-gradient GEMMs read back are stored as fp16 (2 bytes instead
 * of 4) and those GEMMs run single fp16 MFMAs: 8 GB less HBM traffic per 4096-ray step, parameter gradients at ~2e-4
 * relative error instead of fp32 grade.  A separate accuracy class (BASELINE configs[4] style mixed precision). */
#define VIPNERF_PREC_FP16X3H 4
/* Single-pass 16-bit modes (BASELINE configs[4] style mixed precision): ONE MFMA per product in the forward and
 * data-gradient GEMMs, operands rounded once to fp16 (11-bit significand; power-of-two scaling as in FP16X3) or bf16
 * (8 bits, fp32's range), fp32 accumulation, fp32 master weights / biases / encodings / heads / compositing / losses.
 * Both store EVERY operand of the weight-gradient GEMMs (activations, gradients, encodings, head seeds) as 16-bit values in
 * 16-point x 16-feature tiles inside the `acts` / `bwd_ws` workspaces (vip-nerf_amd/csrc/vipnerf_bf16n.h: store_t16) and run all
 * weight-gradient GEMMs as single 16-bit MFMAs fed by DMA + ds_read_b64_tr_b16 (vipnerf_wgrad16.hip): 26 GB of HBM traffic per
 * 4096-ray step instead of 55.  Training calls in these modes need n_rays * samples of each level to be a multiple of 32 (refused with the reason otherwise).  Errors
 * ~1e-3 (fp16) / ~1e-2 (bf16) relative -- a separate accuracy class from everything above (BASELINE configs[4]). */
#define VIPNERF_PREC_FP16   5
#define VIPNERF_PREC_BF16   6

/* Lane layout.  Every kernel runs the NARROW layout (16-point waves, two per SIMD); WIDE (32-point waves, one per SIMD: the round-1 kernel
 * generation) was RETIRED with ABI 5 and is refused with VIPNERF_E_UNSUPPORTED.  DEFAULT == NARROW. */
#define VIPNERF_LAYOUT_DEFAULT 0
#define VIPNERF_LAYOUT_WIDE    1
#define VIPNERF_LAYOUT_NARROW  2

/* vipnerf_config.head_variant: which outputs the trunk head / the view branch predict (MLP.__init__, VipNeRF01.py:467-491) */
#define VIPNERF_HEAD_RGB_TRUNK     1   /* mlp 'view_dependent_rgb' = False: pts_output_linear is [sigma, rgb(3)], the view branch predicts visibility only */
#define VIPNERF_HEAD_NO_VISIBILITY 2   /* mlp 'predict_visibility' = False: no visibility outputs, n_sec must be 0; with RGB_TRUNK: no view branch at all */

#define VIPNERF_MAX_SEC 3            /* secondary views V = nf-1 <= 3 (reference configs use nf in {2,3,4}) */
#define VIPNERF_N_PARAMS 24          /* tensors of one MLP */

typedef void *vipnerf_stream_t;      /* hipStream_t */

/* Static configuration of one render call.  Mirrors the keys VipNeRF01.py reads from its `configs` dict
 * (src/models/VipNeRF01.py:16-19,53,180-185,206-207,363,470). */
typedef struct vipnerf_config {
    int32_t ndc;          /* configs['data_loader']['ndc']: sample in NDC space, depths converted back */
    int32_t n_coarse;     /* coarse_mlp.num_samples (64); any count 2..256 (>= 3 with n_fine > 0), as in the reference */
    int32_t n_fine;       /* fine_mlp.num_samples (128); 0 = coarse pass only; n_coarse + n_fine <= 256 */
    int32_t n_sec;        /* V: secondary views whose visibility is predicted this call (0 if !sec_views_vis) */
    int32_t train;        /* model.training: sigma noise (if noise_std > 0) */
    int32_t lindisp;      /* configs['model']['lindisp'] */
    int32_t white_bkgd;   /* configs['model']['white_bkgd'] */
    int32_t save_acts;    /* keep layer activations in the `acts` workspace for vipnerf_render_backward */
    float   noise_std;    /* configs['model']['raw_noise_std'] (used when train) */
    int32_t given_z_fine; /* parity tests (teacher forcing): out->fine.z_vals already holds the fine depths on
                             entry; importance sampling is skipped.  0 in production. */
    int32_t perturb;      /* configs['model']['perturb'] && training: stratified jitter + random inverse-CDF draws */
    int32_t precision;    /* VIPNERF_PREC_*: arithmetic of the MLP GEMMs */
    int32_t bf16_layout;  /* VIPNERF_LAYOUT_DEFAULT (0) or VIPNERF_LAYOUT_NARROW: the one lane layout left; kept for struct compatibility */
    /* MLP topology (coarse and fine alike), configs['model']['*_mlp'][netdepth, netwidth, points_/views_positional_encoding_degree]
     * (VipNeRF01.py:458-470).  0 = the default.  The hand-written MFMA kernels are specialised on 8 / 256 / 10 / 4 -- what every
     * shipped reference config uses; any other topology (netdepth <= 8, netwidth <= 256 and a multiple of 8, degrees <= 16 / 8;
     * e.g. BASELINE configs[0]'s 4x64 network) runs the generic per-layer kernels (csrc/vipnerf_generic.hip), fp32 only. */
    int32_t netdepth;     /* 0 -> 8 */
    int32_t netwidth;     /* 0 -> 256 */
    int32_t pe_degrees;   /* points degree | views degree << 8;  0 -> 10 | 4 << 8 */
    int32_t head_variant; /* VIPNERF_HEAD_* bits; 0 = sigma | rgb + visibility (every shipped / BASELINE config).  Any other value runs the
                             generic per-layer kernels (fp32), whatever the trunk's size; the parameter tensors a variant lacks are NULL */
} vipnerf_config;

/* One ray batch (render_rays' input_dict, src/models/VipNeRF01.py:74-98).  N = n_rays. */
typedef struct vipnerf_rays {
    int64_t n_rays;
    const float *rays_o;      /* (N,3) world-space origin        input_dict['rays_o'] */
    const float *rays_d;      /* (N,3) world-space direction     input_dict['rays_d'] */
    const float *rays_o_s;    /* (N,3) sampling-space origin: rays_o_ndc if ndc else rays_o */
    const float *rays_d_s;    /* (N,3) sampling-space direction: rays_d_ndc if ndc else rays_d */
    const float *view_dirs;   /* (N,3) unit viewing direction    input_dict['view_dirs'] */
    const float *near;        /* (N)   near_ndc if ndc else near */
    const float *far;         /* (N)   far_ndc  if ndc else far */
    const float *rays_o2;     /* (N,V,3) secondary camera centres (VipNeRF01.py:84-98); NULL if n_sec == 0 */
} vipnerf_rays;

/* Random numbers.  The reference draws them on the CPU generator inside the loop (VipNeRF01.py:200,242,551);
 * here they are either supplied (parity tests feed the reference's recorded draws) or, where a pointer is
 * NULL and cfg.train != 0, generated on device from a Philox4x32-10 stream keyed by (seed, offset). */
typedef struct vipnerf_rng {
    const float *t_rand;        /* (N,n_coarse) U[0,1)  stratified jitter */
    const float *u;             /* (N,n_fine)   U[0,1)  inverse-CDF draws */
    const float *noise_coarse;  /* (N,n_coarse) N(0,1)  sigma noise, coarse pass */
    const float *noise_fine;    /* (N,n_coarse+n_fine)  sigma noise, fine pass */
    uint64_t seed;
    uint64_t offset;
    uint64_t ray_base;          /* index of this call's ray 0 in the (global) batch the stream is drawn for: draw (n, k) of a
                                   stream is keyed by (ray_base + n)*S + k, so a rank that renders rays [r0, r0+N) of a batch,
                                   or a call that renders one chunk of it, sees the numbers a single call over the whole
                                   batch would (per-rank Philox offset = rank * shard). 0 for a whole batch. */
    const int64_t *ray_ids;     /* (N) or NULL: explicit global row index of every ray (replaces ray_base + n) -- for shards
                                   that are not one contiguous range, e.g. a rank's share of the nerf rows followed by its
                                   share of the sparse-depth rows */
} vipnerf_rng;

/* The 24 parameter tensors of one MLP in the reference's construction order (VipNeRF01.py:472-491),
 * nn.Linear layout weight[out,in], bias[out]:
 *   0..15  pts_linears[i].weight, pts_linears[i].bias, i = 0..7   (256x63, 256x256 x4, 256x319, 256x256 x2)
 *   16,17  views_linears[0].weight (128x283), .bias
 *   18,19  pts_output_linear.weight (1x256), .bias
 *   20,21  feature_linear.weight (256x256), .bias
 *   22,23  views_output_linear.weight (4x128), .bias */
typedef struct vipnerf_mlp_params {
    const float *p[VIPNERF_N_PARAMS];
} vipnerf_mlp_params;

typedef struct vipnerf_mlp_grads {
    float *g[VIPNERF_N_PARAMS];  /* same shapes; OVERWRITTEN with dLoss/dparam */
} vipnerf_mlp_grads;

/* Outputs of one level (coarse or fine), S = samples of that level (n_coarse, or n_coarse+n_fine).
 * Keys of render_rays' return dict (VipNeRF01.py:128-133,161-166 and volume_rendering :366-383). */
typedef struct vipnerf_level_out {
    float *z_vals;         /* (N,S)   z_vals_L */
    float *raw_sigma;      /* (N,S)   raw_sigma_L[...,0]  (post-ReLU density) */
    float *raw_rgb;        /* (N,S,3) raw_rgb_L */
    float *raw_vis;        /* (N,S)   raw_visibility_L[...,0] */
    float *raw_vis2;       /* (N,S,V) raw_visibility2_L[...,0]; NULL if n_sec == 0 */
    float *alpha;          /* (N,S) */
    float *visibility;     /* (N,S)   transmittance T */
    float *weights;        /* (N,S) */
    float *rgb;            /* (N,3) */
    float *acc;            /* (N) */
    float *depth;          /* (N)   metric depth */
    float *depth_var;      /* (N) */
    float *depth_ndc;      /* (N)   only if ndc, else may be NULL */
    float *depth_var_ndc;  /* (N)   only if ndc */
    float *vis2;           /* (N,V) visibility2_L; NULL if n_sec == 0 */
} vipnerf_level_out;

typedef struct vipnerf_outputs {
    vipnerf_level_out coarse;
    vipnerf_level_out fine;     /* ignored if n_fine == 0 */
    int32_t *sample_inds;       /* (N,n_fine) searchsorted(cdf,u,right=True) indices, for the bit-exact check; may be NULL */
    float   *z_samples;         /* (N,n_fine) importance samples before the merge; may be NULL */
} vipnerf_outputs;

/* dLoss/d(output) for the outputs the reference's losses differentiate.  Any pointer may be
 * NULL (= zero gradient).  Same shapes as vipnerf_level_out. */
typedef struct vipnerf_level_grads {
    const float *rgb;          /* (N,3) */
    const float *acc;          /* (N) */
    const float *depth;        /* (N) */
    const float *depth_ndc;    /* (N) */
    const float *vis2;         /* (N,V) */
    const float *visibility;   /* (N,S) dLoss/dT */
    const float *weights;      /* (N,S) */
    const float *alpha;        /* (N,S) */
    const float *raw_sigma;    /* (N,S) */
    const float *raw_rgb;      /* (N,S,3) */
    const float *raw_vis;      /* (N,S) */
    const float *raw_vis2;     /* (N,S,V) */
    const float *depth_var;    /* (N) */
    const float *depth_var_ndc;/* (N) */
} vipnerf_level_grads;

typedef struct vipnerf_out_grads {
    vipnerf_level_grads coarse;
    vipnerf_level_grads fine;
} vipnerf_out_grads;

/* Loss inputs/outputs for the fused loss kernel (src/loss_functions/{MSE01,VisibilityLoss01,
 * VisibilityPriorLoss01,SparseDepthMSE01}.py). */
typedef struct vipnerf_loss_in {
    const float   *target_rgb;        /* (N,3)   input_dict['target_rgb'] */
    const uint8_t *mask_nerf;         /* (N)     input_dict['indices_mask_nerf'] */
    const float   *prior;             /* (N,V)   visibility_prior_masks / _weights; NULL = ones */
    const uint8_t *mask_sparse;       /* (N)     indices_mask_sparse_depth; NULL = SparseDepthMSE is 0 */
    const float   *sparse_depth;      /* (N)     sparse_depth_values[:,0] */
} vipnerf_loss_in;

/* loss_values[8]: [0] MSE coarse, [1] MSE fine, [2] VisibilityLoss coarse, [3] fine, [4] VisibilityPrior
 * coarse, [5] fine, [6] SparseDepthMSE, [7] unused.  Seeds are the UNWEIGHTED dLoss_k/d(output), written into
 * caller buffers with vipnerf_level_out shapes: seed_rgb (N,3), seed_T (N,S), seed_raw_vis (N,S),
 * seed_vis2 (N,V), seed_depth (N, fine level only). */
typedef struct vipnerf_loss_level_seeds {
    float *rgb; float *visibility; float *raw_vis; float *vis2; float *depth;
} vipnerf_loss_level_seeds;

typedef struct vipnerf_loss_out {
    float *loss_values;               /* (8) device */
    vipnerf_loss_level_seeds coarse, fine;
    float *scratch;                   /* (8 * N) device scratch for the deterministic two-pass reduction */
} vipnerf_loss_out;

/* ---- library / error ------------------------------------------------------------------------------------ */
int32_t vipnerf_abi_version(void);
/* How the loaded library was built: "libvipnerf_hip abi=4 arch=gfx950 VN_EXP=unset VN_...=..." -- every build-time switch of
 * vip-nerf_amd/csrc/vipnerf_knobs.h with its value (static string).  vipnerf_build_is_experiment() != 0 for a TIMING-ONLY experiment build
 * (-DVN_EXP=n: stores, encodings or MFMAs left out, results are garbage): bindings warn, benchmarks and smoke tests must refuse it. */
const char *vipnerf_build_info(void);
int32_t vipnerf_build_is_experiment(void);
/* copies the calling thread's last error text (NUL terminated, truncated to n) */
int32_t vipnerf_last_error(char *buf, size_t n);

/* ---- weights -------------------------------------------------------------------------------------------- */
/* Bytes of the packed (MFMA fragment order) image of one MLP for precision FP32 (== vipnerf_packed_weights_bytes_p(VIPNERF_PREC_FP32));
 * valid for every entry point called with precision FP32. */
size_t  vipnerf_packed_weights_bytes(void);
/* Re-lay one MLP's nn.Linear tensors into the streaming order the kernels consume (forward image, transposed
 * image for dgrad, LDS-resident heads/biases).  Replaces nothing in the reference; it is what lets
 * MLP.forward (VipNeRF01.py:509-596) run as one kernel.  Call after every optimizer step. */
int32_t vipnerf_pack_weights(const vipnerf_mlp_params *params, void *packed, vipnerf_stream_t stream);
/* Same for a given precision: ONE image per precision (forward stages, transposed stages for the data gradient, LDS-resident block).  A
 * buffer packed for one precision must be used with that precision (cfg.precision / the _p argument) only.  bytes_p returns 0 for an
 * unknown or retired precision. */
size_t  vipnerf_packed_weights_bytes_p(int32_t precision);
int32_t vipnerf_pack_weights_p(const vipnerf_mlp_params *params, int32_t precision, void *packed, vipnerf_stream_t stream);
/* The same by configuration.  Fused topology: exactly the _p call for cfg->precision (the whole buffer is written; valid for every entry
 * point called with that precision).  Any other topology: the flat fp32 parameter buffer of the generic kernels (unused slots of params
 * may be NULL: layers >= netdepth). */
size_t  vipnerf_packed_weights_bytes_c(const vipnerf_config *cfg);
int32_t vipnerf_pack_weights_c(const vipnerf_config *cfg, const vipnerf_mlp_params *params, void *packed, vipnerf_stream_t stream);
/* Two MLPs of one configuration (the coarse and the fine model: VipNeRF.__init__, reference src/models/VipNeRF01.py:17-27) packed by ONE
 * launch; the images are those of two vipnerf_pack_weights_c calls. */
int32_t vipnerf_pack_weights2_c(const vipnerf_config *cfg, const vipnerf_mlp_params *params_a, void *packed_a,
                                const vipnerf_mlp_params *params_b, void *packed_b, vipnerf_stream_t stream);

/* ---- workspace ------------------------------------------------------------------------------------------ */
/* acts_bytes: per-call activation store written by render_forward when cfg.save_acts (0 otherwise), read by
 * render_backward.  bwd_bytes: scratch of render_backward.  Both cover coarse + fine.  The generic-topology kernels run
 * layer by layer through HBM: for them acts_bytes is non-zero (and `acts` required) in eval calls as well. */
int32_t vipnerf_query_workspace(const vipnerf_config *cfg, int64_t n_rays, size_t *acts_bytes, size_t *bwd_bytes);

/* ---- the hot path --------------------------------------------------------------------------------------- */
/* VipNeRF.render_rays (src/models/VipNeRF01.py:74-171) for any number of rays (subsumes batchify_rays
 * :47-72 and batchify :295-329): coarse depths -> MLP -> compositing -> inverse-CDF sampling -> fine MLP ->
 * compositing.  `rng` may be NULL when !cfg.train. `acts` may be NULL when !cfg.save_acts. */
int32_t vipnerf_render_forward(const vipnerf_config *cfg, const vipnerf_rays *rays, const vipnerf_rng *rng,
                               const void *packed_coarse, const void *packed_fine,
                               const vipnerf_outputs *out, void *acts, vipnerf_stream_t stream);

/* Backward of the above w.r.t. the MLP parameters (autograd of VipNeRF01.py:74-171; no gradient flows to
 * rays, depths or sampling, VipNeRF01.py:213).  `out` must hold the forward's outputs, `acts` its activation
 * store. */
int32_t vipnerf_render_backward(const vipnerf_config *cfg, const vipnerf_rays *rays,
                                const void *packed_coarse, const void *packed_fine,
                                const vipnerf_outputs *out, const vipnerf_out_grads *gout,
                                const void *acts, void *bwd_ws,
                                const vipnerf_mlp_grads *grads_coarse, const vipnerf_mlp_grads *grads_fine,
                                vipnerf_stream_t stream);

/* MSE01.compute_loss, VisibilityLoss01.compute_loss, VisibilityPriorLoss01.compute_loss and
 * SparseDepthMSE01.compute_loss in one pass: unweighted loss values + unweighted gradient seeds. */
int32_t vipnerf_losses_forward(const vipnerf_config *cfg, int64_t n_rays, const vipnerf_loss_in *in,
                               const vipnerf_outputs *out, const vipnerf_loss_out *lout,
                               vipnerf_stream_t stream);

/* vipnerf_losses_forward + TotalLoss in the same launches: total[0] = sum_k weights[k] * loss_values[k] (k = 0..7 in order, every product
 * and sum rounded to float: the arithmetic of vipnerf_train_step's total_loss, bit for bit) -- LossComputer.compute_losses' accumulation
 * `total_loss += loss_weight * loss_dict['loss_value']` (reference src/loss_functions/LossComputer01.py:33-44) without one multiply and one
 * add kernel per loss -- and named[j] = loss_values[2 j] + loss_values[2 j + 1] (j = 0..3: MSE, VisibilityLoss, VisibilityPriorLoss,
 * SparseDepthMSE as the reference logs them, coarse + fine).  weights: 8 floats on the HOST (this iteration's loss weights by slot);
 * total (1) and named (4, may be NULL) on the device. */
int32_t vipnerf_losses_forward_w(const vipnerf_config *cfg, int64_t n_rays, const vipnerf_loss_in *in,
                                 const vipnerf_outputs *out, const vipnerf_loss_out *lout, const float *weights,
                                 float *total, float *named, vipnerf_stream_t stream);

/* The backward of the fused losses: out_k[i] = g[slot_k] * in_k[i] for up to VIPNERF_MAX_SCALE_SEGS arrays in ONE launch -- the loss
 * kernel's unweighted gradient seeds times the upstream gradient of their loss value (autograd of TotalLoss = sum_k weight_k * loss_k,
 * reference src/loss_functions/LossComputer01.py:33-44, which PyTorch evaluates as one multiplication per seed tensor).  g: the 8 upstream
 * gradients of loss_values, on the device (no host synchronisation); in and out may be the same array. */
#define VIPNERF_MAX_SCALE_SEGS 16
typedef struct vipnerf_scale_seg {
    const float *in; float *out;      /* device */
    int64_t numel;
    int32_t slot;                     /* index into g, 0..7 */
    int32_t reserved;
} vipnerf_scale_seg;
int32_t vipnerf_scale_segments(int32_t n_segs, const vipnerf_scale_seg *segs, const float *g, vipnerf_stream_t stream);
/* The same for TotalLoss of vipnerf_losses_forward_w: out_k[i] = (g_total[0] * weights[slot_k]) * in_k[i] -- g_total: the upstream gradient
 * of TotalLoss, ONE float on the device (autograd's root gradient 1.0: then exactly vipnerf_train_step's weights x seeds); weights: 8 floats
 * on the host. */
int32_t vipnerf_scale_segments_w(int32_t n_segs, const vipnerf_scale_seg *segs, const float *g_total, const float *weights,
                                 vipnerf_stream_t stream);

/* One Adam step on flat fp32 buffers (parameters, both moments, gradients: n elements each) in ONE launch -- the update of
 * torch.optim.Adam (the reference's optimizer, src/Trainer01.py:505-515: betas (0.9, 0.999), no weight decay, no amsgrad), evaluated with
 * the roundings of torch's single-tensor path (torch/optim/adam.py::_single_tensor_adam), whose five elementwise kernels it replaces:
 *     m <- lerp(m, g, lerp_w);  v <- v * beta2;  v <- v + sq_w * (g * g);  d <- sqrt(v) * inv_sqrt_bc2 + eps;  p <- p + neg_step * (m / d)
 * The caller passes the scalars torch derives on the host (in double, then rounded to float): lerp_w = 1 - beta1, sq_w = 1 - beta2,
 * inv_sqrt_bc2 = (float)(1.0 / sqrt(1 - beta2^t)) (the reciprocal in double: ATen's division by a host scalar), neg_step = -lr / (1 - beta1^t).  fma_mask: which of the three fused multiply-adds torch's
 * kernels contract (bit 0 the lerp, bit 1 v's update, bit 2 the parameter's); -1 = the combination tests/ found bit-identical on gfx950. */
int32_t vipnerf_adam_step(int64_t n, float *param, float *exp_avg, float *exp_avg_sq, const float *grad, float lerp_w, float beta2, float sq_w,
                          float inv_sqrt_bc2, float eps, float neg_step, int32_t fma_mask, vipnerf_stream_t stream);

/* ---- one training iteration in ONE call -------------------------------------------------------------------- */
/* The reference trainer's per-iteration sequence (src/Trainer01.py:61-107: model(input_batch) -> LossComputer.compute_losses ->
 * TotalLoss.backward() -> optimizer.step()) queued by a single library call: [secondary camera centres] -> pack both MLPs' weights ->
 * vipnerf_render_forward -> vipnerf_losses_forward -> gradient seeds x loss weights (+ TotalLoss) -> vipnerf_render_backward -> [Adam].
 * Exactly the kernels, launch shapes and order of the separate calls above (results are bit-identical to them; only TotalLoss, which the
 * separate path sums with a library dot product, may differ in the last bit) -- what it removes is the host: ~45 launches enqueued from
 * C++ in ~0.2 ms instead of five Python -> ctypes round trips plus autograd glue (1.6 ms), which bounds the step at the batch sizes the
 * reference's shipped configs train at (1024, 2048 + 2048 rays: NerfLlffTrainerTester01.py:251,261,617).  Stream ordered, no
 * synchronisation, no allocation: every buffer is the caller's and may be reused from call to call.
 *
 * cfg: train and save_acts must be set.  rays->rays_o2 may be produced by the call itself: give poses / pixel_id / n_frames and a
 * rays_o2_out buffer (N, n_frames-1, 3) that rays->rays_o2 points to (vipnerf_secondary_origins), or leave poses NULL.
 * loss_weights[k]: weight of lout->loss_values[k] in TotalLoss (LossComputer01.py:33-44; 0 for a loss that is not configured).
 * The seed arrays of `lout` hold the WEIGHTED seeds afterwards.  grads_*: dLoss/dparam, overwritten (24 + 24 views of one flat buffer
 * when an optimizer step or an all-reduce follows).  adam_n > 0: the update of vipnerf_adam_step on the flat buffers after the backward
 * pass (a multi-GPU caller passes 0, all-reduces the flat gradient and calls vipnerf_adam_step itself). */
typedef struct vipnerf_train_step_args {
    const vipnerf_config *cfg;
    const vipnerf_rays *rays;
    const vipnerf_rng *rng;
    const vipnerf_loss_in *loss_in;
    float loss_weights[8];
    const vipnerf_mlp_params *params_coarse, *params_fine;   /* fp32 master weights (fine: NULL if cfg->n_fine == 0) */
    void *packed_coarse, *packed_fine;                       /* vipnerf_packed_weights_bytes_c(cfg) bytes each: rewritten every call */
    const vipnerf_outputs *out;                              /* every output of vipnerf_render_forward */
    const vipnerf_loss_out *lout;                            /* loss_values (8), seeds, scratch */
    float *total_loss;                                       /* (1) device: sum_k loss_weights[k] * loss_values[k]; may be NULL */
    void *acts, *bwd_ws;                                     /* vipnerf_query_workspace(cfg with save_acts) */
    const vipnerf_mlp_grads *grads_coarse, *grads_fine;
    /* optional: secondary camera centres (VipNeRF01.py:84-98) */
    const float *poses; const void *pixel_id; int32_t pixel_id_is_int64; int32_t n_frames; float *rays_o2_out;
    /* optional: Adam on flat buffers (see vipnerf_adam_step) */
    int64_t adam_n; float *adam_param, *adam_exp_avg, *adam_exp_avg_sq; const float *adam_grad;
    float lerp_w, beta2, sq_w, inv_sqrt_bc2, eps, neg_step; int32_t fma_mask; int32_t reserved;
} vipnerf_train_step_args;
int32_t vipnerf_train_step(const vipnerf_train_step_args *args, vipnerf_stream_t stream);

/* ---- stage-wise entry points (used by the parity tests; each is also a valid standalone op) ------------- */
/* VipNeRF.get_z_vals_coarse (VipNeRF01.py:173-203).  t_rand NULL = no jitter. */
int32_t vipnerf_coarse_depths(int64_t n_rays, int32_t n_samples, int32_t lindisp, const float *near,
                              const float *far, const float *t_rand, float *z_out, vipnerf_stream_t stream);
/* VipNeRF.get_z_vals_fine + sample_pdf (VipNeRF01.py:205-262): u NULL = deterministic linspace.
 * z_fine (N,n_coarse+n_fine) ascending; inds (N,n_fine) int32; z_samples (N,n_fine). */
int32_t vipnerf_sample_fine(int64_t n_rays, int32_t n_coarse, int32_t n_fine, const float *z_coarse,
                            const float *weights_coarse, const float *u, float *z_fine, int32_t *inds,
                            float *z_samples, vipnerf_stream_t stream);
/* MLP.forward (VipNeRF01.py:509-596) on explicit points: pts (P,3), view_dirs (P,3), view_dirs2 (P,V,3) or
 * NULL, noise (P) or NULL -> sigma (P), rgb (P,3), vis (P), vis2 (P,V). */
int32_t vipnerf_mlp_forward(int64_t n_points, int32_t n_sec, const float *pts, const float *view_dirs,
                            const float *view_dirs2, const float *noise, float noise_std, const void *packed,
                            float *sigma, float *rgb, float *vis, float *vis2, vipnerf_stream_t stream);
/* vipnerf_mlp_forward with the GEMM arithmetic of `precision` (packed from vipnerf_pack_weights_p); the unsuffixed call is its FP32 case. */
int32_t vipnerf_mlp_forward_p(int64_t n_points, int32_t n_sec, const float *pts, const float *view_dirs,
                              const float *view_dirs2, const float *noise, float noise_std, int32_t precision,
                              const void *packed, float *sigma, float *rgb, float *vis, float *vis2,
                              vipnerf_stream_t stream);
/* VipNeRF.volume_rendering (+ convert_depth_from_ndc) (VipNeRF01.py:331-403) on explicit network outputs
 * held in lvl->raw_* and lvl->z_vals; fills the remaining fields of *lvl. */
int32_t vipnerf_composite(const vipnerf_config *cfg, const vipnerf_rays *rays, int32_t n_samples,
                          const vipnerf_level_out *lvl, vipnerf_stream_t stream);

/* ---- callers' sides of the path (f-1 ray generation + batch gather, f-2 frame post-processing) - */
/* One camera.  kinv = inverse of the float32 intrinsic matrix; pose = camera-to-world [3][4]; ndc_cx/cy are the
 * float32 values of the reference's `-1. / (w / (2. * fx))`, `-1. / (h / (2. * fy))`
 * (src/data_preprocessors/DataPreprocessor01.py:364-370). */
typedef struct vipnerf_camera {
    float kinv[9];
    float pose[12];
    float ndc_cx, ndc_cy;
    float pad[2];
} vipnerf_camera;

typedef struct vipnerf_raygen {
    int32_t height, width, n_frames, ndc;
    float near, far, near_ndc, far_ndc;
    const vipnerf_camera *cameras;   /* device array [n_frames] */
    const int64_t *indices;          /* (N) flat ray ids  frame*H*W + y*W + x  (the reference's shuffled `indices`,
                                        DataPreprocessor01.py:538); NULL = first_index .. first_index+N-1 */
    int64_t first_index;
    const float *images;             /* (n_frames,H,W,3) float32 in [0,1] or NULL: source of target_rgb */
    const float *prior;              /* (n_frames,n_frames-1,H,W) float32 or NULL: visibility prior masks / weights */
    /* sparse-depth rows (select_batch_indices / load_sparse_depth_cached_batch, DataPreprocessor01.py:544-563, :635-681):
     * rows with row_is_sparse[n] != 0 get rays / pixel_id / bounds like any row, -1 as target_rgb and prior, and their
     * entries of the dense per-pixel tables below; the other rows get -1 in the sparse_* outputs.  NULL = no such rows. */
    const uint8_t *row_is_sparse;    /* (N) or NULL */
    const float *sparse_depths;      /* (n_frames*H*W) sparse_depth_data['depths'] (-1 where unknown) or NULL */
    const float *sparse_errors;      /* (n_frames*H*W) ['reprojection_errors'] or NULL */
    const float *sparse_depths_ndc;  /* (n_frames*H*W) ['depths_ndc'] or NULL */
} vipnerf_raygen;

/* The ray batch the model consumes (load_nerf_cached_batch, DataPreprocessor01.py:566-615).  Any pointer except
 * rays_o / rays_d may be NULL. */
typedef struct vipnerf_ray_batch {
    float *rays_o, *rays_d, *view_dirs;        /* (N,3) */
    float *rays_o_ndc, *rays_d_ndc;            /* (N,3) if ndc */
    float *near, *far, *near_ndc, *far_ndc;    /* (N) */
    int32_t *pixel_id;                         /* (N,3): frame, x, y */
    float *target_rgb;                         /* (N,3) */
    float *prior;                              /* (N,n_frames-1) */
    float *rays_o2;                            /* (N,n_frames-1,3) secondary camera centres (VipNeRF01.py:88-98) */
    float *sparse_depth_values;                /* (N) sparse_depth_values[:,0]; -1 on rows that are not sparse-depth rows */
    float *sparse_depth_errors;                /* (N) */
    float *sparse_depth_values_ndc;            /* (N) */
} vipnerf_ray_batch;

/* DataPreprocessor.get_rays / get_ndc_rays / get_view_dirs (DataPreprocessor01.py:335-378) for the selected
 * pixels + the batch gather of load_nerf_cached_batch (:566-615) and load_visibility_prior_cached_batch (:702-724),
 * recomputed from the cameras instead of gathered from a per-scene ray cache. */
int32_t vipnerf_generate_rays(const vipnerf_raygen *gen, int64_t n_rays, const vipnerf_ray_batch *out,
                              vipnerf_stream_t stream);

/* DataPreprocessor.retrieve_inference_outputs (DataPreprocessor01.py:866-894, :1074-1084): uint8 image
 * (clip to [0,1], x255, round half to even) and non-negative depth maps.  Any in/out pair may be NULL. */
int32_t vipnerf_postprocess_frame(int64_t n_pixels, const float *rgb, const float *depth, const float *depth_var,
                                  const float *depth_ndc, const float *depth_var_ndc, uint8_t *image,
                                  float *o_depth, float *o_depth_var, float *o_depth_ndc, float *o_depth_var_ndc,
                                  vipnerf_stream_t stream);

/* Visibility-prior generator: plane-sweep-volume visibility weights of frame 1 w.r.t. frame 2,
 * VisibilityWeightsComputer.compute_weights (src/prior_generators/visibility/VisibilityMask02_NeRF_LLFF.py:27-162).
 * All geometry in float64 like the reference.  k1_inv = inv(intrinsic1); transform = extrinsic2 @ inv(extrinsic1)
 * (rows 0..2); planes = 1 / linspace(1/min_depth, 1/max_depth, n_planes) -- computed by the host binding with
 * numpy exactly as the reference does (:38-41, :56, :65).  weights64 / weights32 / mask (h*w each) may be NULL;
 * mask = weights > 0.5 (:275-279). */
typedef struct vipnerf_psv {
    int32_t height, width, n_planes, pad;
    double k1_inv[9], transform[12], k2[9];
    double temperature;
    const double *planes;       /* device, (n_planes) */
    const uint8_t *frame1;      /* device, (h,w,3) */
    const uint8_t *frame2;      /* device, (h,w,3) */
} vipnerf_psv;
int32_t vipnerf_visibility_prior(const vipnerf_psv *psv, double *weights64, float *weights32, uint8_t *mask,
                                 vipnerf_stream_t stream);

/* ---- stage exports used by the parity tests of the production-only pieces ------------------------------------ */
/* VipNeRF.compute_other_view_dirs (VipNeRF01.py:218-226) exactly as the MLP kernels evaluate it:
 * z (N,S) sampling-space depths, rays->rays_o/rays_d/rays_o2 -> dirs2 (N,S,V,3) unit vectors. */
int32_t vipnerf_secondary_dirs(const vipnerf_config *cfg, const vipnerf_rays *rays, int32_t n_samples, const float *z,
                               float *dirs2, vipnerf_stream_t stream);
/* The secondary camera centres of every row (VipNeRF.render_rays' index glue, VipNeRF01.py:84-98): rays_o2[n][v] = translation of
 * poses[v + (v >= frame(n))], frame(n) = pixel_id[n][0].  poses (n_frames,4,4) camera-to-world, pixel_id (N,3) int32 or int64
 * -> rays_o2 (N, n_frames-1, 3), what vipnerf_rays.rays_o2 takes.  One launch instead of the reference's per-view gathers. */
int32_t vipnerf_secondary_origins(int64_t n_rays, int32_t n_frames, const float *poses, const void *pixel_id, int32_t pixel_id_is_int64,
                                  float *rays_o2, vipnerf_stream_t stream);
/* The on-device generator behind vipnerf_rng (the reference draws torch.rand / torch.randn on the CPU generator,
 * VipNeRF01.py:200,242,551).  philox4x32_10: out[i] = Philox4x32-10(counter[i], key[i]) (Random123 known-answer
 * vectors).  rng_draw: the n numbers idx = first_idx .. first_idx+n-1 of stream `stream_id` (1 t_rand, 2 u, 3 sigma
 * noise coarse, 4 sigma noise fine) for (seed, offset): kind 0 = U[0,1) (24 bits), kind 1 = N(0,1) (Box-Muller). */
int32_t vipnerf_philox4x32_10(int64_t n, const uint32_t *counters, const uint32_t *keys, uint32_t *out, vipnerf_stream_t stream);
int32_t vipnerf_rng_draw(int32_t kind, uint64_t seed, uint64_t offset, uint32_t stream_id, uint64_t first_idx, int64_t n,
                         float *out, vipnerf_stream_t stream);

/* ---- measurement --------------------------------------------------------------------------------------- */
/* Per-stage device time from HIP events recorded on the launch stream around each kernel (group) the calls
 * above queue.  Off by default.  profile_read waits for the recorded events, aggregates them by stage name
 * ("mlp_fwd_fine", "mlp_dgrad_fine", "wgrad_256x256", ...), returns up to max_entries and clears the log. */
typedef struct vipnerf_profile_entry {
    char    name[32];
    int32_t count;
    float   total_ms;
} vipnerf_profile_entry;
int32_t vipnerf_profile_enable(int32_t on);
int32_t vipnerf_profile_read(vipnerf_profile_entry *entries, int32_t max_entries, int32_t *n_out);

#ifdef __cplusplus
}
#endif
#endif /* VIPNERF_HIP_H */
