"""ctypes binding of libvipnerf_hip.so (include/vipnerf_hip.h).

The library is the product path; there is no fallback.  If it cannot be loaded every entry point raises
`VipNerfHipError` -- nothing in this package routes through the CPU oracle.
"""
import ctypes as C
import os

VIPNERF_MAX_SEC = 3
VIPNERF_N_PARAMS = 24
ABI_VERSION = 6

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('VIPNERF_HIP_LIB') or os.path.join(os.path.dirname(_HERE), 'lib', 'libvipnerf_hip.so')

c_f = C.c_void_p   # device pointers travel as plain addresses


class VipNerfHipError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [('ndc', C.c_int32), ('n_coarse', C.c_int32), ('n_fine', C.c_int32), ('n_sec', C.c_int32),
                ('train', C.c_int32), ('lindisp', C.c_int32), ('white_bkgd', C.c_int32), ('save_acts', C.c_int32),
                ('noise_std', C.c_float), ('given_z_fine', C.c_int32), ('perturb', C.c_int32), ('precision', C.c_int32), ('bf16_layout', C.c_int32),
                ('netdepth', C.c_int32), ('netwidth', C.c_int32), ('pe_degrees', C.c_int32), ('head_variant', C.c_int32)]


class Rays(C.Structure):
    _fields_ = [('n_rays', C.c_int64), ('rays_o', c_f), ('rays_d', c_f), ('rays_o_s', c_f), ('rays_d_s', c_f),
                ('view_dirs', c_f), ('near', c_f), ('far', c_f), ('rays_o2', c_f)]


class Rng(C.Structure):
    _fields_ = [('t_rand', c_f), ('u', c_f), ('noise_coarse', c_f), ('noise_fine', c_f),
                ('seed', C.c_uint64), ('offset', C.c_uint64), ('ray_base', C.c_uint64), ('ray_ids', c_f)]


class MlpParams(C.Structure):
    _fields_ = [('p', c_f * VIPNERF_N_PARAMS)]


class MlpGrads(C.Structure):
    _fields_ = [('g', c_f * VIPNERF_N_PARAMS)]


LEVEL_OUT_FIELDS = ['z_vals', 'raw_sigma', 'raw_rgb', 'raw_vis', 'raw_vis2', 'alpha', 'visibility', 'weights',
                    'rgb', 'acc', 'depth', 'depth_var', 'depth_ndc', 'depth_var_ndc', 'vis2']


class LevelOut(C.Structure):
    _fields_ = [(k, c_f) for k in LEVEL_OUT_FIELDS]


class Outputs(C.Structure):
    _fields_ = [('coarse', LevelOut), ('fine', LevelOut), ('sample_inds', c_f), ('z_samples', c_f)]


LEVEL_GRAD_FIELDS = ['rgb', 'acc', 'depth', 'depth_ndc', 'vis2', 'visibility', 'weights', 'alpha', 'raw_sigma',
                     'raw_rgb', 'raw_vis', 'raw_vis2', 'depth_var', 'depth_var_ndc']


class LevelGrads(C.Structure):
    _fields_ = [(k, c_f) for k in LEVEL_GRAD_FIELDS]


class OutGrads(C.Structure):
    _fields_ = [('coarse', LevelGrads), ('fine', LevelGrads)]


class LossIn(C.Structure):
    _fields_ = [('target_rgb', c_f), ('mask_nerf', c_f), ('prior', c_f), ('mask_sparse', c_f), ('sparse_depth', c_f)]


class LossLevelSeeds(C.Structure):
    _fields_ = [('rgb', c_f), ('visibility', c_f), ('raw_vis', c_f), ('vis2', c_f), ('depth', c_f)]


class LossOut(C.Structure):
    _fields_ = [('loss_values', c_f), ('coarse', LossLevelSeeds), ('fine', LossLevelSeeds), ('scratch', c_f)]


class ScaleSeg(C.Structure):
    _fields_ = [('in_', c_f), ('out', c_f), ('numel', C.c_int64), ('slot', C.c_int32), ('reserved', C.c_int32)]


class TrainStepArgs(C.Structure):
    _fields_ = [('cfg', C.POINTER(Config)), ('rays', C.POINTER(Rays)), ('rng', C.POINTER(Rng)), ('loss_in', C.POINTER(LossIn)),
                ('loss_weights', C.c_float * 8),
                ('params_coarse', C.POINTER(MlpParams)), ('params_fine', C.POINTER(MlpParams)),
                ('packed_coarse', c_f), ('packed_fine', c_f),
                ('out', C.POINTER(Outputs)), ('lout', C.POINTER(LossOut)), ('total_loss', c_f),
                ('acts', c_f), ('bwd_ws', c_f),
                ('grads_coarse', C.POINTER(MlpGrads)), ('grads_fine', C.POINTER(MlpGrads)),
                ('poses', c_f), ('pixel_id', c_f), ('pixel_id_is_int64', C.c_int32), ('n_frames', C.c_int32), ('rays_o2_out', c_f),
                ('adam_n', C.c_int64), ('adam_param', c_f), ('adam_exp_avg', c_f), ('adam_exp_avg_sq', c_f), ('adam_grad', c_f),
                ('lerp_w', C.c_float), ('beta2', C.c_float), ('sq_w', C.c_float), ('inv_sqrt_bc2', C.c_float), ('eps', C.c_float),
                ('neg_step', C.c_float), ('fma_mask', C.c_int32), ('reserved', C.c_int32)]


class Camera(C.Structure):
    _fields_ = [('kinv', C.c_float * 9), ('pose', C.c_float * 12), ('ndc_cx', C.c_float), ('ndc_cy', C.c_float),
                ('pad', C.c_float * 2)]


class RayGen(C.Structure):
    _fields_ = [('height', C.c_int32), ('width', C.c_int32), ('n_frames', C.c_int32), ('ndc', C.c_int32),
                ('near', C.c_float), ('far', C.c_float), ('near_ndc', C.c_float), ('far_ndc', C.c_float),
                ('cameras', c_f), ('indices', c_f), ('first_index', C.c_int64), ('images', c_f), ('prior', c_f),
                ('row_is_sparse', c_f), ('sparse_depths', c_f), ('sparse_errors', c_f), ('sparse_depths_ndc', c_f)]


RAY_BATCH_FIELDS = ['rays_o', 'rays_d', 'view_dirs', 'rays_o_ndc', 'rays_d_ndc', 'near', 'far', 'near_ndc', 'far_ndc',
                    'pixel_id', 'target_rgb', 'prior', 'rays_o2', 'sparse_depth_values', 'sparse_depth_errors',
                    'sparse_depth_values_ndc']


class RayBatch(C.Structure):
    _fields_ = [(k, c_f) for k in RAY_BATCH_FIELDS]


class Psv(C.Structure):
    _fields_ = [('height', C.c_int32), ('width', C.c_int32), ('n_planes', C.c_int32), ('pad', C.c_int32),
                ('k1_inv', C.c_double * 9), ('transform', C.c_double * 12), ('k2', C.c_double * 9),
                ('temperature', C.c_double), ('planes', c_f), ('frame1', c_f), ('frame2', c_f)]


class ProfileEntry(C.Structure):
    _fields_ = [('name', C.c_char * 32), ('count', C.c_int32), ('total_ms', C.c_float)]


# every symbol include/vipnerf_hip.h declares: name -> (restype, argtypes)
P = C.POINTER
SYMBOLS = {
    'vipnerf_abi_version': (C.c_int32, []),
    'vipnerf_last_error': (C.c_int32, [C.c_char_p, C.c_size_t]),
    'vipnerf_build_info': (C.c_char_p, []),
    'vipnerf_build_is_experiment': (C.c_int32, []),
    'vipnerf_packed_weights_bytes': (C.c_size_t, []),
    'vipnerf_pack_weights': (C.c_int32, [P(MlpParams), c_f, c_f]),
    'vipnerf_packed_weights_bytes_p': (C.c_size_t, [C.c_int32]),
    'vipnerf_pack_weights_p': (C.c_int32, [P(MlpParams), C.c_int32, c_f, c_f]),
    'vipnerf_packed_weights_bytes_c': (C.c_size_t, [P(Config)]),
    'vipnerf_pack_weights_c': (C.c_int32, [P(Config), P(MlpParams), c_f, c_f]),
    'vipnerf_pack_weights2_c': (C.c_int32, [P(Config), P(MlpParams), c_f, P(MlpParams), c_f, c_f]),
    'vipnerf_mlp_forward_p': (C.c_int32, [C.c_int64, C.c_int32, c_f, c_f, c_f, c_f, C.c_float, C.c_int32, c_f, c_f, c_f,
                                          c_f, c_f, c_f]),
    'vipnerf_query_workspace': (C.c_int32, [P(Config), C.c_int64, P(C.c_size_t), P(C.c_size_t)]),
    'vipnerf_render_forward': (C.c_int32, [P(Config), P(Rays), P(Rng), c_f, c_f, P(Outputs), c_f, c_f]),
    'vipnerf_render_backward': (C.c_int32, [P(Config), P(Rays), c_f, c_f, P(Outputs), P(OutGrads), c_f, c_f,
                                            P(MlpGrads), P(MlpGrads), c_f]),
    'vipnerf_losses_forward': (C.c_int32, [P(Config), C.c_int64, P(LossIn), P(Outputs), P(LossOut), c_f]),
    'vipnerf_scale_segments': (C.c_int32, [C.c_int32, P(ScaleSeg), c_f, c_f]),
    'vipnerf_losses_forward_w': (C.c_int32, [P(Config), C.c_int64, P(LossIn), P(Outputs), P(LossOut), C.POINTER(C.c_float), c_f, c_f, c_f]),
    'vipnerf_scale_segments_w': (C.c_int32, [C.c_int32, P(ScaleSeg), c_f, C.POINTER(C.c_float), c_f]),
    'vipnerf_train_step': (C.c_int32, [P(TrainStepArgs), c_f]),
    'vipnerf_adam_step': (C.c_int32, [C.c_int64, c_f, c_f, c_f, c_f, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, c_f]),
    'vipnerf_coarse_depths': (C.c_int32, [C.c_int64, C.c_int32, C.c_int32, c_f, c_f, c_f, c_f, c_f]),
    'vipnerf_sample_fine': (C.c_int32, [C.c_int64, C.c_int32, C.c_int32, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    'vipnerf_mlp_forward': (C.c_int32, [C.c_int64, C.c_int32, c_f, c_f, c_f, c_f, C.c_float, c_f, c_f, c_f, c_f,
                                        c_f, c_f]),
    'vipnerf_composite': (C.c_int32, [P(Config), P(Rays), C.c_int32, P(LevelOut), c_f]),
    'vipnerf_generate_rays': (C.c_int32, [P(RayGen), C.c_int64, P(RayBatch), c_f]),
    'vipnerf_postprocess_frame': (C.c_int32, [C.c_int64, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f]),
    'vipnerf_visibility_prior': (C.c_int32, [P(Psv), c_f, c_f, c_f, c_f]),
    'vipnerf_secondary_dirs': (C.c_int32, [P(Config), P(Rays), C.c_int32, c_f, c_f, c_f]),
    'vipnerf_secondary_origins': (C.c_int32, [C.c_int64, C.c_int32, c_f, c_f, C.c_int32, c_f, c_f]),
    'vipnerf_philox4x32_10': (C.c_int32, [C.c_int64, c_f, c_f, c_f, c_f]),
    'vipnerf_rng_draw': (C.c_int32, [C.c_int32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_int64, c_f, c_f]),
    'vipnerf_profile_enable': (C.c_int32, [C.c_int32]),
    'vipnerf_profile_read': (C.c_int32, [P(ProfileEntry), C.c_int32, P(C.c_int32)]),
}

_lib = None


def load():
    """Load (once) and return the CDLL.  Raises VipNerfHipError if the library or any symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VipNerfHipError(f'{LIB_PATH} not found: build it with vip-nerf_amd/build.sh '
                              f'(or __graft_entry__.build()); there is no CPU fallback')
    # PyTorch-ROCm brings its own HIP runtime (torch/lib/libamdhip64.so); the device memory this library works on is PyTorch's, so
    # that runtime must be the one in the process.  Loaded the other way round -- this library first, resolving its libamdhip64
    # from /opt/rocm, torch afterwards -- the process holds two runtimes and the second one to touch the driver reports "no
    # ROCm-capable device is detected" (seen with build() followed by smoke() in one interpreter).
    import torch  # noqa: F401
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise VipNerfHipError(f'cannot load {LIB_PATH}: {e}') from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise VipNerfHipError(f'{LIB_PATH} does not export {name}') from e
        fn.restype = res
        fn.argtypes = args
    if lib.vipnerf_abi_version() != ABI_VERSION:
        raise VipNerfHipError(f'ABI mismatch: library {lib.vipnerf_abi_version()} != binding {ABI_VERSION}')
    if lib.vipnerf_build_is_experiment():
        import warnings
        warnings.warn(f'{LIB_PATH} is a TIMING-ONLY experiment build ({lib.vipnerf_build_info().decode()}): its results are garbage; '
                      f'bench.py and smoke() refuse it', RuntimeWarning, stacklevel=2)
    _lib = lib
    return lib


def build_info() -> str:
    """The loaded library's own account of how it was built (every VN_* switch with its value)."""
    return load().vipnerf_build_info().decode()


def require_product_build(who: str):
    """Raise if the loaded library is a timing-only experiment build (-DVN_EXP=n)."""
    lib = load()
    if lib.vipnerf_build_is_experiment():
        raise VipNerfHipError(f'{who}: {LIB_PATH} is a timing-only experiment build ({lib.vipnerf_build_info().decode()}); '
                              f'rebuild with vip-nerf_amd/build.sh (no VN_EXP)')


def check(rc, what):
    if rc != 0:
        buf = C.create_string_buffer(512)
        load().vipnerf_last_error(buf, 512)
        raise VipNerfHipError(f'{what} failed (rc={rc}): {buf.value.decode(errors="replace")}')
