"""Tensor-level wrappers over the C ABI.  PyTorch is used here for device memory and the current stream only;
all arithmetic happens in libvipnerf_hip.so."""
import ctypes as C
import functools
from typing import Dict, List, Optional

import torch

from . import _lib as L

TAIL_PARAMS = ['views_linears.0.weight', 'views_linears.0.bias', 'pts_output_linear.weight', 'pts_output_linear.bias',
               'feature_linear.weight', 'feature_linear.bias', 'views_output_linear.weight', 'views_output_linear.bias']
DEFAULT_TOPOLOGY = (8, 256, 10, 4)            # netdepth, netwidth, points / views positional-encoding degree [, head variant]
HEAD_RGB_TRUNK, HEAD_NO_VISIBILITY = 1, 2     # VIPNERF_HEAD_*: mlp 'view_dependent_rgb' = False / 'predict_visibility' = False


def head_variant(mlp_configs: dict) -> int:
    """vipnerf_config.head_variant of one MLP's config dict (MLP.__init__, VipNeRF01.py:467-469)."""
    return (0 if mlp_configs.get('view_dependent_rgb', True) else HEAD_RGB_TRUNK) | \
        (0 if mlp_configs.get('predict_visibility', True) else HEAD_NO_VISIBILITY)


def _topo5(topology):
    """int depth | (depth, width, l_pts, l_view) | (..., heads) -> the 5-tuple."""
    if isinstance(topology, int):
        return (topology,) + DEFAULT_TOPOLOGY[1:] + (0,)
    t = tuple(int(v) for v in topology)
    return t if len(t) == 5 else t + (0,)


def head_outputs(topology):
    """(rows of pts_output_linear, rows of views_output_linear); 0 view rows = no feature / view layers at all."""
    heads = _topo5(topology)[4]
    return (4 if heads & HEAD_RGB_TRUNK else 1), (0 if heads & HEAD_RGB_TRUNK else 3) + (0 if heads & HEAD_NO_VISIBILITY else 1)


def _tail(topology):
    """(name, ABI slot) of the non-trunk tensors this topology has, in the reference's construction order."""
    n_view = head_outputs(topology)[1]
    return [(n, 16 + i) for i, n in enumerate(TAIL_PARAMS) if n_view > 0 or n.startswith('pts_output_linear')]


@functools.lru_cache(maxsize=None)
def _param_order_cached(topo5):
    return tuple([f'pts_linears.{i}.{wb}' for i in range(topo5[0]) for wb in ('weight', 'bias')] + [n for n, _ in _tail(topo5)])


def param_order(topology=8):
    """Parameter names of one MLP in the reference's construction order (VipNeRF01.py:472-491).  topology: netdepth or a topology tuple."""
    return list(_param_order_cached(_topo5(topology)))


def param_slots(topology=8):
    """vipnerf_mlp_params slot of each entry of param_order(topology): trunk layer i -> 2i, 2i+1; the rest -> 16..23."""
    return list(_param_slots_cached(_topo5(topology)))


@functools.lru_cache(maxsize=None)
def _param_slots_cached(topo5):
    return tuple(list(range(2 * topo5[0])) + [s for _, s in _tail(topo5)])


def param_shapes(topology=DEFAULT_TOPOLOGY):
    return list(_param_shapes_cached(_topo5(topology)))


@functools.lru_cache(maxsize=None)
def _param_shapes_cached(topology):
    depth, width, l_pts, l_view, _ = _topo5(topology)
    dp, dv = 3 + 6 * l_pts, 3 + 6 * l_view
    n_trunk, n_view = head_outputs(topology)
    trunk = []
    for i in range(depth):
        k = dp if i == 0 else (width + dp if (i == 5 and depth > 5) else width)
        trunk += [(width, k), (width,)]
    tail = {'views_linears.0.weight': (width // 2, width + dv), 'views_linears.0.bias': (width // 2,),
            'pts_output_linear.weight': (n_trunk, width), 'pts_output_linear.bias': (n_trunk,),
            'feature_linear.weight': (width, width), 'feature_linear.bias': (width,),
            'views_output_linear.weight': (n_view, width // 2), 'views_output_linear.bias': (n_view,)}
    return tuple(trunk + [tail[n] for n, _ in _tail(topology)])


PARAM_ORDER = param_order(8)
PARAM_SHAPES = param_shapes(DEFAULT_TOPOLOGY)


def _stream(dev=None):
    """hipStream_t of PyTorch's current stream ON THE DEVICE THE TENSORS LIVE ON (not on the thread's current device:
    the reference builds `cuda:{device[0]}` without torch.cuda.set_device, Trainer01.py:58 / CommonUtils01.py:15-27)."""
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


class on_device:
    """`with on_device(t0, t1, ...) as dev:` -- makes the tensors' device the current HIP device for the enclosed ABI
    calls (the library launches on the calling thread's current device) and rejects inputs on different devices."""

    def __init__(self, *tensors):
        devs = {t.device for t in tensors if isinstance(t, torch.Tensor)}
        if len(devs) != 1:
            raise L.VipNerfHipError(f'all tensors of one call must live on one GPU (got {sorted(str(d) for d in devs)})')
        self.dev = devs.pop()
        if self.dev.type != 'cuda':
            raise L.VipNerfHipError(f'tensors must live on the GPU (got {self.dev}); there is no CPU fallback')
        self.guard = torch.cuda.device(self.dev)

    def __enter__(self):
        self.guard.__enter__()
        return self.dev

    def __exit__(self, *exc):
        return self.guard.__exit__(*exc)


def _p(t: Optional[torch.Tensor], dtype=torch.float32, name='tensor'):
    if t is None:
        return None
    if not t.is_cuda:
        raise L.VipNerfHipError(f'{name} must live on the GPU (got {t.device})')
    if t.dtype != dtype:
        raise L.VipNerfHipError(f'{name} must be {dtype} (got {t.dtype})')
    if not t.is_contiguous():
        raise L.VipNerfHipError(f'{name} must be contiguous')
    return t.data_ptr()


def f32c(t: torch.Tensor) -> torch.Tensor:
    # (the common case costs one attribute test instead of three tensor calls: ~70 calls per training step, and at the reference's 1024
    # rays per iteration the 16-bit step is bound by the host's enqueue time, docs/HISTORY.md)
    if t.dtype is torch.float32 and t.is_contiguous():
        return t.detach() if t.requires_grad else t
    return t.detach().to(torch.float32).contiguous()


def as_u8(t: torch.Tensor) -> torch.Tensor:
    """A row mask as the uint8 array the ABI takes: a bool tensor is reinterpreted (same bytes, no launch), anything else converted."""
    if t.dtype is torch.bool and t.is_contiguous():
        return t.view(torch.uint8)
    if t.dtype is torch.uint8 and t.is_contiguous():
        return t
    return t.to(torch.uint8).contiguous()


def make_config(ndc, n_coarse, n_fine, n_sec, train, noise_std=0.0, lindisp=False, white_bkgd=False,
                save_acts=False, perturb=None, precision=0, bf16_layout=0, topology=DEFAULT_TOPOLOGY) -> L.Config:
    c = L.Config()
    c.netdepth, c.netwidth = int(topology[0]), int(topology[1])
    c.pe_degrees = int(topology[2]) | (int(topology[3]) << 8)
    c.head_variant = _topo5(topology)[4]
    c.precision = int(precision)
    c.bf16_layout = int(bf16_layout)
    c.perturb = int(bool(train if perturb is None else perturb))
    c.ndc, c.n_coarse, c.n_fine, c.n_sec = int(bool(ndc)), int(n_coarse), int(n_fine), int(n_sec)
    c.train, c.lindisp, c.white_bkgd, c.save_acts = int(bool(train)), int(bool(lindisp)), int(bool(white_bkgd)), int(bool(save_acts))
    c.noise_std = float(noise_std)
    return c


# ('bf16x3', 'bf16x6' and the 'wide' layout were retired with ABI 5: the library refuses them with the reason -- the names stay so that an old
# configuration fails loudly there instead of with a KeyError here)
PRECISIONS = {'fp32': 0, 'bf16x3': 1, 'bf16x6': 2, 'fp16x3': 3, 'fp16x3h': 4, 'fp16': 5, 'bf16': 6}
LAYOUTS = {'default': 0, 'wide': 1, 'narrow': 2}


def packed_bytes(precision: int = 0) -> int:
    return L.load().vipnerf_packed_weights_bytes_p(int(precision))


def topology_of(cfg: L.Config):
    """The 4-tuple for the default heads (what every caller passed so far), the 5-tuple otherwise."""
    t = (cfg.netdepth or 8, cfg.netwidth or 256, (cfg.pe_degrees & 0xff) if cfg.pe_degrees else 10,
         ((cfg.pe_degrees >> 8) & 0xff) if cfg.pe_degrees else 4)
    return t + (cfg.head_variant,) if cfg.head_variant else t


def pack_weights(params: List[torch.Tensor], out: Optional[torch.Tensor] = None, precision: int = 0,
                 cfg: Optional[L.Config] = None) -> torch.Tensor:
    """params: the tensors of one MLP in param_order(depth) (24 for the default topology).  With `cfg` the image is the one
    cfg's kernels consume (cfg.precision; the flat fp32 buffer of the generic kernels for a non-default topology)."""
    lib = L.load()
    if cfg is None:
        cfg = make_config(True, 64, 0, 0, False, precision=precision)
    topo = topology_of(cfg)
    names, slots, shapes = param_order(topo), param_slots(topo), param_shapes(topo)
    if len(params) != len(names):
        raise L.VipNerfHipError(f'expected {len(names)} parameter tensors for netdepth {topo[0]}, got {len(params)}')
    mp = L.MlpParams()
    keep = []
    for t, name, slot, shp in zip(params, names, slots, shapes):
        if tuple(t.shape) != shp:
            raise L.VipNerfHipError(f'parameter {name} has shape {tuple(t.shape)}, expected {shp} for topology {topo}')
        tc = f32c(t)
        keep.append(tc)
        mp.p[slot] = _p(tc, name=name)
    with on_device(*keep) as dev:
        nbytes = lib.vipnerf_packed_weights_bytes_c(C.byref(cfg))
        if nbytes == 0:
            L.check(-2, 'vipnerf_packed_weights_bytes_c')
        if out is None:
            out = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
        L.check(lib.vipnerf_pack_weights_c(C.byref(cfg), C.byref(mp), _p(out), _stream(dev)), 'vipnerf_pack_weights_c')
    return out


def _mlp_struct(params: List[torch.Tensor], topo, keep: list) -> L.MlpParams:
    names, slots, shapes = param_order(topo), param_slots(topo), param_shapes(topo)
    if len(params) != len(names):
        raise L.VipNerfHipError(f'expected {len(names)} parameter tensors for netdepth {topo[0]}, got {len(params)}')
    mp = L.MlpParams()
    for t, name, slot, shp in zip(params, names, slots, shapes):
        if tuple(t.shape) != shp:
            raise L.VipNerfHipError(f'parameter {name} has shape {tuple(t.shape)}, expected {shp} for topology {topo}')
        tc = f32c(t)
        keep.append(tc)
        mp.p[slot] = _p(tc, name=name)
    return mp


def pack_weights2(params_a: List[torch.Tensor], params_b: List[torch.Tensor], cfg: L.Config):
    """The coarse and the fine MLP of one configuration packed by ONE launch (vipnerf_pack_weights2_c) -> (image_a, image_b), the images of
    two pack_weights(cfg=cfg) calls."""
    lib = L.load()
    topo, keep = topology_of(cfg), []
    ma, mb = _mlp_struct(params_a, topo, keep), _mlp_struct(params_b, topo, keep)
    with on_device(*keep) as dev:
        nbytes = lib.vipnerf_packed_weights_bytes_c(C.byref(cfg))
        if nbytes == 0:
            L.check(-2, 'vipnerf_packed_weights_bytes_c')
        both = torch.empty(2, nbytes // 4, dtype=torch.float32, device=dev)
        L.check(lib.vipnerf_pack_weights2_c(C.byref(cfg), C.byref(ma), _p(both[0]), C.byref(mb), _p(both[1]), _stream(dev)), 'vipnerf_pack_weights2_c')
    return both[0], both[1]


def query_workspace(cfg: L.Config, n_rays: int):
    a, b = C.c_size_t(0), C.c_size_t(0)
    L.check(L.load().vipnerf_query_workspace(C.byref(cfg), n_rays, C.byref(a), C.byref(b)), 'vipnerf_query_workspace')
    return a.value, b.value


def _rays_struct(cfg: L.Config, b: Dict[str, torch.Tensor], keep: list) -> L.Rays:
    r = L.Rays()
    n = b['rays_o'].shape[0]
    r.n_rays = n

    def put(field, t, name):
        tc = f32c(t)
        keep.append(tc)
        setattr(r, field, _p(tc, name=name))

    put('rays_o', b['rays_o'], 'rays_o')
    put('rays_d', b['rays_d'], 'rays_d')
    if cfg.ndc:
        put('rays_o_s', b['rays_o_ndc'], 'rays_o_ndc'); put('rays_d_s', b['rays_d_ndc'], 'rays_d_ndc')
        put('near', b['near_ndc'].reshape(n), 'near_ndc'); put('far', b['far_ndc'].reshape(n), 'far_ndc')
    else:
        put('rays_o_s', b['rays_o'], 'rays_o'); put('rays_d_s', b['rays_d'], 'rays_d')
        put('near', b['near'].reshape(n), 'near'); put('far', b['far'].reshape(n), 'far')
    put('view_dirs', b['view_dirs'], 'view_dirs')
    if cfg.n_sec > 0:
        o2 = b['rays_o2']
        if tuple(o2.shape) != (n, cfg.n_sec, 3):
            raise L.VipNerfHipError(f'rays_o2 has shape {tuple(o2.shape)}, expected {(n, cfg.n_sec, 3)}')
        put('rays_o2', o2, 'rays_o2')
    return r


def alloc_level(n, S, V, ndc, dev) -> Dict[str, torch.Tensor]:
    e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    d = {'z_vals': e(n, S), 'raw_sigma': e(n, S), 'raw_rgb': e(n, S, 3), 'raw_vis': e(n, S), 'alpha': e(n, S),
         'visibility': e(n, S), 'weights': e(n, S), 'rgb': e(n, 3), 'acc': e(n), 'depth': e(n), 'depth_var': e(n)}
    if V > 0:
        d['raw_vis2'] = e(n, S, V)
        d['vis2'] = e(n, V)
    if ndc:
        d['depth_ndc'] = e(n)
        d['depth_var_ndc'] = e(n)
    return d


def _level_struct(d: Dict[str, torch.Tensor]) -> L.LevelOut:
    lo = L.LevelOut()
    for k in L.LEVEL_OUT_FIELDS:
        setattr(lo, k, _p(d.get(k), name=k))
    return lo


def render_forward(cfg: L.Config, batch: Dict[str, torch.Tensor], rng: Optional[dict], packed_coarse, packed_fine,
                   acts: Optional[torch.Tensor] = None, z_fine: Optional[torch.Tensor] = None):
    """-> (coarse dict, fine dict or None, extras dict).  `rng`: {'t_rand','u','noise_coarse','noise_fine'} tensors
    (any may be missing -> device Philox) and optional 'seed'/'offset' ints."""
    lib = L.load()
    keep = []
    rays = _rays_struct(cfg, batch, keep)
    n = rays.n_rays
    dev = batch['rays_o'].device
    coarse = alloc_level(n, cfg.n_coarse, cfg.n_sec, cfg.ndc, dev)
    fine = alloc_level(n, cfg.n_coarse + cfg.n_fine, cfg.n_sec, cfg.ndc, dev) if cfg.n_fine > 0 else None
    out = L.Outputs()
    out.coarse = _level_struct(coarse)
    extras = {}
    cfg.given_z_fine = int(z_fine is not None)
    if fine is not None:
        if z_fine is not None:          # teacher forcing (parity tests)
            fine['z_vals'].copy_(z_fine)
        out.fine = _level_struct(fine)
        extras['sample_inds'] = torch.empty(n, cfg.n_fine, dtype=torch.int32, device=dev)
        extras['z_samples'] = torch.empty(n, cfg.n_fine, dtype=torch.float32, device=dev)
        out.sample_inds = _p(extras['sample_inds'], torch.int32)
        out.z_samples = _p(extras['z_samples'])
    rs = None
    if rng is not None:
        rs = L.Rng()
        need = {'t_rand': n * cfg.n_coarse, 'u': n * cfg.n_fine, 'noise_coarse': n * cfg.n_coarse,
                'noise_fine': n * (cfg.n_coarse + cfg.n_fine)}
        for k in ('t_rand', 'u', 'noise_coarse', 'noise_fine'):
            if rng.get(k) is not None:
                if rng[k].numel() != need[k]:       # the library reads n_rays rows: a short array would be read past its end
                    raise RuntimeError(f"render_forward: rng['{k}'] has {rng[k].numel()} elements, {need[k]} expected "
                                       f"({n} rays)")
                tc = f32c(rng[k])
                keep.append(tc)
                setattr(rs, k, _p(tc, name=k))
        rs.seed = int(rng.get('seed', 0)) & 0xFFFFFFFFFFFFFFFF
        rs.offset = int(rng.get('offset', 0)) & 0xFFFFFFFFFFFFFFFF
        rs.ray_base = int(rng.get('ray_base', 0))
        if rng.get('ray_ids') is not None:
            ids = rng['ray_ids'].to(device=dev, dtype=torch.int64).contiguous()
            if ids.numel() != n:
                raise RuntimeError(f"render_forward: rng['ray_ids'] has {ids.numel()} elements, {n} expected")
            keep.append(ids)
            rs.ray_ids = _p(ids, torch.int64, name='ray_ids')
    if n == 0:                       # an empty batch: empty outputs (their NULL data pointers are not handed to the library)
        extras['_keep'] = keep
        return coarse, fine, extras
    with on_device(*keep, packed_coarse, packed_fine, acts) as dev:
        L.check(lib.vipnerf_render_forward(C.byref(cfg), C.byref(rays), C.byref(rs) if rs is not None else None,
                                           _p(packed_coarse), _p(packed_fine) if packed_fine is not None else None,
                                           C.byref(out), _p(acts) if acts is not None else None, _stream(dev)),
                'vipnerf_render_forward')
    extras['_keep'] = keep
    return coarse, fine, extras


def render_backward(cfg: L.Config, batch, packed_coarse, packed_fine, coarse, fine, grads_coarse: dict,
                    grads_fine: Optional[dict], acts, bwd_ws, gparams_coarse: List[torch.Tensor],
                    gparams_fine: Optional[List[torch.Tensor]]):
    lib = L.load()
    keep = []
    rays = _rays_struct(cfg, batch, keep)
    out = L.Outputs()
    out.coarse = _level_struct(coarse)
    if fine is not None:
        out.fine = _level_struct(fine)
    og = L.OutGrads()

    def fill(lg, d):
        for k in L.LEVEL_GRAD_FIELDS:
            t = d.get(k) if d else None
            if t is not None:
                tc = f32c(t)
                keep.append(tc)
                setattr(lg, k, _p(tc, name='grad_' + k))
    fill(og.coarse, grads_coarse)
    fill(og.fine, grads_fine)
    gc, gf = L.MlpGrads(), L.MlpGrads()
    slots = param_slots(topology_of(cfg))
    for s, t in zip(slots, gparams_coarse):
        gc.g[s] = _p(t, name=f'grad param slot {s}')
    if gparams_fine is not None:
        for s, t in zip(slots, gparams_fine):
            gf.g[s] = _p(t, name=f'grad param slot {s}')
    if rays.n_rays == 0:             # no ray, no gradient (the library OVERWRITES the gradient tensors: zeros here)
        for t in list(gparams_coarse) + list(gparams_fine or []):
            t.zero_()
        return keep
    with on_device(*keep, packed_coarse, packed_fine, acts, bwd_ws, *gparams_coarse) as dev:
        L.check(lib.vipnerf_render_backward(C.byref(cfg), C.byref(rays), _p(packed_coarse),
                                            _p(packed_fine) if packed_fine is not None else None, C.byref(out),
                                            C.byref(og), _p(acts), _p(bwd_ws), C.byref(gc),
                                            C.byref(gf) if gparams_fine is not None else None, _stream(dev)),
                'vipnerf_render_backward')
    return keep


# ------------------------------------------------------------------------------------------------ stage-wise ops
def coarse_depths(near, far, n_samples, t_rand=None, lindisp=False):
    n = near.shape[0]
    near, far = f32c(near.reshape(n)), f32c(far.reshape(n))
    tr = f32c(t_rand) if t_rand is not None else None
    z = torch.empty(n, n_samples, dtype=torch.float32, device=near.device)
    with on_device(near, far, tr) as dev:
        L.check(L.load().vipnerf_coarse_depths(n, n_samples, int(lindisp), _p(near), _p(far), _p(tr), _p(z), _stream(dev)),
                'vipnerf_coarse_depths')
    return z


def sample_fine(z_coarse, w_coarse, n_fine, u=None):
    n, sc = z_coarse.shape
    zc, wc = f32c(z_coarse), f32c(w_coarse)
    uu = f32c(u) if u is not None else None
    dev = zc.device
    zf = torch.empty(n, sc + n_fine, dtype=torch.float32, device=dev)
    inds = torch.empty(n, n_fine, dtype=torch.int32, device=dev)
    zs = torch.empty(n, n_fine, dtype=torch.float32, device=dev)
    with on_device(zc, wc, uu) as dev:
        L.check(L.load().vipnerf_sample_fine(n, sc, n_fine, _p(zc), _p(wc), _p(uu), _p(zf), _p(inds, torch.int32), _p(zs),
                                             _stream(dev)), 'vipnerf_sample_fine')
    return zf, inds, zs


def mlp_forward(packed, pts, view_dirs, view_dirs2=None, noise=None, noise_std=1.0, precision=0):
    P = pts.shape[0]
    V = 0 if view_dirs2 is None else view_dirs2.shape[1]
    dev = pts.device
    pts, vd = f32c(pts), f32c(view_dirs)
    vd2 = f32c(view_dirs2) if view_dirs2 is not None else None
    nz = f32c(noise) if noise is not None else None
    e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    sigma, rgb, vis = e(P), e(P, 3), e(P)
    vis2 = e(P, V) if V > 0 else None
    if P == 0:
        return {'sigma': sigma, 'rgb': rgb, 'visibility': vis, 'visibility2': vis2}
    with on_device(pts, vd, vd2, nz, packed) as dev:
        L.check(L.load().vipnerf_mlp_forward_p(P, V, _p(pts), _p(vd), _p(vd2), _p(nz), float(noise_std), int(precision),
                                               _p(packed), _p(sigma), _p(rgb), _p(vis), _p(vis2), _stream(dev)), 'vipnerf_mlp_forward_p')
    return {'sigma': sigma, 'rgb': rgb, 'visibility': vis, 'visibility2': vis2}


def composite(cfg: L.Config, batch, z, sigma, rgb, vis2=None):
    """volume_rendering on explicit network outputs; returns the level dict."""
    keep = []
    rays = _rays_struct(cfg, batch, keep)
    n, S = z.shape
    lvl = alloc_level(n, S, cfg.n_sec, cfg.ndc, z.device)
    lvl['z_vals'] = f32c(z); lvl['raw_sigma'] = f32c(sigma); lvl['raw_rgb'] = f32c(rgb)
    if cfg.n_sec > 0:
        lvl['raw_vis2'] = f32c(vis2)
    lo = _level_struct(lvl)
    with on_device(*keep, *lvl.values()) as dev:
        L.check(L.load().vipnerf_composite(C.byref(cfg), C.byref(rays), S, C.byref(lo), _stream(dev)), 'vipnerf_composite')
    return lvl


def secondary_dirs(cfg: L.Config, batch, z):
    """compute_other_view_dirs as the MLP kernels evaluate it: z (N,S) -> (N,S,V,3)."""
    keep = []
    rays = _rays_struct(cfg, batch, keep)
    n, S = z.shape
    zc = f32c(z)
    out = torch.empty(n, S, cfg.n_sec, 3, dtype=torch.float32, device=zc.device)
    with on_device(*keep, zc) as dev:
        L.check(L.load().vipnerf_secondary_dirs(C.byref(cfg), C.byref(rays), S, _p(zc), _p(out), _stream(dev)),
                'vipnerf_secondary_dirs')
    return out


def secondary_origins(poses: torch.Tensor, pixel_id: torch.Tensor, n_frames: int) -> torch.Tensor:
    """VipNeRF.render_rays' index glue (VipNeRF01.py:84-98) in one launch: poses (nf,4,4), pixel_id (N,3) int32 / int64 ->
    rays_o2 (N, nf-1, 3), the centres of the other cameras of every row."""
    n = pixel_id.shape[0]
    if poses.dim() != 3 or tuple(poses.shape[1:]) not in ((4, 4), (3, 4)):
        raise L.VipNerfHipError(f'secondary_origins: poses must be (num_frames, 4, 4) or (num_frames, 3, 4) camera-to-world matrices, got {tuple(poses.shape)}')
    if poses.device != pixel_id.device:
        raise L.VipNerfHipError(f'secondary_origins: poses on {poses.device}, pixel_id on {pixel_id.device}')
    if poses.shape[1] == 3:                     # the kernel strides by 16 floats per camera: a 3 x 4 pose gets its [0, 0, 0, 1] row
        last = torch.tensor([0., 0., 0., 1.], dtype=poses.dtype, device=poses.device).expand(poses.shape[0], 1, 4)
        poses = torch.cat([poses, last], dim=1)
    pc = f32c(poses)
    pid = pixel_id if pixel_id.dtype in (torch.int32, torch.int64) else pixel_id.to(torch.int64)
    pid = pid.contiguous()
    out = torch.empty(n, max(n_frames - 1, 0), 3, dtype=torch.float32, device=pid.device)
    if n and n_frames > 1:
        if pc.shape[0] < n_frames:
            raise L.VipNerfHipError(f'secondary_origins: {pc.shape[0]} poses for num_frames={n_frames}')
        with on_device(pc, pid) as dev:
            L.check(L.load().vipnerf_secondary_origins(n, int(n_frames), _p(pc), _p(pid, pid.dtype), int(pid.dtype == torch.int64), _p(out),
                                                       _stream(dev)), 'vipnerf_secondary_origins')
    return out


def philox4x32_10(counters: torch.Tensor, keys: torch.Tensor) -> torch.Tensor:
    """counters (n,4), keys (n,2) int32 bit patterns on the GPU -> (n,4) int32 bit patterns."""
    n = counters.shape[0]
    c, k = counters.contiguous(), keys.contiguous()
    out = torch.empty(n, 4, dtype=torch.int32, device=c.device)
    with on_device(c, k) as dev:
        L.check(L.load().vipnerf_philox4x32_10(n, _p(c, torch.int32), _p(k, torch.int32), _p(out, torch.int32), _stream(dev)),
                'vipnerf_philox4x32_10')
    return out


def rng_draw(kind: str, seed: int, offset: int, stream_id: int, first_idx: int, n: int, device) -> torch.Tensor:
    """The production generator: n draws idx = first_idx.. of stream `stream_id` (1 t_rand, 2 u, 3 / 4 sigma noise)."""
    out = torch.empty(n, dtype=torch.float32, device=device)
    with on_device(out) as dev:
        L.check(L.load().vipnerf_rng_draw({'uniform': 0, 'normal': 1}[kind], seed & 0xFFFFFFFFFFFFFFFF,
                                          offset & 0xFFFFFFFFFFFFFFFF, stream_id, first_idx, n, _p(out), _stream(dev)),
                'vipnerf_rng_draw')
    return out


def losses_forward(cfg: L.Config, n_rays, target_rgb, mask_nerf, prior, mask_sparse, sparse_depth, coarse, fine, weights=None):
    """-> (loss_values (8,), seeds_coarse dict, seeds_fine dict).  weights (8 host floats: this iteration's loss weights by slot): the same
    launches also write TotalLoss = sum_k weights[k] * loss_values[k] and the four per-loss sums (vipnerf_losses_forward_w) -> (loss_values,
    seeds_coarse, seeds_fine, total (1,), named (4,))."""
    dev = target_rgb.device
    keep = []
    li = L.LossIn()
    t = f32c(target_rgb); keep.append(t); li.target_rgb = _p(t)
    if mask_nerf is not None:
        m = as_u8(mask_nerf); keep.append(m); li.mask_nerf = _p(m, torch.uint8)
    if prior is not None:
        pr = f32c(prior); keep.append(pr); li.prior = _p(pr)
    if mask_sparse is not None:
        m2 = as_u8(mask_sparse); keep.append(m2); li.mask_sparse = _p(m2, torch.uint8)
        sd = f32c(sparse_depth.reshape(n_rays)); keep.append(sd); li.sparse_depth = _p(sd)
    out = L.Outputs()
    out.coarse = _level_struct(coarse)
    if fine is not None:
        out.fine = _level_struct(fine)
    lo = L.LossOut()
    vals = torch.empty(8, dtype=torch.float32, device=dev)
    scratch = torch.empty(8 * n_rays + 8, dtype=torch.float32, device=dev)
    lo.loss_values, lo.scratch = _p(vals), _p(scratch)
    seeds = []
    for lvl, S, st in ((coarse, cfg.n_coarse, lo.coarse), (fine, cfg.n_coarse + cfg.n_fine, lo.fine)):
        if lvl is None:
            seeds.append(None)
            continue
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        d = {'rgb': e(n_rays, 3), 'visibility': e(n_rays, S), 'raw_vis': e(n_rays, S)}
        if cfg.n_sec > 0:
            d['vis2'] = e(n_rays, cfg.n_sec)
        last = (lvl is fine) or (fine is None)
        if last:
            d['depth'] = e(n_rays)                # (the loss kernel writes every row's seed on the last level: 0 where no sparse depth)
        for k, v in d.items():
            setattr(st, k, _p(v))
        seeds.append(d)
    if weights is not None:
        tn = torch.empty(5, dtype=torch.float32, device=dev)          # [TotalLoss, four per-loss sums]
        w8 = (C.c_float * 8)(*[float(w) for w in weights])
        with on_device(*keep, vals) as dev:
            L.check(L.load().vipnerf_losses_forward_w(C.byref(cfg), n_rays, C.byref(li), C.byref(out), C.byref(lo), w8, _p(tn[0:1]), _p(tn[1:5]),
                                                      _stream(dev)), 'vipnerf_losses_forward_w')
        return vals, seeds[0], seeds[1], tn[0:1], tn[1:5]
    with on_device(*keep, vals) as dev:
        L.check(L.load().vipnerf_losses_forward(C.byref(cfg), n_rays, C.byref(li), C.byref(out), C.byref(lo), _stream(dev)),
                'vipnerf_losses_forward')
    return vals, seeds[0], seeds[1]


def scale_segments(tensors: List[torch.Tensor], slots: List[int], g: torch.Tensor, weights=None) -> List[torch.Tensor]:
    """-> [g[slot_k] * tensors[k]] in ONE launch (the fused losses' backward: seeds times the upstream gradients of their loss values).
    The results are views of one buffer.  weights (8 host floats): g is the upstream gradient of the weighted TotalLoss instead, ONE value
    on the device, and the factors are (g[0] * weights[slot_k]) (vipnerf_scale_segments_w)."""
    if not tensors:
        return []
    if len(tensors) > 16:
        raise L.VipNerfHipError(f'scale_segments: {len(tensors)} tensors (at most 16)')
    ins = [f32c(t) for t in tensors]
    gc = f32c(g).reshape(-1)
    if gc.numel() < (8 if weights is None else 1):
        raise L.VipNerfHipError('scale_segments: g must hold 8 values (1 with weights)')
    offs, total = [], 0
    for t in ins:
        offs.append(total)
        total += (t.numel() + 3) & ~3
    flat = torch.empty(total, dtype=torch.float32, device=ins[0].device)
    outs = [flat[o:o + t.numel()].view(t.shape) for o, t in zip(offs, ins)]
    segs = (L.ScaleSeg * len(ins))()
    for k, (t, o, sl) in enumerate(zip(ins, outs, slots)):
        segs[k].in_, segs[k].out, segs[k].numel, segs[k].slot = _p(t) if t.numel() else None, _p(o) if t.numel() else None, t.numel(), int(sl)
    with on_device(*ins, gc, flat) as dev:
        if weights is None:
            L.check(L.load().vipnerf_scale_segments(len(ins), segs, _p(gc), _stream(dev)), 'vipnerf_scale_segments')
        else:
            w8 = (C.c_float * 8)(*[float(w) for w in weights])
            L.check(L.load().vipnerf_scale_segments_w(len(ins), segs, _p(gc), w8, _stream(dev)), 'vipnerf_scale_segments_w')
    return outs


def adam_step(param: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, grad: torch.Tensor, lr: float, beta1: float, beta2: float,
              eps: float, step: int, fma_mask: int = -1):
    """One Adam step on flat fp32 device buffers in ONE launch (in place), with torch.optim.Adam's single-tensor roundings; the host-side
    scalars are derived exactly as torch/optim/adam.py::_single_tensor_adam derives them (Python doubles, rounded to float by the kernels'
    scalar arguments)."""
    import numpy as np
    for t in (param, exp_avg, exp_avg_sq, grad):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.dim() != 1 or t.numel() != param.numel():
            raise L.VipNerfHipError('adam_step: flat contiguous fp32 buffers of one size')
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    inv_s = float(np.float32(1.0 / bc2 ** 0.5))     # torch divides by a host scalar as a multiplication by its reciprocal, taken in double, rounded to float
    with on_device(param, exp_avg, exp_avg_sq, grad) as dev:
        L.check(L.load().vipnerf_adam_step(param.numel(), _p(param), _p(exp_avg), _p(exp_avg_sq), _p(grad), 1 - beta1, beta2, 1 - beta2, inv_s, eps,
                                           -(lr / bc1), int(fma_mask), _stream(dev)), 'vipnerf_adam_step')


# ------------------------------------------------------------------------------------------------ measurement
def profile_enable(on: bool):
    L.check(L.load().vipnerf_profile_enable(int(on)), 'vipnerf_profile_enable')


def profile_read() -> Dict[str, tuple]:
    """-> {stage name: (launch count, total ms)} since the last read (waits for the recorded events)."""
    arr = (L.ProfileEntry * 32)()
    n = C.c_int32(0)
    L.check(L.load().vipnerf_profile_read(arr, 32, C.byref(n)), 'vipnerf_profile_read')
    return {arr[i].name.decode(): (arr[i].count, arr[i].total_ms) for i in range(n.value)}
