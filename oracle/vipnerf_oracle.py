"""CPU oracle for the ViP-NeRF per-ray hot path.  TEST INFRASTRUCTURE ONLY.

This module is a from-the-math restatement (PyTorch CPU, fp32, autograd) of the reference's per-ray
loop.  It exists so that the HIP path can be checked against something that (a) travels to the GPU box
and (b) has itself been pinned against the real reference: `oracle/gen_golden.py` imports the reference
from /root/reference/src in the build container, runs it on seeded inputs with its RNG draws captured, and
commits the inputs/outputs under tests/golden/; tests/test_oracle_golden.py then checks this file against
those vectors.  Parity status: PINNED by direct import of the reference (the reference ships no tests or
golden vectors of its own).

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this file.  The product
path (vip-nerf_amd/) never does, and raises if the HIP library is missing.

Reference lines each function follows (paths relative to /root/reference/):
  positional_encode      src/models/VipNeRF01.py:416-448, 494-507
  mlp_forward            src/models/VipNeRF01.py:509-596
  coarse_depths          src/models/VipNeRF01.py:173-203
  sample_pdf             src/models/VipNeRF01.py:229-262
  fine_depths            src/models/VipNeRF01.py:205-216
  secondary_dirs         src/models/VipNeRF01.py:218-226
  ndc_to_metric_depth    src/models/VipNeRF01.py:386-403
  composite              src/models/VipNeRF01.py:331-384
  render_rays            src/models/VipNeRF01.py:74-171
  losses                 src/loss_functions/{MSE01,VisibilityLoss01,VisibilityPriorLoss01,SparseDepthMSE01}.py
  total_loss             src/loss_functions/LossComputer01.py:33-69

Differences from the reference that are deliberate: random numbers are *inputs* (`rng` dict with `t_rand`
(N,Sc), `u` (N,Sf), `noise_coarse` (N,Sc), `noise_fine` (N,Sc+Sf)) instead of draws on the CPU generator;
there is no chunk/netchunk host loop unless `chunk` is given (it changes nothing numerically except the
GEMM blocking inside the BLAS).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

EPS_T = 1e-10   # inside the transmittance product
EPS_ACC = 1e-6  # depth / visibility2 normalisation
EPS_PDF = 1e-5  # importance-sampling weight floor and denominator switch


# ----------------------------------------------------------------------------------------------- parameters
def mlp_param_shapes(depth: int = 8, width: int = 256, l_pts: int = 10, l_view: int = 4,
                     view_dep_rgb: bool = True, predict_vis: bool = True):
    """Names and shapes of one MLP's parameters, in the reference's construction order
    (VipNeRF01.py:472-491): pts_linears[0..D-1], views_linears[0], pts_output_linear, feature_linear,
    views_output_linear.  nn.Linear layout: weight[out, in], bias[out].  view_dep_rgb / predict_vis: the mlp configs'
    `view_dependent_rgb` / `predict_visibility` (VipNeRF01.py:467-469, 477-491): rgb moves to the trunk head, visibility is
    dropped, and without either there is no view branch."""
    d_pts = 3 + 6 * l_pts
    d_view = 3 + 6 * l_view
    n_trunk = 1 + (0 if view_dep_rgb else 3)
    n_view = (3 if view_dep_rgb else 0) + (1 if predict_vis else 0)
    shapes = []
    for i in range(depth):
        if i == 0:
            k = d_pts
        elif i == 5:            # layer after the skip concat (skip index 4)
            k = width + d_pts
        else:
            k = width
        shapes.append((f'pts_linears.{i}.weight', (width, k)))
        shapes.append((f'pts_linears.{i}.bias', (width,)))
    if n_view:
        shapes.append(('views_linears.0.weight', (width // 2, width + d_view)))
        shapes.append(('views_linears.0.bias', (width // 2,)))
    shapes.append(('pts_output_linear.weight', (n_trunk, width)))
    shapes.append(('pts_output_linear.bias', (n_trunk,)))
    if n_view:
        shapes.append(('feature_linear.weight', (width, width)))
        shapes.append(('feature_linear.bias', (width,)))
        shapes.append(('views_output_linear.weight', (n_view, width // 2)))
        shapes.append(('views_output_linear.bias', (n_view,)))
    return shapes


def init_params(seed: int, depth: int = 8, width: int = 256, l_pts: int = 10, l_view: int = 4,
                levels=('coarse', 'fine'), scale: float = 1.0, sigma_bias: float = 0.0,
                view_dep_rgb: bool = True, predict_vis: bool = True) -> Dict[str, np.ndarray]:
    """Deterministic, platform-independent parameter set (numpy PCG64), U(-1/sqrt(in), 1/sqrt(in)) like
    nn.Linear's default.  Keys follow the reference's state_dict: `coarse_model.pts_linears.0.weight`, ..."""
    rng = np.random.default_rng(seed)
    out = {}
    shapes = mlp_param_shapes(depth, width, l_pts, l_view, view_dep_rgb, predict_vis)
    for level in levels:
        for name, shape in shapes:
            fan_in = shape[1] if len(shape) == 2 else dict(shapes)[name.replace('bias', 'weight')][1]
            bound = scale / math.sqrt(fan_in)
            out[f'{level}_model.{name}'] = rng.uniform(-bound, bound, size=shape).astype(np.float32)
        out[f'{level}_model.pts_output_linear.bias'][0] += np.float32(sigma_bias)   # lets eval-mode tests see non-zero density
    return out


def params_to_torch(params: Dict[str, np.ndarray], requires_grad: bool = False) -> Dict[str, torch.Tensor]:
    return {k: torch.tensor(v, dtype=torch.float32, requires_grad=requires_grad) for k, v in params.items()}


# ----------------------------------------------------------------------------------------------- encoding / MLP
def positional_encode(x: torch.Tensor, degree: int) -> torch.Tensor:
    """gamma(x) = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)], blocks of 3."""
    parts = [x]
    for l in range(degree):
        f = float(2 ** l)
        parts.append(torch.sin(x * f))
        parts.append(torch.cos(x * f))
    return torch.cat(parts, dim=-1)


def _linear(p, prefix, h):
    return torch.addmm(p[prefix + '.bias'], h, p[prefix + '.weight'].t()) if h.dim() == 2 else \
        torch.matmul(h, p[prefix + '.weight'].t()) + p[prefix + '.bias']


def mlp_forward(p: Dict[str, torch.Tensor], level: str, pts: torch.Tensor, view_dirs: torch.Tensor,
                view_dirs2: Optional[torch.Tensor], noise: Optional[torch.Tensor],
                depth: int = 8, l_pts: int = 10, l_view: int = 4, noise_std: float = 1.0,
                want_feature: bool = False, view_dep_rgb: bool = True, predict_vis: bool = True):
    """pts (P,3), view_dirs (P,3), view_dirs2 (P,V,3) or None, noise (P,) or None ->
    dict(sigma (P,), rgb (P,3), visibility (P,) [if predicted], visibility2 (P,V) [if predicted and view_dirs2 given]).
    MLP.forward, VipNeRF01.py:509-596."""
    pre = f'{level}_model.'
    g = positional_encode(pts, l_pts)
    h = g
    for i in range(depth):
        h = torch.relu(_linear(p, f'{pre}pts_linears.{i}', h))
        if i == 4:
            h = torch.cat([g, h], dim=-1)
    trunk = _linear(p, f'{pre}pts_output_linear', h)
    s_raw = trunk[..., 0]
    if noise is not None:
        s_raw = s_raw + noise * noise_std
    out = {'sigma': torch.relu(s_raw)}
    if not view_dep_rgb:
        out['rgb'] = torch.sigmoid(trunk[..., 1:4])
    if not (view_dep_rgb or predict_vis):
        return out
    feat = _linear(p, f'{pre}feature_linear', h)

    def head(dirs):
        gd = positional_encode(dirs, l_view)
        f = feat if dirs.dim() == 2 else feat[:, None, :].expand(-1, dirs.shape[1], -1)
        hv = torch.relu(_linear(p, f'{pre}views_linears.0', torch.cat([f, gd], dim=-1)))
        return torch.sigmoid(_linear(p, f'{pre}views_output_linear', hv))

    q = head(view_dirs)
    c = 0
    if view_dep_rgb:
        out['rgb'] = q[..., 0:3]
        c = 3
    if predict_vis:
        out['visibility'] = q[..., c]
        if view_dirs2 is not None:
            out['visibility2'] = head(view_dirs2)[..., c]
    if want_feature:
        out['feature'] = feat
    return out


def coarse_depths(near: torch.Tensor, far: torch.Tensor, n_samples: int, t_rand: Optional[torch.Tensor],
                  lindisp: bool = False) -> torch.Tensor:
    """near, far (N,1) -> (N,S).  t_rand (N,S) in [0,1) switches stratified jitter on."""
    tau = torch.linspace(0., 1., steps=n_samples)
    if lindisp:
        z = 1. / (1. / near * (1. - tau) + 1. / far * tau)
    else:
        z = near * (1. - tau) + far * tau
    z = z.expand(near.shape[0], n_samples)
    if t_rand is not None:
        mid = .5 * (z[:, 1:] + z[:, :-1])
        hi = torch.cat([mid, z[:, -1:]], dim=-1)
        lo = torch.cat([z[:, :1], mid], dim=-1)
        z = lo + (hi - lo) * t_rand
    return z


def sample_pdf(bins: torch.Tensor, weights: torch.Tensor, u: torch.Tensor):
    """bins (N,B), weights (N,B-1), u (N,J) -> samples (N,J), inds (N,J) int64 (count of cdf entries <= u)."""
    w = weights + EPS_PDF
    pdf = w / torch.sum(w, dim=-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, dim=-1)], dim=-1)   # (N,B)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    lo = torch.clamp(inds - 1, min=0)
    hi = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_lo, cdf_hi = torch.gather(cdf, 1, lo), torch.gather(cdf, 1, hi)
    b_lo, b_hi = torch.gather(bins, 1, lo), torch.gather(bins, 1, hi)
    den = cdf_hi - cdf_lo
    den = torch.where(den < EPS_PDF, torch.ones_like(den), den)
    t = (u - cdf_lo) / den
    return b_lo + t * (b_hi - b_lo), inds


def fine_depths(z_coarse: torch.Tensor, w_coarse: torch.Tensor, n_fine: int, u: Optional[torch.Tensor]):
    """-> z_fine (N,Sc+Sf) ascending, inds (N,Sf), samples (N,Sf).  u=None -> deterministic linspace."""
    mid = .5 * (z_coarse[:, 1:] + z_coarse[:, :-1])
    if u is None:
        u = torch.linspace(0., 1., steps=n_fine).expand(z_coarse.shape[0], n_fine)
    samples, inds = sample_pdf(mid, w_coarse[:, 1:-1], u)
    samples = samples.detach()
    z, _ = torch.sort(torch.cat([z_coarse, samples], dim=-1), dim=-1)
    return z, inds, samples


def secondary_dirs(z: torch.Tensor, rays_o: torch.Tensor, rays_d: torch.Tensor, rays_o2: torch.Tensor,
                   ndc: bool) -> torch.Tensor:
    """z (N,S), o,d (N,3) world, o2 (N,V,3) -> unit directions (N,S,V,3) from each secondary camera."""
    oz, dz = rays_o[:, None, 2], rays_d[:, None, 2]
    if ndc:
        tn = -(1 + rays_o[:, 2]) / rays_d[:, 2]
        z = (((oz + tn[:, None] * dz) / (1 - z + 1e-6)) - oz) / dz
    p = rays_o[:, None, :] + z[..., None] * rays_d[:, None, :]
    v = p[:, :, None, :] - rays_o2[:, None, :, :]
    return v / torch.norm(v, dim=-1, keepdim=True)


def ndc_to_metric_depth(z_ndc: torch.Tensor, rays_o: torch.Tensor, rays_d: torch.Tensor) -> torch.Tensor:
    oz, dz = rays_o[:, 2:3], rays_d[:, 2:3]
    tn = -(1 + oz) / dz
    c = torch.where(z_ndc == 1., 1e-3, 0.)
    return (oz + tn * dz) / dz * (1 / (1 - z_ndc + c) - 1) + tn


# ----------------------------------------------------------------------------------------------- compositing
def composite(net: Dict[str, torch.Tensor], z: torch.Tensor, dir_for_norm: torch.Tensor, ndc: bool,
              rays_o: Optional[torch.Tensor] = None, rays_d: Optional[torch.Tensor] = None,
              white_bkgd: bool = False) -> Dict[str, torch.Tensor]:
    """net: sigma (N,S), rgb (N,S,3)[, visibility2 (N,S,V)].  z is z_ndc when ndc.  dir_for_norm is rays_d
    (non-NDC) or rays_d_ndc (NDC)."""
    last = torch.full_like(z[:, :1], 1. if ndc else 1e10)
    dist = torch.cat([z, last], dim=-1)
    dist = dist[:, 1:] - dist[:, :-1]
    delta = dist * torch.norm(dir_for_norm[:, None, :], dim=-1)
    alpha = 1. - torch.exp(-net['sigma'] * delta)
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1. - alpha + EPS_T], dim=-1), dim=-1)[:, :-1]
    w = alpha * T
    out = {'rgb': torch.sum(w[..., None] * net['rgb'], dim=-2), 'acc': torch.sum(w, dim=-1),
           'alpha': alpha, 'visibility': T, 'weights': w}
    acc = out['acc']

    def depth_stats(zz):
        d = torch.sum(w * zz, dim=-1) / (acc + EPS_ACC)
        return d, torch.sum(w * torch.square(zz - d[:, None]), dim=-1)

    if ndc:
        out['depth_ndc'], out['depth_var_ndc'] = depth_stats(z)
        out['depth'], out['depth_var'] = depth_stats(ndc_to_metric_depth(z, rays_o, rays_d))
    else:
        out['depth'], out['depth_var'] = depth_stats(z)
    if white_bkgd:
        out['rgb'] = out['rgb'] + (1. - acc[:, None])
    if 'visibility2' in net:
        out['visibility2'] = torch.sum(w[..., None] * net['visibility2'], dim=-2) / (acc[:, None] + EPS_ACC)
    return out


# ----------------------------------------------------------------------------------------------- the ray loop
def secondary_origins(poses: torch.Tensor, image_id: torch.Tensor, num_frames: int) -> torch.Tensor:
    """rays_o2[n, i] = centre of camera (i + (i >= image_id[n])), i = 0..nf-2 (VipNeRF01.py:88-98)."""
    cols = []
    for i in range(num_frames - 1):
        other = i + (i >= image_id).long()
        cols.append(poses[other][:, :3, 3])
    return torch.stack(cols, dim=1)


def render_rays(p: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor], cfg: dict,
                rng: Optional[Dict[str, torch.Tensor]], train: bool, sec_views: bool,
                chunk: Optional[int] = None, netchunk: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """cfg: ndc, n_coarse, n_fine (0 = coarse only), depth, width(unused, implied by p), l_pts, l_view,
    noise_std, lindisp, white_bkgd.  rng: None in eval.  Returns the reference's training-mode key set
    (callers drop what `retraw=False` would drop).  chunk / netchunk: the reference's host loops over ray chunks
    (batchify_rays, VipNeRF01.py:47-72) and over point chunks of the MLP (batchify, :295-329); they change nothing
    numerically except the GEMM blocking, and are what bench.py's cpu_baseline leg runs (structure-equivalent timing)."""
    if chunk is not None and batch['rays_o'].shape[0] > chunk:
        n = batch['rays_o'].shape[0]
        outs = []
        for s in range(0, n, chunk):
            sub = {k: (v[s:s + chunk] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == n else v)
                   for k, v in batch.items()}
            sub_rng = None if rng is None else {k: v[s:s + chunk] for k, v in rng.items()}
            outs.append(render_rays(p, sub, cfg, sub_rng, train, sec_views, None, netchunk))
        return {k: torch.cat([o[k] for o in outs], dim=0) for k in outs[0]}

    ndc = cfg['ndc']
    o, d = batch['rays_o'], batch['rays_d']
    if ndc:
        o_s, d_s, near, far = batch['rays_o_ndc'], batch['rays_d_ndc'], batch['near_ndc'], batch['far_ndc']
    else:
        o_s, d_s, near, far = o, d, batch['near'], batch['far']
    vdir = batch['view_dirs']
    o2 = None
    if sec_views and cfg.get('predict_vis', True):
        o2 = batch['rays_o2'] if 'rays_o2' in batch else secondary_origins(
            batch['poses'], batch['pixel_id'][:, 0].long(), int(batch['num_frames']))
    kw = dict(depth=cfg.get('depth', 8), l_pts=cfg.get('l_pts', 10), l_view=cfg.get('l_view', 4),
              noise_std=cfg.get('noise_std', 1.0), view_dep_rgb=cfg.get('view_dep_rgb', True), predict_vis=cfg.get('predict_vis', True))
    use_noise = train and rng is not None and cfg.get('noise_std', 1.0) > 0 and 'noise_coarse' in rng
    ret = {}

    def level_pass(level, z, noise):
        n, s = z.shape
        pts = (o_s[:, None, :] + d_s[:, None, :] * z[..., None]).reshape(-1, 3)
        vd = vdir[:, None, :].expand(n, s, 3).reshape(-1, 3)
        vd2 = None
        if o2 is not None:
            vd2 = secondary_dirs(z, o, d, o2, ndc).reshape(n * s, -1, 3)
        nz = None if noise is None else noise.reshape(-1)
        if netchunk is None or pts.shape[0] <= netchunk:
            net = mlp_forward(p, level, pts, vd, vd2, nz, **kw)
        else:
            parts = [mlp_forward(p, level, pts[i:i + netchunk], vd[i:i + netchunk], None if vd2 is None else vd2[i:i + netchunk],
                                 None if nz is None else nz[i:i + netchunk], **kw) for i in range(0, pts.shape[0], netchunk)]
            net = {k: torch.cat([q[k] for q in parts], dim=0) for k in parts[0]}
        net = {k: v.reshape(n, s, *v.shape[1:]) for k, v in net.items()}
        comp = composite(net, z, d_s, ndc, o, d, cfg.get('white_bkgd', False))
        ret[f'z_vals_{level}'] = z
        for k, v in comp.items():
            ret[f'{k}_{level}'] = v
        ret[f'raw_sigma_{level}'] = net['sigma'][..., None]
        ret[f'raw_rgb_{level}'] = net['rgb']
        ret[f"raw_rgb_view_{'dependent' if cfg.get('view_dep_rgb', True) else 'independent'}_{level}"] = net['rgb']
        if 'visibility' in net:
            ret[f'raw_visibility_{level}'] = net['visibility'][..., None]
        if 'visibility2' in net:
            ret[f'raw_visibility2_{level}'] = net['visibility2'][..., None]
        return comp

    t_rand = rng['t_rand'] if (train and rng is not None and 't_rand' in rng) else None
    z_c = coarse_depths(near, far, cfg['n_coarse'], t_rand, cfg.get('lindisp', False))
    comp_c = level_pass('coarse', z_c, rng['noise_coarse'] if use_noise else None)
    if cfg.get('n_fine', 0) > 0:
        u = rng['u'] if (train and rng is not None and 'u' in rng) else None
        z_f, inds, samples = fine_depths(z_c, comp_c['weights'], cfg['n_fine'], u)
        ret['sample_inds'] = inds
        ret['z_samples'] = samples
        if rng is not None and rng.get('z_fine') is not None:
            z_f = rng['z_fine']                       # teacher forcing (tests of multi-step trajectories): the given merged fine depths
        level_pass('fine', z_f, rng['noise_fine'] if use_noise else None)
    return ret


# ----------------------------------------------------------------------------------------------- losses
def loss_mse(batch, out, levels):
    m = batch['indices_mask_nerf']
    tot = 0
    for lv in levels:
        e = out[f'rgb_{lv}'][m] - batch['target_rgb'][m]
        tot = tot + (torch.mean(torch.mean(torch.square(e), dim=1)) if e.numel() > 0 else 0)
    return tot


def loss_visibility(batch, out, levels):
    tot = 0
    for lv in levels:
        pred, tgt = out[f'raw_visibility_{lv}'][..., 0], out[f'visibility_{lv}']
        tot = tot + torch.mean(torch.mean(torch.abs(pred - tgt.detach()), dim=1)) \
            + torch.mean(torch.mean(torch.abs(pred.detach() - tgt), dim=1))
    return tot


def loss_visibility_prior(batch, out, levels):
    if any(f'raw_visibility2_{lv}' not in out for lv in levels):
        return None
    m = batch['indices_mask_nerf']
    if 'visibility_prior_masks' in batch:
        pw = batch['visibility_prior_masks']
    elif 'visibility_prior_weights' in batch:
        pw = batch['visibility_prior_weights']
    else:
        pw = torch.ones((batch['rays_o'].shape[0], int(batch['num_frames']) - 1))
    tot = 0
    for lv in levels:
        v2 = out[f'visibility2_{lv}'][m]
        per_ray = torch.sum(pw[m] * (1 - v2), dim=1)
        tot = tot + (torch.mean(per_ray) if v2.numel() > 0 else 0)
    return tot


def loss_sparse_depth(batch, out, levels):
    if 'indices_mask_sparse_depth' not in batch:
        return torch.zeros(())
    m = batch['indices_mask_sparse_depth']
    lv = 'fine' if 'fine' in levels else 'coarse'
    e = out[f'depth_{lv}'][m] - batch['sparse_depth_values'][:, 0][m]
    return torch.mean(torch.square(e)) if e.numel() > 0 else torch.zeros(())


def schedule_weight(loss_cfg: dict, iter_num: int) -> float:
    if 'weight' in loss_cfg:
        return loss_cfg['weight']
    best = None
    for k in sorted((int(k) for k in loss_cfg['iter_weights']), reverse=True):
        if iter_num >= k:
            best = loss_cfg['iter_weights'][str(k)]
            break
    if best is None:
        raise RuntimeError('no loss weight for iteration %d' % iter_num)
    return best


LOSS_FNS = {'MSE01': loss_mse, 'VisibilityLoss01': loss_visibility,
            'VisibilityPriorLoss01': loss_visibility_prior, 'SparseDepthMSE01': loss_sparse_depth}


def total_loss(batch, out, loss_cfgs, iter_num: int, levels=('coarse', 'fine')):
    vals, tot = {}, 0
    for lc in loss_cfgs:
        v = LOSS_FNS[lc['name']](batch, out, levels)
        if v is None:
            continue
        vals[lc['name']] = v
        tot = tot + schedule_weight(lc, iter_num) * v
    vals['TotalLoss'] = tot
    return vals


# ----------------------------------------------------------------------------------------------- synthetic data
def synthetic_batch(n_rays: int, seed: int, scene: str = 'fern', nf: int = 2, n_sparse: int = 0) -> Dict[str, torch.Tensor]:
    """Synthetic ray batch with the reference's batch-dict keys (DataPreprocessor01.py:576-615), built from
    the camera model of SURVEY.md §8(d).  Geometry constants come from the reference's committed
    ModelConfigs.json files (numbers, not code)."""
    g = np.random.default_rng(seed)
    scenes = {
        # H, W, f, near, far, ndc
        'fern': (756, 1008, 815.1316, 1.0, 5.1731, True),
        'realestate': (576, 1024, 900.0, 1.0, 133.33, True),
        'dtu': (300, 400, 361.54, 0.09, 5.0, False),
        'toy': (64, 64, 80.0, 2.0, 6.0, False),
    }
    H, W, f, near, far, ndc = scenes[scene]
    centres = np.zeros((nf, 3), np.float32)
    centres[:, 0] = np.linspace(-0.1, 0.1, nf)
    poses = np.tile(np.eye(4, dtype=np.float32), (nf, 1, 1))
    poses[:, :3, 3] = centres
    n = n_rays + n_sparse
    img = g.integers(0, nf, size=n)
    px = g.integers(0, W, size=n).astype(np.float32)
    py = g.integers(0, H, size=n).astype(np.float32)
    dirs = np.stack([(px - W / 2) / f, -(py - H / 2) / f, -np.ones(n, np.float32)], -1).astype(np.float32)
    rays_d = dirs                                  # identity rotation
    rays_o = centres[img]
    b = {
        'rays_o': rays_o, 'rays_d': rays_d,
        'view_dirs': rays_d / np.linalg.norm(rays_d, axis=-1, keepdims=True),
        'near': np.full((n, 1), near, np.float32), 'far': np.full((n, 1), far, np.float32),
        'pixel_id': np.stack([img, py.astype(np.int64), px.astype(np.int64)], -1).astype(np.int32),
        'target_rgb': g.random((n, 3), dtype=np.float32),
        'visibility_prior_masks': (g.random((n, nf - 1)) < 0.5).astype(np.float32),
        'indices_mask_nerf': np.arange(n) < n_rays,
    }
    if ndc:
        t = -(near + rays_o[:, 2]) / rays_d[:, 2]
        oo = rays_o + t[:, None] * rays_d
        o0 = -1. / (W / (2. * f)) * oo[:, 0] / oo[:, 2]
        o1 = -1. / (H / (2. * f)) * oo[:, 1] / oo[:, 2]
        o2 = 1. + 2. * near / oo[:, 2]
        d0 = -1. / (W / (2. * f)) * (rays_d[:, 0] / rays_d[:, 2] - oo[:, 0] / oo[:, 2])
        d1 = -1. / (H / (2. * f)) * (rays_d[:, 1] / rays_d[:, 2] - oo[:, 1] / oo[:, 2])
        d2 = -2. * near / oo[:, 2]
        b['rays_o_ndc'] = np.stack([o0, o1, o2], -1).astype(np.float32)
        b['rays_d_ndc'] = np.stack([d0, d1, d2], -1).astype(np.float32)
        b['near_ndc'] = np.zeros((n, 1), np.float32)
        b['far_ndc'] = np.ones((n, 1), np.float32)
    if n_sparse > 0:
        b['indices_mask_sparse_depth'] = np.arange(n) >= n_rays
        b['sparse_depth_values'] = g.uniform(near, min(far, 10 * near), size=(n, 1)).astype(np.float32)
    out = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in b.items()}
    out['poses'] = torch.from_numpy(poses)
    out['num_frames'] = nf
    out['ndc'] = ndc
    return out


def synthetic_rng(n_rays: int, n_coarse: int, n_fine: int, seed: int) -> Dict[str, torch.Tensor]:
    g = np.random.default_rng(seed)
    return {
        't_rand': torch.from_numpy(g.random((n_rays, n_coarse), dtype=np.float32)),
        'u': torch.from_numpy(g.random((n_rays, n_fine), dtype=np.float32)),
        'noise_coarse': torch.from_numpy(g.standard_normal((n_rays, n_coarse), dtype=np.float32)),
        'noise_fine': torch.from_numpy(g.standard_normal((n_rays, n_coarse + n_fine), dtype=np.float32)),
    }
