"""Generate the golden vectors under tests/golden/ by running the REAL reference (imported from
/root/reference/src, never copied) on seeded synthetic inputs.  Runs only in the build container.

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz
    python oracle/check_goldens.py         # regenerates into a temporary directory and compares with the committed files, bit for bit

What is captured: F1 sample_pdf (+ searchsorted indices), F2 MLP.forward, F3
volume_rendering / convert_depth_from_ndc / compute_other_view_dirs, F4 render_rays in eval mode, F5 one
training step (outputs, the four losses at iter 0 and 40000, gradient digests, parameters after one Adam
step).  The reference draws its random numbers on the CPU generator inside the loop; they are recorded by
wrapping torch.rand / torch.randn while the reference runs and stored in the fixture so that the oracle and
the HIP path can be fed the same draws.

Parameters are produced by oracle.vipnerf_oracle.init_params (numpy PCG64, platform independent) and
loaded into the reference model with load_state_dict, so fixtures only need to store the seed.
"""
import contextlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference/src')

from oracle import vipnerf_oracle as vo  # noqa: E402
from models.ModelFactory import get_model  # noqa: E402  (reference)
from models.VipNeRF01 import VipNeRF, MLP  # noqa: E402  (reference)
from loss_functions.LossComputer01 import LossComputer  # noqa: E402  (reference)

GOLD = os.environ.get('VIPNERF_GOLDEN_OUT') or os.path.join(ROOT, 'tests', 'golden')     # oracle/check_goldens.py redirects it
os.makedirs(GOLD, exist_ok=True)


@contextlib.contextmanager
def record_rng(log):
    r0, n0 = torch.rand, torch.randn

    def rand(*a, **k):
        t = r0(*a, **k)
        log.append(('rand', t.clone()))
        return t

    def randn(*a, **k):
        t = n0(*a, **k)
        log.append(('randn', t.clone()))
        return t

    torch.rand, torch.randn = rand, randn
    try:
        yield
    finally:
        torch.rand, torch.randn = r0, n0


@contextlib.contextmanager
def record_searchsorted(log):
    s0 = torch.searchsorted

    def ss(*a, **k):
        t = s0(*a, **k)
        log.append(t.clone())
        return t

    torch.searchsorted = ss
    try:
        yield
    finally:
        torch.searchsorted = s0


def ref_configs(ndc, depth=8, width=256, n_coarse=64, n_fine=128, netchunk=4096, chunk=4096, sparse=False,
                view_dep_rgb=True, predict_vis=True):
    def mlp(ns):
        return {'num_samples': ns, 'netdepth': depth, 'netwidth': width,
                'points_positional_encoding_degree': 10, 'views_positional_encoding_degree': 4,
                'use_view_dirs': True, 'view_dependent_rgb': view_dep_rgb, 'predict_visibility': predict_vis}
    model = {'name': 'VipNeRF01', 'coarse_mlp': mlp(n_coarse), 'chunk': chunk, 'lindisp': False,
             'netchunk': netchunk, 'perturb': True, 'raw_noise_std': 1.0, 'white_bkgd': False}
    if n_fine > 0:
        model['fine_mlp'] = mlp(n_fine)
    losses = [{'name': 'MSE01', 'weight': 1}, {'name': 'VisibilityLoss01', 'weight': 0.1},
              {'name': 'VisibilityPriorLoss01', 'iter_weights': {'0': 0, '30000': 0.001}}]
    if not predict_vis:
        losses = losses[:1]                      # the visibility losses read visibility2_*, which such a model does not output
    if sparse:
        losses.append({'name': 'SparseDepthMSE01', 'weight': 0.1})
    return {'data_loader': {'ndc': ndc}, 'model': model, 'losses': losses, 'device': [0]}


def ref_model(cfg, params):
    m = get_model(cfg, None)
    sd = {k: torch.from_numpy(v.copy()) for k, v in params.items()}
    m.load_state_dict(sd, strict=True)
    return m


def ref_batch(b, iter_num):
    """oracle synthetic batch -> the reference's batch dict (DataPreprocessor01.py:498-529, 576-615)."""
    rb = {k: v for k, v in b.items() if k not in ('poses', 'ndc')}
    rb['common_data'] = {'poses': b['poses'][None].clone()}
    rb['iter_num'] = iter_num
    return rb


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(GOLD, name + '.npz')
    np.savez_compressed(path, **out)
    print('%-34s %8.1f KB' % (name + '.npz', os.path.getsize(path) / 1024))


def digest(t: torch.Tensor):
    """Compact pin of a large tensor: [sum, l2, first 64 flat, 192 strided samples]."""
    f = t.detach().reshape(-1).double()
    n = f.numel()
    idx = (torch.arange(192, dtype=torch.long) * 7919) % n
    return torch.cat([f.sum()[None], f.norm()[None], f[:64] if n >= 64 else torch.cat([f, f.new_zeros(64 - n)]),
                      f[idx]]).numpy()


# ------------------------------------------------------------------------------------------------ F1
def gen_f1():
    g = np.random.default_rng(101)
    n = 96
    z = np.sort(g.uniform(0, 1, size=(n, 64)).astype(np.float32), axis=1)
    bins = torch.from_numpy(.5 * (z[:, 1:] + z[:, :-1]))
    w = g.random((n, 62), dtype=np.float32) ** 4
    w[:8] = 0.                       # flat cdf rows
    w[8:16, 5:] = 0.                 # mass in the first bins only
    w[16:24, :] = 0.
    w[16:24, 40] = 1.                # a single spike
    w = torch.from_numpy(w)
    # perturb mode: the reference draws u itself
    rlog, slog = [], []
    torch.manual_seed(7)
    with record_rng(rlog), record_searchsorted(slog):
        s_rand = VipNeRF.sample_pdf(bins, w, 128, det=False)
    u_rand, inds_rand = rlog[0][1], slog[0]
    slog = []
    with record_searchsorted(slog):
        s_det = VipNeRF.sample_pdf(bins, w, 128, det=True)
    inds_det = slog[0]
    # margin of each u to the nearest cdf entry (fp64) so tests can select rows that are not ulp-fragile
    wd = (w.double() + 1e-5)
    cdf = torch.cat([torch.zeros(n, 1, dtype=torch.float64), torch.cumsum(wd / wd.sum(-1, keepdim=True), -1)], -1)
    margin = (u_rand.double()[:, :, None] - cdf[:, None, :]).abs().min(-1).values
    npz('f1_sample_pdf', bins=bins, weights=w, u=u_rand, samples_rand=s_rand, inds_rand=inds_rand,
        samples_det=s_det, inds_det=inds_det, margin=margin.float())


# ------------------------------------------------------------------------------------------------ F2
def gen_f2():
    for V, seed in ((1, 201), (2, 202)):
        params = vo.init_params(seed, levels=('coarse',))
        cfg = ref_configs(True)
        mlp = MLP(cfg, cfg['model']['coarse_mlp'])
        mlp.load_state_dict({k[len('coarse_model.'):]: torch.from_numpy(v.copy()) for k, v in params.items()})
        g = np.random.default_rng(seed)
        P = 256
        pts = torch.from_numpy(g.uniform(-1.2, 1.2, size=(P, 3)).astype(np.float32))
        vd = g.standard_normal((P, 3)).astype(np.float32)
        vd /= np.linalg.norm(vd, axis=-1, keepdims=True)
        vd2 = g.standard_normal((P, V, 3)).astype(np.float32)
        vd2 /= np.linalg.norm(vd2, axis=-1, keepdims=True)
        inp = {'pts': pts, 'view_dirs': torch.from_numpy(vd), 'view_dirs2': torch.from_numpy(vd2)}
        mlp.train()
        rlog = []
        torch.manual_seed(seed)
        with record_rng(rlog):
            out_t = mlp({k: v.clone() for k, v in inp.items()})
        noise = rlog[0][1][:, 0]
        mlp.eval()
        out_e = mlp({k: v.clone() for k, v in inp.items()})
        enc = mlp.pts_pos_enc_fn(pts)
        npz(f'f2_mlp_v{V}', seed=seed, pts=pts, view_dirs=vd, view_dirs2=vd2, noise=noise, enc_pts=enc,
            sigma_train=out_t['sigma'][:, 0], rgb_train=out_t['rgb'], vis_train=out_t['visibility'][:, 0],
            vis2_train=out_t['visibility2'][..., 0],
            sigma_eval=out_e['sigma'][:, 0], rgb_eval=out_e['rgb'], vis_eval=out_e['visibility'][:, 0],
            vis2_eval=out_e['visibility2'][..., 0])


# ------------------------------------------------------------------------------------------------ F3
def gen_f3():
    for scene, nf in (('fern', 2), ('dtu', 3)):
        b = vo.synthetic_batch(40, 300 + nf, scene=scene, nf=nf)
        ndc = b['ndc']
        cfg = ref_configs(ndc)
        model = ref_model(cfg, vo.init_params(1))
        g = np.random.default_rng(31 + nf)
        n, S, V = 40, 64, nf - 1
        if ndc:
            z = np.sort(g.uniform(0, 1, size=(n, S)).astype(np.float32), axis=1)
            z[:4, -1] = 1.0                                      # exact z_ndc == 1 guard
        else:
            z = np.sort(g.uniform(0.09, 5.0, size=(n, S)).astype(np.float32), axis=1)
        z = torch.from_numpy(z)
        sigma = g.gamma(0.6, 8.0, size=(n, S)).astype(np.float32)
        sigma[:, ::5] = 0.                                       # zero-density samples
        sigma[4:8, 10] = 1e4                                     # alpha -> 1 saturation
        rgb = g.random((n, S, 3), dtype=np.float32)
        vis2 = g.random((n, S, V, 1), dtype=np.float32)
        net = {'sigma': torch.from_numpy(sigma)[..., None], 'rgb': torch.from_numpy(rgb),
               'visibility2': torch.from_numpy(vis2)}
        o2 = vo.secondary_origins(b['poses'], b['pixel_id'][:, 0].long(), nf)
        if ndc:
            out = model.volume_rendering(net, z_vals_ndc=z, rays_d_ndc=b['rays_d_ndc'], rays_o=b['rays_o'],
                                         rays_d=b['rays_d'], sec_views_vis=True)
            metric = VipNeRF.convert_depth_from_ndc(z, b['rays_o'], b['rays_d'])
        else:
            out = model.volume_rendering(net, z_vals=z, rays_d=b['rays_d'], sec_views_vis=True)
            metric = z
        dirs2 = model.compute_other_view_dirs(z, b['rays_o'], b['rays_d'], o2)
        extra = {k: b[k] for k in ('rays_o', 'rays_d') + (('rays_o_ndc', 'rays_d_ndc') if ndc else ())}
        npz(f'f3_composite_{scene}', ndc=int(ndc), z=z, sigma=sigma, rgb=rgb, vis2=vis2[..., 0], rays_o2=o2,
            metric_depth=metric, dirs2=dirs2, **extra, **{'out_' + k: v for k, v in out.items()})


# ------------------------------------------------------------------------------------------------ F4 / F5
PER_RAY = ['rgb', 'acc', 'depth', 'depth_var', 'depth_ndc', 'depth_var_ndc', 'visibility2']
PER_SAMPLE = ['z_vals', 'alpha', 'visibility', 'weights', 'raw_sigma', 'raw_rgb', 'raw_visibility',
              'raw_visibility2']


def pack_outputs(out, levels):
    d = {}
    for lv in levels:
        for k in PER_RAY + PER_SAMPLE:
            kk = f'{k}_{lv}'
            if kk in out:
                d['out_' + kk] = out[kk]
    return d


def split_rng(rlog, n, n_coarse, n_fine):
    """Reassemble the recorded draws (order: rand(N,Sc), randn per coarse netchunk..., rand(N,Sf), randn per
    fine netchunk...) into (N,S) tensors."""
    rands = [t for k, t in rlog if k == 'rand']
    randns = [t for k, t in rlog if k == 'randn']
    rng = {'t_rand': rands[0]}
    flat = torch.cat([t.reshape(-1) for t in randns]) if randns else None
    pc = n * n_coarse
    if flat is not None:
        rng['noise_coarse'] = flat[:pc].reshape(n, n_coarse)
    if n_fine > 0:
        rng['u'] = rands[1]
        if flat is not None:
            rng['noise_fine'] = flat[pc:].reshape(n, n_coarse + n_fine)
    return rng


def gen_f4():
    n = 48
    b = vo.synthetic_batch(n, 400, scene='fern', nf=2)
    cfg = ref_configs(True, netchunk=2048, chunk=32)            # exercise both host loops
    params = vo.init_params(11, scale=1.6, sigma_bias=0.6)
    model = ref_model(cfg, params).eval()
    slog = []
    with torch.no_grad():
        with record_searchsorted(slog):                                      # the reference's own inverse-CDF indices, chunk by chunk
            out_plain = model(ref_batch(b, 0))                               # retraw False, no secondary
        out_raw = model(ref_batch(b, 0), retraw=True, sec_views_vis=True)    # validation of a train frame
    keys_plain = sorted(out_plain.keys())
    d = pack_outputs(out_raw, ('coarse', 'fine'))
    npz('f4_eval_fern', seed_params=11, scale_params=1.6, sigma_bias=0.6, seed_batch=400, n=n,
        keys_plain=np.array(keys_plain), plain_sample_inds=torch.cat(slog, 0).to(torch.int32),
        **{'plain_' + k: v for k, v in out_plain.items()}, **d)


def gen_f5(tag, scene, nf, n, n_sparse, seed, depth=8, width=256, n_fine=128, pscale=1.6, white_bkgd=False, lindisp=False,
           view_dep_rgb=True, predict_vis=True):
    b = vo.synthetic_batch(n, seed, scene=scene, nf=nf, n_sparse=n_sparse)
    ndc = b['ndc']
    levels = ('coarse', 'fine') if n_fine > 0 else ('coarse',)
    cfg = ref_configs(ndc, depth=depth, width=width, n_fine=n_fine, netchunk=1024, chunk=4096,
                      sparse=n_sparse > 0, view_dep_rgb=view_dep_rgb, predict_vis=predict_vis)
    cfg['model'].update(white_bkgd=white_bkgd, lindisp=lindisp)      # branches no shipped config takes (VipNeRF01.py:190-193, 379-380)
    params = vo.init_params(seed + 1, depth=depth, width=width, levels=levels, scale=pscale, view_dep_rgb=view_dep_rgb, predict_vis=predict_vis)
    model = ref_model(cfg, params).train()
    lossc = LossComputer(cfg)
    opt = torch.optim.Adam(model.parameters(), lr=5e-4, betas=(0.9, 0.999))
    rlog, slog = [], []
    torch.manual_seed(seed)
    with record_rng(rlog), record_searchsorted(slog):
        out = model(ref_batch(b, 40000))
    rng = split_rng(rlog, n + n_sparse, 64, n_fine)
    l40k = lossc.compute_losses(ref_batch(b, 40000), out)
    l0 = lossc.compute_losses(ref_batch(b, 0), out)
    opt.zero_grad(set_to_none=True)
    l40k['TotalLoss'].backward()
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    opt.step()
    after = {k: p.detach().clone() for k, p in model.named_parameters()}
    d = pack_outputs(out, levels)
    for nm, lv in (('l40k', l40k), ('l0', l0)):
        for k, v in lv.items():
            d[f'{nm}_{k}'] = (v['loss_value'] if isinstance(v, dict) else v).detach().reshape(())
    for k, v in grads.items():
        if v.numel() <= 4096:
            d['grad_' + k] = v
        d['gdig_' + k] = digest(v)
    for k, v in after.items():
        d['adig_' + k] = digest(v)
    npz(f'f5_train_{tag}', scene=scene, nf=nf, n=n, n_sparse=n_sparse, seed_batch=seed, seed_params=seed + 1,
        scale_params=pscale, depth=depth, width=width, n_fine=n_fine, white_bkgd=white_bkgd, lindisp=lindisp,
        **({} if view_dep_rgb and predict_vis else {'view_dep_rgb': view_dep_rgb, 'predict_vis': predict_vis, 'out_keys': np.array(sorted(out.keys()))}),
        **({'sample_inds': torch.cat(slog, 0).to(torch.int32)} if slog else {}),
        **{'rng_' + k: v for k, v in rng.items()}, **d)


# ------------------------------------------------------------------------------------------------ F6
def gen_f6():
    """Ray generation (get_rays / get_ndc_rays / get_view_dirs) and inference post-processing of the reference's
    DataPreprocessor, on a 2-camera 24x32 scene with generic (non-axis-aligned) poses."""
    import types
    from types import SimpleNamespace
    for m in ('skimage', 'skimage.transform', 'skimage.io'):
        sys.modules.setdefault(m, types.ModuleType(m))           # absent here; only used for optional down-scaling
    from data_preprocessors.DataPreprocessor01 import DataPreprocessor
    g = np.random.default_rng(61)
    res = (24, 32)
    K = np.array([[40.3, 0, 16.2], [0, 39.7, 11.9], [0, 0, 1.]]).astype('float32')
    poses = []
    for i in range(3):
        q, _ = np.linalg.qr(g.standard_normal((3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] *= -1
        q = 0.15 * q + 0.85 * np.eye(3)                             # mostly forward-facing so that d_z stays away from 0
        u, _, vt = np.linalg.svd(q)
        p = np.eye(4)
        p[:3, :3] = u @ vt
        p[:3, 3] = g.uniform(-0.3, 0.3, 3)
        poses.append(p.astype('float32'))
    poses = np.stack(poses)
    out = {'resolution': np.array(res), 'intrinsic': K, 'poses': poses, 'near': 1.0}
    stub = SimpleNamespace(mip_nerf_used=False)
    for i in range(3):
        o, d = DataPreprocessor.get_rays(stub, res, K, poses[i])
        vd = DataPreprocessor.get_view_dirs(d)
        on, dn = DataPreprocessor.get_ndc_rays(o, d, res, K, 1.0)
        out.update({f'rays_o_{i}': o, f'rays_d_{i}': d, f'view_dirs_{i}': vd, f'rays_o_ndc_{i}': on, f'rays_d_ndc_{i}': dn})
    rgb = g.uniform(-0.2, 1.2, size=(res[0] * res[1], 3)).astype('float32')
    rgb[:16] = (np.arange(48).reshape(16, 3).astype('float32') + 0.5) / 255          # exact .5 ties: round half to even
    depth = g.uniform(-1, 5, size=(res[0] * res[1],)).astype('float32')
    out.update(pp_rgb=rgb, pp_depth=depth, pp_image=DataPreprocessor.post_process_image(rgb.reshape(res[0], res[1], 3)),
               pp_depth_out=DataPreprocessor.post_process_depth(depth.reshape(res)))
    npz('f6_raygen', **out)


# ------------------------------------------------------------------------------------------------ F6b
def gen_f6b():
    """The reference's whole training-side batch loader (DataPreprocessor in 'train' mode, cached batching) on a 3-camera
    24x32 scene with sparse depth and visibility-prior masks: the index schedule (shuffles on numpy's global generator,
    pre-crop, epoch wrap; DataPreprocessor01.py:248-265, 532-563) and, per iteration, every array of the batch dict
    (:566-615 nerf rows, :635-681 sparse-depth rows, :702-724 visibility prior)."""
    import types
    import pandas
    for m in ('skimage', 'skimage.transform', 'skimage.io'):
        sys.modules.setdefault(m, types.ModuleType(m))
    from data_preprocessors.DataPreprocessor01 import DataPreprocessor
    g = np.random.default_rng(62)
    n, h, w = 3, 24, 32
    K = np.array([[40.3, 0, 16.2], [0, 39.7, 11.9], [0, 0, 1.]])
    extr = []
    for i in range(n):
        q, _ = np.linalg.qr(g.standard_normal((3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] *= -1
        q = 0.1 * q + 0.9 * np.eye(3)
        u, _, vt = np.linalg.svd(q)
        e = np.eye(4)
        e[:3, :3] = u @ vt
        e[:3, 3] = g.uniform(-0.3, 0.3, 3)
        extr.append(e)
    extr = np.stack(extr)
    images = g.integers(0, 256, size=(n, h, w, 3)).astype(np.uint8)
    masks = g.random((n, n - 1, h, w)) < 0.6
    sparse = {}
    for fn in (0, 2):                                   # frame 1 has no sparse depth (the reference fills -1 there)
        k = 40
        xs = g.uniform(0, w - 1.01, k)
        ys = g.uniform(0, h - 1.01, k)
        sparse[fn] = pandas.DataFrame({'x': xs, 'y': ys, 'depth': g.uniform(2.0, 9.0, k), 'reprojection_error': g.uniform(0.1, 2.0, k)})
    raw = {'frame_nums': np.arange(n),
           'nerf_data': {'images': images, 'extrinsics': extr, 'intrinsics': np.stack([K] * n), 'bounds': np.array([1.6, 11.0]),
                         'resolution': (h, w)},
           'sparse_depth_data': sparse, 'visibility_prior_data': {'masks': masks}}
    num_rays, num_sd, iters = 40, 8, 17
    configs = {'device': 'cpu', 'model': {'white_bkgd': False},
               'data_loader': {'bd_factor': 0.75, 'batching': True, 'ndc': True, 'downsampling_factor': 1, 'num_rays': num_rays,
                               'sparse_depth': {'num_rays': num_sd}, 'visibility_prior': {'load_masks': True, 'load_weights': False},
                               'recenter_camera_poses': True, 'spherify': False, 'precrop_fraction': 0.5, 'precrop_iterations': 3}}
    np.random.seed(620)
    dp = DataPreprocessor(configs, 'train', raw)
    pd_ = dp.preprocessed_data_dict
    out = {'n': n, 'h': h, 'w': w, 'num_rays': num_rays, 'num_rays_sparse': num_sd, 'iters': iters, 'numpy_seed': 620,
           'precrop_fraction': 0.5, 'precrop_iterations': 3,
           'poses': np.asarray(pd_['nerf_data']['poses']), 'intrinsics': np.asarray(pd_['nerf_data']['intrinsics']),
           'near': float(pd_['nerf_data']['near']), 'far': float(pd_['nerf_data']['far']),
           'near_ndc': float(pd_['nerf_data']['near_ndc']), 'far_ndc': float(pd_['nerf_data']['far_ndc']),
           'images': np.asarray(pd_['nerf_data']['images']).astype(np.float32),
           'masks': masks.astype(np.float32),
           'sparse_depths': pd_['sparse_depth_data']['depths'].numpy().reshape(n, h, w),
           'sparse_errors': pd_['sparse_depth_data']['reprojection_errors'].numpy().reshape(n, h, w),
           'sparse_depths_ndc': pd_['sparse_depth_data']['depths_ndc'].numpy().reshape(n, h, w),
           'indices0': np.asarray(pd_['indices']).copy(), 'indices_sd0': np.asarray(pd_['sparse_depth_data']['indices']).copy()}
    for it in range(iters):
        b = dp.get_next_batch(it)
        for k, v in b.items():
            if isinstance(v, torch.Tensor):
                out[f'it{it}_{k}'] = v.numpy()
            elif k == 'common_data':
                out[f'it{it}_poses'] = v['poses'].numpy()
            else:
                out[f'it{it}_{k}'] = v
    npz('f6b_batches', **out)


# ------------------------------------------------------------------------------------------------ F8
def gen_f8():
    """state_dict key / shape manifest of the reference model as its trainer saves it (wrapped in DataParallel,
    Trainer01.py:517, :352-366): what CheckpointHip01 must read and write."""
    import json
    model = torch.nn.DataParallel(get_model(ref_configs(True), None))
    sd = model.state_dict()
    man = {'keys': list(sd.keys()), 'shapes': {k: list(v.shape) for k, v in sd.items()}}
    with open(os.path.join(GOLD, 'f8_reference_state_dict_manifest.json'), 'w') as f:
        json.dump(man, f, indent=0)
    print('f8_reference_state_dict_manifest.json', len(man['keys']), 'keys')


# ------------------------------------------------------------------------------------------------ F9
def gen_f9(n=128, first_iter=29996, iters=8, seed=900):
    """A K-step TRAJECTORY of the reference's training loop: the reference's own `Trainer.train_one_iter` (Trainer01.py:61-107: next batch ->
    zero_grad -> forward -> compute_losses -> TotalLoss.backward -> optimizer.step) driven the way `Trainer.train` drives it (:292-298:
    `lr_decayer.get_updated_learning_rate(iter_num)` written into every param group, then the iteration), with the reference's
    NeRFLearningRateDecayer (lr_decayers/NeRFLearningRateDecayer01.py:19-23), torch.optim.Adam as Trainer01.py:519-520 builds it, and the
    LossComputer's iteration-dependent weights (LossComputer01.py:46-60) -- the window 29996..30003 is chosen so that the visibility-prior
    weight switches 0 -> 0.001 inside it.  A fresh batch of `n` fern rays per iteration; the CPU generator's draws and the fine depths of
    every iteration are recorded (teacher forcing: a one-ulp CDF tie must not turn the comparison of an optimizer trajectory into one of
    index flips).  Stored: per-iteration lr, loss values, min |gradient| per sampled element (which parameters' Adam updates are
    determined above rounding level), digests of the parameters after every iteration and the small tensors after the last.

    Trainer01 imports tensorboard / skimage / simplejson / deepdiff at module level for its file IO; they are absent here and none of them
    is touched by train_one_iter, so empty modules stand in for the import only (as gen_f6 / gen_f7 do for skimage)."""
    import types
    for m, attrs in (('simplejson', ()), ('skimage', ()), ('skimage.io', ()), ('skimage.transform', ()), ('deepdiff', ('DeepDiff',)),
                     ('torch.utils.tensorboard', ('SummaryWriter',))):
        if m not in sys.modules:
            try:
                __import__(m)
            except ImportError:
                mod = types.ModuleType(m)
                for a in attrs:
                    setattr(mod, a, None)
                sys.modules[m] = mod
    from Trainer01 import Trainer                                                   # (reference)
    from lr_decayers.LearningRateDecayerFactory import get_lr_decayer               # (reference)
    cfg = ref_configs(True, netchunk=8192, chunk=4096)
    cfg['optimizer'] = {'lr_decayer_name': 'NeRFLearningRateDecayer01', 'lr_initial': 0.0005, 'lr_decay': 250, 'beta1': 0.9, 'beta2': 0.999}
    params = vo.init_params(seed + 1, scale=1.6)
    model = ref_model(cfg, params).train()
    batches = [vo.synthetic_batch(n, seed + 10 + i, scene='fern', nf=2) for i in range(iters)]

    class Loader:                                                                   # what train_one_iter asks of the data loader: the next batch dict
        def get_next_batch(self, iter_num):
            return ref_batch(batches[iter_num - first_iter], iter_num)

    out_log, rng_log = [], []
    real_forward = model.forward

    def forward(input_batch, *a, **k):
        rlog = []
        with record_rng(rlog):
            o = real_forward(input_batch, *a, **k)
        rng_log.append(split_rng(rlog, n, 64, 128))
        out_log.append({kk: o[kk].detach().clone() for kk in ('z_vals_fine', 'rgb_fine', 'rgb_coarse')})
        return o

    model.forward = forward
    tr = Trainer.__new__(Trainer)                                                   # no output directories, no tensorboard: only what train_one_iter reads
    tr.configs, tr.model, tr.train_data_loader = cfg, model, Loader()
    tr.loss_computer = LossComputer(cfg)
    tr.optimizer = torch.optim.Adam(list(model.parameters()), lr=cfg['optimizer']['lr_initial'],
                                    betas=(cfg['optimizer']['beta1'], cfg['optimizer']['beta2']))
    tr.lr_decayer = get_lr_decayer(cfg)
    d = {}
    names = [k for k, _ in model.named_parameters()]
    gmin = {k: None for k in names}
    torch.manual_seed(seed)
    for i in range(iters):
        iter_num = first_iter + i
        iter_lr = tr.lr_decayer.get_updated_learning_rate(iter_num)                 # Trainer01.py:293-295
        for param_group in tr.optimizer.param_groups:
            param_group['lr'] = iter_lr
        losses = tr.train_one_iter(iter_num)
        d[f'it{i}_lr'] = np.float64(iter_lr)
        for k, v in losses.items():
            d[f'it{i}_loss_{k}'] = np.float64(v)
        for k, v in rng_log[i].items():
            d[f'it{i}_rng_{k}'] = v
        d[f'it{i}_z_vals_fine'] = out_log[i]['z_vals_fine']
        d[f'it{i}_rgb_fine'] = out_log[i]['rgb_fine']
        for k, p in model.named_parameters():
            g = digest(p.grad)[2:]
            gmin[k] = np.abs(g) if gmin[k] is None else np.minimum(gmin[k], np.abs(g))
            d[f'it{i}_pdig_{k}'] = digest(p)[:2]                                     # [sum, l2] of the parameter after this iteration's step
    for k, p in model.named_parameters():
        d['gmin_' + k] = gmin[k]
        d['adig_' + k] = digest(p)
        if p.numel() <= 4096:
            d['after_' + k] = p.detach().clone()
    npz('f9_trajectory_fern', n=n, first_iter=first_iter, iters=iters, seed=seed, seed_params=seed + 1, scale_params=1.6,
        loss_names=np.array(sorted(k for k in losses)), **d)


# ------------------------------------------------------------------------------------------------ F7
def gen_f7():
    """Visibility-prior generator of the reference (plane-sweep volume) on a 40x56 two-camera toy scene."""
    import types
    for m in ('skimage', 'skimage.transform', 'skimage.io', 'simplejson'):
        sys.modules.setdefault(m, types.ModuleType(m))           # absent here; used only for file IO
    sys.path.insert(0, '/root/reference/src/prior_generators/visibility')
    from VisibilityMask02_NeRF_LLFF import VisibilityWeightsComputer
    g = np.random.default_rng(71)
    h, w = 40, 56
    base = g.integers(0, 256, size=(h // 4 + 2, w // 4 + 2, 3)).astype(np.float64)
    yy, xx = np.meshgrid(np.linspace(0, h // 4, h), np.linspace(0, w // 4, w), indexing='ij')
    y0, x0 = yy.astype(int), xx.astype(int)
    ty, tx = (yy - y0)[..., None], (xx - x0)[..., None]
    img = (1 - ty) * ((1 - tx) * base[y0, x0] + tx * base[y0, x0 + 1]) + ty * ((1 - tx) * base[y0 + 1, x0] + tx * base[y0 + 1, x0 + 1])
    frame1 = np.clip(np.round(img), 0, 255).astype(np.uint8)
    frame2 = np.clip(np.round(np.roll(img, 3, axis=1) + g.normal(0, 6, img.shape)), 0, 255).astype(np.uint8)
    K = np.array([[60., 0, 28.], [0, 60., 20.], [0, 0, 1.]])
    E1 = np.eye(4)
    E2 = np.eye(4)
    ang = 0.04
    E2[:3, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    E2[:3, 3] = [0.25, -0.03, 0.02]
    comp = VisibilityWeightsComputer({'num_depth_planes': 64, 'temperature': 10})
    w12 = comp.compute_weights(frame1, frame2, E1, E2, K, K, 1.5, 9.0)
    w21 = comp.compute_weights(frame2, frame1, E2, E1, K, K, 1.5, 9.0)
    npz('f7_visibility_prior', frame1=frame1, frame2=frame2, K=K, E1=E1, E2=E2, min_depth=1.5, max_depth=9.0,
        n_planes=64, temperature=10.0, weights12=w12, weights21=w21)


if __name__ == '__main__':
    torch.set_num_threads(8)
    gen_f1()
    gen_f2()
    gen_f3()
    gen_f4()
    gen_f5('llff', 'fern', 2, 32, 0, 500)
    gen_f5('realestate', 'realestate', 3, 16, 16, 510)
    gen_f5('dtu', 'dtu', 3, 24, 0, 520)
    gen_f5('toy', 'toy', 2, 64, 0, 530, depth=4, width=64, n_fine=0, pscale=1.0)
    gen_f5('dtu4wl', 'dtu', 4, 20, 0, 540, white_bkgd=True, lindisp=True)      # V = 3 secondary views, white background, lindisp
    # head variants no shipped config uses (MLP.__init__, VipNeRF01.py:467-491): rgb from the trunk head / no visibility / neither
    gen_f5('toy_rgbtrunk', 'toy', 2, 48, 0, 550, depth=4, width=64, n_fine=0, pscale=1.0, view_dep_rgb=False)
    gen_f5('dtu_novis', 'dtu', 3, 16, 0, 560, depth=6, width=64, n_fine=32, pscale=1.3, predict_vis=False)
    gen_f5('fern_plain', 'fern', 2, 16, 0, 570, depth=8, width=32, n_fine=64, pscale=1.6, view_dep_rgb=False, predict_vis=False)
    gen_f6()
    gen_f6b()
    gen_f7()
    gen_f8()
    gen_f9()
