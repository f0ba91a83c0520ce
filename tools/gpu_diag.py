"""Stage-by-stage HIP-vs-oracle error report (diagnostic; the asserting versions live in tests/)."""
import os, sys, time, traceback
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd'))
from oracle import vipnerf_oracle as vo
from vipnerf_hip import ops, _lib

dev = torch.device('cuda:0')
GOLD = os.path.join(ROOT, 'tests', 'golden')


def load(name):
    return {k: v for k, v in np.load(os.path.join(GOLD, name + '.npz')).items()}


def err(a, b):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    d = np.abs(a - b)
    return 'max_abs %.3e  max_rel %.3e  (ref max %.3e)%s' % (d.max(), (d / (np.abs(b) + 1e-6)).max(), np.abs(b).max(),
                                                          '  NAN!' if not np.isfinite(a).all() else '')


def cu(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev) if isinstance(x, np.ndarray) else x.to(dev)


def section(f):
    print('=' * 10, f.__name__)
    try:
        f()
    except Exception:
        traceback.print_exc()


def pack(params, level):
    names = ops.PARAM_ORDER
    return ops.pack_weights([cu(params[f'{level}_model.{n}']) for n in names])


def t_coarse():
    b = vo.synthetic_batch(64, 1, 'fern')
    rng = vo.synthetic_rng(64, 64, 128, 2)
    for tr in (None, rng['t_rand']):
        zo = vo.coarse_depths(b['near_ndc'], b['far_ndc'], 64, tr)
        zh = ops.coarse_depths(cu(b['near_ndc']), cu(b['far_ndc']), 64, cu(tr) if tr is not None else None)
        print('coarse z (jitter=%s):' % (tr is not None), err(zh, zo), 'exact=%s' % bool((zh.cpu() == zo).all()))
    b = vo.synthetic_batch(64, 1, 'dtu', nf=3)
    zo = vo.coarse_depths(b['near'], b['far'], 64, rng['t_rand'])
    zh = ops.coarse_depths(cu(b['near']), cu(b['far']), 64, cu(rng['t_rand']))
    print('coarse z dtu:', err(zh, zo), 'exact=%s' % bool((zh.cpu() == zo).all()))


def t_sample():
    g = load('f1_sample_pdf')
    z = np.zeros((g['bins'].shape[0], 64), np.float32)  # rebuild z whose mids are the bins is not possible; use oracle path
    gg = np.random.default_rng(5)
    n = 128
    zc = np.sort(gg.uniform(0, 1, size=(n, 64)).astype(np.float32), axis=1)
    w = (gg.random((n, 64), dtype=np.float32) ** 4)
    w[:8] = 0
    u = gg.random((n, 128), dtype=np.float32)
    for uu in (u, None):
        zf_o, inds_o, s_o = vo.fine_depths(torch.from_numpy(zc), torch.from_numpy(w), 128, torch.from_numpy(uu) if uu is not None else None)
        zf, inds, s = ops.sample_fine(cu(zc), cu(w), 128, cu(uu) if uu is not None else None)
        mism = (inds.cpu().long() != inds_o).sum().item()
        print('sample_fine det=%s: inds mismatches %d / %d; samples' % (uu is None, mism, inds_o.numel()), err(s, s_o), '; z_fine', err(zf, zf_o))


def t_mlp():
    for V in (1, 2):
        g = load(f'f2_mlp_v{V}')
        params = vo.init_params(int(g['seed']), levels=('coarse',))
        pk = pack(params, 'coarse')
        for mode, noise in (('train', g['noise']), ('eval', None)):
            o = ops.mlp_forward(pk, cu(g['pts']), cu(g['view_dirs']), cu(g['view_dirs2']), cu(noise) if noise is not None else None, 1.0)
            torch.cuda.synchronize()
            print(f'mlp V={V} {mode}: sigma', err(o['sigma'], g[f'sigma_{mode}']))
            print(f'mlp V={V} {mode}: rgb  ', err(o['rgb'], g[f'rgb_{mode}']))
            print(f'mlp V={V} {mode}: vis  ', err(o['visibility'], g[f'vis_{mode}']))
            print(f'mlp V={V} {mode}: vis2 ', err(o['visibility2'], g[f'vis2_{mode}']))


def t_composite():
    for scene in ('fern', 'dtu'):
        g = load(f'f3_composite_{scene}')
        ndc = bool(g['ndc'])
        V = g['vis2'].shape[-1]
        cfg = ops.make_config(ndc, 64, 0, V, False)
        b = {'rays_o': cu(g['rays_o']), 'rays_d': cu(g['rays_d']), 'view_dirs': cu(g['rays_d']),
             'rays_o2': cu(g['rays_o2'])}
        n = g['z'].shape[0]
        if ndc:
            b.update(rays_o_ndc=cu(g['rays_o_ndc']), rays_d_ndc=cu(g['rays_d_ndc']), near_ndc=torch.zeros(n, device=dev), far_ndc=torch.ones(n, device=dev))
        else:
            b.update(near=torch.zeros(n, device=dev), far=torch.ones(n, device=dev))
        lvl = ops.composite(cfg, b, cu(g['z']), cu(g['sigma']), cu(g['rgb']), cu(g['vis2']))
        torch.cuda.synchronize()
        for k, kk in (('rgb', 'rgb'), ('acc', 'acc'), ('alpha', 'alpha'), ('visibility', 'visibility'), ('weights', 'weights'),
                      ('depth', 'depth'), ('depth_var', 'depth_var'), ('vis2', 'visibility2'), ('depth_ndc', 'depth_ndc'), ('depth_var_ndc', 'depth_var_ndc')):
            if 'out_' + kk in g and k in lvl:
                print(f'composite {scene} {k}:', err(lvl[k], g['out_' + kk]))


def batch_to_dev(b):
    o = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    if 'poses' in b:
        o['rays_o2'] = vo.secondary_origins(b['poses'], b['pixel_id'][:, 0].long(), int(b['num_frames'])).to(dev)
    return o


KEYMAP = {'rgb': 'rgb', 'acc': 'acc', 'depth': 'depth', 'depth_var': 'depth_var', 'depth_ndc': 'depth_ndc',
          'depth_var_ndc': 'depth_var_ndc', 'visibility2': 'vis2', 'z_vals': 'z_vals', 'alpha': 'alpha',
          'visibility': 'visibility', 'weights': 'weights', 'raw_sigma': 'raw_sigma', 'raw_rgb': 'raw_rgb',
          'raw_visibility': 'raw_vis', 'raw_visibility2': 'raw_vis2'}


def cmp_levels(tag, coarse, fine, g):
    for lv, d in (('coarse', coarse), ('fine', fine)):
        if d is None:
            continue
        for rk, hk in KEYMAP.items():
            gk = f'out_{rk}_{lv}'
            if gk in g and hk in d:
                ref = g[gk]
                print(f'{tag} {rk}_{lv}:', err(d[hk].reshape(ref.shape), ref))


def t_render_eval():
    g = load('f4_eval_fern')
    b = vo.synthetic_batch(int(g['n']), int(g['seed_batch']), scene='fern', nf=2)
    params = vo.init_params(int(g['seed_params']), scale=float(g['scale_params']))
    pc, pf = pack(params, 'coarse'), pack(params, 'fine')
    cfg = ops.make_config(True, 64, 128, 1, False)
    c, f, ex = ops.render_forward(cfg, batch_to_dev(b), None, pc, pf)
    torch.cuda.synchronize()
    cmp_levels('eval', c, f, g)


def t_render_train():
    for tag in ('llff', 'realestate', 'dtu'):
        g = load(f'f5_train_{tag}')
        b = vo.synthetic_batch(int(g['n']), int(g['seed_batch']), scene=str(g['scene']), nf=int(g['nf']), n_sparse=int(g['n_sparse']))
        params = vo.init_params(int(g['seed_params']), scale=float(g['scale_params']))
        pc, pf = pack(params, 'coarse'), pack(params, 'fine')
        rng = {k[4:]: cu(v) for k, v in g.items() if k.startswith('rng_')}
        cfg = ops.make_config(b['ndc'], 64, 128, int(g['nf']) - 1, True, noise_std=1.0)
        c, f, ex = ops.render_forward(cfg, batch_to_dev(b), rng, pc, pf)
        torch.cuda.synchronize()
        cmp_levels('train-' + tag, c, f, g)


def t_speed():
    n = 4096
    b = vo.synthetic_batch(n, 7, scene='fern', nf=2)
    params = vo.init_params(3)
    pc, pf = pack(params, 'coarse'), pack(params, 'fine')
    bd = batch_to_dev(b)
    for train in (False, True):
        cfg = ops.make_config(True, 64, 128, 1, train, noise_std=1.0 if train else 0.0, save_acts=train)
        acts = None
        if train:
            ab, bb = ops.query_workspace(cfg, n)
            print('acts GB %.2f  bwd GB %.2f' % (ab / 1e9, bb / 1e9))
            acts = torch.empty(ab // 4, dtype=torch.float32, device=dev)
        rng = {'seed': 1, 'offset': 0} if train else None
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.time()
            ops.render_forward(cfg, bd, rng, pc, pf, acts)
            torch.cuda.synchronize(); dt = time.time() - t0
            fl = 630272 * 256 * 2 * n if train else 630272 * 256 * 2 * n
            print('forward train=%s: %.2f ms  -> %.1f TFLOP/s (algorithmic, V=1)' % (train, dt * 1e3, fl / dt / 1e12))


if __name__ == '__main__':
    print(torch.cuda.get_device_name(0))
    for f in (t_coarse, t_sample, t_mlp, t_composite, t_render_eval, t_render_train, t_speed):
        section(f)
