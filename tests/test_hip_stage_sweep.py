"""Seeded sweeps of the stages whose results are claimed BIT-EXACT (reference src/models/VipNeRF01.py:172-257: get_z_vals_coarse,
sample_pdf / searchsorted, the sorted merge) over what the goldens' fixed shapes do not reach: other sample counts, lindisp, jitter
values at the ends of [0, 1), weight rows that are all zero / a single spike / denormal-small / huge, inverse-CDF draws at exactly 0,
just below 1 and exactly on CDF knots, duplicate coarse depths.  The oracle (pinned to the reference by F1 / F4 / F5) is the judge."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vip-nerf_amd'), os.path.join(ROOT, 'vip-nerf_amd', 'src'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import vipnerf_oracle as vo          # noqa: E402  (the checker)
import test_hip_parity as tp                     # noqa: E402


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.mark.parametrize('seed', range(8))
def test_coarse_depths_sweep_bit_exact(dev, seed):
    ops = tp.hip_ops()
    rs = np.random.default_rng(900 + seed)
    n, S = int(rs.integers(1, 200)), int(32 * rs.integers(1, 9))
    near = rs.uniform(0.05, 2.0, size=(n, 1)).astype(np.float32)
    far = (near + rs.uniform(0.5, 150.0, size=(n, 1))).astype(np.float32)
    if seed % 3 == 0:
        near[:], far[:] = 0.0, 1.0                                           # the NDC bounds
    t = rs.random((n, S), dtype=np.float32)
    t[rs.random((n, S)) < 0.05] = 0.0
    t[rs.random((n, S)) < 0.05] = np.nextafter(np.float32(1), np.float32(0))
    for lindisp in (False, True):
        if lindisp and seed % 3 == 0:
            continue                                                        # 1 / near with near = 0
        for tr in (None, t):
            zo = vo.coarse_depths(torch.from_numpy(near), torch.from_numpy(far), S, torch.from_numpy(tr) if tr is not None else None,
                                  lindisp=lindisp)
            zh = ops.coarse_depths(tp.cu(near, dev), tp.cu(far, dev), S, tp.cu(tr, dev) if tr is not None else None, lindisp=lindisp)
            assert torch.equal(zh.cpu(), zo), f'seed {seed} S {S} lindisp {lindisp} jitter {tr is not None}'


def _weights(rs, n, S, kind):
    w = np.zeros((n, S), np.float32)
    if kind == 'smooth':
        w = rs.random((n, S), dtype=np.float32) ** 4
    elif kind == 'spike':
        w[np.arange(n), rs.integers(1, S - 1, size=n)] = rs.uniform(0.1, 1.0, size=n).astype(np.float32)
    elif kind == 'tiny':
        w = (rs.random((n, S), dtype=np.float32) * 1e-38).astype(np.float32)    # denormal-small: the 1e-5 offset dominates
    elif kind == 'huge':
        w = (rs.random((n, S), dtype=np.float32) * 1e30).astype(np.float32)
    elif kind == 'sparse':
        w = rs.random((n, S), dtype=np.float32) * (rs.random((n, S)) < 0.1)
    elif kind == 'zero':
        pass
    return w.astype(np.float32)


@pytest.mark.parametrize('kind', ['smooth', 'spike', 'tiny', 'huge', 'sparse', 'zero'])
@pytest.mark.parametrize('shape', [(64, 128), (32, 96), (96, 160), (128, 128), (32, 224)])
def test_sample_fine_sweep_bit_exact(dev, kind, shape):
    ops = tp.hip_ops()
    Sc, Sf = shape
    rs = np.random.default_rng(sum(map(ord, kind)) * 1000 + Sc * 7 + Sf)          # a fixed seed per case (hash() of a str is salted)
    n = 257
    zc = np.sort(rs.uniform(0, 1, size=(n, Sc)).astype(np.float32), axis=1)
    dup = rs.random(n) < 0.2                                                 # rows with runs of equal coarse depths
    zc[dup, 5:9] = zc[dup, 5:6]
    w = _weights(rs, n, Sc, kind)
    u = rs.random((n, Sf), dtype=np.float32)
    u[:, 0] = 0.0
    u[:, 1] = np.nextafter(np.float32(1), np.float32(0))
    # draws exactly on knots of this row's CDF (searchsorted right=True must step past them exactly as torch does)
    wt = torch.from_numpy(w[:, 1:-1]) + 1e-5
    cdf = torch.cumsum(wt / wt.sum(-1, keepdim=True), -1).numpy()
    u[:, 2:6] = cdf[:, rs.integers(0, Sc - 2, size=4)]
    u = np.clip(u, 0, np.nextafter(np.float32(1), np.float32(0))).astype(np.float32)
    for uu in (u, None):
        zf_o, inds_o, s_o = vo.fine_depths(torch.from_numpy(zc), torch.from_numpy(w), Sf, torch.from_numpy(uu) if uu is not None else None)
        zf, inds, s = ops.sample_fine(tp.cu(zc, dev), tp.cu(w, dev), Sf, tp.cu(uu, dev) if uu is not None else None)
        assert torch.equal(inds.cpu().long(), inds_o), f'{kind} {shape} det={uu is None}: sample indices'
        assert torch.equal(s.cpu(), s_o), f'{kind} {shape} det={uu is None}: samples'
        assert torch.equal(zf.cpu(), zf_o), f'{kind} {shape} det={uu is None}: sorted merge'
        assert zf.shape == (n, Sc + Sf) and bool((zf[:, 1:] >= zf[:, :-1]).all())


@pytest.mark.parametrize('seed', range(10))
def test_composite_sweep_vs_oracle(dev, seed):
    """volume_rendering + convert_depth_from_ndc (reference VipNeRF01.py:259-338) on explicit network outputs: sample counts 32..256,
    densities from exactly 0 over 1e-6 to 1e4 (alpha -> 1 in one sample), NDC depths that reach exactly 1 and repeat, white
    background, 0 / 1 / 3 secondary views."""
    from test_hip_round2 import assert_close_few_outliers
    ops = tp.hip_ops()
    rs = np.random.default_rng(1300 + seed)
    n, S = int(rs.integers(1, 150)), int(32 * rs.integers(1, 9))
    ndc, white, V = bool(seed % 2), bool(seed % 3 == 0), int([0, 1, 3][seed % 3])
    z = np.sort(rs.uniform(0.02 if not ndc else 0.0, 1.0 if ndc else 6.0, size=(n, S)).astype(np.float32), axis=1)
    if ndc:
        z[rs.random(n) < 0.3, -1] = 1.0                                       # far plane hit exactly: the 1e-3 guard of :329
        z[rs.random(n) < 0.2, -3:] = 1.0
    rep = rs.random(n) < 0.2
    z[rep, 4:7] = z[rep, 4:5]                                                # repeated depths (zero-length intervals), still sorted
    mag = rs.choice([0.0, 1e-6, 1e-2, 1.0, 50.0, 1e4], size=(n, S), p=[0.3, 0.1, 0.2, 0.25, 0.1, 0.05])
    sigma = (mag * rs.random((n, S))).astype(np.float32)
    sigma[rs.random(n) < 0.1] = 0.0                                          # empty rays: acc = 0, depth = 0 / (0 + 1e-10)
    rgb = rs.random((n, S, 3), dtype=np.float32)
    vis2 = rs.random((n, S, max(V, 1)), dtype=np.float32)[:, :, :V]
    rays_d = np.stack([rs.uniform(-0.5, 0.5, n), rs.uniform(-0.4, 0.4, n), -np.ones(n)], -1).astype(np.float32)
    rays_o = (rs.normal(size=(n, 3)) * 0.2).astype(np.float32)
    rays_o[:, 2] = rs.uniform(-0.1, 0.1, n).astype(np.float32)
    o_ndc, d_ndc = rays_o.copy(), (rs.normal(size=(n, 3)) * 0.7).astype(np.float32)
    t = torch.from_numpy
    net = {'sigma': t(sigma), 'rgb': t(rgb)}
    if V:
        net['visibility2'] = t(vis2)
    ref = vo.composite(net, t(z), t(d_ndc if ndc else rays_d), ndc, t(rays_o), t(rays_d), white_bkgd=white)
    cfg = ops.make_config(ndc, S, 0, V, False, white_bkgd=white)
    b = {'rays_o': tp.cu(rays_o, dev), 'rays_d': tp.cu(rays_d, dev), 'view_dirs': tp.cu(rays_d, dev),
         'rays_o2': tp.cu(np.zeros((n, max(V, 1), 3), np.float32)[:, :V], dev)}
    zero = torch.zeros(n, device=dev)
    if ndc:
        b.update(rays_o_ndc=tp.cu(o_ndc, dev), rays_d_ndc=tp.cu(d_ndc, dev), near_ndc=zero, far_ndc=zero + 1)
    else:
        b.update(near=zero, far=zero + 1)
    lvl = ops.composite(cfg, b, tp.cu(z, dev), tp.cu(sigma, dev), tp.cu(rgb, dev), tp.cu(vis2, dev) if V else None)
    for hk, rk in (('rgb', 'rgb'), ('acc', 'acc'), ('alpha', 'alpha'), ('visibility', 'visibility'), ('weights', 'weights'),
                   ('depth', 'depth'), ('depth_var', 'depth_var'), ('vis2', 'visibility2'), ('depth_ndc', 'depth_ndc'),
                   ('depth_var_ndc', 'depth_var_ndc')):
        if rk not in ref:
            continue
        a_, b_ = lvl[hk].cpu().double().numpy().reshape(ref[rk].shape), ref[rk].double().numpy()
        assert np.isfinite(a_).all(), f'seed {seed} {rk}: non-finite'
        if rk.startswith('depth') or rk == 'visibility2':                   # ratios by acc: a nearly empty ray is ill-conditioned
            assert_close_few_outliers(lvl[hk], ref[rk], 2e-4, f'seed {seed} {rk}', max_frac=max(0.02, 1.5 / n))
        else:
            tol = 1e-4 * np.abs(b_) + max(1e-5 * np.abs(b_).max(), 2e-7)    # alpha carries an ulp of 1.0 absolute (test_hip_sweep.py)
            assert np.all(np.abs(a_ - b_) <= tol), f'seed {seed} {rk}: max abs err {np.abs(a_ - b_).max():.3e} (ref max {np.abs(b_).max():.3e})'
