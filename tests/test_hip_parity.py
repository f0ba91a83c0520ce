"""Parity of the HIP path (through the C ABI / the drop-in module) against the CPU oracle and the golden
vectors captured from the reference.  Needs an MI355X: every test is marked gpu.

Tolerances (north_star: 1e-4 relative fp32, bit-exact sample indices):
  * stage-wise and teacher-forced end-to-end outputs: rtol 1e-4 with an absolute floor of 1e-5 * max|ref| (per-sample
    alphas of 1e-9 carry 6e-8 absolute rounding, which is not 1e-4 relative to themselves);
  * searchsorted indices, coarse depths, sorted fine depths given identical inputs: bit-exact;
  * gradients: 1e-3 of the tensor's max magnitude (they are sums over ~1e5 fp32 products in a different order).
The hierarchical sampler is ill-conditioned (the reference in fp32 vs itself in fp64 differs by 3e-4
in z_vals_fine), so fine-level parity is asserted with the reference's fine depths fed in (cfg.given_z_fine) and the
free-running difference is only reported.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd', 'src'))

from oracle import vipnerf_oracle as vo  # noqa: E402


def load(name):
    return {k: v for k, v in np.load(os.path.join(GOLD, name + '.npz')).items()}


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    return torch.device('cuda:0')


def cu(x, dev):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    return x.to(dev)


def assert_close(a, b, rtol=1e-4, floor=1e-5, what=''):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, np.float64)
    a = a.reshape(b.shape)
    assert np.isfinite(a).all(), f'{what}: non-finite values'
    tol = rtol * np.abs(b) + floor * max(np.abs(b).max(), 1e-30)
    bad = np.abs(a - b) > tol
    assert not bad.any(), f'{what}: {bad.sum()} / {bad.size} beyond tolerance; max abs err {np.abs(a - b).max():.3e} (ref max {np.abs(b).max():.3e})'


def hip_ops():
    from vipnerf_hip import ops
    return ops


def pack(params, level, dev):
    ops = hip_ops()
    return ops.pack_weights([cu(params[f'{level}_model.{n}'], dev) for n in ops.PARAM_ORDER])


def batch_to_dev(b, dev):
    o = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    if 'poses' in b:
        o['rays_o2'] = vo.secondary_origins(b['poses'], b['pixel_id'][:, 0].long(), int(b['num_frames'])).to(dev)
    return o


KEYMAP = {'rgb': 'rgb', 'acc': 'acc', 'depth': 'depth', 'depth_var': 'depth_var', 'depth_ndc': 'depth_ndc',
          'depth_var_ndc': 'depth_var_ndc', 'visibility2': 'vis2', 'z_vals': 'z_vals', 'alpha': 'alpha',
          'visibility': 'visibility', 'weights': 'weights', 'raw_sigma': 'raw_sigma', 'raw_rgb': 'raw_rgb',
          'raw_visibility': 'raw_vis', 'raw_visibility2': 'raw_vis2'}


TIE_ULP = 4.0          # a differing free-running index must have its draw within this many fp32 ulp of the separating CDF entry
GRAD_LOG = []          # (what, rel L2, max err / max|g|, bound) of every grad_close call of the session; VIPNERF_GRAD_LOG=<file> appends them there


def grad_close(a, ref, what, scale=0.0, l2_tol=None, rows=0):
    """Parameter gradients against the oracle's / the reference's: relative L2 error and largest element error over the largest element.

    Bounds = about twice what is MEASURED (the first GPU pass of round 6 recorded every tensor of every test:
    profiles/r06_grad_close_measured.txt).  The measurements are bimodal.  Most cases sit at pure summation-order rounding, 1e-6 .. 7e-5:
    they get the default, 1.5e-4 (6e-4 at >= 4096 rows: sums over ~1e6 products, measured 2.8e-4).  A minority sits at 1e-4 .. 2e-3 -- ALL
    tensors of a level at once, by about the same relative amount: one sample of the batch whose contribution changes sign or vanishes.  The
    path has real kinks: the visibility loss is an L1 term whose seed is sign(T^ - T) (VisibilityLoss01.py:60-66; T = 1 - 1e-7 against
    T^ = 1 flips it), and a ReLU whose pre-activation is at rounding level lets a point through or not.  With 16 .. 1024 rays such an event
    is 1e-4 .. 2e-3 of a tensor's norm; those test families pass their own bound explicitly, each with its measured value beside it.  An
    injected relative error of 1e-3 fails the default, the 4096-row and the 1024-ray bounds (tests/test_tolerances_cpu.py); the 16-bit
    modes have their own classes.  The element bound is 10 x the L2 bound."""
    if l2_tol is None:
        l2_tol = 6e-4 if rows >= 4096 else 1.5e-4
    a, ref = np.asarray(a, np.float64).reshape(-1), np.asarray(ref, np.float64).reshape(-1)
    assert np.isfinite(a).all(), what
    nrm = max(np.linalg.norm(ref), scale * np.sqrt(ref.size), 1e-30)
    l2 = np.linalg.norm(a - ref) / nrm
    mx = np.abs(a - ref).max() / max(np.abs(ref).max(), scale, 1e-30)
    GRAD_LOG.append((what, l2, mx, l2_tol))
    if os.environ.get('VIPNERF_GRAD_LOG'):
        with open(os.environ['VIPNERF_GRAD_LOG'], 'a') as f:
            f.write('%-70s l2 %.3e max %.3e bound %.1e ratio %.3f\n' % (what, l2, mx, l2_tol, l2 / l2_tol))
    if os.environ.get('VIPNERF_GRAD_NOASSERT'):          # (a measurement pass: record every tensor, decide the bounds afterwards)
        return
    assert l2 <= l2_tol and mx <= 10 * l2_tol, f'{what}: rel L2 err {l2:.3e} (bound {l2_tol:.1e}), max err / max|g| {mx:.3e}'


def cdf_tie_gaps(w_coarse, u, inds, ref_inds):
    """north_star says "bit-exact for sample indices".  On identical inputs the sampler IS bit-exact (F1, the stage sweeps); free-running, the
    CDF inherits the coarse MLP's last-bit rounding, so an index may differ only where the draw u sits ON a CDF value: for every differing
    (ray, draw) the distance from u to the oracle's CDF entry that separates the two answers, in units of u's fp32 spacing.  searchsorted(cdf,
    u, right=True) counts the entries <= u, so answers k and k + 1 differ exactly on which side of u entry k lies.
    w_coarse (N, Sc) the ORACLE's coarse weights; u (N, J); inds / ref_inds (N, J).  -> float array of gaps in ulp (empty when all agree)."""
    w = w_coarse[:, 1:-1].float() + 1e-5
    pdf = w / torch.sum(w, dim=-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, dim=-1)], dim=-1).double()
    rows, cols = torch.nonzero(inds != ref_inds, as_tuple=True)
    if rows.numel() == 0:
        return np.zeros(0)
    edge = torch.minimum(inds[rows, cols], ref_inds[rows, cols]).clamp(max=cdf.shape[1] - 1)
    uu = u[rows, cols].float()
    gap = (cdf[rows, edge] - uu.double()).abs().numpy()
    return gap / np.spacing(np.maximum(uu.numpy(), np.float32(2.0 ** -24)))


# ------------------------------------------------------------------------------------------------ stage-wise
def test_coarse_depths_bit_exact(dev):
    ops = hip_ops()
    rng = vo.synthetic_rng(96, 64, 128, 2)
    for scene, nf in (('fern', 2), ('dtu', 3)):
        b = vo.synthetic_batch(96, 1, scene, nf=nf)
        near, far = (b['near_ndc'], b['far_ndc']) if b['ndc'] else (b['near'], b['far'])
        for tr in (None, rng['t_rand']):
            zo = vo.coarse_depths(near, far, 64, tr)
            zh = ops.coarse_depths(cu(near, dev), cu(far, dev), 64, cu(tr, dev) if tr is not None else None)
            assert torch.equal(zh.cpu(), zo), f'{scene} jitter={tr is not None}'


def test_sample_pdf_golden_indices_bit_exact(dev):
    """F1: the reference's sample_pdf on adversarial rows (flat cdf, single spike).  bins are arbitrary mid points,
    so drive the kernel through z_coarse whose mids equal them is impossible in general; instead check the oracle
    (already pinned to F1) against the kernel on the same rows."""
    ops = hip_ops()
    g = load('f1_sample_pdf')
    n = g['weights'].shape[0]
    rs = np.random.default_rng(3)
    zc = np.sort(rs.uniform(0, 1, size=(n, 64)).astype(np.float32), axis=1)
    w = np.zeros((n, 64), np.float32)
    w[:, 1:-1] = g['weights']                                   # the fixture's weight rows (incl. all-zero rows)
    for u in (g['u'], None):
        zf_o, inds_o, s_o = vo.fine_depths(torch.from_numpy(zc), torch.from_numpy(w), 128,
                                           torch.from_numpy(u) if u is not None else None)
        zf, inds, s = ops.sample_fine(cu(zc, dev), cu(w, dev), 128, cu(u, dev) if u is not None else None)
        assert torch.equal(inds.cpu().long(), inds_o), 'sample indices must be bit-exact'
        assert torch.equal(s.cpu(), s_o), 'samples'
        assert torch.equal(zf.cpu(), zf_o), 'sorted merge'


def test_sample_fine_random_rows(dev):
    ops = hip_ops()
    rs = np.random.default_rng(5)
    n = 512
    zc = np.sort(rs.uniform(0, 1, size=(n, 64)).astype(np.float32), axis=1)
    w = rs.random((n, 64), dtype=np.float32) ** 4
    u = rs.random((n, 128), dtype=np.float32)
    for uu in (u, None):
        zf_o, inds_o, s_o = vo.fine_depths(torch.from_numpy(zc), torch.from_numpy(w), 128,
                                           torch.from_numpy(uu) if uu is not None else None)
        zf, inds, s = ops.sample_fine(cu(zc, dev), cu(w, dev), 128, cu(uu, dev) if uu is not None else None)
        assert torch.equal(inds.cpu().long(), inds_o)
        assert torch.equal(zf.cpu(), zf_o)
        assert bool((zf[:, 1:] >= zf[:, :-1]).all())


@pytest.mark.parametrize('V', [1, 2])
def test_mlp_forward_golden(dev, V):
    ops = hip_ops()
    g = load(f'f2_mlp_v{V}')
    pk = pack(vo.init_params(int(g['seed']), levels=('coarse',)), 'coarse', dev)
    for mode, noise in (('train', g['noise']), ('eval', None)):
        o = ops.mlp_forward(pk, cu(g['pts'], dev), cu(g['view_dirs'], dev), cu(g['view_dirs2'], dev),
                            cu(noise, dev) if noise is not None else None, 1.0)
        assert_close(o['sigma'], g[f'sigma_{mode}'], what=f'sigma {mode}')
        assert_close(o['rgb'], g[f'rgb_{mode}'], what=f'rgb {mode}')
        assert_close(o['visibility'], g[f'vis_{mode}'], what=f'vis {mode}')
        assert_close(o['visibility2'], g[f'vis2_{mode}'], what=f'vis2 {mode}')


def test_mlp_forward_ragged_and_empty(dev):
    """point counts that are not a multiple of the 128-point workgroup tile, and zero points"""
    ops = hip_ops()
    params = vo.init_params(9, levels=('coarse',))
    pk = pack(params, 'coarse', dev)
    p = vo.params_to_torch(params)
    rs = np.random.default_rng(1)
    for P in (1, 37, 129, 300):
        pts = torch.from_numpy(rs.uniform(-1, 1, size=(P, 3)).astype(np.float32))
        vd = torch.nn.functional.normalize(torch.from_numpy(rs.standard_normal((P, 3)).astype(np.float32)), dim=-1)
        ref = vo.mlp_forward(p, 'coarse', pts, vd, None, None)
        o = ops.mlp_forward(pk, pts.to(dev), vd.to(dev))
        assert_close(o['rgb'], ref['rgb'], what=f'rgb P={P}')
        assert_close(o['sigma'], ref['sigma'], what=f'sigma P={P}')
    o = ops.mlp_forward(pk, torch.zeros(0, 3, device=dev), torch.zeros(0, 3, device=dev))
    assert o['rgb'].shape == (0, 3)


@pytest.mark.parametrize('scene', ['fern', 'dtu'])
def test_composite_golden(dev, scene):
    ops = hip_ops()
    g = load(f'f3_composite_{scene}')
    ndc = bool(g['ndc'])
    V = g['vis2'].shape[-1]
    n = g['z'].shape[0]
    cfg = ops.make_config(ndc, 64, 0, V, False)
    b = {'rays_o': cu(g['rays_o'], dev), 'rays_d': cu(g['rays_d'], dev), 'view_dirs': cu(g['rays_d'], dev),
         'rays_o2': cu(g['rays_o2'], dev)}
    z = torch.zeros(n, device=dev)
    if ndc:
        b.update(rays_o_ndc=cu(g['rays_o_ndc'], dev), rays_d_ndc=cu(g['rays_d_ndc'], dev), near_ndc=z, far_ndc=z + 1)
    else:
        b.update(near=z, far=z + 1)
    lvl = ops.composite(cfg, b, cu(g['z'], dev), cu(g['sigma'], dev), cu(g['rgb'], dev), cu(g['vis2'], dev))
    for hk, rk in (('rgb', 'rgb'), ('acc', 'acc'), ('alpha', 'alpha'), ('visibility', 'visibility'), ('weights', 'weights'),
                   ('depth', 'depth'), ('depth_var', 'depth_var'), ('vis2', 'visibility2'), ('depth_ndc', 'depth_ndc'),
                   ('depth_var_ndc', 'depth_var_ndc')):
        if 'out_' + rk in g:
            assert_close(lvl[hk], g['out_' + rk], what=f'{scene} {rk}')


# ------------------------------------------------------------------------------------------------ the module
def make_model(dev, ndc, params=None, n_fine=128, losses=True, sparse=False):
    from models.ModelFactory import get_model
    mlp = lambda ns: {'num_samples': ns, 'netdepth': 8, 'netwidth': 256, 'points_positional_encoding_degree': 10,
                      'views_positional_encoding_degree': 4, 'use_view_dirs': True, 'view_dependent_rgb': True,
                      'predict_visibility': True}
    cfg = {'data_loader': {'ndc': ndc},
           'model': {'name': 'VipNeRFHip01', 'coarse_mlp': mlp(64), 'chunk': 4096, 'netchunk': 16384, 'lindisp': False,
                     'perturb': True, 'raw_noise_std': 1.0, 'white_bkgd': False},
           'losses': [{'name': 'MSEHip01', 'weight': 1}, {'name': 'VisibilityLossHip01', 'weight': 0.1},
                      {'name': 'VisibilityPriorLossHip01', 'iter_weights': {'0': 0, '30000': 0.001}}],
           'device': [0]}
    if n_fine:
        cfg['model']['fine_mlp'] = mlp(n_fine)
    if sparse:
        cfg['losses'].append({'name': 'SparseDepthMSEHip01', 'weight': 0.1})
    model = get_model(cfg, None)
    if params is not None:
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()}, strict=True)
    return model.to(dev), cfg


def ref_batch(b, dev, iter_num):
    rb = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items() if k not in ('poses', 'ndc')}
    rb['common_data'] = {'poses': b['poses'][None].clone().to(dev)}
    rb['iter_num'] = iter_num
    return rb


def test_eval_render_golden(dev):
    g = load('f4_eval_fern')
    b = vo.synthetic_batch(int(g['n']), int(g['seed_batch']), scene='fern', nf=2)
    params = vo.init_params(int(g['seed_params']), scale=float(g['scale_params']), sigma_bias=float(g['sigma_bias']))
    model, _ = make_model(dev, True, params)
    model.eval()
    with torch.no_grad():
        plain = model(ref_batch(b, dev, 0))
        assert sorted(plain.keys()) == sorted(str(k) for k in g['keys_plain'])     # the reference's eval key set
        inds = model.last_extras['sample_inds']
        assert inds.min() >= 1 and inds.max() <= 63
        # FREE-RUNNING parity, measured (north_star: "bit-exact for sample indices"): the inverse-CDF indices of the free-running HIP
        # render -- its own coarse MLP, weights, cdf -- against the indices the reference's own searchsorted returned in this render
        # Measured (tools/free_running_diag.py): every index of the columns u < 1 agrees; the eval draw u = linspace(0, 1, 128) ends with
        # u = 1.0 EXACTLY, and searchsorted(cdf, 1.0, right=True) there is decided by whether cdf[-1] -- a 63-term fp32 cumulative sum -- came
        # out as 1 - 1 ulp, 1 or 1 + 1 ulp: a third of the rays differ by one in that single column (15 of 48 here), with no effect on the
        # sample (t = 1 in one bin or t = 0 in the next: the same depth), which the fine depths below pin to 2e-6.
        ref_inds = torch.from_numpy(g['plain_sample_inds'].astype(np.int64))
        same = inds.cpu().long() == ref_inds
        agree, agree_open = float(same.float().mean()), float(same[:, :-1].float().mean())
        print(f'free-running sample indices equal to the reference\'s: {agree:.5f} of {ref_inds.numel()} ({agree_open:.5f} over the columns u < 1)')
        assert agree_open >= 0.9995 and agree >= 0.99, (agree_open, agree)
        assert int((inds.cpu().long() - ref_inds).abs().max()) <= 1
        plain_raw = model(ref_batch(b, dev, 0), retraw=True)
        dz = float((plain_raw['z_vals_fine'].cpu() - torch.from_numpy(g['out_z_vals_fine'])).abs().max())
        print(f'free-running fine depths: max abs difference to the reference\'s {dz:.2e} (NDC, [0, 1])')
        assert dz <= 2e-6
        model.injected_z_fine = cu(g['out_z_vals_fine'], dev)
        out = model(ref_batch(b, dev, 0), retraw=True, sec_views_vis=True)
        model.injected_z_fine = None
    for lv in ('coarse', 'fine'):
        for rk in KEYMAP:
            gk = f'out_{rk}_{lv}'
            if gk in g:
                assert_close(out[f'{rk}_{lv}'], g[gk], what=f'{rk}_{lv}')
    # free-running fine pass against the claim (north_star: rendered RGB within 1e-4 of the reference): NO ray beyond 1e-4
    e = (plain['rgb_fine'].cpu() - torch.from_numpy(g['plain_rgb_fine'])).abs().max(dim=-1).values
    err, beyond = float(e.max()), float((e > 1e-4).float().mean())
    print(f'free-running rgb_fine: max abs err {err:.3e}, rays beyond 1e-4: {beyond:.4f}')
    assert err <= 1e-5 and beyond == 0.0, (err, beyond)          # measured 3.6e-7 (48 rays); the statistics at 1024 rays: tests/test_hip_freerun.py
    assert_close(plain['rgb_coarse'], g['plain_rgb_coarse'], what='plain rgb_coarse')


@pytest.mark.parametrize('tag', ['llff', 'realestate', 'dtu', 'dtu4wl'])
def test_train_step_golden(dev, tag):
    """F5: forward outputs, the losses at iter 0 / 40000, parameter gradients and one Adam step, against the
    reference.  RNG draws and fine depths are the reference's (teacher forcing)."""
    from loss_functions.LossComputerHip01 import LossComputerHip
    g = load(f'f5_train_{tag}')
    n_sparse = int(g['n_sparse'])
    b = vo.synthetic_batch(int(g['n']), int(g['seed_batch']), scene=str(g['scene']), nf=int(g['nf']), n_sparse=n_sparse)
    params = vo.init_params(int(g['seed_params']), scale=float(g['scale_params']))
    model, cfg = make_model(dev, b['ndc'], params, sparse=n_sparse > 0)
    cfg['model'].update(white_bkgd=bool(g.get('white_bkgd', False)), lindisp=bool(g.get('lindisp', False)))   # 'dtu4wl': nf = 4, both on
    model.train()
    lossc = LossComputerHip(cfg)
    opt = torch.optim.Adam(model.parameters(), lr=5e-4, betas=(0.9, 0.999))
    model.injected_rng = {k[4:]: cu(v, dev) for k, v in g.items() if k.startswith('rng_')}
    model.injected_z_fine = cu(g['out_z_vals_fine'], dev)
    out = model(ref_batch(b, dev, 40000))
    for lv in ('coarse', 'fine'):
        for rk in KEYMAP:
            gk = f'out_{rk}_{lv}'
            if gk in g:
                assert_close(out[f'{rk}_{lv}'], g[gk], what=f'{tag} {rk}_{lv}')
    names = {'MSEHip01': 'MSE01', 'VisibilityLossHip01': 'VisibilityLoss01', 'VisibilityPriorLossHip01': 'VisibilityPriorLoss01',
             'SparseDepthMSEHip01': 'SparseDepthMSE01', 'TotalLoss': 'TotalLoss'}
    l40k = lossc.compute_losses(ref_batch(b, dev, 40000), out)
    for k, v in l40k.items():
        val = v['loss_value'] if isinstance(v, dict) else v
        assert_close(val, g[f'l40k_{names[k]}'], rtol=1e-4, floor=1e-6, what=f'{tag} loss {k}')
    assert not any(k.startswith('_') for k in out), 'the output dict must hold the reference\'s keys only (Trainer01.py:147-172)'
    l0 = lossc.compute_losses(ref_batch(b, dev, 0), dict(out))
    assert_close(l0['TotalLoss'], g['l0_TotalLoss'], rtol=1e-4, floor=1e-6, what=f'{tag} TotalLoss iter 0')
    opt.zero_grad(set_to_none=True)
    l40k['TotalLoss'].backward()

    def digest(t):
        f = t.detach().reshape(-1).double().cpu()
        nn = f.numel()
        idx = (torch.arange(192, dtype=torch.long) * 7919) % nn
        return torch.cat([f.sum()[None], f.norm()[None], f[:64] if nn >= 64 else torch.cat([f, f.new_zeros(64 - nn)]), f[idx]]).numpy()

    for k, p in model.named_parameters():
        gd, dg = g['gdig_' + k], digest(p.grad)
        assert np.isfinite(dg).all(), k
        np.testing.assert_allclose(dg[1], gd[1], rtol=1e-3, atol=1e-9, err_msg=f'{tag} |grad| of {k}')
        gtol = {'realestate': 2.5e-3, 'dtu4wl': 3e-4}.get(tag)      # measured: llff 1.5e-5, dtu 3.9e-6, dtu4wl 1.3e-4, realestate 1.1e-3 (16 + 16 rays: one kink event)
        grad_close(dg[2:], gd[2:], f'{tag} grad samples of {k}', scale=max(abs(gd[1]) / np.sqrt(p.numel()), 1e-12), l2_tol=gtol)
        if 'grad_' + k in g:
            grad_close(p.grad.cpu().numpy(), g['grad_' + k], f'{tag} grad of {k}', l2_tol=gtol)
    opt.step()
    # The first Adam update is -lr g / (|g| + 1e-8): +-5e-4 whatever the gradient's size, so a bound of one update proves nothing.  Elements
    # whose reference gradient is above rounding level (|g| > 1e-6: the update's sign and size are then determined to < 1e-8) must land
    # on the reference's parameter to 1e-6; only the others (an update whose gradient is at rounding level may flip sign) get the loose bound.
    tight = total = 0
    for k, p in model.named_parameters():
        after, ref_after, gref = digest(p)[2:], g['adig_' + k][2:], g['gdig_' + k][2:]
        firm = np.abs(gref) > 1e-6
        np.testing.assert_allclose(after[firm], ref_after[firm], rtol=0, atol=1e-6, err_msg=f'{tag} Adam step of {k}')
        np.testing.assert_allclose(after[~firm], ref_after[~firm], rtol=0, atol=1.1e-3, err_msg=f'{tag} Adam step of {k} (rounding-level gradients)')
        tight += int(firm.sum()); total += firm.size
    assert tight > 0.5 * total, f'{tag}: only {tight} of {total} sampled parameters have a gradient above rounding level'


@pytest.mark.parametrize('tag', ['llff', 'dtu'])
def test_train_step_free_running_indices_golden(dev, tag):
    """A TRAINING forward left free-running (the reference's recorded random draws injected, but NOT its fine depths): the inverse-CDF
    indices the HIP path computes from its own coarse pass against the indices the reference's searchsorted returned (north_star:
    bit-exact for sample indices), and the fine colour against the reference's."""
    g = load(f'f5_train_{tag}')
    b = vo.synthetic_batch(int(g['n']), int(g['seed_batch']), scene=str(g['scene']), nf=int(g['nf']), n_sparse=int(g['n_sparse']))
    params = vo.init_params(int(g['seed_params']), scale=float(g['scale_params']))
    model, cfg = make_model(dev, b['ndc'], params)
    model.train()
    model.injected_rng = {k[4:]: cu(v, dev) for k, v in g.items() if k.startswith('rng_')}
    with torch.no_grad():
        out = model(ref_batch(b, dev, 40000))
    ref_inds = torch.from_numpy(g['sample_inds'].astype(np.int64))
    inds = model.last_extras['sample_inds'].cpu().long()
    agree = float((inds == ref_inds).float().mean())
    e = (out['rgb_fine'].cpu() - torch.from_numpy(g['out_rgb_fine'])).abs().max(dim=-1).values
    print(f'{tag}: free-running training indices equal to the reference\'s: {agree:.5f} of {ref_inds.numel()}; rgb_fine max abs err {float(e.max()):.3e}, '
          f'rays beyond 1e-4: {float((e > 1e-4).float().mean()):.4f}')
    assert agree >= 0.9999
    assert float(e.max()) <= 1e-4, 'north_star: rendered RGB within 1e-4 of the reference (measured 8.2e-5 / 2.4e-5: no ray beyond)'


def test_gradients_arrive_in_one_flat_buffer(dev):
    """With .grad dropped beforehand, the 48 parameter gradients of a step are consecutive views of ONE buffer
    (no per-tensor fill / add / copy; one collective reduces it), and they equal the accumulate-into-bucket path."""
    from loss_functions.LossComputerHip01 import LossComputerHip
    from vipnerf_hip import dist as vdist
    b = vo.synthetic_batch(384, 21, scene='fern', nf=2)
    params = vo.init_params(9, scale=1.6)
    model, cfg = make_model(dev, b['ndc'], params)
    model.train()
    lossc = LossComputerHip(cfg)
    rng = {k: cu(v.numpy(), dev) for k, v in vo.synthetic_rng(384, 64, 128, 4).items()}
    bucket = vdist.FlatGradBucket(model.parameters())

    def run():
        model.injected_rng = rng
        rb = ref_batch(b, dev, 40000)
        lossc.compute_losses(rb, model(rb))['TotalLoss'].backward()

    bucket.zero()
    run()
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))
    ref = bucket.flat.clone()
    bucket.release()
    run()
    flat = bucket.adopted()
    assert flat is not None, 'gradients were copied instead of adopted'
    assert flat.numel() == 1191946 and flat.data_ptr() == bucket.params[0].grad.data_ptr()
    assert torch.equal(flat, ref), 'same kernels, same inputs: the two paths must agree bit for bit'


def test_short_rng_array_is_rejected(dev):
    """Injected random numbers must cover every ray: a short array is an error, not an out-of-bounds read."""
    b = vo.synthetic_batch(64, 3, scene='fern', nf=2)
    model, _ = make_model(dev, b['ndc'], vo.init_params(4, scale=1.6))
    model.train()
    model.injected_rng = {k: cu(v.numpy(), dev) for k, v in vo.synthetic_rng(32, 64, 128, 5).items()}
    with pytest.raises(RuntimeError, match='rng'):
        model(ref_batch(b, dev, 0))


def test_fused_total_loss_equals_generic_path(dev):
    """LossComputerHip's one-dot TotalLoss (all losses fused) against its generic weight * value accumulation:
    same per-loss values, same total, same parameter gradients (to fp32 rounding of the different summation order)."""
    from loss_functions.LossComputerHip01 import LossComputerHip
    for scene, nf, n_sparse in (('fern', 2, 0), ('realestate', 3, 128)):
        b = vo.synthetic_batch(256, 31, scene=scene, nf=nf, n_sparse=n_sparse)
        params = vo.init_params(11, scale=1.6)
        model, cfg = make_model(dev, b['ndc'], params, sparse=n_sparse > 0)
        model.train()
        rng = {k: cu(v.numpy(), dev) for k, v in vo.synthetic_rng(int(b['rays_o'].shape[0]), 64, 128, 5).items()}
        res = {}
        for fused in (True, False):
            lossc = LossComputerHip(cfg)
            lossc.fused_total = fused
            model.zero_grad(set_to_none=True)
            model.injected_rng = rng
            rb = ref_batch(b, dev, 40000)
            lv = lossc.compute_losses(rb, model(rb))
            lv['TotalLoss'].backward()
            res[fused] = ({k: float(v['loss_value'] if k != 'TotalLoss' else v) for k, v in lv.items()},
                          torch.cat([p.grad.flatten() for p in model.parameters()]).clone())
        assert res[True][0].keys() == res[False][0].keys()
        for k in res[True][0]:
            np.testing.assert_allclose(res[True][0][k], res[False][0][k], rtol=2e-6, atol=1e-9, err_msg=f'{scene} {k}')
        d = (res[True][1] - res[False][1]).norm() / res[False][1].norm()
        assert float(d) < 1e-6, f'{scene}: gradients differ by {float(d):.2e}'


def test_backward_all_cotangents_vs_oracle(dev):
    """Every differentiable output gets a random cotangent; parameter gradients vs the oracle's autograd."""
    n = 40
    for scene, nf in (('fern', 2), ('dtu', 3)):
        b = vo.synthetic_batch(n, 77, scene=scene, nf=nf)
        params = vo.init_params(78, scale=1.6)
        rng = vo.synthetic_rng(n, 64, 128, 79)
        p = vo.params_to_torch(params, requires_grad=True)
        cfg_o = {'ndc': b['ndc'], 'n_coarse': 64, 'n_fine': 128, 'noise_std': 1.0}
        ref = vo.render_rays(p, b, cfg_o, rng, train=True, sec_views=True)
        model, _ = make_model(dev, b['ndc'], params)
        model.train()
        model.injected_rng = {k: v.to(dev) for k, v in rng.items()}
        model.injected_z_fine = ref['z_vals_fine'].detach().to(dev)
        out = model(ref_batch(b, dev, 0))
        keys = ['rgb', 'acc', 'depth', 'visibility2', 'visibility', 'weights', 'alpha', 'raw_sigma', 'raw_rgb',
                'raw_visibility', 'raw_visibility2'] + (['depth_ndc'] if b['ndc'] else [])
        gen = torch.Generator().manual_seed(5)
        tot_o, tot_h = 0, 0
        for lv in ('coarse', 'fine'):
            for k in keys:
                kk = f'{k}_{lv}'
                ct = torch.randn(ref[kk].shape, generator=gen) / ref[kk].numel() ** 0.5
                tot_o = tot_o + (ref[kk] * ct).sum()
                tot_h = tot_h + (out[kk] * ct.to(dev)).sum()
        tot_o.backward()
        tot_h.backward()
        worst = 0.0
        for k, t in model.named_parameters():
            grad_close(t.grad.cpu().numpy(), p[k].grad.numpy(), f'{scene} {k}', l2_tol=1.2e-3)      # measured 5.5e-4 (fern: every coarse tensor by 4.3 .. 5.5e-4, one kink event)
            worst = max(worst, float((t.grad.cpu() - p[k].grad).norm() / p[k].grad.norm()))
        print(f'{scene}: worst relative L2 gradient error over 48 tensors {worst:.3e}')


# ------------------------------------------------------------------------------------------------ full size
def test_full_size_properties(dev):
    """BASELINE config 2 sizes (4096 rays x 64+128): determinism, ray-independence (a batch rendered whole equals
    the batch rendered in two halves, bit for bit), ordering and partition-of-unity properties."""
    n = 4096
    b = vo.synthetic_batch(n, 11, scene='fern', nf=2)
    model, _ = make_model(dev, True, vo.init_params(12, scale=1.6, sigma_bias=0.5))
    model.eval()
    with torch.no_grad():
        a1 = model(ref_batch(b, dev, 0), retraw=True, sec_views_vis=True)     # (the model unpacks common_data in place,
        a2 = model(ref_batch(b, dev, 0), retraw=True, sec_views_vis=True)     #  like the reference: fresh dict per call)
        halves = []
        for sl in (slice(0, n // 2), slice(n // 2, n)):
            hb = {k: (v[sl] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == n else v)
                  for k, v in ref_batch(b, dev, 0).items()}
            halves.append(model(hb, retraw=True, sec_views_vis=True))
    for k in a1:
        assert torch.equal(a1[k], a2[k]), f'non-deterministic: {k}'
        assert torch.equal(a1[k], torch.cat([halves[0][k], halves[1][k]], 0)), f'ray independence: {k}'
        assert torch.isfinite(a1[k]).all(), k
    zf = a1['z_vals_fine']
    assert bool((zf[:, 1:] >= zf[:, :-1]).all())
    for lv in ('coarse', 'fine'):
        assert bool((a1[f'acc_{lv}'] <= 1 + 1e-5).all()) and bool((a1[f'weights_{lv}'] >= 0).all())
        T = a1[f'visibility_{lv}']
        assert bool((T[:, 1:] <= T[:, :-1] * (1 + 1e-6) + 1e-9).all())                   # transmittance is non-increasing
        assert torch.allclose(a1[f'weights_{lv}'].sum(-1), a1[f'acc_{lv}'], rtol=1e-5, atol=1e-6)


def test_full_size_training_gradient_additivity(dev):
    """Gradients of a sum-type loss are additive over rays: grad(batch) == grad(first half) + grad(second half)."""
    n = 2048
    b = vo.synthetic_batch(n, 21, scene='fern', nf=2)
    model, _ = make_model(dev, True, vo.init_params(22, scale=1.6))
    model.train()
    rng = {k: v.to(dev) for k, v in vo.synthetic_rng(n, 64, 128, 23).items()}
    tgt = b['target_rgb'].to(dev)

    def run(sl):
        hb = {k: (v[sl] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == n else v)
              for k, v in ref_batch(b, dev, 0).items()}
        model.injected_rng = {k: v[sl] for k, v in rng.items()}
        model.zero_grad(set_to_none=True)
        out = model(hb)
        loss = ((out['rgb_fine'] - tgt[sl]) ** 2).sum() + ((out['rgb_coarse'] - tgt[sl]) ** 2).sum() \
            + (out['raw_visibility_fine'][..., 0] - out['visibility_fine']).abs().sum() * 1e-3
        loss.backward()
        return {k: p.grad.clone() for k, p in model.named_parameters()}
    whole = run(slice(0, n))
    h1, h2 = run(slice(0, n // 2)), run(slice(n // 2, n))
    for k in whole:
        s = h1[k] + h2[k]
        mx = whole[k].abs().max().item()
        assert (whole[k] - s).abs().max().item() <= 2e-4 * mx + 1e-9, k
