"""The split-precision MFMA modes (cfg.precision / configs['model']['hip_precision'] = 'bf16x3' | 'bf16x6' | 'fp16x3') against
the same golden vectors and the same tolerances as the fp32 path (tests/test_hip_parity.py): outputs within 1e-4
relative (+1e-5 of the tensor's max), gradients within 2e-3 relative L2."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd', 'src'))

from oracle import vipnerf_oracle as vo  # noqa: E402
import test_hip_parity as tp  # noqa: E402

# (the split-bf16 arithmetics bf16x3 / bf16x6 and the 'wide' lane layout were retired with ABI 5: test_retired_modes_are_refused below)
PRECS = ['fp16x3']
MODES = ['fp16x3', 'fp16x3h']
# relative-L2 tolerance on parameter gradients: fp32 grade (the same bar as the fp32 path)
GRAD_TOL = {'fp16x3': 1.5e-3, 'fp16x3h': 1.5e-3, 'fp32': 1.5e-3}      # the goldens in the fp32-grade split arithmetics, measured: fp16x3 1.9e-4, fp16x3h 6.2e-4 (old bound: 2e-3)


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.mark.parametrize('prec', PRECS)
@pytest.mark.parametrize('V', [1, 2])
def test_mlp_forward_golden(dev, prec, V):
    ops = tp.hip_ops()
    g = tp.load(f'f2_mlp_v{V}')
    params = vo.init_params(int(g['seed']), levels=('coarse',))
    pr = ops.PRECISIONS[prec]
    pk = ops.pack_weights([tp.cu(params[f'coarse_model.{n}'], dev) for n in ops.PARAM_ORDER], precision=pr)
    for mode, noise in (('train', g['noise']), ('eval', None)):
        o = ops.mlp_forward(pk, tp.cu(g['pts'], dev), tp.cu(g['view_dirs'], dev), tp.cu(g['view_dirs2'], dev),
                            tp.cu(noise, dev) if noise is not None else None, 1.0, precision=pr)
        tp.assert_close(o['sigma'], g[f'sigma_{mode}'], what=f'{prec} sigma {mode}')
        tp.assert_close(o['rgb'], g[f'rgb_{mode}'], what=f'{prec} rgb {mode}')
        tp.assert_close(o['visibility'], g[f'vis_{mode}'], what=f'{prec} vis {mode}')
        tp.assert_close(o['visibility2'], g[f'vis2_{mode}'], what=f'{prec} vis2 {mode}')


def test_retired_modes_are_refused(dev):
    """bf16x3, bf16x6 and hip_bf16_layout 'wide' fail LOUDLY (VIPNERF_E_UNSUPPORTED with the reason) in every entry point that takes a
    precision or a configuration -- no silent fallback to another arithmetic."""
    from vipnerf_hip._lib import VipNerfHipError
    ops = tp.hip_ops()
    params = vo.init_params(3, levels=('coarse',))
    tensors = [tp.cu(params[f'coarse_model.{n}'], dev) for n in ops.PARAM_ORDER]
    for prec in ('bf16x3', 'bf16x6'):
        with pytest.raises(VipNerfHipError, match='retired'):
            ops.pack_weights(tensors, precision=ops.PRECISIONS[prec])
        b = vo.synthetic_batch(16, 5, scene='fern', nf=2)
        model, _ = make_model(dev, True, vo.init_params(3), prec)
        with pytest.raises(VipNerfHipError, match='retired'):
            model(tp.ref_batch(b, dev, 0))
    pk = ops.pack_weights(tensors)
    with pytest.raises(VipNerfHipError, match='retired'):
        ops.mlp_forward(pk, torch.zeros(4, 3, device=dev), torch.zeros(4, 3, device=dev), precision=ops.PRECISIONS['bf16x6'])
    model, _ = make_model(dev, True, vo.init_params(3), 'fp32-wide')
    with pytest.raises(VipNerfHipError, match='retired'):
        model(tp.ref_batch(vo.synthetic_batch(16, 5, scene='fern', nf=2), dev, 0))


def make_model(dev, ndc, params, mode, sparse=False):
    model, cfg = tp.make_model(dev, ndc, params, sparse=sparse)
    prec, _, layout = mode.partition('-')
    model.configs['model']['hip_precision'] = prec
    model.configs['model']['hip_bf16_layout'] = layout or 'narrow'
    return model, cfg


def grad_tol(mode):
    return GRAD_TOL[mode.partition('-')[0]]


@pytest.mark.parametrize('prec', MODES)
def test_eval_render_golden(dev, prec):
    g = tp.load('f4_eval_fern')
    b = vo.synthetic_batch(int(g['n']), int(g['seed_batch']), scene='fern', nf=2)
    params = vo.init_params(int(g['seed_params']), scale=float(g['scale_params']), sigma_bias=float(g['sigma_bias']))
    model, _ = make_model(dev, True, params, prec)
    model.eval()
    with torch.no_grad():
        model.injected_z_fine = tp.cu(g['out_z_vals_fine'], dev)
        out = model(tp.ref_batch(b, dev, 0), retraw=True, sec_views_vis=True)
    for lv in ('coarse', 'fine'):
        for rk in tp.KEYMAP:
            gk = f'out_{rk}_{lv}'
            if gk in g:
                tp.assert_close(out[f'{rk}_{lv}'], g[gk], what=f'{prec} {rk}_{lv}')


@pytest.mark.parametrize('prec', MODES)
@pytest.mark.parametrize('tag', ['llff', 'realestate', 'dtu'])
def test_train_step_golden(dev, prec, tag):
    """F5 in every split arithmetic -- incl. the RealEstate case (NDC, V = 2, sparse-depth rows + SparseDepthMSE), which is
    BASELINE configs[2]'s shape."""
    from loss_functions.LossComputerHip01 import LossComputerHip
    g = tp.load(f'f5_train_{tag}')
    n_sparse = int(g['n_sparse'])
    b = vo.synthetic_batch(int(g['n']), int(g['seed_batch']), scene=str(g['scene']), nf=int(g['nf']), n_sparse=n_sparse)
    params = vo.init_params(int(g['seed_params']), scale=float(g['scale_params']))
    model, cfg = make_model(dev, b['ndc'], params, prec, sparse=n_sparse > 0)
    model.train()
    lossc = LossComputerHip(cfg)
    model.injected_rng = {k[4:]: tp.cu(v, dev) for k, v in g.items() if k.startswith('rng_')}
    model.injected_z_fine = tp.cu(g['out_z_vals_fine'], dev)
    out = model(tp.ref_batch(b, dev, 40000))
    for lv in ('coarse', 'fine'):
        for rk in tp.KEYMAP:
            gk = f'out_{rk}_{lv}'
            if gk in g:
                tp.assert_close(out[f'{rk}_{lv}'], g[gk], what=f'{prec} {tag} {rk}_{lv}')
    l40k = lossc.compute_losses(tp.ref_batch(b, dev, 40000), out)
    tp.assert_close(l40k['TotalLoss'], g['l40k_TotalLoss'], rtol=1e-4, floor=1e-6, what=f'{prec} {tag} TotalLoss')
    l40k['TotalLoss'].backward()

    def digest(t):
        f = t.detach().reshape(-1).double().cpu()
        nn = f.numel()
        idx = (torch.arange(192, dtype=torch.long) * 7919) % nn
        return torch.cat([f.sum()[None], f.norm()[None], f[:64] if nn >= 64 else torch.cat([f, f.new_zeros(64 - nn)]), f[idx]]).numpy()
    worst = 0.0
    for k, p in model.named_parameters():
        gd, dg = g['gdig_' + k], digest(p.grad)
        np.testing.assert_allclose(dg[1], gd[1], rtol=grad_tol(prec), atol=1e-9, err_msg=f'{prec} {tag} |grad| of {k}')
        tp.grad_close(dg[2:], gd[2:], f'{prec} {tag} grad samples of {k}', scale=max(abs(gd[1]) / np.sqrt(p.numel()), 1e-12),
                      l2_tol=grad_tol(prec))
        if 'grad_' + k in g:
            tp.grad_close(p.grad.cpu().numpy(), g['grad_' + k], f'{prec} {tag} grad of {k}', l2_tol=grad_tol(prec))
            worst = max(worst, float(np.linalg.norm(p.grad.cpu().numpy() - g['grad_' + k]) / max(np.linalg.norm(g['grad_' + k]), 1e-30)))
    print(f'{prec} {tag}: worst rel L2 error over the fully stored gradient tensors {worst:.2e}')


@pytest.mark.parametrize('prec', MODES)
def test_backward_vs_oracle(dev, prec):
    n = 40
    b = vo.synthetic_batch(n, 77, scene='fern', nf=2)
    params = vo.init_params(78, scale=1.6)
    rng = vo.synthetic_rng(n, 64, 128, 79)
    p = vo.params_to_torch(params, requires_grad=True)
    ref = vo.render_rays(p, b, {'ndc': True, 'n_coarse': 64, 'n_fine': 128, 'noise_std': 1.0}, rng, train=True, sec_views=True)
    model, _ = make_model(dev, True, params, prec)
    model.train()
    model.injected_rng = {k: v.to(dev) for k, v in rng.items()}
    model.injected_z_fine = ref['z_vals_fine'].detach().to(dev)
    out = model(tp.ref_batch(b, dev, 0))
    gen = torch.Generator().manual_seed(5)
    tot_o, tot_h = 0, 0
    for lv in ('coarse', 'fine'):
        for k in ['rgb', 'acc', 'depth', 'visibility2', 'visibility', 'weights', 'raw_sigma', 'raw_rgb', 'raw_visibility', 'raw_visibility2']:
            kk = f'{k}_{lv}'
            ct = torch.randn(ref[kk].shape, generator=gen) / ref[kk].numel() ** 0.5
            tot_o = tot_o + (ref[kk] * ct).sum()
            tot_h = tot_h + (out[kk] * ct.to(dev)).sum()
    tot_o.backward()
    tot_h.backward()
    worst = 0.0
    for k, t in model.named_parameters():
        tp.grad_close(t.grad.cpu().numpy(), p[k].grad.numpy(), f'{prec} {k}', l2_tol=grad_tol(prec))
        worst = max(worst, float((t.grad.cpu() - p[k].grad).norm() / p[k].grad.norm()))
    print(f'{prec}: worst relative L2 gradient error over 48 tensors {worst:.3e}')


@pytest.mark.parametrize('prec', PRECS)
def test_mlp_forward_ragged_and_empty(dev, prec):
    """point counts that are not a multiple of the 128-point workgroup tile (16-point waves: 1, 15, 17, ...), and zero
    points, in every split mode"""
    ops = tp.hip_ops()
    params = vo.init_params(9, levels=('coarse',))
    pr = ops.PRECISIONS[prec]
    pk = ops.pack_weights([tp.cu(params[f'coarse_model.{n}'], dev) for n in ops.PARAM_ORDER], precision=pr)
    p = vo.params_to_torch(params)
    rs = np.random.default_rng(1)
    for P in (1, 15, 17, 37, 129, 300):
        pts = torch.from_numpy(rs.uniform(-1, 1, size=(P, 3)).astype(np.float32))
        vd = torch.nn.functional.normalize(torch.from_numpy(rs.standard_normal((P, 3)).astype(np.float32)), dim=-1)
        ref = vo.mlp_forward(p, 'coarse', pts, vd, None, None)
        o = ops.mlp_forward(pk, pts.to(dev), vd.to(dev), precision=pr)
        tp.assert_close(o['rgb'], ref['rgb'], what=f'{prec} rgb P={P}')
        tp.assert_close(o['sigma'], ref['sigma'], what=f'{prec} sigma P={P}')
    o = ops.mlp_forward(pk, torch.zeros(0, 3, device=dev), torch.zeros(0, 3, device=dev), precision=pr)
    assert o['rgb'].shape == (0, 3)


@pytest.mark.parametrize('prec', ['fp16x3', 'fp16x3h'])
def test_full_size_step_properties(dev, prec):
    """BASELINE config 2 sizes (4096 rays x 64+128) in the bench arithmetics: determinism of a whole training step
    (outputs and all 48 gradients bit-identical run to run), ray independence of the eval render, finite values, and
    gradients that agree with the exact-fp32 MFMA path to the mode's tolerance."""
    n = 4096
    b = vo.synthetic_batch(n, 31, scene='fern', nf=2)
    params = vo.init_params(32, scale=1.6, sigma_bias=0.5)
    rng = {k: v.to(dev) for k, v in vo.synthetic_rng(n, 64, 128, 33).items()}

    def step(mode):
        model, _ = make_model(dev, True, params, mode) if mode != 'fp32' else tp.make_model(dev, True, params)
        model.train()
        model.injected_rng = rng
        out = model(tp.ref_batch(b, dev, 0))
        tgt = b['target_rgb'].to(dev)
        loss = ((out['rgb_fine'] - tgt) ** 2).mean() + ((out['rgb_coarse'] - tgt) ** 2).mean() \
            + (out['raw_visibility_fine'][..., 0] - out['visibility_fine']).abs().mean() * 0.1
        loss.backward()
        return model, out, {k: p.grad.clone() for k, p in model.named_parameters()}

    m1, o1, g1 = step(prec)
    _, o2, g2 = step(prec)
    for k in ('rgb_fine', 'rgb_coarse', 'depth_fine', 'visibility_fine'):
        assert torch.equal(o1[k], o2[k]), f'{prec}: non-deterministic {k}'
        assert torch.isfinite(o1[k]).all(), k
    for k in g1:
        assert torch.equal(g1[k], g2[k]), f'{prec}: non-deterministic gradient {k}'
        assert torch.isfinite(g1[k]).all(), k
    _, _, gref = step('fp32')
    worst = max(float((g1[k] - gref[k]).norm() / gref[k].norm().clamp_min(1e-30)) for k in g1)
    # z_vals_fine is sampled from the coarse weights, so a last-bit change of a coarse weight can move a fine sample: the
    # comparison tolerates that, the golden-vector tests (teacher-forced depths) hold the tight bound
    assert worst < 2e-2, worst
    m1.eval()
    with torch.no_grad():
        whole = m1(tp.ref_batch(b, dev, 0), retraw=True, sec_views_vis=True)
        halves = []
        for sl in (slice(0, n // 2), slice(n // 2, n)):
            hb = {k: (v[sl] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == n else v)
                  for k, v in tp.ref_batch(b, dev, 0).items()}
            halves.append(m1(hb, retraw=True, sec_views_vis=True))
    for k in whole:
        assert torch.equal(whole[k], torch.cat([halves[0][k], halves[1][k]], 0)), f'{prec}: ray independence {k}'


def test_fp16x3_range(dev):
    """fp16 fragments have 5 exponent bits.  Inside the range (activations < 65504: here a 6x over-scaled network whose
    raw sigma reaches ~7e3) fp16x3 agrees with the exact fp32 path to fp32 rounding; beyond it (10x: activations ~1e5)
    the outputs are NaN -- a loud failure, never finite garbage."""
    ops = tp.hip_ops()
    rs = np.random.default_rng(0)
    P = 2048
    pts = tp.cu(rs.uniform(-1, 1, size=(P, 3)).astype(np.float32), dev)
    vd = torch.nn.functional.normalize(tp.cu(rs.standard_normal((P, 3)).astype(np.float32), dev), dim=-1)

    def run(scale, prec):
        params = vo.init_params(3, levels=('coarse',), scale=scale)
        pr = ops.PRECISIONS[prec]
        pk = ops.pack_weights([tp.cu(params[f'coarse_model.{n}'], dev) for n in ops.PARAM_ORDER], precision=pr)
        return ops.mlp_forward(pk, pts, vd, precision=pr)

    ref, o = run(6.0, 'fp32'), run(6.0, 'fp16x3')
    assert float(ref['sigma'].max()) > 1e3
    assert float((o['sigma'] - ref['sigma']).abs().max() / ref['sigma'].abs().max()) < 1e-5
    assert float((o['rgb'] - ref['rgb']).abs().max()) < 1e-4
    big = run(10.0, 'fp16x3')
    assert torch.isfinite(run(10.0, 'fp32')['sigma']).all()
    assert not torch.isfinite(big['sigma']).all() and not torch.isfinite(big['rgb']).all()


# ------------------------------------------------------------------------------------------------ single-MFMA 16-bit modes
# 'fp16' / 'bf16' (VIPNERF_PREC_FP16 / BF16): operands rounded ONCE to 16 bits, one MFMA per product -- BASELINE configs[4]'s
# mixed precision, a separate accuracy class: (output rtol, output floor relative to the tensor's max, gradient rel. L2).
# Measured on the goldens (round-2 diagnostics, docs/HISTORY.md): fp16 rgb 3e-5 abs, sigma 3e-4 of max, gradients 1e-3
# median / 2.6e-2 worst tensor; bf16 2.6e-4, 2.4e-3, 1.6e-2 / 8.4e-2.
SINGLE = {'fp16': (5e-3, 2e-3, 6e-2), 'bf16': (4e-2, 1.5e-2, 0.2)}


@pytest.mark.parametrize('prec', list(SINGLE))
@pytest.mark.parametrize('V', [1, 2])
def test_single_mfma_mlp_forward_golden(dev, prec, V):
    ops = tp.hip_ops()
    rtol, floor, _ = SINGLE[prec]
    g = tp.load(f'f2_mlp_v{V}')
    params = vo.init_params(int(g['seed']), levels=('coarse',))
    pr = ops.PRECISIONS[prec]
    pk = ops.pack_weights([tp.cu(params[f'coarse_model.{n}'], dev) for n in ops.PARAM_ORDER], precision=pr)
    for mode, noise in (('train', g['noise']), ('eval', None)):
        o = ops.mlp_forward(pk, tp.cu(g['pts'], dev), tp.cu(g['view_dirs'], dev), tp.cu(g['view_dirs2'], dev),
                            tp.cu(noise, dev) if noise is not None else None, 1.0, precision=pr)
        for k, gk in (('sigma', 'sigma'), ('rgb', 'rgb'), ('visibility', 'vis'), ('visibility2', 'vis2')):
            tp.assert_close(o[k], g[f'{gk}_{mode}'], rtol=rtol, floor=floor, what=f'{prec} {k} {mode}')


@pytest.mark.parametrize('prec', list(SINGLE))
def test_single_mfma_mlp_forward_ragged_and_empty(dev, prec):
    """The two-point-tile kernels (32 points per wave, 256 per workgroup) on point counts that cut a wave's first tile, its second tile, a
    workgroup -- 1, 15, 17, 33, 255, 257, 300 -- with V = 1 secondary direction, and zero points: against the oracle at the mode's tolerance, and
    every count's rows equal to the same rows of the 300-point call bit for bit (a lane beyond P must not disturb its wave)."""
    ops = tp.hip_ops()
    rtol, floor, _ = SINGLE[prec]
    params = vo.init_params(9, levels=('coarse',))
    pr = ops.PRECISIONS[prec]
    pk = ops.pack_weights([tp.cu(params[f'coarse_model.{n}'], dev) for n in ops.PARAM_ORDER], precision=pr)
    p = vo.params_to_torch(params)
    rs = np.random.default_rng(1)
    pts = torch.from_numpy(rs.uniform(-1, 1, size=(300, 3)).astype(np.float32))
    vd = torch.nn.functional.normalize(torch.from_numpy(rs.standard_normal((300, 3)).astype(np.float32)), dim=-1)
    vd2 = torch.nn.functional.normalize(torch.from_numpy(rs.standard_normal((300, 1, 3)).astype(np.float32)), dim=-1)
    ref = vo.mlp_forward(p, 'coarse', pts, vd, vd2, None)
    full = ops.mlp_forward(pk, pts.to(dev), vd.to(dev), vd2.to(dev), precision=pr)
    for k in ('rgb', 'sigma', 'visibility', 'visibility2'):
        tp.assert_close(full[k], ref[k], rtol=rtol, floor=floor, what=f'{prec} {k} P=300')
    for P in (1, 15, 17, 33, 255, 257):
        o = ops.mlp_forward(pk, pts[:P].to(dev), vd[:P].to(dev), vd2[:P].to(dev), precision=pr)
        for k in ('rgb', 'sigma', 'visibility', 'visibility2'):
            assert torch.equal(o[k], full[k][:P]), f'{prec} {k}: the {P}-point call differs from the first {P} rows of the 300-point call'
    o = ops.mlp_forward(pk, torch.zeros(0, 3, device=dev), torch.zeros(0, 3, device=dev), precision=pr)
    assert o['rgb'].shape == (0, 3)


@pytest.mark.parametrize('prec', list(SINGLE))
@pytest.mark.parametrize('tag', ['llff', 'realestate', 'dtu'])
def test_single_mfma_train_step_golden(dev, prec, tag):
    from loss_functions.LossComputerHip01 import LossComputerHip
    rtol, floor, gtol = SINGLE[prec]
    g = tp.load(f'f5_train_{tag}')
    n_sparse = int(g['n_sparse'])
    b = vo.synthetic_batch(int(g['n']), int(g['seed_batch']), scene=str(g['scene']), nf=int(g['nf']), n_sparse=n_sparse)
    params = vo.init_params(int(g['seed_params']), scale=float(g['scale_params']))
    model, cfg = make_model(dev, b['ndc'], params, prec, sparse=n_sparse > 0)
    model.train()
    lossc = LossComputerHip(cfg)
    model.injected_rng = {k[4:]: tp.cu(v, dev) for k, v in g.items() if k.startswith('rng_')}
    model.injected_z_fine = tp.cu(g['out_z_vals_fine'], dev)
    out = model(tp.ref_batch(b, dev, 40000))
    for lv in ('coarse', 'fine'):
        for rk in ('rgb', 'acc', 'visibility2', 'raw_sigma', 'raw_rgb', 'raw_visibility', 'raw_visibility2', 'weights', 'visibility'):
            gk = f'out_{rk}_{lv}'
            if gk in g:
                tp.assert_close(out[f'{rk}_{lv}'], g[gk], rtol=rtol, floor=floor, what=f'{prec} {tag} {rk}_{lv}')
    l40k = lossc.compute_losses(tp.ref_batch(b, dev, 40000), out)
    tp.assert_close(l40k['TotalLoss'], g['l40k_TotalLoss'], rtol=4 * rtol, floor=1e-6, what=f'{prec} {tag} TotalLoss')
    l40k['TotalLoss'].backward()
    worst = 0.0
    for k, p in model.named_parameters():
        assert torch.isfinite(p.grad).all(), k
        if 'grad_' + k in g:
            tp.grad_close(p.grad.cpu().numpy(), g['grad_' + k], f'{prec} {tag} grad of {k}', l2_tol=gtol)
            worst = max(worst, float(np.linalg.norm(p.grad.cpu().numpy() - g['grad_' + k]) / max(np.linalg.norm(g['grad_' + k]), 1e-30)))
    print(f'{prec} {tag}: worst rel L2 error over the fully stored gradient tensors {worst:.2e}')
