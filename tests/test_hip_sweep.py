"""A seeded sweep over the configuration space the header says is honoured -- sample counts (multiples of 32 up to 256 per
level), 1..3 secondary views, NDC / metric scenes, sparse-depth rows, lindisp, white_bkgd, noise on / off, ragged ray counts --
one teacher-forced training step each against the CPU oracle (outputs, per-loss values, all 48 gradients).  The cases are drawn
from a fixed seed, so the sweep is the same on every run; it exists to catch an assumption that only the shipped 64 + 128 / V = 1
shape satisfies (reference: the same code path serves every config, src/models/VipNeRF01.py:74-170)."""
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
from test_hip_round2 import _oracle_and_hip_step, assert_close_few_outliers  # noqa: E402


def _cases(n_cases=24, seed=20260929):
    g = np.random.RandomState(seed)
    out = []
    for i in range(n_cases):
        scene = ['fern', 'dtu', 'realestate'][g.randint(3)]
        nco = int(32 * g.randint(1, 5))                           # 32 .. 128
        total = int(32 * g.randint(nco // 32 + 1, 9))              # nco + 32 .. 256
        out.append(dict(i=i, scene=scene, nf=int(g.randint(2, 5)), n=int(g.randint(1, 41)), nco=nco, nfi=total - nco,
                        n_sparse=int(g.randint(1, 9)) if scene == 'realestate' and g.rand() < 0.7 else 0,
                        white=bool(scene == 'dtu' and g.rand() < 0.5), lindisp=bool(scene == 'dtu' and g.rand() < 0.5),
                        noise=float([0.0, 1.0][g.randint(2)]), iter_num=int([0, 12000, 40000][g.randint(3)])))
    return out


CASES = _cases()


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.mark.parametrize('c', CASES, ids=lambda c: '%(i)d-%(scene)s-nf%(nf)d-n%(n)d-%(nco)d+%(nfi)d' % c)
def test_config_sweep_vs_oracle(dev, c):
    sparse = c['n_sparse'] > 0
    b = vo.synthetic_batch(c['n'], 7000 + c['i'], scene=c['scene'], nf=c['nf'], n_sparse=c['n_sparse'])
    n_rows = c['n'] + c['n_sparse']
    params = vo.init_params(7100 + c['i'], scale=1.6)
    rng = vo.synthetic_rng(n_rows, c['nco'], c['nfi'], 7200 + c['i'])
    upd = {'white_bkgd': c['white'], 'lindisp': c['lindisp'], 'raw_noise_std': c['noise']}
    cfg_o = {'ndc': b['ndc'], 'n_coarse': c['nco'], 'n_fine': c['nfi'], 'noise_std': c['noise'], 'white_bkgd': c['white'],
             'lindisp': c['lindisp']}
    (ref, lref, p), (out, lh, model) = _oracle_and_hip_step(dev, b, params, rng, upd, cfg_o, iter_num=c['iter_num'], sparse=sparse)
    what = 'case %d' % c['i']
    assert out['rgb_fine'].shape == (n_rows, 3) and out['visibility2_fine'].shape == (n_rows, c['nf'] - 1)
    assert out['z_vals_fine'].shape == (n_rows, c['nco'] + c['nfi'])
    assert torch.equal(out['z_vals_coarse'].cpu(), ref['z_vals_coarse']), what + ': coarse depths must be bit-identical'
    for k in ref:
        if k in out and k not in ('z_vals_coarse', 'z_vals_fine'):
            if k.startswith('depth'):        # near-empty rays: depth = sum w z / sum w is ill-conditioned for a ray or two
                assert_close_few_outliers(out[k], ref[k], 2e-4, f'{what} {k}', max_frac=max(0.005, 1.5 / n_rows))   # one ray may
            else:
                # alpha = 1 - exp(-sigma delta) carries one ulp of 1.0 (6e-8) ABSOLUTE in any fp32 evaluation, and the weights
                # and everything composited from them inherit it; in a nearly empty volume (every alpha < 1e-2: random weights
                # on a far-reaching scene) that exceeds a floor taken relative to the outputs' own maximum, so the floor is
                # at least 2e-7 (a few ulps of 1.0) here
                a_, b_ = out[k].detach().cpu().double().numpy().reshape(ref[k].shape), ref[k].detach().double().numpy()
                tol = 1e-4 * np.abs(b_) + max(1e-5 * np.abs(b_).max(), 2e-7)
                assert np.isfinite(a_).all() and np.all(np.abs(a_ - b_) <= tol), \
                    f'{what} {k}: max abs err {np.abs(a_ - b_).max():.3e} (ref max {np.abs(b_).max():.3e})'
    for k, v in lref.items():
        hk = k.replace('01', 'Hip01') if k != 'TotalLoss' else k
        if hk not in lh:
            continue
        hv = lh[hk]
        hv = hv['loss_value'] if isinstance(hv, dict) else hv
        rv = v['loss_value'] if isinstance(v, dict) else v
        tp.assert_close(hv, rv, rtol=2e-4, floor=1e-6, what=f'{what} loss {k}')
    for k, t in model.named_parameters():
        tp.grad_close(t.grad.cpu().numpy(), p[k].grad.numpy(), f'{what} grad {k}', l2_tol=4e-3)      # 24 seeded cases of 8 .. 96 rays: 18 measure <= 6e-5, the worst (case 0) 1.9e-3: kink events in tiny batches


# (output rtol, floor relative to the tensor's max, absolute floor, gradient rel. L2 per tensor, median over the 48 tensors): the 1024-ray
# classes of tests/test_hip_round2.py::ARITH16 with room for what these cases add -- batches of 2 .. 48 rows instead of 1024 (a tensor's
# gradient averages the 16-bit roundings of 25 .. 500 x fewer points: measured medians up to 1.3e-2 fp16 / 4.5e-2 bf16 at 2 + 6 rows) and
# noise-free cases, where the densities are small (max 0.2) and the 16-bit pre-activation's ABSOLUTE error (2e-4 fp16, 1.2e-3 bf16) is what
# alpha / weights carry (1.4e-5 / 8e-5 on values <= 7e-3)
SWEEP16 = {'fp16': (5e-3, 5e-4, 5e-5, 1e-1, 2e-2), 'bf16': (4e-2, 5e-3, 3e-4, 3e-1, 6e-2),
           # fp16x3h: fp32-grade forward / data gradients (3 fp16 MFMAs per product), 16-bit tile storage + single-MFMA weight gradients
           'fp16x3h': (2e-4, 1e-5, 2e-7, 1e-2, 1e-3),      # measured: medians 1.7e-5 .. 1.7e-4, worst tensor 2.1e-3
           # fp16x3: fp32-grade everywhere (3 fp16 MFMAs per product in every GEMM, presplit operand storage, split weight-gradient kernels)
           'fp16x3': (2e-4, 1e-5, 2e-7, 5e-3, 1e-5)}        # measured: medians 3e-7 .. 2.3e-6; a tensor or two per case at 1e-4 .. 2e-3 (ReLU flips of single points in 1 .. 48-row batches)


@pytest.mark.parametrize('prec', list(SWEEP16))
@pytest.mark.parametrize('c', CASES[:12], ids=lambda c: '%(i)d-%(scene)s-nf%(nf)d-n%(n)d-%(nco)d+%(nfi)d' % c)
def test_config_sweep_16bit_vs_oracle(dev, c, prec):
    """The same seeded cases through the single-MFMA 16-bit modes (two point tiles per wave, 16-bit tile storage, DMA-fed weight
    gradients): sample counts from 64 to 256 per ray, V = 1..3, batches down to one row (a 256-point workgroup with 32 valid points),
    sparse-depth rows, lindisp, white background -- at the accuracy class of one 16-bit rounding per operand."""
    sparse = c['n_sparse'] > 0
    b = vo.synthetic_batch(c['n'], 7000 + c['i'], scene=c['scene'], nf=c['nf'], n_sparse=c['n_sparse'])
    n_rows = c['n'] + c['n_sparse']
    params = vo.init_params(7100 + c['i'], scale=1.6)
    rng = vo.synthetic_rng(n_rows, c['nco'], c['nfi'], 7200 + c['i'])
    upd = {'white_bkgd': c['white'], 'lindisp': c['lindisp'], 'raw_noise_std': c['noise']}
    cfg_o = {'ndc': b['ndc'], 'n_coarse': c['nco'], 'n_fine': c['nfi'], 'noise_std': c['noise'], 'white_bkgd': c['white'],
             'lindisp': c['lindisp']}
    rtol, floor, afloor, gtol, gmed = SWEEP16[prec]
    (ref, lref, p), (out, lh, model) = _oracle_and_hip_step(dev, b, params, rng, upd, cfg_o, iter_num=c['iter_num'], sparse=sparse, prec=prec)
    what = '%s case %d' % (prec, c['i'])
    assert torch.equal(out['z_vals_coarse'].cpu(), ref['z_vals_coarse']), what + ': coarse depths must be bit-identical'
    for k in ref:
        if k in out and k not in ('z_vals_coarse', 'z_vals_fine'):
            assert torch.isfinite(out[k]).all(), f'{what} {k}'
            if k.startswith('depth'):
                assert_close_few_outliers(out[k], ref[k], rtol, f'{what} {k}', floor=floor, max_frac=max(0.01, 1.5 / n_rows))
            else:
                tp.assert_close(out[k], ref[k], rtol=rtol, floor=max(floor, (10 * afloor if 'sigma' in k else afloor) / max(float(ref[k].abs().max()), 1e-30)), what=f'{what} {k}')
    tp.assert_close(lh['TotalLoss'], lref['TotalLoss'], rtol=4 * rtol, floor=1e-6, what=f'{what} TotalLoss')
    errs = []
    for k, t in model.named_parameters():
        tp.grad_close(t.grad.cpu().numpy(), p[k].grad.numpy(), f'{what} grad {k}', l2_tol=gtol)
        errs.append(float((t.grad.cpu() - p[k].grad).norm() / p[k].grad.norm().clamp_min(1e-30)))
    errs.sort()
    print(f'{what}: gradient rel. L2 median {errs[len(errs) // 2]:.2e}, worst {errs[-1]:.2e}')
    assert errs[len(errs) // 2] <= gmed
    # the same weights in eval mode with secondary views (a validation frame: the kernels without activation stores, ragged point counts):
    # the coarse level against the oracle's eval render (the fine level samples from the coarse weights: finite, and close where it is compared)
    model.eval()
    model.injected_rng = model.injected_z_fine = None
    with torch.no_grad():
        ev = model(tp.ref_batch(b, dev, 0), retraw=True, sec_views_vis=True)
    ro = vo.render_rays(vo.params_to_torch(params), b, cfg_o, None, train=False, sec_views=True)
    for k in ('rgb_coarse', 'acc_coarse', 'visibility2_coarse', 'raw_sigma_coarse', 'raw_visibility2_coarse', 'weights_coarse'):
        tp.assert_close(ev[k], ro[k], rtol=rtol, floor=max(floor, (10 * afloor if 'sigma' in k else afloor) / max(float(ro[k].abs().max()), 1e-30)), what=f'{what} eval {k}')
    assert all(torch.isfinite(v).all() for v in ev.values())
