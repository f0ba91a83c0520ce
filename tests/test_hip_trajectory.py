"""F9: a K-step TRAJECTORY of the reference's training loop (tests/golden/f9_trajectory_fern.npz, made by oracle/gen_golden.py::gen_f9
from the reference's own Trainer.train_one_iter, NeRFLearningRateDecayer, LossComputer and torch Adam: reference src/Trainer01.py:61-107,
:292-298, lr_decayers/NeRFLearningRateDecayer01.py:19-23, loss_functions/LossComputer01.py:46-60).  Eight iterations 29996..30003 of 128
fern rays each -- lr decays every iteration, the visibility-prior weight switches 0 -> 0.001 at 30000 inside the window, Adam's moments
build up over eight steps.  Three drivers of the HIP path are held to it in fp32, each fed the reference's recorded draws and fine depths:

    module contract   VipNeRFHip.forward -> LossComputerHip.compute_losses -> TotalLoss.backward() -> optimizer.step(), with torch's own
                      Adam and with FlatAdam;
    one call          vipnerf_hip.step.FusedTrainStep (vipnerf_train_step) + FlatAdam;
    the trainer       TrainerHip.train() resumed from a reference-format checkpoint written at iteration 29996.

Bounds: every loss of every iteration to 1e-4 relative (north_star; measured printed), parameters after the eighth step to 1e-5 where
the reference's gradient stayed above rounding level in all eight iterations (|g| > 1e-6: Adam's update lr * m / (sqrt(v) + 1e-8) is then
determined by the gradient, not by its rounding), 3e-5 elsewhere.
"""
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

from oracle import vipnerf_oracle as vo  # noqa: E402  (checker only)

REF_NAMES = {'MSEHip01': 'MSE01', 'VisibilityLossHip01': 'VisibilityLoss01', 'VisibilityPriorLossHip01': 'VisibilityPriorLoss01',
             'TotalLoss': 'TotalLoss'}


def _load():
    return {k: v for k, v in np.load(os.path.join(ROOT, 'tests', 'golden', 'f9_trajectory_fern.npz')).items()}


def _digest(t):
    f = t.detach().reshape(-1).double().cpu()
    n = f.numel()
    idx = (torch.arange(192, dtype=torch.long) * 7919) % n
    return torch.cat([f[:64] if n >= 64 else torch.cat([f, f.new_zeros(64 - n)]), f[idx]]).numpy()


def _setup(g, dev, flat_adam):
    from test_hip_parity import make_model
    from vipnerf_hip.optim import FlatAdam
    params = vo.init_params(int(g['seed_params']), scale=float(g['scale_params']))
    model, cfg = make_model(dev, True, params)
    cfg['optimizer'] = {'lr_initial': 0.0005, 'lr_decay': 250, 'beta1': 0.9, 'beta2': 0.999}
    model.train()
    opt = (FlatAdam if flat_adam else torch.optim.Adam)(model.parameters(), lr=5e-4, betas=(0.9, 0.999))
    return model, cfg, opt


def _batch(g, i, dev):
    from test_hip_parity import ref_batch
    b = vo.synthetic_batch(int(g['n']), int(g['seed']) + 10 + i, scene='fern', nf=2)
    return ref_batch(b, dev, int(g['first_iter']) + i)


def _inject(model, g, i, dev):
    pre = f'it{i}_rng_'
    model.injected_rng = {k[len(pre):]: torch.from_numpy(v).to(dev) for k, v in g.items() if k.startswith(pre)}
    model.injected_z_fine = torch.from_numpy(g[f'it{i}_z_vals_fine']).to(dev)


def _check_losses(g, i, losses, what, worst):
    for k, v in losses.items():
        v = v['loss_value'] if isinstance(v, dict) else v
        ref = float(g[f'it{i}_loss_{REF_NAMES[k]}'])
        rel = abs(float(v) - ref) / max(abs(ref), 1e-12)
        worst[0] = max(worst[0], rel)
        assert rel <= 1e-4, f'{what}: iteration {i} {k}: {float(v):.8g} vs the reference\'s {ref:.8g} (rel {rel:.2e})'


def _check_params(g, model, what):
    firm_err = soft_err = 0.0
    n_firm = n_all = 0
    for k, p in model.named_parameters():
        after, ref, firm = _digest(p), g['adig_' + k][2:], g['gmin_' + k] > 1e-6
        d = np.abs(after - ref)
        firm_err, soft_err = max(firm_err, d[firm].max() if firm.any() else 0.0), max(soft_err, d[~firm].max() if (~firm).any() else 0.0)
        n_firm += int(firm.sum()); n_all += firm.size
        assert np.isfinite(after).all(), k
        assert (d[firm] <= 1e-5).all(), f'{what}: {k} after 8 iterations: {d[firm].max():.3e} from the reference\'s'
        assert (d[~firm] <= 3e-5).all(), f'{what}: {k} (rounding-level gradients): {d[~firm].max():.3e}'
        if 'after_' + k in g:
            assert np.abs(p.detach().cpu().numpy() - g['after_' + k]).max() <= 3e-5, k
    assert n_firm > 0.5 * n_all
    return firm_err, soft_err


@pytest.mark.parametrize('flat_adam', [False, True])
def test_trajectory_module_contract(flat_adam):
    """Trainer01.py:292-298 + :61-107 restated with the drop-in classes: lr into every param group, zero_grad, forward, losses, backward, step."""
    from loss_functions.LossComputerHip01 import LossComputerHip
    dev = torch.device('cuda:0')
    g = _load()
    model, cfg, opt = _setup(g, dev, flat_adam)
    lossc = LossComputerHip(cfg)
    worst = [0.0]
    for i in range(int(g['iters'])):
        it = int(g['first_iter']) + i
        lr = cfg['optimizer']['lr_initial'] * (0.1 ** (it / (cfg['optimizer']['lr_decay'] * 1000)))
        assert lr == float(g[f'it{i}_lr'])
        for grp in opt.param_groups:
            grp['lr'] = lr
        b = _batch(g, i, dev)
        _inject(model, g, i, dev)
        opt.zero_grad(set_to_none=True)
        out = model(b)
        e = float((out['rgb_fine'].detach().cpu() - torch.from_numpy(g[f'it{i}_rgb_fine'])).abs().max())
        assert e <= 1e-5, f'iteration {i}: rgb_fine {e:.3e} from the reference\'s'
        losses = lossc.compute_losses(b, out)
        _check_losses(g, i, losses, 'module contract', worst)
        losses['TotalLoss'].backward()
        opt.step()
    fe, se = _check_params(g, model, 'module contract')
    print(f'module contract ({"FlatAdam" if flat_adam else "torch.optim.Adam"}): worst loss rel err {worst[0]:.2e} (bound 1e-4); parameters after 8 steps: '
          f'{fe:.2e} (bound 1e-5) where |g| > 1e-6 throughout, {se:.2e} (bound 3e-5) elsewhere')


def test_trajectory_one_call_step():
    from vipnerf_hip.step import FusedTrainStep, named_losses
    dev = torch.device('cuda:0')
    g = _load()
    model, cfg, opt = _setup(g, dev, True)
    step = FusedTrainStep(model, cfg, opt)
    worst = [0.0]
    for i in range(int(g['iters'])):
        lr = float(g[f'it{i}_lr'])
        for grp in opt.param_groups:
            grp['lr'] = lr
        _inject(model, g, i, dev)
        res = named_losses(step(_batch(g, i, dev)))
        _check_losses(g, i, res, 'one call', worst)
    fe, se = _check_params(g, model, 'one call')
    print(f'vipnerf_train_step: worst loss rel err {worst[0]:.2e}; parameters after 8 steps: {fe:.2e} / {se:.2e}')


@pytest.mark.parametrize('one_call', [True, False])
def test_trajectory_trainer_resumed_from_checkpoint(one_call, tmp_path):
    """TrainerHip.train(): resumes at 29996 from a reference-format checkpoint (fresh Adam state, as the fixture's run starts), sets the
    decayed lr, switches the prior weight, steps -- its logged history against the reference's eight iterations."""
    import CheckpointHip01 as ckpt
    import TrainerHip01 as T
    dev = torch.device('cuda:0')
    g = _load()
    first, iters = int(g['first_iter']), int(g['iters'])
    model0, cfg, opt0 = _setup(g, dev, True)
    ckpt.save_model(model0, opt0, first, tmp_path)
    cfg.update(num_iterations=first + iters, one_call_step=one_call)

    class Gen:                                       # what the trainer asks of its ray generator: the iteration's batch on the device
        device, n, images = dev, 2, None

        def get_next_batch(self, iter_num, scheduler=None):
            _inject(tr.model, g, iter_num - first, dev)
            return _batch(g, iter_num - first, dev)

    torch.manual_seed(123)                           # the trainer builds its own model: different initial weights until the checkpoint loads
    tr = T.TrainerHip(cfg, Gen(), None, output_dirpath=tmp_path)
    hist = tr.train()
    assert len(hist) == iters and (tr.stepper is not None) == one_call
    worst = [0.0]
    for i, h in enumerate(hist):
        assert h['lr'] == float(g[f'it{i}_lr']) and h.get('iter', first + i) == first + i
        _check_losses(g, i, {k: torch.tensor(v) for k, v in h.items() if k in REF_NAMES}, 'trainer', worst)
    fe, se = _check_params(g, tr.model, 'trainer')
    print(f'TrainerHip.train (one_call_step={one_call}): worst loss rel err {worst[0]:.2e}; parameters after 8 steps: {fe:.2e} / {se:.2e}')
