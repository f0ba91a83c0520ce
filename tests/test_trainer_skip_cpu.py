"""TrainerHip.train() through SKIPPED iterations (CPU, no library call): a multi-GPU run whose epoch-end batch trims to zero rows skips the
iteration on every rank (train_one_iter returns {}); the log must skip it too instead of crashing at the next flush -- first in a block,
in the middle of one, last of the run (TrainerHip01.py:80)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd'))
sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd', 'src'))


def _trainer(total, flush_every, skipped):
    import TrainerHip01 as T
    tr = object.__new__(T.TrainerHip)             # the loop alone: no model, no GPU
    tr.configs = {'num_iterations': total, 'log_flush_interval': flush_every}
    tr.rank, tr.world, tr.output_dirpath, tr.lr_init, tr.lr_decay_steps = 0, 2, None, 5e-4, 250000.0
    tr.optimizer = type('O', (), {'param_groups': [{'lr': 0.0}]})()
    tr.model = type('M', (), {'train': lambda self: None})()
    tr.calls = []

    def train_one_iter(it):
        tr.calls.append(it)
        return {} if it in skipped else {'MSEHip01': torch.tensor(float(it)), 'TotalLoss': torch.tensor(2.0 * it)}
    tr.train_one_iter = train_one_iter
    return tr


def test_train_skips_empty_iterations_in_the_log():
    for skipped in ((0, 3, 6), (1,), (2, 3, 4, 5), tuple(range(7)), ()):
        tr = _trainer(7, 3, skipped)
        hist = tr.train(log_every=2)
        assert tr.calls == list(range(7))                                   # every iteration still runs (and skips identically on every rank)
        kept = [i for i in range(7) if i not in skipped]
        assert [h['MSEHip01'] for h in hist] == [float(i) for i in kept]
        assert [h['TotalLoss'] for h in hist] == [2.0 * i for i in kept]
        assert all(abs(h['lr'] - tr.learning_rate(i)) < 1e-12 for h, i in zip(hist, kept))
        assert [h['iter'] for h in hist] == kept                             # every row names its iteration


def test_skipped_iteration_on_a_validation_boundary_keeps_its_psnr():
    """a skipped iteration that falls on a validation boundary still runs the validation; its psnr must reach the history (a
    loss-less row naming the iteration) instead of being written into an entry nobody keeps."""
    tr = _trainer(8, 3, skipped=(3, 4))
    tr.configs['validation_interval'] = 4
    tr.run_validation = lambda: {0: {'psnr': 20.0 + len(tr.calls)}}
    hist = tr.train()
    assert [h['iter'] for h in hist] == [0, 1, 2, 3, 5, 6, 7]
    by_iter = {h['iter']: h for h in hist}
    assert by_iter[3]['validation_psnr'] == 24.0 and 'MSEHip01' not in by_iter[3]
    assert by_iter[7]['validation_psnr'] == 28.0 and by_iter[7]['MSEHip01'] == 7.0
    assert all('validation_psnr' not in by_iter[i] for i in (0, 1, 2, 5, 6))


def test_learning_rate_and_loss_weight_follow_the_reference_trajectory():
    """F9 (the reference's lr decayer and LossComputer weights over iterations 29996..30003): the host logic that needs no GPU."""
    import numpy as np
    from vipnerf_hip.step import loss_weight
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'f9_trajectory_fern.npz'))
    tr = _trainer(1, 1, ())
    first = int(g['first_iter'])
    lc = {'name': 'VisibilityPriorLossHip01', 'iter_weights': {'0': 0, '30000': 0.001}}
    for i in range(int(g['iters'])):
        assert tr.learning_rate(first + i) == float(g[f'it{i}_lr'])
        w = loss_weight(lc, first + i)
        total = float(g[f'it{i}_loss_MSE01']) + 0.1 * float(g[f'it{i}_loss_VisibilityLoss01']) + w * float(g[f'it{i}_loss_VisibilityPriorLoss01'])
        assert abs(total - float(g[f'it{i}_loss_TotalLoss'])) <= 2e-7 * abs(total), (i, w)
        assert w == (0.001 if first + i >= 30000 else 0)


def test_named_losses_reports_a_diverged_loss_under_its_own_name_only():
    """one NaN loss slot must not turn every logged name into NaN (the reference's LossComputer reports per loss)."""
    from vipnerf_hip.step import named_losses
    lv = torch.tensor([0.25, 0.5, float('nan'), 1.0, 2.0, 4.0, 8.0, 0.0])
    res = {'TotalLoss': torch.tensor([3.0]), 'loss_values': lv, 'two_levels': True,
           'loss_slots': {'MSEHip01': (0, 1), 'VisibilityLossHip01': (2, 3), 'VisibilityPriorLossHip01': (4, 5), 'SparseDepthMSEHip01': (6, 7)}}
    out = named_losses(res)
    assert float(out['MSEHip01']) == 0.75 and float(out['VisibilityPriorLossHip01']) == 6.0 and float(out['SparseDepthMSEHip01']) == 8.0
    assert torch.isnan(out['VisibilityLossHip01']) and float(out['TotalLoss']) == 3.0
    res['loss_slots']['SparseDepthMSEHip01'] = None           # a loss that reported nothing this iteration
    res['two_levels'] = False
    out = named_losses(res)
    assert float(out['MSEHip01']) == 0.25 and float(out['SparseDepthMSEHip01']) == 0.0 and torch.isnan(out['VisibilityLossHip01'])
