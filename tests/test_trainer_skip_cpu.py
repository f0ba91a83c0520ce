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
