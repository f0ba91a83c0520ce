"""Relative L2 error of the parameter gradients of every MLP arithmetic against the F5 golden (reference, CPU fp32)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd')); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd', 'src'))
from oracle import vipnerf_oracle as vo
import test_hip_parity as tp
from loss_functions.LossComputerHip01 import LossComputerHip
dev = torch.device('cuda:0')
for tag in ('llff', 'dtu'):
    g = tp.load(f'f5_train_{tag}')
    n_sparse = int(g['n_sparse'])
    for prec in ('fp32', 'bf16x6', 'fp16x3', 'fp16x3h', 'bf16x3', 'fp16', 'bf16'):
        b = vo.synthetic_batch(int(g['n']), int(g['seed_batch']), scene=str(g['scene']), nf=int(g['nf']), n_sparse=n_sparse)
        params = vo.init_params(int(g['seed_params']), scale=float(g['scale_params']))
        model, cfg = tp.make_model(dev, b['ndc'], params, sparse=n_sparse > 0)
        model.configs['model']['hip_precision'] = prec
        model.train()
        lossc = LossComputerHip(cfg)
        model.injected_rng = {k[4:]: tp.cu(v, dev) for k, v in g.items() if k.startswith('rng_')}
        model.injected_z_fine = tp.cu(g['out_z_vals_fine'], dev)
        rb = tp.ref_batch(b, dev, 40000)
        out = model(rb)
        lossc.compute_losses(rb, out)['TotalLoss'].backward()
        errs = []
        for k, p in model.named_parameters():
            if 'grad_' + k in g:
                ref = g['grad_' + k]; a = p.grad.cpu().numpy()
                errs.append((np.linalg.norm(a - ref) / max(np.linalg.norm(ref), 1e-30), k))
        oerr = max(float(np.abs(out[f'rgb_{lv}'].detach().cpu().numpy() - g[f'out_rgb_{lv}']).max()) for lv in ('coarse', 'fine'))
        errs.sort()
        print('%-5s %-7s rgb max abs err %.2e | grad rel-L2: median %.2e  max %.2e (%s)' % (tag, prec, oerr, errs[len(errs) // 2][0], errs[-1][0], errs[-1][1]))
