"""Per-tensor gradient errors of one case of tests/test_hip_sweep.py against the CPU oracle, in several arithmetics:
   python tools/sweep_case_diag.py <case index> [precisions...]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vip-nerf_amd'), os.path.join(ROOT, 'vip-nerf_amd', 'src'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from oracle import vipnerf_oracle as vo          # the checker
from test_hip_sweep import CASES
from test_hip_round2 import _oracle_and_hip_step
c = CASES[int(sys.argv[1])]
print(c)
dev = torch.device('cuda:0')
for prec in sys.argv[2:] or ['fp32', 'fp16', 'bf16']:
    sparse = c['n_sparse'] > 0
    b = vo.synthetic_batch(c['n'], 7000 + c['i'], scene=c['scene'], nf=c['nf'], n_sparse=c['n_sparse'])
    n_rows = c['n'] + c['n_sparse']
    params = vo.init_params(7100 + c['i'], scale=1.6)
    rng = vo.synthetic_rng(n_rows, c['nco'], c['nfi'], 7200 + c['i'])
    upd = {'white_bkgd': c['white'], 'lindisp': c['lindisp'], 'raw_noise_std': c['noise']}
    cfg_o = {'ndc': b['ndc'], 'n_coarse': c['nco'], 'n_fine': c['nfi'], 'noise_std': c['noise'], 'white_bkgd': c['white'], 'lindisp': c['lindisp']}
    (ref, lref, p), (out, lh, model) = _oracle_and_hip_step(dev, b, params, rng, upd, cfg_o, iter_num=c['iter_num'], sparse=sparse, prec=prec)
    print(prec, 'TotalLoss', float(lh['TotalLoss']), float(lref['TotalLoss']))
    for k in ('rgb_fine', 'raw_sigma_fine', 'raw_rgb_fine', 'raw_visibility2_fine', 'weights_fine', 'alpha_coarse'):
        a, r = out[k].detach().cpu().double().reshape(ref[k].shape), ref[k].detach().double()
        print('   %-22s max abs err %.3e  ref max %.3e' % (k, float((a - r).abs().max()), float(r.abs().max())))
    for k, t in model.named_parameters():
        g, r = t.grad.cpu().double(), p[k].grad.double()
        print('   %-44s rel L2 %.3e   |ref| %.3e' % (k, float((g - r).norm() / r.norm().clamp_min(1e-30)), float(r.norm())))
