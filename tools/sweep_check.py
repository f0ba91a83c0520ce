"""Randomised agreement sweep: every split mode against the exact-fp32 HIP path over ray counts, secondary-view counts,
coarse-only / coarse+fine, NDC on/off -- outputs and all parameter gradients.  (The fp32 path is pinned to the reference
by the golden tests; this sweep looks for shape-dependent bugs in the other kernels.)"""
import itertools, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd')); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd', 'src'))
from oracle import vipnerf_oracle as vo
import test_hip_parity as tp
dev = torch.device('cuda:0')
bad = 0
cases = list(itertools.product((1, 7, 33, 130, 515), (2, 3, 4), (128, 0), ('fern', 'dtu')))
for n, nf, n_fine, scene in cases:
    b = vo.synthetic_batch(n, 100 + n, scene=scene, nf=nf)
    params = vo.init_params(7 + nf, scale=1.6) if n_fine else vo.init_params(7 + nf, scale=1.6, levels=('coarse',))
    rng = {k: v.to(dev) for k, v in vo.synthetic_rng(n, 64, max(n_fine, 1), 5 + n).items()}
    res = {}
    for prec in ('fp32', 'fp16x3', 'fp16x3h', 'bf16x6', 'bf16x3', 'bf16x6-wide'):
        model, cfg = tp.make_model(dev, b['ndc'], params, n_fine=n_fine)
        p, _, lay = prec.partition('-')
        model.configs['model']['hip_precision'] = p
        if lay: model.configs['model']['hip_bf16_layout'] = lay
        model.train()
        model.injected_rng = rng
        out = model(tp.ref_batch(b, dev, 0))
        lv = 'fine' if n_fine else 'coarse'
        tgt = b['target_rgb'].to(dev)
        loss = ((out[f'rgb_{lv}'] - tgt) ** 2).mean() + ((out['rgb_coarse'] - tgt) ** 2).mean() + out[f'depth_{lv}'].mean() * 1e-2
        if f'visibility2_{lv}' in out: loss = loss + out[f'visibility2_{lv}'].mean() * 1e-2
        loss.backward()
        res[prec] = (out[f'rgb_{lv}'].detach().clone(), {k: q.grad.clone() for k, q in model.named_parameters() if q.grad is not None})
    ref_o, ref_g = res['fp32']
    for prec, (o, g) in res.items():
        if prec == 'fp32': continue
        eo = float((o - ref_o).abs().max())
        eg = max(float((g[k] - ref_g[k]).norm() / ref_g[k].norm().clamp_min(1e-30)) for k in ref_g)
        fin = bool(torch.isfinite(o).all()) and all(bool(torch.isfinite(v).all()) for v in g.values())
        tol_g = {'fp16x3': 2e-2, 'fp16x3h': 2e-2, 'bf16x6': 2e-2, 'bf16x6-wide': 2e-2, 'bf16x3': 5e-2}[prec]   # fine depths resample on last-bit changes
        ok = fin and eo < 2e-3 and eg < tol_g and set(g) == set(ref_g)
        if not ok:
            bad += 1
            print('MISMATCH n=%d nf=%d n_fine=%d %s %s: rgb %.2e grad %.2e finite=%s' % (n, nf, n_fine, scene, prec, eo, eg, fin))
print('cases %d x 5 modes, mismatches %d' % (len(cases), bad))
