"""Trunk weight-gradient error (fine / coarse layer 0 and 7, relative L2 against the CPU oracle) over small ray counts and sample counts:
   python tools/ragged_diag.py [precision]      (VIPNERF_HIP_LIB selects a library variant)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vip-nerf_amd'), os.path.join(ROOT, 'vip-nerf_amd', 'src'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from oracle import vipnerf_oracle as vo          # the checker
from test_hip_round2 import _oracle_and_hip_step
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
dev = torch.device('cuda:0')
print(os.environ.get('VIPNERF_HIP_LIB', 'default'), prec)
for scene, nf in (('fern', 2), ('dtu', 4)):
    for nco, nfi in ((64, 128), (128, 32), (32, 32), (64, 64)):
        for n in (1, 2, 3, 5, 8):
            b = vo.synthetic_batch(n, 9000 + n, scene=scene, nf=nf)
            params = vo.init_params(9100, scale=1.6)
            rng = vo.synthetic_rng(n, nco, nfi, 9200 + n)
            cfg_o = {'ndc': b['ndc'], 'n_coarse': nco, 'n_fine': nfi, 'noise_std': 1.0}
            (ref, lref, p), (out, lh, model) = _oracle_and_hip_step(dev, b, params, rng, {}, cfg_o, prec=prec)
            e = {}
            for k, t in model.named_parameters():
                if k.endswith('pts_linears.0.weight') or k.endswith('pts_linears.7.weight') or k.endswith('feature_linear.weight'):
                    e[k.replace('_model.pts_linears', '').replace('.weight', '').replace('_model.feature_linear', '.f')] = float((t.grad.cpu() - p[k].grad).norm() / p[k].grad.norm())
            print('%-5s nf %d  %3d+%-3d n %d  Pc %5d Pf %5d (mod 256: %3d %3d) ' % (scene, nf, nco, nfi, n, n * nco, n * (nco + nfi), n * nco % 256, n * (nco + nfi) % 256),
                  ' '.join('%s %.1e' % kv for kv in e.items()))
