"""Forward error of every MLP arithmetic against the F2 golden (reference fp32 CPU) + eval kernel time."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd'))
from oracle import vipnerf_oracle as vo
from vipnerf_hip import ops
dev = torch.device('cuda:0'); cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'f2_mlp_v1.npz')))
params = vo.init_params(int(g['seed']), levels=('coarse',))
for name, pr in ops.PRECISIONS.items():
    pk = ops.pack_weights([cu(params[f'coarse_model.{n}']) for n in ops.PARAM_ORDER], precision=pr)
    o = ops.mlp_forward(pk, cu(g['pts']), cu(g['view_dirs']), cu(g['view_dirs2']), None, 1.0, precision=pr)
    line = '%-7s' % name
    for k, gk in (('sigma', 'sigma_eval'), ('rgb', 'rgb_eval'), ('visibility', 'vis_eval'), ('visibility2', 'vis2_eval')):
        a, b = o[k].cpu().numpy().reshape(-1), g[gk].reshape(-1)
        line += '  %s: max abs %.2e (rel to max %.2e)' % (k, np.abs(a - b).max(), np.abs(a - b).max() / np.abs(b).max())
    print(line)
# timing on 128*2048 points
P = 128 * 2048
pts = torch.rand(P, 3, device=dev) * 2 - 1; vd = torch.nn.functional.normalize(torch.randn(P, 3, device=dev), dim=-1)
for name, pr in ops.PRECISIONS.items():
    pk = ops.pack_weights([cu(params[f'coarse_model.{n}']) for n in ops.PARAM_ORDER], precision=pr)
    for _ in range(2): ops.mlp_forward(pk, pts, vd, precision=pr)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): ops.mlp_forward(pk, pts, vd, precision=pr)
    e1.record(); torch.cuda.synchronize()
    print('%-7s eval %.3f ms per 262144 points' % (name, e0.elapsed_time(e1) / 5))
for name, pr in ops.PRECISIONS.items():
    pk = ops.pack_weights([cu(params[f'coarse_model.{n}']) for n in ops.PARAM_ORDER], precision=pr)
    o = ops.mlp_forward(pk, cu(g['pts']), cu(g['view_dirs']), cu(g['view_dirs2']), None, 1.0, precision=pr)
    print(name, 'sigma', o['sigma'].cpu().numpy().reshape(-1)[:6], 'rgb', o['rgb'].cpu().numpy().reshape(-1)[:4])
print('gold    sigma', g['sigma_eval'].reshape(-1)[:6], 'rgb', g['rgb_eval'].reshape(-1)[:4])
