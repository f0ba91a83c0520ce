"""Accuracy and speed of the split-bf16 forward kernels vs fp32 / the golden vectors."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd')); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd', 'src'))
from oracle import vipnerf_oracle as vo
from vipnerf_hip import ops
dev = torch.device('cuda:0')
cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
def err(a, b):
    a = a.detach().cpu().double().numpy(); b = np.asarray(b, np.float64).reshape(a.shape)
    d = np.abs(a - b); tol = 1e-4 * np.abs(b) + 1e-5 * np.abs(b).max()
    return 'max_abs %.2e  max_rel(floor) %.2e  viol %d%s' % (d.max(), (d / (np.abs(b) + 1e-5 * np.abs(b).max())).max(), (d > tol).sum(), ' NAN' if not np.isfinite(a).all() else '')
for V in (1, 2):
    g = dict(np.load(os.path.join(ROOT, 'tests/golden/f2_mlp_v%d.npz' % V)))
    params = vo.init_params(int(g['seed']), levels=('coarse',))
    for prec in (0, 1, 2):
        pk = ops.pack_weights([cu(params['coarse_model.' + n]) for n in ops.PARAM_ORDER], precision=prec)
        o = ops.mlp_forward(pk, cu(g['pts']), cu(g['view_dirs']), cu(g['view_dirs2']), cu(g['noise']), 1.0, precision=prec)
        torch.cuda.synchronize()
        print('V=%d prec=%d sigma: %s | rgb: %s | vis2: %s' % (V, prec, err(o['sigma'], g['sigma_train']), err(o['rgb'], g['rgb_train']), err(o['visibility2'], g['vis2_train'])))
# larger-magnitude weights (scale 1.6) on random points vs the oracle
params = vo.init_params(77, levels=('coarse',), scale=1.6, sigma_bias=0.5)
p = vo.params_to_torch(params)
rs = np.random.default_rng(0)
P = 4096
pts = torch.from_numpy(rs.uniform(-1, 1, (P, 3)).astype(np.float32)); vd = torch.nn.functional.normalize(torch.from_numpy(rs.standard_normal((P, 3)).astype(np.float32)), dim=-1)
ref = vo.mlp_forward(p, 'coarse', pts, vd, None, None)
for prec in (0, 1, 2):
    pk = ops.pack_weights([cu(params['coarse_model.' + n]) for n in ops.PARAM_ORDER], precision=prec)
    o = ops.mlp_forward(pk, pts.to(dev), vd.to(dev), precision=prec)
    print('scale1.6 prec=%d sigma: %s | rgb: %s' % (prec, err(o['sigma'], ref['sigma'].numpy()), err(o['rgb'], ref['rgb'].numpy())))
# speed: eval render of 4096 rays and train forward
sys.path.insert(0, ROOT)
import bench
b = vo.synthetic_batch(4096, 7, scene='fern', nf=2)
bd = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
bd['rays_o2'] = vo.secondary_origins(b['poses'], b['pixel_id'][:, 0].long(), 2).to(dev)
pa = vo.init_params(3)
for prec in (0, 1, 2):
    pc = ops.pack_weights([cu(pa['coarse_model.' + n]) for n in ops.PARAM_ORDER], precision=prec)
    pf = ops.pack_weights([cu(pa['fine_model.' + n]) for n in ops.PARAM_ORDER], precision=prec)
    for train in (False, True):
        cfg = ops.make_config(True, 64, 128, 1, train, noise_std=1.0 if train else 0.0, save_acts=train, precision=prec)
        acts = None
        if train:
            ab, _ = ops.query_workspace(cfg, 4096); acts = torch.empty(ab // 4, dtype=torch.float32, device=dev)
        rng = {'seed': 1, 'offset': 0} if train else None
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.time()
            ops.render_forward(cfg, bd, rng, pc, pf, acts)
            torch.cuda.synchronize(); dt = time.time() - t0
        print('prec=%d train=%s forward 4096 rays: %.2f ms' % (prec, train, dt * 1e3))
