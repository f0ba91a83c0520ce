"""Device time of the eval-mode MLP kernels (HIP events) for A/B builds:
   VIPNERF_HIP_LIB=.../libvipnerf_hip_expN.so HIP_PRECISION=fp32 python tools/eval_time.py"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd'))
from oracle import vipnerf_oracle as vo
from vipnerf_hip import ops
dev = torch.device('cuda:0'); cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
name = os.environ.get('HIP_PRECISION', 'fp32')
prec = ops.PRECISIONS[name]
n = int(os.environ.get('RAYS', 32768))
b = vo.synthetic_batch(n, 7, scene='fern', nf=2)
bd = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
pa = vo.init_params(3)
pc = ops.pack_weights([cu(pa['coarse_model.' + k]) for k in ops.PARAM_ORDER], precision=prec)
pf = ops.pack_weights([cu(pa['fine_model.' + k]) for k in ops.PARAM_ORDER], precision=prec)
cfg = ops.make_config(True, 64, 128, 0, False, precision=prec)
for _ in range(2):
    ops.render_forward(cfg, bd, None, pc, pf)
ops.profile_enable(True); ops.profile_read()
K = 3
for _ in range(K):
    ops.render_forward(cfg, bd, None, pc, pf)
torch.cuda.synchronize()
pr = ops.profile_read()
ms = sum(v[1] for k, v in pr.items() if k.startswith('mlp_fwd')) / K
tf = 593536 * 2.0 * 256 * n / (ms * 1e-3) / 1e12
print('%-60s %s rays %d: mlp kernels %.3f ms = %.1f TFLOP/s algorithmic (%.3f of 157.3, %.3f of 2500)' % (
    os.path.basename(os.environ.get('VIPNERF_HIP_LIB', 'default')), name, n, ms, tf, tf / 157.3, tf / 2500))
