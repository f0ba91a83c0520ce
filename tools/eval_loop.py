"""Eval-mode render of 4096 rays x N iterations at a given precision (for rocprofv3 counter passes)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd'))
from oracle import vipnerf_oracle as vo
from vipnerf_hip import ops
dev = torch.device('cuda:0'); cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
prec = ops.PRECISIONS[os.environ.get('HIP_PRECISION', 'bf16x3')]
b = vo.synthetic_batch(4096, 7, scene='fern', nf=2)
bd = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
pa = vo.init_params(3)
pc = ops.pack_weights([cu(pa['coarse_model.' + n]) for n in ops.PARAM_ORDER], precision=prec)
pf = ops.pack_weights([cu(pa['fine_model.' + n]) for n in ops.PARAM_ORDER], precision=prec)
cfg = ops.make_config(True, 64, 128, 0, False, precision=prec)
for _ in range(int(os.environ.get('ITERS', 3))):
    ops.render_forward(cfg, bd, None, pc, pf)
torch.cuda.synchronize()
