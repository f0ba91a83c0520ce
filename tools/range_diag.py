"""fp16x3 vs the exact fp32 path as the weights (hence the activations) grow: where does fp16's range end?"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd'))
from oracle import vipnerf_oracle as vo
from vipnerf_hip import ops
dev = torch.device('cuda:0'); cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
rs = np.random.default_rng(0)
P = 4096
pts = cu(rs.uniform(-1, 1, size=(P, 3)).astype(np.float32)); vd = torch.nn.functional.normalize(cu(rs.standard_normal((P, 3)).astype(np.float32)), dim=-1)
for scale in (1.0, 1.6, 2.5, 4.0, 6.0, 10.0):
    params = vo.init_params(3, levels=('coarse',), scale=scale)
    outs = {}
    for name in ('fp32', 'fp16x3', 'bf16x6'):
        pr = ops.PRECISIONS[name]
        pk = ops.pack_weights([cu(params[f'coarse_model.{n}']) for n in ops.PARAM_ORDER], precision=pr)
        outs[name] = ops.mlp_forward(pk, pts, vd, precision=pr)
    ref = outs['fp32']
    line = 'weight scale %5.1f  max sigma %.3e ' % (scale, float(ref['sigma'].max()))
    for name in ('fp16x3', 'bf16x6'):
        o = outs[name]
        fin = all(bool(torch.isfinite(o[k]).all()) for k in ('sigma', 'rgb'))
        es = float((o['sigma'] - ref['sigma']).abs().max() / ref['sigma'].abs().max().clamp_min(1e-30))
        er = float((o['rgb'] - ref['rgb']).abs().max())
        line += '| %s finite=%s sigma rel %.1e rgb abs %.1e ' % (name, fin, es, er)
    print(line)
