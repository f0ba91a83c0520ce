"""BASELINE configs 3 and 5 at full batch size (RealEstate 3 views + sparse depth, 2048 + 2048 rays; DTU 3 views, non-NDC,
4096 rays): one training step per arithmetic, outputs / loss / gradients against the exact-fp32 MFMA path of the same
library, and the step time.  (The small-n versions of these cases are pinned to the reference by tests/golden/f5_*.)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd')); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd', 'src')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench
from oracle import vipnerf_oracle as vo          # synthetic batches only
from models.ModelFactory import get_model
from loss_functions.LossComputerHip01 import LossComputerHip
dev = torch.device('cuda:0')
def to_dev(b, it):
    rb = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items() if k not in ('poses', 'ndc')}
    rb['common_data'] = {'poses': b['poses'][None].clone().to(dev)}
    rb['iter_num'] = it
    return rb
for scene, nf, n, n_sparse in (('realestate', 3, 2048, 2048), ('dtu', 3, 4096, 0)):
    b = vo.synthetic_batch(n, 7, scene=scene, nf=nf, n_sparse=n_sparse)
    res = {}
    for prec in ('fp32', 'fp16x3', 'fp16x3h', 'fp16', 'bf16'):
        cfg = bench.model_configs(); cfg['model']['hip_precision'] = prec
        cfg['data_loader']['ndc'] = bool(b['ndc'])
        if n_sparse:
            cfg['losses'] = list(cfg['losses']) + [{'name': 'SparseDepthMSEHip01', 'weight': 0.1}]
        torch.manual_seed(0)
        model = get_model(cfg, None).to(dev).train()
        lossc = LossComputerHip(cfg)
        def step():
            rb = to_dev(b, 40000)
            model.zero_grad(set_to_none=True)
            torch.manual_seed(5); model._calls = 0
            out = model(rb); lv = lossc.compute_losses(rb, out); lv['TotalLoss'].backward()
            return out, lv
        out, lv = step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): out, lv = step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        g = torch.cat([p.grad.flatten() for p in model.parameters()])
        res[prec] = (out['rgb_fine'].detach().clone(), float(lv['TotalLoss']), g.clone(), dt)
        assert torch.isfinite(g).all() and torch.isfinite(out['rgb_fine']).all()
    r = res['fp32']
    for prec, (rgb, loss, g, dt) in res.items():
        print('%-10s %-8s %d rays V=%d: %.2f ms/step | rgb_fine max abs dev %.2e | loss %.7f (dev %.1e) | grad rel-L2 dev %.2e' % (
            scene, prec, n + n_sparse, nf - 1, dt * 1e3, float((rgb - r[0]).abs().max()), loss, abs(loss - r[1]), float((g - r[2]).norm() / r[2].norm())))
