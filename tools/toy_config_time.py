"""BASELINE configs[0] on the GPU: 64 x 64 toy scene, 2 views, 1024 rays per iteration, 4-layer x 64 coarse-only MLP -- the generic-
topology kernels (docs/HISTORY.md 4.6).  Training-step time and the CPU oracle's time for the same step.   python tools/toy_config_time.py"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'vip-nerf_amd'), os.path.join(ROOT, 'vip-nerf_amd', 'src')):
    sys.path.insert(0, p)
from oracle import vipnerf_oracle as vo          # synthetic batch + the CPU timing
import test_hip_parity as tp
from test_hip_round2 import _generic_model
from loss_functions.LossComputerHip01 import LossComputerHip
dev = torch.device('cuda:0')
n = 1024
b = vo.synthetic_batch(n, 5, scene='toy', nf=2)
params = vo.init_params(6, depth=4, width=64, levels=('coarse',), scale=1.6)
model, cfg = _generic_model(dev, b['ndc'], params, 4, 64, 0)
model.train()
lossc = LossComputerHip(cfg)
opt = torch.optim.Adam(model.parameters(), lr=5e-4, fused=True)
rb = tp.ref_batch(b, dev, 40000)
def step(i):
    bb = dict(rb); bb['common_data'] = {'poses': rb['common_data']['poses']}; bb['iter_num'] = 40000 + i
    opt.zero_grad(set_to_none=True)
    lossc.compute_losses(bb, model(bb))['TotalLoss'].backward()
    opt.step()
for i in range(10): step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 200
for i in range(K): step(10 + i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
# the CPU oracle on the same step
torch.set_num_threads(16)
p = vo.params_to_torch(params, requires_grad=True)
o = torch.optim.Adam(list(p.values()), lr=5e-4)
rng = vo.synthetic_rng(n, 64, 0, 7)
lcfg = [{'name': 'MSE01', 'weight': 1}, {'name': 'VisibilityLoss01', 'weight': 0.1}, {'name': 'VisibilityPriorLoss01', 'iter_weights': {'0': 0, '30000': 0.001}}]
def cstep():
    o.zero_grad(set_to_none=True)
    out = vo.render_rays(p, b, {'ndc': b['ndc'], 'n_coarse': 64, 'n_fine': 0, 'noise_std': 1.0, 'depth': 4}, rng, train=True, sec_views=True)
    vo.total_loss(b, out, lcfg, 40000, levels=('coarse',))['TotalLoss'].backward(); o.step()
for _ in range(3): cstep()
t0 = time.perf_counter()
for _ in range(10): cstep()
ct = (time.perf_counter() - t0) / 10
print(f'configs[0] (4 x 64 coarse-only, 1024 rays x 64 samples): GPU generic kernels {dt * 1e3:.3f} ms per step = {n / dt / 1e3:.0f} k rays/s; '
      f'CPU oracle (16 threads) {ct * 1e3:.1f} ms = {n / ct / 1e3:.1f} k rays/s; ratio {ct / dt:.0f}x')
from vipnerf_hip import ops
ops.profile_enable(True); ops.profile_read()
for i in range(20): step(500 + i)
torch.cuda.synchronize()
pr = ops.profile_read(); ops.profile_enable(False)
print('library stages, ms per step:', {k: round(v[1] / 20, 3) for k, v in sorted(pr.items())}, 'sum', round(sum(v[1] for v in pr.values()) / 20, 3))
