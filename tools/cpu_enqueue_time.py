"""Host time to ENQUEUE one training step (model forward, fused losses, backward, flat Adam) vs the step's device time:
   RAYS=1024 HIP_PRECISION=bf16 python tools/cpu_enqueue_time.py"""
import os, sys, time, cProfile, pstats
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd')); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd', 'src'))
import bench
from models.ModelFactory import get_model
from loss_functions.LossComputerHip01 import LossComputerHip
from vipnerf_hip.optim import FlatAdam
dev = torch.device('cuda:0')
cfg = bench.model_configs(True)
cfg['model']['hip_precision'] = os.environ.get('HIP_PRECISION', 'bf16')
torch.manual_seed(0)
model = get_model(cfg, None).to(dev).train()
lossc = LossComputerHip(cfg)
opt = FlatAdam(model.parameters(), lr=5e-4)
n = int(os.environ.get('RAYS', 1024))
b0 = bench.make_batch(bench.make_scene('fern', dev), n, 1000)
def step(i):
    b = dict(b0); b['common_data'] = {'poses': b0['common_data']['poses']}; b['iter_num'] = 40000 + i
    opt.zero_grad(set_to_none=True)
    out = model(b); lossc.compute_losses(b, out)['TotalLoss'].backward(); opt.step()
for i in range(10): step(i)
torch.cuda.synchronize()
K = 100
t0 = time.perf_counter()
for i in range(K): step(10 + i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'{n} rays {cfg["model"]["hip_precision"]}: enqueue {1e3 * (t1 - t0) / K:.3f} ms per step (host), {1e3 * (t2 - t0) / K:.3f} ms per step incl. the final wait')
if os.environ.get('PROFILE'):
    pr = cProfile.Profile(); pr.enable()
    for i in range(50): step(200 + i)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
