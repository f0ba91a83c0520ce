"""Per-stage device time of one training step (HIP events), for A/B experiments:
   VIPNERF_HIP_LIB=.../libvipnerf_hip_expN.so python tools/stage_times.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd')); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd', 'src'))
import bench
from vipnerf_hip import ops
from models.ModelFactory import get_model
from loss_functions.LossComputerHip01 import LossComputerHip
dev = torch.device('cuda:0')
scene = os.environ.get('SCENE', 'fern')
cfg = bench.model_configs(bench.SCENES[scene][5])
cfg['model']['hip_precision'] = os.environ.get('HIP_PRECISION', 'fp32')
torch.manual_seed(0)
model = get_model(cfg, None).to(dev).train()
lossc = LossComputerHip(cfg)
b0 = bench.make_batch(bench.make_scene(scene, dev), int(os.environ.get('RAYS', 4096)), 1000)
def step():
    b = dict(b0); b['common_data'] = {'poses': b0['common_data']['poses']}
    model.zero_grad(set_to_none=True)
    out = model(b); lossc.compute_losses(b, out)['TotalLoss'].backward()
for _ in range(3): step()
ops.profile_enable(True); ops.profile_read()
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 5
for _ in range(K): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
pr = ops.profile_read()
print(os.environ.get('VIPNERF_HIP_LIB', 'default'), cfg['model']['hip_precision'], 'step %.2f ms |' % (dt * 1e3), ' '.join('%s %.2f' % (k, v[1] / K) for k, v in sorted(pr.items()) if v[1] / K > 0.05))
