"""List the PyTorch-side (non-library) kernels of one training step with their call sites (torch.profiler)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd')); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd', 'src'))
import bench
from oracle import vipnerf_oracle as vo
from models.ModelFactory import get_model
from loss_functions.LossComputerHip01 import LossComputerHip
from vipnerf_hip import dist as vdist
dev = torch.device('cuda:0')
cfg = bench.model_configs(); cfg['model']['hip_precision'] = os.environ.get('HIP_PRECISION', 'fp16x3')
model = get_model(cfg, None).to(dev); model.train()
lossc = LossComputerHip(cfg)
opt = torch.optim.Adam(model.parameters(), lr=5e-4, fused=True)
bucket = vdist.FlatGradBucket(model.parameters())
batches = [bench.make_batch_oracle(vo, 4096, 1000 + i, dev) for i in range(4)]
def step(i):
    b = dict(batches[i]); b['common_data'] = {'poses': batches[i]['common_data']['poses']}
    bucket.release(); out = model(b); losses = lossc.compute_losses(b, out); losses['TotalLoss'].backward(); opt.step()
for i in range(2): step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(2); torch.cuda.synchronize()
evs = [e for e in prof.key_averages(group_by_stack_n=6) if e.key in ('aten::fill_', 'aten::zero_', 'aten::zeros', 'aten::copy_', 'aten::mul', 'aten::add', 'aten::add_', 'aten::_to_copy', 'aten::sum', 'aten::select_backward')]
for e in sorted(evs, key=lambda e: -e.count)[:40]:
    st = [s for s in e.stack if 'site-packages' not in s and 'dist-packages' not in s][:3]
    print('%-22s x%-3d cuda %.0f us | %s' % (e.key, e.count, e.device_time_total, ' <- '.join(x.strip()[-70:] for x in st)))
