"""Which PyTorch-side device work is left in a module-contract training step (rocblas dot, reduce, elementwise, fills, copies)?
Runs a few 1024-ray steps under torch.profiler and prints, for every device kernel / memcpy / memset that is NOT one of the library's (vn::*), the
count per step and the Python frames that launched it.   python tools/torch_ops_in_step.py [fp32|bf16] [module|onecall]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vip-nerf_amd'), os.path.join(ROOT, 'vip-nerf_amd', 'src')):
    sys.path.insert(0, p)
import bench  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
api = sys.argv[2] if len(sys.argv) > 2 else 'module'
dev = torch.device('cuda:0')
from models.ModelFactory import get_model  # noqa: E402
from loss_functions.LossComputerHip01 import LossComputerHip  # noqa: E402
from vipnerf_hip import dist as vdist  # noqa: E402
from vipnerf_hip.optim import FlatAdam  # noqa: E402

cfg = bench.model_configs(True)
cfg['model']['hip_precision'] = prec
torch.manual_seed(0)
model = get_model(cfg, None).to(dev).train()
lossc = LossComputerHip(cfg)
opt = FlatAdam(model.parameters(), lr=5e-4, betas=(0.9, 0.999))
bucket = vdist.FlatGradBucket(model.parameters())
gen = bench.make_scene('fern', dev)
batches = [bench.make_batch(gen, 1024, 1000 + i) for i in range(4)]
stepper = None
if api == 'onecall':
    from vipnerf_hip.step import FusedTrainStep
    stepper = FusedTrainStep(model, cfg, opt)


def step(i):
    src = batches[i % 4]
    b = dict(src)
    b['common_data'] = {'poses': src['common_data']['poses']}
    b['iter_num'] = 40000 + i
    if stepper is not None:
        stepper(b)
        return
    bucket.release()
    out = model(b)
    losses = lossc.compute_losses(b, out)
    losses['TotalLoss'].backward()
    opt.step()


for i in range(4):
    step(i)
torch.cuda.synchronize()
STEPS = 6
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(STEPS):
        step(4 + i)
    torch.cuda.synchronize()
ev = prof.events()
byname = collections.Counter()
stacks = collections.defaultdict(collections.Counter)
lib = collections.Counter()
for e in ev:
    if e.device_type is not None and 'CUDA' in str(e.device_type):
        name = e.name
        if name.startswith('vn::') or ' vn::' in name:
            lib[name.split('(')[0][:60]] += 1
            continue
        byname[name[:110]] += 1
# CPU-side ops that launch device work: aten ops with a stack
for e in ev:
    if 'CPU' in str(e.device_type) and e.name.startswith('aten::') and e.stack:
        fr = [f for f in e.stack if '/repo/' in f or 'bench' in f][:3]
        stacks[e.name][' <- '.join(os.path.basename(f.split(',')[0]) + ':' + f.split('(')[-1].split(')')[0][-4:] if False else f[-70:] for f in fr)] += 1
print(f'== {prec} {api}: device activities per step that are not the library\'s kernels ({STEPS} steps profiled)')
for k, v in byname.most_common():
    print('%6.2f  %s' % (v / STEPS, k))
print('== library kernels per step: %d launches of %d kinds' % (sum(lib.values()) // STEPS, len(lib)))
for k, v in sorted(lib.items(), key=lambda kv: -kv[1]):
    print('%6.2f  %s' % (v / STEPS, k))
print('== aten ops called from repo code (per step), with the calling frames')
for op, c in sorted(stacks.items(), key=lambda kv: -sum(kv[1].values())):
    tot = sum(c.values())
    if tot < STEPS:
        continue
    print('%6.2f  %s' % (tot / STEPS, op))
    for st, n in c.most_common(3):
        print('          %5.2f  %s' % (n / STEPS, st))
