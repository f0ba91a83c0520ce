"""Soak: N training steps with changing batch sizes, iteration numbers, secondary-view counts and arithmetics through the module +
fused losses + Adam; asserts finite losses and that PyTorch's allocated / reserved memory stops growing (no leak through the
autograd contexts, the workspace cache or the profiling scopes).   python tools/soak.py [steps=1500]"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd')); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd', 'src'))
import bench
from oracle import vipnerf_oracle as vo          # synthetic batches only
from models.ModelFactory import get_model
from loss_functions.LossComputerHip01 import LossComputerHip
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device('cuda:0')
rs = np.random.default_rng(0)
cfg = bench.model_configs()
model = get_model(cfg, None).to(dev).train()
lossc = LossComputerHip(cfg)
opt = torch.optim.Adam(model.parameters(), lr=5e-4, fused=True)
sizes = [1, 37, 256, 1000, 1024, 2048, 3000, 4096]
batches = {(n, nf): bench.make_batch_oracle(vo, n, 7 + n + nf, dev) for n in sizes for nf in (2,)}
for (n, nf), b in list(batches.items()):
    pass
precs = ["fp32", "fp16x3", "fp16", "bf16", "fp16x3h"]
mem = []
t0 = time.time()
for i in range(steps):
    n = sizes[rs.integers(len(sizes))]
    b = dict(batches[(n, 2)]); b['common_data'] = {'poses': batches[(n, 2)]['common_data']['poses']}
    b['iter_num'] = int(rs.integers(0, 50000))
    cfg['model']['hip_precision'] = precs[(i // 50) % len(precs)]
    if i % 97 == 0:
        model.eval()
        with torch.no_grad():
            out = model(b)
        assert torch.isfinite(out['rgb_fine']).all()
        model.train()
        continue
    opt.zero_grad(set_to_none=True)
    loss = lossc.compute_losses(b, model(b))['TotalLoss']
    loss.backward()
    opt.step()
    if i % 100 == 99:
        torch.cuda.synchronize()
        assert torch.isfinite(loss).all(), i
        mem.append((torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20))
torch.cuda.synchronize()
print(f'{steps} steps in {time.time() - t0:.1f} s; (allocated, reserved) MiB every 100 steps: {mem}')
half = len(mem) // 2
assert max(m[0] for m in mem[half:]) <= max(m[0] for m in mem[:half]) + 64, 'allocated memory keeps growing'
assert max(m[1] for m in mem[half:]) <= max(m[1] for m in mem[:half]) + 1024, 'reserved memory keeps growing'
print('soak ok')
