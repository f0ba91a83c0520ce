"""Short training runs from the same weights and batches in several arithmetics: the losses must track the exact-fp32
path (identical to printing precision over the first steps, both decreasing) -- an end-to-end check that the fast modes
train the same model.  python tools/train_compare.py [steps]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd')); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd', 'src'))
import bench
from oracle import vipnerf_oracle as vo          # synthetic batches only
from models.ModelFactory import get_model
from loss_functions.LossComputerHip01 import LossComputerHip
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device('cuda:0')
batches = [bench.make_batch_oracle(vo, 1024, 5000 + (i % 8), dev) for i in range(8)]   # 8 batches of one synthetic scene, cycled
curves = {}
for prec in ('fp32', 'fp16x3', 'fp16x3h', 'fp16', 'bf16'):
    cfg = bench.model_configs(); cfg['model']['hip_precision'] = prec
    torch.manual_seed(0)
    model = get_model(cfg, None).to(dev).train()
    lossc = LossComputerHip(cfg)
    opt = torch.optim.Adam(model.parameters(), lr=5e-4, fused=True)
    torch.manual_seed(1)
    vals = []
    for i in range(steps):
        b = dict(batches[i % 8]); b['common_data'] = {'poses': batches[i % 8]['common_data']['poses']}
        opt.zero_grad(set_to_none=True)
        loss = lossc.compute_losses(b, model(b))['TotalLoss']
        loss.backward(); opt.step()
        vals.append(loss.detach())
    curves[prec] = torch.stack(vals).cpu()
    assert torch.isfinite(curves[prec]).all(), prec
ref = curves['fp32']
print('step   ' + '  '.join('%-12s' % k for k in curves))
for i in list(range(0, min(steps, 10))) + list(range(19, steps, 20)):
    print('%5d  ' % (i + 1) + '  '.join('%-12.6f' % float(c[i]) for c in curves.values()))
for k, c in curves.items():
    d = ((c - ref).abs() / ref.abs())
    print('%-8s first 10 steps: max rel. deviation from fp32 %.2e; mean of last 20 losses %.5f (first %.5f)' % (k, float(d[:10].max()), float(c[-20:].mean()), float(c[0])))
