import os, sys, torch
sys.path.insert(0, '/root/repo/vip-nerf_amd')
from vipnerf_hip import ops
dev = torch.device('cuda:0')
torch.manual_seed(21)
n = 1 << 20
p0 = torch.randn(n, device=dev) * 0.1
g = torch.randn(n, device=dev) * 10.0 ** torch.randint(-9, 3, (n,), device=dev).float()
ref = torch.nn.Parameter(p0.clone())
opt = torch.optim.Adam([ref], lr=5e-4, betas=(0.9, 0.999), eps=1e-8, foreach=False, fused=False)
m0 = torch.randn(n, device=dev) * 0.01; v0 = torch.rand(n, device=dev) * 1e-3
ref.grad = g.clone(); opt.step()          # step 1 from zero state
st = opt.state[ref]
# second step from a non-trivial state
ref.grad = (g * 0.5).clone(); 
pm, mm, vm = ref.detach().clone(), st['exp_avg'].clone(), st['exp_avg_sq'].clone()
opt.step()
for mask in range(8):
    p, m, v = pm.clone(), mm.clone(), vm.clone()
    ops.adam_step(p, m, v, g * 0.5, 5e-4, 0.9, 0.999, 1e-8, 2, fma_mask=mask)
    print(mask, 'm eq %.6f' % (m == st['exp_avg']).float().mean().item(), 'v eq %.6f' % (v == st['exp_avg_sq']).float().mean().item(),
          'p eq %.6f' % (p == ref.detach()).float().mean().item())
# pieces: denominators
v1 = st['exp_avg_sq']
bc2 = 1 - 0.999 ** 2
d_t = (v1.sqrt() / (bc2 ** 0.5)).add_(1e-8)
import numpy as np
inv = float(np.float32(1.0) / np.float32(bc2 ** 0.5))
d_a = v1.sqrt() * inv + 1e-8
d_b = v1.sqrt() / np.float32(bc2 ** 0.5).item() + 1e-8
print('denom: mul-by-inverse eq %.6f' % (d_a == d_t).float().mean().item(), ' true div eq %.6f' % ((v1.sqrt() / torch.tensor(bc2 ** 0.5, device=dev, dtype=torch.float32)).add(1e-8) == d_t).float().mean().item())
# is torch's device division / square root correctly rounded?  (numpy float32 arithmetic is)
a = (torch.randn(n, device=dev) * 10.0 ** torch.randint(-6, 3, (n,), device=dev).float())
b = (torch.rand(n, device=dev) + 1e-3) * 10.0 ** torch.randint(-6, 3, (n,), device=dev).float()
q_gpu = (a / b).cpu().numpy(); q_np = a.cpu().numpy() / b.cpu().numpy()
s_gpu = b.sqrt().cpu().numpy(); s_np = np.sqrt(b.cpu().numpy())
print('torch device division == IEEE: %.6f   sqrt == IEEE: %.6f' % ((q_gpu == q_np).mean(), (s_gpu == s_np).mean()))
ac = torch.addcdiv(torch.zeros(n, device=dev), a, b, value=1.0).cpu().numpy()
print('addcdiv(0, a, b, 1) == IEEE a / b: %.6f ; == torch a / b: %.6f' % ((ac == q_np).mean(), (ac == q_gpu).mean()))
# the five-step sequence of tests/test_hip_fullsize.py::test_fused_adam_step_is_torch_adam, per step
torch.manual_seed(21)
n = 1191946
p0 = torch.randn(n, device=dev) * 0.1
grads = []
for it in range(5):
    gg = torch.randn(n, device=dev) * 10.0 ** torch.randint(-9, 3, (n,), device=dev).float()
    gg[torch.rand(n, device=dev) < 0.05] = 0.0
    grads.append(gg)
lrs = [5e-4 * 0.97 ** it for it in range(5)]
ref = torch.nn.Parameter(p0.clone())
opt = torch.optim.Adam([ref], lr=5e-4, betas=(0.9, 0.999), eps=1e-8, foreach=False, fused=False)
p, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
for it in range(5):
    ref.grad = grads[it].clone(); opt.param_groups[0]['lr'] = lrs[it]; opt.step()
    st = opt.state[ref]
    ops.adam_step(p, m, v, grads[it], lrs[it], 0.9, 0.999, 1e-8, it + 1, fma_mask=7)
    bad = (p != ref.detach())
    print('step', it + 1, 'm eq %.7f v eq %.7f p eq %.7f' % ((m == st['exp_avg']).float().mean().item(), (v == st['exp_avg_sq']).float().mean().item(), (~bad).float().mean().item()))
    if bad.any():
        i = bad.nonzero()[0, 0]
        print('   first bad: g %.9e m %.9e v %.9e p_mine %.9e p_torch %.9e' % (grads[it][i].item(), m[i].item(), v[i].item(), p[i].item(), ref.detach()[i].item()))
    p.copy_(ref.detach()); m.copy_(st['exp_avg']); v.copy_(st['exp_avg_sq'])
print('--- scalar forensics per step')
for step in range(1, 6):
    bc1, bc2 = 1 - 0.9 ** step, 1 - 0.999 ** step
    vv = torch.rand(1 << 20, device=dev) * 10.0 ** torch.randint(-12, 0, (1 << 20,), device=dev).float()
    mm = torch.randn(1 << 20, device=dev) * 0.1
    pp = torch.randn(1 << 20, device=dev) * 0.1
    d_t = (vv.sqrt() / (bc2 ** 0.5)).add_(1e-8)
    inv = float(np.float32(1.0 / bc2 ** 0.5))
    d_m = (vv.sqrt() * inv).add_(1e-8)
    d_div = (vv.sqrt() / torch.tensor(bc2 ** 0.5, dtype=torch.float32, device=dev)).add_(1e-8)
    lr = 5e-4 * 0.97 ** (step - 1)
    p_t = pp.clone().addcdiv_(mm, d_t, value=-(lr / bc1))
    a32 = float(np.float32(-(lr / bc1)))
    q = mm / d_t
    p_fma = torch.addcmul(pp, q, torch.ones_like(q), value=a32)          # pp + a32 * (q * 1): contracted like the kernel's fma?
    p_nofma = pp + a32 * q
    print(step, 'denom: mul-by-inverse %.6f true div %.6f | p: fma-like %.6f  mul-then-add %.6f' % ((d_m == d_t).float().mean().item(), (d_div == d_t).float().mean().item(),
          (p_fma == p_t).float().mean().item(), (p_nofma == p_t).float().mean().item()))
