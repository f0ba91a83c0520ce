"""Achievable HBM write / read / copy bandwidth on this box (torch kernels, 4 GiB buffers): the ceilings the MLP kernels'
activation stores and the weight-gradient kernels' operand streams are up against."""
import time, torch
dev = torch.device('cuda:0')
n = 1 << 30                                   # 4 GiB of fp32
x = torch.empty(n, dtype=torch.float32, device=dev)
y = torch.empty(n, dtype=torch.float32, device=dev)
def t(f, reps=5):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
w = t(lambda: x.zero_())
r = t(lambda: x.sum())
c = t(lambda: y.copy_(x))
print('write (zero_)  %.2f TB/s' % (4 * n / w / 1e12))
print('read  (sum)    %.2f TB/s' % (4 * n / r / 1e12))
print('copy  (r + w)  %.2f TB/s of traffic' % (8 * n / c / 1e12))
