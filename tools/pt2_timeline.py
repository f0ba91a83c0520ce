"""Where one workgroup of k_mlp_fwd_pt2 spends its cycles (experiment build -DVN_EXP=50: s_memtime at the phase boundaries, per wave):
    tools/build_variant.sh exp50 "-DVN_EXP=50" vipnerf_mlp_fwd_pt2
    VIPNERF_HIP_LIB=vip-nerf_amd/lib/libvipnerf_hip_exp50.so python tools/pt2_timeline.py [eval|train]
The recorded workgroup is the one in the middle of the FINE level's grid of the last launch."""
import ctypes as C, os, sys, warnings
warnings.filterwarnings('ignore')
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd'))
from oracle import vipnerf_oracle as vo
from vipnerf_hip import _lib as L, ops
mode = sys.argv[1] if len(sys.argv) > 1 else 'eval'
dev = torch.device('cuda:0'); cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
prec = ops.PRECISIONS[os.environ.get('HIP_PRECISION', 'bf16')]
n = int(os.environ.get('RAYS', 8192))
b = vo.synthetic_batch(n, 7, scene='fern', nf=2)
bd = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
pa = vo.init_params(3)
pc = ops.pack_weights([cu(pa['coarse_model.' + k]) for k in ops.PARAM_ORDER], precision=prec)
pf = ops.pack_weights([cu(pa['fine_model.' + k]) for k in ops.PARAM_ORDER], precision=prec)
train = mode == 'train'
V = 1 if train else 0
cfg = ops.make_config(True, 64, 128, V, train, noise_std=1.0 if train else 0.0, precision=prec, save_acts=train)
if train:
    bd['rays_o2'] = torch.zeros(n, 1, 3, device=dev) + torch.tensor([0.2, 0.0, 0.0], device=dev)
acts = None
if train:
    ab, _ = ops.query_workspace(cfg, n)
    acts = torch.empty(ab // 4, dtype=torch.float32, device=dev)
for _ in range(3):
    ops.render_forward(cfg, bd, {'seed': 1, 'offset': 2} if train else None, pc, pf, acts)
torch.cuda.synchronize()
lib = L.load()
lib.vipnerf_exp_timeline.restype = C.c_int
buf = (C.c_ulonglong * 1024)()
assert lib.vipnerf_exp_timeline(buf, 1024) == 0
t = np.array(buf, dtype=np.uint64).reshape(8, 128).astype(np.int64)
labels = ['entry', 'resident', 'pe', ('L0', 1)] + [(f'L{l}.{j}', 1) for l in range(1, 9) for j in ((0, 1, 'pe') if l == 5 else (0, 1))]
print(f'{mode} {os.environ.get("HIP_PRECISION", "bf16")}: cycles (s_memtime ticks) of the recorded workgroup, per wave')
rows = []
for w in range(8):
    x = t[w]; i = 3
    rec = {'resident': x[1] - x[0], 'pe': x[2] - x[1], 'wait': 0, 'gemm': 0, 'between': 0}
    stages = []
    for s in range(1 + 16 + 1):            # layer 0, 8 layers x 2 stages, layer 5's gamma(x) stage
        pre, post, end = x[i], x[i + 1], x[i + 2]; i += 3
        stages.append((post - pre, end - post))
        rec['wait'] += post - pre; rec['gemm'] += end - post
        if s: rec['between'] += pre - prev_end
        prev_end = end
    pre, post, vdone, last = x[i], x[i + 1], x[i + 2], x[i + 3]
    rec['between'] += pre - prev_end; rec['wait'] += post - pre; rec['gemm'] += vdone - post
    rec['tail'] = last - vdone; rec['total'] = last - x[0]
    rows.append(rec)
    print(f'wave {w}: total {rec["total"]:7d} | resident load {rec["resident"]:5d}  gamma(x) {rec["pe"]:5d} | 19 stages: MFMA loops {rec["gemm"]:7d}  waits+barriers {rec["wait"]:6d}  '
          f'between stages (epilogues, sigma head, operand reloads) {rec["between"]:6d} | view tail {rec["tail"]:6d}')
    if w in (0, 4):
        print('        per stage (wait, loop):', ' '.join(f'{a}/{b}' for a, b in stages))
m = {k: int(np.mean([r[k] for r in rows])) for k in rows[0]}
print('mean over waves:', m, '| shares of total:', {k: round(v / m['total'], 3) for k, v in m.items() if k != 'total'})
print('MFMA pipe time of the workgroup per SIMD at 16 cycles per MFMA: 2 waves x 2336 MFMAs x 16 =', 2 * 2336 * 16)
