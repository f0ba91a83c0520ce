"""Where one workgroup of k_mlp_fwd_pt2 / k_mlp_bwd_pt2 spends its cycles (experiment build -DVN_EXP=50: s_memtime at the phase boundaries,
per wave, tagged -- vipnerf_mlp_pt2.h TS_AT):
    tools/build_variant.sh exp50 "-DVN_EXP=50" vipnerf_mlp_fwd_pt2 vipnerf_mlp_bwd_pt2
    VIPNERF_HIP_LIB=vip-nerf_amd/lib/libvipnerf_hip_exp50.so [HIP_PRECISION=fp32] python tools/pt2_timeline.py [eval|train|bwd]
(HIP_PRECISION=fp32: the 16-point exact-fp32 forward kernel k_mlp_fwd_bf16n, eval / train)
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
train = mode in ('train', 'bwd', 'wg')
V = 1 if train else 0
cfg = ops.make_config(True, 64, 128, V, train, noise_std=1.0 if train else 0.0, precision=prec, save_acts=train)
if train:
    bd['rays_o2'] = torch.zeros(n, 1, 3, device=dev) + torch.tensor([0.2, 0.0, 0.0], device=dev)
acts = None
if train:
    ab, bb = ops.query_workspace(cfg, n)
    acts = torch.empty(ab // 4, dtype=torch.float32, device=dev)
for _ in range(3):
    c, f, _ = ops.render_forward(cfg, bd, {'seed': 1, 'offset': 2} if train else None, pc, pf, acts)
if mode in ('bwd', 'wg'):
    if mode == 'bwd': os.environ['VIPNERF_EXP_SKIP_WGRAD'] = '1'
    bwd_ws = torch.empty(bb // 4, dtype=torch.float32, device=dev)
    shapes = ops.param_shapes(ops.topology_of(cfg))
    gc = [torch.zeros(s, device=dev) for s in shapes]; gf = [torch.zeros(s, device=dev) for s in shapes]
    gen = torch.Generator(device=dev).manual_seed(3)
    g_rgb = torch.randn(n, 3, device=dev, generator=gen) * 1e-4
    g = {'rgb': g_rgb, 'visibility': torch.randn(n, 64, device=dev, generator=gen) * 1e-5, 'vis2': torch.randn(n, V, device=dev, generator=gen) * 1e-4}
    g['raw_vis'] = g['visibility']
    gfine = dict(g, visibility=torch.randn(n, 192, device=dev, generator=gen) * 1e-5); gfine['raw_vis'] = gfine['visibility']
    for _ in range(3):
        ops.render_backward(cfg, bd, pc, pf, c, f, g, gfine, acts, bwd_ws, gc, gf)
torch.cuda.synchronize()
lib = L.load()
if mode == 'wg':          # the exact-fp32 256 x 256 weight-gradient kernel: per 32-point block -- loads issued / MFMA loop / LDS stores / barrier
    fn = lib.vipnerf_exp_timeline_wg; fn.restype = C.c_int
    buf = (C.c_ulonglong * 2048)(); assert fn(buf, 2048) == 0
    raw = np.array(buf, dtype=np.uint64).reshape(8, 256)
    for w in range(8):
        ev = [(int(x >> np.uint64(56)), int(x & np.uint64((1 << 56) - 1))) for x in raw[w] if x][:-1]
        acc = {'top (next block loads issued)': [], 'mfma loop': [], 'lds stores': [], 'barrier': []}
        prev = ev[0][1]
        for tag, t in ev[1:]:
            key = {4: 'top (next block loads issued)', 5: 'mfma loop', 8: 'lds stores', 3: 'barrier'}.get(tag)
            if key: acc[key].append(t - prev)
            prev = t
        print(f'wave {w}:', {k: (int(np.mean(v[2:])) if len(v) > 2 else None) for k, v in acc.items()}, 'blocks', len(acc['mfma loop']))
    sys.exit(0)
narrow = os.environ.get('HIP_PRECISION', 'bf16') not in ('bf16', 'fp16')      # the 16-point kernels (exact fp32, split arithmetics): forward only
fn = ((lib.vipnerf_exp_timeline_f32b if (os.environ.get('HIP_PRECISION') == 'fp32' and hasattr(lib, 'vipnerf_exp_timeline_f32b')) else lib.vipnerf_exp_timeline_nb) if mode == 'bwd' else (lib.vipnerf_exp_timeline_f32f if (os.environ.get('HIP_PRECISION') == 'fp32' and hasattr(lib, 'vipnerf_exp_timeline_f32f')) else lib.vipnerf_exp_timeline_n)) if narrow else (lib.vipnerf_exp_timeline_bwd if mode == 'bwd' else lib.vipnerf_exp_timeline)
fn.restype = C.c_int
buf = (C.c_ulonglong * 2048)()
assert fn(buf, 2048) == 0
raw = np.array(buf, dtype=np.uint64).reshape(8, 256)
ENTRY, RESIDENT, HEAD, PRE, POST, END, VIEW, LAST, EPI_A, EPI_B = range(10)
print(f'{mode} {os.environ.get("HIP_PRECISION", "bf16")}: cycles (s_memtime ticks) of the recorded workgroup, per wave')
rows = []
for w in range(8):
    ev = [(int(x >> np.uint64(56)), int(x & np.uint64((1 << 56) - 1))) for x in raw[w] if x]
    assert ev[0][0] == ENTRY and ev[-1][0] == LAST, [e[0] for e in ev]
    rec = {'resident': 0, 'head': 0, 'wait': 0, 'gemm': 0, 'between': 0, 'tail': 0, 'epi_operands': 0, 'epi_valu': 0}
    stages = []; fine = {}
    prev_tag, prev_t = ev[0]
    for tag, t in ev[1:]:
        d = t - prev_t
        if tag == RESIDENT: rec['resident'] += d
        elif tag == HEAD: rec['head'] += d                  # forward: gamma(x); backward: the heads and the view hidden layer of every direction
        elif tag == PRE: rec['between'] += d                # epilogue of the layer before, accumulator set-up, operand reloads
        elif tag == POST: rec['wait'] += d; stages.append([d, 0])
        elif tag in (END, VIEW): rec['gemm'] += d; stages[-1][1] = d
        elif tag == EPI_A: rec['epi_operands'] += d; rec['between'] += d      # (backward) the epilogue's operands ready: last accumulators, ReLU bits
        elif tag == EPI_B: rec['epi_valu'] += d; rec['between'] += d           # (backward) conversion + ReLU bits applied
        elif tag == LAST: rec['tail'] += d                  # forward: per-direction view tail; backward: the last epilogue and dY_0's stores
        if tag >= 10: fine[tag] = fine.get(tag, 0) + d      # finer markers of an experiment build (the cycles up to each marker from the one before it)
        prev_tag, prev_t = tag, t
    rec['total'] = ev[-1][1] - ev[0][1]
    rows.append(rec)
    print(f'wave {w}: total {rec["total"]:7d} | resident load {rec["resident"]:5d}  head {rec["head"]:6d} | {len(stages)} stages: MFMA loops {rec["gemm"]:7d}  waits+barriers {rec["wait"]:6d}  '
          f'between stages {rec["between"]:6d} | tail {rec["tail"]:6d}')
    if fine: print('        fine markers (tag: cycles before it):', fine)
    if w in (0, 4):
        print('        per stage (wait, loop):', ' '.join(f'{a}/{b}' for a, b in stages))
m = {k: int(np.mean([r[k] for r in rows])) for k in rows[0]}
print('mean over waves:', m, '| shares of total:', {k: round(v / m['total'], 3) for k, v in m.items() if k != 'total'})
if not narrow: print('MFMA pipe time of the workgroup per SIMD at 16 cycles per MFMA: 2 waves x %d MFMAs x 16 = %d' % ((2176 if mode == 'bwd' else 2336), 2 * 16 * (2176 if mode == 'bwd' else 2336)))
