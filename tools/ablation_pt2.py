"""Store / MFMA ablation of the two-point-tile 16-bit MLP kernels, with the socket's power and shader clock beside every figure.

    VIPNERF_HIP_LIB=vip-nerf_amd/lib/libvipnerf_hip_<variant>.so python tools/ablation_pt2.py [seconds]

Runs (a) the TRAINING forward (activation stores on) and (b) the data-gradient pass alone (experiment builds honour
VIPNERF_EXP_SKIP_WGRAD=1: no weight-gradient launches) back to back for `seconds` each on a 4096-ray batch, reads the kernels' device
times from the library's HIP events and samples rocm-smi four times a second meanwhile.  One JSON line per library; the table in
profiles/r04_ablation_pt2.md is made from the lines of the variants full / no stores (VN_EXP 40) / no MFMAs (43) / neither (44)."""
import json, os, re, subprocess, sys, threading, time
os.environ.setdefault('VIPNERF_EXP_SKIP_WGRAD', '1')
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd')); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd', 'src'))
import bench
from vipnerf_hip import _lib as L, ops
from models.ModelFactory import get_model

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
prec = os.environ.get('HIP_PRECISION', 'bf16')
rays = int(os.environ.get('RAYS', 4096))
scene = os.environ.get('SCENE', 'fern')
dev = torch.device('cuda:0')


class Smi:
    """rocm-smi power / sclk samples while a loop runs."""
    def __init__(self):
        self.rows, self._stop = [], False
    def __enter__(self):
        def run():
            while not self._stop:
                try:
                    out = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True, timeout=5).stdout
                    w = re.search(r'Power \(W\):\s*([0-9.]+)', out)
                    c = re.search(r'sclk clock level:.*?\((\d+)Mhz\)', out)
                    if w and c:
                        self.rows.append((float(w.group(1)), int(c.group(1))))
                except Exception:
                    pass
                time.sleep(0.15)
        self.t = threading.Thread(target=run, daemon=True); self.t.start(); return self
    def __exit__(self, *e):
        self._stop = True; self.t.join(timeout=6)
    def summary(self):
        r = self.rows[1:] if len(self.rows) > 2 else self.rows          # the first sample may predate the load
        if not r:
            return {'samples': 0}
        pw, ck = sorted(x[0] for x in r), sorted(x[1] for x in r)
        return {'samples': len(r), 'power_w_median': pw[len(pw) // 2], 'power_w_max': pw[-1], 'sclk_mhz_median': ck[len(ck) // 2], 'sclk_mhz_min': ck[0]}


cfgd = bench.model_configs(bench.SCENES[scene][5])
cfgd['model']['hip_precision'] = prec
torch.manual_seed(0)
model = get_model(cfgd, None).to(dev).train()
b = bench.make_batch(bench.make_scene(scene, dev), rays, 1000)
V = bench.SCENES[scene][6] - 1
m = cfgd['model']
cfg = ops.make_config(bench.SCENES[scene][5], 64, 128, V, train=True, noise_std=1.0, perturb=True, precision=ops.PRECISIONS[prec], save_acts=True)
batch = {k: b[k] for k in ('rays_o', 'rays_d', 'view_dirs', 'rays_o_ndc', 'rays_d_ndc', 'near_ndc', 'far_ndc', 'near', 'far') if k in b}
batch['rays_o2'] = ops.secondary_origins(b['common_data']['poses'][0] if b['common_data']['poses'].dim() == 4 else b['common_data']['poses'], b['pixel_id'], int(b['num_frames']))
pc = ops.pack_weights(model.coarse_model.ordered_params(), cfg=cfg)
pf = ops.pack_weights(model.fine_model.ordered_params(), cfg=cfg)
ab, bb = ops.query_workspace(cfg, rays)
acts = torch.empty(ab // 4, dtype=torch.float32, device=dev)
bwd_ws = torch.empty(bb // 4, dtype=torch.float32, device=dev)
rng = {'seed': 1, 'offset': 2, 'ray_base': 0}
shapes = ops.param_shapes(ops.topology_of(cfg))
gc = [torch.zeros(s, device=dev) for s in shapes]; gf = [torch.zeros(s, device=dev) for s in shapes]


def loop(fn, seconds):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ops.profile_enable(True); ops.profile_read()
    n = 0
    with Smi() as smi:
        t0 = time.time()
        while time.time() - t0 < seconds:
            for _ in range(10):
                fn()
            torch.cuda.synchronize(); n += 10
    pr = ops.profile_read(); ops.profile_enable(False)
    return {k: round(v[1] / n, 4) for k, v in sorted(pr.items()) if v[1] / n > 0.02}, smi.summary(), n


state = {}
def fwd():
    state['c'], state['f'], _ = ops.render_forward(cfg, batch, rng, pc, pf, acts, None)
def bwd():
    c, f = state['c'], state['f']
    g = {'rgb': state['g_rgb'], 'visibility': state['g_T'], 'raw_vis': state['g_T'], 'vis2': state['g_v2']}
    gfine = {'rgb': state['g_rgb'], 'visibility': state['g_Tf'], 'raw_vis': state['g_Tf'], 'vis2': state['g_v2']}
    ops.render_backward(cfg, batch, pc, pf, c, f, g, gfine, acts, bwd_ws, gc, gf)

fwd()
g = torch.Generator(device=dev).manual_seed(3)
state['g_rgb'] = torch.randn(rays, 3, device=dev, generator=g) * 1e-4
state['g_T'] = torch.randn(rays, 64, device=dev, generator=g) * 1e-5
state['g_Tf'] = torch.randn(rays, 192, device=dev, generator=g) * 1e-5
state['g_v2'] = torch.randn(rays, V, device=dev, generator=g) * 1e-4
res = {'lib': os.path.basename(L.LIB_PATH), 'build': L.build_info().split(' arch=')[1][:60], 'precision': prec, 'rays': rays, 'scene': scene}
res['forward_ms'], res['forward_smi'], res['forward_iters'] = loop(fwd, secs)
fwd()
res['backward_ms'], res['backward_smi'], res['backward_iters'] = loop(bwd, secs)
res['skip_wgrad_honoured'] = 'wgrad_256x256' not in res['backward_ms']
print(json.dumps(res))
