"""Per-workgroup time of the eval MLP kernel vs how many CUs are busy (discriminates per-CU vs chip-wide limits)."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd'))
from oracle import vipnerf_oracle as vo
from vipnerf_hip import ops
dev = torch.device('cuda:0'); cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
pa = vo.init_params(3, levels=('coarse',))
rs = np.random.default_rng(0)
for prec in (0, 2, 1):
    pk = ops.pack_weights([cu(pa['coarse_model.' + n]) for n in ops.PARAM_ORDER], precision=prec)
    for nwg in ((64, 2048) if os.environ.get("WG_SHORT") else (8, 64, 256, 512, 1024, 4096)):
        P = 128 * nwg
        pts = torch.rand(P, 3, device=dev) * 2 - 1; vd = torch.nn.functional.normalize(torch.randn(P, 3, device=dev), dim=-1)
        for _ in range(2): ops.mlp_forward(pk, pts, vd, precision=prec)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5): ops.mlp_forward(pk, pts, vd, precision=prec)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        rounds = max(1, nwg / 256)
        print('prec=%d  workgroups=%5d  %.3f ms  -> %.1f us per round of <=256 WGs' % (prec, nwg, ms, ms * 1e3 / rounds))
