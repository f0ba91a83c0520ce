"""Which sample counts do the kernels REALLY need?  With a library built -DVN_SAMPLE_STEP=8 (check_cfg relaxed) one teacher-forced
training step per (n_coarse, n_fine, rays, arithmetic) against the oracle: worst output error, loss error, worst gradient rel L2 -- or the exception.
   tools/build_variant.sh step8 "-DVN_SAMPLE_STEP=8" vipnerf_api && VIPNERF_HIP_LIB=.../libvipnerf_hip_step8.so python tools/exp/sample_counts_probe.py"""
import os, sys, traceback
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'vip-nerf_amd'), os.path.join(ROOT, 'vip-nerf_amd', 'src'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from oracle import vipnerf_oracle as vo
import test_hip_round2 as r2
dev = torch.device('cuda:0')
CASES = [(50, 77, 33), (17, 30, 20), (64, 127, 16), (33, 31, 64), (5, 11, 40), (100, 156, 8), (63, 1, 32)]
for prec in ('fp32', 'bf16', 'fp16x3'):
    for nco, nfi, n in CASES:
        tag = '%-7s n_coarse %3d n_fine %3d rays %3d (P = %5d / %5d)' % (prec, nco, nfi, n, n * nco, n * (nco + nfi))
        try:
            b = vo.synthetic_batch(n, 303, scene='dtu', nf=3)
            params = vo.init_params(304, scale=1.6)
            rng = vo.synthetic_rng(n, nco, nfi, 305)
            cfg_o = {'ndc': b['ndc'], 'n_coarse': nco, 'n_fine': nfi, 'noise_std': 1.0, 'white_bkgd': False, 'lindisp': False}
            (ref, lref, p), (out, lh, model) = r2._oracle_and_hip_step(dev, b, params, rng, {}, cfg_o, prec=prec)
            torch.cuda.synchronize()
            zc = bool(torch.equal(out['z_vals_coarse'].cpu(), ref['z_vals_coarse']))
            eo = max(float((out[k].detach().cpu() - ref[k].detach()).abs().max() / max(float(ref[k].abs().max()), 1e-9)) for k in ref if k in out and k not in ('z_vals_coarse', 'z_vals_fine'))
            el = abs(float(lh['TotalLoss']) - float(lref['TotalLoss'])) / abs(float(lref['TotalLoss']))
            eg = max(float((t.grad.cpu() - p[k].grad).norm() / p[k].grad.norm().clamp_min(1e-30)) for k, t in model.named_parameters())
            # free-running indices as well (no injected fine depths)
            model.injected_z_fine = None
            with torch.no_grad():
                model(r2.tp.ref_batch(b, dev, 40000))
            same = float((model.last_extras['sample_inds'].cpu().long() == ref['sample_inds'].long()).float().mean())
            print('%s  z_coarse bit-exact %s  outputs %.1e  loss %.1e  grads %.1e  free-running indices equal %.5f' % (tag, zc, eo, el, eg, same), flush=True)
        except Exception as e:
            print('%s  %s: %s' % (tag, type(e).__name__, str(e)[:160]), flush=True)
