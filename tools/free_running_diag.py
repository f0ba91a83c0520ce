import os, sys, numpy as np, torch
ROOT='/root/repo'
for p in (ROOT, ROOT+'/tests', ROOT+'/vip-nerf_amd', ROOT+'/vip-nerf_amd/src'): sys.path.insert(0,p)
import test_hip_parity as tp
from oracle import vipnerf_oracle as vo
dev=torch.device('cuda:0')
g=tp.load('f4_eval_fern')
b=vo.synthetic_batch(int(g['n']), int(g['seed_batch']), scene='fern', nf=2)
params=vo.init_params(int(g['seed_params']), scale=float(g['scale_params']), sigma_bias=float(g['sigma_bias']))
model,_=tp.make_model(dev, True, params); model.eval()
with torch.no_grad(): plain=model(tp.ref_batch(b,dev,0), retraw=True)
inds=model.last_extras['sample_inds'].cpu().long(); ref=torch.from_numpy(g['plain_sample_inds'].astype(np.int64))
mm=(inds!=ref)
print('mismatch cols', mm.nonzero()[:, 1].tolist()); print('hip', inds[mm].tolist()); print('ref', ref[mm].tolist())
zf=plain['z_vals_fine'].cpu(); zr=torch.from_numpy(g['out_z_vals_fine'])
print('z_vals_fine max abs diff', float((zf-zr).abs().max()), 'bit-equal frac', float((zf==zr).float().mean()))
for tag in ('llff','dtu'):
    g=tp.load(f'f5_train_{tag}')
    b=vo.synthetic_batch(int(g['n']), int(g['seed_batch']), scene=str(g['scene']), nf=int(g['nf']), n_sparse=int(g['n_sparse']))
    params=vo.init_params(int(g['seed_params']), scale=float(g['scale_params']))
    model,cfg=tp.make_model(dev,b['ndc'],params); model.train()
    model.injected_rng={k[4:]: tp.cu(v,dev) for k,v in g.items() if k.startswith('rng_')}
    with torch.no_grad(): out=model(tp.ref_batch(b,dev,40000))
    inds=model.last_extras['sample_inds'].cpu().long(); ref=torch.from_numpy(g['sample_inds'].astype(np.int64))
    mm=(inds!=ref); print(tag,'agree',float((~mm).float().mean()), 'mismatches', int(mm.sum()), 'cols', mm.nonzero()[:,1].tolist()[:20])
    zf=out['z_vals_fine'].cpu(); zr=torch.from_numpy(g['out_z_vals_fine']); print(' z_vals_fine max abs diff', float((zf-zr).abs().max()), 'bit-equal', float((zf==zr).float().mean()))
    e=(out['rgb_fine'].cpu()-torch.from_numpy(g['out_rgb_fine'])).abs().max(-1).values; print(' rgb_fine max err', float(e.max()), 'beyond 1e-4', float((e>1e-4).float().mean()))
