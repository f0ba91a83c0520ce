"""Why does one of EIGHT processes that open the one GPU at the same moment sometimes die with SIGABRT before its first kernel?
(tests/test_hip_dist.py retries such a start once: seen about once in twenty starts.)  Starts N rounds of 8 processes, each of which opens the
device, loads libvipnerf_hip.so and runs one small forward pass, with the HIP runtime's own log (AMD_LOG_LEVEL=3) captured per process; a
process that dies by a signal gets the tail of its log and its stderr saved under gpurun_out/spawn8/.
    python tools/spawn8_probe.py [rounds=40]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out', 'spawn8'); os.makedirs(OUT, exist_ok=True)
CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.path.join(%r, 'vip-nerf_amd')); sys.path.insert(0, %r)
torch.cuda.set_device(0)
x = torch.zeros(1024, device='cuda') + 1
from vipnerf_hip import ops, _lib
_lib.load()
import numpy as np
from oracle import vipnerf_oracle as vo
pa = vo.init_params(3, levels=('coarse',))
pk = ops.pack_weights([torch.from_numpy(np.ascontiguousarray(pa['coarse_model.' + k])).cuda() for k in ops.PARAM_ORDER])
o = ops.mlp_forward(pk, torch.rand(256, 3, device='cuda'), torch.rand(256, 3, device='cuda'))
torch.cuda.synchronize()
print('ok', float(o['sigma'].sum()))
''' % (ROOT, ROOT)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
level = os.environ.get('PROBE_LOG_LEVEL', '3')
died = []
t0 = time.time()
for r in range(rounds):
    procs = []
    for k in range(8):
        env = dict(os.environ, AMD_LOG_LEVEL=level, HSA_ENABLE_IPC_MODE_LEGACY='0')
        err = open(f'/tmp/spawn8_{k}.err', 'w')
        procs.append((subprocess.Popen([sys.executable, '-c', CHILD], stdout=subprocess.PIPE, stderr=err, env=env, text=True), err))
    for k, (p, err) in enumerate(procs):
        out, _ = p.communicate(timeout=300)
        err.close()
        if p.returncode != 0:
            died.append((r, k, p.returncode))
            txt = open(f'/tmp/spawn8_{k}.err', errors='replace').read()
            with open(os.path.join(OUT, f'round{r:02d}_proc{k}_rc{p.returncode}.log'), 'w') as f:
                f.write(txt[-20000:])
            print(f'round {r} process {k}: rc {p.returncode}; stdout {out!r}; last lines of its log:\n' + '\n'.join(txt.splitlines()[-12:]), flush=True)
print(f'{rounds} rounds x 8 processes in {time.time() - t0:.0f} s: {len(died)} died: {died}')
