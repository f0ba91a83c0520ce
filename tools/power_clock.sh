# Shader clock and socket power while an arithmetic's EVAL kernels (or the training step) run back to back:
#   tools/power_clock.sh "fp32 fp16x3 fp16"      (on the GPU box; rocm-smi sampled once a second next to the load)
for P in $1; do
  HIP_PRECISION=$P RAYS=32768 ITERS=400 python - <<'PY' &
import os, sys, numpy as np, torch, time
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd'))
from oracle import vipnerf_oracle as vo
from vipnerf_hip import ops
dev = torch.device('cuda:0'); cu = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
prec = ops.PRECISIONS[os.environ['HIP_PRECISION']]
b = vo.synthetic_batch(int(os.environ['RAYS']), 7, scene='fern', nf=2)
bd = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
pa = vo.init_params(3)
pc = ops.pack_weights([cu(pa['coarse_model.' + n]) for n in ops.PARAM_ORDER], precision=prec)
pf = ops.pack_weights([cu(pa['fine_model.' + n]) for n in ops.PARAM_ORDER], precision=prec)
cfg = ops.make_config(True, 64, 128, 0, False, precision=prec)
t0 = time.time()
while time.time() - t0 < 14:
    for _ in range(5):
        ops.render_forward(cfg, bd, None, pc, pf)
    torch.cuda.synchronize()
PY
  PID=$!
  sleep 7
  echo "== $P (eval kernels, 32768 rays per call, back to back)"
  for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power|power" | tr '\n' ' '; echo; sleep 1; done
  wait $PID
done
