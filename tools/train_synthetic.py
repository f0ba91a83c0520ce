"""Trains the synthetic 3-camera plane scene of tests/test_hip_train_e2e.py with TrainerHip01 in one arithmetic and prints the
loss / PSNR trajectory:  python tools/train_synthetic.py [precision=fp16x3] [iterations=1500] [image side=64] [rays per iteration=1024]"""
import os, sys, time, tempfile
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'vip-nerf_amd'), os.path.join(ROOT, 'vip-nerf_amd', 'src')):
    sys.path.insert(0, p)
import test_hip_train_e2e as e2e
from TrainerHip01 import TrainerHip
from data_preprocessors.RayGeneratorHip01 import BatchIndexScheduler, RayGeneratorHip
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp16x3'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
side = int(sys.argv[3]) if len(sys.argv) > 3 else 64            # image side; focal length scales with it
rays = int(sys.argv[4]) if len(sys.argv) > 4 else 1024           # rays per iteration
# under torchrun (python -m torch.distributed.run --nproc-per-node N tools/train_synthetic.py ...): one rank per GPU, every rank builds
# the same batch schedule and renders its row-class-aware shard; one all-reduce of the flat gradient bucket per iteration
from vipnerf_hip import dist as vdist
rank, world, local = vdist.init_from_env()
dev = torch.device('cuda', local % torch.cuda.device_count())
torch.cuda.set_device(dev)
n, h, w = 3, side, side
K, poses, images_u8 = e2e.synthetic_scene(n, h, w, f=80.0 * side / 64)
images = torch.from_numpy(images_u8.astype(np.float32) / 255)
torch.manual_seed(0); np.random.seed(0)
cfg = e2e.configs(iters, prec)
cfg['model_save_interval'] = 0
cfg['validation_interval'] = 250 if iters <= 3000 else 1000
gen = RayGeneratorHip((h, w), K[None], poses, 2.0, 4.0, False, dev, images=images, visibility_prior=torch.ones(n, n - 1, h, w))
if rays > 1024 or os.environ.get('ONE_SUB'):
    cfg['sub_batch_size'] = 0      # one sub-batch (the default configuration splits its 1024 rays into two of 512)
cfg['one_call_step'] = os.environ.get('ONE_CALL', '1') != '0'     # ONE_CALL=0: the module-contract sequence instead of vipnerf_train_step
tr = TrainerHip(cfg, gen, BatchIndexScheduler(n, h, w, num_rays=rays), output_dirpath=tempfile.mkdtemp(), rank=rank, world=world)
torch.cuda.synchronize(); t0 = time.time()
hist = tr.train()
torch.cuda.synchronize(); dt = time.time() - t0
mse = np.array([x['MSEHip01'] for x in hist])
psnr = [(i + 1, round(x['validation_psnr'], 2)) for i, x in enumerate(hist) if 'validation_psnr' in x]
if world > 1:
    torch.distributed.barrier()
if rank == 0:
  print((f'[{world} ranks] ' if world > 1 else '') + ('one-call step, ' if tr.stepper is not None and cfg['sub_batch_size'] == 0 else 'module contract, ') + f'{prec}: {iters} iterations of {rays} rays ({side} x {side} images) in {dt:.1f} s ({iters * rays / dt / 1e3:.0f} k rays/s incl. validation renders); '
        f'MSE {mse[:10].mean():.4f} -> {mse[-50:].mean():.5f}; all finite: {bool(np.isfinite(mse).all())}; PSNR of the training views: {psnr}')
