"""Trains the synthetic 3-camera plane scene of tests/test_hip_train_e2e.py with TrainerHip01 in one arithmetic and prints the
loss / PSNR trajectory:  python tools/train_synthetic.py [precision=fp16x3] [iterations=1500]"""
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
dev = torch.device('cuda:0')
n, h, w = 3, 64, 64
K, poses, images_u8 = e2e.synthetic_scene(n, h, w, f=80.0)
images = torch.from_numpy(images_u8.astype(np.float32) / 255)
torch.manual_seed(0); np.random.seed(0)
cfg = e2e.configs(iters, prec)
cfg['model_save_interval'] = 0
cfg['validation_interval'] = 250
gen = RayGeneratorHip((h, w), K[None], poses, 2.0, 4.0, False, dev, images=images, visibility_prior=torch.ones(n, n - 1, h, w))
tr = TrainerHip(cfg, gen, BatchIndexScheduler(n, h, w, num_rays=1024), output_dirpath=tempfile.mkdtemp())
torch.cuda.synchronize(); t0 = time.time()
hist = tr.train()
torch.cuda.synchronize(); dt = time.time() - t0
mse = np.array([x['MSEHip01'] for x in hist])
psnr = [(i + 1, round(x['validation_psnr'], 2)) for i, x in enumerate(hist) if 'validation_psnr' in x]
print(f'{prec}: {iters} iterations of 1024 rays in {dt:.1f} s ({iters * 1024 / dt / 1e3:.0f} k rays/s incl. validation renders); '
      f'MSE {mse[:10].mean():.4f} -> {mse[-50:].mean():.5f}; all finite: {bool(np.isfinite(mse).all())}; PSNR of the training views: {psnr}')
