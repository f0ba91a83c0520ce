"""Timing of the §8f caller-side kernels (HBM/latency-bound): ray generation + batch gather, frame post-processing,
visibility-prior generator.  Prints one JSON line per kernel with achieved GB/s (algorithmic bytes / HIP-event time)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd')); sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd', 'src'))
from vipnerf_hip import ops
from data_preprocessors.RayGeneratorHip01 import RayGeneratorHip
from prior_generators.VisibilityMaskHip02 import VisibilityWeightsComputerHip
dev = torch.device('cuda:0')
H, W, NF = 756, 1008, 2
K = np.array([[815.1316, 0, 504.], [0, 815.1316, 378.], [0, 0, 1.]], dtype=np.float32)
poses = np.tile(np.eye(4, dtype=np.float32), (NF, 1, 1)); poses[:, 0, 3] = [-0.1, 0.1]
rs = np.random.default_rng(0)
images = torch.from_numpy(rs.random((NF, H, W, 3), dtype=np.float32))
prior = torch.from_numpy((rs.random((NF, NF - 1, H, W)) < 0.5).astype(np.float32))
gen = RayGeneratorHip((H, W), K[None], poses, 1.0, 5.1731, True, dev, images=images, visibility_prior=prior)


def timed(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

idx = torch.from_numpy(rs.permutation(NF * H * W)[:4096]).to(dev)
ms = timed(lambda: gen.get_next_batch(0, idx))
print(json.dumps({'kernel': 'gen_rays batch gather (4096 rays, shuffled indices)', 'ms': round(ms, 4), 'note': 'launch-latency bound: 4096 x ~140 B'}))
ms = timed(lambda: gen.create_test_data(0))
by = H * W * (9 * 12 + 4 * 4 + 12)
print(json.dumps({'kernel': 'gen_rays full frame (762048 rays)', 'ms': round(ms, 4), 'GB/s': round(by / ms / 1e6, 1), 'bytes': by}))
out = {'rgb_fine': torch.rand(H * W, 3, device=dev), 'depth_fine': torch.rand(H * W, device=dev), 'depth_var_fine': torch.rand(H * W, device=dev),
       'depth_ndc_fine': torch.rand(H * W, device=dev), 'depth_var_ndc_fine': torch.rand(H * W, device=dev)}
ms = timed(lambda: gen.retrieve_inference_outputs(out))
by = H * W * (12 + 16 + 3 + 16)
print(json.dumps({'kernel': 'postprocess_frame (762048 px)', 'ms': round(ms, 4), 'GB/s': round(by / ms / 1e6, 1), 'bytes': by}))
f1 = rs.integers(0, 256, size=(H, W, 3)).astype(np.uint8); f2 = np.roll(f1, 5, axis=1)
comp = VisibilityWeightsComputerHip({'num_depth_planes': 64, 'temperature': 10}, dev)
E2 = np.eye(4); E2[0, 3] = 0.2
ops.profile_enable(True); ops.profile_read()
for _ in range(5): comp.compute_masks(f1, f2, np.eye(4), E2, K.astype(np.float64), K.astype(np.float64), 1.0, 5.17)
pr = ops.profile_read()['visibility_prior']; ops.profile_enable(False)
ms = pr[1] / pr[0]
taps = H * W * 64 * 4 * 3
print(json.dumps({'kernel': 'visibility_prior (756x1008, 64 planes)', 'ms': round(ms, 4), 'gathered_GB/s': round(taps / ms / 1e6, 1),
                  'note': 'L2-resident gathers (2.3 MB frame); HBM traffic ~9 MB per call'}))
t0 = time.time()
sys.path.insert(0, ROOT)
from oracle import psv_oracle as po
po.compute_weights(f1[:189, :252], f2[:189, :252], np.eye(4), E2, K.astype(np.float64), K.astype(np.float64), 1.0, 5.17)
print(json.dumps({'cpu_oracle_visibility_prior_ms_for_1/16_frame': round((time.time() - t0) * 1e3, 1)}))
