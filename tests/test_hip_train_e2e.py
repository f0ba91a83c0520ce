"""End to end on the GPU with this tree's own pieces only: images + cameras of a synthetic scene -> on-device ray generation with
the reference's index schedule (RayGeneratorHip, BatchIndexScheduler) -> VipNeRFHip + fused losses -> Adam with the reference's
learning-rate decay -> reference-format checkpoint -> resume -> full-frame eval render to uint8 (predict_frame), driven by
TrainerHip01 (the reference trainer's sequence, src/Trainer01.py:61-107, 265-311).  What is asserted is what training is for:
the loss falls and the rendered training views approach the images."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vip-nerf_amd'), os.path.join(ROOT, 'vip-nerf_amd', 'src')):
    if p not in sys.path:
        sys.path.insert(0, p)


def synthetic_scene(n=3, h=32, w=32, f=40.0, depth=3.0):
    """n cameras on a line, all looking down -z at a fronto-parallel textured plane z = -depth (NeRF camera convention, identity
    rotation): the images are the analytic ray / plane intersections of a smooth texture."""
    K = np.array([[f, 0, w / 2], [0, f, h / 2], [0, 0, 1]], np.float32)
    poses = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    poses[:, 0, 3] = np.linspace(-0.3, 0.3, n)
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing='ij')
    images = []
    for i in range(n):
        dx, dy = (xs - w / 2) / f, -(ys - h / 2) / f
        X, Y = poses[i, 0, 3] + depth * dx, depth * dy
        rgb = np.stack([0.5 + 0.4 * np.sin(2.2 * X + 0.3) * np.cos(1.7 * Y), 0.5 + 0.4 * np.cos(1.5 * X) * np.sin(2.4 * Y + 0.5),
                        0.5 + 0.4 * np.sin(1.1 * X + 1.9 * Y)], -1)
        images.append(np.round(np.clip(rgb, 0, 1) * 255).astype(np.uint8))
    return K, poses, np.stack(images)


def configs(num_iterations, precision='fp16x3'):
    mlp = lambda ns: {'num_samples': ns, 'netdepth': 8, 'netwidth': 256, 'points_positional_encoding_degree': 10,
                      'views_positional_encoding_degree': 4, 'use_view_dirs': True, 'view_dependent_rgb': True, 'predict_visibility': True}
    return {'data_loader': {'ndc': False},
            'model': {'name': 'VipNeRFHip01', 'coarse_mlp': mlp(64), 'fine_mlp': mlp(128), 'chunk': 4096, 'netchunk': 16384,
                      'lindisp': False, 'perturb': True, 'raw_noise_std': 1.0, 'white_bkgd': False, 'hip_precision': precision},
            'losses': [{'name': 'MSEHip01', 'weight': 1}, {'name': 'VisibilityLossHip01', 'weight': 0.1},
                       {'name': 'VisibilityPriorLossHip01', 'iter_weights': {'0': 0, '100': 0.001}}],
            'optimizer': {'lr_initial': 5e-4, 'lr_decay': 250, 'beta1': 0.9, 'beta2': 0.999},
            'num_iterations': num_iterations, 'sub_batch_size': 512, 'model_save_interval': 100, 'device': [0]}


def test_training_a_synthetic_scene_end_to_end(tmp_path):
    from TrainerHip01 import TrainerHip
    from data_preprocessors.RayGeneratorHip01 import BatchIndexScheduler, RayGeneratorHip
    dev = torch.device('cuda:0')
    n, h, w = 3, 32, 32
    K, poses, images_u8 = synthetic_scene(n, h, w)
    images = torch.from_numpy(images_u8.astype(np.float32) / 255)
    prior = torch.ones(n, n - 1, h, w)                        # a fronto-parallel plane is visible from every camera
    torch.manual_seed(0)
    np.random.seed(0)

    def build(iters):
        gen = RayGeneratorHip((h, w), K[None], poses, 2.0, 4.0, False, dev, images=images, visibility_prior=prior)
        sched = BatchIndexScheduler(n, h, w, num_rays=1024)
        return TrainerHip(configs(iters), gen, sched, output_dirpath=tmp_path)

    tr = build(100)
    psnr0 = np.mean([v['psnr'] for v in tr.run_validation().values()])
    hist = tr.train()
    assert len(hist) == 100 and abs(hist[-1]['lr'] - 5e-4 * 0.1 ** (99 / 250000)) < 1e-12
    assert (tmp_path / 'saved_models' / 'Model_Iter000100.tar').exists()
    # resume: a fresh trainer picks the checkpoint up (reference layout, `module.` keys) and continues to iteration 200
    tr2 = build(200)
    w_before = torch.cat([p.detach().flatten() for p in tr.model.parameters()])
    assert tr2.load_model() == 100
    assert torch.equal(torch.cat([p.detach().flatten() for p in tr2.model.parameters()]), w_before)
    hist2 = tr2.train()
    assert len(hist2) == 100
    mse = [x['MSEHip01'] for x in hist + hist2]
    first, last = np.mean(mse[:10]), np.mean(mse[-10:])
    val = tr2.run_validation()
    psnr1 = np.mean([v['psnr'] for v in val.values()])
    print(f'MSE (coarse + fine, 2 sub-batches summed) {first:.4f} -> {last:.4f}; PSNR of the training views {psnr0:.1f} -> {psnr1:.1f} dB')
    assert all(np.isfinite(list(x.values())).all() for x in hist + hist2)
    assert last < 0.25 * first, (first, last)
    assert psnr1 > psnr0 + 6 and psnr1 > 17, (psnr0, psnr1)
    img = val[1]['image']
    assert img.shape == (h, w, 3) and img.dtype == torch.uint8 and float(val[1]['depth'].min()) >= 0
    # the learnt geometry: the plane is 3 units away (depth of the fine level, metric, centre pixel)
    d = float(val[1]['depth'][h // 2, w // 2])
    assert 2.3 < d < 3.7, d
