"""SURVEY.md §8f rows f-1 / f-2 on the GPU: ray generation + batch gather and frame post-processing through the C
ABI, against the golden vectors of the reference's DataPreprocessor (bit-exact) and the CPU oracle."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')
sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd', 'src'))

from oracle import raygen_oracle as ro  # noqa: E402
from oracle import vipnerf_oracle as vo  # noqa: E402


def load(name):
    return {k: v for k, v in np.load(os.path.join(GOLD, name + '.npz')).items()}


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def make_gen(g, dev, images=None, prior=None, ndc=True):
    from data_preprocessors.RayGeneratorHip01 import RayGeneratorHip
    res = tuple(int(v) for v in g['resolution'])
    return RayGeneratorHip(res, g['intrinsic'][None], g['poses'], float(g['near']), 6.0, ndc, dev, images=images, visibility_prior=prior), res


def test_ray_generation_bit_exact(dev):
    g = load('f6_raygen')
    gen, res = make_gen(g, dev)
    for i in range(3):
        b = gen.create_test_data(i, secondary=True)
        for hk, gk in (('rays_o', 'rays_o'), ('rays_d', 'rays_d'), ('view_dirs', 'view_dirs'), ('rays_o_ndc', 'rays_o_ndc'), ('rays_d_ndc', 'rays_d_ndc')):
            ref = g[f'{gk}_{i}'].reshape(-1, 3)
            assert np.array_equal(b[hk].cpu().numpy(), ref), f'{hk} frame {i}'
        img_id = np.full(res[0] * res[1], i)
        assert np.array_equal(b['rays_o2'].cpu().numpy(), ro.secondary_origins(g['poses'], img_id))
        pid = b['pixel_id'].cpu().numpy()
        assert (pid[:, 0] == i).all() and pid[:, 1].max() == res[1] - 1 and pid[:, 2].max() == res[0] - 1
        assert float(b['near'].min()) == 1.0 and float(b['far_ndc'].max()) == 1.0


@pytest.mark.parametrize('seed', range(6))
def test_ray_generation_camera_sweep_bit_exact(dev, seed):
    """Seeded cameras away from the golden's 24 x 32 frame: odd sizes, fx != fy, off-centre principal point, arbitrary
    rotations and translations, several near planes, NDC on and off -- rays, view directions, NDC rays and secondary camera
    centres bit-identical to the numpy restatement of DataPreprocessor01.py:335-378 (itself pinned by f6_raygen)."""
    from data_preprocessors.RayGeneratorHip01 import RayGeneratorHip
    rs = np.random.default_rng(500 + seed)
    h, w = int(rs.integers(5, 90)), int(rs.integers(5, 130))
    K = np.array([[rs.uniform(20, 900), 0, rs.uniform(0.3, 0.7) * w], [0, rs.uniform(20, 900), rs.uniform(0.3, 0.7) * h], [0, 0, 1]], np.float32)
    nfr = int(rs.integers(2, 5))
    poses = np.tile(np.eye(4, dtype=np.float32), (nfr, 1, 1))
    for f in range(nfr):
        q, _ = np.linalg.qr(rs.normal(size=(3, 3)))
        if np.linalg.det(q) < 0:
            q[:, 0] *= -1
        ang = rs.uniform(0, 0.3)                                   # a moderate rotation: q^ang would need a log; blend and re-orthonormalise
        r, _ = np.linalg.qr((1 - ang) * np.eye(3) + ang * q)
        r *= np.sign(np.diag(r))[None, :]                          # columns oriented like the identity's
        poses[f, :3, :3] = r.astype(np.float32)
        poses[f, :3, 3] = rs.normal(size=3).astype(np.float32) * 0.3
    near = float(rs.choice([0.5, 1.0, 2.0]))
    ndc = bool(seed % 2 == 0)
    gen = RayGeneratorHip((h, w), K[None], poses, near, 6.0, ndc, dev)
    for f in range(nfr):
        b = gen.create_test_data(f, secondary=True)
        ro_, rd_ = ro.get_rays((h, w), K, poses[f])
        assert np.array_equal(b['rays_o'].cpu().numpy(), ro_.reshape(-1, 3)), f'rays_o frame {f}'
        assert np.array_equal(b['rays_d'].cpu().numpy(), rd_.reshape(-1, 3)), f'rays_d frame {f}'
        assert np.array_equal(b['view_dirs'].cpu().numpy(), ro.get_view_dirs(rd_).reshape(-1, 3)), f'view_dirs frame {f}'
        if ndc:
            on, dn = ro.get_ndc_rays(ro_, rd_, (h, w), K, near)
            assert np.array_equal(b['rays_o_ndc'].cpu().numpy(), on.reshape(-1, 3)), f'rays_o_ndc frame {f}'
            assert np.array_equal(b['rays_d_ndc'].cpu().numpy(), dn.reshape(-1, 3)), f'rays_d_ndc frame {f}'
        assert np.array_equal(b['rays_o2'].cpu().numpy(), ro.secondary_origins(poses, np.full(h * w, f)))


def test_batch_gather_matches_cached_gather(dev):
    """shuffled indices: the on-device recomputation equals gathering rows of the full per-scene ray cache"""
    g = load('f6_raygen')
    res = tuple(int(v) for v in g['resolution'])
    rs = np.random.default_rng(8)
    n, hw = 3, res[0] * res[1]
    images = rs.random((n, res[0], res[1], 3), dtype=np.float32)
    prior = (rs.random((n, n - 1, res[0], res[1])) < 0.5).astype(np.float32)
    gen, _ = make_gen(g, dev, torch.from_numpy(images), torch.from_numpy(prior))
    idx = rs.permutation(n * hw)[:1000]
    b = gen.get_next_batch(7, torch.from_numpy(idx))
    cache = {k: np.concatenate([g[f'{k}_{i}'].reshape(-1, 3) for i in range(n)]) for k in ('rays_o', 'rays_d', 'view_dirs', 'rays_o_ndc', 'rays_d_ndc')}
    for k in cache:
        assert np.array_equal(b[k].cpu().numpy(), cache[k][idx]), k
    assert np.array_equal(b['target_rgb'].cpu().numpy(), images.reshape(-1, 3)[idx])
    masks = np.transpose(prior, [0, 2, 3, 1]).reshape(-1, n - 1)            # DataPreprocessor01.py:475-476
    assert np.array_equal(b['visibility_prior_masks'].cpu().numpy(), masks[idx])
    assert b['iter_num'] == 7 and b['num_frames'] == 3 and b['common_data']['poses'].shape == (1, 3, 4, 4)
    e = gen.get_next_batch(0, torch.zeros(0, dtype=torch.int64))            # empty batch
    assert e['rays_o'].shape == (0, 3)


def test_postprocess_bit_exact(dev):
    g = load('f6_raygen')
    gen, res = make_gen(g, dev, ndc=False)
    out = {'rgb_fine': torch.from_numpy(g['pp_rgb']).to(dev), 'depth_fine': torch.from_numpy(g['pp_depth']).to(dev),
           'depth_var_fine': torch.from_numpy(g['pp_depth']).to(dev)}
    r = gen.retrieve_inference_outputs(out)
    assert r['image'].dtype == torch.uint8 and np.array_equal(r['image'].cpu().numpy(), g['pp_image'])
    assert np.array_equal(r['depth'].cpu().numpy(), g['pp_depth_out'])


def test_predict_frame_against_oracle(dev):
    """camera -> image entirely on the GPU vs the oracle render of the same rays (uint8 image within one level)."""
    from data_preprocessors.RayGeneratorHip01 import predict_frame
    from models.ModelFactory import get_model
    g = load('f6_raygen')
    gen, res = make_gen(g, dev)
    mlp = lambda ns: {'num_samples': ns, 'netdepth': 8, 'netwidth': 256, 'points_positional_encoding_degree': 10,
                      'views_positional_encoding_degree': 4, 'use_view_dirs': True, 'view_dependent_rgb': True, 'predict_visibility': True}
    cfg = {'data_loader': {'ndc': True}, 'model': {'name': 'VipNeRFHip01', 'coarse_mlp': mlp(64), 'fine_mlp': mlp(128), 'lindisp': False,
                                                    'perturb': True, 'raw_noise_std': 1.0, 'white_bkgd': False}}
    params = vo.init_params(31, scale=1.6, sigma_bias=0.6)
    model = get_model(cfg, None)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    model = model.to(dev).eval()
    r = predict_frame(model, gen, frame=1)
    o, d = ro.get_rays(res, g['intrinsic'], g['poses'][1])
    on, dn = ro.get_ndc_rays(o, d, res, g['intrinsic'], 1.0)
    hw = res[0] * res[1]
    b = {'rays_o': torch.from_numpy(o.reshape(-1, 3).copy()), 'rays_d': torch.from_numpy(d.reshape(-1, 3).copy()),
         'view_dirs': torch.from_numpy(ro.get_view_dirs(d).reshape(-1, 3)), 'rays_o_ndc': torch.from_numpy(on.reshape(-1, 3)),
         'rays_d_ndc': torch.from_numpy(dn.reshape(-1, 3)), 'near_ndc': torch.zeros(hw, 1), 'far_ndc': torch.ones(hw, 1)}
    with torch.no_grad():
        ref = vo.render_rays(vo.params_to_torch(params), b, {'ndc': True, 'n_coarse': 64, 'n_fine': 128}, None, train=False, sec_views=False)
    img = ro.post_process_image(ref['rgb_fine'].numpy().reshape(res[0], res[1], 3)).astype(np.int32)
    diff = np.abs(r['image'].cpu().numpy().astype(np.int32) - img)
    assert diff.max() <= 1 and (diff > 0).mean() < 0.02, (diff.max(), (diff > 0).mean())
    dep = ro.post_process_depth(ref['depth_fine'].numpy().reshape(res))
    assert np.abs(r['depth'].cpu().numpy() - dep).max() <= 2e-3 * dep.max()


def test_frame_strips_equal_the_whole_frame(dev):
    """A frame rendered as strips of rows (what N GPUs do, one strip each: predict_frame_sharded) is bit-identical to the frame
    rendered in one call -- rays are independent and the kernels' results do not depend on the launch size."""
    from data_preprocessors.RayGeneratorHip01 import frame_strip, predict_frame
    from models.ModelFactory import get_model
    g = load('f6_raygen')
    gen, res = make_gen(g, dev)
    mlp = lambda ns: {'num_samples': ns, 'netdepth': 8, 'netwidth': 256, 'points_positional_encoding_degree': 10,
                      'views_positional_encoding_degree': 4, 'use_view_dirs': True, 'view_dependent_rgb': True, 'predict_visibility': True}
    cfg = {'data_loader': {'ndc': True}, 'model': {'name': 'VipNeRFHip01', 'coarse_mlp': mlp(64), 'fine_mlp': mlp(128), 'lindisp': False,
                                                    'perturb': True, 'raw_noise_std': 1.0, 'white_bkgd': False}}
    params = vo.init_params(33, scale=1.6, sigma_bias=0.6)
    model = get_model(cfg, None)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    model = model.to(dev).eval()
    whole = predict_frame(model, gen, frame=2, secondary=True)
    for world in (2, 3, 5, 8, res[0] + 8):                  # the last one: more ranks than rows, some strips are empty
        strips = [frame_strip(res[0], r, world) for r in range(world)]
        assert strips[0][0] == 0 and strips[-1][1] == res[0] and all(a[1] == b[0] for a, b in zip(strips, strips[1:]))
        parts = [predict_frame(model, gen, frame=2, secondary=True, rows=rw) for rw in strips]
        for k, v in whole.items():
            cat = torch.cat([p[k] for p in parts], dim=1 if k == 'visibility2' else 0)
            assert torch.equal(cat, v), f'{k} with {world} strips'
    with pytest.raises(Exception):
        gen.create_test_data(0, rows=(5, res[0] + 1))


def test_visibility_prior_generator_golden(dev):
    """f-3: plane-sweep visibility weights on the GPU vs the reference's VisibilityWeightsComputer (float64 geometry:
    1e-9 relative on the weights, identical masks)."""
    from prior_generators.VisibilityMaskHip02 import VisibilityWeightsComputerHip
    g = load('f7_visibility_prior')
    comp = VisibilityWeightsComputerHip({'num_depth_planes': int(g['n_planes']), 'temperature': float(g['temperature'])}, dev)
    for a, b, ea, eb, key in (('frame1', 'frame2', 'E1', 'E2', 'weights12'), ('frame2', 'frame1', 'E2', 'E1', 'weights21')):
        w, m = comp.compute_masks(g[a], g[b], g[ea], g[eb], g['K'], g['K'], float(g['min_depth']), float(g['max_depth']))
        np.testing.assert_allclose(w, g[key], rtol=1e-9, atol=1e-12)
        assert np.array_equal(m, g[key] > 0.5)
        assert w.dtype == np.float64 and w.shape == g[key].shape


def test_visibility_prior_scene_generation_and_disk_layout(dev, tmp_path):
    """A whole scene (every ordered pair, reference :248-279): the (n, n-1, h, w) array in the loader's order, each entry equal
    to the pairwise call; the files written in the reference's layout read back -- `{f1:04}_{f2:04}.png` == 255, as
    NerfLlffDataLoader01.read_mask does -- to the same masks, and the .npy files hold the bool masks / float64 weights."""
    from prior_generators.VisibilityMaskHip02 import VisibilityWeightsComputerHip, load_scene_masks
    g = load('f7_visibility_prior')
    cfgp = {'num_depth_planes': int(g['n_planes']), 'temperature': float(g['temperature'])}
    comp = VisibilityWeightsComputerHip(cfgp, dev)
    rs = np.random.default_rng(4)
    f3 = np.clip(g['frame1'].astype(np.int32) + rs.integers(-20, 20, size=g['frame1'].shape), 0, 255).astype(np.uint8)
    E3 = 0.5 * (g['E1'] + g['E2'])
    E3[:3, :3] = g['E1'][:3, :3]
    frames, E, K = [g['frame1'], g['frame2'], f3], [g['E1'], g['E2'], E3], [g['K']] * 3
    nums = [3, 11, 20]
    lo, hi = float(g['min_depth']), float(g['max_depth'])
    masks, weights = comp.generate_scene(frames, E, K, lo, hi, frame_nums=nums, output_dirpath=tmp_path)
    assert masks.shape == (3, 2) + g['frame1'].shape[:2] and masks.dtype == bool and weights.dtype == np.float64
    np.testing.assert_allclose(weights[0, 0], g['weights12'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(weights[1, 0], g['weights21'], rtol=1e-9, atol=1e-12)
    for i in range(3):
        for c, j in enumerate([x for x in range(3) if x != i]):
            w, m = comp.compute_masks(frames[i], frames[j], E[i], E[j], K[i], K[j], lo, hi)
            assert np.array_equal(weights[i, c], w) and np.array_equal(masks[i, c], m), (i, j)
            stem = f'{nums[i]:04}_{nums[j]:04}'
            assert np.array_equal(np.load(tmp_path / f'visibility_masks/{stem}.npy'), m)
            assert np.array_equal(np.load(tmp_path / f'visibility_weights/{stem}.npy'), w)
            assert (tmp_path / f'visibility_weights/{stem}.png').exists()
    assert np.array_equal(load_scene_masks(tmp_path, nums), masks)
    dm, dw = comp.generate_scene(frames, E, K, lo, hi, keep_on_device=True)
    assert dm.is_cuda and np.array_equal(dm.cpu().numpy(), masks) and np.array_equal(dw.cpu().numpy(), weights)


def test_visibility_prior_full_frame_properties(dev):
    """LLFF-sized frame (756x1008, 64 planes): identical frames and cameras are fully visible (w = 1); a camera
    displaced far to the side sees nothing (all taps outside -> warped = 0 -> error = mean intensity)."""
    from prior_generators.VisibilityMaskHip02 import VisibilityWeightsComputerHip
    rs = np.random.default_rng(2)
    h, w = 756, 1008
    img = rs.integers(1, 256, size=(h, w, 3)).astype(np.uint8)
    K = np.array([[815.13, 0, 504.], [0, 815.13, 378.], [0, 0, 1.]])
    comp = VisibilityWeightsComputerHip({'num_depth_planes': 64, 'temperature': 10}, dev)
    wts = comp.compute_weights(img, img, np.eye(4), np.eye(4), K, K, 1.0, 5.0)
    assert wts.shape == (h, w) and np.abs(wts - 1.0).max() < 1e-9
    E2 = np.eye(4)
    E2[0, 3] = 1e5                                                       # projects every plane far outside frame 2
    wts = comp.compute_weights(img, img, np.eye(4), E2, K, K, 1.0, 5.0)
    ref = np.exp(-img.astype(np.float64).mean(-1) / 10)
    inside = (slice(2, h - 2), slice(2, w - 2))
    assert np.abs(wts[inside] - ref[inside]).max() < 1e-9


def test_training_batches_equal_the_reference_loader(dev):
    """f-1 end to end against golden F6b -- the batch dicts the reference's DataPreprocessor (train mode, cached batching,
    sparse depth + visibility-prior masks, pre-crop) hands its trainer over 17 iterations: host index schedule
    (BatchIndexScheduler) -> ONE vipnerf_generate_rays launch per iteration.  Every array bit for bit: the rays of nerf AND
    sparse-depth rows, -1 fill of target_rgb / prior on sparse-depth rows and of the sparse_* columns on nerf rows, the
    NDC depths of the sparse points (computed once per scene as preprocess_sparse_depth_data does), ragged end-of-epoch
    batches."""
    from data_preprocessors.RayGeneratorHip01 import BatchIndexScheduler, RayGeneratorHip
    g = load('f6b_batches')
    n, h, w = int(g['n']), int(g['h']), int(g['w'])
    gen = RayGeneratorHip((h, w), g['intrinsics'], g['poses'], float(g['near']), float(g['far']), True, dev,
                          near_ndc=float(g['near_ndc']), far_ndc=float(g['far_ndc']), images=torch.from_numpy(g['images']),
                          visibility_prior=torch.from_numpy(g['masks']), sparse_depths=g['sparse_depths'],
                          sparse_errors=g['sparse_errors'])
    assert np.array_equal(gen.sparse_depths_ndc.cpu().numpy().reshape(n, h, w), g['sparse_depths_ndc']), 'depths_ndc table'
    np.random.seed(int(g['numpy_seed']))
    sched = BatchIndexScheduler(n, h, w, int(g['num_rays']), float(g['precrop_fraction']), int(g['precrop_iterations']),
                                g['sparse_depths'], int(g['num_rays_sparse']))
    keys = ['rays_o', 'rays_d', 'view_dirs', 'pixel_id', 'target_rgb', 'near', 'far', 'rays_o_ndc', 'rays_d_ndc', 'near_ndc',
            'far_ndc', 'sparse_depth_values', 'sparse_depth_errors', 'sparse_depth_values_ndc', 'visibility_prior_masks',
            'indices', 'indices_mask_nerf', 'indices_mask_sparse_depth']
    for it in range(int(g['iters'])):
        b = gen.get_next_batch(it, scheduler=sched)
        for k in keys:
            ref = g[f'it{it}_{k}']
            got = b[k].cpu().numpy()
            assert got.shape == ref.shape and got.dtype == ref.dtype, (it, k, got.shape, ref.shape, got.dtype, ref.dtype)
            assert np.array_equal(got, ref), (it, k)
        assert b['iter_num'] == it and b['num_frames'] == n
        assert np.array_equal(b['common_data']['poses'][0].cpu().numpy(), g[f'it{it}_poses'][0])
    # the model consumes such a batch as it is (sparse-depth rows included)
    from models.ModelFactory import get_model
    import test_hip_parity as tp
    model, cfg = tp.make_model(dev, True, vo.init_params(3, scale=1.6), sparse=True)
    model.train()
    out = model(b)
    from loss_functions.LossComputerHip01 import LossComputerHip
    lv = LossComputerHip(cfg).compute_losses(b, out)
    assert torch.isfinite(lv['TotalLoss']) and float(lv['SparseDepthMSEHip01']['loss_value']) > 0
