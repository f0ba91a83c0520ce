"""The N>1 path on CPU: two gloo processes, ray-sharded, one all-reduce of the flat gradient bucket.  The model
here is the CPU oracle (the HIP module cannot run without a GPU); what is under test is vipnerf_hip.dist -- bucket
views, the single collective, shard arithmetic -- and the claim of SURVEY.md §8e that with equal shards the averaged
per-rank gradients equal the single-process gradients of the whole batch."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd'))

N_RAYS = 16
CFG = {'ndc': True, 'n_coarse': 64, 'n_fine': 128, 'noise_std': 1.0}
LOSSES = [{'name': 'MSE01', 'weight': 1}, {'name': 'VisibilityLoss01', 'weight': 0.1},
          {'name': 'VisibilityPriorLoss01', 'iter_weights': {'0': 0, '30000': 0.001}}]


class OracleModule(torch.nn.Module):
    def __init__(self, params):
        super().__init__()
        self.names = list(params.keys())
        self.ps = torch.nn.ParameterList([torch.nn.Parameter(torch.from_numpy(v.copy())) for v in params.values()])

    def pdict(self):
        return dict(zip(self.names, self.ps))


def step_grads(model, batch, rng, sl):
    from oracle import vipnerf_oracle as vo
    n = batch['rays_o'].shape[0]
    sub = {k: (v[sl] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == n else v) for k, v in batch.items()}
    out = vo.render_rays(model.pdict(), sub, CFG, {k: v[sl] for k, v in rng.items()}, train=True, sec_views=True)
    vo.total_loss(sub, out, LOSSES, 40000)['TotalLoss'].backward()


def worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2 if world <= 2 else 1)
    from oracle import vipnerf_oracle as vo
    from vipnerf_hip import dist as vdist
    r, w, _ = vdist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    params = vo.init_params(5 + rank, scale=1.6)            # deliberately different: broadcast must fix it
    model = OracleModule(params)
    vdist.broadcast_parameters(model, src=0)
    bucket = vdist.FlatGradBucket(model.parameters())
    batch = vo.synthetic_batch(N_RAYS, 3, scene='fern', nf=2)
    rng = vo.synthetic_rng(N_RAYS, 64, 128, 4)
    bucket.zero()
    step_grads(model, batch, rng, vdist.shard_rows(N_RAYS, rank, world))
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views)), 'grads left the bucket'
    bucket.all_reduce_mean()
    ret[rank] = bucket.flat.clone().numpy()
    # (b) gradients dropped first (release): the oracle's backward leaves 48 separate tensors, which the bucket
    #     gathers before its single collective
    bucket.release()
    step_grads(model, batch, rng, vdist.shard_rows(N_RAYS, rank, world))
    assert bucket.adopted() is None
    bucket.all_reduce_mean()
    ret[f'gathered{rank}'] = torch.cat([p.grad.reshape(-1) for p in bucket.params]).numpy()
    # (c) gradients handed over as consecutive views of one foreign buffer (what the HIP backward returns): reduced
    #     in place, no copy
    bucket.release()
    step_grads(model, batch, rng, vdist.shard_rows(N_RAYS, rank, world))
    foreign = torch.cat([p.grad.reshape(-1) for p in bucket.params])
    o = 0
    for p in bucket.params:
        p.grad = foreign[o:o + p.numel()].view_as(p)
        o += p.numel()
    adopted = bucket.adopted()
    assert adopted is not None and adopted.data_ptr() == foreign.data_ptr() and adopted.numel() == foreign.numel()
    bucket.all_reduce_mean()
    assert bucket.params[0].grad.data_ptr() == foreign.data_ptr(), 'the adopted buffer must be reduced in place'
    ret[f'adopted{rank}'] = foreign.numpy().copy()
    dist.destroy_process_group()


def test_two_rank_sharded_step_matches_single_process():
    from oracle import vipnerf_oracle as vo
    from vipnerf_hip import dist as vdist
    world, port = 2, 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(worker, args=(world, port, ret), nprocs=world, join=True)
    assert np.array_equal(ret[0], ret[1]), 'ranks disagree after the all-reduce'
    for r in range(world):
        assert np.array_equal(ret[f'gathered{r}'], ret[0]), 'gathered path differs from the in-bucket path'
        assert np.array_equal(ret[f'adopted{r}'], ret[0]), 'adopted path differs from the in-bucket path'
    model = OracleModule(vo.init_params(5, scale=1.6))
    bucket = vdist.FlatGradBucket(model.parameters())
    bucket.zero()
    step_grads(model, vo.synthetic_batch(N_RAYS, 3, scene='fern', nf=2), vo.synthetic_rng(N_RAYS, 64, 128, 4), slice(0, N_RAYS))
    ref = bucket.flat.numpy()
    assert ref.size == 1191946
    err = np.linalg.norm(ret[0] - ref) / np.linalg.norm(ref)
    assert err < 1e-5, err


def worker8(rank, world, port, ret):
    """One of 8 ranks: the shard arithmetic of the two 8-GPU statements (65,536 / 131,072 rows, 2048 + 2048 rows, uneven classes with trim)
    checked ON the rank with its own (rank, world) from the environment, then a real 8-way all-reduce of the oracle's gradients of this
    rank's two rows of a 16-row config-2 style batch (nerf + sparse-depth rows) drawn with the global rows' random numbers."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from oracle import vipnerf_oracle as vo
    from vipnerf_hip import dist as vdist
    r, w, _ = vdist.init_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    # BASELINE configs[3] / configs[4]: contiguous equal shards, rank r owns rows [r N / 8, (r + 1) N / 8)
    for n_rows in (65536, 131072):
        sl = vdist.shard_rows(n_rows, r, w)
        assert (sl.start, sl.stop) == (r * n_rows // w, (r + 1) * n_rows // w)
        ids = vdist.shard_row_ids({'rays_o': torch.zeros(n_rows, 3), 'row_class_counts': (n_rows, 0)}, r, w)
        assert ids.numel() == n_rows // w and int(ids[0]) == sl.start and int(ids[-1]) == sl.stop - 1
    # configs[2]'s batch: 2048 nerf rows then 2048 sparse-depth rows -> 256 + 256 per rank
    ids = vdist.shard_row_ids({'rays_o': torch.zeros(4096, 3), 'row_class_counts': (2048, 2048)}, r, w)
    assert ids.tolist() == list(range(256 * r, 256 * (r + 1))) + list(range(2048 + 256 * r, 2048 + 256 * (r + 1)))
    # uneven classes (the short last batch of an epoch, an odd number of sparse-depth pixels): trimmed per class, never an empty collective
    ids = vdist.shard_row_ids({'rays_o': torch.zeros(1000 + 77, 3), 'row_class_counts': (1000, 77)}, r, w, uneven='trim')
    assert ids.tolist() == list(range(125 * r, 125 * (r + 1))) + list(range(1000 + 9 * r, 1000 + 9 * (r + 1)))
    everyone = [None] * w
    dist.all_gather_object(everyone, ids.tolist())
    flat = sorted(i for part in everyone for i in part)
    assert flat == list(range(1000)) + list(range(1000, 1072)), 'trimmed shards must cover each class\'s first count - count % world rows once'
    # the gradient of the global batch = the mean of the 8 rank gradients (one all-reduce of the flat bucket)
    model = OracleModule(vo.init_params(5 + rank, scale=1.6))
    vdist.broadcast_parameters(model, src=0)
    bucket = vdist.FlatGradBucket(model.parameters())
    batch = vo.synthetic_batch(8, 3, scene='realestate', nf=3, n_sparse=8)
    rng = vo.synthetic_rng(16, 64, 128, 4)
    shard = vdist.shard_batch(batch, r, w)
    rows = shard['rng_ray_ids']
    assert rows.tolist() == [r, 8 + r]
    bucket.zero()
    out = vo.render_rays(model.pdict(), shard, CFG, {k: v[rows] for k, v in rng.items()}, train=True, sec_views=True)
    vo.total_loss(shard, out, LOSSES8, 40000)['TotalLoss'].backward()
    bucket.all_reduce_mean()
    ret[rank] = bucket.flat.clone().numpy()
    dist.destroy_process_group()


LOSSES8 = LOSSES + [{'name': 'SparseDepthMSE01', 'weight': 0.1}]


def test_eight_rank_shards_and_reduction_match_single_process():
    """SURVEY 8e at the world size the 8-GPU statements use: 8 gloo ranks, each with its row-class-aware shard (one nerf row + one
    sparse-depth row) and the global rows' random numbers; the reduced gradient is the same on every rank and equals the single-process
    gradient of the 16-row batch."""
    from oracle import vipnerf_oracle as vo
    from vipnerf_hip import dist as vdist
    world, port = 8, 27500 + (os.getpid() % 2000)
    ret = mp.Manager().dict()
    mp.spawn(worker8, args=(world, port, ret), nprocs=world, join=True)
    for r in range(1, world):
        assert np.array_equal(ret[0], ret[r]), f'rank {r} disagrees with rank 0 after the all-reduce'
    model = OracleModule(vo.init_params(5, scale=1.6))
    bucket = vdist.FlatGradBucket(model.parameters())
    bucket.zero()
    batch = vo.synthetic_batch(8, 3, scene='realestate', nf=3, n_sparse=8)
    out = vo.render_rays(model.pdict(), batch, CFG, vo.synthetic_rng(16, 64, 128, 4), train=True, sec_views=True)
    vo.total_loss(batch, out, LOSSES8, 40000)['TotalLoss'].backward()
    ref = bucket.flat.numpy()
    err = np.linalg.norm(ret[0] - ref) / np.linalg.norm(ref)
    assert err < 2e-5, err


def test_shard_rows_partition():
    from vipnerf_hip import dist as vdist
    for world in (1, 2, 4, 8):
        rows = np.concatenate([np.arange(4096)[vdist.shard_rows(4096, r, world)] for r in range(world)])
        assert np.array_equal(rows, np.arange(4096))


def test_row_class_aware_sharding():
    """nerf rows and sparse-depth rows are split separately, so that each rank holds N_nerf/R + N_sd/R rows
    and the mean of the rank means of every loss equals its global mean (config 3: 2048 + 2048 rows)."""
    from oracle import vipnerf_oracle as vo
    from vipnerf_hip import dist as vdist
    b = vo.synthetic_batch(2048, 3, scene='realestate', nf=3, n_sparse=2048)
    n = 4096
    for world in (2, 4, 8):
        seen = []
        for r in range(world):
            ids = vdist.shard_row_ids(b, r, world)
            assert int(b['indices_mask_nerf'][ids].sum()) == 2048 // world
            assert int(b['indices_mask_sparse_depth'][ids].sum()) == 2048 // world
            sb = vdist.shard_batch(b, r, world)
            assert sb['rays_o'].shape[0] == n // world and torch.equal(sb['rng_ray_ids'], ids)
            assert torch.equal(sb['sparse_depth_values'], b['sparse_depth_values'][ids]) and sb['num_frames'] == b['num_frames']
            seen.append(ids)
        assert torch.equal(torch.sort(torch.cat(seen))[0], torch.arange(n))
    # the losses' means: rank means average to the global mean exactly when the classes are split separately ...
    out = {'rgb_coarse': torch.rand(n, 3), 'rgb_fine': torch.rand(n, 3), 'depth_fine': torch.rand(n) * 5}
    g_mse, g_sd = vo.loss_mse(b, out, ('coarse', 'fine')), vo.loss_sparse_depth(b, out, ('coarse', 'fine'))
    parts = [vdist.shard_batch({**b, **out}, r, 4) for r in range(4)]
    m_mse = sum(vo.loss_mse(p, p, ('coarse', 'fine')) for p in parts) / 4
    m_sd = sum(vo.loss_sparse_depth(p, p, ('coarse', 'fine')) for p in parts) / 4
    assert abs(float(m_mse - g_mse)) < 1e-6 and abs(float(m_sd - g_sd)) < 1e-5
    # ... which a contiguous split of the mixed batch does not give (ranks 0-1 would hold no sparse-depth row at all)
    with pytest.raises(ValueError):
        vdist.shard_rows(4095, 0, 2)


def test_short_batches_trim_per_row_class_on_the_host():
    """the short last batch of an epoch (or an odd number of sparse-depth pixels) used to raise mid-training with more
    than one rank; the trainer now trims every row class to a multiple of the ranks, from host-side class counts (no device sync)."""
    import torch
    from vipnerf_hip import dist as vdist
    batch = {'rays_o': torch.zeros(12, 3), 'row_class_counts': (7, 5)}
    with pytest.raises(ValueError):
        vdist.shard_row_ids(batch, 0, 2)
    ids = [vdist.shard_row_ids(batch, r, 2, uneven='trim').tolist() for r in range(2)]
    assert ids == [[0, 1, 2, 7, 8], [3, 4, 5, 9, 10]]
    sh = vdist.shard_batch(dict(batch, target=torch.arange(12.)), 1, 2, uneven='trim')
    assert sh['target'].tolist() == [3., 4., 5., 9., 10.] and sh['row_class_counts'] == (3, 2) and sh['rng_ray_ids'].tolist() == ids[1]


# ------------------------------------------------------------------------------------------------ the scaling line's self-check
def worker_verify(rank, world, port, ret):
    """What bench.py --gpus N runs before / after its timed region, under gloo with the oracle as the model: vdist.verify_sharded_gradient
    (shard gradients all-reduced vs the whole global batch's gradient computed on every rank) and vdist.params_identical."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2 if world <= 2 else 1)
    from oracle import vipnerf_oracle as vo
    from vipnerf_hip import dist as vdist
    r, w, _ = vdist.init_from_env(backend='gloo')
    model = OracleModule(vo.init_params(5 + rank, scale=1.6))
    vdist.broadcast_parameters(model, src=0)
    n_nerf, n_sd = world, world                                   # one nerf row + one sparse-depth row per rank
    batch = vo.synthetic_batch(n_nerf, 3, scene='realestate', nf=3, n_sparse=n_sd)
    rng = vo.synthetic_rng(n_nerf + n_sd, 64, 128, 4)             # a row's draws: by its GLOBAL index, whatever batch it arrives in

    def grad_fn(b):
        rows = b['rng_ray_ids'] if 'rng_ray_ids' in b else torch.arange(b['rays_o'].shape[0])
        for p in model.parameters():
            p.grad = None
        out = vo.render_rays(model.pdict(), b, CFG, {k: v[rows] for k, v in rng.items()}, train=True, sec_views=True)
        vo.total_loss(b, out, LOSSES8, 40000)['TotalLoss'].backward()
        return torch.cat([p.grad.reshape(-1) for p in model.parameters()])

    res = vdist.verify_sharded_gradient(grad_fn, batch, r, w)
    same0 = vdist.params_identical(model.parameters())
    # a broken reduction must be caught: this rank's shard gradient scaled by (1 + rank) before the collective
    bad = vdist.verify_sharded_gradient(lambda b: grad_fn(b) * (1.0 + (r if 'rng_ray_ids' in b else 0)), batch, r, w)
    # ranks that drifted apart must be caught: one element of one rank's parameters moved by one ulp
    if r == w - 1:
        with torch.no_grad():
            p = list(model.parameters())[3]
            p.view(-1)[7] = torch.nextafter(p.view(-1)[7], torch.tensor(10.0))
    same1 = vdist.params_identical(model.parameters())
    ret[rank] = (res, same0, bad, same1)
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 8])
def test_sharding_self_check_under_gloo(world):
    port = 25500 + (os.getpid() % 2000) + world
    ret = mp.Manager().dict()
    mp.spawn(worker_verify, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        res, same0, bad, same1 = ret[r]
        assert res['ranks'] == world and res['rel_l2'] == ret[0][0]['rel_l2'], 'every rank must hold the same verdict'
        assert res['rel_l2'] <= 2e-5 and res['rel_l2_rank'] <= res['rel_l2'], res
        assert same0 == {'identical': True, 'max_bits_diff': 0, 'max_sum_diff': 0.0}
        assert bad['rel_l2'] > 0.1, bad                                     # the scaled shards are not the whole batch's gradient
        assert same1['identical'] is False and same1['max_bits_diff'] >= 1, same1
