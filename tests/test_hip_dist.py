"""The N>1 path with the HIP module itself (tests/test_dist_gloo.py drives the CPU oracle): two ranks share ONE GPU (gloo
carries the collective; on a multi-GPU node the same code runs one rank per GPU over RCCL), each renders its row-class-aware
shard of a config-3 style batch (nerf rows + sparse-depth rows) with the ON-DEVICE random numbers, the 48 gradients arrive
as consecutive views of one buffer that FlatGradBucket adopts and reduces with a single all-reduce -- and the averaged
gradients equal the single-process gradients of the whole batch."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vip-nerf_amd'), os.path.join(ROOT, 'vip-nerf_amd', 'src'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

N_NERF, N_SD, SEED, ITER = 384, 128, 4242, 40000


def _global_batch(dev):
    from oracle import vipnerf_oracle as vo
    import test_hip_parity as tp
    b = vo.synthetic_batch(N_NERF, 11, scene='realestate', nf=3, n_sparse=N_SD)
    return b, tp.ref_batch(b, dev, ITER)


def _step(model, cfg, batch):
    from loss_functions.LossComputerHip01 import LossComputerHip
    torch.manual_seed(SEED)                    # every rank: the same seed, like the reference's init_seeds
    model._last_iter = None
    out = model(batch)
    LossComputerHip(cfg).compute_losses(batch, out)['TotalLoss'].backward()
    return out


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                      VIPNERF_DIST_BACKEND='gloo')
    from oracle import vipnerf_oracle as vo
    from vipnerf_hip import dist as vdist
    import test_hip_parity as tp
    r, w, _ = vdist.init_from_env(backend='gloo')
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    b, rb = _global_batch(dev)
    model, cfg = tp.make_model(dev, True, vo.init_params(20 + rank, scale=1.6), sparse=True)    # different on purpose
    vdist.broadcast_parameters(model, src=0)
    model.train()
    bucket = vdist.FlatGradBucket(model.parameters())
    shard = vdist.shard_batch(rb, r, w)
    assert int(shard['indices_mask_nerf'].sum()) == N_NERF // w and int(shard['indices_mask_sparse_depth'].sum()) == N_SD // w
    bucket.release()
    out = _step(model, cfg, shard)
    flat = bucket.adopted()
    assert flat is not None and flat.numel() == 1191946, 'the HIP backward must hand over one flat gradient buffer'
    ptr = flat.data_ptr()
    bucket.all_reduce_mean()
    assert bucket.params[0].grad.data_ptr() == ptr, 'reduced in place'
    ret[rank] = flat.cpu().numpy()
    ret[f'ids{rank}'] = shard['rng_ray_ids'].cpu().numpy()
    ret[f'z{rank}'] = out['z_vals_fine'].detach().cpu().numpy()
    torch.distributed.destroy_process_group()


def test_two_hip_ranks_on_one_gpu_equal_one_process():
    assert torch.cuda.is_available()
    world, port = 2, 31000 + (os.getpid() % 2000)
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert np.array_equal(ret[0], ret[1]), 'ranks disagree after the all-reduce'
    from oracle import vipnerf_oracle as vo
    import test_hip_parity as tp
    dev = torch.device('cuda:0')
    _, rb = _global_batch(dev)
    model, cfg = tp.make_model(dev, True, vo.init_params(20, scale=1.6), sparse=True)
    model.train()
    out = _step(model, cfg, rb)
    ref = torch.cat([p.grad.flatten() for p in model.parameters()]).cpu().numpy()
    zf = out['z_vals_fine'].detach().cpu().numpy()
    for r in range(world):      # the shards drew, ray for ray, the random numbers of the whole batch (keyed by global row index)
        assert np.array_equal(ret[f'z{r}'], zf[ret[f'ids{r}']]), f'rank {r}: fine depths differ from the single-process run'
    err = np.linalg.norm(ret[0] - ref) / np.linalg.norm(ref)
    assert err < 1e-5, f'averaged rank gradients vs single-process gradients: {err:.2e}'


def test_eight_hip_ranks_on_one_gpu_equal_one_process():
    """The same at the world size of BASELINE configs[3] / configs[4]: EIGHT ranks of the HIP module share the GPU (gloo carries the
    all-reduce), 48 + 16 rows each; every rank's fine depths are the global rows' (Philox keyed by global row index), the reduced gradient
    is identical on all ranks and equals the single-process one."""
    assert torch.cuda.is_available()
    world, port = 8, 30000 + (os.getpid() % 900)
    ret = mp.Manager().dict()
    try:
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    except mp.ProcessExitedException as e:
        # eight processes SHARING one device: about one start in twenty-five a rank's queue is aborted by the runtime (see
        # test_bench_self_spawns_eight_ranks_strong_scaling and profiles/r05_spawn8_abort.log); such a start is repeated once, on another port
        import warnings
        warnings.warn(f'a rank died while eight processes opened the device ({e}); repeating the launch once')
        ret = mp.Manager().dict()
        mp.spawn(_worker, args=(world, port + 901, ret), nprocs=world, join=True)
    from oracle import vipnerf_oracle as vo
    import test_hip_parity as tp
    dev = torch.device('cuda:0')
    _, rb = _global_batch(dev)
    model, cfg = tp.make_model(dev, True, vo.init_params(20, scale=1.6), sparse=True)
    model.train()
    out = _step(model, cfg, rb)
    ref = torch.cat([p.grad.flatten() for p in model.parameters()]).cpu().numpy()
    zf = out['z_vals_fine'].detach().cpu().numpy()
    seen = []
    for r in range(world):
        assert np.array_equal(ret[0], ret[r]), f'rank {r} disagrees after the all-reduce'
        assert ret[f'ids{r}'].shape[0] == (N_NERF + N_SD) // world
        assert np.array_equal(ret[f'z{r}'], zf[ret[f'ids{r}']]), f'rank {r}: fine depths differ from the single-process run'
        seen += ret[f'ids{r}'].tolist()
    assert sorted(seen) == list(range(N_NERF + N_SD))
    err = np.linalg.norm(ret[0] - ref) / np.linalg.norm(ref)
    assert err < 1e-5, f'averaged rank gradients vs single-process gradients: {err:.2e}'


def test_bench_self_spawns_eight_ranks_strong_scaling():
    """`python bench.py --gpus 8` WITHOUT a launcher: bench.py becomes the launcher (torch.distributed.run, 8 ranks -- here all on this GPU
    over gloo), runs the ray-sharded strong-scaling statement (8192 rays per iteration / 8) with the render leg as 8 strips of the 756-row
    frame (756 / 8 is not an integer), and reports n_gpus = the ranks that took part in its collectives."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env['VIPNERF_DIST_BACKEND'] = 'gloo'
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '2', '--warmup', '1', '--scaling', 'strong', '--global-rays', '8192',
           '--no-configs4', '--no-configs2', '--also', 'bf16']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    if r.returncode != 0 and 'Signal 6 (SIGABRT)' in r.stderr:
        # eight HIP processes SHARING one device: about one start in twenty-five a rank dies by SIGABRT.  Round 5 ran the start 24 times
        # with the runtime's log on (tools/spawn8_bench_probe.sh, profiles/r05_spawn8_abort.log): the rank's hardware queue is aborted with
        # HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION while the dispatch in flight is PyTorch's own FillFunctor kernel (a torch.zeros of the
        # 1.19 M-float gradient bucket) -- not one of this library's kernels -- and 320 single-GPU process starts that load the library and run
        # its forward kernel never showed it (tools/spawn8_probe.py): the runtime under eight processes on one device, not the code under
        # test.  The statement under test is bench.py's launcher and rank accounting, so such a start is repeated once -- a second abort
        # fails the test.  (On a real 8-GPU node every rank owns its device.)
        import warnings
        warnings.warn('a rank aborted while eight processes opened the device; repeating the launch once')
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    if r.returncode != 0:            # what the failing rank itself said sits far above the launcher's summary
        import re
        told = [ln for ln in r.stderr.splitlines() if re.search(r'abort|terminate|what\(\)|HSA_|hipError|HIP error|Assertion|Segmentation|memory|Traceback|Error:', ln)]
        raise AssertionError('\n'.join(told[:40]) + '\n...\n' + r.stderr[-1500:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res['n_gpus'] == 8 and res['ranks_reduced'] == 8 and res['scaling'] == 'strong'
    assert res['config']['rays_per_gpu'] == 1024 and res['config']['global_rays'] == 8192 and res['value'] > 0 and res['value_bf16'] > 0
    assert len(lines[0]) < 4096                                             # the ONE compact stdout line (tests/test_bench_line_cpu.py)
    # attribution of the collective and of a slow rank: present at N > 1
    assert res['allreduce_ms_per_step'] > 0 and res['allreduce_calls_per_step'] == 1.0
    assert 0 < res['rank_ms_per_step_min'] <= res['rank_ms_per_step_max'] <= res['ms_per_step'] * 1.001
    full = json.load(open(os.path.join(ROOT, res['full_report'])))          # the nested blocks live in the full report (the line names the file)
    assert full['value'] == res['value'] and full['n_gpus'] == 8
    # the scaling line's own parity evidence: reduced shard gradients == whole-batch gradient, identical parameters after K steps
    print('8 ranks on one GPU: reduced shard gradients vs whole-batch gradient, rel L2', res['grad_allreduce_vs_whole_batch'])
    assert 0 <= res['grad_allreduce_vs_whole_batch'] <= 1e-5 and res['ranks_param_identical'] is True
    assert full['sharding_check']['ranks'] == 8 and full['sharding_check']['global_rows'] == 8192 and full['sharding_check']['passed'] is True
    assert full['render']['fp32']['row_strips'] == 8 and res['render_ms_per_frame'] > 0
    assert 'VN_EXP=unset' in full['build_info']


def test_bench_two_ranks_on_one_gpu():
    """The driver's N > 1 launch line (torch.distributed.run, one rank per GPU) with two ranks sharing this GPU over gloo:
    bench.py must come back with ONE JSON line from rank 0 -- every rank has to take part in every collective of the
    timed region AND of the clock-sampling steps behind it."""
    import json
    import subprocess
    env = dict(os.environ, VIPNERF_DIST_BACKEND='gloo')
    port = 32000 + (os.getpid() % 2000)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--configs4-rays', '512',
           '--rays', '1024']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['scaling'] == 'weak' and res['value'] > 0 and res['config']['global_rays'] == 2048
    # N > 1 lines carry the configs[4] arithmetic beside `value`, and the configs[4] block (every rank its own DTU shard)
    full = json.load(open(os.path.join(ROOT, res['full_report'])))          # the nested blocks live in the full report, scalars of them in the line
    print('2 ranks on one GPU: reduced shard gradients vs whole-batch gradient, rel L2', res['grad_allreduce_vs_whole_batch'])
    assert 0 <= res['grad_allreduce_vs_whole_batch'] <= 1e-5 and res['ranks_param_identical'] is True and full['sharding_check']['ranks'] == 2
    assert len(full['runs_ms_per_step']) == 3 and sorted(full['runs_ms_per_step'])[1] == res['ms_per_step']      # the line reports the median run
    assert res['value_bf16'] > 0 and full['configs4_dtu']['n_gpus'] == 2 and full['configs4_dtu']['global_rays'] == 1024
    assert full['configs4_dtu']['bf16']['value'] > 0 and full['configs4_dtu']['bf16']['roofline']['bound'] == 'mfma'
    assert res['configs4_bf16_ms'] == full['configs4_dtu']['bf16']['ms_per_step']
    assert res['dtype'] == 'f32' and res['roofline']['bound'] == 'mfma' and 0 < res['roofline']['frac'] < 1
    assert res['allreduce_ms_per_step'] > 0 and res['rank_ms_per_step_max'] >= res['rank_ms_per_step_min'] > 0
    # the render leg at N > 1: one strip of the 756-row frame per rank, barrier-bracketed maximum over the ranks
    assert full['render']['fp32']['row_strips'] == 2 and res['render_ms_per_frame'] > 0
    # strong scaling: configs[3]'s 65,536 rays would not leave room for two ranks on one GPU; the flag itself with a small total
    cmd2 = cmd[:-2] + ['--scaling', 'strong', '--global-rays', '2048']
    r = subprocess.run(cmd2, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])
    assert res['scaling'] == 'strong' and res['config']['rays_per_gpu'] == 1024 and res['config']['global_rays'] == 2048


# ------------------------------------------------------------------------------------------------ the trainer on two ranks
TR_ITERS = 6


def _trainer(dev, rank, world, tmp):
    import test_hip_train_e2e as e2e
    from TrainerHip01 import TrainerHip
    from data_preprocessors.RayGeneratorHip01 import BatchIndexScheduler, RayGeneratorHip
    n, h, w = 3, 32, 32
    K, poses, images_u8 = e2e.synthetic_scene(n, h, w)
    images = torch.from_numpy(images_u8.astype(np.float32) / 255)
    torch.manual_seed(0)                       # identical initial weights and RNG key on every rank / in the single process
    np.random.seed(0)                          # identical index schedule
    cfg = e2e.configs(TR_ITERS, precision='fp32')
    cfg['sub_batch_size'] = 0                  # one sub-batch: the reference SUMS the sub-batches' mean gradients (Trainer01.py:78-104)
    cfg['model_save_interval'] = 0
    gen = RayGeneratorHip((h, w), K[None], poses, 2.0, 4.0, False, dev, images=images, visibility_prior=torch.ones(n, n - 1, h, w))
    sched = BatchIndexScheduler(n, h, w, num_rays=1024)
    return TrainerHip(cfg, gen, sched, output_dirpath=tmp, rank=rank, world=world)


def _trainer_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                      VIPNERF_DIST_BACKEND='gloo')
    from vipnerf_hip import dist as vdist
    r, w, _ = vdist.init_from_env(backend='gloo')
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    tr = _trainer(dev, r, w, None)
    hist = tr.train()
    ret[rank] = torch.cat([p.detach().flatten() for p in tr.model.parameters()]).cpu().numpy()
    ret[f'mse{rank}'] = [x['MSEHip01'] for x in hist]
    torch.distributed.destroy_process_group()


def test_trainer_two_ranks_equal_one_process():
    """TrainerHip01 with world = 2 (each rank: its row-class-aware shard of the scheduler's batch, the batch's own random
    numbers by global ray index, one all-reduce of the flat gradient bucket, the same Adam step) against world = 1 on the
    whole batch: after 6 iterations the two ranks hold bit-identical parameters, and they equal the single-process
    parameters to rounding (the mean of two half-batch means is the whole-batch mean; only the summation order differs:
    measured 2e-5 relative on the accumulated update)."""
    assert torch.cuda.is_available()
    world, port = 2, 33000 + (os.getpid() % 2000)
    ret = mp.Manager().dict()
    mp.spawn(_trainer_worker, args=(world, port, ret), nprocs=world, join=True)
    assert np.array_equal(ret[0], ret[1]), 'the ranks drifted apart'
    tr = _trainer(torch.device('cuda:0'), 0, 1, None)
    hist = tr.train()
    ref = torch.cat([p.detach().flatten() for p in tr.model.parameters()]).cpu().numpy()
    # Adam's first steps move every weight by ~lr whatever the gradient's size: compare the MOVES, not the weights
    tr0 = _trainer(torch.device('cuda:0'), 0, 1, None)
    w0 = torch.cat([p.detach().flatten() for p in tr0.model.parameters()]).cpu().numpy()
    move_ref, move_2 = ref - w0, ret[0] - w0
    err = np.linalg.norm(move_2 - move_ref) / np.linalg.norm(move_ref)
    print(f'two-rank vs single-process parameter update: relative L2 {err:.2e}')
    assert err < 1e-3, f'two-rank parameter update vs single process: {err:.2e}'
    # rank 0's logged MSE is the mean over ITS half of the rows: the two halves average to the single-process value
    both = 0.5 * (np.array(ret['mse0']) + np.array(ret['mse1']))
    np.testing.assert_allclose(both, [x['MSEHip01'] for x in hist], rtol=5e-3)


# ------------------------------------------------------------------------------------------------ a frame on two ranks
def _render_setup(dev):
    from data_preprocessors.RayGeneratorHip01 import RayGeneratorHip
    from oracle import vipnerf_oracle as vo
    import test_hip_parity as tp
    K = np.array([[60.0, 0, 20.0], [0, 60.0, 15.0], [0, 0, 1]], np.float32)
    poses = np.tile(np.eye(4, dtype=np.float32), (2, 1, 1))
    poses[:, 0, 3] = [-0.1, 0.1]
    gen = RayGeneratorHip((31, 40), K[None], poses, 1.0, 6.0, True, dev)          # 31 rows: uneven strips
    model, _ = tp.make_model(dev, True, vo.init_params(41, scale=1.6, sigma_bias=0.6))
    return model.eval(), gen


def _render_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                      VIPNERF_DIST_BACKEND='gloo')
    from vipnerf_hip import dist as vdist
    from data_preprocessors.RayGeneratorHip01 import predict_frame_sharded
    r, w, _ = vdist.init_from_env(backend='gloo')
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    model, gen = _render_setup(dev)
    full = predict_frame_sharded(model, gen, frame=1, rank=r, world=w)
    ret[rank] = {k: v.numpy() for k, v in full.items()}
    torch.distributed.destroy_process_group()


def test_frame_rendered_by_two_ranks_equals_one_process():
    """predict_frame_sharded: each rank renders its strip of rows (no data-path collective), the strips are gathered on the host
    -- every rank ends up with the frame one process renders, bit for bit."""
    from data_preprocessors.RayGeneratorHip01 import predict_frame
    world, port = 2, 34000 + (os.getpid() % 2000)
    ret = mp.Manager().dict()
    mp.spawn(_render_worker, args=(world, port, ret), nprocs=world, join=True)
    model, gen = _render_setup(torch.device('cuda:0'))
    whole = predict_frame(model, gen, frame=1)
    for r in range(world):
        assert set(ret[r]) == set(whole)
        for k, v in whole.items():
            assert np.array_equal(ret[r][k], v.cpu().numpy()), f'rank {r}: {k}'
    assert ret[0]['image'].shape == (31, 40, 3) and ret[0]['image'].dtype == np.uint8


# ---------------------------------------------------------------------------------------------------- RCCL, first contact
def _rccl_worker(rank, world, port, ret):
    """WORLD_SIZE = 1 over the `nccl` (= RCCL) backend with the single-process shortcuts switched off (vipnerf_hip.dist.FORCE): backend
    initialisation on the real GPU, parameter broadcast, the all-reduce of the flat gradient buffer the HIP backward hands over (in
    place, on the adopted storage), the barrier pattern of bench.py, a second step on the reduced gradients."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                      VIPNERF_DIST_BACKEND='nccl', VIPNERF_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    from oracle import vipnerf_oracle as vo
    from vipnerf_hip import dist as vdist
    import test_hip_parity as tp
    r, w, _ = vdist.init_from_env()
    assert torch.distributed.is_initialized() and torch.distributed.get_backend() == 'nccl' and (r, w) == (0, 1) and vdist._active()
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    b, rb = _global_batch(dev)
    model, cfg = tp.make_model(dev, True, vo.init_params(20, scale=1.6), sparse=True)
    before = [p.detach().clone() for p in model.parameters()]
    vdist.broadcast_parameters(model, src=0)
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, model.parameters()))
    model.train()
    bucket = vdist.FlatGradBucket(model.parameters())
    opt = torch.optim.Adam(model.parameters(), lr=5e-4, fused=True)
    shard = vdist.shard_batch(rb, r, w)
    poses = shard['common_data']['poses']
    for it in range(2):
        bucket.release()
        shard['common_data'] = {'poses': poses}       # a fresh common_data per iteration, as the trainer's loader provides (forward unpacks it in place)
        _step(model, cfg, shard)
        flat = bucket.adopted()
        assert flat is not None and flat.numel() == 1191946
        ref = flat.clone()
        ptr = flat.data_ptr()
        vdist.barrier()
        torch.cuda.synchronize()
        bucket.all_reduce_mean()                 # RCCL all-reduce (sum over one rank) + 1/world on the adopted buffer
        vdist.barrier()
        torch.cuda.synchronize()
        assert bucket.params[0].grad.data_ptr() == ptr, 'reduced in place'
        assert torch.equal(flat, ref), 'a one-rank all-reduce must return its input'
        opt.step()
    t = torch.tensor([3.5], dtype=torch.float64, device=dev)          # the max-over-ranks reduction of bench.py
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    ret['ok'] = float(t.item()) == 3.5
    torch.distributed.destroy_process_group()


def test_rccl_single_rank_collectives_on_the_adopted_gradient_buffer():
    assert torch.cuda.is_available()
    port = 33000 + (os.getpid() % 2000)
    ret = mp.Manager().dict()
    mp.spawn(_rccl_worker, args=(1, port, ret), nprocs=1, join=True)
    assert ret.get('ok') is True


def test_bench_force_dist_single_rank_rccl():
    """bench.py --force-dist: the N > 1 code path of the benchmark (process group over RCCL, broadcast, all-reduce inside the timed
    step, barriers, max-over-ranks) with one rank on the real GPU."""
    import json
    import subprocess
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(34000 + (os.getpid() % 2000)), HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('VIPNERF_DIST_BACKEND', None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--force-dist', '--steps', '3', '--warmup', '1', '--rays', '1024', '--also', '',
           '--no-cpu-baseline', '--no-render']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res['n_gpus'] == 1 and res['value'] > 0 and res['config'].get('collectives') == 'forced (nccl, world_size 1)'
    # the self-check of the sharded step runs on this path too (one rank: the shard is the batch, the collective is RCCL's): fields present, run passed
    assert 0 <= res['grad_allreduce_vs_whole_batch'] <= 1e-7 and res['ranks_param_identical'] is True


def test_bench_self_check_at_the_eight_gpu_global_batch_size():
    """The self-check of `bench.py --gpus 8` has every rank compute the gradient of the WHOLE global batch -- 8 x 4096 = 32,768 rays -- through the
    re-rendering backward in <= 8192-ray chunks under its workspace cap.  One GPU runs that pass at that size here (`--verify-rows 32768`; with one rank
    the shard is the batch, so both passes take it and must agree bit for bit): the first 8-GPU run must not be the first time this size executes."""
    import json
    import subprocess
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(36000 + (os.getpid() % 2000)), HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('VIPNERF_DIST_BACKEND', None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--force-dist', '--verify-rows', '32768', '--steps', '2', '--warmup', '1', '--repeats', '1',
           '--also', '', '--no-cpu-baseline', '--no-render', '--no-configs4', '--no-configs2', '--no-sizes']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])
    full = json.load(open(os.path.join(ROOT, res['full_report'])))
    print('self-check at 32,768 rows on one rank:', full['sharding_check'])
    assert full['sharding_check']['global_rows'] == 32768 and full['sharding_check']['passed'] is True
    assert res['grad_allreduce_vs_whole_batch'] <= 1e-7 and full['sharding_check']['whole_norm'] > 0


def test_uneven_row_classes_trim_instead_of_raising():
    """the short last batch of an epoch / an odd number of sparse-depth pixels must not stop a multi-rank run."""
    from vipnerf_hip import dist as vdist
    dev = torch.device('cuda:0')
    n_nerf, n_sd = 7, 5
    batch = {'rays_o': torch.zeros(n_nerf + n_sd, 3, device=dev), 'row_class_counts': (n_nerf, n_sd),
             'indices_mask_nerf': torch.tensor([True] * n_nerf + [False] * n_sd, device=dev),
             'indices_mask_sparse_depth': torch.tensor([False] * n_nerf + [True] * n_sd, device=dev)}
    with pytest.raises(ValueError):
        vdist.shard_row_ids(batch, 0, 2)
    ids = [vdist.shard_row_ids(batch, r, 2, uneven='trim').cpu().tolist() for r in range(2)]
    assert ids == [[0, 1, 2, 7, 8], [3, 4, 5, 9, 10]]
    no_counts = {k: v for k, v in batch.items() if k != 'row_class_counts'}      # the device-mask path agrees
    assert [vdist.shard_row_ids(no_counts, r, 2, uneven='trim').cpu().tolist() for r in range(2)] == ids
