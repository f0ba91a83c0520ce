#!/usr/bin/env python
"""Benchmark of the ViP-NeRF per-ray hot path on MI355X (BASELINE.json: train rays/sec + full-frame render ms,
LLFF-fern 2-view geometry, synthetic data).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one training iteration over one 4096-ray batch per GPU (BASELINE config 2: 64 + 128 samples, coarse + fine
8x256 MLP, fp32): forward -> fused losses (MSE 1, Visibility 0.1, VisibilityPrior 0.001 @ iter 40000) -> backward ->
[RCCL all-reduce of the flat gradient bucket] -> Adam.  Batches are generated and resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd'))
sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd', 'src'))

MAC_PER_POINT = 630272          # trunk+sigma+feature 556,800 + 2 x 36,736 view-branch evaluations (V = 1)
POINTS_PER_RAY = 64 + 192
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16)
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E peak (about 6300 GB/s achievable)
# Weight-gradient stage: bytes the GEMMs of a level must read per point (fp32 operands as stored, each GEMM reading its two
# operands once; V = 1): 8 x (256+256) + 2 x (256+64) [gamma(x)] + (128+256) [view, feature cols] + 1 [sigma head: rides in
# the feature layer's GEMM, only d(sigma) is extra] + 2 x (128+32) [view, direction cols] + 2 x (8+128) [output head] floats
# = 5713 floats = 22,852 B   
WGRAD_BYTES_PER_POINT = 4 * (8 * 512 + 2 * 320 + 384 + 1 + 2 * 160 + 2 * 136)
# fp32-equivalent peak of each arithmetic: the split modes spend 6 / 3 bf16 MFMAs per fp32 multiply-add
PEAK = {'fp32': (FP32_MFMA_PEAK_TFLOPS, 'fp32 MFMA peak (v_mfma_f32_32x32x2_f32)'),
        'bf16x6': (BF16_MFMA_PEAK_TFLOPS / 6, 'dense bf16 MFMA peak 2500 TFLOP/s / 6 cross terms per fp32-grade product'),
        'bf16x3': (BF16_MFMA_PEAK_TFLOPS / 3, 'dense bf16 MFMA peak 2500 TFLOP/s / 3 cross terms per product'),
        'fp16x3': (BF16_MFMA_PEAK_TFLOPS / 3, 'dense fp16 MFMA peak 2500 TFLOP/s / 3 cross terms per fp32-grade product'),
        'fp16x3h': (BF16_MFMA_PEAK_TFLOPS / 3, 'dense fp16 MFMA peak 2500 TFLOP/s / 3 cross terms per product (forward / data gradients)')}
DTYPE = {'fp32': 'f32', 'bf16x6': 'f32 via 3-way bf16 split (6 bf16 MFMAs per product, fp32 accumulate; fp32-grade error)',
         'bf16x3': 'f32 via 2-way bf16 split (3 bf16 MFMAs per product, fp32 accumulate; ~5e-6 relative error)',
         'fp16x3': 'f32 via 2-way fp16 split (3 fp16 MFMAs per product, fp32 accumulate, power-of-two operand scaling; '
                   'fp32-grade error)',
         'fp16x3h': 'mixed: fp16x3 forward / data gradients, trunk activations and gradients stored as fp16 for the weight '
                    'gradients (single fp16 MFMA, ~2e-4 relative gradient error)'}


def model_configs(n_views=2):
    mlp = lambda ns: {'num_samples': ns, 'netdepth': 8, 'netwidth': 256, 'points_positional_encoding_degree': 10,
                      'views_positional_encoding_degree': 4, 'use_view_dirs': True, 'view_dependent_rgb': True,
                      'predict_visibility': True}
    return {'data_loader': {'ndc': True},
            'model': {'name': 'VipNeRFHip01', 'coarse_mlp': mlp(64), 'fine_mlp': mlp(128), 'chunk': 4096,
                      'netchunk': 16384, 'lindisp': False, 'perturb': True, 'raw_noise_std': 1.0, 'white_bkgd': False},
            'losses': [{'name': 'MSEHip01', 'weight': 1}, {'name': 'VisibilityLossHip01', 'weight': 0.1},
                       {'name': 'VisibilityPriorLossHip01', 'iter_weights': {'0': 0, '30000': 0.001}}],
            'device': [0]}


def make_batch(vo, n_rays, seed, dev, iter_num=40000):
    b = vo.synthetic_batch(n_rays, seed, scene='fern', nf=2)
    rb = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items() if k not in ('poses', 'ndc')}
    rb['common_data'] = {'poses': b['poses'][None].clone().to(dev)}
    rb['iter_num'] = iter_num
    return rb


def cpu_baseline(vo, n_rays, steps):
    """The CPU oracle (a PyTorch-eager restatement pinned to the reference, kind='port') doing the same training
    step on the host cores, on a bounded ray sample."""
    torch.manual_seed(0)
    params = vo.params_to_torch(vo.init_params(0), requires_grad=True)
    opt = torch.optim.Adam(list(params.values()), lr=5e-4, betas=(0.9, 0.999))
    lcfg = [{'name': 'MSE01', 'weight': 1}, {'name': 'VisibilityLoss01', 'weight': 0.1},
            {'name': 'VisibilityPriorLoss01', 'iter_weights': {'0': 0, '30000': 0.001}}]
    cfg = {'ndc': True, 'n_coarse': 64, 'n_fine': 128, 'noise_std': 1.0}
    times = []
    for it in range(steps + 1):
        b = vo.synthetic_batch(n_rays, 1000 + it, scene='fern', nf=2)
        rng = vo.synthetic_rng(n_rays, 64, 128, 2000 + it)
        t0 = time.time()
        opt.zero_grad(set_to_none=True)
        out = vo.render_rays(params, b, cfg, rng, train=True, sec_views=True, chunk=4096)
        vo.total_loss(b, out, lcfg, 40000)['TotalLoss'].backward()
        opt.step()
        times.append(time.time() - t0)
    t = sorted(times[1:])[len(times[1:]) // 2]          # median after one warm-up
    return n_rays / t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--rays', type=int, default=4096, help='rays per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-rays', type=int, default=1024)
    ap.add_argument('--cpu-steps', type=int, default=2)
    ap.add_argument('--no-render', action='store_true')
    ap.add_argument('--precision', default='fp16x3', choices=['fp32', 'bf16x6', 'bf16x3', 'fp16x3', 'fp16x3h'],
                    help='MLP GEMM arithmetic of the headline number (all three are parity-tested; see DESIGN.md)')
    ap.add_argument('--no-other-precisions', action='store_true')
    args = ap.parse_args()

    from vipnerf_hip import dist as vdist
    from vipnerf_hip import ops
    rank, world, local = vdist.init_from_env()
    if args.gpus != world and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    dev = torch.device(f'cuda:{local % torch.cuda.device_count()}')   # (one GPU per rank; the modulo only serves the
                                                                      # 2-ranks-on-1-GPU gloo smoke run of the N>1 code path)
    torch.cuda.set_device(dev)

    from oracle import vipnerf_oracle as vo       # synthetic-data generator + cpu_baseline leg only
    from models.ModelFactory import get_model
    from loss_functions.LossComputerHip01 import LossComputerHip

    cfg = model_configs()
    cfg['model']['hip_precision'] = args.precision
    torch.manual_seed(0)
    model = get_model(cfg, None).to(dev)
    vdist.broadcast_parameters(model)
    model.train()
    lossc = LossComputerHip(cfg)
    opt = torch.optim.Adam(model.parameters(), lr=5e-4, betas=(0.9, 0.999), fused=True)   # same update, one kernel
    bucket = vdist.FlatGradBucket(model.parameters())

    n_batches = args.steps + args.warmup
    batches = [make_batch(vo, args.rays, 1000 + rank * 100003 + i, dev) for i in range(n_batches)]
    torch.cuda.synchronize()

    def step(i):
        b = dict(batches[i])
        b['common_data'] = {'poses': batches[i]['common_data']['poses']}
        bucket.release()
        out = model(b)
        losses = lossc.compute_losses(b, out)
        losses['TotalLoss'].backward()
        bucket.all_reduce_mean()
        opt.step()

    # One-time initialisation that is not a property of the steady state: the first calls load kernels, size the caching
    # allocator's multi-GB workspace blocks and create Adam's state.  Two untimed passes (reported as `init_steps`), then
    # the W warm-up steps and the K timed steps the contract asks for.
    INIT_STEPS = 2
    for i in range(INIT_STEPS):
        step(i % n_batches)
    for i in range(args.warmup):
        step(i)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    ops.profile_enable(True)
    ops.profile_read()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    prof = ops.profile_read()
    ops.profile_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank != 0:
        torch.distributed.barrier()              # rank 0 finishes its report, then everybody leaves together
        torch.distributed.destroy_process_group()
        return
    rays_total = args.rays * world * args.steps
    value = rays_total / elapsed

    # roofline of the dominant stage: algorithmic FLOP (reference-equivalent work, 3x convention: forward, data
    # gradient and weight gradient each count 630,272 MAC/point) / device time from HIP events on the launch stream
    groups = {
        'mlp_fwd': ('mlp_fwd_coarse', 'mlp_fwd_fine'),
        'mlp_dgrad': ('mlp_dgrad_coarse', 'mlp_dgrad_fine'),
        'wgrad': ('wgrad_256x256', 'wgrad_small'),
    }
    stage_ms = {g: sum(prof.get(k, (0, 0.0))[1] for k in ks) / args.steps for g, ks in groups.items()}
    other_ms = sum(v[1] for k, v in prof.items() if not any(k in ks for ks in groups.values())) / args.steps
    dom = max(stage_ms, key=stage_ms.get)
    flop_per_launch = MAC_PER_POINT * 2.0 * POINTS_PER_RAY * args.rays
    achieved = flop_per_launch / (stage_ms[dom] * 1e-3) / 1e12 if stage_ms[dom] > 0 else 0.0
    peak, peak_note = PEAK[args.precision]
    if dom == 'wgrad':
        # the weight-gradient GEMMs stream both operands (4 bytes per element: fp32, or fp16 hi/lo pairs) from HBM once per GEMM
        # and are bound by that, not by MFMA
        bytes_per_step = WGRAD_BYTES_PER_POINT * POINTS_PER_RAY * args.rays
        gbs = bytes_per_step / (stage_ms[dom] * 1e-3) / 1e9 if stage_ms[dom] > 0 else 0.0
        roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(gbs / HBM_PEAK_GBS, 4), 'traffic': None,
                    'peak_note': 'HBM3E peak; algorithmic bytes = %d B/point x %d points per step (operands of the GEMMs of '
                                 'each level, read once per GEMM)' % (WGRAD_BYTES_PER_POINT, POINTS_PER_RAY * args.rays)}
    else:
        roofline = {'bound': 'mfma', 'kernel': dom, 'achieved': round(achieved, 2), 'peak': round(peak, 1),
                    'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4), 'traffic': None, 'peak_note': peak_note,
                    'frac_of_fp32_mfma_peak': round(achieved / FP32_MFMA_PEAK_TFLOPS, 4)}
    # the MFMA-bound stages, always reported next to the dominant one
    roofline['mfma_stages'] = {k: {'achieved_tflops': round(flop_per_launch / (stage_ms[k] * 1e-3) / 1e12, 1),
                                   'frac': round(flop_per_launch / (stage_ms[k] * 1e-3) / 1e12 / peak, 4)}
                               for k in ('mlp_fwd', 'mlp_dgrad') if stage_ms[k] > 0}
    roofline.update({'stage_ms_per_step': {k: round(v, 3) for k, v in stage_ms.items()},
                     'other_kernels_ms_per_step': round(other_ms, 3),
                     'step_flop_frac': round(3 * flop_per_launch / (elapsed / args.steps) / 1e12 / peak, 4)})
    # HBM bytes of the dominant stage per step: PMC counters cannot be collected from inside this process; the value
    # is the committed rocprofv3 --pmc measurement of this very command (profiles/r01_pmc_traffic.json), used only
    # when the workload matches the one profiled
    try:
        tr = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')))
        if tr['workload']['rays_per_gpu'] == args.rays and tr['workload']['precision'] == args.precision:
            roofline['traffic'] = tr['bytes_per_step'][dom]['total']
            roofline['traffic_note'] = ('HBM bytes per step of this stage, FETCH_SIZE x2 + WRITE_SIZE from separate rocprofv3 '
                                        '--pmc passes (profiles/r01_pmc_*_fp16x3.txt, profiles/r01_pmc_traffic.json)')
    except (OSError, KeyError, ValueError):
        pass

    result = {
        'metric': 'train_rays_per_sec', 'value': round(value, 1), 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'init_steps': INIT_STEPS, 'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': DTYPE[args.precision], 'data': 'synthetic',
        'config': {'workload': 'LLFF-fern 2-view geometry, %d rays/iter/GPU x (64+128) samples, coarse+fine 8x256 MLP, '
                               'V=1 secondary view, losses MSE+Visibility+VisibilityPrior, Adam' % args.rays,
                   'rays_per_gpu': args.rays, 'parallelism': f'ray-sharded dp{world}', 'gemm_arithmetic': args.precision},
        'roofline': roofline,
    }

    if world == 1 and not args.no_other_precisions:
        # the same step in the other arithmetics (same process, same batches; best of three 5-step groups each)
        others = {}
        for prec in ('fp32', 'bf16x6', 'bf16x3', 'fp16x3', 'fp16x3h'):
            if prec == args.precision:
                continue
            model.configs['model']['hip_precision'] = prec
            torch.cuda.empty_cache()             # the workspace sizes differ between the arithmetics: without this the caching
            for i in range(5):                   # allocator keeps splitting / re-allocating multi-GB blocks inside the timed steps
                step(i % n_batches)
            groups = []                          # best of three groups of five steps: right after a switch of arithmetic the
            for gi in range(3):                  # allocator still re-shapes its multi-GB blocks now and then (one 80 ms step)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for i in range(5):
                    step((args.warmup + 5 * gi + i) % n_batches)
                torch.cuda.synchronize(); groups.append((time.perf_counter() - t0) / 5)
            dt = min(groups)
            others[prec] = {'rays_per_sec': round(args.rays / dt, 1), 'ms_per_step': round(dt * 1e3, 3)}
        model.configs['model']['hip_precision'] = args.precision
        torch.cuda.empty_cache()
        result['other_precisions'] = others
        if 'fp32' in others:      # the exact-fp32 MFMA path (BASELINE configs[1] says fp32), next to the headline arithmetic
            result['value_fp32_mfma'] = others['fp32']['rays_per_sec']

    if world == 1 and not args.no_render:
        # full-frame eval render, camera -> uint8 image on the GPU (756 x 1008 rays, no secondary views):
        # on-device ray generation -> coarse+fine eval pass -> post-processing (Tester01.predict_frame's job)
        from data_preprocessors.RayGeneratorHip01 import RayGeneratorHip, predict_frame
        import numpy as np
        model.eval()
        K = np.array([[815.1316, 0, 504.], [0, 815.1316, 378.], [0, 0, 1.]], dtype=np.float32)
        poses = np.tile(np.eye(4, dtype=np.float32), (2, 1, 1))
        poses[:, 0, 3] = [-0.1, 0.1]
        gen = RayGeneratorHip((756, 1008), K[None], poses, 1.0, 5.1731, True, dev)
        ops.profile_enable(True); ops.profile_read()
        for _ in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            frame = predict_frame(model, gen, frame=0)
            torch.cuda.synchronize(); rt = time.perf_counter() - t0
        rp = ops.profile_read(); ops.profile_enable(False)
        assert frame['image'].shape == (756, 1008, 3) and frame['image'].dtype == torch.uint8
        n = 756 * 1008
        result['render_ms_per_frame'] = round(rt * 1e3, 1)
        result['render_rays_per_sec'] = round(n / rt, 1)
        result['render_stage_ms'] = {k: round(v[1] / 2, 3) for k, v in sorted(rp.items())}
        model.train()

    if world == 1 and not args.no_cpu_baseline:
        cores = torch.get_num_threads()
        v = cpu_baseline(vo, args.cpu_rays, args.cpu_steps)
        result['cpu_baseline'] = {'value': round(v, 1), 'unit': 'rays/s', 'cores': cores, 'kind': 'port',
                                  'sample': '%d training steps of %d rays (same synthetic workload, CPU oracle, fp32, '
                                            'os.cpu_count=%d)' % (args.cpu_steps, args.cpu_rays, os.cpu_count())}
    print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
