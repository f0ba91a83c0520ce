#!/usr/bin/env python
"""Benchmark of the ViP-NeRF per-ray hot path on MI355X (BASELINE.json: train rays/sec + full-frame render ms,
LLFF-fern 2-view geometry, synthetic data).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one training iteration over one ray batch per GPU (BASELINE configs[1]: 4096 rays x (64 + 128) samples, coarse +
fine 8x256 MLP, fp32): forward -> fused losses (MSE 1, Visibility 0.1, VisibilityPrior 0.001 @ iter 40000) -> backward ->
[RCCL all-reduce of the flat gradient bucket] -> Adam.  Batches are generated and resident in HBM before the timed region;
the random numbers of a step are drawn on the device (Philox) inside it.  Rank 0 prints ONE JSON line.

`value` / `dtype` are the EXACT-fp32 MFMA arithmetic (v_mfma_f32_16x16x4_f32 / 32x32x2_f32), as configs[1] says; the faster split
arithmetics are timed by the same procedure (W warm-up + K timed steps each) and reported beside it (`value_fp16x3`, ...),
each with its own `roofline` block (MFMA-bound path, algorithmic 630,272 MAC/point against the dense MFMA peak
of the operand dtype).  `--scaling strong` runs BASELINE configs[3]'s statement (65,536 rays per iteration split over the
ranks) instead of the weak-scaling default (4096 rays per GPU).
"""
import argparse
import json
import os
import sys
import threading
import time

# multi-process GPU work on this platform needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails with the legacy mode); the environment
# exports it already -- kept here for launchers that build their own environment.  Must be set before the HSA runtime starts.
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd'))
sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd', 'src'))

MAC_PER_POINT = 630272          # trunk+sigma+feature 556,800 + 2 x 36,736 view-branch evaluations (V = 1)
POINTS_PER_RAY = 64 + 192
# SURVEY.md 8d algorithmic bytes: ~105 B/ray in + ~72 B/ray per-ray outputs + 10,240 B/ray of per-sample training outputs,
# + the weights once per launch and the gradients once per step (4.77 MB each)
ALGO_BYTES_PER_RAY = 105 + 72 + 10240
ALGO_BYTES_FIXED = 2 * 4767784
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 4 SIMDs x 256 FLOP/clk... @ 2.4 GHz
F16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense fp16 / bf16 MFMA
PEAK_CLOCK_MHZ = 2400.0         # the clock both peaks assume
# What the chip SUSTAINS for dense 16-bit MFMA from registers alone, operands that look like data, over seconds (tools/
# mfma_f16_sustained.hip next to rocm-smi, profiles/r02_mfma_f16_sustained.txt): 1686 (fp16) / 1708 (bf16) TFLOP/s -- power management
# holds the 16-bit MFMA rate at 0.68 of the nominal figure whatever the kernel does.  Reported beside the contract's frac, never instead.
F16_MFMA_SUSTAINED_TFLOPS = 1690.0
HBM_PEAK_GBS = 8000.0
# arithmetic -> (dtype string, dense MFMA peak of the operand dtype, MFMAs issued per multiply-add, note)
ARITH = {
    'fp32': ('f32', FP32_MFMA_PEAK_TFLOPS, 1, 'exact fp32 operands, products and accumulation: v_mfma_f32_16x16x4_f32 in the MLP kernels (two '
             'waves per SIMD), v_mfma_f32_32x32x2_f32 in the weight-gradient GEMMs; one MFMA per product'),
    'fp16x3': ('f32 emulated: operands split into 2 fp16 parts, 3 fp16 MFMAs per product, fp32 accumulate (22-bit operands, '
               'fp16 exponent range with power-of-two scaling; fp32-grade error on the goldens)', F16_MFMA_PEAK_TFLOPS, 3, ''),
    'bf16x6': ('f32 emulated: operands split into 3 bf16 parts, 6 bf16 MFMAs per product, fp32 accumulate', F16_MFMA_PEAK_TFLOPS, 6, ''),
    'bf16x3': ('f32 emulated: 2 bf16 parts, 3 bf16 MFMAs per product (~5e-6 relative error)', F16_MFMA_PEAK_TFLOPS, 3, ''),
    'fp16x3h': ('mixed: fp16x3 forward / data gradients; activations and gradients stored as fp16 for the weight gradients '
                '(1 fp16 MFMA per product there)', F16_MFMA_PEAK_TFLOPS, 3, ''),
    'fp16': ('f16 operands (rounded once, power-of-two scaling), ONE fp16 MFMA per product, fp32 accumulate; trunk activations / '
             'gradients stored as fp16; fp32 master weights, encodings, heads, compositing, losses (BASELINE configs[4]-style mixed '
             'precision; ~1e-3 relative gradient error)', F16_MFMA_PEAK_TFLOPS, 1, ''),
    'bf16': ('bf16 operands (rounded once), ONE bf16 MFMA per product in the forward / data-gradient GEMMs, fp32 accumulate and '
             'storage (BASELINE configs[4]-style mixed precision; ~1e-2 relative gradient error)', F16_MFMA_PEAK_TFLOPS, 1, ''),
}
# kernels of a step (profile scopes of the library) and the share of a pass's algorithmic MACs each one carries: the eight 256x256
# weight-gradient GEMMs of an MLP are 8 x 65,536 of the 630,272 MAC/point, the thin GEMMs (encoding columns, view branch, heads) the rest
STAGE_GROUPS = {'mlp_fwd': ('mlp_fwd_coarse', 'mlp_fwd_fine'), 'mlp_dgrad': ('mlp_dgrad_coarse', 'mlp_dgrad_fine'),
                'wgrad_256x256': ('wgrad_256x256',), 'wgrad_small': ('wgrad_small',)}
STAGE_MACS = {'mlp_fwd': MAC_PER_POINT, 'mlp_dgrad': MAC_PER_POINT, 'wgrad_256x256': 8 * 65536, 'wgrad_small': MAC_PER_POINT - 8 * 65536}
STAGE_KERNEL = {'mlp_fwd': 'k_mlp_fwd* (MLP forward: one launch per level)', 'mlp_dgrad': 'k_mlp_bwd* (MLP data gradients: one launch per level)',
                'wgrad_256x256': 'k_wgrad<2,8,4> / k_wgrad_*_256 (the eight 256x256 weight-gradient GEMMs: one launch per level)',
                'wgrad_small': 'the thin weight-gradient GEMMs + the chunk reduction (six launches per level)'}


def model_configs():
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


# ---------------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(vo, n_rays=4096, warm=2, timed=3, sweep=(8, 16, 24, 32, 48, 64, 128), sweep_rays=1024):
    """The CPU oracle (a PyTorch-eager restatement pinned to the reference, kind='port'; structure-equivalent incl. the
    reference's chunk = 4096 / netchunk = 16384 host loops) doing the SAME training step on the host cores (SURVEY.md 8d:
    the 4096-ray batch, 2 warm-ups -- the first steps are page-fault bound --, >= 3 timed).  The thread count is chosen by
    a quick sweep on 1024-ray steps: eager PyTorch on 2 x 64 cores is not fastest with every hardware thread."""
    torch.manual_seed(0)
    params = vo.params_to_torch(vo.init_params(0), requires_grad=True)
    opt = torch.optim.Adam(list(params.values()), lr=5e-4, betas=(0.9, 0.999))
    lcfg = [{'name': 'MSE01', 'weight': 1}, {'name': 'VisibilityLoss01', 'weight': 0.1},
            {'name': 'VisibilityPriorLoss01', 'iter_weights': {'0': 0, '30000': 0.001}}]
    cfg = {'ndc': True, 'n_coarse': 64, 'n_fine': 128, 'noise_std': 1.0}

    def step(n, it):
        b = vo.synthetic_batch(n, 1000 + it, scene='fern', nf=2)
        rng = vo.synthetic_rng(n, 64, 128, 2000 + it)
        t0 = time.time()
        opt.zero_grad(set_to_none=True)
        out = vo.render_rays(params, b, cfg, rng, train=True, sec_views=True, chunk=4096, netchunk=16384)
        vo.total_loss(b, out, lcfg, 40000)['TotalLoss'].backward()
        opt.step()
        return time.time() - t0

    avail = os.cpu_count() or 1
    default = torch.get_num_threads()
    cands = sorted({t for t in sweep if t <= avail} | {min(default, avail)})
    for i in range(2):
        step(sweep_rays, i)                          # process-level warm-up (allocator growth, page faults)
    rates = {}
    for t in cands:                                  # ascending; stop once more threads clearly lose (each probe costs seconds)
        torch.set_num_threads(t)
        step(sweep_rays, 10)
        rates[t] = sweep_rays / min(step(sweep_rays, 11), step(sweep_rays, 12))
        if rates[t] < 0.75 * max(rates.values()):
            break
    best = max(rates, key=rates.get)
    torch.set_num_threads(best)
    for i in range(warm):
        step(n_rays, 20 + i)
    times = sorted(step(n_rays, 30 + i) for i in range(timed))
    torch.set_num_threads(default)
    return {'value': round(n_rays / times[len(times) // 2], 1), 'unit': 'rays/s', 'cores': best, 'kind': 'port',
            'sample': '%d warm-up + %d timed training steps of %d rays (median; same synthetic workload as the GPU step, CPU oracle in '
                      'PyTorch eager fp32 with the reference\'s chunk 4096 / netchunk 16384 loops); os.cpu_count=%d, thread sweep on '
                      '%d-ray steps: %s' % (warm, timed, n_rays, avail, sweep_rays, {k: round(v, 1) for k, v in rates.items()})}


# ---------------------------------------------------------------------------------------------------- helpers
class ClockSampler:
    """Shader clock under load, sampled from the driver (amdsmi through torch.cuda.clock_rate) by a side thread while a few
    extra steps run AFTER the timed region -- every '... of the 2.4 GHz peak' figure has the DVFS state in it."""

    def __init__(self, dev):
        self.dev, self.samples, self._stop = dev, [], False

    def __enter__(self):
        def run():
            while not self._stop:
                try:
                    self.samples.append(float(torch.cuda.clock_rate(self.dev)))
                except Exception:
                    return
                time.sleep(0.004)
        self.t = threading.Thread(target=run, daemon=True)
        self.t.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        self.t.join(timeout=2)

    def median(self):
        s = sorted(x for x in self.samples if x > 0)
        return s[len(s) // 2] if s else None


def pmc_reference(prec, rays):
    """HBM bytes per step from the committed rocprofv3 --pmc passes of this very command (profiles/r02_pmc_traffic_<prec>.json;
    counters cannot be collected from inside the process).  Used only when the workload matches the one profiled."""
    for name in ('r02_pmc_traffic_%s.json' % prec,):
        try:
            tr = json.load(open(os.path.join(ROOT, 'profiles', name)))
            if tr['workload']['rays_per_gpu'] == rays and tr['workload']['precision'] == prec:
                return tr, name
        except (OSError, KeyError, ValueError):
            pass
    return None, None


def roofline_block(prec, prof, steps, rays, ms_per_step, sclk_mhz):
    """the path is MFMA-bound; frac = algorithmic FLOP of the dominant kernel's launches / its device time
    (HIP events on the launch stream inside the timed region) / dense MFMA peak of the operand dtype."""
    dtype, peak, issued, _ = ARITH[prec]
    stage_ms = {g: sum(prof.get(k, (0, 0.0))[1] for k in ks) / steps for g, ks in STAGE_GROUPS.items()}
    launches = {g: sum(prof.get(k, (0, 0.0))[0] for k in ks) / steps for g, ks in STAGE_GROUPS.items()}
    other_ms = sum(v[1] for k, v in prof.items() if not any(k in ks for ks in STAGE_GROUPS.values())) / steps
    points = POINTS_PER_RAY * rays
    flop_pass = MAC_PER_POINT * 2.0 * points                          # one pass (forward, data gradient or weight gradient) per step
    dom = max(stage_ms, key=stage_ms.get)                             # the kernel with the largest device time per step
    tf = lambda g: STAGE_MACS[g] * 2.0 * points / (stage_ms[g] * 1e-3) / 1e12 if stage_ms[g] > 0 else 0.0
    wg_ms = stage_ms['wgrad_256x256'] + stage_ms['wgrad_small']
    r = {'bound': 'mfma', 'kernel': STAGE_KERNEL[dom], 'achieved': round(tf(dom), 2), 'peak': peak, 'unit': 'TFLOP/s',
         'frac': round(tf(dom) / peak, 4),
         'definition': 'algorithmic MACs of the dominant kernel (%d MAC/point of the pass\'s 630,272) x 2 x %d points per step / device time '
                       'of its launches in a step (mean over %d timed steps, HIP events on the launch stream) / dense MFMA peak of the operand '
                       'dtype; forward, data-gradient and weight-gradient passes each count 630,272 MAC/point (SURVEY.md 8d, the 3x convention)'
                       % (STAGE_MACS[dom], points, steps),
         'avg_launch_ms': round(stage_ms[dom] / max(launches[dom], 1), 4), 'launches_per_step': launches[dom],
         'mfmas_issued_per_product': issued, 'frac_issued': round(min(tf(dom) * issued / peak, 9.99), 4),
         'stages': dict({g: {'ms_per_step': round(stage_ms[g], 3), 'achieved_tflops': round(tf(g), 1), 'frac': round(tf(g) / peak, 4)}
                         for g in stage_ms},
                        wgrad={'ms_per_step': round(wg_ms, 3), 'achieved_tflops': round(flop_pass / (wg_ms * 1e-3) / 1e12, 1) if wg_ms > 0 else 0.0,
                               'frac': round(flop_pass / (wg_ms * 1e-3) / 1e12 / peak, 4) if wg_ms > 0 else 0.0}),
         'other_kernels_ms_per_step': round(other_ms, 3),
         'step_frac': round(3 * flop_pass / (ms_per_step * 1e-3) / 1e12 / peak, 4),
         'sclk_mhz': sclk_mhz, 'traffic': None}
    if sclk_mhz:
        r['frac_at_measured_sclk'] = round(r['frac'] * PEAK_CLOCK_MHZ / sclk_mhz, 4)
    if peak == F16_MFMA_PEAK_TFLOPS:
        r['sustained_peak'] = F16_MFMA_SUSTAINED_TFLOPS
        r['frac_of_sustained'] = round(tf(dom) / F16_MFMA_SUSTAINED_TFLOPS, 4)
        r['frac_issued_of_sustained'] = round(min(tf(dom) * issued / F16_MFMA_SUSTAINED_TFLOPS, 9.99), 4)
        r['sustained_note'] = 'dense 16-bit MFMA rate this chip sustains from registers alone (tools/mfma_f16_sustained.hip: 1.69 PFLOP/s ' \
                              'at 1.75-1.9 GHz, 1.2-1.3 kW): the practical ceiling behind the nominal 2.5 PFLOP/s'
    algo_bytes = ALGO_BYTES_PER_RAY * rays + ALGO_BYTES_FIXED
    r['algorithmic_bytes_per_step'] = algo_bytes
    tr, name = pmc_reference(prec, rays)
    if tr is not None:
        r['traffic'] = tr['bytes_per_step'][dom if dom in tr['bytes_per_step'] else 'wgrad']['total']
        step_bytes = sum(v['total'] for v in tr['bytes_per_step'].values())
        r['traffic_step'] = step_bytes
        r['traffic_ratio'] = round(step_bytes / algo_bytes, 1)
        r['hbm_gbs_step'] = round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1)
        r['traffic_note'] = 'HBM bytes per step (dominant kernel / whole step): FETCH_SIZE x2 + WRITE_SIZE from separate rocprofv3 --pmc ' \
                            'passes of this command, profiles/%s; traffic_ratio = whole step / SURVEY 8d algorithmic bytes' % name
        mb = tr.get('mfma_busy', {})
        if mb:
            r['mfma_busy_pmc'] = mb
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--rays', type=int, default=4096, help='rays per GPU per step (weak scaling)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'])
    ap.add_argument('--global-rays', type=int, default=65536, help='rays per step over all GPUs (strong scaling: BASELINE configs[3])')
    ap.add_argument('--precision', default='fp32', choices=list(ARITH), help='arithmetic of `value` (BASELINE configs[1] says fp32)')
    ap.add_argument('--also', default='fp16x3,fp16', help='comma list of further arithmetics timed the same way ("" = none, "all")')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-rays', type=int, default=4096)
    ap.add_argument('--no-render', action='store_true')
    ap.add_argument('--no-other-precisions', action='store_true', help='same as --also ""')
    args = ap.parse_args()

    from vipnerf_hip import dist as vdist
    from vipnerf_hip import ops
    rank, world, local = vdist.init_from_env()
    if args.gpus != world and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    dev = torch.device(f'cuda:{local % torch.cuda.device_count()}')   # (one GPU per rank; the modulo only serves the
    torch.cuda.set_device(dev)                                        # 2-ranks-on-1-GPU gloo run of the N>1 code path)

    from oracle import vipnerf_oracle as vo       # synthetic-data generator + cpu_baseline leg only
    from models.ModelFactory import get_model
    from loss_functions.LossComputerHip01 import LossComputerHip

    strong = args.scaling == 'strong'
    if strong and args.global_rays % world:
        raise SystemExit(f'--global-rays {args.global_rays} does not split over {world} ranks')
    rays = args.global_rays // world if strong else args.rays
    cfg = model_configs()
    cfg['model']['hip_precision'] = args.precision
    torch.manual_seed(0)
    model = get_model(cfg, None).to(dev)
    vdist.broadcast_parameters(model)
    model.train()
    lossc = LossComputerHip(cfg)
    opt = torch.optim.Adam(model.parameters(), lr=5e-4, betas=(0.9, 0.999), fused=True)   # same update, one kernel
    bucket = vdist.FlatGradBucket(model.parameters())

    n_batches = min(args.steps + args.warmup, 8 if rays > 8192 else 32)      # distinct resident batches, cycled
    batches = [make_batch(vo, rays, 1000 + rank * 100003 + i, dev) for i in range(n_batches)]
    # ray-sharded ranks draw the random numbers of their own rows of the global batch (Philox keyed by global ray index)
    for b in batches:
        b['rng_ray_base'] = rank * rays
    torch.cuda.synchronize()
    it_counter = [40000]

    def step(i):
        b = dict(batches[i % n_batches])
        b['common_data'] = {'poses': batches[i % n_batches]['common_data']['poses']}
        b['iter_num'] = it_counter[0]               # a new iteration number per step: new Philox offset (loss weights: > 30000)
        it_counter[0] += 1
        bucket.release()
        out = model(b)
        losses = lossc.compute_losses(b, out)
        losses['TotalLoss'].backward()
        bucket.all_reduce_mean()
        opt.step()

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    INIT_STEPS = 2

    def timed_run(prec):
        """The contract's procedure for one arithmetic: [2 untimed initialisation passes: kernel loading, the caching
        allocator's multi-GB workspace blocks, Adam's state] W warm-up steps, then EXACTLY K steps between barrier +
        synchronize pairs; max over ranks."""
        model.configs['model']['hip_precision'] = prec
        torch.cuda.empty_cache()                 # the workspace sizes differ between the arithmetics
        for i in range(INIT_STEPS):
            step(i)
        for i in range(args.warmup):
            step(i)
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
        # shader clock under load: EVERY rank runs the extra steps (they contain the collective); rank 0 samples
        cs = ClockSampler(dev) if rank == 0 else None
        if cs is not None:
            cs.__enter__()
        for i in range(max(4, args.steps // 2)):
            step(i)
        torch.cuda.synchronize()
        if cs is not None:
            cs.__exit__(None, None, None)
        barrier()
        return elapsed, prof, (cs.median() if cs is not None else None)

    elapsed, prof, sclk = timed_run(args.precision)
    also = [] if (args.no_other_precisions or world > 1) else \
        ([p for p in ARITH if p != args.precision] if args.also == 'all' else [p for p in args.also.split(',') if p and p != args.precision])
    others = {p: timed_run(p) for p in also}
    model.configs['model']['hip_precision'] = args.precision

    def max_over_ranks(x):
        if world > 1:
            t = torch.tensor([x], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            return float(t.item())
        return x

    def render_bench():
        # full-frame eval render, camera -> uint8 image on the GPU (756 x 1008 rays, no secondary views):
        # on-device ray generation -> coarse+fine eval pass -> post-processing (Tester01.predict_frame's job)
        # N > 1: the frame is cut into N strips of rows, one per rank, no data-path collective; ms_per_frame is
        # the barrier-bracketed maximum over the ranks
        from data_preprocessors.RayGeneratorHip01 import RayGeneratorHip, frame_strip, predict_frame
        import numpy as np
        model.eval()
        K = np.array([[815.1316, 0, 504.], [0, 815.1316, 378.], [0, 0, 1.]], dtype=np.float32)
        poses = np.tile(np.eye(4, dtype=np.float32), (2, 1, 1))
        poses[:, 0, 3] = [-0.1, 0.1]
        gen = RayGeneratorHip((756, 1008), K[None], poses, 1.0, 5.1731, True, dev)
        n = 756 * 1008
        rows = frame_strip(756, rank, world)
        render = {}
        for prec in [args.precision] + [p for p in also if p in ('fp16x3',)]:
            model.configs['model']['hip_precision'] = prec
            torch.cuda.empty_cache()
            ops.profile_enable(True); ops.profile_read()
            for _ in range(2):
                barrier(); t0 = time.perf_counter()
                frame = predict_frame(model, gen, frame=0, rows=rows)
                torch.cuda.synchronize(); rt = max_over_ranks(time.perf_counter() - t0)
            rp = ops.profile_read(); ops.profile_enable(False)
            assert frame['image'].shape == (rows[1] - rows[0], 1008, 3) and frame['image'].dtype == torch.uint8
            mlp_ms = sum(v[1] for k, v in rp.items() if k.startswith('mlp_fwd')) / 2 * world   # rank 0's strip x N: the frame's kernel time
            eval_flop = 593536 * 2.0 * POINTS_PER_RAY * n          # 231.6 TFLOP per frame
            render[prec] = {'ms_per_frame': round(rt * 1e3, 1), 'rays_per_sec': round(n / rt, 1),
                            'mlp_kernel_ms': round(mlp_ms, 1), 'achieved_tflops': round(eval_flop / (mlp_ms * 1e-3) / 1e12, 1),
                            'frac': round(eval_flop / (mlp_ms * 1e-3) / 1e12 / ARITH[prec][1], 4),
                            'frac_issued_of_sustained': round(eval_flop * ARITH[prec][2] / (mlp_ms * 1e-3) / 1e12 /
                                                              (F16_MFMA_SUSTAINED_TFLOPS if ARITH[prec][1] == F16_MFMA_PEAK_TFLOPS else ARITH[prec][1]), 4),
                            'stage_ms': {k: round(v[1] / 2, 3) for k, v in sorted(rp.items())},
                            'row_strips': world}            # N > 1: one strip of rows per GPU, stage_ms = rank 0's strip
        model.configs['model']['hip_precision'] = args.precision
        model.train()
        return render

    render = None if args.no_render else render_bench()          # every rank: its strip of the frame

    if rank != 0:
        torch.distributed.barrier()              # rank 0 finishes its report, then everybody leaves together
        torch.distributed.destroy_process_group()
        return
    ms = elapsed / args.steps * 1e3
    value = rays * world * args.steps / elapsed
    cfg_name = 'configs[3] (65,536 rays/iter ray-sharded over the GPUs)' if strong else 'configs[1]'
    result = {
        'metric': 'train_rays_per_sec', 'value': round(value, 1), 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'init_steps': INIT_STEPS, 'ms_per_step': round(ms, 3), 'higher_is_better': True,
        'scaling': args.scaling, 'vs_baseline': None, 'dtype': ARITH[args.precision][0], 'data': 'synthetic',
        'config': {'workload': 'BASELINE %s: LLFF-fern 2-view geometry, %d rays/iter/GPU x (64+128) samples, coarse+fine 8x256 MLP, '
                               'V=1 secondary view, losses MSE+Visibility+VisibilityPrior, Adam; random numbers drawn on device'
                               % (cfg_name, rays),
                   'rays_per_gpu': rays, 'global_rays': rays * world, 'parallelism': f'ray-sharded dp{world}',
                   'gemm_arithmetic': args.precision, 'arithmetic_note': ARITH[args.precision][3]},
        'roofline': roofline_block(args.precision, prof, args.steps, rays, ms, sclk),
    }
    for p, (el, pr, sc) in others.items():
        pms = el / args.steps * 1e3
        result['value_' + p] = round(rays * args.steps / el, 1)
        result['ms_per_step_' + p] = round(pms, 3)
        result['dtype_' + p] = ARITH[p][0]
        result['roofline_' + p] = roofline_block(p, pr, args.steps, rays, pms, sc)

    if render is not None:
        result['render_ms_per_frame'] = render[args.precision]['ms_per_frame']
        result['render'] = render

    if world == 1 and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline(vo, n_rays=args.cpu_rays)
    print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
