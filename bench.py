#!/usr/bin/env python
"""Benchmark of the ViP-NeRF per-ray hot path on MI355X (BASELINE.json: train rays/sec + full-frame render ms,
LLFF-fern 2-view geometry, synthetic data).

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: bench.py spawns its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             (a WORLD_SIZE that contradicts --gpus is an error)

A step = one training iteration over one ray batch per GPU (BASELINE configs[1]: 4096 rays x (64 + 128) samples, coarse +
fine 8x256 MLP, fp32): forward -> fused losses (MSE 1, Visibility 0.1, VisibilityPrior 0.001 @ iter 40000) -> backward ->
[RCCL all-reduce of the flat gradient bucket] -> Adam.  Batches are generated and resident in HBM before the timed region;
the random numbers of a step are drawn on the device (Philox) inside it.  Rank 0 prints ONE JSON line.

`value` / `dtype` are the EXACT-fp32 MFMA arithmetic (v_mfma_f32_16x16x4_f32 / 32x32x2_f32), as configs[1] says; the faster
arithmetics (`--also`: bf16 and fp16x3 by default; fp16x3h, fp16 on request) are timed by the same procedure (W warm-up + K timed steps each) and reported
beside it (`value_fp16x3`, ...), each with its own `roofline` block (MFMA-bound path, algorithmic 630,272 MAC/point
against the dense MFMA peak of the operand dtype).  At N = 1 the line also carries `configs4_dtu`: BASELINE configs[4]'s per-GPU
shard (DTU geometry, non-NDC, 3 views = 2 secondary views, 131,072 / 8 = 16,384 rays per iteration, bf16 / fp16 mixed precision)
timed the same way, `configs2_realestate`: BASELINE configs[2] (RealEstate geometry, 3 views, 2048 nerf + 2048 sparse-depth rows, the
sparse-depth loss in the list; fp32 and bf16), and `sizes`: the batch sizes the reference's shipped configs train at (1024 rays;
2048 + 2048) through the module contract and through the one-call step (vipnerf_train_step).  `--workload fern|realestate|dtu` selects the scene of `value` itself; `--scaling strong` runs the
ray-sharded statement of configs[3] / configs[4] (`--global-rays` rays per iteration split over the ranks) instead of the
weak-scaling default (`--rays` per GPU).  `--force-dist` makes a single process take the multi-rank code path (RCCL process group
with world_size 1, broadcast, all-reduce of the flat gradient bucket inside the timed step, barriers).

The batches come from the product's own on-device batch builder (RayGeneratorHip: synthetic cameras, random images and visibility
priors, rays generated on the GPU); `oracle/` is imported for the `cpu_baseline` leg only.
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

MAC_TRUNK = 556800              # trunk + sigma head + feature layer, per point
MAC_VIEW = 36736                # one view-branch evaluation (1 + V of them per point)
MAC_PER_POINT = MAC_TRUNK + 2 * MAC_VIEW      # 630,272 with V = 1 (configs[1])
POINTS_PER_RAY = 64 + 192
# synthetic scenes: geometry constants of the reference's shipped ModelConfigs (numbers, not code) -- h, w, focal, near, far, ndc, views
SCENES = {'fern': (756, 1008, 815.1316, 1.0, 5.1731, True, 2), 'realestate': (576, 1024, 900.0, 1.0, 133.33, True, 3),
          'dtu': (300, 400, 361.54, 0.09, 5.0, False, 3)}
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
    'fp16x3h': ('mixed: fp16x3 forward / data gradients (fp32-grade outputs and losses); every weight-gradient operand stored as fp16 tiles, '
                'weight gradients 1 fp16 MFMA per product (gradients ~1e-4 relative)', F16_MFMA_PEAK_TFLOPS, 3, ''),
    'fp16': ('f16 operands (rounded once, power-of-two scaling), ONE fp16 MFMA per product, fp32 accumulate; trunk activations / '
             'gradients stored as fp16; fp32 master weights, encodings, heads, compositing, losses (BASELINE configs[4]-style mixed '
             'precision; ~1e-3 relative gradient error)', F16_MFMA_PEAK_TFLOPS, 1, ''),
    'bf16': ('bf16 operands (rounded once), ONE bf16 MFMA per product in every GEMM (forward, data gradients, weight gradients), fp32 '
             'accumulate; activations / gradients stored as the bf16 operands; fp32 master weights, encodings, heads, compositing, losses '
             '(BASELINE configs[4]: bf16 mixed precision; ~1e-2 relative gradient error)', F16_MFMA_PEAK_TFLOPS, 1, ''),
}
# kernels of a step (profile scopes of the library) and the share of a pass's algorithmic MACs each one carries: the eight 256x256
# weight-gradient GEMMs of an MLP are 8 x 65,536 of the 630,272 MAC/point, the thin GEMMs (encoding columns, view branch, heads) the rest
STAGE_GROUPS = {'mlp_fwd': ('mlp_fwd_coarse', 'mlp_fwd_fine'), 'mlp_dgrad': ('mlp_dgrad_coarse', 'mlp_dgrad_fine'),
                'wgrad_256x256': ('wgrad_256x256',), 'wgrad_small': ('wgrad_small',)}
STAGE_KERNEL = {'mlp_fwd': 'k_mlp_fwd* (MLP forward: one launch per level)', 'mlp_dgrad': 'k_mlp_bwd* (MLP data gradients: one launch per level)',
                'wgrad_256x256': 'k_wgrad256_w8 / k_wgrad_*_256 / k_wg16<16,16> (the eight 256x256 weight-gradient GEMMs: one launch per level)',
                'wgrad_small': 'the thin weight-gradient GEMMs + the chunk reduction (six launches per level)'}


def stage_macs(n_sec):
    per_point = MAC_TRUNK + (1 + n_sec) * MAC_VIEW
    return {'mlp_fwd': per_point, 'mlp_dgrad': per_point, 'wgrad_256x256': 8 * 65536, 'wgrad_small': per_point - 8 * 65536}


def stage_macs_executed(n_sec):
    """MACs per point the exact-fp32 kernels EXECUTE (vipnerf_mlp_{fwd,bwd}_f32.hip): forward 36 weight stages x 256 MFMAs x 1024 MAC / 16 points
    (gamma(x) padded to K = 64, the view layer's 256 feature columns evaluated ONCE for all directions) + one 8-tile k-step per direction; data
    gradient 34 stages (the directions' hidden gradients are summed before the one W_v^T product).  The 3x convention prices every pass at the
    algorithmic 630,272: these are what the MFMA pipe really does."""
    return {'mlp_fwd': 36 * 256 * 1024 // 16 + (1 + n_sec) * 4096, 'mlp_dgrad': 34 * 256 * 1024 // 16}


def model_configs(ndc=True, sparse_depth=False):
    mlp = lambda ns: {'num_samples': ns, 'netdepth': 8, 'netwidth': 256, 'points_positional_encoding_degree': 10,
                      'views_positional_encoding_degree': 4, 'use_view_dirs': True, 'view_dependent_rgb': True,
                      'predict_visibility': True}
    losses = [{'name': 'MSEHip01', 'weight': 1}, {'name': 'VisibilityLossHip01', 'weight': 0.1},
              {'name': 'VisibilityPriorLossHip01', 'iter_weights': {'0': 0, '30000': 0.001}}]
    if sparse_depth:            # BASELINE configs[2]: the sparse-depth prior at the reference's weight (RealEstateTrainerTester01.py:249-259)
        losses.append({'name': 'SparseDepthMSEHip01', 'weight': 0.1})
    return {'data_loader': {'ndc': ndc},
            'model': {'name': 'VipNeRFHip01', 'coarse_mlp': mlp(64), 'fine_mlp': mlp(128), 'chunk': 4096,
                      'netchunk': 16384, 'lindisp': False, 'perturb': True, 'raw_noise_std': 1.0, 'white_bkgd': False},
            'losses': losses, 'device': [0]}


def make_scene(name, dev, seed=0, sparse_depth=False):
    """The product's on-device batch builder over a synthetic scene: `views` cameras on a 0.2-wide baseline looking down -z, random
    images and random visibility-prior masks (there is no dataset in the container).  sparse_depth: ~2 % of the pixels carry a
    COLMAP-style sparse depth (uniform in [near, far]) and a reprojection error, -1 elsewhere.  -> RayGeneratorHip"""
    import numpy as np
    from data_preprocessors.RayGeneratorHip01 import RayGeneratorHip
    h, w, f, near, far, ndc, nf = SCENES[name]
    K = np.array([[f, 0, w / 2.], [0, f, h / 2.], [0, 0, 1.]], dtype=np.float32)
    poses = np.tile(np.eye(4, dtype=np.float32), (nf, 1, 1))
    poses[:, 0, 3] = np.linspace(-0.1, 0.1, nf)
    g = torch.Generator(device=dev).manual_seed(seed)
    images = torch.rand(nf, h, w, 3, generator=g, device=dev)
    prior = (torch.rand(nf, nf - 1, h, w, generator=g, device=dev) < 0.5).float()
    sd = se = None
    if sparse_depth:
        rs = np.random.RandomState(seed + 17)
        has = rs.rand(nf, h, w) < 0.02
        sd = np.where(has, rs.uniform(near, min(far, 20.0), (nf, h, w)), -1.0).astype(np.float32)
        se = np.where(has, rs.uniform(0.1, 2.0, (nf, h, w)), -1.0).astype(np.float32)
    return RayGeneratorHip((h, w), K[None], poses, near, far, ndc, dev, images=images, visibility_prior=prior, sparse_depths=sd, sparse_errors=se)


def make_batch(gen, n_rays, seed, iter_num=40000, n_sparse=0):
    """One resident training batch of n_rays random pixels of the scene (rays generated on the GPU by vipnerf_generate_rays); n_sparse > 0
    appends that many sparse-depth rows -- random pixels that carry a sparse depth -- behind them (the reference's 2048 + 2048 layout,
    select_batch_indices, DataPreprocessor01.py:544-563)."""
    import numpy as np
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, gen.n * gen.h * gen.w, (n_rays,), generator=g).numpy()
    if not n_sparse:
        return gen.get_next_batch(iter_num, indices=ids)
    pool = np.nonzero(gen.sparse_depths.cpu().numpy() > 0)[0] if not hasattr(gen, '_sd_pool') else gen._sd_pool
    gen._sd_pool = pool
    pick = pool[torch.randint(0, pool.shape[0], (n_sparse,), generator=g).numpy()]
    return gen.get_next_batch(iter_num, indices=np.concatenate([ids, pick]),
                              row_is_sparse=np.concatenate([np.zeros(n_rays, dtype=bool), np.ones(n_sparse, dtype=bool)]))


def make_batch_oracle(vo, n_rays, seed, dev, iter_num=40000, scene='fern', nf=2):
    """A batch from the ORACLE's generator (tools/ diagnostics that compare against the oracle on the same rays); not used by the benchmark."""
    b = vo.synthetic_batch(n_rays, seed, scene=scene, nf=nf)
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


def pmc_reference(prec, rays, workload):
    """HBM bytes per step from the committed rocprofv3 --pmc passes of this very command (profiles/r04_pmc_traffic_<prec>.json;
    counters cannot be collected from inside the process).  Used only when the workload matches the one profiled; the block
    says which file, of which commit and date, it quotes -- the figure goes stale when a kernel changes without re-profiling."""
    for name in ('r06_pmc_traffic_%s.json' % prec, 'r05_pmc_traffic_%s.json' % prec, 'r04_pmc_traffic_%s.json' % prec, 'r03_pmc_traffic_%s.json' % prec):
        try:
            tr = json.load(open(os.path.join(ROOT, 'profiles', name)))
            if tr['workload']['rays_per_gpu'] == rays and tr['workload']['precision'] == prec and tr['workload'].get('scene', 'fern') == workload:
                return tr, name
        except (OSError, KeyError, ValueError):
            pass
    return None, None


def roofline_block(prec, prof, steps, rays, ms_per_step, sclk_mhz, n_sec=1, workload='fern'):
    """the path is MFMA-bound; frac = algorithmic FLOP of the dominant kernel's launches / its device time
    (HIP events on the launch stream inside the timed region) / dense MFMA peak of the operand dtype."""
    dtype, peak, issued, _ = ARITH[prec]
    macs = stage_macs(n_sec)
    stage_ms = {g: sum(prof.get(k, (0, 0.0))[1] for k in ks) / steps for g, ks in STAGE_GROUPS.items()}
    launches = {g: sum(prof.get(k, (0, 0.0))[0] for k in ks) / steps for g, ks in STAGE_GROUPS.items()}
    other_ms = sum(v[1] for k, v in prof.items() if not any(k in ks for ks in STAGE_GROUPS.values())) / steps
    points = POINTS_PER_RAY * rays
    flop_pass = macs['mlp_fwd'] * 2.0 * points                        # one pass (forward, data gradient or weight gradient) per step
    dom = max(stage_ms, key=stage_ms.get)                             # the kernel with the largest device time per step
    tf = lambda g: macs[g] * 2.0 * points / (stage_ms[g] * 1e-3) / 1e12 if stage_ms[g] > 0 else 0.0
    wg_ms = stage_ms['wgrad_256x256'] + stage_ms['wgrad_small']
    r = {'bound': 'mfma', 'kernel': STAGE_KERNEL[dom], 'achieved': round(tf(dom), 2), 'peak': peak, 'unit': 'TFLOP/s',
         'frac': round(tf(dom) / peak, 4), 'mac_per_point': macs[dom], 'points_per_step': points,
         'avg_launch_ms': round(stage_ms[dom] / max(launches[dom], 1), 4), 'launches_per_step': launches[dom],
         'mfmas_issued_per_product': issued, 'frac_issued': round(min(tf(dom) * issued / peak, 9.99), 4),
         'stages': dict({g: dict({'ms_per_step': round(stage_ms[g], 3), 'achieved_tflops': round(tf(g), 1), 'frac': round(tf(g) / peak, 4)},
                                 **({'mac_per_point_executed': stage_macs_executed(n_sec)[g],
                                     'frac_executed': round(stage_macs_executed(n_sec)[g] * 2.0 * points / (stage_ms[g] * 1e-3) / 1e12 / peak, 4)}
                                    if prec == 'fp32' and g in ('mlp_fwd', 'mlp_dgrad') and stage_ms[g] > 0 else {}))
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
    algo_bytes = ALGO_BYTES_PER_RAY * rays + ALGO_BYTES_FIXED
    r['algorithmic_bytes_per_step'] = algo_bytes
    tr, name = pmc_reference(prec, rays, workload)
    if tr is not None:
        r['traffic'] = tr['bytes_per_step'][dom if dom in tr['bytes_per_step'] else 'wgrad']['total']
        step_bytes = sum(v['total'] for v in tr['bytes_per_step'].values())
        r['traffic_step'] = step_bytes
        r['traffic_ratio'] = round(step_bytes / algo_bytes, 1)
        r['hbm_gbs_step'] = round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1)
        r['traffic_source'] = 'profiles/%s @ %s %s' % (name, str(tr.get('git_head', 'unknown'))[:12], tr.get('date', 'unknown'))
        try:                                     # is the quoted profile one of the CURRENT kernel sources?
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            from pmc_traffic import csrc_sha16
            r['traffic_profile_is_of_current_kernels'] = tr.get('csrc_sha16') == csrc_sha16()
        except Exception:
            r['traffic_profile_is_of_current_kernels'] = None
        mb = tr.get('mfma_busy', {})
        if mb:
            r['mfma_busy_pmc'] = mb
        # the same stages against the HBM roofline: PMC bytes of the stage / its device time in THIS run (a stage that moves its bytes at
        # the part's measured plateau -- 5.5-5.9 TB/s written, 6-6.3 read -- is HBM-bound whatever its MFMA fraction says)
        hv = {}
        for g, b in tr['bytes_per_step'].items():
            ms = wg_ms if g == 'wgrad' else stage_ms.get(g, 0.0)
            if ms > 0:
                hv[g] = {'bytes': b['total'], 'gb_per_s': round(b['total'] / (ms * 1e-3) / 1e9, 1), 'frac_of_8000': round(b['total'] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        r['hbm_view'] = hv
    return r


# Said ONCE in the full report (bench_full.json), never in the stdout line: what the fields of every roofline block mean.
NOTES = {
    'roofline': 'SURVEY.md 8d: the path is MFMA-bound.  achieved = algorithmic MACs of the dominant kernel (mac_per_point) x 2 x points_per_step / '
                'device time of its launches in a step (mean over the timed steps, HIP events on the launch stream inside the timed region); frac = '
                'achieved / dense MFMA peak of the operand dtype (157.3 TFLOP/s fp32, 2500 fp16 / bf16: always the nominal figure); forward, '
                'data-gradient and weight-gradient passes each count the full MAC/point (the 3x convention): step_frac = 3 x pass FLOP / ms_per_step / peak',
    'sustained': 'sustained_peak / frac_of_sustained: what a seconds-long dense 16-bit MFMA stream sustains on this part under its power limit '
                 '(tools/mfma_f16_sustained.hip, docs/HISTORY.md 4.1b); informational -- frac is against the nominal 2.5 PFLOP/s',
    'traffic': 'HBM bytes per step (dominant kernel / whole step): FETCH_SIZE + WRITE_SIZE from separate rocprofv3 --pmc passes of this command, '
               'collected and corrected as MI355X_MICROARCH.md prescribes; quoted from the committed file named in traffic_source -- NOT a measurement '
               'of this run (counters cannot be collected from inside the process); traffic_profile_is_of_current_kernels compares a hash of csrc/; '
               'traffic_ratio = whole step / SURVEY 8d algorithmic bytes; hbm_view = PMC bytes of a stage / its device time in THIS run',
    'sizes': 'ms per training step (K timed steps, same procedure as `value`, per-kernel events off); module = VipNeRFHip.forward -> compute_losses -> '
             'backward -> FlatAdam.step (the reference trainer\'s sequence); onecall = vipnerf_train_step (the same kernels queued by ONE library call)',
    'allreduce': 'allreduce_ms_per_step: HIP events on the launch stream around the one all-reduce (mean over ranks folded into the collective with RCCL) of '
                 'the flat 4.77 MB gradient bucket, per step, rank 0; at world_size 1 (--force-dist) it is the collective\'s latency floor',
    'sharding_check': 'N > 1 / --force-dist, before the timed region: every rank renders ITS shard of one global batch of n_gpus x rays rows and the shard '
                      'gradients are all-reduced (the step\'s own collective); every rank also computes the whole global batch\'s gradient locally; '
                      'grad_allreduce_vs_whole_batch = max over ranks of |reduced - whole| / |whole| (bound GRAD_SHARD_TOL).  After the K timed steps: '
                      'ranks_param_identical = every rank\'s parameters have rank 0\'s bits (integer checksum of the bit patterns + fp64 sum, MAX-reduced).  '
                      'Either failing is reported here and the run exits with status 3',
    'runs': 'runs_ms_per_step: every timed region of K steps run for the arithmetic (--repeats, default 3); value / ms_per_step / roofline are the MEDIAN run\'s',
}

GRAD_SHARD_TOL = 1e-5           # rel L2 of (all-reduced gradient of the ranks' shards) against (gradient of the whole global batch); measured 1e-6..4e-6
COMPACT_LIMIT = 4096            # bytes of the ONE stdout line (the driver reads it with a bounded parser; r04's 27 KB line was not parsed)


def _short(x, nd=4):
    return round(x, nd) if isinstance(x, float) else x


def compact_line(full: dict) -> dict:
    """The ONE stdout line: the contract's keys, the headline `roofline` and `cpu_baseline` objects with scalar members only, and scalar
    extras for everything else the full report (bench_full.json, stderr) holds.  Pure function of the full report (tests/test_bench_line_cpu.py)."""
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data')
    c = {k: full[k] for k in keep if k in full}
    cf = full.get('config', {})
    c['config'] = {k: cf[k] for k in ('workload', 'rays_per_gpu', 'global_rays', 'parallelism', 'gemm_arithmetic', 'collectives') if k in cf}
    r = full.get('roofline') or {}
    c['roofline'] = {k: _short(r[k]) for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_ms', 'launches_per_step', 'step_frac',
                                              'sclk_mhz', 'traffic', 'traffic_ratio', 'algorithmic_bytes_per_step', 'traffic_source',
                                              'traffic_profile_is_of_current_kernels') if k in r}
    for g, st in (r.get('stages') or {}).items():
        c['roofline']['ms_' + g] = st['ms_per_step']
        if 'frac_executed' in st:
            c['roofline']['frac_executed_' + g] = st['frac_executed']
    cb = full.get('cpu_baseline')
    if cb:
        c['cpu_baseline'] = {'value': cb['value'], 'unit': cb['unit'], 'cores': cb['cores'], 'kind': cb['kind'], 'sample': cb['sample'][:160]}
        c['gpu_over_cpu'] = round(full['value'] / cb['value'], 1) if cb['value'] else None
    for k, v in full.items():                         # value_<arith>, ms_per_step_<arith>: scalars as they are
        if (k.startswith('value_') or k.startswith('ms_per_step_')) and isinstance(v, (int, float)):
            c[k] = v
        if k.startswith('roofline_') and isinstance(v, dict):
            c['frac_' + k[9:]] = v.get('frac')
            c['step_frac_' + k[9:]] = v.get('step_frac')
    if 'render' in full:
        for prec, rd in full['render'].items():
            c['render_ms_per_frame' + ('' if prec == full.get('config', {}).get('gemm_arithmetic') else '_' + prec)] = rd['ms_per_frame']
    elif 'render_ms_per_frame' in full:
        c['render_ms_per_frame'] = full['render_ms_per_frame']
    for blk, tag in (('configs2_realestate', 'configs2'), ('configs4_dtu', 'configs4')):
        for prec, b in (full.get(blk) or {}).items():
            if isinstance(b, dict) and 'ms_per_step' in b:
                c['%s_%s_ms' % (tag, prec)] = b['ms_per_step']
                c['%s_%s_frac' % (tag, prec)] = (b.get('roofline') or {}).get('frac')
    for label, b in (full.get('sizes') or {}).items():
        if not isinstance(b, dict):
            continue
        tag = 'sizes_%d' % b.get('rows', 0)
        for prec in ('fp32', 'bf16'):
            for api, e in (b.get(prec) or {}).items():
                c['%s_%s_%s_ms' % (tag, prec, api)] = e['ms_per_step']
    for k in ('ranks_reduced', 'grad_allreduce_vs_whole_batch', 'ranks_param_identical', 'step_api', 'allreduce_ms_per_step', 'allreduce_calls_per_step', 'rank_ms_per_step_min', 'rank_ms_per_step_max',
              'build_info_sha16', 'csrc_sha16', 'full_report'):
        if k in full:
            c[k] = full[k]
    size = lambda d: len(json.dumps(d, separators=(',', ':')))
    if size(c) > COMPACT_LIMIT:                       # never print an unparseable line: shed the extras, keep the contract
        must = set(keep) | {'config', 'roofline', 'cpu_baseline', 'ranks_reduced', 'grad_allreduce_vs_whole_batch', 'ranks_param_identical',
                            'allreduce_ms_per_step', 'ms_per_step_bf16', 'value_bf16', 'render_ms_per_frame', 'full_report'}
        for k in [k for k in c if k not in must]:
            c.pop(k)
        c['truncated'] = True
    if size(c) > COMPACT_LIMIT:                       # still too long (a config / sample string grew): the bare contract with bounded members
        c = {k: c[k] for k in c if k in keep or k in ('ranks_reduced', 'grad_allreduce_vs_whole_batch', 'ranks_param_identical', 'truncated')}
        c['config'] = {'workload': str(cf.get('workload', ''))[:300]}
        c['roofline'] = {k: _short(r[k]) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic') if k in r}
        if cb:
            c['cpu_baseline'] = {'value': cb['value'], 'unit': cb['unit'], 'cores': cb['cores'], 'kind': cb['kind'], 'sample': cb['sample'][:80]}
    assert size(c) <= COMPACT_LIMIT, 'bench.py: the stdout line does not fit %d bytes even as the bare contract' % COMPACT_LIMIT
    return c


def respawn_cmd(gpus, environ, argv):
    """The launcher command `python bench.py --gpus N` turns itself into when N > 1 and no launcher set WORLD_SIZE, else None.
    Raises SystemExit when a launcher's WORLD_SIZE contradicts --gpus."""
    ws = environ.get('WORLD_SIZE')
    if ws is not None:
        if int(ws) != gpus:
            raise SystemExit(f'--gpus {gpus} but WORLD_SIZE={ws}: bench.py reports n_gpus = the ranks that ran, launch it with matching numbers')
        return None
    if gpus <= 1:
        return None
    import socket
    with socket.socket() as sk:                  # a free rendezvous port on the loopback interface
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={gpus}', '--master-addr', '127.0.0.1',
            '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--rays', type=int, default=4096, help='rays per GPU per step (weak scaling)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'])
    ap.add_argument('--global-rays', type=int, default=None, help='rays per step over all GPUs (strong scaling; default 65536 = BASELINE '
                    'configs[3], 131072 = configs[4] with --workload dtu)')
    ap.add_argument('--workload', default='fern', choices=list(SCENES), help='scene geometry of `value` (BASELINE configs[1] / [3]: fern; '
                    'configs[2]: realestate; configs[4]: dtu)')
    ap.add_argument('--precision', default='fp32', choices=list(ARITH), help='arithmetic of `value` (BASELINE configs[1] says fp32)')
    ap.add_argument('--also', default='bf16,fp16x3', help='comma list of further arithmetics timed the same way ("" = none, "all"; default: BASELINE configs[4]\'s bf16 and the fp32-grade 3 x fp16 mode)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-rays', type=int, default=4096)
    ap.add_argument('--no-render', action='store_true')
    ap.add_argument('--no-other-precisions', action='store_true', help='same as --also ""')
    ap.add_argument('--no-configs4', action='store_true', help='skip the configs[4] block (DTU shard in bf16 / fp16)')
    ap.add_argument('--configs4-rays', type=int, default=16384, help='rays per GPU of the configs[4] block (131,072 / 8)')
    ap.add_argument('--optimizer', default='flat', choices=['flat', 'torch-fused'], help='flat: vipnerf_hip.optim.FlatAdam (torch.optim.Adam\'s own '
                    'update on the flat buffers, bit-identical, one launch); torch-fused: torch.optim.Adam(fused=True) (2 x ~100 us multi_tensor_apply)')
    ap.add_argument('--force-dist', action='store_true', help='take the multi-rank code path (process group, broadcast, all-reduce, '
                    'barriers) even with one rank')
    ap.add_argument('--no-configs2', action='store_true', help='skip the configs[2] block (RealEstate geometry, 3 views, 2048 + 2048 sparse-depth rows)')
    ap.add_argument('--no-sizes', action='store_true', help='skip the `sizes` block (the reference\'s shipped batch sizes through vipnerf_train_step)')
    ap.add_argument('--step-api', default='module', choices=['module', 'onecall'], help='how `value` itself steps: module = the reference\'s module '
                    'contract (model() -> compute_losses -> backward -> optimizer.step, Trainer01.py:61-107); onecall = vipnerf_train_step')
    ap.add_argument('--report-path', default=None, help='where the full report goes (default: gpurun_out/bench_full.json when gpurun_out/ exists, else '
                    'bench_full.json next to bench.py); the line names it in `full_report`')
    ap.add_argument('--repeats', type=int, default=3, help='timed regions of K steps each for `value` (and the --also arithmetics): the line reports the MEDIAN '
                    'run (its K steps, its ms_per_step); every run is in the full report under `runs_ms_per_step`')
    ap.add_argument('--verify-rows', type=int, default=None, help='rows of the self-check\'s global batch (default: rays per GPU x ranks); lets ONE GPU run the '
                    'check at the size an 8-GPU run gives it (tests/test_hip_dist.py)')
    ap.add_argument('--check-ranks', action='store_true', help='spawn / join the ranks, count them with one all-reduce, print {"n_gpus", "ranks_reduced"} '
                    'and exit without touching a GPU (the CPU test of the --gpus N launcher)')
    args = ap.parse_args()

    # `python bench.py --gpus N` with N > 1 and no launcher around it: become the launcher (one rank per GPU under torch.distributed.run).  A
    # WORLD_SIZE that contradicts --gpus is an error: the line never reports an n_gpus other than the number of ranks that reduced.
    cmd = respawn_cmd(args.gpus, os.environ, sys.argv[1:])
    if cmd is not None:
        import subprocess
        sys.stdout.flush()
        raise SystemExit(subprocess.call(cmd))

    # stdout carries ONE JSON line, from rank 0, and nothing else: fd 1 is pointed at stderr for the whole run (what a native library prints
    # there -- RCCL's version banner under the platform's NCCL_DEBUG=VERSION, once per rank, buffered until exit -- lands on stderr), and the
    # JSON line is written to the saved original stdout
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    from vipnerf_hip import dist as vdist
    from vipnerf_hip import ops
    rank, world, local = vdist.init_from_env(force=True if args.force_dist else None)
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    ranks_reduced = world
    if vdist._active():                          # how many ranks really take part: one all-reduce of ones
        t = torch.ones(1, device='cuda' if torch.distributed.get_backend() == 'nccl' else 'cpu')
        torch.distributed.all_reduce(t)
        ranks_reduced = int(t.item())
        if ranks_reduced != world:
            raise SystemExit(f'{ranks_reduced} ranks reduced but WORLD_SIZE={world}')
    if args.check_ranks:
        if rank == 0:
            os.write(json_fd, (json.dumps({'n_gpus': world, 'ranks_reduced': ranks_reduced, 'check_ranks': True}) + '\n').encode())
        if torch.distributed.is_initialized():
            vdist.barrier()
            torch.distributed.destroy_process_group()
        return
    from vipnerf_hip import _lib as vlib
    vlib.require_product_build('bench.py')       # a timing-only experiment build (-DVN_EXP=n) would time "faster" unnoticed
    dev = torch.device(f'cuda:{local % torch.cuda.device_count()}')   # (one GPU per rank; the modulo only serves the
    torch.cuda.set_device(dev)                                        # 2-ranks-on-1-GPU gloo run of the N>1 code path)
    collectives = vdist._active()

    from models.ModelFactory import get_model
    from loss_functions.LossComputerHip01 import LossComputerHip

    strong = args.scaling == 'strong'
    global_rays = args.global_rays if args.global_rays is not None else (131072 if args.workload == 'dtu' else 65536)
    if strong and global_rays % world:
        raise SystemExit(f'--global-rays {global_rays} does not split over {world} ranks')
    rays = global_rays // world if strong else args.rays

    def barrier():
        vdist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if collectives:
            t = torch.tensor([x], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            return float(t.item())
        return x

    INIT_STEPS = 2

    class Workload:
        """One scene + ray count: model, optimizer, resident batches, and the contract's timing procedure per arithmetic."""

        def __init__(self, scene, n_rays, precision, n_sparse=0, step_api='module'):
            self.scene, self.rays, self.n_sparse, self.step_api = scene, n_rays, n_sparse, step_api
            self.rows = n_rays + n_sparse             # rows of a batch: nerf rows + sparse-depth rows (configs[2]: 2048 + 2048)
            self.n_sec = SCENES[scene][6] - 1
            self.cfg = model_configs(SCENES[scene][5], sparse_depth=n_sparse > 0)
            self.cfg['model']['hip_precision'] = precision
            torch.manual_seed(0)
            self.model = get_model(self.cfg, None).to(dev)
            vdist.broadcast_parameters(self.model)
            self.model.train()
            self.lossc = LossComputerHip(self.cfg)
            if args.optimizer == 'flat' or step_api == 'onecall':     # torch's single-tensor Adam expressions on ONE flat parameter / moment / gradient buffer
                from vipnerf_hip.optim import FlatAdam                       # (bit-identical to torch.optim.Adam; one library launch per step)
                self.opt = FlatAdam(self.model.parameters(), lr=5e-4, betas=(0.9, 0.999))
            else:
                self.opt = torch.optim.Adam(self.model.parameters(), lr=5e-4, betas=(0.9, 0.999), fused=True)
            self.bucket = vdist.FlatGradBucket(self.model.parameters())
            self.stepper = None
            if step_api == 'onecall':                # ONE library call per iteration (vipnerf_train_step); ranks reduce the flat gradient between backward and Adam
                from vipnerf_hip.step import FusedTrainStep
                self.stepper = FusedTrainStep(self.model, self.cfg, self.opt, bucket_reduce=vdist.all_reduce_mean_flat if collectives else None)
            self.gen = make_scene(scene, dev, sparse_depth=n_sparse > 0)
            self.n_batches = min(args.steps + args.warmup, 8 if self.rows > 8192 else 32)      # distinct resident batches, cycled
            self.batches = [make_batch(self.gen, n_rays, 1000 + rank * 100003 + i, n_sparse=n_sparse) for i in range(self.n_batches)]
            for b in self.batches:         # ray-sharded ranks draw the random numbers of their own rows of the global batch (Philox
                b['rng_ray_base'] = rank * self.rows   # keyed by global ray index)
            torch.cuda.synchronize()
            self.it = 40000

        def step(self, i):
            src = self.batches[i % self.n_batches]
            b = dict(src)
            b['common_data'] = {'poses': src['common_data']['poses']}
            b['iter_num'] = self.it                   # a new iteration number per step: new Philox offset (loss weights: > 30000)
            self.it += 1
            if self.stepper is not None:
                self.stepper(b)
                return
            self.bucket.release()
            out = self.model(b)
            losses = self.lossc.compute_losses(b, out)
            losses['TotalLoss'].backward()
            self.bucket.all_reduce_mean()
            self.opt.step()

        def timed_run(self, prec, profile=True, repeats=1):
            """The contract's procedure for one arithmetic: [2 untimed initialisation passes: kernel loading, the caching
            allocator's multi-GB workspace blocks, Adam's state] W warm-up steps, then EXACTLY K steps between barrier +
            synchronize pairs; max over ranks.  repeats > 1: that timed region `repeats` times back to back (each its own barrier pair and K
            steps); the run reported is the MEDIAN one -- its elapsed time, its per-kernel events, its collective times -- and every run's
            ms per step is kept in self.last_runs (box noise of +- 0.15 ms per step is as large as a kernel change worth keeping)."""
            self.model.configs['model']['hip_precision'] = prec
            if self.stepper is not None:
                self.stepper.release()               # the one-call path's persistent buffers of the previous arithmetic
            torch.cuda.empty_cache()                 # the workspace sizes differ between the arithmetics
            for i in range(INIT_STEPS):
                self.step(i)
            for i in range(args.warmup):
                self.step(i)
            runs = []
            for rep in range(max(1, repeats)):
                ops.profile_enable(profile)
                ops.profile_read()
                vdist.timing_enable(profile and collectives)
                barrier()
                t0 = time.perf_counter()
                for i in range(args.steps):
                    self.step(args.warmup + rep * args.steps + i)
                torch.cuda.synchronize()
                own = time.perf_counter() - t0           # this rank's own K steps (before the closing barrier): min / max over ranks attribute a slow rank
                barrier()
                elapsed = time.perf_counter() - t0
                prof = ops.profile_read()
                ops.profile_enable(False)
                n_ar, ar_ms = vdist.timing_read()
                vdist.timing_enable(False)
                elapsed = max_over_ranks(elapsed)
                extra = {'allreduce_ms_per_step': round(ar_ms / args.steps, 4) if n_ar else None, 'allreduce_calls_per_step': n_ar / args.steps,
                         'rank_ms_per_step_min': round(-max_over_ranks(-own) / args.steps * 1e3, 3),
                         'rank_ms_per_step_max': round(max_over_ranks(own) / args.steps * 1e3, 3)}
                runs.append((elapsed, prof, extra))
            order = sorted(range(len(runs)), key=lambda j: runs[j][0])     # (elapsed is already the max over ranks: every rank picks the same run)
            elapsed, prof, self.last_extra = runs[order[(len(runs) - 1) // 2]]
            self.last_runs = [round(r[0] / args.steps * 1e3, 3) for r in runs]
            if not profile:
                return elapsed, prof, None
            # shader clock under load: EVERY rank runs the extra steps (they contain the collective); rank 0 samples
            cs = ClockSampler(dev) if rank == 0 else None
            if cs is not None:
                cs.__enter__()
            for i in range(max(4, args.steps // 2)):
                self.step(i)
            torch.cuda.synchronize()
            if cs is not None:
                cs.__exit__(None, None, None)
            barrier()
            return elapsed, prof, (cs.median() if cs is not None else None)

        def verify_sharding(self, precision):
            """Before the timed region of a multi-rank run: the all-reduced gradient of the ranks' shards of ONE global batch (world x rays rows,
            the same on every rank) against the gradient of the whole global batch computed locally by every rank (vdist.verify_sharded_gradient)."""
            self.model.configs['model']['hip_precision'] = precision
            gb = make_batch(self.gen, args.verify_rows or self.rays * world, 777, n_sparse=self.n_sparse * world)
            # The whole-batch pass (world x rays rows on EVERY rank) runs through the re-rendering backward in chunks of <= 8192 rays (autograd.py:
            # the path tests/test_hip_fullsize.py holds at 65,536 rays) under a workspace cap, instead of keeping 5.4 MB of activations per ray
            # for 32,768+ rays at once.  Ranks SHARING a device (the one-GPU tests of this path) each get an equal share of half of it.
            sharing = -(-world // max(torch.cuda.device_count(), 1))
            cap_key, m = 'hip_max_workspace_bytes', self.model.configs['model']
            old_cap = m.get(cap_key)
            m[cap_key] = int(min(48 << 30, torch.cuda.mem_get_info(dev)[1] // (2 * sharing)))

            def grad_fn(batch):
                b = dict(batch)
                b['common_data'] = {'poses': batch['common_data']['poses']}
                b['iter_num'] = 40000
                self.model.injected_rng = {'offset': 40000 << 16}      # both passes draw from the same Philox offset (a row's numbers depend on its global index only)
                try:
                    self.bucket.release()
                    out = self.model(b)
                    self.lossc.compute_losses(b, out)['TotalLoss'].backward()
                finally:
                    self.model.injected_rng = None
                flat = self.bucket.adopted()
                return flat if flat is not None else torch.cat([p.grad.reshape(-1) for p in self.bucket.params])

            try:
                res = vdist.verify_sharded_gradient(grad_fn, gb, rank, world)
            finally:
                if old_cap is None:
                    m.pop(cap_key, None)
                else:
                    m[cap_key] = old_cap
            self.bucket.release()
            torch.cuda.empty_cache()
            return res

        def release(self):
            if self.stepper is not None:
                self.stepper.release()
            self.model = self.opt = self.bucket = self.batches = self.gen = self.stepper = None
            torch.cuda.empty_cache()

    main_wl = Workload(args.workload, rays, args.precision, step_api=args.step_api)
    model = main_wl.model
    # N > 1 (or --force-dist): the scaling line carries its own parity evidence -- the reduced sharded gradient against the whole-batch gradient
    # BEFORE the timed region, bit-identical parameters on every rank AFTER it; a failure is reported in the line and the run exits non-zero
    verify = main_wl.verify_sharding(args.precision) if collectives else None
    elapsed, prof, sclk = main_wl.timed_run(args.precision, repeats=args.repeats)
    main_extra, main_runs = main_wl.last_extra, main_wl.last_runs
    same = vdist.params_identical(model.parameters()) if collectives else None
    verify_failed = bool(collectives and (not (verify['rel_l2'] <= GRAD_SHARD_TOL) or not same['identical']))
    # N > 1 (the driver's scaling runs): `value` plus the configs[4] arithmetic only, unless --also is given explicitly
    also_arg = args.also if (world == 1 or '--also' in sys.argv) else 'bf16'
    also = [] if args.no_other_precisions else \
        ([p for p in ARITH if p != args.precision] if also_arg == 'all' else [p for p in also_arg.split(',') if p and p != args.precision])
    others, other_runs = {}, {}
    for p in also:
        others[p] = main_wl.timed_run(p, repeats=args.repeats)
        other_runs[p] = main_wl.last_runs
    model.configs['model']['hip_precision'] = args.precision

    def render_bench():
        # full-frame eval render, camera -> uint8 image on the GPU (756 x 1008 rays, no secondary views):
        # on-device ray generation -> coarse+fine eval pass -> post-processing (Tester01.predict_frame's job)
        # N > 1: the frame is cut into N strips of rows, one per rank, no data-path collective; ms_per_frame is
        # the barrier-bracketed maximum over the ranks
        from data_preprocessors.RayGeneratorHip01 import frame_strip, predict_frame
        model.eval()
        gen = main_wl.gen
        n = gen.h * gen.w
        rows = frame_strip(gen.h, rank, world)
        render = {}
        for prec in [args.precision] + [p for p in also if p in ('fp16x3', 'bf16')]:
            model.configs['model']['hip_precision'] = prec
            torch.cuda.empty_cache()
            ops.profile_enable(True); ops.profile_read()
            for _ in range(2):
                barrier(); t0 = time.perf_counter()
                frame = predict_frame(model, gen, frame=0, rows=rows)
                torch.cuda.synchronize(); rt = max_over_ranks(time.perf_counter() - t0)
            rp = ops.profile_read(); ops.profile_enable(False)
            assert frame['image'].shape == (rows[1] - rows[0], gen.w, 3) and frame['image'].dtype == torch.uint8
            mlp_ms = sum(v[1] for k, v in rp.items() if k.startswith('mlp_fwd')) / 2 * world   # rank 0's strip x N: the frame's kernel time
            eval_flop = (MAC_TRUNK + MAC_VIEW) * 2.0 * POINTS_PER_RAY * n          # 231.6 TFLOP per 756 x 1008 frame
            render[prec] = {'ms_per_frame': round(rt * 1e3, 1), 'rays_per_sec': round(n / rt, 1),
                            'mlp_kernel_ms': round(mlp_ms, 1), 'achieved_tflops': round(eval_flop / (mlp_ms * 1e-3) / 1e12, 1),
                            'frac': round(eval_flop / (mlp_ms * 1e-3) / 1e12 / ARITH[prec][1], 4),
                            'stage_ms': {k: round(v[1] / 2, 3) for k, v in sorted(rp.items())},
                            'row_strips': world}            # N > 1: one strip of rows per GPU, stage_ms = rank 0's strip
        model.configs['model']['hip_precision'] = args.precision
        model.train()
        return render

    render = None if args.no_render else render_bench()          # every rank: its strip of the frame

    # BASELINE configs[4] at its per-GPU shard size, every rank its own 16,384 rays: at N = 8 this IS configs[4]'s statement (131,072 rays
    # per iteration ray-sharded over 8 GPUs, one all-reduce per step); `--workload dtu --scaling strong --precision bf16` puts it into `value`
    c4 = None
    if not args.no_configs4 and not (args.workload == 'dtu' and rays == args.configs4_rays):
        main_wl.release()
        wl4 = Workload('dtu', args.configs4_rays, 'bf16')
        c4 = {'workload': 'BASELINE configs[4] per-GPU shard: DTU geometry (non-NDC), 3 views (V = 2 secondary views), 131,072 / 8 = %d rays/iter x '
                          '(64+128) samples, coarse+fine 8x256 MLP, mixed precision (16-bit MFMA operands and activation storage, fp32 master '
                          'weights / accumulation / losses), Adam' % args.configs4_rays, 'rays_per_gpu': args.configs4_rays,
              'global_rays': args.configs4_rays * world, 'n_gpus': world}
        for p in (('bf16', 'fp16') if world == 1 else ('bf16',)):
            el, pr, sc = wl4.timed_run(p)
            pms = el / args.steps * 1e3
            c4[p] = {'value': round(args.configs4_rays * world * args.steps / el, 1), 'unit': 'rays/s', 'ms_per_step': round(pms, 3), 'dtype': ARITH[p][0],
                     'roofline': roofline_block(p, pr, args.steps, args.configs4_rays, pms, sc, n_sec=wl4.n_sec, workload='dtu')}
        wl4.release()

    # BASELINE configs[2]: RealEstate geometry, 3 input views (V = 2), visibility + sparse-depth priors: the reference's batch of 2048 nerf rows
    # + 2048 sparse-depth rows (RealEstateTrainerTester01.py:249-259; SparseDepthMSE at weight 0.1, SparseDepthMSE01.py:58-63), every rank its own batch
    c2 = None
    if not args.no_configs2 and not (args.workload == 'realestate'):
        if c4 is None:
            main_wl.release()
        wl2 = Workload('realestate', 2048, 'fp32', n_sparse=2048)
        c2 = {'workload': 'BASELINE configs[2]: RealEstate geometry (NDC), 3 views (V = 2 secondary views), 2048 nerf rows + 2048 sparse-depth rows per '
                          'iteration x (64+128) samples, coarse+fine 8x256 MLP, losses MSE 1 + Visibility 0.1 + VisibilityPrior 0.001 + SparseDepthMSE 0.1, Adam',
              'rows_per_gpu': 4096, 'nerf_rows': 2048, 'sparse_depth_rows': 2048, 'n_gpus': world, 'mac_per_point': stage_macs(wl2.n_sec)['mlp_fwd']}
        for p in (('fp32', 'bf16') if world == 1 else ('bf16',)):
            el, pr, sc = wl2.timed_run(p)
            pms = el / args.steps * 1e3
            c2[p] = {'value': round(4096 * world * args.steps / el, 1), 'unit': 'rays/s', 'ms_per_step': round(pms, 3), 'dtype': ARITH[p][0],
                     'roofline': roofline_block(p, pr, args.steps, 4096, pms, sc, n_sec=wl2.n_sec, workload='realestate')}
        wl2.release()

    # The batch sizes the reference's shipped configs train at (`num_rays` 1024, or 2048 + 2048 with sparse depth --
    # NerfLlffTrainerTester01.py:251,261,617), through the module contract AND through the one-call step (vipnerf_train_step): at these
    # sizes the 16-bit step is as long as the host needs to enqueue it through five Python -> ctypes calls
    sizes = None
    if not args.no_sizes and world == 1:
        if c4 is None and c2 is None:
            main_wl.release()
        sizes = {'note': 'ms per training step (K timed steps, same procedure as `value`); module = VipNeRFHip.forward -> compute_losses -> backward -> '
                         'FlatAdam.step (the reference trainer\'s sequence); onecall = vipnerf_train_step (the same kernels queued by ONE library call)'}
        for label, scene, n, nsd in (('fern_1024', 'fern', 1024, 0), ('realestate_2048+2048sd', 'realestate', 2048, 2048)):
            sizes[label] = {'rows': n + nsd, 'sparse_depth_rows': nsd, 'scene': scene}
            for api in ('module', 'onecall'):
                wls = Workload(scene, n, 'fp32', n_sparse=nsd, step_api=api)
                for p in ('fp32', 'bf16'):
                    el, pr, sc = wls.timed_run(p)            # (with the per-kernel HIP events on: the kernel time of a step)
                    kern = sum(v[1] for v in pr.values()) / args.steps
                    el = wls.timed_run(p, profile=False)[0]  # the step time itself WITHOUT them: at 1.6 ms per step ~50 event records are 2 % of it
                    sizes[label].setdefault(p, {})[api] = {'ms_per_step': round(el / args.steps * 1e3, 3), 'rays_per_sec': round((n + nsd) * args.steps / el, 1),
                                                           'kernel_ms_per_step': round(kern, 3)}
                wls.release()

    if rank != 0:
        vdist.barrier()                          # rank 0 finishes its report, then everybody leaves together
        torch.distributed.destroy_process_group()
        if verify_failed:
            raise SystemExit(3)
        return
    ms = elapsed / args.steps * 1e3
    value = rays * world * args.steps / elapsed
    n_sec = main_wl.n_sec
    cfg_name = {'fern': 'configs[3] (65,536 rays/iter ray-sharded over the GPUs)' if strong else 'configs[1]',
                'realestate': 'configs[2] geometry', 'dtu': 'configs[4] (131,072 rays/iter ray-sharded over the GPUs)' if strong else 'configs[4] geometry'}[args.workload]
    result = {
        'metric': 'train_rays_per_sec', 'value': round(value, 1), 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'init_steps': INIT_STEPS, 'ms_per_step': round(ms, 3), 'higher_is_better': True,
        'scaling': args.scaling, 'vs_baseline': None, 'dtype': ARITH[args.precision][0], 'data': 'synthetic',
        'config': {'workload': 'BASELINE %s: %s geometry (%s), %d views, %d rays/iter/GPU x (64+128) samples, coarse+fine 8x256 MLP, '
                               'V=%d secondary view(s), losses MSE+Visibility+VisibilityPrior, Adam; rays generated and random numbers drawn on device'
                               % (cfg_name, args.workload, 'NDC' if SCENES[args.workload][5] else 'non-NDC', SCENES[args.workload][6], rays, n_sec),
                   'rays_per_gpu': rays, 'global_rays': rays * world, 'parallelism': f'ray-sharded dp{world}',
                   'gemm_arithmetic': args.precision, 'arithmetic_note': ARITH[args.precision][3],
                   'optimizer': 'Adam(lr 5e-4, betas 0.9 / 0.999): ' + ('vipnerf_hip.optim.FlatAdam -- torch.optim.Adam\'s single-tensor update on one flat '
                                 'parameter / moment / gradient buffer, one launch (vipnerf_adam_step; bit-identical per parameter)' if args.optimizer == 'flat' else 'torch.optim.Adam(fused=True)')},
        'roofline': roofline_block(args.precision, prof, args.steps, rays, ms, sclk, n_sec=n_sec, workload=args.workload),
        'runs_ms_per_step': main_runs, 'runs_reported': 'median of %d timed regions of %d steps' % (len(main_runs), args.steps),
    }
    if collectives and world == 1:
        result['config']['collectives'] = 'forced (%s, world_size 1)' % torch.distributed.get_backend()
    for p, (el, pr, sc) in others.items():
        pms = el / args.steps * 1e3
        result['value_' + p] = round(rays * world * args.steps / el, 1)
        result['ms_per_step_' + p] = round(pms, 3)
        result['runs_ms_per_step_' + p] = other_runs[p]
        result['dtype_' + p] = ARITH[p][0]
        result['roofline_' + p] = roofline_block(p, pr, args.steps, rays, pms, sc, n_sec=n_sec, workload=args.workload)

    if render is not None:
        result['render_ms_per_frame'] = render[args.precision]['ms_per_frame']
        result['render'] = render
    if c4 is not None:
        result['configs4_dtu'] = c4
    if c2 is not None:
        result['configs2_realestate'] = c2
    if sizes is not None:
        result['sizes'] = sizes
    result['ranks_reduced'] = ranks_reduced
    result['step_api'] = args.step_api
    if collectives:
        result.update({k: v for k, v in main_extra.items() if v is not None})
        result['grad_allreduce_vs_whole_batch'] = float('%.3e' % verify['rel_l2'])
        result['ranks_param_identical'] = same['identical']
        result['sharding_check'] = dict(verify, tolerance=GRAD_SHARD_TOL, global_rows=args.verify_rows or (rays + main_wl.n_sparse) * world, params=same,
                                        passed=not verify_failed)
    import hashlib
    bi = vlib.build_info()
    result['build_info'] = bi
    result['build_info_sha16'] = hashlib.sha256(json.dumps(bi, sort_keys=True).encode()).hexdigest()[:16]
    try:
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        from pmc_traffic import csrc_sha16
        result['csrc_sha16'] = csrc_sha16()
    except Exception:
        pass

    if world == 1 and not args.no_cpu_baseline:
        from oracle import vipnerf_oracle as vo       # the checker, as the reported CPU baseline only
        result['cpu_baseline'] = cpu_baseline(vo, n_rays=args.cpu_rays)
    result['notes'] = NOTES
    # the full report: a file next to bench.py (and under gpurun_out/ when that exists) and stderr; stdout gets ONE compact line
    report_path = args.report_path or os.path.join(ROOT, 'gpurun_out' if os.path.isdir(os.path.join(ROOT, 'gpurun_out')) else '', 'bench_full.json')
    try:
        result['full_report'] = os.path.relpath(report_path, ROOT) if os.path.abspath(report_path).startswith(ROOT + os.sep) else report_path
        os.makedirs(os.path.dirname(os.path.abspath(report_path)), exist_ok=True)
        with open(report_path, 'w') as f:
            f.write(json.dumps(result, indent=1) + '\n')
    except OSError as e:
        result['full_report'] = 'not written: %s' % e
    # (stderr: indented, one member per line -- no line of it is a JSON document a line-oriented reader of merged output could mistake for THE line)
    sys.stderr.write('bench.py full report (also in %s):\n' % result['full_report'] + json.dumps(result, indent=1) + '\n')
    sys.stderr.flush()
    os.write(json_fd, (json.dumps(compact_line(result), separators=(',', ':')) + '\n').encode())
    if torch.distributed.is_initialized():
        vdist.barrier()
        torch.distributed.destroy_process_group()
    if verify_failed:
        sys.stderr.write('bench.py: the sharded step failed its self-check: %s\n' % json.dumps(result['sharding_check']))
        raise SystemExit(3)


if __name__ == '__main__':
    main()
