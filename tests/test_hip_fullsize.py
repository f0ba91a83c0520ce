"""BASELINE configs[3] / configs[4] at their per-GPU shard sizes (65,536 / 8 = 8192 fern rays in exact fp32; 131,072 / 8 = 16,384 DTU rays,
non-NDC, 3 views = 2 secondary views, in bf16 and fp16 mixed precision) -- sizes the CPU oracle cannot reach in seconds -- through
size-independent properties of the path:

  * every output, the losses and every parameter gradient finite;
  * determinism: the same call twice gives bit-identical outputs AND gradients (ordered reductions, no atomics);
  * ray independence: the rows of the whole batch equal, bit for bit, the same rows rendered as two half batches
    (Philox streams keyed by global ray index: rng_ray_base) -- the property ray sharding across GPUs rests on;
  * gradient additivity: every loss is a mean over rows, so grad(whole) = (grad(half 0) + grad(half 1)) / 2 up to summation order --
    what the all-reduce of the rank gradients computes.
The accuracy of the same kernels against the oracle is pinned at 1024 rays (tests/test_hip_round2.py) and against the reference's goldens.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'vip-nerf_amd'), os.path.join(ROOT, 'vip-nerf_amd', 'src'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def _step(model, cfg, batch, base, it=40000):
    from loss_functions.LossComputerHip01 import LossComputerHip
    b = dict(batch)
    b['common_data'] = {'poses': batch['common_data']['poses']}
    b['iter_num'], b['rng_ray_base'] = it, base
    model._last_iter = None                     # the same Philox offset on every call
    model.zero_grad(set_to_none=True)
    out = model(b)
    losses = LossComputerHip(cfg).compute_losses(b, out)
    losses['TotalLoss'].backward()
    grads = torch.cat([p.grad.flatten() for p in model.parameters()]).clone()
    keep = {k: out[k].detach().clone() for k in ('rgb_fine', 'rgb_coarse', 'acc_fine', 'depth_fine', 'visibility2_fine', 'weights_fine', 'z_vals_fine')}
    return keep, float(losses['TotalLoss']), grads


# (the last three: ray counts whose point counts are no multiple of the 256-point workgroups / of the weight-gradient chunk plans, halves with
# an odd number of 32-point blocks)
@pytest.mark.parametrize('scene,n,prec,add_tol', [('fern', 8192, 'fp32', 2e-5), ('dtu', 16384, 'bf16', 2e-4), ('dtu', 16384, 'fp16', 2e-4),
                                                  ('fern', 3002, 'bf16', 2e-4), ('dtu', 1006, 'fp16', 2e-4), ('fern', 1502, 'fp16x3h', 2e-5)])
def test_shard_size_properties(scene, n, prec, add_tol):
    import bench
    import test_hip_parity as tp
    from oracle import vipnerf_oracle as vo
    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    gen = bench.make_scene(scene, dev, seed=3)
    batch = bench.make_batch(gen, n, 77)
    model, cfg = tp.make_model(dev, bench.SCENES[scene][5], vo.init_params(31, scale=1.6))
    cfg['model']['hip_precision'] = prec
    model.train()

    out_a, loss_a, g_a = _step(model, cfg, batch, 0)
    assert np.isfinite(loss_a) and torch.isfinite(g_a).all() and all(torch.isfinite(v).all() for v in out_a.values())
    assert out_a['visibility2_fine'].shape == (n, bench.SCENES[scene][6] - 1)
    assert float(g_a.abs().max()) > 0

    out_b, loss_b, g_b = _step(model, cfg, batch, 0)                                   # determinism
    assert loss_a == loss_b and torch.equal(g_a, g_b) and all(torch.equal(out_a[k], out_b[k]) for k in out_a)

    half = n // 2                                                                      # ray independence + gradient additivity
    g_sum = torch.zeros_like(g_a)
    for h in range(2):
        sl = slice(h * half, (h + 1) * half)
        sb = {k: (v[sl] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == n else v) for k, v in batch.items()}
        sb['common_data'] = batch['common_data']
        sb.pop('row_class_counts', None)
        out_h, _, g_h = _step(model, cfg, sb, h * half)
        for k in out_a:
            assert torch.equal(out_h[k], out_a[k][sl]), f'{scene} {prec}: rows {sl} of {k} differ between the half batch and the whole batch'
        g_sum += g_h
    err = float((g_sum / 2 - g_a).norm() / g_a.norm())
    print(f'{scene} {n} rays {prec}: gradient additivity over two halves, relative L2 {err:.2e}')
    assert err <= add_tol


def test_flat_adam_trains_like_torch_adam():
    """vipnerf_hip.optim.FlatAdam on the device: three training steps (on-device batches and random numbers, the HIP backward's flat gradient
    buffer adopted without a copy) end with parameters bit-identical to torch.optim.Adam's single-tensor path on a second copy of the model;
    the parameters keep their identity / names / shapes, only their storage moves into one buffer (which pack_weights reads through)."""
    import bench
    import test_hip_parity as tp
    from oracle import vipnerf_oracle as vo
    from vipnerf_hip.optim import FlatAdam
    dev = torch.device('cuda:0')
    torch.manual_seed(11)
    gen = bench.make_scene('fern', dev, seed=5)
    batches = [bench.make_batch(gen, 512, 900 + i) for i in range(3)]
    finals = []
    for kind in ('torch', 'flat'):
        model, cfg = tp.make_model(dev, True, vo.init_params(41, scale=1.6))
        model.train()
        names = [k for k, _ in model.named_parameters()]
        opt = torch.optim.Adam(model.parameters(), lr=5e-4, betas=(0.9, 0.999), foreach=False, fused=False) if kind == 'torch' \
            else FlatAdam(model.parameters(), lr=5e-4, betas=(0.9, 0.999))
        from loss_functions.LossComputerHip01 import LossComputerHip
        lossc = LossComputerHip(cfg)
        for i, b0 in enumerate(batches):
            b = dict(b0)
            b['common_data'] = {'poses': b0['common_data']['poses']}
            b['iter_num'] = 40000 + i
            model._last_iter = None
            opt.zero_grad(set_to_none=True)
            out = model(b)
            lossc.compute_losses(b, out)['TotalLoss'].backward()
            opt.step()
        assert [k for k, _ in model.named_parameters()] == names
        finals.append(torch.cat([p.detach().flatten() for p in model.parameters()]).clone())
        if kind == 'flat':
            assert opt._flat_grad().data_ptr() == next(iter(model.parameters())).grad.data_ptr(), 'the adopted gradient buffer is used as it is'
    assert torch.equal(finals[0], finals[1])


def test_fused_adam_step_is_torch_adam():
    """vipnerf_adam_step (one launch for the whole flat update; FlatAdam's default on the GPU) takes, bit for bit, the steps of torch.optim.Adam's
    single-tensor path on the device: 1,191,946 values, gradients spanning 12 orders of magnitude incl. exact zeros, five steps with a changing
    learning rate.  Which of torch's three update expressions are contracted into an fma is a property of its compiled kernels: the library's
    default combination is the one that matches here (all eight are tried and the matching ones printed)."""
    from vipnerf_hip import ops
    dev = torch.device('cuda:0')
    torch.manual_seed(21)
    n = 1191946
    p0 = torch.randn(n, device=dev) * 0.1
    grads = []
    for it in range(5):
        g = torch.randn(n, device=dev) * 10.0 ** torch.randint(-9, 3, (n,), device=dev).float()
        g[torch.rand(n, device=dev) < 0.05] = 0.0
        grads.append(g)
    lrs = [5e-4 * 0.97 ** it for it in range(5)]
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=5e-4, betas=(0.9, 0.999), eps=1e-8, foreach=False, fused=False)
    states = []
    for it in range(5):
        ref.grad = grads[it].clone()
        opt.param_groups[0]['lr'] = lrs[it]
        opt.step()
        st = opt.state[ref]
        states.append((ref.detach().clone(), st['exp_avg'].clone(), st['exp_avg_sq'].clone()))
    matching = []
    for mask in list(range(8)) + [-1]:
        p, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        ok = True
        for it in range(5):
            ops.adam_step(p, m, v, grads[it], lrs[it], 0.9, 0.999, 1e-8, it + 1, fma_mask=mask)
            ok = ok and torch.equal(p, states[it][0]) and torch.equal(m, states[it][1]) and torch.equal(v, states[it][2])
        if ok:
            matching.append(mask)
    print('fma masks whose five steps equal torch.optim.Adam bit for bit:', matching)
    assert -1 in matching, f'the library default does not reproduce torch.optim.Adam; matching masks: {matching}'
