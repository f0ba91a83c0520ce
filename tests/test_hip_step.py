"""vipnerf_train_step -- one training iteration as ONE library call (vipnerf_hip/step.py::FusedTrainStep) -- against the module-contract
path (VipNeRFHip.forward -> LossComputerHip.compute_losses -> TotalLoss.backward() -> FlatAdam.step(), the sequence of reference
src/Trainer01.py:61-107).  The call queues exactly the kernels of the five-call path, so over several iterations from the same weights:
every output, the eight loss values, the flat gradient and the parameters after each Adam step must be BIT-IDENTICAL; TotalLoss too
(both take it from the library in the same order)."""
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

OUT_KEYS = ('rgb_coarse', 'rgb_fine', 'acc_fine', 'depth_fine', 'depth_var_fine', 'weights_fine', 'z_vals_fine', 'alpha_coarse',
            'visibility_fine', 'vis2_fine')


def _build(scene, prec, sparse, dev, seed=11):
    import bench
    from models.ModelFactory import get_model
    from vipnerf_hip.optim import FlatAdam
    cfg = bench.model_configs(bench.SCENES[scene][5], sparse_depth=sparse)
    cfg['model']['hip_precision'] = prec
    torch.manual_seed(seed)
    model = get_model(cfg, None).to(dev).train()
    opt = FlatAdam(model.parameters(), lr=5e-4, betas=(0.9, 0.999))
    return cfg, model, opt


def _batches(scene, n, n_sparse, dev, count):
    import bench
    gen = bench.make_scene(scene, dev, seed=2, sparse_depth=n_sparse > 0)
    return [bench.make_batch(gen, n, 500 + i, n_sparse=n_sparse) for i in range(count)]


def _fresh(b, it):
    c = dict(b)
    c['common_data'] = {'poses': b['common_data']['poses']}
    c['iter_num'] = it
    return c


@pytest.mark.parametrize('scene,n,n_sparse,prec', [('fern', 1024, 0, 'fp32'), ('fern', 1024, 0, 'bf16'), ('realestate', 256, 256, 'fp32'),
                                                   ('realestate', 512, 512, 'bf16'), ('dtu', 300, 0, 'fp16'), ('fern', 96, 0, 'fp16x3'),
                                                   ('realestate', 2048, 2048, 'bf16')])        # BASELINE configs[2] at its full batch: 2048 nerf + 2048 sparse-depth rows
def test_one_call_step_is_the_five_call_step(scene, n, n_sparse, prec):
    from loss_functions.FusedLossesHip01 import VECTOR_ATTR
    from loss_functions.LossComputerHip01 import LossComputerHip
    from vipnerf_hip.step import FusedTrainStep, named_losses
    dev = torch.device('cuda:0')
    batches = _batches(scene, n, n_sparse, dev, 3)
    # A: the module contract
    cfg_a, model_a, opt_a = _build(scene, prec, n_sparse > 0, dev)
    lossc = LossComputerHip(cfg_a)
    # B: one call per iteration
    cfg_b, model_b, opt_b = _build(scene, prec, n_sparse > 0, dev)
    assert torch.equal(opt_a.flat, opt_b.flat)
    step = FusedTrainStep(model_b, cfg_b, opt_b)
    for i, b in enumerate(batches):
        it = 40000 + i
        ba, bb = _fresh(b, it), _fresh(b, it)
        for p in model_a.parameters():
            p.grad = None
        out_a = model_a(ba)
        losses_a = lossc.compute_losses(ba, out_a)
        vec_a = getattr(out_a['rgb_coarse'], VECTOR_ATTR).detach().clone()
        losses_a['TotalLoss'].backward()
        grad_a = torch.cat([p.grad.flatten() for p in model_a.parameters()]).clone()
        opt_a.step()
        res = step(bb)
        torch.cuda.synchronize()
        # outputs
        o = step.outputs
        names = {'vis2_fine': 'visibility2_fine'}
        for k in OUT_KEYS:
            ka = names.get(k, k)
            if ka not in out_a:
                continue
            assert torch.equal(o[k].reshape(-1), out_a[ka].detach().reshape(-1)), f'{scene} {prec} iter {i}: output {k} differs'
        assert torch.equal(res['loss_values'][:7], vec_a[:7]), f'loss values differ: {res["loss_values"].tolist()} vs {vec_a.tolist()}'
        # TotalLoss: since round 6 the module path takes it from the loss kernel in the one-call step's order and roundings (no torch.dot): the same bits
        assert torch.equal(losses_a['TotalLoss'].detach().reshape(1), res['TotalLoss']), (float(losses_a['TotalLoss']), float(res['TotalLoss'][0]))
        if 'rays_o2' not in bb and step._bufs and n_sparse == 0:          # the other cameras' centres, written by the step's first launch (k_coarse_z)
            from vipnerf_hip import ops
            B = next(reversed(step._bufs.values()))
            assert torch.equal(B.o2, ops.secondary_origins(bb['common_data']['poses'], bb['pixel_id'], int(bb['num_frames'])))
        grad_b = torch.cat([p.grad.flatten() for p in model_b.parameters()])
        assert float(grad_a.abs().max()) > 0
        assert torch.equal(grad_a, grad_b), f'{scene} {prec} iter {i}: gradients differ by {(grad_a - grad_b).abs().max().item():.3e}'
        assert torch.equal(opt_a.flat, opt_b.flat), f'{scene} {prec} iter {i}: parameters after the Adam step differ'
        assert torch.equal(opt_a.exp_avg_sq, opt_b.exp_avg_sq) and opt_a.t == opt_b.t
        nl = named_losses(res)
        for name in ('MSEHip01', 'VisibilityLossHip01'):
            assert abs(float(nl[name]) - float(losses_a[name]['loss_value'])) <= 1e-6 * max(1.0, abs(float(nl[name])))
        if n_sparse:
            assert float(nl['SparseDepthMSEHip01']) > 0


def test_one_call_step_with_external_reduce_equals_local_adam():
    """bucket_reduce (the multi-GPU hook: the call stops after the backward pass, the flat gradient is reduced, FlatAdam steps) with an
    identity reduction is the local-Adam call, bit for bit; with a x0.5 reduction the parameters move exactly as FlatAdam moves them."""
    from vipnerf_hip.step import FusedTrainStep
    dev = torch.device('cuda:0')
    batches = _batches('fern', 512, 0, dev, 2)
    cfg_a, model_a, opt_a = _build('fern', 'bf16', False, dev)
    cfg_b, model_b, opt_b = _build('fern', 'bf16', False, dev)
    seen = []
    step_a = FusedTrainStep(model_a, cfg_a, opt_a)
    step_b = FusedTrainStep(model_b, cfg_b, opt_b, bucket_reduce=lambda flat: seen.append(flat.data_ptr()))
    for i, b in enumerate(batches):
        step_a(_fresh(b, 40000 + i)); step_b(_fresh(b, 40000 + i))
        torch.cuda.synchronize()
        assert torch.equal(opt_a.flat, opt_b.flat) and torch.equal(opt_a.exp_avg, opt_b.exp_avg)
    assert len(seen) == 2 and seen[0] == seen[1] == model_b.coarse_model.pts_linears[0].weight.grad.data_ptr()


def test_one_call_step_rejects_what_it_cannot_do():
    from vipnerf_hip import _lib as L
    from vipnerf_hip.step import FusedTrainStep
    dev = torch.device('cuda:0')
    cfg, model, opt = _build('fern', 'fp32', False, dev)
    with pytest.raises(L.VipNerfHipError):
        FusedTrainStep(model, cfg, torch.optim.Adam(model.parameters()))
    step = FusedTrainStep(model, cfg, opt)
    b = _batches('fern', 64, 0, dev, 1)[0]
    model.eval()
    with pytest.raises(L.VipNerfHipError):
        step(_fresh(b, 1))
    model.train()
    bad = dict(cfg, losses=[{'name': 'SomeOtherLoss01', 'weight': 1}])
    with pytest.raises(L.VipNerfHipError):
        FusedTrainStep(model, bad, opt)


def test_build_info_names_every_switch_and_is_a_product_build():
    from vipnerf_hip import _lib as L
    info = L.build_info()
    assert info.startswith('libvipnerf_hip abi=%d arch=gfx950 VN_EXP=unset' % L.ABI_VERSION), info
    for k in ('VN_PT2_SPREAD=', 'VN_T16=', 'VN_WG16_HYBRID=', 'VN_DMA_MODE=', 'VN_ADAM_FMA_MASK='):
        assert k in info
    L.require_product_build('test')


def test_trainer_one_call_iterations_equal_the_module_contract():
    """TrainerHip01 steps through vipnerf_train_step by default (one sub-batch per iteration: the reference's shipped configs); with
    `one_call_step: False` it walks the module contract.  Same scene, schedule and seeds: bit-identical parameters after six iterations,
    the same logged losses."""
    import test_hip_dist as thd
    dev = torch.device('cuda:0')
    runs = {}
    for one_call in (True, False):
        tr = thd._trainer(dev, 0, 1, None)
        tr.configs['one_call_step'] = one_call
        if not one_call:
            tr.stepper = None
        assert (tr.stepper is not None) == one_call
        hist = tr.train()
        runs[one_call] = (torch.cat([p.detach().flatten() for p in tr.model.parameters()]).clone(), [h['MSEHip01'] for h in hist], [h['TotalLoss'] for h in hist])
    assert torch.equal(runs[True][0], runs[False][0]), 'parameters after six iterations differ between the two entry points'
    np.testing.assert_allclose(runs[True][1], runs[False][1], rtol=1e-6)
    np.testing.assert_allclose(runs[True][2], runs[False][2], rtol=1e-6)
