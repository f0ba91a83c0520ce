"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol include/vipnerf_hip.h
declares, argument validation works without a GPU, and the product path refuses to run without the library / on
CPU tensors (no silent fallback)."""
import ctypes as C
import math
import os
import re
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd'))
sys.path.insert(0, os.path.join(ROOT, 'vip-nerf_amd', 'src'))


def header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'vipnerf_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(vipnerf_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    from vipnerf_hip import _lib
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/vipnerf_hip.h but not exported'
    assert set(names) == set(_lib.SYMBOLS), 'ctypes binding and header disagree'
    assert lib.vipnerf_abi_version() == 6          # 6: + vipnerf_losses_forward_w, vipnerf_scale_segments_w, vipnerf_pack_weights2_c (append-only; no struct changed)
    image = 4 * (36 * 16384 + 34 * 16384 + 7424)      # ONE image per precision (ABI 5): 36 forward + 34 data-gradient stages of 64 KiB + the resident block
    assert lib.vipnerf_packed_weights_bytes() == lib.vipnerf_packed_weights_bytes_p(0) == image
    assert lib.vipnerf_packed_weights_bytes_p(1) == 0 and lib.vipnerf_packed_weights_bytes_p(2) == 0      # the retired split-bf16 arithmetics
    assert lib.vipnerf_packed_weights_bytes_p(3) > 0 and lib.vipnerf_packed_weights_bytes_p(6) > 0 and lib.vipnerf_packed_weights_bytes_p(7) == 0


def test_struct_sizes_match_header():
    from vipnerf_hip import _lib
    assert C.sizeof(_lib.Config) == 17 * 4
    assert C.sizeof(_lib.Rays) == 8 + 8 * 8
    assert C.sizeof(_lib.LevelOut) == 15 * 8
    assert C.sizeof(_lib.Outputs) == 2 * 15 * 8 + 16
    assert C.sizeof(_lib.LevelGrads) == 14 * 8
    assert C.sizeof(_lib.Rng) == 4 * 8 + 3 * 8 + 8
    assert C.sizeof(_lib.RayGen) == 4 * 4 + 4 * 4 + 2 * 8 + 8 + 2 * 8 + 4 * 8
    assert C.sizeof(_lib.RayBatch) == 16 * 8
    assert C.sizeof(_lib.ScaleSeg) == 2 * 8 + 8 + 2 * 4


def test_argument_validation_without_gpu():
    from vipnerf_hip import _lib, ops
    lib = _lib.load()
    cfg = ops.make_config(True, 64, 128, 1, False)
    a, b = C.c_size_t(), C.c_size_t()
    assert lib.vipnerf_query_workspace(C.byref(cfg), 4096, C.byref(a), C.byref(b)) == 0
    assert a.value == 0 and b.value > 0
    cfg.save_acts = 1
    assert lib.vipnerf_query_workspace(C.byref(cfg), 4096, C.byref(a), C.byref(b)) == 0
    assert a.value == 4 * 4096 * 256 * (9 * 256 + 8 * 8 + 2 * 128 + 64 + 2 * 32 + 2 * 4)      # (+ 2 x 4 words: the view hidden's ReLU bits per direction)
    assert lib.vipnerf_query_workspace(C.byref(ops.make_config(True, 60, 127, 1, False)), 16, C.byref(a), C.byref(b)) == 0      # any sample counts (round 6), as in the reference
    bad = ops.make_config(True, 200, 128, 1, False)                         # > 256 samples per ray on the fine level
    assert lib.vipnerf_query_workspace(C.byref(bad), 16, C.byref(a), C.byref(b)) == -2
    buf = C.create_string_buffer(256)
    lib.vipnerf_last_error(buf, 256)
    assert b'n_coarse' in buf.value
    toy = ops.make_config(False, 64, 0, 1, False, topology=(4, 64, 10, 4))          # generic kernels: scratch even in eval
    assert lib.vipnerf_query_workspace(C.byref(toy), 1024, C.byref(a), C.byref(b)) == 0
    assert a.value >= 4 * 1024 * 64 * (63 + 5 * 64 + 2 * (27 + 32 + 4)) and b.value > 0
    assert lib.vipnerf_packed_weights_bytes_c(C.byref(toy)) == 4 * (64 * 63 + 64 + 3 * (64 * 64 + 64) + 32 * 91 + 32 + 64 + 1 + 64 * 64 + 64 + 128 + 4)
    assert lib.vipnerf_packed_weights_bytes_c(C.byref(cfg)) == lib.vipnerf_packed_weights_bytes_p(0)
    toy.precision = 3
    assert lib.vipnerf_query_workspace(C.byref(toy), 16, C.byref(a), C.byref(b)) == -2
    lib.vipnerf_last_error(buf, 256)
    assert b'fp32 only' in buf.value
    with pytest.raises(_lib.VipNerfHipError):
        _lib.check(lib.vipnerf_pack_weights(None, None, None), 'pack')


def test_product_path_has_no_cpu_fallback():
    from vipnerf_hip import _lib
    from models.ModelFactory import get_model
    mlp = {'num_samples': 64, 'netdepth': 8, 'netwidth': 256, 'points_positional_encoding_degree': 10,
           'views_positional_encoding_degree': 4, 'use_view_dirs': True, 'view_dependent_rgb': True,
           'predict_visibility': True}
    cfg = {'data_loader': {'ndc': False}, 'model': {'name': 'VipNeRFHip01', 'coarse_mlp': dict(mlp), 'fine_mlp': dict(mlp, num_samples=128),
                                                      'perturb': True, 'raw_noise_std': 1.0, 'lindisp': False, 'white_bkgd': False}}
    m = get_model(cfg, None)
    names = [k for k, _ in m.named_parameters()]
    assert names[0] == 'coarse_model.pts_linears.0.weight' and len(names) == 48
    assert sum(p.numel() for p in m.parameters()) == 1191946
    z = torch.zeros(4, 3)
    with pytest.raises(_lib.VipNerfHipError):
        m({'rays_o': z, 'rays_d': z, 'view_dirs': z, 'near': torch.zeros(4, 1), 'far': torch.ones(4, 1)})
    # other topologies are served by the generic kernels (BASELINE configs[0]: 4 x 64 coarse-only) ...
    small = get_model({'data_loader': {'ndc': False}, 'model': {'name': 'VipNeRFHip01', 'coarse_mlp': dict(mlp, netwidth=64, netdepth=4)}}, None)
    assert [k for k, _ in small.named_parameters()][-1] == 'coarse_model.views_output_linear.bias' and len(list(small.parameters())) == 16
    assert small.coarse_model.pts_linears[3].weight.shape == (64, 64) and small.fine_model is None
    six = get_model({'data_loader': {'ndc': False}, 'model': {'name': 'VipNeRFHip01', 'coarse_mlp': dict(mlp, netwidth=128, netdepth=6)}}, None)
    assert six.coarse_model.pts_linears[5].weight.shape == (128, 128 + 63)          # gamma(x) re-enters after layer 4
    # ... but not in a split arithmetic, and not what no config uses
    with pytest.raises(_lib.VipNerfHipError):
        get_model({'data_loader': {'ndc': False}, 'model': {'name': 'VipNeRFHip01', 'hip_precision': 'fp16x3',
                                                             'coarse_mlp': dict(mlp, netwidth=64, netdepth=4)}}, None)
    plain = get_model({'data_loader': {'ndc': False}, 'model': {'name': 'VipNeRFHip01', 'coarse_mlp': dict(mlp, use_view_dirs=False, view_dependent_rgb=False,
                                                                                                         predict_visibility=False)}}, None)
    assert len(list(plain.parameters())) == 18                  # 8 trunk layers + the 4-row trunk head: use_view_dirs = False needs no more
    for bad in (dict(mlp, netwidth=100), dict(mlp, netdepth=9), dict(mlp, use_view_dirs=False), dict(mlp, use_view_dirs=False, predict_visibility=False)):
        with pytest.raises(_lib.VipNerfHipError):
            get_model({'data_loader': {'ndc': False}, 'model': {'name': 'VipNeRFHip01', 'coarse_mlp': bad}}, None)
    # the head variants (view_dependent_rgb / predict_visibility = False, VipNeRF01.py:467-491): the reference's parameter set (the oracle's
    # shapes are pinned to the reference by goldens F5 toy_rgbtrunk / dtu_novis / fern_plain), generic kernels, fp32 only
    from oracle import vipnerf_oracle as vo
    from vipnerf_hip import ops
    for vd, pv in ((False, True), (True, False), (False, False)):
        m = get_model({'data_loader': {'ndc': False}, 'model': {'name': 'VipNeRFHip01', 'coarse_mlp': dict(mlp, view_dependent_rgb=vd, predict_visibility=pv)}}, None)
        want = [('coarse_model.' + k, s) for k, s in vo.mlp_param_shapes(view_dep_rgb=vd, predict_vis=pv)]
        assert [(k, tuple(p.shape)) for k, p in m.named_parameters()] == want
        topo = m.topology
        assert topo[4] == (0 if vd else ops.HEAD_RGB_TRUNK) | (0 if pv else ops.HEAD_NO_VISIBILITY)
        assert ops.param_order(topo) == [k for k, _ in vo.mlp_param_shapes(view_dep_rgb=vd, predict_vis=pv)]
        assert ops.param_shapes(topo) == [s for _, s in vo.mlp_param_shapes(view_dep_rgb=vd, predict_vis=pv)]
        assert len(ops.param_slots(topo)) == len(want) and m.predict_visibility == pv
        with pytest.raises(_lib.VipNerfHipError):
            get_model({'data_loader': {'ndc': False}, 'model': {'name': 'VipNeRFHip01', 'hip_precision': 'bf16',
                                                                 'coarse_mlp': dict(mlp, view_dependent_rgb=vd, predict_visibility=pv)}}, None)


def test_oracle_is_not_imported_by_the_product():
    import subprocess
    out = subprocess.run(['grep', '-rl', 'oracle', os.path.join(ROOT, 'vip-nerf_amd', 'vipnerf_hip'),
                          os.path.join(ROOT, 'vip-nerf_amd', 'src'), os.path.join(ROOT, 'vip-nerf_amd', 'csrc')],
                         capture_output=True, text=True).stdout.split()
    out = [f for f in out if not f.endswith('.pyc')]
    offenders = []
    for f in out:
        for line in open(f, errors='ignore'):
            if re.search(r'^\s*(from|import)\s+oracle', line):
                offenders.append(f)
    assert not offenders, offenders


def test_checkpoint_roundtrip_with_reference_key_layout(tmp_path):
    """f-4: state-dict keys / shapes / order equal the reference's DataParallel-wrapped VipNeRF (manifest captured
    from the real reference by importing it), and the save/load shim round-trips through the reference's file format."""
    import json
    import CheckpointHip01 as ck
    from models.ModelFactory import get_model
    man = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'f8_reference_state_dict_manifest.json')))
    mlp = {'num_samples': 64, 'netdepth': 8, 'netwidth': 256, 'points_positional_encoding_degree': 10,
           'views_positional_encoding_degree': 4, 'use_view_dirs': True, 'view_dependent_rgb': True, 'predict_visibility': True}
    cfg = {'data_loader': {'ndc': True}, 'model': {'name': 'VipNeRFHip01', 'coarse_mlp': dict(mlp), 'fine_mlp': dict(mlp, num_samples=128)}}
    torch.manual_seed(3)
    m = get_model(cfg, None)
    sd = ck.add_prefix(m.state_dict())
    assert list(sd.keys()) == man['keys']
    assert all(list(v.shape) == man['shapes'][k] for k, v in sd.items())
    opt = torch.optim.Adam(m.parameters(), lr=5e-4)
    path = ck.save_model(m, opt, 50000, tmp_path)
    assert path.name == 'Model_Iter050000.tar' and (tmp_path / 'saved_models' / 'Model_Latest.tar').is_symlink()
    raw = torch.load(str(path), weights_only=False)
    assert set(raw.keys()) == {'iteration_num', 'model_state_dict', 'optimizer_state_dict'}
    assert all(k.startswith('module.') for k in raw['model_state_dict'])
    torch.manual_seed(4)
    m2 = get_model(cfg, None)
    it = ck.load_model(torch.nn.DataParallel(m2), tmp_path / 'saved_models' / 'Model_Latest.tar')
    assert it == 50000
    for (k1, p1), (k2, p2) in zip(m.named_parameters(), m2.named_parameters()):
        assert k1 == k2 and torch.equal(p1, p2)


def test_common_utils_contract_on_cpu():
    """get_device / move_to_device (reference src/utils/CommonUtils01.py:15-42): without a GPU the answer is the CPU
    device, like the reference's; the HIP model then refuses CPU tensors (test_product_path_has_no_cpu_fallback)."""
    from utils.CommonUtilsHip01 import get_device, move_to_device
    assert get_device(None) == torch.device('cpu') and get_device('') == torch.device('cpu')
    if not torch.cuda.is_available():
        assert get_device([0]) == torch.device('cpu')
    d = move_to_device({'a': torch.zeros(2), 'b': [torch.ones(1), 'x'], 'c': 5}, torch.device('cpu'))
    assert d['a'].device.type == 'cpu' and d['b'][1] == 'x' and d['c'] == 5


def test_batch_index_schedule_equals_the_reference():
    """f-1 host side: BatchIndexScheduler against the index schedule recorded from the reference's DataPreprocessor
    (golden F6b: shuffles on numpy's global generator, pre-crop incl. the reference's discarded re-generation at
    precrop_iterations, short last batch of an epoch, epoch reshuffle of both index arrays, sparse-depth rows appended)."""
    import numpy as np
    from data_preprocessors.RayGeneratorHip01 import BatchIndexScheduler
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'f6b_batches.npz'))
    np.random.seed(int(g['numpy_seed']))
    s = BatchIndexScheduler(int(g['n']), int(g['h']), int(g['w']), int(g['num_rays']), float(g['precrop_fraction']),
                            int(g['precrop_iterations']), g['sparse_depths'], int(g['num_rays_sparse']))
    assert np.array_equal(s.indices, g['indices0']) and np.array_equal(s.indices_sparse, g['indices_sd0'])
    sizes = set()
    for it in range(int(g['iters'])):
        idx, sp = s.next(it)
        assert np.array_equal(idx, g[f'it{it}_indices']), it
        assert np.array_equal(sp, g[f'it{it}_indices_mask_sparse_depth']) and np.array_equal(~sp, g[f'it{it}_indices_mask_nerf'])
        sizes.add(idx.shape[0])
    assert len(sizes) > 1, 'the fixture must contain a short end-of-epoch batch'


def test_philox_restatement_reproduces_random123_vectors():
    import numpy as np
    from oracle import philox_oracle as po
    for c, k, want in po.KAT:
        assert tuple(int(x) for x in po.philox4x32_10(np.array([c], np.uint32), np.array([k], np.uint32))[0]) == want


def test_flat_adam_is_torch_adam():
    """vipnerf_hip.optim.FlatAdam (one flat parameter / moment / gradient buffer, six elementwise kernels per step) takes, bit for bit, the
    steps of torch.optim.Adam's single-tensor path -- the reference's optimizer (Trainer01.py:505-515) -- with gradients that are views of
    one buffer (the HIP backward's layout) as well as with separate gradient tensors.  A parameter WITHOUT a gradient is where the two differ by
    design (torch skips it, one flat update cannot): FlatAdam raises instead of diverging silently."""
    import torch
    from vipnerf_hip.optim import FlatAdam
    torch.manual_seed(0)
    shapes = [(256, 63), (256,), (256, 319), (4, 128), (1,)]
    ref = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone()) for p in ref]
    o_ref = torch.optim.Adam(ref, lr=5e-4, betas=(0.9, 0.999), foreach=False, fused=False)
    o_mine = FlatAdam(mine, lr=5e-4, betas=(0.9, 0.999))
    n = sum(p.numel() for p in ref)
    for it in range(5):
        flat = torch.randn(n) * 10 ** (-it)                 # gradients of very different sizes from step to step
        off = 0
        for k, (a, b) in enumerate(zip(ref, mine)):
            g = flat[off:off + a.numel()].view(a.shape)
            off += a.numel()
            a.grad = g.clone()
            b.grad = g if it % 2 == 0 else g.clone()         # even steps: views of ONE buffer (adopted, no copy); odd steps: separate tensors
        if it == 3:                                          # a missing gradient: refused, by name
            keep, mine[1].grad = mine[1].grad, None
            with pytest.raises(RuntimeError, match=r'parameter\(s\) \[1\] have no gradient'):
                o_mine.step()
            mine[1].grad = keep
        for grp in o_ref.param_groups:
            grp['lr'] = 5e-4 * 0.9 ** it
        o_mine.param_groups[0]['lr'] = 5e-4 * 0.9 ** it
        o_ref.step()
        o_mine.step()
        for a, b in zip(ref, mine):
            assert torch.equal(a.detach(), b.detach()), f'step {it}'
    assert all(p.data_ptr() == o_mine.flat[sum(q.numel() for q in mine[:i]):].data_ptr() for i, p in enumerate(mine))


def test_flat_adam_refuses_checkpoints_it_cannot_represent():
    """a torch.optim.Adam state in which the parameters sit at different step counts, or some have no state at all, cannot be
    one flat update with one step count: load_state_dict says which parameters are the problem instead of raising KeyError."""
    import torch
    from vipnerf_hip.optim import FlatAdam
    ps = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(2))]
    ot = torch.optim.Adam(ps, lr=1e-3, foreach=False, fused=False)
    for it in range(3):
        ps[0].grad, ps[1].grad = torch.randn(4, 3), torch.randn(5)
        ps[2].grad = torch.randn(2) if it < 2 else None     # the last step skips parameter 2
        ot.step()
    sd = ot.state_dict()
    mine = FlatAdam([torch.nn.Parameter(p.detach().clone()) for p in ps], lr=1e-3)
    with pytest.raises(ValueError, match=r'different step counts \(most at 3; parameter\(s\) \[2\] at \[2\]\)'):
        mine.load_state_dict(sd)
    sd2 = {'state': {k: v for k, v in sd['state'].items() if k != 1}, 'param_groups': sd['param_groups']}
    with pytest.raises(ValueError, match=r'parameter\(s\) \[1\] have no \(or partial\) optimizer state'):
        mine.load_state_dict(sd2)


def test_secondary_origins_rejects_malformed_poses():
    """the kernel strides by 16 floats per camera; anything but (nf, 4, 4) / (nf, 3, 4) poses is refused before any launch."""
    import torch
    from vipnerf_hip import _lib, ops
    pid = torch.zeros(5, 3, dtype=torch.int32)
    for bad in (torch.zeros(3, 16), torch.zeros(3, 4, 3), torch.zeros(4, 4)):
        with pytest.raises(_lib.VipNerfHipError, match='poses must be'):
            ops.secondary_origins(bad, pid, 3)


def test_flat_adam_state_dict_is_torch_adams():
    """FlatAdam.state_dict() / load_state_dict() use torch.optim.Adam's format: a run checkpointed with one optimizer resumes with the
    other and continues bit for bit (the `optimizer_state_dict` of the reference's checkpoints, Trainer01.py:352-381)."""
    import io
    import torch
    from vipnerf_hip.optim import FlatAdam
    torch.manual_seed(1)
    shapes = [(64, 63), (64,), (4, 32), (1,)]
    n = sum(math.prod(s) for s in shapes)
    grads = [torch.randn(n) * 0.1 for _ in range(6)]

    def run(params, opt, its):
        for it in its:
            off = 0
            for p in params:
                p.grad = grads[it][off:off + p.numel()].view(p.shape).clone()
                off += p.numel()
            opt.step()

    def through_file(sd):
        f = io.BytesIO()
        torch.save(sd, f)
        f.seek(0)
        return torch.load(f, weights_only=False)

    init = [torch.randn(s) for s in shapes]
    straight = [torch.nn.Parameter(t.clone()) for t in init]
    run(straight, torch.optim.Adam(straight, lr=5e-4, foreach=False, fused=False), range(6))

    a = [torch.nn.Parameter(t.clone()) for t in init]                     # torch -> FlatAdam -> torch
    oa = torch.optim.Adam(a, lr=5e-4, foreach=False, fused=False)
    run(a, oa, range(2))
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    ob = FlatAdam(b, lr=1.0)
    ob.load_state_dict(through_file(oa.state_dict()))
    assert ob.t == 2 and ob.param_groups[0]['lr'] == 5e-4
    run(b, ob, range(2, 4))
    c = [torch.nn.Parameter(p.detach().clone()) for p in b]
    oc = torch.optim.Adam(c, lr=1.0, foreach=False, fused=False)
    oc.load_state_dict(through_file(ob.state_dict()))
    run(c, oc, range(4, 6))
    for x, y in zip(straight, c):
        assert torch.equal(x.detach(), y.detach())

    fresh = FlatAdam([torch.nn.Parameter(t.clone()) for t in init])        # before the first step: empty state, like torch
    assert fresh.state_dict()['state'] == {}
    ob.load_state_dict(fresh.state_dict())
    assert ob.t == 0 and float(ob.exp_avg.abs().max()) == 0
    with pytest.raises(ValueError):
        ob.load_state_dict(torch.optim.Adam(a[:2]).state_dict())


def test_argument_checks_need_no_gpu():
    """Every entry point validates its arguments before it touches the device: the error paths return VIPNERF_E_* (and a message through
    vipnerf_last_error) on a machine without a GPU."""
    from vipnerf_hip import _lib
    lib = _lib.load()

    def last():
        buf = C.create_string_buffer(512)
        lib.vipnerf_last_error(buf, 512)
        return buf.value.decode()

    segs = (_lib.ScaleSeg * 17)()
    assert lib.vipnerf_scale_segments(17, segs, None, None) < 0 and 'n_segs' in last()
    assert lib.vipnerf_scale_segments(0, None, None, None) == 0
    assert lib.vipnerf_scale_segments(1, segs, None, None) < 0 and 'NULL' in last()
    # the weighted forms (TotalLoss without PyTorch arithmetic): host weights and a device total are required
    w8 = (C.c_float * 8)(*[1.0] * 8)
    assert lib.vipnerf_scale_segments_w(17, segs, None, w8, None) < 0 and 'n_segs' in last()
    assert lib.vipnerf_scale_segments_w(0, None, None, None, None) == 0
    assert lib.vipnerf_scale_segments_w(1, segs, None, w8, None) < 0 and 'NULL' in last()
    assert lib.vipnerf_adam_step(-1, None, None, None, None, 0.1, 0.999, 0.001, 1.0, 1e-8, -1e-3, -1, None) < 0
    assert lib.vipnerf_adam_step(0, None, None, None, None, 0.1, 0.999, 0.001, 1.0, 1e-8, -1e-3, -1, None) == 0
    assert lib.vipnerf_adam_step(8, None, None, None, None, 0.1, 0.999, 0.001, 1.0, 1e-8, -1e-3, -1, None) < 0 and 'NULL' in last()
    from vipnerf_hip import ops
    bad = ops.make_config(True, 64, 128, 1, True, topology=(8, 256, 10, 4, 7))
    a, b = C.c_size_t(0), C.c_size_t(0)
    assert lib.vipnerf_query_workspace(C.byref(bad), 16, C.byref(a), C.byref(b)) < 0 and 'head_variant' in last()
    bad = ops.make_config(True, 1, 0, 1, True)                     # (any sample counts since round 6; a single coarse sample is not a ray)
    assert lib.vipnerf_query_workspace(C.byref(bad), 16, C.byref(a), C.byref(b)) < 0 and 'n_coarse' in last()
    bad = ops.make_config(True, 2, 14, 1, True)                    # importance sampling needs three coarse samples (sample_pdf's bins)
    assert lib.vipnerf_query_workspace(C.byref(bad), 16, C.byref(a), C.byref(b)) < 0 and 'n_fine' in last()
