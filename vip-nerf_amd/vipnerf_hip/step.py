"""One training iteration as ONE library call (vipnerf_train_step): the launch-latency path for the batch sizes the reference's shipped
configs train at (1024 rays, or 2048 + 2048 with sparse depth: NerfLlffTrainerTester01.py:251,261,617).

The module contract (VipNeRFHip.forward -> LossComputerHip.compute_losses -> TotalLoss.backward() -> optimizer.step(), the sequence of
reference src/Trainer01.py:61-107) costs the host ~1.6 ms per iteration: five Python -> ctypes round trips, ~70 tensor allocations, two
autograd.Function nodes.  At 1024 rays the 16-bit kernels of a step take 1.6 ms, so the host IS the step.  FusedTrainStep keeps every
buffer of an iteration alive between calls (outputs, seeds, activation store, backward scratch, packed weights, the flat gradient), fills
one argument block and makes one call; the library queues exactly the kernels of the five-call path, so outputs, loss values, gradients
and the Adam update are bit-identical to it (tests/test_hip_step.py) -- only TotalLoss, which the module path sums with torch.dot, may
differ in its last bit.

    step = FusedTrainStep(model, configs, optimizer)        # model: VipNeRFHip (train mode), optimizer: vipnerf_hip.optim.FlatAdam
    res = step(batch)                                       # {'TotalLoss': (1,), 'loss_values': (8,), '<loss name>': 0-dim ...}
    step.outputs['rgb_fine'] ...                            # this iteration's outputs (views of persistent buffers: overwritten next call)

Multi-GPU: pass `bucket_reduce=fn` (e.g. lambda flat: all-reduce mean); the call then stops after the backward pass, fn(flat_grad) runs,
and the optimizer steps on the reduced buffer.
"""
import ctypes as C
from typing import Callable, Dict, Optional

import numpy as np
import torch

from . import _lib as L
from . import ops

LOSS_SLOTS = {'MSE': (0, 1), 'VisibilityLoss': (2, 3), 'VisibilityPriorLoss': (4, 5), 'SparseDepthMSE': (6, 7)}


TrainStepArgs = L.TrainStepArgs


def loss_weight(loss_configs: dict, iter_num: int):
    """LossComputer.get_loss_weight (reference src/loss_functions/LossComputer01.py:46-60)."""
    if 'weight' in loss_configs:
        return loss_configs['weight']
    if 'iter_weights' in loss_configs:
        for k in sorted((int(k) for k in loss_configs['iter_weights']), reverse=True):
            if iter_num >= k:
                return loss_configs['iter_weights'][str(k)]
    raise RuntimeError(f'loss_weight is None for {loss_configs.get("name")} at iter {iter_num}')


class _Buffers:
    """Everything one iteration of a given (ray count, V, sparse-depth) shape writes: allocated once, reused every call."""

    def __init__(self, cfg: L.Config, n: int, dev, n_params_flat: int, shapes, slots):
        two = cfg.n_fine > 0
        V = cfg.n_sec
        e = lambda *s, dt=torch.float32: torch.empty(*s, dtype=dt, device=dev)
        self.coarse = ops.alloc_level(n, cfg.n_coarse, V, cfg.ndc, dev)
        self.fine = ops.alloc_level(n, cfg.n_coarse + cfg.n_fine, V, cfg.ndc, dev) if two else None
        self.out = L.Outputs()
        self.out.coarse = ops._level_struct(self.coarse)
        if two:
            self.out.fine = ops._level_struct(self.fine)
            self.sample_inds, self.z_samples = e(n, cfg.n_fine, dt=torch.int32), e(n, cfg.n_fine)
            self.out.sample_inds, self.out.z_samples = self.sample_inds.data_ptr(), self.z_samples.data_ptr()
        self.loss_values, self.total, self.scratch = e(8), e(1), e(8 * n + 8)
        self.lout = L.LossOut()
        self.lout.loss_values, self.lout.scratch = self.loss_values.data_ptr(), self.scratch.data_ptr()
        self.seeds = []
        for lvl, S, st in ((self.coarse, cfg.n_coarse, self.lout.coarse), (self.fine, cfg.n_coarse + cfg.n_fine, self.lout.fine)):
            if lvl is None:
                self.seeds.append(None)
                continue
            d = {'rgb': e(n, 3), 'visibility': e(n, S), 'raw_vis': e(n, S)}
            if V > 0:
                d['vis2'] = e(n, V)
            if (lvl is self.fine) or not two:
                d['depth'] = e(n)                     # written by the loss kernel on the last level (zero without sparse-depth rows)
            for k, v in d.items():
                setattr(st, k, v.data_ptr())
            self.seeds.append(d)
        c = L.Config.from_buffer_copy(cfg)
        c.save_acts = 1
        ab, bb = ops.query_workspace(c, n)
        self.acts, self.bwd_ws = e(ab // 4 + 4), e(bb // 4 + 4)      # activation store, backward scratch
        self.acts_ptr, self.bwd_ptr = self.acts.data_ptr(), self.bwd_ws.data_ptr()
        nb = L.load().vipnerf_packed_weights_bytes_c(C.byref(cfg))
        self.packed_c = e(nb // 4)
        self.packed_f = e(nb // 4) if two else None
        self.o2 = e(n, V, 3) if V > 0 else None
        # the gradients: ONE flat buffer in parameter order (coarse, then fine) -- what FlatAdam and FlatGradBucket adopt
        self.flat_grad = e(n_params_flat)
        self.grad_views, o = [], 0
        self.gc, self.gf = L.MlpGrads(), L.MlpGrads()
        for lv in range(2 if two else 1):
            for s, shp in zip(slots, shapes):
                k = int(np.prod(shp))
                v = self.flat_grad[o:o + k].view(shp)
                self.grad_views.append(v)
                (self.gf if lv else self.gc).g[s] = v.data_ptr()
                o += k


class FusedTrainStep:
    MAX_BUFFER_SETS = 2          # persistent buffer sets kept alive (most recently used batch shapes)

    def __init__(self, model, configs: dict, optimizer, bucket_reduce: Optional[Callable] = None):
        from .optim import FlatAdam
        if not isinstance(optimizer, FlatAdam):
            raise L.VipNerfHipError('FusedTrainStep needs vipnerf_hip.optim.FlatAdam (flat parameter / moment buffers)')
        # the flat gradient is laid out in ABI order (coarse.ordered_params(), then fine) and handed to ONE Adam launch next to the
        # optimizer's flat parameter buffer: the two orders must be the same tensors in the same order
        ordered = list(model.coarse_model.ordered_params()) + (list(model.fine_model.ordered_params()) if model.fine_mlp_needed else [])
        if len(optimizer.params) != len(ordered) or any(a is not b for a, b in zip(optimizer.params, ordered)) \
                or optimizer.flat.numel() != sum(p.numel() for p in ordered):
            raise L.VipNerfHipError('FusedTrainStep: the optimizer\'s parameters are not coarse.ordered_params() + fine.ordered_params() '
                                    '(a frozen / extra parameter, or another order): use the module-contract path')
        if model.topology != ops.DEFAULT_TOPOLOGY:
            # (the generic kernels run through the same entry point; nothing here depends on the topology except the tests that pin it)
            pass
        self.model, self.configs, self.opt, self.bucket_reduce = model, configs, optimizer, bucket_reduce
        self.lib = L.load()
        self._bufs: Dict[tuple, _Buffers] = {}
        self.outputs: Dict[str, torch.Tensor] = {}
        self.losses = []
        for lc in configs['losses']:
            base = lc['name'][:-2]
            base = base[:-3] if base.endswith('Hip') else base        # 'MSEHip01' / 'MSE01' -> 'MSE'
            if base not in LOSS_SLOTS:
                raise L.VipNerfHipError(f'FusedTrainStep: loss {lc["name"]} is not one of the fused ray losses {sorted(LOSS_SLOTS)}')
            self.losses.append((lc['name'], LOSS_SLOTS[base], lc))
        self._params = None
        self.args = TrainStepArgs()

    # ------------------------------------------------------------------------------------------------------------------
    def _param_structs(self):
        pc = self.model.coarse_model.ordered_params()
        pf = self.model.fine_model.ordered_params() if self.model.fine_mlp_needed else None
        key = (pc[0].data_ptr(), pc[-1].data_ptr(), pf[0].data_ptr() if pf else 0)
        if self._params is None or self._params[0] != key:
            slots = ops.param_slots(self.model.topology)
            mc, mf = L.MlpParams(), L.MlpParams()
            for s, t in zip(slots, pc):
                mc.p[s] = ops._p(t, name='parameter')
            if pf:
                for s, t in zip(slots, pf):
                    mf.p[s] = ops._p(t, name='parameter')
            self._params = (key, mc, mf if pf else None)
        return self._params[1], self._params[2]

    def release(self):
        """Give the persistent buffers (tens of GB at 4096 rays in fp32) back to the allocator."""
        self._bufs.clear()
        self.outputs = {}

    def __call__(self, input_batch: dict) -> dict:
        model, m = self.model, self.configs['model']
        if not model.training:
            raise L.VipNerfHipError('FusedTrainStep: the model must be in training mode')
        if 'common_data' in input_batch:                        # the reference's in-place unpack (VipNeRF01.py:29-33)
            cd = input_batch['common_data']
            for key in cd:
                if isinstance(cd[key], torch.Tensor) and key == 'poses' and cd[key].dim() == 4:
                    cd[key] = cd[key][0]
        rays_o = input_batch['rays_o']
        if not rays_o.is_cuda:
            raise L.VipNerfHipError('FusedTrainStep runs on the GPU only; there is no CPU fallback')
        dev, n = rays_o.device, rays_o.shape[0]
        if n == 0:                                          # before anything is touched: Adam's step count, the buffers, the argument block
            raise L.VipNerfHipError('FusedTrainStep: empty batch')
        ndc = model.ndc
        two = model.fine_mlp_needed
        V = 0
        poses = pid = None
        if model.predict_visibility:
            if 'rays_o2' in input_batch:
                V = input_batch['rays_o2'].shape[1]
            else:
                V = int(input_batch['num_frames']) - 1
                poses, pid = ops.f32c(input_batch['common_data']['poses']), input_batch['pixel_id']
                if poses.dim() != 3 or tuple(poses.shape[1:]) != (4, 4) or poses.shape[0] < V + 1:
                    raise L.VipNerfHipError(f'FusedTrainStep: poses must be (num_frames, 4, 4), got {tuple(poses.shape)}')
                if pid.dtype not in (torch.int32, torch.int64):
                    pid = pid.to(torch.int64)
                pid = pid.contiguous()
        prec = ops.PRECISIONS[m.get('hip_precision', 'fp32')]
        cfg = ops.make_config(ndc, m['coarse_mlp']['num_samples'], m['fine_mlp']['num_samples'] if two else 0, V, train=True,
                              noise_std=float(m.get('raw_noise_std', 0.0)), lindisp=m.get('lindisp', False),
                              white_bkgd=m.get('white_bkgd', False), save_acts=True, perturb=bool(m.get('perturb', False)),
                              precision=prec, bf16_layout=ops.LAYOUTS[m.get('hip_bf16_layout', 'default')], topology=model.topology)
        z_inj = model.injected_z_fine
        cfg.given_z_fine = int(z_inj is not None and two)
        key = (n, V, prec, cfg.n_coarse, cfg.n_fine, bool(ndc), dev.index)
        B = self._bufs.pop(key, None)
        if B is None:
            # a trainer sends batches of varying size (the short last batch of an epoch, per-rank trimmed shards): only the MAX_BUFFER_SETS
            # most recently used shapes keep their buffers (~5 MB per ray in fp32), the oldest set goes back to the allocator first
            while len(self._bufs) >= self.MAX_BUFFER_SETS:
                self._bufs.pop(next(iter(self._bufs)))
            self.outputs = {}
            shapes, slots = ops.param_shapes(model.topology), ops.param_slots(model.topology)
            B = _Buffers(cfg, n, dev, self.opt.flat.numel(), shapes, slots)
        self._bufs[key] = B                                  # (re-inserted: dict order = recency)
        keep = []
        batch = {k: input_batch[k] for k in ('rays_o', 'rays_d', 'view_dirs') if k in input_batch}
        if 'view_dirs' not in batch:
            batch['view_dirs'] = batch['rays_d']
        if ndc:
            for k in ('rays_o_ndc', 'rays_d_ndc', 'near_ndc', 'far_ndc'):
                batch[k] = input_batch[k]
        else:
            batch['near'], batch['far'] = input_batch['near'], input_batch['far']
        if V > 0:
            batch['rays_o2'] = input_batch['rays_o2'] if poses is None else B.o2
        rays = ops._rays_struct(cfg, batch, keep)
        # random numbers: the module path's key (seed, iteration-derived offset, global ray index)
        seed, offset, ray_base = model._rng_key(input_batch, dev)
        rs = L.Rng()
        inj = model.injected_rng or {}
        for k in ('t_rand', 'u', 'noise_coarse', 'noise_fine'):
            if inj.get(k) is not None:
                tc = ops.f32c(inj[k]); keep.append(tc); setattr(rs, k, ops._p(tc, name=k))
        rs.seed, rs.offset, rs.ray_base = int(inj.get('seed', seed)) & (2 ** 64 - 1), int(inj.get('offset', offset)) & (2 ** 64 - 1), int(inj.get('ray_base', ray_base))
        ids = input_batch.get('rng_ray_ids')
        if ids is not None:
            ids = ids.to(device=dev, dtype=torch.int64).contiguous(); keep.append(ids); rs.ray_ids = ops._p(ids, torch.int64)
        if cfg.given_z_fine:
            B.fine['z_vals'].copy_(z_inj)
        # loss inputs and this iteration's weights
        li = L.LossIn()
        t = ops.f32c(input_batch['target_rgb']); keep.append(t); li.target_rgb = ops._p(t)
        mk = input_batch.get('indices_mask_nerf')
        if mk is not None:
            mk = ops.as_u8(mk)
            keep.append(mk); li.mask_nerf = ops._p(mk, torch.uint8)
        if V > 0:
            pr = input_batch.get('visibility_prior_masks', input_batch.get('visibility_prior_weights'))
            if pr is not None:
                pr = ops.f32c(pr); keep.append(pr); li.prior = ops._p(pr)
        msd = input_batch.get('indices_mask_sparse_depth')
        if msd is not None:
            msd = ops.as_u8(msd)
            sdv = ops.f32c(input_batch['sparse_depth_values'][:, 0]); keep += [msd, sdv]
            li.mask_sparse, li.sparse_depth = ops._p(msd, torch.uint8), ops._p(sdv)
        it = int(input_batch['iter_num'])
        w8 = [0.0] * 8
        present = {}
        for name, (a, b), lc in self.losses:
            if a == 4 and V == 0:
                continue                                  # VisibilityPriorLoss reports None without secondary views
            if a == 6 and msd is None:
                present[name] = None
                continue
            w = float(loss_weight(lc, it))
            w8[a] = w
            if two and b != 7:
                w8[b] = w
            present[name] = (a, b)
        A = self.args
        mc, mf = self._param_structs()
        A.cfg, A.rays, A.rng, A.loss_in = C.pointer(cfg), C.pointer(rays), C.pointer(rs), C.pointer(li)
        for k in range(8):
            A.loss_weights[k] = w8[k]
        A.params_coarse, A.params_fine = C.pointer(mc), (C.pointer(mf) if mf is not None else None)
        A.packed_coarse, A.packed_fine = B.packed_c.data_ptr(), (B.packed_f.data_ptr() if two else None)
        A.out, A.lout, A.total_loss = C.pointer(B.out), C.pointer(B.lout), B.total.data_ptr()
        A.acts, A.bwd_ws = B.acts_ptr, B.bwd_ptr
        A.grads_coarse, A.grads_fine = C.pointer(B.gc), (C.pointer(B.gf) if two else None)
        if poses is not None and V > 0:
            keep += [poses, pid]
            A.poses, A.pixel_id, A.pixel_id_is_int64 = ops._p(poses), ops._p(pid, pid.dtype), int(pid.dtype == torch.int64)
            A.n_frames, A.rays_o2_out = V + 1, B.o2.data_ptr()
        else:
            A.poses = A.pixel_id = A.rays_o2_out = None
        opt = self.opt
        local_adam = self.bucket_reduce is None
        if local_adam:
            grp = opt.param_groups[0]
            lr, (b1, b2), eps = grp['lr'], grp['betas'], grp['eps']
            t_next = opt.t + 1                           # committed only once the library call has succeeded (a raising step must not
            bc1, bc2 = 1 - b1 ** t_next, 1 - b2 ** t_next    # advance Adam's bias correction / the checkpointed step count)
            A.adam_n = opt.flat.numel()
            A.adam_param, A.adam_exp_avg, A.adam_exp_avg_sq, A.adam_grad = opt.flat.data_ptr(), opt.exp_avg.data_ptr(), opt.exp_avg_sq.data_ptr(), B.flat_grad.data_ptr()
            A.lerp_w, A.beta2, A.sq_w, A.inv_sqrt_bc2 = 1 - b1, b2, 1 - b2, float(np.float32(1.0 / bc2 ** 0.5))
            A.eps, A.neg_step, A.fma_mask = eps, -(lr / bc1), -1
        else:
            A.adam_n = 0
        with ops.on_device(rays_o, opt.flat, B.flat_grad) as d:
            L.check(self.lib.vipnerf_train_step(C.byref(A), ops._stream(d)), 'vipnerf_train_step')
        if local_adam:
            opt.t = t_next
        g0 = opt.params[0].grad
        if g0 is None or g0.data_ptr() != B.grad_views[0].data_ptr():
            for p, v in zip(opt.params, B.grad_views):       # the gradients, for whoever looks (a bucket, a test, gradient clipping)
                p.grad = v
        if not local_adam:
            self.bucket_reduce(B.flat_grad)
            opt.step()
        # this iteration's outputs under the module contract's names
        out = {}
        for lv, d in (('coarse', B.coarse), ('fine', B.fine)):
            if d is None:
                continue
            for k, v in d.items():
                out[f'{k}_{lv}'] = v
        self.outputs = out
        self._keep = keep
        # no per-loss tensor arithmetic here (each would be a launch): named_losses() derives the per-name values when somebody logs them
        return {'TotalLoss': B.total, 'loss_values': B.loss_values, 'loss_slots': present, 'two_levels': two}


_NAMED_IDX = {}


def named_losses(res: dict) -> Dict[str, torch.Tensor]:
    """{loss name: value} like LossComputer.compute_losses reports them (coarse + fine per loss), from a FusedTrainStep result.  Three
    launches whatever the number of losses (two index_selects into the step's loss vector and their sum: fresh tensors -- the step's own
    buffers are overwritten by the next call) and one copy of TotalLoss; the per-name values are views of the sum.  Selection by INDEX, not
    by a 0 / 1 matrix product: one diverged loss (NaN / Inf in its slot) shows up under its own name only, as in the reference's
    LossComputer, instead of turning every logged name into NaN through 0 * NaN."""
    lv = res['loss_values']
    names = list(res['loss_slots'])
    key = (tuple((n, res['loss_slots'][n]) for n in names), bool(res['two_levels']), lv.device)
    idx = _NAMED_IDX.get(key)
    if idx is None:
        i0, i1 = [], []
        for n in names:
            slots = res['loss_slots'][n]
            if slots is None:
                i0.append(7); i1.append(8)               # a loss that reported nothing: slot 7 (written 0 by the loss kernel); 8 = the appended zero
            else:
                i0.append(slots[0])
                i1.append(slots[1] if (res['two_levels'] and slots[1] != 7) else 8)
        idx = _NAMED_IDX[key] = (torch.tensor(i0 or [8], dtype=torch.long, device=lv.device), torch.tensor(i1 or [8], dtype=torch.long, device=lv.device),
                                 torch.zeros(1, device=lv.device))
    ext = torch.cat([lv, idx[2]])                        # [8 loss slots, 0.0]
    v = ext.index_select(0, idx[0]) + ext.index_select(0, idx[1])
    out = {n: v[i] for i, n in enumerate(names)}
    out['TotalLoss'] = res['TotalLoss'].clone()[0]
    return out
