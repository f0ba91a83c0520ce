"""torch.autograd glue around the C ABI: one Function for the whole render (forward = vipnerf_render_forward,
backward = vipnerf_render_backward) and one for the fused losses.  PyTorch provides the tape, the optimizer and
device memory; no arithmetic of the path happens here."""
from typing import Dict, List, Optional

import torch

from . import _lib as L
from . import ops

# outputs of one level, in the order RenderFunction returns them
LEVEL_KEYS = ['z_vals', 'raw_sigma', 'raw_rgb', 'raw_vis', 'raw_vis2', 'alpha', 'visibility', 'weights', 'rgb',
              'acc', 'depth', 'depth_var', 'depth_ndc', 'depth_var_ndc', 'vis2']
NON_DIFF = {'z_vals'}   # sample depths carry no gradient in the reference either (VipNeRF01.py:213 detaches the samples)


class RenderState:
    """Per-call, non-tensor state handed through RenderFunction.apply."""

    def __init__(self, cfg: L.Config, batch: Dict[str, torch.Tensor], rng: Optional[dict], z_fine=None,
                 max_workspace_bytes: Optional[int] = None):
        self.cfg, self.batch, self.rng, self.z_fine = cfg, batch, rng, z_fine
        self.max_workspace_bytes = max_workspace_bytes   # None = what the device has free right now
        self.recompute_chunk = 0                         # > 0: backward re-renders the batch in ray chunks of this size
        self.grad_enabled = torch.is_grad_enabled()     # captured outside Function.forward, where it is always off
        self.keys: List[str] = []        # '<key>_<level>' of every returned tensor, in order
        self.extras: dict = {}


class RenderFunction(torch.autograd.Function):
    """apply(state, *params): params = 24 coarse tensors (+ 24 fine tensors if cfg.n_fine > 0), in ops.PARAM_ORDER.
    Returns a flat tuple of tensors; state.keys names them."""

    @staticmethod
    def forward(ctx, state: RenderState, *params):
        cfg = state.cfg
        two = cfg.n_fine > 0
        n_mlp = len(ops.param_order(ops.topology_of(cfg)))        # tensors per MLP (24 for the default topology)
        if two:                                                   # both MLPs' images by one launch
            pc, pf = ops.pack_weights2(list(params[:n_mlp]), list(params[n_mlp:]), cfg)
        else:
            pc, pf = ops.pack_weights(list(params[:n_mlp]), cfg=cfg), None
        need_bwd = state.grad_enabled and any(ctx.needs_input_grad)   # grad mode as seen by the caller of apply()
        cfg.save_acts = int(need_bwd)
        acts = None
        n = state.batch['rays_o'].shape[0]
        if need_bwd:
            ab, bb = ops.query_workspace(cfg, n)
            state.recompute_chunk = _recompute_chunk(state, n, ab, bb, params[0].device)
            if not state.recompute_chunk:
                try:
                    acts = torch.empty(ab // 4, dtype=torch.float32, device=params[0].device)
                except torch.OutOfMemoryError:
                    # the shortcut in _recompute_chunk (a workspace under a tenth of the device's memory is taken without asking the driver)
                    # met a device that other tensors / processes have filled: ask now and fall back to the re-rendering backward
                    torch.cuda.empty_cache()
                    state.recompute_chunk = _recompute_chunk(state, n, ab, bb, params[0].device, ask_driver=True) or min(n, 256)
            if state.recompute_chunk:
                cfg.save_acts = 0        # the activation store would not fit: backward re-renders chunk by chunk
        if acts is None:
            ab, _ = ops.query_workspace(cfg, n)     # non-zero without save_acts for the generic-topology kernels only: their
            if ab:                                   # layers run through HBM
                acts = torch.empty(ab // 4, dtype=torch.float32, device=params[0].device)
        coarse, fine, extras = ops.render_forward(cfg, state.batch, state.rng, pc, pf, acts, state.z_fine)
        state.extras = {k: v for k, v in extras.items() if not k.startswith('_')}
        outs, keys, nondiff = [], [], []
        for lv, d in (('coarse', coarse), ('fine', fine)):
            if d is None:
                continue
            for k in LEVEL_KEYS:
                if k in d:
                    outs.append(d[k])
                    keys.append(f'{k}_{lv}')
                    if k in NON_DIFF:
                        nondiff.append(d[k])
        state.keys = keys
        ctx.mark_non_differentiable(*nondiff)
        ctx.set_materialize_grads(False)          # outputs no loss touches arrive as None (= NULL = zero in the ABI)
        # what backward needs of the outputs is kept as DETACHED aliases: the returned tensors carry this node as grad_fn, so
        # holding them on ctx would be a reference cycle through C++ that only the cyclic GC frees (~50 MB of (N,S) outputs
        # per 4096-ray step lingering for a nondeterministic time)
        det = lambda d: None if d is None else {k: v.detach() for k, v in d.items()}
        ctx.state, ctx.coarse, ctx.fine, ctx.acts, ctx.packed = state, det(coarse), det(fine), acts, (pc, pf)
        ctx.n_params = len(params)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        state, cfg = ctx.state, ctx.state.cfg
        if state.recompute_chunk:
            return _chunked_backward(ctx, gouts)
        if ctx.acts is None:
            raise L.VipNerfHipError('backward called on a forward that did not save activations')
        gl = {'coarse': {}, 'fine': {}}
        for key, g in zip(state.keys, gouts):
            if g is None:
                continue
            k, lv = key.rsplit('_', 1)
            gl[lv][k] = g
        n = state.batch['rays_o'].shape[0]
        dev = ctx.coarse['rgb'].device
        _, bb = ops.query_workspace(cfg, n)
        bwd_ws = torch.empty(bb // 4, dtype=torch.float32, device=dev)
        # all parameter gradients of the call live in ONE buffer, in parameter order (coarse, then fine): with
        # .grad = None beforehand autograd adopts the views as they are (no per-tensor fill / add / copy), and
        # dist.FlatGradBucket reduces the buffer with a single collective
        shapes = ops.param_shapes(ops.topology_of(cfg))
        sizes = [int(torch.Size(s).numel()) for s in shapes]
        levels = 2 if ctx.fine is not None else 1
        flat = torch.empty(levels * sum(sizes), dtype=torch.float32, device=dev)
        views, o = [], 0
        for _ in range(levels):
            for s, k in zip(shapes, sizes):
                views.append(flat[o:o + k].view(s))
                o += k
        gc = views[:len(sizes)]
        gf = views[len(sizes):] if ctx.fine is not None else None
        del views, flat
        ops.render_backward(cfg, state.batch, ctx.packed[0], ctx.packed[1], ctx.coarse, ctx.fine, gl['coarse'],
                            gl['fine'] if ctx.fine is not None else None, ctx.acts, bwd_ws, gc, gf)
        ctx.acts = ctx.coarse = ctx.fine = ctx.packed = None
        return (None, *gc, *(gf or []))


MAX_RECOMPUTE_CHUNK = 8192


def _recompute_chunk(state: RenderState, n: int, acts_bytes: int, bwd_bytes: int, dev, ask_driver: bool = False) -> int:
    """0 if the training workspace of an n-ray call (activation store + backward scratch, ~ 5 MB per ray) fits the
    budget, else the ray-chunk size for a re-rendering backward.  The reference bounds memory with its `chunk` host loop
    only in eval (every chunk's autograd graph stays alive in training, VipNeRF01.py:47-72); here a call that does not fit
    keeps NO activations in the forward and its backward re-renders the rays chunk by chunk (one extra forward), with the
    very same random numbers: the Philox streams are keyed by global ray index (vipnerf_rng.ray_base / ray_ids)."""
    limit = state.max_workspace_bytes
    if limit is None:
        # a workspace under a tenth of the device's memory is taken without asking the driver (mem_get_info is a host <-> driver round
        # trip on every training forward); the allocator raises the usual out-of-memory error should even that not fit
        if not ask_driver and acts_bytes + bwd_bytes <= torch.cuda.get_device_properties(dev).total_memory // 10:
            return 0
        free, _ = torch.cuda.mem_get_info(dev)
        cached = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
        limit = int(0.7 * (free + cached))      # one allocation (activation store + scratch) must fit next to what lives already
    if acts_bytes + bwd_bytes <= limit or n <= 256:
        return 0
    per_ray = (acts_bytes + bwd_bytes) / n
    # chunks of at most 8192 rays (~42 GB): large enough to run the kernels at full efficiency, small enough that the caching
    # allocator keeps one such block from step to step (allocating and releasing 200 GB per step costs seconds)
    chunk = min(int(limit / per_ray), MAX_RECOMPUTE_CHUNK) // 256 * 256
    c = L.Config.from_buffer_copy(state.cfg)
    c.save_acts = 1
    while chunk >= 256 and sum(ops.query_workspace(c, chunk)) > limit:      # the scratch is not exactly linear in n
        chunk -= 256
    if chunk < 256:
        raise L.VipNerfHipError(f'training workspace: {acts_bytes + bwd_bytes} bytes for {n} rays ({per_ray / 1e6:.1f} MB per ray) and '
                                f'not even a 256-ray chunk fits the {limit} bytes available; lower configs["sub_batch_size"]')
    return 0 if chunk >= n else chunk


def _slice_rows(d: Optional[dict], n: int, s: int, e: int) -> Optional[dict]:
    if d is None:
        return None
    return {k: (v[s:e] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == n else v) for k, v in d.items()}


def _chunked_backward(ctx, gouts):
    """Backward of a forward that kept no activations: per ray chunk, re-render WITH the activation store (same weights,
    same random numbers, the forward's fine depths), run the fused backward, and add up the parameter gradients."""
    state, cfg = ctx.state, ctx.state.cfg
    n = state.batch['rays_o'].shape[0]
    dev = ctx.coarse['rgb'].device
    two = ctx.fine is not None
    gl = {'coarse': {}, 'fine': {}}
    for key, g in zip(state.keys, gouts):
        if g is not None:
            k, lv = key.rsplit('_', 1)
            gl[lv][k] = g
    shapes = ops.param_shapes(ops.topology_of(cfg))
    sizes = [int(torch.Size(s).numel()) for s in shapes]
    levels = 2 if two else 1

    def flat_views():
        flat = torch.empty(levels * sum(sizes), dtype=torch.float32, device=dev)
        views, o = [], 0
        for _ in range(levels):
            for s, k in zip(shapes, sizes):
                views.append(flat[o:o + k].view(s))
                o += k
        return flat, views

    total, views = flat_views()
    tmp, tviews = flat_views()
    base = int(state.rng.get('ray_base', 0)) if state.rng else 0
    chunk = state.recompute_chunk
    # ONE workspace for all chunks (activation store + backward scratch of the largest chunk), allocated after handing the
    # allocator's cached blocks back: tens of GB each, and a stale block of the wrong size would otherwise count twice
    cmax = L.Config.from_buffer_copy(cfg)
    cmax.save_acts = 1
    ab_max, bb_max = ops.query_workspace(cmax, min(chunk, n))
    try:
        ws = torch.empty((ab_max + bb_max) // 4, dtype=torch.float32, device=dev)
    except torch.OutOfMemoryError:
        torch.cuda.empty_cache()                 # cached blocks of other sizes (a smaller earlier batch's workspace, ...)
        ws = torch.empty((ab_max + bb_max) // 4, dtype=torch.float32, device=dev)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        c = L.Config.from_buffer_copy(cfg)
        c.save_acts = 1
        sub = _slice_rows(state.batch, n, s, e)
        rng = None
        if state.rng is not None:
            rng = {k: (v.reshape(n, -1)[s:e] if k in ('t_rand', 'u', 'noise_coarse', 'noise_fine', 'ray_ids') and v is not None
                       else v) for k, v in state.rng.items()}
            if rng.get('ray_ids') is not None:
                rng['ray_ids'] = rng['ray_ids'].reshape(-1)
            rng['ray_base'] = base + s
        ab, bb = ops.query_workspace(c, e - s)
        acts, bwd_ws = ws[:ab // 4], ws[ab_max // 4:ab_max // 4 + bb // 4]
        zf = ctx.fine['z_vals'][s:e] if two else None
        coarse, fine, _ = ops.render_forward(c, sub, rng, ctx.packed[0], ctx.packed[1], acts, zf)
        dst = views if s == 0 else tviews
        ops.render_backward(c, sub, ctx.packed[0], ctx.packed[1], coarse, fine, _slice_rows(gl['coarse'], n, s, e),
                            _slice_rows(gl['fine'], n, s, e) if two else None, acts, bwd_ws, dst[:len(sizes)],
                            dst[len(sizes):] if two else None)
        if s > 0:
            total.add_(tmp)
        del acts, bwd_ws, coarse, fine
    del ws
    ctx.acts = ctx.coarse = ctx.fine = ctx.packed = None
    return (None, *views)


class FusedLossFunction(torch.autograd.Function):
    """apply(cfg, n_rays, target_rgb, mask_nerf, prior, mask_sparse, sparse_depth,
             rgb_c, T_c, rawvis_c, vis2_c, depth_c, rgb_f, T_f, rawvis_f, vis2_f, depth_f) -> loss_values (8,)
    Tensors of an absent level / absent vis2 are None.  Values: [mse_c, mse_f, vis_c, vis_f, prior_c, prior_f, sd, 0]."""

    @staticmethod
    def forward(ctx, cfg, n_rays, target_rgb, mask_nerf, prior, mask_sparse, sparse_depth, *lv):
        def level(t):
            rgb, T, rv, v2, dep = t
            if rgb is None:
                return None
            d = {'rgb': ops.f32c(rgb), 'visibility': ops.f32c(T), 'raw_vis': ops.f32c(rv), 'depth': ops.f32c(dep)}
            if v2 is not None:
                d['vis2'] = ops.f32c(v2)
            return d
        coarse, fine = level(lv[0:5]), level(lv[5:10])
        vals, sc, sf = ops.losses_forward(cfg, n_rays, target_rgb, mask_nerf, prior, mask_sparse, sparse_depth, coarse, fine)
        ctx.seeds = (sc, sf)
        ctx.present = [t is not None for t in lv]
        return vals

    @staticmethod
    def backward(ctx, g):
        sc, sf = ctx.seeds
        out = [None] * 10
        todo = []                                    # (position in out, seed tensor, slot of loss_values whose upstream gradient scales it)
        for li, sd in enumerate((sc, sf)):
            if sd is None:
                continue
            for j, (key, slot) in enumerate((('rgb', 0 + li), ('visibility', 2 + li), ('raw_vis', 2 + li), ('vis2', 4 + li), ('depth', 6))):
                if key in sd and ctx.present[5 * li + j]:
                    todo.append((5 * li + j, sd[key], slot))
        scaled = ops.scale_segments([t for _, t, _ in todo], [s for _, _, s in todo], g)      # one launch for all of them
        for (pos, _, _), t in zip(todo, scaled):
            out[pos] = t
        return (None, None, None, None, None, None, None, *out)


class FusedLossTotalFunction(torch.autograd.Function):
    """apply(cfg, n_rays, weights8, target_rgb, mask_nerf, prior, mask_sparse, sparse_depth, <the ten level tensors of FusedLossFunction>)
        -> (TotalLoss 0-dim, loss_values (8,), named (4,))
    LossComputer.compute_losses' weighted total (reference src/loss_functions/LossComputer01.py:33-44) evaluated by the loss kernels themselves
    (vipnerf_losses_forward_w): TotalLoss = sum_k weights8[k] * loss_values[k] in vipnerf_train_step's order and roundings -- bit-identical to
    the one-call step's --, named = the four per-loss sums for logging.  Only TotalLoss carries a gradient; its backward is ONE launch
    (vipnerf_scale_segments_w: seeds x (upstream gradient x weight)).  No PyTorch arithmetic: the module-contract step's dot product, pair
    sums and their backward kernels (round 5: rocblas dot, a reduce and two elementwise kernels per step) are gone."""

    @staticmethod
    def forward(ctx, cfg, n_rays, weights8, target_rgb, mask_nerf, prior, mask_sparse, sparse_depth, *lv):
        def level(t):
            rgb, T, rv, v2, dep = t
            if rgb is None:
                return None
            d = {'rgb': ops.f32c(rgb), 'visibility': ops.f32c(T), 'raw_vis': ops.f32c(rv), 'depth': ops.f32c(dep)}
            if v2 is not None:
                d['vis2'] = ops.f32c(v2)
            return d
        coarse, fine = level(lv[0:5]), level(lv[5:10])
        vals, sc, sf, total, named = ops.losses_forward(cfg, n_rays, target_rgb, mask_nerf, prior, mask_sparse, sparse_depth, coarse, fine,
                                                        weights=weights8)
        ctx.seeds, ctx.weights = (sc, sf), [float(w) for w in weights8]
        ctx.present = [t is not None for t in lv]
        ctx.mark_non_differentiable(vals, named)
        ctx.set_materialize_grads(False)          # (the two value vectors carry no gradient: without this the engine fills a zero tensor for each, every step)
        return total.view(()), vals, named

    @staticmethod
    def backward(ctx, g_total, _g_vals, _g_named):
        sc, sf = ctx.seeds
        out = [None] * 10
        todo = []
        for li, sd in enumerate((sc, sf)):
            if sd is None:
                continue
            for j, (key, slot) in enumerate((('rgb', 0 + li), ('visibility', 2 + li), ('raw_vis', 2 + li), ('vis2', 4 + li), ('depth', 6))):
                if key in sd and ctx.present[5 * li + j]:
                    todo.append((5 * li + j, sd[key], slot))
        scaled = ops.scale_segments([t for _, t, _ in todo], [s for _, _, s in todo], g_total, weights=ctx.weights)
        for (pos, _, _), t in zip(todo, scaled):
            out[pos] = t
        return (None, None, None, None, None, None, None, None, *out)
