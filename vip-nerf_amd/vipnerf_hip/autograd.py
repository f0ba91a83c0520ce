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
NON_DIFF = {'z_vals', 'depth_var', 'depth_var_ndc'}   # no loss of the reference differentiates these


class RenderState:
    """Per-call, non-tensor state handed through RenderFunction.apply."""

    def __init__(self, cfg: L.Config, batch: Dict[str, torch.Tensor], rng: Optional[dict], z_fine=None):
        self.cfg, self.batch, self.rng, self.z_fine = cfg, batch, rng, z_fine
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
        pc = ops.pack_weights(list(params[:L.VIPNERF_N_PARAMS]), precision=cfg.precision)
        pf = ops.pack_weights(list(params[L.VIPNERF_N_PARAMS:]), precision=cfg.precision) if two else None
        need_bwd = state.grad_enabled and any(ctx.needs_input_grad)   # grad mode as seen by the caller of apply()
        cfg.save_acts = int(need_bwd)
        acts = None
        n = state.batch['rays_o'].shape[0]
        if need_bwd:
            ab, _ = ops.query_workspace(cfg, n)
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
        ctx.state, ctx.coarse, ctx.fine, ctx.acts, ctx.packed = state, coarse, fine, acts, (pc, pf)
        ctx.n_params = len(params)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        state, cfg = ctx.state, ctx.state.cfg
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
        sizes = [int(torch.Size(s).numel()) for s in ops.PARAM_SHAPES]
        levels = 2 if ctx.fine is not None else 1
        flat = torch.empty(levels * sum(sizes), dtype=torch.float32, device=dev)
        views, o = [], 0
        for _ in range(levels):
            for s, k in zip(ops.PARAM_SHAPES, sizes):
                views.append(flat[o:o + k].view(s))
                o += k
        gc = views[:len(sizes)]
        gf = views[len(sizes):] if ctx.fine is not None else None
        del views, flat
        ops.render_backward(cfg, state.batch, ctx.packed[0], ctx.packed[1], ctx.coarse, ctx.fine, gl['coarse'],
                            gl['fine'] if ctx.fine is not None else None, ctx.acts, bwd_ws, gc, gf)
        ctx.acts = None
        return (None, *gc, *(gf or []))


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
        out = []
        for li, sd in enumerate((sc, sf)):
            if sd is None:
                out += [None] * 5
                continue
            out.append(g[0 + li] * sd['rgb'])
            out.append(g[2 + li] * sd['visibility'])
            out.append(g[2 + li] * sd['raw_vis'])
            out.append(g[4 + li] * sd['vis2'] if 'vis2' in sd else None)
            out.append(g[6] * sd['depth'] if 'depth' in sd else None)
        out = [o if p else None for o, p in zip(out, ctx.present)]
        return (None, None, None, None, None, None, None, *out)
