"""Drop-in model for the reference's name-based factory: set configs['model']['name'] = 'VipNeRFHip01' and
`models.ModelFactory.get_model` (reference src/models/ModelFactory.py:10-22) instantiates `VipNeRFHip` with the
same `(configs, model_configs)` constructor, `forward(input_batch, retraw, sec_views_vis) -> dict` and
`render_rays(input_dict, retraw, sec_views_vis) -> dict` contract as `VipNeRF`
(reference src/models/VipNeRF01.py:11-171).  Parameters are ordinary nn.Linear modules with the reference's names and
shapes (`coarse_model.pts_linears.0.weight`, ...), created in the reference's construction order
(VipNeRF01.py:472-491), so optimizers, checkpoints (incl. the `module.` DataParallel prefix) and seeded
initialisation carry over.

All arithmetic of the path runs in libvipnerf_hip.so (MI355X / gfx950).  There is no CPU fallback: on a box
without the library or without a GPU tensor this module raises.
"""
import os
import sys
from pathlib import Path

import torch

try:
    import vipnerf_hip  # noqa: F401
except ImportError:
    for cand in (os.environ.get('VIPNERF_HIP_ROOT'), str(Path(__file__).resolve().parents[2])):
        if cand and cand not in sys.path:
            sys.path.insert(0, cand)
    import vipnerf_hip  # noqa: F401
from vipnerf_hip import _lib as L
from vipnerf_hip import ops
from vipnerf_hip.autograd import RenderFunction, RenderState


class MLPParams(torch.nn.Module):
    """Parameter container with MLP's layout (reference VipNeRF01.py:451-492: netdepth trunk layers of netwidth, gamma(x)
    re-injected after layer 4 -- `self.skips = [4]` --, netwidth/2-wide view layer, rgb + visibility head).  No forward of
    its own: the network is evaluated inside the HIP kernels -- the fused MFMA ones for the topology every shipped reference
    config uses (8 x 256, degrees 10 / 4, view-dependent rgb + visibility), the generic per-layer ones for any other (e.g.
    BASELINE configs[0]'s 4 x 64; `view_dependent_rgb` / `predict_visibility` = False, VipNeRF01.py:467-491)."""

    def __init__(self, configs, mlp_configs):
        super().__init__()
        if not mlp_configs.get('use_view_dirs') and (mlp_configs['view_dependent_rgb'] or mlp_configs['predict_visibility']):
            # MLP.forward reads input_batch['view_dirs'] for any view-dependent output (VipNeRF01.py:519-520), which run_network only
            # passes with use_view_dirs (:273-277): the reference raises KeyError on this combination
            raise L.VipNerfHipError("VipNeRFHip: use_view_dirs=False needs view_dependent_rgb=False and predict_visibility=False "
                                    "(the reference's MLP.forward fails on any other combination)")
        self.configs, self.mlp_configs = configs, mlp_configs
        D, W = int(mlp_configs['netdepth']), int(mlp_configs['netwidth'])
        lp, lv = int(mlp_configs['points_positional_encoding_degree']), int(mlp_configs['views_positional_encoding_degree'])
        if not (1 <= D <= 8 and 8 <= W <= 256 and W % 8 == 0 and 0 <= lp <= 16 and 0 <= lv <= 8):
            raise L.VipNerfHipError(f'VipNeRFHip: netdepth={D} netwidth={W} degrees {lp}/{lv} unsupported (depth 1..8, width 8..256 and '
                                    f'a multiple of 8, degrees <= 16 / 8)')
        heads = ops.head_variant(mlp_configs)
        self.topology = (D, W, lp, lv) + ((heads,) if heads else ())
        self.view_dep_rgb = bool(mlp_configs['view_dependent_rgb'])
        self.predict_visibility = bool(mlp_configs['predict_visibility'])
        n_trunk, n_view = ops.head_outputs(self.topology)
        d_pts, d_view = 3 + 6 * lp, 3 + 6 * lv
        self.pts_linears = torch.nn.ModuleList(
            [torch.nn.Linear(d_pts, W)] +
            [torch.nn.Linear(W, W) if i != 4 else torch.nn.Linear(W + d_pts, W) for i in range(D - 1)])
        if n_view:
            self.views_linears = torch.nn.ModuleList([torch.nn.Linear(d_view + W, W // 2)])
        self.pts_output_linear = torch.nn.Linear(W, n_trunk)
        if n_view:
            self.feature_linear = torch.nn.Linear(W, W)
            self.views_output_linear = torch.nn.Linear(W // 2, n_view)

    def ordered_params(self):
        """The parameter tensors in the ABI's order, by attribute path: on a torch.nn.DataParallel REPLICA the weights are plain (non-leaf)
        tensor attributes -- the broadcast copies autograd reduces back onto the master -- and named_parameters() is empty there."""
        replica = getattr(self, '_is_replica', False)
        cached = None if replica else self.__dict__.get('_ordered_cache')
        if cached is not None and all(owner._parameters.get(attr) is t for owner, attr, t in cached[0]):
            return cached[1]                         # the same nn.Parameter objects as last time (optimizers change their .data, not them)
        out, where = [], []
        for name in ops.param_order(self.topology):
            owner, t = None, self
            for part in name.split('.'):
                owner, t = t, getattr(t, part)
            out.append(t)
            where.append((owner, name.rsplit('.', 1)[1], t))
        if not replica:
            self.__dict__['_ordered_cache'] = (where, out)
        return out


class VipNeRFHip(torch.nn.Module):
    def __init__(self, configs: dict, model_configs: dict = None):
        super().__init__()
        self.configs, self.model_configs = configs, model_configs
        self.ndc = configs['data_loader']['ndc']
        m = configs['model']
        if 'coarse_mlp' not in m:
            raise L.VipNerfHipError('VipNeRFHip needs a coarse_mlp')
        self.coarse_mlp_needed = True
        self.fine_mlp_needed = 'fine_mlp' in m
        self.coarse_model = MLPParams(configs, m['coarse_mlp'])
        self.fine_model = MLPParams(configs, m['fine_mlp']) if self.fine_mlp_needed else None
        if self.fine_model is not None and self.fine_model.topology != self.coarse_model.topology:
            raise L.VipNerfHipError('VipNeRFHip: coarse and fine MLP must share one topology '
                                    f'({self.coarse_model.topology} vs {self.fine_model.topology})')
        self.topology = self.coarse_model.topology
        self.predict_visibility = self.coarse_model.predict_visibility          # VipNeRF01.py:19 (one topology for both levels here)
        if self.topology != ops.DEFAULT_TOPOLOGY and m.get('hip_precision', 'fp32') != 'fp32':
            raise L.VipNerfHipError(f"hip_precision={m.get('hip_precision')!r}: the generic-topology kernels (netdepth/netwidth/degrees "
                                    f"{self.topology}) are fp32 only")
        self._calls = 0                 # Philox offset when the caller supplies no iter_num
        self._last_iter, self._sub = None, 0
        # parity hooks (tests): injected random numbers / teacher-forced fine depths for the next forward
        self.injected_rng = None
        self.injected_z_fine = None

    # ---- reference contract ------------------------------------------------------------------------------
    def forward(self, input_batch: dict, retraw: bool = False, sec_views_vis: bool = False):
        if 'common_data' in input_batch.keys():
            for key in input_batch['common_data'].keys():           # same in-place unpacking as the reference (VipNeRF01.py:35-39) ...
                v = input_batch['common_data'][key]
                if isinstance(v, torch.Tensor) and not (key == 'poses' and v.dim() == 3):
                    # ... except that poses already unpacked to (num_frames, 4, 4) -- the same dict handed to forward() a second time --
                    # stay as they are: the reference would take poses[0] again and index cameras out of a single matrix
                    input_batch['common_data'][key] = v[0]
        return self.render_rays(input_batch, retraw or self.training, sec_views_vis or self.training)

    def _rng_key(self, input_dict: dict, dev: torch.device):
        """(seed, offset, ray_base) of this call's Philox streams (the reference draws torch.rand / torch.randn on the CPU
        generator inside the loop, VipNeRF01.py:200,242,551).  seed = the process's torch seed; offset = a pure function of
        `iter_num` and the index of the call within the iteration (the trainer's sub-batches, Trainer01.py:83-97), so a
        resumed run continues the stream and nothing depends on module state that DataParallel replicas or checkpoints
        would lose; ray_base = position of this call's ray 0 in the global batch (`rng_ray_base`, set by
        vipnerf_hip.dist.shard_batch for ray-sharded ranks: R ranks draw what one process would draw for the whole batch)."""
        seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
        replica = getattr(self, '_is_replica', False)            # multi-device nn.DataParallel rebuilds the replicas from the untouched
        if replica and 'pixel_id' not in input_dict and dev.index:   # master on every forward: no module state survives there.  Rows are
            seed = (seed ^ (dev.index * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF   # keyed by pixel (render_rays); without pixel ids: by device
        it = input_dict.get('iter_num')
        if it is None:
            offset = (1 << 62) | self._calls
            self._calls += 1
        else:
            it = int(it)
            if it == self._last_iter:
                self._sub += 1
            else:
                self._last_iter, self._sub = it, 0
            offset = (it << 16) | (self._sub & 0xFFFF)
        return seed, offset, int(input_dict.get('rng_ray_base', 0))

    def render(self, input_dict: dict, retraw: bool, sec_views_vis: bool):
        return self.render_rays(input_dict, retraw, sec_views_vis)

    def render_rays(self, input_dict: dict, retraw: bool, sec_views_vis: bool):
        """One fused launch sequence for any number of rays (the reference's chunk / netchunk host loops,
        VipNeRF01.py:47-72,295-329, are not needed)."""
        m = self.configs['model']
        rays_o = input_dict['rays_o']
        if not rays_o.is_cuda:
            raise L.VipNerfHipError('VipNeRFHip runs on the GPU only (rays_o is on %s); there is no CPU fallback' % rays_o.device)
        n = rays_o.shape[0]
        batch = {k: input_dict[k] for k in ('rays_o', 'rays_d', 'view_dirs') if k in input_dict}
        if 'view_dirs' not in batch and not m['coarse_mlp'].get('use_view_dirs'):
            batch['view_dirs'] = batch['rays_d']                # a network without view-dependent outputs reads no direction (ABI: non-NULL)
        if self.ndc:
            for k in ('rays_o_ndc', 'rays_d_ndc', 'near_ndc', 'far_ndc'):
                batch[k] = input_dict[k]
        else:
            batch['near'], batch['far'] = input_dict['near'], input_dict['far']
        V = 0
        if sec_views_vis and self.predict_visibility:           # VipNeRF01.py:84,114,148: secondary views only where visibility is predicted
            if 'rays_o2' in input_dict:
                o2 = input_dict['rays_o2']
            else:                                               # VipNeRF01.py:88-98: the other cameras' centres per row, one launch
                o2 = ops.secondary_origins(input_dict['common_data']['poses'], input_dict['pixel_id'], int(input_dict['num_frames']))
            V = o2.shape[1]
            batch['rays_o2'] = o2
        n_fine = m['fine_mlp']['num_samples'] if self.fine_mlp_needed else 0
        train = bool(self.training)
        perturb = bool(m.get('perturb', False)) and train
        noise_std = float(m.get('raw_noise_std', 0.0)) if train else 0.0
        cfg = ops.make_config(self.ndc, m['coarse_mlp']['num_samples'], n_fine, V, train=train, noise_std=noise_std,
                              lindisp=m.get('lindisp', False), white_bkgd=m.get('white_bkgd', False), perturb=perturb,
                              precision=ops.PRECISIONS[m.get('hip_precision', 'fp32')],
                              bf16_layout=ops.LAYOUTS[m.get('hip_bf16_layout', 'default')], topology=self.topology)
        rng = None
        if train:
            rng = dict(self.injected_rng) if self.injected_rng is not None else {}
            seed, offset, ray_base = self._rng_key(input_dict, rays_o.device)
            rng.setdefault('seed', seed)
            rng.setdefault('offset', offset)
            rng.setdefault('ray_base', ray_base)
            if input_dict.get('rng_ray_ids') is not None:
                rng.setdefault('ray_ids', input_dict['rng_ray_ids'])
            elif getattr(self, '_is_replica', False) and 'pixel_id' in input_dict and 'rng_ray_base' not in input_dict:
                # thread-per-device DataParallel replica (reference Trainer01.py:517 with 'device': [0, 1]): the call sees an arbitrary
                # slice of some sub-batch and keeps no state, so every row's streams are keyed by its pixel (frame, y, x) -- the same
                # numbers however the trainer cuts the batch into sub-batches and devices
                pid = input_dict['pixel_id'].long()
                rng.setdefault('ray_ids', (pid[:, 0] << 40) | (pid[:, 2] << 20) | pid[:, 1])
        max_ws = m.get('hip_max_workspace_bytes', os.environ.get('VIPNERF_MAX_WORKSPACE_BYTES'))
        state = RenderState(cfg, batch, rng, self.injected_z_fine, None if max_ws is None else int(max_ws))
        params = self.coarse_model.ordered_params() + (self.fine_model.ordered_params() if self.fine_mlp_needed else [])
        outs = RenderFunction.apply(state, *params)
        d = dict(zip(state.keys, outs))
        ret = {}
        for lv in ('coarse', 'fine') if self.fine_mlp_needed else ('coarse',):
            S = d[f'z_vals_{lv}'].shape[1]
            ret[f'z_vals_{lv}'] = d[f'z_vals_{lv}']
            for k in ('rgb', 'acc', 'alpha', 'visibility', 'weights', 'depth', 'depth_var'):
                ret[f'{k}_{lv}'] = d[f'{k}_{lv}']
            if self.ndc:
                ret[f'depth_ndc_{lv}'] = d[f'depth_ndc_{lv}']
                ret[f'depth_var_ndc_{lv}'] = d[f'depth_var_ndc_{lv}']
            if V > 0:
                ret[f'visibility2_{lv}'] = d[f'vis2_{lv}']
            if retraw:
                ret[f'raw_sigma_{lv}'] = d[f'raw_sigma_{lv}'].unsqueeze(-1)
                ret[f"raw_rgb_view_{'dependent' if self.coarse_model.view_dep_rgb else 'independent'}_{lv}"] = d[f'raw_rgb_{lv}']
                if self.predict_visibility:
                    ret[f'raw_visibility_{lv}'] = d[f'raw_vis_{lv}'].unsqueeze(-1)
                if V > 0:
                    ret[f'raw_visibility2_{lv}'] = d[f'raw_vis2_{lv}'].unsqueeze(-1)
                ret[f'raw_rgb_{lv}'] = d[f'raw_rgb_{lv}']
        if not retraw:                                           # VipNeRF01.py:168-170
            for lv in ('coarse', 'fine'):
                for k in ('z_vals', 'visibility', 'weights'):
                    ret.pop(f'{k}_{lv}', None)
        self.last_extras = state.extras
        return ret
