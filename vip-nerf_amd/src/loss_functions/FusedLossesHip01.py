"""Shared evaluation of the four ray losses on the GPU.  The first drop-in loss class asked for its value in an
iteration runs ONE fused kernel (vipnerf_losses_forward) over the model's output dict and caches the eight
loss scalars; the other classes read them.  The cache is an attribute of the `rgb_coarse` output tensor, NOT an entry of
output_dict: the reference's trainer walks every key of the output dict (Trainer01.py:147-172 merge_output_batches_
raises on anything that is not a Tensor or dict, and would torch.cat an (8,) vector across chunks as if it were per-ray
data), so the dict must keep exactly the reference's key set.  Autograd sees a single node whose backward scales the
kernel's precomputed gradient seeds by each loss's weight."""
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
from vipnerf_hip import ops
from vipnerf_hip.autograd import FusedLossFunction, FusedLossTotalFunction

VECTOR_ATTR = '_vipnerf_hip_loss_vector'      # attribute of output_dict['rgb_coarse'] after fused_loss_total: the (8,) loss vector, detached
CACHE_ATTR = '_vipnerf_hip_fused_losses'     # attribute of output_dict['rgb_coarse']: (vector (8,), its unbind() tuple)


def _loss_inputs(configs: dict, input_dict: dict, output_dict: dict):
    m = configs['model']
    fine = 'fine_mlp' in m
    n = output_dict['rgb_coarse'].shape[0]
    V = output_dict['visibility2_coarse'].shape[1] if 'visibility2_coarse' in output_dict else 0
    cfg = ops.make_config(configs['data_loader']['ndc'], m['coarse_mlp']['num_samples'],
                          m['fine_mlp']['num_samples'] if fine else 0, V, train=False)
    prior = None
    if V > 0:
        if 'visibility_prior_masks' in input_dict:
            prior = input_dict['visibility_prior_masks']
        elif 'visibility_prior_weights' in input_dict:
            prior = input_dict['visibility_prior_weights']
    mask_sd = input_dict.get('indices_mask_sparse_depth')
    sd = input_dict['sparse_depth_values'][:, 0] if mask_sd is not None else None

    def level(lv):
        if f'rgb_{lv}' not in output_dict:
            return (None,) * 5
        # raw_visibility (N,S,1) -> (N,S) by squeeze: a view both ways ([..., 0] costs a zeros + a copy kernel in backward)
        rv = output_dict.get(f'raw_visibility_{lv}')
        # a model that predicts no visibility (mlp predict_visibility = False) has none: the kernel's visibility slots then hold a
        # number no loss class may read (VisibilityLossHip raises the reference's KeyError instead)
        rv = rv.squeeze(-1) if rv is not None else torch.zeros_like(output_dict[f'visibility_{lv}'])
        return (output_dict[f'rgb_{lv}'], output_dict[f'visibility_{lv}'], rv, output_dict.get(f'visibility2_{lv}'), output_dict[f'depth_{lv}'])
    return cfg, n, (input_dict['target_rgb'], input_dict['indices_mask_nerf'], prior, mask_sd, sd), (*level('coarse'), *(level('fine') if fine else (None,) * 5))


def fused_loss_values(configs: dict, input_dict: dict, output_dict: dict):
    """-> 8 scalar tensors: [mse_c, mse_f, vis_c, vis_f, prior_c, prior_f, sparse_depth, 0] (unweighted).  They are
    the unbind() of the kernel's result vector, so that autograd sees one Unbind node instead of one Select node (a
    zeros + a copy kernel) per value a loss class picks."""
    anchor = output_dict['rgb_coarse']
    cached = getattr(anchor, CACHE_ATTR, None)
    if cached is not None:
        return cached[1]
    cfg, n, loss_in, levels = _loss_inputs(configs, input_dict, output_dict)
    vals = FusedLossFunction.apply(cfg, n, *loss_in, *levels)
    parts = vals.unbind(0)
    setattr(anchor, CACHE_ATTR, (vals, parts))
    return parts


def fused_loss_total(configs: dict, input_dict: dict, output_dict: dict, weights8):
    """-> (TotalLoss 0-dim with the graph, loss_values (8,) detached, named (4,) detached): the weighted total of
    LossComputer.compute_losses evaluated by the loss kernels themselves (FusedLossTotalFunction; weights8 = this iteration's weight of
    every slot of the loss vector).  No PyTorch arithmetic, forward or backward."""
    cfg, n, loss_in, levels = _loss_inputs(configs, input_dict, output_dict)
    res = FusedLossTotalFunction.apply(cfg, n, [float(w) for w in weights8], *loss_in, *levels)
    setattr(output_dict['rgb_coarse'], VECTOR_ATTR, res[1])      # (for whoever wants the eight raw values: tests, diagnostics; no graph attached)
    return res


def fused_loss_vector(configs: dict, input_dict: dict, output_dict: dict):
    """The same eight values as ONE tensor of shape (8,) (what a weighted total can be taken from with a single dot)."""
    fused_loss_values(configs, input_dict, output_dict)
    return getattr(output_dict['rgb_coarse'], CACHE_ATTR)[0]
