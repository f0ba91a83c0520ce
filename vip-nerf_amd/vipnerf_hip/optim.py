"""Adam over ONE flat parameter buffer -- the optimizer-side counterpart of dist.FlatGradBucket.

The path has 48 small parameter tensors (1,191,946 floats).  torch.optim.Adam walks them with multi_tensor_apply (fused: two launches of
~100 us each on MI355X -- chunked over the tensor list, ~60 workgroups on a 256-CU chip) or tensor by tensor (~300 launches): 0.2 ms of a
6 ms bf16 step.  The HIP backward already hands over all 48 gradients as consecutive views of one buffer, so here the parameters (and
Adam's two moments) are laid out the same way -- every nn.Parameter keeps its identity, shape and name, its storage becomes a view into
the flat buffer -- and one step is the six elementwise kernels of torch's own single-tensor Adam applied to the flat tensors:

    exp_avg.lerp_(g, 1 - beta1);  exp_avg_sq.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    denom = (exp_avg_sq.sqrt() / sqrt(1 - beta2^t)).add_(eps);  param.addcdiv_(exp_avg, denom, value=-lr / (1 - beta1^t))

(on the GPU by default ONE launch: the library's vipnerf_adam_step evaluates the same chain with the roundings of those kernels -- which
of them contract into an fma, the division by a host scalar as a multiplication by its double-precision reciprocal rounded to float --
found and pinned against torch on the device, tests/test_hip_fullsize.py::test_fused_adam_step_is_torch_adam; fused=False keeps the six)
-- the same expressions, in the same order, as torch.optim.Adam(foreach=False, fused=False) (torch/optim/adam.py::_single_tensor_adam):
elementwise, so bit-identical to it per parameter (tests/test_abi_cpu.py::test_flat_adam_is_torch_adam) -- PROVIDED every parameter
receives a gradient on every step, which the path's backward guarantees: torch skips a parameter whose .grad is None, one flat update cannot,
so step() raises in that case instead of silently diverging, and load_state_dict() refuses a checkpoint with per-parameter step counts.  The reference's optimizer
(torch.optim.Adam, Trainer01.py:505-515, betas (0.9, 0.999), no weight decay, no amsgrad) is this update.

state_dict() / load_state_dict() speak torch.optim.Adam's format (per-parameter 'step' / 'exp_avg' / 'exp_avg_sq', one param group), so
the `optimizer_state_dict` of a checkpoint written with either optimizer -- the reference's (Trainer01.py:352-381) included -- resumes
with the other.
"""
import math
from typing import Iterable

import torch


class FlatAdam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 5e-4, betas=(0.9, 0.999), eps: float = 1e-8, fused: bool = None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError('FlatAdam: no parameters')
        dev, dt = self.params[0].device, self.params[0].dtype
        if any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError('FlatAdam: parameters must share one device and dtype')
        self.flat = torch.cat([p.detach().reshape(-1) for p in self.params])
        o = 0
        for p in self.params:                       # same Parameter objects (names, shapes, identity): only their storage moves
            n = p.numel()
            p.data = self.flat[o:o + n].view(p.shape)
            o += n
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self._grad = torch.zeros_like(self.flat)    # gather target when the gradients are not already one buffer
        self.param_groups = [{'lr': float(lr), 'betas': (float(betas[0]), float(betas[1])), 'eps': float(eps)}]
        self.t = 0
        # fused: the whole step as ONE launch of the library's vipnerf_adam_step (same roundings; device buffers only).  None = where possible
        self.fused = self.flat.is_cuda if fused is None else bool(fused)
        if self.fused and not self.flat.is_cuda:
            raise ValueError('FlatAdam(fused=True) needs the parameters on the GPU')

    def _flat_grad(self) -> torch.Tensor:
        """The gradients as one tensor in parameter order: the buffer they already are consecutive views of (the HIP backward's
        layout, or a FlatGradBucket), else a gathered copy."""
        g0 = self.params[0].grad
        if g0 is not None and g0.is_contiguous():
            st, o = g0.untyped_storage(), g0.storage_offset()
            ok = True
            for p in self.params:
                g = p.grad
                if (g is None or g.dtype != self.flat.dtype or not g.is_contiguous() or g.storage_offset() != o
                        or g.untyped_storage().data_ptr() != st.data_ptr()):
                    ok = False
                    break
                o += g.numel()
            if ok:
                return torch.empty(0, dtype=self.flat.dtype, device=self.flat.device).set_(st, g0.storage_offset(), (self.flat.numel(),))
        # torch.optim.Adam SKIPS a parameter whose .grad is None (its moments do not decay, it does not move); one flat update cannot skip a
        # slice, and treating None as a zero gradient would keep the parameter moving on stale momentum (1.2e-2 apart from torch after two
        # steps): every parameter must have received a gradient -- which the path's backward always provides (all 48 tensors, every call)
        missing = [i for i, p in enumerate(self.params) if p.grad is None]
        if missing:
            raise RuntimeError(f'FlatAdam.step: parameter(s) {missing[:8]}{"..." if len(missing) > 8 else ""} have no gradient; unlike '
                               f'torch.optim.Adam this optimizer cannot skip individual parameters (one flat update) -- give every parameter a gradient '
                               f'(zeros to freeze its direction but decay its moments) or use torch.optim.Adam')
        o = 0
        for p in self.params:
            n = p.numel()
            self._grad[o:o + n].copy_(p.grad.reshape(-1))
            o += n
        return self._grad

    @torch.no_grad()
    def step(self):
        if self.params[0].data_ptr() != self.flat.data_ptr() or self.params[-1].data_ptr() != self.flat[self.flat.numel() - self.params[-1].numel():].data_ptr():
            raise RuntimeError('FlatAdam: the parameters no longer live in the flat buffer (module.to() / load into new tensors after the optimizer '
                               'was built?) -- build the optimizer after the model is on its device')
        grp = self.param_groups[0]
        lr, (b1, b2), eps = grp['lr'], grp['betas'], grp['eps']
        g = self._flat_grad()
        self.t += 1
        if self.fused:
            from . import ops
            ops.adam_step(self.flat, self.exp_avg, self.exp_avg_sq, g.contiguous(), lr, b1, b2, eps, self.t)
            return
        self.exp_avg.lerp_(g, 1 - b1)
        self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        denom = (self.exp_avg_sq.sqrt() / math.sqrt(bc2)).add_(eps)
        self.flat.addcdiv_(self.exp_avg, denom, value=-(lr / bc1))

    def state_dict(self) -> dict:
        state, o = {}, 0
        for i, p in enumerate(self.params):
            n = p.numel()
            if self.t > 0:                          # like torch: no per-parameter state before the first step
                state[i] = {'step': torch.tensor(float(self.t)), 'exp_avg': self.exp_avg[o:o + n].view(p.shape).clone(),
                            'exp_avg_sq': self.exp_avg_sq[o:o + n].view(p.shape).clone()}
            o += n
        g = self.param_groups[0]
        group = {'lr': g['lr'], 'betas': g['betas'], 'eps': g['eps'], 'weight_decay': 0, 'amsgrad': False, 'maximize': False,
                 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None, 'params': list(range(len(self.params)))}
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd: dict):
        groups = sd['param_groups']
        ids = [i for g in groups for i in g['params']]
        if len(groups) != 1 or len(ids) != len(self.params):
            raise ValueError(f'FlatAdam.load_state_dict: expected one param group over {len(self.params)} parameters, got {len(groups)} group(s) over {len(ids)}')
        g = groups[0]
        if g.get('weight_decay', 0) != 0 or g.get('amsgrad', False) or g.get('maximize', False):
            raise ValueError('FlatAdam.load_state_dict: weight decay / amsgrad / maximize are not part of this update')
        self.param_groups[0].update(lr=float(g['lr']), betas=(float(g['betas'][0]), float(g['betas'][1])), eps=float(g['eps']))
        state = sd['state']
        if not state:
            self.t = 0
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            return
        absent = [i for i in ids if i not in state or any(k not in state[i] for k in ('step', 'exp_avg', 'exp_avg_sq'))]
        if absent:
            raise ValueError(f'FlatAdam.load_state_dict: parameter(s) {absent[:8]} have no (or partial) optimizer state while others do -- a '
                             f'torch.optim.Adam checkpoint in which some parameters never received a gradient; one flat update keeps ONE step count '
                             f'for all parameters and cannot represent that')
        steps = {i: int(float(state[i]['step'])) for i in ids}
        if len(set(steps.values())) != 1:
            common = max(set(steps.values()), key=list(steps.values()).count)
            odd = [i for i, t in steps.items() if t != common]
            raise ValueError(f'FlatAdam.load_state_dict: parameters at different step counts (most at {common}; parameter(s) {odd[:8]} at '
                             f'{[steps[i] for i in odd[:8]]}): one flat update keeps ONE step count for all parameters')
        self.t, o = next(iter(steps.values())), 0
        for i, p in zip(ids, self.params):
            n = p.numel()
            if tuple(state[i]['exp_avg'].shape) != tuple(p.shape):
                raise ValueError(f'FlatAdam.load_state_dict: parameter {i} has shape {tuple(p.shape)}, its state {tuple(state[i]["exp_avg"].shape)}')
            self.exp_avg[o:o + n].copy_(state[i]['exp_avg'].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(state[i]['exp_avg_sq'].reshape(-1))
            o += n

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()
