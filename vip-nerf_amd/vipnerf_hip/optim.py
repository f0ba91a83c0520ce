"""Adam over ONE flat parameter buffer -- the optimizer-side counterpart of dist.FlatGradBucket.

The path has 48 small parameter tensors (1,191,946 floats).  torch.optim.Adam walks them with multi_tensor_apply (fused: two launches of
~100 us each on MI355X -- chunked over the tensor list, ~60 workgroups on a 256-CU chip) or tensor by tensor (~300 launches): 0.2 ms of a
6 ms bf16 step.  The HIP backward already hands over all 48 gradients as consecutive views of one buffer, so here the parameters (and
Adam's two moments) are laid out the same way -- every nn.Parameter keeps its identity, shape and name, its storage becomes a view into
the flat buffer -- and one step is the six elementwise kernels of torch's own single-tensor Adam applied to the flat tensors:

    exp_avg.lerp_(g, 1 - beta1);  exp_avg_sq.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    denom = (exp_avg_sq.sqrt() / sqrt(1 - beta2^t)).add_(eps);  param.addcdiv_(exp_avg, denom, value=-lr / (1 - beta1^t))

-- the same expressions, in the same order, as torch.optim.Adam(foreach=False, fused=False) (torch/optim/adam.py::_single_tensor_adam):
elementwise, so bit-identical to it per parameter (tests/test_abi_cpu.py::test_flat_adam_is_torch_adam).  The reference's optimizer
(torch.optim.Adam, Trainer01.py:505-515, betas (0.9, 0.999), no weight decay, no amsgrad) is this update.
"""
import math
from typing import Iterable

import torch


class FlatAdam:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 5e-4, betas=(0.9, 0.999), eps: float = 1e-8):
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
        o = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self._grad[o:o + n].zero_()
            else:
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
        self.exp_avg.lerp_(g, 1 - b1)
        self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        denom = (self.exp_avg_sq.sqrt() / math.sqrt(bc2)).add_(eps)
        self.flat.addcdiv_(self.exp_avg, denom, value=-(lr / bc1))

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()
