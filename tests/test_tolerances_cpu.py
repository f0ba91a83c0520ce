"""The gradient bound must be able to FAIL: a parameter gradient that loses a term at the 1e-3 level -- a bias or head
gradient missing one small contribution -- is rejected by tests/test_hip_parity.py::grad_close at its fp32-grade bounds (1e-4 on the goldens
and small cases, 5e-4 at >= 4096 rows), while rounding-level differences pass.  CPU: the reference's golden gradients, perturbed."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def test_an_injected_1e3_bias_gradient_error_fails():
    import test_hip_parity as tp
    g = tp.load('f5_train_llff')
    shown = False
    for k in ('grad_fine_model.pts_linears.3.bias', 'grad_coarse_model.views_output_linear.bias', 'grad_fine_model.pts_output_linear.bias'):
        ref = g[k].astype(np.float64)
        for rows in (0, 4096):
            with pytest.raises(AssertionError) as e:
                tp.grad_close(ref * (1 + 1e-3), ref, f'injected 1e-3 error in {k}', rows=rows)          # every element 0.1 % too large
            bad = ref.copy()
            bad[np.argmax(np.abs(ref))] *= 1 - 2e-2                                                   # one element loses 2 % (a dropped term)
            with pytest.raises(AssertionError):
                tp.grad_close(bad, ref, f'one element of {k} loses 2 %', rows=rows)
            if not shown:
                print('the bound fails as it must:', e.value)
                shown = True
            tp.grad_close(ref * (1 + 3e-5) + 1e-7 * np.abs(ref).max(), ref, f'rounding-level error in {k}', rows=rows)     # what is measured passes
    # the old bound (2e-3) would have let the injected error through
    tp.grad_close(g['grad_fine_model.pts_linears.3.bias'] * (1 + 1e-3), g['grad_fine_model.pts_linears.3.bias'], 'old bound', l2_tol=2e-3)
