"""Compile-time guards on the generated gfx950 code (no GPU needed: hipcc cross-compiles)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import isa_exec_mfma_scan as scan          # noqa: E402


def test_scanner_sees_a_predicated_mfma():
    asm = '\n'.join(['_Zk:', '\ts_and_saveexec_b64 s[6:7], s[8:9]', '\tv_mfma_f32_16x16x32_bf16 v[18:21], v[184:187], v[220:223], v[18:21]',
                     '\ts_or_b64 exec, exec, s[6:7]', '\ts_and_saveexec_b64 s[6:7], s[8:9]', '\ts_cbranch_execz .LBB0_1',
                     '\tv_mfma_f32_16x16x32_bf16 v[18:21], v[184:187], v[220:223], v[18:21]', '.LBB0_1:', '\ts_endpgm'])
    assert scan.scan_asm(asm) == [('_Zk', 3)]


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='hipcc not on PATH')
@pytest.mark.parametrize('src', ['vipnerf_wgrad16.hip', 'vipnerf_wgrad.hip'])
def test_no_exec_predicated_mfma_in_the_weight_gradient_kernels(src):
    """MFMA ignores EXEC (tools/isa_exec_mfma_scan.py): the kernels whose waves take different MFMAs (extra tiles, head products) must branch on
    conditions the compiler can prove wave-uniform.  k_wg16's sigma tile had the predicated form until round 5."""
    hits = scan.scan_source(os.path.join(ROOT, 'vip-nerf_amd', 'csrc', src))
    assert not hits, hits
