"""Compile-time guards on the generated gfx950 code (no GPU needed: hipcc cross-compiles)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import isa_exec_mfma_scan as scan          # noqa: E402


_ASM = {}


def _asm(src):
    if src not in _ASM:
        _ASM[src] = scan.assemble(os.path.join(ROOT, 'vip-nerf_amd', 'csrc', src))
    return _ASM[src]


def test_scanner_sees_a_predicated_mfma():
    asm = '\n'.join(['_Zk:', '\ts_and_saveexec_b64 s[6:7], s[8:9]', '\tv_mfma_f32_16x16x32_bf16 v[18:21], v[184:187], v[220:223], v[18:21]',
                     '\ts_or_b64 exec, exec, s[6:7]', '\ts_and_saveexec_b64 s[6:7], s[8:9]', '\ts_cbranch_execz .LBB0_1',
                     '\tv_mfma_f32_16x16x32_bf16 v[18:21], v[184:187], v[220:223], v[18:21]', '.LBB0_1:', '\ts_endpgm'])
    assert scan.scan_asm(asm) == [('_Zk', 3)]


def test_scanner_keeps_the_narrowed_state_across_labels():
    """an MFMA inside a loop body / behind a label within a saveexec region (no exec branch) is the same hazard."""
    asm = '\n'.join(['_Zk:', '\ts_and_saveexec_b64 s[6:7], s[8:9]'] + ['\tv_add_f32 v1, v2, v3'] * 120 + ['.LBB0_3:',
                     '\tv_mfma_f32_16x16x32_bf16 v[18:21], v[184:187], v[220:223], v[18:21]', '\ts_cbranch_scc1 .LBB0_3',
                     '\ts_or_b64 exec, exec, s[6:7]', '\tv_mfma_f32_16x16x32_bf16 v[18:21], v[184:187], v[220:223], v[18:21]', '\ts_endpgm'])
    assert scan.scan_asm(asm) == [('_Zk', 124)]


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='hipcc not on PATH')
@pytest.mark.parametrize('src', sorted(f for f in os.listdir(os.path.join(ROOT, 'vip-nerf_amd', 'csrc'))
                                        if f.endswith('.hip') and 'mfma' in open(os.path.join(ROOT, 'vip-nerf_amd', 'csrc', f)).read()))
def test_no_exec_predicated_mfma_in_any_mfma_kernel(src):
    """MFMA ignores EXEC (tools/isa_exec_mfma_scan.py): the kernels whose waves take different MFMAs (extra tiles, head products) must branch on
    conditions the compiler can prove wave-uniform.  k_wg16's sigma tile had the predicated form until round 5."""
    hits = scan.scan_asm(_asm(src))
    assert not hits, hits


def test_scanner_counts_the_loads_behind_a_dma():
    dma = ['\tglobal_load_lds_dwordx4 v[2:3], off']
    load = '\tglobal_load_dwordx4 v[8:11], v[4:5], off'
    wait = '\ts_waitcnt vmcnt(3) ; dma-landed-wait'
    good = '\n'.join(['_Zk:', '\ts_cbranch_scc1 .LBB0_9'] + dma + [load, load, load, '\ts_cbranch_scc0 .LBB0_2', load, '.LBB0_2:', '.LBB0_9:', wait, '\ts_barrier'])
    assert scan.scan_counted_waits(good) == ([], 1)
    # the same loads, each predicated (an execz-skipped branch of its own): a wave may issue none of them
    pred = []
    for i in range(4):
        pred += ['\ts_cbranch_execz .LBB0_%d' % (10 + i), load, '.LBB0_%d:' % (10 + i)]
    bad = '\n'.join(['_Zk:'] + dma + pred + [wait, '\ts_barrier'])
    assert scan.scan_counted_waits(bad) == ([('_Zk', 15, 3, 0)], 1)
    assert scan.scan_counted_waits('\n'.join(['_Zk:', load, load, load, wait]))[0] == [('_Zk', 5, 3, -1)]      # a tagged wait with no DMA in front of it


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='hipcc not on PATH')
def test_counted_waits_behind_the_head_dma_are_covered_by_unconditional_loads():
    """ADVICE r05 (vipnerf_wgrad.hip, k_wgrad_view<NV, HEADS = true>): `s_waitcnt vmcnt(NV + 2)` in front of the block barrier proves the head
    DMA landed only if every wave issues NV + 2 younger register loads -- which the heads-fused instantiation now issues unconditionally
    (view_gload<NV, FULL = true>); the generated code is checked, not the source."""
    bad, checked = scan.scan_counted_waits(_asm('vipnerf_wgrad.hip'))
    assert checked >= 4 and not bad, (checked, bad)              # two instantiations (NV = 1, 2) x two block parities
