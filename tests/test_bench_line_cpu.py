"""bench.py prints ONE stdout line that a bounded reader can parse: the compact line is a pure function of the full report
(bench.compact_line), stays under bench.COMPACT_LIMIT bytes for N = 1 and N = 8, round-trips through json.loads and carries the
contract's keys, the headline `roofline` and `cpu_baseline` objects.  Canned input: a real full report (round 4's 27 KB line, which the
driver could NOT parse -- profiles/r04_bench_1gpu.json.log), with and without the multi-rank fields."""
import copy
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config')


def _canned(n_gpus):
    full = json.loads(open(os.path.join(ROOT, 'profiles', 'r04_bench_1gpu.json.log')).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 20000                 # the input IS the oversized report
    full = copy.deepcopy(full)
    if n_gpus > 1:
        full['n_gpus'] = n_gpus
        full['value'] *= n_gpus
        full['config']['global_rays'] = full['config']['rays_per_gpu'] * n_gpus
        full['config']['parallelism'] = 'ray-sharded dp%d' % n_gpus
        full.pop('cpu_baseline')
        full.pop('sizes')
        full.update({'ranks_reduced': n_gpus, 'allreduce_ms_per_step': 0.1234, 'allreduce_calls_per_step': 1.0, 'rank_ms_per_step_min': 29.1,
                     'rank_ms_per_step_max': 29.4, 'grad_allreduce_vs_whole_batch': 2.5e-6, 'ranks_param_identical': True,
                     'sharding_check': {'rel_l2': 2.5e-6, 'ranks': n_gpus, 'passed': True}})
    full['build_info_sha16'] = '0123456789abcdef'
    full['csrc_sha16'] = 'fedcba9876543210'
    return full


@pytest.mark.parametrize('n_gpus', [1, 8])
def test_compact_line_is_small_and_parses(n_gpus):
    import bench
    full = _canned(n_gpus)
    c = bench.compact_line(full)
    line = json.dumps(c, separators=(',', ':'))
    assert len(line) < bench.COMPACT_LIMIT <= 6000, len(line)
    back = json.loads(line)
    assert back == c and '\n' not in line
    for k in CONTRACT:
        assert k in back, k
    assert back['value'] == full['value'] and back['ms_per_step'] == full['ms_per_step'] and back['dtype'] == 'f32'
    assert back['n_gpus'] == n_gpus and back['config']['global_rays'] == 4096 * n_gpus
    assert 'configs[1]' in back['config']['workload']
    r = back['roofline']
    for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'avg_launch_ms', 'launches_per_step', 'step_frac', 'sclk_mhz', 'traffic',
              'traffic_ratio', 'algorithmic_bytes_per_step'):
        assert k in r, k
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    assert all(not isinstance(v, (dict, list)) for v in r.values())          # scalar members only
    if n_gpus == 1:
        cb = back['cpu_baseline']
        assert cb['kind'] == 'port' and cb['cores'] >= 1 and cb['value'] > 0 and cb['unit'] == 'rays/s' and len(cb['sample']) <= 160
        assert back['sizes_1024_bf16_onecall_ms'] == full['sizes']['fern_1024']['bf16']['onecall']['ms_per_step']
    else:
        assert 'cpu_baseline' not in back
        assert back['allreduce_ms_per_step'] == 0.1234 and back['rank_ms_per_step_max'] == 29.4 and back['ranks_reduced'] == 8
        assert back['grad_allreduce_vs_whole_batch'] == 2.5e-6 and back['ranks_param_identical'] is True and 'sharding_check' not in back
    assert back['ms_per_step_bf16'] == full['ms_per_step_bf16'] and back['configs2_fp32_ms'] == full['configs2_realestate']['fp32']['ms_per_step']
    assert back['configs4_bf16_ms'] == full['configs4_dtu']['bf16']['ms_per_step'] and back['render_ms_per_frame'] == full['render_ms_per_frame']
    # every value outside config / roofline / cpu_baseline is a scalar
    assert all(not isinstance(v, (dict, list)) for k, v in back.items() if k not in ('config', 'roofline', 'cpu_baseline'))


def test_compact_line_sheds_extras_rather_than_growing():
    import bench
    full = _canned(1)
    for i in range(400):                                  # a report that keeps growing must not grow the line
        full['value_arith%03d' % i] = 1.0 * i
        full['roofline_arith%03d' % i] = {'frac': 0.5, 'step_frac': 0.4}
    c = bench.compact_line(full)
    assert c.get('truncated') is True and len(json.dumps(c, separators=(',', ':'))) < bench.COMPACT_LIMIT
    for k in CONTRACT + ('roofline', 'cpu_baseline'):
        assert k in c


def test_compact_line_falls_back_to_the_bare_contract():
    """after shedding the extras the line is measured AGAIN; members that are kept unconditionally (config, roofline, cpu_baseline)
    and grew past the limit are cut to bounded forms instead of printing a line the driver cannot parse."""
    import bench
    for n_gpus in (1, 8):
        full = _canned(n_gpus)
        full['config']['workload'] = full['config']['workload'] + ' x' * 3000
        full['config']['parallelism'] = 'p' * 2000
        c = bench.compact_line(full)
        line = json.dumps(c, separators=(',', ':'))
        assert len(line) <= bench.COMPACT_LIMIT and c['truncated'] is True and json.loads(line) == c
        for k in CONTRACT + ('roofline',):
            assert k in c
        assert c['value'] == full['value'] and c['roofline']['frac'] == round(full['roofline']['frac'], 4) and c['roofline']['bound'] == 'mfma'
        if n_gpus == 8:
            assert c['grad_allreduce_vs_whole_batch'] == 2.5e-6 and c['ranks_param_identical'] is True and c['ranks_reduced'] == 8
        else:
            assert c['cpu_baseline']['value'] == full['cpu_baseline']['value']
