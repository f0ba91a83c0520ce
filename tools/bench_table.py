"""Markdown rows for DESIGN.md section 5 from a bench.py JSON line:  python tools/bench_table.py gpurun_out/round/bench.json.log"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def st(r):
    s = r['stages']
    return '%.2f / %.2f / %.2f (%.2f + %.2f)' % (s['mlp_fwd']['ms_per_step'], s['mlp_dgrad']['ms_per_step'], s['wgrad']['ms_per_step'], s['wgrad_256x256']['ms_per_step'], s['wgrad_small']['ms_per_step'])
print('| arithmetic | step | rays/s | forward / data grad / weight grad (256×256 + thin) ms | roofline (§8(d), dominant kernel) |')
print('|---|---|---|---|---|')
r = d['roofline']
print('| **`fp32`** (headline, `value`) | **%.2f ms** | **%.1f k** | %s | %s: %.1f TFLOP/s = **%.3f** of 157.3; whole step %.3f; sclk %s MHz; %.1f GB per step (PMC) |' % (
    d['ms_per_step'], d['value'] / 1e3, st(r), r['kernel'].split(' (')[0], r['achieved'], r['frac'], r['step_frac'], r['sclk_mhz'], (r.get('traffic_step') or 0) / 1e9))
for p in ('fp16x3', 'fp16x3h', 'fp16', 'bf16'):
    if 'value_' + p in d:
        r = d['roofline_' + p]
        print('| `%s` | %.2f ms | %.0f k | %s | %s: %.0f TFLOP/s = %.3f of 2500; whole step %.3f%s |' % (
            p, d['ms_per_step_' + p], d['value_' + p] / 1e3, st(r), r['kernel'].split(' (')[0], r['achieved'], r['frac'], r['step_frac'],
            ('; %.1f GB per step (PMC)' % (r['traffic_step'] / 1e9)) if r.get('traffic_step') else ''))
print()
print('| other quantities | value |')
print('|---|---|')
c4 = d.get('configs4_dtu')
if c4:
    print('| `configs4_dtu` (DTU, non-NDC, V = 2, %d rays = 131 072 / 8) | %s |' % (c4['rays_per_gpu'], '; '.join('%s %.2f ms per step = %.0f k rays/s (dominant kernel %.3f of 2500)' % (p, c4[p]['ms_per_step'], c4[p]['value'] / 1e3, c4[p]['roofline']['frac']) for p in ('bf16', 'fp16') if p in c4)))
c2 = d.get('configs2_realestate')
if c2:
    print('| **`configs2_realestate`** (RealEstate geometry, V = 2, 2048 nerf + 2048 sparse-depth rows, SparseDepthMSE 0.1; 667 008 MAC / point) | %s |' % '; '.join(
        '%s %.2f ms per step = %.1f k rays/s (dominant kernel %.3f of %s)' % (p, c2[p]['ms_per_step'], c2[p]['value'] / 1e3, c2[p]['roofline']['frac'], c2[p]['roofline']['peak']) for p in ('fp32', 'bf16') if p in c2))
s = d.get('sizes')
if s:
    for k, v in s.items():
        if not isinstance(v, dict):
            continue
        parts = []
        for p in ('fp32', 'bf16'):
            m, o = v[p]['module'], v[p]['onecall']
            parts.append('%s: module contract %.3f ms (%.1f k rays/s), **one call %.3f ms (%.1f k rays/s)**, kernels %.3f ms' % (p, m['ms_per_step'], m['rays_per_sec'] / 1e3, o['ms_per_step'], o['rays_per_sec'] / 1e3, o['kernel_ms_per_step']))
        print('| **`sizes.%s`** (%d rows per iteration, %d of them sparse-depth) | %s |' % (k, v['rows'], v['sparse_depth_rows'], '; '.join(parts)))
rd = d.get('render')
if rd:
    print('| full-frame render, camera → uint8 image, 756×1008 | %s |' % '; '.join('%s %.1f ms (%.3f of %s)' % (p, v['ms_per_frame'], v['frac'], '157.3' if p == 'fp32' else '2500') for p, v in rd.items()))
cb = d.get('cpu_baseline')
if cb:
    print('| CPU baseline (oracle, %d of the box\'s host threads) | %.1f rays/s → GPU / CPU = %.0f× (fp32), %.0f× (bf16) |' % (cb['cores'], cb['value'], d['value'] / cb['value'], d.get('value_bf16', 0) / cb['value']))
