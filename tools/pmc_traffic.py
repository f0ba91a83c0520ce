"""profiles/r06_pmc_traffic_<prec>.json from the per-counter summaries tools/profile_round.sh leaves (tools/pmc_summary.py
tables): HBM bytes per training step and stage, MFMA-pipe busy fraction and shader clock per kernel.
    python tools/pmc_traffic.py gpurun_out/round fp32 profiles/r06_pmc_traffic_fp32.json"""
import datetime
import json
import subprocess
import sys


def table(path):
    rows = {}
    for ln in open(path).read().splitlines()[1:]:
        parts = ln.rsplit(None, 3)
        if len(parts) == 4:
            rows[parts[0].strip()] = (int(parts[1]), float(parts[2]), float(parts[3]))   # calls, total_us, counter sum
    return rows


STAGES = {'mlp_fwd': ('k_mlp_fwd',), 'mlp_dgrad': ('k_mlp_bwd',), 'wgrad': ('k_wgrad', 'k_wg16')}


def csrc_sha16():
    """sha256 (first 16 hex digits) over the kernel sources: bench.py recomputes it and says whether the profile is of the current kernels."""
    import glob, hashlib, os
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'vip-nerf_amd', 'csrc')
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, '*.hip')) + glob.glob(os.path.join(root, '*.h'))):
        h.update(os.path.basename(f).encode()); h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def main(d, prec, out):
    f, w = table(f'{d}/pmc_FETCH_SIZE_{prec}.txt'), table(f'{d}/pmc_WRITE_SIZE_{prec}.txt')
    # steps in the profiled run (timed + warm-up + initialisation): the MLP forward kernel runs twice per step (coarse, fine)
    steps_profiled = max(v[0] for k, v in f.items() if 'k_mlp_fwd' in k) // 2
    m, b = table(f'{d}/pmc_SQ_VALU_MFMA_BUSY_CYCLES_{prec}.txt'), table(f'{d}/pmc_SQ_BUSY_CYCLES_{prec}.txt')
    per = {}
    for st, keys in STAGES.items():
        rd = sum(v[2] for k, v in f.items() if any(x in k for x in keys)) * 1024 * 2 / steps_profiled
        wr = sum(v[2] for k, v in w.items() if any(x in k for x in keys)) * 1024 / steps_profiled
        per[st] = {'read': int(rd), 'written': int(wr), 'total': int(rd + wr)}
    busy = {}
    for k, (calls, us, cyc) in m.items():
        if 'vn::k_mlp' in k or 'k_wgrad_split16_256' in k or 'k_wgrad_bf16x3_256' in k or 'k_wgrad_h16_256' in k or 'k_wgrad<2, 8, 4>' in k or 'k_wgrad256_w8' in k or 'k_wg16<' in k:
            bus = b.get(k)
            if not bus:
                continue
            clk_ghz = bus[2] / bus[1] / 32 / 1e3            # SQ_BUSY_CYCLES is summed over the 32 shader engines
            busy[k] = {'sclk_ghz': round(clk_ghz, 2), 'mfma_busy_frac': round(cyc / (1024 * us * 1e-6 * clk_ghz * 1e9), 3)}
    res = {
        'workload': {'rays_per_gpu': 4096, 'precision': prec, 'scene': 'fern', 'layout': 'narrow', 'steps_profiled': steps_profiled},
        'git_head': subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip() or 'unknown',
        'date': datetime.date.today().isoformat(),
        'csrc_sha16': csrc_sha16(),
        'source': 'rocprofv3 --kernel-trace --pmc <COUNTER> (one counter per pass, tools/profile_round.sh) of `python bench.py --steps 3 '
                  '--warmup 1 --precision %s`; profiles/r05_pmc_<COUNTER>_%s.txt' % (prec, prec),
        'corrections': 'FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md '
                       'HBM section); WRITE_SIZE taken as is (k_pack_bf16n, whose output size is known, reads 1.00x)',
        'bytes_per_step': per,
        'mfma_busy': busy,
        'mfma_busy_note': 'SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel time x shader clock); clock = SQ_BUSY_CYCLES / 32 SEs / kernel time',
    }
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3])
