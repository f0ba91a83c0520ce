"""bench.py's launcher logic on CPU: `python bench.py --gpus N` without a launcher around it becomes one (torch.distributed.run, N ranks);
a WORLD_SIZE that contradicts --gpus is an error; the line's n_gpus is the number of ranks that took part in a collective."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_respawn_cmd():
    import bench
    assert bench.respawn_cmd(1, {}, []) is None                                  # single process: run in place
    assert bench.respawn_cmd(8, {'WORLD_SIZE': '8'}, []) is None                 # already under a launcher
    cmd = bench.respawn_cmd(8, {}, ['--gpus', '8', '--steps', '3'])
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node=8' in cmd and cmd[-4:] == ['--gpus', '8', '--steps', '3']
    assert '127.0.0.1' in cmd and os.path.basename(cmd[cmd.index('--steps') - 3]) == 'bench.py'
    for ws, gpus in (('1', 8), ('8', 1), ('4', 8)):
        with pytest.raises(SystemExit):
            bench.respawn_cmd(gpus, {'WORLD_SIZE': ws}, [])


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'MASTER_ADDR')}
    env['VIPNERF_DIST_BACKEND'] = 'gloo'
    return env


@pytest.mark.parametrize('n', [2, 8])
def test_gpus_n_without_launcher_runs_n_ranks(n):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--check-ranks'], capture_output=True, text=True,
                       timeout=300, env=_clean_env(), cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, (r.stdout, r.stderr[-1000:])
    res = json.loads(lines[0])
    assert res == {'n_gpus': n, 'ranks_reduced': n, 'check_ranks': True}


def test_world_size_mismatch_is_an_error():
    env = dict(_clean_env(), WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--check-ranks'], capture_output=True, text=True, timeout=120,
                       env=env, cwd=ROOT)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
