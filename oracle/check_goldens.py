"""Regenerates every golden fixture from the REAL reference (oracle/gen_golden.py, /root/reference must exist: build container
only) into a temporary directory and compares it with the committed tests/golden/ file: same keys, every array bit for bit.
TEST INFRASTRUCTURE.  Exit status 0 = the committed fixtures are exactly what the reference produces today.

    python oracle/check_goldens.py
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, 'tests', 'golden')


def compare(committed_dir: str, fresh_dir: str):
    """-> list of (file, problem) for every committed fixture that the fresh run does not reproduce exactly."""
    bad = []
    names = sorted(set(os.listdir(committed_dir)) | set(os.listdir(fresh_dir)))
    for f in names:
        pa, pb = os.path.join(committed_dir, f), os.path.join(fresh_dir, f)
        if not os.path.exists(pa):
            bad.append((f, 'generated but not committed'))
        elif not os.path.exists(pb):
            bad.append((f, 'committed but not generated'))
        elif f.endswith('.npz'):
            x, y = np.load(pa), np.load(pb)
            if set(x.files) != set(y.files):
                bad.append((f, 'key sets differ: %s' % sorted(set(x.files) ^ set(y.files))))
            else:
                diff = [k for k in x.files if x[k].shape != y[k].shape or x[k].dtype != y[k].dtype or not np.array_equal(x[k], y[k])]
                if diff:
                    bad.append((f, 'arrays differ: %s' % diff[:8]))
        elif f.endswith('.json'):
            if json.load(open(pa)) != json.load(open(pb)):
                bad.append((f, 'json differs'))
    return bad


def main():
    if not os.path.isdir('/root/reference/src'):
        print('no /root/reference here: nothing to check against')
        return 2
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, VIPNERF_GOLDEN_OUT=tmp)
        subprocess.run([sys.executable, os.path.join(HERE, 'gen_golden.py')], check=True, env=env, cwd=ROOT, stdout=subprocess.DEVNULL)
        bad = compare(GOLD, tmp)
    for f, why in bad:
        print('MISMATCH %s: %s' % (f, why))
    print('%d fixtures, %d mismatches' % (len(os.listdir(GOLD)), len(bad)))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
