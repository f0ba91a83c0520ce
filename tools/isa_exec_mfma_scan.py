"""MFMA ignores EXEC.  The compiler does not know: `if (cond) acc = mfma(a, b, acc)` with a condition it takes for per-lane (anything derived
from threadIdx without a readfirstlane) becomes  s_and_saveexec / v_mfma / s_or exec  -- and the MFMA runs in every wave, on whatever the
skipped loads left in its operand registers (round 5: profiles/r05_ab_wg16_view.log).  This scans the gfx950 assembly of
kernel sources for an MFMA that follows an EXEC write with no exec-branch in between.

    python tools/isa_exec_mfma_scan.py [file.hip ...]        (default: every vip-nerf_amd/csrc/*.hip that contains an MFMA)
prints one line per hit and exits 1 if there is any."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wno-unused-result', '-I' + os.path.join(ROOT, 'include'),
         '-S', '--cuda-device-only']
EXEC_WRITE = re.compile(r's_and_saveexec|s_or_saveexec|s_andn2_saveexec|s_(and|andn2|or|xor|mov)_b64 exec')


def scan_asm(text):
    """[(kernel, line number)] of MFMAs issued under a narrowed EXEC without an exec-branch around them."""
    hits, kern, lines = [], None, text.split('\n')
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\S+):', l)
        if m:
            kern = m.group(1)
        if re.search(r's_\w+_saveexec|s_(and|andn2)_b64 exec', l):
            for j in range(i + 1, min(i + 80, len(lines))):
                t = lines[j]
                if 's_cbranch_exec' in t or EXEC_WRITE.search(t) or t.startswith('.LBB') or 's_endpgm' in t:
                    break
                if 'v_mfma' in t or 'v_smfmac' in t:
                    hits.append((kern, j + 1))
                    break
    return hits


def scan_source(path):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'k.s')
        subprocess.check_call(['hipcc'] + FLAGS + [path, '-o', out], stderr=subprocess.DEVNULL)
        return scan_asm(open(out).read())


if __name__ == '__main__':
    files = sys.argv[1:] or [f for f in sorted(glob.glob(os.path.join(ROOT, 'vip-nerf_amd', 'csrc', '*.hip'))) if 'mfma' in open(f).read()]
    bad = 0
    for f in files:
        for kern, line in scan_source(f):
            print('%s: %s: MFMA under a narrowed EXEC at assembly line %d' % (os.path.basename(f), kern, line))
            bad += 1
    print('%d file(s) scanned, %d EXEC-predicated MFMA(s)' % (len(files), bad))
    sys.exit(1 if bad else 0)
