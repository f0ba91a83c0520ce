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
NARROW = re.compile(r's_(and|andn2|or|xor|xnor|nand|nor)_saveexec_b64|s_(and|andn2|xor)_b64 exec')      # EXEC may lose lanes
RESTORE = re.compile(r's_or_b64 exec|s_mov_b64 exec')                                                   # lanes come back (end of the if / else region)


def scan_asm(text):
    """[(kernel, line number)] of MFMAs issued under a narrowed EXEC without an exec-branch around them.

    EXEC state is tracked LINEARLY through the whole kernel (round 5's scanner gave up at the first label or after 80 lines): a
    narrowing write opens a region that lasts -- across labels and loop bodies -- until EXEC is restored (s_or_b64 / s_mov_b64 exec); an
    s_cbranch_execz / execnz inside it means the region is skipped when no lane is left, which is what a wave-uniform condition compiles to
    (all lanes or none: an MFMA inside is then right); an MFMA in a region WITHOUT that branch is the predicated form -- it executes whatever
    EXEC says."""
    hits, kern, state = [], None, 'full'
    for i, l in enumerate(text.split('\n')):
        m = re.match(r'^(_Z\S+):', l)
        if m:
            kern, state = m.group(1), 'full'
            continue
        t = l.strip()
        if not t or t.startswith(('.', ';', '//')) and not t.startswith('.LBB'):
            continue
        if 's_endpgm' in t:
            state = 'full'
        elif NARROW.search(t):
            state = 'narrow'
        elif RESTORE.search(t):
            state = 'full'
        elif 's_cbranch_exec' in t and state == 'narrow':
            state = 'branched'
        elif ('v_mfma' in t or 'v_smfmac' in t) and state == 'narrow':
            hits.append((kern, i + 1))
    return hits


def scan_counted_waits(text, tag='dma-landed-wait'):
    """Counted waits behind an LDS DMA (k_wgrad_view<NV, HEADS = true>): `s_waitcnt vmcnt(K)` in front of a barrier proves that the
    `global_load_lds` pieces issued EARLIER have landed only if every wave issues at least K younger loads behind them.  The source tags such
    waits with an assembly comment (`tag`).  For each: the register loads between the last DMA piece and the wait, taken at the nesting level
    common to ALL of them (the branch spans every load sits in -- e.g. the wave-uniform `b + 3 < nblk` the wait itself is under), must
    number at least K: a load inside a span of its own -- an execz-skipped predicated load, a wave-role branch -- is not counted.
    -> ([(kernel, line of the wait, K, loads counted)] for every wait that fails, number of waits checked)."""
    bad, checked, kern, lines = [], 0, None, text.split('\n')
    owner = []
    for l in lines:
        m = re.match(r'^(_Z\S+):', l)
        if m:
            kern = m.group(1)
        owner.append(kern)
    for i, l in enumerate(lines):
        m = re.search(r's_waitcnt vmcnt\((\d+)\)', l)
        if not m or tag not in l:
            continue
        k = int(m.group(1))
        j = i - 1
        while j >= 0 and owner[j] == owner[i] and 'global_load_lds' not in lines[j] and 's_barrier' not in lines[j]:
            j -= 1
        checked += 1
        if j < 0 or 'global_load_lds' not in lines[j]:
            bad.append((owner[i], i + 1, k, -1))          # no DMA in front of a tagged wait: the protocol is not what the source says
            continue
        pending, spans = set(), []
        for q in range(j + 1, i):
            t = lines[q].strip()
            lab = re.match(r'^(\.LBB\S+):', t)
            if lab:
                pending.discard(lab.group(1))
            br = re.match(r's_cbranch_\w+\s+(\.LBB\S+)', t)
            if br:
                pending.add(br.group(1))
            if re.match(r'(global|buffer|flat)_load_dword', t) and 'lds' not in t:
                spans.append(frozenset(pending))
        common = frozenset.intersection(*spans) if spans else frozenset()
        loads = sum(1 for sp in spans if sp == common)
        if loads < k:
            bad.append((owner[i], i + 1, k, loads))
    return bad, checked


def assemble(path):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'k.s')
        subprocess.check_call(['hipcc'] + FLAGS + [path, '-o', out], stderr=subprocess.DEVNULL)
        return open(out).read()


def scan_source(path):
    return scan_asm(assemble(path))


if __name__ == '__main__':
    files = sys.argv[1:] or [f for f in sorted(glob.glob(os.path.join(ROOT, 'vip-nerf_amd', 'csrc', '*.hip'))) if 'mfma' in open(f).read()]
    bad = 0
    for f in files:
        for kern, line in scan_source(f):
            print('%s: %s: MFMA under a narrowed EXEC at assembly line %d' % (os.path.basename(f), kern, line))
            bad += 1
    print('%d file(s) scanned, %d EXEC-predicated MFMA(s)' % (len(files), bad))
    sys.exit(1 if bad else 0)
