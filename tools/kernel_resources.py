"""Registers / spills / scratch / LDS of every kernel of a .hip file, from the compiler's own remarks (no GPU needed):
   python tools/kernel_resources.py vip-nerf_amd/csrc/vipnerf_mlp_fwd_pt2.hip [-DVN_X=v ...]"""
import re, subprocess, sys
src, flags = sys.argv[1], sys.argv[2:]
cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Rpass-analysis=kernel-resource-usage',
       '-c', src, '-o', '/dev/null'] + flags
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r'remark: +(.*?) \[-Rpass', line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        cur = subprocess.run(['c++filt', t.split(':', 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
    elif cur and ':' in t:
        k, v = t.split(':', 1)
        rows[cur][k.strip()] = v.strip()
print('%-70s %6s %6s %8s %8s %8s' % ('kernel', 'VGPRs', 'AGPRs', 'spillV', 'scratch', 'LDS'))
for k, r in rows.items():
    print('%-70s %6s %6s %8s %8s %8s' % (k[:70], r.get('VGPRs'), r.get('AGPRs'), r.get('VGPRs Spill'), r.get('ScratchSize [bytes/lane]'), r.get('LDS Size [bytes/block]')))
