"""Per-kernel register / LDS / occupancy table of one .hip file as build.py compiles it (hipcc -Rpass-analysis).
    python tools/kres.py chamfer.hip [filter]"""
import os
import re
import subprocess
import sys
import importlib.util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('vpn_build', os.path.join(ROOT, 'volumetric-primitives-net_amd', 'build.py'))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
cmd = [b.hipcc()] + b.COMMON + b.PER_FILE.get(src, []) + ['-c', os.path.join(b.CSRC, src), '-o', '/dev/null',
                                                           '-Rpass-analysis=kernel-resource-usage']
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r'remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)', line)
    if not m:
        continue
    if m.group(1) == 'Function Name':
        cur = subprocess.run(['c++filt', m.group(2)], capture_output=True, text=True).stdout.strip().split('(')[0]
        rows[cur] = {}
    elif cur:
        rows[cur][m.group(1).split(' ')[0]] = m.group(2)
for k, v in rows.items():
    if flt in k:
        print('%-60s sgpr %-4s vgpr %-4s agpr %-3s scratch %-4s occ %-2s lds %s' % (k[-60:], v.get('TotalSGPRs'), v.get('VGPRs'), v.get('AGPRs'),
              v.get('ScratchSize'), v.get('Occupancy'), v.get('LDS')))
