#!/usr/bin/env python3
"""Register / LDS / scratch use of every kernel of one csrc/*.hip (no GPU needed).

    python scripts/kernel_resources.py sweep.hip [substring]

Compiles the file with the library's flags plus -Rpass-analysis=kernel-resource-usage and prints one line per kernel.
"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphlearning_amd import _build

src = os.path.join(_build.CSRC, sys.argv[1])
match = sys.argv[2] if len(sys.argv) > 2 else ''
res = subprocess.run([_build._hipcc()] + _build._flags() + ['-c', src, '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'],
                     capture_output=True, text=True)
cur = None
rows = []
for line in res.stderr.splitlines():
    m = re.search(r'remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass', line) or re.search(r'remark:\s+(.*?) \[-Rpass', line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith('Function Name:') or t.startswith('Name:'):
        cur = {'name': t.split(':', 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip() or r['name']
    if match and match not in name:
        continue
    print('%-90s VGPR %-4s AGPR %-3s SGPR %-4s occ %-2s scratch %-5s LDS %s' % (
        name[:90], r.get('VGPRs', '?'), r.get('AGPRs', '?'), r.get('TotalSGPRs', r.get('SGPRs', '?')),
        r.get('Occupancy [waves/SIMD]', '?'), r.get('ScratchSize [bytes/lane]', '?'), r.get('LDS Size [bytes/block]', '?')))
