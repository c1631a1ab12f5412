#!/usr/bin/env python
"""Registers / spills / LDS / occupancy of every kernel, from the compiler's resource-usage remarks.
usage: hipcc ... -Rpass-analysis=kernel-resource-usage 2>&1 | python scripts/kernel_resources.py [filter]"""
import re
import subprocess
import sys

flt = sys.argv[1] if len(sys.argv) > 1 else ""
rows, cur = [], {}
for l in sys.stdin:
    m = re.search(r'Function Name: (\S+)', l)
    if m:
        if cur:
            rows.append(cur)
        cur = {'name': m.group(1)}
    for k, kk in (('VGPRs', 'vgprs'), ('VGPRs Spill', 'spill'), (r'LDS Size \[bytes/block\]', 'lds'), (r'Occupancy \[waves/SIMD\]', 'waves')):
        m = re.search(k + r': (\d+)', l)
        if m:
            cur[kk] = int(m.group(1))
if cur:
    rows.append(cur)
for r in rows:
    n = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip().replace('(anonymous namespace)::', '').split('(')[0]
    if flt in n:
        print("%-52s vgprs %3d  spill %3d  lds %6d  waves/SIMD %d" % (n[:52], r.get('vgprs', 0), r.get('spill', 0), r.get('lds', 0), r.get('waves', 0)))
