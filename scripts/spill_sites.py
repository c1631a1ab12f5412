#!/usr/bin/env python
"""Where a kernel's spills sit: scratch loads / stores of its ISA with the innermost loop around each.

usage: hipcc ... --cuda-device-only -S -o k.s device/fjgpu_kernels.hip; scripts/spill_sites.py k.s MANGLED_PREFIX
"""
import re
import sys

s = open(sys.argv[1]).read()
m = re.search(r'^(%s[^\n:]*):[^\n]*\n' % re.escape(sys.argv[2]), s, re.M)
name = m.group(1)
start = m.end()
end = s.index('.Lfunc_end', start)
body = s[start:end].splitlines()
labels = {}
for i, l in enumerate(body):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        labels[mm.group(1)] = i
loops = []
for i, l in enumerate(body):
    mm = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        loops.append((labels[mm.group(1)], i))
sc = [(i, l.strip()) for i, l in enumerate(body) if 'scratch_' in l]
print(name, len(body), 'lines,', len(sc), 'scratch ops,', len(loops), 'loops')
by = {}
for i, l in sc:
    inn = sorted([(b - a, a, b) for a, b in loops if a <= i <= b])
    key = inn[0][1:] if inn else None
    by.setdefault(key, []).append((i, l.split()[0]))
for key, v in sorted(by.items(), key=lambda kv: (kv[0] is None, kv[0])):
    ld = sum(1 for _, op in v if 'load' in op)
    print('innermost loop', key, 'len', (key[1] - key[0]) if key else None, ': loads', ld, 'stores', len(v) - ld, 'at', [i for i, _ in v][:12])
