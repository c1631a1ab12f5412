#!/usr/bin/env python
"""How long hipMalloc takes on this box, by size, in a fresh process per size (what a cold frame's work-arena allocation waits for).
usage (GPU box): python scripts/malloc_times.py"""
import subprocess
import sys

child = r'''
import ctypes, time, sys
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
hip.hipSetDevice(0)
p = ctypes.c_void_p()
hip.hipMalloc(ctypes.byref(p), 1 << 20); hip.hipFree(p)          # runtime up
gb = float(sys.argv[1]); parts = int(sys.argv[2])
t = time.perf_counter()
ps = []
for k in range(parts):
    q = ctypes.c_void_p()
    rc = hip.hipMalloc(ctypes.byref(q), int(gb * (1 << 30) / parts))
    ps.append(q)
dt = time.perf_counter() - t
hip.hipDeviceSynchronize()
t2 = time.perf_counter()
for q in ps: hip.hipFree(q)
print("%6.1f GB in %2d allocations: hipMalloc %.3f s, hipFree %.3f s (rc %d)" % (gb, parts, dt, time.perf_counter() - t2, rc))
'''
import os
cases = ((1, 1), (4, 1), (14, 12), (14, 1), (55, 12), (110, 12), (110, 1), (14, 12), (1, 1))
if os.environ.get('FJ_MALLOC_CASES'):
    cases = tuple(tuple(float(v) if i == 0 else int(v) for i, v in enumerate(c.split('x'))) for c in os.environ['FJ_MALLOC_CASES'].split(','))
for gb, parts in cases:
    r = subprocess.run([sys.executable, "-c", child, str(gb), str(parts)], capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr.strip()[-200:], flush=True)
