#!/usr/bin/env python3
"""Does a large hipMalloc on a second host thread stall kernels that run meanwhile?  (what a background growth of the work arena would rely on)

    python scripts/malloc_overlap_probe.py [GB]

The main thread runs a short torch kernel + synchronize in a loop and records each round trip; a second thread allocates GB (default 100) in
1 GB pieces through hipMalloc (ctypes: the GIL is released).  Printed: round-trip times before / during / after the allocation."""
import ctypes
import sys
import threading
import time

import torch

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
if "--dirty" in sys.argv:
    # a previous tenant: a process that allocates and writes most of the memory and ends (what the driver hands out next it clears first)
    import subprocess
    subprocess.run([sys.executable, "-c", "import torch; xs=[torch.ones(1<<28, device='cuda') for _ in range(240)]; torch.cuda.synchronize()"], check=False)
import os
_tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")       # (the runtime torch itself loaded)
hip = ctypes.CDLL(_tl if os.path.exists(_tl) else "/opt/rocm/lib/libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
x = torch.zeros(64 << 20, device=dev)
torch.cuda.synchronize()
state = {"t0": None, "t1": None}


def grow():
    hip.hipSetDevice(0)
    state["t0"] = time.perf_counter()
    ps = []
    for _ in range(int(gb)):
        p = ctypes.c_void_p()
        if hip.hipMalloc(ctypes.byref(p), 1 << 30) != 0:
            break
        ps.append(p)
    state["t1"] = time.perf_counter()
    state["n"] = len(ps)
    time.sleep(0.5)
    for p in ps:
        hip.hipFree(p)


rt = []
th = threading.Thread(target=grow)
t_start = time.perf_counter()
started = False
while True:
    now = time.perf_counter()
    if not started and now - t_start > 0.5:
        th.start()
        started = True
    t = time.perf_counter()
    x.add_(1.0)
    torch.cuda.synchronize()
    rt.append((t, time.perf_counter() - t))
    if started and state["t1"] is not None and time.perf_counter() - state["t1"] > 0.3:
        break
th.join()


def stats(sel):
    v = sorted(d for _, d in sel)
    return "n %5d  median %7.3f ms  p99 %7.3f ms  max %8.3f ms" % (len(v), v[len(v) // 2] * 1e3, v[int(len(v) * .99)] * 1e3, v[-1] * 1e3) if v else "none"


# ... and is the memory, once cleared for this process and freed by it, handed out again without the wait?
t = time.perf_counter()
ps2 = []
for _ in range(int(gb)):
    p = ctypes.c_void_p()
    if hip.hipMalloc(ctypes.byref(p), 1 << 30) != 0:
        break
    ps2.append(p)
t_again = time.perf_counter() - t
q = ctypes.c_void_p()
t = time.perf_counter()
for p in ps2:
    hip.hipFree(p)
rc1 = hip.hipMalloc(ctypes.byref(q), int(gb) << 30)
t_one = time.perf_counter() - t
print("the same %d GB again after this process freed them: %.3f s in 1 GB pieces; then freed and as ONE allocation: %.3f s (rc %d)" % (len(ps2), t_again, t_one, rc1))
print("hipMalloc of %d x 1 GB on a second thread took %.3f s" % (state.get("n", 0), state["t1"] - state["t0"]))
print("kernel + synchronize round trips  before:", stats([r for r in rt if r[0] < state["t0"]]))
print("                                  during:", stats([r for r in rt if state["t0"] <= r[0] <= state["t1"]]))
print("                                   after:", stats([r for r in rt if r[0] > state["t1"]]))
