#!/usr/bin/env python
"""One GPU, the frame rendered by 1 / 2 / 3 / 4 scene replicas on streams of their own
(fjgpu_render_frame_multi with a repeated device index): do the tails of one replica's
persistent kernels fill with the other's work?  usage: scripts/two_streams.py [workload] [ranks]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fujiyama_renderer_amd import gpu, host, workloads  # noqa: E402

w = sys.argv[1] if len(sys.argv) > 1 else "dragon"
ranks = int(sys.argv[2]) if len(sys.argv) > 2 else 1
text = workloads.BUILDERS[w](workloads.default_asset_dir())
host.run_scene_text(text, deferred=True)
sp, rd = host.get_desc()
nt = gpu.tile_count(rd)
tiles = [t for t in range(nt) if t % ranks == 0]
for n in (1, 2, 3, 4):
    ms = gpu.MultiScene(sp, [0] * n)
    ms.render_frame(rd, tiles)
    t0 = time.perf_counter()
    for _ in range(3):
        fb, st = ms.render_frame(rd, tiles)
    dt = (time.perf_counter() - t0) / 3 * 1e3
    print("%s tiles %d/%d replicas %d: %.1f ms per frame (device totals %s)" % (w, len(tiles), nt, n, dt, ["%.1f" % s.total_ms for s in st]), flush=True)
    ms.close()
