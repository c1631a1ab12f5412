#!/usr/bin/env python
"""HBM held by the work arena (fjgpu_scene_query "work_bytes") after one frame of a few renders, small and large.
usage (GPU box): python scripts/arena_sizes.py"""
import os
import sys

root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from fujiyama_renderer_amd import gpu, host, workloads  # noqa: E402

a = workloads.default_asset_dir()
for name, text in (("teapot 64x64 2x2 spp (32 point lights)", workloads.teapot(a, res=(64, 64), spp=(2, 2))),
                   ("ibl 64x64 2x2 spp, buddha, dome light of 256 samples", workloads.ibl(a, res=(64, 64), spp=(2, 2))),
                   ("ibl 1920x1080 8x8 spp (C6)", workloads.ibl(a)),
                   ("dragon 1920x1080 8x8 spp (C3)", workloads.dragon(a))):
    host.run_scene_text(text, deferred=True)
    sp, rd = host.get_desc()
    gs = gpu.Scene(sp)
    fb, st = gs.render_frame(rd)
    print("%-58s work arena %8.2f GB   scene %6.2f GB   frame %.1f ms" % (name, gs.query("work_bytes") / 1e9, gs.query("scene_bytes") / 1e9, st.total_ms), flush=True)
    gs.close()
