"""frame time vs number of instances (instance-level BVH check)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fujiyama_renderer_amd import workloads, host, gpu
for n in (16, 64, 400, 1600):
    host.run_scene_text(workloads.crowd(workloads.default_asset_dir(), res=(640, 480), spp=(3, 3), mesh="tiny", n=n, nlights=8), deferred=True)
    sp, rd = host.get_desc()
    gs = gpu.Scene(sp)
    gs.render_frame(rd)
    fb, st = gs.render_frame(rd)
    print("crowd n=%d" % n, "ms %.1f" % st.total_ms, "rays %.1fM" % (st.rays.total() / 1e6))
    gs.close()
