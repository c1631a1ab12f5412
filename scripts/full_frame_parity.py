"""One-off check at the headline size: the WHOLE C3 frame (1920x1080, 8x8 spp, 7.2 M triangles) on
the device against the CPU oracle (all host threads; a few minutes), every pixel and every ray
count.  usage (GPU box): python scripts/full_frame_parity.py [workload]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_ffi  # noqa: E402  (the checker; never on the product path)
from fujiyama_renderer_amd import gpu, host, workloads  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
host.run_scene_text(workloads.BUILDERS[name](workloads.default_asset_dir()), deferred=True)
sp, rd = host.get_desc()
gs = gpu.Scene(sp)
fb, st = gs.render_frame(rd)
fb, st = gs.render_frame(rd)
gs.close()
print("device: %.1f ms, rays %s" % (st.total_ms, st.rays.as_dict()), flush=True)
t0 = time.perf_counter()
osc = oracle_ffi.OracleScene(sp)
ref, rc = osc.render(rd, threads=os.cpu_count() or 1)
osc.close()
dt = time.perf_counter() - t0
print("oracle: %.1f s (build + render, %d threads), rays %s" % (dt, os.cpu_count() or 1, rc.as_dict()), flush=True)
rel = np.abs(fb - ref) / np.maximum(np.abs(ref), 1e-3)
print("ray counts equal:", st.rays.as_dict() == rc.as_dict())
print("pixels %d x %d: max relative error %.3g, max absolute %.3g, pixels above 1e-5 relative: %d, identical: %.2f %%" % (
    rd.xres, rd.yres, float(rel.max()), float(np.abs(fb - ref).max()), int((rel.max(axis=2) > 1e-5).sum()),
    100.0 * float((fb == ref).all(axis=2).mean())))
