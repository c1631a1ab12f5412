"""per-frame wall times of a workload (first-frame costs vs steady state)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fujiyama_renderer_amd import workloads, host, gpu
name = sys.argv[1]
kw = {}
if name == "cornell":
    kw["spp"] = (6, 6)
t0 = time.perf_counter()
host.run_scene_text(workloads.BUILDERS[name](workloads.default_asset_dir(), **kw), deferred=True)
sp, rd = host.get_desc()
gs = gpu.Scene(sp)
t1 = time.perf_counter()
fb = torch.zeros((rd.yres, rd.xres, 4), dtype=torch.float32, device="cuda")
out = []
for i in range(4):
    torch.cuda.synchronize()
    a = time.perf_counter()
    st = gs.render_tiles(rd, list(range(gpu.tile_count(rd))), fb.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out.append((time.perf_counter() - a) * 1e3)
print(name, "prepare %.2f s" % (t1 - t0), "frames ms:", ["%.1f" % x for x in out], "device total_ms last: %.1f" % st.total_ms)
