import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from fujiyama_renderer_amd import gpu, host, workloads
text = workloads.dragon(workloads.default_asset_dir())
t=time.perf_counter(); host.run_scene_text(text, deferred=True); sp, rd = host.get_desc(); print("parse+assets %.3f" % (time.perf_counter()-t))
t=time.perf_counter(); gs = gpu.Scene(sp); print("scene create %.3f" % (time.perf_counter()-t))
dev = torch.device("cuda", 0)
fb = torch.zeros((rd.yres, rd.xres, 4), dtype=torch.float32, device=dev)
host_fb = torch.empty((rd.yres, rd.xres, 4), dtype=torch.float32).pin_memory()
torch.cuda.synchronize()
stream = torch.cuda.current_stream(dev).cuda_stream
print("free before", torch.cuda.mem_get_info(dev)[0] / 1e9)
for k in range(3):
    t0 = time.perf_counter(); st = gs.render_tiles(rd, list(range(gpu.tile_count(rd))), fb.data_ptr(), stream); torch.cuda.synchronize(); t1 = time.perf_counter()
    host_fb.copy_(fb, non_blocking=False); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("frame %d: render_tiles wall %.1f ms (device %.1f ms), copy %.1f ms" % (k, (t1-t0)*1e3, st.total_ms, (t2-t1)*1e3))
print("free after", torch.cuda.mem_get_info(dev)[0] / 1e9, "work GB", gs.query("work_bytes")/1e9, "scene GB", gs.query("scene_bytes")/1e9)
