"""C3 (dragon, 1920x1080) with the adaptive grid sampler next to the fixed 8x8 grid:
frame time, camera rays, mean absolute difference between the two images"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fujiyama_renderer_amd import workloads, host, gpu


def run(extra, label):
    host.run_scene_text(workloads.dragon(workloads.default_asset_dir(), extra=extra), deferred=True)
    sp, rd = host.get_desc()
    gs = gpu.Scene(sp)
    fb = torch.zeros((rd.yres, rd.xres, 4), dtype=torch.float32, device="cuda")
    ms = []
    for i in range(3):
        torch.cuda.synchronize()
        a = time.perf_counter()
        st = gs.render_tiles(rd, list(range(gpu.tile_count(rd))), fb.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - a) * 1e3)
    print(label, "frames ms:", ["%.1f" % x for x in ms], "camera rays %.1f M" % (st.rays.camera / 1e6),
          "all rays %.1f M" % (st.rays.total() / 1e6), "launches", st.trace_launches, "batches", st.batches, flush=True)
    gs.close()
    return fb.clone()


ref = run((), "fixed 8x8")
for depth, thr in ((3, .05), (3, .02), (2, .05), (1, .05)):
    img = run((("sampler_type", (1,)), ("adaptive_max_subdivision", (depth,)), ("adaptive_subdivision_threshold", (thr,))),
              "adaptive depth %d threshold %.2f" % (depth, thr))
    print("   mean |difference to fixed 8x8| %.5f  max %.3f" % (float((img - ref).abs().mean()), float((img - ref).abs().max())), flush=True)
