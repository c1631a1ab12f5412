"""Ad-hoc GPU parity check: HIP core vs CPU oracle on a small workload."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from fujiyama_renderer_amd import workloads, host, gpu
import oracle_ffi

def check(which, res, spp, **kw):
    txt = getattr(workloads, which)(workloads.default_asset_dir(), res=res, spp=spp, **kw)
    host.run_scene_text(txt, deferred=True)
    sp, rd = host.get_desc()
    t0 = time.time(); osc = oracle_ffi.OracleScene(sp); t1 = time.time()
    ofb, orc = osc.render(rd); t2 = time.time()
    gs = gpu.Scene(sp); t3 = time.time()
    gfb, st = gs.render_frame(rd); t4 = time.time()
    gfb, st = gs.render_frame(rd); t5 = time.time()
    d = np.abs(gfb - ofb); rel = d / np.maximum(np.abs(ofb), 1e-3)
    print("== %s %s spp %s" % (which, res, spp))
    print("oracle build %.2fs render %.2fs | gpu build %.2fs render %.3fs (2nd %.3fs)" % (t1-t0, t2-t1, t3-t2, t4-t3, t5-t4))
    print("rays oracle", orc.as_dict(), "gpu", st.rays.as_dict())
    print("max abs %.3e max rel %.3e  px>1e-4: %d of %d" % (d.max(), rel.max(), (rel.max(-1) > 1e-4).sum(), d.shape[0]*d.shape[1]))
    print("gpu ms: total %.2f trace %.2f shade %.2f gen %.2f resolve %.2f launches %d batches %d" % (st.total_ms, st.trace_ms, st.shade_ms, st.gen_ms, st.resolve_ms, st.trace_launches, st.batches))
    print("nodes %d prims %d insts %d traced %d" % (st.nodes_visited, st.prims_tested, st.insts_tested, st.rays_traced))
    # trace parity on camera-like random rays
    rng = np.random.RandomState(1)
    n = 20000
    o = np.tile(np.array([0, 1.5, 7.0]), (n, 1)) + rng.uniform(-.5, .5, (n, 3))
    dd = rng.normal(size=(n, 3)) * [0.5, 0.3, 0.1] + [0, -0.1, -1]
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    rays = np.concatenate([o, dd, np.full((n, 1), .01), np.full((n, 1), 1000.)], axis=1)
    # groups: last user+implicit all-objects group is rd target -> use group index from desc: try all-objects = n_groups-1? use 1
    for grp in (0, 1):
        try:
            tg, ig, uvg, _ = gs.trace(grp, rays)
            to, io, ao = osc.trace(grp, rays)
        except Exception as e:
            print("trace group", grp, "skipped:", e); continue
        same_t = (tg == to).mean(); same_id = (ig == io).all(axis=1).mean()
        print("trace group %d: hit frac %.3f  t bit-equal %.6f  ids equal %.6f" % (grp, (io[:, 0] >= 0).mean(), same_t, same_id))
    gs.close(); osc.close()

if __name__ == "__main__":
    print("devices:", gpu.device_count())
    if len(sys.argv) > 1 and sys.argv[1] == "cornell":
        check("cornell", (64, 48), (4, 4), mesh="tiny"); check("cornell", (160, 120), (4, 4)); sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "furry":
        check("furry", (64, 48), (2, 2), mesh="furball", nlights=4); check("furry", (160, 120), (3, 3), mesh="furball"); sys.exit(0)
    check("teapot", (64, 64), (2, 2))
    check("teapot", (256, 256), (1, 1))
    check("buddhas", (160, 90), (2, 2), mesh="bunny")
    check("dragon", (160, 90), (3, 3), mesh="small")

def check_furry():
    check("furry", (64, 48), (2, 2), mesh="furball", nlights=4)
