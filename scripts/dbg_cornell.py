import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from fujiyama_renderer_amd import workloads, host, gpu
import oracle_ffi
for objs, extra in (((), ()), (("happy",), ()), (("bunny",), ()), (("sphere",), ()), ((), (("max_diffuse_depth", (1,)),)), ((), (("max_diffuse_depth", (0,)),))):
    host.run_scene_text(workloads.cornell(workloads.default_asset_dir(), res=(64, 48), spp=(2, 2), mesh="tiny", objects=objs, extra=extra), deferred=True)
    sp, rd = host.get_desc()
    gs = gpu.Scene(sp); fb, st = gs.render_frame(rd); gs.close()
    osc = oracle_ffi.OracleScene(sp); ref, rc = osc.render(rd); osc.close()
    d = np.abs(fb - ref)
    print(objs, extra, "oracle", rc.as_dict(), "gpu", st.rays.as_dict(), "maxabs %.3g npx %d" % (d.max(), (d.max(-1) > 1e-5).sum()))
