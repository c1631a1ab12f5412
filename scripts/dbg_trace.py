import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from fujiyama_renderer_amd import workloads, host, gpu
import oracle_ffi
host.run_scene_text(workloads.buddhas(workloads.default_asset_dir(), res=(64, 36), spp=(1, 1), mesh="bunny"), deferred=True)
sp, rd = host.get_desc()
rng = np.random.RandomState(5)
n = 30000
o = rng.normal(size=(n, 3)) * [4, 2, 4] + [0, 1.5, 0]
tgt = rng.normal(size=(n, 3)) * [2, 1, 2] + [-1, 1, -1]
d = tgt - o
d /= np.linalg.norm(d, axis=1, keepdims=True)
d[::97] = [0, -1, 0]
d[1::97] = [1, 0, 0]
tmax = np.where(rng.uniform(size=n) < .3, rng.uniform(.1, 6, size=n), 1000.)
rays = np.concatenate([o, d, np.full((n, 1), 1e-4), tmax[:, None]], axis=1)
gs = gpu.Scene(sp); osc = oracle_ffi.OracleScene(sp)
for group in (0, 1):
    t, ids, uv, _ = gs.trace(group, rays)
    to, io, ao = osc.trace(group, rays)
    bad = np.nonzero((t != to) | (ids != io).any(1))[0]
    print("group", group, "mismatches", len(bad))
    for i in bad[:12]:
        print(i, "ray", rays[i], "gpu t %r ids %s" % (t[i], ids[i]), "oracle t %r ids %s" % (to[i], io[i]))
