import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from fujiyama_renderer_amd import workloads, host, gpu
import oracle_ffi
host.run_scene_text(workloads.cornell(workloads.default_asset_dir(), res=(64, 48), spp=(2, 2), mesh="tiny", objects=()), deferred=True)
sp, rd = host.get_desc()
rng = np.random.RandomState(3)
n = 200000
o = np.tile(np.array([0, .5, 1.85]), (n, 1))
tgt = np.stack([rng.uniform(-.6, .6, n), rng.uniform(-.1, 1.1, n), rng.uniform(-.5, .5, n)], 1)
d = tgt - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.concatenate([o, d, np.full((n, 1), .01), np.full((n, 1), 1000.)], axis=1)
gs = gpu.Scene(sp); osc = oracle_ffi.OracleScene(sp)
t, ids, uv, _ = gs.trace(0, rays); to, io, ao = osc.trace(0, rays)
bad = np.nonzero((t != to) | (ids != io).any(1))[0]
print("mismatches", len(bad), "of", n)
import collections
print("gpu inst of bad", collections.Counter(ids[bad, 0].tolist()), "oracle inst of bad", collections.Counter(io[bad, 0].tolist()))
for i in bad[:8]:
    print(i, rays[i, 3:6], "gpu", t[i], ids[i], "oracle", to[i], io[i])
