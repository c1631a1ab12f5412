#!/usr/bin/env python
"""End-to-end wall time of the product's bin/scene on a workload (one process: parse, assets, BLAS build, upload, ONE cold frame, .fb written),
under a list of environments.  usage: scripts/e2e_scene.py WORKLOAD REPEATS 'label|ENV=1 ENV2=x' ..."""
import os
import subprocess
import sys
import time

root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from fujiyama_renderer_amd import workloads  # noqa: E402

wl, reps = sys.argv[1], int(sys.argv[2])
tmp = os.path.join(workloads.default_asset_dir(), "e2e_" + wl)
text = workloads.BUILDERS[wl](workloads.default_asset_dir())
open(tmp + ".scn", "w").write(text + "SaveFrameBuffer fb1 %s.fb\n" % tmp)
exe = os.path.join(root, "fujiyama-renderer_amd", "bin", "scene")
for spec in sys.argv[3:]:
    label, envs = (spec.split("|") + [""])[:2]
    env = dict(os.environ)
    for kv in envs.split():
        k, v = kv.split("=", 1)
        env[k] = v
    rows = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = subprocess.run([exe, tmp + ".scn"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
        wall = time.perf_counter() - t0
        rs = [l for l in r.stdout.splitlines() if l.startswith("# RenderScene")]
        w = rs[0].split() if rs else None
        rows.append((wall, float(w[2]) if w else -1, float(w[5]) if w else -1, r.returncode))
    print("%-28s %s" % (label, "  ".join("wall %.2f s (prepare %.2f, frame %.3f)%s" % (a, c, b, "" if rc == 0 else " rc=%d" % rc) for a, b, c, rc in rows)), flush=True)
