#!/usr/bin/env python
"""Run a list of bench.py variants on the GPU box and print one summary line each.

usage: scripts/exp.py OUTNAME  'label|ENV=1 ENV2=x|--bench --args'  ...
Writes gpurun_out/OUTNAME.txt (summary) and gpurun_out/OUTNAME.<label>.{json,err}.
"""
import json
import os
import subprocess
import sys

root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join(root, "gpurun_out")
os.makedirs(out, exist_ok=True)
name = sys.argv[1]
lines = []
for spec in sys.argv[2:]:
    label, envs, args = (spec.split("|") + ["", ""])[:3]
    env = dict(os.environ)
    for kv in envs.split():
        k, v = kv.split("=", 1)
        if k == "FJGPU_LIBDIR" and not os.path.isabs(v):
            v = os.path.join(root, v)
        env[k] = v
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--cpu-tiles", "0"] + args.split()
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    with open(os.path.join(out, "%s.%s.err" % (name, label)), "w") as f:
        f.write(p.stderr)
    line = "%-28s rc=%d" % (label, p.returncode)
    try:
        j = json.loads(p.stdout.strip().splitlines()[-1])
        with open(os.path.join(out, "%s.%s.json" % (name, label)), "w") as f:
            json.dump(j, f)
        k = {a: b for a, b in j["roofline"]["kernel_ms_per_frame_rank0"].items() if a.startswith(("k_trace", "k_shadow"))}
        c = j["config"]["counters_counting_frame_rank0"]
        m = j["config"]["ms_last_frame_rank0"]
        line += "  ms/frame %.1f  Mray/s %.0f | %s | shade %.1f gen %.1f resolve %.1f | nodes %.3g prims %.3g shtrav %.3g" % (
            j["ms_per_step"], j["value"], "  ".join("%s %.1f" % (a.replace("k_", ""), b) for a, b in k.items()),
            m["shade"], m["gen"], m["resolve"], c["nodes"], c["prims"], c["shadow_traversed"])
    except Exception as e:  # noqa: BLE001
        line += "  (no json: %s) %s" % (e, p.stderr.strip().splitlines()[-1:] )
    ph = [l for l in p.stderr.splitlines() if l.startswith(("fjgpu phase", "fjgpu curve"))][-12:]
    if ph:
        line += "\n    " + "\n    ".join(ph)
    print(line, flush=True)
    lines.append(line)
with open(os.path.join(out, name + ".txt"), "w") as f:
    f.write("\n".join(lines) + "\n")
