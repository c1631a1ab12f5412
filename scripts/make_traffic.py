#!/usr/bin/env python
"""profiles/r01_traffic.json from a PMC summary written by scripts/prof.sh.

usage: scripts/make_traffic.py gpurun_out/NAME_pmc.json profiles/NAME_pmc.json(for the source note)
HBM bytes per frame of the traversal-side kernels (closest hit, light loop, shadow walk): FETCH_SIZE
and WRITE_SIZE from their own --pmc passes.  The PMC run renders one timed frame and the untimed
counting frame: kernels with a counting instantiation (<..., true, ...>) are told apart by name,
the light loop is the same kernel in both frames and is halved.
"""
import json
import sys

src = sys.argv[1]
note = sys.argv[2] if len(sys.argv) > 2 else src
c = json.load(open(src))["counters"]


def pick(prefix, exclude=None):
    out = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
    for k, v in c.items():
        if prefix in k and not (exclude and exclude in k):
            for n in out:
                out[n] += v.get(n, 0.0)
    return out


closest = pick("k_trace_closest<false, false")
anyhit = pick("k_shadow_anyhit<false>")
general = pick("k_shadow_trace<", exclude=", true,")      # scenes that use the general shadow walk
cull = pick("k_shadow_cull")
cull = {n: v / 2 for n, v in cull.items()}
kern = {"k_trace_closest": closest, "k_shadow_cull": cull, "k_shadow_anyhit": anyhit}
if general["FETCH_SIZE"]:
    kern["k_shadow_trace"] = general
fetch = sum(v["FETCH_SIZE"] for v in kern.values())
write = sum(v["WRITE_SIZE"] for v in kern.values())
out = {
    "source": note + " (rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, python bench.py --steps 1 --warmup 0 --cpu-tiles 0)",
    "workload": "dragon-class scene, 1920x1080, 8x8 spp",
    "kernels": {k: {"fetch_kb_per_frame": v["FETCH_SIZE"], "write_kb_per_frame": v["WRITE_SIZE"]} for k, v in kern.items()},
    "fetch_kb_per_frame": fetch,
    "write_kb_per_frame": write,
    "correction": "gfx950 rocprofv3 tallies the 128-B L2->fabric read requests at 64 B: FETCH_SIZE is doubled before use (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is used as reported (uncalibrated)",
    "hbm_bytes_per_frame": (2 * fetch + write) * 1024.0,
}
json.dump(out, open("profiles/r01_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
