#!/usr/bin/env python
"""GPU box: calibrate rocprofv3's FETCH_SIZE on the access pattern of the BVH walks.

Runs fujiyama-renderer_amd/bin/hbm_gather_calib (known byte counts: a streaming read, random
64-byte / 128-byte / 36-byte record gathers over a 4 GiB array) once plain for its timings and
once per counter set under `rocprofv3 --kernel-trace --pmc ...` (never combined with other trace
domains), and writes gpurun_out/<tag>_fetch_size_calibration.{json,csv}:

    factor[kernel] = bytes the kernel must read per launch / (FETCH_SIZE per launch x 1024)

bench.py multiplies the FETCH_SIZE of a traversal kernel by factor["k_calib_gather64"] (its node
records are gathered exactly like that) instead of the constant 2 the guide measured for streaming
reads.  usage: scripts/fetch_calibration.py [tag]
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
out = os.path.join(root, "gpurun_out")
os.makedirs(out, exist_ok=True)
tool = os.path.join(root, "fujiyama-renderer_amd", "bin", "hbm_gather_calib")
env = dict(os.environ, TMPDIR="/tmp")
plain = json.loads(subprocess.run([tool], stdout=subprocess.PIPE, check=True, text=True, cwd="/tmp", env=env).stdout.strip().splitlines()[-1])
sets = (("FETCH_SIZE",), ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_BUBBLE_sum"), ("TCC_HIT_sum", "TCC_MISS_sum"))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(int)
rows = []
for k, cs in enumerate(sets):
    d = "/tmp/fjcal_%d" % k
    shutil.rmtree(d, ignore_errors=True)
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + list(cs) + ["--output-format", "csv", "-d", d, "--", tool],
                   cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, timeout=900)
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            if not name.startswith("k_calib_") or name == "k_calib_fill":
                continue
            agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
            rows.append((name, r["Counter_Name"], r["Counter_Value"]))
            if r["Counter_Name"] == cs[0]:
                launches[(name, k)] += 1
res = {"tool": plain, "note": "counters summed over the launches of a kernel in one rocprofv3 pass, divided by that number of launches; "
       "FETCH_SIZE in KB as rocprofv3 reports it", "kernels": {}}
for name, c in sorted(agg.items()):
    nl0 = max(1, launches[(name, 0)])
    nl1 = max(1, launches[(name, 1)])
    need = plain["kernels"][name]["bytes_needed_per_launch"]
    fetch_b = c.get("FETCH_SIZE", 0.0) * 1024.0 / nl0
    rd = c.get("TCC_EA0_RDREQ_sum", 0.0) / nl1
    res["kernels"][name] = {
        "bytes_needed_per_launch": need, "FETCH_SIZE_bytes_per_launch": fetch_b,
        "factor_needed_over_FETCH_SIZE": need / fetch_b if fetch_b else None,
        "TCC_EA0_RDREQ_per_launch": rd, "TCC_EA0_RDREQ_32B_per_launch": c.get("TCC_EA0_RDREQ_32B_sum", 0.0) / nl1,
        "bytes_needed_per_RDREQ": need / rd if rd else None,
        "TCC_BUBBLE_per_launch": c.get("TCC_BUBBLE_sum", 0.0) / nl1,
        # a request carries 32, 64 or 128 bytes; TCC_BUBBLE counts the 128-byte ones
        "bytes_from_request_counters": (128.0 * c.get("TCC_BUBBLE_sum", 0.0) + 64.0 * (c.get("TCC_EA0_RDREQ_sum", 0.0) - c.get("TCC_BUBBLE_sum", 0.0) -
                                        c.get("TCC_EA0_RDREQ_32B_sum", 0.0)) + 32.0 * c.get("TCC_EA0_RDREQ_32B_sum", 0.0)) / nl1,
        "L2_hit_rate": c.get("TCC_HIT_sum", 0.0) / max(1.0, c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0)),
        "needed_over_bytes_from_request_counters": None,
        "ms": plain["kernels"][name]["ms"], "GBps_needed": plain["kernels"][name]["GBps_needed"]}
for v in res["kernels"].values():
    if v["bytes_from_request_counters"]:
        v["needed_over_bytes_from_request_counters"] = v["bytes_needed_per_launch"] / v["bytes_from_request_counters"]
with open(os.path.join(out, tag + "_fetch_size_calibration.json"), "w") as f:
    json.dump(res, f, indent=1)
with open(os.path.join(out, tag + "_fetch_size_calibration.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(("kernel", "counter", "value_per_dispatch"))
    w.writerows(rows)
print(json.dumps(res, indent=1))
