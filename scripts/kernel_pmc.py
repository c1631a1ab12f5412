#!/usr/bin/env python3
"""Fabric traffic and VALU issue of ANY kernel of a workload (bench.py's own counter passes look at the dominant walk only).

    python scripts/kernel_pmc.py cornell k_shade [k_trace_closest_flat ...]

One rocprofv3 --kernel-trace --pmc pass per counter set (never combined with other trace domains), each over a one-step run of
bench.py; per kernel-name substring: launches, summed duration, bytes read (64 B x TCC_EA0_RDREQ: the lower bound, see bench.py) and
written (WRITE_SIZE), and from them GB/s over the kernel's own time; VALU busy and lane efficiency as bench.py computes them."""
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETS = (("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"), ("WRITE_SIZE",),
        ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_ANY"))


def main():
    global SETS
    if os.environ.get("KPMC_SETS"):          # e.g. KPMC_SETS="SQ_INSTS_SALU SQ_INSTS_SMEM,SQ_INSTS_VMEM_RD SQ_INSTS_LDS": extra passes, printed raw
        SETS = SETS + tuple(tuple(x.split()) for x in os.environ["KPMC_SETS"].split(","))
    workload, names = sys.argv[1], sys.argv[2:]
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    tmp = tempfile.mkdtemp(prefix="kpmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", FJGPU_COLD_START="0")
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "1", "--warmup", "0", "--cpu-tiles", "0",
             "--no-pmc", "--no-e2e"]
    acc = {n: {} for n in names}
    for k, cs in enumerate(SETS):
        d = os.path.join(tmp, "p%d" % k)
        try:
            # (own process group and a short limit: a counter set the profiler cannot schedule has hung a pass for its whole timeout before)
            subprocess.run([exe, "--kernel-trace", "--pmc"] + list(cs) + ["--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp", env=env,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=float(os.environ.get("KPMC_PASS_TIMEOUT", "240")), check=True)
        except (subprocess.TimeoutExpired, subprocess.CalledProcessError) as e:
            print("  pass %s FAILED: %s" % (" ".join(cs), type(e).__name__), flush=True)
            continue
        for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                for n in names:
                    if n in r["Kernel_Name"]:
                        a = acc[n]
                        a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        if k == 2 and os.environ.get("KPMC_PER_LAUNCH"):
            # per launch: how long its waves live on average against how long the launch takes (what is left is its tail)
            per, dur = {}, {}
            for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
                for r in csv.DictReader(open(f)):
                    if any(n in r["Kernel_Name"] for n in names):
                        per.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
            for f in glob.glob(os.path.join(d, "*", "*kernel_trace.csv")):
                for r in csv.DictReader(open(f)):
                    if r["Dispatch_Id"] in per:
                        dur[r["Dispatch_Id"]] = (float(r["Start_Timestamp"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
            ghz = float(os.environ.get("KPMC_GHZ", "2.37"))
            for did in sorted(dur, key=lambda x: dur[x][0]):
                p = per[did]
                if p.get("SQ_WAVES"):
                    life = p["SQ_WAVE_CYCLES"] / p["SQ_WAVES"] * 4.0 / (ghz * 1e9)
                    print("  launch %-44s %9.3f ms  %7d waves, average wave alive %5.1f %% of it  VALU instr %.3g lanes %.2f" %
                          (p["name"].split("(")[0][-44:], dur[did][1] * 1e-6, int(p["SQ_WAVES"]), 100. * life / (dur[did][1] * 1e-9), p.get("SQ_INSTS_VALU", 0),
                           p.get("SQ_THREAD_CYCLES_VALU", 0) / (64. * p["SQ_ACTIVE_INST_VALU"]) if p.get("SQ_ACTIVE_INST_VALU") else 0), flush=True)
        if k == 0:
            for f in glob.glob(os.path.join(d, "*", "*kernel_trace.csv")):
                for r in csv.DictReader(open(f)):
                    for n in names:
                        if n in r["Kernel_Name"]:
                            a = acc[n]
                            a["_ns"] = a.get("_ns", 0.0) + float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                            a["_launches"] = a.get("_launches", 0.0) + 1
    shutil.rmtree(tmp, ignore_errors=True)
    for n in names:
        a = acc[n]
        if not a.get("_ns"):
            print("%s: no launch" % n)
            continue
        rd, rd32 = a.get("TCC_EA0_RDREQ_sum", 0.0), a.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        rb = 64.0 * (rd - rd32) + 32.0 * rd32
        wb = a.get("WRITE_SIZE", 0.0) * 1024.0
        s = a["_ns"] * 1e-9      # (durations under the counter pass: a few per cent longer than in a plain run)
        line = "%-28s %4d launches  %8.2f ms  read >= %7.1f GB  written %7.1f GB  -> >= %6.0f GB/s" % (n, a["_launches"], s * 1e3, rb / 1e9, wb / 1e9, (rb + wb) / s / 1e9)
        if a.get("SQ_ACTIVE_INST_VALU") and a.get("SQ_WAVE_CYCLES") and a.get("SQ_WAVES"):
            lane = a["SQ_THREAD_CYCLES_VALU"] / (64.0 * a["SQ_ACTIVE_INST_VALU"])
            line += "  | VALU instr %.3g, lanes %.2f, waiting on memory %.2f of wave cycles" % (a.get("SQ_INSTS_VALU", 0.0), lane, a.get("SQ_WAIT_ANY", 0.0) / a["SQ_WAVE_CYCLES"])
        print(line, flush=True)
        extra = {k: v for k, v in a.items() if not k.startswith("_") and k not in sum((list(x) for x in SETS[:3]), [])}
        if extra:
            print("    " + "  ".join("%s %.4g" % kv for kv in sorted(extra.items())) + "  | SQ_WAVES %.4g SQ_WAVE_CYCLES %.4g SQ_BUSY_CYCLES %.4g SQ_ACTIVE_INST_VALU %.4g" %
                  (a.get("SQ_WAVES", 0), a.get("SQ_WAVE_CYCLES", 0), a.get("SQ_BUSY_CYCLES", 0), a.get("SQ_ACTIVE_INST_VALU", 0)), flush=True)


if __name__ == "__main__":
    main()
