#!/usr/bin/env python3
"""Per-launch timeline of a rocprofv3 --kernel-trace run: every dispatch in start order with its grid and duration.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python bench.py --workload buddhas --steps 1 --warmup 1 --no-pmc --cpu-tiles 0
    python scripts/launch_timeline.py gpurun_out/tl [--last-frame] [--min-us 50]

`--last-frame` keeps the dispatches from the last k_gen_camera on (one frame of a whole-frame batch)."""
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.match(r"(?:void )?(?:\(anonymous namespace\)::)?([A-Za-z_0-9]+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    last = "--last-frame" in sys.argv
    min_us = 0.
    for k, a in enumerate(sys.argv):
        if a == "--min-us":
            min_us = float(sys.argv[k + 1])
            args = [x for x in args if x != sys.argv[k + 1]]
    files = glob.glob(os.path.join(args[0], "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no *kernel_trace.csv under " + args[0])
    rows = []
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                             int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1), int(r["Workgroup_Size_X"])))
    rows.sort()
    if last:
        gens = [k for k, r in enumerate(rows) if "k_gen_camera" in r[2]]
        if gens:
            rows = rows[gens[-1]:]
    t0 = rows[0][0]
    total = 0.
    for s, e, name, grid, wg in rows:
        us = (e - s) / 1e3
        total += us
        if us >= min_us:
            print("%10.3f ms  +%9.1f us  grid %10d (%7d blocks of %4d)  %s" % ((s - t0) / 1e6, us, grid, grid // max(wg, 1), wg, short(name)))
    print("# %d dispatches, %.3f ms of kernels, %.3f ms from first start to last end" % (len(rows), total / 1e3, (rows[-1][1] - t0) / 1e6))


if __name__ == "__main__":
    main()
