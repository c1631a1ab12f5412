#!/usr/bin/env python
"""Shape of the persistent walks' launch tails, from a -DFJ_WAVE_TIMELINE build.

usage: FJGPU_LIBDIR=fujiyama-renderer_amd/lib_var/timeline FJGPU_TIMELINE=/tmp/tl.txt python bench.py ... ; scripts/wave_timeline.py /tmp/tl.txt

The build's launchers append, per walk launch, one line per wave: wave id, 100 MHz wall clock at its start, at its first
iteration after the queue ran dry for it, at its end, its iterations, its iterations after the queue ran dry.
Printed per launch: duration, when the FIRST wave found the queue empty (from then on the launch is "in its tail"), how many
waves are still alive at fractions of the tail, and the tail's occupancy integral (1.0 = every wave busy until the end).
"""
import sys
import numpy as np


hists = {}
splits = {}


def launches(path):
    name, rows = None, []
    for line in open(path):
        if line.startswith("#split"):
            splits[name] = line[1:].strip()
            continue
        if line.startswith("#hist"):
            hists[name] = line.split()[1:]
            continue
        if line.startswith("#"):
            if name is not None and rows:
                yield name, np.array(rows, dtype=np.float64)
            name, rows = line[1:].strip(), []
        else:
            rows.append([float(x) for x in line.split()])
    if name is not None and rows:
        yield name, np.array(rows, dtype=np.float64)


def main():
    path = sys.argv[1]
    last = int(sys.argv[2]) if len(sys.argv) > 2 else 0       # only the last N launches
    ls = list(launches(path))
    if last:
        ls = ls[-last:]
    for name, a in ls:
        t0, tdry, t1 = a[:, 1], a[:, 2], a[:, 3]
        iters, tail_it = a[:, 4], a[:, 5]
        start, end = t0.min(), t1.max()
        us = lambda x: x / 100.0            # 100 MHz ticks -> microseconds
        dur = us(end - start)
        first_dry = us(tdry.min() - start)
        tail = dur - first_dry
        line = "%-28s waves %5d  %8.1f us  start spread %6.1f us | queue dry for the first wave at %8.1f us (tail %7.1f us = %4.1f %%)" % (
            name, len(a), dur, us(t0.max() - start), first_dry, tail, 100.0 * tail / dur)
        # waves alive at fractions of the tail
        fr = [0.0, 0.1, 0.25, 0.5, 0.75, 0.9]
        alive = [int((t1 > tdry.min() + f * (end - tdry.min())).sum()) for f in fr]
        # occupancy integral of the tail
        integ = float(np.clip(t1 - tdry.min(), 0, None).sum()) / max(1.0, len(a) * (end - tdry.min()))
        line += "\n    alive at tail fraction " + "  ".join("%.2f:%d" % (f, n) for f, n in zip(fr, alive))
        line += "  | tail occupancy %.3f | a wave's own tail (dry -> end): median %.1f us, p90 %.1f, max %.1f | iterations per wave: median %d, after dry: median %d, max %d" % (
            integ, us(np.median(t1 - tdry)), us(np.percentile(t1 - tdry, 90)), us((t1 - tdry).max()), int(np.median(iters)), int(np.median(tail_it)), int(tail_it.max()))
        print(line)
        if a.shape[1] >= 9:
            wt, wi, nc = a[:, 6], a[:, 7], a[:, 8]
            o = np.argsort(-wt)[:8]
            print("    claims per wave: median %d | a wave's LONGEST claim: median %.0f us / %d iterations, p99 %.0f us, max %.0f us; the eight longest: %s" % (
                int(np.median(nc)), us(np.median(wt)), int(np.median(wi)), us(np.percentile(wt, 99)), us(wt.max()),
                "  ".join("%.0f us / %d it (%.2f us per it)" % (us(wt[k]), int(wi[k]), us(wt[k]) / max(1, wi[k])) for k in o)))
        if splits.get(name):
            print("    ray splitting: " + splits[name])
        h = hists.get(name)
        if h and any(x != "0:0" for x in h):
            tot_r = sum(int(x.split(":")[0]) for x in h) or 1
            tot_s = sum(int(x.split(":")[1]) for x in h) or 1
            print("    inner steps per ray, [2^b, 2^(b+1)): share of rays / share of steps  " + "  ".join(
                "%d: %.3f%%/%.1f%%" % (b, 100.0 * int(x.split(":")[0]) / tot_r, 100.0 * int(x.split(":")[1]) / tot_s) for b, x in enumerate(h) if x != "0:0"))


if __name__ == "__main__":
    main()
