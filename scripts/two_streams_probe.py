#!/usr/bin/env python3
"""What would two half-batches on two streams buy?  (probe, not product)

The launches of a batch end in TAILS -- a persistent walk whose queue has run dry waits for its few longest rays with the chip
almost empty (profiles/r04_wave_timeline_*.txt) -- and a batch is a chain of dependent launches, so nothing fills them.  Two
half-batches on two streams would: the one's bulk under the other's tails.  This probe measures the bound of that with what
exists: TWO device scenes of the same workload on ONE GPU (the geometry twice: worse for the caches than the real thing
would be), each rendering half of the tile list from a host thread of its own, against one scene rendering all of it.

    python scripts/two_streams_probe.py [--workload dragon] [--share 8] [--frames 5]
"""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import fujiyama_renderer_amd  # noqa: E402,F401
from fujiyama_renderer_amd import distributed as fjdist  # noqa: E402
from fujiyama_renderer_amd import gpu, host, workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="dragon")
    ap.add_argument("--share", type=int, nargs="*", default=[1, 8])
    ap.add_argument("--frames", type=int, default=5)
    ap.add_argument("--parts", type=int, nargs="*", default=[2])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    text = workloads.BUILDERS[args.workload](workloads.default_asset_dir())
    host.run_scene_text(text, deferred=True)
    scene_ptr, render = host.get_desc()
    nmax = max(args.parts)
    scenes = [gpu.Scene(scene_ptr, device=0) for _ in range(nmax)]
    gpu.global_option("cold_start", 0)
    n_tiles = gpu.tile_count(render)
    nx = -(-render.xres // render.tile_w)
    fb = torch.zeros((render.yres, render.xres, 4), dtype=torch.float32, device=dev)
    fbs = [torch.zeros_like(fb) for _ in range(nmax)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nmax)]

    def timed(fn):
        ms = []
        for _ in range(args.frames + 2):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize(dev)
            ms.append((time.perf_counter() - t0) * 1e3)
        ms = sorted(ms[2:])
        return ms[len(ms) // 2]

    for share in args.share:
        tiles = fjdist.deal_tiles(n_tiles, share, nx, "auto")[0] if share > 1 else list(range(n_tiles))
        one = timed(lambda: scenes[0].render_tiles(render, tiles, fb.data_ptr(), streams[0].cuda_stream))
        ref = fb.clone()
        print("share 1/%d (%d tiles): one scene, one stream          %8.2f ms" % (share, len(tiles), one), flush=True)
        for parts in args.parts:
            for how in ("halves", "interleaved"):
                if how == "halves":
                    n = -(-len(tiles) // parts)
                    lists = [tiles[k * n:(k + 1) * n] for k in range(parts)]
                else:
                    lists = [tiles[k::parts] for k in range(parts)]

                def run():
                    th = [threading.Thread(target=lambda k=k: scenes[k].render_tiles(render, lists[k], fbs[k].data_ptr(), streams[k].cuda_stream))
                          for k in range(parts)]
                    for t in th:
                        t.start()
                    for t in th:
                        t.join()
                ms = timed(run)
                if os.environ.get("PROBE_SEQUENTIAL_CHECK"):      # the parts one after the other: is a difference the concurrency's?
                    for k in range(parts):
                        fbs[k].zero_()
                        scenes[k].render_tiles(render, lists[k], fbs[k].data_ptr(), streams[k].cuda_stream)
                        torch.cuda.synchronize(dev)
                # the same pixels: every part wrote its own tiles
                got = torch.zeros_like(fb)
                for k in range(parts):
                    for t in lists[k]:
                        x0, y0, x1, y1 = gpu.tile_rect(render, t)
                        got[y0:y1, x0:x1] = fbs[k][y0:y1, x0:x1]
                mask = torch.zeros_like(fb, dtype=torch.bool)
                for t in tiles:
                    x0, y0, x1, y1 = gpu.tile_rect(render, t)
                    mask[y0:y1, x0:x1] = True
                same = bool(torch.equal(got[mask], ref[mask]))
                if not same:
                    d = (got - ref).abs() * mask
                    bad = (d.amax(dim=2) > 0)
                    ys, xs = torch.nonzero(bad, as_tuple=True)
                    print("   differing pixels %d of %d, max |d| %.3g, x %d..%d y %d..%d; got==0 there: %d" %
                          (int(bad.sum()), int(mask[..., 0].sum()), float(d.max()), int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max()),
                           int((got.abs().amax(dim=2)[bad] == 0).sum())), flush=True)
                print("                      %d scenes, %d streams, %-11s  %8.2f ms  (%+.1f %%)  pixels %s" %
                      (parts, parts, how, ms, (ms / one - 1) * 100, "equal" if same else "DIFFER"), flush=True)


if __name__ == "__main__":
    main()
