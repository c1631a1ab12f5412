#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X ray-intersection/integrator core.

Metric (BASELINE.json): Mray/s, primary + secondary rays, and ms/frame at
1920x1080, 64 spp.  One "step" = one pass of the hot path over one frame of
the dragon-class workload (configs[2]: xyzrgb_dragon-class dense mesh of
7.22 M triangles, plastic + constant shaders, 32 point lights): sample
generation -> camera rays -> traversal -> shading / shadow / reflection rays ->
gaussian pixel filter -> framebuffer in HOST memory (D2H included; with N > 1
also the RCCL tile gather to rank 0).  Scene parse, BLAS build and upload are
outside the timed region, like the reference's build_accelerators()
(reference src/fj_scene_interface.cc:264-270).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --scale 1,2,4,8        # the scaling curve: one compact JSON line per N, then the speed-ups

With N > 1 the line also carries "multi_gpu": one untimed frame taken apart (every rank's render time, the exchange's time)
and a self-check -- the frame assembled from the N ranks against rank 0's own render of the whole frame.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra
objects: "roofline" -- for the kernel that takes the largest share of the frame BY
MEASURED TIME (HIP events): its HBM traffic from rocprofv3's FETCH_SIZE / WRITE_SIZE
(calibrated on the walks' own access pattern, profiles/*_fetch_size_calibration.json)
over its time against the 8 TB/s HBM3E peak, its VALU issue utilisation, and -- kept
apart, never divided by the HBM peak -- the algorithmic bytes of SURVEY 8(d) -- and
"cpu_baseline" (the compiled reference -- or the CPU restatement -- timed on the
host cores on a bounded tile sample of the same workload).
"""
import argparse
import json
import os
import struct
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (read by the HSA runtime when the first HIP call initialises it -- long before the process group: RCCL's buffers cross
# processes through dmabuf IPC only on this host driver)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from fujiyama_renderer_amd import distributed as fjdist  # noqa: E402
from fujiyama_renderer_amd import gpu, host, workloads  # noqa: E402

TRAFFIC_OVER_ALGORITHMIC_LAST_MEASURED = 0.33   # fabric-side bytes / algorithmic bytes of the dominant walk, last counter run (roofline.frac_source)
HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# algorithmic bytes per traversal event (SURVEY.md 8d / DESIGN.md 7)
# (node and triangle record sizes are those of the built layout: fjgpu_scene_query)
# instance test = its 48-B box (the 96-B inverse matrix read on entry is not counted);
# closest-hit ray = DRay 48 B in (origin, direction: the range follows from the ray's class since round 6) + DHit 32 B out; a shadow ray that survives the
# instance-box cull = its 80-B queue entry (culled ones are never materialised)
S_INST, S_RAY_IN, S_HIT_OUT, S_SHADOW = 48, 48, 32, 80


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="dragon", choices=sorted(workloads.BUILDERS))
    ap.add_argument("--mesh", default=None, help="mesh class override (smaller = quicker; not a valid headline run)")
    ap.add_argument("--res", type=int, nargs=2, default=None)
    ap.add_argument("--spp", type=int, nargs=2, default=None)
    ap.add_argument("--lights", type=int, default=0, help="number of point lights (experiments; not a valid headline run)")
    ap.add_argument("--cpu-tiles", type=int, default=-1,
                    help="tiles in the CPU-baseline sample (-1: two per host core so every core stays busy; 0 disables)")
    ap.add_argument("--cpu-port-frame", action="store_true",
                    help="cpu_baseline_port on the WHOLE frame (about a minute at the headline size) instead of every 8th tile")
    ap.add_argument("--batch-tiles", type=int, default=0)
    ap.add_argument("--as-rank-of", type=int, default=0,
                    help="diagnostic: on ONE GPU render only the tiles rank 0 of an N-GPU job would own (per-rank cost)")
    ap.add_argument("--device-build", type=int, nargs="?", const=1, default=0, choices=[0, 1, 2],
                    help="build the BLAS on the GPU instead of the host SAH build: 1 locally-ordered clustering, 2 radix tree of the Morton codes")
    ap.add_argument("--rank-costs", type=int, default=0,
                    help="diagnostic: on ONE GPU time the tile share of EVERY rank of an N-GPU job (max over ranks = the N-GPU frame "
                         "time before the gather)")
    ap.add_argument("--deal", default=os.environ.get("FJ_TILE_DEAL", "auto"), choices=("auto", "lattice", "interleave"),
                    help="static tile deal of an N-GPU job: tile id %% N, the (tx + s ty) %% N lattice, or (auto) the first unless the "
                         "frame's row length makes it stripes (fujiyama_renderer_amd/distributed.py)")
    ap.add_argument("--balance-frames", type=int, default=int(os.environ.get("FJ_BALANCE_FRAMES", "8")),
                    help="N-GPU job: the first this-many frames (warm-up frames included) end with an exchange of the ranks' frame "
                         "times, and every second one with a re-deal of tiles from the slowest ranks to the fastest; 0: the static "
                         "deal throughout")
    ap.add_argument("--dry-ranks", type=int, default=0,
                    help="self-check of the N > 1 code path on ONE GPU: re-runs this script as N torch.distributed ranks (gloo, all on "
                         "cuda:0, rendering one after the other), gathers their tiles exactly like an N-GPU job and compares the "
                         "assembled frame with rank 0's own full render; timings of such a run mean nothing")
    ap.add_argument("--launcher", action="store_true",
                    help="start the ranks through torch.distributed.run (nccl) even for --gpus 1: the RCCL bring-up on one GPU; "
                         "--gpus N > 1 without a launcher around the script always does")
    ap.add_argument("--scale", default=None,
                    help="the scaling curve in ONE command, e.g. --scale 1,2,4,8: this script once per N (as `--gpus N`, which starts its own N nccl "
                         "ranks), one compact JSON line per N -- value, ms_per_step, rccl_ranks, the slowest rank's render ms, the exchange's ms, "
                         "the N > 1 self-check -- and a last line with the speed-ups over the first N; an N beyond the visible GPUs is reported as skipped")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the end-to-end leg (the product's bin/scene run as a process of its own on this workload: one cold frame, .fb written)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the rocprofv3 counter passes (HBM traffic, VALU issue) that feed the roofline object")
    return ap.parse_args()


def algorithmic_bytes(st, s_node_closest, s_prim, s_node_shadow):
    """SURVEY 8(d) bytes of one frame's traversal side: every walk's nodes at the record size THAT walk reads"""
    closest = st.rays_traced - st.rays.shadow
    nodes = (st.nodes_visited - st.shadow_nodes) * s_node_closest + st.shadow_nodes * s_node_shadow
    return (nodes + st.prims_tested * s_prim + st.insts_tested * S_INST +
            closest * (S_RAY_IN + S_HIT_OUT) + st.shadow_traversed * S_SHADOW)


def closest_algorithmic_bytes(st, s_node_closest, s_prim):
    closest = st.rays_traced - st.rays.shadow
    return ((st.nodes_visited - st.shadow_nodes) * s_node_closest + (st.prims_tested - st.shadow_prims) * s_prim +
            (st.insts_tested - st.shadow_insts) * S_INST + closest * (S_RAY_IN + S_HIT_OUT))


def fetch_calibration():
    """factor = bytes a kernel needed / (rocprofv3 FETCH_SIZE x 1024), measured on this chip for random 64-byte record
    gathers (the node fetch of the walks) by scripts/fetch_calibration.py; the newest committed file under profiles/.
    Without one the guide's streaming-read factor 2 is used and the line says so."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_fetch_size_calibration.json"))):
        try:
            k = json.load(open(f))["kernels"]
            best = {"factor": float(k["k_calib_gather64"]["factor_needed_over_FETCH_SIZE"]), "source": os.path.relpath(f, ROOT),
                    "factor_stream": float(k["k_calib_stream"]["factor_needed_over_FETCH_SIZE"]),
                    "factor_gather36": float(k.get("k_calib_gather36", {}).get("factor_needed_over_FETCH_SIZE") or 0) or None,
                    "pattern": "random 64-byte records, 4 x global_load_dwordx4 per lane, 4 GiB array"}
        except Exception:  # noqa: BLE001
            continue
    return best or {"factor": 2.0, "source": "MI355X_MICROARCH.md (streaming reads; NOT calibrated for gathers)", "pattern": None}


def cpu_baseline(args, scene_text_fn, render, scene_ptr, sample_ids, sample_rays):
    """Time the CPU path on `sample_ids` (a block of tiles around the image centre).

    kind "reference": the unmodified reference compiled from /root/reference by
    oracle/Makefile (oracle/_ref/ref_render), restricted to the sample with its own
    `render_region` property; its frame time is taken with steady_clock between the
    frame start / done callbacks.  kind "port": the CPU restatement
    (oracle/liboracle.so) on the same tiles.  Rays = the rays the device counted
    for exactly these tiles (per-context counts are identical at parity).
    """
    cores = min(os.cpu_count() or 1, 64) if args.workload in ("cornell",) else (os.cpu_count() or 1)
    rects = [gpu.tile_rect(render, t) for t in sample_ids]
    region = (min(r[0] for r in rects), min(r[1] for r in rects), max(r[2] for r in rects), max(r[3] for r in rects))
    desc = "%d tiles (%dx%d px block at %d,%d) of the %dx%d frame, %dx%d spp" % (
        len(sample_ids), region[2] - region[0], region[3] - region[1], region[0], region[1],
        render.xres, render.yres, render.rate_x, render.rate_y)
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "ref_render")
    if os.path.exists(ref_bin):
        try:
            text = scene_text_fn(extra=(("render_region", region),))
            tmp = os.path.join(workloads.default_asset_dir(), "bench_cpu_baseline")
            with open(tmp + ".scn", "w") as f:
                f.write(text)
            env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "oracle", "_ref"))
            subprocess.run([ref_bin, tmp + ".scn", tmp + ".fjfb"], env=env, check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=1500)
            with open(tmp + ".fjfb", "rb") as f:
                head = f.read(24)
            seconds = struct.unpack("<d", head[16:24])[0]
            return {"value": sample_rays / seconds / 1e6, "unit": "Mray/s",
                    "cores": min(os.cpu_count() or 1, len(sample_ids)), "host_cores": os.cpu_count() or 1, "cpu_model": cpu_model(),
                    "kind": "reference", "sample": desc, "seconds": seconds, "rays": int(sample_rays),
                    "threads_spawned": os.cpu_count() or 1,
                    "note": "the reference's use_max_thread default spawns one worker per hardware thread; at most one per tile of the "
                            "sample (= `cores`) is ever busy.  The sample is a block around the image centre, denser than the frame "
                            "average: a rate on that block, not a frame rate"}
        except Exception as e:  # fall through to the port
            sys.stderr.write("cpu_baseline: reference run failed (%s); using the CPU restatement\n" % e)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi  # test infrastructure, used here only as the timed CPU baseline
    osc = oracle_ffi.OracleScene(scene_ptr)
    t0 = time.perf_counter()
    _, rc = osc.render(render, tile_ids=sample_ids, threads=cores)
    seconds = time.perf_counter() - t0
    osc.close()
    return {"value": rc.total() / seconds / 1e6, "unit": "Mray/s", "cores": cores, "host_cores": os.cpu_count() or 1,
            "cpu_model": cpu_model(), "kind": "port", "sample": desc, "seconds": seconds, "rays": int(rc.total())}


def cpu_baseline_port(args, render, scene_ptr, n_tiles):
    """The CPU restatement of the path (oracle/liboracle.so, kind "port") on every 8th tile of the frame -- a sample with the
    frame's own mix of tiles, unlike the centre block the reference is timed on -- or, with --cpu-port-frame, on the whole
    frame; at most 64 threads (its scaling beyond that was not measured).  Scene set-up (the reference-style grid accelerators)
    is not timed, as prepare_render is not in the reference's own frame time."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi  # test infrastructure, used here only as a timed CPU baseline
    # every hardware thread of the box (64 for workloads with a PathtracingShader: rng[64]); a sample of about two tiles per thread
    cores = min(os.cpu_count() or 1, 64) if args.workload in ("cornell",) else (os.cpu_count() or 1)
    every = max(1, min(8, n_tiles // max(1, 2 * cores)))
    ids = list(range(n_tiles)) if args.cpu_port_frame else list(range(0, n_tiles, every))
    osc = oracle_ffi.OracleScene(scene_ptr)
    t0 = time.perf_counter()
    _, rc = osc.render(render, tile_ids=ids, threads=cores)
    seconds = time.perf_counter() - t0
    osc.close()
    return {"value": rc.total() / seconds / 1e6, "unit": "Mray/s", "cores": cores, "host_cores": os.cpu_count() or 1,
            "cpu_model": cpu_model(), "kind": "port",
            "sample": ("the whole frame" if args.cpu_port_frame else "every %d. tile of the frame (%d of %d tiles)" % (every, len(ids), n_tiles)) +
                      ", %dx%d, %dx%d spp" % (render.xres, render.yres, render.rate_x, render.rate_y),
            "seconds": seconds, "rays": int(rc.total()),
            "frame_seconds_extrapolated": seconds * n_tiles / max(1, len(ids))}


# read side: the L2's fabric read requests.  On gfx950 a request carries 64 or 128 bytes and no counter tells them apart
# (rocprofv3's FETCH_SIZE = 64 B x RDREQ; TCC_BUBBLE, the 128-byte count of gfx942, reads 0) -- calibrated on known byte
# counts (profiles/r03_fetch_size_calibration.json): random 64-byte record gathers, the node fetch of the walks, move 64 B per
# request; coalesced streaming reads 128 B.  So 64 x RDREQ is a LOWER bound of the bytes read, 128 x RDREQ an upper one,
# and the estimate adds, to the lower bound, half of the bytes the kernel reads as coalesced streams (its queue records).
PMC_SETS = (("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"), ("WRITE_SIZE",),
            ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU",
             "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"))


def kernel_matcher(kname):
    """production instantiation (event-counting template argument false) of a traversal kernel, by its mangled-free name"""
    def match(nm):
        if kname == "k_shadow_anyhit":
            return "k_shadow_anyhit<false" in nm
        if kname == "k_shadow_anyhit_curves":
            return "k_shadow_anyhit_curves<false" in nm
        if kname == "k_shadow_trace":
            return any(("k_shadow_trace<%s, false" % c) in nm for c in ("true", "false"))
        if kname == "k_trace_closest_phased":
            return "k_trace_closest_phased<false" in nm
        if kname == "k_trace_closest_flat":
            return "k_trace_closest_flat<false" in nm
        if kname == "k_trace_closest":
            return any(("k_trace_closest<%s, false" % c) in nm for c in ("true", "false"))
        return kname in nm
    return match


def pmc_passes(args, kname):
    """Hardware counters of the dominant kernel for THIS build, measured now: one rocprofv3
    --kernel-trace --pmc pass per counter set (FETCH_SIZE and WRITE_SIZE do not fit one pass,
    /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots"; never combined with other
    trace domains), each running this script as a child for ONE timed frame.  Returns
    {counter: sum over the launches of the production instantiation in that frame} (the child may
    cut the frame into another number of launches than the parent: per-frame sums are comparable,
    per-launch ones are not), or None when rocprofv3 is not usable here."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out, tmp = {}, tempfile.mkdtemp(prefix="fjpmc_", dir="/tmp")
    match = kernel_matcher(kname)
    child = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--cpu-tiles", "0", "--no-pmc",
             "--workload", args.workload]
    if args.mesh:
        child += ["--mesh", args.mesh]
    if args.res:
        child += ["--res"] + [str(v) for v in args.res]
    if args.spp:
        child += ["--spp"] + [str(v) for v in args.spp]
    if args.device_build:
        child += ["--device-build", str(args.device_build)]
    env = dict(os.environ, TMPDIR="/tmp", FJGPU_COLD_START="0")       # (ONE production frame in the child: its sums are per frame)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        env.pop(k, None)                   # (a parent started through the launcher: the counter child is a plain one-GPU run)
    try:
        for k, cs in enumerate(PMC_SETS):
            d = os.path.join(tmp, "p%d" % k)
            cmd = [exe, "--kernel-trace", "--pmc"] + list(cs) + ["--output-format", "csv", "-d", d, "--"] + child
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600, check=True)
            for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
                for r in csv.DictReader(open(f)):
                    # the production instantiation: the event-counting template argument is false
                    # (k_shadow_anyhit<count, multi>, k_shadow_trace<curves, count, motion>)
                    if not match(r["Kernel_Name"]):
                        continue
                    out[r["Counter_Name"]] = out.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                    if r["Counter_Name"] == "SQ_WAVES":
                        out["_launches"] = out.get("_launches", 0.0) + 1.0
        return out or None
    except Exception as e:  # noqa: BLE001
        sys.stderr.write("bench: rocprofv3 counter passes failed (%s): roofline.traffic is null\n" % e)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def scene_binary_end_to_end(scene_text_fn):
    """What a user of the reference actually runs: ONE frame per `bin/scene` process (tools/scene_parser/main.cc:34-43) --
    parse the scene text, read the PLY assets, build the accelerators, upload, allocate the work buffers, render the frame
    cold, copy it to the host and write the .fb file.  Wall time of the product's own bin/scene on this workload's text with a
    SaveFrameBuffer appended, plus the parts the binary reports itself."""
    exe = os.path.join(ROOT, "fujiyama-renderer_amd", "bin", "scene")
    if not os.path.exists(exe):
        return None
    try:
        tmp = os.path.join(workloads.default_asset_dir(), "bench_e2e")
        with open(tmp + ".scn", "w") as f:
            f.write(scene_text_fn() + "SaveFrameBuffer fb1 %s.fb\n" % tmp)
        def run(env):
            t0 = time.perf_counter()
            r = subprocess.run([exe, tmp + ".scn"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
            wall = time.perf_counter() - t0
            if r.returncode != 0:
                return {"error": (r.stderr or r.stdout)[-300:]}
            o = {"wall_seconds": wall}
            for line in r.stdout.splitlines():
                if line.startswith("# RenderScene"):
                    w = line.split()
                    o["render_scene_seconds"] = float(w[2])
                    o["prepare_seconds"] = float(w[5])
            return o
        out = run(dict(os.environ))
        if "error" in out:
            return out
        out["fb_bytes"] = os.path.getsize(tmp + ".fb") if os.path.exists(tmp + ".fb") else None
        out["what"] = ("bin/scene <file>: parse + assets + BLAS build + upload + ONE cold frame + .fb written (one process, one GPU); SiRenderScene renders "
                       "its one frame in batches of 16 M samples (FJ_BATCH_SAMPLES): a work arena of a few GB instead of ~110 GB")
        # ... and the same with the core's default batch (the whole frame at once: what the timed steps above run): the cold frame then
        # waits for ~110 GB of work buffers, up to seconds when the driver has to clear that memory first
        out["with_whole_frame_batches"] = run(dict(os.environ, FJ_BATCH_SAMPLES="0"))
        for ext in (".scn", ".fb"):
            try:
                os.remove(tmp + ext)
            except OSError:
                pass
        return out
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)[-300:]}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def dry_ranks_parent(args):
    """--dry-ranks N: this script again as N ranks under torch.distributed.run, all on cuda:0 over gloo"""
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.dry_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), "--gpus", str(args.dry_ranks),
           "--steps", str(args.steps), "--warmup", str(args.warmup), "--cpu-tiles", "0", "--no-pmc", "--workload", args.workload]
    if args.mesh:
        cmd += ["--mesh", args.mesh]
    if args.res:
        cmd += ["--res"] + [str(v) for v in args.res]
    if args.spp:
        cmd += ["--spp"] + [str(v) for v in args.spp]
    env = dict(os.environ, FJ_BENCH_DRY="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    # N processes share ONE GPU here, each with a scene replica and work buffers of its own: the shadow queue (up to 41 GB per process
    # when it has a GPU to itself) is held to a share of it, or eight ranks at the headline size do not fit the 288 GB
    env.setdefault("FJGPU_SQUEUE_M", str(max(8, 256 // max(1, args.dry_ranks))))
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: this script again as N ranks, one per GPU, under
    torch.distributed.run with the nccl (= RCCL) backend -- the command shape the driver uses for N > 1, spelled by the
    script itself so that a bare `--gpus N` measures N GPUs (the reference's worker pool sizes itself the same way,
    src/fj_renderer.cc:632-641).  `--launcher` forces it for N = 1 as well (the RCCL bring-up on one GPU)."""
    import socket
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the fjgpu core has no CPU fallback")
    have = torch.cuda.device_count()
    if have < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node (one rank per GPU; no oversubscription -- "
                         "--dry-ranks N is the one-GPU code-path check)" % (args.gpus, have))
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    argv = [a for a in sys.argv[1:] if a != "--launcher"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FJ_BENCH_LAUNCHED="1")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def scale_points(text):
    pts = []
    for tok in str(text).replace(" ", "").split(","):
        if tok:
            n = int(tok)
            if n < 1:
                raise ValueError("--scale: counts are >= 1")
            pts.append(n)
    if not pts:
        raise ValueError("--scale: empty list")
    return pts


def scale_line(n, line):
    """the compact per-N record of a --scale run from the full bench line of `--gpus n`"""
    mg = line.get("multi_gpu") or {}
    return {"scale_point": n, "metric": line["metric"], "value": line["value"], "unit": line["unit"], "n_gpus": line["n_gpus"],
            "rccl_ranks": line["rccl_ranks"], "backend": line["backend"], "steps": line["steps"], "warmup": line["warmup"],
            "ms_per_step": line["ms_per_step"], "slowest_rank_render_ms": mg.get("slowest_rank_render_ms"),
            "exchange_ms": mg.get("exchange_ms"), "render_ms_by_rank": mg.get("render_ms_by_rank"),
            "self_check": mg.get("self_check"), "tiles_per_rank": ((line["config"].get("tile_balance") or {}).get("tiles_per_rank")),
            "workload": line["config"]["workload"]}


def scale_parent(args, run=None):
    """--scale: one child per N (its own process tree: `--gpus N` starts N ranks over RCCL), rank 0's lines collected here.
    `run(n) -> dict | None` is injectable (tests/test_bench_contract.py drives the plumbing without a GPU)."""
    pts = scale_points(args.scale)
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0

    def child(n):
        argv, skip = [], False
        for a in sys.argv[1:]:
            if skip:
                skip = False
                continue
            if a == "--scale":
                skip = True
                continue
            if a.startswith("--scale=") or a in ("--gpus",):
                skip = a == "--gpus"
                continue
            if a.startswith("--gpus="):
                continue
            argv.append(a)
        for flag in ("--no-pmc", "--no-e2e"):
            if flag not in argv:
                argv.append(flag)
        if "--cpu-tiles" not in argv:
            argv += ["--cpu-tiles", "0"]
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", str(n)] + argv, stdout=subprocess.PIPE, text=True)
        if p.returncode != 0:
            return {"scale_point": n, "error": "bench.py --gpus %d exited with %d" % (n, p.returncode)}
        for ln in reversed(p.stdout.strip().splitlines()):
            if ln.startswith("{"):
                return scale_line(n, json.loads(ln))
        return {"scale_point": n, "error": "no JSON line"}
    run = run or child
    lines = []
    for n in pts:
        if run is child and n > have:
            rec = {"scale_point": n, "skipped": "%d GPU(s) visible on this node" % have}
        else:
            rec = run(n)
        lines.append(rec)
        print(json.dumps(rec), flush=True)
    ok = [r for r in lines if "value" in r]
    summary = {"scale_summary": True, "points": [r["scale_point"] for r in ok], "value": [r["value"] for r in ok],
               "ms_per_step": [r["ms_per_step"] for r in ok],
               "speedup_over_first": [ok[0]["ms_per_step"] / r["ms_per_step"] for r in ok] if ok else [],
               "skipped": [r["scale_point"] for r in lines if "skipped" in r], "failed": [r["scale_point"] for r in lines if "error" in r],
               "note": "strong scaling of one frame; efficiency is for the reader to compute (speed-up / N)"}
    print(json.dumps(summary), flush=True)
    return 1 if summary["failed"] else 0


def main():
    args = parse_args()
    if args.scale:
        raise SystemExit(scale_parent(args))
    dry = os.environ.get("FJ_BENCH_DRY") == "1"
    if args.dry_ranks > 1 and not dry:
        dry_ranks_parent(args)
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not under_launcher and not dry and (args.gpus > 1 or args.launcher):
        launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if dry else int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the fjgpu core has no CPU fallback")
    if under_launcher and not dry and args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s): the two must agree" % (args.gpus, world))
    if not dry and torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU of its own (%d visible): one rank per GPU" % (rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)            # before the process group: RCCL binds to the current device
    device = torch.device("cuda", local_rank)
    backend = None
    if world > 1 or under_launcher:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if dry:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)
            # RCCL bring-up check before any frame: one all-reduce across the ranks (also with ONE rank: the communicator
            # is created and a kernel of the collective library runs on this GPU)
            probe = torch.ones(1, dtype=torch.float64, device=device)
            dist.all_reduce(probe)
            if int(probe.item()) != world:
                raise SystemExit("bench.py: RCCL all-reduce over %d rank(s) returned %r" % (world, probe.item()))
            # ... and the frame exchange's own collective (dist.gather of equal slabs to rank 0) on a token slab
            tok = torch.full((4,), float(rank), dtype=torch.float32, device=device)
            got = [torch.empty_like(tok) for _ in range(world)] if rank == 0 else None
            dist.gather(tok, gather_list=got, dst=0)
            if rank == 0 and [int(g[0].item()) for g in got] != list(range(world)):
                raise SystemExit("bench.py: RCCL gather over %d rank(s) returned the slabs out of order" % world)
        backend = dist.get_backend()

    # ---------------- untimed: assets, scene, BLAS build, upload
    kw = {}
    if args.mesh:
        kw["mesh"] = args.mesh
    if args.res:
        kw["res"] = tuple(args.res)
    if args.spp:
        kw["spp"] = tuple(args.spp)
    if args.lights:
        kw["nlights"] = args.lights
    builder = workloads.BUILDERS[args.workload]
    asset_dir = workloads.default_asset_dir()
    if world > 1:
        if rank == 0:
            builder(asset_dir, **kw)          # generate assets once
        dist.barrier()

    def scene_text(extra=()):
        return builder(asset_dir, extra=extra, **kw)

    t_prep = time.perf_counter()
    host.run_scene_text(scene_text(), deferred=True)
    scene_ptr, render = host.get_desc()
    if args.device_build:
        gpu.global_option("device_build", args.device_build)
    gs = gpu.Scene(scene_ptr, device=local_rank)
    s_node, s_prim = gs.query("node_record_bytes"), gs.query("tri_record_bytes")
    # the lean any-hit walk reads the quantised 64-byte twin of a node; so does the closest-hit walk of scenes
    # without curve sets / motion (fjgpu_dev_traverse.h)
    s_node_walk = gs.query("anyhit_node_record_bytes") if (gs.query("lean_anyhit") or gs.query("curve_anyhit")) else s_node     # (both read the 64-byte quantised nodes)
    s_node_closest = gs.query("closest_node_record_bytes")
    if args.batch_tiles:
        gs.set_option("batch_tiles", args.batch_tiles)
    if os.environ.get("FJGPU_OVERLAP"):            # experiment switch: light loop + shadow walk on a second stream
        gs.set_option("overlap_shadow", int(os.environ["FJGPU_OVERLAP"]))
    prep_seconds = time.perf_counter() - t_prep

    n_tiles = gpu.tile_count(render)
    tiles_per_row = -(-render.xres // render.tile_w)
    capacity = fjdist.slab_capacity(n_tiles, world) if args.balance_frames > 0 else None
    balance = fjdist.TileBalance(fjdist.deal_tiles(n_tiles, world, tiles_per_row, args.deal), capacity,
                                 frames=args.balance_frames if world > 1 else 0)
    my_tiles = balance.lists[rank]
    if args.as_rank_of > 1 and world == 1:
        my_tiles = fjdist.deal_tiles(n_tiles, args.as_rank_of, tiles_per_row, args.deal)[0]
    tile_rects = [gpu.tile_rect(render, t) for t in range(n_tiles)]
    fb = torch.zeros((render.yres, render.xres, 4), dtype=torch.float32, device=device)
    host_fb = torch.empty((render.yres, render.xres, 4), dtype=torch.float32).pin_memory()
    stream = torch.cuda.current_stream(device).cuda_stream

    def measure_tail(G, deal):
        """what an N-GPU frame adds after its slowest rank is done, less the exchange itself: packing a rank's tiles into
        the slab, scattering the N slabs into the frame, the frame's copy to the host -- measured on this GPU"""
        slabs = fjdist._DeviceSlabs(fb, tile_rects, n_tiles, render.tile_w, render.tile_h, 0, G, deal, fjdist.slab_capacity(n_tiles, G))
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        for _ in range(5):
            gpu.pack_tiles(fb.data_ptr(), render.xres, slabs.mine.data_ptr(), slabs.per_rank, slabs.tile_px, slabs.slab.data_ptr(), stream)
            gpu.unpack_tiles(fb.data_ptr(), render.xres, slabs.all.data_ptr(), G * slabs.per_rank, slabs.tile_px, slabs.recv.data_ptr(), stream)
            host_fb.copy_(fb, non_blocking=False)
        torch.cuda.synchronize(device)
        return (time.perf_counter() - t1) / 5 * 1e3, slabs.slab.numel() * 4

    presize_ms = None
    if world > 1 and balance.adapting:
        # the work buffers follow the longest tile list a call has seen, and growing them costs seconds: one untimed frame
        # of as many tiles as the balance may ever hand to this rank
        own = set(my_tiles)
        spare = [t for t in range(n_tiles) if t not in own]
        t1 = time.perf_counter()
        gs.render_tiles(render, (my_tiles + spare)[:capacity], fb.data_ptr(), stream)
        torch.cuda.synchronize(device)
        presize_ms = (time.perf_counter() - t1) * 1e3

    first = {}

    def step():
        nonlocal my_tiles
        cold = not first
        if cold:
            torch.cuda.synchronize(device)
            first["t0"] = time.perf_counter()
        st = step_()
        if cold:
            _t = time.perf_counter()
            torch.cuda.synchronize(device)
            if os.environ.get("FJ_BENCH_TRACE"):
                sys.stderr.write("bench trace: cold step %.1f ms before the closing synchronize, %.1f ms in it\n" % ((_t - first["t0"]) * 1e3, (time.perf_counter() - _t) * 1e3))
            first["ms"] = (time.perf_counter() - first["t0"]) * 1e3
            first["device_ms"] = st.total_ms if st is not None else None
            # the core's first call renders in cold-start batches (a small work arena); the SECOND call grows the arena to its steady-state
            # size -- one untimed frame here, so that neither the warm-up count nor the timed steps see the allocation
            # (FJGPU_COLD_START=0, the counter children: one call, one whole-frame arena, as before the policy)
            if os.environ.get("FJGPU_COLD_START", "1") != "0":
                _t = time.perf_counter()
                st = step_()
                torch.cuda.synchronize(device)
                first["second_ms"] = (time.perf_counter() - _t) * 1e3
        return st

    def step_():
        nonlocal my_tiles
        if dry:
            # ranks share ONE device here: they render one after the other
            st = None
            for turn in range(world):
                if turn == rank:
                    st = gs.render_tiles(render, my_tiles, fb.data_ptr(), stream)
                    torch.cuda.synchronize(device)
                dist.barrier()
        else:
            _t = time.perf_counter()
            st = gs.render_tiles(render, my_tiles, fb.data_ptr(), stream)
            if os.environ.get("FJ_BENCH_TRACE"):
                torch.cuda.synchronize(device)
                sys.stderr.write("bench trace: render_tiles %.1f ms wall, %.1f ms device\n" % ((time.perf_counter() - _t) * 1e3, st.total_ms))
        adapting = balance.adapting
        frame = fjdist.gather_frame(fb, n_tiles, render.tile_w, render.tile_h, rank, world, rects=tile_rects,
                                    lists=balance.lists, capacity=capacity)
        if rank == 0:
            _t = time.perf_counter()
            host_fb.copy_(frame, non_blocking=False)       # framebuffer resident in host memory
            if os.environ.get("FJ_BENCH_TRACE"):
                sys.stderr.write("bench trace: frame copy to host %.1f ms\n" % ((time.perf_counter() - _t) * 1e3))
        if adapting:
            # the ranks' render times -> the next frame's deal (the same arithmetic on every rank; what follows the renders --
            # exchange, scatter, the frame's copy to the host -- waits for the slowest rank whoever that is)
            balance.update(fjdist.share_times(st.total_ms, rank, world, device))
            my_tiles = balance.lists[rank]
        return st

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    if world > 1 and rank == 0 and os.environ.get("FJGPU_VERBOSE"):
        sys.stderr.write("bench: %d ranks, backend %s, %d of %d tiles on rank 0\n" % (world, dist.get_backend(), len(my_tiles), n_tiles))

    # The feedback deal converges BEFORE the timed region: its frames (each ends with an exchange of the ranks' times, every
    # second one with a re-deal) are untimed, so the K timed steps all run the final deal and nothing but the frame's own
    # exchange sits inside them (adapting deals mixed into ms_per_step).  The line says how many there were.
    balance_frames_untimed = 0
    while world > 1 and balance.adapting:
        step()
        balance_frames_untimed += 1
    closed = False
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    stats = []
    for _ in range(args.steps):
        stats.append(step())
    sync()
    elapsed = time.perf_counter() - t0

    # One more, UNTIMED frame in the counting instantiation of the traversal kernels: the event
    # counters (nodes / triangles / instance boxes / rays) cost registers and issue slots, so
    # the timed frames run without them.  The frame is deterministic: the counts of this frame
    # are the counts of every timed frame; kernel times come from the timed frames only.
    gs.set_option("count_nodes", 1)
    counted = step()
    sync()
    gs.set_option("count_nodes", 0)

    # N > 1: one more UNTIMED frame taken apart -- every rank's render on its own clock, then the exchange (pack, gather over RCCL, scatter,
    # the frame's copy to the host) between two barriers: the two numbers a scaling curve is read with (--scale prints them per N)
    multi = None
    if world > 1:
        sync()
        render_ms = 0.0
        for turn in range(world if dry else 1):          # (--dry-ranks: the ranks share one GPU and render in turns)
            if not dry or turn == rank:
                t1 = time.perf_counter()
                gs.render_tiles(render, my_tiles, fb.data_ptr(), stream)
                torch.cuda.synchronize(device)
                render_ms = (time.perf_counter() - t1) * 1e3
            if dry:
                dist.barrier()
        by_rank = fjdist.share_times(render_ms, rank, world, device)
        sync()
        t1 = time.perf_counter()
        frame = fjdist.gather_frame(fb, n_tiles, render.tile_w, render.tile_h, rank, world, rects=tile_rects, lists=balance.lists, capacity=capacity)
        if rank == 0:
            host_fb.copy_(frame, non_blocking=False)
        sync()
        multi = {"render_ms_by_rank": [float(v) for v in by_rank], "slowest_rank_render_ms": float(max(by_rank)),
                 "exchange_ms": (time.perf_counter() - t1) * 1e3,
                 "note": "one untimed frame after the timed ones: each rank's render_tiles on its own clock; exchange = pack + gather + scatter + D2H between barriers"}

    # ---------------- aggregate over ranks
    rays_local = float(sum(s.rays.total() for s in stats))
    trace_ms_local = float(sum(s.trace_ms for s in stats))
    alg_bytes_local = float(algorithmic_bytes(counted, s_node_closest, s_prim, s_node_walk)) * len(stats)
    launches_local = float(sum(s.trace_launches for s in stats))
    agg = torch.tensor([rays_local, alg_bytes_local, launches_local, elapsed, trace_ms_local], dtype=torch.float64, device=device)
    mx = agg.clone()
    if dist.is_initialized():
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    total_rays = float(agg[0])
    elapsed_max = float(mx[3])

    # HBM this rank holds now (scene + work arena + framebuffers; the arena only grows, so this is the peak): the driver's view
    # (hipMemGetInfo through torch) and the core's own books
    free_b, total_b = torch.cuda.mem_get_info(device)
    hbm_use = {"device_used_bytes": int(total_b - free_b), "device_total_bytes": int(total_b),
               "scene_bytes": int(gs.query("scene_bytes")), "work_arena_bytes": int(gs.query("work_bytes")),
               "note": "rank 0's GPU after the timed frames; device_used counts every allocation on the GPU (this process's HIP / "
                       "torch / RCCL contexts and the PMC children are gone by then)"}
    if rank == 0:
        s0 = stats[-1]
        per = {k: int(getattr(s0.rays, k)) for k in ("camera", "shadow", "diffuse", "reflect", "refract")}
        # ---- roofline of the DOMINANT KERNEL on rank 0: the traversal kernel with the largest share of the
        # frame by MEASURED time (HIP events of exactly its launches in the timed frames)
        nf = len(stats)
        shadow_name = "k_shadow_anyhit" if gs.query("lean_anyhit") else ("k_shadow_anyhit_curves" if gs.query("curve_anyhit") else "k_shadow_trace")
        closest_name = {1: "k_trace_closest_phased", 4: "k_trace_closest_flat"}.get(int(gs.query("closest_kernel")), "k_trace_closest")
        cand = {
            shadow_name: {"ms": float(sum(s.shadow_walk_ms for s in stats)), "launches": float(sum(s.shadow_walk_launches for s in stats)),
                          "alg": float(counted.shadow_nodes * s_node_walk + counted.shadow_prims * s_prim + counted.shadow_insts * S_INST +
                                       counted.shadow_traversed * S_SHADOW) * nf,
                          "rays": float(counted.shadow_traversed) * nf,
                          "streamed_bytes": float(counted.shadow_traversed) * S_SHADOW * nf},
            closest_name: {"ms": float(sum(s.closest_ms - s.sort_ms for s in stats)), "launches": float(sum(s.closest_launches for s in stats)),
                           "alg": float(closest_algorithmic_bytes(counted, s_node_closest, s_prim)) * nf,
                           "rays": float(counted.rays_traced - counted.rays.shadow) * nf,
                           "streamed_bytes": float(counted.rays_traced - counted.rays.shadow) * S_RAY_IN * nf},
        }
        kname = max(cand, key=lambda k: cand[k]["ms"])
        K = cand[kname]
        walk_ms, walk_nl, walk_alg = K["ms"], K["launches"], K["alg"]
        frame_ms_sum = max(1e-9, float(sum(s.total_ms for s in stats)))
        # the whole traversal side (closest-hit walk + light loop + shadow walk), algorithmic figure only
        alg = float(algorithmic_bytes(counted, s_node_closest, s_prim, s_node_walk)) * nf
        tms = float(sum(s.trace_ms for s in stats))
        nl = float(sum(s.trace_launches for s in stats))
        # Hardware counters of the SAME build, measured now by rocprofv3 child passes of this script
        # (pmc_passes): HBM-side traffic = cal x FETCH_SIZE + WRITE_SIZE (both in KB), cal = the factor
        # scripts/fetch_calibration.py measured on this chip for random 64-byte record gathers
        # (profiles/r*_fetch_size_calibration.json; MI355X_MICROARCH.md "HBM": calibrate on your own pattern);
        # VALU issue = ACTIVE_INST_VALU quad-cycles / (SIMDs x elapsed quad-cycles); lane efficiency =
        # THREAD_CYCLES_VALU / (64 x ACTIVE_INST_VALU).
        cal = fetch_calibration()
        traffic, pmc, issue, hbm = None, None, None, None
        if world == 1 and not args.no_pmc and args.as_rank_of <= 1:
            # the counter children render the same frame in processes of their own: this process's work arena (110 GB at the headline size)
            # goes back to the driver first, so that a child finds the memory the timed frames found and cuts its frame into the same batches
            gs.set_option("release_work", 1)
            pmc = pmc_passes(args, kname)
        launches_per_frame = walk_nl / nf if nf else 1.0
        avg_ms = walk_ms / walk_nl if walk_nl else None
        if pmc and "TCC_EA0_RDREQ_sum" in pmc and "WRITE_SIZE" in pmc and launches_per_frame and avg_ms:
            rd, rd32 = pmc["TCC_EA0_RDREQ_sum"], pmc.get("TCC_EA0_RDREQ_32B_sum", 0.0)
            read_low = 64.0 * (rd - rd32) + 32.0 * rd32            # every request a 64-byte one (= FETCH_SIZE)
            read_high = 128.0 * (rd - rd32) + 32.0 * rd32          # every request a 128-byte one
            # coalesced record streams of this kernel per frame (128-byte requests, tallied at 64 B in the lower bound)
            streamed = K["streamed_bytes"] / nf
            read_bytes = read_low + min(0.5 * streamed, read_low)
            how = ("64 B x TCC_EA0_RDREQ (lower bound: right for the 64-byte node gathers, calibrated) + half of the %.1f GB per frame "
                   "the kernel reads as coalesced streams (128-byte requests)" % (streamed / 1e9))
            # (per frame in the child -> per launch of this run)
            traffic = (read_bytes + pmc["WRITE_SIZE"] * 1024.0) / launches_per_frame
            gbps = traffic / (avg_ms * 1e-3) / 1e9
            hbm = {"GBps": gbps, "frac_of_peak": gbps / HBM_PEAK_GBPS, "traffic_bytes_per_launch": traffic,
                   "read_bytes_per_frame": read_bytes, "read_bytes_from": how,
                   "read_bytes_bounds_per_frame": [read_low, read_high],
                   "frac_of_peak_bounds": [(read_low + pmc["WRITE_SIZE"] * 1024.0) / launches_per_frame / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                           min(1.0, (read_high + pmc["WRITE_SIZE"] * 1024.0) / launches_per_frame / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS)],
                   "read_requests_per_frame": {"all": rd, "32B": rd32},
                   "launches_in_the_counter_child": pmc.get("_launches"), "launches_per_frame_here": launches_per_frame,
                   "FETCH_SIZE_KB_per_frame_as_rocprofv3_reports_it": rd * 64.0 / 1024.0, "WRITE_SIZE_KB_per_frame": pmc["WRITE_SIZE"],
                   "calibration": cal,
                   "traffic_over_algorithmic": traffic / (walk_alg / walk_nl) if walk_alg else None,
                   "note": "bytes the L2s requested from the fabric (Infinity Cache hits are in it: an upper bound of DRAM traffic); "
                           "WRITE_SIZE as reported (uncalibrated; 3 % of this kernel's traffic)"}
        if pmc and pmc.get("SQ_WAVE_CYCLES") and pmc.get("SQ_WAVES") and pmc.get("SQ_ACTIVE_INST_VALU"):
            props = torch.cuda.get_device_properties(device)
            simds = props.multi_processor_count * 4
            # persistent waves live for the whole launch: wave cycles / waves per launch = the time (in the
            # counter's own unit) the kernel's launches of one frame were resident, summed over the launches
            elapsed_q = pmc["SQ_WAVE_CYCLES"] * max(pmc.get("_launches", 1.0), 1.0) / pmc["SQ_WAVES"]
            # (waves that leave a launch early make elapsed_q an underestimate: the share is capped at 1)
            busy = min(1.0, pmc["SQ_ACTIVE_INST_VALU"] / (simds * elapsed_q))
            lane = pmc["SQ_THREAD_CYCLES_VALU"] / (64.0 * pmc["SQ_ACTIVE_INST_VALU"])
            issue = {"valu_busy": busy, "lane_efficiency": lane, "useful_valu_issue": busy * lane,
                     "waves_waiting_on_memory": pmc.get("SQ_WAIT_ANY", 0.0) / pmc["SQ_WAVE_CYCLES"],
                     "waves_stalled_at_issue": pmc.get("SQ_WAIT_INST_ANY", 0.0) / pmc["SQ_WAVE_CYCLES"],
                     "valu_wave_instructions_per_frame": pmc.get("SQ_INSTS_VALU"), "simds": simds,
                     "note": "one VALU per SIMD, 4 cycles per wave instruction (FP64 and packed FP32 alike): busy = share of "
                             "the chip's VALU issue cycles this kernel used, lane_efficiency = active lanes per issued instruction"}
        # `frac` is a utilisation of a hardware peak measured by counters, <= 1 by construction: the HBM-side
        # bandwidth of this kernel over the 8 TB/s peak.  Which resource is nearer its roof is said beside it
        # (`binding_resource`: "valu" when the VALU issue utilisation is the larger fraction).  The algorithmic
        # bytes of SURVEY 8(d) are reported under `algorithmic` and never priced against the HBM peak.
        bound, frac, achieved, peak, unit = "hbm", None, None, HBM_PEAK_GBPS, "GB/s"
        frac_source = "hardware counters (rocprofv3 --pmc child passes of this run)"
        if hbm:
            achieved = min(hbm["GBps"], HBM_PEAK_GBPS)
            frac = achieved / HBM_PEAK_GBPS
        elif walk_alg and avg_ms:
            # no counter passes in this run (rocprofv3 missing, --no-pmc, a rank-share run or world > 1): the contract's fields
            # stay numbers -- an ESTIMATE from this run's algorithmic bytes and the traffic / algorithmic ratio the counters gave
            # for this kernel family last time they ran (profiles/r03_bench_dragon1080p.json: 0.33) -- and say so
            traffic = TRAFFIC_OVER_ALGORITHMIC_LAST_MEASURED * walk_alg / walk_nl
            achieved = min(traffic / (avg_ms * 1e-3) / 1e9, HBM_PEAK_GBPS)
            frac = achieved / HBM_PEAK_GBPS
            frac_source = ("UNCALIBRATED estimate: algorithmic bytes of this run x %.2f (the traffic / algorithmic ratio of the last counter run, "
                           "profiles/r03_bench_dragon1080p.json); the counter passes did not run" % TRAFFIC_OVER_ALGORITHMIC_LAST_MEASURED)
        # The yardstick for THIS access pattern: the walks gather 64-byte records, and a kernel that does nothing else reaches ~56 G requests/s
        # (3.5 TB/s) at the scene's footprint on this part, not the 8 TB/s of a stream (profiles/r06_memory_system_notes.txt).  Measured live by the
        # calibration tool on an array of the scene's size (rounded up to a power of two, >= 256 MiB), and set against the requests the dominant
        # kernel's L2s sent to the fabric per second (TCC_EA0_RDREQ / launch time).  Reported BESIDE `frac`, which stays the contract's HBM fraction.
        gather = None
        if hbm and avg_ms:
            try:
                gib = 0.25
                while gib * (1 << 30) < float(gs.query("scene_bytes")) and gib < 64:
                    gib *= 2
                exe = os.path.join(ROOT, "fujiyama-renderer_amd", "bin", "hbm_gather_calib")
                cj = json.loads(subprocess.run([exe, str(gib), "64"], cwd="/tmp", stdout=subprocess.PIPE, text=True, timeout=120).stdout.strip().splitlines()[-1])
                ceil_req = cj["kernels"]["k_calib_gather64"]["GBps_needed"] / 64.0            # G requests / s
                ach_req = hbm["read_requests_per_frame"]["all"] / launches_per_frame / (avg_ms * 1e-3) / 1e9
                gather = {"achieved_Grequests_per_s": ach_req, "ceiling_Grequests_per_s": ceil_req, "frac_of_gather_ceiling": ach_req / ceil_req,
                          "calibration_array_GiB": gib, "ceiling_GBps_64B": cj["kernels"]["k_calib_gather64"]["GBps_needed"],
                          "stream_GBps_same_tool": cj["kernels"]["k_calib_stream"]["GBps_needed"],
                          "note": "requests past the L2 per second of the dominant kernel against what a pure random 64-byte gather reaches on this GPU at the "
                                  "scene's footprint (bin/hbm_gather_calib, run now); includes the kernel's coalesced queue reads, so it overstates the gathers"}
            except Exception as e:  # noqa: BLE001
                gather = {"error": str(e)}
        binding = None
        if issue and frac is not None:
            binding = "valu" if issue["valu_busy"] > frac else "hbm"
        roof = {"bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": frac,
                "traffic": traffic, "traffic_unit": "bytes per launch (fabric side of L2, calibrated FETCH_SIZE + WRITE_SIZE)",
                "kernel": kname, "launches": int(walk_nl), "avg_launch_ms": avg_ms,
                "share_of_frame": walk_ms / frame_ms_sum,
                "picked_by": "largest measured HIP-event time among the traversal kernels",
                "binding_resource": binding, "frac_source": frac_source, "uncalibrated": hbm is None,
                "hbm_counters": hbm, "valu_issue": issue, "gather_yardstick": gather,
                "algorithmic": {"bytes_per_launch": walk_alg / walk_nl if walk_nl else None,
                                "GBps": walk_alg / (walk_ms * 1e-3) / 1e9 if walk_ms > 0 else None,
                                "bytes_per_ray": walk_alg / max(1.0, K["rays"]),
                                "note": "SURVEY 8(d): event counts of this kernel (counting frame) x record sizes = the bytes the walk "
                                        "must READ; caches serve most of them, so this is data turned over, not an HBM fraction",
                                "record_bytes": {"node_closest_walk": s_node_closest, "node_shadow_walk": s_node_walk, "tri": s_prim,
                                                 "instance_box": S_INST, "ray_in": S_RAY_IN, "hit_out": S_HIT_OUT, "shadow_ray": S_SHADOW},
                                "all_traversal_kernels": {"kernel": closest_name + "+k_shadow_cull+" + shadow_name, "launches": int(nl),
                                                          "GBps": alg / (tms * 1e-3) / 1e9 if tms > 0 else 0.0,
                                                          "bytes_per_frame": alg / nf,
                                                          "bytes_per_ray": alg / max(1.0, float(counted.rays_traced) * nf)}},
                "kernel_ms_per_frame_rank0": {closest_name: cand[closest_name]["ms"] / nf,
                                              "ray_sort": float(sum(s.sort_ms for s in stats)) / nf,
                                              "k_shadow_cull": float(sum(s.light_loop_ms for s in stats)) / nf,
                                              shadow_name: cand[shadow_name]["ms"] / nf,
                                              "k_shade": float(sum(s.shade_ms for s in stats)) / nf,
                                              "k_gen_camera": float(sum(s.gen_ms for s in stats)) / nf,
                                              "k_resolve": float(sum(s.resolve_ms for s in stats)) / nf}}
        # rays that actually walk a BLAS (camera / reflect / ... closest-hit rays + the shadow rays
        # that survive the instance-box cull), next to the SlTrace-event count of the metric
        walked = float(sum((s.rays.total() - s.rays.shadow) for s in stats)) + float(counted.shadow_traversed) * nf
        out = {
            "metric": "Mray/s primary+secondary (and ms/frame) at 1920x1080 64spp",
            "value": total_rays / elapsed_max / 1e6,
            "unit": "Mray/s",
            "n_gpus": world, "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1, "backend": backend,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3,
            # the process's FIRST frame, cold: work-buffer allocation, first launches, the exchange's first call (steady-state
            # frames follow; with N > 1 a rank's first render is the untimed one that sizes its buffers: presize_ms)
            "first_frame_ms": presize_ms if presize_ms is not None else first.get("ms"),
            "first_frame_device_ms": first.get("device_ms"),
            # ... and its second: the call in which the work arena grows from the cold-start size to the steady-state one (untimed, before the warm-up)
            "second_frame_ms": first.get("second_ms"),
            "peak_hbm_bytes": hbm_use,
            "traversed_Mray_s": walked / elapsed_max / 1e6 if world == 1 else None,
            "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s-class scene, %dx%d, %dx%d spp, tile %dx%d, %d tiles, 32 point lights"
                                   % (args.workload, render.xres, render.yres, render.rate_x, render.rate_y,
                                      render.tile_w, render.tile_h, n_tiles),
                       "mesh": args.mesh or {"dragon": "dragon", "buddhas": "buddha", "teapot": "teapot",
                                             "furry": "furbunny"}.get(args.workload, args.workload),
                       "rays_per_frame_rank0": per, "parallelism": "tiles%%%d" % world, "tile_deal": args.deal,
                       "tile_balance": {"frames": args.balance_frames, "untimed_frames_before_warmup": balance_frames_untimed,
                                        "tiles_per_rank": [len(l) for l in balance.lists],
                                        "slowest_rank_ms_by_deal": balance.history} if world > 1 else None,
                       "blas_build": ("host binned SAH", "device clustering (PLOC)", "device radix tree (LBVH)")[args.device_build],
                       "prepare_seconds": prep_seconds,
                       "counters_counting_frame_rank0": {"nodes": int(counted.nodes_visited), "prims": int(counted.prims_tested),
                                                         "insts": int(counted.insts_tested), "traced": int(counted.rays_traced),
                                                         "shadow_traversed": int(counted.shadow_traversed)},
                       "ms_last_frame_rank0": {"trace": s0.trace_ms, "shade": s0.shade_ms, "gen": s0.gen_ms,
                                               "resolve": s0.resolve_ms, "total": s0.total_ms,
                                               "ray_sort": s0.sort_ms, "rays_sorted": int(s0.rays_sorted)}},
            "roofline": roof,
        }
        if dry:
            # the assembled frame of the N-rank job against this rank's own render of the whole frame
            whole = torch.zeros_like(fb)
            gs.render_tiles(render, list(range(n_tiles)), whole.data_ptr(), stream)
            torch.cuda.synchronize(device)
            a, b = host_fb.numpy(), whole.cpu().numpy()
            rel = float((np.abs(a - b) / np.maximum(np.abs(b), 1e-3)).max())
            out["dry_ranks"] = {"ranks": world, "backend": dist.get_backend(), "max_rel_err_vs_single_render": rel, "ok": bool(rel <= 1e-5),
                                "note": "all ranks on cuda:0, rendering in turns: a code-path check, its timings mean nothing"}
        if multi is not None and dry:
            out["multi_gpu"] = multi                     # (the self-check of a dry run is `dry_ranks` above)
        elif multi is not None:
            # self-check of the N-GPU frame: the assembled frame (host_fb, the exchange above) against THIS rank's own render of the whole frame.
            # The other ranks are not held up: every collective of the run lies behind (the closing barrier waits for rank 0 either way).
            whole = torch.zeros_like(fb)
            gs.render_tiles(render, list(range(n_tiles)), whole.data_ptr(), stream)
            torch.cuda.synchronize(device)
            a, b = host_fb.numpy(), whole.cpu().numpy()
            rel = float((np.abs(a - b) / np.maximum(np.abs(b), 1e-3)).max())
            multi["self_check"] = {"max_rel_err_vs_rank0_whole_render": rel, "ok": bool(rel <= 1e-5)}
            out["multi_gpu"] = multi
            if not multi["self_check"]["ok"]:
                sys.stderr.write("bench.py: the frame assembled from %d ranks differs from rank 0's own render (max rel err %.3g)\n" % (world, rel))
        if world == 1 and args.rank_costs > 1:
            # per-rank cost of an N-GPU job, every rank's tile share timed on this one GPU
            # (tile t -> rank t % N): the slowest share bounds the N-GPU frame before the gather
            costs, parts = [], []
            G = args.rank_costs
            deal = fjdist.deal_tiles(n_tiles, G, tiles_per_row, args.deal)
            for rk in range(G):
                tiles_r = deal[rk]
                gs.render_tiles(render, tiles_r, fb.data_ptr(), stream)      # warm
                torch.cuda.synchronize(device)
                each = []
                for _ in range(5):                                            # (the median of five: one frame in ten is an outlier of +1..2 ms)
                    t1 = time.perf_counter()
                    rst = gs.render_tiles(render, tiles_r, fb.data_ptr(), stream)
                    torch.cuda.synchronize(device)
                    each.append((time.perf_counter() - t1) * 1e3)
                costs.append(sorted(each)[len(each) // 2])
                parts.append({"closest": rst.closest_ms, "light_loop": rst.light_loop_ms, "shadow_walk": rst.shadow_walk_ms,
                              "shade": rst.shade_ms, "gen": rst.gen_ms, "resolve": rst.resolve_ms, "device_total": rst.total_ms,
                              "launches": int(rst.trace_launches), "batches": int(rst.batches)})
            slow = max(range(len(costs)), key=lambda k: costs[k])
            # what an N-GPU frame adds to its slowest rank: packing the rank's tiles into a slab, the exchange, the
            # scatter into the frame, the frame's D2H -- all but the exchange itself measured on this GPU (the 7 slabs
            # of <= 4.1 MB each arrive over separate xGMI links: priced at 50 GB/s per link, a third of the link rate)
            tail_ms, slab_bytes = measure_tail(G, deal)
            xgmi_ms = slab_bytes / 50e9 * 1e3
            # ... and the feedback deal (TileBalance) played through on this GPU: every frame of an N-GPU job is one round of
            # timing each rank's current share (one frame each, as the job would see it)
            bal = None
            if args.balance_frames > 0:
                tb = fjdist.TileBalance(deal, fjdist.slab_capacity(n_tiles, G), frames=args.balance_frames)

                gs.render_tiles(render, list(range(tb.capacity)), fb.data_ptr(), stream)      # (buffers for the longest list, as the job does)
                torch.cuda.synchronize(device)

                def share_ms(tiles_r, reps):
                    each = []
                    for _ in range(reps):
                        each.append(gs.render_tiles(render, tiles_r, fb.data_ptr(), stream).total_ms)     # (what the job's ranks report)
                    return sorted(each)[len(each) // 2]
                while tb.adapting:
                    tb.update([share_ms(tb.lists[rk], 1) for rk in range(G)])
                final = []
                for rk in range(G):
                    each = []
                    for _ in range(5):
                        t1 = time.perf_counter()
                        gs.render_tiles(render, tb.lists[rk], fb.data_ptr(), stream)
                        torch.cuda.synchronize(device)
                        each.append((time.perf_counter() - t1) * 1e3)
                    final.append(sorted(each)[2])
                frame_ms = max(final) + tail_ms + xgmi_ms
                bal = {"frames": args.balance_frames, "slowest_rank_ms_by_frame": tb.history, "per_rank": final,
                       "tiles_per_rank": [len(l) for l in tb.lists], "projected_frame_ms": frame_ms,
                       "projected_speedup": out["ms_per_step"] / frame_ms}
            out["config"]["rank_costs_ms"] = {"ranks": G, "deal": args.deal, "balanced": bal, "per_rank": costs, "max": max(costs),
                                              "speedup_before_gather": out["ms_per_step"] / max(costs),
                                              "pack_unpack_d2h_ms_measured": tail_ms, "xgmi_exchange_ms_estimated": xgmi_ms,
                                              "projected_frame_ms": max(costs) + tail_ms + xgmi_ms,
                                              "projected_speedup": out["ms_per_step"] / (max(costs) + tail_ms + xgmi_ms),
                                              "slowest_rank_kernels_ms": parts[slow]}
        if world == 1 and args.cpu_tiles != 0:
            # bounded CPU sample: a block of tiles in the middle of the frame, two tiles per
            # host core (the reference hands whole tiles to its worker threads)
            nx = -(-render.xres // render.tile_w)
            ny = -(-render.yres // render.tile_h)
            # Default sample: ONE tile per hardware thread of the box, so that every worker thread the reference spawns
            # (use_max_thread) has a tile -- the stated baseline is "the same box's host cores", all of them.  (Round 4 capped
            # the sample at 64 tiles = 64 busy threads of 256; 2 tiles per thread, 512 tiles, took 142 s of wall time on the
            # GPU box -- the reference allocates per ray and its heap traffic scales badly -- which no default run can afford.)
            # Workloads with a PathtracingShader keep at most 64 tiles: the plugin owns rng[64], one generator per worker
            # thread id (shaders/pathtracing_shader/pathtracing_shader.cc), so 64 is the most threads it may run with.
            pt_cap = args.workload in ("cornell",)
            want = args.cpu_tiles if args.cpu_tiles > 0 else (min(2 * (os.cpu_count() or 1), 64) if pt_cap else (os.cpu_count() or 1))
            want = max(1, min(want, nx * ny))
            bw = max(1, min(nx, 32, want))
            bh = max(1, min(ny, -(-want // bw)))
            x0, y0 = (nx - bw) // 2, (ny - bh) // 2
            sample = [(y0 + j) * nx + (x0 + i) for j in range(bh) for i in range(bw)]
            sfb = torch.zeros_like(fb)
            sst = gs.render_tiles(render, sample, sfb.data_ptr(), stream)
            out["cpu_baseline"] = cpu_baseline(args, scene_text, render, scene_ptr, sample, float(sst.rays.total()))
            out["cpu_baseline"]["gpu_ms_same_sample"] = sst.total_ms
            # ... and the restatement (the faster CPU code: no per-ray heap traffic) on a sample with the frame's own mix of tiles
            out["cpu_baseline_port"] = cpu_baseline_port(args, render, scene_ptr, n_tiles)
        if world == 1 and args.cpu_tiles != 0 and not args.no_e2e and args.as_rank_of <= 1:
            gs.close()          # (its ~100 GB of work buffers go back first: the binary is a process of its own on the same GPU)
            closed = True
            out["scene_binary_end_to_end"] = scene_binary_end_to_end(scene_text)
        print(json.dumps(out))
    if not closed:
        gs.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
