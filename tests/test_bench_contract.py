"""bench.py keeps its output contract: ONE JSON line with the driver's keys, a per-kernel
`roofline` object and a `cpu_baseline` object (here on the small teapot workload, seconds)."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "teapot", "--steps", "2", "--warmup", "1"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                      # exactly one line on stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["value"] > 0 and d["ms_per_step"] > 0 and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "launches", "avg_launch_ms"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
    # an HBM fraction <= 1 by construction, never null: counter-measured (rocprofv3 child passes of the same run) where
    # rocprofv3 exists, else an estimate that says so (roofline.uncalibrated)
    assert r["traffic"] is not None and r["frac"] is not None and r["achieved"] is not None
    assert 0 < r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    if shutil.which("rocprofv3"):
        assert r["uncalibrated"] is False, "the rocprofv3 counter passes did not run: " + r["frac_source"]
        assert r["hbm_counters"]["read_bytes_per_frame"] > 0 and r["valu_issue"]["valu_busy"] <= 1.0 + 1e-6
        assert r["binding_resource"] in ("hbm", "valu")
        # the counter child renders ONE production frame (no cold-start frame beside it): its sums are per frame of this run.  (It may cut
        # that frame into more batches than this process did -- the parent's arena is still allocated while it runs -- but on this small
        # workload both render it in one.)
        assert r["hbm_counters"]["launches_in_the_counter_child"] == r["hbm_counters"]["launches_per_frame_here"], r["hbm_counters"]
        assert d["first_frame_ms"] > 0 and d["second_frame_ms"] > 0
    else:
        assert r["uncalibrated"] is True and "UNCALIBRATED" in r["frac_source"]
    # the dominant kernel by measured time, named; the SURVEY 8(d) bytes are kept apart
    assert r["kernel"] in ("k_shadow_anyhit", "k_shadow_anyhit_curves", "k_shadow_trace", "k_trace_closest", "k_trace_closest_phased", "k_trace_closest_flat")
    assert r["launches"] >= 1 and r["avg_launch_ms"] > 0 and r["algorithmic"]["bytes_per_launch"] > 0
    assert max(r["kernel_ms_per_frame_rank0"], key=lambda k: r["kernel_ms_per_frame_rank0"][k] if k.startswith("k_trace") or k.startswith("k_shadow_a") or k.startswith("k_shadow_t") else -1) == r["kernel"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1


@pytest.mark.gpu
def test_bench_multi_rank_code_path_on_one_gpu():
    """bench.py --dry-ranks 3: the N > 1 path of the driver's contract (process group, tile deal and its feedback, per-rank
    render, the one gather, scatter, D2H) run as three torch.distributed ranks that share cuda:0 over gloo; the
    assembled frame equals a single render and the line keeps its contract fields with n_gpus = 3"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "teapot", "--dry-ranks", "3", "--steps", "2", "--warmup", "1"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["scaling"] == "strong" and d["value"] > 0
    assert d["dry_ranks"]["ok"] and d["dry_ranks"]["ranks"] == 3, d["dry_ranks"]
    assert d["config"]["parallelism"] == "tiles%3"
    # the feedback deal ran (a re-deal after every second frame; the last frames are re-dealt ones) and kept every tile
    tb = d["config"]["tile_balance"]
    assert tb["frames"] == 8 and len(tb["slowest_rank_ms_by_deal"]) >= 1 and len(tb["tiles_per_rank"]) == 3
    # the frame taken apart (what --scale prints per N): every rank's render time and the exchange between two barriers
    mg = d["multi_gpu"]
    assert len(mg["render_ms_by_rank"]) == 3 and mg["slowest_rank_render_ms"] == max(mg["render_ms_by_rank"]) > 0 and mg["exchange_ms"] > 0


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_over_rccl():
    """`python bench.py --gpus N` with no launcher around it re-runs itself as N ranks under torch.distributed.run with the nccl
    (= RCCL) backend -- what the driver's scaling run needs from a bare `--gpus N`.  On the one GPU of
    this box: `--launcher` forces that path for N = 1, so the RCCL communicator is created, an all-reduce and the frame
    exchange's gather run on real hardware, and the line says which backend carried it; asking for more GPUs than the node
    has fails loudly instead of rendering on one."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "teapot", "--gpus", "1", "--launcher", "--steps", "2",
                        "--warmup", "1", "--cpu-tiles", "0", "--no-pmc"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["backend"] == "nccl" and d["rccl_ranks"] == 1 and d["value"] > 0
    assert d["first_frame_ms"] > 0 and d["peak_hbm_bytes"]["work_arena_bytes"] > 0
    import torch
    more = torch.cuda.device_count() + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "teapot", "--gpus", str(more), "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "GPU(s) visible" in (p.stderr + p.stdout), (p.returncode, p.stderr[-500:])
    assert not [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]          # no line from a run that did not happen


def test_bench_without_a_gpu_fails_loudly():
    """no CPU fallback anywhere: on a machine without a GPU bench.py (with or without --gpus N) stops with a message, prints no line"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    for extra in ([], ["--gpus", "2"]):
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "teapot", "--steps", "1", "--warmup", "0"] + extra,
                           cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert p.returncode != 0 and "needs a GPU" in (p.stderr + p.stdout), (p.returncode, p.stderr[-500:])
        assert not [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]


def _import_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("fj_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_scale_sweep_plumbing_without_a_gpu(capsys):
    """`bench.py --scale 1,2,4,8`: the list is parsed, every N gives ONE compact line with the fields a scaling curve
    is read with, a failing point is reported and counted, and the last line carries the speed-ups -- with the per-N runner injected (no GPU)."""
    bench = _import_bench()
    assert bench.scale_points("1,2, 4,8") == [1, 2, 4, 8]
    for bad in ("", "0,1", "a"):
        with pytest.raises(ValueError):
            bench.scale_points(bad)

    def full_line(n):
        return {"metric": "Mray/s primary+secondary (and ms/frame) at 1920x1080 64spp", "value": 24000.0 * n * 0.8, "unit": "Mray/s", "n_gpus": n,
                "rccl_ranks": n, "backend": "nccl" if n > 1 else None, "steps": 3, "warmup": 1, "ms_per_step": 114.0 / (n * 0.8),
                "config": {"workload": "dragon-class scene", "tile_balance": {"tiles_per_rank": [2040 // n] * n} if n > 1 else None},
                "multi_gpu": {"render_ms_by_rank": [100.0 / n] * n, "slowest_rank_render_ms": 100.0 / n, "exchange_ms": 0.9,
                              "self_check": {"max_rel_err_vs_rank0_whole_render": 0.0, "ok": True}} if n > 1 else None}

    def run(n):
        if n == 4:
            return {"scale_point": 4, "error": "bench.py --gpus 4 exited with 1"}
        return bench.scale_line(n, full_line(n))

    class A:
        scale = "1,2,4,8"
    rc = bench.scale_parent(A(), run=run)
    lines = [json.loads(ln) for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")]
    assert rc == 1 and len(lines) == 5
    assert [ln.get("scale_point") for ln in lines[:4]] == [1, 2, 4, 8]
    one, two, four, eight, summ = lines
    assert one["n_gpus"] == 1 and one["slowest_rank_render_ms"] is None and one["self_check"] is None
    assert two["rccl_ranks"] == 2 and two["exchange_ms"] == 0.9 and two["self_check"]["ok"] and two["tiles_per_rank"] == [1020, 1020]
    assert "error" in four and eight["value"] > two["value"]
    assert summ["scale_summary"] and summ["points"] == [1, 2, 8] and summ["failed"] == [4] and summ["skipped"] == []
    assert summ["speedup_over_first"][0] == 1.0 and abs(summ["speedup_over_first"][2] - 8 * 0.8 / 0.8) < 1e-9


@pytest.mark.gpu
def test_scale_sweep_on_this_box():
    """--scale 1,2 on the GPUs this box has: N = 1 is measured (a bench line's worth), an N beyond the visible GPUs is reported as skipped, not
    faked; on a node with two GPUs the second point runs over RCCL and must pass its self-check"""
    import torch
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "teapot", "--steps", "2", "--warmup", "1", "--scale", "1,2"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 3
    assert lines[0]["scale_point"] == 1 and lines[0]["value"] > 0 and lines[0]["n_gpus"] == 1
    if torch.cuda.device_count() >= 2:
        assert lines[1]["rccl_ranks"] == 2 and lines[1]["self_check"]["ok"] and lines[1]["exchange_ms"] > 0
        assert lines[2]["points"] == [1, 2]
    else:
        assert "skipped" in lines[1] and lines[2]["points"] == [1] and lines[2]["skipped"] == [2]
