"""GPU suite (-m gpu): the HIP core, called through the C ABI (lib/libfjgpu.so,
lib/libfjscene.so), against
  * the CPU oracle on the same seeded inputs,
  * the golden frames / vectors produced by the compiled reference,
  * size-independent properties at BASELINE.json's full size.

Tolerances: traversal results (t, instance, primitive, barycentrics) and ray
counts are BIT-EXACT / equal; per-pixel RGBA is within 1e-4 relative (north_star),
denominator floored at 1e-3 for near-black pixels.  The residual (~4e-7) comes
from f32 accumulation order only: light sums are reduced as a tree and terms are
added to a sample in queue order, where the reference adds them sequentially.
"""
import os

import numpy as np
import pytest

import golden_io
import oracle_ffi
from fujiyama_renderer_amd import gpu, host, synth, workloads
from fujiyama_renderer_amd.fujiyama import SceneInterface

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4


def rel_err(a, ref):
    return np.abs(a - ref) / np.maximum(np.abs(ref), 1e-3)


def prepare(text):
    host.run_scene_text(text, deferred=True)
    return host.get_desc()


_last = {"adaptive": False}


def close_enough(a, st_a, b, st_b, adaptive, tol):
    """pixels within `tol` and identical ray counts.  Under the ADAPTIVE sampler a subdivision
    decision compares f32 sample values with the threshold, and the device's sums differ from a
    sequential sum in the last bits (order of the atomic adds): a corner spread within ~1e-7 of
    the threshold may decide differently -- one rectangle, a few samples, a few pixels.  Such a
    flip (never observed so far) is tolerated there; anything larger is not."""
    exact = (st_a is None or st_a == st_b) and float(rel_err(a, b).max()) <= tol
    if exact or not adaptive:
        return exact
    bad = int((rel_err(a, b) > tol).any(axis=2).sum())
    if st_a is None:          # two device renders of one frame: pixels only
        return bad <= max(16, a.shape[0] * a.shape[1] // 500)
    cam_a, cam_b = st_a["camera"], st_b["camera"]
    return abs(cam_a - cam_b) <= max(64, cam_b // 1000) and bad <= max(16, a.shape[0] * a.shape[1] // 500)


def render_both(text, threads=None):
    sp, rd = prepare(text)
    _last["adaptive"] = rd.sampler_type == 1
    gs = gpu.Scene(sp)
    gs.set_option("count_nodes", 1)        # the counting instantiation of the traversal kernels
    fb, st = gs.render_frame(rd)
    gs.set_option("count_nodes", 0)        # ... and the production one: same pixels
    fb2, _ = gs.render_frame(rd)
    assert np.array_equal(fb, fb2) or close_enough(fb, None, fb2, None, _last["adaptive"], 1e-6)
    gs.close()
    osc = oracle_ffi.OracleScene(sp)
    ref, rc = osc.render(rd, threads=threads)
    osc.close()
    return fb, st, ref, rc


def assert_parity(fb, st, ref, rc):
    """the device frame of the last render_both() against the oracle's: same SlTrace events per
    context, pixels within REL_TOL"""
    assert close_enough(fb, st.rays.as_dict(), ref, rc.as_dict(), _last["adaptive"], REL_TOL), (
        st.rays.as_dict(), rc.as_dict(), float(rel_err(fb, ref).max()))


from frame_cases import FRAMES as CASES  # noqa: E402  (incl. the adaptive grid sampler's frames)


@pytest.mark.parametrize("kw", [dict(res=(64, 48), spp=(3, 3), mesh="tiny"),
                                dict(res=(48, 32), spp=(2, 2), mesh="tiny", extra=(("max_diffuse_depth", (1,)),)),
                                dict(res=(48, 32), spp=(2, 2), mesh="tiny", objects=("bunny",),
                                     extra=(("max_reflect_depth", (2,)), ("max_refract_depth", (1,))))],
                         ids=["c4_cornell_64x48_3spp", "diffuse_depth_1", "glass_only_depth_limits"])
def test_pathtracing_matches_oracle(kw, asset_dir):
    """C4: PathtracingShader with the counter-based bounce stream (uid, path key): the same
    rays (per-context counts equal), the same pixels.  Also exercises the reference's
    mirrored instance box for the negatively scaled light blob (fjgpu_build.cc)."""
    fb, st, ref, rc = render_both(workloads.cornell(asset_dir, **kw))
    assert st.rays.as_dict() == rc.as_dict()
    assert rc.diffuse > 0 and rc.shadow == 0
    assert float(rel_err(fb, ref).max()) <= REL_TOL


@pytest.mark.parametrize("bits", [1, 4, 7])
def test_sorted_ray_queues_change_nothing(bits, asset_dir):
    """the ray-queue sort in front of the closest-hit walk (fjgpu_raysort.hip; secondary rays in
    (direction octant, origin cell) order, hits written back to the ray's own slot): same rays per
    context, same pixels as the oracle, for pathtracing (3 children per hit) and glass scenes; the
    statistics say that the rays went through it"""
    gpu.global_option("ray_sort", bits)
    gpu.global_option("ray_sort_min", 1)
    try:
        fb, st, ref, rc = render_both(workloads.cornell(asset_dir, res=(64, 48), spp=(3, 3), mesh="tiny"))
        assert st.rays.as_dict() == rc.as_dict()
        assert float(rel_err(fb, ref).max()) <= REL_TOL
        assert st.rays_sorted == rc.diffuse + rc.reflect + rc.refract and st.sort_ms > 0
        fb, st, ref, rc = render_both(workloads.teapot(asset_dir, res=(64, 64), spp=(2, 2)))
        assert st.rays.as_dict() == rc.as_dict()
        assert float(rel_err(fb, ref).max()) <= REL_TOL
        assert st.rays_sorted == rc.reflect + rc.refract
    finally:
        gpu.global_option("ray_sort", -1)
        gpu.global_option("ray_sort_min", 1 << 16)


@pytest.mark.parametrize("n,key_bits", [(1, 6), (63, 15), (64, 8), (4096, 15), (4097, 15), (70001, 6), (70001, 9), (262144 + 37, 15),
                                        (1 << 20, 18), (300017, 24), (300017, 30), (50000, 32), (3000003, 15)])
def test_ray_sort_is_a_stable_sort(n, key_bits):
    """the hand-written wave64 radix sort (fjgpu_raysort.hip: tile histograms, row scans, match-any ranking by ballots, tiles sorted
    in LDS) against numpy's stable argsort: keys ascending, perm the stable permutation -- one tile and many, ragged last tiles, one
    to four passes, digits narrower than 8 bits, keys with runs of equal values and bits above key_bits that must be ignored"""
    rng = np.random.RandomState(n % 9973 + key_bits)
    keys = rng.randint(0, 1 << min(key_bits, 31), size=n, dtype=np.int64).astype(np.uint32)
    if key_bits == 32:
        keys |= (rng.randint(0, 2, size=n).astype(np.uint32) << np.uint32(31))
    if n > 1000:
        keys[n // 3:n // 3 + 700] = keys[n // 3]          # a long run of one key across tile / wave / round boundaries
    junk = np.uint32(0)
    if key_bits < 32:
        junk = (rng.randint(0, 1 << 8, size=n).astype(np.uint32) << np.uint32(key_bits)) if key_bits <= 24 else np.uint32(0)
    raw = keys | junk
    out, perm, ms = gpu.sort_pairs(raw, key_bits)
    want = np.argsort(keys, kind="stable").astype(np.uint32)
    assert np.array_equal(perm, want)
    assert np.array_equal(out, raw[want])
    assert ms >= 0


def test_ray_sort_of_nothing_and_bad_arguments():
    out, perm, ms = gpu.sort_pairs(np.zeros(0, dtype=np.uint32), 15)
    assert out.size == 0 and perm.size == 0
    with pytest.raises(gpu.GpuError):
        gpu.sort_pairs(np.zeros(4, dtype=np.uint32), 0)
    with pytest.raises(gpu.GpuError):
        gpu.sort_pairs(np.zeros(4, dtype=np.uint32), 33)


@pytest.mark.parametrize("kind", ["grid", "sphere", "both"])
def test_area_lights_match_oracle(kind, asset_dir):
    """RectangleLight / SphereLight with the per-event counter-based stream: the device draws
    the same positions as the oracle -- same shadow rays (counts equal), same pixels."""
    fb, st, ref, rc = render_both(workloads.arealights(asset_dir, res=(64, 48), spp=(3, 3), mesh="tiny", kind=kind))
    assert st.rays.as_dict() == rc.as_dict()
    assert rc.shadow > rc.camera
    assert float(rel_err(fb, ref).max()) <= REL_TOL


@pytest.mark.parametrize("name", sorted(CASES))
def test_frames_match_oracle_and_reference_golden(name, asset_dir, golden_dir):
    builder, kw = CASES[name]
    fb, st, ref, rc = render_both(workloads.BUILDERS[builder](asset_dir, **kw))
    assert_parity(fb, st, ref, rc)
    golden = np.load(os.path.join(golden_dir, "frames.npz"))[name]   # rendered by the compiled reference
    assert fb.shape == golden.shape
    assert close_enough(fb, None, golden, None, _last["adaptive"], REL_TOL)
    assert st.nodes_visited > 0 and st.prims_tested > 0 and st.rays_traced == st.rays.total()


def test_trace_bit_exact_against_reference_grid_vectors(asset_dir, golden_dir):
    """fjgpu_trace on the golden ray set: t / primitive ids produced by the REFERENCE's
    GridAccelerator + Mesh::ray_intersect are reproduced bit-exactly by the BVH path."""
    import test_oracle_golden as tg
    vec = golden_io.read_vectors(os.path.join(golden_dir, "ref_vectors.bin"))
    sp, _ = prepare(tg._mesh_scene(asset_dir))
    rays = np.load(os.path.join(golden_dir, "mesh_trace_rays.npy"))
    gs = gpu.Scene(sp)
    gs.set_option("count_nodes", 1)
    t, ids, uv, st = gs.trace(0, rays)
    gs.close()
    assert np.array_equal(t, vec["grid_t"])
    assert np.array_equal(ids[:, 1], vec["grid_prim"])
    assert st.rays_traced == rays.shape[0]


@pytest.mark.parametrize("builder,kw", [("teapot", dict(res=(64, 64), spp=(1, 1))),
                                        ("buddhas", dict(res=(64, 36), spp=(1, 1), mesh="bunny"))])
def test_trace_groups_bit_exact_against_oracle(builder, kw, asset_dir):
    """every group of the scene (shadow group + all-objects group), random ray soup incl.
    short tmax, origins inside geometry and axis-aligned directions"""
    sp, _ = prepare(workloads.BUILDERS[builder](asset_dir, **kw))
    rng = np.random.RandomState(5)
    n = 30000
    o = rng.normal(size=(n, 3)) * [4, 2, 4] + [0, 1.5, 0]
    tgt = rng.normal(size=(n, 3)) * [2, 1, 2] + [-1, 1, -1]
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[::97] = [0, -1, 0]
    d[1::97] = [1, 0, 0]
    tmax = np.where(rng.uniform(size=n) < .3, rng.uniform(.1, 6, size=n), 1000.)
    rays = np.concatenate([o, d, np.full((n, 1), 1e-4), tmax[:, None]], axis=1)
    gs = gpu.Scene(sp)
    osc = oracle_ffi.OracleScene(sp)
    for group in (0, 1):
        t, ids, uv, _ = gs.trace(group, rays)
        to, io, _ = osc.trace(group, rays)
        assert np.array_equal(t, to), group
        assert np.array_equal(ids, io), group
        assert (io[:, 0] >= 0).mean() > 0.02
    gs.close()
    osc.close()


def test_curve_trace_bit_exact_against_oracle(asset_dir):
    """C5 primitives: Bezier ribbons (Nakamaru-Ono subdivision) incl. the reference grid's
    cell-listing acceptance rule; t and instance bit-exact (the reference leaves prim_id of a
    curve hit at 0, so primitive ids are compared for mesh hits only)"""
    sp, _ = prepare(workloads.furry(asset_dir, res=(32, 24), spp=(1, 1), mesh="furball", nlights=1))
    rng = np.random.RandomState(9)
    n = 40000
    o = rng.normal(size=(n, 3)) * .4 + [0.3, .25, .3]
    tgt = rng.normal(size=(n, 3)) * .07 + [0, .12, 0]
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([o, d, np.full((n, 1), 1e-4), np.full((n, 1), 1000.)], axis=1)
    gs = gpu.Scene(sp)
    osc = oracle_ffi.OracleScene(sp)
    for group in (0, 1):
        t, ids, uv, _ = gs.trace(group, rays)
        to, io, _ = osc.trace(group, rays)
        assert np.array_equal(t, to), group
        assert np.array_equal(ids[:, 0], io[:, 0]), group
    curve_inst = 3                                   # bunny1, floor1, dome1, curve1
    hits_curve = (io[:, 0] == curve_inst)
    assert hits_curve.sum() > 500
    mesh_hit = (io[:, 0] >= 0) & ~hits_curve
    assert np.array_equal(ids[mesh_hit, 1], io[mesh_hit, 1])
    gs.close()
    osc.close()


ADAPTIVE = (("sampler_type", (1,)), ("adaptive_max_subdivision", (2,)), ("adaptive_subdivision_threshold", (.03,)))


@pytest.mark.parametrize("builder,kw", [
    ("cornell", dict(res=(48, 32), spp=(2, 2), mesh="tiny", extra=ADAPTIVE)),
    ("arealights", dict(res=(64, 48), spp=(3, 3), mesh="tiny", kind="both", extra=ADAPTIVE)),
    ("furry", dict(res=(64, 48), spp=(2, 2), mesh="furball", nlights=4, extra=ADAPTIVE)),
    ("ibl", dict(res=(64, 48), spp=(2, 2), mesh="small", sample_count=48, extra=ADAPTIVE)),
    # no filter margin (width 1 -> ceil(0) pixels), no jitter, ragged 20x20 tiles
    ("teapot", dict(res=(70, 50), spp=(1, 1), extra=ADAPTIVE + (("filterwidth", (1, 1)), ("sample_jitter", (0,)), ("tilesize", (20, 20))))),
    ("teapot", dict(res=(64, 64), spp=(1, 1), extra=(("sampler_type", (1,)), ("adaptive_max_subdivision", (4,)),
                                                      ("adaptive_subdivision_threshold", (.1,)), ("filterwidth", (2.5, 4))))),
], ids=["pathtracing", "area_lights", "curves", "dome_light", "no_margin_no_jitter", "depth4_wide_filter"])
def test_adaptive_sampler_with_every_shading_path(builder, kw, asset_dir):
    """AdaptiveGridSampler (sampler_type 1) as a level-synchronous wavefront: the same samples
    are traced as by the reference's stack walk (camera-ray counts equal the oracle's, whose
    walk is pinned bit-exactly against the reference), the same pixels come out.  Here with
    the shading paths whose random streams are keyed by the sample's index in its tile."""
    fb, st, ref, rc = render_both(workloads.BUILDERS[builder](asset_dir, **kw))
    assert_parity(fb, st, ref, rc)


def test_adaptive_sampler_batches_and_subdivision_depths(asset_dir):
    """tiles stay independent under the adaptive sampler (any batch split: same pixels), a
    threshold nothing exceeds traces only the pixel corners, threshold 0 traces the lattice of
    every pixel whose corners differ at all, and depth 0 has nothing to subdivide"""
    def scene(depth, threshold, res=(96, 64)):
        return workloads.dragon(asset_dir, res=res, spp=(2, 2), mesh="tiny", extra=(
            ("sampler_type", (1,)), ("adaptive_max_subdivision", (depth,)), ("adaptive_subdivision_threshold", (threshold,))))
    sp, rd = prepare(scene(3, .04))
    gs = gpu.Scene(sp)
    full, st_full = gs.render_frame(rd)
    for bt in (1, 2, 5):
        gs.set_option("batch_tiles", bt)
        fb, st = gs.render_frame(rd)
        assert st.batches == -(-gpu.tile_count(rd) // bt)
        assert close_enough(fb, st.rays.as_dict(), full, st_full.rays.as_dict(), True, 1e-6)
    gs.close()
    osc = oracle_ffi.OracleScene(sp)
    ref, rc = osc.render(rd)
    assert close_enough(full, st_full.rays.as_dict(), ref, rc.as_dict(), True, REL_TOL)
    corners = sum((gpu.tile_rect(rd, t)[2] - gpu.tile_rect(rd, t)[0] + 3) * (gpu.tile_rect(rd, t)[3] - gpu.tile_rect(rd, t)[1] + 3)
                  for t in range(gpu.tile_count(rd)))        # (tile + 2 margin pixels + 1)^2 lattice corners
    lattice = sum((8 * (gpu.tile_rect(rd, t)[2] - gpu.tile_rect(rd, t)[0] + 2) + 1) * (8 * (gpu.tile_rect(rd, t)[3] - gpu.tile_rect(rd, t)[1] + 2) + 1)
                  for t in range(gpu.tile_count(rd)))
    assert corners < rc.camera < lattice
    osc.close()
    for depth, threshold, expect in ((3, 1e9, corners), (0, 0., corners)):
        sp, rd = prepare(scene(depth, threshold))
        gs = gpu.Scene(sp)
        fb, st = gs.render_frame(rd)
        gs.close()
        osc = oracle_ffi.OracleScene(sp)
        ref, rc = osc.render(rd)
        osc.close()
        assert st.rays.camera == rc.camera == expect
        assert float(rel_err(fb, ref).max()) <= REL_TOL


def test_tile_subsets_and_batching_are_consistent(asset_dir):
    """tiles are independent units: any subset / batch split gives the same pixels, and
    pixels of tiles that were not listed stay untouched"""
    import torch
    sp, rd = prepare(workloads.dragon(asset_dir, res=(160, 90), spp=(2, 2), mesh="small"))
    gs = gpu.Scene(sp)
    full, st_full = gs.render_frame(rd)
    n = gpu.tile_count(rd)
    assert n == 15
    sub = [1, 7, 14, 3]
    fb = torch.full((rd.yres, rd.xres, 4), -1.0, dtype=torch.float32, device="cuda")
    gs.set_option("batch_tiles", 3)
    st = gs.render_tiles(rd, sub, fb.data_ptr())
    out = fb.cpu().numpy()
    touched = np.zeros((rd.yres, rd.xres), dtype=bool)
    for t in sub:
        x0, y0, x1, y1 = gpu.tile_rect(rd, t)
        touched[y0:y1, x0:x1] = True
        assert float(rel_err(out[y0:y1, x0:x1], full[y0:y1, x0:x1]).max()) <= 1e-6
    assert (out[~touched] == -1.0).all()
    assert st.batches == 2 and st.rays.camera < st_full.rays.camera
    gs.set_option("batch_tiles", 1)
    one_by_one, st1 = gs.render_frame(rd)
    assert float(rel_err(one_by_one, full).max()) <= 1e-6
    assert st1.rays.as_dict() == st_full.rays.as_dict() and st1.batches == 15
    # ... and by a sample budget (option "batch_samples": what SiRenderScene sets for its one-frame scenes): batches of whole tiles
    # that hold at most that many samples, batch_tiles = 0 meaning "not set"
    gs.set_option("batch_tiles", 0)
    per_tile = (2 * rd.tile_w + 2 * 2) * (2 * rd.tile_h + 2 * 2)          # (upper bound of a tile's samples: rate 2, filter margin <= 2)
    gs.set_option("batch_samples", 4 * per_tile)
    by_samples, st2 = gs.render_frame(rd)
    assert float(rel_err(by_samples, full).max()) <= 1e-6
    assert st2.rays.as_dict() == st_full.rays.as_dict() and 4 <= st2.batches <= 8
    gs.set_option("batch_samples", 0)
    whole, st3 = gs.render_frame(rd)
    assert st3.batches == 1 and float(rel_err(whole, full).max()) <= 1e-6
    gs.close()


def test_cold_start_renders_the_same_frame_from_a_small_arena(asset_dir):
    """global option "cold_start" (default on): a scene's FIRST render call works in batches of "cold_batch_samples" samples -- a small work
    arena, the first image without waiting for the whole-frame allocation -- and the second call sizes its batches by memory; the pixels and
    the ray counts are those of a scene that never batched"""
    sp, rd = prepare(workloads.dragon(asset_dir, res=(160, 90), spp=(2, 2), mesh="small"))
    per_tile = (2 * rd.tile_w + 2 * 2) * (2 * rd.tile_h + 2 * 2)          # (upper bound of a tile's samples: rate 2, filter margin <= 2)
    gpu.global_option("cold_start", 0)
    try:
        ref_scene = gpu.Scene(sp)
        ref, st_ref = ref_scene.render_frame(rd)
        assert st_ref.batches == 1
        whole_arena = ref_scene.query("work_bytes")
        ref_scene.close()
        gpu.global_option("cold_start", 1)
        gpu.global_option("cold_batch_samples", 4 * per_tile)
        gs = gpu.Scene(sp)
        first, st1 = gs.render_frame(rd)
        small_arena = gs.query("work_bytes")
        second, st2 = gs.render_frame(rd)
        assert 4 <= st1.batches <= 8 and st2.batches == 1
        assert st1.rays.as_dict() == st_ref.rays.as_dict() == st2.rays.as_dict()
        assert float(rel_err(first, ref).max()) <= 1e-6 and float(rel_err(second, ref).max()) <= 1e-6
        # (the arena only grows; at this size its floors decide, so equality is allowed: profiles/r05_arena_sizes.txt has the 14 -> 110 GB of C3)
        assert small_arena <= gs.query("work_bytes") and gs.query("work_bytes") >= whole_arena
        # option "release_work": the arena back to the driver, the scene stays; the next call allocates again and renders the same frame
        gs.set_option("release_work", 1)
        assert gs.query("work_bytes") == 0
        third, st4 = gs.render_frame(rd)
        assert gs.query("work_bytes") > 0 and st4.rays.as_dict() == st_ref.rays.as_dict() and float(rel_err(third, ref).max()) <= 1e-6
        # an explicit batch size is the caller's business: the policy stays out of it
        g2 = gpu.Scene(sp)
        g2.set_option("batch_tiles", 15)
        _, st3 = g2.render_frame(rd)
        assert st3.batches == 1
        g2.close()
        gs.close()
    finally:
        gpu.global_option("cold_start", 1)
        gpu.global_option("cold_batch_samples", 0)


@pytest.mark.parametrize("scene", ["teapot", "dragon"])
def test_speculative_walk_changes_nothing(asset_dir, scene):
    """global option "speculative_walk" (default on): the closest-hit walk of recursion level L + 1 is enqueued behind level L's shading launch and
    reads its ray count from device memory, the counters come back on a side stream -- the same rays, the same launches that have rays, the same
    pixels as the loop that waits for the host at every level (glass: two children per hit; plastic mirror: one)"""
    if scene == "teapot":
        sp, rd = prepare(workloads.teapot(asset_dir, res=(96, 96), spp=(2, 2)))
    else:
        sp, rd = prepare(workloads.dragon(asset_dir, res=(160, 90), spp=(2, 2), mesh="small"))
    out = {}
    try:
        for on in (0, 1):
            gpu.global_option("speculative_walk", on)
            gs = gpu.Scene(sp)
            fb, st = gs.render_frame(rd)
            fb2, st2 = gs.render_frame(rd)
            assert st.rays.as_dict() == st2.rays.as_dict()
            out[on] = (fb2, st2)
            gs.close()
    finally:
        gpu.global_option("speculative_walk", 1)
    assert out[0][1].rays.as_dict() == out[1][1].rays.as_dict()
    assert out[0][1].rays.reflect > 0
    assert float(rel_err(out[1][0], out[0][0]).max()) <= 1e-6


from edge_scenes import EDGE_CASES, custom_scene as _custom_scene  # noqa: E402


@pytest.mark.parametrize("name", sorted(EDGE_CASES))
def test_edge_cases_match_oracle(name, asset_dir):
    fb, st, ref, rc = render_both(_custom_scene(asset_dir, **EDGE_CASES[name]))
    assert st.rays.as_dict() == rc.as_dict(), name
    assert float(rel_err(fb, ref).max()) <= REL_TOL, name
    if name == "no_shadows":
        assert st.rays.shadow == 0
    if name == "zero_depth":
        assert st.rays.reflect == 0 and st.rays.refract == 0
    if name == "no_lights":
        assert st.rays.shadow == 0 and ref.max() > 0      # dome / reflections still contribute


def test_si_render_scene_end_to_end(asset_dir):
    """the drop-in path: scene text -> Si* API -> SiRenderScene -> HIP core -> FrameBuffer"""
    text = workloads.teapot(asset_dir, res=(64, 64), spp=(2, 2))
    host.run_scene_text(text, deferred=False)
    fb = host.framebuffer(0)
    st = host.last_stats()
    sp, rd = host.get_desc()
    osc = oracle_ffi.OracleScene(sp)
    ref, rc = osc.render(rd)
    osc.close()
    assert float(rel_err(fb, ref).max()) <= REL_TOL
    assert st.rays.as_dict() == rc.as_dict() and st.render_seconds > 0


@pytest.mark.parametrize("mode", [1, 2], ids=["clustering", "radix_tree"])
def test_device_blas_build_gives_the_same_hits_and_pixels(asset_dir, golden_dir, mode):
    """BLAS built on the device (fjgpu_lbvh.hip: locally-ordered clustering, or the radix tree of
    the Morton codes) instead of the host's binned-SAH tree: closest hits do not depend on the
    culling structure, so t / ids stay bit-exact against the reference grid vectors and a frame
    matches the oracle."""
    import test_oracle_golden as tg
    vec = golden_io.read_vectors(os.path.join(golden_dir, "ref_vectors.bin"))
    gpu.global_option("device_build", mode)
    try:
        sp, _ = prepare(tg._mesh_scene(asset_dir))
        rays = np.load(os.path.join(golden_dir, "mesh_trace_rays.npy"))
        gs = gpu.Scene(sp)
        assert gs.query("blas_nodes") > 0
        t, ids, uv, st = gs.trace(0, rays)
        gs.close()
        assert np.array_equal(t, vec["grid_t"])
        assert np.array_equal(ids[:, 1], vec["grid_prim"])
        fb, st, ref, rc = render_both(workloads.dragon(asset_dir, res=(96, 54), spp=(3, 3), mesh="small"))
        assert st.rays.as_dict() == rc.as_dict()
        assert float(rel_err(fb, ref).max()) <= REL_TOL
        fb, st, ref, rc = render_both(workloads.motion(asset_dir, res=(64, 48), spp=(2, 2), mesh="tiny", kind="velocity+object"))
        assert st.rays.as_dict() == rc.as_dict()
        assert float(rel_err(fb, ref).max()) <= REL_TOL
    finally:
        gpu.global_option("device_build", -1)


def test_device_tlas_equals_host_tlas_node_for_node(asset_dir):
    """the instance level of every group is built on the device (fjgpu_tlas.hip: reference topology,
    sort by centroid on the cycling axis + find_median, threaded depth first); option tlas_verify makes
    scene creation compare it byte for byte with the host's build of the same list.  A frame rendered
    with either list is the same frame, and it is the oracle's."""
    scenes = [workloads.crowd(asset_dir, res=(64, 48), spp=(2, 2), mesh="tiny", n=150),
              workloads.crowd(asset_dir, res=(64, 48), spp=(2, 2), mesh="tiny", n=777, nlights=2),
              workloads.cornell(asset_dir, res=(48, 32), spp=(2, 2), mesh="tiny"),
              workloads.buddhas(asset_dir, res=(64, 36), spp=(2, 2), mesh="tiny")]
    gpu.global_option("tlas_verify", 1)
    try:
        for k, text in enumerate(scenes):
            sp, rd = prepare(text)
            gs = gpu.Scene(sp)                       # fails if the two builds differ
            fb_dev, st_dev = gs.render_frame(rd)
            gs.close()
            gpu.global_option("device_tlas", 0)
            gs = gpu.Scene(sp)
            fb_host, st_host = gs.render_frame(rd)
            gs.close()
            gpu.global_option("device_tlas", 1)
            assert st_dev.rays.as_dict() == st_host.rays.as_dict()
            assert float(rel_err(fb_dev, fb_host).max()) <= 1e-6
            if k == 0:
                osc = oracle_ffi.OracleScene(sp)
                ref, rc = osc.render(rd)
                osc.close()
                assert st_dev.rays.as_dict() == rc.as_dict()
                assert float(rel_err(fb_dev, ref).max()) <= REL_TOL
    finally:
        gpu.global_option("tlas_verify", 0)
        gpu.global_option("device_tlas", 1)


def test_shadow_rays_split_per_candidate_instance(asset_dir):
    """shadow groups of several instances, lean any-hit walk: a ray whose world-space test passes k instance
    boxes is queued k times (one entry per instance, one join slot), and the entry that completes the count
    of unoccluded ones adds the light (option split_shadow, the default); off, the walk steps through the
    group's instance level itself.  Either way: the oracle's ray counts and pixels."""
    for text in (workloads.crowd(asset_dir, res=(64, 48), spp=(2, 2), mesh="tiny", n=40),
                 workloads.buddhas(asset_dir, res=(96, 54), spp=(2, 2), mesh="tiny")):
        frames = []
        for split in (1, 0):
            gpu.global_option("split_shadow", split)
            try:
                fb, st, ref, rc = render_both(text)
            finally:
                gpu.global_option("split_shadow", 1)
            assert st.rays.as_dict() == rc.as_dict()
            assert float(rel_err(fb, ref).max()) <= REL_TOL
            frames.append((fb, st))
        assert float(rel_err(frames[0][0], frames[1][0]).max()) <= 1e-5


@pytest.mark.parametrize("builder,kw", [
    ("cornell", dict(res=(96, 54), spp=(3, 3), mesh="tiny")),          # the phase-scheduled walk (39 / 20 / 12)
    ("buddhas", dict(res=(96, 54), spp=(2, 2), mesh="tiny")),          # k_trace_closest + the light loop's candidate search
    ("crowd", dict(res=(64, 48), spp=(2, 2), mesh="tiny", n=30)),      # 30 instances: beyond the phased walk's budget, inside the big one
    ("crowd", dict(res=(64, 48), spp=(2, 2), mesh="tiny", n=150)),     # fits no walk's budget (the light loop's node copy only)
    ("arealights", dict(res=(64, 48), spp=(2, 2), mesh="tiny", kind="both")),
    ("furry", dict(res=(64, 48), spp=(2, 2), mesh="furball", nlights=4)),   # the curve instantiations' own budget (15 / 8 / 12)
])
def test_instance_level_in_lds_changes_nothing(builder, kw, asset_dir):
    """option inst_lds: the walks of a scene whose instance level fits their budget read it from an LDS copy
    (DInstEntry); off, from global memory.  Same ray counts, same instance-level events, same pixels, and the oracle's."""
    text = getattr(workloads, builder)(asset_dir, **kw)
    out = []
    for on in (1, 0):
        gpu.global_option("inst_lds", on)
        try:
            fb, st, ref, rc = render_both(text)
        finally:
            gpu.global_option("inst_lds", 1)
        assert_parity(fb, st, ref, rc)
        out.append((fb, st))
    (fb1, st1), (fb0, st0) = out
    assert st1.rays.as_dict() == st0.rays.as_dict()
    # instance-level events are per ray and order-free: equal.  Node / triangle counts of the ANY-HIT walk are not a function
    # of the ray alone: a lane sets a leaf aside and walks on while its wave runs inner steps (FJ_ANYHIT_POSTPONE), so an
    # occluded ray visits a few nodes more or less depending on which rays the light loop's atomics queued next to it
    # (measured on the box: 75 of 412 663 between two runs) -- same order of magnitude is all that can be asserted.
    # (curve scenes: with the instance level in LDS the shadow rays run k_shadow_anyhit_curves, without it the general walk -- other visiting
    # order (no distance sort), postponed leaves, another moment at which an occluded ray learns it: the same results from other event counts)
    # (cornell: its groups are FLAT -- with the instance level in LDS the closest-hit rays run k_trace_closest_flat, one world-space tree per group,
    # without it the phase-scheduled walk with its instance loop: other trees, other event counts, the same hits)
    other_kernel = builder in ("furry", "cornell")
    tol = 0.5 if builder == "cornell" else (0.10 if builder == "furry" else 0.01)
    if not other_kernel:
        assert st1.insts_tested == st0.insts_tested
    if builder != "cornell":       # (one world-space tree instead of nine object-space ones: another node count altogether)
        assert abs(st1.nodes_visited - st0.nodes_visited) <= tol * st0.nodes_visited
    assert abs(st1.prims_tested - st0.prims_tested) <= tol * st0.prims_tested
    assert float(rel_err(fb1, fb0).max()) <= 1e-5


@pytest.mark.parametrize("kw", [
    dict(res=(64, 48), spp=(2, 2), mesh="furball", nlights=4),
    dict(res=(96, 64), spp=(3, 3), mesh="furball", nlights=9),           # more lights than a wave iteration's worth of pending curves
])
def test_curve_anyhit_walk_changes_nothing(kw, asset_dir):
    """option curve_anyhit: the shadow rays of a curve scene whose occluders are all opaque run k_shadow_anyhit_curves (phase-scheduled,
    pending curves, ribbon phase); off, the general walk.  Same ray counts per context, same pixels, and the oracle's
    (Curve::ray_intersect, src/fj_curve.cc:187-232, through SlIlluminance, src/fj_shading.cc:296-359)."""
    text = workloads.furry(asset_dir, **kw)
    out = []
    _last["adaptive"] = False
    for on in (1, 0):
        gpu.global_option("curve_anyhit", on)
        try:
            sp, rd = prepare(text)
            gs = gpu.Scene(sp)
            assert int(gs.query("curve_anyhit")) == on
            fb, st = gs.render_frame(rd)
            gs.close()
        finally:
            gpu.global_option("curve_anyhit", 1)
        out.append((fb, st))
    osc = oracle_ffi.OracleScene(sp)
    ref, rc = osc.render(rd)
    osc.close()
    for fb, st in out:
        assert_parity(fb, st, ref, rc)
    assert out[0][1].rays.as_dict() == out[1][1].rays.as_dict()
    assert float(rel_err(out[0][0], out[1][0]).max()) <= 1e-5


@pytest.mark.parametrize("builder,kw", [
    ("cornell", dict(res=(96, 54), spp=(3, 3), mesh="tiny")),
    ("teapot", dict(res=(64, 64), spp=(2, 2), mesh="tiny")),                               # glass: reflect + refract children, shadow rays beside
    ("cornell", dict(res=(64, 36), spp=(2, 2), mesh="tiny", objects=("bunny",))),
    # exact ties in t ACROSS instances in a scene with a glass shader: the instance earlier in the group's order keeps the hit
    ("edge", dict(obj_shader="glass_shader", twins=3, lights=2)),
    ("edge", dict(obj_shader="glass_shader", twins=5, lights=1, with_object=False, spp=(1, 1))),
])
def test_flat_groups_change_nothing(builder, kw, asset_dir):
    """option flat_groups: scenes with incoherent closest-hit rays whose groups hold small static meshes walk ONE world-space culling tree per group
    (k_trace_closest_flat); off, the phase-scheduled walk with its instance loop.  Same rays per context, same pixels, and the oracle's
    (src/fj_bvh_accelerator.cc:164-241, src/fj_object_instance.cc:213-243)."""
    if builder == "edge":
        from edge_scenes import custom_scene
        text = custom_scene(asset_dir, **kw)
    else:
        text = getattr(workloads, builder)(asset_dir, **kw)
    out = []
    _last["adaptive"] = False
    for on in (1, 0):
        gpu.global_option("flat_groups", on)
        try:
            sp, rd = prepare(text)
            gs = gpu.Scene(sp)
            assert (int(gs.query("closest_kernel")) == 4) == (on == 1)
            fb, st = gs.render_frame(rd)
            gs.close()
        finally:
            gpu.global_option("flat_groups", 1)
        out.append((fb, st))
    osc = oracle_ffi.OracleScene(sp)
    ref, rc = osc.render(rd)
    osc.close()
    for fb, st in out:
        assert_parity(fb, st, ref, rc)
    assert out[0][1].rays.as_dict() == out[1][1].rays.as_dict()
    assert float(rel_err(out[0][0], out[1][0]).max()) <= 1e-5


@pytest.mark.parametrize("builder,kw", [
    ("dragon", dict(res=(96, 54), spp=(2, 2), mesh="tiny", nlights=5)),                    # lean any-hit walk, one instance per entry
    ("buddhas", dict(res=(96, 54), spp=(2, 2), mesh="tiny", nlights=4)),                   # rays split per candidate instance (join slots in tindex)
    ("furry", dict(res=(64, 48), spp=(2, 2), mesh="furball", nlights=4)),                  # the curve scenes' any-hit walk
    ("ibl", dict(res=(64, 36), spp=(2, 2), mesh="tiny", sample_count=16)),                 # dome light samples
])
def test_compact_shadow_queue_changes_nothing(builder, kw, asset_dir):
    """option compact_squeue: 56-byte shadow-queue records without direction and distance -- the any-hit walks rebuild them from the origin and the
    light sample's position with the light loop's own statements (SlIlluminance, src/fj_shading.cc:296-359) -- against the 80-byte records:
    the same rays per context, the same pixels, and the oracle's"""
    text = getattr(workloads, builder)(asset_dir, **kw)
    out = []
    _last["adaptive"] = False
    for on in (1, 0):
        gpu.global_option("compact_squeue", on)
        try:
            sp, rd = prepare(text)
            gs = gpu.Scene(sp)
            fb, st = gs.render_frame(rd)
            gs.close()
        finally:
            gpu.global_option("compact_squeue", 1)
        out.append((fb, st))
    osc = oracle_ffi.OracleScene(sp)
    ref, rc = osc.render(rd)
    osc.close()
    for fb, st in out:
        assert_parity(fb, st, ref, rc)
    assert out[0][1].rays.as_dict() == out[1][1].rays.as_dict()
    assert float(rel_err(out[0][0], out[1][0]).max()) <= 1e-6


def test_hair_shader_declared_in_an_all_opaque_mesh_scene_with_split_shadow_rays(asset_dir):
    """a HairShader in a scene WITHOUT curves whose shadow groups hold several instances: the light loop runs its
    hair instantiation, and that one must queue rays per candidate instance exactly like the plain one does (the
    lean any-hit walk reads every entry as (instance, join slot)); both with the shader merely declared and with
    it shading a mesh -- the oracle's ray counts and pixels, split on and off"""
    base = workloads.crowd(asset_dir, res=(64, 48), spp=(2, 2), mesh="tiny", n=12)
    declared = base.replace("NewCamera cam1", "OpenPlugin hair_shader HairShader.so\nNewShader hair1 hair_shader\nNewCamera cam1", 1)
    assert declared != base
    used = declared.replace("AssignShader obj3 DEFAULT_SHADING_GROUP obj_shader1", "AssignShader obj3 DEFAULT_SHADING_GROUP hair1")
    assert used != declared
    for text in (declared, used):
        for split in (1, 0):
            gpu.global_option("split_shadow", split)
            try:
                fb, st, ref, rc = render_both(text)
            finally:
                gpu.global_option("split_shadow", 1)
            assert st.rays.as_dict() == rc.as_dict()
            assert rc.shadow > rc.camera
            assert float(rel_err(fb, ref).max()) <= REL_TOL


@pytest.mark.parametrize("builder,kw", [
    # 80 x 60 tiles of 8 x 8 pixels = 4800 tiles (> 4096: the tile id no longer fits 12 bits of the sample uid)
    ("motion", dict(res=(640, 480), spp=(1, 1), mesh="tiny", kind="both", extra=(("tilesize", (8, 8)),))),
    ("arealights", dict(res=(640, 480), spp=(1, 1), mesh="tiny", kind="both", extra=(("tilesize", (8, 8)),))),
    ("cornell", dict(res=(640, 480), spp=(1, 1), mesh="tiny", extra=(("tilesize", (8, 8)),))),
    # one tile of (2 * 520 + 4)^2 = 1.09 M samples (> 2^20: the sample's time index no longer fits 20 bits of the uid)
    ("motion", dict(res=(520, 520), spp=(2, 2), mesh="tiny", kind="object", extra=(("tilesize", (520, 520)),))),
], ids=["motion_4800_tiles", "area_lights_4800_tiles", "pathtracing_4800_tiles", "motion_tile_of_1M_samples"])
def test_frames_beyond_4096_tiles_and_tiles_beyond_2_20_samples(builder, kw, asset_dir):
    """sample times and random streams are keyed by (tile id, sample index in the tile): the uid is the 64-bit
    tile * 2^20 + index folded to 32 bits, the time index travels on its own (DPath.flags) -- a 1080p frame with
    16 x 16 tiles (8160 tiles) renders like any other.  Device == oracle (same contract)."""
    fb, st, ref, rc = render_both(workloads.BUILDERS[builder](asset_dir, **kw))
    assert st.rays.as_dict() == rc.as_dict()
    assert float(rel_err(fb, ref).max()) <= REL_TOL


def test_unsupported_features_fail_loudly(asset_dir):
    """features outside the device path: explicit error naming the feature, never a silent
    approximation or a CPU fallback"""
    base = _custom_scene(asset_dir, lights=1)
    deep = base.replace("RenderScene ren1", "SetProperty1 ren1 sampler_type 1\nSetProperty1 ren1 adaptive_max_subdivision 9\nRenderScene ren1")
    sp, rd = prepare(deep)
    gs = gpu.Scene(sp)
    with pytest.raises(gpu.GpuError) as e:
        gs.render_frame(rd)
    gs.close()
    assert "adaptive_max_subdivision" in str(e.value)
    with pytest.raises(Exception) as e:
        prepare(base.replace("OpenPlugin plastic_shader PlasticShader.so", "OpenPlugin plastic_shader VolumeShader.so"))
    assert "no device implementation" in str(e.value) or "VolumeShader" in str(e.value)


def test_full_size_headline_properties(asset_dir):
    """BASELINE.json configs[2] at full size (7.22 M triangles, 1920x1080, 8x8 spp):
    size-independent properties + oracle parity on two whole tiles."""
    import torch
    sp, rd = prepare(workloads.dragon(asset_dir))
    assert (rd.xres, rd.yres, rd.rate_x, rd.rate_y) == (1920, 1080, 8, 8)
    n = gpu.tile_count(rd)
    assert n == 2040
    gs = gpu.Scene(sp)
    # a band of tiles through the dragon, rendered in two different batch splits
    nx = 60
    band = [17 * nx + x for x in range(20, 40)]
    fb = torch.zeros((rd.yres, rd.xres, 4), dtype=torch.float32, device="cuda")
    st = gs.render_tiles(rd, band, fb.data_ptr())
    a = fb.cpu().numpy()
    assert st.rays.camera == len(band) * 264 * 264          # (8*32 + 2*4)^2 samples per full tile
    assert st.rays.shadow > 10 * st.rays.camera and st.rays.reflect > 0 and st.rays.refract == 0
    gs.set_option("batch_tiles", 7)
    fb.zero_()
    st2 = gs.render_tiles(rd, band[::-1], fb.data_ptr())
    b = fb.cpu().numpy()
    assert st2.rays.as_dict() == st.rays.as_dict()
    assert float(rel_err(b, a).max()) <= 1e-6
    assert np.isfinite(a).all() and a.min() >= 0
    y0 = 17 * 32
    assert np.abs(a[y0:y0 + 32, 20 * 32:40 * 32, 3] - 1.0).max() < 1e-6   # every camera ray hits (dome) -> alpha 1
    assert not a[:y0].any() and not a[y0 + 32:].any()
    # oracle (reference grid accelerator) on two of those tiles
    pick = [band[3], band[11]]
    osc = oracle_ffi.OracleScene(sp)
    ref, rc = osc.render(rd, tile_ids=pick)
    osc.close()
    fb.zero_()
    st3 = gs.render_tiles(rd, pick, fb.data_ptr())
    c = fb.cpu().numpy()
    gs.close()
    assert st3.rays.as_dict() == rc.as_dict()
    for t in pick:
        x0, yy0, x1, y1 = gpu.tile_rect(rd, t)
        assert float(rel_err(c[yy0:y1, x0:x1], ref[yy0:y1, x0:x1]).max()) <= REL_TOL


def _round_number():
    """the build round: seeds the tile draw below, so that every round's GPU run checks OTHER full-size tiles than the last
    one did.  The driver leaves one BENCH_rNN.json per finished round at the repo root: this round is the highest NN + 1
    (FJ_ROUND overrides; no such file: round 1).  (Until round 4 this parsed the title of VERDICT.md -- brittle.)"""
    import glob
    import re
    if os.environ.get("FJ_ROUND", "").isdigit():
        return int(os.environ["FJ_ROUND"])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    done = [int(m.group(1)) for m in (re.search(r"BENCH_r(\d+)\.json$", f) for f in glob.glob(os.path.join(root, "BENCH_r*.json"))) if m]
    return max(done) + 1 if done else 1


def _drawn_tiles(builder, n_tiles, count, nx=None):
    """`count` tile ids, seeded with the round: half of them from the window in the middle of the frame (columns 30-70 %, rows 25-80 %), where
    every workload has its objects -- C6 is a statue and two balls in front of a sky dome: a draw over the whole frame came up with twelve
    tiles of sky in round 5 --, the rest from anywhere"""
    rng = np.random.RandomState(1000 * _round_number() + sum(ord(c) for c in builder))
    if not nx:
        return sorted(int(t) for t in rng.choice(n_tiles, size=count, replace=False))
    ny = -(-n_tiles // nx)
    mid = [y * nx + x for y in range(ny) for x in range(nx) if .30 * nx <= x + .5 <= .70 * nx and .25 * ny <= y + .5 <= .80 * ny and y * nx + x < n_tiles]
    a = [int(t) for t in rng.choice(mid, size=min(len(mid), (count + 1) // 2), replace=False)]
    rest = [t for t in range(n_tiles) if t not in set(a)]
    b = [int(t) for t in rng.choice(rest, size=count - len(a), replace=False)]
    return sorted(a + b)


@pytest.mark.parametrize("builder,expect,count", [
    ("dragon", (1920, 1080, 8, 8), 12),            # C3 (headline): 7.22 M triangles
    ("buddhas", (1280, 720, 4, 4), 12),            # C2: glass + plastic, 1.09 M triangles x 16 instances
    ("cornell", (1920, 1080, 16, 16), 12),         # C4: pathtracing, 256 spp
    ("furry", (1920, 1080, 8, 8), 8),              # C5: fur curves + hair shader (the oracle's slowest: 8 tiles)
    ("ibl", (1920, 1080, 8, 8), 12),               # C6: dome light, 256 samples
], ids=["c3_dragon", "c2_buddhas", "c4_pathtracing", "c5_furry", "c6_ibl"])
def test_full_size_configs_match_oracle_on_whole_tiles(builder, expect, count, asset_dir):
    """every BASELINE.json configuration at its FULL size: whole tiles, drawn afresh every round from a seeded generator
    (tile ids in the assertion message), against the oracle (reference accelerators, reference recursion) -- same rays
    per context, same pixels"""
    import torch
    sp, rd = prepare(workloads.BUILDERS[builder](asset_dir))
    assert (rd.xres, rd.yres, rd.rate_x, rd.rate_y) == expect
    pick = _drawn_tiles(builder, gpu.tile_count(rd), count, nx=-(-rd.xres // rd.tile_w))
    gs = gpu.Scene(sp)
    fb = torch.zeros((rd.yres, rd.xres, 4), dtype=torch.float32, device="cuda")
    st = gs.render_tiles(rd, pick, fb.data_ptr())
    out = fb.cpu().numpy()
    gs.close()
    osc = oracle_ffi.OracleScene(sp)
    ref, rc = osc.render(rd, tile_ids=pick)
    osc.close()
    assert st.rays.as_dict() == rc.as_dict(), (pick, st.rays.as_dict(), rc.as_dict())
    # (the drawn tiles hold real work, not only sky: secondary rays outnumber the camera rays several times over -- a draw of eight
    # tiles around the bunny's rim reached 8.2 x in round 5, where the check still said 10 x)
    assert rc.total() > 2 * rc.camera or builder == "cornell"
    for t in pick:
        x0, y0, x1, y1 = gpu.tile_rect(rd, t)
        # (a tile may be legitimately empty: camera rays that leave between the floor and the dome's rim hit nothing)
        assert float(rel_err(out[y0:y1, x0:x1], ref[y0:y1, x0:x1]).max()) <= REL_TOL, (builder, t, pick)
    assert out.any() and ref.any()


@pytest.mark.skipif(os.environ.get("FJ_SKIP_WHOLE_FRAME") == "1", reason="opted out (FJ_SKIP_WHOLE_FRAME=1)")
def test_whole_frame_c2_matches_oracle(asset_dir):
    """C2 (happy_buddhas-class, 1280x720, 16 spp) as a WHOLE frame against the oracle on the host threads (~40 s of CPU):
    every one of the 920 tiles, ray counts per context equal, every pixel within tolerance"""
    sp, rd = prepare(workloads.buddhas(asset_dir))
    gs = gpu.Scene(sp)
    fb, st = gs.render_frame(rd)
    gs.close()
    osc = oracle_ffi.OracleScene(sp)
    ref, rc = osc.render(rd)
    osc.close()
    assert st.rays.as_dict() == rc.as_dict()
    assert float(rel_err(fb, ref).max()) <= REL_TOL


def _whole_frame(text):
    sp, rd = prepare(text)
    gs = gpu.Scene(sp)
    fb, st = gs.render_frame(rd)
    gs.close()
    osc = oracle_ffi.OracleScene(sp)
    ref, rc = osc.render(rd)
    osc.close()
    assert st.rays.as_dict() == rc.as_dict()
    assert float(rel_err(fb, ref).max()) <= REL_TOL


def test_whole_frame_c3_matches_oracle(asset_dir):
    """C3, the HEADLINE configuration (xyzrgb_dragon-class, 7.22 M triangles, 1920x1080, 64 spp, 32 lights), as a WHOLE frame against the
    oracle on the host threads (~70 s of CPU on the GPU box): all 2040 tiles, 2.77 G rays -- ray counts per context equal, every pixel
    within tolerance (execute_rendering, src/fj_renderer.cc:747-791)"""
    _whole_frame(workloads.dragon(asset_dir))


@pytest.mark.skipif(os.environ.get("FJ_SKIP_WHOLE_FRAME") == "1", reason="opted out (FJ_SKIP_WHOLE_FRAME=1)")
@pytest.mark.parametrize("builder", ["ibl", "cornell"], ids=["c6_ibl", "c4_pathtracing"])
def test_whole_frame_c6_c4_match_oracle(builder, asset_dir):
    """C6 (dome light, 256 light samples) and C4 (pathtracing, 256 spp) as WHOLE frames in the default run since round 6: ~200 / ~140 s of
    oracle time on the GPU box's host threads each; ray counts per context equal, every pixel within tolerance"""
    _whole_frame(workloads.BUILDERS[builder](asset_dir))


@pytest.mark.skipif(not os.environ.get("FJ_TEST_WHOLE_FRAMES"), reason="~8 minutes of oracle time: set FJ_TEST_WHOLE_FRAMES=1 (scripts/full_frame_parity.py prints the table)")
def test_whole_frame_c5_matches_oracle(asset_dir):
    """C5 (fur) as a whole frame (478 s of oracle time on 64 threads)"""
    _whole_frame(workloads.BUILDERS["furry"](asset_dir))


@pytest.mark.parametrize("shape", ["tilesize16", "uhd_in_batches", "region_tilesize48x20"])
def test_full_size_c3_on_other_shapes(shape, asset_dir):
    """the headline scene away from its default shape, full-size tiles against the oracle: 16 x 16 tiles (8160 of them: four times the
    tiles per frame, a quarter of the samples per tile); 3840 x 2160 with the drawn tiles forced through batches of 5 (`batch_tiles`:
    the regime a frame larger than the work arena runs in); a render region that is not a multiple of its ragged 48 x 20 tiles"""
    import torch
    kw = {"tilesize16": dict(extra=(("tilesize", (16, 16)),)),
          "uhd_in_batches": dict(res=(3840, 2160)),
          "region_tilesize48x20": dict(extra=(("tilesize", (48, 20)), ("render_region", (301, 211, 1700, 1003))))}[shape]
    sp, rd = prepare(workloads.dragon(asset_dir, **kw))
    n = gpu.tile_count(rd)
    nx = -(-rd.region[2] // rd.tile_w) - rd.region[0] // rd.tile_w          # (tiles sit on the frame's tile grid: a region starts and ends with partial tiles)
    assert n == {"tilesize16": 8160, "uhd_in_batches": 8160, "region_tilesize48x20": 30 * 41}[shape]
    count = 24 if shape == "tilesize16" else 12
    pick = _drawn_tiles("dragon_" + shape, n, count, nx=nx)
    gs = gpu.Scene(sp)
    if shape == "uhd_in_batches":
        gs.set_option("batch_tiles", 5)
    fb = torch.zeros((rd.yres, rd.xres, 4), dtype=torch.float32, device="cuda")
    st = gs.render_tiles(rd, pick, fb.data_ptr())
    out = fb.cpu().numpy()
    gs.close()
    if shape == "uhd_in_batches":
        assert st.batches == 3
    osc = oracle_ffi.OracleScene(sp)
    ref, rc = osc.render(rd, tile_ids=pick)
    osc.close()
    assert st.rays.as_dict() == rc.as_dict(), (shape, pick, st.rays.as_dict(), rc.as_dict())
    assert rc.total() > 2 * rc.camera
    for t in pick:
        x0, y0, x1, y1 = gpu.tile_rect(rd, t)
        assert float(rel_err(out[y0:y1, x0:x1], ref[y0:y1, x0:x1]).max()) <= REL_TOL, (shape, t, pick)


@pytest.mark.parametrize("builder,kw", [("dragon", dict(res=(160, 90), spp=(3, 3), mesh="teapot")),
                                        ("buddhas", dict(res=(96, 54), spp=(2, 2), mesh="bunny")),
                                        ("ibl", dict(res=(64, 48), spp=(2, 2), mesh="small", sample_count=48))])
def test_anyhit_exact_phase_under_load(builder, kw, asset_dir):
    """the lean any-hit walk's EXACT phase (the reference's FP64 triangle test on a ray rebuilt from the queue entry: what settles the 0.1 %
    of leaf tests its f32 filter leaves undecided) with EVERY test sent through it (global option anyhit_filter_off): parked triangles, lanes
    that find a second one while the first is pending, rays that end with one pending -- same ray counts, same pixels as the oracle, and
    the frame of the filtered walk"""
    sp, rd = prepare(workloads.BUILDERS[builder](asset_dir, **kw))
    gs = gpu.Scene(sp)
    assert gs.query("lean_anyhit") == 1
    fb0, st0 = gs.render_frame(rd)
    gpu.global_option("anyhit_filter_off", 1)
    try:
        fb1, st1 = gs.render_frame(rd)
    finally:
        gpu.global_option("anyhit_filter_off", 0)
        gs.close()
    osc = oracle_ffi.OracleScene(sp)
    ref, rc = osc.render(rd)
    osc.close()
    assert st0.rays.as_dict() == rc.as_dict() and st1.rays.as_dict() == rc.as_dict()
    assert float(rel_err(fb1, ref).max()) <= REL_TOL and float(rel_err(fb0, ref).max()) <= REL_TOL
    assert float(rel_err(fb1, fb0).max()) <= 1e-5


def test_multi_device_frame_equals_single_device_frame(asset_dir):
    """fjgpu_render_frame_multi (the worker pool with GPUs for workers, src/fj_renderer.cc:747-791):
    two replicas -- on one device here -- deal the tiles k % 2, pack, peer-copy and scatter their
    slabs; the frame, the tile subsets and the summed ray counts are those of one device."""
    text = workloads.teapot(asset_dir, res=(100, 76), spp=(2, 2), mesh="tiny", extra=(("tilesize", (16, 16)),))
    sp, rd = prepare(text)
    gs = gpu.Scene(sp)
    one, st1 = gs.render_frame(rd)
    gs.close()
    n = gpu.tile_count(rd)
    assert n == 7 * 5
    for replicas in (2, 3):
        ms = gpu.MultiScene(sp, [0] * replicas)
        fb, sts = ms.render_frame(rd)
        assert float(rel_err(fb, one).max()) <= 1e-6       # (f32 atomics: the order of the adds differs run to run)
        total = {k: sum(getattr(s.rays, k) for s in sts) for k in ("camera", "shadow", "diffuse", "reflect", "refract")}
        assert total == st1.rays.as_dict()
        assert all(s.rays.camera > 0 for s in sts)         # every replica rendered its share
        # a tile subset (the cancel path of SiRenderScene): listed tiles as in the full frame, the rest 0
        ids = [0, 3, 4, 9, 17, 33, 34]
        sub, _ = ms.render_frame(rd, ids)
        mask = np.zeros(one.shape[:2], dtype=bool)
        for t in ids:
            x0, y0, x1, y1 = gpu.tile_rect(rd, t)
            mask[y0:y1, x0:x1] = True
        assert float(rel_err(sub[mask], one[mask]).max()) <= 1e-6 and not sub[~mask].any()
        ms.close()


def test_rccl_spelling_of_the_slab_exchange(asset_dir):
    """the C++ core's RCCL path (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd for the tile slabs of
    fjgpu_render_frame_multi, option "multi_exchange" 1): librccl.so loads through the core's own loader and the exchange's calls run on
    this GPU (a communicator of one rank that sends to itself: fjgpu_dev_rccl_selftest); with the option on, a one-device frame and a
    frame of replicas that share a device -- RCCL refuses two ranks on one GPU, the core keeps the peer copies there -- are unchanged;
    the multi-GPU form needs a node with several GPUs (the driver's scaling run uses the per-process form of bench.py --gpus N)"""
    gpu.rccl_selftest(0, 1 << 18)
    gpu.rccl_selftest(0, 1)
    with pytest.raises(gpu.GpuError):
        gpu.global_option("multi_exchange", 2)
    text = workloads.teapot(asset_dir, res=(100, 76), spp=(2, 2), mesh="tiny", extra=(("tilesize", (16, 16)),))
    sp, rd = prepare(text)
    gs = gpu.Scene(sp)
    one, st1 = gs.render_frame(rd)
    gpu.global_option("multi_exchange", 1)
    try:
        again, st2 = gs.render_frame(rd)
        assert float(rel_err(again, one).max()) <= 1e-6 and st2.rays.as_dict() == st1.rays.as_dict()
        ms = gpu.MultiScene(sp, [0, 0])
        fb, sts = ms.render_frame(rd)
        ms.close()
        assert float(rel_err(fb, one).max()) <= 1e-6
        assert sum(s.rays.total() for s in sts) == st1.rays.total()
    finally:
        gpu.global_option("multi_exchange", 0)
        gs.close()


def test_multi_device_frame_on_distinct_devices(asset_dir):
    """fjgpu_render_frame_multi on DISTINCT devices, wherever the node has more than one (the driver's scaling node; skipped on a one-GPU box):
    the assembled frame against device 0's own render, with the peer copies and with the RCCL exchange (one group after the join, collective
    over success: ADVICE round 5), and a tile subset shorter than the device list (devices without a tile neither send nor are waited for)"""
    import torch
    nd = torch.cuda.device_count()
    if nd < 2:
        pytest.skip("one GPU visible: the N-device exchange needs a node with several (replicas on one device: the test above)")
    text = workloads.teapot(asset_dir, res=(100, 76), spp=(2, 2), mesh="tiny", extra=(("tilesize", (16, 16)),))
    sp, rd = prepare(text)
    gs = gpu.Scene(sp)
    one, st1 = gs.render_frame(rd)
    gs.close()
    devs = list(range(min(nd, 8)))
    for exchange in (0, 1):
        gpu.global_option("multi_exchange", exchange)
        try:
            ms = gpu.MultiScene(sp, devs)
            fb, sts = ms.render_frame(rd)
            assert float(rel_err(fb, one).max()) <= 1e-6, exchange
            assert sum(s.rays.total() for s in sts) == st1.rays.total()
            # fewer tiles than devices (entry k of the list goes to device k % G): the last devices hold nothing
            ids = [5, 1] if len(devs) > 2 else [5]
            sub, _ = ms.render_frame(rd, ids)
            for t in ids:
                x0, y0, x1, y1 = gpu.tile_rect(rd, t)
                assert float(rel_err(sub[y0:y1, x0:x1], one[y0:y1, x0:x1]).max()) <= 1e-6
            ms.close()
        finally:
            gpu.global_option("multi_exchange", 0)


def test_si_callbacks_and_interrupts(asset_dir):
    """SiSetFrameReportCallback / SiSetTileReportCallback (src/fj_callback.h:15-98) on the GPU path:
    one tile_start / tile_done per tile, TileInfo.framebuffer readable in tile_done, and the
    reference's interrupt semantics -- a tile_start hook returning CALLBACK_INTERRUPT stops the
    queue (that tile and later ones are not rendered, earlier ones are), frame_done still fires and
    RenderScene succeeds (src/fj_renderer.cc:787-790,1098-1121); a frame_start interrupt fails the
    render before any tile (src/fj_renderer.cc:774-777)."""
    text = workloads.teapot(asset_dir, res=(96, 64), spp=(2, 2), mesh="tiny", extra=(("tilesize", (32, 32)),))
    host.run_scene_text(text, deferred=False)
    full = host.framebuffer(0)
    assert full[..., 3].max() > 0
    host.run_scene_text(text, deferred=True)
    ev = {"frame_start": 0, "frame_done": 0, "abort": 0, "start": [], "done": [], "sample": 0, "lit": []}

    def tile_done(info):
        ev["done"].append(info.region_id)
        x0, y0, x1, y1 = tuple(info.tile_region)
        assert info.total_region_count == 6 and info.framebuffer
        ev["lit"].append(float(np.abs(host.framebuffer(0)[y0:y1, x0:x1]).max()))
        return host.CALLBACK_CONTINUE

    def count(key):
        def fn(*a):
            ev[key] += 1
            return host.CALLBACK_CONTINUE
        return fn

    rc = host.render_with_callbacks(frame_start=count("frame_start"), frame_done=count("frame_done"), frame_abort=count("abort"),
                                    tile_start=lambda i: ev["start"].append(i.region_id) or 0, tile_done=tile_done,
                                    sample_done=count("sample"))
    assert rc == 0 and ev["frame_start"] == 1 and ev["frame_done"] == 1 and ev["abort"] == 0
    assert ev["start"] == list(range(6)) and sorted(ev["done"]) == list(range(6)) and ev["sample"] >= 1
    assert min(ev["lit"]) > 0                                # every finished tile was readable in its hook
    assert float(rel_err(host.framebuffer(0), full).max()) <= 1e-6

    # interrupt at the start of tile 2: tiles 0 and 1 are rendered, nothing else is touched
    host.run_scene_text(text, deferred=True)
    ev2 = {"done": [], "frame_done": 0}
    rc = host.render_with_callbacks(tile_start=lambda i: host.CALLBACK_INTERRUPT if i.region_id == 2 else host.CALLBACK_CONTINUE,
                                    tile_done=lambda i: ev2["done"].append(i.region_id) or 0,
                                    frame_done=lambda i: ev2.__setitem__("frame_done", ev2["frame_done"] + 1) or 0)
    part = host.framebuffer(0)
    assert rc == 0 and sorted(ev2["done"]) == [0, 1] and ev2["frame_done"] == 1
    assert float(rel_err(part[:32, :64], full[:32, :64]).max()) <= 1e-6
    assert not part[:32, 64:].any() and not part[32:].any()

    # interrupt at frame start: SI_FAIL, no tile hooks, framebuffer untouched (all zero after the resize)
    host.run_scene_text(text, deferred=True)
    ev3 = {"tiles": 0}
    rc = host.render_with_callbacks(frame_start=lambda i: host.CALLBACK_INTERRUPT,
                                    tile_start=lambda i: ev3.__setitem__("tiles", ev3["tiles"] + 1) or 0)
    assert rc == -1 and ev3["tiles"] == 0 and not host.framebuffer(0).any()

    # sample_done fires once per BATCH of tiles (fjgpu_set_batch_callback); an interrupt from it ends the frame
    # after that batch (integrate_samples -> LoopStatus::Cancel, src/fj_renderer.cc:1061-1121): tiles 0-3 of 6
    gpu.global_option("batch_tiles", 2)
    try:
        host.run_scene_text(text, deferred=True)
        ev4 = {"sample": 0, "done": []}
        rc = host.render_with_callbacks(sample_done=lambda: ev4.__setitem__("sample", ev4["sample"] + 1) or 0,
                                        tile_done=lambda i: ev4["done"].append(i.region_id) or 0)
        assert rc == 0 and ev4["sample"] == 3 and ev4["done"] == list(range(6))
        assert float(rel_err(host.framebuffer(0), full).max()) <= 1e-6
        host.run_scene_text(text, deferred=True)
        ev5 = {"sample": 0, "done": [], "frame_done": 0}

        def stop_after_two():
            ev5["sample"] += 1
            return host.CALLBACK_INTERRUPT if ev5["sample"] == 2 else host.CALLBACK_CONTINUE
        rc = host.render_with_callbacks(sample_done=stop_after_two, tile_done=lambda i: ev5["done"].append(i.region_id) or 0,
                                        frame_done=lambda i: ev5.__setitem__("frame_done", 1) or 0)
        part = host.framebuffer(0)
        assert rc == 0 and ev5["sample"] == 2 and ev5["done"] == [0, 1, 2, 3] and ev5["frame_done"] == 1
        assert float(rel_err(part[:32], full[:32]).max()) <= 1e-6 and float(rel_err(part[32:, :32], full[32:, :32]).max()) <= 1e-6
        assert not part[32:, 32:].any()
    finally:
        gpu.global_option("batch_tiles", 0)

    # a tile_start interrupt on the very FIRST tile: nothing is rendered, no tile_done, the frame still completes
    host.run_scene_text(text, deferred=True)
    ev6 = {"done": 0, "frame_done": 0}
    rc = host.render_with_callbacks(tile_start=lambda i: host.CALLBACK_INTERRUPT,
                                    tile_done=lambda i: ev6.__setitem__("done", ev6["done"] + 1) or 0,
                                    frame_done=lambda i: ev6.__setitem__("frame_done", 1) or 0)
    assert rc == 0 and ev6["done"] == 0 and ev6["frame_done"] == 1 and not host.framebuffer(0).any()
    host.close_scene()
