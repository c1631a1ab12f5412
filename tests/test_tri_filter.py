"""The f32 triangle filter of the lean any-hit walk (fujiyama-renderer_amd/csrc/device/fjgpu_tri_filter.h) must never decide against the
reference's FP64 test: FJ_TRI_MISS only where TriRayIntersect (src/fj_triangle.cc:81-153) misses or reports a t outside [tmin, tmax],
FJ_TRI_HIT only where it hits inside the range.  The header the kernels compile is built for the host (lib/libfj_tri_filter_host.so, ROCm
clang++: the same packed statements, fmaf = one rounding like v_pk_fma_f32) and held against the oracle's pinned TriRayIntersect on random
and adversarial (ray, triangle) pairs: targets 1e-12 .. 1e-3 from edges and vertices, origins on the triangle (t around tmin), grazing rays,
unnormalised directions, coordinates from 1e-3 to 1e4, triangles from 1e-6 to 0.3 of the scene, tmax at the hit distance, huge and infinite.
(On the device a -DFJ_TRI_FILTER_VALIDATE build re-runs the exact test behind every verdict of whole frames: profiles/r06_anyhit_f32_filter.txt.)"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MISS, HIT, MAYBE = 0, 1, 2


def _lib():
    path = os.path.join(ROOT, "fujiyama-renderer_amd", "lib", "libfj_tri_filter_host.so")
    if not os.path.exists(path):
        pytest.skip("lib/libfj_tri_filter_host.so is not built (needs the ROCm clang++)")
    return C.CDLL(path)


def _run(tris, rays, bound, want_hit=1):
    n = len(tris)
    tris = np.ascontiguousarray(tris, np.float32)
    rays = np.ascontiguousarray(rays, np.float64)
    bound = np.ascontiguousarray(bound, np.float32)
    out = np.empty(n, np.int8)
    _lib().fj_tri_filter_batch(C.c_int64(n), tris.ctypes.data_as(C.c_void_p), rays.ctypes.data_as(C.c_void_p), bound.ctypes.data_as(C.c_void_p),
                               want_hit, out.ctypes.data_as(C.c_void_p))
    inp = np.ascontiguousarray(np.concatenate([tris.astype(np.float64), rays[:, :6]], axis=1))
    hit = np.empty(n, np.int32)
    tuv = np.empty((n, 3), np.float64)
    oracle_ffi.lib().fjo_tri_ray(C.c_int(n), inp.ctypes.data_as(C.c_void_p), hit.ctypes.data_as(C.c_void_p), tuv.ctypes.data_as(C.c_void_p))
    exact = (hit != 0) & (rays[:, 6] <= tuv[:, 0]) & (tuv[:, 0] <= rays[:, 7])
    return out, exact


def _pairs(n, rng, scale, tri_size, mode):
    c = rng.uniform(-1, 1, (n, 3)) * scale
    tri = (c[:, None, :] + rng.normal(0, 1, (n, 3, 3)) * tri_size * scale).astype(np.float32)
    t64 = tri.astype(np.float64)
    bu = rng.uniform(-.3, 1.3, n)
    bv = rng.uniform(-.3, 1.3, n)
    k = rng.integers(0, 6, n)
    eps = 10.0 ** rng.uniform(-12, -3, n) * rng.choice([-1, 1], n)
    bu = np.where(k == 1, eps, bu)
    bv = np.where(k == 2, eps, bv)
    bv = np.where(k == 3, 1 - bu + eps, bv)
    bu = np.where(k == 4, eps, bu)
    bv = np.where(k == 4, eps * rng.uniform(-1, 1, n), bv)
    target = t64[:, 0] + bu[:, None] * (t64[:, 1] - t64[:, 0]) + bv[:, None] * (t64[:, 2] - t64[:, 0])
    dirn = rng.normal(0, 1, (n, 3))
    dirn /= np.linalg.norm(dirn, axis=1)[:, None]
    if mode == "graze":
        nrm = np.cross(t64[:, 1] - t64[:, 0], t64[:, 2] - t64[:, 0])
        nrm = nrm / np.maximum(np.linalg.norm(nrm, axis=1)[:, None], 1e-300)
        dirn = dirn - (dirn * nrm).sum(1)[:, None] * nrm * (1 - 10.0 ** rng.uniform(-8, -1, n))[:, None]
        dirn /= np.maximum(np.linalg.norm(dirn, axis=1)[:, None], 1e-300)
    dist = 10.0 ** rng.uniform(-6, 2, n) * scale
    if mode == "surface":
        dist = 10.0 ** rng.uniform(-9, -2, n) * rng.choice([-1, 1, 1], n)       # origin (almost) on the triangle: t around tmin
    o = target - dirn * dist[:, None]
    dscale = 10.0 ** rng.uniform(-2, 2, n) if mode == "dscale" else np.ones(n)
    d = dirn * dscale[:, None]
    tmax = np.abs(dist) / dscale * rng.choice([.5, .999999, 1.0, 1.000001, 2., 10.], n)
    big = rng.uniform(0, 1, n)
    tmax = np.where(big < .4, 10.0 ** rng.uniform(-3, 3, n) * scale, tmax)
    tmax = np.where(big > .9, rng.choice([3.4e38, 3.5e38, 1e300, np.inf], n), tmax)      # dome-light samples sit at dir x FLT_MAX
    rays = np.concatenate([o, d, np.full((n, 1), 1e-4), tmax[:, None]], axis=1)
    bound = (np.abs(tri).reshape(n, -1).max(1) * rng.uniform(1, 4, n)).astype(np.float32)
    return tri.reshape(n, 9), rays, bound


@pytest.mark.parametrize("mode", ["plain", "graze", "surface", "dscale"])
def test_filter_never_decides_against_the_exact_test(mode):
    rng = np.random.default_rng(20260930 + len(mode))
    decided = 0
    for scale in (1e-3, 1., 50., 1e4):
        for ts in (1e-6, 1e-4, 1e-3, 1e-2, .3):
            tri, rays, bound = _pairs(60000, rng, scale, ts, mode)
            out, exact = _run(tri, rays, bound)
            assert not ((out == MISS) & exact).any(), (mode, scale, ts, "a hit of the exact test was rejected")
            assert not ((out == HIT) & ~exact).any(), (mode, scale, ts, "a miss of the exact test was accepted")
            decided += int((out != MAYBE).sum())
            # closest-hit form: hits are never claimed
            out2, _ = _run(tri[:5000], rays[:5000], bound[:5000], want_hit=0)
            assert not (out2 == HIT).any() and np.array_equal(out2 == MISS, out[:5000] == MISS)
    assert decided > 0.5 * 20 * 60000          # (and it does decide: these pairs are adversarial, production sees 99.9 % decided)


def test_filter_settles_ordinary_pairs_and_poisons_what_it_cannot_bound():
    rng = np.random.default_rng(7)
    n = 200000
    c = rng.uniform(-1, 1, (n, 3))
    tri = (c[:, None, :] + rng.normal(0, 1, (n, 3, 3)) * 1e-3).astype(np.float32)
    t64 = tri.astype(np.float64)
    bu, bv = rng.uniform(-1, 2, n), rng.uniform(-1, 2, n)
    target = t64[:, 0] + bu[:, None] * (t64[:, 1] - t64[:, 0]) + bv[:, None] * (t64[:, 2] - t64[:, 0])
    dirn = rng.normal(0, 1, (n, 3))
    dirn /= np.linalg.norm(dirn, axis=1)[:, None]
    o = target - dirn * (10.0 ** rng.uniform(-3, 0, n))[:, None]
    rays = np.concatenate([o, dirn, np.full((n, 1), 1e-4), np.full((n, 1), 1e3)], axis=1)
    out, exact = _run(tri.reshape(n, 9), rays, np.full(n, 2., np.float32))
    assert not ((out == MISS) & exact).any() and not ((out == HIT) & ~exact).any()
    assert (out == MAYBE).mean() < 2e-3                       # a dragon-class mesh seen from inside its own bounds
    assert ((out == HIT) & exact).sum() > .98 * exact.sum()
    # magnitudes outside the error analysis (|d| > 2^30, |o| + bound > 2^29) and NaN: every verdict undecided
    huge = rays.copy()
    huge[:, 3:6] *= 4e9
    out, _ = _run(tri.reshape(n, 9)[:1000], huge[:1000], np.full(1000, 2., np.float32))
    assert (out == MAYBE).all()
    out, _ = _run(tri.reshape(n, 9)[:1000], rays[:1000], np.full(1000, 6e8, np.float32))
    assert (out == MAYBE).all()
    bad = rays[:1000].copy()
    bad[::2, 0] = np.nan
    bad[1::2, 7] = np.nan
    out, exact = _run(tri.reshape(n, 9)[:1000], bad, np.full(1000, 2., np.float32))
    # (a NaN origin or tmax: never a claimed hit; a miss only where the exact test misses too -- |det| < EPSILON needs neither)
    assert not (out == HIT).any() and not ((out == MISS) & exact).any()
