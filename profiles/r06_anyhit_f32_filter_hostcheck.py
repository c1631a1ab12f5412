import ctypes, numpy as np, sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import oracle_ffi
L = ctypes.CDLL('/tmp/tf/libtf.so')
O = oracle_ffi.lib()
def run(tris, rays, bound, want_hit=1):
    n = len(tris)
    tris = np.ascontiguousarray(tris, np.float32); rays = np.ascontiguousarray(rays, np.float64); bound = np.ascontiguousarray(bound, np.float32)
    out = np.empty(n, np.int8)
    L.fj_tri_filter_batch(ctypes.c_int64(n), tris.ctypes.data_as(ctypes.c_void_p), rays.ctypes.data_as(ctypes.c_void_p), bound.ctypes.data_as(ctypes.c_void_p), want_hit, out.ctypes.data_as(ctypes.c_void_p))
    inp = np.concatenate([tris.astype(np.float64), rays[:, :6]], axis=1); inp = np.ascontiguousarray(inp)
    hit = np.empty(n, np.int32); tuv = np.empty((n, 3), np.float64)
    O.fjo_tri_ray(ctypes.c_int(n), inp.ctypes.data_as(ctypes.c_void_p), hit.ctypes.data_as(ctypes.c_void_p), tuv.ctypes.data_as(ctypes.c_void_p))
    exact = (hit != 0) & (rays[:, 6] <= tuv[:, 0]) & (tuv[:, 0] <= rays[:, 7])
    return out, exact
def gen(n, rng, scale, tri_size, mode):
    c = rng.uniform(-1, 1, (n, 3)) * scale
    tri = (c[:, None, :] + rng.normal(0, 1, (n, 3, 3)) * tri_size[:, None, None]).astype(np.float32)
    t64 = tri.astype(np.float64)
    # target point: barycentric, sometimes near / on edges and vertices
    bu = rng.uniform(-.3, 1.3, n); bv = rng.uniform(-.3, 1.3, n)
    k = rng.integers(0, 6, n)
    eps = 10.0 ** rng.uniform(-12, -3, n) * rng.choice([-1, 1], n)
    bu = np.where(k == 1, eps, bu); bv = np.where(k == 2, eps, bv)
    bv = np.where(k == 3, 1 - bu + eps, bv)
    bu = np.where(k == 4, eps, bu); bv = np.where(k == 4, eps * rng.uniform(-1, 1, n), bv)
    target = t64[:, 0] + bu[:, None] * (t64[:, 1] - t64[:, 0]) + bv[:, None] * (t64[:, 2] - t64[:, 0])
    dirn = rng.normal(0, 1, (n, 3)); dirn /= np.linalg.norm(dirn, axis=1)[:, None]
    if mode == 'graze':
        nrm = np.cross(t64[:, 1] - t64[:, 0], t64[:, 2] - t64[:, 0]); nl = np.linalg.norm(nrm, axis=1)[:, None]; nrm = nrm / np.maximum(nl, 1e-300)
        dirn = dirn - (dirn * nrm).sum(1)[:, None] * nrm * (1 - 10.0 ** rng.uniform(-8, -1, n))[:, None]
        dirn /= np.maximum(np.linalg.norm(dirn, axis=1)[:, None], 1e-300)
    dist = 10.0 ** rng.uniform(-6, 2, n) * scale
    if mode == 'surface': dist = 10.0 ** rng.uniform(-9, -2, n) * rng.choice([-1, 1, 1], n)   # origin (almost) on the triangle: t around tmin
    o = target - dirn * dist[:, None]
    dscale = 10.0 ** rng.uniform(-2, 2, n) if mode == 'dscale' else np.ones(n)
    d = dirn * dscale[:, None]
    tmax = np.abs(dist) / dscale * rng.choice([.5, .999999, 1.0, 1.000001, 2., 10.], n)
    tmax = np.where(rng.uniform(0, 1, n) < .5, 10.0 ** rng.uniform(-3, 3, n) * scale, tmax)
    rays = np.concatenate([o, d, np.full((n, 1), 1e-4), tmax[:, None]], axis=1)
    bound = (np.abs(tri).reshape(n, -1).max(1) * rng.uniform(1, 4, n)).astype(np.float32)
    return tri.reshape(n, 9), rays, bound
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
tot = {}
for mode in ['plain', 'graze', 'surface', 'dscale']:
    for scale in [1e-3, 1., 50., 1e4]:
        for ts in [1e-6, 1e-4, 1e-3, 1e-2, .3]:
            n = 400000
            tri, rays, bound = gen(n, rng, scale, np.full(n, ts * scale), mode)
            out, exact = run(tri, rays, bound)
            bad_miss = int(((out == 0) & exact).sum()); bad_hit = int(((out == 1) & ~exact).sum())
            print("%-8s scale %-7g tri %-7g: miss %.3f hit %.3f maybe %.4f | exact hits %.3f | WRONG miss %d hit %d" % (mode, scale, ts, (out == 0).mean(), (out == 1).mean(), (out == 2).mean(), exact.mean(), bad_miss, bad_hit), flush=True)
            assert bad_miss == 0 and bad_hit == 0
