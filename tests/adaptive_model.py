"""Model of the adaptive grid sampler's two formulations (no product code, no oracle):

  sequential()  the reference's stack walk (src/fj_adaptive_grid_sampler.cc:124-170,224-343),
                as restated in oracle/restate/fjo_render.cc (AdaptiveGrid)
  levelsync()   the level-by-level formulation the device runs
                (fujiyama-renderer_amd/csrc/device/fjgpu_dev_adaptive.h)

tests/test_adaptive_model.py checks that they produce the same sample values and trace the
same samples on random lattices: the argument in fjgpu_dev_adaptive.h, executed.
"""
import numpy as np

def sequential(W0, H0, D, thr, color):
    div = 1 << D
    nx, ny = div * W0 + 1, div * H0 + 1
    data = np.zeros((ny, nx, 4)); state = -np.ones((ny, nx), int)
    traced = np.zeros((ny, nx), bool)
    stack = [(x * div, y * div, (x + 1) * div, (y + 1) * div) for y in range(H0) for x in range(W0)]
    while stack:
        x0, y0, x1, y1 = stack.pop()
        for (cx, cy) in ((x0, y0), (x1, y0), (x0, y1), (x1, y1)):
            if state[cy, cx] < 0:
                state[cy, cx] = 1; data[cy, cx] = color(cx, cy); traced[cy, cx] = True
        split = False
        if x1 - x0 >= 2:
            c = np.stack([data[y0, x0], data[y0, x1], data[y1, x0], data[y1, x1]])
            split = bool(((c.max(0) - c.min(0)) > thr).any())
        if split:
            xm, ym = (x0 + x1) // 2, (y0 + y1) // 2
            stack += [(x0, y0, xm, ym), (xm, y0, x1, ym), (x0, ym, xm, y1), (xm, ym, x1, y1)]
        else:
            c00, c10, c01, c11 = data[y0, x0].copy(), data[y0, x1].copy(), data[y1, x0].copy(), data[y1, x1].copy()
            for y in range(y0, y1 + 1):
                ty = 1. * (y - y0) / (y1 - y0)
                for x in range(x0, x1 + 1):
                    tx = 1. * (x - x0) / (x1 - x0)
                    l = (1 - ty) * c00 + ty * c01; r = (1 - ty) * c10 + ty * c11
                    data[y, x] = (1 - tx) * l + tx * r
                    if state[y, x] < 0: state[y, x] = 0
    return data, traced

def morton(ly, lx, D):
    m = 0
    for b in range(D):
        m |= ((lx >> b) & 1) << (2 * b) | ((ly >> b) & 1) << (2 * b + 1)
    return m

def levelsync(W0, H0, D, thr, color):
    div = 1 << D
    nx, ny = div * W0 + 1, div * H0 + 1
    st = [np.zeros((H0 << k, W0 << k), int) for k in range(D + 1)]
    seen = np.zeros((ny, nx, 4)); has_seen = np.zeros((ny, nx), bool)
    traced = np.zeros((ny, nx), bool); tval = np.zeros((ny, nx, 4))
    def key(k, cx, cy):
        X0, Y0 = cx >> k, cy >> k
        lx = (cx << (D - k)) & (div - 1); ly = (cy << (D - k)) & (div - 1)
        return ((Y0 * W0 + X0) << (2 * D)) | morton(ly, lx, D)
    def rect(k, cx, cy):
        s = div >> k
        return cx * s, cy * s, (cx + 1) * s, (cy + 1) * s
    def interp(k, cx, cy, fx, fy):
        x0, y0, x1, y1 = rect(k, cx, cy)
        c00, c10, c01, c11 = seen[y0, x0], seen[y0, x1], seen[y1, x0], seen[y1, x1]
        assert has_seen[y0, x0] and has_seen[y0, x1] and has_seen[y1, x0] and has_seen[y1, x1]
        ty = 1. * (fy - y0) / (y1 - y0); tx = 1. * (fx - x0) / (x1 - x0)
        l = (1 - ty) * c00 + ty * c01; r = (1 - ty) * c10 + ty * c11
        return (1 - tx) * l + tx * r
    def side(k1, cx, cy):   # classify level-k1 lattice cell: ('none'|'split'|'leaf', level, cx, cy)
        if cx < 0 or cy < 0 or cx >= (W0 << k1) or cy >= (H0 << k1): return ('none',)
        j = k1
        while True:
            s = st[j][cy, cx]
            if s == 2: assert j == k1; return ('split', j, cx, cy)
            if s == 1: return ('leaf', j, cx, cy)
            j -= 1; cx >>= 1; cy >>= 1
    for k in range(D + 1):
        s = div >> k
        # points
        for py in range((H0 << k) + 1):
            for px in range((W0 << k) + 1):
                fx, fy = px * s, py * s
                if k == 0:
                    traced[fy, fx] = True; continue
                ox, oy = px & 1, py & 1
                if not (ox or oy): continue
                if ox and oy:
                    if st[k - 1][py >> 1, px >> 1] == 2: traced[fy, fx] = True
                    continue
                if ox: sides = [side(k - 1, px >> 1, py // 2 - 1), side(k - 1, px >> 1, py // 2)]
                else: sides = [side(k - 1, px // 2 - 1, py >> 1), side(k - 1, px // 2, py >> 1)]
                kinds = [t[0] for t in sides]
                if 'split' not in kinds: continue
                if kinds.count('split') == 2 or 'none' in kinds: traced[fy, fx] = True; continue
                A = sides[kinds.index('split')]; L = sides[kinds.index('leaf')]
                if key(*L[1:]) > key(*A[1:]):
                    seen[fy, fx] = interp(L[1], L[2], L[3], fx, fy); has_seen[fy, fx] = True
                else:
                    traced[fy, fx] = True
        # trace
        for fy in range(0, ny, s):
            for fx in range(0, nx, s):
                if traced[fy, fx] and not has_seen[fy, fx]:
                    tval[fy, fx] = color(fx, fy); seen[fy, fx] = tval[fy, fx]; has_seen[fy, fx] = True
        # decide
        for cy in range(H0 << k):
            for cx in range(W0 << k):
                if k > 0 and st[k - 1][cy >> 1, cx >> 1] != 2: continue
                if k == D: st[k][cy, cx] = 1; continue
                x0, y0, x1, y1 = rect(k, cx, cy)
                c = np.stack([seen[y0, x0], seen[y0, x1], seen[y1, x0], seen[y1, x1]])
                st[k][cy, cx] = 2 if ((c.max(0) - c.min(0)) > thr).any() else 1
    # final
    data = np.zeros((ny, nx, 4))
    def leaf_of(fcx, fcy):
        for k in range(D + 1):
            cx, cy = fcx >> (D - k), fcy >> (D - k)
            if st[k][cy, cx] == 1: return (k, cx, cy)
        raise AssertionError
    for fy in range(ny):
        for fx in range(nx):
            best = None
            for dy in (-1, 0):
                for dx in (-1, 0):
                    cx, cy = fx + dx, fy + dy
                    if cx < 0 or cy < 0 or cx >= div * W0 or cy >= div * H0: continue
                    L = leaf_of(cx, cy)
                    x0, y0, x1, y1 = rect(*L)
                    if (fx == x0 or fx == x1) and (fy == y0 or fy == y1): continue
                    if best is None or key(*L) < key(*best): best = L
            if best is None:
                assert has_seen[fy, fx] and traced[fy, fx]
                data[fy, fx] = seen[fy, fx]
            else:
                data[fy, fx] = interp(best[0], best[1], best[2], fx, fy)
    return data, traced
