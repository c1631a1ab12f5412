"""Seeded synthetic assets for the BASELINE.json workloads.

The reference's scenes read Stanford PLY scans and HDR environment maps that
are not in its repository (reference INSTALL:56-65); there is no network here,
so every asset is synthesised from a fixed seed with matched triangle counts:

  dragon-class   closed bumpy genus-0 surface, 1900 x 1900 quads -> 7.22 M tris
  buddha-class   738 x 738  -> 1.09 M tris
  bunny-class    186 x 186  -> 69 k tris
  teapot-class   56 x 56    -> 6.3 k tris
  floor.ply      10 x 10 quads spanning +-10 at y = 0
  dome.ply       hemisphere 32 x 16 quads, r = 100, with uv

Files are written in the formats the reference's own loaders accept:
  * PLY: `x y z [uv1 uv2]` vertex properties + `vertex_indices` face lists
    (procedures/stanfordply_procedure/ply2mesh.cc:32-49,76-139), binary
    little endian, float32 coordinates like the Stanford scans;
  * .mip: "MIPM", version 1, width, height, nchannels, tilesize, then
    tilesize x tilesize x nchannels float32 tiles in row-major tile order
    (src/fj_mipmap.cc:124-151, 285-312).
"""
import os
import struct

import numpy as np

SEED = 20260928


def write_ply(path, verts, faces_quads=None, faces_tris=None, uv=None):
    """verts [n,3] float32; quads [m,4] / tris [k,3] int32; uv [n,2] float32."""
    verts = np.asarray(verts, dtype="<f4")
    n = verts.shape[0]
    nq = 0 if faces_quads is None else len(faces_quads)
    nt = 0 if faces_tris is None else len(faces_tris)
    hdr = ["ply", "format binary_little_endian 1.0", "element vertex %d" % n,
           "property float x", "property float y", "property float z"]
    if uv is not None:
        hdr += ["property float uv1", "property float uv2"]
    hdr += ["element face %d" % (nq + nt),
            "property list uchar int vertex_indices", "end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(hdr) + "\n").encode("ascii"))
        if uv is not None:
            rec = np.empty(n, dtype=[("p", "<f4", 3), ("t", "<f4", 2)])
            rec["p"] = verts
            rec["t"] = np.asarray(uv, dtype="<f4")
            rec.tofile(f)
        else:
            verts.tofile(f)
        if nt:
            rec = np.empty(nt, dtype=[("n", "u1"), ("v", "<i4", 3)])
            rec["n"] = 3
            rec["v"] = faces_tris
            rec.tofile(f)
        if nq:
            rec = np.empty(nq, dtype=[("n", "u1"), ("v", "<i4", 4)])
            rec["n"] = 4
            rec["v"] = faces_quads
            rec.tofile(f)


def bumpy_sphere(nu, nv, seed=SEED, bumps=12, amp=0.18, radius=1.0):
    """Closed genus-0 surface r(theta,phi) = 1 + sum a_k sin(f_k theta + p_k) sin(g_k phi)^2...

    nu segments around (periodic), nv segments pole to pole; poles are single
    vertices with triangle fans, the rest quads: nu*(nv-2) quads + 2*nu tris
    -> 2*nu*(nv-1) triangles. Rests on y = 0.
    """
    rng = np.random.RandomState(seed)
    f = rng.randint(1, 9, size=bumps)
    g = rng.randint(1, 9, size=bumps)
    p = rng.uniform(0, 2 * np.pi, size=bumps)
    q = rng.uniform(0, 2 * np.pi, size=bumps)
    a = amp * rng.uniform(0.2, 1.0, size=bumps) / np.sqrt(f * g)

    theta = (np.arange(nu, dtype=np.float64) / nu) * 2 * np.pi
    phi = (np.arange(1, nv, dtype=np.float64) / nv) * np.pi          # interior rings
    T, P = np.meshgrid(theta, phi)                                    # [nv-1, nu]
    r = np.ones_like(T)
    for k in range(bumps):
        # sin(phi)^2 envelope keeps r single-valued at the poles
        r += a[k] * np.sin(f[k] * T + p[k]) * np.cos(g[k] * P + q[k]) * np.sin(P) ** 2
    x = r * np.sin(P) * np.cos(T)
    y = r * np.cos(P)
    z = r * np.sin(P) * np.sin(T)
    ring = np.stack([x, y, z], axis=-1).reshape(-1, 3)
    verts = np.concatenate([[[0.0, 1.0, 0.0]], ring, [[0.0, -1.0, 0.0]]], axis=0)
    verts[:, 1] -= verts[:, 1].min()
    if radius != 1.0:
        verts *= radius

    nr = nv - 1
    idx = 1 + np.arange(nr * nu).reshape(nr, nu)
    nxt = np.roll(idx, -1, axis=1)
    # outward-facing winding
    quads = np.stack([idx[:-1], nxt[:-1], nxt[1:], idx[1:]], axis=-1).reshape(-1, 4)
    top = np.stack([np.zeros(nu, dtype=np.int64), nxt[0], idx[0]], axis=-1)
    south = 1 + nr * nu
    bot = np.stack([np.full(nu, south), idx[-1], nxt[-1]], axis=-1)
    tris = np.concatenate([top, bot], axis=0)
    return verts.astype(np.float32), quads.astype(np.int32), tris.astype(np.int32)


def floor_grid(n=10, half=10.0):
    g = np.linspace(-half, half, n + 1)
    X, Z = np.meshgrid(g, g)
    verts = np.stack([X, np.zeros_like(X), Z], axis=-1).reshape(-1, 3)
    idx = np.arange((n + 1) * (n + 1)).reshape(n + 1, n + 1)
    # winding so that the geometric normal is +y
    quads = np.stack([idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]], axis=-1).reshape(-1, 4)
    return verts.astype(np.float32), quads.astype(np.int32)


def dome(nu=32, nv=16, radius=100.0):
    """Upper hemisphere seen from inside, with lat-long uv."""
    theta = np.linspace(0, 2 * np.pi, nu + 1)
    phi = np.linspace(0, 0.5 * np.pi, nv + 1)              # 0 = zenith
    T, P = np.meshgrid(theta, phi)
    x = radius * np.sin(P) * np.cos(T)
    y = radius * np.cos(P)
    z = radius * np.sin(P) * np.sin(T)
    verts = np.stack([x, y, z], axis=-1).reshape(-1, 3)
    uv = np.stack([T / (2 * np.pi), 1.0 - P / np.pi], axis=-1).reshape(-1, 2)
    idx = np.arange((nv + 1) * (nu + 1)).reshape(nv + 1, nu + 1)
    quads = np.stack([idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]], axis=-1).reshape(-1, 4)
    return verts.astype(np.float32), quads.astype(np.int32), uv.astype(np.float32)


def sky_image(width=1024, height=512, seed=SEED):
    """RGB float 'HDR': vertical gradient + three gaussian suns, values 0..50."""
    rng = np.random.RandomState(seed + 1)
    v = np.linspace(1.0, 0.0, height)[:, None]
    u = np.linspace(0.0, 1.0, width, endpoint=False)[None, :]
    img = np.empty((height, width, 3), dtype=np.float64)
    img[..., 0] = 0.25 + 0.45 * (1 - v)
    img[..., 1] = 0.35 + 0.40 * (1 - v)
    img[..., 2] = 0.55 + 0.35 * v
    for _ in range(3):
        cu, cv = rng.uniform(0.1, 0.9), rng.uniform(0.55, 0.9)
        s = rng.uniform(0.01, 0.03)
        du = np.minimum(np.abs(u - cu), 1 - np.abs(u - cu))
        blob = np.exp(-(du ** 2 + (v - cv) ** 2) / (2 * s * s))
        img += (rng.uniform(10, 50) * blob)[..., None] * np.array([1.0, 0.9, 0.7])
    return img.astype(np.float32)


def rock_image(size=512, seed=SEED):
    rng = np.random.RandomState(seed + 2)
    img = np.zeros((size, size), dtype=np.float64)
    yy, xx = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    for o in range(5):
        fx, fy = rng.randint(1, 6) * 2 ** o, rng.randint(1, 6) * 2 ** o
        img += np.sin(2 * np.pi * (fx * xx + fy * yy) / size + rng.uniform(0, 6.28)) / 2 ** o
    img = (img - img.min()) / (img.max() - img.min())
    return np.stack([0.3 + 0.6 * img, 0.25 + 0.5 * img, 0.2 + 0.4 * img], axis=-1).astype(np.float32)


def write_mip(path, img, tilesize=64):
    """img [h,w,c] float32, h and w multiples of tilesize."""
    img = np.ascontiguousarray(img, dtype="<f4")
    if img.ndim == 2:
        img = img[..., None]
    h, w, c = img.shape
    assert h % tilesize == 0 and w % tilesize == 0
    with open(path, "wb") as f:
        f.write(b"MIPM")
        f.write(struct.pack("<iiiii", 1, w, h, c, tilesize))
        tiles = img.reshape(h // tilesize, tilesize, w // tilesize, tilesize, c).transpose(0, 2, 1, 3, 4)
        np.ascontiguousarray(tiles).tofile(f)


def write_obj_groups(path, verts, quads, tris):
    """Wavefront OBJ with FACE GROUPS and no normals (the reader accumulates point normals): coordinates are f32 values printed
    exactly, faces before the first `g` stay in the default group, then bands A B C D by height, and a second stretch of A."""
    v = np.asarray(verts, dtype=np.float32)
    faces = [list(q) for q in quads] + [list(t) for t in tris]
    ymid = [float(np.mean([v[i][1] for i in f])) for f in faces]
    lo, hi = min(ymid), max(ymid)
    band = [min(5, int(6 * (y - lo) / (hi - lo + 1e-12))) for y in ymid]
    names = {0: None, 1: "A", 2: "B", 3: "C", 4: "D", 5: "A"}
    with open(path, "w") as f:
        f.write("# synthetic: face groups (default, A, B, C, D, A again)\n")
        for p in v:
            f.write("v %.17g %.17g %.17g\n" % (float(p[0]), float(p[1]), float(p[2])))
        for b in range(6):
            if names[b]:
                f.write("g %s extra_word_ignored\n" % names[b])
            for face, fb in zip(faces, band):
                if fb == b:
                    f.write("f " + " ".join(str(i + 1) for i in face) + "\n")


def write_obj_normals(path, verts, quads, tris, seed=SEED):
    """Wavefront OBJ with `vn` and per-CORNER normal indices: smooth normals on most faces, the face's own normal on every third
    face (a crease: one point, several normals), texture coordinates that the reader drops, corners written as v//vn, v/vt/vn and
    with NEGATIVE (relative) indices, two groups.  Coordinates are arbitrary doubles (not f32 values)."""
    rng = np.random.RandomState(seed + 7)
    v = np.asarray(verts, dtype=np.float64) * (1.0 / 3.0) + [.01, .02, .03]
    faces = [list(q) for q in quads] + [list(t) for t in tris]
    n_smooth = v - v.mean(axis=0)
    n_smooth /= np.linalg.norm(n_smooth, axis=1, keepdims=True)
    normals = [tuple(n) for n in n_smooth]
    with open(path, "w") as f:
        f.write("# synthetic: per-corner normals\n")
        for p in v:
            f.write("v %.17g %.17g %.17g 1.0\n" % tuple(p))
        for k in range(len(v)):
            f.write("vt %.6f %.6f\n" % (rng.uniform(), rng.uniform()))
        for n in normals:
            f.write("vn %.17g %.17g %.17g\n" % n)
        nv, nn = len(v), len(normals)
        for k, face in enumerate(faces):
            if k == len(faces) // 2:
                f.write("g upper\n")
            if k % 3 == 0:
                a, b, c = v[face[0]], v[face[1]], v[face[2]]
                ng = np.cross(b - a, c - a)
                ng = ng / max(np.linalg.norm(ng), 1e-300) * (1.5 if k % 2 else 1.0)       # (unnormalised on purpose: the reference does not renormalise here)
                f.write("vn %.17g %.17g %.17g\n" % tuple(ng))
                nn += 1
                corners = ["%d//%d" % (i + 1, -1) for i in face]                               # the normal just written, relative index
            elif k % 3 == 1:
                corners = ["%d/%d/%d" % (i + 1, i + 1, i + 1) for i in face]
            else:
                corners = ["%d//%d" % (i - nv, i + 1) for i in face]                           # negative position indices
            f.write("f " + " ".join(corners) + "\n")


MESH_CLASSES = {
    # name: (nu, nv) -> 2*nu*(nv-1) triangles
    "dragon": (1900, 1901),    # 7 220 000
    "buddha": (738, 738),      # 1 087 812
    "bunny": (186, 188),       # 69 564
    "teapot": (56, 57),        # 6 272
    "tiny": (12, 9),           # 192 (unit tests)
    "small": (100, 101),       # 20 000 (golden traversal vectors)
    # furred meshes (config 5): CurveGeneratorProcedure grows int(1e5 * area) curves per face,
    # so the curve count is set by the surface area, i.e. by the radius
    "furbunny": (186, 188, 0.5),   # 69 564 tris, area ~3.3  -> ~330 k curves
    "furball": (24, 17, 0.12),     # 768 tris,    area ~0.19 -> ~19 k curves (tests)
}


def ensure_assets(root, meshes=("teapot",), textures=True):
    """Generate (once) the named assets under `root`; returns dict name -> path."""
    os.makedirs(root, exist_ok=True)
    out = {}
    for name in meshes:
        path = os.path.join(root, "%s.ply" % name)
        if not os.path.exists(path):
            cls = MESH_CLASSES[name]
            v, q, t = bumpy_sphere(cls[0], cls[1], seed=SEED + len(name), radius=cls[2] if len(cls) > 2 else 1.0)
            tmp = "%s.%d.tmp" % (path, os.getpid())
            write_ply(tmp, v, faces_quads=q, faces_tris=t)
            os.replace(tmp, path)
        out[name] = path
    # OBJ twins of the smallest mesh (WavefrontObjProcedure: face groups / per-corner normals)
    for key, writer in (("tiny_groups_obj", write_obj_groups), ("tiny_normals_obj", write_obj_normals)):
        if key in meshes:
            continue
        path = os.path.join(root, key.replace("_obj", ".obj"))
        if "tiny" in meshes and not os.path.exists(path):
            cls = MESH_CLASSES["tiny"]
            v, q, t = bumpy_sphere(cls[0], cls[1], seed=SEED + len("tiny"))
            tmp = "%s.%d.tmp" % (path, os.getpid())
            writer(tmp, v, q, t)
            os.replace(tmp, path)
        if os.path.exists(path):
            out[key] = path
    # (every file appears under its name complete or not at all: several ranks may ask for the same assets)
    path = os.path.join(root, "floor.ply")
    if not os.path.exists(path):
        v, q = floor_grid()
        tmp = "%s.%d.tmp" % (path, os.getpid())
        write_ply(tmp, v, faces_quads=q)
        os.replace(tmp, path)
    out["floor"] = path
    path = os.path.join(root, "dome.ply")
    if not os.path.exists(path):
        v, q, uv = dome()
        tmp = "%s.%d.tmp" % (path, os.getpid())
        write_ply(tmp, v, faces_quads=q, uv=uv)
        os.replace(tmp, path)
    out["dome"] = path
    if textures:
        for name, image in (("sky", sky_image), ("rock", rock_image)):
            path = os.path.join(root, name + ".mip")
            if not os.path.exists(path):
                tmp = "%s.%d.tmp" % (path, os.getpid())
                write_mip(tmp, image())
                os.replace(tmp, path)
            out[name] = path
    return out
