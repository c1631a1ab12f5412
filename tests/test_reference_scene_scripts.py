"""The reference's own scene SCRIPTS run unchanged through the Python-3 emitter.

north_star: "scenes/*.py ... load unchanged".  The reference's emitter
(tools/python_api/fujiyama.py:137-333) is Python 2 and only formats commands; the build's
counterpart is fujiyama-renderer_amd/fujiyama.py.  Here the scripts under /root/reference/scenes
are executed AS THEY ARE with `import fujiyama` resolving to that module, and

  * for the four scenes the reference maintains by hand as `.scn` twins
    (scenes/{happy_buddhas,xyzrgb_dragon,furry_bunny}.scn; teapot.scn has no .py), the emitted
    command stream equals the twin -- modulo comments, whitespace, the `.so` suffix of plugin
    paths, float digits and the texture path (`.hdr` -> converted `.mip`);
  * every in-scope script of the directory (motion blur, area lights, ...)
    emits a stream the product's command parser accepts up to the first asset it would read
    (grammar, arity, names, plugin names, property names and types).

Build container only: the scripts are read from /root/reference at test time and nothing of
them is copied into the repository or travels to the GPU box.
"""
import os
import runpy
import sys

import pytest

from fujiyama_renderer_amd import fujiyama as emitter
from fujiyama_renderer_amd import host

REF_SCENES = "/root/reference/scenes"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_SCENES), reason="needs the reference tree (build container)")

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _write_hdr(path):
    from test_host_boundary import _write_hdr as w
    import numpy as np
    rng = np.random.RandomState(3)
    w(path, rng.uniform(0, 4, size=(64, 128, 3)).astype(np.float32), False)


def run_script(name, tmp_path, monkeypatch, argv=()):
    """exec scenes/<name>.py with the Py3 emitter as `fujiyama`; returns the emitted text.
    The script runs in <tmp>/scenes/x so that its relative asset paths (../../hdr/*.hdr) resolve
    to synthetic stand-ins: NewTexture really converts a .hdr on the way, as the reference's does."""
    cwd = tmp_path / "scenes" / "x"
    cwd.mkdir(parents=True, exist_ok=True)
    src = open(os.path.join(REF_SCENES, name + ".py")).read()
    (tmp_path / "hdr").mkdir(exist_ok=True)
    import re
    for hdr in set(re.findall(r"\.\./\.\./hdr/([\w\-.]+\.hdr)", src)):
        if not (tmp_path / "hdr" / hdr).exists():
            _write_hdr(str(tmp_path / "hdr" / hdr))
    captured = {}

    def fake_run(self):
        captured["text"] = self.text()
        return 0
    monkeypatch.setattr(emitter.SceneInterface, "Run", fake_run)
    monkeypatch.setattr(emitter.SceneInterface, "Print", fake_run)
    monkeypatch.setitem(sys.modules, "fujiyama", emitter)
    monkeypatch.setattr(sys, "argv", [name + ".py"] + list(argv))
    monkeypatch.chdir(cwd)
    runpy.run_path(os.path.join(REF_SCENES, name + ".py"), run_name="__main__")
    assert "text" in captured, "the script never called Run() / Print()"
    return captured["text"]


def _num(tok):
    try:
        return float(tok)
    except ValueError:
        return None


def normalise(text):
    """command lines as token lists: comments and blank lines dropped (tools/scene_parser/parser.cc:52-59),
    plugin paths without their DSO suffix (the reference's OpenPlugin appends it, fujiyama.py:137-151),
    texture paths as the image's base name (the emitter converts foo.hdr to <temp>/<uuid>_foo.mip)"""
    out = []
    for line in text.splitlines():
        tok = line.split()
        if not tok or tok[0].startswith("#"):
            continue
        if tok[0] == "OpenPlugin" and tok[2].endswith(".so"):
            tok[2] = tok[2][:-3]
        if tok[0] == "NewTexture":
            base = os.path.splitext(os.path.basename(tok[2]))[0]
            tok[2] = base.split("_", 1)[1] if len(base) > 37 and base[36] == "_" else base      # <uuid4>_<name>
        out.append(tok)
    return out


def same_stream(a, b):
    assert len(a) == len(b), (len(a), len(b))
    for la, lb in zip(a, b):
        assert len(la) == len(lb), (la, lb)
        for ta, tb in zip(la, lb):
            if ta == tb:
                continue
            na, nb = _num(ta), _num(tb)
            assert na is not None and nb is not None, (la, lb)
            assert abs(na - nb) <= 1e-9 * max(1.0, abs(na), abs(nb)), (la, lb)


@pytest.mark.parametrize("name", ["happy_buddhas", "xyzrgb_dragon", "furry_bunny"])
def test_reference_scripts_emit_their_hand_maintained_scn_twins(name, tmp_path, monkeypatch):
    emitted = normalise(run_script(name, tmp_path, monkeypatch))
    twin = normalise(open(os.path.join(REF_SCENES, name + ".scn")).read())
    same_stream(emitted, twin)
    assert len(emitted) > 100 and emitted[-2][0] == "RenderScene" and emitted[-1][0] == "SaveFrameBuffer"


def test_command_line_overrides_of_the_reference_scripts(tmp_path, monkeypatch):
    """-R / -S as the reference's emitter applies them: two SetProperty2 lines in front of RenderScene
    (tools/python_api/fujiyama.py:153-166); -P prints instead of running"""
    base = normalise(run_script("xyzrgb_dragon", tmp_path, monkeypatch))
    over = normalise(run_script("xyzrgb_dragon", tmp_path, monkeypatch, argv=("-R", "1920", "1080", "-S", "8", "8")))
    k = [i for i, l in enumerate(over) if l[0] == "RenderScene"][0]
    assert over[k - 2] == ["SetProperty2", "ren1", "resolution", "1920", "1080"]
    assert over[k - 1] == ["SetProperty2", "ren1", "pixelsamples", "8", "8"]
    assert over[:k - 2] + over[k:] == base


# every scene script whose features are on the device path: it must at least speak the
# command language the product's parser accepts.  Out of scope (volumes, point clouds, SSS, .obj
# scenes): pyro_ball, point_cloud, spline_wisps, surface_wisps, volume_and_bunny, subsurface_scattering, rungholt.
# mis (MaterialShader) and velocity_attribute_blur (PointcloudGenerator) likewise: SURVEY 2 marks those plugins out of scope.
IN_SCOPE = ["bump_mapping", "camera_motion_blur", "dome_light1", "dome_light2", "furry_bunny", "glassy_happy",
            "grid_light", "hair_velocity_blur", "happy_buddhas", "mesh_velocity_blur", "pathtracing",
            "sphere_light", "teapot2", "transform_motion_blur", "xyzrgb_dragon"]


@pytest.mark.parametrize("name", IN_SCOPE)
def test_in_scope_reference_scripts_speak_the_products_command_language(name, tmp_path, monkeypatch, asset_dir):
    """the emitted stream, with only ASSET paths replaced by synthetic stand-ins (meshes -> the tiny
    bumpy sphere / floor / dome of synth.py, textures -> the synthetic .mip) and the frame shrunk, runs
    through the product's parser + Si* API without an error, up to and including RenderScene (deferred:
    no GPU here) -- plugin names, property names / arities / types and the object graph are the
    reference's own."""
    from fujiyama_renderer_amd import synth
    a = synth.ensure_assets(asset_dir, ("tiny",))
    text = run_script(name, tmp_path, monkeypatch, argv=("-R", "48", "32", "-S", "2", "2"))
    out = []
    for line in text.splitlines():
        tok = line.split()
        if len(tok) == 4 and tok[0] == "SetStringProperty" and tok[2] == "filepath":
            stem = os.path.basename(tok[3])
            tok[3] = a["floor"] if "floor" in stem else (a["dome"] if "dome" in stem else a["tiny"])
        if tok and tok[0] == "NewTexture" and not os.path.exists(tok[2]):
            tok[2] = a["sky"]
        if tok and tok[0] == "SaveFrameBuffer":
            tok[2] = str(tmp_path / (name + ".fb"))
        # (hair_velocity_blur loads its head from a Wavefront .obj: an asset loader, SURVEY 2 out of scope -- the
        # stand-in mesh comes through the PLY procedure, whose properties have the same names)
        if tok and tok[0] == "OpenPlugin" and "WavefrontObjProcedure" in tok[2]:
            continue
        if tok and tok[0] == "NewProcedure" and tok[2] == "wavefrontobj_procedure":
            tok[2] = "stanfordply_procedure"
        out.append(" ".join(tok))
    host.run_scene_text("\n".join(out) + "\n", deferred=True)
    sp, rd = host.get_desc()
    assert (rd.xres, rd.yres, rd.rate_x, rd.rate_y) == (48, 32, 2, 2)
    host.close_scene()


# ---------------------------------------------------------------------------------------------------
# Scene zoo: the in-scope scene scripts of the reference beyond the five BASELINE configurations, with the
# reference's OWN parameter choices (cameras, lights, transforms and their time samples, shader settings,
# sampler settings), rendered by the compiled reference and by the product's parser + the CPU restatement.
REF_RENDER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "ref_render")
# scripts whose random streams depend on the worker schedule in the reference (PathtracingShader's per-thread
# generator, the area lights' shared one): pinned through ONE worker thread and the restatement's serial-stream mode
SERIAL = {"pathtracing", "grid_light", "sphere_light"}
ZOO = ["bump_mapping", "camera_motion_blur", "dome_light1", "dome_light2", "glassy_happy", "grid_light", "hair_velocity_blur",
       "mesh_velocity_blur", "pathtracing", "sphere_light", "teapot2", "transform_motion_blur"]


@pytest.mark.skipif(not os.path.exists(REF_RENDER), reason="needs oracle/_ref (the compiled reference)")
@pytest.mark.parametrize("name", ZOO)
def test_scene_zoo_reference_scripts_render_like_the_reference(name, tmp_path, monkeypatch, asset_dir):
    """scenes/<name>.py as it is -> the Py3 emitter (-R 64 48 -S 2 2) -> only asset paths replaced by synthetic
    stand-ins -> (a) the unmodified compiled reference renders the stream (oracle/_ref/ref_render), (b) the
    product's parser reads the same stream and the CPU restatement renders its flat description: the same
    pixels BIT FOR BIT.  (The GPU suite covers the same features against the restatement on the build's own
    workloads: motion kinds, area lights, dome lights, glass, bump maps, hair with velocities.)"""
    import struct
    import subprocess
    import numpy as np
    import oracle_ffi
    from fujiyama_renderer_amd import synth
    a = synth.ensure_assets(asset_dir, ("tiny", "furball"))
    # (stand-ins sized like the assets they replace where the scene depends on it: pathtracing.py builds its box from a
    # floor.ply spanning +-5 scaled by .1; hair_velocity_blur.py grows 1e5 hairs per unit of its head's area)
    floor5 = os.path.join(asset_dir, "floor5.ply")
    if not os.path.exists(floor5):
        v, q = synth.floor_grid(10, 5.0)
        synth.write_ply(floor5 + ".tmp", v, faces_quads=q)
        os.replace(floor5 + ".tmp", floor5)
    # (... and hangs its light -- a squashed sphere.ply, centred on the origin -- half through the ceiling)
    sphere0 = os.path.join(asset_dir, "sphere_centred.ply")
    if not os.path.exists(sphere0):
        v, q, t = synth.bumpy_sphere(12, 9, seed=7)
        v = v.copy()
        v[:, 1] -= 1.0
        synth.write_ply(sphere0 + ".tmp", v, faces_quads=q, faces_tris=t)
        os.replace(sphere0 + ".tmp", sphere0)
    text = run_script(name, tmp_path, monkeypatch, argv=("-R", "64", "48", "-S", "2", "2"))
    fb_path = str(tmp_path / (name + ".fb"))
    out = []
    for line in text.splitlines():
        tok = line.split()
        if len(tok) == 4 and tok[0] == "SetStringProperty" and tok[2] == "filepath":
            stem = os.path.basename(tok[3])
            tok[3] = a["floor"] if "floor" in stem else (a["dome"] if "dome" in stem else a["tiny"])
            if name == "pathtracing" and "floor" in stem:
                tok[3] = floor5
            if name == "pathtracing" and stem.startswith("sphere"):
                tok[3] = sphere0
            if name == "hair_velocity_blur" and stem.endswith(".obj"):
                tok[3] = a["furball"]
        if tok and tok[0] == "NewTexture" and not os.path.exists(tok[2]):
            tok[2] = a["rock"]                       # (.jpg maps: the jpg converter is out of scope; a synthetic .mip stands in)
        if tok and tok[0] == "SaveFrameBuffer":
            tok[2] = fb_path
        if tok and tok[0] == "OpenPlugin" and "WavefrontObjProcedure" in tok[2]:
            continue                                 # (asset loader of hair_velocity_blur's head: the PLY procedure stands in)
        if tok and tok[0] == "NewProcedure" and tok[2] == "wavefrontobj_procedure":
            tok[2] = "stanfordply_procedure"
        if tok and tok[0] == "RenderScene" and name in SERIAL:
            out.append("SetProperty1 %s use_max_thread 0" % tok[1])
            out.append("SetProperty1 %s thread_count 1" % tok[1])
        out.append(" ".join(tok))
    text = "\n".join(out) + "\n"
    scn = str(tmp_path / (name + ".scn"))
    with open(scn, "w") as f:
        f.write(text)
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.dirname(REF_RENDER))
    subprocess.run([REF_RENDER, scn, scn + ".fjfb"], check=True, env=env, stdout=subprocess.DEVNULL, timeout=900)
    with open(scn + ".fjfb", "rb") as f:
        b = f.read()
    w, h, c = struct.unpack("<iii", b[4:16])
    ref = np.frombuffer(b[24:], dtype=np.float32).reshape(h, w, c).copy()
    assert (w, h, c) == (64, 48, 4) and ref[..., :3].max() > 0
    host.run_scene_text(text.replace(fb_path, str(tmp_path / "deferred.fb")), deferred=True)
    sp, rd = host.get_desc()
    osc = oracle_ffi.OracleScene(sp)
    fb, _ = osc.render_serial(rd) if name in SERIAL else osc.render(rd, threads=8)
    osc.close()
    host.close_scene()
    assert np.array_equal(fb, ref), (name, float(np.abs(fb - ref).max()), int((fb != ref).any(axis=2).sum()))
