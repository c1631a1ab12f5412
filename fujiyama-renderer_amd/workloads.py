"""Workload definitions: the five BASELINE.json configs (+ reduced variants).

Each builder returns scene-description text (the command language of the
reference's `scene` parser) for a scene with the same structure as the
reference's scenes/{teapot,happy_buddhas,xyzrgb_dragon}.scn -- same cameras,
object transforms, shader assignments, shadow groups, 32 point lights of
intensity 1/32 at y = 12 -- but with the seeded synthetic assets of synth.py in
place of the downloadable scans / HDR maps.
"""
import math
import os

import numpy as np

from . import synth
from .fujiyama import SceneInterface


# The 32 point lights of the reference's scenes, (x, z) at y = 12, intensity 1/32 each: the same
# set in scenes/teapot.scn:16-111, happy_buddhas.scn, xyzrgb_dragon.scn:20-115 and
# furry_bunny.scn (scene DATA the workloads keep verbatim).
REFERENCE_LIGHTS_XZ = (
    (0.900771, 4.09137), (2.02315, 5.28021), (10.69, 13.918), (4.28027, 7.58462),
    (12.9548, 1.19914), (6.55808, 2.31772), (0.169064, 10.9623), (1.25002, 4.51314),
    (2.46758, 5.73382), (3.55644, 6.84334), (4.76112, 8.00264), (13.3267, 9.10333),
    (14.4155, 2.68084), (8.10755, 3.79629), (9.21103, 4.9484), (2.83469, 6.09221),
    (4.00945, 7.18302), (12.6072, 0.832089), (6.21169, 1.98055), (7.39599, 10.5563),
    (8.52421, 4.15086), (9.5891, 5.39715), (3.18967, 13.9542), (4.41432, 0.082813),
    (5.48803, 1.21856), (6.57647, 2.31432), (0.265098, 10.9453), (8.84422, 12.1117),
    (10.0154, 5.67625), (11.0907, 14.4043), (4.71726, 7.98851), (13.3907, 9.08986),
)


def point_lights(si, n=32):
    """the first n of the reference's 32 point lights (y = 12), intensity 1/n each"""
    for i in range(n):
        x, z = REFERENCE_LIGHTS_XZ[i % 32]
        if i >= 32:                       # more lights than the reference has: a second, shifted ring
            x, z = x + .37 * (i // 32), z + .53 * (i // 32)
        name = "light%d" % i
        si.NewLight(name, "PointLight")
        si.SetProperty3(name, "translate", x, 12, z)
        si.SetProperty1(name, "intensity", 1.0 / n)


def _ply(si, mesh, path):
    proc = mesh + "_proc"
    si.NewMesh(mesh)
    si.NewProcedure(proc, "stanfordply_procedure")
    si.AssignMesh(proc, "mesh", mesh)
    si.SetStringProperty(proc, "filepath", path)
    si.SetStringProperty(proc, "io_mode", "r")
    si.RunProcedure(proc)


def _renderer(si, res, spp, extra=()):
    si.NewFrameBuffer("fb1", "rgba")
    si.NewRenderer("ren1")
    si.AssignCamera("ren1", "cam1")
    si.AssignFrameBuffer("ren1", "fb1")
    si.SetProperty2("ren1", "resolution", res[0], res[1])
    si.SetProperty2("ren1", "pixelsamples", spp[0], spp[1])
    for name, vals in extra:
        getattr(si, "SetProperty%d" % len(vals))("ren1", name, *vals)
    si.RenderScene("ren1")


def _stage(si, assets, dome_rotate=None, floor_translate=None, sky=True):
    """floor (plastic) + sky dome (constant shader with the sky texture)."""
    si.NewShader("floor_shader", "plastic_shader")
    si.NewShader("dome_shader", "constant_shader")
    _ply(si, "floor_mesh", assets["floor"])
    _ply(si, "dome_mesh", assets["dome"])
    si.NewObjectInstance("floor1", "floor_mesh")
    if floor_translate:
        si.SetProperty3("floor1", "translate", *floor_translate)
    si.AssignShader("floor1", "DEFAULT_SHADING_GROUP", "floor_shader")
    si.NewObjectInstance("dome1", "dome_mesh")
    si.SetProperty3("dome1", "scale", .5, .5, .5)
    if dome_rotate:
        si.SetProperty3("dome1", "rotate", *dome_rotate)
    si.AssignShader("dome1", "DEFAULT_SHADING_GROUP", "dome_shader")
    if sky:
        si.NewTexture("tex1", assets["sky"])
        si.AssignTexture("dome_shader", "texture", "tex1")


def teapot(asset_dir, res=(256, 256), spp=(1, 1), mesh="teapot", nlights=32, extra=()):
    """C1: glass object on a plastic floor under a textured dome (teapot.scn)."""
    a = synth.ensure_assets(asset_dir, (mesh,))
    si = SceneInterface(parse_args=False)
    si.OpenPlugin("plastic_shader", "PlasticShader")
    si.OpenPlugin("glass_shader", "GlassShader")
    si.OpenPlugin("constant_shader", "ConstantShader")
    si.OpenPlugin("stanfordply_procedure", "StanfordPlyProcedure")
    si.NewCamera("cam1", "PerspectiveCamera")
    si.SetProperty3("cam1", "translate", 0, 1, 7)
    si.SetProperty3("cam1", "rotate", -5.710593137499643, 0, 0)
    point_lights(si, nlights)
    si.NewShader("teapot_shader", "glass_shader")
    _ply(si, "teapot_mesh", a[mesh])
    si.NewObjectInstance("teapot1", "teapot_mesh")
    si.AssignShader("teapot1", "DEFAULT_SHADING_GROUP", "teapot_shader")
    _stage(si, a, dome_rotate=(0, 30, 0), floor_translate=(-2, 0, -2))
    si.NewObjectGroup("group1")
    si.AddObjectToGroup("group1", "teapot1")
    si.AssignObjectGroup("teapot1", "shadow_target", "group1")
    si.AssignObjectGroup("floor1", "shadow_target", "group1")
    _renderer(si, res, spp, extra)
    return si.text()


def buddhas(asset_dir, res=(1280, 720), spp=(4, 4), mesh="buddha", nlights=32, extra=()):
    """C2: 4 x 4 instances of one mesh (BVH over instances), happy_buddhas.scn."""
    a = synth.ensure_assets(asset_dir, (mesh,))
    si = SceneInterface(parse_args=False)
    si.OpenPlugin("plastic_shader", "PlasticShader")
    si.OpenPlugin("constant_shader", "ConstantShader")
    si.OpenPlugin("stanfordply_procedure", "StanfordPlyProcedure")
    si.NewCamera("cam1", "PerspectiveCamera")
    si.SetProperty3("cam1", "translate", 5, 4, 5)
    si.SetProperty3("cam1", "rotate", -19.471220634490692, 45, 0)
    point_lights(si, nlights)
    rng = np.random.RandomState(11)
    for i in range(16):
        s = "buddha_shader%d" % i
        si.NewShader(s, "plastic_shader")
        c = rng.uniform(0, .5, size=3)
        si.SetProperty3(s, "diffuse", *[float(np.float32(v)) for v in c])
    _ply(si, "buddha_mesh", a[mesh])
    for i in range(16):
        o = "buddha%d" % i
        si.NewObjectInstance(o, "buddha_mesh")
        si.SetProperty3(o, "translate", -1.5 * (i // 4), 0, -1.5 * (i % 4))
        si.SetProperty3(o, "rotate", 0, 30 * i, 0)
        si.SetProperty3(o, "scale", .6, .6, .6)
        si.AssignShader(o, "DEFAULT_SHADING_GROUP", "buddha_shader%d" % i)
    _stage(si, a, floor_translate=(-2, 0, -2))
    si.NewObjectGroup("group1")
    for i in range(16):
        si.AddObjectToGroup("group1", "buddha%d" % i)
    for i in range(16):
        si.AssignObjectGroup("buddha%d" % i, "shadow_target", "group1")
    si.AssignObjectGroup("floor1", "shadow_target", "group1")
    _renderer(si, res, spp, extra)
    return si.text()


def dragon(asset_dir, res=(1920, 1080), spp=(8, 8), mesh="dragon", nlights=32, extra=()):
    """C3 (headline): one dense mesh, mirror-like plastic (ior 50, diffuse 0)."""
    a = synth.ensure_assets(asset_dir, (mesh,))
    si = SceneInterface(parse_args=False)
    si.OpenPlugin("plastic_shader", "PlasticShader")
    si.OpenPlugin("constant_shader", "ConstantShader")
    si.OpenPlugin("stanfordply_procedure", "StanfordPlyProcedure")
    si.NewCamera("cam1", "PerspectiveCamera")
    si.SetProperty3("cam1", "translate", 0, 1.5, 7)
    si.SetProperty3("cam1", "rotate", -5.710593137499643, 0, 0)
    point_lights(si, nlights)
    si.NewShader("dragon_shader0", "plastic_shader")
    si.SetProperty1("dragon_shader0", "ior", 50)
    si.SetProperty3("dragon_shader0", "diffuse", 0, 0, 0)
    _ply(si, "dragon_mesh", a[mesh])
    si.NewObjectInstance("dragon1", "dragon_mesh")
    si.SetProperty3("dragon1", "scale", .5, .5, .5)
    si.SetProperty3("dragon1", "rotate", 0, -35, 0)
    si.SetProperty3("dragon1", "translate", .2, 0, 0)
    si.AssignShader("dragon1", "DEFAULT_SHADING_GROUP", "dragon_shader0")
    _stage(si, a, dome_rotate=(0, 180, 0))
    si.NewObjectGroup("group1")
    si.AddObjectToGroup("group1", "dragon1")
    si.AssignObjectGroup("dragon1", "shadow_target", "group1")
    si.AssignObjectGroup("floor1", "shadow_target", "group1")
    _renderer(si, res, spp, extra)
    return si.text()


def motion(asset_dir, res=(640, 480), spp=(9, 9), mesh="small", kind="object", nlights=4, extra=()):
    """Motion blur by time-sampled transforms (scenes/transform_motion_blur.py,
    scenes/camera_motion_blur.py): `kind` = "object" (the mesh rotates and moves over the
    shutter), "camera" (the camera dollies and pans), "both", "scale" (an object that also
    grows: three samples per channel, not evenly spaced), "velocity" (per-vertex velocities
    from VelocityGeneratorProcedure, scenes/mesh_velocity_blur.py) or "velocity+object"."""
    a = synth.ensure_assets(asset_dir, (mesh,))
    si = SceneInterface(parse_args=False)
    si.OpenPlugin("plastic_shader", "PlasticShader")
    si.OpenPlugin("constant_shader", "ConstantShader")
    si.OpenPlugin("stanfordply_procedure", "StanfordPlyProcedure")
    si.NewCamera("cam1", "PerspectiveCamera")
    if kind in ("camera", "both"):
        si.SetSampleProperty3("cam1", "translate", 0, 1.5, 7, 0)
        si.SetSampleProperty3("cam1", "translate", .4, 1.6, 6.3, 1)
        si.SetSampleProperty3("cam1", "rotate", -5.710593137499643, 0, 0, 0)
        si.SetSampleProperty3("cam1", "rotate", -7, 4, 2, 1)
    else:
        si.SetSampleProperty3("cam1", "translate", 0, 1.5, 7, 0)
        si.SetProperty3("cam1", "rotate", -5.710593137499643, 0, 0)
    point_lights(si, nlights)
    si.NewShader("dragon_shader0", "plastic_shader")
    si.SetProperty3("dragon_shader0", "diffuse", .7, .05, .1)
    _ply(si, "dragon_mesh", a[mesh])
    if kind in ("velocity", "velocity+object"):
        # per-vertex velocities (scenes/mesh_velocity_blur.py)
        si.OpenPlugin("velocity_generator_procedure", "VelocityGeneratorProcedure")
        si.NewProcedure("velgen_proc", "velocity_generator_procedure")
        si.AssignMesh("velgen_proc", "mesh", "dragon_mesh")
        si.RunProcedure("velgen_proc")
    si.NewObjectInstance("dragon1", "dragon_mesh")
    if kind in ("object", "both", "velocity+object"):
        si.SetSampleProperty3("dragon1", "translate", .2, 0, 0, 0)
        si.SetSampleProperty3("dragon1", "translate", .5, .2, 0, 1)
        si.SetSampleProperty3("dragon1", "rotate", 0, -35, 0, 0)
        si.SetSampleProperty3("dragon1", "rotate", 0, -20, 20, 1)
        si.SetProperty3("dragon1", "scale", .5, .5, .5)
    elif kind == "scale":
        si.SetSampleProperty3("dragon1", "translate", .2, 0, 0, 0)
        si.SetSampleProperty3("dragon1", "translate", .3, .1, .2, .25)
        si.SetSampleProperty3("dragon1", "translate", -.2, .3, 0, 1)
        si.SetSampleProperty3("dragon1", "rotate", 0, -35, 0, 0)
        si.SetSampleProperty3("dragon1", "rotate", 10, 0, 0, 1)
        si.SetSampleProperty3("dragon1", "scale", .5, .5, .5, 0)
        si.SetSampleProperty3("dragon1", "scale", .6, .4, .5, .5)
        si.SetSampleProperty3("dragon1", "scale", .7, .7, .3, 1)
    else:
        si.SetProperty3("dragon1", "scale", .5, .5, .5)
        si.SetProperty3("dragon1", "rotate", 0, -35, 0)
        si.SetProperty3("dragon1", "translate", .2, 0, 0)
    si.AssignShader("dragon1", "DEFAULT_SHADING_GROUP", "dragon_shader0")
    _stage(si, a, dome_rotate=(0, 180, 0))
    si.NewObjectGroup("group1")
    si.AddObjectToGroup("group1", "dragon1")
    si.AssignObjectGroup("dragon1", "shadow_target", "group1")
    si.AssignObjectGroup("floor1", "shadow_target", "group1")
    _renderer(si, res, spp, (("sample_time_range", (0, 1)),) + tuple(extra))
    return si.text()


def arealights(asset_dir, res=(640, 480), spp=(6, 6), mesh="small", kind="grid", extra=()):
    """Area lights (scenes/grid_light.py, scenes/sphere_light.py): `kind` = "grid" (a
    rectangle light above the stage), "sphere", or "both" (+ one point light, so static and
    per-event light samples are interleaved in the loop)."""
    a = synth.ensure_assets(asset_dir, (mesh,))
    si = SceneInterface(parse_args=False)
    si.OpenPlugin("plastic_shader", "PlasticShader")
    si.OpenPlugin("constant_shader", "ConstantShader")
    si.OpenPlugin("stanfordply_procedure", "StanfordPlyProcedure")
    si.NewCamera("cam1", "PerspectiveCamera")
    si.SetSampleProperty3("cam1", "translate", 0, 1.5, 7, 0)
    si.SetProperty3("cam1", "rotate", -5.710593137499643, 0, 0)
    if kind in ("grid", "both"):
        si.NewLight("light1", "GridLight")
        si.SetProperty3("light1", "translate", 0, 6, 0)
        si.SetProperty3("light1", "rotate", 0, 0, 180)
        si.SetProperty3("light1", "scale", 5, 5, 5)
        si.SetProperty1("light1", "sample_count", 8)
    if kind == "both":
        si.NewLight("light2", "PointLight")
        si.SetProperty3("light2", "translate", -6, 8, 6)
        si.SetProperty1("light2", "intensity", .3)
    if kind in ("sphere", "both"):
        si.NewLight("light3", "SphereLight")
        si.SetProperty3("light3", "translate", 2, 2.5, 1.5)
        si.SetProperty3("light3", "scale", .5, .5, .5)
        si.SetProperty1("light3", "intensity", 1.5 if kind == "both" else 3)
        si.SetProperty1("light3", "sample_count", 6)
    si.NewShader("dragon_shader0", "plastic_shader")
    si.SetProperty3("dragon_shader0", "diffuse", .7, .5, .2)
    _ply(si, "dragon_mesh", a[mesh])
    si.NewObjectInstance("dragon1", "dragon_mesh")
    si.SetProperty3("dragon1", "scale", .5, .5, .5)
    si.SetProperty3("dragon1", "rotate", 0, -35, 0)
    si.SetProperty3("dragon1", "translate", .2, 0, 0)
    si.AssignShader("dragon1", "DEFAULT_SHADING_GROUP", "dragon_shader0")
    _stage(si, a, dome_rotate=(0, 180, 0))
    si.NewObjectGroup("group1")
    si.AddObjectToGroup("group1", "dragon1")
    si.AssignObjectGroup("dragon1", "shadow_target", "group1")
    si.AssignObjectGroup("floor1", "shadow_target", "group1")
    _renderer(si, res, spp, extra)
    return si.text()


def crowd(asset_dir, res=(640, 480), spp=(3, 3), mesh="tiny", n=64, nlights=4, extra=()):
    """Many instances of one mesh on a floor (n x-z grid positions, varied rotation and scale):
    exercises the instance level -- the reference's BVH over ObjectInstances -- beyond the
    3-18 instances of the shipped scenes."""
    a = synth.ensure_assets(asset_dir, (mesh,))
    si = SceneInterface(parse_args=False)
    si.OpenPlugin("plastic_shader", "PlasticShader")
    si.OpenPlugin("constant_shader", "ConstantShader")
    si.OpenPlugin("stanfordply_procedure", "StanfordPlyProcedure")
    si.NewCamera("cam1", "PerspectiveCamera")
    si.SetProperty3("cam1", "translate", 0, 6, 12)
    si.SetProperty3("cam1", "rotate", -25, 0, 0)
    point_lights(si, nlights)
    si.NewShader("obj_shader0", "plastic_shader")
    si.SetProperty3("obj_shader0", "diffuse", .7, .4, .2)
    si.NewShader("obj_shader1", "plastic_shader")
    si.SetProperty3("obj_shader1", "diffuse", .2, .5, .7)
    si.SetProperty3("obj_shader1", "reflect", 0, 0, 0)
    _ply(si, "obj_mesh", a[mesh])
    side = int(math.ceil(math.sqrt(n)))
    names = []
    for k in range(n):
        ix, iz = k % side, k // side
        name = "obj%d" % k
        si.NewObjectInstance(name, "obj_mesh")
        sc = .25 + .15 * ((k * 7) % 5) / 4.
        si.SetProperty3(name, "scale", sc, sc * (1.2 if k % 3 else .8), sc)
        si.SetProperty3(name, "rotate", 0, (k * 37) % 360, (k * 11) % 30)
        si.SetProperty3(name, "translate", (ix - .5 * (side - 1)) * 1.3, 0, (iz - .5 * (side - 1)) * 1.3)
        si.AssignShader(name, "DEFAULT_SHADING_GROUP", "obj_shader%d" % (k % 2))
        names.append(name)
    _stage(si, a, dome_rotate=(0, 180, 0))
    si.NewObjectGroup("group1")
    for name in names:
        si.AddObjectToGroup("group1", name)
    for name in names + ["floor1"]:
        si.AssignObjectGroup(name, "shadow_target", "group1")
    _renderer(si, res, spp, extra)
    return si.text()


def furry(asset_dir, res=(1920, 1080), spp=(8, 8), mesh="furbunny", nlights=32, extra=(), hair=False):
    """C5: fur (cubic Bezier curves + HairShader) grown on a mesh by
    CurveGeneratorProcedure; plastic mesh and floor without reflection (furry_bunny.scn)."""
    a = synth.ensure_assets(asset_dir, (mesh,))
    si = SceneInterface(parse_args=False)
    si.OpenPlugin("plastic_shader", "PlasticShader")
    si.OpenPlugin("constant_shader", "ConstantShader")
    si.OpenPlugin("hair_shader", "HairShader")
    si.OpenPlugin("curve_generator_procedure", "CurveGeneratorProcedure")
    si.OpenPlugin("stanfordply_procedure", "StanfordPlyProcedure")
    si.NewCamera("cam1", "PerspectiveCamera")
    si.SetProperty3("cam1", "translate", 1.6, .8, 1.8)
    si.SetProperty3("cam1", "rotate", -8.0494669755283983, 45, 0)
    point_lights(si, nlights)
    si.NewShader("curve_shader", "hair_shader")
    si.NewShader("bunny_shader", "plastic_shader")
    si.SetProperty3("bunny_shader", "diffuse", .8, .5, .3)
    si.SetProperty3("bunny_shader", "reflect", 0, 0, 0)
    si.NewCurve("curve_data")
    _ply(si, "bunny_mesh", a[mesh])
    si.NewProcedure("bunny_hair_gen", "curve_generator_procedure")
    si.AssignMesh("bunny_hair_gen", "mesh", "bunny_mesh")
    si.AssignCurve("bunny_hair_gen", "curve", "curve_data")
    if hair:
        # strands of five chained cubics with per-vertex velocities (scenes/hair_velocity_blur.py)
        si.SetProperty1("bunny_hair_gen", "is_hair", 1)
    si.RunProcedure("bunny_hair_gen")
    si.NewObjectInstance("bunny1", "bunny_mesh")
    si.AssignShader("bunny1", "DEFAULT_SHADING_GROUP", "bunny_shader")
    _stage(si, a, floor_translate=(3, 0, 3))
    si.SetProperty3("floor_shader", "diffuse", .3, .35, .4)
    si.SetProperty3("floor_shader", "reflect", 0, 0, 0)
    si.NewObjectInstance("curve1", "curve_data")
    si.AssignShader("curve1", "DEFAULT_SHADING_GROUP", "curve_shader")
    si.NewObjectGroup("group1")
    si.AddObjectToGroup("group1", "bunny1")
    si.AddObjectToGroup("group1", "curve1")
    for o in ("bunny1", "curve1", "floor1"):
        si.AssignObjectGroup(o, "shadow_target", "group1")
    _renderer(si, res, spp, extra)
    return si.text()


def ibl(asset_dir, res=(1920, 1080), spp=(8, 8), mesh="buddha", sample_count=256, extra=()):
    """Image-based lighting check (scenes/dome_light1.py): one DomeLight with an environment
    map (stratified importance samples, deterministic), a plastic object, a mirror-like and a
    diffuse ball, the dome mirrored by a negative scale, an EMPTY shadow group for the balls
    and per-object reflect groups."""
    a = synth.ensure_assets(asset_dir, (mesh, "tiny"))
    si = SceneInterface(parse_args=False)
    si.OpenPlugin("constant_shader", "ConstantShader")
    si.OpenPlugin("plastic_shader", "PlasticShader")
    si.OpenPlugin("stanfordply_procedure", "StanfordPlyProcedure")
    si.NewCamera("cam1", "PerspectiveCamera")
    si.SetSampleProperty3("cam1", "translate", 0, 1, 8.5, 0)
    rot = 110
    si.NewLight("light1", "DomeLight")
    si.SetProperty3("light1", "rotate", 0, rot, 0)
    si.SetProperty1("light1", "sample_count", sample_count)
    si.NewTexture("tex1", a["sky"])
    si.AssignTexture("light1", "environment_map", "tex1")
    si.NewShader("happy_shader", "plastic_shader")
    si.SetProperty3("happy_shader", "diffuse", .8, .8, .8)
    si.NewShader("dome_shader", "constant_shader")
    si.AssignTexture("dome_shader", "texture", "tex1")
    si.NewShader("sphere_shader1", "plastic_shader")
    si.SetProperty3("sphere_shader1", "diffuse", 0, 0, 0)
    si.SetProperty1("sphere_shader1", "ior", 40)
    si.NewShader("sphere_shader2", "plastic_shader")
    si.SetProperty3("sphere_shader2", "diffuse", .5, .5, .5)
    si.SetProperty3("sphere_shader2", "reflect", 0, 0, 0)
    _ply(si, "happy_mesh", a[mesh])
    _ply(si, "dome_mesh", a["dome"])
    _ply(si, "sphere_mesh", a["tiny"])
    si.NewObjectInstance("happy1", "happy_mesh")
    si.AssignShader("happy1", "DEFAULT_SHADING_GROUP", "happy_shader")
    si.NewObjectInstance("dome1", "dome_mesh")
    si.SetProperty3("dome1", "rotate", 0, rot, 0)
    si.SetProperty3("dome1", "scale", -.5, .5, .5)
    si.AssignShader("dome1", "DEFAULT_SHADING_GROUP", "dome_shader")
    si.NewObjectInstance("sphere1", "sphere_mesh")
    si.AssignShader("sphere1", "DEFAULT_SHADING_GROUP", "sphere_shader1")
    si.SetProperty3("sphere1", "translate", -1.5, -.5, 0)
    si.SetProperty3("sphere1", "scale", .5, .5, .5)
    si.NewObjectInstance("sphere2", "sphere_mesh")
    si.AssignShader("sphere2", "DEFAULT_SHADING_GROUP", "sphere_shader2")
    si.SetProperty3("sphere2", "translate", 1.5, -.5, 0)
    si.SetProperty3("sphere2", "scale", .5, .5, .5)
    si.NewObjectGroup("group1")
    si.AddObjectToGroup("group1", "happy1")
    si.AssignObjectGroup("happy1", "shadow_target", "group1")
    si.NewObjectGroup("group2")
    si.AssignObjectGroup("sphere1", "shadow_target", "group2")
    si.AssignObjectGroup("sphere2", "shadow_target", "group2")
    si.NewObjectGroup("group3")
    si.AddObjectToGroup("group3", "dome1")
    si.AssignObjectGroup("sphere1", "reflect_target", "group3")
    si.NewObjectGroup("group4")
    si.AddObjectToGroup("group4", "dome1")
    si.AddObjectToGroup("group4", "happy1")
    si.AssignObjectGroup("happy1", "reflect_target", "group4")
    _renderer(si, res, spp, extra)
    return si.text()


def cornell(asset_dir, res=(1920, 1080), spp=(16, 16), mesh="bunny", extra=(), objects=("bunny", "sphere", "happy")):
    """C4: closed box lit by an emissive blob, PathtracingShader everywhere
    (scenes/pathtracing.py): diffuse walls, a glass object (reflect + refract + transmit
    filter), a rock-textured bumpy ball, a glossy diffuse+reflect object.  No lights."""
    a = synth.ensure_assets(asset_dir, (mesh, "tiny", "small"))
    si = SceneInterface(parse_args=False)
    si.OpenPlugin("pathtracing_shader", "PathtracingShader")
    si.OpenPlugin("stanfordply_procedure", "StanfordPlyProcedure")
    si.NewCamera("cam1", "PerspectiveCamera")
    si.SetSampleProperty3("cam1", "translate", 0, 0.5, 1.85, 0)
    si.SetProperty1("cam1", "fov", 40)
    si.NewTexture("rock_tex1", a["rock"])
    diff = .8
    for name, col in (("floor_shader1", (diff, diff, diff)), ("ceiling_shader1", (diff, diff, diff)),
                      ("wall_shader1", (diff, 0, 0)), ("wall_shader2", (0, diff, 0)), ("wall_shader3", (diff, diff, diff))):
        si.NewShader(name, "pathtracing_shader")
        si.SetProperty3(name, "diffuse", *col)
    si.NewShader("plastic_shader1", "pathtracing_shader")
    si.SetProperty3("plastic_shader1", "diffuse", .2, .4, .8)
    si.SetProperty3("plastic_shader1", "reflect", 1, 1, 1)
    si.SetProperty3("plastic_shader1", "specular", .1, .1, .1)
    si.NewShader("textured_shader1", "pathtracing_shader")
    si.AssignTexture("textured_shader1", "diffuse_map", "rock_tex1")
    si.AssignTexture("textured_shader1", "bump_map", "rock_tex1")
    si.SetProperty1("textured_shader1", "bump_amplitude", 3)
    si.NewShader("glass_shader1", "pathtracing_shader")
    si.SetProperty3("glass_shader1", "diffuse", 0, 0, 0)
    si.SetProperty3("glass_shader1", "reflect", 1, 1, 1)
    si.SetProperty3("glass_shader1", "refract", 1, 1, 1)
    si.SetProperty3("glass_shader1", "transmit", .2, .1, .0)
    si.NewShader("light_shader1", "pathtracing_shader")
    si.SetProperty3("light_shader1", "diffuse", 0, 0, 0)
    si.SetProperty3("light_shader1", "emission", 1.2 * 20, 1.1 * 20, 0.9 * 20)
    _ply(si, "happy_mesh", a["small"])
    _ply(si, "bunny_mesh", a[mesh])
    _ply(si, "floor_mesh", a["floor"])
    _ply(si, "sphere_mesh", a["dome"])          # has uv (needed by the textured / bumpy ball)
    walls = (("floor1", "floor_shader1", None, None), ("ceiling1", "ceiling_shader1", (0, 0, 180), (0, 1, 0)),
             ("wall1", "wall_shader1", (0, 0, -90), (-.5, .5, 0)), ("wall2", "wall_shader2", (0, 0, 90), (.5, .5, 0)),
             ("wall3", "wall_shader3", (90, 0, 0), (0, .5, -.5)))
    for name, shader, rotate, translate in walls:
        si.NewObjectInstance(name, "floor_mesh")
        si.AssignShader(name, "DEFAULT_SHADING_GROUP", shader)
        si.SetProperty3(name, "scale", .05, .05, .05)      # floor.ply spans +-10 -> +-0.5
        if rotate:
            si.SetProperty3(name, "rotate", *rotate)
        if translate:
            si.SetProperty3(name, "translate", *translate)
    if "bunny" in objects:
        si.NewObjectInstance("bunny1", "bunny_mesh")
        si.AssignShader("bunny1", "DEFAULT_SHADING_GROUP", "glass_shader1")
        si.SetProperty3("bunny1", "translate", -.23, 0, .21)
        si.SetProperty3("bunny1", "scale", .12, .12, .12)
    if "sphere" in objects:
        si.NewObjectInstance("sphere2", "sphere_mesh")
        si.AssignShader("sphere2", "DEFAULT_SHADING_GROUP", "textured_shader1")
        si.SetProperty3("sphere2", "translate", .3, .0, .23)
        si.SetProperty3("sphere2", "scale", .0015, .0015, .0015)   # dome.ply hemisphere r = 100
        si.SetProperty3("sphere2", "rotate", -15, 0, 0)
    if "happy" in objects:
        si.NewObjectInstance("happy1", "happy_mesh")
        si.AssignShader("happy1", "DEFAULT_SHADING_GROUP", "plastic_shader1")
        si.SetProperty3("happy1", "translate", .0, 0, -.1)
        si.SetProperty3("happy1", "scale", .15, .15, .15)
    si.NewObjectInstance("light_source1", "sphere_mesh")
    si.AssignShader("light_source1", "DEFAULT_SHADING_GROUP", "light_shader1")
    si.SetProperty3("light_source1", "translate", 0, 1.02, 0)
    si.SetProperty3("light_source1", "scale", .002, -.0005, .002)  # flattened dome hanging from the ceiling
    _renderer(si, res, spp, extra)
    return si.text()


BUILDERS = {"teapot": teapot, "buddhas": buddhas, "dragon": dragon, "furry": furry, "ibl": ibl, "cornell": cornell,
            "motion": motion, "arealights": arealights, "crowd": crowd}


def default_asset_dir():
    return os.environ.get("FJ_ASSET_DIR", "/tmp/fj_assets")
