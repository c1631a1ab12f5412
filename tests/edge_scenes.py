"""Edge-case scenes shared by the CPU suite (oracle vs reference goldens), the GPU
suite (HIP core vs oracle) and tests/golden/make_golden.py."""
from fujiyama_renderer_amd import synth, workloads
from fujiyama_renderer_amd.fujiyama import SceneInterface


def custom_scene(asset_dir, res=(64, 48), spp=(2, 2), lights=3, floor_props=(), obj_shader="plastic_shader",
                  obj_props=(), ren_props=(), textures=(), with_object=True, dome=True, twins=0, obj_file=None):
    a = synth.ensure_assets(asset_dir, ("tiny",))
    si = SceneInterface(parse_args=False)
    si.OpenPlugin("plastic_shader", "PlasticShader")
    si.OpenPlugin("glass_shader", "GlassShader")
    si.OpenPlugin("constant_shader", "ConstantShader")
    si.OpenPlugin("stanfordply_procedure", "StanfordPlyProcedure")
    if obj_file:
        si.OpenPlugin("wavefrontobj_procedure", "WavefrontObjProcedure")
    si.NewCamera("cam1", "PerspectiveCamera")
    si.SetProperty3("cam1", "translate", 0.5, 2.0, 6)
    si.SetProperty3("cam1", "rotate", -12, 4, 0)
    si.SetProperty1("cam1", "fov", 40)
    workloads.point_lights(si, lights)
    tex_ids = {}
    for name, path_key in textures:
        si.NewTexture(name, a[path_key])
        tex_ids[name] = True
    si.NewShader("floor_shader", "plastic_shader")
    for p in floor_props:
        p(si, "floor_shader")
    si.NewShader("obj_shader", obj_shader)
    for p in obj_props:
        p(si, "obj_shader")
    si.NewShader("dome_shader", "constant_shader")
    for mesh, key in (("floor_mesh", "floor"), ("dome_mesh", "dome")) + (() if obj_file else (("obj_mesh", "tiny"),)):
        workloads._ply(si, mesh, a[key])
    if obj_file:
        # the object comes from an OBJ file (WavefrontObjProcedure, the reference's one producer of face groups and per-corner normals)
        si.NewMesh("obj_mesh")
        si.NewProcedure("obj_mesh_proc", "wavefrontobj_procedure")
        si.AssignMesh("obj_mesh_proc", "mesh", "obj_mesh")
        si.SetStringProperty("obj_mesh_proc", "filepath", a[obj_file])
        si.SetStringProperty("obj_mesh_proc", "io_mode", "r")
        si.RunProcedure("obj_mesh_proc")
    si.NewObjectInstance("floor1", "floor_mesh")
    si.AssignShader("floor1", "DEFAULT_SHADING_GROUP", "floor_shader")
    if with_object:
        si.NewObjectInstance("obj1", "obj_mesh")
        si.SetProperty3("obj1", "rotate", 10, 25, -5)
        si.SetProperty3("obj1", "scale", .9, 1.2, .8)
        si.SetProperty3("obj1", "translate", .3, .1, -.4)
        si.SetProperty1("obj1", "transform_order", 4)      # ORDER_TRS
        si.SetProperty1("obj1", "rotate_order", 7)          # ORDER_XZY
        si.AssignShader("obj1", "DEFAULT_SHADING_GROUP", "obj_shader")
        if obj_file == "tiny_groups_obj":
            # groups of the file: "" (0), A (1), B (2), C (3), D (4).  A and C get shaders of their own; B is an UNASSIGNED slot inside the
            # instance's shader list and D lies beyond its end: both shade with slot 0 (ObjectInstance::GetShader, src/fj_object_instance.cc:177-191);
            # a name the mesh does not have lands on slot 0 as well (SiAssignShader, src/fj_scene_interface.cc:731-737) and REPLACES the default
            si.NewShader("group_a_shader", "plastic_shader")
            si.SetProperty3("group_a_shader", "diffuse", .9, .15, .1)
            si.SetProperty3("group_a_shader", "reflect", 0, 0, 0)
            si.NewShader("group_c_shader", "glass_shader")
            si.SetProperty3("group_c_shader", "filter_color", .3, .9, .4)
            si.NewShader("late_default_shader", "plastic_shader")
            si.SetProperty3("late_default_shader", "diffuse", .2, .3, .9)
            si.AssignShader("obj1", "A", "group_a_shader")
            si.AssignShader("obj1", "C", "group_c_shader")
            si.AssignShader("obj1", "no_such_group", "late_default_shader")
        if obj_file == "tiny_normals_obj":
            si.NewShader("upper_shader", "plastic_shader")
            si.SetProperty3("upper_shader", "diffuse", .8, .7, .2)
            si.AssignShader("obj1", "upper", "upper_shader")
    # `twins` more instances of the object's mesh under the SAME transform, each with a colour of its own:
    # every hit on the object is an exact tie in t between instances, so the picture shows which instance
    # the instance level visits first (the reference: depth-first order of its BVH over the instances)
    for k in range(twins):
        sh = "twin_shader%d" % (k + 1)
        si.NewShader(sh, "plastic_shader")
        si.SetProperty3(sh, "diffuse", *[(.9, .1, .1), (.1, .9, .1), (.1, .1, .9), (.9, .9, .1)][k % 4])
        si.SetProperty3(sh, "reflect", 0, 0, 0)
        name = "twin%d" % (k + 1)
        si.NewObjectInstance(name, "obj_mesh")
        si.SetProperty3(name, "rotate", 10, 25, -5)
        si.SetProperty3(name, "scale", .9, 1.2, .8)
        si.SetProperty3(name, "translate", .3, .1, -.4)
        si.SetProperty1(name, "transform_order", 4)
        si.SetProperty1(name, "rotate_order", 7)
        si.AssignShader(name, "DEFAULT_SHADING_GROUP", sh)
    if dome:
        si.NewObjectInstance("dome1", "dome_mesh")
        si.SetProperty3("dome1", "scale", .5, .5, .5)
        si.AssignShader("dome1", "DEFAULT_SHADING_GROUP", "dome_shader")
        if "sky" in tex_ids:
            si.AssignTexture("dome_shader", "texture", "sky")
    workloads._renderer(si, res, spp, ren_props)
    return si.text()


def _set3(name, *v):
    return lambda si, sh: si.SetProperty3(sh, name, *v)


def _set1(name, v):
    return lambda si, sh: si.SetProperty1(sh, name, v)


def _tex(prop, tex):
    return lambda si, sh: si.AssignTexture(sh, prop, tex)


EDGE_CASES = {
    # translucent occluder: plastic opacity < 1 -> closest-hit shadow path, (1 - Os) attenuation
    "translucent_occluder": dict(obj_props=(_set1("opacity", .35), _set3("reflect", 0, 0, 0))),
    "no_shadows": dict(ren_props=(("cast_shadow", (0,)),)),
    "depth_limits": dict(obj_shader="glass_shader", ren_props=(("max_reflect_depth", (1,)), ("max_refract_depth", (2,)))),
    "zero_depth": dict(obj_shader="glass_shader", ren_props=(("max_reflect_depth", (0,)), ("max_refract_depth", (0,)))),
    "glass_color_filter": dict(obj_shader="glass_shader", obj_props=(_set3("filter_color", .2, .5, .1), _set1("ior", 1.7))),
    "no_lights": dict(lights=0),
    "one_light_nonpow2_five": dict(lights=5),
    "many_lights_70": dict(lights=70, spp=(1, 1)),
    "textures_diffuse_and_bump": dict(textures=(("sky", "sky"), ("rock", "rock")),
                                      floor_props=(_tex("diffuse_map", "rock"), _tex("bump_map", "rock"), _set1("bump_amplitude", 2.5))),
    "ragged_frame_and_region": dict(res=(70, 50), ren_props=(("render_region", (3, 5, 66, 47)),)),
    "no_jitter_wide_filter": dict(ren_props=(("sample_jitter", (0,)), ("filterwidth", (3, 3)), ("pixelsamples", (3, 2)))),
    "empty_hit_free_scene": dict(with_object=False, dome=False, lights=1),
    # exact ties in t ACROSS instances: the first instance visited in the reference BVH's depth-first order keeps the hit
    "coincident_instances_2": dict(twins=2, lights=2),
    "coincident_instances_5": dict(twins=5, lights=2, with_object=False),
    # WavefrontObjProcedure meshes: five face groups on one mesh with an unassigned slot, an out-of-range id and an unknown group name;
    # per-corner ("vertex") normals with creases, relative indices and fan-triangulated polygons (arbitrary f64 coordinates)
    "obj_face_groups": dict(obj_file="tiny_groups_obj", lights=2),
    "obj_vertex_normals": dict(obj_file="tiny_normals_obj", lights=2),
}
