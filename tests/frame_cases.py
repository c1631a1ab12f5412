"""The golden frame set: scene builder + arguments of every frame the compiled reference renders
into tests/golden/frames.npz (tests/golden/make_golden.py) and the tests render again."""

# name -> (builder, kwargs): the golden frame set
FRAMES = {
    "c1_teapot_256_1spp": ("teapot", dict(res=(256, 256), spp=(1, 1))),
    "teapot_64_2spp": ("teapot", dict(res=(64, 64), spp=(2, 2))),
    "c2_buddhas_96x54_2spp_bunny": ("buddhas", dict(res=(96, 54), spp=(2, 2), mesh="bunny")),
    "c3_dragon_96x54_3spp_small": ("dragon", dict(res=(96, 54), spp=(3, 3), mesh="small")),
    "c5_furry_64x48_2spp_furball": ("furry", dict(res=(64, 48), spp=(2, 2), mesh="furball", nlights=4)),
    "c6_ibl_dome_light_64x48_2spp": ("ibl", dict(res=(64, 48), spp=(2, 2), mesh="small", sample_count=48)),
    "crowd_40_instances_64x48_2spp": ("crowd", dict(res=(64, 48), spp=(2, 2), mesh="tiny", n=40)),
    "crowd_150_instances_64x48_2spp": ("crowd", dict(res=(64, 48), spp=(2, 2), mesh="tiny", n=150)),
    "c5_hair_vertex_velocity_64x48_2spp": ("furry", dict(res=(64, 48), spp=(2, 2), mesh="furball", nlights=4, hair=True)),
    "motion_object_64x48_3spp": ("motion", dict(res=(64, 48), spp=(3, 3), mesh="tiny", kind="object")),
    "motion_camera_64x48_3spp": ("motion", dict(res=(64, 48), spp=(3, 3), mesh="tiny", kind="camera")),
    "motion_both_64x48_2spp": ("motion", dict(res=(64, 48), spp=(2, 2), mesh="tiny", kind="both")),
    "motion_scale_3samples_64x48_2spp": ("motion", dict(res=(64, 48), spp=(2, 2), mesh="tiny", kind="scale")),
    "motion_vertex_velocity_64x48_3spp": ("motion", dict(res=(64, 48), spp=(3, 3), mesh="tiny", kind="velocity")),
    "motion_velocity_and_object_64x48_2spp": ("motion", dict(res=(64, 48), spp=(2, 2), mesh="tiny", kind="velocity+object")),
    "dragon_region_tilesize16": ("dragon", dict(res=(80, 48), spp=(2, 2), mesh="tiny",
                                  extra=(("tilesize", (16, 16)), ("render_region", (16, 16, 64, 48)),
                                         ("filterwidth", (3, 2.5))))),
    # AdaptiveGridSampler (sampler_type 1): default subdivision, deeper trees, low threshold,
    # ragged tiles + wide filter (margin 2), glass recursion, motion blur, no subdivision at all
    "adaptive_teapot_64_subd1": ("teapot", dict(res=(64, 64), spp=(1, 1), extra=(("sampler_type", (1,)),))),
    "adaptive_teapot_96x80_subd2": ("teapot", dict(res=(96, 80), spp=(1, 1), extra=(
        ("sampler_type", (1,)), ("adaptive_max_subdivision", (2,))))),
    "adaptive_dragon_80x48_subd3_thr02": ("dragon", dict(res=(80, 48), spp=(2, 2), mesh="tiny", extra=(
        ("sampler_type", (1,)), ("adaptive_max_subdivision", (3,)), ("adaptive_subdivision_threshold", (.02,))))),
    "adaptive_dragon_region_tilesize16_subd2": ("dragon", dict(res=(80, 48), spp=(2, 2), mesh="tiny", extra=(
        ("sampler_type", (1,)), ("adaptive_max_subdivision", (2,)), ("tilesize", (16, 16)),
        ("render_region", (16, 16, 64, 48)), ("filterwidth", (3, 2.5))))),
    "adaptive_motion_64x48_subd2": ("motion", dict(res=(64, 48), spp=(3, 3), mesh="tiny", kind="both", extra=(
        ("sampler_type", (1,)), ("adaptive_max_subdivision", (2,))))),
    "adaptive_crowd_64x48_subd0": ("crowd", dict(res=(64, 48), spp=(2, 2), mesh="tiny", n=40, extra=(
        ("sampler_type", (1,)), ("adaptive_max_subdivision", (0,))))),
}
