/* fj_scene_desc.h -- flat, C-ABI description of a Fujiyama scene at the moment
 * SiRenderScene() is called.
 *
 * This is the data half of the drop-in boundary: a host
 * that owns a Fujiyama `Scene` (reference: src/fj_scene.h, filled through the
 * Si* API of src/fj_scene_interface.h:64-127) walks its objects once and fills
 * these structs with plain pointers and sizes; include/fjgpu.h consumes them.
 * Everything is SOURCE-level data (what the user set through the Si* API and
 * what the geometry procedures produced); acceleration structures, matrices
 * and sample tables are derived from it by the consumer.
 *
 * No C++ types, no torch types, no ownership transfer: the caller keeps every
 * pointer alive until fjgpu_scene_create() returns (the GPU core copies).
 *
 * Each struct cites the reference type it flattens.
 */
#ifndef FJ_SCENE_DESC_H
#define FJ_SCENE_DESC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FJ_MAX_XFORM_SAMPLES 8     /* MAX_PROPERTY_SAMPLES, src/fj_property.h:117-137 */
#define FJ_MAX_SHADING_GROUPS 8

/* SiTransformOrder, src/fj_scene_interface.h:32-47 (same numeric values) */
enum {
  FJ_ORDER_SRT = 0, FJ_ORDER_STR, FJ_ORDER_RST, FJ_ORDER_RTS, FJ_ORDER_TRS, FJ_ORDER_TSR,
  FJ_ORDER_XYZ, FJ_ORDER_XZY, FJ_ORDER_YXZ, FJ_ORDER_YZX, FJ_ORDER_ZXY, FJ_ORDER_ZYX
};

/* SiLightType, src/fj_scene_interface.h:49-54 */
enum { FJ_POINT_LIGHT = 0, FJ_GRID_LIGHT, FJ_SPHERE_LIGHT, FJ_DOME_LIGHT };

/* shader plugins recognised by PluginInfo.plugin_name */
enum {
  FJ_SHADER_NONE = 0,        /* NULL shader -> NO_SHADER_COLOR, src/fj_shading.cc:24,553-556 */
  FJ_SHADER_PLASTIC,         /* shaders/plastic_shader/plastic_shader.cc */
  FJ_SHADER_CONSTANT,        /* shaders/constant_shader/constant_shader.cc */
  FJ_SHADER_GLASS,           /* shaders/glass_shader/glass_shader.cc */
  FJ_SHADER_HAIR,            /* shaders/hair_shader/hair_shader.cc */
  FJ_SHADER_PATHTRACING      /* shaders/pathtracing_shader/pathtracing_shader.cc */
};

enum { FJ_PRIMSET_MESH = 0, FJ_PRIMSET_CURVE = 1 };

/* One time sample of a TRS channel: PropertySample, src/fj_property.h:117-128 */
typedef struct fj_xform_sample {
  double v[3];
  double time;
} fj_xform_sample;

/* TransformSampleList, src/fj_transform.h (translate/rotate/scale channels,
 * each 1..8 samples sorted by time; orders as SiTransformOrder values) */
typedef struct fj_xform_desc {
  int32_t transform_order;   /* default FJ_ORDER_SRT */
  int32_t rotate_order;      /* default FJ_ORDER_ZXY */
  int32_t n_translate, n_rotate, n_scale;
  int32_t _pad;
  fj_xform_sample translate[FJ_MAX_XFORM_SAMPLES];
  fj_xform_sample rotate[FJ_MAX_XFORM_SAMPLES];
  fj_xform_sample scale[FJ_MAX_XFORM_SAMPLES];
} fj_xform_desc;

/* Mesh, src/fj_mesh.h:200-216 (AoS vectors flattened; Real = double) */
typedef struct fj_mesh_desc {
  int32_t n_points;
  int32_t n_faces;
  const double  *P;          /* [n_points][3]                       */
  const double  *N;          /* [n_points][3] point normals or NULL */
  const float   *uv;         /* [n_points][2] or NULL               */
  const double  *velocity;   /* [n_points][3] or NULL               */
  const int32_t *indices;    /* [n_faces][3]                        */
  const int32_t *face_group; /* [n_faces] or NULL (-> group 0)      */
  double bounds[6];          /* Mesh::ComputeBounds: min xyz, max xyz */
  const double  *vertex_N;   /* [n_faces][3][3] per-CORNER normals (Mesh::GetVertexNormal(3 f + k), src/fj_mesh.cc:100-106:
                              * a mesh that HasVertexNormal() shades with them instead of N, :108-120) or NULL */
} fj_mesh_desc;

/* Curve, src/fj_curve.h (cubic Bezier ribbons: 4 control points per curve) */
typedef struct fj_curve_desc {
  int32_t n_points;          /* control points                       */
  int32_t n_curves;
  const double  *P;          /* [n_points][3]                        */
  const double  *width;      /* [n_points]                           */
  const float   *Cd;         /* [n_points][3]                        */
  const float   *uv;         /* [n_points][2] or NULL                */
  const double  *velocity;   /* [n_points][3] or NULL                */
  const int32_t *indices;    /* [n_curves] first control point index */
  double bounds[6];
} fj_curve_desc;

/* Texture = one .mip file resident in memory, src/fj_mipmap.cc:124-170:
 * tiles of tilesize x tilesize x nchannels float, tile-major, row-major tiles.
 * width == 0 means "file not open" -> NO_TEXTURE_COLOR (src/fj_texture.cc:15). */
typedef struct fj_texture_desc {
  int32_t width, height, nchannels, tilesize;
  const float *tiles;
} fj_texture_desc;

/* Shader instance parameters captured from SiSetProperty* on a shader ID
 * (after the plugin setter's clamps, e.g. plastic_shader.cc:181-273). */
typedef struct fj_shader_desc {
  int32_t type;              /* FJ_SHADER_* */
  float diffuse[3];
  float specular[3];
  float ambient[3];
  float reflect[3];
  float refract[3];
  float emission[3];
  float filter_color[3];
  float roughness;
  float ior;
  float opacity;
  float bump_amplitude;
  int32_t do_reflect;
  int32_t do_color_filter;
  int32_t diffuse_map;       /* texture index or -1 */
  int32_t bump_map;          /* texture index or -1 */
  int32_t texture;           /* constant shader "texture", or -1 */
} fj_shader_desc;

/* DomeSample, src/fj_importance_sampling.h (host-side preprocess output) */
typedef struct fj_dome_sample {
  double dir[3];
  float color[3];
  float uv[2];
} fj_dome_sample;

/* Light + subclass, src/fj_light.h:31-90 */
typedef struct fj_light_desc {
  int32_t type;              /* FJ_*_LIGHT */
  int32_t sample_count;
  int32_t double_sided;
  int32_t environment_map;   /* texture index or -1 */
  float color[3];
  float intensity;
  fj_xform_desc xform;
  int32_t n_dome_samples;
  int32_t _pad;
  const fj_dome_sample *dome_samples;
} fj_light_desc;

/* ObjectInstance, src/fj_object_instance.h */
typedef struct fj_instance_desc {
  int32_t primset_type;      /* FJ_PRIMSET_* */
  int32_t primset;           /* index into meshes / curves */
  int32_t n_shaders;         /* shader_list_.size() */
  int32_t shaders[FJ_MAX_SHADING_GROUPS];  /* shader index or -1 */
  int32_t reflect_target;    /* group index (never -1 after prepare_render) */
  int32_t refract_target;
  int32_t shadow_target;
  fj_xform_desc xform;
} fj_instance_desc;

/* ObjectGroup / ObjectSet, src/fj_object_group.cc:21-79: ordered instance list */
typedef struct fj_group_desc {
  int32_t n_instances;
  int32_t _pad;
  const int32_t *instances;
} fj_group_desc;

/* Camera, src/fj_camera.h */
typedef struct fj_camera_desc {
  fj_xform_desc xform;
  double fov, znear, zfar;
} fj_camera_desc;

/* Renderer settings, src/fj_renderer.h + property defaults
 * src/internal/fj_property_list_include.cc:451-473 */
typedef struct fj_render_desc {
  int32_t xres, yres;
  int32_t tile_w, tile_h;
  int32_t rate_x, rate_y;        /* pixelsamples */
  float   filter_w, filter_h;    /* gaussian pixel filter widths */
  int32_t region[4];             /* xmin ymin xmax ymax */
  float   jitter;
  int32_t cast_shadow;
  double  time_start, time_end;  /* sample_time_range */
  int32_t max_diffuse_depth, max_reflect_depth, max_refract_depth;
  int32_t sampler_type;          /* 0 fixed grid, 1 adaptive grid (src/fj_renderer.cc:487-499) */
  int32_t adaptive_max_subdivision;       /* Renderer::SetMaxSubdivision, default 1 */
  float   adaptive_subdivision_threshold; /* Renderer::SetSubdivisionThreshold, default .05 */
  int32_t _pad_render;
} fj_render_desc;

typedef struct fj_scene_desc {
  int32_t n_meshes, n_curves, n_textures, n_shaders, n_lights, n_instances, n_groups;
  int32_t target_group;          /* renderer target: implicit all-objects group */
  const fj_mesh_desc     *meshes;
  const fj_curve_desc    *curves;
  const fj_texture_desc  *textures;
  const fj_shader_desc   *shaders;
  const fj_light_desc    *lights;
  const fj_instance_desc *instances;
  const fj_group_desc    *groups;
  fj_camera_desc camera;
} fj_scene_desc;

/* Per-context ray counts: SlTrace calls that pass has_reached_bounce_limit and
 * reach trace_surface (src/fj_shading.cc:154-160,538-539). BASELINE.md 3. */
typedef struct fj_ray_counts {
  uint64_t camera, shadow, diffuse, reflect, refract;
} fj_ray_counts;

#ifdef __cplusplus
}
#endif
#endif /* FJ_SCENE_DESC_H */
