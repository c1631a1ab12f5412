// fj_host.h -- internal scene model of libfjscene.so (host side of the
// drop-in boundary).  Thin bookkeeping only: it records
// what the Si* API was told, runs the geometry procedures, and flattens the
// result into fj_scene_desc for the HIP core.  No rendering code lives here.
#ifndef FJ_HOST_H
#define FJ_HOST_H

#include "fj_scene_interface.h"
#include "fj_plugin_abi.h"
#include "fjgpu.h"

#include <map>
#include <memory>
#include <string>
#include <vector>

namespace fjhost {

// entry types, numbering of reference src/fj_scene_interface.cc:46-64
enum EntryType {
  Type_Begin = 0, Type_ObjectInstance = 1, Type_Accelerator, Type_FrameBuffer, Type_ObjectGroup,
  Type_PointCloud, Type_Turbulence, Type_Procedure, Type_Renderer, Type_Texture, Type_Camera,
  Type_Plugin, Type_Shader, Type_Volume, Type_Curve, Type_Light, Type_Mesh, Type_End
};
static const long TYPE_ID_OFFSET = 10000000;
inline long encode_id(int type, int index) { return TYPE_ID_OFFSET * type + index; }

struct XformSamples {
  fj_xform_desc d;
  XformSamples();
  void push(fj_xform_sample *list, int32_t *count, double x, double y, double z, double time);
};

struct Mesh {
  std::vector<double> P, N, velocity;
  std::vector<double> vertex_N;      // [n_faces][3][3] per-corner normals (OBJ files with `vn`), else empty
  std::vector<float> uv;
  std::vector<int32_t> indices, face_group;
  std::map<std::string, int> face_group_name;
  double bounds[6];
  Mesh();
  int point_count() const { return (int) (P.size() / 3); }
  int face_count() const { return (int) (indices.size() / 3); }
  void ComputeNormals();
  void ComputeBounds();
};

struct Curve {
  std::vector<double> P, width, velocity;
  std::vector<float> Cd, uv;
  std::vector<int32_t> indices;
  double bounds[6];
  Curve();
  void ComputeBounds();
};

struct Texture {
  int width, height, nchannels, tilesize;
  std::vector<float> tiles;
  std::string filename;
  Texture() : width(0), height(0), nchannels(0), tilesize(0) {}
  int LoadFile(const std::string &path);
};

// a plugin DSO opened through the reference's protocol (fj_host_plugin.cc)
struct LoadedPlugin {
  void *dso;
  fj::PluginInfo info;
  std::vector<void *> instances;       // created through info.create_instance, deleted on close
  LoadedPlugin() : dso(nullptr) {}
};
int OpenPluginDso(const char *filename, LoadedPlugin *out);   // 0 or a fj::PlgErrorNo
void ClosePluginDso(LoadedPlugin *p);

enum PluginKind { PLUGIN_SHADER, PLUGIN_PROCEDURE };
struct Plugin {
  std::string name;       // PluginInfo.plugin_name, e.g. "PlasticShader"
  PluginKind kind;
  int shader_type;        // FJ_SHADER_* for shader plugins: the device implementation
  std::unique_ptr<LoadedPlugin> loaded;   // the DSO when one was found; null: built in, known by name
  ~Plugin() { if (loaded) ClosePluginDso(loaded.get()); }
};

struct Shader {
  Plugin *plugin;
  void *instance;         // the DSO's own instance (its setters are called like the reference calls them) or null
  fj_shader_desc d;
  Shader() : plugin(nullptr), instance(nullptr) {}
};

struct Procedure {
  const Plugin *plugin;
  int mesh, curve;        // assigned indices or -1
  std::map<std::string, std::string> strings;
  std::map<std::string, std::vector<double> > numbers;
  Procedure() : plugin(0), mesh(-1), curve(-1) {}
};

struct Light {
  fj_light_desc d;
  XformSamples xf;
  std::vector<fj_dome_sample> dome_samples;
};

struct Instance {
  int primset_type, primset;
  XformSamples xf;
  std::vector<int> shaders;    // shader index or -1; size >= 1
  int reflect_target, refract_target, shadow_target;   // group index or -1
};

struct Group { std::vector<int> instances; };

struct Camera { XformSamples xf; double fov, znear, zfar; };

struct Renderer {
  fj_render_desc d;
  int camera, framebuffer;
  bool use_max_thread;
  int thread_count;
  void *frame_data, *tile_data;
  fj::FrameStartCallback frame_start;
  fj::FrameAbortCallback frame_abort;
  fj::FrameDoneCallback frame_done;
  fj::TileStartCallback tile_start;
  fj::SampleDoneCallback sample_done;
  fj::TileDoneCallback tile_done;
};

struct Scene {
  std::vector<std::unique_ptr<Mesh> > meshes;
  std::vector<std::unique_ptr<Curve> > curves;
  std::vector<std::unique_ptr<Texture> > textures;
  std::vector<std::unique_ptr<Plugin> > plugins;
  std::vector<std::unique_ptr<Shader> > shaders;
  std::vector<std::unique_ptr<Procedure> > procedures;
  std::vector<std::unique_ptr<Light> > lights;
  std::vector<std::unique_ptr<Instance> > instances;
  std::vector<std::unique_ptr<Group> > groups;
  std::vector<std::unique_ptr<Camera> > cameras;
  std::vector<std::unique_ptr<Renderer> > renderers;
  std::vector<std::unique_ptr<fj::FrameBuffer> > framebuffers;
  int n_user_groups;      // groups created through the API (implicit ones follow)

  // flattened description (valid after prepare)
  fj_scene_desc desc;
  fj_render_desc render;
  std::vector<fj_mesh_desc> d_meshes;
  std::vector<fj_curve_desc> d_curves;
  std::vector<fj_texture_desc> d_textures;
  std::vector<fj_shader_desc> d_shaders;
  std::vector<fj_light_desc> d_lights;
  std::vector<fj_instance_desc> d_instances;
  std::vector<fj_group_desc> d_groups;
  bool has_desc;
  Scene() : n_user_groups(-1), has_desc(false) {}
};

Scene *get_scene();

// procedures (fj_host_procedures.cc)
int ReadPlyFile(const std::string &path, Mesh *mesh, std::string *err);
int ReadObjFile(const std::string &path, Mesh *mesh, std::string *err);      // WavefrontObjProcedure (face groups, per-corner normals)
int RunProcedure(Scene *sc, Procedure *proc, std::string *err);
int RunVelocityGenerator(Scene *sc, Procedure *proc, std::string *err);

// dome light importance sampling (fj_host_dome.cc)
int PreprocessDomeLight(Scene *sc, Light *light);

// framebuffer text writer (fj_host_io.cc)
int WriteFrameBuffer(const std::string &filename, const fj::FrameBuffer &fb);

extern bool g_deferred_render;
extern fj_render_stats g_last_stats;
extern std::string g_last_error;

}  // namespace fjhost
#endif
