// fj_host_scene.cc -- the Si* scene API of libfjscene.so.
//
// Own implementation of the reference's flat scene interface
// (reference src/fj_scene_interface.cc): same function names, ID encoding,
// property names / arities / defaults / clamps, implicit-group rules before a
// render.  It keeps only what the HIP core needs to know; SiRenderScene()
// flattens the scene (fj_scene_desc) and hands it to include/fjgpu.h.
#include "fj_host.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>

namespace fjhost {

static Scene *the_scene = nullptr;
static int si_errno = fj::SI_ERR_NONE;
bool g_deferred_render = false;
fj_render_stats g_last_stats;
std::string g_last_error;

Scene *get_scene() { return the_scene; }

struct Entry { int type, index; };
static Entry decode_id(long id)
{
  Entry e{-1, -1};
  if (id < 0) return e;
  const int type = (int) (id / TYPE_ID_OFFSET);
  if (!(type > Type_Begin && type < Type_End)) return e;
  e.type = type;
  e.index = (int) (id - type * TYPE_ID_OFFSET);
  return e;
}

template <class T>
static T *get(std::vector<std::unique_ptr<T> > &v, int i) { return (i >= 0 && i < (int) v.size()) ? v[i].get() : nullptr; }

// ------------------------------------------------------------ transform samples
XformSamples::XformSamples()
{
  // XfmInitTransformSampleList, reference src/fj_transform.cc:238-251
  std::memset(&d, 0, sizeof(d));
  d.transform_order = FJ_ORDER_SRT;
  d.rotate_order = FJ_ORDER_ZXY;
  d.n_translate = d.n_rotate = d.n_scale = 1;
  d.scale[0].v[0] = d.scale[0].v[1] = d.scale[0].v[2] = 1;
}

// PropPushSample, reference src/fj_property.cc:295-315: replace a sample with the
// same time, else append and keep sorted by time; silently full at 8.
void XformSamples::push(fj_xform_sample *list, int32_t *count, double x, double y, double z, double time)
{
  fj_xform_sample s;
  s.v[0] = x; s.v[1] = y; s.v[2] = z; s.time = time;
  for (int i = 0; i < *count; i++)
    if (list[i].time == time) { list[i] = s; return; }
  if (*count >= FJ_MAX_XFORM_SAMPLES) return;
  list[(*count)++] = s;
  std::stable_sort(list, list + *count, [](const fj_xform_sample &a, const fj_xform_sample &b) { return a.time < b.time; });
}

// ------------------------------------------------------------------ geometry
Mesh::Mesh() { face_group_name[""] = 0; for (double &b : bounds) b = 0; }
Curve::Curve() { for (double &b : bounds) b = 0; }

static inline void sub3(const double *a, const double *b, double *o) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static inline void normalize3(double *v)
{
  const double len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (len == 0) return;
  const double inv = 1. / len;
  v[0] *= inv; v[1] *= inv; v[2] *= inv;
}

// Mesh::ComputeNormals, reference src/fj_mesh.cc:195-233: unweighted sum of unit
// face normals per point, then normalise.
void Mesh::ComputeNormals()
{
  if (P.empty() || indices.empty()) return;
  N.assign(P.size(), 0.);
  const int nf = face_count(), np = point_count();
  // The reference adds every face's unit normal to its three point normals in FACE ORDER (one thread), and the last bits of a sum depend
  // on that order.  Here every host thread owns a range of the POINTS and walks all faces for it: a point's normal still receives its faces'
  // normals in face order -- the same additions, the same bits --, a face normal is computed by the (at most three) threads that own one of
  // its points.  (7.2 M triangles: 0.3 s of a 0.6 s scene assembly on one thread.)
  const unsigned hc = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
  const unsigned nt = (unsigned) std::max(1, std::min<int>((int) hc, np / 4096 + 1));
  auto work = [&](unsigned t) {
    const int v0 = (int) ((long long) np * t / nt), v1 = (int) ((long long) np * (t + 1) / nt);
    for (int f = 0; f < nf; f++) {
      const int32_t *ix = &indices[3 * (size_t) f];
      const bool m0 = ix[0] >= v0 && ix[0] < v1, m1 = ix[1] >= v0 && ix[1] < v1, m2 = ix[2] >= v0 && ix[2] < v1;
      if (!(m0 || m1 || m2)) continue;
      double a[3], b[3], ng[3];
      sub3(&P[3 * (size_t) ix[1]], &P[3 * (size_t) ix[0]], a);
      sub3(&P[3 * (size_t) ix[2]], &P[3 * (size_t) ix[0]], b);
      ng[0] = a[1] * b[2] - a[2] * b[1];
      ng[1] = a[2] * b[0] - a[0] * b[2];
      ng[2] = a[0] * b[1] - a[1] * b[0];
      normalize3(ng);
      // the three point normals are read first and then written back as N_k + Ng
      // (a face that repeats an index therefore adds Ng to it once, like the reference)
      if (m0) for (int c = 0; c < 3; c++) N[3 * (size_t) ix[0] + c] += ng[c];
      if (m1 && ix[1] != ix[0]) for (int c = 0; c < 3; c++) N[3 * (size_t) ix[1] + c] += ng[c];
      if (m2 && ix[2] != ix[0] && ix[2] != ix[1]) for (int c = 0; c < 3; c++) N[3 * (size_t) ix[2] + c] += ng[c];
    }
    for (int i = v0; i < v1; i++) normalize3(&N[3 * (size_t) i]);
  };
  if (nt == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(work, t);
    for (auto &t : th) t.join();
  }
}

// Mesh::ComputeBounds, reference src/fj_mesh.cc:235-244 (union of triangle bounds)
void Mesh::ComputeBounds()
{
  const double big = 1.7976931348623157e308;
  const int nf = face_count();
  const unsigned hc = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
  const unsigned nt = (unsigned) std::max(1, std::min<int>((int) hc, nf / 65536 + 1));
  std::vector<double> part((size_t) nt * 6);
  auto work = [&](unsigned t) {
    double mn[3] = {big, big, big}, mx[3] = {-big, -big, -big};
    const int f0 = (int) ((long long) nf * t / nt), f1 = (int) ((long long) nf * (t + 1) / nt);
    for (int f = f0; f < f1; f++)
      for (int k = 0; k < 3; k++) {
        const size_t p = (size_t) indices[3 * (size_t) f + k];
        for (int c = 0; c < 3; c++) {
          double pos[2] = {P[3 * p + c], P[3 * p + c]};
          if (!velocity.empty()) pos[1] = P[3 * p + c] + velocity[3 * p + c];
          for (double q : pos) { if (q < mn[c]) mn[c] = q; if (q > mx[c]) mx[c] = q; }
        }
      }
    for (int c = 0; c < 3; c++) { part[6 * (size_t) t + c] = mn[c]; part[6 * (size_t) t + 3 + c] = mx[c]; }
  };
  if (nt == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(work, t);
    for (auto &t : th) t.join();
  }
  double mn[3] = {big, big, big}, mx[3] = {-big, -big, -big};
  for (unsigned t = 0; t < nt; t++)
    for (int c = 0; c < 3; c++) { mn[c] = std::min(mn[c], part[6 * (size_t) t + c]); mx[c] = std::max(mx[c], part[6 * (size_t) t + 3 + c]); }
  for (int c = 0; c < 3; c++) { bounds[c] = mn[c]; bounds[3 + c] = mx[c]; }
}

// ------------------------------------------------------------------- texture
// MipInput::ReadHeader + whole-file read, reference src/fj_mipmap.cc:124-170
int Texture::LoadFile(const std::string &path)
{
  filename = path;
  FILE *fp = std::fopen(path.c_str(), "rb");
  if (!fp) return -1;
  char magic[4];
  int32_t hdr[5];
  int ok = std::fread(magic, 1, 4, fp) == 4 && std::memcmp(magic, "MIPM", 4) == 0;
  ok = ok && std::fread(hdr, sizeof(int32_t), 5, fp) == 5 && hdr[0] == 1;
  if (!ok || hdr[1] <= 0 || hdr[2] <= 0 || hdr[3] <= 0 || hdr[4] <= 0) { std::fclose(fp); return -1; }
  width = hdr[1]; height = hdr[2]; nchannels = hdr[3]; tilesize = hdr[4];
  const size_t ntiles = (size_t) (width / tilesize) * (height / tilesize);
  tiles.assign(ntiles * tilesize * tilesize * nchannels, 0.f);
  const size_t got = std::fread(tiles.data(), sizeof(float), tiles.size(), fp);
  std::fclose(fp);
  if (got != tiles.size()) { width = height = 0; tiles.clear(); return -1; }
  return 0;
}

// ---------------------------------------------------------- shader defaults
// property defaults of the five device shaders (their MyPropertyList tables)
static void shader_defaults(fj_shader_desc *d, int type)
{
  std::memset(d, 0, sizeof(*d));
  d->type = type;
  d->diffuse_map = d->bump_map = d->texture = -1;
  auto set3 = [](float *a, float x, float y, float z) { a[0] = x; a[1] = y; a[2] = z; };
  set3(d->filter_color, 1, 1, 1);
  d->opacity = 1;
  d->bump_amplitude = 1;
  switch (type) {
  case FJ_SHADER_PLASTIC:      // plastic_shader.cc:50-62
    set3(d->diffuse, .8f, .8f, .8f); set3(d->specular, 1, 1, 1); set3(d->ambient, 1, 1, 1);
    d->roughness = .1f; set3(d->reflect, 1, 1, 1); d->do_reflect = 1; d->ior = 1.4f;
    break;
  case FJ_SHADER_CONSTANT:     // constant_shader.cc:28-32
    set3(d->diffuse, 1, 1, 1);
    break;
  case FJ_SHADER_GLASS:        // glass_shader.cc:38-46
    set3(d->diffuse, 0, 0, 0); set3(d->specular, 1, 1, 1); set3(d->ambient, 1, 1, 1);
    d->roughness = .1f; d->ior = 1.4f;
    break;
  case FJ_SHADER_HAIR:         // hair_shader.cc:39-46
    set3(d->diffuse, 1, 1, 1); set3(d->specular, 1, 1, 1); set3(d->ambient, 1, 1, 1);
    d->roughness = .1f; set3(d->reflect, 1, 1, 1);
    break;
  case FJ_SHADER_PATHTRACING:  // pathtracing_shader.cc:69-84
    set3(d->diffuse, .8f, .8f, .8f); set3(d->ambient, 1, 1, 1);
    d->roughness = .1f; d->ior = 1.4f;
    break;
  }
}

static inline float fmax0(double v) { return (float) (v > 0 ? v : 0); }   // Max(0, x) then -> float

// numeric shader property; returns 0 / -1 (unknown name or wrong arity) following
// PropFind(type, name) semantics (reference src/fj_scene_interface.cc:1237-1248)
static int set_shader_property(Shader *s, const std::string &name, int n, const double *v)
{
  fj_shader_desc &d = s->d;
  auto col = [&](float *dst) { dst[0] = fmax0(v[0]); dst[1] = fmax0(v[1]); dst[2] = fmax0(v[2]); };
  const int t = d.type;
  const bool pl = t == FJ_SHADER_PLASTIC, gl = t == FJ_SHADER_GLASS, ha = t == FJ_SHADER_HAIR, pt = t == FJ_SHADER_PATHTRACING;
  if (n == 3) {
    if (name == "diffuse") { col(d.diffuse); return 0; }
    if (name == "specular" && (pl || gl || ha || pt)) { col(d.specular); return 0; }
    if (name == "ambient" && (pl || gl || ha || pt)) { col(d.ambient); return 0; }
    if (name == "emission" && pt) { col(d.emission); return 0; }
    if (name == "refract" && pt) { col(d.refract); return 0; }
    if (name == "reflect" && (pl || ha || pt)) {
      col(d.reflect);
      if (pl) d.do_reflect = (d.reflect[0] > 0 || d.reflect[1] > 0 || d.reflect[2] > 0);   // plastic_shader.cc:216-233
      return 0;
    }
    if ((name == "filter_color" && gl) || (name == "transmit" && pt)) {   // glass_shader.cc:174-192
      for (int i = 0; i < 3; i++) d.filter_color[i] = (float) (v[i] > .001 ? v[i] : .001);
      d.do_color_filter = !(d.filter_color[0] == 1 && d.filter_color[1] == 1 && d.filter_color[2] == 1);
      return 0;
    }
    return -1;
  }
  if (n == 1) {
    if (name == "roughness" && (pl || gl || ha || pt)) { float r = (float) v[0]; d.roughness = fmax0(r); return 0; }
    if (name == "ior" && (pl || pt)) { float i = (float) v[0]; d.ior = (float) (i > .001 ? (double) i : .001); return 0; }   // Max(.001, ior)
    if (name == "ior" && gl) { float i = (float) v[0]; d.ior = fmax0(i); return 0; }                                         // Max(0, ior)
    if (name == "opacity" && (pl || pt)) { float o = (float) v[0]; d.opacity = o < 0 ? 0 : (o > 1 ? 1 : o); return 0; }
    if (name == "bump_amplitude" && (pl || pt)) { d.bump_amplitude = (float) v[0]; return 0; }
    return -1;
  }
  return -1;
}

static int set_shader_texture(Shader *s, const std::string &name, int tex)
{
  const int t = s->d.type;
  if (name == "texture" && t == FJ_SHADER_CONSTANT) { s->d.texture = tex; return 0; }
  if (name == "diffuse_map" && (t == FJ_SHADER_PLASTIC || t == FJ_SHADER_PATHTRACING)) { s->d.diffuse_map = tex; return 0; }
  if (name == "bump_map" && (t == FJ_SHADER_PLASTIC || t == FJ_SHADER_PATHTRACING)) { s->d.bump_map = tex; return 0; }
  return -1;
}

// -------------------------------------------------------- built-in properties
static void renderer_defaults(Renderer *r)
{
  // Renderer ctor (reference src/fj_renderer.cc:363-424) followed by the property
  // defaults SiNewRenderer applies (src/internal/fj_property_list_include.cc:451-473)
  std::memset(&r->d, 0, sizeof(r->d));
  r->d.xres = 320; r->d.yres = 240;
  r->d.region[0] = 0; r->d.region[1] = 0; r->d.region[2] = 320; r->d.region[3] = 240;
  r->d.tile_w = 32; r->d.tile_h = 32;
  r->d.filter_w = 2; r->d.filter_h = 2;
  r->d.rate_x = 3; r->d.rate_y = 3;
  r->d.jitter = 1;
  r->d.cast_shadow = 1;
  r->d.time_start = 0; r->d.time_end = 1;
  r->d.max_diffuse_depth = 3; r->d.max_reflect_depth = 3; r->d.max_refract_depth = 3;
  r->d.sampler_type = 0;
  r->d.adaptive_max_subdivision = 1;
  r->d.adaptive_subdivision_threshold = .05f;
  r->camera = r->framebuffer = -1;
  r->use_max_thread = true;
  r->thread_count = 8;
  r->frame_data = r->tile_data = nullptr;
  r->frame_start = nullptr; r->frame_abort = nullptr; r->frame_done = nullptr;
  r->tile_start = nullptr; r->sample_done = nullptr; r->tile_done = nullptr;
}

static int set_xform_property(XformSamples *xf, const std::string &name, int n, const double *v, double time, bool allow_scale)
{
  if (n == 1 && name == "transform_order") { const int o = (int) v[0]; if (o < 0 || o >= 6) return -1; xf->d.transform_order = o; return 0; }
  if (n == 1 && name == "rotate_order") { const int o = (int) v[0]; if (o < 6 || o >= 12) return -1; xf->d.rotate_order = o; return 0; }
  if (n == 3 && name == "translate") { xf->push(xf->d.translate, &xf->d.n_translate, v[0], v[1], v[2], time); return 0; }
  if (n == 3 && name == "rotate") { xf->push(xf->d.rotate, &xf->d.n_rotate, v[0], v[1], v[2], time); return 0; }
  if (n == 3 && name == "scale" && allow_scale) { xf->push(xf->d.scale, &xf->d.n_scale, v[0], v[1], v[2], time); return 0; }
  return 1;   // not a transform property
}

static int set_renderer_property(Renderer *r, const std::string &name, int n, const double *v)
{
  fj_render_desc &d = r->d;
  if (n == 1) {
    if (name == "sample_jitter") { d.jitter = (float) v[0]; return 0; }
    if (name == "cast_shadow") { d.cast_shadow = (int) v[0]; return 0; }
    if (name == "max_diffuse_depth") { d.max_diffuse_depth = (int) v[0]; return 0; }
    if (name == "max_reflect_depth") { d.max_reflect_depth = (int) v[0]; return 0; }
    if (name == "max_refract_depth") { d.max_refract_depth = (int) v[0]; return 0; }
    if (name == "raymarch_step" || name == "raymarch_shadow_step" || name == "raymarch_diffuse_step" ||
        name == "raymarch_reflect_step" || name == "raymarch_refract_step") return 0;   // volumes are out of scope
    if (name == "sampler_type") { const int t = (int) v[0]; d.sampler_type = (t == 0 || t == 1) ? t : 0; return 0; }
    // Renderer::SetMaxSubdivision / SetSubdivisionThreshold (src/fj_renderer.cc:508-518; asserts there)
    if (name == "adaptive_max_subdivision") { if ((int) v[0] < 0) return -1; d.adaptive_max_subdivision = (int) v[0]; return 0; }
    if (name == "adaptive_subdivision_threshold") { if (v[0] < 0) return -1; d.adaptive_subdivision_threshold = (float) v[0]; return 0; }
    if (name == "use_max_thread") { r->use_max_thread = ((int) v[0]) != 0; return 0; }
    if (name == "thread_count") { r->thread_count = (int) v[0] < 1 ? 1 : (int) v[0]; return 0; }
  } else if (n == 2) {
    if (name == "sample_time_range") { d.time_start = v[0]; d.time_end = v[1]; return 0; }
    if (name == "resolution") {   // SetResolution also resets the render region (src/fj_renderer.cc:437-446)
      d.xres = (int) v[0]; d.yres = (int) v[1];
      d.region[0] = 0; d.region[1] = 0; d.region[2] = d.xres; d.region[3] = d.yres;
      return 0;
    }
    if (name == "tilesize") { d.tile_w = (int) v[0]; d.tile_h = (int) v[1]; return 0; }
    if (name == "filterwidth") { d.filter_w = (float) v[0]; d.filter_h = (float) v[1]; return 0; }
    if (name == "pixelsamples") { d.rate_x = (int) v[0]; d.rate_y = (int) v[1]; return 0; }
  } else if (n == 4) {
    if (name == "render_region") { for (int i = 0; i < 4; i++) d.region[i] = (int) v[i]; return 0; }
  }
  return -1;
}

static int set_property(long id, const std::string &name, int n, const double *v, double time)
{
  Scene *sc = get_scene();
  if (!sc) return -1;
  const Entry e = decode_id(id);
  switch (e.type) {
  case Type_ObjectInstance: {
    Instance *o = get(sc->instances, e.index);
    if (!o) return -1;
    const int r = set_xform_property(&o->xf, name, n, v, time, true);
    return r == 1 ? -1 : r;
  }
  case Type_Camera: {
    Camera *c = get(sc->cameras, e.index);
    if (!c) return -1;
    const int r = set_xform_property(&c->xf, name, n, v, time, false);
    if (r != 1) return r;
    if (n == 1 && name == "fov") { c->fov = v[0]; return 0; }
    if (n == 1 && name == "znear") { c->znear = v[0]; return 0; }
    if (n == 1 && name == "zfar") { c->zfar = v[0]; return 0; }
    return -1;
  }
  case Type_Light: {
    Light *l = get(sc->lights, e.index);
    if (!l) return -1;
    const int r = set_xform_property(&l->xf, name, n, v, time, true);
    if (r != 1) return r;
    if (n == 1 && name == "intensity") { l->d.intensity = (float) v[0]; return 0; }
    if (n == 3 && name == "color") { for (int i = 0; i < 3; i++) l->d.color[i] = (float) v[i]; return 0; }
    if (n == 1 && name == "sample_count") { const int c = (int) v[0]; l->d.sample_count = c < 1 ? 1 : c; return 0; }
    if (n == 1 && name == "double_sided") { l->d.double_sided = v[0] != 0; return 0; }
    return -1;
  }
  case Type_Renderer: {
    Renderer *r = get(sc->renderers, e.index);
    return r ? set_renderer_property(r, name, n, v) : -1;
  }
  case Type_Shader: {
    Shader *s = get(sc->shaders, e.index);
    if (!s) return -1;
    if (s->plugin && s->plugin->loaded) {
      // PropFind(type, name) in the DSO's table, then its setter (src/fj_scene_interface.cc:1237-1275)
      static const int types[5] = {fj::PROP_NONE, fj::PROP_SCALAR, fj::PROP_VECTOR2, fj::PROP_VECTOR3, fj::PROP_VECTOR4};
      const fj::Property *q = fj::PropFind(s->plugin->loaded->info.property_list, types[n], name.c_str());
      if (!q) return -1;
      fj::PropertyValue pv;
      pv.type = types[n];
      pv.vector = fj::Vector4(v[0], n > 1 ? v[1] : 0, n > 2 ? v[2] : 0, n > 3 ? v[3] : 0);
      if (s->instance && q->SetValue(s->instance, pv)) return -1;
      (void) set_shader_property(s, name, n, v);
      return 0;
    }
    return set_shader_property(s, name, n, v);
  }
  case Type_Procedure: {
    Procedure *p = get(sc->procedures, e.index);
    if (!p) return -1;
    p->numbers[name] = std::vector<double>(v, v + n);
    return 0;
  }
  default:
    return -1;
  }
}

// ------------------------------------------------------------ prepare_render
// reference src/fj_scene_interface.cc:1077-1222.  Accelerator builds happen in the
// GPU core (fjgpu_scene_create); here: implicit groups + flatten.
static void create_implicit_groups(Scene *sc)
{
  if (sc->n_user_groups >= 0) sc->groups.resize(sc->n_user_groups);   // re-render: drop old implicit groups
  sc->n_user_groups = (int) sc->groups.size();
  const int all_objects = (int) sc->groups.size();
  sc->groups.emplace_back(new Group());
  for (int i = 0; i < (int) sc->instances.size(); i++) sc->groups[all_objects]->instances.push_back(i);
  (void) all_objects;
}

static void fill_xform(const XformSamples &xf, fj_xform_desc *out) { *out = xf.d; }

static int flatten(Scene *sc, Renderer *ren)
{
  create_implicit_groups(sc);
  const int all_objects = sc->n_user_groups;

  sc->d_meshes.clear();
  for (auto &m : sc->meshes) {
    fj_mesh_desc d;
    std::memset(&d, 0, sizeof(d));
    d.n_points = m->point_count();
    d.n_faces = m->face_count();
    d.P = m->P.data();
    d.N = m->N.empty() ? nullptr : m->N.data();
    d.uv = m->uv.empty() ? nullptr : m->uv.data();
    d.velocity = m->velocity.empty() ? nullptr : m->velocity.data();
    d.indices = m->indices.data();
    d.face_group = m->face_group.empty() ? nullptr : m->face_group.data();
    d.vertex_N = m->vertex_N.empty() ? nullptr : m->vertex_N.data();
    std::memcpy(d.bounds, m->bounds, sizeof(d.bounds));
    sc->d_meshes.push_back(d);
  }
  sc->d_curves.clear();
  for (auto &c : sc->curves) {
    fj_curve_desc d;
    std::memset(&d, 0, sizeof(d));
    d.n_points = (int) (c->P.size() / 3);
    d.n_curves = (int) c->indices.size();
    d.P = c->P.data();
    d.width = c->width.data();
    d.Cd = c->Cd.empty() ? nullptr : c->Cd.data();
    d.uv = c->uv.empty() ? nullptr : c->uv.data();
    d.velocity = c->velocity.empty() ? nullptr : c->velocity.data();
    d.indices = c->indices.data();
    std::memcpy(d.bounds, c->bounds, sizeof(d.bounds));
    sc->d_curves.push_back(d);
  }
  sc->d_textures.clear();
  for (auto &t : sc->textures) {
    fj_texture_desc d;
    d.width = t->width; d.height = t->height; d.nchannels = t->nchannels; d.tilesize = t->tilesize;
    d.tiles = t->tiles.empty() ? nullptr : t->tiles.data();
    sc->d_textures.push_back(d);
  }
  sc->d_shaders.clear();
  for (auto &s : sc->shaders) sc->d_shaders.push_back(s->d);
  sc->d_lights.clear();
  for (auto &l : sc->lights) {
    if (l->d.type == FJ_DOME_LIGHT && PreprocessDomeLight(sc, l.get())) return -1;   // Light::Preprocess
    fj_light_desc d = l->d;
    fill_xform(l->xf, &d.xform);
    d.n_dome_samples = (int) l->dome_samples.size();
    d.dome_samples = l->dome_samples.empty() ? nullptr : l->dome_samples.data();
    sc->d_lights.push_back(d);
  }
  sc->d_instances.clear();
  for (auto &o : sc->instances) {
    fj_instance_desc d;
    std::memset(&d, 0, sizeof(d));
    d.primset_type = o->primset_type;
    d.primset = o->primset;
    d.n_shaders = (int) o->shaders.size();
    if (d.n_shaders > FJ_MAX_SHADING_GROUPS) { g_last_error = "too many shading groups on one object"; return -1; }
    for (int i = 0; i < FJ_MAX_SHADING_GROUPS; i++) d.shaders[i] = i < d.n_shaders ? o->shaders[i] : -1;
    // null targets default to the all-objects group (create_implicit_groups)
    d.reflect_target = o->reflect_target < 0 ? all_objects : o->reflect_target;
    d.refract_target = o->refract_target < 0 ? all_objects : o->refract_target;
    d.shadow_target = o->shadow_target < 0 ? all_objects : o->shadow_target;
    fill_xform(o->xf, &d.xform);
    sc->d_instances.push_back(d);
  }
  sc->d_groups.clear();
  for (auto &g : sc->groups) {
    fj_group_desc d;
    d.n_instances = (int) g->instances.size();
    d._pad = 0;
    d.instances = g->instances.data();
    sc->d_groups.push_back(d);
  }
  fj_scene_desc &D = sc->desc;
  std::memset(&D, 0, sizeof(D));
  D.n_meshes = (int) sc->d_meshes.size(); D.meshes = sc->d_meshes.data();
  D.n_curves = (int) sc->d_curves.size(); D.curves = sc->d_curves.data();
  D.n_textures = (int) sc->d_textures.size(); D.textures = sc->d_textures.data();
  D.n_shaders = (int) sc->d_shaders.size(); D.shaders = sc->d_shaders.data();
  D.n_lights = (int) sc->d_lights.size(); D.lights = sc->d_lights.data();
  D.n_instances = (int) sc->d_instances.size(); D.instances = sc->d_instances.data();
  D.n_groups = (int) sc->d_groups.size(); D.groups = sc->d_groups.data();
  D.target_group = all_objects;
  Camera *cam = get(sc->cameras, ren->camera);
  if (!cam) { g_last_error = "renderer has no camera"; return -1; }
  fill_xform(cam->xf, &D.camera.xform);
  D.camera.fov = cam->fov; D.camera.znear = cam->znear; D.camera.zfar = cam->zfar;
  sc->render = ren->d;
  sc->has_desc = true;
  return 0;
}

static double now_s()
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static int render_scene(Scene *sc, Renderer *ren)
{
  fj::FrameBuffer *fb = get(sc->framebuffers, ren->framebuffer);
  if (!fb) { g_last_error = "renderer has no framebuffer"; return -1; }
  const double t0 = now_s();
  if (flatten(sc, ren)) return -1;
  fb->Resize(ren->d.xres, ren->d.yres, 4);          // preprocess_framebuffer, src/fj_renderer.cc:805-816
  if (g_deferred_render) return 0;

  // Workers are GPUs: `thread_count` devices, or every visible one with `use_max_thread`
  // (Renderer::GetThreadCount, src/fj_renderer.cc:632-641); FJ_GPU_DEVICES caps it.
  int ndev = fjgpu_device_count();
  if (ndev < 1) { g_last_error = "no HIP device visible: the fjgpu core has no CPU fallback"; return -1; }
  int want = ren->use_max_thread ? ndev : std::min(ndev, std::max(1, ren->thread_count));
  if (const char *e = getenv("FJ_GPU_DEVICES")) want = std::max(1, std::min(want, atoi(e)));
  std::vector<int> devices(want);
  for (int k = 0; k < want; k++) devices[k] = k;
  std::vector<fjgpu_scene *> gs(want, nullptr);
  // (one frame per device scene: the BLAS of the meshes is built on the GPU unless the "device_build" option says otherwise, fjgpu.h)
  fjgpu_global_option("single_frame_build", 1);
  int err = fjgpu_scene_create_multi(&sc->desc, devices.data(), want, gs.data());
  fjgpu_global_option("single_frame_build", 0);
  if (err) { g_last_error = std::string("fjgpu_scene_create: ") + fjgpu_last_error(); return -1; }
  auto destroy_all = [&]() { for (fjgpu_scene *g : gs) fjgpu_scene_destroy(g); };
  // One frame per device scene here (it is destroyed below), like one frame per bin/scene process in the reference
  // (tools/scene_parser/main.cc:34-43): every frame is a COLD frame, and the core's default -- the whole frame as one batch, ~110 GB of
  // wavefront queues at 1080p / 64 spp -- would spend up to seconds waiting for the driver to hand out (and clear) memory that is used for
  // 0.1 s.  Batches of 16 M samples (FJ_BATCH_SAMPLES, 0 = the core's default) cost a 1080p frame ~10 ms in launch tails.
  {
    long bs = 16l << 20;
    if (const char *e = getenv("FJ_BATCH_SAMPLES")) bs = std::max(0l, atol(e));
    for (fjgpu_scene *g : gs) fjgpu_set_option(g, "batch_samples", bs);
  }
  const double t1 = now_s();

  const int ntiles = fjgpu_tile_count(&sc->render);
  fj::FrameInfo finfo;
  finfo.frame_id = 1; finfo.worker_count = want; finfo.tile_count = ntiles;
  finfo.xres = ren->d.xres; finfo.yres = ren->d.yres;
  finfo.frame_region.min.x = ren->d.region[0]; finfo.frame_region.min.y = ren->d.region[1];
  finfo.frame_region.max.x = ren->d.region[2]; finfo.frame_region.max.y = ren->d.region[3];
  finfo.framebuffer = fb;
  // render_frame_start: an interrupt here ends execute_rendering with an error before any tile
  // (src/fj_renderer.cc:774-777)
  if (ren->frame_start && ren->frame_start(ren->frame_data, &finfo) == fj::CALLBACK_INTERRUPT) { destroy_all(); return -1; }

  auto tile_info = [&](int t) {
    fj::TileInfo ti;
    int32_t rect[4];
    fjgpu_tile_rect(&sc->render, t, rect);
    ti.frame_id = 1; ti.worker_id = t % want; ti.region_id = t; ti.total_region_count = ntiles;
    ti.tile_region.min.x = rect[0]; ti.tile_region.min.y = rect[1]; ti.tile_region.max.x = rect[2]; ti.tile_region.max.y = rect[3];
    ti.framebuffer = fb;
    return ti;
  };
  // Tile start hooks, in queue order.  The reference's worker pool stops handing out tiles once
  // a tile_start hook returns CALLBACK_INTERRUPT (render_tile -> LoopStatus::Cancel,
  // src/fj_renderer.cc:1098-1121; src/fj_multi_thread.cc:75-79,127-130): that tile and all
  // later ones are never rendered, the tiles started before it are finished and reported, and
  // execute_rendering still reports the frame as done and returns 0 (src/fj_renderer.cc:787-790;
  // nothing in the reference ever invokes the frame_abort hook).  Same here: the frame that is
  // submitted holds exactly the tiles whose hook said CONTINUE.
  std::vector<int32_t> tile_ids;
  for (int t = 0; t < ntiles; t++) {
    const fj::TileInfo ti = tile_info(t);
    if (ren->tile_start && ren->tile_start(ren->tile_data, &ti) == fj::CALLBACK_INTERRUPT) break;
    tile_ids.push_back(t);
  }

  std::vector<fjgpu_stats> st(want);
  std::memset(st.data(), 0, sizeof(fjgpu_stats) * want);
  // sample_done: the reference reports every sample from its worker threads (CbReportSampleDone in
  // integrate_samples, src/fj_renderer.cc:1061-1096) and an interrupt ends the pool after the tile in
  // flight.  A host call per sample is infeasible here: the hook fires once per BATCH of
  // tiles a device completes (fjgpu_set_batch_callback; from that device's host thread, as the reference
  // fires it from its workers), and an interrupt stops every device after the batch it is rendering.
  struct BatchCtx { Renderer *ren; std::mutex mu; std::vector<int32_t> done; bool cancel = false; } bctx;
  bctx.ren = ren;
  const fjgpu_batch_fn on_batch = [](void *user, int, const int32_t *ids, int n) -> int {
    BatchCtx *c = static_cast<BatchCtx *>(user);
    std::lock_guard<std::mutex> lock(c->mu);
    c->done.insert(c->done.end(), ids, ids + n);
    if (c->ren->sample_done && c->ren->sample_done(c->ren->tile_data) == fj::CALLBACK_INTERRUPT) c->cancel = true;
    return c->cancel ? 1 : 0;
  };
  for (fjgpu_scene *g : gs) fjgpu_set_batch_callback(g, on_batch, &bctx);
  const double t2 = now_s();
  // (a tile_start interrupt on the very first tile leaves nothing to render: the frame stays as resized)
  if (!tile_ids.empty())
    err = fjgpu_render_frame_multi(gs.data(), want, &sc->render, tile_ids.data(), (int) tile_ids.size(), fb->GetWritable(0, 0, 0), st.data());
  const double t3 = now_s();
  if (err) g_last_error = std::string("fjgpu_render_frame_multi: ") + fjgpu_last_error();
  destroy_all();
  if (err) return -1;

  // tile_done once per rendered tile, in queue order, when the frame is in host memory
  // (TileInfo.framebuffer is readable for the finished tile, src/fj_callback.h:39-59)
  // (the core may report a batch twice -- it renders the tiles again after a split-shadow queue overflow, include/fjgpu.h --
  // while the reference reports a tile exactly once)
  std::sort(bctx.done.begin(), bctx.done.end());
  bctx.done.erase(std::unique(bctx.done.begin(), bctx.done.end()), bctx.done.end());
  for (int32_t t : bctx.done) {
    const fj::TileInfo ti = tile_info(t);
    if (ren->tile_done) ren->tile_done(ren->tile_data, &ti);
  }
  if (ren->frame_done) ren->frame_done(ren->frame_data, &finfo);

  g_last_stats.prepare_seconds = t1 - t0;
  g_last_stats.render_seconds = t3 - t2;
  std::memset(&g_last_stats.rays, 0, sizeof(g_last_stats.rays));
  for (const fjgpu_stats &x : st) {
    g_last_stats.rays.camera += x.rays.camera; g_last_stats.rays.shadow += x.rays.shadow; g_last_stats.rays.diffuse += x.rays.diffuse;
    g_last_stats.rays.reflect += x.rays.reflect; g_last_stats.rays.refract += x.rays.refract;
  }
  return 0;
}

}  // namespace fjhost

// =====================================================================
//                               Si* API
// =====================================================================
namespace fj {
using namespace fjhost;

static void set_errno(int e) { si_errno = e; }
static Status status_of(int err) { return err ? SI_FAIL : SI_SUCCESS; }

int SiGetErrorNo(void) { return si_errno; }

// SiOpenPlugin (reference src/fj_scene_interface.cc:193-222 -> Plugin::Open, src/fj_plugin.cc:28-69).
// The DSO is opened through the reference's own protocol when the loader finds it (fj_host_plugin.cc):
// dlopen(name [+ ".so"]), Initialize(PluginInfo *), validation; PluginInfo.plugin_name then selects
// the DEVICE implementation of a shader, and the DSO's Property table gives names, types and
// defaults.  A shader DSO whose plugin_name has no device twin is refused loudly -- there is no
// host shading path.  When no DSO of that name can be loaded, the plugins this build carries
// itself (the five device shaders and the three geometry procedures of the hot path) are found by
// the file's base name, e.g. ".../PlasticShader.so" or "PlasticShader".
static const struct KnownPlugin { const char *name; PluginKind kind; int shader; } kKnownPlugins[] = {
  {"PlasticShader", PLUGIN_SHADER, FJ_SHADER_PLASTIC}, {"ConstantShader", PLUGIN_SHADER, FJ_SHADER_CONSTANT},
  {"GlassShader", PLUGIN_SHADER, FJ_SHADER_GLASS}, {"HairShader", PLUGIN_SHADER, FJ_SHADER_HAIR},
  {"PathtracingShader", PLUGIN_SHADER, FJ_SHADER_PATHTRACING},
  {"StanfordPlyProcedure", PLUGIN_PROCEDURE, 0}, {"CurveGeneratorProcedure", PLUGIN_PROCEDURE, 0},
  {"VelocityGeneratorProcedure", PLUGIN_PROCEDURE, 0}, {"WavefrontObjProcedure", PLUGIN_PROCEDURE, 0},
};

static std::string table_mismatch(const std::string &plugin_name, const Property *theirs);

ID SiOpenPlugin(const char *filename)
{
  Scene *sc = get_scene();
  if (!sc || !filename) return SI_BADID;
  std::unique_ptr<LoadedPlugin> lp(new LoadedPlugin());
  const int perr = OpenPluginDso(filename, lp.get());
  if (perr == 0) {
    const std::string name(lp->info.plugin_name), type(lp->info.plugin_type);
    for (const auto &k : kKnownPlugins)
      if (name == k.name && type == (k.kind == PLUGIN_SHADER ? "Shader" : "Procedure")) {
        const std::string diff = k.kind == PLUGIN_SHADER ? table_mismatch(name, lp->info.property_list) : std::string();
        if (!diff.empty()) {
          ClosePluginDso(lp.get());
          g_last_error = "plugin '" + name + "' (" + filename + ") is not the shader the device code of that name implements: " + diff +
              " -- refused (a shader DSO is rendered by the device twin of its plugin_name, never by its own evaluate())";
          std::fprintf(stderr, "libfjscene: %s\n", g_last_error.c_str());
          set_errno(SI_ERR_FAILLOAD);
          return SI_BADID;
        }
        Plugin *p = new Plugin();
        p->name = k.name; p->kind = k.kind; p->shader_type = k.shader;
        p->loaded = std::move(lp);
        sc->plugins.emplace_back(p);
        set_errno(SI_ERR_NONE);
        return encode_id(Type_Plugin, (int) sc->plugins.size() - 1);
      }
    ClosePluginDso(lp.get());
    g_last_error = "plugin '" + name + "' (" + type + ", " + filename + ") has no device implementation";
    set_errno(SI_ERR_FAILLOAD);
    return SI_BADID;
  }
  if (perr != fj::PLG_ERR_PLUGIN_NOT_FOUND) {
    // a DSO was found but is not a valid plugin: the reference's error numbers
    static const int map[] = {SI_ERR_NONE, SI_ERR_PLUGIN_NOT_FOUND, SI_ERR_INIT_PLUGIN_FUNC_NOT_EXIST, SI_ERR_INIT_PLUGIN_FUNC_FAIL,
        SI_ERR_BAD_PLUGIN_INFO, SI_ERR_CLOSE_PLUGIN_FAIL, SI_ERR_NO_MEMORY};
    g_last_error = std::string("plugin '") + filename + "' is not a valid plugin";
    set_errno(map[perr]);
    return SI_BADID;
  }
  std::string base(filename);
  const size_t slash = base.find_last_of("/\\");
  if (slash != std::string::npos) base = base.substr(slash + 1);
  const size_t dot = base.find_last_of('.');
  if (dot != std::string::npos) base = base.substr(0, dot);
  for (const auto &k : kKnownPlugins)
    if (base == k.name) {
      Plugin *p = new Plugin();
      p->name = k.name; p->kind = k.kind; p->shader_type = k.shader;
      sc->plugins.emplace_back(p);
      set_errno(SI_ERR_NONE);
      return encode_id(Type_Plugin, (int) sc->plugins.size() - 1);
    }
  g_last_error = "plugin '" + base + "' not found, and the build has no device implementation of that name";
  set_errno(SI_ERR_PLUGIN_NOT_FOUND);
  return SI_BADID;
}

Status SiOpenScene(void)
{
  delete the_scene;
  the_scene = new Scene();
  set_errno(SI_ERR_NONE);
  return SI_SUCCESS;
}

Status SiCloseScene(void)
{
  delete the_scene;
  the_scene = nullptr;
  set_errno(SI_ERR_NONE);
  return SI_SUCCESS;
}

Status SiRenderScene(ID renderer)
{
  Scene *sc = get_scene();
  const Entry e = decode_id(renderer);
  if (!sc || e.type != Type_Renderer) return SI_FAIL;
  Renderer *r = get(sc->renderers, e.index);
  if (!r) return SI_FAIL;
  if (render_scene(sc, r)) return SI_FAIL;
  set_errno(SI_ERR_NONE);
  return SI_SUCCESS;
}

Status SiSaveFrameBuffer(ID framebuffer, const char *filename)
{
  Scene *sc = get_scene();
  const Entry e = decode_id(framebuffer);
  if (!sc || e.type != Type_FrameBuffer) return SI_FAIL;
  FrameBuffer *fb = get(sc->framebuffers, e.index);
  if (!fb) return SI_FAIL;
  if (WriteFrameBuffer(filename, *fb)) return SI_FAIL;
  set_errno(SI_ERR_NONE);
  return SI_SUCCESS;
}

Status SiRunProcedure(ID procedure)
{
  Scene *sc = get_scene();
  const Entry e = decode_id(procedure);
  if (!sc || e.type != Type_Procedure) return SI_FAIL;
  Procedure *p = get(sc->procedures, e.index);
  if (!p) return SI_FAIL;
  std::string err;
  if (RunProcedure(sc, p, &err)) { g_last_error = err; return SI_FAIL; }
  set_errno(SI_ERR_NONE);
  return SI_SUCCESS;
}

Status SiAddObjectToGroup(ID group, ID object)
{
  Scene *sc = get_scene();
  const Entry g = decode_id(group), o = decode_id(object);
  if (!sc || g.type != Type_ObjectGroup || o.type != Type_ObjectInstance) return SI_FAIL;
  Group *gp = get(sc->groups, g.index);
  if (!gp || !get(sc->instances, o.index)) return SI_FAIL;
  gp->instances.push_back(o.index);
  set_errno(SI_ERR_NONE);
  return SI_SUCCESS;
}

ID SiNewObjectInstance(ID primset)
{
  Scene *sc = get_scene();
  const Entry e = decode_id(primset);
  if (!sc) return SI_BADID;
  Instance *o = new Instance();
  if (e.type == Type_Mesh && get(sc->meshes, e.index)) { o->primset_type = FJ_PRIMSET_MESH; o->primset = e.index; }
  else if (e.type == Type_Curve && get(sc->curves, e.index)) { o->primset_type = FJ_PRIMSET_CURVE; o->primset = e.index; }
  else { delete o; set_errno(SI_ERR_BADTYPE); return SI_BADID; }   // volumes / point clouds: out of scope
  o->shaders.assign(1, -1);
  o->reflect_target = o->refract_target = o->shadow_target = -1;
  sc->instances.emplace_back(o);
  set_errno(SI_ERR_NONE);
  return encode_id(Type_ObjectInstance, (int) sc->instances.size() - 1);
}

ID SiNewFrameBuffer(const char *)
{
  Scene *sc = get_scene();
  if (!sc) return SI_BADID;
  sc->framebuffers.emplace_back(new FrameBuffer());
  set_errno(SI_ERR_NONE);
  return encode_id(Type_FrameBuffer, (int) sc->framebuffers.size() - 1);
}

ID SiNewObjectGroup(void)
{
  Scene *sc = get_scene();
  if (!sc) return SI_BADID;
  if (sc->n_user_groups >= 0) { sc->groups.resize(sc->n_user_groups); sc->n_user_groups = -1; }
  sc->groups.emplace_back(new Group());
  set_errno(SI_ERR_NONE);
  return encode_id(Type_ObjectGroup, (int) sc->groups.size() - 1);
}

// point clouds, turbulence and volumes are outside the hot path
ID SiNewPointCloud(void) { set_errno(SI_ERR_BADTYPE); g_last_error = "PointCloud is outside the device path"; return SI_BADID; }
ID SiNewTurbulence(void) { set_errno(SI_ERR_BADTYPE); g_last_error = "Turbulence is outside the device path"; return SI_BADID; }
ID SiNewVolume(void) { set_errno(SI_ERR_BADTYPE); g_last_error = "Volume is outside the device path"; return SI_BADID; }

ID SiNewProcedure(ID plugin)
{
  Scene *sc = get_scene();
  const Entry e = decode_id(plugin);
  if (!sc || e.type != Type_Plugin) return SI_BADID;
  Plugin *p = get(sc->plugins, e.index);
  if (!p || p->kind != PLUGIN_PROCEDURE) return SI_BADID;
  Procedure *proc = new Procedure();
  proc->plugin = p;
  sc->procedures.emplace_back(proc);
  set_errno(SI_ERR_NONE);
  return encode_id(Type_Procedure, (int) sc->procedures.size() - 1);
}

ID SiNewRenderer(void)
{
  Scene *sc = get_scene();
  if (!sc) return SI_BADID;
  Renderer *r = new Renderer();
  renderer_defaults(r);
  sc->renderers.emplace_back(r);
  set_errno(SI_ERR_NONE);
  return encode_id(Type_Renderer, (int) sc->renderers.size() - 1);
}

ID SiNewTexture(const char *filename)
{
  Scene *sc = get_scene();
  if (!sc) return SI_BADID;
  fjhost::Texture *t = new fjhost::Texture();
  sc->textures.emplace_back(t);
  if (t->LoadFile(filename ? filename : "")) {
    g_last_error = std::string("cannot load texture ") + (filename ? filename : "");
    set_errno(SI_ERR_FAILLOAD);
    return SI_FAIL;   // the reference returns SI_FAIL (== SI_BADID) here, src/fj_scene_interface.cc:537-540
  }
  set_errno(SI_ERR_NONE);
  return encode_id(Type_Texture, (int) sc->textures.size() - 1);
}

ID SiNewCamera(const char *)
{
  Scene *sc = get_scene();
  if (!sc) return SI_BADID;
  Camera *c = new Camera();
  c->fov = 30; c->znear = .01; c->zfar = 1000;
  sc->cameras.emplace_back(c);
  set_errno(SI_ERR_NONE);
  return encode_id(Type_Camera, (int) sc->cameras.size() - 1);
}

ID SiNewShader(ID plugin)
{
  Scene *sc = get_scene();
  const Entry e = decode_id(plugin);
  if (!sc || e.type != Type_Plugin) return SI_BADID;
  Plugin *p = get(sc->plugins, e.index);
  if (!p || p->kind != PLUGIN_SHADER) return SI_BADID;
  fjhost::Shader *s = new fjhost::Shader();
  s->plugin = p;
  shader_defaults(&s->d, p->shader_type);
  if (p->loaded) {
    // the DSO's instance (its create function applies the property defaults through its own
    // setters, PropSetAllDefaultValues) and the DSO's defaults mirrored into the device parameters
    s->instance = p->loaded->info.create_instance();
    if (s->instance) p->loaded->instances.push_back(s->instance);
    for (const Property *q = p->loaded->info.property_list; q && q->IsValid(); q++) {
      const Vector4 &dv = q->GetDefaultValue();
      const double v[4] = {dv.x, dv.y, dv.z, dv.w};
      const int n = q->GetType() == PROP_SCALAR ? 1 : q->GetType() == PROP_VECTOR2 ? 2 : q->GetType() == PROP_VECTOR3 ? 3 :
          q->GetType() == PROP_VECTOR4 ? 4 : 0;
      if (n) (void) set_shader_property(s, q->GetName(), n, v);     // a property the device twin does not read is ignored
    }
  }
  sc->shaders.emplace_back(s);
  set_errno(SI_ERR_NONE);
  return encode_id(Type_Shader, (int) sc->shaders.size() - 1);
}

ID SiNewCurve(void)
{
  Scene *sc = get_scene();
  if (!sc) return SI_BADID;
  sc->curves.emplace_back(new fjhost::Curve());
  set_errno(SI_ERR_NONE);
  return encode_id(Type_Curve, (int) sc->curves.size() - 1);
}

ID SiNewLight(int light_type)
{
  Scene *sc = get_scene();
  if (!sc || light_type < SI_POINT_LIGHT || light_type > SI_DOME_LIGHT) return SI_BADID;
  fjhost::Light *l = new fjhost::Light();
  std::memset(&l->d, 0, sizeof(l->d));
  l->d.type = light_type;
  l->d.color[0] = l->d.color[1] = l->d.color[2] = 1;
  l->d.intensity = 1;
  l->d.sample_count = 16;
  l->d.double_sided = 0;
  l->d.environment_map = -1;
  sc->lights.emplace_back(l);
  set_errno(SI_ERR_NONE);
  return encode_id(Type_Light, (int) sc->lights.size() - 1);
}

ID SiNewMesh(void)
{
  Scene *sc = get_scene();
  if (!sc) return SI_BADID;
  sc->meshes.emplace_back(new fjhost::Mesh());
  set_errno(SI_ERR_NONE);
  return encode_id(Type_Mesh, (int) sc->meshes.size() - 1);
}

Status SiAssignFrameBuffer(ID renderer, ID framebuffer)
{
  Scene *sc = get_scene();
  const Entry r = decode_id(renderer), f = decode_id(framebuffer);
  if (!sc || r.type != Type_Renderer || f.type != Type_FrameBuffer) return SI_FAIL;
  Renderer *rp = get(sc->renderers, r.index);
  if (!rp || !get(sc->framebuffers, f.index)) return SI_FAIL;
  rp->framebuffer = f.index;
  return SI_SUCCESS;
}

Status SiAssignCamera(ID renderer, ID camera)
{
  Scene *sc = get_scene();
  const Entry r = decode_id(renderer), c = decode_id(camera);
  if (!sc || r.type != Type_Renderer || c.type != Type_Camera) return SI_FAIL;
  Renderer *rp = get(sc->renderers, r.index);
  if (!rp || !get(sc->cameras, c.index)) return SI_FAIL;
  rp->camera = c.index;
  return SI_SUCCESS;
}

Status SiAssignObjectGroup(ID id, const char *name, ID group)
{
  Scene *sc = get_scene();
  const Entry e = decode_id(id), g = decode_id(group);
  if (!sc || g.type != Type_ObjectGroup || !get(sc->groups, g.index)) return SI_FAIL;
  if (e.type != Type_ObjectInstance) return SI_FAIL;
  Instance *o = get(sc->instances, e.index);
  if (!o || !name) return SI_FAIL;
  const std::string n(name);
  if (n == "reflect_target") o->reflect_target = g.index;
  else if (n == "refract_target") o->refract_target = g.index;
  else if (n == "shadow_target") o->shadow_target = g.index;
  else return SI_FAIL;
  return SI_SUCCESS;
}

Status SiAssignPointCloud(ID, const char *, ID) { return SI_FAIL; }
Status SiAssignTurbulence(ID, const char *, ID) { return SI_FAIL; }
Status SiAssignVolume(ID, const char *, ID) { return SI_FAIL; }

Status SiAssignTexture(ID id, const char *name, ID texture)
{
  Scene *sc = get_scene();
  const Entry e = decode_id(id), t = decode_id(texture);
  if (!sc || !name || t.type != Type_Texture || !get(sc->textures, t.index)) return SI_FAIL;
  if (e.type == Type_Shader) {
    fjhost::Shader *s = get(sc->shaders, e.index);
    if (!s) return SI_FAIL;
    if (s->plugin && s->plugin->loaded) {
      const Property *q = PropFind(s->plugin->loaded->info.property_list, PROP_TEXTURE, name);
      if (!q) return SI_FAIL;
      // (the DSO only stores the pointer; it is never dereferenced: evaluate() does not run)
      if (s->instance && q->SetValue(s->instance, PropTexture(reinterpret_cast<fj::Texture *>(get(sc->textures, t.index))))) return SI_FAIL;
      (void) set_shader_texture(s, name, t.index);
      return SI_SUCCESS;
    }
    return status_of(set_shader_texture(s, name, t.index));
  }
  if (e.type == Type_Light) {
    fjhost::Light *l = get(sc->lights, e.index);
    if (!l || std::string(name) != "environment_map") return SI_FAIL;
    l->d.environment_map = t.index;
    return SI_SUCCESS;
  }
  return SI_FAIL;
}

Status SiAssignShader(ID object, const char *shading_group, ID shader)
{
  Scene *sc = get_scene();
  const Entry o = decode_id(object), s = decode_id(shader);
  if (!sc || o.type != Type_ObjectInstance || s.type != Type_Shader) return SI_FAIL;
  Instance *op = get(sc->instances, o.index);
  if (!op || !get(sc->shaders, s.index)) return SI_FAIL;
  int gid = 0;
  if (op->primset_type == FJ_PRIMSET_MESH) {   // Mesh::LookupFaceGroup, unknown -> 0
    const fjhost::Mesh *m = get(sc->meshes, op->primset);
    auto it = m->face_group_name.find(shading_group ? shading_group : "");
    gid = (it != m->face_group_name.end()) ? it->second : 0;
  }
  if ((int) op->shaders.size() <= gid) op->shaders.resize(gid + 1, -1);
  op->shaders[gid] = s.index;
  return SI_SUCCESS;
}

Status SiAssignCurve(ID id, const char *name, ID curve)
{
  Scene *sc = get_scene();
  const Entry e = decode_id(id), c = decode_id(curve);
  if (!sc || !name || c.type != Type_Curve || !get(sc->curves, c.index) || e.type != Type_Procedure) return SI_FAIL;
  Procedure *p = get(sc->procedures, e.index);
  if (!p || std::string(name) != "curve") return SI_FAIL;
  p->curve = c.index;
  return SI_SUCCESS;
}

Status SiAssignMesh(ID id, const char *name, ID mesh)
{
  Scene *sc = get_scene();
  const Entry e = decode_id(id), m = decode_id(mesh);
  if (!sc || !name || m.type != Type_Mesh || !get(sc->meshes, m.index) || e.type != Type_Procedure) return SI_FAIL;
  Procedure *p = get(sc->procedures, e.index);
  if (!p || std::string(name) != "mesh") return SI_FAIL;
  p->mesh = m.index;
  return SI_SUCCESS;
}

Status SiSetProperty1(ID id, const char *name, double v0) { const double v[4] = {v0, 0, 0, 0}; return status_of(set_property(id, name ? name : "", 1, v, 0)); }
Status SiSetProperty2(ID id, const char *name, double v0, double v1) { const double v[4] = {v0, v1, 0, 0}; return status_of(set_property(id, name ? name : "", 2, v, 0)); }
Status SiSetProperty3(ID id, const char *name, double v0, double v1, double v2) { const double v[4] = {v0, v1, v2, 0}; return status_of(set_property(id, name ? name : "", 3, v, 0)); }
Status SiSetProperty4(ID id, const char *name, double v0, double v1, double v2, double v3) { const double v[4] = {v0, v1, v2, v3}; return status_of(set_property(id, name ? name : "", 4, v, 0)); }

Status SiSetSampleProperty3(ID id, const char *name, double v0, double v1, double v2, double time)
{
  const double v[4] = {v0, v1, v2, 0};
  return status_of(set_property(id, name ? name : "", 3, v, time));
}

Status SiSetStringProperty(ID id, const char *name, const char *string)
{
  Scene *sc = get_scene();
  const Entry e = decode_id(id);
  if (!sc || !name || !string || e.type != Type_Procedure) return SI_FAIL;
  Procedure *p = get(sc->procedures, e.index);
  if (!p) return SI_FAIL;
  p->strings[name] = string;
  return SI_SUCCESS;
}

// Property tables as the reference exposes them (SiGetPropertyList, src/fj_scene_interface.cc:1047-1051
// -> get_property_list: a built-in type's table, else the table of the opened plugin of that name).
// The built-in tables carry names, types and defaults (src/internal/fj_property_list_include.cc);
// the setters live behind SiSetProperty* here, so the entries' own setter is null.
namespace {
struct PropRow { const char *name; int n; double v[4]; };
const Property *make_table(const PropRow *rows, std::vector<Property> *keep)
{
  for (const PropRow *r = rows; r->name; r++) {
    PropertyValue pv = r->n == 1 ? PropScalar(r->v[0]) : r->n == 2 ? PropVector2(r->v[0], r->v[1]) :
        r->n == 3 ? PropVector3(r->v[0], r->v[1], r->v[2]) : r->n == 4 ? PropVector4(r->v[0], r->v[1], r->v[2], r->v[3]) :
        PropTexture(nullptr);
    keep->push_back(Property(r->name, pv, nullptr));
  }
  keep->push_back(Property());
  return keep->data();
}
}  // namespace

// the table this build carries for a built-in type or a plugin with a device implementation (null: none of that name); *is_plugin: the latter
static const Property *builtin_table(const std::string &n, bool *is_plugin)
{
  static const PropRow renderer_props[] = {
    {"sample_jitter", 1, {1}}, {"cast_shadow", 1, {1}}, {"max_diffuse_depth", 1, {3}}, {"max_reflect_depth", 1, {3}},
    {"max_refract_depth", 1, {3}}, {"sample_time_range", 2, {0, 1}}, {"resolution", 2, {320, 240}}, {"tilesize", 2, {32, 32}},
    {"filterwidth", 2, {2, 2}}, {"sampler_type", 1, {0}}, {"adaptive_max_subdivision", 1, {1}},
    {"adaptive_subdivision_threshold", 1, {.05}}, {"pixelsamples", 2, {3, 3}}, {"render_region", 4, {0, 0, 320, 240}},
    {"use_max_thread", 1, {1}}, {"thread_count", 1, {8}}, {nullptr, 0, {0}}};
  static const PropRow object_props[] = {
    {"transform_order", 1, {0}}, {"rotate_order", 1, {10}}, {"translate", 3, {0, 0, 0}}, {"rotate", 3, {0, 0, 0}},
    {"scale", 3, {1, 1, 1}}, {nullptr, 0, {0}}};
  static const PropRow camera_props[] = {
    {"transform_order", 1, {0}}, {"rotate_order", 1, {10}}, {"translate", 3, {0, 0, 0}}, {"rotate", 3, {0, 0, 0}},
    {"fov", 1, {30}}, {"znear", 1, {.01}}, {"zfar", 1, {1000}}, {nullptr, 0, {0}}};
  static const PropRow light_props[] = {
    {"transform_order", 1, {0}}, {"rotate_order", 1, {10}}, {"translate", 3, {0, 0, 0}}, {"rotate", 3, {0, 0, 0}},
    {"scale", 3, {1, 1, 1}}, {"intensity", 1, {1}}, {"color", 3, {1, 1, 1}}, {"sample_count", 1, {16}},
    {"double_sided", 1, {0}}, {nullptr, 0, {0}}};
  // the built-in shaders' tables = the Property tables of the reference's plugins (n = 0: a texture)
  static const PropRow plastic_props[] = {      // shaders/plastic_shader/plastic_shader.cc:50-62
    {"diffuse", 3, {.8, .8, .8}}, {"specular", 3, {1, 1, 1}}, {"ambient", 3, {1, 1, 1}}, {"roughness", 1, {.1}},
    {"reflect", 3, {1, 1, 1}}, {"ior", 1, {1.4}}, {"opacity", 1, {1}}, {"diffuse_map", 0, {0}}, {"bump_map", 0, {0}},
    {"bump_amplitude", 1, {1}}, {nullptr, 0, {0}}};
  static const PropRow constant_props[] = {     // shaders/constant_shader/constant_shader.cc:28-32
    {"diffuse", 3, {1, 1, 1}}, {"texture", 0, {0}}, {nullptr, 0, {0}}};
  static const PropRow glass_props[] = {        // shaders/glass_shader/glass_shader.cc:38-46
    {"diffuse", 3, {0, 0, 0}}, {"specular", 3, {1, 1, 1}}, {"ambient", 3, {1, 1, 1}}, {"filter_color", 3, {1, 1, 1}},
    {"roughness", 1, {.1}}, {"ior", 1, {1.4}}, {nullptr, 0, {0}}};
  static const PropRow hair_props[] = {         // shaders/hair_shader/hair_shader.cc:39-46
    {"diffuse", 3, {1, 1, 1}}, {"specular", 3, {1, 1, 1}}, {"ambient", 3, {1, 1, 1}}, {"roughness", 1, {.1}},
    {"reflect", 3, {1, 1, 1}}, {nullptr, 0, {0}}};
  static const PropRow pathtracing_props[] = {  // shaders/pathtracing_shader/pathtracing_shader.cc:69-84
    {"emission", 3, {0, 0, 0}}, {"diffuse", 3, {.8, .8, .8}}, {"specular", 3, {0, 0, 0}}, {"ambient", 3, {1, 1, 1}},
    {"transmit", 3, {1, 1, 1}}, {"roughness", 1, {.1}}, {"reflect", 3, {0, 0, 0}}, {"refract", 3, {0, 0, 0}}, {"ior", 1, {1.4}},
    {"opacity", 1, {1}}, {"diffuse_map", 0, {0}}, {"bump_map", 0, {0}}, {"bump_amplitude", 1, {1}}, {nullptr, 0, {0}}};
  static const struct { const char *type; const PropRow *rows; } builtin[] = {
    {"Renderer", renderer_props}, {"ObjectInstance", object_props}, {"Camera", camera_props}, {"Light", light_props},
    {"PlasticShader", plastic_props}, {"ConstantShader", constant_props}, {"GlassShader", glass_props},
    {"HairShader", hair_props}, {"PathtracingShader", pathtracing_props}};
  static std::vector<Property> tables[sizeof(builtin) / sizeof(builtin[0])];
  for (size_t i = 0; i < sizeof(builtin) / sizeof(builtin[0]); i++)
    if (n == builtin[i].type) {
      if (is_plugin) *is_plugin = i >= 4;
      if (tables[i].empty()) { tables[i].reserve(32); make_table(builtin[i].rows, &tables[i]); }
      return tables[i].data();
    }
  return nullptr;
}

// A shader DSO runs as the DEVICE code of its plugin_name, never as its own evaluate(): one whose property table is not the stock
// shader's (a modified shader that kept the name) would render as the stock one, silently.  "" if the tables agree, else what differs.
static std::string table_mismatch(const std::string &plugin_name, const Property *theirs)
{
  const Property *ours = builtin_table(plugin_name, nullptr);
  if (!ours) return "";                      // (procedures: no table of ours to compare with)
  if (!theirs) return "no property table";
  for (;; ours++, theirs++) {
    if (!ours->IsValid() || !theirs->IsValid()) {
      if (ours->IsValid()) return std::string("property '") + ours->GetName() + "' is missing";
      if (theirs->IsValid()) return std::string("extra property '") + theirs->GetName() + "'";
      return "";
    }
    if (std::strcmp(ours->GetName(), theirs->GetName()) != 0) return std::string("property '") + theirs->GetName() + "' where '" + ours->GetName() + "' is expected";
    if (ours->GetType() != theirs->GetType()) return std::string("property '") + ours->GetName() + "' has another type";
    const Vector4 &a = ours->GetDefaultValue(), &b = theirs->GetDefaultValue();
    if (a.x != b.x || a.y != b.y || a.z != b.z || a.w != b.w) return std::string("property '") + ours->GetName() + "' has another default";
  }
}

const Property *SiGetPropertyList(const char *type_name)
{
  if (!type_name) return nullptr;
  const std::string n(type_name);
  // an opened plugin DSO speaks for itself
  if (Scene *sc = get_scene())
    for (const auto &p : sc->plugins)
      if (p->loaded && n == p->loaded->info.plugin_name) return p->loaded->info.property_list;
  bool is_plugin = false;
  const Property *t = builtin_table(n, &is_plugin);
  if (t && is_plugin) {      // a plugin's table exists once the plugin is opened (reference: get_property_list)
    bool opened = false;
    if (Scene *sc = get_scene()) for (const auto &p : sc->plugins) if (p->name == n) opened = true;
    if (!opened) return nullptr;
  }
  return t;
}

Status SiSetFrameReportCallback(ID id, void *data, FrameStartCallback frame_start, FrameAbortCallback frame_abort, FrameDoneCallback frame_done)
{
  Scene *sc = get_scene();
  const Entry e = decode_id(id);
  if (!sc || e.type != Type_Renderer) return SI_FAIL;
  Renderer *r = get(sc->renderers, e.index);
  if (!r) return SI_FAIL;
  r->frame_data = data; r->frame_start = frame_start; r->frame_abort = frame_abort; r->frame_done = frame_done;
  return SI_SUCCESS;
}

Status SiSetTileReportCallback(ID id, void *data, TileStartCallback tile_start, SampleDoneCallback sample_done, TileDoneCallback tile_done)
{
  Scene *sc = get_scene();
  const Entry e = decode_id(id);
  if (!sc || e.type != Type_Renderer) return SI_FAIL;
  Renderer *r = get(sc->renderers, e.index);
  if (!r) return SI_FAIL;
  r->tile_data = data; r->tile_start = tile_start; r->sample_done = sample_done; r->tile_done = tile_done;
  return SI_SUCCESS;
}

}  // namespace fj
