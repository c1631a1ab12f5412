// oracle/restate/fjo_render.cc -- TEST INFRASTRUCTURE ONLY (CPU oracle).
// Restatement of the reference's renderer loop, sampler, camera, shading
// runtime (SlTrace / SlIlluminance), the shader plugins and the pixel filter.
#include "fjo_render.h"

#include <cfloat>
#include <cmath>
#include <cstring>
#include <mutex>
#include <thread>

namespace fjo {

// =============================================================== tiler (a2)
// src/fj_tiler.cc:56-113
void GenerateTiles(const fj_render_desc &r, std::vector<Tile> *tiles)
{
  const int xmin = r.region[0], ymin = r.region[1], xmax = r.region[2], ymax = r.region[3];
  const int XMIN = (int) std::floor(Max(0, xmin) / (double) r.tile_w);
  const int YMIN = (int) std::floor(Max(0, ymin) / (double) r.tile_h);
  const int XMAX = (int) std::ceil(Min(r.xres, xmax) / (double) r.tile_w);
  const int YMAX = (int) std::ceil(Min(r.yres, ymax) / (double) r.tile_h);
  tiles->clear();
  int id = 0;
  for (int y = YMIN; y < YMAX; y++)
    for (int x = XMIN; x < XMAX; x++) {
      Tile t;
      t.id = id++;
      t.xmin = (int) Max(x * r.tile_w, xmin);
      t.ymin = (int) Max(y * r.tile_h, ymin);
      t.xmax = (int) Min((x + 1) * r.tile_w, xmax);
      t.ymax = (int) Min((y + 1) * r.tile_h, ymax);
      tiles->push_back(t);
    }
}

// ============================================================= sampler (a5)
// src/fj_fixed_grid_sampler.cc:33-84,131-146
void SamplerMargin(const fj_render_desc &r, int margin[2])
{
  const double fw[2] = {(double) r.filter_w, (double) r.filter_h};
  const int rate[2] = {r.rate_x, r.rate_y};
  for (int i = 0; i < 2; i++)
    margin[i] = static_cast<int>(std::ceil(((fw[i] - 1) * rate[i]) * .5));
}

void GenerateSamples(const fj_render_desc &r, const Tile &tile, std::vector<Sample> *samples, int nsamples[2])
{
  int margin[2];
  SamplerMargin(r, margin);
  const int rate[2] = {r.rate_x, r.rate_y};
  const int res[2] = {r.xres, r.yres};
  nsamples[0] = rate[0] * (tile.xmax - tile.xmin) + 2 * margin[0];
  nsamples[1] = rate[1] * (tile.ymax - tile.ymin) + 2 * margin[1];
  samples->resize(static_cast<size_t>(nsamples[0]) * nsamples[1]);

  XorShift rng, rng_time;
  const double jitter = r.jitter;            // Real jitter_ <- float renderer jitter_
  const bool jittered = jitter > 0;          // src/fj_sampler.cc:68-74
  const double udelta = 1. / (rate[0] * res[0]);
  const double vdelta = 1. / (rate[1] * res[1]);
  const int xoffset = tile.xmin * rate[0] - margin[0];
  const int yoffset = tile.ymin * rate[1] - margin[1];

  Sample *s = samples->data();
  for (int y = 0; y < nsamples[1]; y++)
    for (int x = 0; x < nsamples[0]; x++) {
      s->uv[0] = (.5 + x + xoffset) * udelta;
      s->uv[1] = 1 - (.5 + y + yoffset) * vdelta;
      if (jittered) {
        const double u_jitter = rng.NextFloat01() * jitter;
        const double v_jitter = rng.NextFloat01() * jitter;
        s->uv[0] += udelta * (u_jitter - .5);
        s->uv[1] += vdelta * (v_jitter - .5);
      }
      // SetSampleTimeRange is always called by init_worker (src/fj_renderer.cc:895-896)
      const double rnd = rng_time.NextFloat01();
      s->time = Fit(rnd, 0, 1, r.time_start, r.time_end);
      s->data[0] = s->data[1] = s->data[2] = s->data[3] = 0;
      s++;
    }
}

// =================================================== adaptive grid sampler (a5')
// src/fj_adaptive_grid_sampler.cc.  Samples sit on the corners of a lattice of
// 2^max_subdivision cells per pixel (so neighbouring cells share samples); the rectangles
// of the lattice are taken from a LIFO stack, their four corners are traced if nothing has
// written them yet, and the rectangle is either split in four (corner values differ by more
// than the threshold in some channel) or filled by bilinear interpolation, overwriting
// whatever its border samples held.
struct AdaptiveGrid {
  struct Rect { int x0, y0, x1, y1; };
  std::vector<Sample> samples;
  std::vector<signed char> state;      // subd_flag_: < 0 until traced or interpolated
  std::vector<Rect> stack;
  int nx = 0, ny = 0, div = 1, corner = 0;
  int margin[2] = {0, 0};
  double threshold = 0;

  // count_samples_in_margin / compute_num_pixel_division / count_samples_in_region, :195-222
  static void Counts(const fj_render_desc &r, int *div, int margin[2])
  {
    *div = static_cast<int>(std::pow(2, r.adaptive_max_subdivision));
    margin[0] = static_cast<int>(std::ceil((double) r.filter_w - 1));
    margin[1] = static_cast<int>(std::ceil((double) r.filter_h - 1));
  }

  void Generate(const fj_render_desc &r, const Tile &tile)         // generate_samples, :35-110
  {
    Counts(r, &div, margin);
    threshold = (double) r.adaptive_subdivision_threshold;          // float member of the Renderer -> Real
    const int tw = tile.xmax - tile.xmin + 2 * margin[0], th = tile.ymax - tile.ymin + 2 * margin[1];
    nx = div * tw + 1;
    ny = div * th + 1;
    samples.assign(static_cast<size_t>(nx) * ny, Sample());
    state.assign(samples.size(), -1);
    XorShift rng, rng_time;
    const double jitter = r.jitter;
    const double udelta = 1. / (div * r.xres), vdelta = 1. / (div * r.yres);
    const int xoffset = (tile.xmin - margin[0]) * div, yoffset = (tile.ymin - margin[1]) * div;
    Sample *s = samples.data();
    for (int y = 0; y < ny; y++)
      for (int x = 0; x < nx; x++, s++) {
        s->uv[0] = (x + xoffset) * udelta;                         // lattice corners: no half-cell offset
        s->uv[1] = 1 - (y + yoffset) * vdelta;
        if (jitter > 0) {
          const double uj = rng.NextFloat01() * jitter;
          const double vj = rng.NextFloat01() * jitter;
          s->uv[0] += udelta * (uj - .5);
          s->uv[1] += vdelta * (vj - .5);
        }
        s->time = Fit(rng_time.NextFloat01(), 0, 1, r.time_start, r.time_end);
        s->data[0] = s->data[1] = s->data[2] = s->data[3] = 0;
      }
    stack.clear();
    corner = 0;
    for (int y = 0; y < th; y++)
      for (int x = 0; x < tw; x++) stack.push_back(Rect{x * div, y * div, (x + 1) * div, (y + 1) * div});
  }

  size_t At(int x, int y) const { return static_cast<size_t>(y) * nx + x; }

  // get_next_sample, :124-170: index of the next sample to trace, or -1 when the tile is done
  long Next()
  {
    while (!stack.empty()) {
      const Rect q = stack.back();
      if (corner == 4) {
        corner = 0;
        stack.pop_back();
        if (NeedsSplit(q)) Split(q); else Fill(q);
        continue;
      }
      const int c = corner++;
      const size_t k = At((c & 1) ? q.x1 : q.x0, (c & 2) ? q.y1 : q.y0);   // corner order: (x0,y0) (x1,y0) (x0,y1) (x1,y1)
      if (state[k] < 0) { state[k] = 1; return (long) k; }
    }
    return -1;
  }

  bool NeedsSplit(const Rect &q) const                            // compare_corners, :224-262
  {
    if (q.x1 - q.x0 < 2 || q.y1 - q.y0 < 2) return false;
    const double *d[4] = {samples[At(q.x0, q.y0)].data, samples[At(q.x1, q.y0)].data,
                          samples[At(q.x0, q.y1)].data, samples[At(q.x1, q.y1)].data};
    for (int c = 0; c < 4; c++) {
      double lo = d[0][c], hi = d[0][c];
      for (int i = 1; i < 4; i++) { lo = Min(lo, d[i][c]); hi = Max(hi, d[i][c]); }
      if (hi - lo > threshold) return true;
    }
    return false;
  }

  void Split(const Rect &q)                                       // subdivide_rect, :264-302
  {
    const int xm = (q.x0 + q.x1) / 2, ym = (q.y0 + q.y1) / 2;
    stack.push_back(Rect{q.x0, q.y0, xm, ym});
    stack.push_back(Rect{xm, q.y0, q.x1, ym});
    stack.push_back(Rect{q.x0, ym, xm, q.y1});
    stack.push_back(Rect{xm, ym, q.x1, q.y1});
  }

  void Fill(const Rect &q)                                        // interpolate_rect, :304-330
  {
    double c00[4], c10[4], c01[4], c11[4];
    for (int c = 0; c < 4; c++) {
      c00[c] = samples[At(q.x0, q.y0)].data[c]; c10[c] = samples[At(q.x1, q.y0)].data[c];
      c01[c] = samples[At(q.x0, q.y1)].data[c]; c11[c] = samples[At(q.x1, q.y1)].data[c];
    }
    for (int y = q.y0; y <= q.y1; y++) {
      const double ty = 1. * (y - q.y0) / (q.y1 - q.y0);
      for (int x = q.x0; x <= q.x1; x++) {
        const double tx = 1. * (x - q.x0) / (q.x1 - q.x0);
        const size_t k = At(x, y);
        for (int c = 0; c < 4; c++) {
          const double left = (1 - ty) * c00[c] + ty * c01[c];     // Lerp(Vector4), src/fj_vector.h:515-518
          const double right = (1 - ty) * c10[c] + ty * c11[c];
          samples[k].data[c] = (1 - tx) * left + tx * right;
        }
        if (state[k] < 0) state[k] = 0;
      }
    }
  }
};

// ============================================================== camera (a7)
// src/fj_camera.cc:79-110
void CameraGetRay(const CameraState &cam, const double uv[2], double time, Ray *ray)
{
  Xfm lerped;
  const Xfm *x = &cam.xfm_static;
  if (!cam.is_static) { LerpXfm(cam.d->xform, time, &lerped); x = &lerped; }
  const V3 target((uv[0] - .5) * cam.uv_size[0], (uv[1] - .5) * cam.uv_size[1], -1);
  const V3 tw = MatTransformPoint(x->matrix, target);
  const V3 eye = MatTransformPoint(x->matrix, V3());
  ray->dir = Normalize(tw - eye);
  ray->orig = eye;
  ray->tmin = cam.d->znear;
  ray->tmax = cam.d->zfar;
}

void CameraInit(const fj_camera_desc *d, int xres, int yres, CameraState *cam)
{
  cam->d = d;
  const double aspect = xres / (double) yres;            // src/fj_renderer.cc:799
  cam->uv_size[1] = 2 * std::tan(Radian(d->fov / 2.));   // src/fj_camera.cc:98-102
  cam->uv_size[0] = cam->uv_size[1] * aspect;
  cam->is_static = (d->xform.n_translate == 1 && d->xform.n_rotate == 1 && d->xform.n_scale == 1);
  LerpXfm(d->xform, 0, &cam->xfm_static);
}

// ============================================================== filter (a35)
double GaussianFilter(double xwidth, double ywidth, double x, double y)   // src/fj_filter.cc:49-58
{
  const double xx = 2 * x / xwidth;
  const double yy = 2 * y / ywidth;
  return std::exp(-2 * (xx * xx + yy * yy));
}

// ============================================================= texture (a33)
static const Col4 NO_TEXTURE_COLOR(1, .63f, .63f, 1);   // src/fj_texture.cc:15

Col4 TextureLookup(const fj_texture_desc &tex, float u, float v)          // src/fj_texture.cc:51-78
{
  if (tex.width == 0 || tex.tiles == nullptr) return NO_TEXTURE_COLOR;
  const int ts = tex.tilesize;
  const int xntiles = tex.width / ts, yntiles = tex.height / ts;
  const float tu = u - std::floor(u);
  const float tv = v - std::floor(v);
  const float su = tu * xntiles;
  const float sv = (1 - tv) * yntiles;
  const int xtile = static_cast<int>(std::floor(su));
  const int ytile = static_cast<int>(std::floor(sv));
  // MipInput::ReadTile clamps the tile index, src/fj_mipmap.cc:153-170
  const int x = static_cast<int>(Clamp(xtile, 0, xntiles - 1));
  const int y = static_cast<int>(Clamp(ytile, 0, yntiles - 1));
  const float *tile = tex.tiles + static_cast<size_t>(y * xntiles + x) * ts * ts * tex.nchannels;
  const int xpxl = (int) ((su - std::floor(su)) * 64);
  const int ypxl = (int) ((sv - std::floor(sv)) * 64);
  // FrameBuffer::GetColor on the ts x ts tile buffer: out of range -> Color4()
  if (xpxl < 0 || xpxl >= ts || ypxl < 0 || ypxl >= ts) return Col4();
  const float *p = tile + (ypxl * ts + xpxl) * tex.nchannels;
  switch (tex.nchannels) {
  case 1: return Col4(p[0], p[0], p[0], 1);
  case 3: return Col4(p[0], p[1], p[2], 1);
  case 4: return Col4(p[0], p[1], p[2], p[3]);
  default: return Col4();
  }
}

// ======================================================= shading runtime (a9..)
enum { CXT_CAMERA_RAY = 0, CXT_SHADOW_RAY, CXT_DIFFUSE_RAY, CXT_REFLECT_RAY, CXT_REFRACT_RAY };

struct Cxt {
  int ray_context;
  int diffuse_depth, reflect_depth, refract_depth;
  int max_diffuse_depth, max_reflect_depth, max_refract_depth;
  int cast_shadow;
  double time;
  float opacity_threshold;
  int trace_target;       // group index
  // counter-based RNG contract of the pathtracing restatement:
  uint32_t sample_uid;    // tile id * 2^20 + sample index inside the tile (64-bit, folded to 32: see render_tile)
  uint32_t path_key;      // 0 for the camera ray; child k of a ray with key p has 4 p + k
};

struct SurfIn {
  V3 P, N;
  Col Cd;
  float u, v;
  V3 I, dPdu, dPdv;
  int shaded_object;
};

struct RenderState {
  const Scene *sc;
  fj_ray_counts counts;
  // Serial-stream mode (pins the restatement to a ONE-thread reference render bit for bit):
  // the reference's own generators in its own draw order -- PathtracingShader's rng[thread 0],
  // one per shader instance (shaders/pathtracing_shader/pathtracing_shader.cc:50,186-188), and
  // one XorShift per area light (src/fj_rectangle_light.h:23, src/fj_sphere_light.cc:9,33),
  // all default seeded, never re-seeded, drawn from in shading-call order over the whole frame
  // (tiles in queue order, samples in raster order, recursion depth first).  Off: the
  // counter-based contract shared with the device (see pt_draw2 / area_stream).
  bool serial_rng = false;
  std::vector<XorShift> shader_rng, light_rng;
  double cos_half_pi, cos_pi;
};

static int SlTrace(RenderState *rs, const Cxt &cxt, const V3 &orig, const V3 &dir,
    double tmin, double tmax, Col4 *out, double *t_hit);

static V3 Faceforward(const V3 &I, const V3 &N) { return (Dot(I, N) < 0) ? N : V3(-N.x, -N.y, -N.z); }  // src/fj_shading.cc:42-51

static double Fresnel(const V3 &I, const V3 &N, double ior)   // :53-73
{
  double cos = -1 * Dot(I, N);
  double eta;
  if (cos > 0) eta = ior;
  else { eta = 1. / ior; cos *= -1; }
  const double k2 = .0;
  const double F0 = ((1. - eta) * (1. - eta) + k2) / ((1. + eta) * (1. + eta) + k2);
  return F0 + (1. - F0) * std::pow(1. - cos, 5.);
}

static V3 Reflect(const V3 &I, const V3 &N)                    // :90-98
{
  const double cos = -1 * Dot(I, N);
  return V3(I.x + 2 * cos * N.x, I.y + 2 * cos * N.y, I.z + 2 * cos * N.z);
}

static V3 Refract(const V3 &I, const V3 &N, double ior)        // :100-138
{
  V3 n;
  double eta;
  double cos1 = -1 * Dot(I, N);
  if (cos1 < 0) { cos1 *= -1; eta = 1 / ior; n = V3(-N.x, -N.y, -N.z); }
  else { eta = ior; n = N; }
  const double radicand = 1 - eta * eta * (1 - cos1 * cos1);
  if (radicand < 0.) return Reflect(I, N);   // total internal reflection
  const double ncoeff = eta * cos1 - std::sqrt(radicand);
  return V3(eta * I.x + ncoeff * n.x, eta * I.y + ncoeff * n.y, eta * I.z + ncoeff * n.z);
}

static float Luminance4(const Col4 &c) { return .298912 * c.r + .586611 * c.g + .114478 * c.b; }   // src/fj_color.h:280-283 (f64 weights)

// src/fj_shading.cc:418-464
static V3 BumpMapping(const fj_texture_desc &bump, const V3 &dPdu, const V3 &dPdv,
    float tu, float tv, double amplitude, const V3 &N)
{
  const int xres = bump.width, yres = bump.height;
  if (xres == 0 || yres == 0) return N;   // N_bump left untouched; callers pass Nf
  const float du = 1. / xres;
  const float dv = 1. / yres;
  float val0 = Luminance4(TextureLookup(bump, tu - du, tv));
  float val1 = Luminance4(TextureLookup(bump, tu + du, tv));
  const float Bu = (val0 - val1) / (2 * du);
  val0 = Luminance4(TextureLookup(bump, tu, tv - dv));
  val1 = Luminance4(TextureLookup(bump, tu, tv + dv));
  const float Bv = (val0 - val1) / (2 * dv);
  V3 N_dPdu = Cross(N, dPdu);
  V3 N_dPdv = Cross(N, dPdv);
  N_dPdu = V3(N_dPdu.x * du, N_dPdu.y * du, N_dPdu.z * du);
  N_dPdv = V3(N_dPdv.x * du, N_dPdv.y * du, N_dPdv.z * du);
  V3 nb(N.x + amplitude * (Bv * N_dPdu.x - Bu * N_dPdv.x),
        N.y + amplitude * (Bv * N_dPdu.y - Bu * N_dPdv.y),
        N.z + amplitude * (Bv * N_dPdu.z - Bu * N_dPdv.z));
  return Normalize(nb);
}

// Light::Illuminate per type (a32)
static Col Illuminate(const fj_light_desc &L, const LightSample &s, const V3 &Ps)
{
  switch (L.type) {
  case FJ_GRID_LIGHT: {    // RectangleLight::illuminate, src/fj_rectangle_light.cc:47-59
    const V3 Ln = Normalize(Ps - s.P);
    double dot = Dot(Ln, s.N);
    dot = L.double_sided ? std::abs(dot) : Max(dot, 0.);
    const float si = L.intensity / L.sample_count;
    const float k = dot * si;                       // Real * float, then operator*(float, Color)
    return Col(k * L.color[0], k * L.color[1], k * L.color[2]);
  }
  case FJ_SPHERE_LIGHT: {  // SphereLight::illuminate, src/fj_sphere_light.cc:47-59
    const V3 Ln = Normalize(Ps - s.P);
    if (Dot(Ln, s.N) > 0) {
      const float si = L.intensity / L.sample_count;
      return Col(si * L.color[0], si * L.color[1], si * L.color[2]);
    }
    return Col();
  }
  case FJ_POINT_LIGHT:     // src/fj_point_light.cc:36-39
    return Col(L.intensity * L.color[0], L.intensity * L.color[1], L.intensity * L.color[2]);
  case FJ_DOME_LIGHT: {    // src/fj_dome_light.cc:53-56; sample_intensity = intensity / sample_count
    const float si = L.intensity / L.sample_count;
    return Col(si * s.color.r, si * s.color.g, si * s.color.b);
  }
  default:
    return Col();
  }
}

struct LightOut { Col Cl; V3 Ln; double distance; };

// SlIlluminance, src/fj_shading.cc:296-359
static int Illuminance(RenderState *rs, const Cxt &cxt, const LightSample &smp, const V3 &Ps,
    const V3 &axis, double cosangle_limit, const SurfIn &in, LightOut *out)
{
  out->Cl = Col();
  out->Ln = V3(smp.P.x - Ps.x, smp.P.y - Ps.y, smp.P.z - Ps.z);
  out->distance = Length(out->Ln);
  if (out->distance > 0) {
    const double inv = 1. / out->distance;
    out->Ln = V3(out->Ln.x * inv, out->Ln.y * inv, out->Ln.z * inv);
  }
  const V3 nml_axis = Normalize(axis);
  const double cosangle = Dot(nml_axis, out->Ln);
  if (cosangle < cosangle_limit) return 0;

  Col lc = Illuminate(rs->sc->d->lights[smp.light], smp, Ps);
  if (lc.r < .0001 && lc.g < .0001 && lc.b < .0001) return 0;
  if (cxt.ray_context == CXT_SHADOW_RAY) return 0;

  if (cxt.cast_shadow) {
    Cxt sh = cxt;                                   // SlShadowContext, :266-279
    sh.ray_context = CXT_SHADOW_RAY;
    sh.max_diffuse_depth = sh.max_reflect_depth = sh.max_refract_depth = 0;
    sh.trace_target = rs->sc->d->instances[in.shaded_object].shadow_target;
    Col4 C_occl;
    double t_hit = FLT_MAX;
    const int hit = SlTrace(rs, sh, Ps, out->Ln, .0001, out->distance, &C_occl, &t_hit);
    if (hit) {
      const float ac = 1 - C_occl.a;
      lc.r *= ac; lc.g *= ac; lc.b *= ac;
    }
  }
  out->Cl = lc;
  return 1;
}

struct SurfOut { Col Cs; float Os; };

// ---- area lights (RectangleLight / SphereLight get_samples, src/fj_rectangle_light.cc:20-45,
// src/fj_sphere_light.cc:22-45).  The reference draws the sample positions from ONE XorShift
// per light, shared unsynchronised by all worker threads: its image depends on the schedule.
// RNG contract shared with the device: per shading event (sample_uid, path_key)
// and light index L the stream is the reference's seeded XorShift with
//   seed = mix(mix(uid, key) ^ 0x51ED270B, L),  four warm-up draws,
// then the draws of that light's samples in order (rectangle: x, z per sample; sphere:
// HollowSphereRand's rejection loop, three draws per attempt).
static uint32_t pt_mix(uint32_t uid, uint32_t key);
static XorShift area_stream(uint32_t uid, uint32_t key, int light)
{
  uint32_t seed = pt_mix(pt_mix(uid, key) ^ 0x51ED270Bu, (uint32_t) light);
  XorShift r;
  for (uint32_t i = 0; i < 4; i++) r.s[i] = seed = 1812433253U * (seed ^ (seed >> 30)) + i;   // XorShift(unsigned), src/fj_random.cc:18-24
  for (int i = 0; i < 4; i++) r.NextInteger();
  return r;
}

// SlNewLightSamples for one shading event: the static table, with the samples of area
// lights generated in place (same order: lights in scene order, samples in draw order)
static const std::vector<LightSample> &event_light_samples(RenderState *rs, const Cxt &cxt, std::vector<LightSample> *tmp)
{
  const Scene &sc = *rs->sc;
  if (!sc.has_area_lights) return sc.light_samples;
  *tmp = sc.light_samples;
  int cur = -1;
  XorShift event_rng;
  for (LightSample &s : *tmp) {
    const fj_light_desc &L = sc.d->lights[s.light];
    if (L.type != FJ_GRID_LIGHT && L.type != FJ_SPHERE_LIGHT) continue;
    if (s.light != cur && !rs->serial_rng) { cur = s.light; event_rng = area_stream(cxt.sample_uid, cxt.path_key, cur); }
    XorShift &rng = rs->serial_rng ? rs->light_rng[s.light] : event_rng;
    const Xfm &x = sc.light_xfm[s.light];
    if (L.type == FJ_GRID_LIGHT) {
      const double px = rng.NextFloat01() - .5;
      const double pz = rng.NextFloat01() - .5;
      s.P = MatTransformPoint(x.matrix, V3(px, 0, pz));
      s.N = Normalize(MatTransformVector(x.matrix, V3(0, 1, 0)));
    } else {
      V3 o;
      double dot;
      for (;;) {                                      // XorShift::HollowSphereRand, src/fj_random.cc:69-87
        o.x = 2 * rng.NextFloat01() - 1;
        o.y = 2 * rng.NextFloat01() - 1;
        o.z = 2 * rng.NextFloat01() - 1;
        dot = Dot(o, o);
        if (dot > 0 && dot <= 1) break;
      }
      const V3 p = o / std::sqrt(dot);
      s.P = MatTransformPoint(x.matrix, p);
      s.N = Normalize(MatTransformVector(x.matrix, p));
    }
  }
  return *tmp;
}

static Cxt ReflectCxt(const RenderState *rs, const Cxt &c, int obj)   // :230-240
{
  Cxt r = c; r.reflect_depth++; r.ray_context = CXT_REFLECT_RAY; r.path_key = 4 * c.path_key + 2;
  r.trace_target = rs->sc->d->instances[obj].reflect_target; return r;
}
static Cxt RefractCxt(const RenderState *rs, const Cxt &c, int obj)   // :242-252
{
  Cxt r = c; r.refract_depth++; r.ray_context = CXT_REFRACT_RAY; r.path_key = 4 * c.path_key + 3;
  r.trace_target = rs->sc->d->instances[obj].refract_target; return r;
}
static Cxt DiffuseCxt(const RenderState *rs, const Cxt &c, int obj)   // :218-228 (reflect target!)
{
  Cxt r = c; r.diffuse_depth++; r.ray_context = CXT_DIFFUSE_RAY; r.path_key = 4 * c.path_key + 1;
  r.trace_target = rs->sc->d->instances[obj].reflect_target; return r;
}

// shaders/plastic_shader/plastic_shader.cc:101-179
static void PlasticEvaluate(RenderState *rs, const fj_shader_desc &sh, const Cxt &cxt, const SurfIn &in, SurfOut *out)
{
  const fj_scene_desc *d = rs->sc->d;
  Col diff, spec;
  Col4 diff_map(1, 1, 1, 1);
  V3 Nf = Faceforward(in.I, in.N);
  if (sh.bump_map >= 0)
    Nf = BumpMapping(d->textures[sh.bump_map], in.dPdu, in.dPdv, in.u, in.v, sh.bump_amplitude, Nf);

  std::vector<LightSample> tmp_samples;
  const std::vector<LightSample> &samples = event_light_samples(rs, cxt, &tmp_samples);
  for (size_t i = 0; i < samples.size(); i++) {
    LightOut L;
    Illuminance(rs, cxt, samples[i], in.P, Nf, rs->cos_half_pi, in, &L);
    float Kd = Dot(Nf, L.Ln);
    Kd = Max(0, Kd);
    diff.r += Kd * L.Cl.r;
    diff.g += Kd * L.Cl.g;
    diff.b += Kd * L.Cl.b;
  }
  if (sh.diffuse_map >= 0) diff_map = TextureLookup(d->textures[sh.diffuse_map], in.u, in.v);

  out->Cs.r = diff.r * sh.diffuse[0] * diff_map.r + spec.r;
  out->Cs.g = diff.g * sh.diffuse[1] * diff_map.g + spec.g;
  out->Cs.b = diff.b * sh.diffuse[2] * diff_map.b + spec.b;

  if (sh.do_reflect) {
    Col4 C_refl;
    double t_hit = REAL_MAX;
    const Cxt rc = ReflectCxt(rs, cxt, in.shaded_object);
    const V3 R = Normalize(Reflect(in.I, Nf));
    SlTrace(rs, rc, in.P, R, .001, 1000, &C_refl, &t_hit);
    const double Kr = Fresnel(in.I, Nf, 1 / sh.ior);     // 1/ior in f32
    out->Cs.r += Kr * C_refl.r * sh.reflect[0];
    out->Cs.g += Kr * C_refl.g * sh.reflect[1];
    out->Cs.b += Kr * C_refl.b * sh.reflect[2];
  }
  out->Os = sh.opacity;
}

// shaders/constant_shader/constant_shader.cc:72-96
static void ConstantEvaluate(RenderState *rs, const fj_shader_desc &sh, const SurfIn &in, SurfOut *out)
{
  Col4 C;
  if (sh.texture >= 0) {
    C = TextureLookup(rs->sc->d->textures[sh.texture], in.u, in.v);
    C.r *= sh.diffuse[0]; C.g *= sh.diffuse[1]; C.b *= sh.diffuse[2];
  } else {
    C.r = sh.diffuse[0]; C.g = sh.diffuse[1]; C.b = sh.diffuse[2];
  }
  out->Cs = Col(C.r, C.g, C.b);
  out->Os = 1;
}

// shaders/glass_shader/glass_shader.cc:88-130
static void GlassEvaluate(RenderState *rs, const fj_shader_desc &sh, const Cxt &cxt, const SurfIn &in, SurfOut *out)
{
  Col4 C_refl, C_refr;
  double t_hit = REAL_MAX;
  out->Cs = Col();
  const double Kr = Fresnel(in.I, in.N, 1 / sh.ior);
  const double Kt = 1 - Kr;

  const Cxt rc = ReflectCxt(rs, cxt, in.shaded_object);
  const V3 R = Normalize(Reflect(in.I, in.N));
  SlTrace(rs, rc, in.P, R, .0001, 1000, &C_refl, &t_hit);
  out->Cs.r += Kr * C_refl.r;
  out->Cs.g += Kr * C_refl.g;
  out->Cs.b += Kr * C_refl.b;

  const Cxt tc = RefractCxt(rs, cxt, in.shaded_object);
  const V3 T = Normalize(Refract(in.I, in.N, 1 / sh.ior));
  SlTrace(rs, tc, in.P, T, .0001, 1000, &C_refr, &t_hit);
  if (sh.do_color_filter && Dot(in.I, in.N) < 0) {
    C_refr.r *= std::pow(sh.filter_color[0], t_hit);
    C_refr.g *= std::pow(sh.filter_color[1], t_hit);
    C_refr.b *= std::pow(sh.filter_color[2], t_hit);
  }
  out->Cs.r += Kt * C_refr.r;
  out->Cs.g += Kt * C_refr.g;
  out->Cs.b += Kt * C_refr.b;
  out->Os = 1;
}

static void HairEvaluate(RenderState *rs, const fj_shader_desc &sh, const Cxt &cxt, const SurfIn &in, SurfOut *out);
static void PathtracingEvaluate(RenderState *rs, const fj_shader_desc &sh, const Cxt &cxt, const SurfIn &in, SurfOut *out);

// has_reached_bounce_limit, src/fj_shading.cc:467-499
static int reached_bounce_limit(const Cxt &c)
{
  int cur = 0, mx = 0;
  switch (c.ray_context) {
  case CXT_CAMERA_RAY: case CXT_SHADOW_RAY: cur = 0; mx = 1; break;
  case CXT_DIFFUSE_RAY: cur = c.diffuse_depth; mx = c.max_diffuse_depth; break;
  case CXT_REFLECT_RAY: cur = c.reflect_depth; mx = c.max_reflect_depth; break;
  case CXT_REFRACT_RAY: cur = c.refract_depth; mx = c.max_refract_depth; break;
  }
  return cur > mx;
}

// SlTrace + trace_surface, src/fj_shading.cc:140-179,527-572.  No volumes are
// in scope: with an empty volume set raymarch_volume returns 0 and the
// composite `vol + surf*(1-vol.a)` is the surface colour.
static int SlTrace(RenderState *rs, const Cxt &cxt, const V3 &orig, const V3 &dir,
    double tmin, double tmax, Col4 *out, double *t_hit)
{
  *out = Col4();
  if (reached_bounce_limit(cxt)) return 0;
  switch (cxt.ray_context) {
  case CXT_CAMERA_RAY: rs->counts.camera++; break;
  case CXT_SHADOW_RAY: rs->counts.shadow++; break;
  case CXT_DIFFUSE_RAY: rs->counts.diffuse++; break;
  case CXT_REFLECT_RAY: rs->counts.reflect++; break;
  case CXT_REFRACT_RAY: rs->counts.refract++; break;
  }
  Ray ray{orig, dir, tmin, tmax};
  Isect isect;
  if (!GroupIntersect(*rs->sc, cxt.trace_target, ray, cxt.time, &isect)) return 0;

  SurfIn in;
  in.shaded_object = isect.object;
  in.P = isect.P; in.N = isect.N; in.Cd = isect.Cd; in.u = isect.u; in.v = isect.v;
  in.I = ray.dir; in.dPdu = isect.dPdu; in.dPdv = isect.dPdv;

  // Intersection::GetShader -> ObjectInstance::GetShader, src/fj_object_instance.cc:177-191
  const fj_instance_desc &inst = rs->sc->d->instances[isect.object];
  int sid;
  if (isect.shading_group_id < 0 || isect.shading_group_id >= inst.n_shaders) sid = inst.shaders[0];
  else { sid = inst.shaders[isect.shading_group_id]; if (sid < 0) sid = inst.shaders[0]; }

  SurfOut so;
  so.Cs = Col(.5f, 1.f, 0.f);   // NO_SHADER_COLOR, :24
  so.Os = 1;
  if (sid >= 0) {
    const fj_shader_desc &sh = rs->sc->d->shaders[sid];
    switch (sh.type) {
    case FJ_SHADER_PLASTIC: PlasticEvaluate(rs, sh, cxt, in, &so); break;
    case FJ_SHADER_CONSTANT: ConstantEvaluate(rs, sh, in, &so); break;
    case FJ_SHADER_GLASS: GlassEvaluate(rs, sh, cxt, in, &so); break;
    case FJ_SHADER_HAIR: HairEvaluate(rs, sh, cxt, in, &so); break;
    case FJ_SHADER_PATHTRACING: PathtracingEvaluate(rs, sh, cxt, in, &so); break;
    default: break;
    }
  }
  so.Os = Clamp(so.Os, 0, 1);
  *out = Col4(so.Cs.r, so.Cs.g, so.Cs.b, so.Os);
  *t_hit = isect.t_hit;
  return 1;
}

// shaders/hair_shader/hair_shader.cc:87-117,184-206 (Kajiya-Kay; tangent = dPdv;
// the illuminance axis is in.N, which Curve::ray_intersect leaves at zero, so the
// cone test against cos(PI) always passes)
static void HairEvaluate(RenderState *rs, const fj_shader_desc &sh, const Cxt &cxt, const SurfIn &in, SurfOut *out)
{
  out->Cs = Col();
  std::vector<LightSample> tmp_samples;
  const std::vector<LightSample> &samples = event_light_samples(rs, cxt, &tmp_samples);
  for (size_t i = 0; i < samples.size(); i++) {
    LightOut L;
    L.Cl = Col(); L.Ln = V3(); L.distance = 0;
    Illuminance(rs, cxt, samples[i], in.P, in.N, rs->cos_pi, in, &L);
    const V3 tangent = Normalize(in.dPdv);
    // kajiya_diffuse / kajiya_specular
    const float TL = Dot(tangent, L.Ln);
    // the plugin calls the C library's double sqrt / pow on float arguments
    // (hair_shader.cc includes no <cmath>): products and the sum are formed in f64
    const float diff = ::sqrt((double) (1 - TL * TL));
    const float roughness = .05;
    const float TI = Dot(tangent, in.I);
    float spec = ::sqrt((double) (1 - TL * TL)) * ::sqrt((double) (1 - TI * TI)) + TL * TI;
    spec = ::pow((double) spec, (double) (1 / roughness));
    out->Cs.r += (in.Cd.r * sh.diffuse[0] * diff + spec) * L.Cl.r;
    out->Cs.g += (in.Cd.g * sh.diffuse[1] * diff + spec) * L.Cl.g;
    out->Cs.b += (in.Cd.b * sh.diffuse[2] * diff + spec) * L.Cl.b;
  }
  out->Os = 1;
}

// shaders/pathtracing_shader/pathtracing_shader.cc:125-257.
//
// RNG contract.  The plugin draws from `mutable XorShift rng[64]` indexed by thread id
// (:50,186-188): its stream depends on the global shading order and is not reproducible
// in parallel, not even by the reference itself.  This restatement -- and the
// device path, identically -- replaces it by a counter-based stream: the two numbers of a
// diffuse bounce come from the reference's own seeded generator XorShift(seed)
// (src/fj_random.cc:18-24) with seed = mix(sample_uid, path_key), after four warm-up
// draws.  Everything else (ONB, cosine-weighted direction, the extra N.D factor, Fresnel
// weights, per-type depth limits, transmit colour filter) follows the plugin line by line.
// Parity: GPU == this restatement (1e-4); this restatement ~ reference statistically
// (tests/test_oracle_golden.py::test_pathtracing_matches_reference_statistically).
static uint32_t pt_mix(uint32_t uid, uint32_t key)
{
  uint32_t h = uid * 0x9E3779B1u ^ (key + 0x7F4A7C15u) * 0x85EBCA77u;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}

static void pt_draw2(RenderState *rs, int shader, uint32_t uid, uint32_t key, double *x1, double *x2)
{
  if (rs->serial_rng) {          // rng[MtGetThreadID() = 0] of this shader instance, :186-188
    *x1 = rs->shader_rng[shader].NextFloat01();
    *x2 = rs->shader_rng[shader].NextFloat01();
    return;
  }
  uint32_t st[4];
  uint32_t seed = pt_mix(uid, key);
  for (uint32_t i = 0; i < 4; i++) st[i] = seed = 1812433253U * (seed ^ (seed >> 30)) + i;   // XorShift(unsigned)
  XorShift r;
  for (int i = 0; i < 4; i++) r.s[i] = st[i];
  for (int i = 0; i < 4; i++) r.NextInteger();
  *x1 = r.NextFloat01();
  *x2 = r.NextFloat01();
}

static float Luminance3(const float c[3]) { return .298912 * c[0] + .586611 * c[1] + .114478 * c[2]; }

static void PathtracingEvaluate(RenderState *rs, const fj_shader_desc &sh, const Cxt &cxt, const SurfIn &in0, SurfOut *out)
{
  const fj_scene_desc *d = rs->sc->d;
  SurfIn in = in0;
  if (sh.diffuse_map >= 0) {
    const Col4 c = TextureLookup(d->textures[sh.diffuse_map], in0.u, in0.v);
    in.Cd.r *= c.r; in.Cd.g *= c.g; in.Cd.b *= c.b;
  }
  if (sh.bump_map >= 0)
    in.N = BumpMapping(d->textures[sh.bump_map], in0.dPdu, in0.dPdv, in0.u, in0.v, sh.bump_amplitude, in0.N);

  Col Lo(sh.emission[0], sh.emission[1], sh.emission[2]);

  if (Luminance3(sh.diffuse) > 0.) {                         // integrate_diffuse, :170-203
    const V3 w = in.N;
    V3 u = std::abs(w.x) > .001 ? V3(0, 1, 0) : V3(1, 0, 0);
    u = Normalize(Cross(u, w));
    const V3 v = Cross(w, u);
    double x1, x2;
    pt_draw2(rs, (int) (&sh - d->shaders), cxt.sample_uid, cxt.path_key, &x1, &x2);
    const double r1 = 2. * PI * x1;
    const double r2 = x2;
    const double r2sqrt = std::sqrt(r2);
    const V3 D = Normalize(u * std::cos(r1) * r2sqrt + v * std::sin(r1) * r2sqrt + w * std::sqrt(1. - r2));
    const double Kd = Dot(in.N, D);
    Col4 C;
    double t_hit = REAL_MAX;
    const Cxt dc = DiffuseCxt(rs, cxt, in.shaded_object);
    SlTrace(rs, dc, in.P, D, .001, 1000, &C, &t_hit);
    const float kd = Kd;                                      // Color * Real -> Color * float
    Lo.r += in.Cd.r * kd * sh.diffuse[0] * C.r;
    Lo.g += in.Cd.g * kd * sh.diffuse[1] * C.g;
    Lo.b += in.Cd.b * kd * sh.diffuse[2] * C.b;
  }
  if (Luminance3(sh.reflect) > 0.) {                         // integrate_reflect, :205-224
    const V3 R = Normalize(Reflect(in.I, in.N));
    const double Kr = Fresnel(in.I, in.N, 1. / sh.ior);
    Col4 C;
    double t_hit = REAL_MAX;
    const Cxt rc = ReflectCxt(rs, cxt, in.shaded_object);
    SlTrace(rs, rc, in.P, R, .001, 1000, &C, &t_hit);
    const float kr = Kr;
    Lo.r += kr * sh.reflect[0] * C.r;
    Lo.g += kr * sh.reflect[1] * C.g;
    Lo.b += kr * sh.reflect[2] * C.b;
  }
  if (Luminance3(sh.refract) > 0.) {                         // integrate_refract, :226-257
    const V3 T = Normalize(Refract(in.I, in.N, 1. / sh.ior));
    const double Kr = Fresnel(in.I, in.N, 1 / sh.ior);       // 1/ior in f32 here, as in the plugin
    const double Kt = 1 - Kr;
    Col4 C;
    double t_hit = REAL_MAX;
    const Cxt tc = RefractCxt(rs, cxt, in.shaded_object);
    SlTrace(rs, tc, in.P, T, .0001, 1000, &C, &t_hit);
    if (sh.do_color_filter && Dot(in.I, in.N) < 0) {
      C.r *= std::pow(sh.filter_color[0], t_hit);
      C.g *= std::pow(sh.filter_color[1], t_hit);
      C.b *= std::pow(sh.filter_color[2], t_hit);
    }
    const float kt = Kt;
    Lo.r += kt * sh.refract[0] * C.r;
    Lo.g += kt * sh.refract[1] * C.g;
    Lo.b += kt * sh.refract[2] * C.b;
  }
  out->Cs = Lo;
  out->Os = 1;
}

// =============================================================== light samples
// SlNewLightSamples (src/fj_shading.cc:380-404) for the deterministic light
// types: PointLight (src/fj_point_light.cc:21-34) and DomeLight
// (src/fj_dome_light.cc:30-51).  Both evaluate the light transform at time 0,
// so the sample array is identical for every shading event: built once.
static int build_light_samples(Scene *sc)
{
  const fj_scene_desc *d = sc->d;
  sc->light_samples.clear();
  sc->light_xfm.resize(d->n_lights);
  sc->has_area_lights = false;
  for (int i = 0; i < d->n_lights; i++) {
    const fj_light_desc &L = d->lights[i];
    Xfm x;
    LerpXfm(L.xform, 0, &x);
    sc->light_xfm[i] = x;
    if (L.type == FJ_POINT_LIGHT) {
      LightSample s;
      s.light = i; s.P = x.translate; s.N = V3();
      sc->light_samples.push_back(s);
    } else if (L.type == FJ_DOME_LIGHT) {
      const int n = L.sample_count < L.n_dome_samples ? L.sample_count : L.n_dome_samples;
      for (int k = 0; k < n; k++) {
        const fj_dome_sample &ds = L.dome_samples[k];
        LightSample s;
        s.light = i;
        const V3 dir(ds.dir[0], ds.dir[1], ds.dir[2]);
        s.P = MatTransformPoint(x.matrix, dir * FLT_MAX);
        s.N = MatTransformVector(x.matrix, -1 * dir);
        s.color = Col(ds.color[0], ds.color[1], ds.color[2]);
        sc->light_samples.push_back(s);
      }
    } else if (L.type == FJ_GRID_LIGHT || L.type == FJ_SPHERE_LIGHT) {
      // get_sample_count() = sample density; positions are drawn per shading event
      for (int k = 0; k < L.sample_count; k++) {
        LightSample s;
        s.light = i;
        sc->light_samples.push_back(s);
      }
      sc->has_area_lights = true;
    } else {
      return -1;
    }
  }
  return 0;
}

// ================================================================ tile loop
static void render_tile(RenderState *rs, const fj_render_desc &r, const CameraState &cam,
    const Tile &tile, float *fb, std::vector<Sample> *samples, AdaptiveGrid *grid)
{
  const bool adaptive = r.sampler_type == 1;
  int ns[2], margin[2];
  if (adaptive) grid->Generate(r, tile);
  else GenerateSamples(r, tile, samples, ns);

  // integrate_samples, src/fj_renderer.cc:1061-1096
  Cxt cxt;                                 // SlCameraContext + init_worker overrides
  cxt.ray_context = CXT_CAMERA_RAY;
  cxt.diffuse_depth = cxt.reflect_depth = cxt.refract_depth = 0;
  cxt.max_diffuse_depth = r.max_diffuse_depth;
  cxt.max_reflect_depth = r.max_reflect_depth;
  cxt.max_refract_depth = r.max_refract_depth;
  cxt.cast_shadow = r.cast_shadow;
  cxt.opacity_threshold = .995f;
  cxt.trace_target = rs->sc->d->target_group;
  auto integrate = [&](Sample &s, uint32_t index_in_tile) {
    {
      // tile id * 2^20 + sample index as a 64-bit number, the high word folded into the low one: the plain
      // 32-bit sum for frames of up to 4096 tiles of up to 2^20 samples, distinct streams beyond
      const unsigned long long u = ((unsigned long long) (uint32_t) tile.id << 20) + index_in_tile;
      cxt.sample_uid = (uint32_t) u ^ ((uint32_t) (u >> 32) * 0x9E3779B1u);
    }
    cxt.path_key = 0;
    Ray ray;
    CameraGetRay(cam, s.uv, s.time, &ray);
    cxt.time = s.time;
    Col4 C;
    double t_hit = FLT_MAX;
    const int hit = SlTrace(rs, cxt, ray.orig, ray.dir, ray.tmin, ray.tmax, &C, &t_hit);
    if (hit) { s.data[0] = C.r; s.data[1] = C.g; s.data[2] = C.b; s.data[3] = C.a; }
    else { s.data[0] = s.data[1] = s.data[2] = s.data[3] = 0; }
  };
  // window of one pixel in the sample array: get_sampleset_in_pixel of either sampler
  int npx[2], step[2];
  const Sample *all;
  if (adaptive) {
    for (long k; (k = grid->Next()) >= 0; ) integrate(grid->samples[(size_t) k], (uint32_t) k);
    ns[0] = grid->nx; ns[1] = grid->ny;
    step[0] = step[1] = grid->div;         // src/fj_adaptive_grid_sampler.cc:172-193,210-213
    npx[0] = grid->div * (1 + 2 * grid->margin[0]) + 1;
    npx[1] = grid->div * (1 + 2 * grid->margin[1]) + 1;
    all = grid->samples.data();
  } else {
    uint32_t sample_k = 0;
    for (Sample &s : *samples) integrate(s, sample_k++);
    SamplerMargin(r, margin);
    step[0] = r.rate_x; step[1] = r.rate_y;
    npx[0] = r.rate_x + 2 * margin[0]; npx[1] = r.rate_y + 2 * margin[1];
    all = samples->data();
  }

  // reconstruct_image + apply_pixel_filter, src/fj_renderer.cc:939-995
  const double fw = (double) r.filter_w, fh = (double) r.filter_h;
  for (int y = tile.ymin; y < tile.ymax; y++)
    for (int x = tile.xmin; x < tile.xmax; x++) {
      const Sample *src = all + static_cast<size_t>(y - tile.ymin) * step[1] * ns[0] + (x - tile.xmin) * step[0];
      float px[4] = {0, 0, 0, 0};
      float wgt_sum = 0.f;
      for (int sy = 0; sy < npx[1]; sy++)
        for (int sx = 0; sx < npx[0]; sx++) {
          const Sample &s = src[static_cast<size_t>(sy) * ns[0] + sx];
          const double filtx = r.xres * s.uv[0] - (x + .5);
          const double filty = r.yres * (1 - s.uv[1]) - (y + .5);
          const double wgt = GaussianFilter(fw, fh, filtx, filty);
          px[0] += wgt * s.data[0];
          px[1] += wgt * s.data[1];
          px[2] += wgt * s.data[2];
          px[3] += wgt * s.data[3];
          wgt_sum += wgt;
        }
      const float inv_sum = 1.f / wgt_sum;
      float *dst = fb + (static_cast<size_t>(y) * r.xres + x) * 4;   // FrameBuffer::SetColor, src/fj_framebuffer.cc:103-128
      for (int c = 0; c < 4; c++) dst[c] = px[c] * inv_sum;
    }
}

int RenderTiles(Scene *sc, const fj_render_desc &r, const int32_t *tile_ids, int n_tiles,
    float *fb, int nthreads, fj_ray_counts *counts_out, int serial_rng)
{
  if (serial_rng) nthreads = 1;                    // one worker, tiles in queue order: the reference with thread_count 1
  if (r.sampler_type != 0 && r.sampler_type != 1) return -2;
  if (r.sampler_type == 1 && (r.adaptive_max_subdivision < 0 || r.adaptive_max_subdivision > 8)) return -2;
  if (build_light_samples(sc)) return -3;
  std::vector<Tile> tiles;
  GenerateTiles(r, &tiles);
  std::vector<int> ids;
  if (tile_ids) ids.assign(tile_ids, tile_ids + n_tiles);
  else for (size_t i = 0; i < tiles.size(); i++) ids.push_back((int) i);
  for (int id : ids) if (id < 0 || id >= (int) tiles.size()) return -4;

  CameraState cam;
  CameraInit(&sc->d->camera, r.xres, r.yres, &cam);

  if (nthreads < 1) nthreads = 1;
  std::atomic<size_t> next(0);                     // dynamic tile queue, src/fj_multi_thread.cc:86-132
  std::vector<fj_ray_counts> counts(nthreads);
  auto worker = [&](int tid) {
    RenderState rs;
    rs.sc = sc;
    std::memset(&rs.counts, 0, sizeof(rs.counts));
    rs.cos_half_pi = std::cos(PI / 2.);
    rs.cos_pi = std::cos(PI);
    rs.serial_rng = serial_rng != 0;
    if (rs.serial_rng) {
      rs.shader_rng.assign((size_t) sc->d->n_shaders, XorShift());
      rs.light_rng.assign((size_t) sc->d->n_lights, XorShift());
    }
    std::vector<Sample> samples;
    AdaptiveGrid grid;
    for (;;) {
      const size_t k = next.fetch_add(1);
      if (k >= ids.size()) break;
      render_tile(&rs, r, cam, tiles[ids[k]], fb, &samples, &grid);
    }
    counts[tid] = rs.counts;
  };
  std::vector<std::thread> th;
  for (int t = 1; t < nthreads; t++) th.emplace_back(worker, t);
  worker(0);
  for (auto &t : th) t.join();
  if (counts_out) {
    std::memset(counts_out, 0, sizeof(*counts_out));
    for (auto &c : counts) {
      counts_out->camera += c.camera; counts_out->shadow += c.shadow; counts_out->diffuse += c.diffuse;
      counts_out->reflect += c.reflect; counts_out->refract += c.refract;
    }
  }
  return 0;
}

}  // namespace fjo
