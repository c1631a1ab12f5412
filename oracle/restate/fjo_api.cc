// oracle/restate/fjo_api.cc -- TEST INFRASTRUCTURE ONLY (CPU oracle).
//
// C entry points of liboracle.so (loaded with ctypes by tests/, smoke() and
// bench.py's cpu_baseline leg -- never by the product).  Two layers:
//   * whole-path: build the reference's accelerators from a fj_scene_desc and
//     render tiles / trace ray batches on the CPU;
//   * function-level: the individual reference functions that have golden
//     vectors in tests/golden/ (generated from the compiled reference by
//     oracle/ref_vectors.cc).
#include "fjo_render.h"

#include <cstring>

using namespace fjo;

extern "C" {

// ------------------------------------------------------------- whole path
void *fjo_scene_create(const fj_scene_desc *d)
{
  Scene *sc = new Scene();
  BuildScene(d, sc);
  return sc;
}

void fjo_scene_destroy(void *h) { delete static_cast<Scene *>(h); }

int fjo_scene_render(void *h, const fj_render_desc *r, const int32_t *tile_ids, int n_tiles,
    float *fb, int nthreads, fj_ray_counts *counts)
{
  return RenderTiles(static_cast<Scene *>(h), *r, tile_ids, n_tiles, fb, nthreads, counts);
}

// the same with the reference's serial random streams (RenderTiles: serial_rng)
int fjo_scene_render_serial(void *h, const fj_render_desc *r, const int32_t *tile_ids, int n_tiles,
    float *fb, fj_ray_counts *counts)
{
  return RenderTiles(static_cast<Scene *>(h), *r, tile_ids, n_tiles, fb, 1, counts, 1);
}

// closest hit of n rays against one group. rays: [n][8] = orig, dir, tmin, tmax.
// out_t[n] (REAL_MAX on miss), out_ids[n][2] = instance, prim (-1 on miss),
// out_attr[n][8] = N.xyz, u, v, P.xyz
int fjo_scene_trace(void *h, int group, int n, const double *rays, double time,
    double *out_t, int32_t *out_ids, double *out_attr)
{
  const Scene *sc = static_cast<Scene *>(h);
  if (group < 0 || group >= (int) sc->groups.size()) return -1;
  for (int i = 0; i < n; i++) {
    const double *r = rays + 8 * i;
    Ray ray{V3(r[0], r[1], r[2]), V3(r[3], r[4], r[5]), r[6], r[7]};
    Isect is;
    const bool hit = GroupIntersect(*sc, group, ray, time, &is);
    out_t[i] = hit ? is.t_hit : REAL_MAX;
    out_ids[2 * i] = hit ? is.object : -1;
    out_ids[2 * i + 1] = hit ? is.prim_id : -1;
    if (out_attr) {
      double *a = out_attr + 8 * i;
      if (hit) { a[0] = is.N.x; a[1] = is.N.y; a[2] = is.N.z; a[3] = is.u; a[4] = is.v; a[5] = is.P.x; a[6] = is.P.y; a[7] = is.P.z; }
      else std::memset(a, 0, 8 * sizeof(double));
    }
  }
  return 0;
}

// grid statistics of mesh i (for DESIGN.md / tests): ncells[3], total list entries
int fjo_scene_grid_info(void *h, int mesh, int32_t *ncells, int64_t *entries)
{
  const Scene *sc = static_cast<Scene *>(h);
  if (mesh < 0 || mesh >= (int) sc->meshes.size()) return -1;
  for (int i = 0; i < 3; i++) ncells[i] = sc->meshes[mesh].grid.ncells[i];
  *entries = (int64_t) sc->meshes[mesh].grid.cell_prims.size();
  return 0;
}

// instance world bounds (merge_sampled_bounds) -> out[6]
int fjo_scene_instance_bounds(void *h, int inst, double *out)
{
  const Scene *sc = static_cast<Scene *>(h);
  if (inst < 0 || inst >= (int) sc->instances.size()) return -1;
  const Box &b = sc->instances[inst].bounds;
  out[0] = b.min.x; out[1] = b.min.y; out[2] = b.min.z; out[3] = b.max.x; out[4] = b.max.y; out[5] = b.max.z;
  return 0;
}

// --------------------------------------------------------- function level
void fjo_xorshift_u32(int n, uint32_t *out)
{
  XorShift r;
  for (int i = 0; i < n; i++) out[i] = r.NextInteger();
}

void fjo_xorshift_f01(int n, double *out)
{
  XorShift r;
  for (int i = 0; i < n; i++) out[i] = r.NextFloat01();
}

// in: [n][14] = box min, box max, orig, dir, tmin, tmax ; out_hit[n], out_t[n][2]
void fjo_box_ray(int n, const double *in, int32_t *out_hit, double *out_t)
{
  for (int i = 0; i < n; i++) {
    const double *a = in + 14 * i;
    Box b(V3(a[0], a[1], a[2]), V3(a[3], a[4], a[5]));
    double t0 = 0, t1 = 0;
    const bool hit = BoxRayIntersect(b, V3(a[6], a[7], a[8]), V3(a[9], a[10], a[11]), a[12], a[13], &t0, &t1);
    out_hit[i] = hit;
    out_t[2 * i] = hit ? t0 : 0;
    out_t[2 * i + 1] = hit ? t1 : 0;
  }
}

// in: [n][15] = v0, v1, v2, orig, dir ; out_hit[n], out_tuv[n][3]
void fjo_tri_ray(int n, const double *in, int32_t *out_hit, double *out_tuv)
{
  for (int i = 0; i < n; i++) {
    const double *a = in + 15 * i;
    double t = 0, u = 0, v = 0;
    const bool hit = TriRayIntersect(V3(a[0], a[1], a[2]), V3(a[3], a[4], a[5]), V3(a[6], a[7], a[8]),
        V3(a[9], a[10], a[11]), V3(a[12], a[13], a[14]), &t, &u, &v);
    out_hit[i] = hit;
    out_tuv[3 * i] = hit ? t : 0; out_tuv[3 * i + 1] = hit ? u : 0; out_tuv[3 * i + 2] = hit ? v : 0;
  }
}

// trs: t xyz, r xyz, s xyz ; out matrix[16], inverse[16]
void fjo_make_transform(int transform_order, int rotate_order, const double *trs, double *m, double *inv)
{
  Xfm x;
  XfmSetTransform(&x, transform_order, rotate_order, trs[0], trs[1], trs[2], trs[3], trs[4], trs[5], trs[6], trs[7], trs[8]);
  std::memcpy(m, x.matrix.e, sizeof(double) * 16);
  std::memcpy(inv, x.inverse.e, sizeof(double) * 16);
}

void fjo_mat_inverse(const double *m, double *inv)
{
  Mat a, b;
  std::memcpy(a.e, m, sizeof(a.e));
  MatInverse(&b, a);
  std::memcpy(inv, b.e, sizeof(b.e));
}

// uvt: [n][3] = u, v, time ; out: [n][8] orig, dir, tmin, tmax
void fjo_camera_rays(const fj_camera_desc *cam, int xres, int yres, int n, const double *uvt, double *out)
{
  CameraState cs;
  CameraInit(cam, xres, yres, &cs);
  for (int i = 0; i < n; i++) {
    Ray r;
    CameraGetRay(cs, uvt + 3 * i, uvt[3 * i + 2], &r);
    double *o = out + 8 * i;
    o[0] = r.orig.x; o[1] = r.orig.y; o[2] = r.orig.z; o[3] = r.dir.x; o[4] = r.dir.y; o[5] = r.dir.z; o[6] = r.tmin; o[7] = r.tmax;
  }
}

// tiles of a frame: out [max][5] = id xmin ymin xmax ymax ; returns count
int fjo_tiles(const fj_render_desc *r, int32_t *out, int max_tiles)
{
  std::vector<Tile> t;
  GenerateTiles(*r, &t);
  for (size_t i = 0; i < t.size() && (int) i < max_tiles; i++) {
    out[5 * i] = t[i].id; out[5 * i + 1] = t[i].xmin; out[5 * i + 2] = t[i].ymin; out[5 * i + 3] = t[i].xmax; out[5 * i + 4] = t[i].ymax;
  }
  return (int) t.size();
}

// samples of one tile rect: out [n][3] = u, v, time ; returns n (= nx*ny), dims in nxy
int fjo_tile_samples(const fj_render_desc *r, const int32_t *rect, double *out, int max_samples, int32_t *nxy)
{
  Tile t{0, rect[0], rect[1], rect[2], rect[3]};
  std::vector<Sample> s;
  int ns[2];
  GenerateSamples(*r, t, &s, ns);
  nxy[0] = ns[0]; nxy[1] = ns[1];
  for (size_t i = 0; i < s.size() && (int) i < max_samples; i++) {
    out[3 * i] = s[i].uv[0]; out[3 * i + 1] = s[i].uv[1]; out[3 * i + 2] = s[i].time;
  }
  return (int) s.size();
}

void fjo_gaussian(int n, double xwidth, double ywidth, const double *xy, double *out)
{
  for (int i = 0; i < n; i++) out[i] = GaussianFilter(xwidth, ywidth, xy[2 * i], xy[2 * i + 1]);
}

void fjo_texture_lookup(const fj_texture_desc *tex, int n, const float *uv, float *out)
{
  for (int i = 0; i < n; i++) {
    const Col4 c = TextureLookup(*tex, uv[2 * i], uv[2 * i + 1]);
    out[4 * i] = c.r; out[4 * i + 1] = c.g; out[4 * i + 2] = c.b; out[4 * i + 3] = c.a;
  }
}

}  // extern "C"
