// oracle/ref_vectors.cc -- TEST INFRASTRUCTURE ONLY.
//
// Golden-vector generator: calls the UNMODIFIED reference's own functions
// (linked from oracle/_ref/libscene.so, headers included from /root/reference/src
// at compile time by oracle/Makefile) on seeded inputs and writes inputs +
// outputs as one small binary container.  tests/golden/make_golden.py runs it
// in the build container and commits the result (tests/golden/ref_vectors.bin);
// tests/test_oracle_golden.py replays the inputs through oracle/liboracle.so
// and requires bit-identical outputs.
//
// usage: ref_vectors out.bin [mesh.bin rays.bin]
//   mesh.bin: int32 n_points, n_faces; f64 P[n_points*3]; int32 idx[n_faces*3]
//   rays.bin: int32 n; f64 rays[n*8] (orig, dir, tmin, tmax)
#include "fj_box.h"
#include "fj_camera.h"
#include "fj_filter.h"
#include "fj_fixed_grid_sampler.h"
#include "fj_grid_accelerator.h"
#include "fj_intersection.h"
#include "fj_matrix.h"
#include "fj_mesh.h"
#include "fj_random.h"
#include "fj_ray.h"
#include "fj_rectangle.h"
#include "fj_tiler.h"
#include "fj_transform.h"
#include "fj_triangle.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace fj;

static FILE *g_out = NULL;

static void put(const std::string &name, char dtype, const std::vector<uint32_t> &dims, const void *data, size_t elem)
{
  const uint32_t nl = (uint32_t) name.size();
  fwrite(&nl, 4, 1, g_out);
  fwrite(name.data(), 1, nl, g_out);
  fwrite(&dtype, 1, 1, g_out);
  const uint32_t nd = (uint32_t) dims.size();
  fwrite(&nd, 4, 1, g_out);
  size_t n = 1;
  for (size_t i = 0; i < dims.size(); i++) { fwrite(&dims[i], 4, 1, g_out); n *= dims[i]; }
  fwrite(data, elem, n, g_out);
}
static void put_d(const std::string &n, const std::vector<uint32_t> &dims, const std::vector<double> &v) { put(n, 'd', dims, v.data(), 8); }
static void put_i(const std::string &n, const std::vector<uint32_t> &dims, const std::vector<int32_t> &v) { put(n, 'i', dims, v.data(), 4); }
static void put_u(const std::string &n, const std::vector<uint32_t> &dims, const std::vector<uint32_t> &v) { put(n, 'u', dims, v.data(), 4); }

// our own input generator (splitmix64)
struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next() { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
  double uni() { return (next() >> 11) * (1.0 / 9007199254740992.0); }
  double range(double a, double b) { return a + (b - a) * uni(); }
};

static void gen_xorshift()
{
  XorShift r;
  std::vector<uint32_t> u(4096);
  for (size_t i = 0; i < u.size(); i++) u[i] = r.NextInteger();
  put_u("xorshift_u32", {4096}, u);
  XorShift r2;
  std::vector<double> f(512);
  for (size_t i = 0; i < f.size(); i++) f[i] = r2.NextFloat01();
  put_d("xorshift_f01", {512}, f);
}

static void gen_box()
{
  Rng g(101);
  const int N = 3006;
  std::vector<double> in(N * 14), t(N * 2, 0.);
  std::vector<int32_t> hit(N);
  // six hand-built cases in the spirit of the reference's tests/box_test.cc:
  // inside, outside-front, tmax clipping, behind, short ray, reversed-infinite box
  const double hand[6][14] = {
    {-1, -1, -1, 1, 1, 1,   0, 0, 0,    0, 0, 1,   .001, 1000},
    {-1, -1, -1, 1, 1, 1,   0, 0, -5,   0, 0, 1,   .001, 1000},
    {-1, -1, -1, 1, 1, 1,   0, 0, -5,   0, 0, 1,   .001, 4.5},
    {-1, -1, -1, 1, 1, 1,   0, 0, 5,    0, 0, 1,   .001, 1000},
    {-1, -1, -1, 1, 1, 1,   0, 0, -5,   0, 0, 1,   .001, 3},
    {1.7976931348623157e308, 1.7976931348623157e308, 1.7976931348623157e308,
     -1.7976931348623157e308, -1.7976931348623157e308, -1.7976931348623157e308,   0, 0, 0,   0, 0, 1,   .001, 1000},
  };
  for (int i = 0; i < N; i++) {
    double *a = &in[i * 14];
    if (i < 6) { memcpy(a, hand[i], sizeof(hand[i])); }
    else {
      for (int k = 0; k < 3; k++) { const double c = g.range(-5, 5), h = g.range(.01, 3); a[k] = c - h; a[3 + k] = c + h; }
      for (int k = 0; k < 3; k++) a[6 + k] = g.range(-8, 8);
      // aim at a point near the box so that about half the rays hit
      double tgt[3], len = 0;
      for (int k = 0; k < 3; k++) { tgt[k] = g.range(a[k] - 1.5, a[3 + k] + 1.5); a[9 + k] = tgt[k] - a[6 + k]; len += a[9 + k] * a[9 + k]; }
      len = std::sqrt(len);
      for (int k = 0; k < 3; k++) a[9 + k] /= len;
      const int mode = (int) (g.next() % 16);
      if (mode == 0) a[9 + (g.next() % 3)] = 0;                       // axis-parallel component
      if (mode == 1) { a[9] = 0; a[10] = 0; a[11] = 1; }               // axis-aligned ray
      if (mode == 2) a[6 + (g.next() % 3)] = a[g.next() % 3];          // origin on a slab plane
      if (mode == 3) { for (int k = 0; k < 3; k++) a[6 + k] = .5 * (a[k] + a[3 + k]); }   // origin inside
      a[12] = (mode == 4) ? g.range(0, 5) : .001;
      a[13] = (mode == 5) ? g.range(0, 5) : 1000;
    }
    Box b;
    b.min = Vector(a[0], a[1], a[2]);
    b.max = Vector(a[3], a[4], a[5]);
    double t0 = 0, t1 = 0;
    hit[i] = BoxRayIntersect(b, Vector(a[6], a[7], a[8]), Vector(a[9], a[10], a[11]), a[12], a[13], &t0, &t1);
    if (hit[i]) { t[2 * i] = t0; t[2 * i + 1] = t1; }
  }
  put_d("box_in", {(uint32_t) N, 14}, in);
  put_i("box_hit", {(uint32_t) N}, hit);
  put_d("box_t", {(uint32_t) N, 2}, t);
}

static void gen_tri()
{
  Rng g(202);
  const int N = 3000;
  std::vector<double> in(N * 15), tuv(N * 3, 0.);
  std::vector<int32_t> hit(N);
  for (int i = 0; i < N; i++) {
    double *a = &in[i * 15];
    for (int k = 0; k < 9; k++) a[k] = g.range(-2, 2);
    for (int k = 0; k < 3; k++) a[9 + k] = g.range(-6, 6);
    // target: barycentric point, sometimes exactly on an edge / vertex / outside
    double b0 = g.uni(), b1 = g.uni() * (1 - b0);
    const int mode = (int) (g.next() % 10);
    if (mode == 0) b1 = 0;
    if (mode == 1) { b0 = 0; }
    if (mode == 2) { b0 = 1; b1 = 0; }
    if (mode == 3) { b0 = g.range(-.3, 1.3); b1 = g.range(-.3, 1.3); }
    double tgt[3], len = 0;
    for (int k = 0; k < 3; k++) {
      tgt[k] = a[k] + b0 * (a[3 + k] - a[k]) + b1 * (a[6 + k] - a[k]);
      a[12 + k] = tgt[k] - a[9 + k];
      len += a[12 + k] * a[12 + k];
    }
    len = std::sqrt(len);
    for (int k = 0; k < 3; k++) a[12 + k] /= len;
    if (mode == 4) {   // ray (almost) in the triangle's plane
      for (int k = 0; k < 3; k++) { a[9 + k] = a[k] + 3 * (a[3 + k] - a[k]); a[12 + k] = a[k] - a[9 + k] + 1e-7 * g.uni(); }
    }
    double t = 0, u = 0, v = 0;
    hit[i] = TriRayIntersect(Vector(a[0], a[1], a[2]), Vector(a[3], a[4], a[5]), Vector(a[6], a[7], a[8]),
        Vector(a[9], a[10], a[11]), Vector(a[12], a[13], a[14]), 0 /* DO_NOT_CULL_BACKFACES */, &t, &u, &v);
    if (hit[i]) { tuv[3 * i] = t; tuv[3 * i + 1] = u; tuv[3 * i + 2] = v; }
  }
  put_d("tri_in", {(uint32_t) N, 15}, in);
  put_i("tri_hit", {(uint32_t) N}, hit);
  put_d("tri_tuv", {(uint32_t) N, 3}, tuv);
}

static void gen_transform()
{
  Rng g(303);
  const int R = 3, N = 36 * R;
  std::vector<int32_t> orders(N * 2);
  std::vector<double> trs(N * 9), M(N * 16), Minv(N * 16);
  int i = 0;
  for (int to = 0; to < 6; to++)
    for (int ro = 6; ro < 12; ro++)
      for (int r = 0; r < R; r++, i++) {
        orders[2 * i] = to; orders[2 * i + 1] = ro;
        double *p = &trs[9 * i];
        for (int k = 0; k < 3; k++) p[k] = g.range(-10, 10);
        for (int k = 0; k < 3; k++) p[3 + k] = g.range(-360, 360);
        for (int k = 0; k < 3; k++) p[6 + k] = g.range(.2, 3) * ((g.next() & 7) ? 1 : -1);
        if (r == 0) { p[3] = 0; p[4] = -35; p[5] = 0; p[6] = p[7] = p[8] = .5; }
        Transform x;
        XfmSetTransform(&x, to, ro, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]);
        memcpy(&M[16 * i], x.matrix.e, sizeof(double) * 16);
        memcpy(&Minv[16 * i], x.inverse.e, sizeof(double) * 16);
      }
  put_i("xfm_orders", {(uint32_t) N, 2}, orders);
  put_d("xfm_trs", {(uint32_t) N, 9}, trs);
  put_d("xfm_M", {(uint32_t) N, 16}, M);
  put_d("xfm_Minv", {(uint32_t) N, 16}, Minv);
}

static void gen_camera()
{
  Rng g(404);
  // translate, rotate, fov, xres, yres of the C1/C2/C3 cameras + one odd one
  const double cams[4][9] = {
    {0, 1, 7, -5.710593137499643, 0, 0, 30, 256, 256},
    {5, 4, 5, -19.471220634490692, 45, 0, 30, 1280, 720},
    {0, 1.5, 7, -5.710593137499643, 0, 0, 30, 1920, 1080},
    {-3, 2.5, -4, 12, 200, 33, 55, 640, 480},
  };
  const int NC = 4, N = 500;
  std::vector<double> cam(NC * 9), uvt(NC * N * 3), rays(NC * N * 8);
  for (int c = 0; c < NC; c++) {
    memcpy(&cam[9 * c], cams[c], sizeof(cams[c]));
    Camera C;
    C.SetTranslate(cams[c][0], cams[c][1], cams[c][2], 0);
    C.SetRotate(cams[c][3], cams[c][4], cams[c][5], 0);
    C.SetFov(cams[c][6]);
    C.SetAspect(cams[c][7] / cams[c][8]);
    for (int i = 0; i < N; i++) {
      double *u = &uvt[(c * N + i) * 3];
      u[0] = g.uni(); u[1] = g.uni(); u[2] = g.uni();
      Ray r;
      C.GetRay(Vector2(u[0], u[1]), u[2], &r);
      double *o = &rays[(c * N + i) * 8];
      o[0] = r.orig.x; o[1] = r.orig.y; o[2] = r.orig.z; o[3] = r.dir.x; o[4] = r.dir.y; o[5] = r.dir.z; o[6] = r.tmin; o[7] = r.tmax;
    }
  }
  put_d("cam_params", {(uint32_t) NC, 9}, cam);
  put_d("cam_uvt", {(uint32_t) NC, (uint32_t) N, 3}, uvt);
  put_d("cam_rays", {(uint32_t) NC, (uint32_t) N, 8}, rays);
}

static void gen_sampler()
{
  // xres yres ratex ratey fw fh jitter xmin ymin xmax ymax t0 t1
  const double cfg[7][13] = {
    {256, 256, 1, 1, 2, 2, 1,    0, 0, 32, 32,   0, 1},
    {1280, 720, 4, 4, 2, 2, 1,   1248, 704, 1280, 720,  0, 1},
    {1920, 1080, 8, 8, 2, 2, 1,  960, 1056, 968, 1080,  0, 1},
    {640, 480, 3, 3, 2, 2, 1,    32, 64, 64, 96,   0, 1},
    {640, 480, 3, 2, 3, 2.5, .5, 608, 448, 640, 480,  .25, .75},
    {320, 240, 2, 2, 2, 2, 0,    0, 0, 32, 24,   0, 1},
    {1920, 1080, 16, 16, 2, 2, 1, 0, 0, 4, 4,  0, 1},
  };
  for (int c = 0; c < 7; c++) {
    FixedGridSampler s;
    s.SetResolution(Int2((int) cfg[c][0], (int) cfg[c][1]));
    s.SetPixelSamples(Int2((int) cfg[c][2], (int) cfg[c][3]));
    s.SetFilterWidth(Vector2(cfg[c][4], cfg[c][5]));
    s.SetJitter(cfg[c][6]);
    s.SetSampleTimeRange(cfg[c][11], cfg[c][12]);
    Rectangle r;
    r.min = Int2((int) cfg[c][7], (int) cfg[c][8]);
    r.max = Int2((int) cfg[c][9], (int) cfg[c][10]);
    s.GenerateSamples(r);
    std::vector<double> out;
    Sample *smp;
    while ((smp = s.GetNextSample()) != NULL) { out.push_back(smp->uv[0]); out.push_back(smp->uv[1]); out.push_back(smp->time); }
    char name[64];
    snprintf(name, sizeof(name), "sampler%d_cfg", c);
    put_d(name, {13}, std::vector<double>(cfg[c], cfg[c] + 13));
    snprintf(name, sizeof(name), "sampler%d_uvt", c);
    put_d(name, {(uint32_t) (out.size() / 3), 3}, out);
  }
}

static void gen_filter()
{
  Rng g(505);
  const int N = 4000;
  Filter f;
  f.SetFilterType(FLT_GAUSSIAN, 2, 2);
  Filter f2;
  f2.SetFilterType(FLT_GAUSSIAN, 3, 2.5);
  std::vector<double> xy(N * 2), w(N), w2(N);
  for (int i = 0; i < N; i++) {
    xy[2 * i] = g.range(-2, 2); xy[2 * i + 1] = g.range(-2, 2);
    w[i] = f.Evaluate(xy[2 * i], xy[2 * i + 1]);
    w2[i] = f2.Evaluate(xy[2 * i], xy[2 * i + 1]);
  }
  put_d("gauss_xy", {(uint32_t) N, 2}, xy);
  put_d("gauss_w_2_2", {(uint32_t) N}, w);
  put_d("gauss_w_3_2p5", {(uint32_t) N}, w2);
}

static void gen_tiler()
{
  // xres yres tw th  region xmin ymin xmax ymax
  const int cfg[5][8] = {
    {256, 256, 32, 32, 0, 0, 256, 256},
    {1280, 720, 32, 32, 0, 0, 1280, 720},
    {1920, 1080, 32, 32, 0, 0, 1920, 1080},
    {640, 480, 64, 48, 100, 50, 500, 333},
    {100, 70, 32, 32, 0, 0, 100, 70},
  };
  for (int c = 0; c < 5; c++) {
    Tiler t;
    t.Divide(cfg[c][0], cfg[c][1], cfg[c][2], cfg[c][3]);
    Rectangle r;
    r.min = Int2(cfg[c][4], cfg[c][5]);
    r.max = Int2(cfg[c][6], cfg[c][7]);
    t.GenerateTiles(r);
    std::vector<int32_t> out;
    for (int i = 0; i < t.GetTileCount(); i++) {
      const Tile *tl = t.GetTile(i);
      out.push_back(tl->id); out.push_back(tl->xmin); out.push_back(tl->ymin); out.push_back(tl->xmax); out.push_back(tl->ymax);
    }
    char name[64];
    snprintf(name, sizeof(name), "tiler%d_cfg", c);
    put_i(name, {8}, std::vector<int32_t>(cfg[c], cfg[c] + 8));
    snprintf(name, sizeof(name), "tiler%d_tiles", c);
    put_i(name, {(uint32_t) (out.size() / 5), 5}, out);
  }
}

static int gen_mesh_trace(const char *mesh_path, const char *rays_path)
{
  FILE *fm = fopen(mesh_path, "rb"), *fr = fopen(rays_path, "rb");
  if (!fm || !fr) { fprintf(stderr, "ref_vectors: cannot open mesh / rays file\n"); return -1; }
  int32_t np = 0, nf = 0, nr = 0;
  if (fread(&np, 4, 1, fm) != 1 || fread(&nf, 4, 1, fm) != 1) return -1;
  std::vector<double> P((size_t) np * 3);
  std::vector<int32_t> idx((size_t) nf * 3);
  if (fread(P.data(), 8, P.size(), fm) != P.size() || fread(idx.data(), 4, idx.size(), fm) != idx.size()) return -1;
  if (fread(&nr, 4, 1, fr) != 1) return -1;
  std::vector<double> rays((size_t) nr * 8);
  if (fread(rays.data(), 8, rays.size(), fr) != rays.size()) return -1;
  fclose(fm); fclose(fr);

  Mesh mesh;
  mesh.SetPointCount(np);
  mesh.AddPointPosition();
  for (int i = 0; i < np; i++) mesh.SetPointPosition(i, Vector(P[3 * i], P[3 * i + 1], P[3 * i + 2]));
  mesh.SetFaceCount(nf);
  mesh.AddFaceIndices();
  for (int i = 0; i < nf; i++) mesh.SetFaceIndices(i, Index3(idx[3 * i], idx[3 * i + 1], idx[3 * i + 2]));
  mesh.ComputeNormals();
  mesh.ComputeBounds();
  GridAccelerator acc;
  acc.SetPrimitiveSet(&mesh);
  acc.ComputeBounds();
  acc.Build();

  std::vector<double> t(nr), attr((size_t) nr * 6, 0.);
  std::vector<int32_t> prim(nr);
  for (int i = 0; i < nr; i++) {
    const double *r = &rays[(size_t) i * 8];
    Ray ray;
    ray.orig = Vector(r[0], r[1], r[2]);
    ray.dir = Vector(r[3], r[4], r[5]);
    ray.tmin = r[6]; ray.tmax = r[7];
    Intersection is;
    const bool hit = acc.Intersect(ray, 0, &is);
    t[i] = hit ? is.t_hit : 1.7976931348623157e308;
    prim[i] = hit ? is.prim_id : -1;
    if (hit) { double *a = &attr[(size_t) i * 6]; a[0] = is.N.x; a[1] = is.N.y; a[2] = is.N.z; a[3] = is.P.x; a[4] = is.P.y; a[5] = is.P.z; }
  }
  std::vector<double> nrm((size_t) np * 3);
  for (int i = 0; i < np; i++) { const Vector n = mesh.GetPointNormal(i); nrm[3 * i] = n.x; nrm[3 * i + 1] = n.y; nrm[3 * i + 2] = n.z; }
  const Box &b = mesh.GetBounds();
  put_d("mesh_normals", {(uint32_t) np, 3}, nrm);
  put_d("mesh_bounds", {6}, {b.min.x, b.min.y, b.min.z, b.max.x, b.max.y, b.max.z});
  put_d("grid_t", {(uint32_t) nr}, t);
  put_i("grid_prim", {(uint32_t) nr}, prim);
  put_d("grid_attr", {(uint32_t) nr, 6}, attr);
  return 0;
}

int main(int argc, const char **argv)
{
  if (argc != 2 && argc != 4) { fprintf(stderr, "usage: ref_vectors out.bin [mesh.bin rays.bin]\n"); return 2; }
  g_out = fopen(argv[1], "wb");
  if (!g_out) return 2;
  fwrite("FJGV", 1, 4, g_out);
  gen_xorshift();
  gen_box();
  gen_tri();
  gen_transform();
  gen_camera();
  gen_sampler();
  gen_filter();
  gen_tiler();
  int rc = 0;
  if (argc == 4) rc = gen_mesh_trace(argv[2], argv[3]);
  fclose(g_out);
  return rc ? 1 : 0;
}
