// fjgpu_xform.cc -- host math that feeds geometry to the device.
//
// Every transcendental that influences a hit/miss decision is evaluated HERE,
// on the host, with the reference's own operation order (libm sin/cos/tan,
// no FP contraction), so the matrices and tables the kernels consume are
// bit-identical to what the reference computes per ray:
//   make_transform_matrix   reference src/fj_transform.cc:335-391
//   MatRotateX/Y/Z, MatMultiply, MatInverse (Cramer)   src/fj_matrix.cc:66-206
//   PropLerpSamples         src/fj_property.cc:317-345
//   Tiler::GenerateTiles    src/fj_tiler.cc:56-113
//   XorShift                src/fj_random.cc:10-43
// Compiled with -ffp-contract=off.
#include "fjgpu_build.h"

#include <cmath>
#include <cstring>

namespace fjgpu {

namespace {

const double kPi = 3.14159265358979323846;

struct M44 { double e[16]; };

M44 identity()
{
  M44 m;
  for (int i = 0; i < 16; i++) m.e[i] = (i % 5 == 0) ? 1. : 0.;
  return m;
}

// c[j][i] = ((((0 + a[j][0] b[0][i]) + a[j][1] b[1][i]) + a[j][2] b[2][i]) + a[j][3] b[3][i])
M44 mul(const M44 &a, const M44 &b)
{
  M44 c;
  for (int j = 0; j < 4; j++)
    for (int i = 0; i < 4; i++) {
      double s = 0.;
      s += a.e[4 * j + 0] * b.e[0 + i];
      s += a.e[4 * j + 1] * b.e[4 + i];
      s += a.e[4 * j + 2] * b.e[8 + i];
      s += a.e[4 * j + 3] * b.e[12 + i];
      c.e[4 * j + i] = s;
    }
  return c;
}

M44 rows(double a, double b, double c, double d, double e, double f, double g, double h, double i, double j, double k, double l)
{
  M44 m;
  const double v[16] = {a, b, c, d, e, f, g, h, i, j, k, l, 0., 0., 0., 1.};
  std::memcpy(m.e, v, sizeof(v));
  return m;
}

M44 rot(int axis, double deg)
{
  const double rad = deg * kPi / 180.;
  const double s = std::sin(rad), c = std::cos(rad);
  if (axis == 0) return rows(1., 0., 0., 0.,  0., c, -s, 0.,  0., s, c, 0.);
  if (axis == 1) return rows(c, 0., s, 0.,  0., 1., 0., 0.,  -s, 0., c, 0.);
  return rows(c, -s, 0., 0.,  s, c, 0., 0.,  0., 0., 1., 0.);
}

// three-term cofactor row: (p0 + p1 + p2) - (m0 + m1 + m2), left to right
inline double cof(double p0, double p1, double p2, double m0, double m1, double m2)
{
  double r = p0 + p1 + p2;
  r -= m0 + m1 + m2;
  return r;
}

// Cramer's rule on the transposed matrix, term order of src/fj_matrix.cc:117-206
M44 inverse(const M44 &a)
{
  double s[16];
  for (int i = 0; i < 4; i++) { s[i] = a.e[4 * i]; s[i + 4] = a.e[4 * i + 1]; s[i + 8] = a.e[4 * i + 2]; s[i + 12] = a.e[4 * i + 3]; }
  M44 d;
  double t[12];
  t[0] = s[10] * s[15]; t[1] = s[11] * s[14]; t[2] = s[9] * s[15]; t[3] = s[11] * s[13];
  t[4] = s[9] * s[14];  t[5] = s[10] * s[13]; t[6] = s[8] * s[15]; t[7] = s[11] * s[12];
  t[8] = s[8] * s[14];  t[9] = s[10] * s[12]; t[10] = s[8] * s[13]; t[11] = s[9] * s[12];
  d.e[0] = cof(t[0] * s[5], t[3] * s[6], t[4] * s[7],   t[1] * s[5], t[2] * s[6], t[5] * s[7]);
  d.e[1] = cof(t[1] * s[4], t[6] * s[6], t[9] * s[7],   t[0] * s[4], t[7] * s[6], t[8] * s[7]);
  d.e[2] = cof(t[2] * s[4], t[7] * s[5], t[10] * s[7],  t[3] * s[4], t[6] * s[5], t[11] * s[7]);
  d.e[3] = cof(t[5] * s[4], t[8] * s[5], t[11] * s[6],  t[4] * s[4], t[9] * s[5], t[10] * s[6]);
  d.e[4] = cof(t[1] * s[1], t[2] * s[2], t[5] * s[3],   t[0] * s[1], t[3] * s[2], t[4] * s[3]);
  d.e[5] = cof(t[0] * s[0], t[7] * s[2], t[8] * s[3],   t[1] * s[0], t[6] * s[2], t[9] * s[3]);
  d.e[6] = cof(t[3] * s[0], t[6] * s[1], t[11] * s[3],  t[2] * s[0], t[7] * s[1], t[10] * s[3]);
  d.e[7] = cof(t[4] * s[0], t[9] * s[1], t[10] * s[2],  t[5] * s[0], t[8] * s[1], t[11] * s[2]);
  t[0] = s[2] * s[7]; t[1] = s[3] * s[6]; t[2] = s[1] * s[7]; t[3] = s[3] * s[5];
  t[4] = s[1] * s[6]; t[5] = s[2] * s[5]; t[6] = s[0] * s[7]; t[7] = s[3] * s[4];
  t[8] = s[0] * s[6]; t[9] = s[2] * s[4]; t[10] = s[0] * s[5]; t[11] = s[1] * s[4];
  d.e[8]  = cof(t[0] * s[13], t[3] * s[14], t[4] * s[15],   t[1] * s[13], t[2] * s[14], t[5] * s[15]);
  d.e[9]  = cof(t[1] * s[12], t[6] * s[14], t[9] * s[15],   t[0] * s[12], t[7] * s[14], t[8] * s[15]);
  d.e[10] = cof(t[2] * s[12], t[7] * s[13], t[10] * s[15],  t[3] * s[12], t[6] * s[13], t[11] * s[15]);
  d.e[11] = cof(t[5] * s[12], t[8] * s[13], t[11] * s[14],  t[4] * s[12], t[9] * s[13], t[10] * s[14]);
  d.e[12] = cof(t[2] * s[10], t[5] * s[11], t[1] * s[9],    t[4] * s[11], t[0] * s[9], t[3] * s[10]);
  d.e[13] = cof(t[8] * s[11], t[0] * s[8], t[7] * s[10],    t[6] * s[10], t[9] * s[11], t[1] * s[8]);
  d.e[14] = cof(t[6] * s[9], t[11] * s[11], t[3] * s[8],    t[10] * s[11], t[2] * s[8], t[7] * s[9]);
  d.e[15] = cof(t[10] * s[10], t[4] * s[8], t[9] * s[9],    t[8] * s[9], t[11] * s[10], t[5] * s[8]);
  double det = s[0] * d.e[0] + s[1] * d.e[1] + s[2] * d.e[2] + s[3] * d.e[3];
  det = 1. / det;
  for (int j = 0; j < 16; j++) d.e[j] *= det;
  return d;
}

double fit01(double x, double s0, double s1)
{
  if (x <= s0) return 0;
  if (x >= s1) return 1;
  return 0 + (1 - 0) * ((x - s0) / (s1 - s0));
}

void lerp_samples(const fj_xform_sample *s, int n, double time, double out[3])
{
  const fj_xform_sample *pick = nullptr;
  if (s[0].time >= time || n == 1) pick = &s[0];
  else if (s[n - 1].time <= time) pick = &s[n - 1];
  if (pick) { for (int i = 0; i < 3; i++) out[i] = pick->v[i]; return; }
  for (int k = 0; k < n; k++) {
    if (s[k].time == time) { for (int i = 0; i < 3; i++) out[i] = s[k].v[i]; return; }
    if (s[k].time > time) {
      const double t = fit01(time, s[k - 1].time, s[k].time);
      for (int i = 0; i < 3; i++) out[i] = (1 - t) * s[k - 1].v[i] + t * s[k].v[i];
      return;
    }
  }
}

}  // namespace

void MakeTransform(const fj_xform_desc &x, double time, double M[16], double Minv[16])
{
  double T[3], R[3], S[3];
  lerp_samples(x.translate, x.n_translate, time, T);
  lerp_samples(x.rotate, x.n_rotate, time, R);
  lerp_samples(x.scale, x.n_scale, time, S);

  const M44 mt = rows(1., 0., 0., T[0],  0., 1., 0., T[1],  0., 0., 1., T[2]);
  const M44 ms = rows(S[0], 0., 0., 0.,  0., S[1], 0., 0.,  0., 0., S[2], 0.);
  const M44 r[3] = {rot(0, R[0]), rot(1, R[1]), rot(2, R[2])};

  // rotate_order 6..11 = XYZ XZY YXZ YZX ZXY ZYX ; applied as R = q2 * q1 * q0
  static const int rorder[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  const int *ro = rorder[(x.rotate_order >= 6 && x.rotate_order < 12) ? x.rotate_order - 6 : 0];
  M44 mr = identity();
  for (int i = 0; i < 3; i++) mr = mul(r[ro[i]], mr);

  // transform_order 0..5 = SRT STR RST RTS TRS TSR ; M = q2 * q1 * q0
  const M44 *by_letter[3] = {&ms, &mr, &mt};   // S R T
  static const int torder[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 1, 0}, {2, 0, 1}};
  const int *to = torder[(x.transform_order >= 0 && x.transform_order < 6) ? x.transform_order : 0];
  M44 m = identity();
  for (int i = 0; i < 3; i++) m = mul(*by_letter[to[i]], m);

  const M44 inv = inverse(m);
  std::memcpy(M, m.e, sizeof(m.e));
  std::memcpy(Minv, inv.e, sizeof(inv.e));
}

// MatTransformBounds, src/fj_matrix.cc:224-249
void TransformBounds(const double M[16], const double in[6], double out[6])
{
  const double big = 1.7976931348623157e308;
  double mn[3] = {big, big, big}, mx[3] = {-big, -big, -big};
  for (int c = 0; c < 8; c++) {
    const double p[3] = {(c & 1) ? in[3] : in[0], (c & 2) ? in[4] : in[1], (c & 4) ? in[5] : in[2]};
    for (int k = 0; k < 3; k++) {
      const double q = M[4 * k] * p[0] + M[4 * k + 1] * p[1] + M[4 * k + 2] * p[2] + M[4 * k + 3];
      if (q < mn[k]) mn[k] = q;
      if (q > mx[k]) mx[k] = q;
    }
  }
  for (int k = 0; k < 3; k++) { out[k] = mn[k]; out[3 + k] = mx[k]; }
}

void GenerateTiles(const fj_render_desc &r, std::vector<TileRect> *tiles)
{
  auto imax = [](int a, int b) { return a > b ? a : b; };
  auto imin = [](int a, int b) { return a < b ? a : b; };
  const int xmin = r.region[0], ymin = r.region[1], xmax = r.region[2], ymax = r.region[3];
  const int X0 = (int) std::floor(imax(0, xmin) / (double) r.tile_w);
  const int Y0 = (int) std::floor(imax(0, ymin) / (double) r.tile_h);
  const int X1 = (int) std::ceil(imin(r.xres, xmax) / (double) r.tile_w);
  const int Y1 = (int) std::ceil(imin(r.yres, ymax) / (double) r.tile_h);
  tiles->clear();
  for (int y = Y0; y < Y1; y++)
    for (int x = X0; x < X1; x++) {
      TileRect t;
      t.id = (int) tiles->size();
      t.xmin = imax(x * r.tile_w, xmin);
      t.ymin = imax(y * r.tile_h, ymin);
      t.xmax = imin((x + 1) * r.tile_w, xmax);
      t.ymax = imin((y + 1) * r.tile_h, ymax);
      tiles->push_back(t);
    }
}

// count_samples_in_margin, src/fj_fixed_grid_sampler.cc:131-136
void SamplerMargin(const fj_render_desc &r, int margin[2])
{
  margin[0] = (int) std::ceil((((double) r.filter_w - 1) * r.rate_x) * .5);
  margin[1] = (int) std::ceil((((double) r.filter_h - 1) * r.rate_y) * .5);
}

void XorShiftTable(size_t n, std::vector<double> *out)
{
  uint32_t s0 = 123456789u, s1 = 362436069u, s2 = 521288629u, s3 = 88675123u;
  out->resize(n);
  for (size_t i = 0; i < n; i++) {
    const uint32_t t = s0 ^ (s0 << 11);
    s0 = s1; s1 = s2; s2 = s3;
    s3 = (s3 ^ (s3 >> 19)) ^ (t ^ (t >> 8));
    (*out)[i] = static_cast<double>(s3) / 4294967295u;   // / UINT32_MAX
  }
}

double CameraUvSizeY(double fov)   // src/fj_camera.cc:98-102
{
  return 2 * std::tan((fov / 2.) * kPi / 180.);
}

}  // namespace fjgpu

// ---- diagnostics (include/fjgpu.h): let the CPU test-suite pin the host math
// against the reference's golden vectors without a GPU
extern "C" {

void fjgpu_host_make_transform(int transform_order, int rotate_order, const double *trs, double *M, double *Minv)
{
  fj_xform_desc x;
  std::memset(&x, 0, sizeof(x));
  x.transform_order = transform_order;
  x.rotate_order = rotate_order;
  x.n_translate = x.n_rotate = x.n_scale = 1;
  for (int k = 0; k < 3; k++) { x.translate[0].v[k] = trs[k]; x.rotate[0].v[k] = trs[3 + k]; x.scale[0].v[k] = trs[6 + k]; }
  fjgpu::MakeTransform(x, 0, M, Minv);
}

void fjgpu_host_xorshift_f01(int n, double *out)
{
  std::vector<double> t;
  fjgpu::XorShiftTable((size_t) (n > 0 ? n : 0), &t);
  for (int i = 0; i < n; i++) out[i] = t[i];
}

void fjgpu_host_sampler_margin(const fj_render_desc *r, int32_t *margin)
{
  int m[2];
  fjgpu::SamplerMargin(*r, m);
  margin[0] = m[0]; margin[1] = m[1];
}

double fjgpu_host_camera_uv_size_y(double fov) { return fjgpu::CameraUvSizeY(fov); }

}  // extern "C"
