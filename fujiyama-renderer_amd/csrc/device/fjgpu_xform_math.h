// fjgpu_xform_math.h -- TransformSampleList evaluation, shared by the host build
// (fjgpu_xform.cc: static transforms, instance bounds) and the device kernels
// (time-sampled camera / instance transforms are evaluated per ray, like the reference
// does in Camera::GetRay and ObjectInstance::RayIntersect).  One source, so the operation
// order is the same on both sides:
//   make_transform_matrix   reference src/fj_transform.cc:335-391
//   MatRotateX/Y/Z, MatMultiply, MatInverse (Cramer)   src/fj_matrix.cc:66-206
//   PropLerpSamples         src/fj_property.cc:317-345
// On the host sin / cos are glibc's (what the reference calls); on the device they are
// the ROCm device library's (<= 2 ulp): time-sampled transforms therefore agree with the
// reference to rounding, static ones (host-built) bit for bit.
#ifndef FJGPU_XFORM_MATH_H
#define FJGPU_XFORM_MATH_H

#include "fj_scene_desc.h"
#include <math.h>

#if defined(__HIPCC__)
#define FJ_HD __host__ __device__
#else
#define FJ_HD
#endif

namespace fjx {

#define FJX_PI 3.14159265358979323846

struct M44 { double e[16]; };

FJ_HD inline M44 identity()
{
  M44 m;
  for (int i = 0; i < 16; i++) m.e[i] = (i % 5 == 0) ? 1. : 0.;
  return m;
}

// c[j][i] = ((((0 + a[j][0] b[0][i]) + a[j][1] b[1][i]) + a[j][2] b[2][i]) + a[j][3] b[3][i])
FJ_HD inline M44 mul(const M44 &a, const M44 &b)
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

FJ_HD inline M44 rows(double a, double b, double c, double d, double e, double f, double g, double h, double i, double j, double k, double l)
{
  M44 m;
  const double v[16] = {a, b, c, d, e, f, g, h, i, j, k, l, 0., 0., 0., 1.};
  for (int q = 0; q < 16; q++) m.e[q] = v[q];
  return m;
}

FJ_HD inline M44 rot(int axis, double deg)
{
  const double rad = deg * FJX_PI / 180.;
  const double s = ::sin(rad), c = ::cos(rad);
  if (axis == 0) return rows(1., 0., 0., 0.,  0., c, -s, 0.,  0., s, c, 0.);
  if (axis == 1) return rows(c, 0., s, 0.,  0., 1., 0., 0.,  -s, 0., c, 0.);
  return rows(c, -s, 0., 0.,  s, c, 0., 0.,  0., 0., 1., 0.);
}

// three-term cofactor row: (p0 + p1 + p2) - (m0 + m1 + m2), left to right
FJ_HD inline double cof(double p0, double p1, double p2, double m0, double m1, double m2)
{
  double r = p0 + p1 + p2;
  r -= m0 + m1 + m2;
  return r;
}

// Cramer's rule on the transposed matrix, term order of src/fj_matrix.cc:117-206
FJ_HD inline M44 inverse(const M44 &a)
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

FJ_HD inline double fit01(double x, double s0, double s1)
{
  if (x <= s0) return 0;
  if (x >= s1) return 1;
  return 0 + (1 - 0) * ((x - s0) / (s1 - s0));
}

FJ_HD inline void lerp_samples(const fj_xform_sample *s, int n, double time, double out[3])
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


FJ_HD inline void make_transform(const fj_xform_desc &x, double time, double M[16], double Minv[16])
{
  double T[3] = {0, 0, 0}, R[3] = {0, 0, 0}, S[3] = {1, 1, 1};
  lerp_samples(x.translate, x.n_translate, time, T);
  lerp_samples(x.rotate, x.n_rotate, time, R);
  lerp_samples(x.scale, x.n_scale, time, S);

  const M44 mt = rows(1., 0., 0., T[0],  0., 1., 0., T[1],  0., 0., 1., T[2]);
  const M44 ms = rows(S[0], 0., 0., 0.,  0., S[1], 0., 0.,  0., 0., S[2], 0.);
  const M44 r[3] = {rot(0, R[0]), rot(1, R[1]), rot(2, R[2])};

  // rotate_order 6..11 = XYZ XZY YXZ YZX ZXY ZYX ; applied as R = q2 * q1 * q0
  const int rorder[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  const int *ro = rorder[(x.rotate_order >= 6 && x.rotate_order < 12) ? x.rotate_order - 6 : 0];
  M44 mr = identity();
  for (int i = 0; i < 3; i++) mr = mul(r[ro[i]], mr);

  // transform_order 0..5 = SRT STR RST RTS TRS TSR ; M = q2 * q1 * q0
  const M44 *by_letter[3] = {&ms, &mr, &mt};   // S R T
  const int torder[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 1, 0}, {2, 0, 1}};
  const int *to = torder[(x.transform_order >= 0 && x.transform_order < 6) ? x.transform_order : 0];
  M44 m = identity();
  for (int i = 0; i < 3; i++) m = mul(*by_letter[to[i]], m);

  const M44 inv = inverse(m);
  for (int q = 0; q < 16; q++) { M[q] = m.e[q]; Minv[q] = inv.e[q]; }
}


}  // namespace fjx
#endif
