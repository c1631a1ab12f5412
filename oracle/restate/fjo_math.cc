// oracle/restate/fjo_math.cc -- TEST INFRASTRUCTURE ONLY (CPU oracle).
// Restatement of src/fj_box.cc, src/fj_matrix.cc, src/fj_transform.cc math in
// the reference's operation order (see fjo_math.h).
#include "fjo_math.h"

namespace fjo {

// src/fj_box.cc:73-138.  Slab test; per-axis branch on dir >= 0; six real
// divisions; early outs after y and z; accepted iff tmin < ray_tmax and
// tmax > ray_tmin.  NaN (0/0) comparisons are false exactly as in the
// reference because the expression structure is the same.
bool BoxRayIntersect(const Box &box, const V3 &o, const V3 &d,
    double ray_tmin, double ray_tmax, double *hit_tmin, double *hit_tmax)
{
  double tmin, tmax, tymin, tymax, tzmin, tzmax;

  if (d.x >= 0) { tmin = (box.min.x - o.x) / d.x; tmax = (box.max.x - o.x) / d.x; }
  else          { tmin = (box.max.x - o.x) / d.x; tmax = (box.min.x - o.x) / d.x; }

  if (d.y >= 0) { tymin = (box.min.y - o.y) / d.y; tymax = (box.max.y - o.y) / d.y; }
  else          { tymin = (box.max.y - o.y) / d.y; tymax = (box.min.y - o.y) / d.y; }

  if ((tmin > tymax) || (tymin > tmax)) return false;
  if (tymin > tmin) tmin = tymin;
  if (tymax < tmax) tmax = tymax;

  if (d.z >= 0) { tzmin = (box.min.z - o.z) / d.z; tzmax = (box.max.z - o.z) / d.z; }
  else          { tzmin = (box.max.z - o.z) / d.z; tzmax = (box.min.z - o.z) / d.z; }

  if ((tmin > tzmax) || (tzmin > tmax)) return false;
  if (tzmin > tmin) tmin = tzmin;
  if (tzmax < tmax) tmax = tzmax;

  const bool hit = (tmin < ray_tmax) && (tmax > ray_tmin);
  if (hit) { *hit_tmin = tmin; *hit_tmax = tmax; }
  return hit;
}

void MatIdentity(Mat *m)
{
  for (int i = 0; i < 16; i++) m->e[i] = (i % 5 == 0) ? 1. : 0.;
}

// src/fj_matrix.cc:102-115: c = 0; c += a[j][k]*b[k][i], k ascending
void MatMultiply(Mat *dst, const Mat &a, const Mat &b)
{
  Mat c;
  for (int j = 0; j < 4; j++)
    for (int i = 0; i < 4; i++) {
      double acc = 0.;
      for (int k = 0; k < 4; k++) acc += a.e[4 * j + k] * b.e[4 * k + i];
      c.e[4 * j + i] = acc;
    }
  *dst = c;
}

// src/fj_matrix.cc:117-206: Cramer's rule on the transposed source.  The
// reference spells out 2 x 12 pair products and 16 cofactor rows of the form
// (p0 + p1 + p2) - (m0 + m1 + m2); the tables below hold the same index
// pattern, evaluated in the same left-to-right order.
void MatInverse(Mat *dst, const Mat &a)
{
  double src[16], tmp[12];
  for (int i = 0; i < 4; i++) {
    src[i] = a.e[i * 4];
    src[i + 4] = a.e[i * 4 + 1];
    src[i + 8] = a.e[i * 4 + 2];
    src[i + 12] = a.e[i * 4 + 3];
  }
  static const int pair_a[2][12][2] = {
    {{10,15},{11,14},{9,15},{11,13},{9,14},{10,13},{8,15},{11,12},{8,14},{10,12},{8,13},{9,12}},
    {{2,7},{3,6},{1,7},{3,5},{1,6},{2,5},{0,7},{3,4},{0,6},{2,4},{0,5},{1,4}}};
  // rows: {tmp,src} x3 added, {tmp,src} x3 subtracted
  static const int cof[16][12] = {
    {0,5, 3,6, 4,7,   1,5, 2,6, 5,7},
    {1,4, 6,6, 9,7,   0,4, 7,6, 8,7},
    {2,4, 7,5, 10,7,  3,4, 6,5, 11,7},
    {5,4, 8,5, 11,6,  4,4, 9,5, 10,6},
    {1,1, 2,2, 5,3,   0,1, 3,2, 4,3},
    {0,0, 7,2, 8,3,   1,0, 6,2, 9,3},
    {3,0, 6,1, 11,3,  2,0, 7,1, 10,3},
    {4,0, 9,1, 10,2,  5,0, 8,1, 11,2},
    {0,13, 3,14, 4,15,   1,13, 2,14, 5,15},
    {1,12, 6,14, 9,15,   0,12, 7,14, 8,15},
    {2,12, 7,13, 10,15,  3,12, 6,13, 11,15},
    {5,12, 8,13, 11,14,  4,12, 9,13, 10,14},
    {2,10, 5,11, 1,9,    4,11, 0,9, 3,10},
    {8,11, 0,8, 7,10,    6,10, 9,11, 1,8},
    {6,9, 11,11, 3,8,    10,11, 2,8, 7,9},
    {10,10, 4,8, 9,9,    8,9, 11,10, 5,8}};

  for (int half = 0; half < 2; half++) {
    for (int k = 0; k < 12; k++)
      tmp[k] = src[pair_a[half][k][0]] * src[pair_a[half][k][1]];
    for (int r = 8 * half; r < 8 * half + 8; r++) {
      const int *c = cof[r];
      double plus = tmp[c[0]] * src[c[1]] + tmp[c[2]] * src[c[3]] + tmp[c[4]] * src[c[5]];
      const double minus = tmp[c[6]] * src[c[7]] + tmp[c[8]] * src[c[9]] + tmp[c[10]] * src[c[11]];
      plus -= minus;
      dst->e[r] = plus;
    }
  }
  double det = src[0] * dst->e[0] + src[1] * dst->e[1] + src[2] * dst->e[2] + src[3] * dst->e[3];
  det = 1. / det;
  for (int j = 0; j < 16; j++) dst->e[j] *= det;
}

// src/fj_matrix.cc:224-249 (corner order: mmm Mmm mMm mmM mMM MmM MMm MMM)
void MatTransformBounds(const Mat &m, Box *bounds)
{
  static const int pick[8][3] = {{0,0,0},{1,0,0},{0,1,0},{0,0,1},{0,1,1},{1,0,1},{1,1,0},{1,1,1}};
  Box box;
  box.ReverseInfinite();
  for (int c = 0; c < 8; c++) {
    V3 pt(pick[c][0] ? bounds->max.x : bounds->min.x,
          pick[c][1] ? bounds->max.y : bounds->min.y,
          pick[c][2] ? bounds->max.z : bounds->min.z);
    box.AddPoint(MatTransformPoint(m, pt));
  }
  *bounds = box;
}

static void set16(Mat *m, double a0, double a1, double a2, double a3, double a4, double a5,
    double a6, double a7, double a8, double a9, double a10, double a11)
{
  const double v[16] = {a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, 0., 0., 0., 1.};
  for (int i = 0; i < 16; i++) m->e[i] = v[i];
}

// src/fj_transform.cc:335-391 + src/fj_matrix.cc:48-100
static void make_transform_matrix(int transform_order, int rotate_order,
    double tx, double ty, double tz, double rx, double ry, double rz,
    double sx, double sy, double sz, Mat *out)
{
  Mat T, R, S, RX, RY, RZ;
  set16(&T, 1., 0., 0., tx, 0., 1., 0., ty, 0., 0., 1., tz);
  {
    const double s = std::sin(Radian(rx)), c = std::cos(Radian(rx));
    set16(&RX, 1., 0., 0., 0., 0., c, -s, 0., 0., s, c, 0.);
  }
  {
    const double s = std::sin(Radian(ry)), c = std::cos(Radian(ry));
    set16(&RY, c, 0., s, 0., 0., 1., 0., 0., -s, 0., c, 0.);
  }
  {
    const double s = std::sin(Radian(rz)), c = std::cos(Radian(rz));
    set16(&RZ, c, -s, 0., 0., s, c, 0., 0., 0., 0., 1., 0.);
  }
  set16(&S, sx, 0., 0., 0., 0., sy, 0., 0., 0., 0., sz, 0.);

  const Mat *q[3] = {&RX, &RY, &RZ};
  switch (rotate_order) {
  case 6:  q[0] = &RX; q[1] = &RY; q[2] = &RZ; break;  // XYZ
  case 7:  q[0] = &RX; q[1] = &RZ; q[2] = &RY; break;  // XZY
  case 8:  q[0] = &RY; q[1] = &RX; q[2] = &RZ; break;  // YXZ
  case 9:  q[0] = &RY; q[1] = &RZ; q[2] = &RX; break;  // YZX
  case 10: q[0] = &RZ; q[1] = &RX; q[2] = &RY; break;  // ZXY
  case 11: q[0] = &RZ; q[1] = &RY; q[2] = &RX; break;  // ZYX
  default: break;
  }
  MatIdentity(&R);
  for (int i = 0; i < 3; i++) MatMultiply(&R, *q[i], R);

  switch (transform_order) {
  case 0: q[0] = &S; q[1] = &R; q[2] = &T; break;  // SRT
  case 1: q[0] = &S; q[1] = &T; q[2] = &R; break;  // STR
  case 2: q[0] = &R; q[1] = &S; q[2] = &T; break;  // RST
  case 3: q[0] = &R; q[1] = &T; q[2] = &S; break;  // RTS
  case 4: q[0] = &T; q[1] = &R; q[2] = &S; break;  // TRS
  case 5: q[0] = &T; q[1] = &S; q[2] = &R; break;  // TSR
  default: break;
  }
  MatIdentity(out);
  for (int i = 0; i < 3; i++) MatMultiply(out, *q[i], *out);
}

void XfmSetTransform(Xfm *x, int transform_order, int rotate_order,
    double tx, double ty, double tz, double rx, double ry, double rz,
    double sx, double sy, double sz)
{
  x->translate = V3(tx, ty, tz);
  x->rotate = V3(rx, ry, rz);
  x->scale = V3(sx, sy, sz);
  make_transform_matrix(transform_order, rotate_order, tx, ty, tz, rx, ry, rz, sx, sy, sz, &x->matrix);
  MatInverse(&x->inverse, x->matrix);
}

}  // namespace fjo
