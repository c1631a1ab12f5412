// oracle/restate/fjo_math.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
//
// Restatement of the reference's level-0 math, in the reference's exact
// operation order (FP64 geometry, FP32 colour); compiled -ffp-contract=off.
// Each block cites the reference lines it follows.  Not part of the product:
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load liboracle.so.
#ifndef FJO_MATH_H
#define FJO_MATH_H

#include <cmath>
#include <cstdint>
#include <limits>

namespace fjo {

static const double PI = 3.14159265358979323846;            // src/fj_numeric.h:14
static const double REAL_MAX = std::numeric_limits<double>::max();  // :15

struct V3 {
  double x, y, z;
  V3() : x(0), y(0), z(0) {}
  V3(double a, double b, double c) : x(a), y(b), z(c) {}
  double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  double &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
};

// src/fj_vector.h:262-324 (component-wise; a / s == a * (1./s))
static inline V3 operator+(const V3 &a, const V3 &b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(const V3 &a, const V3 &b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator*(const V3 &a, const V3 &b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 operator/(const V3 &a, const V3 &b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline V3 operator*(const V3 &a, double s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline V3 operator*(double s, const V3 &a) { return a * s; }
static inline V3 operator/(const V3 &a, double s) { const double inv = 1. / s; return a * inv; }
static inline double Dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 Cross(const V3 &a, const V3 &b)       // src/fj_vector.h:326-332
{
  return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline double Length(const V3 &a) { return std::sqrt(Dot(a, a)); }  // :334-337
static inline V3 Normalize(const V3 &a)                                    // :339-345
{
  const double len = Length(a);
  if (len == 0) return a;
  return a / len;
}

// src/fj_numeric.h:42-59,96-105
static inline double Min(double x, double y) { return x < y ? x : y; }
static inline double Max(double x, double y) { return x > y ? x : y; }
static inline double Clamp(double x, double a, double b) { return x < a ? a : (x > b ? b : x); }
static inline double Radian(double deg) { return deg * PI / 180.; }
static inline double Fit(double x, double s0, double s1, double d0, double d1)
{
  if (x <= s0) return d0;
  if (x >= s1) return d1;
  return d0 + (d1 - d0) * ((x - s0) / (s1 - s0));
}

struct Col { float r, g, b; Col() : r(0), g(0), b(0) {} Col(float a, float c, float d) : r(a), g(c), b(d) {} };
struct Col4 { float r, g, b, a; Col4() : r(0), g(0), b(0), a(0) {} Col4(float x, float y, float z, float w) : r(x), g(y), b(z), a(w) {} };

// src/fj_box.{h,cc}
struct Box {
  V3 min, max;
  Box() {}
  Box(const V3 &a, const V3 &b) : min(a), max(b) {}
  void Expand(double d) { min = min - V3(d, d, d); max = max + V3(d, d, d); }   // :24-28
  void ReverseInfinite()                                                       // :30-34
  {
    min = V3(REAL_MAX, REAL_MAX, REAL_MAX);
    max = V3(-REAL_MAX, -REAL_MAX, -REAL_MAX);
  }
  bool ContainsPoint(const V3 &p) const                                        // :36-43
  {
    if ((p.x < min.x) || (max.x < p.x)) return false;
    if ((p.y < min.y) || (max.y < p.y)) return false;
    if ((p.z < min.z) || (max.z < p.z)) return false;
    return true;
  }
  void AddPoint(const V3 &p)                                                   // :45-53
  {
    min.x = Min(min.x, p.x); min.y = Min(min.y, p.y); min.z = Min(min.z, p.z);
    max.x = Max(max.x, p.x); max.y = Max(max.y, p.y); max.z = Max(max.z, p.z);
  }
  void AddBox(const Box &o)                                                    // :55-63
  {
    min.x = Min(min.x, o.min.x); min.y = Min(min.y, o.min.y); min.z = Min(min.z, o.min.z);
    max.x = Max(max.x, o.max.x); max.y = Max(max.y, o.max.y); max.z = Max(max.z, o.max.z);
  }
  V3 Centroid() const { return .5 * (min + max); }                             // :65-68
  V3 Diagonal() const { return max - min; }                                    // :70-73
};

bool BoxRayIntersect(const Box &box, const V3 &o, const V3 &d,
    double ray_tmin, double ray_tmax, double *hit_tmin, double *hit_tmax);
static inline bool BoxBoxIntersect(const Box &a, const Box &b)                 // :140-152
{
  return !(a.max.x < b.min.x || a.min.x > b.max.x ||
           a.max.y < b.min.y || a.min.y > b.max.y ||
           a.max.z < b.min.z || a.min.z > b.max.z);
}

// src/fj_random.cc:10-43
struct XorShift {
  uint32_t s[4];
  XorShift() { s[0] = 123456789; s[1] = 362436069; s[2] = 521288629; s[3] = 88675123; }
  uint32_t NextInteger()
  {
    const uint32_t t = (s[0] ^ (s[0] << 11));
    s[0] = s[1]; s[1] = s[2]; s[2] = s[3];
    s[3] = (s[3] ^ (s[3] >> 19)) ^ (t ^ (t >> 8));
    return s[3];
  }
  double NextFloat01() { return static_cast<double>(NextInteger()) / UINT32_MAX; }
};

// src/fj_matrix.{h,cc}: row-major 4x4 f64
struct Mat { double e[16]; };
void MatIdentity(Mat *m);
void MatMultiply(Mat *dst, const Mat &a, const Mat &b);
void MatInverse(Mat *dst, const Mat &a);
static inline V3 MatTransformPoint(const Mat &m, const V3 &p)                  // :208-214
{
  return V3(m.e[0] * p.x + m.e[1] * p.y + m.e[2] * p.z + m.e[3],
            m.e[4] * p.x + m.e[5] * p.y + m.e[6] * p.z + m.e[7],
            m.e[8] * p.x + m.e[9] * p.y + m.e[10] * p.z + m.e[11]);
}
static inline V3 MatTransformVector(const Mat &m, const V3 &v)                 // :216-222
{
  return V3(m.e[0] * v.x + m.e[1] * v.y + m.e[2] * v.z,
            m.e[4] * v.x + m.e[5] * v.y + m.e[6] * v.z,
            m.e[8] * v.x + m.e[9] * v.y + m.e[10] * v.z);
}
void MatTransformBounds(const Mat &m, Box *bounds);

// Transform = matrix + inverse, src/fj_transform.cc:324-391
struct Xfm { Mat matrix, inverse; V3 translate, rotate, scale; };
void XfmSetTransform(Xfm *x, int transform_order, int rotate_order,
    double tx, double ty, double tz, double rx, double ry, double rz,
    double sx, double sy, double sz);

}  // namespace fjo
#endif
