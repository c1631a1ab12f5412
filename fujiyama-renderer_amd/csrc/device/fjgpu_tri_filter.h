// fjgpu_tri_filter.h -- conservative f32 filter in front of the reference's FP64 triangle test.
//
// TriRayIntersect (reference src/fj_triangle.cc:81-153, non-culling branch; tri_ray of fjgpu_dev_math.h) decides
// hit / miss in FP64 with the reference's non-fused statements.  Nearly every triangle a walk tests is missed by a wide
// margin, so the walks first evaluate the same determinants in packed f32 FMAs together with a bound of their distance
// from the reference's FP64 values, and settle what that bound settles:
//   FJ_TRI_MISS   the reference's test returns false, or reports a t outside [tmin, tmax] (proved)
//   FJ_TRI_HIT    the reference's test returns true with tmin <= t <= tmax (proved; any-hit rays need no more)
//   FJ_TRI_MAYBE  undecided: the caller runs the reference's statements (tri_ray / tri_ray_anyhit)
// The contract is slab32q_test's: never decide against the exact test.  tests/test_tri_filter.py compiles THIS file for
// the host (clang++, same packed statements) and checks the contract on random and adversarial (ray, triangle) pairs;
// a -DFJ_TRI_FILTER_VALIDATE build of the kernels re-runs the exact test behind every decision of a whole frame.
//
// Inputs (object space of the instance).  Vertices are f32 values (the walks that call this read f32 triangle records:
// every coordinate exactly representable).  The FP64 ray (o, d) is held as
//   oh = fl32(o), ol = fl32(o - oh)   (o = oh + ol up to 2^-48 |o|),   d32 = fl32(d),
//   tfl = 2^-22 max|o_i| (rounded up): floor of the size of tvec below which the 2^-48 |o| term would count.
// u = 2^-24.  With Tm = max(|tvec_i|, tfl), Dm = max|d_i|, E1m = max|e1_i|, E2m = max|e2_i| (all of the computed f32
// values) the computed quantities lie within
//   |det - det*| <= 40 u Dm E1m E2m,   |U - U*| <= 52 u Tm Dm E2m,   |V - V*| <= 52 u Tm Dm E1m,   |T - T*| <= 52 u Tm E1m E2m
// of the reference's (x* = the FP64 statement's value; its own rounding, 2^-52 of the same products, is inside the
// slack), derived term by term in DESIGN.md 4 ("f32 triangle filter"): tvec = (oh - v0) + ol is good to 3 u Tm, a cross
// product component a b - c d by fma to 7.1 u max|a b| (inputs good to u), a three-term fma dot product adds 6 u max|term|.
// The constants used are 48 u (det) and 64 u (U, V, T) -- 20 % above the derivation -- plus an absolute 2^-60 that
// covers underflow (flushed or gradual: an underflowed product is off by <= 2^-126, amplified by at most two factors
// <= 2^30).  The caller guarantees the magnitudes: max|d_i| <= 2^30 and max|o_i| + max|vertex coordinate| <= 2^29
// (tri_filter_ray_ok); a ray outside gets NaN constants, for which every comparison below is false -> FJ_TRI_MAYBE.
#ifndef FJGPU_TRI_FILTER_H
#define FJGPU_TRI_FILTER_H

#include <stdint.h>

#ifdef __HIPCC__
#define FJ_TF_FN __device__ __forceinline__
#else
#define FJ_TF_FN static inline
#endif

typedef float fj_tf2 __attribute__((ext_vector_type(2)));

enum { FJ_TRI_MISS = 0, FJ_TRI_HIT = 1, FJ_TRI_MAYBE = 2 };

// the ray as the filter holds it (12 floats: the lean any-hit walk keeps them in LDS)
struct TriFilterRay {
  float ohx, ohy, ohz;         // fl32(o)
  float olx, oly, olz;         // fl32(o - oh)
  float dx, dy, dz;            // fl32(d)
  float tfl;                   // 2^-22 max|o_i|, rounded up
  float tmin_lo, tmin_hi;      // <= tmin (1 - 2^-19), >= tmin (1 + 2^-19)
  float tmax_lo, tmax_hi;      // <= tmax (1 - 2^-19), >= tmax (1 + 2^-19)
};

FJ_TF_FN float tf_absf(float x) { return __builtin_fabsf(x); }
FJ_TF_FN float tf_max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
FJ_TF_FN uint32_t tf_bits(float x) { union { float f; uint32_t u; } c; c.f = x; return c.u; }
FJ_TF_FN float tf_float(uint32_t x) { union { float f; uint32_t u; } c; c.u = x; return c.f; }
// next float towards +inf / -inf of a finite positive-or-negative value (no libm on the device path)
FJ_TF_FN float tf_up(float x) { if (x != x || x == __builtin_inff()) return x; if (x == 0.f) return tf_float(1u); const uint32_t b = tf_bits(x); return tf_float(x > 0.f ? b + 1u : b - 1u); }
FJ_TF_FN float tf_down(float x) { return -tf_up(-x); }
FJ_TF_FN float tf_above(double x) { const float f = (float) x; return (double) f >= x ? f : tf_up(f); }
FJ_TF_FN float tf_below(double x) { const float f = (float) x; return (double) f <= x ? f : tf_down(f); }

// bound_abs >= |any vertex coordinate| of the primitive set (object space).  A ray whose magnitudes the error analysis
// does not cover gets NaN constants (every test undecided).
FJ_TF_FN TriFilterRay tri_filter_ray(double ox, double oy, double oz, double dx, double dy, double dz, double tmin, double tmax, float bound_abs)
{
  TriFilterRay r;
  r.ohx = (float) ox; r.ohy = (float) oy; r.ohz = (float) oz;
  r.olx = (float) (ox - (double) r.ohx); r.oly = (float) (oy - (double) r.ohy); r.olz = (float) (oz - (double) r.ohz);
  r.dx = (float) dx; r.dy = (float) dy; r.dz = (float) dz;
  const float om = tf_max3(tf_absf(r.ohx), tf_absf(r.ohy), tf_absf(r.ohz)) * 1.0000002f;     // >= max|o_i|
  const float dm = tf_max3(tf_absf(r.dx), tf_absf(r.dy), tf_absf(r.dz));
  r.tfl = om * 2.3841864e-7f;                       // 2^-22 (1 + 2^-20)
  r.tmin_lo = tf_below(tmin * (1.0 - 1.9073486328125e-06)); r.tmin_hi = tf_above(tmin * (1.0 + 1.9073486328125e-06));
  r.tmax_lo = tf_below(tmax * (1.0 - 1.9073486328125e-06)); r.tmax_hi = tf_above(tmax * (1.0 + 1.9073486328125e-06));
  // (written so that a NaN anywhere fails the test)
  // (tmax may be huge or +inf -- dome-light samples sit at dir x FLT_MAX --: tmax_hi x D overflowing to +inf makes "t > tmax" unprovable and
  //  "t < tmax" true, which is what they are; a NaN tmax fails both comparisons)
  const bool ok = dm <= 1073741824.f && om + bound_abs <= 536870912.f && r.tmin_lo > 0.f;
  if (!ok) { const float q = __builtin_nanf(""); r.dx = r.dy = r.dz = q; r.tfl = q; }
  return r;
}

#define FJ_TF_CU_DET 2.8610229e-06f       // 48 x 2^-24
#define FJ_TF_CU 3.8146973e-06f           // 64 x 2^-24
#define FJ_TF_FLOOR 8.6736174e-19f        // 2^-60
#define FJ_TF_EPS_LO 9.9999905e-07f       // < 1e-6 (1 - 2^-20): below it |det*| < EPSILON for sure
#define FJ_TF_EPS_HI 1.0000010e-06f       // > 1e-6 (1 + 2^-20)
#define FJ_TF_ONE_P 1.0000010f            // 1 + 2^-20 (rounded up)
#define FJ_TF_ONE_M 0.99999899f           // 1 - 2^-20 (rounded down)

// p: the nine floats of the triangle record (v0 v1 v2).  kWantHit = false: hits are left undecided (closest-hit walks
// need the reference's t, u, v and run the exact statements for them anyway).
template <bool kWantHit, class FloatPtr>
FJ_TF_FN int tri_filter32(FloatPtr p, const TriFilterRay &r)
{
  const float v0x = p[0], v0y = p[1], v0z = p[2];
  // (e1 | e2), component by component
  fj_tf2 ex, ey, ez;
  ex.x = p[3]; ex.y = p[6]; ey.x = p[4]; ey.y = p[7]; ez.x = p[5]; ez.y = p[8];
  ex = ex - (fj_tf2) (v0x); ey = ey - (fj_tf2) (v0y); ez = ez - (fj_tf2) (v0z);
  // tvec = o - v0
  const float tx = (r.ohx - v0x) + r.olx, ty = (r.ohy - v0y) + r.oly, tz = (r.ohz - v0z) + r.olz;
  // (pvec | qvec) = (d | tvec) x (e2 | e1)
  fj_tf2 ax, ay, az;
  ax.x = r.dx; ax.y = tx; ay.x = r.dy; ay.y = ty; az.x = r.dz; az.y = tz;
  const fj_tf2 bx = ex.yx, by = ey.yx, bz = ez.yx;
  const fj_tf2 cx = __builtin_elementwise_fma(ay, bz, -(az * by));
  const fj_tf2 cy = __builtin_elementwise_fma(az, bx, -(ax * bz));
  const fj_tf2 cz = __builtin_elementwise_fma(ax, by, -(ay * bx));
  // (det | T) = (e1 | e2) . (pvec | qvec),   (U | V) = (tvec | d) . (pvec | qvec)
  const fj_tf2 dt = __builtin_elementwise_fma(ex, cx, __builtin_elementwise_fma(ey, cy, ez * cz));
  const fj_tf2 uv = __builtin_elementwise_fma(ax.yx, cx, __builtin_elementwise_fma(ay.yx, cy, az.yx * cz));
  // magnitudes
  const float e1m = tf_max3(tf_absf(ex.x), tf_absf(ey.x), tf_absf(ez.x));
  const float e2m = tf_max3(tf_absf(ex.y), tf_absf(ey.y), tf_absf(ez.y));
  const float tm = tf_max3(tf_absf(tx), tf_absf(ty), __builtin_fmaxf(tf_absf(tz), r.tfl));
  const float dm = tf_max3(tf_absf(r.dx), tf_absf(r.dy), tf_absf(r.dz));
  // bounds, packed: x = (48 u Dm | 64 u Tm);  (dD | dT) = x E1m E2m + floor;  (dV | dU) = 64 u Tm (Dm E1m | Dm E2m) + floor
  fj_tf2 em, x, cu;
  em.x = e1m; em.y = e2m;
  x.x = dm; x.y = tm;
  cu.x = FJ_TF_CU_DET; cu.y = FJ_TF_CU;
  x = x * cu;
  const fj_tf2 dDT = __builtin_elementwise_fma(x, (fj_tf2) (e1m * e2m), (fj_tf2) (FJ_TF_FLOOR));
  const fj_tf2 dvu = __builtin_elementwise_fma(x.yy, em * (fj_tf2) (dm), (fj_tf2) (FJ_TF_FLOOR));
  const float dD = dDT.x, dT = dDT.y;
  // everything on the side of det > 0
  const uint32_t s = tf_bits(dt.x) & 0x80000000u;
  const float D = tf_absf(dt.x);
  const float a = tf_float(tf_bits(uv.x) ^ s), b = tf_float(tf_bits(uv.y) ^ s), c = tf_float(tf_bits(dt.y) ^ s);
  const float Dhi = D + dD, Dlo = D - dD;
  const float am = a - dvu.y, bm = b - dvu.x, cm = c - dT, cp = c + dT;
  const float Dhi1 = Dhi * FJ_TF_ONE_P;
  // sure misses: |det| < EPSILON; u < 0; v < 0; u > 1; u + v > 1; t < tmin; t > tmax
  const bool sign_sure = Dlo > 0.f;
  const bool miss = (Dhi < FJ_TF_EPS_LO) |
      (sign_sure & ((a + dvu.y < 0.f) | (b + dvu.x < 0.f) | (am > Dhi1) | (am + bm > Dhi1) | (cp < r.tmin_lo * Dlo) | (cm > r.tmax_hi * Dhi)));
  if (miss) return FJ_TRI_MISS;
  if (kWantHit) {
    const float ap = a + dvu.y, bp = b + dvu.x;
    const bool hit = (Dlo > FJ_TF_EPS_HI) & (am > 0.f) & (bm > 0.f) & (ap + bp < Dlo * FJ_TF_ONE_M) & (cm > r.tmin_hi * Dhi) & (cp < r.tmax_lo * Dlo);
    if (hit) return FJ_TRI_HIT;
  }
  return FJ_TRI_MAYBE;
}

#endif
