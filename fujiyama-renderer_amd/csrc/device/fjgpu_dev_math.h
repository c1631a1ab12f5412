// fjgpu_dev_math.h -- vectors, culling and exact box tests, time-sampled transforms, Moller-Trumbore, wave helpers.
// Part of the kernels translation unit: included by fjgpu_kernels.hip only (device code,
// compiled with -ffp-contract=off; see the header of that file).
#ifndef FJGPU_DEV_MATH_H
#define FJGPU_DEV_MATH_H

// ------------------------------------------------------------------ vectors
struct V3 { double x, y, z; };
__device__ __forceinline__ V3 mk(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// Normalize, reference src/fj_vector.h:339-345: a * (1./len), len == 0 -> a
__device__ __forceinline__ V3 normalize(V3 a)
{
  const double len = sqrt(dot(a, a));
  if (len == 0) return a;
  const double inv = 1. / len;
  return a * inv;
}
// A pointer read from a record in memory is "generic" to the compiler, which then emits FLAT
// loads: they count against the LDS (LGKM) counter as well, so every wait for the LDS traversal
// stack also waits for the node or triangle fetch in flight.  These arrays are global memory.
#define FJ_GLOBAL __attribute__((address_space(1)))
typedef float fj_v4f __attribute__((ext_vector_type(4)));
typedef float fj_v2f __attribute__((ext_vector_type(2)));
typedef uint32_t fj_v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ V3 ld3(const double *p) { return mk(p[0], p[1], p[2]); }
__device__ __forceinline__ V3 ld3(const FJ_GLOBAL double *p) { return mk(p[0], p[1], p[2]); }
// the same array as the compiler should see it: global memory
#define FJ_G(T, ptr) ((const FJ_GLOBAL T *) (ptr))
// MatTransformPoint / MatTransformVector, reference src/fj_matrix.cc:208-222
__device__ __forceinline__ V3 xpoint(const double *m, V3 p)
{
  return mk(m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3],
            m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7],
            m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]);
}
__device__ __forceinline__ V3 xvector(const double *m, V3 v)
{
  return mk(m[0] * v.x + m[1] * v.y + m[2] * v.z,
            m[4] * v.x + m[5] * v.y + m[6] * v.z,
            m[8] * v.x + m[9] * v.y + m[10] * v.z);
}

__device__ __forceinline__ double clampd(double x, double a, double b) { return x < a ? a : (x > b ? b : x); }

// ------------------------------------------------------------ culling tests
// Conservative slab test (culling only).  NaN from 0 * inf is ignored by
// fmin/fmax, which return the non-NaN operand.
__device__ __forceinline__ bool slab(const double bmin[3], const double bmax[3], V3 o, V3 inv,
    double tmin, double tmax, double *tnear)
{
  const double x0 = (bmin[0] - o.x) * inv.x, x1 = (bmax[0] - o.x) * inv.x;
  const double y0 = (bmin[1] - o.y) * inv.y, y1 = (bmax[1] - o.y) * inv.y;
  const double z0 = (bmin[2] - o.z) * inv.z, z1 = (bmax[2] - o.z) * inv.z;
  const double tn = fmax(fmax(fmin(x0, x1), fmin(y0, y1)), fmax(fmin(z0, z1), tmin));
  const double tf = fmin(fmin(fmax(x0, x1), fmax(y0, y1)), fmin(fmax(z0, z1), tmax));
  *tnear = tn;
  return tn <= tf;
}

// the f64 test on a DNode child box ((min, max) pairs); kept for validation builds
__device__ __forceinline__ bool slab_f32box(fj_v2f px, fj_v2f py, fj_v2f pz, V3 o, V3 inv,
    double tmin, double tmax, double *tnear)
{
  const double mn[3] = {(double) px.x, (double) py.x, (double) pz.x};
  const double mx[3] = {(double) px.y, (double) py.y, (double) pz.y};
  return slab(mn, mx, o, inv, tmin, tmax, tnear);
}

// ---- conservative f32 slab test on the BLAS node boxes (culling only).
// DNode stores a child's box as three (min, max) pairs, so one packed fma yields both plane
// distances of an axis.  Per (ray, instance) and axis a the entry code keeps
//   i = (float)(1/od_a),  o = (float)(oo_a/od_a),  e = 3.6e-7 (Bmax_a |i| + |o|),
//   l = -(o + e),  h = e - o,     Bmax_a >= |any box coordinate| of the primitive set.
// For a box plane b (f32) the exact distance is t = (b - oo_a)/od_a.  t~ = fmaf(b, i, -o)
// differs from t by at most |b/od| 2^-24 (rounding of i) + |oo/od| 2^-24 (rounding of o; the
// f64 reciprocal and product before it are good to 2^-49) + |t~| 2^-24 (the fma's rounding)
// <= 2^-23 (Bmax |i| + |o|)(1 + 2^-18) =: E.  e is 3 E; folding it into the addend costs
// another 2^-24 |o + e|, so fmaf(b, i, l) <= t - E' and fmaf(b, i, h) >= t + E' with E' > E:
// [min over the two planes of the l-variant, max of the h-variant] contains the exact slab
// interval, and the f64 interval of slab_f32box (own error ~2^-52).  Whatever the f64 test
// accepts this one accepts: it can only cull less.  An axis whose e is not a finite number
// below 1e30 (direction component zero or denormal: 1/od = inf) gets i = 0, l = -1e30,
// h = 1e30: interval [-1e30, 1e30], it never culls.  With e < 1e30 every product is finite or
// an infinity of one sign: no NaN can arise (box coordinates are finite).
// one axis: t_lo = fmaf(b, i, l), t_hi = fmaf(b, i, h).  (Plain values, no pointers into a
// struct: the compiler kept a by-reference Slab32 in scratch memory.)
struct Slab32Axis { float i, l, h; };
struct Slab32 { Slab32Axis x, y, z; };

// inv = 1/od good to 2^-49 (filter_rcp) or exact
__device__ __forceinline__ Slab32Axis slab32_axis(double inv, double oo, double bmax_abs)
{
  const float i = (float) inv, o = (float) (oo * inv);
  // (node boxes are rounded outward from the f64 bounds by up to 2 ulp: the factor covers 8)
  const float e = 3.6e-7f * ((float) bmax_abs * 1.000001f * fabsf(i) + fabsf(o));
  const bool ok = e < 1e30f;
  Slab32Axis a;
  a.i = ok ? i : 0.f;
  a.l = ok ? -(o + e) : -1e30f;
  a.h = ok ? e - o : 1e30f;
  return a;
}

// bounds = {min xyz, max xyz} of the primitive set (every node box lies inside)
__device__ __forceinline__ Slab32 slab32_setup(V3 oo, V3 inv, const double *bounds)
{
  Slab32 s;
  s.x = slab32_axis(inv.x, oo.x, fmax(fabs(bounds[0]), fabs(bounds[3])));
  s.y = slab32_axis(inv.y, oo.y, fmax(fabs(bounds[1]), fabs(bounds[4])));
  s.z = slab32_axis(inv.z, oo.z, fmax(fabs(bounds[2]), fabs(bounds[5])));
  return s;
}

// The same constants for QUANTISED boxes (DNodeQ): a plane is g0 + q cell with q an integer in
// [0, 65535], so t = (g0 + q cell - oo)/od = q (cell/od) + (g0 - oo)/od, and
//   t~ = fmaf((float) q, A, B),  A = (float)(cell/od),  B = (float)((g0 - oo)/od)
// is off by at most 2^-23 (65535 |A| + |B|)(1 + 2^-18) (roundings of A, of B, of the fma; (float) q
// is exact).  i = A, l = B - e, h = B + e with e three times that bound, as above.
//
// FJ_SLAB_PERM (default): the integer reaches the fma WITHOUT a conversion instruction.  A byte permute
// (v_perm_b32) drops the 16-bit coordinate into the mantissa of 2^23: as_float(0x4B000000 | q) IS the
// number 8388608 + q, exactly.  The fma then computes (8388608 + q) A + (c - 8388608 A) -- product and sum
// exact, ONE rounding of the result, which has the magnitude of t again -- so all that is new is the
// rounding of the shifted addend c' = fmaf(-8388608, A, c): half an ulp of a number of magnitude <= 8388608 |A| + |c|,
// i.e. <= |A| / 2 + 2^-24 |c| (the second term is inside e already).  l' and h' are moved outward by FJ_SLAB_PERM_PAD |A|
// (0.51) for it: the box grows by half a grid cell (of 65536 per axis).  Per child and axis: two permutes instead of a
// rotate and two SDWA conversions.  (Measured on C3 with a pad of 2 cells: 1 % more boxes pass, +2.4 % nodes, +10 % triangle
// tests -- leaf boxes are only ~80 cells wide; validated: 48.5 G box tests, 0 lost, profiles/r03_anyhit_wide8_and_perm.txt.)
#ifndef FJ_SLAB_PERM
#define FJ_SLAB_PERM 1
#endif
#ifndef FJ_SLAB_PERM_PAD
#define FJ_SLAB_PERM_PAD .51f
#endif
__device__ __forceinline__ Slab32Axis slab32q_axis(double inv, double oo, double g0, double cell)
{
  const float A = (float) (cell * inv), B = (float) ((g0 - oo) * inv);
  const float e = 3.6e-7f * (65535.f * 1.000001f * fabsf(A) + fabsf(B));
  const bool ok = e < 1e30f;
  Slab32Axis a;
  a.i = ok ? A : 0.f;
#if FJ_SLAB_PERM
  a.l = ok ? fmaf(-8388608.f, A, B - e) - FJ_SLAB_PERM_PAD * fabsf(A) : -1e30f;
  a.h = ok ? fmaf(-8388608.f, A, B + e) + FJ_SLAB_PERM_PAD * fabsf(A) : 1e30f;
#else
  a.l = ok ? B - e : -1e30f;
  a.h = ok ? B + e : 1e30f;
#endif
  return a;
}
__device__ __forceinline__ Slab32 slab32q_setup(V3 oo, V3 inv, const double *g0, const double *cell)
{
  Slab32 s;
  s.x = slab32q_axis(inv.x, oo.x, g0[0], cell[0]);
  s.y = slab32q_axis(inv.y, oo.y, g0[1], cell[1]);
  s.z = slab32q_axis(inv.z, oo.z, g0[2], cell[2]);
  return s;
}
// (min, max) pair of grid coordinates packed in one 32-bit word -> two floats
__device__ __forceinline__ fj_v2f unpack_q(uint32_t w) { fj_v2f p; p.x = (float) (w & 0xffffu); p.y = (float) (w >> 16); return p; }

// px py pz = the (min, max) pairs of a child box; tmin32 <= tmin and tmax32 >= tmax of the ray.
// *tnear = conservative entry distance (an ordering key for the closest-hit walk).
__device__ __forceinline__ bool slab32_test(fj_v2f px, fj_v2f py, fj_v2f pz, const Slab32 s, float tmin32, float tmax32, float *tnear)
{
  const fj_v2f lx = __builtin_elementwise_fma(px, (fj_v2f) (s.x.i), (fj_v2f) (s.x.l));
  const fj_v2f hx = __builtin_elementwise_fma(px, (fj_v2f) (s.x.i), (fj_v2f) (s.x.h));
  const fj_v2f ly = __builtin_elementwise_fma(py, (fj_v2f) (s.y.i), (fj_v2f) (s.y.l));
  const fj_v2f hy = __builtin_elementwise_fma(py, (fj_v2f) (s.y.i), (fj_v2f) (s.y.h));
  const fj_v2f lz = __builtin_elementwise_fma(pz, (fj_v2f) (s.z.i), (fj_v2f) (s.z.l));
  const fj_v2f hz = __builtin_elementwise_fma(pz, (fj_v2f) (s.z.i), (fj_v2f) (s.z.h));
  const float tn = fmaxf(fmaxf(fmaxf(fminf(lx.x, lx.y), fminf(ly.x, ly.y)), fminf(lz.x, lz.y)), tmin32);
  const float tf = fminf(fminf(fminf(fmaxf(hx.x, hx.y), fmaxf(hy.x, hy.y)), fmaxf(hz.x, hz.y)), tmax32);
  *tnear = tn;
  return tn <= tf;
}
// The same test on a QUANTISED box with the direction signs of the ray used up front: wx wy wz are the
// (min, max) words of one child; sh* (slab32_shift) say which half of a word is the NEAR plane on that
// axis -- the low one unless the ray runs towards smaller coordinates.  With the near plane in the first
// and the far plane in the second element, one packed fma per axis yields (entry, exit) directly --
// fma(b, i, l) is monotonic in b, hence min(fma(min, i, l), fma(max, i, l)) IS fma(near, i, l) and
// likewise for the exit side: the same numbers as slab32_test, 4 instead of 6 instructions per axis.
//   FJ_SLAB_PERM 1: sh = the v_perm_b32 selector that builds as_float(0x4B000000 | near half) -- result bytes 3, 2 =
//                   bytes 3, 2 of the constant (selector values 7, 6), bytes 1, 0 = the word's (1, 0) or (3, 2); the
//                   far half's selector is sh ^ 0x0202
//   FJ_SLAB_PERM 0: sh = 16 or 0, the word is rotated (v_alignbit_b32) and converted (2 SDWA cvt)
#if FJ_SLAB_PERM
__device__ __forceinline__ uint32_t slab32_shift(float i) { return 0x07060100u ^ ((__float_as_uint(i) >> 31) * 0x0202u); }
__device__ __forceinline__ fj_v2f slab32q_planes(uint32_t w, uint32_t sh)
{
  fj_v2f p;
  p.x = __uint_as_float(__builtin_amdgcn_perm(0x4B000000u, w, sh));
  p.y = __uint_as_float(__builtin_amdgcn_perm(0x4B000000u, w, sh ^ 0x0202u));
  return p;
}
#else
__device__ __forceinline__ uint32_t slab32_shift(float i) { return (__float_as_uint(i) >> 31) << 4; }
__device__ __forceinline__ fj_v2f slab32q_planes(uint32_t w, uint32_t sh) { return unpack_q(__builtin_amdgcn_alignbit(w, w, sh)); }
#endif
__device__ __forceinline__ bool slab32q_test(uint32_t wx, uint32_t wy, uint32_t wz, const Slab32 s, uint32_t shx, uint32_t shy, uint32_t shz,
    float tmin32, float tmax32, float *tnear = nullptr)
{
  const fj_v2f px = slab32q_planes(wx, shx);
  const fj_v2f py = slab32q_planes(wy, shy);
  const fj_v2f pz = slab32q_planes(wz, shz);
  fj_v2f cx, cy, cz;
  cx.x = s.x.l; cx.y = s.x.h; cy.x = s.y.l; cy.y = s.y.h; cz.x = s.z.l; cz.y = s.z.h;
  const fj_v2f tx = __builtin_elementwise_fma(px, (fj_v2f) (s.x.i), cx);
  const fj_v2f ty = __builtin_elementwise_fma(py, (fj_v2f) (s.y.i), cy);
  const fj_v2f tz = __builtin_elementwise_fma(pz, (fj_v2f) (s.z.i), cz);
  const float tn = fmaxf(fmaxf(fmaxf(tx.x, ty.x), tz.x), tmin32);
  const float tf = fminf(fminf(fminf(tx.y, ty.y), tz.y), tmax32);
  if (tnear) *tnear = tn;
  return tn <= tf;
}
// the same with the slab constants held the way the packed fma wants them: (l, h) of an axis as ONE two-float value (a register pair: built
// per test from two scalars it cost the any-hit walk ten moves per iteration), the slope as a scalar (broadcast by the instruction's op_sel)
struct Slab32P { float ix, iy, iz; fj_v2f lhx, lhy, lhz; };
__device__ __forceinline__ Slab32P slab32p(const Slab32 s)
{
  Slab32P p;
  p.ix = s.x.i; p.iy = s.y.i; p.iz = s.z.i;
  p.lhx.x = s.x.l; p.lhx.y = s.x.h; p.lhy.x = s.y.l; p.lhy.y = s.y.h; p.lhz.x = s.z.l; p.lhz.y = s.z.h;
  return p;
}
__device__ __forceinline__ bool slab32q_test(uint32_t wx, uint32_t wy, uint32_t wz, const Slab32P s, uint32_t shx, uint32_t shy, uint32_t shz,
    float tmin32, float tmax32)
{
  const fj_v2f tx = __builtin_elementwise_fma(slab32q_planes(wx, shx), (fj_v2f) (s.ix), s.lhx);
  const fj_v2f ty = __builtin_elementwise_fma(slab32q_planes(wy, shy), (fj_v2f) (s.iy), s.lhy);
  const fj_v2f tz = __builtin_elementwise_fma(slab32q_planes(wz, shz), (fj_v2f) (s.iz), s.lhz);
  const float tn = fmaxf(fmaxf(fmaxf(tx.x, ty.x), tz.x), tmin32);
  const float tf = fminf(fminf(fminf(tx.y, ty.y), tz.y), tmax32);
  return tn <= tf;
}
// f32 bounds of an f64 ray range: down / up to the neighbouring float
__device__ __forceinline__ float f32_below(double x) { const float f = (float) x; return (double) f <= x ? f : nextafterf(f, -INFINITY); }
__device__ __forceinline__ float f32_above(double x) { const float f = (float) x; return (double) f >= x ? f : nextafterf(f, INFINITY); }

// Reference quirk kept for identical results: BoxRayIntersect (src/fj_box.cc:73-138)
// branches on `dir >= 0`, which is true for -0.0, and then divides by -0.0: the
// slab interval comes out reversed (+inf, -inf) and EVERY box test on that ray
// fails -- the group's bounds test in world space, an instance's accelerator
// bounds test in object space.  A ray with a negative-zero direction component
// therefore hits nothing there.  (+0.0 behaves normally.)
__device__ __forceinline__ bool has_negative_zero(V3 d)
{
  return (d.x == 0 && signbit(d.x)) || (d.y == 0 && signbit(d.y)) || (d.z == 0 && signbit(d.z));
}

// ---- time-sampled transforms (motion blur).  The sample's time is draw k of the per-tile
// time stream mapped to sample_time_range (FixedGridSampler, src/fj_fixed_grid_sampler.cc:
// 73-77: Fit(rnd, 0, 1, start, end)); k = sample index in the tile = low 20 bits of uid.
__device__ __forceinline__ double sample_time(const DScene &S, uint32_t tindex)
{
  const double x = S.time_tab[tindex];
  if (x <= 0) return S.time_start;
  if (x >= 1) return S.time_end;
  return S.time_start + (S.time_end - S.time_start) * ((x - 0) / (1 - 0));
}

// XfmLerpTransformSample + matrix + Cramer inverse at `time` (fjgpu_xform_math.h: the host's
// source, compiled for the device).  Out of line on purpose: it is rare and register hungry.
__device__ __noinline__ void xform_at(const fj_xform_desc *x, double time, double *M, double *Minv)
{
  double m[16], mi[16];
  fjx::make_transform(*x, time, m, mi);
  for (int k = 0; k < 12; k++) { M[k] = m[k]; Minv[k] = mi[k]; }
}

// BoxRayIntersect, reference src/fj_box.cc:73-138, operation for operation.  Used where
// the reference's own box decides the RESULT (instance bounds, which are not always
// conservative -- see fjgpu_build.cc), as opposed to pure culling.
#ifndef FJ_BOXREF_ATTR
#define FJ_BOXREF_ATTR __forceinline__
#endif
// (BoxPtr: `const double *`, or a pointer with an address space -- a box behind a pointer that came out of a record is generic to the
// compiler: FLAT loads, each waited for with both counters)
template <class BoxPtr>
__device__ FJ_BOXREF_ATTR bool box_ray_ref(BoxPtr b, V3 o, V3 d, double ray_tmin, double ray_tmax)
{
  double tmin, tmax, tymin, tymax, tzmin, tzmax;
  if (d.x >= 0) { tmin = (b[0] - o.x) / d.x; tmax = (b[3] - o.x) / d.x; }
  else          { tmin = (b[3] - o.x) / d.x; tmax = (b[0] - o.x) / d.x; }
  if (d.y >= 0) { tymin = (b[1] - o.y) / d.y; tymax = (b[4] - o.y) / d.y; }
  else          { tymin = (b[4] - o.y) / d.y; tymax = (b[1] - o.y) / d.y; }
  if ((tmin > tymax) || (tymin > tmax)) return false;
  if (tymin > tmin) tmin = tymin;
  if (tymax < tmax) tmax = tymax;
  if (d.z >= 0) { tzmin = (b[2] - o.z) / d.z; tzmax = (b[5] - o.z) / d.z; }
  else          { tzmin = (b[5] - o.z) / d.z; tzmax = (b[2] - o.z) / d.z; }
  if ((tmin > tzmax) || (tzmin > tmax)) return false;
  if (tzmin > tmin) tmin = tzmin;
  if (tzmax < tmax) tmax = tzmax;
  return (tmin < ray_tmax) && (tmax > ray_tmin);
}

// Same decision as box_ray_ref at a fraction of the cost: the slab interval from the
// per-ray reciprocal differs from the reference's divisions by a few ulp, so it settles
// every case that is not within 1e-12 (relative) of a boundary; the rest takes the exact
// path.  `plain` = no direction component is zero (else 0 * inf = NaN: exact path);
// overflow makes m infinite, which also lands in the exact path.
template <class BoxPtr>
__device__ __forceinline__ bool box_ray_ref_fast(BoxPtr b, V3 o, V3 d, V3 inv, bool plain, double ray_tmin, double ray_tmax)
{
  const double x0 = (b[0] - o.x) * inv.x, x1 = (b[3] - o.x) * inv.x;
  const double y0 = (b[1] - o.y) * inv.y, y1 = (b[4] - o.y) * inv.y;
  const double z0 = (b[2] - o.z) * inv.z, z1 = (b[5] - o.z) * inv.z;
  const double lo = fmax(fmax(fmin(x0, x1), fmin(y0, y1)), fmin(z0, z1));
  const double hi = fmin(fmin(fmax(x0, x1), fmax(y0, y1)), fmax(z0, z1));
  const double m = 1e-12 * ((fabs(lo) + fabs(hi)) + (fabs(ray_tmin) + fmin(fabs(ray_tmax), 1e300)));
  const double g = fmin(fmin(hi - lo, ray_tmax - lo), hi - ray_tmin);
  if (plain) {
    if (g > m) return true;
    if (g < -m) return false;
  }
  return box_ray_ref(b, o, d, ray_tmin, ray_tmax);
}
__device__ __forceinline__ bool plain_dir(V3 d) { return d.x != 0 && d.y != 0 && d.z != 0; }

// reciprocal for the FILTER only (box_ray_ref_fast decides nothing within 1e-12 of a boundary,
// and asks box_ray_ref there): hardware estimate + two Newton steps, ~1e-16 relative, a
// quarter of the instructions of a correctly rounded division.  Zero gives inf / NaN, which
// plain_dir() has already routed to the exact path.
__device__ __forceinline__ double filter_rcp(double x)
{
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
  r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
  return r;
}

// vertices of leaf slot i: f64 as stored, or f32 widened (exact) -- see DPrimSet
__device__ __forceinline__ void load_tri(const double *t64, const float *t32, uint32_t i, V3 *v0, V3 *v1, V3 *v2)
{
  if (t32) {
    const FJ_GLOBAL float *p = (const FJ_GLOBAL float *) t32 + (size_t) i * 9;
    *v0 = mk((double) p[0], (double) p[1], (double) p[2]);
    *v1 = mk((double) p[3], (double) p[4], (double) p[5]);
    *v2 = mk((double) p[6], (double) p[7], (double) p[8]);
  } else {
    const FJ_GLOBAL double *p = (const FJ_GLOBAL double *) t64 + (size_t) i * 9;
    *v0 = mk(p[0], p[1], p[2]); *v1 = mk(p[3], p[4], p[5]); *v2 = mk(p[6], p[7], p[8]);
  }
}

// ------------------------------------------------------- triangle test (a21)
// TriRayIntersect, reference src/fj_triangle.cc:81-153, non-culling branch,
// EPSILON 1e-6 (:12); no t-sign test here -- the range test is the caller's
// (PrimitiveSet::RayIntersect, src/fj_primitive_set.cc:10-26).
__device__ __forceinline__ bool tri_ray(V3 v0, V3 v1, V3 v2, V3 orig, V3 dir, double *t, double *u, double *v)
{
  const V3 edge1 = v1 - v0;
  const V3 edge2 = v2 - v0;
  const V3 pvec = cross(dir, edge2);
  const double det = dot(edge1, pvec);
  if (det > -1e-6 && det < 1e-6) return false;
  const double inv_det = 1.0 / det;
  const V3 tvec = orig - v0;
  const double uu = dot(tvec, pvec) * inv_det;
  if (uu < 0.0 || uu > 1.0) return false;
  const V3 qvec = cross(tvec, edge1);
  const double vv = dot(dir, qvec) * inv_det;
  if (vv < 0.0 || uu + vv > 1.0) return false;
  *t = dot(edge2, qvec) * inv_det;
  *u = uu;
  *v = vv;
  return true;
}

// The same decision and the same t, u, v with the misses settled before the division where the reference's own arithmetic provably rejects
// (the rules of tri_ray_anyhit, fjgpu_dev_anyhit.h): u = U * fl(1 / det) < 0 when U and det differ in sign and the product cannot underflow
// (|U| > 1e-100 |det|); u > 1 when |U| > |det| (1 + 1e-10); the same for v.  Whatever is left takes tri_ray's statements as they are.
__device__ __forceinline__ bool tri_ray_early(V3 v0, V3 v1, V3 v2, V3 orig, V3 dir, double *t, double *u, double *v)
{
  const V3 edge1 = v1 - v0;
  const V3 edge2 = v2 - v0;
  const V3 pvec = cross(dir, edge2);
  const double det = dot(edge1, pvec);
  if (det > -1e-6 && det < 1e-6) return false;
  const double adet = fabs(det);
  const V3 tvec = orig - v0;
  const double U = dot(tvec, pvec);
  const double aU = fabs(U);
  if (((U < 0.0) != (det < 0.0)) ? aU > 1e-100 * adet : aU > adet * (1.0 + 1e-10)) return false;
  const V3 qvec = cross(tvec, edge1);
  const double V = dot(dir, qvec);
  const double aV = fabs(V);
  if (((V < 0.0) != (det < 0.0)) ? aV > 1e-100 * adet : aV > adet * (1.0 + 1e-10)) return false;
  const double inv_det = 1.0 / det;
  const double uu = U * inv_det;
  if (uu < 0.0 || uu > 1.0) return false;
  const double vv = V * inv_det;
  if (vv < 0.0 || uu + vv > 1.0) return false;
  *t = dot(edge2, qvec) * inv_det;
  *u = uu;
  *v = vv;
  return true;
}

// ----------------------------------------------------------------- traversal
struct Best { double t, u, v; int inst, prim; };

struct LocalCounters { uint32_t nodes, prims, insts; };

// sum over the 64 lanes of the wave (butterfly; every lane gets the total)
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v)
{
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned lo = __shfl_xor((unsigned) (v & 0xffffffffull), off);
    const unsigned hi = __shfl_xor((unsigned) (v >> 32), off);
    v += ((unsigned long long) hi << 32) | lo;
  }
  return v;
}

// one global atomic per counter per WAVE, issued once at the end of a
// persistent kernel (a per-thread atomic on one address serialises in L2)
__device__ __forceinline__ void flush_counters(DCounters *cnt, unsigned long long nodes, unsigned long long prims,
    unsigned long long insts, unsigned long long traced, unsigned long long shadow)
{
  nodes = wave_sum(nodes); prims = wave_sum(prims); insts = wave_sum(insts);
  traced = wave_sum(traced); shadow = wave_sum(shadow);
  if (__lane_id() == 0) {
    if (nodes) atomicAdd(&cnt->nodes, nodes);
    if (prims) atomicAdd(&cnt->prims, prims);
    if (insts) atomicAdd(&cnt->insts, insts);
    if (traced) atomicAdd(&cnt->traced, traced);
    if (shadow) atomicAdd(&cnt->rays[CXT_SHADOW_RAY], shadow);
  }
}

// the shadow walk's events, counted a second time on their own (per-kernel roofline)
__device__ __forceinline__ void flush_shadow_walk_counters(DCounters *cnt, unsigned long long nodes, unsigned long long prims,
    unsigned long long insts)
{
  nodes = wave_sum(nodes); prims = wave_sum(prims); insts = wave_sum(insts);
  if (__lane_id() == 0) {
    if (nodes) atomicAdd(&cnt->sh_nodes, nodes);
    if (prims) atomicAdd(&cnt->sh_prims, prims);
    if (insts) atomicAdd(&cnt->sh_insts, insts);
  }
}

#endif
