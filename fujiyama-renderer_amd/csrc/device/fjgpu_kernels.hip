// fjgpu_kernels.hip -- hand-written HIP kernels of the MI355X (gfx950, wave64)
// ray-intersection + integrator core.  Compiled with -ffp-contract=off: every
// FP64 expression that decides a hit or feeds geometry keeps the reference's
// operation order (citations per function); culling arithmetic (slab tests on
// widened boxes) is free to differ because it can only reject provable misses.
//
// Kernel map:
//   k_gen_camera     FixedGridSampler::generate_samples + Camera::GetRay
//   k_trace_closest  Accelerator::Intersect over a group: instance loop ->
//                    BLAS (BVH2, LDS traversal stack) -> FP64 Moller-Trumbore
//   k_shade          trace_surface's attribute setup + the shader plugins;
//                    emits light records and child rays (ballot compaction)
//   k_shadow         SlIlluminance: (record, light) pairs, shadow rays,
//                    wave-segment reduction, accumulation into the sample
//   k_resolve        reconstruct_image / apply_pixel_filter
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "fjgpu_types.h"
#include "fjgpu_kernels.h"
#include "fjgpu_xform_math.h"

#define BLOCK 256

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
__device__ __forceinline__ V3 ld3(const double *p) { return mk(p[0], p[1], p[2]); }
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

__device__ __forceinline__ bool slab_f32box(const float *bmin, const float *bmax, V3 o, V3 inv,
    double tmin, double tmax, double *tnear)
{
  const double mn[3] = {(double) bmin[0], (double) bmin[1], (double) bmin[2]};
  const double mx[3] = {(double) bmax[0], (double) bmax[1], (double) bmax[2]};
  return slab(mn, mx, o, inv, tmin, tmax, tnear);
}

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
__device__ FJ_BOXREF_ATTR bool box_ray_ref(const double *b, V3 o, V3 d, double ray_tmin, double ray_tmax)
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
__device__ __forceinline__ bool box_ray_ref_fast(const double *b, V3 o, V3 d, V3 inv, bool plain, double ray_tmin, double ray_tmax)
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
    const float *p = t32 + (size_t) i * 9;
    *v0 = mk((double) p[0], (double) p[1], (double) p[2]);
    *v1 = mk((double) p[3], (double) p[4], (double) p[5]);
    *v2 = mk((double) p[6], (double) p[7], (double) p[8]);
  } else {
    const double *p = t64 + (size_t) i * 9;
    *v0 = ld3(p); *v1 = ld3(p + 3); *v2 = ld3(p + 6);
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

// --------------------------------------------------------- curve test (a23)
// Curve::ray_intersect + converge_bezier3, reference src/fj_curve.cc:187-232,300-390
// (Nakamaru-Ono subdivision in ray space).  The reference recurses with a
// Bezier3 per level; here every leaf segment is re-derived from the root by
// the same sequence of split_bezier3 calls (identical arithmetic, no per-lane
// stack of control points), and subtrees whose ancestor fails the reference's
// bounds test are skipped.  Children of a node start from a fresh "no hit"
// (t = REAL_MAX) in the reference, so nothing is pruned by depth; the combine
// rule `t_left < t_right ? left : right` selects the RIGHTMOST leaf among those
// with the smallest z, which is what `z <= best` in a left-to-right sweep does.
struct Bz { V3 c0, c1, c2, c3; double w0, w1; };

__device__ __forceinline__ V3 bez_eval(const Bz &b, double t)       // eval_bezier3, :464-472
{
  const double u = 1 - t;
  const double a = u * u * u;
  const double bb = 3 * u * u * t;
  const double c = 3 * u * t * t;
  const double d = t * t * t;
  return a * b.c0 + bb * b.c1 + c * b.c2 + d * b.c3;
}
__device__ __forceinline__ V3 mid_point(V3 a, V3 b) { return (a + b) * .5; }
__device__ __forceinline__ double dmax(double x, double y) { return x > y ? x : y; }   // Max, src/fj_numeric.h
__device__ __forceinline__ double dmin(double x, double y) { return x < y ? x : y; }
__device__ __forceinline__ double dot_xy(V3 a, V3 b) { return a.x * b.x + a.y * b.y; }

__device__ bool curve_ray(const double *cpw, double w0, double w1, int depth, V3 oo, V3 od, double *t_out, double *v_out)
{
  // nml_ray.dir = ray.dir / |ray.dir|  (Vector /= Real  ==  *= 1./s)
  const double ray_scale = sqrt(dot(od, od));
  const double sinv = 1. / ray_scale;
  const V3 nd = od * sinv;
  // compute_world_to_ray_matrix, :268-295: dst = rotate * translate
  const double lx = nd.x, ly = nd.y, lz = nd.z;
  const double d = sqrt(lx * lx + lz * lz);
  const double d_inv = 1. / d;
  const V3 r0 = mk(lz * d_inv, 0, -lx * d_inv);
  const V3 r1 = mk(-lx * ly * d_inv, d, -ly * lz * d_inv);
  const V3 r2 = mk(lx, ly, lz);
  const double nox = -oo.x, noy = -oo.y, noz = -oo.z;
  const double m03 = r0.x * nox + r0.y * noy + r0.z * noz;
  const double m13 = r1.x * nox + r1.y * noy + r1.z * noz;
  const double m23 = r2.x * nox + r2.y * noy + r2.z * noz;
  Bz root;
  {
    V3 p[4];
    for (int k = 0; k < 4; k++) {
      const V3 q = ld3(cpw + 3 * k);
      p[k] = mk(r0.x * q.x + r0.y * q.y + r0.z * q.z + m03,
                r1.x * q.x + r1.y * q.y + r1.z * q.z + m13,
                r2.x * q.x + r2.y * q.y + r2.z * q.z + m23);
    }
    root.c0 = p[0]; root.c1 = p[1]; root.c2 = p[2]; root.c3 = p[3];
    root.w0 = w0; root.w1 = w1;
  }
  double best_z = DBL_MAX, best_v = DBL_MAX;
  bool any = false;
  const uint32_t nleaf = 1u << depth;
  uint32_t j = 0;
  while (j < nleaf) {
    Bz b = root;
    double v0 = 0, vn = 1;
    bool pruned = false;
    for (int L = 0;; L++) {
      // converge_bezier3 entry test: get_bezier3_bounds (cp bounds +- max radius)
      const double radius = .5 * dmax(b.w0, b.w1);
      const double mnx = dmin(dmin(dmin(b.c0.x, b.c1.x), b.c2.x), b.c3.x) - radius;
      const double mxx = dmax(dmax(dmax(b.c0.x, b.c1.x), b.c2.x), b.c3.x) + radius;
      const double mny = dmin(dmin(dmin(b.c0.y, b.c1.y), b.c2.y), b.c3.y) - radius;
      const double mxy = dmax(dmax(dmax(b.c0.y, b.c1.y), b.c2.y), b.c3.y) + radius;
      const double mxz = dmax(dmax(dmax(b.c0.z, b.c1.z), b.c2.z), b.c3.z) + radius;
      if (mnx >= radius || mxx <= -radius || mny >= radius || mxy <= -radius || mxz <= 1e-6) {
        const uint32_t span = 1u << (depth - L);
        j = ((j / span) + 1) * span;
        pruned = true;
        break;
      }
      if (L == depth) break;
      // split_bezier3, :488-508, keeping the half selected by bit (depth-L-1) of j
      const V3 midP = bez_eval(b, .5);
      const V3 midCP = mid_point(b.c1, b.c2);
      const double vm = (v0 + vn) * .5;
      const double wm = (b.w0 + b.w1) * .5;
      if (((j >> (depth - L - 1)) & 1u) == 0) {
        const V3 l1 = mid_point(b.c0, b.c1);
        const V3 l2 = mid_point(l1, midCP);
        b.c1 = l1; b.c2 = l2; b.c3 = midP;
        b.w1 = wm;
        vn = vm;
      } else {
        const V3 q2 = mid_point(b.c3, b.c2);
        const V3 q1 = mid_point(q2, midCP);
        b.c0 = midP; b.c1 = q1; b.c2 = q2;
        b.w0 = wm;
        v0 = vm;
      }
    }
    if (pruned) continue;
    j++;
    // depth == 0 block of converge_bezier3
    const V3 dir = b.c3 - b.c0;
    V3 dP0 = b.c1 - b.c0;
    if (dot_xy(dir, dP0) < 0) dP0 = dP0 * -1;
    if (-1 * dot_xy(dP0, b.c0) < 0) continue;
    V3 dPn = b.c3 - b.c2;
    if (dot_xy(dir, dPn) < 0) dPn = dPn * -1;
    if (dot_xy(dPn, b.c3) < 0) continue;
    double w = dir.x * dir.x + dir.y * dir.y;
    if (fabs(w) < 1e-6) continue;
    w = -(b.c0.x * dir.x + b.c0.y * dir.y) / w;
    w = clampd(w, 0, 1);
    const double v = v0 * (1 - w) + vn * w;
    const double radius_w = .5 * ((1 - w) * b.w0 + w * b.w1);
    const V3 vP = bez_eval(b, w);
    if (vP.x * vP.x + vP.y * vP.y >= radius_w * radius_w) continue;
    if (vP.z <= 1e-6) continue;
    if (vP.z <= best_z) { best_z = vP.z; best_v = v; any = true; }
  }
  if (!any) return false;
  *t_out = best_z / ray_scale;
  *v_out = best_v;
  return true;
}

// The reference's GridAccelerator accepts a primitive hit only when the hit point
// lies inside the cell being walked (src/fj_grid_accelerator.cc:253-260), and a curve
// is listed in a cell only if one of its 32 depth-5 sub-segments' control-point boxes
// overlaps the cell (Curve::box_intersect, src/fj_curve.cc:234-242,399-462) -- WITHOUT
// the ribbon radius.  A ribbon hit whose ray point falls in a neighbouring cell that
// does not list the curve is therefore rejected by the reference.  The same rule is
// applied here so the two renderers see the same fur.
__device__ bool curve_listed_in_cell_of(const DPrimSet *P, const double *cpw, V3 hitp)
{
  int ci[3];
  double cmin[3], cmax[3];
  const double hp[3] = {hitp.x, hitp.y, hitp.z};
  for (int a = 0; a < 3; a++) {
    int c = (int) floor((hp[a] - P->bounds[a]) / P->grid_cell[a]);
    c = c < 0 ? 0 : (c > P->grid_n[a] - 1 ? P->grid_n[a] - 1 : c);
    ci[a] = c;
    cmin[a] = P->bounds[a] + (double) c * P->grid_cell[a];       // get_grid_cell, :334-343
    cmax[a] = cmin[a] + P->grid_cell[a];
    if (hp[a] < cmin[a] || cmax[a] < hp[a]) return false;        // Box::ContainsPoint (inclusive)
  }
  (void) ci;
  // box_bezier3_intersect_recursive(cell, bezier, 5) with zero velocity
  const V3 r0 = ld3(cpw), r1 = ld3(cpw + 3), r2 = ld3(cpw + 6), r3 = ld3(cpw + 9);
  const uint32_t depth = 5, nleaf = 32;
  uint32_t j = 0;
  while (j < nleaf) {
    V3 c0 = r0, c1 = r1, c2 = r2, c3 = r3;
    bool pruned = false;
    for (uint32_t L = 0;; L++) {
      // AABB of the control polygon vs the cell (BoxBoxIntersect, inclusive).  Inner
      // levels are tested too: a sub-segment's control points stay inside the parent's hull
      const double mn[3] = {dmin(dmin(dmin(c0.x, c1.x), c2.x), c3.x), dmin(dmin(dmin(c0.y, c1.y), c2.y), c3.y), dmin(dmin(dmin(c0.z, c1.z), c2.z), c3.z)};
      const double mx[3] = {dmax(dmax(dmax(c0.x, c1.x), c2.x), c3.x), dmax(dmax(dmax(c0.y, c1.y), c2.y), c3.y), dmax(dmax(dmax(c0.z, c1.z), c2.z), c3.z)};
      const bool overlap = !(mx[0] < cmin[0] || mn[0] > cmax[0] || mx[1] < cmin[1] || mn[1] > cmax[1] || mx[2] < cmin[2] || mn[2] > cmax[2]);
      if (!overlap) {
        const uint32_t span = 1u << (depth - L);
        j = ((j / span) + 1) * span;
        pruned = true;
        break;
      }
      if (L == depth) return true;
      Bz b;
      b.c0 = c0; b.c1 = c1; b.c2 = c2; b.c3 = c3; b.w0 = b.w1 = 0;
      const V3 midP = bez_eval(b, .5);
      const V3 midCP = mid_point(c1, c2);
      if (((j >> (depth - L - 1)) & 1u) == 0) {
        const V3 l1 = mid_point(c0, c1);
        const V3 l2 = mid_point(l1, midCP);
        c1 = l1; c2 = l2; c3 = midP;
      } else {
        const V3 q2 = mid_point(c3, c2);
        const V3 q1 = mid_point(q2, midCP);
        c0 = midP; c1 = q1; c2 = q2;
      }
    }
    if (!pruned) j++;
  }
  return false;
}

// ----------------------------------------------------- persistent traversal
// One traversal engine for closest-hit and any-hit rays, written as a per-lane
// state machine so that a lane that finishes its ray is refilled from the
// wave's slice of the queue instead of idling until the slowest lane of the
// wave is done (ray costs are heavy tailed: most shadow rays leave the BLAS
// after a few nodes, a few walk hundreds).  Each wave owns a contiguous slice
// of the ray queue (static split, no global work counter) and hands indices to
// its idle lanes with ballot + prefix popcount.
//
// Semantics reproduced: a hit counts iff tmin <= t <= tmax with
// the ORIGINAL ray range (RayInRange, src/fj_ray.h:29-32); the closest one wins
// with strict '<' (src/fj_bvh_accelerator.cc:183, src/fj_grid_accelerator.cc:263);
// at exactly equal t inside one mesh the larger primitive id wins (the grid's
// LIFO cell lists test it first).  Instances of the group are visited in group
// order (ObjectInstance::RayIntersect, src/fj_object_instance.cc:213-243: the ray
// goes to object space with M^-1 and dir is NOT renormalised, so t is preserved).
#define TRAV_DONE 0xffffffffu
// tunables (defaults measured on C3; overridable with FJGPU_TRAV_{REFILL,STEPS,GRAB})
struct TravTune { uint32_t refill, steps, grab; };
#define TRAV_REFILL tune.refill   // refill when at least this many lanes are idle
#define TRAV_STEPS (int) tune.steps   // inner-node steps between leaf / refill checks
#define TRAV_GRAB tune.grab       // queue entries a wave claims per global atomic

// Claim size.  A wave takes `grab` consecutive rays per atomic; with few rays per launch (a
// rank of an 8-GPU job, a deep recursion level) whole claims decide the load balance -- 4
// claims per wave leave the slowest wave ~25 % behind -- so the claim shrinks until every
// wave gets at least ~16 of them (never below 16 rays: a wave has 64 lanes to fill).
__device__ __forceinline__ uint32_t adaptive_grab(uint32_t grab, uint32_t n)
{
  const uint32_t waves = gridDim.x * (BLOCK / 64);
  const uint32_t want = n / (waves * 16u);
  return want >= grab ? grab : (want < 16u ? 16u : want);
}

struct RayIn { V3 o, d; double tmin, tmax, time; int group; bool anyhit; };

// Per-lane traversal stack: the first FJ_STACK_LDS entries in LDS ([depth][thread], lane
// consecutive, conflict free), deeper ones -- the builder reports the worst case of the
// scene's trees -- in a global overflow area ([depth][global thread]).
struct TravStack {
  uint32_t *lds;        // s_stack + threadIdx.x
  uint32_t *ovf;        // overflow base + global thread id (null when no tree needs it)
  uint32_t ovf_stride;  // threads in the grid
  __device__ __forceinline__ void push(int &sp, uint32_t v) const
  {
    if (sp < FJ_STACK_LDS) lds[sp * BLOCK] = v;
    else ovf[(size_t) (sp - FJ_STACK_LDS) * ovf_stride] = v;
    sp++;
  }
  __device__ __forceinline__ uint32_t pop(int &sp) const
  {
    --sp;
    return sp < FJ_STACK_LDS ? lds[sp * BLOCK] : ovf[(size_t) (sp - FJ_STACK_LDS) * ovf_stride];
  }
};
__device__ __forceinline__ TravStack make_stack(uint32_t *s_stack, uint32_t *ovf)
{
  TravStack st;
  st.lds = s_stack + threadIdx.x;
  st.ovf_stride = gridDim.x * BLOCK;
  st.ovf = ovf ? ovf + (size_t) blockIdx.x * BLOCK + threadIdx.x : nullptr;
  return st;
}

template <bool kCurves, bool kCount, bool kMotion, class Policy>
__device__ void traverse_persistent(const DScene &S, Policy &pol, TravTune tune, uint32_t n, uint32_t *head, TravStack stk, LocalCounters *lc)
{
  const unsigned lane = __lane_id();
  bool head_live = true;                   // wave-uniform: the global head still has entries
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  // work distribution: waves claim TRAV_GRAB consecutive queue entries at a time
  // from a global head (one atomic per 1024 rays).  Static per-wave slices were
  // measurably worse: neighbouring rays have correlated cost, so whole slices
  // end up cheap or expensive and the slowest wave sets the kernel time.
  uint32_t next = 0, range_end = 0;        // wave-uniform
  tune.grab = adaptive_grab(tune.grab, n);
  bool have = false;
  uint32_t idx = 0;
  V3 o = mk(0, 0, 0), oo = o, od = o, inv = o, d = o, winv = o;
  double tmin = 0, tmax = 0, rtime = 0;
  Best best;
  best.t = DBL_MAX; best.u = best.v = 0; best.inst = -1; best.prim = -1;
  int gfirst = 0, gcount = 0, gi = 0, ii = -1;
  const double *gsb = nullptr;
  bool anyhit = false, dead_ray = false, plain = false;
  const DPrimSet *P = nullptr;
  uint32_t cur = TRAV_DONE;
  uint32_t last_curve = 0xffffffffu;      // curve tested last for this (ray, instance)
  int sp = 0;

  for (;;) {
    // ---- refill idle lanes from the wave's slice
    const unsigned long long idle = __ballot(!have);
    if (next >= range_end && head_live && (idle == ~0ull || (unsigned) __popcll(idle) >= TRAV_REFILL)) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(head, (uint32_t) TRAV_GRAB);
      base = __shfl(base, 0);
      if (base >= n) head_live = false;
      else { next = base; range_end = (n - base < TRAV_GRAB) ? n : base + TRAV_GRAB; }
    }
    if (idle == ~0ull || ((unsigned) __popcll(idle) >= TRAV_REFILL && next < range_end)) {
      if (!have) {
        const uint32_t my = next + (uint32_t) __popcll(idle & lt_mask);
        if (my < range_end) {
          RayIn r;
          r.o = r.d = mk(0, 0, 1); r.tmin = r.tmax = r.time = 0; r.group = 0; r.anyhit = false;
          have = pol.fetch(my, &r);
          idx = my;
          o = r.o; d = r.d; tmin = r.tmin; tmax = r.tmax; anyhit = r.anyhit;
          if (kMotion) rtime = r.time;
          const DGroup G = S.groups[r.group];
          gfirst = G.first; gcount = G.count; gi = 0;
          gsb = S.groups[r.group].sbounds;
          best.t = DBL_MAX; best.u = best.v = 0; best.inst = -1; best.prim = -1;
          cur = TRAV_DONE; sp = 0;
          dead_ray = has_negative_zero(d);   // every box test of the reference fails (see above)
          winv = mk(filter_rcp(d.x), filter_rcp(d.y), filter_rcp(d.z));
          plain = plain_dir(d);
        }
      }
      next += (uint32_t) __popcll(idle);
      if (__ballot(have) == 0ull) {
        if (next >= range_end && !head_live) break;
        continue;               // only padding slots were fetched / slice exhausted: claim more
      }
    }

    // ---- lanes between instances: enter the next instance or retire the ray
    if (have && cur == TRAV_DONE) {
      bool found = false;
      while (!dead_ray && gi < gcount) {
        ii = S.group_instances[gfirst + gi];
        gi++;
        const DInstance *I = &S.instances[ii];
        if (kCount) lc->insts++;
        double tn;
        const double tfar = anyhit ? tmax : fmin(tmax, best.t);
        // the reference's own (possibly non-enclosing) instance box, full ray range
        if (!box_ray_ref_fast(gcount == 1 ? gsb : I->wbounds, o, d, winv, plain, tmin, tmax)) continue;
        if (kMotion && I->xform >= 0) {
          // ObjectInstance::RayIntersect evaluates a time-sampled transform at the ray's time
          double tm[12], tmi[12];
          xform_at(&S.xforms[I->xform], rtime, tm, tmi);
          oo = xpoint(tmi, o);
          od = xvector(tmi, d);
        } else {
          oo = xpoint(I->Minv, o);
          od = xvector(I->Minv, d);
        }
        if (has_negative_zero(od)) continue;
        inv = mk(1. / od.x, 1. / od.y, 1. / od.z);
        P = &S.primsets[I->primset];
        if (P->n_prims == 0) continue;
        if (!slab(P->bounds, P->bounds + 3, oo, inv, tmin, tfar, &tn)) continue;
        found = true;
        break;
      }
      if (found) { cur = P->root; sp = 0; last_curve = 0xffffffffu; }
      else { pol.finish(idx, best); have = false; }
    }

    // ---- inner nodes: a few steps for every lane that holds one
    for (int step = 0; step < TRAV_STEPS; step++) {
      const bool inner = have && !(cur & FJ_LEAF_FLAG);
      if (__ballot(inner) == 0ull) break;
      if (inner) {
        const float4 *nd = reinterpret_cast<const float4 *>(&P->nodes[cur]);
        if (kCount) lc->nodes++;
        // 128-byte node: eight 16-byte loads (4 child boxes + 4 child refs)
        const float4 q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3], q4 = nd[4], q5 = nd[5];
        const uint4 e = reinterpret_cast<const uint4 *>(nd)[6];
        const float b0[6] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y};
        const float b1[6] = {q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
        const float b2[6] = {q3.x, q3.y, q3.z, q3.w, q4.x, q4.y};
        const float b3[6] = {q4.z, q4.w, q5.x, q5.y, q5.z, q5.w};
        const double tf2 = anyhit ? tmax : fmin(tmax, best.t);
        double t0, t1, t2, t3;
        const bool h0 = slab_f32box(b0, b0 + 3, oo, inv, tmin, tf2, &t0);                          // slot 0 always exists
        const bool h1 = slab_f32box(b1, b1 + 3, oo, inv, tmin, tf2, &t1);                          // slot 1 always exists
        const bool h2 = e.z != FJ_NO_CHILD && slab_f32box(b2, b2 + 3, oo, inv, tmin, tf2, &t2);
        const bool h3 = e.w != FJ_NO_CHILD && slab_f32box(b3, b3 + 3, oo, inv, tmin, tf2, &t3);
        // near-to-far order is a heuristic only: f32 keys, misses sort last
        float k0 = h0 ? fminf((float) t0, FLT_MAX) : INFINITY, k1 = h1 ? fminf((float) t1, FLT_MAX) : INFINITY;
        float k2 = h2 ? fminf((float) t2, FLT_MAX) : INFINITY, k3 = h3 ? fminf((float) t3, FLT_MAX) : INFINITY;
        uint32_t r0 = e.x, r1 = e.y, r2 = e.z, r3 = e.w;
#define FJ_CSWAP(ka, ra, kb, rb) { const bool sw = kb < ka; const float tk = sw ? ka : kb; const uint32_t tr = sw ? ra : rb; ka = sw ? kb : ka; ra = sw ? rb : ra; kb = tk; rb = tr; }
        FJ_CSWAP(k0, r0, k1, r1) FJ_CSWAP(k2, r2, k3, r3) FJ_CSWAP(k0, r0, k2, r2) FJ_CSWAP(k1, r1, k3, r3) FJ_CSWAP(k1, r1, k2, r2)
#undef FJ_CSWAP
        const int nh = (int) h0 + (int) h1 + (int) h2 + (int) h3;
        if (nh == 0) cur = (sp == 0) ? TRAV_DONE : stk.pop(sp);
        else {
          cur = r0;
          if (nh > 3) stk.push(sp, r3);
          if (nh > 2) stk.push(sp, r2);
          if (nh > 1) stk.push(sp, r1);
        }
      }
    }

    // ---- leaves: FP64 Moller-Trumbore on the pre-gathered triangles
    if (have && (cur & FJ_LEAF_FLAG) && cur != TRAV_DONE) {
      const uint32_t first = (cur & 0x7fffffffu) >> 3;
      const uint32_t cnt = (cur & 7u) + 1;
      bool stop = false;
      const bool is_curve = kCurves && P->type == FJ_PRIMSET_CURVE;
      for (uint32_t k = 0; k < cnt; k++) {
        double t, u = 0, v = 0;
        if (kCount && !(kCurves && is_curve)) lc->prims++;
        if (kCurves && is_curve) {
          // hit record of a curve: u = curve parameter v_hit, v = BLAS slot (attribute fetch).
          // BLAS entries are sub-segments of curves: the full ribbon test of a curve runs
          // once, not once per piece entered (same ray, same instance: same result)
          const size_t sl = first + k;
          const uint32_t cid = P->prim_ids[sl];
          if (cid == last_curve) continue;
          last_curve = cid;
          if (kCount) lc->prims++;
          if (!curve_ray(P->curve_cp + sl * 12, P->curve_width[2 * sl], P->curve_width[2 * sl + 1],
                         (int) P->curve_depth[sl], oo, od, &t, &u)) continue;
          if (!curve_listed_in_cell_of(P, P->curve_cp + sl * 12, oo + t * od)) continue;
          v = (double) sl;
        } else {
          V3 v0, v1, v2;
          load_tri(P->tri_verts, P->tri_verts32, first + k, &v0, &v1, &v2);
          if (kMotion && P->tri_vel) {       // Mesh::ray_intersect: P + time * velocity (src/fj_mesh.cc:252-259)
            const double *w = P->tri_vel + (size_t) (first + k) * 9;
            v0 = v0 + rtime * ld3(w); v1 = v1 + rtime * ld3(w + 3); v2 = v2 + rtime * ld3(w + 6);
          }
          if (!tri_ray(v0, v1, v2, oo, od, &t, &u, &v)) continue;
        }
        if (!(tmin <= t && t <= tmax)) continue;
        const int pid = (int) P->prim_ids[first + k];
        if (t < best.t || (t == best.t && best.inst == ii && pid > best.prim)) {
          best.t = t; best.u = u; best.v = v; best.inst = ii; best.prim = pid;
          if (anyhit) { stop = true; break; }
        }
      }
      if (stop) { pol.finish(idx, best); have = false; cur = TRAV_DONE; }
      else cur = (sp == 0) ? TRAV_DONE : stk.pop(sp);
    }
  }
}

// ------------------------------------------------------------------ k_trace
struct ClosestPolicy {
  const DScene *S;
  const DRay *rays;
  const DPath *paths;
  DHit *hits;
  int default_group;
  __device__ bool fetch(uint32_t i, RayIn *r) const
  {
    const DRay q = rays[i];
    r->o = mk(q.o[0], q.o[1], q.o[2]); r->d = mk(q.d[0], q.d[1], q.d[2]);
    r->tmin = q.tmin; r->tmax = q.tmax;
    r->time = (S->has_motion && paths) ? sample_time(*S, paths[i].uid & 0xfffffu) : 0.;   // fjgpu_trace: time 0
    r->group = paths ? paths[i].group : default_group;
    r->anyhit = false;
    return true;
  }
  __device__ void finish(uint32_t i, const Best &b) const
  {
    DHit h;
    h.t = b.t; h.u = b.u; h.v = b.v; h.inst = b.inst; h.prim = b.prim;
    hits[i] = h;
  }
};

#ifndef FJ_CURVE_MINB
#define FJ_CURVE_MINB 2
#endif
#ifndef FJ_CLOSEST_MINB
#define FJ_CLOSEST_MINB 3
#endif
#ifndef FJ_SHADOW_MINB
#define FJ_SHADOW_MINB 1
#endif
template <bool kCurves, bool kCount, bool kMotion>
__global__ void __launch_bounds__(BLOCK, (kCurves || kMotion) ? FJ_CURVE_MINB : FJ_CLOSEST_MINB) k_trace_closest(DScene S, const DRay *rays, const DPath *paths,
    DHit *hits, uint32_t n, DCounters *cnt, TravTune tune)
{
  __shared__ uint32_t s_stack[FJ_STACK_LDS * BLOCK];
  ClosestPolicy pol;
  pol.S = &S; pol.rays = rays; pol.paths = paths; pol.hits = hits; pol.default_group = S.target_group;
  LocalCounters lc = {0, 0, 0};
  traverse_persistent<kCurves, kCount, kMotion>(S, pol, tune, n, &cnt->trace_head, make_stack(s_stack, S.stack_overflow), &lc);
  if (kCount) {
    flush_counters(cnt, lc.nodes, lc.prims, lc.insts, 0, 0);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&cnt->traced, (unsigned long long) n);
  }
}

// --------------------------------------------------------------- k_gen_camera
// FixedGridSampler::generate_samples (src/fj_fixed_grid_sampler.cc:33-84) with the
// per-tile XorShift streams read from host-built tables (the stream restarts
// for every tile, so draw k is the same number in every tile), then
// Camera::GetRay (src/fj_camera.cc:79-110) with the host-built camera matrix.
template <bool kMovingCamera>
__global__ void __launch_bounds__(BLOCK) k_gen_camera(DScene S, GenParams gp, const TileDesc *tiles,
    const double *jitter_tab, const double *time_tab, double *s_uv, DRay *rays, DPath *paths)
{
  const TileDesc T = tiles[blockIdx.y];
  const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
  const uint32_t ns = (uint32_t) T.nx * (uint32_t) T.ny;
  if (k >= ns) return;
  const int x = (int) (k % (uint32_t) T.nx), y = (int) (k / (uint32_t) T.nx);
  const int xoffset = T.xmin * gp.rate_x - gp.margin_x;
  const int yoffset = T.ymin * gp.rate_y - gp.margin_y;

  double u = (.5 + x + xoffset) * gp.udelta;
  double v = 1 - (.5 + y + yoffset) * gp.vdelta;
  if (gp.jittered) {
    const double u_jitter = jitter_tab[2 * (size_t) k] * gp.jitter;
    const double v_jitter = jitter_tab[2 * (size_t) k + 1] * gp.jitter;
    u += gp.udelta * (u_jitter - .5);
    v += gp.vdelta * (v_jitter - .5);
  }
  const uint32_t slot = T.sample_offset + k;
  s_uv[2 * (size_t) slot] = u;
  s_uv[2 * (size_t) slot + 1] = v;
  (void) time_tab;   // (the same table as S.time_tab)

  // Camera::GetRay (src/fj_camera.cc:79-110): a time-sampled camera is evaluated at the
  // sample's time, a static one uses the host-built matrix
  const double *cam = S.cam_M;
  double cm[12], cmi[12];
  if (kMovingCamera) { xform_at(S.cam_xform, sample_time(S, k), cm, cmi); cam = cm; }
  const V3 target = mk((u - .5) * S.cam_uv_size[0], (v - .5) * S.cam_uv_size[1], -1);
  const V3 tw = xpoint(cam, target);
  const V3 eye = mk(cam[3], cam[7], cam[11]);
  const V3 dir = normalize(tw - eye);

  DRay r;
  r.o[0] = eye.x; r.o[1] = eye.y; r.o[2] = eye.z;
  r.d[0] = dir.x; r.d[1] = dir.y; r.d[2] = dir.z;
  r.tmin = S.cam_znear; r.tmax = S.cam_zfar;
  rays[slot] = r;
  DPath p;
  p.sample = slot;
  p.T[0] = p.T[1] = p.T[2] = 1.f;
  p.cxt = CXT_CAMERA_RAY; p.ddepth = p.rdepth = p.tdepth = 0;
  p.group = S.target_group;
  p.fc[0] = p.fc[1] = p.fc[2] = 1.f;
  p.flags = 0; p.rng = 0; p.uid = ((uint32_t) T.id << 20) + k;
  paths[slot] = p;
}

// -------------------------------------------------------------------- shading

// Texture::Lookup, src/fj_texture.cc:51-78 + MipInput::ReadTile clamp (src/fj_mipmap.cc:153-170)
__device__ void tex_lookup(const DTexture &tex, float u, float v, float out[4])
{
  if (tex.width == 0 || tex.tiles == nullptr) { out[0] = 1.f; out[1] = .63f; out[2] = .63f; out[3] = 1.f; return; }
  const int ts = tex.tilesize;
  const int xnt = tex.width / ts, ynt = tex.height / ts;
  const float tu = u - floorf(u);
  const float tv = v - floorf(v);
  const float su = tu * xnt;
  const float sv = (1 - tv) * ynt;
  int xt = (int) floorf(su), yt = (int) floorf(sv);
  xt = xt < 0 ? 0 : (xt > xnt - 1 ? xnt - 1 : xt);
  yt = yt < 0 ? 0 : (yt > ynt - 1 ? ynt - 1 : yt);
  const int xp = (int) ((su - floorf(su)) * 64);
  const int yp = (int) ((sv - floorf(sv)) * 64);
  if (xp < 0 || xp >= ts || yp < 0 || yp >= ts) { out[0] = out[1] = out[2] = out[3] = 0.f; return; }
  const float *p = tex.tiles + ((size_t) (yt * xnt + xt) * ts * ts + (size_t) (yp * ts + xp)) * tex.nchannels;
  switch (tex.nchannels) {
  case 1: out[0] = out[1] = out[2] = p[0]; out[3] = 1.f; break;
  case 3: out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; out[3] = 1.f; break;
  case 4: out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; out[3] = p[3]; break;
  default: out[0] = out[1] = out[2] = out[3] = 0.f; break;
  }
}

__device__ __forceinline__ V3 faceforward(V3 I, V3 N) { return (dot(I, N) < 0) ? N : mk(-N.x, -N.y, -N.z); }   // src/fj_shading.cc:42-51

__device__ double fresnel(V3 I, V3 N, double ior)   // SlFresnel, src/fj_shading.cc:53-73
{
  double c = -1 * dot(I, N);
  double eta;
  if (c > 0) eta = ior;
  else { eta = 1. / ior; c *= -1; }
  const double k2 = .0;
  const double F0 = ((1. - eta) * (1. - eta) + k2) / ((1. + eta) * (1. + eta) + k2);
  return F0 + (1. - F0) * pow(1. - c, 5.);
}

__device__ __forceinline__ V3 reflect(V3 I, V3 N)   // SlReflect, :90-98
{
  const double c = -1 * dot(I, N);
  return mk(I.x + 2 * c * N.x, I.y + 2 * c * N.y, I.z + 2 * c * N.z);
}

__device__ V3 refract(V3 I, V3 N, double ior)        // SlRefract, :100-138
{
  V3 n;
  double eta;
  double c1 = -1 * dot(I, N);
  if (c1 < 0) { c1 *= -1; eta = 1 / ior; n = mk(-N.x, -N.y, -N.z); }
  else { eta = ior; n = N; }
  const double radicand = 1 - eta * eta * (1 - c1 * c1);
  if (radicand < 0.) return reflect(I, N);
  const double nc = eta * c1 - sqrt(radicand);
  return mk(eta * I.x + nc * n.x, eta * I.y + nc * n.y, eta * I.z + nc * n.z);
}

__device__ __forceinline__ float luminance4(const float c[4]) { return (float) (.298912 * c[0] + .586611 * c[1] + .114478 * c[2]); }

__device__ __forceinline__ float luminance3(const float c[3]) { return (float) (.298912 * c[0] + .586611 * c[1] + .114478 * c[2]); }

// Counter-based RNG contract of the pathtracing path: the reference's
// seeded XorShift (src/fj_random.cc:18-43) with seed = mix(sample uid, path key), four
// warm-up draws, then the two numbers of the diffuse bounce.
__device__ __forceinline__ uint32_t pt_mix(uint32_t uid, uint32_t key)
{
  uint32_t h = uid * 0x9E3779B1u ^ (key + 0x7F4A7C15u) * 0x85EBCA77u;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}

// XorShift of src/fj_random.cc:10-43 (state in registers)
struct XS {
  uint32_t a, b, c, d;
  __device__ __forceinline__ uint32_t next()
  {
    const uint32_t t = a ^ (a << 11);
    a = b; b = c; c = d;
    d = (d ^ (d >> 19)) ^ (t ^ (t >> 8));
    return d;
  }
  __device__ __forceinline__ double f01() { return (double) next() / 4294967295u; }
};

// stream of one (shading event, area light): RNG contract of DESIGN.md 4
__device__ __forceinline__ XS area_stream(uint32_t uid, uint32_t key, int light)
{
  uint32_t seed = pt_mix(pt_mix(uid, key) ^ 0x51ED270Bu, (uint32_t) light);
  XS r;
  r.a = seed = 1812433253U * (seed ^ (seed >> 30)) + 0u;
  r.b = seed = 1812433253U * (seed ^ (seed >> 30)) + 1u;
  r.c = seed = 1812433253U * (seed ^ (seed >> 30)) + 2u;
  r.d = seed = 1812433253U * (seed ^ (seed >> 30)) + 3u;
  for (int i = 0; i < 4; i++) r.next();
  return r;
}

__device__ void pt_draw2(uint32_t uid, uint32_t key, double *x1, double *x2)
{
  uint32_t h = uid * 0x9E3779B1u ^ (key + 0x7F4A7C15u) * 0x85EBCA77u;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  uint32_t st[4];
  uint32_t seed = h;
  for (uint32_t i = 0; i < 4; i++) st[i] = seed = 1812433253U * (seed ^ (seed >> 30)) + i;
  double out[2] = {0, 0};
  for (int i = 0; i < 6; i++) {
    const uint32_t t = st[0] ^ (st[0] << 11);
    st[0] = st[1]; st[1] = st[2]; st[2] = st[3];
    st[3] = (st[3] ^ (st[3] >> 19)) ^ (t ^ (t >> 8));
    if (i >= 4) out[i - 4] = (double) st[3] / 4294967295u;
  }
  *x1 = out[0];
  *x2 = out[1];
}

// SlBumpMapping, src/fj_shading.cc:418-464
__device__ V3 bump_mapping(const DTexture &bump, V3 dPdu, V3 dPdv, float tu, float tv, double amplitude, V3 N)
{
  if (bump.width == 0 || bump.height == 0) return N;
  const float du = (float) (1. / bump.width);
  const float dv = (float) (1. / bump.height);
  float c0[4], c1[4];
  tex_lookup(bump, tu - du, tv, c0);
  tex_lookup(bump, tu + du, tv, c1);
  const float Bu = (luminance4(c0) - luminance4(c1)) / (2 * du);
  tex_lookup(bump, tu, tv - dv, c0);
  tex_lookup(bump, tu, tv + dv, c1);
  const float Bv = (luminance4(c0) - luminance4(c1)) / (2 * dv);
  V3 a = cross(N, dPdu), b = cross(N, dPdv);
  a = mk(a.x * du, a.y * du, a.z * du);
  b = mk(b.x * du, b.y * du, b.z * du);
  const V3 nb = mk(N.x + amplitude * (Bv * a.x - Bu * b.x),
                   N.y + amplitude * (Bv * a.y - Bu * b.y),
                   N.z + amplitude * (Bv * a.z - Bu * b.z));
  return normalize(nb);
}

// wave-aggregated append: one atomic per wave, slots by ballot prefix count.
// `tally` (optional) receives the number of appended entries, also once per wave.
__device__ __forceinline__ uint32_t wave_append(bool want, uint32_t *counter, unsigned long long *tally)
{
  const unsigned long long mask = __ballot(want);
  if (!want) return 0xffffffffu;
  const unsigned lane = __lane_id();
  const unsigned leader = (unsigned) __ffsll((long long) mask) - 1;
  uint32_t base = 0;
  if (lane == leader) {
    base = atomicAdd(counter, (uint32_t) __popcll(mask));
    if (tally) atomicAdd(tally, (unsigned long long) __popcll(mask));
  }
  base = __shfl(base, leader);
  return base + (uint32_t) __popcll(mask & ((1ull << lane) - 1ull));
}

struct ChildRay {
  bool want;
  V3 o, d;
  double tmin, tmax;
  float T[3];
  uint8_t dd, rd, td;
  int group;
  float fc[3];
  uint32_t flags;
};

// `cxt` is uniform per call site (reflect / refract / diffuse children are
// emitted by separate calls), so the per-context ray count is one atomic per wave
__device__ __forceinline__ void emit_child(const ChildRay &c, int cxt, uint32_t sample, uint32_t uid, uint32_t key,
    DRay *next_rays, DPath *next_paths, DCounters *cnt, uint32_t capacity)
{
  const uint32_t slot = wave_append(c.want, &cnt->next_count, &cnt->rays[cxt]);
  if (!c.want) return;
  if (slot >= capacity) { cnt->overflow = 1; return; }
  DRay r;
  r.o[0] = c.o.x; r.o[1] = c.o.y; r.o[2] = c.o.z;
  r.d[0] = c.d.x; r.d[1] = c.d.y; r.d[2] = c.d.z;
  r.tmin = c.tmin; r.tmax = c.tmax;
  next_rays[slot] = r;
  DPath p;
  p.sample = sample;
  p.T[0] = c.T[0]; p.T[1] = c.T[1]; p.T[2] = c.T[2];
  p.cxt = (uint8_t) cxt; p.ddepth = c.dd; p.rdepth = c.rd; p.tdepth = c.td;
  p.group = c.group;
  p.fc[0] = c.fc[0]; p.fc[1] = c.fc[1]; p.fc[2] = c.fc[2];
  p.flags = c.flags; p.rng = key; p.uid = uid;
  next_paths[slot] = p;
}

// trace_surface's SurfaceInput setup + Shader::Evaluate for the device shaders.
// Radiance is accumulated as throughput-weighted terms: every shader term of
// the reference is linear in the radiance returned by its child SlTrace calls,
// so `Cs = local + sum_k w_k * C_child_k` unrolls into per-path products
// (DESIGN.md 6).
template <bool kMotion>
__global__ void __launch_bounds__(BLOCK) k_shade(DScene S, ShadeParams sp, const DRay *rays, const DPath *paths,
    const DHit *hits, uint32_t n, float *s_accum, DRay *next_rays, DPath *next_paths,
    DLightRec *lrecs, DCounters *cnt)
{
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  const bool active = i < n;
  DHit h;
  h.inst = -1;
  if (active) h = hits[i];
  const bool hit = active && h.inst >= 0;

  ChildRay c0, c1, c2;   // reflect, refract, diffuse children (glass: c0 + c1, plastic: c0, pathtracing: any)
  c0.want = c1.want = c2.want = false;
  bool want_light = false;
  DLightRec lr;
  DLightHair lh;
  lr.kind = 0;
  uint32_t sample = 0, rng = 0, uid = 0;

  if (hit) {
    const DRay r = rays[i];
    DPath p = paths[i];
    sample = p.sample;
    rng = p.rng;
    uid = p.uid;
    const DInstance *I = &S.instances[h.inst];
    const DPrimSet *P = &S.primsets[I->primset];
    const V3 ro = mk(r.o[0], r.o[1], r.o[2]), rd = mk(r.d[0], r.d[1], r.d[2]);

    // instance matrices: host-built for static instances, evaluated at the ray's time for
    // time-sampled ones (the traversal did the same, so P and N belong to the same pose)
    const double *IM = I->M, *IMinv = I->Minv;
    double tm[12], tmi[12];
    if (kMotion && I->xform >= 0) {
      xform_at(&S.xforms[I->xform], sample_time(S, p.uid & 0xfffffu), tm, tmi);
      IM = tm; IMinv = tmi;
    }
    const V3 oo = xpoint(IMinv, ro);
    const V3 od = xvector(IMinv, rd);
    V3 N = mk(0, 0, 0);
    float tu = 0.f, tv = 0.f;
    V3 dPdu = mk(0, 0, 0), dPdv = mk(0, 0, 0);
    float Cd[3] = {1.f, 1.f, 1.f};                                // Intersection default
    bool has_uv = false;
    float t0u = 0, t0v = 0, t1u = 0, t1v = 0, t2u = 0, t2v = 0;
    int i0 = 0, i1 = 0, i2 = 0;
    int sg = 0;
    if (P->type == FJ_PRIMSET_CURVE) {
      // --- Curve::ray_intersect attribute part (src/fj_curve.cc:211-229): dPdv = curve
      // derivative at v_hit, Cd = lerp of the end colours; N / uv / dPdu stay zero
      const size_t sl = (size_t) h.v;
      const double vhit = h.u;
      const double *cp = P->curve_cp + sl * 12;
      const V3 c0 = ld3(cp), c1 = ld3(cp + 3), c2 = ld3(cp + 6), c3 = ld3(cp + 9);
      const double uu = 1 - vhit;
      const double da = 2 * uu * uu, db = 4 * uu * vhit, dc = 2 * vhit * vhit;
      dPdv = da * (c1 - c0) + db * (c2 - c1) + dc * (c3 - c2);   // derivative_bezier3, :474-486
      const float tl = (float) vhit;
      const float *cd = P->curve_Cd + sl * 6;
      Cd[0] = (1 - tl) * cd[0] + tl * cd[3];
      Cd[1] = (1 - tl) * cd[1] + tl * cd[4];
      Cd[2] = (1 - tl) * cd[2] + tl * cd[5];
    } else {
      // --- Mesh::ray_intersect attribute part (src/fj_mesh.cc:267-305) in object space
      const int32_t *ix = P->indices + 3 * (size_t) h.prim;
      i0 = ix[0]; i1 = ix[1]; i2 = ix[2];
      V3 n0 = mk(0, 0, 0), n1 = n0, n2 = n0;
      if (P->N) { n0 = ld3(P->N + 3 * (size_t) i0); n1 = ld3(P->N + 3 * (size_t) i1); n2 = ld3(P->N + 3 * (size_t) i2); }
      N = (1 - h.u - h.v) * n0 + h.u * n1 + h.v * n2;            // TriComputeNormal, src/fj_triangle.cc:44-49
      has_uv = P->uv != nullptr;
      if (has_uv) {
        t0u = P->uv[2 * (size_t) i0]; t0v = P->uv[2 * (size_t) i0 + 1];
        t1u = P->uv[2 * (size_t) i1]; t1v = P->uv[2 * (size_t) i1 + 1];
        t2u = P->uv[2 * (size_t) i2]; t2v = P->uv[2 * (size_t) i2 + 1];
        const float tt = (float) (1 - h.u - h.v);                  // f32 barycentric, src/fj_mesh.cc:285
        tu = (float) (tt * t0u + h.u * t1u + h.v * t2u);
        tv = (float) (tt * t0v + h.u * t1v + h.v * t2v);
      }
      sg = P->face_group ? P->face_group[h.prim] : 0;
    }
    V3 Pw = oo + h.t * od;                                        // RayPointAt in object space
    // --- ObjectInstance::RayIntersect back-transform (src/fj_object_instance.cc:231-240)
    Pw = xpoint(IM, Pw);
    N = normalize(xvector(IM, N));
    dPdv = xvector(IM, dPdv);

    // --- shader lookup: ObjectInstance::GetShader (src/fj_object_instance.cc:177-191)
    int sid;
    if (sg < 0 || sg >= I->n_shaders) sid = I->shaders[0];
    else { sid = I->shaders[sg]; if (sid < 0) sid = I->shaders[0]; }

    // pending pow(filter, t_hit) of a refraction child (glass_shader.cc:117-121)
    if (p.flags & 1u) {
      p.T[0] = (float) (p.T[0] * pow((double) p.fc[0], h.t));
      p.T[1] = (float) (p.T[1] * pow((double) p.fc[1], h.t));
      p.T[2] = (float) (p.T[2] * pow((double) p.fc[2], h.t));
    }

    float Cs[3] = {.5f, 1.f, 0.f};   // NO_SHADER_COLOR, src/fj_shading.cc:24
    float Os = 1.f;
    bool add_cs = true;
    const V3 Iw = rd;
    if (sid >= 0) {
      const fj_shader_desc *sh = &S.shaders[sid];
      switch (sh->type) {
      case FJ_SHADER_CONSTANT: {   // constant_shader.cc:72-96
        if (sh->texture >= 0) {
          float ct[4];
          tex_lookup(S.textures[sh->texture], tu, tv, ct);
          Cs[0] = ct[0] * sh->diffuse[0]; Cs[1] = ct[1] * sh->diffuse[1]; Cs[2] = ct[2] * sh->diffuse[2];
        } else { Cs[0] = sh->diffuse[0]; Cs[1] = sh->diffuse[1]; Cs[2] = sh->diffuse[2]; }
        Os = 1.f;
        break;
      }
      case FJ_SHADER_PLASTIC: {    // plastic_shader.cc:101-179
        V3 Nf = faceforward(Iw, N);
        if (sh->bump_map >= 0) {
          if (has_uv) {
            // TriComputeDerivatives (src/fj_triangle.cc:51-74) on the object-space
            // vertices, then the instance's M as a vector transform
            V3 p0 = ld3(P->P + 3 * (size_t) i0), p1 = ld3(P->P + 3 * (size_t) i1), p2 = ld3(P->P + 3 * (size_t) i2);
            if (kMotion && P->velocity) {    // the time-sampled vertices of Mesh::ray_intersect
              const double tm_ = sample_time(S, p.uid & 0xfffffu);
              p0 = p0 + tm_ * ld3(P->velocity + 3 * (size_t) i0);
              p1 = p1 + tm_ * ld3(P->velocity + 3 * (size_t) i1);
              p2 = p2 + tm_ * ld3(P->velocity + 3 * (size_t) i2);
            }
            const V3 dP1 = p1 - p0, dP2 = p2 - p0;
            const float du1 = t1u - t0u, du2 = t2u - t0u, dv1 = t1v - t0v, dv2 = t2v - t0v;
            const float determinant = du1 * dv2 - dv1 * du2;
            if (determinant != 0) {
              const float invdet = (float) (1. / determinant);
              dPdu = ((double) dv2 * dP1 - (double) dv1 * dP2) * (double) invdet;
              dPdv = ((double) (-du2) * dP1 + (double) du1 * dP2) * (double) invdet;
            }
            dPdu = xvector(IM, dPdu);
            dPdv = xvector(IM, dPdv);   // (mesh dPdv is zero until here)
          }
          Nf = bump_mapping(S.textures[sh->bump_map], dPdu, dPdv, tu, tv, (double) sh->bump_amplitude, Nf);
        }
        add_cs = false;
        if (S.n_light_samples > 0) {
          float dm[4] = {1.f, 1.f, 1.f, 1.f};
          if (sh->diffuse_map >= 0) tex_lookup(S.textures[sh->diffuse_map], tu, tv, dm);
          want_light = true;
          lr.P[0] = Pw.x; lr.P[1] = Pw.y; lr.P[2] = Pw.z;
          lr.N[0] = Nf.x; lr.N[1] = Nf.y; lr.N[2] = Nf.z;
          lr.W[0] = p.T[0] * (sh->diffuse[0] * dm[0]);
          lr.W[1] = p.T[1] * (sh->diffuse[1] * dm[1]);
          lr.W[2] = p.T[2] * (sh->diffuse[2] * dm[2]);
          lr.sample = sample;
          lr.group = I->shadow_target;
          lr.kind = 0;
          lr.uid = p.uid; lr.key = p.rng;
          if (lr.W[0] == 0.f && lr.W[1] == 0.f && lr.W[2] == 0.f && !sp.count_all_shadow) want_light = false;
        }
        if (sh->do_reflect && (int) p.rdepth + 1 <= sp.max_reflect_depth) {
          const V3 R = normalize(reflect(Iw, Nf));
          const double Kr = fresnel(Iw, Nf, (double) (1.f / sh->ior));
          c0.want = true;
          c0.o = Pw; c0.d = R; c0.tmin = .001; c0.tmax = 1000;
          c0.T[0] = (float) (Kr * sh->reflect[0]) * p.T[0];
          c0.T[1] = (float) (Kr * sh->reflect[1]) * p.T[1];
          c0.T[2] = (float) (Kr * sh->reflect[2]) * p.T[2];
          c0.dd = p.ddepth; c0.rd = p.rdepth + 1; c0.td = p.tdepth;
          c0.group = I->reflect_target;
          c0.fc[0] = c0.fc[1] = c0.fc[2] = 1.f; c0.flags = 0;
        }
        Os = sh->opacity;
        break;
      }
      case FJ_SHADER_GLASS: {      // glass_shader.cc:88-130 (N is not face-forwarded)
        add_cs = false;
        const double Kr = fresnel(Iw, N, (double) (1.f / sh->ior));
        const double Kt = 1 - Kr;
        if ((int) p.rdepth + 1 <= sp.max_reflect_depth) {
          c0.want = true;
          c0.o = Pw; c0.d = normalize(reflect(Iw, N)); c0.tmin = .0001; c0.tmax = 1000;
          c0.T[0] = (float) Kr * p.T[0]; c0.T[1] = (float) Kr * p.T[1]; c0.T[2] = (float) Kr * p.T[2];
          c0.dd = p.ddepth; c0.rd = p.rdepth + 1; c0.td = p.tdepth;
          c0.group = I->reflect_target;
          c0.fc[0] = c0.fc[1] = c0.fc[2] = 1.f; c0.flags = 0;
        }
        if ((int) p.tdepth + 1 <= sp.max_refract_depth) {
          c1.want = true;
          c1.o = Pw; c1.d = normalize(refract(Iw, N, (double) (1.f / sh->ior))); c1.tmin = .0001; c1.tmax = 1000;
          c1.T[0] = (float) Kt * p.T[0]; c1.T[1] = (float) Kt * p.T[1]; c1.T[2] = (float) Kt * p.T[2];
          c1.dd = p.ddepth; c1.rd = p.rdepth; c1.td = p.tdepth + 1;
          c1.group = I->refract_target;
          const bool filt = sh->do_color_filter && dot(Iw, N) < 0;
          c1.fc[0] = sh->filter_color[0]; c1.fc[1] = sh->filter_color[1]; c1.fc[2] = sh->filter_color[2];
          c1.flags = filt ? 1u : 0u;
        }
        Os = 1.f;
        break;
      }
      case FJ_SHADER_HAIR: {       // hair_shader.cc:87-117: Kajiya-Kay over all light samples
        add_cs = false;
        if (S.n_light_samples > 0) {
          const V3 tangent = normalize(dPdv);
          want_light = true;
          lr.P[0] = Pw.x; lr.P[1] = Pw.y; lr.P[2] = Pw.z;
          lr.N[0] = N.x; lr.N[1] = N.y; lr.N[2] = N.z;     // illuminance axis = in.N (zero for curves)
          lh.aux[0] = tangent.x; lh.aux[1] = tangent.y; lh.aux[2] = tangent.z;
          lh.aux[3] = Iw.x; lh.aux[4] = Iw.y; lh.aux[5] = Iw.z;
          lr.W[0] = p.T[0]; lr.W[1] = p.T[1]; lr.W[2] = p.T[2];
          lh.Cd[0] = Cd[0] * sh->diffuse[0]; lh.Cd[1] = Cd[1] * sh->diffuse[1]; lh.Cd[2] = Cd[2] * sh->diffuse[2]; lh.pad = 0;
          lr.sample = sample;
          lr.group = I->shadow_target;
          lr.kind = 1;
          lr.uid = p.uid; lr.key = p.rng;
        }
        Os = 1.f;
        break;
      }
      case FJ_SHADER_PATHTRACING: {   // pathtracing_shader.cc:125-257 with the counter-based RNG contract
        float Cdm[3] = {Cd[0], Cd[1], Cd[2]};
        V3 Np = N;
        if (sh->diffuse_map >= 0) {
          float dm[4];
          tex_lookup(S.textures[sh->diffuse_map], tu, tv, dm);
          Cdm[0] *= dm[0]; Cdm[1] *= dm[1]; Cdm[2] *= dm[2];
        }
        if (sh->bump_map >= 0) {
          if (has_uv) {
            V3 p0 = ld3(P->P + 3 * (size_t) i0), p1 = ld3(P->P + 3 * (size_t) i1), p2 = ld3(P->P + 3 * (size_t) i2);
            if (kMotion && P->velocity) {    // the time-sampled vertices of Mesh::ray_intersect
              const double tm_ = sample_time(S, p.uid & 0xfffffu);
              p0 = p0 + tm_ * ld3(P->velocity + 3 * (size_t) i0);
              p1 = p1 + tm_ * ld3(P->velocity + 3 * (size_t) i1);
              p2 = p2 + tm_ * ld3(P->velocity + 3 * (size_t) i2);
            }
            const V3 dP1 = p1 - p0, dP2 = p2 - p0;
            const float du1 = t1u - t0u, du2 = t2u - t0u, dv1 = t1v - t0v, dv2 = t2v - t0v;
            const float determinant = du1 * dv2 - dv1 * du2;
            if (determinant != 0) {
              const float invdet = (float) (1. / determinant);
              dPdu = ((double) dv2 * dP1 - (double) dv1 * dP2) * (double) invdet;
              dPdv = ((double) (-du2) * dP1 + (double) du1 * dP2) * (double) invdet;
            }
            dPdu = xvector(IM, dPdu);
            dPdv = xvector(IM, dPdv);
          }
          Np = bump_mapping(S.textures[sh->bump_map], dPdu, dPdv, tu, tv, (double) sh->bump_amplitude, N);
        }
        Cs[0] = sh->emission[0]; Cs[1] = sh->emission[1]; Cs[2] = sh->emission[2];   // Le, added below
        if (luminance3(sh->diffuse) > 0.f && (int) p.ddepth + 1 <= sp.max_diffuse_depth) {   // integrate_diffuse
          const V3 w = Np;
          V3 u = fabs(w.x) > .001 ? mk(0, 1, 0) : mk(1, 0, 0);
          u = normalize(cross(u, w));
          const V3 v = cross(w, u);
          double x1, x2;
          pt_draw2(uid, rng, &x1, &x2);
          const double r1 = 2. * 3.14159265358979323846 * x1;
          const double r2 = x2;
          const double r2sqrt = sqrt(r2);
          const V3 D = normalize(u * cos(r1) * r2sqrt + v * sin(r1) * r2sqrt + w * sqrt(1. - r2));
          const float kd = (float) dot(Np, D);
          c2.want = true;
          c2.o = Pw; c2.d = D; c2.tmin = .001; c2.tmax = 1000;
          c2.T[0] = p.T[0] * (Cdm[0] * kd * sh->diffuse[0]);
          c2.T[1] = p.T[1] * (Cdm[1] * kd * sh->diffuse[1]);
          c2.T[2] = p.T[2] * (Cdm[2] * kd * sh->diffuse[2]);
          c2.dd = p.ddepth + 1; c2.rd = p.rdepth; c2.td = p.tdepth;
          c2.group = I->reflect_target;                              // SlDiffuseContext uses the REFLECT target
          c2.fc[0] = c2.fc[1] = c2.fc[2] = 1.f; c2.flags = 0;
        }
        if (luminance3(sh->reflect) > 0.f && (int) p.rdepth + 1 <= sp.max_reflect_depth) {   // integrate_reflect
          const float kr = (float) fresnel(Iw, Np, 1. / (double) sh->ior);
          c0.want = true;
          c0.o = Pw; c0.d = normalize(reflect(Iw, Np)); c0.tmin = .001; c0.tmax = 1000;
          c0.T[0] = p.T[0] * (kr * sh->reflect[0]); c0.T[1] = p.T[1] * (kr * sh->reflect[1]); c0.T[2] = p.T[2] * (kr * sh->reflect[2]);
          c0.dd = p.ddepth; c0.rd = p.rdepth + 1; c0.td = p.tdepth;
          c0.group = I->reflect_target;
          c0.fc[0] = c0.fc[1] = c0.fc[2] = 1.f; c0.flags = 0;
        }
        if (luminance3(sh->refract) > 0.f && (int) p.tdepth + 1 <= sp.max_refract_depth) {   // integrate_refract
          const double Kr = fresnel(Iw, Np, (double) (1 / sh->ior));      // 1/ior in f32, as in the plugin
          const float kt = (float) (1 - Kr);
          c1.want = true;
          c1.o = Pw; c1.d = normalize(refract(Iw, Np, 1. / (double) sh->ior)); c1.tmin = .0001; c1.tmax = 1000;
          c1.T[0] = p.T[0] * (kt * sh->refract[0]); c1.T[1] = p.T[1] * (kt * sh->refract[1]); c1.T[2] = p.T[2] * (kt * sh->refract[2]);
          c1.dd = p.ddepth; c1.rd = p.rdepth; c1.td = p.tdepth + 1;
          c1.group = I->refract_target;
          const bool filt = sh->do_color_filter && dot(Iw, Np) < 0;
          c1.fc[0] = sh->filter_color[0]; c1.fc[1] = sh->filter_color[1]; c1.fc[2] = sh->filter_color[2];
          c1.flags = filt ? 1u : 0u;
        }
        Os = 1.f;
        break;
      }
      default:
        add_cs = false;
        break;
      }
    }
    Os = (float) clampd(Os, 0, 1);
    float *acc = s_accum + 4 * (size_t) sample;
    if (add_cs) {
      const float r0 = p.T[0] * Cs[0], r1 = p.T[1] * Cs[1], r2 = p.T[2] * Cs[2];
      if (r0 != 0.f) atomicAdd(acc + 0, r0);
      if (r1 != 0.f) atomicAdd(acc + 1, r1);
      if (r2 != 0.f) atomicAdd(acc + 2, r2);
    }
    if (p.cxt == CXT_CAMERA_RAY) acc[3] = Os;   // one camera ray per sample
  }

  // ---- compaction: ballot + prefix count, one atomic per wave and queue
  const uint32_t lslot = wave_append(want_light, &cnt->light_count, nullptr);
  if (want_light) {
    if (lslot < sp.light_capacity) {
      lrecs[lslot] = lr;
      if (lr.kind == 1 && S.lrec_hair) S.lrec_hair[lslot] = lh;
    }
    else cnt->overflow = 1;
  }
  emit_child(c2, CXT_DIFFUSE_RAY, sample, uid, 4 * rng + 1, next_rays, next_paths, cnt, sp.ray_capacity);
  emit_child(c0, CXT_REFLECT_RAY, sample, uid, 4 * rng + 2, next_rays, next_paths, cnt, sp.ray_capacity);
  emit_child(c1, CXT_REFRACT_RAY, sample, uid, 4 * rng + 3, next_rays, next_paths, cnt, sp.ray_capacity);
}

// ------------------------------------------------------------------- k_shadow
// SlIlluminance (src/fj_shading.cc:296-359) in two wavefront stages.
//
// k_shadow_cull: every (light record, light sample) pair.  `lanes` consecutive
// lanes (a power of two <= 64) serve one record and stride over the light
// samples.  A pair is tested against the world AABBs of the shadow group's
// instances: if it misses all of them the light is unoccluded and Kd * Cl goes
// into the per-record sum (butterfly reduction inside the lane segment, one
// lane adds W * sum to the sample); otherwise the ray is appended -- ballot +
// prefix count, one atomic per wave -- to the compact shadow-ray queue.
//
// k_shadow_trace: the compact queue only, so every lane of a wave is
// traversing (no lanes idling while a neighbour walks the BLAS).  Adds
// c * (1 - Os_occluder) to the sample, or c when the ray reaches the light.
#define SQ_CHUNK 256u          // shadow-queue slots a wave reserves per global atomic
#define SQ_INVALID 0xffffffffu // DShadowRay.sample of a padding slot

template <bool kHair, bool kArea>
__global__ void __launch_bounds__(BLOCK) k_shadow_cull(DScene S, ShadowParams sp, const DLightRec *lrecs,
    uint32_t rec_begin, uint32_t rec_end, float *s_accum, DShadowRay *squeue, DCounters *cnt, int count_events)
{
  unsigned long long c_insts = 0, c_shadow = 0;
  const unsigned lane = __lane_id();
  const uint32_t n = rec_end - rec_begin;
  // each wave owns a contiguous slice of the records, so the rays it emits --
  // and the shadow-queue chunks it fills -- stay spatially coherent
  const unsigned long long wave = ((unsigned long long) blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const unsigned long long n_waves = ((unsigned long long) gridDim.x * BLOCK) >> 6;
  const uint32_t recs_per_iter = 64u / sp.lanes;
  const uint32_t slice_begin = (uint32_t) ((unsigned long long) n * wave / n_waves);
  const uint32_t slice_end = (uint32_t) ((unsigned long long) n * (wave + 1) / n_waves);
  // queue space is reserved SQ_CHUNK slots at a time: one atomic per 1024 rays
  // instead of one per wave iteration (a single-address atomic per iteration
  // serialised the whole kernel in L2)
  uint32_t chunk_base = 0, chunk_used = SQ_CHUNK;   // wave-uniform; "used == CHUNK" = no chunk yet

  for (uint32_t r0 = slice_begin; r0 < slice_end; r0 += recs_per_iter) {
    const uint32_t rec = rec_begin + r0 + lane / sp.lanes;
    const uint32_t sub = lane % sp.lanes;
    const bool active = (r0 + lane / sp.lanes) < slice_end;

    float sum[3] = {0.f, 0.f, 0.f};
    uint32_t r_sample = 0;
    float W[3] = {0.f, 0.f, 0.f};
    const uint32_t nl = (uint32_t) S.n_light_samples;
    const uint32_t iters = (nl + sp.lanes - 1) / sp.lanes;     // uniform trip count: ballots stay convergent
    DLightRec R;
    DLightHair H;
    for (int q = 0; q < 6; q++) H.aux[q] = 0;
    H.Cd[0] = H.Cd[1] = H.Cd[2] = 0.f;
    R.uid = R.key = 0; R.kind = 0;
    XS xs = {0, 0, 0, 0};
    V3 Ps = mk(0, 0, 0), axis = Ps, nml_axis = Ps;
    int g_first = 0, g_count = 0;
    const double *g_sbounds = nullptr;     // stays a global-memory pointer (a by-value DGroup lands in scratch)
    double cos_limit = 0;
    if (active) {
      R = lrecs[rec];
      if (kHair && R.kind == 1) H = S.lrec_hair[rec];
      Ps = mk(R.P[0], R.P[1], R.P[2]);
      axis = mk(R.N[0], R.N[1], R.N[2]);
      nml_axis = normalize(axis);
      r_sample = R.sample;
      W[0] = R.W[0]; W[1] = R.W[1]; W[2] = R.W[2];
      cos_limit = (!kHair || R.kind == 0) ? sp.cos_half_pi : sp.cos_pi;
      g_first = S.groups[R.group].first; g_count = S.groups[R.group].count;
      g_sbounds = S.groups[R.group].sbounds;
    }
    for (uint32_t it = 0; it < iters; it++) {
      const uint32_t l = sub + it * sp.lanes;
      bool emit = false;
      DShadowRay q;
      if (active && l < nl) {
        const DLightSample LS = S.light_samples[l];
        V3 Pl = mk(LS.P[0], LS.P[1], LS.P[2]);
        float Cl[3] = {LS.Cl[0], LS.Cl[1], LS.Cl[2]};
        if (kArea && (LS.type == FJ_GRID_LIGHT || LS.type == FJ_SPHERE_LIGHT)) {
          // RectangleLight / SphereLight::get_samples + illuminate with the per-event stream
          const DAreaLight *A = &S.area_lights[LS.light];
          if (LS.ordinal == 0) xs = area_stream(R.uid, R.key, LS.light);
          V3 Nl;
          if (LS.type == FJ_GRID_LIGHT) {
            const double px = xs.f01() - .5;
            const double pz = xs.f01() - .5;
            Pl = xpoint(A->M, mk(px, 0, pz));
            Nl = mk(A->N[0], A->N[1], A->N[2]);
          } else {
            V3 o;
            double dd;
            for (;;) {                                      // XorShift::HollowSphereRand
              o.x = 2 * xs.f01() - 1;
              o.y = 2 * xs.f01() - 1;
              o.z = 2 * xs.f01() - 1;
              dd = dot(o, o);
              if (dd > 0 && dd <= 1) break;
            }
            const double inv = 1. / sqrt(dd);
            const V3 p = mk(o.x * inv, o.y * inv, o.z * inv);
            Pl = xpoint(A->M, p);
            Nl = normalize(xvector(A->M, p));
          }
          const V3 Lq = normalize(mk(Ps.x - Pl.x, Ps.y - Pl.y, Ps.z - Pl.z));
          double dl = dot(Lq, Nl);
          float k;
          if (LS.type == FJ_GRID_LIGHT) {
            dl = A->double_sided ? fabs(dl) : (dl > 0. ? dl : 0.);
            k = (float) (dl * (double) A->sample_intensity);
          } else k = dl > 0 ? A->sample_intensity : 0.f;
          Cl[0] = k * A->color[0]; Cl[1] = k * A->color[1]; Cl[2] = k * A->color[2];
        }
        V3 Ln = mk(Pl.x - Ps.x, Pl.y - Ps.y, Pl.z - Ps.z);
        const double distance = sqrt(dot(Ln, Ln));
        if (distance > 0) {
          const double inv = 1. / distance;
          Ln = mk(Ln.x * inv, Ln.y * inv, Ln.z * inv);
        }
        const double cosangle = dot(nml_axis, Ln);
        const bool lit = !(cosangle < cos_limit) && !(Cl[0] < .0001 && Cl[1] < .0001 && Cl[2] < .0001);
        if (lit) {
          float k[3] = {0.f, 0.f, 0.f};
          if (!kHair || R.kind == 0) {   // plastic_shader.cc:131-137
            float Kd = (float) dot(axis, Ln);
            Kd = (float) (Kd > 0 ? (double) Kd : 0.);
            k[0] = Kd * Cl[0]; k[1] = Kd * Cl[1]; k[2] = Kd * Cl[2];
          } else {                       // hair_shader.cc:184-206 (the plugin's sqrt / pow are the C
                                         // library's double versions on float arguments)
            const V3 tangent = mk(H.aux[0], H.aux[1], H.aux[2]);
            const V3 Iv = mk(H.aux[3], H.aux[4], H.aux[5]);
            const float TL = (float) dot(tangent, Ln);
            const float diff = (float) sqrt((double) (1 - TL * TL));
            const float roughness = .05f;
            const float TI = (float) dot(tangent, Iv);
            float spec = (float) (sqrt((double) (1 - TL * TL)) * sqrt((double) (1 - TI * TI)) + (double) (TL * TI));
            spec = (float) pow((double) spec, (double) (1 / roughness));
            k[0] = (H.Cd[0] * diff + spec) * Cl[0];
            k[1] = (H.Cd[1] * diff + spec) * Cl[1];
            k[2] = (H.Cd[2] * diff + spec) * Cl[2];
          }
          bool maybe_occluded = false;
          if (sp.cast_shadow) {
            c_shadow++;
            // group bounds test + leaf bounds of the instance BVH, as culling
            if (!has_negative_zero(Ln)) {
              const V3 winv = mk(filter_rcp(Ln.x), filter_rcp(Ln.y), filter_rcp(Ln.z));
              const bool plain = plain_dir(Ln);
              for (int gi = 0; gi < g_count; gi++) {
                const DInstance *I = &S.instances[S.group_instances[g_first + gi]];
                if (box_ray_ref_fast(g_count == 1 ? g_sbounds : I->wbounds, Ps, Ln, winv, plain, .0001, distance)) { maybe_occluded = true; break; }
                c_insts++;
              }
            }
          }
          if (maybe_occluded) {
            emit = true;
            q.o[0] = Ps.x; q.o[1] = Ps.y; q.o[2] = Ps.z;
            q.d[0] = Ln.x; q.d[1] = Ln.y; q.d[2] = Ln.z;
            q.tmax = distance;
            q.c[0] = W[0] * k[0]; q.c[1] = W[1] * k[1]; q.c[2] = W[2] * k[2];
            q.sample = r_sample; q.group = R.group; q.tindex = R.uid & 0xfffffu;
          } else {
            sum[0] += k[0]; sum[1] += k[1]; sum[2] += k[2];
          }
        }
      }
      // ---- compaction into the wave's current chunk (ballot + prefix popcount)
      const unsigned long long mask = __ballot(emit);
      const uint32_t need = (uint32_t) __popcll(mask);
      if (need) {
        if (chunk_used + need > SQ_CHUNK) {
          // retire the chunk: mark its unused tail as padding, reserve a new one
          if (chunk_used < SQ_CHUNK)
            for (uint32_t k = chunk_used + lane; k < SQ_CHUNK; k += 64)
              if (chunk_base + k < sp.queue_capacity) squeue[chunk_base + k].sample = SQ_INVALID;
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(&cnt->shadow_count, SQ_CHUNK);
          chunk_base = __shfl(base, 0);
          chunk_used = 0;
        }
        if (emit) {
          const uint32_t slot = chunk_base + chunk_used + (uint32_t) __popcll(mask & ((1ull << lane) - 1ull));
          if (slot < sp.queue_capacity) squeue[slot] = q;
          else cnt->overflow = 1;
        }
        chunk_used += need;
      }
    }
    // butterfly reduction inside the lane segment (all 64 lanes participate)
    for (uint32_t off = sp.lanes >> 1; off > 0; off >>= 1) {
      sum[0] += __shfl_xor(sum[0], (int) off);
      sum[1] += __shfl_xor(sum[1], (int) off);
      sum[2] += __shfl_xor(sum[2], (int) off);
    }
    if (active && sub == 0) {
      float *acc = s_accum + 4 * (size_t) r_sample;
      const float r0v = W[0] * sum[0], r1v = W[1] * sum[1], r2v = W[2] * sum[2];
      if (r0v != 0.f) atomicAdd(acc + 0, r0v);
      if (r1v != 0.f) atomicAdd(acc + 1, r1v);
      if (r2v != 0.f) atomicAdd(acc + 2, r2v);
    }
  }
  // pad the tail of the last chunk
  if (chunk_used < SQ_CHUNK)
    for (uint32_t k = chunk_used + lane; k < SQ_CHUNK; k += 64)
      if (chunk_base + k < sp.queue_capacity) squeue[chunk_base + k].sample = SQ_INVALID;
  flush_counters(cnt, 0, 0, count_events ? c_insts : 0, count_events ? c_shadow : 0, c_shadow);
}

struct ShadowPolicy {
  const DScene *S;
  const DShadowRay *squeue;
  float *s_accum;
  __device__ bool fetch(uint32_t i, RayIn *r) const
  {
    const DShadowRay q = squeue[i];
    if (q.sample == SQ_INVALID) return false;      // padding slot of a partially filled chunk
    r->o = mk(q.o[0], q.o[1], q.o[2]); r->d = mk(q.d[0], q.d[1], q.d[2]);
    r->tmin = .0001; r->tmax = q.tmax;
    r->time = S->has_motion ? sample_time(*S, q.tindex) : 0.;
    r->group = q.group;
    r->anyhit = S->groups[q.group].all_opaque != 0;
    return true;
  }
  __device__ void finish(uint32_t i, const Best &b) const
  {
    float ac = 1.f;
    if (b.inst >= 0) {
      // the occluder's shader runs in shadow context and only its Os is used
      // (src/fj_shading.cc:338-355,548-569): opacity for plastic, 1 otherwise
      float Os = 1.f;
      const DInstance *I = &S->instances[b.inst];
      const DPrimSet *P = &S->primsets[I->primset];
      const int sg = (P->face_group && b.prim >= 0) ? P->face_group[b.prim] : 0;
      int sid;
      if (sg < 0 || sg >= I->n_shaders) sid = I->shaders[0];
      else { sid = I->shaders[sg]; if (sid < 0) sid = I->shaders[0]; }
      if (sid >= 0 && S->shaders[sid].type == FJ_SHADER_PLASTIC) Os = S->shaders[sid].opacity;
      Os = (float) clampd(Os, 0, 1);
      ac = 1 - Os;
    }
    if (ac == 0.f) return;
    const DShadowRay *q = &squeue[i];
    const float r0 = q->c[0] * ac, r1 = q->c[1] * ac, r2 = q->c[2] * ac;
    float *acc = s_accum + 4 * (size_t) q->sample;
    if (r0 != 0.f) atomicAdd(acc + 0, r0);
    if (r1 != 0.f) atomicAdd(acc + 1, r1);
    if (r2 != 0.f) atomicAdd(acc + 2, r2);
  }
};

template <bool kCurves, bool kCount, bool kMotion>
__global__ void __launch_bounds__(BLOCK, (kCurves || kMotion) ? FJ_CURVE_MINB : FJ_SHADOW_MINB) k_shadow_trace(DScene S, const DShadowRay *squeue, float *s_accum,
    DCounters *cnt, TravTune tune)
{
  __shared__ uint32_t s_stack[FJ_STACK_LDS * BLOCK];
  const uint32_t n = cnt->shadow_count;         // written by k_shadow_cull earlier on this stream
  ShadowPolicy pol;
  pol.S = &S; pol.squeue = squeue; pol.s_accum = s_accum;
  LocalCounters lc = {0, 0, 0};
  traverse_persistent<kCurves, kCount, kMotion>(S, pol, tune, n, &cnt->shadow_head, make_stack(s_stack, S.stack_overflow_shadow), &lc);
  if (kCount) {
    flush_counters(cnt, lc.nodes, lc.prims, lc.insts, 0, 0);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&cnt->squeued, (unsigned long long) n);
  }
}

// ---- conservative f32 slab test for the any-hit walk.
// Per (ray, instance) and axis a the entry code keeps  i32 = (float)(1/od_a),
// o32 = (float)(oo_a/od_a)  and a margin  E = 1.5 * 2^-22 * (Bmax_a * |i32| + |o32|),
// Bmax_a >= |any box coordinate| of the primitive set.  For a box plane b (f32):
//   t~ = fmaf(b, i32, -o32)  differs from the exact (b - oo_a)/od_a by at most
//   |b/od| 2^-24 (i32 rounding) + |oo/od| 2^-24 (o32 rounding) + |t~| 2^-24 (fma rounding)
//   <= 2^-23 (Bmax |i32| + |o32|) (1 + 2^-22)  <  E,
// so [min(t~0, t~1) - E, max(t~0, t~1) + E] contains the exact slab interval and the f64
// interval of slab_f32box (whose own error is ~2^-52 relative): whatever the f64 test
// accepts this one accepts -- it can only cull less.  An axis whose E is not a finite
// number below 1e30 (direction component zero or denormal: 1/od = inf, 0 * inf = NaN) is
// given i32 = o32 = 0, E = 1e30: t~ = 0, interval [-1e30, 1e30], it never culls.  With
// E < 1e30 every product is below 2.8e36, so no inf and no NaN can arise in the test.
struct Slab32 { float ix, iy, iz, ox, oy, oz, ex, ey, ez; };
#ifdef FJ_EXP_SLAB_VALIDATE
__device__ unsigned long long g_slab_lost, g_slab_extra, g_slab_tests;
#endif

__device__ __forceinline__ void slab32_axis(double inv, double oo, double bmax_abs, float *i32, float *o32, float *e32)
{
  float i = (float) inv, o = (float) (oo * inv);
  float e = 3.6e-7f * ((float) bmax_abs * 1.0000002f * fabsf(i) + fabsf(o));
  if (!(e < 1e30f)) { i = 0.f; o = 0.f; e = 1e30f; }
  *i32 = i; *o32 = o; *e32 = e;
}

__device__ __forceinline__ Slab32 slab32_setup(V3 oo, V3 inv, const double *bounds)
{
  Slab32 s;
  slab32_axis(inv.x, oo.x, fmax(fabs(bounds[0]), fabs(bounds[3])), &s.ix, &s.ox, &s.ex);
  slab32_axis(inv.y, oo.y, fmax(fabs(bounds[1]), fabs(bounds[4])), &s.iy, &s.oy, &s.ey);
  slab32_axis(inv.z, oo.z, fmax(fabs(bounds[2]), fabs(bounds[5])), &s.iz, &s.oz, &s.ez);
  return s;
}

// box = {min xyz, max xyz}; tmin32 <= tmin and tmax32 >= tmax of the ray
__device__ __forceinline__ bool slab32_test(const float *b, const Slab32 &s, float tmin32, float tmax32)
{
  const float x0 = fmaf(b[0], s.ix, -s.ox), x1 = fmaf(b[3], s.ix, -s.ox);
  const float y0 = fmaf(b[1], s.iy, -s.oy), y1 = fmaf(b[4], s.iy, -s.oy);
  const float z0 = fmaf(b[2], s.iz, -s.oz), z1 = fmaf(b[5], s.iz, -s.oz);
  const float lx = fminf(x0, x1) - s.ex, hx = fmaxf(x0, x1) + s.ex;
  const float ly = fminf(y0, y1) - s.ey, hy = fmaxf(y0, y1) + s.ey;
  const float lz = fminf(z0, z1) - s.ez, hz = fmaxf(z0, z1) + s.ez;
  const float tn = fmaxf(fmaxf(lx, ly), fmaxf(lz, tmin32));
  const float tf = fminf(fminf(hx, hy), fminf(hz, tmax32));
  return tn <= tf;
}

// ---- lean any-hit traversal: shadow rays of scenes in which every possible occluder is
// opaque (Os = 1) and no curve set exists -- the common case and the dominant kernel of
// C1-C3.  Same tests, same order of instances, same result (occluded or not) as
// traverse_persistent with anyhit rays; what is gone is the closest-hit bookkeeping
// (best t/u/v/ids, tie rule, range shrinking) and the world-space ray, which is re-read
// from the queue entry on the rare instance switches.  The point is registers: occupancy
// decides throughput on this latency-bound walk.
template <bool kCount>
__device__ void traverse_anyhit(const DScene &S, const DShadowRay *squeue, float *s_accum, TravTune tune,
    uint32_t n, uint32_t *head, TravStack stk, LocalCounters *lc)
{
  const unsigned lane = __lane_id();
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  bool head_live = true;
  uint32_t next = 0, range_end = 0;
  tune.grab = adaptive_grab(tune.grab, n);
  bool have = false, hit = false;
  uint32_t idx = 0;
  V3 oo = mk(0, 0, 0), od = oo;
#if !defined(FJ_EXP_ANYHIT_F32SLAB) || defined(FJ_EXP_SLAB_VALIDATE)
  V3 inv_keep = oo;
#endif
#ifdef FJ_EXP_ANYHIT_F32SLAB
  Slab32 s32 = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  float tmax32 = 0.f;
  const float tmin32 = 9.9999e-5f;      // <= .0001
#endif
  double tmax = 0;
  int gi = 0, gend = 0;
  const DNode *nodes = nullptr;
  const double *tris = nullptr;
  const float *tris32 = nullptr;
  uint32_t cur = TRAV_DONE;
  int sp = 0;
  const double tmin = .0001;

  for (;;) {
    // ---- refill idle lanes (see traverse_persistent)
    const unsigned long long idle = __ballot(!have);
    if (next >= range_end && head_live && (idle == ~0ull || (unsigned) __popcll(idle) >= TRAV_REFILL)) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(head, (uint32_t) TRAV_GRAB);
      base = __shfl(base, 0);
      if (base >= n) head_live = false;
      else { next = base; range_end = (n - base < TRAV_GRAB) ? n : base + TRAV_GRAB; }
    }
    if (idle == ~0ull || ((unsigned) __popcll(idle) >= TRAV_REFILL && next < range_end)) {
      if (!have) {
        const uint32_t my = next + (uint32_t) __popcll(idle & lt_mask);
        if (my < range_end && squeue[my].sample != SQ_INVALID) {
          have = true; hit = false;
          idx = my;
          const int g = squeue[my].group;
          gi = S.groups[g].first;
          gend = gi + S.groups[g].count;
          cur = TRAV_DONE; sp = 0;
        }
      }
      next += (uint32_t) __popcll(idle);
      if (__ballot(have) == 0ull) {
        if (next >= range_end && !head_live) break;
        continue;
      }
    }

    // ---- between instances: retire the ray or enter the next instance
    if (have && cur == TRAV_DONE) {
      const DShadowRay *q = &squeue[idx];
      bool found = false;
      if (!hit) {
        const V3 o = mk(q->o[0], q->o[1], q->o[2]), d = mk(q->d[0], q->d[1], q->d[2]);
        tmax = q->tmax;
        if (!has_negative_zero(d)) {
          const V3 winv = mk(filter_rcp(d.x), filter_rcp(d.y), filter_rcp(d.z));
          const bool plain = plain_dir(d);
          const DGroup *G = &S.groups[q->group];
          const bool single = G->count == 1;
          while (gi < gend) {
            const DInstance *I = &S.instances[S.group_instances[gi]];
            gi++;
            if (kCount) lc->insts++;
            if (!box_ray_ref_fast(single ? G->sbounds : I->wbounds, o, d, winv, plain, tmin, tmax)) continue;
            oo = xpoint(I->Minv, o);
            od = xvector(I->Minv, d);
            if (has_negative_zero(od)) continue;
            const V3 inv = mk(1. / od.x, 1. / od.y, 1. / od.z);
            const DPrimSet *P = &S.primsets[I->primset];
            if (P->n_prims == 0) continue;
            double tn;
            if (!slab(P->bounds, P->bounds + 3, oo, inv, tmin, tmax, &tn)) continue;
#if !defined(FJ_EXP_ANYHIT_F32SLAB) || defined(FJ_EXP_SLAB_VALIDATE)
            inv_keep = inv;
#endif
#ifdef FJ_EXP_ANYHIT_F32SLAB
            s32 = slab32_setup(oo, inv, P->bounds);
            tmax32 = nextafterf((float) tmax, INFINITY);
#endif
            nodes = P->nodes; tris = P->tri_verts; tris32 = P->tri_verts32;
            cur = P->root; sp = 0;
            found = true;
            break;
          }
        }
      }
      if (!found) {
        if (!hit) {      // reached the light: add c (an opaque occluder adds c * (1 - Os) = 0)
          float *acc = s_accum + 4 * (size_t) q->sample;
          const float r0 = q->c[0], r1 = q->c[1], r2 = q->c[2];
          if (r0 != 0.f) atomicAdd(acc + 0, r0);
          if (r1 != 0.f) atomicAdd(acc + 1, r1);
          if (r2 != 0.f) atomicAdd(acc + 2, r2);
        }
        have = false;
      }
    }

    // ---- inner nodes
    for (int step = 0; step < TRAV_STEPS; step++) {
      const bool inner = have && !(cur & FJ_LEAF_FLAG);
      if (__ballot(inner) == 0ull) break;
      if (inner) {
        const float4 *nd = reinterpret_cast<const float4 *>(&nodes[cur]);
        if (kCount) lc->nodes++;
        const float4 q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3], q4 = nd[4], q5 = nd[5];
        const uint4 e = reinterpret_cast<const uint4 *>(nd)[6];
        const float b0[6] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y};
        const float b1[6] = {q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
        const float b2[6] = {q3.x, q3.y, q3.z, q3.w, q4.x, q4.y};
        const float b3[6] = {q4.z, q4.w, q5.x, q5.y, q5.z, q5.w};
#ifndef FJ_EXP_ANYHIT_F32SLAB
        double t0, t1, t2, t3;
        const bool h0 = slab_f32box(b0, b0 + 3, oo, inv_keep, tmin, tmax, &t0);
        const bool h1 = slab_f32box(b1, b1 + 3, oo, inv_keep, tmin, tmax, &t1);
        const bool h2 = e.z != FJ_NO_CHILD && slab_f32box(b2, b2 + 3, oo, inv_keep, tmin, tmax, &t2);
        const bool h3 = e.w != FJ_NO_CHILD && slab_f32box(b3, b3 + 3, oo, inv_keep, tmin, tmax, &t3);
#else
        const bool h0 = slab32_test(b0, s32, tmin32, tmax32);
        const bool h1 = slab32_test(b1, s32, tmin32, tmax32);
        const bool h2 = e.z != FJ_NO_CHILD && slab32_test(b2, s32, tmin32, tmax32);
        const bool h3 = e.w != FJ_NO_CHILD && slab32_test(b3, s32, tmin32, tmax32);
#ifdef FJ_EXP_SLAB_VALIDATE
        {   // every box the f64 test accepts must be accepted by the f32 test
          double tq;
          const bool g0 = slab_f32box(b0, b0 + 3, oo, inv_keep, tmin, tmax, &tq);
          const bool g1 = slab_f32box(b1, b1 + 3, oo, inv_keep, tmin, tmax, &tq);
          const bool g2 = e.z != FJ_NO_CHILD && slab_f32box(b2, b2 + 3, oo, inv_keep, tmin, tmax, &tq);
          const bool g3 = e.w != FJ_NO_CHILD && slab_f32box(b3, b3 + 3, oo, inv_keep, tmin, tmax, &tq);
          const int lost = (int) (g0 && !h0) + (int) (g1 && !h1) + (int) (g2 && !h2) + (int) (g3 && !h3);
          const int extra = (int) (h0 && !g0) + (int) (h1 && !g1) + (int) (h2 && !g2) + (int) (h3 && !g3);
          if (lost) atomicAdd(&g_slab_lost, (unsigned long long) lost);
          if (extra) atomicAdd(&g_slab_extra, (unsigned long long) extra);
          atomicAdd(&g_slab_tests, (unsigned long long) (2 + (e.z != FJ_NO_CHILD) + (e.w != FJ_NO_CHILD)));
        }
#endif
#endif
        // any hit ends the ray, so the visiting order is free: no distance sort; children
        // are stored by decreasing surface area (the builder), larger ones first
        uint32_t r0 = e.x, r1 = e.y, r2 = e.z, r3 = e.w;
        if (!h2) { r2 = r3; }
        if (!h1) { r1 = r2; r2 = r3; }
        if (!h0) { r0 = r1; r1 = r2; r2 = r3; }
        const int nh = (int) h0 + (int) h1 + (int) h2 + (int) h3;
        if (nh == 0) cur = (sp == 0) ? TRAV_DONE : stk.pop(sp);
        else {
          cur = r0;
          if (nh > 3) stk.push(sp, r3);
          if (nh > 2) stk.push(sp, r2);
          if (nh > 1) stk.push(sp, r1);
        }
      }
    }

    // ---- leaves: the first triangle hit inside [tmin, tmax] ends the ray
    if (have && (cur & FJ_LEAF_FLAG) && cur != TRAV_DONE) {
      const uint32_t first = (cur & 0x7fffffffu) >> 3;
      const uint32_t cnt = (cur & 7u) + 1;
      for (uint32_t k = 0; k < cnt; k++) {
        double t, u, v;
        if (kCount) lc->prims++;
        V3 v0, v1, v2;
        load_tri(tris, tris32, first + k, &v0, &v1, &v2);
        if (!tri_ray(v0, v1, v2, oo, od, &t, &u, &v)) continue;
        if (!(tmin <= t && t <= tmax)) continue;
        hit = true;
        break;
      }
      if (hit) { have = false; cur = TRAV_DONE; }
      else cur = (sp == 0) ? TRAV_DONE : stk.pop(sp);
    }
  }
}

#ifndef FJ_ANYHIT_MINB
#define FJ_ANYHIT_MINB 4
#endif
template <bool kCount>
__global__ void __launch_bounds__(BLOCK, FJ_ANYHIT_MINB) k_shadow_anyhit(DScene S, const DShadowRay *squeue, float *s_accum,
    DCounters *cnt, TravTune tune)
{
  __shared__ uint32_t s_stack[FJ_STACK_LDS * BLOCK];
  const uint32_t n = cnt->shadow_count;
  LocalCounters lc = {0, 0, 0};
  traverse_anyhit<kCount>(S, squeue, s_accum, tune, n, &cnt->shadow_head, make_stack(s_stack, S.stack_overflow_shadow), &lc);
  if (kCount) {
    flush_counters(cnt, lc.nodes, lc.prims, lc.insts, 0, 0);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&cnt->squeued, (unsigned long long) n);
  }
}

// ------------------------------------------------------------------ k_resolve
// reconstruct_image + apply_pixel_filter (src/fj_renderer.cc:939-995) with
// eval_gaussian (src/fj_filter.cc:49-58).  One wave per pixel: the lanes take the
// (rate + 2 margin)^2 samples of the pixel's window in row-major order, so a load
// instruction reads whole window rows (npx_x * 16 B contiguous each) instead of 64
// addresses 128 B apart (the one-lane-per-pixel version spent 17.6 ms per C3 frame on L2
// requests, this one is bandwidth bound).  The weighted sums are reduced in f64 by a
// butterfly and rounded to f32 once; the reference adds the same f64 products into f32
// accumulators one by one -- the difference is the accumulators' rounding (~1e-7).
__global__ void __launch_bounds__(BLOCK) k_resolve(ResolveParams rp, const TileDesc *tiles,
    const double *s_uv, const float *s_accum, float *fb)
{
  const TileDesc T = tiles[blockIdx.y];
  const int tw = T.xmax - T.xmin, th = T.ymax - T.ymin;
  const int k = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);     // pixel of this wave
  if (k >= tw * th) return;
  const unsigned lane = __lane_id();
  const int px = T.xmin + k % tw, py = T.ymin + k / tw;
  const size_t base = (size_t) T.sample_offset + (size_t) (py - T.ymin) * rp.rate_y * T.nx + (size_t) (px - T.xmin) * rp.rate_x;
  const int nwin = rp.npx_x * rp.npx_y;
  double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0, wsum = 0;
  for (int w = (int) lane; w < nwin; w += 64) {
    const int sy = w / rp.npx_x, sx = w - sy * rp.npx_x;
    const size_t s = base + (size_t) sy * T.nx + sx;
    const double2 uv = reinterpret_cast<const double2 *>(s_uv)[s];
    const float4 d = reinterpret_cast<const float4 *>(s_accum)[s];
    const double filtx = rp.xres * uv.x - (px + .5);
    const double filty = rp.yres * (1 - uv.y) - (py + .5);
    const double xx = 2 * filtx / rp.fw;
    const double yy = 2 * filty / rp.fh;
    const double wgt = exp(-2 * (xx * xx + yy * yy));
    acc0 += wgt * (double) d.x;
    acc1 += wgt * (double) d.y;
    acc2 += wgt * (double) d.z;
    acc3 += wgt * (double) d.w;
    wsum += wgt;
  }
  for (int off = 32; off > 0; off >>= 1) {
    acc0 += __shfl_xor(acc0, off);
    acc1 += __shfl_xor(acc1, off);
    acc2 += __shfl_xor(acc2, off);
    acc3 += __shfl_xor(acc3, off);
    wsum += __shfl_xor(wsum, off);
  }
  if (lane == 0) {
    const float inv_sum = 1.f / (float) wsum;
    float4 out;
    out.x = (float) acc0 * inv_sum; out.y = (float) acc1 * inv_sum; out.z = (float) acc2 * inv_sum; out.w = (float) acc3 * inv_sum;
    reinterpret_cast<float4 *>(fb)[(size_t) py * rp.xres + px] = out;
  }
}

// ----------------------------------------------------------- host launchers
// persistent launches: at most PERSIST_BLOCKS_PER_CU resident blocks per CU
#define PERSIST_BLOCKS_PER_CU 4
static unsigned persistent_grid(unsigned long long blocks_needed)
{
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  const unsigned long long cap = (unsigned long long) cus * PERSIST_BLOCKS_PER_CU;
  return (unsigned) (blocks_needed < cap ? (blocks_needed ? blocks_needed : 1) : cap);
}

size_t persistent_threads() { return (size_t) persistent_grid(~0ull) * BLOCK; }

static TravTune trav_tune()
{
  static TravTune t = {0, 0, 0};
  if (t.grab == 0) {
    auto env = [](const char *name, uint32_t dflt) { const char *v = getenv(name); return v ? (uint32_t) atoi(v) : dflt; };
    t.refill = env("FJGPU_TRAV_REFILL", 24);
    t.steps = env("FJGPU_TRAV_STEPS", 3);
    t.grab = env("FJGPU_TRAV_GRAB", 128);
    if (t.refill < 1) t.refill = 1;
    if (t.refill > 64) t.refill = 64;
    if (t.steps < 1) t.steps = 1;
    if (t.grab < 64) t.grab = 64;
  }
  return t;
}

#define LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int) e_; } while (0)

int launch_gen_camera(hipStream_t st, const DScene &S, const GenParams &gp, const TileDesc *d_tiles, int n_tiles,
    uint32_t max_tile_samples, const double *jit, const double *tim, double *s_uv, DRay *rays, DPath *paths)
{
  dim3 grid((max_tile_samples + BLOCK - 1) / BLOCK, n_tiles);
  if (S.cam_xform) hipLaunchKernelGGL(k_gen_camera<true>, grid, dim3(BLOCK), 0, st, S, gp, d_tiles, jit, tim, s_uv, rays, paths);
  else hipLaunchKernelGGL(k_gen_camera<false>, grid, dim3(BLOCK), 0, st, S, gp, d_tiles, jit, tim, s_uv, rays, paths);
  LAUNCH_CHECK();
  return 0;
}

int launch_trace_closest(hipStream_t st, const DScene &S, const DRay *rays, const DPath *paths, DHit *hits,
    uint32_t n, DCounters *cnt, int count_events)
{
  if (n == 0) return 0;
  (void) hipMemsetAsync(&cnt->trace_head, 0, sizeof(uint32_t), st);
  // scenes without curve sets run the lean instantiation (the ribbon test costs registers)
  // (the event counters cost registers and issue slots: counting is its own instantiation)
  const dim3 grid(persistent_grid((n + BLOCK - 1) / BLOCK));
#define FJ_LAUNCH_CLOSEST(CURVES, COUNT, MOTION) hipLaunchKernelGGL((k_trace_closest<CURVES, COUNT, MOTION>), grid, dim3(BLOCK), 0, st, S, rays, paths, hits, n, cnt, trav_tune())
  if (S.has_motion) {      // time-sampled instance transforms: one general instantiation
    if (count_events) FJ_LAUNCH_CLOSEST(true, true, true); else FJ_LAUNCH_CLOSEST(true, false, true);
  } else if (S.has_curves) { if (count_events) FJ_LAUNCH_CLOSEST(true, true, false); else FJ_LAUNCH_CLOSEST(true, false, false); }
  else { if (count_events) FJ_LAUNCH_CLOSEST(false, true, false); else FJ_LAUNCH_CLOSEST(false, false, false); }
#undef FJ_LAUNCH_CLOSEST
  LAUNCH_CHECK();
  return 0;
}

int launch_shade(hipStream_t st, const DScene &S, const ShadeParams &sp, const DRay *rays, const DPath *paths,
    const DHit *hits, uint32_t n, float *s_accum, DRay *next_rays, DPath *next_paths, DLightRec *lrecs, DCounters *cnt)
{
  if (n == 0) return 0;
  if (S.has_motion)
    hipLaunchKernelGGL(k_shade<true>, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, S, sp, rays, paths, hits, n,
        s_accum, next_rays, next_paths, lrecs, cnt);
  else
    hipLaunchKernelGGL(k_shade<false>, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, S, sp, rays, paths, hits, n,
        s_accum, next_rays, next_paths, lrecs, cnt);
  LAUNCH_CHECK();
  return 0;
}

// Shadow work in two steps.  The light loop APPENDS the surviving shadow rays of records
// [b, e) to the queue (cnt->shadow_count keeps growing); the traversal consumes the whole
// queue in ONE launch per batch of tiles (or when the queue would overflow) -- a persistent
// kernel ends with a tail of a few slow rays (~1 ms on C3), so it pays to launch it once for
// the rays of every recursion level instead of once per level.
uint32_t shadow_queue_padding()          // every resident wave may leave one partially filled chunk
{
  return persistent_grid(1ull << 30) * (BLOCK / 64) * SQ_CHUNK;
}

void shadow_queue_reset(hipStream_t st, DCounters *cnt)
{
  (void) hipMemsetAsync(&cnt->shadow_count, 0, 2 * sizeof(uint32_t), st);   // shadow_count + shadow_head
}

int launch_shadow_cull(hipStream_t st, const DScene &S, const ShadowParams &sp, const DLightRec *lrecs, uint32_t b, uint32_t e,
    float *s_accum, DShadowRay *squeue, DCounters *cnt, int count_events)
{
  if (e <= b) return 0;
  const unsigned long long threads = (unsigned long long) (e - b) * sp.lanes;
#define FJ_LAUNCH_CULL(HAIR, AREA) hipLaunchKernelGGL((k_shadow_cull<HAIR, AREA>), dim3(persistent_grid((threads + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, S, sp, lrecs, b, e, s_accum, squeue, cnt, count_events)
  if (S.has_area) FJ_LAUNCH_CULL(true, true);          // general instantiation
  else if (S.has_hair) FJ_LAUNCH_CULL(true, false);
  else FJ_LAUNCH_CULL(false, false);
#undef FJ_LAUNCH_CULL
  LAUNCH_CHECK();
  return 0;
}

int launch_shadow_trace(hipStream_t st, const DScene &S, const DShadowRay *squeue, float *s_accum, DCounters *cnt, int count_events)
{
  if (S.all_opaque && !S.has_curves && !S.has_motion) {
    if (count_events)
      hipLaunchKernelGGL(k_shadow_anyhit<true>, dim3(persistent_grid(1ull << 30)), dim3(BLOCK), 0, st, S, squeue, s_accum, cnt, trav_tune());
    else
      hipLaunchKernelGGL(k_shadow_anyhit<false>, dim3(persistent_grid(1ull << 30)), dim3(BLOCK), 0, st, S, squeue, s_accum, cnt, trav_tune());
  } else {
#define FJ_LAUNCH_SHADOW(CURVES, COUNT, MOTION) hipLaunchKernelGGL((k_shadow_trace<CURVES, COUNT, MOTION>), dim3(persistent_grid(1ull << 30)), dim3(BLOCK), 0, st, S, squeue, s_accum, cnt, trav_tune())
    if (S.has_motion) { if (count_events) FJ_LAUNCH_SHADOW(true, true, true); else FJ_LAUNCH_SHADOW(true, false, true); }
    else if (S.has_curves) { if (count_events) FJ_LAUNCH_SHADOW(true, true, false); else FJ_LAUNCH_SHADOW(true, false, false); }
    else { if (count_events) FJ_LAUNCH_SHADOW(false, true, false); else FJ_LAUNCH_SHADOW(false, false, false); }
#undef FJ_LAUNCH_SHADOW
  }
  LAUNCH_CHECK();
  return 0;
}

int launch_resolve(hipStream_t st, const ResolveParams &rp, const TileDesc *d_tiles, int n_tiles, int max_tile_pixels,
    const double *s_uv, const float *s_accum, float *fb)
{
  dim3 grid((max_tile_pixels + (BLOCK / 64) - 1) / (BLOCK / 64), n_tiles);   // one wave per pixel
  hipLaunchKernelGGL(k_resolve, grid, dim3(BLOCK), 0, st, rp, d_tiles, s_uv, s_accum, fb);
  LAUNCH_CHECK();
#ifdef FJ_EXP_SLAB_VALIDATE
  { unsigned long long v[3] = {0, 0, 0}; (void) hipMemcpyFromSymbol(&v[0], HIP_SYMBOL(g_slab_lost), 8); (void) hipMemcpyFromSymbol(&v[1], HIP_SYMBOL(g_slab_extra), 8); (void) hipMemcpyFromSymbol(&v[2], HIP_SYMBOL(g_slab_tests), 8);
    fprintf(stderr, "slab32 validate: lost %llu extra %llu of %llu box tests\n", v[0], v[1], v[2]); }
#endif
  return 0;
}
