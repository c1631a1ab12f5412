// fjgpu_dev_curve.h -- Bezier ribbon test (Curve::ray_intersect) and the reference grid's cell-listing predicate.
// Part of the kernels translation unit: included by fjgpu_kernels.hip only (device code,
// compiled with -ffp-contract=off; see the header of that file).
#ifndef FJGPU_DEV_CURVE_H
#define FJGPU_DEV_CURVE_H

#ifdef FJ_PHASE_STATS
__device__ unsigned long long g_curve_stat[8];
#define FJ_CURVE_STAT(i, v) atomicAdd(&g_curve_stat[i], (unsigned long long) (v))
#else
#define FJ_CURVE_STAT(i, v) do { } while (0)
#endif
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

// Ray space of one (ray, instance): origin at the ray origin, +z along the ray
// (compute_world_to_ray_matrix, src/fj_curve.cc:268-295: dst = rotate * translate), rebuilt in every
// Curve::ray_intersect call like the reference does.
#ifndef FJ_CURVE_CACHE_LEVEL
#define FJ_CURVE_CACHE_LEVEL 1            // 0: no cached node of the subdivision
#endif
#ifndef FJ_CURVE_SKIP_PASSED
#define FJ_CURVE_SKIP_PASSED 1
#endif
// (measured and dropped in rounds 1-3, kept as profiles/r04_curve_dropped_experiments.patch: the frame kept in 12 doubles of LDS per lane instead of
// rebuilt per test; the whole curve's ray-space box tested in the leaf phase before the lane takes the curve along; the second stage shared by
// the wave, curve_ray_coop)
#define FJ_RAYSPACE_DOUBLES (FJ_CURVE_CACHE_LEVEL > 0 ? 14 : 1)
struct RayFrame { V3 r0, r1, r2; double m03, m13, m23, ray_scale; };
__device__ __forceinline__ RayFrame make_ray_frame(V3 oo, V3 od)
{
  RayFrame f;
  // nml_ray.dir = ray.dir / |ray.dir|  (Vector /= Real  ==  *= 1./s)
  f.ray_scale = sqrt(dot(od, od));
  const double sinv = 1. / f.ray_scale;
  const V3 nd = od * sinv;
  const double lx = nd.x, ly = nd.y, lz = nd.z;
  const double d = sqrt(lx * lx + lz * lz);
  const double d_inv = 1. / d;
  f.r0 = mk(lz * d_inv, 0, -lx * d_inv);
  f.r1 = mk(-lx * ly * d_inv, d, -ly * lz * d_inv);
  f.r2 = mk(lx, ly, lz);
  const double nox = -oo.x, noy = -oo.y, noz = -oo.z;
  f.m03 = f.r0.x * nox + f.r0.y * noy + f.r0.z * noz;
  f.m13 = f.r1.x * nox + f.r1.y * noy + f.r1.z * noz;
  f.m23 = f.r2.x * nox + f.r2.y * noy + f.r2.z * noz;
  return f;
}
struct RaySpace {
  double *lds;          // s_rayspace + threadIdx.x: the cached node of the subdivision (curve_ray<true>), or null
  V3 oo, od;            // the ray in the instance's space: the frame is rebuilt from it per test (cheaper than 12 doubles of LDS per lane)
  __device__ __forceinline__ RayFrame frame() const { return make_ray_frame(oo, od); }
};

// The curve (moved to the ray's time) in that space
__device__ __forceinline__ Bz curve_to_ray_space(const FJ_GLOBAL double *cpw, const FJ_GLOBAL double *velw, double time, double w0, double w1,
    const RaySpace &rsp, double *ray_scale_out)
{
  const RayFrame f = rsp.frame();
  const V3 r0 = f.r0, r1 = f.r1, r2 = f.r2;
  const double m03 = f.m03, m13 = f.m13, m23 = f.m23;
  Bz root;
  V3 p[4];
  for (int k = 0; k < 4; k++) {
    V3 q = ld3(cpw + 3 * k);
    if (velw) q = q + time * ld3(velw + 3 * k);       // time_sample, src/fj_curve.cc:392-397
    p[k] = mk(r0.x * q.x + r0.y * q.y + r0.z * q.z + m03,
              r1.x * q.x + r1.y * q.y + r1.z * q.z + m13,
              r2.x * q.x + r2.y * q.y + r2.z * q.z + m23);
  }
  root.c0 = p[0]; root.c1 = p[1]; root.c2 = p[2]; root.c3 = p[3];
  root.w0 = w0; root.w1 = w1;
  *ray_scale_out = f.ray_scale;
  return root;
}

// converge_bezier3's entry test (get_bezier3_bounds: control points +- max radius) against
// the ray axis
__device__ __forceinline__ bool bz_misses_ray(const Bz &b)
{
  const double radius = .5 * dmax(b.w0, b.w1);
  const double mnx = dmin(dmin(dmin(b.c0.x, b.c1.x), b.c2.x), b.c3.x) - radius;
  const double mxx = dmax(dmax(dmax(b.c0.x, b.c1.x), b.c2.x), b.c3.x) + radius;
  const double mny = dmin(dmin(dmin(b.c0.y, b.c1.y), b.c2.y), b.c3.y) - radius;
  const double mxy = dmax(dmax(dmax(b.c0.y, b.c1.y), b.c2.y), b.c3.y) + radius;
  const double mxz = dmax(dmax(dmax(b.c0.z, b.c1.z), b.c2.z), b.c3.z) + radius;
  return mnx >= radius || mxx <= -radius || mny >= radius || mxy <= -radius || mxz <= 1e-6;
}

// before any ribbon arithmetic: is the ray (as a line, object space) within `reach` of the chord
// of the PIECE this BLAS slot stands for?  (fjgpu_curve_build.cc: a conservative capsule; a ray
// outside it cannot hit the curve within this piece.)  Line-to-segment distance: the segment
// parameter of the closest pair clamped to [0, 1] -- the distance is a convex function of it --
// then the distance of that point to the line; 1e-6 of slack covers the arithmetic.
__device__ __forceinline__ bool capsule_may_hit(const FJ_GLOBAL float *cap, V3 oo, V3 od)
{
  const double reach = (double) cap[6];
  if (!(reach < 1e30)) return true;
  const V3 A = mk((double) cap[0], (double) cap[1], (double) cap[2]);
  const V3 u = mk((double) cap[3] - A.x, (double) cap[4] - A.y, (double) cap[5] - A.z);
  const V3 w0 = A - oo;
  const double a = dot(u, u), b = dot(u, od), c = dot(od, od), d = dot(u, w0), e = dot(od, w0);
  const double D = a * c - b * b;
  double sc = 0;
  if (D > 1e-12 * a * c) { sc = (b * e - c * d) * filter_rcp(D); sc = sc < 0 ? 0 : (sc > 1 ? 1 : sc); }
  const V3 S = w0 + sc * u;
  const double tt = dot(S, od) * filter_rcp(c);
  const V3 q = S - tt * od;
  return dot(q, q) <= reach * reach * 1.000001;
}

// the same test on a capsule record already in registers (two 16-byte loads: A xyz, B x | B yz, reach, the curve's id)
__device__ __forceinline__ bool capsule_may_hit_v(fj_v4f c0, fj_v4f c1, V3 oo, V3 od)
{
  const double reach = (double) c1.z;
  if (!(reach < 1e30)) return true;
  const V3 A = mk((double) c0.x, (double) c0.y, (double) c0.z);
  const V3 u = mk((double) c0.w - A.x, (double) c1.x - A.y, (double) c1.y - A.z);
  const V3 w0 = A - oo;
  const double a = dot(u, u), b = dot(u, od), c = dot(od, od), d = dot(u, w0), e = dot(od, w0);
  const double D = a * c - b * b;
  double sc = 0;
  if (D > 1e-12 * a * c) { sc = (b * e - c * d) * filter_rcp(D); sc = sc < 0 ? 0 : (sc > 1 ? 1 : sc); }
  const V3 S = w0 + sc * u;
  const double tt = dot(S, od) * filter_rcp(c);
  const V3 q = S - tt * od;
  return dot(q, q) <= reach * reach * 1.000001;
}

// kCache = false: no LDS behind rsp (the any-hit walk of curve scenes, fjgpu_dev_anyhit_curves.h): every leaf walk starts at the root
template <bool kCache = true>
__device__ bool curve_ray(const FJ_GLOBAL double *cpw, const FJ_GLOBAL double *velw, double time, double w0, double w1, int depth, const RaySpace &rsp, double *t_out, double *v_out)
{
  double ray_scale;
  const Bz root = curve_to_ray_space(cpw, velw, time, w0, w1, rsp, &ray_scale);
  double best_z = DBL_MAX, best_v = DBL_MAX;
  bool any = false;
  const uint32_t nleaf = 1u << depth;
  // One node of the subdivision tree -- the level FJ_CURVE_CACHE_LEVEL ancestor of the current
  // leaf -- is kept in LDS: the leaves (and pruned subtrees) below it start from there instead
  // of from the root.  Same splits, same operands: only the repetition is gone.  (C5, level of
  // the cached node 1 / 2 / 3: 5.19 / 5.26 / 5.46 s per frame, 5.41 s without.)
  const int CL = FJ_CURVE_CACHE_LEVEL;
  const bool use_cache = kCache && CL > 0 && depth > CL;
  double *cache = kCache ? rsp.lds : nullptr;
  uint32_t cached = 0xffffffffu;           // which level-CL node the cache holds
  uint32_t j = 0;
  // levels 0..passed of the walk towards j are known to pass their bounds test: the ancestors a leaf walk shares with the one
  // before it passed THERE -- a failed test skips the whole span below it, so the next walk starts beyond that span.  Same splits, same
  // operands; only the repetition of a test whose outcome is known is gone (FJ_CURVE_SKIP_PASSED: C5 any-hit walk 1130 -> 1074 ms).
  int passed = -1;
  while (j < nleaf) {
    FJ_CURVE_STAT(0, 1);                    // leaf walks (outer iterations)
    Bz b = root;
    double v0 = 0, vn = 1;
    int L0 = 0;
    const uint32_t j_before = j;
    const uint32_t pre = use_cache ? j >> (depth - CL) : 0u;
    if (use_cache && cached == pre) {
      b.c0 = mk(cache[0 * BLOCK], cache[1 * BLOCK], cache[2 * BLOCK]);
      b.c1 = mk(cache[3 * BLOCK], cache[4 * BLOCK], cache[5 * BLOCK]);
      b.c2 = mk(cache[6 * BLOCK], cache[7 * BLOCK], cache[8 * BLOCK]);
      b.c3 = mk(cache[9 * BLOCK], cache[10 * BLOCK], cache[11 * BLOCK]);
      b.w0 = cache[12 * BLOCK]; b.w1 = cache[13 * BLOCK];
      v0 = (double) pre * (1. / (1 << CL));        // the interval of a level-CL node: exact dyadic numbers,
      vn = (double) (pre + 1) * (1. / (1 << CL));  // as the repeated (v0 + vn) * .5 gives them
      L0 = CL;
    }
    bool pruned = false;
    for (int L = L0;; L++) {
      FJ_CURVE_STAT(1, 1);                  // nodes visited (inner iterations)
      // (a node taken from the cache passed this test when it was stored)
      if (!(L0 == CL && L == CL && use_cache) && !(FJ_CURVE_SKIP_PASSED && L <= passed) && bz_misses_ray(b)) {
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
      if (use_cache && L + 1 == CL) {
        // the next iteration tests this node's bounds; if it fails the whole span is skipped
        // and the entry is never asked for
        cache[0 * BLOCK] = b.c0.x; cache[1 * BLOCK] = b.c0.y; cache[2 * BLOCK] = b.c0.z;
        cache[3 * BLOCK] = b.c1.x; cache[4 * BLOCK] = b.c1.y; cache[5 * BLOCK] = b.c1.z;
        cache[6 * BLOCK] = b.c2.x; cache[7 * BLOCK] = b.c2.y; cache[8 * BLOCK] = b.c2.z;
        cache[9 * BLOCK] = b.c3.x; cache[10 * BLOCK] = b.c3.y; cache[11 * BLOCK] = b.c3.z;
        cache[12 * BLOCK] = b.w0; cache[13 * BLOCK] = b.w1;
        cached = pre;
      }
    }
    if (FJ_CURVE_SKIP_PASSED) passed = depth - 32 + (int) __clz((int) (j_before ^ (pruned ? j : j + 1u)));     // common leading bits of the two leaf indices
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
// The same predicate for a curve with vertex velocities: box_bezier3_recursive
// (src/fj_curve.cc:428-462) splits the curve at shutter open and at shutter close alike, gives
// each half the velocity "close half - open half", and at depth 0 tests the box of the four
// control points and of the four points moved by that velocity (box_bezier3, N_STEPS = 1).
// Inner levels are pruned with the hull of both polygons widened by 1e-12 (relative): the
// leaf's "cp + (end - cp)" differs from "end" by rounding only.
__device__ bool curve_listed_in_cell_of_moving(const DPrimSet *P, const FJ_GLOBAL double *cpw, const FJ_GLOBAL double *velw, V3 hitp)
{
  double cmin[3], cmax[3];
  const double hp[3] = {hitp.x, hitp.y, hitp.z};
  for (int a = 0; a < 3; a++) {
    int c = (int) floor((hp[a] - P->bounds[a]) / P->grid_cell[a]);
    c = c < 0 ? 0 : (c > P->grid_n[a] - 1 ? P->grid_n[a] - 1 : c);
    cmin[a] = P->bounds[a] + (double) c * P->grid_cell[a];
    cmax[a] = cmin[a] + P->grid_cell[a];
    if (hp[a] < cmin[a] || cmax[a] < hp[a]) return false;
  }
  const V3 r0 = ld3(cpw), r1 = ld3(cpw + 3), r2 = ld3(cpw + 6), r3 = ld3(cpw + 9);
  const V3 w0 = ld3(velw), w1 = ld3(velw + 3), w2 = ld3(velw + 6), w3 = ld3(velw + 9);
  const uint32_t depth = 5, nleaf = 32;
  uint32_t j = 0;
  while (j < nleaf) {
    V3 c0 = r0, c1 = r1, c2 = r2, c3 = r3;      // control points at shutter open
    V3 v0 = w0, v1 = w1, v2 = w2, v3 = w3;      // velocity of this (sub)curve
    bool pruned = false;
    for (uint32_t L = 0;; L++) {
      // the curve at shutter close: time_sample(&end, 1) = cp + 1 * vel; also the moved
      // points of box_bezier3 (cp + vel / N_STEPS, N_STEPS = 1)
      const V3 m0 = c0 + 1. * v0, m1 = c1 + 1. * v1, m2 = c2 + 1. * v2, m3 = c3 + 1. * v3;
      double mn[3] = {dmin(dmin(dmin(c0.x, c1.x), c2.x), c3.x), dmin(dmin(dmin(c0.y, c1.y), c2.y), c3.y), dmin(dmin(dmin(c0.z, c1.z), c2.z), c3.z)};
      double mx[3] = {dmax(dmax(dmax(c0.x, c1.x), c2.x), c3.x), dmax(dmax(dmax(c0.y, c1.y), c2.y), c3.y), dmax(dmax(dmax(c0.z, c1.z), c2.z), c3.z)};
      const double en[3] = {dmin(dmin(dmin(m0.x, m1.x), m2.x), m3.x), dmin(dmin(dmin(m0.y, m1.y), m2.y), m3.y), dmin(dmin(dmin(m0.z, m1.z), m2.z), m3.z)};
      const double ex[3] = {dmax(dmax(dmax(m0.x, m1.x), m2.x), m3.x), dmax(dmax(dmax(m0.y, m1.y), m2.y), m3.y), dmax(dmax(dmax(m0.z, m1.z), m2.z), m3.z)};
      for (int a = 0; a < 3; a++) { mn[a] = dmin(mn[a], en[a]); mx[a] = dmax(mx[a], ex[a]); }
      if (L < depth) {                     // inner level: pruning only, slightly widened
        for (int a = 0; a < 3; a++) { const double pad = 1e-12 * (fabs(mn[a]) + fabs(mx[a])) + 1e-300; mn[a] -= pad; mx[a] += pad; }
      }
      const bool overlap = !(mx[0] < cmin[0] || mn[0] > cmax[0] || mx[1] < cmin[1] || mn[1] > cmax[1] || mx[2] < cmin[2] || mn[2] > cmax[2]);
      if (!overlap) {
        const uint32_t span = 1u << (depth - L);
        j = ((j / span) + 1) * span;
        pruned = true;
        break;
      }
      if (L == depth) return true;
      // split_bezier3 of the open and of the close curve; keep the half selected by j and
      // give it the velocity "close half - open half"
      Bz b, eb;
      b.c0 = c0; b.c1 = c1; b.c2 = c2; b.c3 = c3; b.w0 = b.w1 = 0;
      eb.c0 = m0; eb.c1 = m1; eb.c2 = m2; eb.c3 = m3; eb.w0 = eb.w1 = 0;
      const V3 midP = bez_eval(b, .5), emidP = bez_eval(eb, .5);
      const V3 midCP = mid_point(c1, c2), emidCP = mid_point(m1, m2);
      V3 e0, e1, e2, e3;
      if (((j >> (depth - L - 1)) & 1u) == 0) {
        const V3 l1 = mid_point(c0, c1), el1 = mid_point(m0, m1);
        const V3 l2 = mid_point(l1, midCP), el2 = mid_point(el1, emidCP);
        c1 = l1; c2 = l2; c3 = midP;
        e0 = m0; e1 = el1; e2 = el2; e3 = emidP;
      } else {
        const V3 q2 = mid_point(c3, c2), eq2 = mid_point(m3, m2);
        const V3 q1 = mid_point(q2, midCP), eq1 = mid_point(eq2, emidCP);
        c0 = midP; c1 = q1; c2 = q2;
        e0 = emidP; e1 = eq1; e2 = eq2; e3 = m3;
      }
      v0 = e0 - c0; v1 = e1 - c1; v2 = e2 - c2; v3 = e3 - c3;
    }
    if (!pruned) j++;
  }
  return false;
}

__device__ bool curve_listed_in_cell_of(const DPrimSet *P, const FJ_GLOBAL double *cpw, V3 hitp)
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

#endif
