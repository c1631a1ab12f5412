// oracle/restate/fjo_curve.cc -- TEST INFRASTRUCTURE ONLY (CPU oracle).
// Restatement of the reference's cubic Bezier ribbon primitive,
// src/fj_curve.cc (Nakamaru-Ono recursive subdivision in ray space).
#include "fjo_scene.h"

#include <cmath>

namespace fjo {

namespace {

struct Bez {
  V3 cp[4];
  V3 vel[4];
  double width[2];
};

inline V3 P3(const double *a, int i) { return V3(a[3 * i], a[3 * i + 1], a[3 * i + 2]); }

void get_bezier3(const fj_curve_desc &c, int prim, Bez *b)        // :548-567
{
  const int i0 = c.indices[prim];
  for (int k = 0; k < 4; k++) {
    b->cp[k] = P3(c.P, i0 + k);
    b->vel[k] = c.velocity ? P3(c.velocity, i0 + k) : V3();
  }
  b->width[0] = c.width[i0];
  b->width[1] = c.width[i0 + 3];
}

inline double max_radius(const Bez &b) { return .5 * Max(b.width[0], b.width[1]); }          // :526-529
inline double width_at(const Bez &b, double t) { return (1 - t) * b.width[0] + t * b.width[1]; }   // Lerp, :531-534

void bezier_bounds(const Bez &b, Box *bounds)                     // :536-546
{
  bounds->ReverseInfinite();
  for (int i = 0; i < 4; i++) bounds->AddPoint(b.cp[i]);
  bounds->Expand(max_radius(b));
}

V3 eval_bezier3(const V3 *cp, double t)                           // :464-472
{
  const double u = 1 - t;
  const double a = u * u * u;
  const double b = 3 * u * u * t;
  const double c = 3 * u * t * t;
  const double d = t * t * t;
  return a * cp[0] + b * cp[1] + c * cp[2] + d * cp[3];
}

V3 derivative_bezier3(const V3 *cp, double t)                     // :474-486
{
  const double u = 1 - t;
  const double a = 2 * u * u;
  const double b = 4 * u * t;
  const double c = 2 * t * t;
  return a * (cp[1] - cp[0]) + b * (cp[2] - cp[1]) + c * (cp[3] - cp[2]);
}

inline V3 mid_point(const V3 &a, const V3 &b) { return (a + b) * .5; }

void split_bezier3(const Bez &b, Bez *l, Bez *r)                  // :488-508
{
  const V3 midP = eval_bezier3(b.cp, .5);
  const V3 midCP = mid_point(b.cp[1], b.cp[2]);
  l->cp[0] = b.cp[0];
  l->cp[1] = mid_point(b.cp[0], b.cp[1]);
  l->cp[2] = mid_point(l->cp[1], midCP);
  l->cp[3] = midP;
  r->cp[3] = b.cp[3];
  r->cp[2] = mid_point(b.cp[3], b.cp[2]);
  r->cp[1] = mid_point(r->cp[2], midCP);
  r->cp[0] = midP;
  l->width[0] = b.width[0];
  l->width[1] = (b.width[0] + b.width[1]) * .5;
  r->width[0] = l->width[1];
  r->width[1] = b.width[1];
}

inline double dot_xy(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y; }

// :300-390
bool converge_bezier3(const Bez &bz, double v0, double vn, int depth, double *v_hit, double *P_hit)
{
  const V3 *cp = bz.cp;
  const double radius = max_radius(bz);
  Box bounds;
  bezier_bounds(bz, &bounds);
  if (bounds.min.x >= radius || bounds.max.x <= -radius ||
      bounds.min.y >= radius || bounds.max.y <= -radius ||
      bounds.min.z >= *P_hit || bounds.max.z <= 1e-6)
    return false;

  if (depth == 0) {
    const V3 dir = cp[3] - cp[0];
    V3 dP0 = cp[1] - cp[0];
    if (dot_xy(dir, dP0) < 0) dP0 = dP0 * -1;
    if (-1 * dot_xy(dP0, cp[0]) < 0) return false;
    V3 dPn = cp[3] - cp[2];
    if (dot_xy(dir, dPn) < 0) dPn = dPn * -1;
    if (dot_xy(dPn, cp[3]) < 0) return false;

    double w = dir.x * dir.x + dir.y * dir.y;
    if (std::abs(w) < 1e-6) return false;
    w = -(cp[0].x * dir.x + cp[0].y * dir.y) / w;
    w = Clamp(w, 0, 1);
    const double v = v0 * (1 - w) + vn * w;
    const double radius_w = .5 * width_at(bz, w);
    const V3 vP = eval_bezier3(cp, w);
    if (vP.x * vP.x + vP.y * vP.y >= radius_w * radius_w) return false;
    if (vP.z <= 1e-6 || *P_hit < vP.z) return false;
    *P_hit = vP.z;
    *v_hit = v;
    return true;
  }
  const double vm = (v0 + vn) * .5;
  Bez l, r;
  split_bezier3(bz, &l, &r);
  double v_left = REAL_MAX, v_right = REAL_MAX, t_left = REAL_MAX, t_right = REAL_MAX;
  const bool hit_left = converge_bezier3(l, v0, vm, depth - 1, &v_left, &t_left);
  const bool hit_right = converge_bezier3(r, vm, vn, depth - 1, &v_right, &t_right);
  if (hit_left || hit_right) {
    if (t_left < t_right) { *P_hit = t_left; *v_hit = v_left; }
    else { *P_hit = t_right; *v_hit = v_right; }
  }
  return hit_left || hit_right;
}

void time_sample(Bez *b, double time)                             // :392-397
{
  for (int i = 0; i < 4; i++) b->cp[i] = b->cp[i] + time * b->vel[i];
}

// compute_world_to_ray_matrix, :268-295 (MatMultiply(rotate, translate))
void world_to_ray(const Ray &ray, Mat *dst)
{
  const double ox = ray.orig.x, oy = ray.orig.y, oz = ray.orig.z;
  const double lx = ray.dir.x, ly = ray.dir.y, lz = ray.dir.z;
  const double d = std::sqrt(lx * lx + lz * lz);
  const double d_inv = 1. / d;
  Mat translate, rotate;
  const double t[16] = {1, 0, 0, -ox, 0, 1, 0, -oy, 0, 0, 1, -oz, 0, 0, 0, 1};
  const double r[16] = {lz * d_inv, 0, -lx * d_inv, 0, -lx * ly * d_inv, d, -ly * lz * d_inv, 0, lx, ly, lz, 0, 0, 0, 0, 1};
  for (int i = 0; i < 16; i++) { translate.e[i] = t[i]; rotate.e[i] = r[i]; }
  MatMultiply(dst, rotate, translate);
}

bool box_bezier3(const Box &box, const Bez &b)                    // :399-426 (N_STEPS = 1)
{
  Box seg(b.cp[0], b.cp[1]);     // Box(P0, P1) orders min/max per axis
  for (int k = 0; k < 3; k++) {
    const double a0 = b.cp[0][k], a1 = b.cp[1][k];
    if (a0 < a1) { seg.min[k] = a0; seg.max[k] = a1; } else { seg.min[k] = a1; seg.max[k] = a0; }
  }
  seg.AddPoint(b.cp[2]);
  seg.AddPoint(b.cp[3]);
  for (int i = 0; i < 4; i++) seg.AddPoint(b.cp[i] + b.vel[i] / 1);
  return BoxBoxIntersect(seg, box);
}

bool box_bezier3_recursive(const Box &box, const Bez &b, int depth)   // :428-462
{
  if (depth == 0) return box_bezier3(box, b);
  Bez l, r;
  split_bezier3(b, &l, &r);
  {
    Bez end = b;
    time_sample(&end, 1);
    Bez el, er;
    split_bezier3(end, &el, &er);
    for (int i = 0; i < 4; i++) { l.vel[i] = el.cp[i] - l.cp[i]; r.vel[i] = er.cp[i] - r.cp[i]; }
  }
  if (box_bezier3_recursive(box, l, depth - 1)) return true;
  if (box_bezier3_recursive(box, r, depth - 1)) return true;
  return false;
}

int split_depth_limit(const V3 *cp, double epsilon)               // :510-524
{
  const int N = 4;
  double L0 = -1.;
  for (int i = 0; i < N - 2; i++) {
    const double x_val = std::fabs(cp[i].x - 2 * cp[i + 1].x + cp[i + 2].x);
    const double y_val = std::fabs(cp[i].y - 2 * cp[i + 1].y + cp[i + 2].y);
    L0 = Max(L0, Max(x_val, y_val));
  }
  return (int) (std::log(std::sqrt(2.) * N * (N - 1) * L0 / (8. * epsilon)) / std::log(4.));
}

}  // namespace

void CurveCacheSplitDepth(PrimSet *ps)                            // :168-185
{
  const fj_curve_desc &c = *ps->curve;
  ps->curve_split_depth.resize(c.n_curves);
  for (int i = 0; i < c.n_curves; i++) {
    Bez b;
    get_bezier3(c, i, &b);
    int depth = split_depth_limit(b.cp, 2 * max_radius(b) / 20.);
    depth = (int) Clamp(depth, 1, 5);
    ps->curve_split_depth[i] = (int8_t) depth;
  }
}

// Curve::ray_intersect, :187-232
bool CurveRayIntersect(const PrimSet &ps, int prim_id, const Ray &ray, double time, Isect *isect)
{
  const fj_curve_desc &c = *ps.curve;
  const double ray_scale = Length(ray.dir);
  Ray nml = ray;
  nml.dir = ray.dir / ray_scale;
  Bez b;
  get_bezier3(c, prim_id, &b);
  const int depth = ps.curve_split_depth[prim_id];
  time_sample(&b, time);
  Mat w2r;
  world_to_ray(nml, &w2r);
  for (int i = 0; i < 4; i++) b.cp[i] = MatTransformPoint(w2r, b.cp[i]);
  double ttmp = REAL_MAX, v_hit = REAL_MAX;
  if (!converge_bezier3(b, 0, 1, depth, &v_hit, &ttmp)) return false;
  isect->t_hit = ttmp / ray_scale;
  isect->P = ray.orig + isect->t_hit * ray.dir;
  Bez orig;
  get_bezier3(c, prim_id, &orig);
  time_sample(&orig, time);
  isect->dPdv = derivative_bezier3(orig.cp, v_hit);
  const int i0 = c.indices[prim_id], i1 = i0 + 3;
  const float t = (float) v_hit;                                   // Lerp(Color, Color, float)
  const float *c0 = c.Cd + 3 * i0, *c1 = c.Cd + 3 * i1;
  isect->Cd = c.Cd ? Col((1 - t) * c0[0] + t * c1[0], (1 - t) * c0[1] + t * c1[1], (1 - t) * c0[2] + t * c1[2]) : Col();
  // N, uv, dPdu keep the Intersection defaults of the caller's scratch object;
  // object / prim_id / shading_group_id are NOT written by Curve::ray_intersect
  isect->N = V3();
  isect->u = isect->v = 0;
  isect->dPdu = V3();
  isect->object = -1;
  isect->prim_id = 0;
  isect->shading_group_id = 0;
  return true;
}

// Curve::get_primitive_bounds, :244-257
void CurvePrimBounds(const fj_curve_desc &c, int prim_id, Box *bounds)
{
  Bez b;
  get_bezier3(c, prim_id, &b);
  bezier_bounds(b, bounds);
  time_sample(&b, 1);
  Box close;
  bezier_bounds(b, &close);
  bounds->AddBox(close);
}

// Curve::box_intersect, :234-242
bool CurveBoxIntersect(const fj_curve_desc &c, int prim_id, const Box &box)
{
  Bez b;
  get_bezier3(c, prim_id, &b);
  return box_bezier3_recursive(box, b, 5);
}

}  // namespace fjo
