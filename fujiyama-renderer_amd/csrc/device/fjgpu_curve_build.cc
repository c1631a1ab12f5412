// fjgpu_curve_build.cc -- BLAS over cubic Bezier curve primitives (config 5).
//
// Leaf payload, pre-gathered in BLAS order: 4 control points (12 f64), the two end
// widths, the two end colours and the cached split depth of
// Curve::cache_split_depth (reference src/fj_curve.cc:168-185,510-524; log / sqrt are
// evaluated here on the host with libm so the depth equals the reference's).
// Culling boxes are Curve::get_primitive_bounds (:244-257): control points +- the
// curve's max radius.
#include "fjgpu_build.h"
#include "fjgpu.h"

#include <algorithm>
#include <cfloat>
#include <cmath>

namespace fjgpu {

static int split_depth_limit(const double *cp, double epsilon)
{
  const int N = 4;
  double L0 = -1.;
  for (int i = 0; i < N - 2; i++) {
    const double x_val = std::fabs(cp[3 * i] - 2 * cp[3 * (i + 1)] + cp[3 * (i + 2)]);
    const double y_val = std::fabs(cp[3 * i + 1] - 2 * cp[3 * (i + 1) + 1] + cp[3 * (i + 2) + 1]);
    const double max_val = x_val > y_val ? x_val : y_val;
    L0 = L0 > max_val ? L0 : max_val;
  }
  return (int) (std::log(std::sqrt(2.) * N * (N - 1) * L0 / (8. * epsilon)) / std::log(4.));
}

int BuildCurveSet(const fj_curve_desc &c, HostPrimSet *ps, std::string *err)
{
  ps->type = FJ_PRIMSET_CURVE;
  ps->mesh = nullptr;
  ps->curve = &c;
  if (c.velocity) { *err = "curve velocity (motion blur) is not on the device path yet"; return FJGPU_EUNSUPPORTED; }
  if (!c.width || !c.P || !c.indices) { *err = "curve set without positions / widths / indices"; return FJGPU_EINVAL; }
  const double ACC_PADDING = .0001;
  for (int k = 0; k < 3; k++) { ps->bounds[k] = c.bounds[k] - ACC_PADDING; ps->bounds[3 + k] = c.bounds[3 + k] + ACC_PADDING; }

  // compute_grid_cellsizes, reference src/fj_grid_accelerator.cc:318-332 (host libm pow)
  {
    const double size[3] = {ps->bounds[3] - ps->bounds[0], ps->bounds[4] - ps->bounds[1], ps->bounds[5] - ps->bounds[2]};
    const double max_width = std::max(std::max(size[0], size[1]), size[2]);
    const double cube_root = 3 * std::pow(c.n_curves, 1. / 3);
    const double per_unit = cube_root / max_width;
    for (int a = 0; a < 3; a++) {
      int n = (int) std::floor(size[a] * per_unit + .5);
      n = n < 1 ? 1 : (n > 512 ? 512 : n);
      ps->grid_n[a] = n;
      ps->grid_cell[a] = (ps->bounds[3 + a] - ps->bounds[a]) / n;
    }
  }

  std::vector<PrimRef> refs(c.n_curves);
  for (int i = 0; i < c.n_curves; i++) {
    const int i0 = c.indices[i];
    if (i0 < 0 || i0 + 3 >= c.n_points) { *err = "curve index out of range"; return FJGPU_EINVAL; }
    const double w0 = c.width[i0], w1 = c.width[i0 + 3];
    const double radius = .5 * (w0 > w1 ? w0 : w1);
    double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (int k = 0; k < 4; k++)
      for (int a = 0; a < 3; a++) {
        const double p = c.P[3 * (i0 + k) + a];
        mn[a] = std::min(mn[a], p);
        mx[a] = std::max(mx[a], p);
      }
    PrimRef &r = refs[i];
    for (int a = 0; a < 3; a++) {
      // the ribbon test accepts points within `radius` of the curve in ray space; pad a
      // little more than the reference's box so the cull can never be the tighter test
      const double pad = radius * 1.0000001 + 1e-12;
      r.bmin[a] = RoundDown2(mn[a] - pad);
      r.bmax[a] = RoundUp2(mx[a] + pad);
      r.c[a] = (float) (.5 * (mn[a] + mx[a]));
    }
    r.id = (uint32_t) i;
  }
  BuildBlas(ps, refs);
  const int n = ps->n_prims;
  ps->curve_cp.resize((size_t) n * 12);
  ps->curve_width.resize((size_t) n * 2);
  ps->curve_Cd.assign((size_t) n * 6, 0.f);
  ps->curve_depth.resize(n);
  for (int s = 0; s < n; s++) {
    const int i0 = c.indices[ps->prim_ids[s]];
    for (int k = 0; k < 12; k++) ps->curve_cp[(size_t) s * 12 + k] = c.P[3 * i0 + k];
    const double w0 = c.width[i0], w1 = c.width[i0 + 3];
    ps->curve_width[2 * (size_t) s] = w0;
    ps->curve_width[2 * (size_t) s + 1] = w1;
    if (c.Cd)
      for (int k = 0; k < 3; k++) {
        ps->curve_Cd[(size_t) s * 6 + k] = c.Cd[3 * i0 + k];
        ps->curve_Cd[(size_t) s * 6 + 3 + k] = c.Cd[3 * (i0 + 3) + k];
      }
    const double radius = .5 * (w0 > w1 ? w0 : w1);
    int depth = split_depth_limit(&c.P[3 * i0], 2 * radius / 20.);
    depth = depth < 1 ? 1 : (depth > 5 ? 5 : depth);
    ps->curve_depth[s] = (int8_t) depth;
  }
  return 0;
}

}  // namespace fjgpu
