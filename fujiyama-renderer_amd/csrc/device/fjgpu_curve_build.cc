// fjgpu_curve_build.cc -- BLAS over cubic Bezier curve primitives (config 5).
//
// Leaf payload, pre-gathered in BLAS order: 4 control points (12 f64), the two end
// widths, the two end colours and the cached split depth of
// Curve::cache_split_depth (reference src/fj_curve.cc:168-185,510-524; log / sqrt are
// evaluated here on the host with libm so the depth equals the reference's).
// Culling boxes are Curve::get_primitive_bounds (:244-257): control points +- the
// curve's max radius.
#include "fjgpu_build.h"

#include <cstdlib>
#include <cstring>
#include <vector>
#include "fjgpu.h"

#include <algorithm>
#include <cfloat>
#include <cmath>

namespace fjgpu {

// largest ribbon radius within piece `sgm` of `S` equal parametric pieces (+ rounding slack)
static double PieceRadius(double w0, double w1, int sgm, int S)
{
  const double va = (double) sgm / S, vb = (double) (sgm + 1) / S;
  const double ra = .5 * ((1 - va) * w0 + va * w1), rb = .5 * ((1 - vb) * w0 + vb * w1);
  return std::max(ra, rb) * (1 + 1e-9) + 1e-15;
}

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
  ps->device_build = false;
  ps->f32_exact = false;
  ps->mesh = nullptr;
  ps->curve = &c;
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

  // BLAS references are SUB-SEGMENTS of the curves: a fur strand is long, thin and diagonal,
  // and the box of the whole cubic is mostly empty (measured on C5: 15.8 ribbon tests per ray
  // with one box per curve).  Each curve is cut into 2^k parametric pieces by de Casteljau
  // splits; a piece's box is the hull of ITS control points + the ribbon radius; every piece
  // refers to the same curve, whose full test runs once per ray (the traversal remembers the
  // curve it tested last), so the result is the reference's whichever piece was entered.
  int seg_depth = 4;      // (with per-piece radii: C5 3.06 s at 2, 2.63 s at 3, 2.58 s at 4; at 3 waves per SIMD 2.56 / 2.14 / 2.06 / 2.14 s at 2 / 3 / 4 / 5 -- and twice the BLAS slots each step)
  if (const char *e = getenv("FJGPU_CURVE_SEGDEPTH")) seg_depth = std::max(0, std::min(5, atoi(e)));
  const int S = 1 << seg_depth;
  std::vector<PrimRef> refs((size_t) c.n_curves * S);
  for (int i = 0; i < c.n_curves; i++) {
    const int i0 = c.indices[i];
    if (i0 < 0 || i0 + 3 >= c.n_points) { *err = "curve index out of range"; return FJGPU_EINVAL; }
    const double w0 = c.width[i0], w1 = c.width[i0 + 3];
    // pieces by repeated midpoint subdivision (control polygons; hull property).  With vertex
    // velocities the same subdivision of the end-of-shutter curve (P + velocity) bounds the
    // piece at time 1; positions are linear in time, so the two hulls bound the whole sweep.
    auto subdivide = [&](const double *cp12, std::vector<double> &pieces) {
      pieces.assign((size_t) S * 12, 0.);
      for (int k = 0; k < 12; k++) pieces[k] = cp12[k];
      for (int d = 0, cnt = 1; d < seg_depth; d++, cnt *= 2)
        for (int q = cnt - 1; q >= 0; q--) {
          double b[12], l[12], r[12];
          for (int k = 0; k < 12; k++) b[k] = pieces[(size_t) q * 12 + k];
          for (int a = 0; a < 3; a++) {
            const double p0 = b[a], p1 = b[3 + a], p2 = b[6 + a], p3 = b[9 + a];
            const double q0 = .5 * (p0 + p1), q1 = .5 * (p1 + p2), q2 = .5 * (p2 + p3);
            const double r0 = .5 * (q0 + q1), r1 = .5 * (q1 + q2);
            const double m = .5 * (r0 + r1);
            l[a] = p0; l[3 + a] = q0; l[6 + a] = r0; l[9 + a] = m;
            r[a] = m; r[3 + a] = r1; r[6 + a] = q2; r[9 + a] = p3;
          }
          for (int k = 0; k < 12; k++) { pieces[(size_t) (2 * q) * 12 + k] = l[k]; pieces[(size_t) (2 * q + 1) * 12 + k] = r[k]; }
        }
    };
    std::vector<double> pieces, pieces_end;
    subdivide(&c.P[3 * i0], pieces);
    if (c.velocity) {
      double endcp[12];
      for (int k = 0; k < 12; k++) endcp[k] = c.P[3 * i0 + k] + c.velocity[3 * i0 + k];
      subdivide(endcp, pieces_end);
    }
    for (int sgm = 0; sgm < S; sgm++) {
      // the ribbon's radius within this piece: the width is linear in the curve parameter
      // (split_bezier3 halves it, src/fj_curve.cc:488-508) and the test prunes and accepts with the
      // widths of the sub-segment at hand, so a piece needs the larger of ITS two end radii, not the
      // curve's (fur tapers from .003 to .0001: the tip pieces are 4-30 times thinner)
      const double piece_radius = PieceRadius(w0, w1, sgm, S);
      double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
      for (int k = 0; k < 4; k++)
        for (int a = 0; a < 3; a++) {
          const double p = pieces[(size_t) sgm * 12 + 3 * k + a];
          mn[a] = std::min(mn[a], p);
          mx[a] = std::max(mx[a], p);
          if (c.velocity) {
            const double pe = pieces_end[(size_t) sgm * 12 + 3 * k + a];
            mn[a] = std::min(mn[a], pe);
            mx[a] = std::max(mx[a], pe);
          }
        }
      PrimRef &r = refs[(size_t) i * S + sgm];
      for (int a = 0; a < 3; a++) {
        // the ribbon test accepts points within `radius` of the curve in ray space; pad a
        // little more (and for the rounding of the subdivision) so the cull is never the
        // tighter test
        const double pad = piece_radius * 1.0000001 + 1e-12 + 1e-9 * (std::fabs(mn[a]) + std::fabs(mx[a]));
        r.bmin[a] = RoundDown2(mn[a] - pad);
        r.bmax[a] = RoundUp2(mx[a] + pad);
        r.c[a] = (float) (.5 * (mn[a] + mx[a]));
      }
      r.id = (uint32_t) i * (uint32_t) S + (uint32_t) sgm;      // curve and piece (split again after the build)
    }
  }
  // a ribbon test costs tens of node steps: ONE curve per leaf (the traversal's two-stage
  // ribbon test relies on it), splits almost always pay
  const int leaf = 1;
  float tc = .05f;
  if (const char *e = getenv("FJGPU_CURVE_TRAVCOST")) tc = (float) atof(e);
  if ((uint64_t) c.n_curves * (uint64_t) S >= (1ull << 32)) { *err = "too many curves"; return FJGPU_EINVAL; }
  BuildBlas(ps, refs, leaf, tc);
  const int n = ps->n_prims;
  // Per BLAS slot, the capsule of ITS piece: every point of the piece lies in the hull of the
  // piece's control points, hence within max(dist(p1, AB), dist(p2, AB)) of the chord AB =
  // (p0, p3) (distance to a segment is convex: its maximum over a hull is at a vertex); a ribbon
  // hit puts the ray within the ribbon radius of a curve point.  A ray farther than
  // `reach` from AB cannot hit the curve WITHIN this piece, and a hit in another piece is
  // found when that piece's box is entered: the walk skips the ribbon test (whose first stage,
  // the whole curve's box in ray space, lets 80 % of the candidates through).  Static curves
  // only (reach = inf otherwise).  A and B are stored as f32; their rounding goes into reach.
  std::vector<int> piece_of(n);
  for (int s = 0; s < n; s++) { piece_of[s] = (int) (ps->prim_ids[s] % (uint32_t) S); ps->prim_ids[s] /= (uint32_t) S; }
  ps->curve_capsule.assign((size_t) n * 8, 0.f);
  for (int s = 0; s < n; s++) {
    float *cap = &ps->curve_capsule[(size_t) s * 8];
    const int i0 = c.indices[ps->prim_ids[s]];
    // (word 7: the curve's id, the bits of prim_ids[s] -- the walk learns from ONE record whether this piece belongs to the curve it tested last)
    { const uint32_t cid = ps->prim_ids[s]; std::memcpy(&cap[7], &cid, sizeof(cid)); }
    if (c.velocity) { cap[6] = INFINITY; continue; }
    double b[12];
    for (int k = 0; k < 12; k++) b[k] = c.P[3 * i0 + k];
    for (int d = seg_depth - 1; d >= 0; d--) {            // the same de Casteljau splits as above, one path
      const bool right = ((piece_of[s] >> d) & 1) != 0;
      double o[12];
      for (int a = 0; a < 3; a++) {
        const double p0 = b[a], p1 = b[3 + a], p2 = b[6 + a], p3 = b[9 + a];
        const double q0 = .5 * (p0 + p1), q1 = .5 * (p1 + p2), q2 = .5 * (p2 + p3);
        const double r0 = .5 * (q0 + q1), r1 = .5 * (q1 + q2);
        const double m = .5 * (r0 + r1);
        if (!right) { o[a] = p0; o[3 + a] = q0; o[6 + a] = r0; o[9 + a] = m; }
        else { o[a] = m; o[3 + a] = r1; o[6 + a] = q2; o[9 + a] = p3; }
      }
      for (int k = 0; k < 12; k++) b[k] = o[k];
    }
    const double *A = b, *B = b + 9;
    auto dist_to_chord = [&](const double *p) {
      double u[3], w[3], uu = 0, uw = 0;
      for (int a = 0; a < 3; a++) { u[a] = B[a] - A[a]; w[a] = p[a] - A[a]; uu += u[a] * u[a]; uw += u[a] * w[a]; }
      double t = uu > 0 ? uw / uu : 0;
      t = t < 0 ? 0 : (t > 1 ? 1 : t);
      double d2 = 0;
      for (int a = 0; a < 3; a++) { const double q = w[a] - t * u[a]; d2 += q * q; }
      return std::sqrt(d2);
    };
    const double w0 = c.width[i0], w1 = c.width[i0 + 3];
    const double radius = PieceRadius(w0, w1, piece_of[s], S);
    double reach = std::max(dist_to_chord(b + 3), dist_to_chord(b + 6)) + radius;
    double slack = 0, scale = 0;
    for (int a = 0; a < 3; a++) {
      cap[a] = (float) A[a]; cap[3 + a] = (float) B[a];
      slack += std::fabs(A[a] - (double) cap[a]) + std::fabs(B[a] - (double) cap[3 + a]);
      scale = std::max(scale, std::max(std::fabs(A[a]), std::fabs(B[a])));
    }
    reach = reach * 1.000001 + slack + 1e-9 * scale + 1e-12;
    cap[6] = std::nextafter((float) reach, INFINITY);
  }
  ps->curve_cp.resize((size_t) n * 12);
  ps->curve_width.resize((size_t) n * 2);
  ps->curve_Cd.assign((size_t) n * 6, 0.f);
  ps->curve_depth.resize(n);
  if (c.velocity) ps->curve_vel.resize((size_t) n * 12);
  for (int s = 0; s < n; s++) {
    const int i0 = c.indices[ps->prim_ids[s]];
    for (int k = 0; k < 12; k++) ps->curve_cp[(size_t) s * 12 + k] = c.P[3 * i0 + k];
    if (c.velocity) for (int k = 0; k < 12; k++) ps->curve_vel[(size_t) s * 12 + k] = c.velocity[3 * i0 + k];
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
