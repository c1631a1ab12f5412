// oracle/restate/fjo_prims.cc -- TEST INFRASTRUCTURE ONLY (CPU oracle).
// Restatement of the reference's triangle / mesh primitive code.
#include "fjo_scene.h"

namespace fjo {

static const double TRI_EPSILON = 1e-6;   // src/fj_triangle.cc:12

// src/fj_triangle.cc:81-153, DO_NOT_CULL_BACKFACES branch (the only one
// Mesh::ray_intersect uses, src/fj_mesh.cc:262-265).
bool TriRayIntersect(const V3 &v0, const V3 &v1, const V3 &v2, const V3 &orig, const V3 &dir,
    double *t, double *u, double *v)
{
  const V3 edge1 = v1 - v0;
  const V3 edge2 = v2 - v0;
  const V3 pvec = Cross(dir, edge2);
  const double det = Dot(edge1, pvec);
  if (det > -TRI_EPSILON && det < TRI_EPSILON) return false;
  const double inv_det = 1.0 / det;
  const V3 tvec = orig - v0;
  *u = Dot(tvec, pvec) * inv_det;
  if (*u < 0.0 || *u > 1.0) return false;
  const V3 qvec = Cross(tvec, edge1);
  *v = Dot(dir, qvec) * inv_det;
  if (*v < 0.0 || *u + *v > 1.0) return false;
  *t = Dot(edge2, qvec) * inv_det;
  return true;
}

static inline V3 P3(const double *a, int i) { return V3(a[3 * i], a[3 * i + 1], a[3 * i + 2]); }

static void face_points(const fj_mesh_desc &m, int f, V3 *p0, V3 *p1, V3 *p2)
{
  const int32_t *ix = m.indices + 3 * f;
  *p0 = P3(m.P, ix[0]); *p1 = P3(m.P, ix[1]); *p2 = P3(m.P, ix[2]);
}

// src/fj_triangle.cc:51-74 (uv deltas / determinant in f32; products promote to f64)
static void tri_derivatives(const V3 &p0, const V3 &p1, const V3 &p2,
    const float *t0, const float *t1, const float *t2, V3 *dPdu, V3 *dPdv)
{
  const V3 dP1 = p1 - p0;
  const V3 dP2 = p2 - p0;
  const float du1 = t1[0] - t0[0];
  const float du2 = t2[0] - t0[0];
  const float dv1 = t1[1] - t0[1];
  const float dv2 = t2[1] - t0[1];
  const float determinant = du1 * dv2 - dv1 * du2;
  if (determinant == 0) { *dPdu = V3(); *dPdv = V3(); return; }
  const float invdet = 1. / determinant;
  *dPdu = (dv2 * dP1 - dv1 * dP2) * invdet;
  *dPdv = (-du2 * dP1 + du1 * dP2) * invdet;
}

// src/fj_mesh.cc:246-308
bool MeshRayIntersect(const fj_mesh_desc &m, int prim_id, const Ray &ray, double time, Isect *isect)
{
  V3 p0, p1, p2;
  face_points(m, prim_id, &p0, &p1, &p2);
  const int32_t *ix = m.indices + 3 * prim_id;
  if (m.velocity) {
    p0 = p0 + time * P3(m.velocity, ix[0]);
    p1 = p1 + time * P3(m.velocity, ix[1]);
    p2 = p2 + time * P3(m.velocity, ix[2]);
  }
  double u, v, t_hit;
  if (!TriRayIntersect(p0, p1, p2, ray.orig, ray.dir, &t_hit, &u, &v)) return false;

  // TriComputeNormal, src/fj_triangle.cc:44-49; missing normals read as zero
  // (bounds-checked getters return Type(), src/fj_mesh.cc:24-47)
  // compute_shading_normal, src/fj_mesh.cc:108-120: per-corner ("vertex") normals win over point normals
  V3 n0, n1, n2;
  if (m.vertex_N) { n0 = P3(m.vertex_N, 3 * prim_id); n1 = P3(m.vertex_N, 3 * prim_id + 1); n2 = P3(m.vertex_N, 3 * prim_id + 2); }
  else if (m.N) { n0 = P3(m.N, ix[0]); n1 = P3(m.N, ix[1]); n2 = P3(m.N, ix[2]); }
  isect->N = (1 - u - v) * n0 + u * n1 + v * n2;

  if (m.uv) {
    const float *t0 = m.uv + 2 * ix[0], *t1 = m.uv + 2 * ix[1], *t2 = m.uv + 2 * ix[2];
    const float t = 1 - u - v;                       // f32 barycentric, :285
    isect->u = t * t0[0] + u * t1[0] + v * t2[0];    // f32*f32 + f64*f32 ... -> f32
    isect->v = t * t0[1] + u * t1[1] + v * t2[1];
    tri_derivatives(p0, p1, p2, t0, t1, t2, &isect->dPdu, &isect->dPdv);
  } else {
    isect->u = 0; isect->v = 0;
    isect->dPdu = V3(); isect->dPdv = V3();
  }
  isect->P = ray.orig + t_hit * ray.dir;
  isect->object = -1;
  isect->prim_id = prim_id;
  isect->shading_group_id = m.face_group ? m.face_group[prim_id] : 0;
  isect->t_hit = t_hit;
  return true;
}

// src/fj_mesh.cc:410-430 + src/fj_triangle.cc:24-33
void MeshPrimBounds(const fj_mesh_desc &m, int prim_id, Box *b)
{
  V3 p0, p1, p2;
  face_points(m, prim_id, &p0, &p1, &p2);
  b->ReverseInfinite();
  b->AddPoint(p0); b->AddPoint(p1); b->AddPoint(p2);
  if (m.velocity) {
    const int32_t *ix = m.indices + 3 * prim_id;
    b->AddPoint(p0 + P3(m.velocity, ix[0]));
    b->AddPoint(p1 + P3(m.velocity, ix[1]));
    b->AddPoint(p2 + P3(m.velocity, ix[2]));
  }
}

// src/fj_mesh.cc:317-340,401-408: recursion depth 0 -> eight velocity segments,
// each an AABB-vs-AABB test (the SAT TriBoxIntersect is compiled out).
bool MeshBoxIntersect(const fj_mesh_desc &m, int prim_id, const Box &box)
{
  V3 p[3], vel[3];
  face_points(m, prim_id, &p[0], &p[1], &p[2]);
  const int32_t *ix = m.indices + 3 * prim_id;
  if (m.velocity) for (int k = 0; k < 3; k++) vel[k] = P3(m.velocity, ix[k]);
  const int N_STEPS = 8;
  V3 step[3];
  for (int k = 0; k < 3; k++) step[k] = vel[k] / N_STEPS;
  for (int i = 0; i < N_STEPS; i++) {
    Box seg;
    seg.ReverseInfinite();
    V3 q[3];
    for (int k = 0; k < 3; k++) { q[k] = p[k] + i * step[k]; seg.AddPoint(q[k]); }
    for (int k = 0; k < 3; k++) seg.AddPoint(q[k] + step[k]);
    if (BoxBoxIntersect(seg, box)) return true;
  }
  return false;
}

}  // namespace fjo
