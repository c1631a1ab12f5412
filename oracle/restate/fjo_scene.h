// oracle/restate/fjo_scene.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
// Prepared scene for the CPU restatement: the reference's accelerators
// (uniform grid per mesh, BVH over instances per group) rebuilt from a flat
// fj_scene_desc.
#ifndef FJO_SCENE_H
#define FJO_SCENE_H

#include "fjo_math.h"
#include "fj_scene_desc.h"

#include <atomic>
#include <cstdint>
#include <vector>

namespace fjo {

struct Ray { V3 orig, dir; double tmin, tmax; };

// Intersection, src/fj_intersection.h:21-55
struct Isect {
  V3 P, N;
  Col Cd;
  float u, v;
  V3 dPdu, dPdv;
  int object;            // instance index or -1
  int prim_id;
  int shading_group_id;
  double t_hit;
  Isect() : Cd(1, 1, 1), u(0), v(0), object(-1), prim_id(0), shading_group_id(0), t_hit(REAL_MAX) {}
};

// GridAccelerator, src/fj_grid_accelerator.cc (cell lists stored CSR, in the
// reference's LIFO list order = descending prim id)
struct Grid {
  Box bounds;            // primset bounds + PADDING
  int ncells[3];
  V3 cellsize;
  std::vector<uint32_t> cell_start;   // ncells+1
  std::vector<int32_t> cell_prims;
};

struct PrimSet {
  int type;              // FJ_PRIMSET_*
  const fj_mesh_desc *mesh;
  const fj_curve_desc *curve;
  Box bounds;            // primset's own bounds
  Box acc_bounds;        // Accelerator::bounds_ = bounds + PADDING
  Grid grid;
  std::vector<int8_t> curve_split_depth;
};

struct Instance {
  const fj_instance_desc *d;
  PrimSet *primset;
  Box bounds;            // world bounds (merge_sampled_bounds)
  bool is_static;        // all TRS channels have one sample
  Xfm xfm_static;
};

// BVHAccelerator over a group's instances, src/fj_bvh_accelerator.cc
struct BvhNode { int left, right; Box bounds; int prim_id; };
struct Group {
  std::vector<int> instances;
  Box acc_bounds;        // ObjectSet bounds + PADDING
  std::vector<BvhNode> nodes;
  int root;
};

struct LightSample { int light; V3 P, N; Col color; };

struct Scene {
  const fj_scene_desc *d;
  std::vector<PrimSet> meshes, curves;
  std::vector<Instance> instances;
  std::vector<Group> groups;
  std::vector<LightSample> light_samples;   // deterministic lights; one placeholder per sample of an
                                            // area light (filled per shading event, fjo_render.cc)
  bool has_area_lights = false;
  std::vector<Xfm> light_xfm;               // per light, time 0
  bool lights_deterministic;
};

void BuildScene(const fj_scene_desc *d, Scene *scene);
bool GroupIntersect(const Scene &sc, int group, const Ray &ray, double time, Isect *isect);
void LerpXfm(const fj_xform_desc &x, double time, Xfm *out);

// primitives (fjo_prims.cc)
bool TriRayIntersect(const V3 &v0, const V3 &v1, const V3 &v2, const V3 &orig, const V3 &dir,
    double *t, double *u, double *v);
bool MeshRayIntersect(const fj_mesh_desc &m, int prim_id, const Ray &ray, double time, Isect *isect);
void MeshPrimBounds(const fj_mesh_desc &m, int prim_id, Box *b);
bool MeshBoxIntersect(const fj_mesh_desc &m, int prim_id, const Box &box);
bool CurveRayIntersect(const PrimSet &ps, int prim_id, const Ray &ray, double time, Isect *isect);
void CurvePrimBounds(const fj_curve_desc &c, int prim_id, Box *b);
bool CurveBoxIntersect(const fj_curve_desc &c, int prim_id, const Box &box);
void CurveCacheSplitDepth(PrimSet *ps);

}  // namespace fjo
#endif
