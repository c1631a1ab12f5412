// fjgpu_build.h -- host-side preparation of the device scene (runs once per
// scene, outside the timed region, like the reference's build_accelerators()).
#ifndef FJGPU_BUILD_H
#define FJGPU_BUILD_H

#include "fjgpu_types.h"

#include <memory>
#include <string>
#include <vector>

namespace fjgpu {

// BLAS node array (uninitialised storage filled by the parallel collapse)
struct NodeArray {
  std::unique_ptr<DNode[]> p;
  size_t n = 0;
  const DNode *data() const { return p.get(); }
  size_t size() const { return n; }
};

struct HostPrimSet {
  int type;
  NodeArray nodes;
  std::vector<double> tri_verts;      // mesh: [n][9]  (empty when tri_verts32 is used)
  std::vector<double> tri_vel;        // mesh: [n][9] vertex velocities in leaf order (empty = static)
  std::vector<float> tri_verts32;     // mesh: [n][9]  all coordinates exactly representable in f32
  std::vector<uint32_t> prim_ids;
  std::vector<double> curve_cp;       // curves: [n][12]
  std::vector<double> curve_width;    // [n][2]
  std::vector<float> curve_Cd;        // [n][6]
  std::vector<int8_t> curve_depth;    // [n]
  std::vector<float> curve_capsule;   // [n][8] per BLAS slot: chord A, chord B, reach of the piece it encloses (static curves)
  std::vector<double> curve_vel;      // [n][12] or empty
  uint32_t root;
  double bounds[6];
  double grid_cell[3];
  int grid_n[3];
  int n_prims;
  int max_depth;
  int stack_need;                     // worst-case traversal stack entries (4-wide tree)
  bool device_build;                  // BLAS left to the device builder (fjgpu_lbvh.hip): nodes / triangles empty
  bool f32_exact;                     // every coordinate of the mesh is exactly representable in f32
  const fj_mesh_desc *mesh;
  const fj_curve_desc *curve;
};

struct HostScene {
  std::vector<HostPrimSet> primsets;          // meshes first, then curve sets
  std::vector<DInstance> instances;
  std::vector<DGroup> groups;
  std::vector<int> group_members;             // instance indices of every group, concatenated in group order (DGroup order;
                                              // group g owns n_instances entries from group_member_first[g])
  std::vector<int> group_member_first;
  std::vector<DTNode> group_nodes;            // threaded instance BVH of every group (DGroup.first / count)
  std::vector<fj_shader_desc> shaders;
  std::vector<fj_xform_desc> xforms;          // time-sampled instance transforms (DInstance.xform)
  fj_xform_desc cam_xform;                    // valid when cam_static is false
  bool cam_static;
  std::vector<DLightSample> light_samples;
  std::vector<DAreaLight> area_lights;        // [n_lights] when any rectangle / sphere light exists
  int n_meshes;
  int device_build_quality;                   // fjgpu_lbvh.hip: 0 radix tree, 1 locally-ordered clustering
  int target_group;
  double cam_M[12];
  double cam_fov, cam_znear, cam_zfar;
};

// BLAS build over arbitrary primitive boxes (meshes and curve sets share it)
struct PrimRef { float bmin[3], bmax[3], c[3]; uint32_t id; };
void BuildBlas(HostPrimSet *ps, std::vector<PrimRef> &refs, int max_leaf, float trav_cost);
// Binned-SAH binary tree over K boxes with ONE box per leaf: the top of a device-built BLAS over the clusters its
// agglomeration has formed so far (fjgpu_lbvh.hip).  nodes[i].left / right >= 0: index of another TopNode; < 0: ~(box index).
struct TopNode { int32_t left, right; float box[6]; };
int BuildTopTree(const float *boxes6, int K, std::vector<TopNode> *nodes, int32_t *root);
float RoundDown2(double v);   // f64 -> f32 toward -inf, one more ulp outward
float RoundUp2(double v);

// threaded instance BVH of one group, appended to *out (fjgpu_build.cc)
void BuildGroupNodes(const std::vector<DInstance> &instances, const std::vector<int> &members, std::vector<DTNode> *out);

// returns 0 or a negative FJGPU_E* code with *err set
int BuildHostScene(const fj_scene_desc *desc, HostScene *out, std::string *err, bool device_mesh_build = false);

// reference-exact host math used while flattening (fjgpu_xform.cc)
void MakeTransform(const fj_xform_desc &x, double time, double M[16], double Minv[16]);
void TransformBounds(const double M[16], const double in[6], double out[6]);

// frame tiling / sampling tables (fjgpu_tables.cc)
struct TileRect { int id, xmin, ymin, xmax, ymax; };
void GenerateTiles(const fj_render_desc &r, std::vector<TileRect> *tiles);
void SamplerMargin(const fj_render_desc &r, int margin[2]);
// first n draws of the default-seeded XorShift as f64 in [0,1]
void XorShiftTable(size_t n, std::vector<double> *out);
double CameraUvSizeY(double fov);

}  // namespace fjgpu
#endif
