// fjgpu_lbvh.h -- device-side BLAS build of one mesh (fjgpu_lbvh.hip)
#ifndef FJGPU_LBVH_H
#define FJGPU_LBVH_H

#include <cstring>
#include <string>

#include "fjgpu_types.h"

struct LbvhOut {               // device allocations (hipMalloc) owned by the caller afterwards
  DNode *nodes;
  size_t n_nodes;
  uint32_t root;
  uint32_t *prim_ids;          // [n_faces] original triangle of leaf slot k
  double *tri_verts;           // [n_faces][9] or null
  float *tri_verts32;          // [n_faces][9] or null (f32_exact)
  double *tri_vel;             // [n_faces][9] or null
  int stack_need;
};

// d_P / d_vel / d_idx: the mesh arrays already on the device; bounds: the primitive set's
// padded bounds (Morton grid); quality: 0 = radix tree of the Morton codes, 1 = locally-ordered
// clustering (slower to build, traces like a SAH tree).  Returns 0, or -1 with *err set (nothing left allocated).
int LbvhBuildMesh(const double *d_P, const double *d_vel, const int32_t *d_idx, int n_faces, int n_points,
    const double bounds[6], bool f32_exact, int quality, LbvhOut *out, std::string *err);

#endif
