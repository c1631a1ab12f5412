// fjgpu_lbvh.hip -- BLAS build ON THE DEVICE: a binary tree over Morton-sorted
// triangles -- locally-ordered clustering with a surface-area distance (Meister & Bittner 2018;
// quality 1, the default) or the plain radix tree of the Morton codes (Karras 2012; quality 0) --
// leaves of <= 4 triangles, then the same collapse to 4-wide 128-byte nodes as the host builder.  Replaces build_accelerators()
// (reference src/fj_scene_interface.cc:1161-1202; grid build src/fj_grid_accelerator.cc:69-160)
// when the scene is created with the "device_build" option: 7.2 M triangles in tens of
// milliseconds instead of 0.4 s on the host threads, at the price of a tree that is not
// SAH-optimised (the default stays the host's binned-SAH build, which traces faster).
//
// The result obeys the same contract as the host tree: child boxes are f32
// rounded OUTWARD (+1 ulp) from the f64 vertex bounds, so culling can only skip provable
// misses and closest hits do not depend on the structure.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <float.h>
#include <math.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "fjgpu_lbvh.h"
#include "fjgpu_build.h"

namespace {

#define LB 256
#define LEAFBIT 0x80000000u

struct Box6 { float mn[3], mx[3]; };

__device__ __forceinline__ float down2(double v) { return nextafterf(__double2float_rd(v), -INFINITY); }
__device__ __forceinline__ float up2(double v) { return nextafterf(__double2float_ru(v), INFINITY); }

__device__ __forceinline__ unsigned long long spread21(unsigned long long x)
{
  x &= 0x1fffffull;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

// per triangle: outward f32 box (swept over the shutter when the mesh has velocities) and
// the 63-bit Morton code of its centre inside the mesh bounds
__global__ void __launch_bounds__(LB) k_prim(const double *P, const double *vel, const int32_t *idx, int n,
    double bx, double by, double bz, double sx, double sy, double sz, Box6 *boxes, unsigned long long *keys, uint32_t *vals, int *bad, int n_points)
{
  const int i = blockIdx.x * LB + threadIdx.x;
  if (i >= n) return;
  double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
  for (int k = 0; k < 3; k++) {
    const int p = idx[3 * (size_t) i + k];
    if (p < 0 || p >= n_points) { *bad = 1; return; }
    for (int c = 0; c < 3; c++) {
      const double q = P[3 * (size_t) p + c];
      mn[c] = fmin(mn[c], q); mx[c] = fmax(mx[c], q);
      if (vel) { const double r = q + vel[3 * (size_t) p + c]; mn[c] = fmin(mn[c], r); mx[c] = fmax(mx[c], r); }
    }
  }
  Box6 b;
  for (int c = 0; c < 3; c++) { b.mn[c] = down2(mn[c]); b.mx[c] = up2(mx[c]); }
  boxes[i] = b;
  const double cx = (.5 * (mn[0] + mx[0]) - bx) * sx, cy = (.5 * (mn[1] + mx[1]) - by) * sy, cz = (.5 * (mn[2] + mx[2]) - bz) * sz;
  const double top = 2097151.;
  const unsigned long long qx = (unsigned long long) fmin(fmax(cx * top, 0.), top);
  const unsigned long long qy = (unsigned long long) fmin(fmax(cy * top, 0.), top);
  const unsigned long long qz = (unsigned long long) fmin(fmax(cz * top, 0.), top);
  keys[i] = (spread21(qx) << 2) | (spread21(qy) << 1) | spread21(qz);
  vals[i] = (uint32_t) i;
}

// Binary tree over the n Morton-sorted leaves, whichever way it was formed.  Node ids: leaf k
// (sorted position) = k, inner node m = n + m.
struct BTree {
  uint32_t *left, *right;    // [2n-1], inner entries used
  uint32_t *count;           // [2n-1] leaves below
  Box6 *box;                 // [2n-1]
  uint32_t *parent;          // [2n-1] (radix-tree path only)
  int *flag;                 // [n]    (radix-tree path only)
};

__global__ void __launch_bounds__(LB) k_leaf_nodes(const Box6 *boxes, const uint32_t *vals, int n, BTree T)
{
  const int i = blockIdx.x * LB + threadIdx.x;
  if (i >= n) return;
  T.box[i] = boxes[vals[i]];
  T.count[i] = 1u;
}

__device__ __forceinline__ int delta(const unsigned long long *keys, int n, int i, int j)
{
  if (j < 0 || j >= n) return -1;
  const unsigned long long a = keys[i], b = keys[j];
  if (a == b) return 64 + __clz((unsigned) (i ^ j));
  return __clzll((long long) (a ^ b));
}

// ---- quality 0: binary radix tree (Karras 2012), one thread per inner node
__global__ void __launch_bounds__(LB) k_hierarchy(const unsigned long long *keys, int n, BTree T)
{
  const int i = blockIdx.x * LB + threadIdx.x;
  if (i >= n - 1) return;
  const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) < 0 ? -1 : 1;
  const int dmin = delta(keys, n, i, i - d);
  int lmax = 2;
  while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
    if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta(keys, n, i, j);
  int s = 0;
  for (int t = (l + 1) / 2; ; t = (t + 1) / 2) {
    if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    if (t <= 1) break;
  }
  const int gamma = i + s * d + (d < 0 ? -1 : 0);
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  const uint32_t lc = (lo == gamma) ? (uint32_t) gamma : (uint32_t) (n + gamma);
  const uint32_t rc = (hi == gamma + 1) ? (uint32_t) (gamma + 1) : (uint32_t) (n + gamma + 1);
  T.left[n + i] = lc; T.right[n + i] = rc;
  T.count[n + i] = (uint32_t) (hi - lo + 1);
  T.parent[lc] = (uint32_t) (n + i);
  T.parent[rc] = (uint32_t) (n + i);
  if (i == 0) T.parent[n] = 0xffffffffu;
}

__device__ __forceinline__ Box6 box_union(const Box6 &a, const Box6 &b)
{
  Box6 u;
  for (int c = 0; c < 3; c++) { u.mn[c] = fminf(a.mn[c], b.mn[c]); u.mx[c] = fmaxf(a.mx[c], b.mx[c]); }
  return u;
}

// bottom-up fit: the second thread to reach a node unions its children
__global__ void __launch_bounds__(LB) k_fit(int n, BTree T)
{
  const int i = blockIdx.x * LB + threadIdx.x;
  if (i >= n) return;
  uint32_t p = T.parent[i];
  while (p != 0xffffffffu) {
    __threadfence();
    if (atomicAdd(&T.flag[p - (uint32_t) n], 1) == 0) return;
    __threadfence();
    T.box[p] = box_union(T.box[T.left[p]], T.box[T.right[p]]);
    p = T.parent[p];
  }
}

__device__ __forceinline__ float half_area(const Box6 &b)
{
  const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
  return dx * dy + dy * dz + dz * dx;
}

// ---- quality 1: parallel locally-ordered clustering (Meister & Bittner 2018).  The clusters
// stay in Morton order; every round each cluster looks PLOC_R neighbours to either side for the
// one whose union with it has the smallest surface area, mutual choices merge, the array is
// compacted.  Agglomerative with a surface-area distance: the tree quality of a top-down SAH
// build without its serial passes.  Everything is decided by index arithmetic and exclusive
// sums, so the tree -- and with it every counter a render reports -- is the same in every run.
#define PLOC_R_MAX 128

// ordering of candidate pairs with equal distance: by a hash of the pair, then by (lower index,
// higher index).  Symmetric in its two members, so the pair that is globally smallest in
// (distance, this order) is always mutual and every round merges; hashed, because plain index order
// turns a run of equal distances (regular tessellations) into a chain with one mutual pair at its end.
__device__ __forceinline__ unsigned long long pair_key(int i, int j)
{
  const uint32_t a = (uint32_t) (i < j ? i : j), b = (uint32_t) (i < j ? j : i);
  uint32_t h = a * 0x9e3779b1u ^ (b + 0x7f4a7c15u) * 0x85ebca6bu;
  h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12;
  return ((unsigned long long) h << 32) | a;      // (hash, lower index); the higher index follows from the window
}
__device__ __forceinline__ bool pair_before(int i, int j, int k)
{
  const unsigned long long kj = pair_key(i, j), kk = pair_key(i, k);
  return kj < kk || (kj == kk && j < k);
}

__global__ void __launch_bounds__(LB) k_ploc_nearest(const uint32_t *C, int c, int radius, const Box6 *box, uint32_t *NN)
{
  __shared__ Box6 sb[LB + 2 * PLOC_R_MAX];
  const int base = (int) (blockIdx.x * LB) - radius;
  for (int t = threadIdx.x; t < LB + 2 * radius; t += LB) {
    const int g = base + t;
    if (g >= 0 && g < c) sb[t] = box[C[g]];
  }
  __syncthreads();
  const int i = blockIdx.x * LB + threadIdx.x;
  if (i >= c) return;
  const Box6 bi = sb[threadIdx.x + radius];
  float best = FLT_MAX;
  int bj = -1;
  for (int d = -radius; d <= radius; d++) {
    const int j = i + d;
    if (d == 0 || j < 0 || j >= c) continue;
    const float a = half_area(box_union(bi, sb[threadIdx.x + radius + d]));
    if (bj < 0 || a < best || (a == best && pair_before(i, j, bj))) { best = a; bj = j; }
  }
  NN[i] = (uint32_t) bj;
}

// per cluster: low word 1 = it disappears (the higher index of a mutual pair), high word 1 = it
// becomes the merged node (the lower index).  An exclusive sum of these gives both the compacted
// position and the new node's id.
__global__ void __launch_bounds__(LB) k_ploc_mark(const uint32_t *NN, int c, unsigned long long *marks)
{
  const int i = blockIdx.x * LB + threadIdx.x;
  if (i >= c) return;
  const int j = (int) NN[i];
  unsigned long long m = 0;
  if ((int) NN[j] == i) m = i < j ? (1ull << 32) : 1ull;
  marks[i] = m;
}

// boxes and leaf counts of the current clusters, packed (the host builds the top of the tree over them)
__global__ void __launch_bounds__(LB) k_cluster_boxes(const uint32_t *C, int c, BTree T, float *boxes6, uint32_t *counts)
{
  const int i = blockIdx.x * LB + threadIdx.x;
  if (i >= c) return;
  const Box6 b = T.box[C[i]];
  for (int k = 0; k < 3; k++) { boxes6[6 * i + k] = b.mn[k]; boxes6[6 * i + 3 + k] = b.mx[k]; }
  counts[i] = T.count[C[i]];
}

__global__ void __launch_bounds__(LB) k_ploc_apply(const uint32_t *C, int c, const uint32_t *NN, const unsigned long long *marks,
    const unsigned long long *sums, uint32_t next_node, BTree T, uint32_t *Cout, unsigned long long *total)
{
  const int i = blockIdx.x * LB + threadIdx.x;
  if (i >= c) return;
  const unsigned long long m = marks[i], s = sums[i];
  if (i == c - 1) *total = s + m;
  if (m == 1ull) return;
  const uint32_t pos = (uint32_t) i - (uint32_t) (s & 0xffffffffull);
  if (m == 0) { Cout[pos] = C[i]; return; }
  const uint32_t p = next_node + (uint32_t) (s >> 32);
  const uint32_t a = C[i], b = C[NN[i]];
  T.left[p] = a; T.right[p] = b;
  T.count[p] = T.count[a] + T.count[b];
  T.box[p] = box_union(T.box[a], T.box[b]);
  Cout[pos] = p;
}

// ---- collapse to 4-wide nodes, top-down one level per launch; the triangles of the subtree
// under a queue entry take the slots [first, first + count) of the final order, so a subtree
// of <= FJ_MAX_LEAF_PRIMS leaves is one leaf of the wide tree with contiguous triangles
struct QEntry { uint32_t node2, node4, first; };

__device__ __forceinline__ void write_leaf_order(const BTree &T, uint32_t n, const uint32_t *vals, uint32_t ref, uint32_t *dst)
{
  uint32_t stack[FJ_MAX_LEAF_PRIMS + 1];
  int sp = 0;
  stack[sp++] = ref;
  while (sp > 0) {
    const uint32_t r = stack[--sp];
    if (r < n) { *dst++ = vals[r]; continue; }
    stack[sp++] = T.right[r];
    stack[sp++] = T.left[r];
  }
}

// Leaf decision of the wide tree, the host builder's rule (fjgpu_build.cc): one or two triangles
// always; three or four only when splitting them once more does not pay -- the children's
// area-weighted triangle counts plus one node step (trav_cost triangle tests) against testing all.
__device__ __forceinline__ bool becomes_leaf(const BTree &T, uint32_t n, uint32_t ref, float trav_cost)
{
  const uint32_t cnt = T.count[ref];
  if (cnt > (uint32_t) FJ_MAX_LEAF_PRIMS) return false;
  if (cnt <= 2u || ref < n) return true;
  const uint32_t a = T.left[ref], b = T.right[ref];
  const float area = half_area(T.box[ref]);
  const float split = half_area(T.box[a]) * (float) T.count[a] + half_area(T.box[b]) * (float) T.count[b];
  return split + trav_cost * area >= area * (float) cnt;
}

__global__ void __launch_bounds__(LB) k_collapse(BTree T, uint32_t n, const uint32_t *vals, const QEntry *in, uint32_t n_in,
    QEntry *out, uint32_t *n_out, uint32_t *n_nodes4, DNode *nodes4, uint32_t *order, float trav_cost)
{
  const uint32_t q = blockIdx.x * LB + threadIdx.x;
  if (q >= n_in) return;
  const QEntry e = in[q];
  uint32_t ref[4];
  Box6 bx[4];
  int k = 2;
  ref[0] = T.left[e.node2]; ref[1] = T.right[e.node2];
  bx[0] = T.box[ref[0]]; bx[1] = T.box[ref[1]];
  while (k < 4) {
    int pick = -1;
    float area = -1.f;
    for (int i = 0; i < k; i++) {
      if (becomes_leaf(T, n, ref[i], trav_cost)) continue;
      const float a = half_area(bx[i]);
      if (a > area) { area = a; pick = i; }
    }
    if (pick < 0) break;
    const uint32_t p = ref[pick];
    ref[pick] = T.left[p]; ref[k] = T.right[p];
    bx[pick] = T.box[ref[pick]]; bx[k] = T.box[ref[k]];
    k++;
  }
  // larger children first (the any-hit walk visits hit children in slot order)
  for (int a = 0; a < k; a++)
    for (int b = a + 1; b < k; b++)
      if (half_area(bx[b]) > half_area(bx[a])) { const Box6 tb = bx[a]; bx[a] = bx[b]; bx[b] = tb; const uint32_t tr = ref[a]; ref[a] = ref[b]; ref[b] = tr; }
  DNode w;
  uint32_t off = e.first;
  for (int i = 0; i < 4; i++) {
    for (int c = 0; c < 3; c++) { w.box[i][2 * c] = i < k ? bx[i].mn[c] : FLT_MAX; w.box[i][2 * c + 1] = i < k ? bx[i].mx[c] : -FLT_MAX; }
    w.pad[i] = 0;
    if (i >= k) { w.child[i] = FJ_NO_CHILD; continue; }
    const uint32_t cnt = T.count[ref[i]];
    if (becomes_leaf(T, n, ref[i], trav_cost)) {
      w.child[i] = FJ_LEAF_FLAG | (off << 3) | (cnt - 1u);
      write_leaf_order(T, n, vals, ref[i], order + off);
    } else {
      const uint32_t slot = atomicAdd(n_nodes4, 1u);
      w.child[i] = slot;
      QEntry ne;
      ne.node2 = ref[i]; ne.node4 = slot; ne.first = off;
      out[atomicAdd(n_out, 1u)] = ne;
    }
    off += cnt;
  }
  nodes4[e.node4] = w;
}

// triangles in leaf order (the traversal's layout): vertices as exact f32 or f64, velocities
__global__ void __launch_bounds__(LB) k_gather(const double *P, const double *vel, const int32_t *idx, const uint32_t *vals, int n,
    float *v32, double *v64, double *tv)
{
  const int i = blockIdx.x * LB + threadIdx.x;
  if (i >= n) return;
  const size_t f = vals[i];
  for (int k = 0; k < 3; k++) {
    const size_t p = (size_t) idx[3 * f + k];
    for (int c = 0; c < 3; c++) {
      const double q = P[3 * p + c];
      if (v32) v32[(size_t) i * 9 + 3 * k + c] = (float) q;
      else v64[(size_t) i * 9 + 3 * k + c] = q;
      if (tv) tv[(size_t) i * 9 + 3 * k + c] = vel[3 * p + c];
    }
  }
}

template <class T> hipError_t dalloc(T **p, size_t n) { return hipMalloc((void **) p, (n ? n : 1) * sizeof(T)); }

}  // namespace

#define LB_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { *err = std::string("device BLAS build: ") + #expr + ": " + hipGetErrorString(e_); goto fail; } } while (0)

int LbvhBuildMesh(const double *d_P, const double *d_vel, const int32_t *d_idx, int n_faces, int n_points,
    const double bounds[6], bool f32_exact, int quality, LbvhOut *out, std::string *err)
{
  const int n = n_faces;
  std::memset(out, 0, sizeof(*out));
  Box6 *boxes = nullptr;
  unsigned long long *keys = nullptr, *keys2 = nullptr;
  uint32_t *vals = nullptr, *vals2 = nullptr;
  void *tmp = nullptr, *scan_tmp = nullptr;
  BTree T = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  uint32_t *c0 = nullptr, *c1 = nullptr, *nn = nullptr;
  unsigned long long *marks = nullptr, *sums = nullptr, *total = nullptr;
  QEntry *qa = nullptr, *qb = nullptr;
  uint32_t *counters = nullptr;     // [0] next-queue count, [1] wide nodes
  DNode *wide = nullptr;
  uint32_t *order = nullptr;
  int *bad = nullptr;
  const unsigned grid = (unsigned) ((n + LB - 1) / LB);
  int levels = 0;
  uint32_t root2 = 0;
  float trav_cost = 1.2f;                        // a node step ~ 1.2 triangle tests (as fjgpu_build.cc)
  if (const char *e = getenv("FJGPU_TRAV_COST")) trav_cost = (float) atof(e);

  if (n <= FJ_MAX_LEAF_PRIMS) {
    // a handful of triangles: the root is one leaf (no nodes)
    LB_TRY(dalloc(&out->nodes, 1));
    out->n_nodes = 1;
    out->root = n > 0 ? (FJ_LEAF_FLAG | (uint32_t) (n - 1)) : FJ_LEAF_FLAG;
    LB_TRY(dalloc(&out->prim_ids, (size_t) n));
    if (n > 0) {
      std::vector<uint32_t> ids(n);
      for (int i = 0; i < n; i++) ids[i] = (uint32_t) i;
      LB_TRY(hipMemcpy(out->prim_ids, ids.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
      if (f32_exact) LB_TRY(dalloc(&out->tri_verts32, (size_t) n * 9)); else LB_TRY(dalloc(&out->tri_verts, (size_t) n * 9));
      if (d_vel) LB_TRY(dalloc(&out->tri_vel, (size_t) n * 9));
      hipLaunchKernelGGL(k_gather, dim3(1), dim3(LB), 0, 0, d_P, d_vel, d_idx, out->prim_ids, n, out->tri_verts32, out->tri_verts, out->tri_vel);
      LB_TRY(hipDeviceSynchronize());
    }
    out->stack_need = 0;
    return 0;
  }

  LB_TRY(dalloc(&boxes, (size_t) n)); LB_TRY(dalloc(&keys, (size_t) n)); LB_TRY(dalloc(&keys2, (size_t) n));
  LB_TRY(dalloc(&vals, (size_t) n)); LB_TRY(dalloc(&vals2, (size_t) n)); LB_TRY(dalloc(&bad, 1));
  LB_TRY(hipMemset(bad, 0, sizeof(int)));
  {
    const double ex = bounds[3] - bounds[0], ey = bounds[4] - bounds[1], ez = bounds[5] - bounds[2];
    hipLaunchKernelGGL(k_prim, dim3(grid), dim3(LB), 0, 0, d_P, d_vel, d_idx, n, bounds[0], bounds[1], bounds[2],
        ex > 0 ? 1. / ex : 0., ey > 0 ? 1. / ey : 0., ez > 0 ? 1. / ez : 0., boxes, keys, vals, bad, n_points);
    int hbad = 0;
    LB_TRY(hipMemcpy(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost));
    if (hbad) { *err = "mesh index out of range"; goto fail; }
  }
  {
    size_t bytes = 0;
    LB_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, keys, keys2, vals, vals2, n, 0, 63, 0));
    LB_TRY(hipMalloc(&tmp, bytes ? bytes : 1));
    LB_TRY(hipcub::DeviceRadixSort::SortPairs(tmp, bytes, keys, keys2, vals, vals2, n, 0, 63, 0));
  }
  LB_TRY(dalloc(&T.left, (size_t) 2 * n)); LB_TRY(dalloc(&T.right, (size_t) 2 * n)); LB_TRY(dalloc(&T.count, (size_t) 2 * n));
  LB_TRY(dalloc(&T.box, (size_t) 2 * n));
  hipLaunchKernelGGL(k_leaf_nodes, dim3(grid), dim3(LB), 0, 0, boxes, vals2, n, T);

  if (quality <= 0) {
    // binary radix tree over the Morton codes + bottom-up fit
    LB_TRY(dalloc(&T.parent, (size_t) 2 * n)); LB_TRY(dalloc(&T.flag, (size_t) n));
    LB_TRY(hipMemset(T.flag, 0, sizeof(int) * (size_t) n));
    hipLaunchKernelGGL(k_hierarchy, dim3(grid), dim3(LB), 0, 0, keys2, n, T);
    hipLaunchKernelGGL(k_fit, dim3(grid), dim3(LB), 0, 0, n, T);
    root2 = (uint32_t) n;
  } else {
    // locally-ordered clustering: rounds of nearest-neighbour search, mark, exclusive sum, merge + compact
    int radius = 16;
    if (const char *e = getenv("FJGPU_PLOC_RADIUS")) radius = atoi(e);
    radius = radius < 1 ? 1 : (radius > PLOC_R_MAX ? PLOC_R_MAX : radius);
    LB_TRY(dalloc(&c0, (size_t) n)); LB_TRY(dalloc(&c1, (size_t) n)); LB_TRY(dalloc(&nn, (size_t) n));
    LB_TRY(dalloc(&marks, (size_t) n)); LB_TRY(dalloc(&sums, (size_t) n)); LB_TRY(dalloc(&total, 1));
    size_t scan_bytes = 0;
    LB_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, marks, sums, n, 0));
    LB_TRY(hipMalloc(&scan_tmp, scan_bytes ? scan_bytes : 1));
    {
      std::vector<uint32_t> ident((size_t) n);
      for (int i = 0; i < n; i++) ident[i] = (uint32_t) i;
      LB_TRY(hipMemcpy(c0, ident.data(), sizeof(uint32_t) * (size_t) n, hipMemcpyHostToDevice));
    }
    int c = n;
    uint32_t next_node = (uint32_t) n;
    int rounds = 0;
    // HYBRID TOP: agglomeration is at its best near the leaves (small clusters, neighbours in the Morton window) and at its
    // worst near the root, where every ray passes.  So the clustering stops once at most `top` clusters are left, and the tree
    // ABOVE them is a binned-SAH build over the clusters' boxes on the host (BuildTopTree: microseconds for 10^4 boxes).
    int top = 65536;                   // C3: 0 / 1 k / 8 k / 64 k / 512 k / 2 M clusters: 137.6 / 138.3 / 135.5 / 133.8 / 134.2 / 131.7 ms per frame (host tree: 129.4),
                                       // prepare 0.49 / . / 0.48 / 0.48 / 0.74 / 1.45 s (host build: 1.0 - 2.5 s)
    if (const char *e = getenv("FJGPU_PLOC_TOP")) top = atoi(e);
    while (c > 1 && c > top) {
      const unsigned g = (unsigned) ((c + LB - 1) / LB);
      hipLaunchKernelGGL(k_ploc_nearest, dim3(g), dim3(LB), 0, 0, c0, c, radius, T.box, nn);
      hipLaunchKernelGGL(k_ploc_mark, dim3(g), dim3(LB), 0, 0, nn, c, marks);
      size_t sb = scan_bytes;
      LB_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp, sb, marks, sums, c, 0));
      hipLaunchKernelGGL(k_ploc_apply, dim3(g), dim3(LB), 0, 0, c0, c, nn, marks, sums, next_node, T, c1, total);
      unsigned long long ht = 0;
      LB_TRY(hipMemcpy(&ht, total, sizeof(ht), hipMemcpyDeviceToHost));
      const uint32_t merged = (uint32_t) (ht >> 32);
      if (merged == 0 || merged != (uint32_t) (ht & 0xffffffffull)) { *err = "device BLAS build: clustering round merged nothing"; goto fail; }
      next_node += merged;
      c -= (int) merged;
      std::swap(c0, c1);
      rounds++;
    }
    if (c > 1) {
      // the top of the tree over the c remaining clusters
      float *d_b6 = nullptr; uint32_t *d_cnt = nullptr;
      LB_TRY(dalloc(&d_b6, (size_t) 6 * c)); LB_TRY(dalloc(&d_cnt, (size_t) c));
      hipLaunchKernelGGL(k_cluster_boxes, dim3((unsigned) ((c + LB - 1) / LB)), dim3(LB), 0, 0, c0, c, T, d_b6, d_cnt);
      std::vector<float> b6((size_t) 6 * c);
      std::vector<uint32_t> cnt((size_t) c), cid((size_t) c);
      LB_TRY(hipMemcpy(b6.data(), d_b6, sizeof(float) * b6.size(), hipMemcpyDeviceToHost));
      LB_TRY(hipMemcpy(cnt.data(), d_cnt, sizeof(uint32_t) * cnt.size(), hipMemcpyDeviceToHost));
      LB_TRY(hipMemcpy(cid.data(), c0, sizeof(uint32_t) * cid.size(), hipMemcpyDeviceToHost));
      (void) hipFree(d_b6); (void) hipFree(d_cnt);
      std::vector<fjgpu::TopNode> tn;
      int32_t troot = 0;
      if (fjgpu::BuildTopTree(b6.data(), c, &tn, &troot) || tn.size() != (size_t) c - 1) { *err = "device BLAS build: top tree failed"; goto fail; }
      const size_t m = tn.size();
      std::vector<uint32_t> hl(m), hr(m), hc(m, 0u);
      std::vector<Box6> hb(m);
      auto id_of = [&](int32_t ch) { return ch >= 0 ? next_node + (uint32_t) ch : cid[(size_t) ~ch]; };
      // counts bottom-up: children of a TopNode have larger indices (pre-order numbering)
      for (size_t i = m; i-- > 0;) {
        const fjgpu::TopNode &t = tn[i];
        hl[i] = id_of(t.left); hr[i] = id_of(t.right);
        hc[i] = (t.left >= 0 ? hc[(size_t) t.left] : cnt[(size_t) ~t.left]) + (t.right >= 0 ? hc[(size_t) t.right] : cnt[(size_t) ~t.right]);
        for (int k = 0; k < 3; k++) { hb[i].mn[k] = t.box[k]; hb[i].mx[k] = t.box[3 + k]; }
      }
      LB_TRY(hipMemcpy(T.left + next_node, hl.data(), sizeof(uint32_t) * m, hipMemcpyHostToDevice));
      LB_TRY(hipMemcpy(T.right + next_node, hr.data(), sizeof(uint32_t) * m, hipMemcpyHostToDevice));
      LB_TRY(hipMemcpy(T.count + next_node, hc.data(), sizeof(uint32_t) * m, hipMemcpyHostToDevice));
      LB_TRY(hipMemcpy(T.box + next_node, hb.data(), sizeof(Box6) * m, hipMemcpyHostToDevice));
      root2 = id_of(troot);
      if (getenv("FJGPU_VERBOSE") && n > 100000) fprintf(stderr, "fjgpu: clustering build: top of the tree over %d clusters by binned SAH\n", c);
    } else
    LB_TRY(hipMemcpy(&root2, c0, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (getenv("FJGPU_VERBOSE") && n > 100000) fprintf(stderr, "fjgpu: clustering build: %d rounds, window radius %d\n", rounds, radius);
  }

  // collapse, level by level from the root
  LB_TRY(dalloc(&qa, (size_t) n)); LB_TRY(dalloc(&qb, (size_t) n)); LB_TRY(dalloc(&counters, 2));
  LB_TRY(dalloc(&wide, (size_t) n)); LB_TRY(dalloc(&order, (size_t) n));
  {
    const QEntry root = {root2, 0u, 0u};
    const uint32_t init[2] = {0u, 1u};
    LB_TRY(hipMemcpy(qa, &root, sizeof(root), hipMemcpyHostToDevice));
    LB_TRY(hipMemcpy(counters, init, sizeof(init), hipMemcpyHostToDevice));
    uint32_t n_in = 1;
    while (n_in > 0) {
      hipLaunchKernelGGL(k_collapse, dim3((n_in + LB - 1) / LB), dim3(LB), 0, 0, T, (uint32_t) n, vals2, qa, n_in, qb, &counters[0], &counters[1], wide, order, trav_cost);
      uint32_t hc[2];
      LB_TRY(hipMemcpy(hc, counters, sizeof(hc), hipMemcpyDeviceToHost));
      n_in = hc[0];
      LB_TRY(hipMemset(&counters[0], 0, sizeof(uint32_t)));
      std::swap(qa, qb);
      levels++;
      out->n_nodes = hc[1];
    }
  }
  // results in exact-size buffers
  LB_TRY(dalloc(&out->nodes, out->n_nodes));
  LB_TRY(hipMemcpy(out->nodes, wide, sizeof(DNode) * out->n_nodes, hipMemcpyDeviceToDevice));
  out->root = 0;
  out->prim_ids = order; order = nullptr;
  if (f32_exact) LB_TRY(dalloc(&out->tri_verts32, (size_t) n * 9)); else LB_TRY(dalloc(&out->tri_verts, (size_t) n * 9));
  if (d_vel) LB_TRY(dalloc(&out->tri_vel, (size_t) n * 9));
  hipLaunchKernelGGL(k_gather, dim3(grid), dim3(LB), 0, 0, d_P, d_vel, d_idx, out->prim_ids, n, out->tri_verts32, out->tri_verts, out->tri_vel);
  LB_TRY(hipDeviceSynchronize());
  out->stack_need = 3 * levels + 1;      // <= 3 siblings pushed per level
  for (void *p : {(void *) boxes, (void *) keys, (void *) keys2, (void *) vals, (void *) vals2, tmp, scan_tmp, (void *) T.left, (void *) T.right,
                  (void *) T.count, (void *) T.box, (void *) T.parent, (void *) T.flag, (void *) c0, (void *) c1, (void *) nn, (void *) marks,
                  (void *) sums, (void *) total, (void *) qa, (void *) qb, (void *) counters, (void *) wide, (void *) bad})
    if (p) (void) hipFree(p);
  return 0;

fail:
  for (void *p : {(void *) boxes, (void *) keys, (void *) keys2, (void *) vals, (void *) vals2, tmp, scan_tmp, (void *) T.left, (void *) T.right,
                  (void *) T.count, (void *) T.box, (void *) T.parent, (void *) T.flag, (void *) c0, (void *) c1, (void *) nn, (void *) marks,
                  (void *) sums, (void *) total, (void *) qa, (void *) qb, (void *) counters, (void *) wide, (void *) order, (void *) bad,
                  (void *) out->nodes, (void *) out->prim_ids, (void *) out->tri_verts, (void *) out->tri_verts32, (void *) out->tri_vel})
    if (p) (void) hipFree(p);
  std::memset(out, 0, sizeof(*out));
  return -1;
}
