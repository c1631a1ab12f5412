// fjgpu_tlas.hip -- the instance level (TLAS) of every group, built on the device.
//
// What is built is the threaded list fjgpu_types.h describes (DTNode): the leaves in the depth-first
// order of the REFERENCE's instance BVH -- build_bvh sorts a range by centroid on the cycling axis
// and splits it where find_median says (src/fj_bvh_accelerator.cc:253-334) -- with an inner node
// (union box + skip link) in front of every subtree of more than FJ_TLAS_FLAT leaves.  The host
// builder (fjgpu_build.cc: BuildGroupNodes) runs the same arithmetic; the two lists are equal byte
// for byte (option "tlas_verify", GPU test).
//
// One workgroup per group, level by level instead of by recursion:
//   * all ranges of a level are sorted at once by ONE bitonic network over the group's array
//     (ascending-only "flip" form, so the array needs no padding to a power of two); a
//     compare-exchange whose two positions lie in different ranges is skipped -- with (range, key)
//     as the composite key it would never swap -- so every range ends up sorted in place;
//   * one thread per range runs find_median and files the two children; elements learn their new
//     range from a table indexed by the old range's first position;
//   * the place of every node in the depth-first list follows from counts, not from a traversal:
//     an inner node over positions [b, e) sits at b + #(inner nodes starting before b) + #(its
//     ancestors starting AT b), the leaf at position i at i + #(inner nodes starting at or before i),
//     and the skip link of [b, e) is e + #(inner nodes starting before e).
// Instance counts are small (3-18 in the shipped scenes, 150 in the crowd test): the build is a few
// microseconds of device time, and re-running it IS the refit (boxes are recomputed from the
// instances' current bounds).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <vector>

#include "fjgpu_tlas.h"

namespace {

constexpr int TB = 256;

struct TlasWork {
  double *cent;        // [members][3] centroids
  int *ord;            // [members]    member slot at each position of the group's array
  int *segb;           // [members]    first position of the range a position belongs to
  int *split_of;       // [members]    by a range's first position: where it was last split (0: never)
  int *curb, *cure, *currank;      // ranges of the current level (size > 1)
  int *nxtb, *nxte, *nxtrank;      // ... and of the next one
  int *eb, *ee, *erank;            // emitted inner nodes: range and the number of emitted ancestors with the same b
  int *hist;           // [members + groups] per group: N + 1 counters (inner nodes by first position -> exclusive prefix)
};

__device__ __forceinline__ bool cent_less(const double *cent, int a, int b, int axis)
{
  const double ca = cent[3 * a + axis], cb = cent[3 * b + axis];
  return ca < cb || (ca == cb && a < b);
}

__device__ int find_median(const double *cent, const int *ord, int begin, int end, int axis)   // src/fj_bvh_accelerator.cc:313-334
{
  int low = begin, high = end - 1, mid = -1;
  const double key = (cent[3 * ord[low] + axis] + cent[3 * ord[high] + axis]) / 2;
  while (low != mid) {
    mid = (low + high) / 2;
    const double c = cent[3 * ord[mid] + axis];
    if (key < c) high = mid;
    else if (c < key) low = mid;
    else break;
  }
  return mid + 1;
}

__global__ void __launch_bounds__(TB) k_tlas_build(const DInstance *inst, const int *members, const int *mfirst, const int *mcount,
    TlasWork W, int *n_emit)
{
  const int g = blockIdx.x, tid = threadIdx.x;
  const int N = mcount[g], off = mfirst[g];
  double *cent = W.cent + 3 * (size_t) off;
  int *ord = W.ord + off, *segb = W.segb + off, *split_of = W.split_of + off;
  int *curb = W.curb + off, *cure = W.cure + off, *currank = W.currank + off;
  int *nxtb = W.nxtb + off, *nxte = W.nxte + off, *nxtrank = W.nxtrank + off;
  int *eb = W.eb + off, *ee = W.ee + off, *erank = W.erank + off;
  __shared__ int s_ncur, s_nnext, s_nemit;
  if (N == 0) { if (tid == 0) n_emit[g] = 0; return; }
  for (int i = tid; i < N; i += TB) {
    const double *b = inst[members[off + i]].wbounds;
    for (int k = 0; k < 3; k++) cent[3 * i + k] = .5 * (b[k] + b[3 + k]);      // Box::Centroid, src/fj_box.cc:63-66
    ord[i] = i; segb[i] = 0; split_of[i] = 0;
  }
  if (tid == 0) {
    curb[0] = 0; cure[0] = N; currank[0] = 0;
    s_ncur = N > 1 ? 1 : 0; s_nnext = 0; s_nemit = 0;
  }
  __syncthreads();
  int P = 1;
  while (P < N) P <<= 1;
  for (int level = 0;; level++) {
    const int ncur = s_ncur;
    if (ncur == 0) break;
    const int axis = level % 3;
    for (int k = 2; k <= P; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < N; i += TB) {
          const int l = (j == (k >> 1)) ? (i ^ (k - 1)) : (i ^ j);
          if (l > i && l < N && segb[i] == segb[l]) {
            const int a = ord[i], b = ord[l];
            if (cent_less(cent, b, a, axis)) { ord[i] = b; ord[l] = a; }
          }
        }
        __syncthreads();
      }
    }
    for (int s = tid; s < ncur; s += TB) {
      const int b = curb[s], e = cure[s], rank = currank[s];
      const int m = find_median(cent, ord, b, e, axis);
      const int emit = (e - b) > FJ_TLAS_FLAT ? 1 : 0;
      if (emit) { const int x = atomicAdd(&s_nemit, 1); eb[x] = b; ee[x] = e; erank[x] = rank; }
      split_of[b] = m;
      if (m - b > 1) { const int x = atomicAdd(&s_nnext, 1); nxtb[x] = b; nxte[x] = m; nxtrank[x] = rank + emit; }
      if (e - m > 1) { const int x = atomicAdd(&s_nnext, 1); nxtb[x] = m; nxte[x] = e; nxtrank[x] = 0; }
    }
    __syncthreads();
    for (int i = tid; i < N; i += TB) {
      const int m = split_of[segb[i]];       // (a stale entry is an ancestor's split: at or beyond this range's end)
      if (m > 0 && i >= m) segb[i] = m;
    }
    __syncthreads();
    { int *t; t = curb; curb = nxtb; nxtb = t; t = cure; cure = nxte; nxte = t; t = currank; currank = nxtrank; nxtrank = t; }
    if (tid == 0) { s_ncur = s_nnext; s_nnext = 0; }
    __syncthreads();
  }
  if (tid == 0) n_emit[g] = s_nemit;
}

__global__ void __launch_bounds__(TB) k_tlas_emit(const DInstance *inst, const int *members, const int *mfirst, const int *mcount,
    TlasWork W, const int *n_emit, const int *node_first, DTNode *nodes)
{
  const int g = blockIdx.x, tid = threadIdx.x;
  const int N = mcount[g], off = mfirst[g], E = n_emit[g], first = node_first[g];
  const int *ord = W.ord + off, *eb = W.eb + off, *ee = W.ee + off, *erank = W.erank + off;
  int *hist = W.hist + off + g;            // N + 1 entries
  for (int i = tid; i <= N; i += TB) hist[i] = 0;
  __syncthreads();
  for (int x = tid; x < E; x += TB) atomicAdd(&hist[eb[x]], 1);
  __syncthreads();
  // exclusive prefix, in place: hist[i] = #(inner nodes starting before position i); the count AT i is the
  // difference to the next entry
  if (tid == 0) { int run = 0; for (int i = 0; i <= N; i++) { const int c = hist[i]; hist[i] = run; run += c; } }
  __syncthreads();
  for (int x = tid; x < E; x += TB) {
    const int b = eb[x], e = ee[x];
    double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (int i = b; i < e; i++) {
      const double *w = inst[members[off + ord[i]]].wbounds;
      for (int k = 0; k < 3; k++) { mn[k] = fmin(mn[k], w[k]); mx[k] = fmax(mx[k], w[3 + k]); }
    }
    DTNode nd;
    for (int k = 0; k < 3; k++) {
      // widened: the walk tests inner boxes with approximate reciprocals (culling only)
      const double pad = 1e-9 * (fabs(mn[k]) + fabs(mx[k])) + 1e-12;
      nd.box[k] = mn[k] - pad; nd.box[3 + k] = mx[k] + pad;
    }
    nd.inst = -1;
    nd.skip = first + e + hist[e];
    nodes[first + b + hist[b] + erank[x]] = nd;
  }
  for (int i = tid; i < N; i += TB) {
    DTNode nd;
    nd.inst = members[off + ord[i]];
    for (int k = 0; k < 6; k++) nd.box[k] = inst[nd.inst].wbounds[k];
    nd.skip = 0;
    nodes[first + i + hist[i + 1]] = nd;     // hist[i + 1] = #(inner nodes starting at or before i)
  }
}

template <class T> bool dalloc(std::vector<void *> *keep, size_t n, T **out)
{
  void *p = nullptr;
  if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return false;
  keep->push_back(p);
  *out = static_cast<T *>(p);
  return true;
}

}  // namespace

int TlasBuildDevice(const DInstance *d_instances, const std::vector<int> &members, const std::vector<int> &member_first,
    const std::vector<int> &member_count, std::vector<int> *node_first, std::vector<int> *node_count, DTNode **d_nodes, std::string *err)
{
  const int G = (int) member_first.size();
  const size_t M = members.size();
  *d_nodes = nullptr;
  node_first->assign(G, 0);
  node_count->assign(G, 0);
  if (G == 0) return 0;
  std::vector<void *> tmp;
  auto done = [&](int rc, const char *why) {
    for (void *p : tmp) (void) hipFree(p);
    if (rc && why) *err = std::string("device TLAS build: ") + why;
    return rc;
  };
  int *d_members, *d_mfirst, *d_mcount, *d_nemit, *d_nfirst;
  TlasWork W;
  bool ok = dalloc(&tmp, M, &d_members) && dalloc(&tmp, (size_t) G, &d_mfirst) && dalloc(&tmp, (size_t) G, &d_mcount) &&
      dalloc(&tmp, (size_t) G, &d_nemit) && dalloc(&tmp, (size_t) G, &d_nfirst) && dalloc(&tmp, 3 * M, &W.cent) &&
      dalloc(&tmp, M, &W.ord) && dalloc(&tmp, M, &W.segb) && dalloc(&tmp, M, &W.split_of) &&
      dalloc(&tmp, M, &W.curb) && dalloc(&tmp, M, &W.cure) && dalloc(&tmp, M, &W.currank) &&
      dalloc(&tmp, M, &W.nxtb) && dalloc(&tmp, M, &W.nxte) && dalloc(&tmp, M, &W.nxtrank) &&
      dalloc(&tmp, M, &W.eb) && dalloc(&tmp, M, &W.ee) && dalloc(&tmp, M, &W.erank) && dalloc(&tmp, M + (size_t) G, &W.hist);
  if (!ok) return done(-1, "out of device memory");
  if (hipMemcpy(d_members, members.data(), M * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d_mfirst, member_first.data(), G * sizeof(int), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d_mcount, member_count.data(), G * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
    return done(-1, "upload failed");
  hipLaunchKernelGGL(k_tlas_build, dim3(G), dim3(TB), 0, 0, d_instances, d_members, d_mfirst, d_mcount, W, d_nemit);
  std::vector<int> nemit(G);
  {
    hipError_t le = hipGetLastError();        // (reading it clears it: keep the value for the message)
    if (le == hipSuccess) le = hipMemcpy(nemit.data(), d_nemit, G * sizeof(int), hipMemcpyDeviceToHost);
    if (le != hipSuccess) return done(-1, hipGetErrorString(le));
  }
  size_t total = 0;
  for (int g = 0; g < G; g++) { (*node_first)[g] = (int) total; (*node_count)[g] = member_count[g] + nemit[g]; total += (size_t) (*node_count)[g]; }
  if (hipMemcpy(d_nfirst, node_first->data(), G * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return done(-1, "upload failed");
  void *nodes = nullptr;
  if (hipMalloc(&nodes, (total ? total : 1) * sizeof(DTNode)) != hipSuccess) return done(-1, "out of device memory");
  hipLaunchKernelGGL(k_tlas_emit, dim3(G), dim3(TB), 0, 0, d_instances, d_members, d_mfirst, d_mcount, W, d_nemit, d_nfirst, (DTNode *) nodes);
  {
    hipError_t le = hipGetLastError();
    if (le == hipSuccess) le = hipDeviceSynchronize();
    if (le != hipSuccess) { (void) hipFree(nodes); return done(-1, hipGetErrorString(le)); }
  }
  *d_nodes = (DTNode *) nodes;
  return done(0, nullptr);
}
