// fjgpu_build.cc -- host-side build of the device scene.
//
// Replaces the reference's build_accelerators() (src/fj_scene_interface.cc:1161-1202):
// instead of a uniform grid per mesh (src/fj_grid_accelerator.cc:69-160) the GPU
// core uses a binned-SAH BVH per primitive set ("BLAS"), built binary and collapsed
// to 4-wide 128-byte nodes, whose leaves hold up to four pre-gathered f64 triangles.
// Closest-hit results do not depend on the culling structure; the
// triangle test itself is the reference's FP64 Moller-Trumbore.
//
// The instance level (BVHAccelerator over ObjectInstances, src/fj_object_group.cc:27) is a
// threaded BVH per group (BuildGroupNodes): a plain list for the usual handful of instances.
#include "fjgpu_build.h"
#include "fjgpu_xform_math.h"
#include "fjgpu.h"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>

namespace fjgpu {

namespace {

const double ACC_PADDING = .0001;     // Accelerator PADDING, src/fj_accelerator.cc:13

// chunked parallel loop over [0, n) on the host threads (scene preparation only)
template <class F> void ParallelFor(size_t n, F fn)
{
  const size_t hc = std::max(1u, std::thread::hardware_concurrency());
  const size_t nt = std::min<size_t>(std::min<size_t>(hc, 64), n / 65536 + 1);
  if (nt <= 1) { fn(0, n); return; }
  std::vector<std::thread> th;
  for (size_t t = 0; t < nt; t++) th.emplace_back([=]() { fn(n * t / nt, n * (t + 1) / nt); });
  for (auto &x : th) x.join();
}

struct StageTimer {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  const bool on = getenv("FJGPU_VERBOSE") != nullptr;
  void lap(const char *what, int n)
  {
    const auto t1 = std::chrono::steady_clock::now();
    if (on && n > 100000) fprintf(stderr, "fjgpu:   %-28s %.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
    t0 = t1;
  }
};

// f64 -> f32 rounded toward -inf / +inf, then one extra ulp outward
inline float down2(double v)
{
  float f = (float) v;
  if ((double) f > v) f = std::nextafterf(f, -INFINITY);
  return std::nextafterf(f, -INFINITY);
}
inline float up2(double v)
{
  float f = (float) v;
  if ((double) f < v) f = std::nextafterf(f, INFINITY);
  return std::nextafterf(f, INFINITY);
}

struct Node2 {                 // binary node of the SAH build, before the 4-wide collapse
  float lmin[3], lmax[3];
  float rmin[3], rmax[3];
  uint32_t lc, rc;
};

// index allocator: a thread takes 1024 consecutive slots of a shared array at a time
struct ChunkAlloc {
  std::atomic<uint32_t> *next;
  uint32_t cur = 0, end = 0;
  explicit ChunkAlloc(std::atomic<uint32_t> *n) : next(n) {}
  uint32_t get()
  {
    if (cur == end) { cur = next->fetch_add(1024); end = cur + 1024; }
    return cur++;
  }
};

struct Builder {
  std::vector<PrimRef> prims;
  std::vector<PrimRef> scratch;            // out-of-place partition of the large top nodes
  std::unique_ptr<Node2[]> nodes;          // uninitialised storage, slots handed out in chunks
  std::atomic<uint32_t> next_node;
  std::atomic<int> max_depth;
  int max_leaf;
  float trav_cost;                         // cost of one node step in primitive tests (SAH leaf decision)

  static void grow(float *mn, float *mx, const float *a, const float *b)
  {
    for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], a[k]); mx[k] = std::max(mx[k], b[k]); }
  }
  static float half_area(const float *mn, const float *mx)
  {
    const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return dx * dy + dy * dz + dz * dx;
  }

  uint32_t leaf_ref(int begin, int count) const { return FJ_LEAF_FLAG | ((uint32_t) begin << 3) | (uint32_t) (count - 1); }

  enum { NB = 16, BIG = 1 << 18 };
  struct Bins {
    int cnt[NB];
    float bmn[NB][3], bmx[NB][3];
    void clear() { for (int b = 0; b < NB; b++) { cnt[b] = 0; for (int k = 0; k < 3; k++) { bmn[b][k] = FLT_MAX; bmx[b][k] = -FLT_MAX; } } }
    void merge(const Bins &o) { for (int b = 0; b < NB; b++) { cnt[b] += o.cnt[b]; grow(bmn[b], bmx[b], o.bmn[b], o.bmx[b]); } }
  };
  static int bin_of(const PrimRef &p, int axis, float lo, float scale)
  {
    const int b = (int) ((p.c[axis] - lo) * scale);
    return b < 0 ? 0 : (b >= NB ? NB - 1 : b);
  }

  // returns child ref for range [begin,end); writes its bounds into mn/mx.
  // Nodes of >= BIG primitives near the top run their passes on all host threads.
  uint32_t build(int begin, int end, int depth, int par_depth, float *mn, float *mx, ChunkAlloc &alloc)
  {
    const int n = end - begin;
    const bool big = par_depth > 0 && n >= BIG;
    for (int k = 0; k < 3; k++) { mn[k] = FLT_MAX; mx[k] = -FLT_MAX; }
    float cmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, cmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (big) {
      std::mutex mu;
      ParallelFor((size_t) n, [&](size_t i0, size_t i1) {
        float a[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, b[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        float c[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, d[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (size_t i = begin + i0; i < begin + i1; i++) { grow(a, b, prims[i].bmin, prims[i].bmax); grow(c, d, prims[i].c, prims[i].c); }
        std::lock_guard<std::mutex> g(mu);
        grow(mn, mx, a, b);
        grow(cmn, cmx, c, d);
      });
    } else {
      for (int i = begin; i < end; i++) {
        grow(mn, mx, prims[i].bmin, prims[i].bmax);
        grow(cmn, cmx, prims[i].c, prims[i].c);
      }
    }
    int d = max_depth.load(std::memory_order_relaxed);
    while (depth > d && !max_depth.compare_exchange_weak(d, depth)) {}
    if (n <= max_leaf) {
      // leaves of <= 4 prims are always accepted at or below max_leaf
      if (n <= std::min(2, max_leaf) || depth >= FJ_BVH_MAX_DEPTH - 2) return leaf_ref(begin, n);
    }

    // binned SAH on the axis with the widest centroid extent
    int axis = 0;
    float ext = cmx[0] - cmn[0];
    for (int k = 1; k < 3; k++) if (cmx[k] - cmn[k] > ext) { ext = cmx[k] - cmn[k]; axis = k; }
    int mid = -1;
    // remaining depth budget: subtrees deeper than this are split at the object median
    const int budget = FJ_BVH_MAX_DEPTH - 2 - depth;
    const double cap = std::ldexp((double) max_leaf, std::max(0, budget - 1));
    const bool force_median = (double) n > cap * 0.5 && budget < 30;
    if (ext > 0 && !force_median) {
      Bins B;
      float scale = NB * (1.f - 1e-6f) / ext;
      float lo = cmn[axis];
      float best = FLT_MAX;
      int best_b = -1;
      // SAH over the bins of one axis: best split plane and its cost
      auto sweep = [&](const Bins &Bx, float *cost_out, int *b_out) {
        float rarea[NB];
        int rcnt[NB];
        float amn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, amx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        int c = 0;
        for (int b = NB - 1; b > 0; b--) {
          if (Bx.cnt[b]) grow(amn, amx, Bx.bmn[b], Bx.bmx[b]);
          c += Bx.cnt[b];
          rcnt[b] = c;
          rarea[b] = c ? half_area(amn, amx) : 0.f;
        }
        for (int k = 0; k < 3; k++) { amn[k] = FLT_MAX; amx[k] = -FLT_MAX; }
        c = 0;
        *cost_out = FLT_MAX; *b_out = -1;
        for (int b = 0; b < NB - 1; b++) {
          if (Bx.cnt[b]) grow(amn, amx, Bx.bmn[b], Bx.bmx[b]);
          c += Bx.cnt[b];
          if (c == 0 || rcnt[b + 1] == 0) continue;
          const float cost = half_area(amn, amx) * c + rarea[b + 1] * rcnt[b + 1];
          if (cost < *cost_out) { *cost_out = cost; *b_out = b; }
        }
      };
      if (big) {
        // the few largest nodes: widest centroid axis only, binned on all threads
        B.clear();
        std::mutex mu;
        ParallelFor((size_t) n, [&](size_t i0, size_t i1) {
          Bins L;
          L.clear();
          for (size_t i = begin + i0; i < begin + i1; i++) {
            const int b = bin_of(prims[i], axis, lo, scale);
            L.cnt[b]++;
            grow(L.bmn[b], L.bmx[b], prims[i].bmin, prims[i].bmax);
          }
          std::lock_guard<std::mutex> g(mu);
          B.merge(L);
        });
        sweep(B, &best, &best_b);
      } else {
        // all three axes, keep the cheapest split
        int best_axis = axis;
        for (int ax = 0; ax < 3; ax++) {
          const float e = cmx[ax] - cmn[ax];
          if (!(e > 0)) continue;
          const float sc = NB * (1.f - 1e-6f) / e;
          Bins Bx;
          Bx.clear();
          for (int i = begin; i < end; i++) {
            const int b = bin_of(prims[i], ax, cmn[ax], sc);
            Bx.cnt[b]++;
            grow(Bx.bmn[b], Bx.bmx[b], prims[i].bmin, prims[i].bmax);
          }
          float cst;
          int bb;
          sweep(Bx, &cst, &bb);
          if (bb >= 0 && cst < best) { best = cst; best_b = bb; best_axis = ax; B = Bx; scale = sc; lo = cmn[ax]; }
        }
        axis = best_axis;
      }
      if (best_b >= 0) {
        // leaf cost vs split cost (traversal cost 1 node ~ 1.2 triangle tests)
        const float parent_area = half_area(mn, mx);
        if (n <= max_leaf && best + trav_cost * parent_area >= parent_area * n) return leaf_ref(begin, n);
        if (big) {
          // out-of-place parallel partition: per-chunk counts, prefix, scatter, copy back
          int nleft = 0;
          for (int b = 0; b <= best_b; b++) nleft += B.cnt[b];
          const size_t nt = 64;
          std::vector<int> lcount(nt + 1, 0);
          ParallelFor(nt, [&](size_t t0, size_t t1) {
            for (size_t t = t0; t < t1; t++) {
              int cnt = 0;
              for (size_t i = begin + (size_t) n * t / nt; i < begin + (size_t) n * (t + 1) / nt; i++) cnt += bin_of(prims[i], axis, lo, scale) <= best_b;
              lcount[t + 1] = cnt;
            }
          });
          for (size_t t = 0; t < nt; t++) lcount[t + 1] += lcount[t];
          std::vector<std::thread> th;
          for (size_t t = 0; t < nt; t++) th.emplace_back([&, t]() {
            const size_t i0 = begin + (size_t) n * t / nt, i1 = begin + (size_t) n * (t + 1) / nt;
            size_t l = begin + lcount[t], r = begin + nleft + ((i0 - begin) - lcount[t]);
            for (size_t i = i0; i < i1; i++) {
              if (bin_of(prims[i], axis, lo, scale) <= best_b) scratch[l++] = prims[i];
              else scratch[r++] = prims[i];
            }
          });
          for (auto &x : th) x.join();
          ParallelFor((size_t) n, [&](size_t i0, size_t i1) { std::memcpy(&prims[begin + i0], &scratch[begin + i0], (i1 - i0) * sizeof(PrimRef)); });
          mid = begin + nleft;
        } else {
          PrimRef *m = std::partition(&prims[begin], &prims[begin] + n, [&](const PrimRef &p) { return bin_of(p, axis, lo, scale) <= best_b; });
          mid = (int) (m - &prims[0]);
        }
      }
    }
    if (mid <= begin || mid >= end) {
      if (n <= max_leaf) return leaf_ref(begin, n);
      mid = begin + n / 2;
      std::nth_element(&prims[begin], &prims[mid], &prims[begin] + n,
          [axis](const PrimRef &a, const PrimRef &b) { return a.c[axis] < b.c[axis]; });
    }

    const uint32_t me = alloc.get();
    float lmn[3], lmx[3], rmn[3], rmx[3];
    uint32_t lc, rc;
    if (par_depth > 0 && n > 20000) {
      std::thread th([&]() { ChunkAlloc a2(&next_node); lc = build(begin, mid, depth + 1, par_depth - 1, lmn, lmx, a2); });
      rc = build(mid, end, depth + 1, par_depth - 1, rmn, rmx, alloc);
      th.join();
    } else {
      lc = build(begin, mid, depth + 1, 0, lmn, lmx, alloc);
      rc = build(mid, end, depth + 1, 0, rmn, rmx, alloc);
    }
    Node2 &nd = nodes[me];
    for (int k = 0; k < 3; k++) { nd.lmin[k] = lmn[k]; nd.lmax[k] = lmx[k]; nd.rmin[k] = rmn[k]; nd.rmax[k] = rmx[k]; }
    nd.lc = lc; nd.rc = rc;
    return me;
  }

  // ---- collapse to 4-wide nodes: starting from a binary node's two children, the inner
  // child with the largest surface area is replaced by its own two children until four
  // slots are filled (or only leaves remain).  Returns the DNode index; *need = worst-case
  // traversal stack entries below this node (k-1 siblings pushed, deepest child first).
  // Subtrees near the top are collapsed on their own threads.
  struct Cand { float mn[3], mx[3]; uint32_t ref; };
  std::unique_ptr<DNode[]> wide;
  std::atomic<uint32_t> next_wide;
  uint32_t collapse(uint32_t ref2, int *need, int par_depth, ChunkAlloc &alloc)
  {
    Cand c[4];
    int k = 2;
    {
      const Node2 &n = nodes[ref2];
      for (int a = 0; a < 3; a++) { c[0].mn[a] = n.lmin[a]; c[0].mx[a] = n.lmax[a]; c[1].mn[a] = n.rmin[a]; c[1].mx[a] = n.rmax[a]; }
      c[0].ref = n.lc; c[1].ref = n.rc;
    }
    while (k < 4) {
      int pick = -1;
      float area = -1.f;
      for (int i = 0; i < k; i++) {
        if (c[i].ref & FJ_LEAF_FLAG) continue;
        const float a = half_area(c[i].mn, c[i].mx);
        if (a > area) { area = a; pick = i; }
      }
      if (pick < 0) break;
      const Node2 &n = nodes[c[pick].ref];
      for (int a = 0; a < 3; a++) { c[pick].mn[a] = n.lmin[a]; c[pick].mx[a] = n.lmax[a]; c[k].mn[a] = n.rmin[a]; c[k].mx[a] = n.rmax[a]; }
      c[pick].ref = n.lc; c[k].ref = n.rc;
      k++;
    }
    // larger children first: the any-hit kernel visits hit children in slot order
    std::sort(c, c + k, [](const Cand &a, const Cand &b) { return half_area(a.mn, a.mx) > half_area(b.mn, b.mx); });
    const uint32_t me = alloc.get();
    int nd[4] = {0, 0, 0, 0};
    uint32_t child[4];
    std::vector<std::thread> th;
    for (int i = 0; i < 4; i++) {
      if (i >= k) { child[i] = FJ_NO_CHILD; continue; }
      if (c[i].ref & FJ_LEAF_FLAG) { child[i] = c[i].ref; continue; }
      if (par_depth > 0) th.emplace_back([&, i]() { ChunkAlloc a2(&next_wide); child[i] = collapse(c[i].ref, &nd[i], par_depth - 1, a2); });
      else child[i] = collapse(c[i].ref, &nd[i], 0, alloc);
    }
    for (auto &x : th) x.join();
    const int worst = std::max(std::max(nd[0], nd[1]), std::max(nd[2], nd[3]));
    DNode &w = wide[me];
    for (int i = 0; i < 4; i++) {
      for (int a = 0; a < 3; a++) {
        w.box[i][2 * a] = i < k ? c[i].mn[a] : FLT_MAX;
        w.box[i][2 * a + 1] = i < k ? c[i].mx[a] : -FLT_MAX;
      }
      w.child[i] = child[i];
      w.pad[i] = 0;
    }
    *need = (k - 1) + worst;
    return me;
  }
};

}  // namespace

int BuildTopTree(const float *boxes6, int K, std::vector<TopNode> *out, int32_t *root)
{
  out->clear();
  if (K < 1) return -1;
  if (K == 1) { *root = ~0; return 0; }
  Builder b;
  b.prims.resize((size_t) K);
  for (int i = 0; i < K; i++) {
    PrimRef &r = b.prims[i];
    for (int c = 0; c < 3; c++) { r.bmin[c] = boxes6[6 * i + c]; r.bmax[c] = boxes6[6 * i + 3 + c]; r.c[c] = .5f * (r.bmin[c] + r.bmax[c]); }
    r.id = (uint32_t) i;
  }
  b.nodes.reset(new Node2[(size_t) K + 2048]);
  b.next_node = 0;
  b.max_depth = 0;
  b.max_leaf = 1;                      // one cluster per leaf
  b.trav_cost = 1.2f;
  float mn[3], mx[3];
  ChunkAlloc alloc(&b.next_node);
  const uint32_t r2 = b.build(0, K, 0, 0, mn, mx, alloc);
  // Node2 tree -> TopNode list (children first is not needed: indices are assigned on the way down)
  struct Walk {
    const Builder &b; std::vector<TopNode> &out;
    int32_t go(uint32_t ref, float *bmn, float *bmx)
    {
      if (ref & FJ_LEAF_FLAG) {
        const uint32_t begin = (ref & 0x7fffffffu) >> 3;       // (count - 1 == 0)
        const PrimRef &p = b.prims[begin];
        for (int c = 0; c < 3; c++) { bmn[c] = p.bmin[c]; bmx[c] = p.bmax[c]; }
        return ~(int32_t) p.id;
      }
      const Node2 &n = b.nodes[ref];
      const int32_t me = (int32_t) out.size();
      out.push_back(TopNode());
      float a[3], c[3], e[3], f[3];
      const int32_t l = go(n.lc, a, c), r = go(n.rc, e, f);
      TopNode &t = out[(size_t) me];
      t.left = l; t.right = r;
      for (int k = 0; k < 3; k++) { bmn[k] = std::min(a[k], e[k]); bmx[k] = std::max(c[k], f[k]); t.box[k] = bmn[k]; t.box[3 + k] = bmx[k]; }
      return me;
    }
  } w{b, *out};
  float bmn[3], bmx[3];
  *root = w.go(r2, bmn, bmx);
  return 0;
}

float RoundDown2(double v) { return down2(v); }
float RoundUp2(double v) { return up2(v); }

void BuildBlas(HostPrimSet *ps, std::vector<PrimRef> &refs, int max_leaf, float trav_cost)
{
  Builder b;
  b.prims.swap(refs);
  const int n = (int) b.prims.size();
  // node slots are handed out in chunks of 1024 per thread: capacity = worst count + slack
  b.nodes.reset(new Node2[(size_t) std::max(n, 1) + 1024 * 132]);
  b.next_node = 0;
  b.max_depth = 0;
  b.max_leaf = std::max(1, std::min(max_leaf, FJ_MAX_LEAF_PRIMS));
  b.trav_cost = trav_cost;
  float mn[3], mx[3];
  if (n == 0) {
    ps->root = FJ_LEAF_FLAG;   // never dereferenced: n_prims == 0 is checked first
  } else {
    // task-parallel over the top levels: 2^par_depth subtrees on their own threads, and
    // the passes of the largest nodes on all of them
    int par_depth = 0;
    for (unsigned hc = std::max(1u, std::thread::hardware_concurrency()); hc > 1 && par_depth < 7; hc >>= 1) par_depth++;
    if (n >= Builder::BIG) b.scratch.resize(b.prims.size());
    StageTimer tb;
    ChunkAlloc alloc(&b.next_node);
    ps->root = b.build(0, n, 0, par_depth, mn, mx, alloc);
    tb.lap("binned SAH binary build", n);
  }
  StageTimer tm;
  ps->stack_need = 0;
  const int cpar = n > 200000 ? 3 : 0;
  const size_t wide_cap = (size_t) b.next_node.load() + 1024 * (cpar ? 128 : 2);
  b.wide.reset(new DNode[wide_cap]);
  b.next_wide = 0;
  if (n > 0 && !(ps->root & FJ_LEAF_FLAG)) {
    ChunkAlloc alloc(&b.next_wide);
    ps->root = b.collapse(ps->root, &ps->stack_need, cpar, alloc);
  }
  tm.lap("collapse to 4-wide", n);
  ps->nodes.n = std::max<size_t>(1, std::min<size_t>(wide_cap, b.next_wide.load()));
  ps->nodes.p = std::move(b.wide);
  ps->max_depth = b.max_depth;
  ps->prim_ids.resize(n);
  ParallelFor((size_t) n, [&](size_t i0, size_t i1) { for (size_t i = i0; i < i1; i++) ps->prim_ids[i] = b.prims[i].id; });
  ps->n_prims = n;
}

namespace {

int build_mesh(const fj_mesh_desc &m, HostPrimSet *ps, std::string *err, bool device_build)
{
  ps->device_build = false;
  ps->f32_exact = false;
  ps->type = FJ_PRIMSET_MESH;
  ps->mesh = &m;
  ps->curve = nullptr;
  if (m.n_faces > (1 << 28)) { *err = "mesh too large"; return FJGPU_EINVAL; }
  for (int k = 0; k < 3; k++) { ps->bounds[k] = m.bounds[k] - ACC_PADDING; ps->bounds[3 + k] = m.bounds[3 + k] + ACC_PADDING; }
  if (device_build) {
    // the device builds the tree and gathers the triangles; the host only decides the layout
    std::atomic<int> inexact0(0);
    ParallelFor((size_t) m.n_points * 3, [&](size_t i0, size_t i1) {
      for (size_t i = i0; i < i1; i++) if ((double) (float) m.P[i] != m.P[i]) { inexact0 = 1; return; }
    });
    ps->device_build = true;
    ps->f32_exact = !inexact0 && !getenv("FJGPU_NO_F32_TRIS");
    ps->root = FJ_LEAF_FLAG; ps->n_prims = m.n_faces; ps->max_depth = 0; ps->stack_need = 0;
    for (int k = 0; k < 3; k++) { ps->grid_cell[k] = 0; ps->grid_n[k] = 0; }
    return 0;
  }
  StageTimer tm;
  std::vector<PrimRef> refs(m.n_faces);
  std::atomic<int> bad(0);
  ParallelFor((size_t) m.n_faces, [&](size_t f0, size_t f1) {
    for (size_t f = f0; f < f1; f++) {
      double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
      for (int k = 0; k < 3; k++) {
        const int p = m.indices[3 * f + k];
        if (p < 0 || p >= m.n_points) { bad = 1; return; }
        for (int c = 0; c < 3; c++) {
          mn[c] = std::min(mn[c], m.P[3 * p + c]); mx[c] = std::max(mx[c], m.P[3 * p + c]);
          if (m.velocity) {      // swept over the shutter [0, 1]: Mesh::get_primitive_bounds, src/fj_mesh.cc:310-340
            const double q = m.P[3 * p + c] + m.velocity[3 * p + c];
            mn[c] = std::min(mn[c], q); mx[c] = std::max(mx[c], q);
          }
        }
      }
      PrimRef &r = refs[f];
      for (int c = 0; c < 3; c++) { r.bmin[c] = down2(mn[c]); r.bmax[c] = up2(mx[c]); r.c[c] = (float) (.5 * (mn[c] + mx[c])); }
      r.id = (uint32_t) f;
    }
  });
  if (bad) { *err = "mesh index out of range"; return FJGPU_EINVAL; }
  tm.lap("primitive boxes", m.n_faces);
  for (int k = 0; k < 3; k++) { ps->grid_cell[k] = 0; ps->grid_n[k] = 0; }
  {
    // experiment knobs: leaf size / SAH cost of a node step in triangle tests
    int max_leaf = FJ_MAX_LEAF_PRIMS;
    float trav_cost = 1.2f;                        // a node step ~ 1.2 triangle tests
    if (const char *e = getenv("FJGPU_MAX_LEAF")) max_leaf = atoi(e);
    if (const char *e = getenv("FJGPU_TRAV_COST")) trav_cost = (float) atof(e);
    BuildBlas(ps, refs, max_leaf, trav_cost);
  }
  tm.lap("BLAS total", m.n_faces);
  // pre-gathered vertices in leaf order; as f32 when that loses nothing (meshes read from
  // PLY files carry f32 coordinates): half the bytes per triangle test, identical operands
  std::atomic<int> inexact(0);
  ParallelFor((size_t) m.n_points * 3, [&](size_t i0, size_t i1) {
    for (size_t i = i0; i < i1; i++) if ((double) (float) m.P[i] != m.P[i]) { inexact = 1; return; }
  });
  const bool f32_exact = !inexact && !getenv("FJGPU_NO_F32_TRIS");
  if (f32_exact) ps->tri_verts32.resize((size_t) ps->n_prims * 9);
  else ps->tri_verts.resize((size_t) ps->n_prims * 9);
  ParallelFor((size_t) ps->n_prims, [&](size_t i0, size_t i1) {
    for (size_t i = i0; i < i1; i++) {
      const int f = (int) ps->prim_ids[i];
      for (int k = 0; k < 3; k++) {
        const int p = m.indices[3 * f + k];
        for (int c = 0; c < 3; c++) {
          if (f32_exact) ps->tri_verts32[i * 9 + 3 * k + c] = (float) m.P[3 * p + c];
          else ps->tri_verts[i * 9 + 3 * k + c] = m.P[3 * p + c];
        }
      }
    }
  });
  if (m.velocity) {
    ps->tri_vel.resize((size_t) ps->n_prims * 9);
    ParallelFor((size_t) ps->n_prims, [&](size_t i0, size_t i1) {
      for (size_t i = i0; i < i1; i++) {
        const int f = (int) ps->prim_ids[i];
        for (int k = 0; k < 3; k++) {
          const int p = m.indices[3 * f + k];
          for (int c = 0; c < 3; c++) ps->tri_vel[i * 9 + 3 * k + c] = m.velocity[3 * p + c];
        }
      }
    });
  }
  tm.lap("triangle gather", m.n_faces);
  return 0;
}

}  // namespace

int BuildCurveSet(const fj_curve_desc &c, HostPrimSet *ps, std::string *err);   // fjgpu_curve_build.cc

// Instance level of one group (the reference's BVHAccelerator over ObjectInstances,
// src/fj_bvh_accelerator.cc:79-107,253-334).  The TOPOLOGY is the reference's own -- sort the range
// by centroid on the cycling axis, split where find_median says -- because its depth-first leaf
// order decides which instance keeps an exactly equal t (the first one visited).  It is stored as a
// threaded list (DTNode): the leaves in that order, and an inner node with the union box in front
// of every subtree of more than TLAS_FLAT leaves (smaller subtrees are scanned).  Equal centroids
// (the reference's std::sort leaves their order open) are ordered by position in the group.
// The device builds the same list with the same arithmetic (fjgpu_tlas.hip); this host version
// serves scene builds without a device and the node-for-node check of the device build.
namespace {
struct TlasItem { double c[3]; int slot; };

int tlas_find_median(const std::vector<TlasItem> &it, const std::vector<int> &ord, int begin, int end, int axis)   // :313-334
{
  int low = begin, high = end - 1, mid = -1;
  const double key = (it[ord[low]].c[axis] + it[ord[high]].c[axis]) / 2;
  while (low != mid) {
    mid = (low + high) / 2;
    if (key < it[ord[mid]].c[axis]) high = mid;
    else if (it[ord[mid]].c[axis] < key) low = mid;
    else break;
  }
  return mid + 1;
}

void emit_group_nodes(const std::vector<DInstance> &instances, const std::vector<int> &members, const std::vector<TlasItem> &it,
    std::vector<int> &ord, int begin, int end, int axis, std::vector<DTNode> *out)
{
  const int n = end - begin;
  if (n == 1) {
    DTNode leaf; std::memset(&leaf, 0, sizeof(leaf)); leaf.inst = members[ord[begin]]; leaf.skip = 0;
    std::memcpy(leaf.box, instances[leaf.inst].wbounds, sizeof(leaf.box));
    out->push_back(leaf);
    return;
  }
  std::sort(ord.begin() + begin, ord.begin() + end, [&](int a, int b) {
    return it[a].c[axis] < it[b].c[axis] || (it[a].c[axis] == it[b].c[axis] && a < b);
  });
  const int median = tlas_find_median(it, ord, begin, end, axis);
  const size_t me = out->size();
  if (n > FJ_TLAS_FLAT) {
    double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (int i = begin; i < end; i++) {
      const double *b = instances[members[ord[i]]].wbounds;
      for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], b[k]); mx[k] = std::max(mx[k], b[3 + k]); }
    }
    DTNode inner;
    std::memset(&inner, 0, sizeof(inner));
    inner.inst = -1;
    for (int k = 0; k < 3; k++) {
      // widened: the walk tests inner boxes with approximate reciprocals (culling only)
      const double pad = 1e-9 * (std::fabs(mn[k]) + std::fabs(mx[k])) + 1e-12;
      inner.box[k] = mn[k] - pad; inner.box[3 + k] = mx[k] + pad;
    }
    out->push_back(inner);
  }
  emit_group_nodes(instances, members, it, ord, begin, median, (axis + 1) % 3, out);
  emit_group_nodes(instances, members, it, ord, median, end, (axis + 1) % 3, out);
  if (n > FJ_TLAS_FLAT) (*out)[me].skip = (int) out->size();
}
}  // namespace

void BuildGroupNodes(const std::vector<DInstance> &instances, const std::vector<int> &members, std::vector<DTNode> *out)
{
  const int n = (int) members.size();
  if (n == 0) return;
  std::vector<TlasItem> it(n);
  std::vector<int> ord(n);
  for (int i = 0; i < n; i++) {
    const double *b = instances[members[i]].wbounds;
    for (int k = 0; k < 3; k++) it[i].c[k] = .5 * (b[k] + b[3 + k]);       // Box::Centroid, src/fj_box.cc:63-66
    it[i].slot = i;
    ord[i] = i;
  }
  emit_group_nodes(instances, members, it, ord, 0, n, 0, out);      // (skip links are indices into *out: DGroup.first included)
}

int BuildHostScene(const fj_scene_desc *d, HostScene *out, std::string *err, bool device_mesh_build)
{
  if (!d) { *err = "null scene description"; return FJGPU_EINVAL; }
  out->n_meshes = d->n_meshes;
  out->primsets.resize(d->n_meshes + d->n_curves);
  for (int i = 0; i < d->n_meshes; i++) {
    const int e = build_mesh(d->meshes[i], &out->primsets[i], err, device_mesh_build);
    if (e) return e;
  }
  for (int i = 0; i < d->n_curves; i++) {
    const int e = BuildCurveSet(d->curves[i], &out->primsets[d->n_meshes + i], err);
    if (e) return e;
  }

  out->instances.resize(d->n_instances);
  for (int i = 0; i < d->n_instances; i++) {
    const fj_instance_desc &s = d->instances[i];
    DInstance &o = out->instances[i];
    std::memset(&o, 0, sizeof(o));
    const fj_xform_desc &x = s.xform;
    for (const int cnt : {x.n_translate, x.n_rotate, x.n_scale})
      if (cnt < 1 || cnt > FJ_MAX_XFORM_SAMPLES) { *err = "transform sample count out of range"; return FJGPU_EINVAL; }
    const bool is_static = x.n_translate == 1 && x.n_rotate == 1 && x.n_scale == 1;
    o.xform = -1;
    if (!is_static) { o.xform = (int) out->xforms.size(); out->xforms.push_back(x); }
    double M[16], Minv[16];
    MakeTransform(x, 0, M, Minv);
    std::memcpy(o.M, M, sizeof(o.M));
    std::memcpy(o.Minv, Minv, sizeof(o.Minv));
    if (s.primset_type == FJ_PRIMSET_MESH) {
      if (s.primset < 0 || s.primset >= d->n_meshes) { *err = "instance mesh index out of range"; return FJGPU_EINVAL; }
      o.primset = s.primset;
    } else if (s.primset_type == FJ_PRIMSET_CURVE) {
      if (s.primset < 0 || s.primset >= d->n_curves) { *err = "instance curve index out of range"; return FJGPU_EINVAL; }
      o.primset = d->n_meshes + s.primset;
    } else { *err = "unknown primitive set type"; return FJGPU_EINVAL; }
    // ObjectInstance::merge_sampled_bounds (reference src/fj_object_instance.cc:313-356).
    // The world box is built from T, R and the largest ABSOLUTE scale over the samples.  With
    // a negative scale it is mirrored and no longer encloses the geometry -- and the
    // reference's instance BVH culls with it, so rays that miss this box miss the instance
    // there.  The device tests the same box with the same BoxRayIntersect arithmetic
    // (DESIGN.md 4).  Time-sampled transforms: a rotating object's box is first replaced by
    // the cube around its bounding sphere, then the boxes of all translate samples are
    // merged (the rotate sample is picked with the TRANSLATE index; slots past the rotate
    // count hold zero samples, as in the reference's PropertySampleList).
    {
      double ob[6];
      std::memcpy(ob, out->primsets[o.primset].bounds, sizeof(ob));
      if (x.n_rotate > 1) {
        const double dx = ob[3] - ob[0], dy = ob[4] - ob[1], dz = ob[5] - ob[2];
        const double half_diagonal = .5 * std::sqrt(dx * dx + dy * dy + dz * dz);      // Length(Diagonal())
        const double c[3] = {(ob[0] + ob[3]) / 2, (ob[1] + ob[4]) / 2, (ob[2] + ob[5]) / 2};   // Centroid()
        for (int k = 0; k < 3; k++) { ob[k] = c[k] - half_diagonal; ob[3 + k] = c[k] + half_diagonal; }
      }
      double S[3] = {0, 0, 0};
      for (int i = 0; i < x.n_scale; i++)
        for (int k = 0; k < 3; k++) S[k] = std::max(S[k], std::fabs(x.scale[i].v[k]));
      const double big = 1.7976931348623157e308;
      double mb[6] = {big, big, big, -big, -big, -big};
      for (int i = 0; i < x.n_translate; i++) {
        fj_xform_desc ax = x;
        ax.n_translate = ax.n_rotate = ax.n_scale = 1;
        ax.translate[0] = x.translate[i];
        for (int k = 0; k < 3; k++) { ax.rotate[0].v[k] = i < x.n_rotate ? x.rotate[i].v[k] : 0.; ax.scale[0].v[k] = S[k]; }
        double Ma[16], Mai[16], sb[6];
        MakeTransform(ax, 0, Ma, Mai);
        TransformBounds(Ma, ob, sb);
        for (int k = 0; k < 3; k++) { mb[k] = std::min(mb[k], sb[k]); mb[3 + k] = std::max(mb[3 + k], sb[3 + k]); }
      }
      std::memcpy(o.wbounds, mb, sizeof(mb));
    }
    o.n_shaders = s.n_shaders;
    for (int k = 0; k < FJ_MAX_SHADING_GROUPS; k++) {
      o.shaders[k] = s.shaders[k];
      if (o.shaders[k] >= d->n_shaders) { *err = "shader index out of range"; return FJGPU_EINVAL; }
    }
    const int tg[3] = {s.reflect_target, s.refract_target, s.shadow_target};
    for (int t : tg) if (t < 0 || t >= d->n_groups) { *err = "trace target group out of range"; return FJGPU_EINVAL; }
    o.reflect_target = tg[0]; o.refract_target = tg[1]; o.shadow_target = tg[2];
  }

  out->shaders.assign(d->shaders, d->shaders + d->n_shaders);
  for (const fj_shader_desc &s : out->shaders) {
    const int tx[3] = {s.diffuse_map, s.bump_map, s.texture};
    for (int t : tx) if (t >= d->n_textures) { *err = "texture index out of range"; return FJGPU_EINVAL; }
  }

  out->groups.resize(d->n_groups);
  out->group_nodes.clear();
  out->group_members.clear();
  out->group_member_first.clear();
  for (int g = 0; g < d->n_groups; g++) {
    DGroup &G = out->groups[g];
    G.first = (int) out->group_nodes.size();
    G.n_instances = d->groups[g].n_instances;
    G.all_opaque = 1;
    for (int k = 0; k < 6; k++) G.sbounds[k] = 0;
    std::vector<int> members;
    for (int k = 0; k < G.n_instances; k++) {
      const int inst = d->groups[g].instances[k];
      if (inst < 0 || inst >= d->n_instances) { *err = "group instance index out of range"; return FJGPU_EINVAL; }
      members.push_back(inst);
      // occluder opacity: only PlasticShader has a settable Os (plastic_shader.cc:177-178)
      const DInstance &I = out->instances[inst];
      for (int s = 0; s < I.n_shaders && s < FJ_MAX_SHADING_GROUPS; s++) {
        const int sid = I.shaders[s];
        if (sid >= 0 && d->shaders[sid].type == FJ_SHADER_PLASTIC && d->shaders[sid].opacity < 1.f) G.all_opaque = 0;
      }
    }
    out->group_member_first.push_back((int) out->group_members.size());
    out->group_members.insert(out->group_members.end(), members.begin(), members.end());
    BuildGroupNodes(out->instances, members, &out->group_nodes);
    G.count = (int) out->group_nodes.size() - G.first;
    // Accelerator::ComputeBounds of a group: ObjectSet bounds + PADDING.  Only needed for
    // single-instance groups, where the instance BVH's root is the leaf and the group box is
    // the only world-space test the reference makes.
    if (G.n_instances == 1) {
      const DInstance &I = out->instances[members[0]];
      for (int k = 0; k < 3; k++) { G.sbounds[k] = I.wbounds[k] - ACC_PADDING; G.sbounds[3 + k] = I.wbounds[3 + k] + ACC_PADDING; }
    }
  }
  if (d->target_group < 0 || d->target_group >= d->n_groups) { *err = "renderer target group out of range"; return FJGPU_EINVAL; }
  out->target_group = d->target_group;

  // SlNewLightSamples for the deterministic lights (src/fj_shading.cc:380-404):
  // PointLight::get_samples (src/fj_point_light.cc:21-34), DomeLight::get_samples
  // (src/fj_dome_light.cc:30-51); transforms are evaluated at time 0.
  out->light_samples.clear();
  for (int i = 0; i < d->n_lights; i++) {
    const fj_light_desc &L = d->lights[i];
    double M[16], Minv[16];
    MakeTransform(L.xform, 0, M, Minv);
    if (L.type == FJ_POINT_LIGHT) {
      DLightSample s;
      // Transform.translate of the sample list evaluated at time 0 (lights are not time
      // sampled in the reference: "TODO time sampling", src/fj_point_light.cc:25-27)
      double T0[3] = {0, 0, 0};
      fjx::lerp_samples(L.xform.translate, L.xform.n_translate, 0., T0);
      s.P[0] = T0[0]; s.P[1] = T0[1]; s.P[2] = T0[2];
      for (int k = 0; k < 3; k++) s.Cl[k] = L.intensity * L.color[k];   // PointLight::illuminate
      s.light = i; s.type = FJ_POINT_LIGHT; s.ordinal = 0;
      out->light_samples.push_back(s);
    } else if (L.type == FJ_DOME_LIGHT) {
      const int n = std::min(L.sample_count, L.n_dome_samples);
      const float si = L.intensity / L.sample_count;                   // Light::sample_intensity_
      for (int k = 0; k < n; k++) {
        const fj_dome_sample &ds = L.dome_samples[k];
        const double p[3] = {ds.dir[0] * FLT_MAX, ds.dir[1] * FLT_MAX, ds.dir[2] * FLT_MAX};
        DLightSample s;
        for (int r = 0; r < 3; r++) s.P[r] = M[4 * r] * p[0] + M[4 * r + 1] * p[1] + M[4 * r + 2] * p[2] + M[4 * r + 3];
        for (int c = 0; c < 3; c++) s.Cl[c] = si * ds.color[c];
        s.light = i; s.type = FJ_DOME_LIGHT; s.ordinal = k;
        out->light_samples.push_back(s);
      }
    } else if (L.type == FJ_GRID_LIGHT || L.type == FJ_SPHERE_LIGHT) {
      // RectangleLight / SphereLight: get_sample_count() positions per shading event, drawn on
      // the device from the counter-based stream of DESIGN.md 4 (the reference's shared,
      // unsynchronised per-light XorShift makes its own image schedule dependent)
      if (out->area_lights.empty()) out->area_lights.resize(d->n_lights);
      DAreaLight &A = out->area_lights[i];
      std::memset(&A, 0, sizeof(A));
      std::memcpy(A.M, M, sizeof(A.M));
      const double n[3] = {M[1], M[5], M[9]};                      // M * (0,1,0) as a vector
      const double len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      const double inv = len > 0 ? 1. / len : 0.;                    // Normalize, src/fj_vector.h
      for (int k = 0; k < 3; k++) { A.N[k] = len > 0 ? n[k] * inv : n[k]; A.color[k] = L.color[k]; }
      A.sample_intensity = L.intensity / L.sample_count;
      A.double_sided = L.double_sided;
      for (int k = 0; k < L.sample_count; k++) {
        DLightSample s;
        std::memset(&s, 0, sizeof(s));
        s.light = i; s.type = L.type; s.ordinal = k;
        out->light_samples.push_back(s);
      }
    } else {
      *err = "unknown light type";
      return FJGPU_EINVAL;
    }
  }

  const fj_xform_desc &cx = d->camera.xform;
  for (const int cnt : {cx.n_translate, cx.n_rotate, cx.n_scale})
    if (cnt < 1 || cnt > FJ_MAX_XFORM_SAMPLES) { *err = "camera transform sample count out of range"; return FJGPU_EINVAL; }
  out->cam_static = cx.n_translate == 1 && cx.n_rotate == 1 && cx.n_scale == 1;
  out->cam_xform = cx;      // time-sampled: Camera::GetRay evaluates it per sample (k_gen_camera)
  double M[16], Minv[16];
  MakeTransform(cx, 0, M, Minv);
  std::memcpy(out->cam_M, M, sizeof(out->cam_M));
  out->cam_fov = d->camera.fov; out->cam_znear = d->camera.znear; out->cam_zfar = d->camera.zfar;
  return 0;
}

}  // namespace fjgpu
