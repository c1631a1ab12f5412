// fjgpu_api.hip -- C ABI of libfjgpu.so (include/fjgpu.h): device scene
// residency, the wavefront batch loop, and the ray-batch trace entry point.
#include <hip/hip_runtime.h>
#include <mutex>
#include <map>
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <chrono>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "fjgpu.h"
#include "fjgpu_build.h"
#include "fjgpu_kernels.h"
#include "fjgpu_lbvh.h"
#include "fjgpu_raysort.h"
#include "fjgpu_tlas.h"

namespace {

thread_local std::string t_last_error;

int fail(int code, const std::string &msg)
{
  t_last_error = msg;
  return code;
}

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
  return fail(FJGPU_ENODEV, std::string(#expr) + ": " + hipGetErrorString(e_)); } while (0)

#define FJ_LREC_BUFS 4

struct DeviceBuffers {
  std::vector<void *> ptrs;
  size_t bytes = 0;
  ~DeviceBuffers() { for (void *p : ptrs) (void) hipFree(p); }
  template <class T> int upload(const T *src, size_t n, const T **out)
  {
    *out = nullptr;
    if (n == 0 || src == nullptr) return 0;
    void *d = nullptr;
    if (hipMalloc(&d, n * sizeof(T)) != hipSuccess) return -1;
    ptrs.push_back(d);
    bytes += n * sizeof(T);
    if (hipMemcpy(d, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return -1;
    *out = static_cast<const T *>(d);
    return 0;
  }
  template <class T> int alloc(size_t n, T **out)
  {
    void *d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return -1;
    ptrs.push_back(d);
    bytes += std::max<size_t>(n, 1) * sizeof(T);
    *out = static_cast<T *>(d);
    return 0;
  }
};

// ---- RCCL, loaded at first use (no link-time dependency: the single-device paths never need it).  The slab exchange of
// fjgpu_render_frame_multi in RCCL's spelling -- ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd, SURVEY 8(e) -- beside the peer copy.
struct Rccl {
  typedef void *comm_t;
  int (*CommInitAll)(comm_t *, int, const int *) = nullptr;
  int (*CommDestroy)(comm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, comm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  void *handle = nullptr;
  bool ok = false;
  static constexpr int kFloat = 7;          // ncclFloat32 (rccl.h)
};

Rccl &rccl()
{
  static Rccl R;
  static std::once_flag once;
  std::call_once(once, []() {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { R.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (R.handle) break; }
    if (!R.handle) return;
    auto sym = [&](const char *n) { return dlsym(R.handle, n); };
    R.CommInitAll = (decltype(R.CommInitAll)) sym("ncclCommInitAll");
    R.CommDestroy = (decltype(R.CommDestroy)) sym("ncclCommDestroy");
    R.GroupStart = (decltype(R.GroupStart)) sym("ncclGroupStart");
    R.GroupEnd = (decltype(R.GroupEnd)) sym("ncclGroupEnd");
    R.Send = (decltype(R.Send)) sym("ncclSend");
    R.Recv = (decltype(R.Recv)) sym("ncclRecv");
    R.GetErrorString = (decltype(R.GetErrorString)) sym("ncclGetErrorString");
    R.ok = R.CommInitAll && R.CommDestroy && R.GroupStart && R.GroupEnd && R.Send && R.Recv;
  });
  return R;
}

// one communicator set per list of devices, created at first use and kept until the process exits (bring-up takes of the order of a second);
// destroyed by an atexit hook -- live communicators at exit are a known source of RCCL shutdown hangs.  A failed bring-up is remembered WITH its
// reason (the next frame falls back to peer copies at once and says why) and is retried after FJ_RCCL_RETRY_CALLS calls.
#define FJ_RCCL_RETRY_CALLS 64
struct RcclSet { std::vector<Rccl::comm_t> comms; std::string why; int calls_since_failure = 0; };
std::mutex g_rccl_mu;
std::map<std::vector<int>, RcclSet> g_rccl_comms;

void rccl_destroy_all()
{
  Rccl &R = rccl();
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  for (auto &kv : g_rccl_comms)
    for (Rccl::comm_t c : kv.second.comms) if (c && R.CommDestroy) (void) R.CommDestroy(c);
  g_rccl_comms.clear();
}

const std::vector<Rccl::comm_t> *rccl_comms(const std::vector<int> &devices, std::string *why)
{
  Rccl &R = rccl();
  if (!R.ok) { *why = "librccl.so not loadable"; return nullptr; }
  std::lock_guard<std::mutex> lock(g_rccl_mu);
  static bool hooked = false;
  if (!hooked) { hooked = true; atexit(rccl_destroy_all); }
  auto it = g_rccl_comms.find(devices);
  if (it != g_rccl_comms.end()) {
    if (!it->second.comms.empty()) return &it->second.comms;
    if (++it->second.calls_since_failure < FJ_RCCL_RETRY_CALLS) { *why = it->second.why + " (remembered)"; return nullptr; }
    g_rccl_comms.erase(it);
  }
  RcclSet set;
  set.comms.assign(devices.size(), nullptr);
  const int e = R.CommInitAll(set.comms.data(), (int) devices.size(), devices.data());
  if (e) { set.why = std::string("ncclCommInitAll: ") + (R.GetErrorString ? R.GetErrorString(e) : "error"); set.comms.clear(); *why = set.why; }
  auto &slot = g_rccl_comms[devices] = set;
  return slot.comms.empty() ? nullptr : &slot.comms;
}

}  // namespace

struct fjgpu_scene {
  int device;
  DScene S;                        // device pointers inside
  DeviceBuffers mem;               // scene-lifetime allocations
  double cam_fov;
  int n_light_samples;
  // options
  long batch_tiles;
  long batch_samples = 0;          // option "batch_samples": samples per batch where batch_tiles is 0 (0: as many as the memory budget holds)
  long render_calls = 0;           // fjgpu_render_tiles calls that rendered something (the first one sizes its batches for a cold start)
  long count_events;
  long count_all_shadow;
  // work buffers (lazily sized)
  std::unique_ptr<DeviceBuffers> work;
  size_t work_samples, work_rays;
  double *d_suv;
  uint32_t *d_stk = nullptr;       // (tile id, index in the tile) per sample slot: DScene.cam_tk (scenes with uid-keyed random streams), or null
  float *d_accum;
  // adaptive grid sampler (allocated on first use, in the `work` arena)
  double *d_aseen = nullptr, *d_afinal = nullptr;
  uint8_t *d_apstate = nullptr, *d_acells = nullptr;
  const void *a_owner = nullptr;
  size_t a_samples = 0, a_cell_bytes = 0;
  struct Level { DRay *rays; DPath *paths; size_t cap; uint32_t *keys; float *fc; };     // keys: sort keys of the level's rays (written by the shading kernel that emits them) or null;
                                   // fc: filter colours of refraction children, 3 floats per slot (scenes with a glass / pathtracing shader) or null
  bool has_filter_rays = false;    // some shader emits refraction children that carry a filter colour (DPath.flags bit 0)
  std::vector<Level> levels;       // ray queue per recursion level (allocated on first use)
  bool uses_sample_uid;            // some random stream or sample time of the scene is keyed by the sample's uid / index in its tile
  int max_children;                // most child rays one shading event can emit in this scene
  bool bounce_diffuse, bounce_reflect, bounce_refract;   // bounce types some shader of the scene emits
  DHit *d_hits;
  DLightRec *d_lrecs[FJ_LREC_BUFS];   // several buffers: the shadow stream consumes one while shading fills another (two are used without the overlap)
  DLightHair *d_lhair[FJ_LREC_BUFS];  // only when the scene has a HairShader
  int lrec_bufs = 2;                  // buffers allocated: 2, or FJ_LREC_BUFS once the overlap option is on
  long overlap;                    // option "overlap_shadow": 0 off, 1 on, 2 by the batch size
  bool overlap_now = false;        // ... what this render call does
  size_t free_cached = 0;          // free HBM + this scene's own work arena, as of the last query
  hipEvent_t ev_frame[2] = {nullptr, nullptr};   // frame start / end
  long early_shadow = 0;           // FJGPU_EARLY_SHADOW = k > 0 (only where the light loops run on their own stream: small batches, a rank's share of a frame): the shadow
                                   // queue is flushed after level k - 1's light loops, so that the one big any-hit walk runs BESIDE the deeper levels' closest-hit walks
                                   // instead of after them.  Round 6, a rank's share: C3 0 / 1 / 2 / 3 -> 18.7 / 18.1-18.6 / 18.3 / 18.5 ms, C6 +0.3, arealights +0.3, C2
                                   // 14.0 -> 15.1 with 1; switched on for single-instance shadow groups only, the SLOWEST of C3's eight shares went 17.95 -> 19.2 ms: off.
  hipStream_t shadow_stream;       // light loop + shadow traversal run here, overlapping the next level's closest-hit work
  hipStream_t read_stream = nullptr;   // the counters of a shading launch come back on this one while the main stream already runs the next level's walk
  hipEvent_t ev_shaded = nullptr;
  hipEvent_t ev_shadow_done[FJ_LREC_BUFS];    // shadow work reading d_lrecs[k] has finished
  std::vector<hipEvent_t> ev_pool; // per-launch timing events (resolved at the end of the frame: no host sync per launch)
  DShadowRay *d_squeue;
  size_t squeue_rec_bytes = 0;       // bytes per entry the queue was allocated for (DShadowRayC or DShadowRay)
  size_t squeue_cap;
  uint32_t *d_join = nullptr; size_t join_cap = 0;   // DScene.shadow_join (work arena)
  bool split_overflowed = false;   // the last render call overflowed a queue while rays were split
  uint32_t split_kfac = 1;         // queue entries the host allows per (light record, light) pair when rays are split
  bool split_shadow = false;       // the lean any-hit walk serves shadow groups of several instances: rays are queued per candidate instance
  DCounters *d_cnt;
  TileDesc *d_tiles;
  double *d_jit, *d_tim;
  size_t tab_len;
  int tiles_cap;
  int stack_need;
  double tri_record_bytes;         // 36 when every mesh is stored as f32 triangles, else 72
  size_t blas_nodes;
  size_t squeue_max;               // shadow-queue entries allowed by the memory budget
  size_t batch_fit_samples = 0;    // 0, or the samples per batch the work buffers were cut down to when an allocation failed (not tried beyond again)
  // ray-queue sort (fjgpu_raysort.hip): levels >= 1 of scenes with incoherent secondary rays
  int ray_sort_bits = 0;           // grid bits per axis; 0 = rays are walked in queue order
  double scene_box[6];             // union of the instances' world boxes (the sort's grid)
  uint32_t *d_sort[4] = {nullptr, nullptr, nullptr, nullptr};   // -, keys_alt, -, perm (in the `work` arena)
  void *d_sort_tmp = nullptr; size_t sort_tmp_bytes = 0; size_t sort_cap = 0;
  // frame-level buffers of fjgpu_render_frame_multi (lazily sized, freed with the scene)
  float *d_frame = nullptr; size_t d_frame_n = 0;      // this device's framebuffer
  float *d_slab = nullptr; size_t d_slab_n = 0;        // packed tiles: own ones (sender) / incoming (first device)
  int32_t *d_rects = nullptr; size_t d_rects_n = 0;    // tile rectangles of a slab
  fjgpu_batch_fn batch_fn = nullptr; void *batch_user = nullptr;   // fjgpu_set_batch_callback
};

static int grow(void **p, size_t *have, size_t want, size_t elem)
{
  if (*have >= want && *p) return 0;
  if (*p) (void) hipFree(*p);
  *p = nullptr; *have = 0;
  if (hipMalloc(p, std::max<size_t>(want, 1) * elem) != hipSuccess) return -1;
  *have = want;
  return 0;
}

// Option "overlap_shadow" (0 off, 1 on, 2 = by the size of the batch, the default): the light loops on a stream of their own, concurrent
// with the NEXT recursion levels' closest-hit walks, and with only FJ_OVERLAP_CULL_BLOCKS resident blocks per CU so that they leave the
// walks room.  It pays where the launches are small -- a rank's share of an 8-GPU frame, a small frame: the closest-hit walks of the deeper
// levels are then a few long rays each (a launch per level: 0.7 ms of dependent fetches and nothing else, profiles/r04_wave_timeline_*.txt),
// and the light loops' VALU work fits underneath: a rank's share of C3 20.9 -> 19.7 ms (profiles/r04_overlap_small_launches.txt).  A whole
// 1080p frame on one GPU loses with it (123 -> 136 ms: its walks fill the chip for most of their time), so "by size" turns it on below
// FJ_OVERLAP_MAX_SAMPLES samples per batch, for scenes whose shaders bounce.  (Round 1-3 kept it off: with the light loop at full
// occupancy the two streams only took turns.  Walking level 0's shadow rays early, under the deeper levels, was measured too
// (FJGPU_EARLY_SHADOW): the persistent walk fills every slot and the shading kernels of the deeper levels wait behind it: 20.3 ms.)
#ifndef FJ_OVERLAP_MAX_SAMPLES
#define FJ_OVERLAP_MAX_SAMPLES ((size_t) 24 << 20)    // (a quarter of a 1080p, 64 spp frame still loses with it: 35.5 -> 36.1 ms; an eighth gains)
#endif
#ifndef FJ_OVERLAP_CULL_BLOCKS
#define FJ_OVERLAP_CULL_BLOCKS 1
#endif
static int enable_overlap(fjgpu_scene *sc)
{
  if (sc->shadow_stream) return 0;
  // (FJGPU_OVERLAP_PRIO: the stream gets the LOWEST priority, so that its blocks only take the slots the main stream's kernels leave)
  int prio_lo = 0, prio_hi = 0;
  if (getenv("FJGPU_OVERLAP_PRIO") && hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) == hipSuccess && prio_lo != prio_hi) {
    if (hipStreamCreateWithPriority(&sc->shadow_stream, hipStreamNonBlocking, prio_lo) != hipSuccess) { sc->shadow_stream = nullptr; return -1; }
  }
  else if (hipStreamCreateWithFlags(&sc->shadow_stream, hipStreamNonBlocking) != hipSuccess) { sc->shadow_stream = nullptr; return -1; }
  for (int k = 0; k < FJ_LREC_BUFS; k++)
    if (hipEventCreateWithFlags(&sc->ev_shadow_done[k], hipEventDisableTiming) != hipSuccess) return -1;
  return 0;
}

// A render description the tiler can work on: positive sizes and a region inside the frame.
// The reference only asserts region min < max (src/fj_tiler.cc:75-76) and would write outside
// its framebuffer for a region past the frame; this is a public C ABI, so it is refused.
static const char *bad_tiling(const fj_render_desc *r)
{
  if (r->xres <= 0 || r->yres <= 0 || r->tile_w <= 0 || r->tile_h <= 0)
    return "bad render settings: resolution and tilesize must be positive";
  if (r->region[0] < 0 || r->region[1] < 0 || r->region[2] > r->xres || r->region[3] > r->yres ||
      r->region[0] >= r->region[2] || r->region[1] >= r->region[3])
    return "bad render settings: render_region must be a non-empty rectangle inside the frame";
  return nullptr;
}
static const char *bad_render(const fj_render_desc *r)
{
  if (const char *why = bad_tiling(r)) return why;
  if (r->rate_x <= 0 || r->rate_y <= 0) return "bad render settings: pixelsamples must be positive";
  return nullptr;
}

// FLAT groups (DFlat, fjgpu_dev_flat.h): when EVERY group of a scene with incoherent closest-hit rays holds only small static meshes built on
// the host, each group gets one world-space culling tree over the triangles of all its instances.  Returns the flats (empty: not applicable)
// and the deepest stack any of the trees needs.
#ifndef FJ_FLAT_MAX_TRIS
#define FJ_FLAT_MAX_TRIS (1 << 21)
#endif

struct HostFlat { fjgpu::HostPrimSet tree; std::vector<DFlatRef> refs; std::vector<double> refbox; double grid[6]; };
// one group: false if it cannot be flattened (an instance that is not a static, host-built, f32-exact mesh; more than 32 instances or max_tris triangles)
static bool build_flat_group(const fjgpu::HostScene &hs, size_t g, size_t max_tris, HostFlat *Fp)
{
  const DGroup &G = hs.groups[g];
  // the instances in the order the walks (and the reference's BVH) visit them: the leaves of the threaded instance level
  std::vector<int> order;
  for (int k = G.first; k < G.first + G.count; k++) if (hs.group_nodes[k].inst >= 0) order.push_back(hs.group_nodes[k].inst);
  if (order.empty() || order.size() > 32 || (int) order.size() != G.n_instances) return false;
  size_t total = 0;
  std::vector<size_t> first_ref(order.size() + 1, 0);
  for (size_t pos = 0; pos < order.size(); pos++) {
    const int inst = order[pos];
    const DInstance &I = hs.instances[inst];
    const fjgpu::HostPrimSet &ps = hs.primsets[I.primset];
    if (ps.type != FJ_PRIMSET_MESH || ps.device_build || I.xform >= 0 || !ps.tri_vel.empty() || inst >= (1 << 24)) return false;
    if (ps.n_prims > 0 && ps.tri_verts32.empty()) return false;          // (the leaf records hold f32 triangles: exact only for such sets)
    first_ref[pos] = total;
    total += (size_t) ps.n_prims;
  }
  first_ref[order.size()] = total;
  if (total > max_tris || total >= ((size_t) 1 << 28)) return false;
  HostFlat &F = *Fp;
  std::vector<fjgpu::PrimRef> prs(total);
  std::vector<DFlatRef> src(total);
  double gmn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, gmx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
  bool finite = true;
  for (size_t pos = 0; pos < order.size(); pos++) {
    const int inst = order[pos];
    const DInstance &I = hs.instances[inst];
    const fjgpu::HostPrimSet &ps = hs.primsets[I.primset];
    const double *rb = order.size() == 1 ? G.sbounds : I.wbounds;
    for (int k = 0; k < 6; k++) F.refbox.push_back(rb[k]);
    const size_t base = first_ref[pos];
    const int n = ps.n_prims;
    // (threads over the triangles of the instance: C2's group is 17.4 M of them)
    const unsigned hc = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    std::vector<char> bad(hc, 0);
    for (unsigned t = 0; t < hc; t++) th.emplace_back([&, t]() {
      for (int k = (int) ((size_t) n * t / hc); k < (int) ((size_t) n * (t + 1) / hc); k++) {
        double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
        for (int v = 0; v < 3; v++) {
          double p[3];
          for (int a = 0; a < 3; a++) p[a] = (double) ps.tri_verts32[(size_t) k * 9 + 3 * v + a];
          for (int a = 0; a < 3; a++) {
            const double w = I.M[4 * a] * p[0] + I.M[4 * a + 1] * p[1] + I.M[4 * a + 2] * p[2] + I.M[4 * a + 3];
            mn[a] = std::min(mn[a], w); mx[a] = std::max(mx[a], w);
          }
        }
        fjgpu::PrimRef r;
        for (int a = 0; a < 3; a++) {
          // the exact test runs in object space: its hit point, carried to the world by o + t d, lies on the image of the triangle up to the
          // roundings of M v and M^-1 (o, d) -- a relative 1e-9 (and 1e-12 absolute) covers them many times over
          if (!std::isfinite(mn[a]) || !std::isfinite(mx[a])) { bad[t] = 1; mn[a] = mx[a] = 0; }
          const double pad = 1e-9 * (std::fabs(mn[a]) + std::fabs(mx[a])) + 1e-12;
          r.bmin[a] = fjgpu::RoundDown2(mn[a] - pad);
          r.bmax[a] = fjgpu::RoundUp2(mx[a] + pad);
          r.c[a] = (float) (.5 * (mn[a] + mx[a]));
        }
        r.id = (uint32_t) (base + (size_t) k);
        prs[base + (size_t) k] = r;
        DFlatRef fr;
        for (int q = 0; q < 9; q++) fr.v[q] = ps.tri_verts32[(size_t) k * 9 + q];
        fr.inst_ord = ((uint32_t) inst << 8) | (uint32_t) pos;
        fr.pid = ps.prim_ids[k];
        fr.pad = 0;
        src[base + (size_t) k] = fr;
      }
    });
    for (auto &t : th) t.join();
    for (char b : bad) if (b) finite = false;
  }
  if (!finite) return false;
  for (const fjgpu::PrimRef &r : prs)
    for (int a = 0; a < 3; a++) { gmn[a] = std::min(gmn[a], (double) r.bmin[a]); gmx[a] = std::max(gmx[a], (double) r.bmax[a]); }
  F.tree.type = FJ_PRIMSET_MESH; F.tree.device_build = false; F.tree.f32_exact = false; F.tree.mesh = nullptr; F.tree.curve = nullptr;
  {
    // (experiment knobs like the mesh builder's: leaf size / SAH cost of a node step in candidate tests)
    int max_leaf = FJ_MAX_LEAF_PRIMS;
    float trav_cost = .7f;           // (a candidate costs three node steps here -- record, M^-1, transform, test: C4 frame 611 / 606 / 605 / 690 ms at 1.2 / .4 / .7 / 2.5)
    if (const char *e = getenv("FJGPU_FLAT_MAX_LEAF")) max_leaf = atoi(e);
    if (const char *e = getenv("FJGPU_FLAT_TRAV_COST")) trav_cost = (float) atof(e);
    fjgpu::BuildBlas(&F.tree, prs, max_leaf, trav_cost);
  }
  F.refs.resize(src.size());
  for (size_t sl = 0; sl < src.size(); sl++) F.refs[sl] = src[F.tree.prim_ids[sl]];
  for (int a = 0; a < 3; a++) {
    if (total == 0) { gmn[a] = 0; gmx[a] = 1; }
    const double pad = 1e-4 * std::max(1., gmx[a] - gmn[a]);
    F.grid[a] = gmn[a] - pad;
    F.grid[3 + a] = std::max(1e-300, (gmx[a] - gmn[a] + 2 * pad) / 65535. * (1 + 1e-9));
    F.tree.bounds[a] = gmn[a] - pad; F.tree.bounds[3 + a] = gmx[a] + pad;
  }
  return true;
}
// every group of the scene, or none
static bool build_flat_groups(const fjgpu::HostScene &hs, std::vector<HostFlat> *out)
{
  out->clear();
  if (hs.groups.empty() || !hs.xforms.empty()) return false;
  std::vector<HostFlat> flats(hs.groups.size());
  size_t max_tris = (size_t) FJ_FLAT_MAX_TRIS;
  if (const char *e = getenv("FJGPU_FLAT_MAX_TRIS_LOG2")) max_tris = (size_t) 1 << std::max(1, std::min(27, atoi(e)));
  for (size_t g = 0; g < hs.groups.size(); g++) if (!build_flat_group(hs, g, max_tris, &flats[g])) return false;
  out->swap(flats);
  return true;
}

static long g_ray_sort = -1;       // "ray_sort": grid bits per axis of the ray-queue sort (fjgpu_raysort.hip); 0 = off, -1 = by scene
static long g_ray_sort_min = FJ_RAY_SORT_MIN;   // "ray_sort_min": smaller launches keep queue order
static long g_device_tlas = 1;     // "device_tlas": the instance level of every group is built on the device (fjgpu_tlas.hip)
static long g_tlas_verify = 0;     // "tlas_verify": ... and compared node for node with the host's build (scene creation fails on a difference)
static long g_split_shadow = 1;    // "split_shadow": shadow rays into groups of several instances are queued once per candidate instance
                                   // (C2: any-hit walk 91 -> 54 ms, the light loop that now lists every candidate 18 -> 41 ms, frame 134 -> 121)
static long g_compact_squeue = 1;  // "compact_squeue": 56-byte shadow-queue records (DShadowRayC) where the walks can rebuild direction and distance
static long g_flat_groups = 1;     // "flat_groups": scenes with incoherent closest-hit rays whose groups hold only small static meshes walk ONE world-space tree per group
static long g_curve_anyhit = 1;    // "curve_anyhit": curve scenes whose occluders are all opaque walk their shadow rays with k_shadow_anyhit_curves
static long g_inst_lds = 1;        // "inst_lds": the walks keep the instance level of scenes that fit their budget in LDS (DInstEntry)
static long g_batch_tiles = 0;     // "batch_tiles": default of the per-scene option of that name for scenes created from now on (0 = by memory)
static long g_device_build = -1;   // "device_build": BLAS of meshes built on the GPU (fjgpu_lbvh.hip): 1 = clustering, 2 = radix tree; 0 = on the host; -1 = not set
static long g_multi_exchange = 0;  // "multi_exchange": how fjgpu_render_frame_multi moves the devices' tile slabs: 0 one hipMemcpyPeer each, 1 RCCL grouped send / recv
static long g_spec_walk = 1;       // "speculative_walk": the next recursion level's closest-hit walk is enqueued before the host has read how many rays it has (the walk reads the count itself)
static long g_cold_start = 1;      // "cold_start": a scene's first render call uses batches of FJ_COLD_BATCH_SAMPLES samples (a small arena: a fast first frame)
#ifndef FJ_COLD_BATCH_SAMPLES
#define FJ_COLD_BATCH_SAMPLES ((size_t) 16 << 20)
#endif
static long g_cold_batch_samples = (long) FJ_COLD_BATCH_SAMPLES;      // "cold_batch_samples": ... of this many samples (0 = the default, 16 M)
// "single_frame_build": scenes created while it is on render ONE frame (SiRenderScene): where device_build is not set they build on the GPU.
// PER THREAD (the caller switches it on around ITS fjgpu_scene_create call, which reads it on the same thread): a scene created at the same time on
// another thread is not touched by the toggle.
static thread_local long g_single_frame = 0;

extern "C" {

int fjgpu_global_option(const char *name, long value)
{
  if (!name) return fail(FJGPU_EINVAL, "bad option call");
  if (std::string(name) == "ray_sort") { g_ray_sort = value < -1 ? -1 : (value > 9 ? 9 : value); return 0; }
  if (std::string(name) == "ray_sort_min") { g_ray_sort_min = value < 1 ? 1 : value; return 0; }
  if (std::string(name) == "device_tlas") { g_device_tlas = value != 0; return 0; }
  if (std::string(name) == "tlas_verify") { g_tlas_verify = value != 0; return 0; }
  if (std::string(name) == "split_shadow") { g_split_shadow = value != 0; return 0; }
  if (std::string(name) == "inst_lds") { g_inst_lds = value != 0; return 0; }
  if (std::string(name) == "curve_anyhit") { g_curve_anyhit = value != 0; return 0; }
  if (std::string(name) == "flat_groups") { g_flat_groups = value < 0 ? 0 : (value > 2 ? 2 : value); return 0; }
  if (std::string(name) == "compact_squeue") { g_compact_squeue = value != 0; return 0; }
  if (std::string(name) == "batch_tiles") { g_batch_tiles = value < 0 ? 0 : value; return 0; }
  if (std::string(name) == "device_build") { g_device_build = value < 0 ? -1 : (value > 2 ? 2 : value); return 0; }
  if (std::string(name) == "single_frame_build") { g_single_frame = value != 0; return 0; }
  if (std::string(name) == "speculative_walk") { g_spec_walk = value != 0; return 0; }
  if (std::string(name) == "anyhit_filter_off") { set_anyhit_filter_off(value); return 0; }
  if (std::string(name) == "cold_start") { g_cold_start = value != 0; return 0; }
  if (std::string(name) == "cold_batch_samples") { g_cold_batch_samples = value > 0 ? value : (long) FJ_COLD_BATCH_SAMPLES; return 0; }
  if (std::string(name) == "multi_exchange") { if (value < 0 || value > 1) return fail(FJGPU_EINVAL, "multi_exchange: 0 peer copies, 1 RCCL send / recv"); g_multi_exchange = value; return 0; }
  return fail(FJGPU_EINVAL, std::string("unknown global option ") + name);
}

const char *fjgpu_last_error(void) { return t_last_error.c_str(); }

int fjgpu_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int fjgpu_tile_count(const fj_render_desc *render)
{
  if (!render) return 0;
  if (const char *why = bad_tiling(render)) { (void) fail(FJGPU_EINVAL, why); return 0; }
  std::vector<fjgpu::TileRect> t;
  fjgpu::GenerateTiles(*render, &t);
  return (int) t.size();
}

int fjgpu_tile_rect(const fj_render_desc *render, int tile_id, int32_t rect[4])
{
  if (!render || !rect) return fail(FJGPU_EINVAL, "null argument");
  if (const char *why = bad_tiling(render)) return fail(FJGPU_EINVAL, why);
  std::vector<fjgpu::TileRect> t;
  fjgpu::GenerateTiles(*render, &t);
  if (tile_id < 0 || tile_id >= (int) t.size()) return FJGPU_EINVAL;
  rect[0] = t[tile_id].xmin; rect[1] = t[tile_id].ymin; rect[2] = t[tile_id].xmax; rect[3] = t[tile_id].ymax;
  return 0;
}

}  // extern "C"

// Upload a built host scene to one device (fjgpu_scene_create = build + upload; the multi-device
// entry builds once and uploads to every device).
static int upload_scene(const fj_scene_desc *desc, fjgpu::HostScene &hs, int device, fjgpu_scene **out)
{
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(FJGPU_ENODEV, "no HIP device visible: the fjgpu core has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(FJGPU_EINVAL, "device index out of range");
  HIP_TRY(hipSetDevice(device));

  const auto t_up0 = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (getenv("FJGPU_VERBOSE")) { (void) hipDeviceSynchronize(); fprintf(stderr, "fjgpu: upload: %-28s at %.3f s\n", what,
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t_up0).count()); }
  };
  std::unique_ptr<fjgpu_scene> sc(new fjgpu_scene());
  sc->device = device;
  sc->batch_tiles = g_batch_tiles;
  sc->count_events = 0;     // traversal event counters are opt-in ("count_nodes"): they cost registers
  sc->count_all_shadow = 1;
  sc->work_samples = sc->work_rays = 0;
  sc->tab_len = 0;
  sc->overlap = 2;
  sc->shadow_stream = nullptr;
  for (int k = 0; k < FJ_LREC_BUFS; k++) { sc->ev_shadow_done[k] = nullptr; sc->d_lrecs[k] = nullptr; sc->d_lhair[k] = nullptr; }
  if (getenv("FJGPU_EARLY_SHADOW")) sc->early_shadow = atoi(getenv("FJGPU_EARLY_SHADOW"));
  if (const char *e = getenv("FJGPU_OVERLAP")) sc->overlap = std::max(0, std::min(2, atoi(e)));
  DeviceBuffers &M = sc->mem;
  int e = 0;

  std::vector<DPrimSet> dps(hs.primsets.size());
  for (size_t i = 0; i < hs.primsets.size(); i++) {
    const fjgpu::HostPrimSet &h = hs.primsets[i];
    DPrimSet &d = dps[i];
    std::memset(&d, 0, sizeof(d));
    d.type = h.type;
    d.root = h.root;
    d.n_prims = h.n_prims;
    std::memcpy(d.bounds, h.bounds, sizeof(d.bounds));
    for (int k = 0; k < 3; k++) { d.grid_cell[k] = h.grid_cell[k]; d.grid_n[k] = h.grid_n[k]; }
    if (!h.device_build) {
      e |= M.upload(h.nodes.data(), h.nodes.size(), &d.nodes);
      e |= M.upload(h.prim_ids.data(), h.prim_ids.size(), &d.prim_ids);
    }
    if (h.type == FJ_PRIMSET_MESH) {
      const fj_mesh_desc &m = *h.mesh;
      e |= M.upload(m.velocity, m.velocity ? (size_t) m.n_points * 3 : 0, &d.velocity);
      e |= M.upload(m.P, (size_t) m.n_points * 3, &d.P);
      e |= M.upload(m.N, m.N ? (size_t) m.n_points * 3 : 0, &d.N);
      e |= M.upload(m.vertex_N, m.vertex_N ? (size_t) m.n_faces * 9 : 0, &d.vN);
      e |= M.upload(m.uv, m.uv ? (size_t) m.n_points * 2 : 0, &d.uv);
      e |= M.upload(m.indices, (size_t) m.n_faces * 3, &d.indices);
      e |= M.upload(m.face_group, m.face_group ? (size_t) m.n_faces : 0, &d.face_group);
      if (h.device_build) {
        // BLAS on the device: Morton sort, clustering (or radix tree + fit), collapse, triangle gather
        if (e) return fail(FJGPU_ENOMEM, "device allocation / upload failed while creating the scene");
        const auto tb0 = std::chrono::steady_clock::now();
        LbvhOut lo;
        std::string lerr;
        if (LbvhBuildMesh(d.P, d.velocity, d.indices, m.n_faces, m.n_points, h.bounds, h.f32_exact, hs.device_build_quality, &lo, &lerr))
          return fail(FJGPU_ENODEV, lerr);
        for (void *p : {(void *) lo.nodes, (void *) lo.prim_ids, (void *) lo.tri_verts, (void *) lo.tri_verts32, (void *) lo.tri_vel})
          if (p) M.ptrs.push_back(p);
        d.nodes = lo.nodes; d.prim_ids = lo.prim_ids; d.root = lo.root;
        d.tri_verts = lo.tri_verts; d.tri_verts32 = lo.tri_verts32; d.tri_vel = lo.tri_vel;
        hs.primsets[i].stack_need = lo.stack_need;
        hs.primsets[i].nodes.n = lo.n_nodes;
        if (lo.tri_verts32) hs.primsets[i].tri_verts32.resize(1);      // (layout flags read below)
        if (lo.tri_vel) hs.primsets[i].tri_vel.resize(1);
        if (getenv("FJGPU_VERBOSE") && m.n_faces > 100000)
          fprintf(stderr, "fjgpu: device BLAS build of %d triangles: %.3f s, %zu wide nodes\n", m.n_faces,
              std::chrono::duration<double>(std::chrono::steady_clock::now() - tb0).count(), lo.n_nodes);
      } else {
        e |= M.upload(h.tri_verts.data(), h.tri_verts.size(), &d.tri_verts);
        e |= M.upload(h.tri_verts32.data(), h.tri_verts32.size(), &d.tri_verts32);
        e |= M.upload(h.tri_vel.data(), h.tri_vel.size(), &d.tri_vel);
      }
    } else {
      e |= M.upload(h.curve_cp.data(), h.curve_cp.size(), &d.curve_cp);
      e |= M.upload(h.curve_width.data(), h.curve_width.size(), &d.curve_width);
      e |= M.upload(h.curve_Cd.data(), h.curve_Cd.size(), &d.curve_Cd);
      e |= M.upload(h.curve_depth.data(), h.curve_depth.size(), &d.curve_depth);
      e |= M.upload(h.curve_capsule.data(), h.curve_capsule.size(), &d.curve_capsule);
      e |= M.upload(h.curve_vel.data(), h.curve_vel.size(), &d.curve_vel);
    }
  }
  lap("primitive sets");
  std::vector<DTexture> dtex(desc->n_textures);
  for (int i = 0; i < desc->n_textures; i++) {
    const fj_texture_desc &t = desc->textures[i];
    dtex[i].width = t.width; dtex[i].height = t.height; dtex[i].nchannels = t.nchannels; dtex[i].tilesize = t.tilesize;
    const size_t n = (t.width && t.tiles) ? (size_t) (t.width / t.tilesize) * (t.height / t.tilesize) * t.tilesize * t.tilesize * t.nchannels : 0;
    e |= M.upload(t.tiles, n, &dtex[i].tiles);
    if (!dtex[i].tiles) dtex[i].width = dtex[i].height = 0;
  }
  DScene &S = sc->S;
  std::memset(&S, 0, sizeof(S));
  {
    // quantised node arrays of the lean any-hit walk (DNodeQ): one per mesh, same node indices;
    // grid = 65536^3 cells over the primitive set's padded bounds
    std::vector<std::array<double, 6>> qgrid(dps.size());
    for (size_t i = 0; i < dps.size(); i++) {
      DPrimSet &P = dps[i];
      P.qnodes = nullptr;
      if (P.type != FJ_PRIMSET_MESH || P.n_prims == 0 || e) continue;
      const size_t n_nodes = std::max<size_t>(1, hs.primsets[i].nodes.size());
      for (int a = 0; a < 3; a++) {
        qgrid[i][a] = P.bounds[a];
        qgrid[i][3 + a] = std::max(1e-300, (P.bounds[3 + a] - P.bounds[a]) / 65535. * (1 + 1e-9));
      }
      DNodeQ *q = nullptr;
      if (M.alloc(n_nodes, &q)) { e = 1; continue; }
      if (launch_quantize_nodes(nullptr, P.nodes, (uint32_t) n_nodes, &qgrid[i][0], &qgrid[i][3], q)) e = 1;
      P.qnodes = q;
    }
    // curve sets: the same quantised twin for the closest-hit / general shadow walks of scenes with curve sets (kept out of
    // DPrimSet.qnodes, which the lean any-hit walk's tables read as "a mesh with pre-gathered triangles").  The grid spans the
    // set's bounds widened by 1e-4 of their size (at least 1e-4), so no node box is clamped at the rim.
    std::vector<const DNodeQ *> cq(dps.size(), nullptr);
    for (size_t i = 0; FJ_CURVE_QNODES && FJ_CLOSEST_QNODES && i < dps.size(); i++) {
      const DPrimSet &P = dps[i];
      if (P.type != FJ_PRIMSET_CURVE || P.n_prims == 0 || e) continue;
      const size_t n_nodes = std::max<size_t>(1, hs.primsets[i].nodes.size());
      for (int a = 0; a < 3; a++) {
        const double pad = 1e-4 * std::max(1., P.bounds[3 + a] - P.bounds[a]);
        qgrid[i][a] = P.bounds[a] - pad;
        qgrid[i][3 + a] = std::max(1e-300, (P.bounds[3 + a] - P.bounds[a] + 2 * pad) / 65535. * (1 + 1e-9));
      }
      DNodeQ *q = nullptr;
      if (M.alloc(n_nodes, &q)) { e = 1; continue; }
      if (launch_quantize_nodes(nullptr, P.nodes, (uint32_t) n_nodes, &qgrid[i][0], &qgrid[i][3], q)) e = 1;
      cq[i] = q;
    }
    lap("quantised nodes");
    e |= M.upload(dps.data(), dps.size(), &S.primsets);      // (after the loops above: DPrimSet.qnodes is set)
    {
      // the instance table, each record with a copy of its primitive set's entry data
      std::vector<DInstance> di = hs.instances;
      for (DInstance &I : di) {
        const DPrimSet &P = dps[I.primset];
        std::memcpy(I.pbounds, P.bounds, sizeof(I.pbounds));
        I.pnodes = P.nodes; I.proot = P.root; I.pn_prims = P.n_prims;
        I.pqnodes = P.qnodes ? P.qnodes : cq[I.primset];
        I.sh_indices = P.indices; I.sh_N = P.N; I.sh_vN = P.vN; I.sh_uv = P.uv; I.sh_face_group = P.face_group; I.sh_type = P.type; I.sh_pad = 0;
        for (int k = 0; k < 3; k++) { I.qorigin[k] = qgrid[I.primset][k]; I.qcell[k] = qgrid[I.primset][3 + k]; }
      }
      e |= M.upload(di.data(), di.size(), &S.instances);
      std::vector<DInstEntry> ie(di.size());
      bool any_curves = false, any_motion = !hs.xforms.empty();       // (DScene.has_curves / has_motion, set below)
      for (const auto &ps : hs.primsets) {
        if (ps.type == FJ_PRIMSET_CURVE && ps.n_prims > 0) any_curves = true;
        if (!ps.tri_vel.empty() || !ps.curve_vel.empty()) any_motion = true;
      }
      for (size_t k = 0; k < di.size(); k++) {
        const DInstance &I = di[k];
        DInstEntry &E = ie[k];
        std::memcpy(E.Minv, I.Minv, sizeof(E.Minv));
        std::memcpy(E.pbounds, I.pbounds, sizeof(E.pbounds));
        // tight world box: the eight corners of the primitive set's padded box through M, widened by 1e-7 of its size (+ 1e-9): it must contain
        // every point o + t d of a hit that the object-space test finds
        for (int a = 0; a < 3; a++) { E.tbounds[a] = -INFINITY; E.tbounds[3 + a] = INFINITY; }
        if (I.xform < 0 && I.pn_prims > 0) {
          double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
          for (int c = 0; c < 8; c++) {
            const double p[3] = {I.pbounds[(c & 1) ? 3 : 0], I.pbounds[(c & 2) ? 4 : 1], I.pbounds[(c & 4) ? 5 : 2]};
            for (int a = 0; a < 3; a++) {
              const double w = I.M[4 * a] * p[0] + I.M[4 * a + 1] * p[1] + I.M[4 * a + 2] * p[2] + I.M[4 * a + 3];
              mn[a] = std::min(mn[a], w); mx[a] = std::max(mx[a], w);
            }
          }
          bool finite = true;
          for (int a = 0; a < 3; a++) finite = finite && std::isfinite(mn[a]) && std::isfinite(mx[a]);
          if (finite)
            for (int a = 0; a < 3; a++) {
              const double pad = 1e-7 * (mx[a] - mn[a]) + 1e-7 * std::max(std::fabs(mn[a]), std::fabs(mx[a])) + 1e-9;
              E.tbounds[a] = mn[a] - pad; E.tbounds[3 + a] = mx[a] + pad;
            }
        }
        for (int a = 0; a < 3; a++) { E.qorigin[a] = I.qorigin[a]; E.qcell[a] = I.qcell[a]; }
        const DPrimSet &P = dps[I.primset];
        const bool quantised = FJ_CLOSEST_QNODES && (!any_curves || FJ_CURVE_QNODES) && !any_motion;     // (the instantiations launch_trace_closest picks)
        E.nodes = quantised ? (const void *) I.pqnodes : (const void *) I.pnodes;
        E.ptype = P.type; E.pad = 0;
        E.proot = I.proot; E.pn_prims = I.pn_prims; E.primset = I.primset; E.xform = I.xform;
        E.tri_verts = P.tri_verts; E.tri_verts32 = P.type == FJ_PRIMSET_CURVE ? P.curve_capsule : P.tri_verts32; E.tri_vel = P.tri_vel; E.prim_ids = P.prim_ids;
      }
      e |= M.upload(ie.data(), ie.size(), &S.inst_entries);
    }
    // flat per-instance records of that walk (static mesh instances): node and triangle arrays
    // as 32-bit offsets from the lowest of their addresses
    uintptr_t lo = UINTPTR_MAX, hi = 0;
    auto tris_of = [](const DPrimSet &P) { return (uintptr_t) (P.tri_verts32 ? (const void *) P.tri_verts32 : (const void *) P.tri_verts); };
    for (size_t i = 0; i < dps.size(); i++) {
      const DPrimSet &P = dps[i];
      if (!P.qnodes) continue;
      const uintptr_t pn = (uintptr_t) P.qnodes, pt = tris_of(P);
      lo = std::min(lo, std::min(pn, pt)); hi = std::max(hi, std::max(pn, pt));
    }
    // (hipMalloc returns 256-byte aligned blocks: offsets in units of 128 B span 512 GB)
    bool fits = lo != UINTPTR_MAX && lo % 128 == 0 && (hi - lo) / 128 < 0xffffffffull;
    for (size_t i = 0; i < dps.size(); i++) {
      const DPrimSet &P = dps[i];
      if (P.qnodes && (((uintptr_t) P.qnodes - lo) % 128 != 0 || (tris_of(P) - lo) % 128 != 0)) fits = false;
    }
    // the lean walk reads f32 triangle records only (every PLY mesh): a mesh that needs f64
    // vertices sends the scene's shadow rays through the general walk
    for (const DPrimSet &P : dps) if (P.qnodes && !P.tri_verts32) fits = false;
    S.blas_base = fits ? (const char *) lo : nullptr;
    if (!fits && lo != UINTPTR_MAX && getenv("FJGPU_VERBOSE"))
      fprintf(stderr, "fjgpu: BLAS arrays span %zu bytes from %p: no 32-bit offsets, the general shadow walk is used\n", (size_t) (hi - lo), (void *) lo);
    std::vector<DAnyInst> ai(hs.instances.size());
    for (size_t i = 0; i < hs.instances.size(); i++) {
      const DInstance &I = hs.instances[i];
      const DPrimSet &P = dps[I.primset];
      DAnyInst &a = ai[i];
      std::memset(&a, 0, sizeof(a));
      std::memcpy(a.Minv, I.Minv, sizeof(a.Minv));
      std::memcpy(a.bounds, P.bounds, sizeof(a.bounds));
      a.root = P.root;
      a.n_prims = (P.qnodes && S.blas_base) ? P.n_prims : 0;
      if (a.n_prims) {
        for (int k = 0; k < 3; k++) { a.qorigin[k] = qgrid[I.primset][k]; a.qcell[k] = qgrid[I.primset][3 + k]; }
        a.node_base = (uint32_t) (((uintptr_t) P.qnodes - lo) / 128);
        a.tri_base = (uint32_t) ((tris_of(P) - lo) / 128);
        a.tris_f32 = P.tri_verts32 ? 1 : 0;
        double bm = 0;
        for (int k = 0; k < 6; k++) bm = std::max(bm, std::fabs(P.bounds[k]));
        a.fbound = std::nextafter((float) bm, INFINITY);
      }
    }
    e |= M.upload(ai.data(), ai.size(), &S.any_insts);
  }
  if (g_device_tlas && e == 0 && !hs.groups.empty()) {
    // instance level on the device, from the instance table just uploaded
    std::vector<int> mcount(hs.groups.size()), nfirst, ncount;
    for (size_t g = 0; g < hs.groups.size(); g++) mcount[g] = hs.groups[g].n_instances;
    DTNode *d_nodes = nullptr;
    std::string terr;
    if (TlasBuildDevice(S.instances, hs.group_members, hs.group_member_first, mcount, &nfirst, &ncount, &d_nodes, &terr))
      return fail(FJGPU_ENODEV, terr);
    M.ptrs.push_back(d_nodes);
    size_t total = 0;
    for (int c : ncount) total += (size_t) c;
    if (g_tlas_verify) {
      std::vector<DTNode> got(total);
      HIP_TRY(hipMemcpy(got.data(), d_nodes, total * sizeof(DTNode), hipMemcpyDeviceToHost));
      bool same = total == hs.group_nodes.size();
      for (size_t g = 0; same && g < hs.groups.size(); g++) same = nfirst[g] == hs.groups[g].first && ncount[g] == hs.groups[g].count;
      if (same && total) same = std::memcmp(got.data(), hs.group_nodes.data(), total * sizeof(DTNode)) == 0;
      if (!same) return fail(FJGPU_ENODEV, "device TLAS build differs from the host's build");
    }
    std::vector<DGroup> groups = hs.groups;
    for (size_t g = 0; g < groups.size(); g++) { groups[g].first = nfirst[g]; groups[g].count = ncount[g]; }
    e |= M.upload(groups.data(), groups.size(), &S.groups);
    S.group_nodes = d_nodes;
    S.n_group_nodes = (int32_t) total;
  } else {
    e |= M.upload(hs.groups.data(), hs.groups.size(), &S.groups);
    e |= M.upload(hs.group_nodes.data(), hs.group_nodes.size(), &S.group_nodes);
    S.n_group_nodes = (int32_t) hs.group_nodes.size();
  }
  static_assert(sizeof(DInstEntry) == 8 * FJ_INST_LDS_ENTRY_WORDS && sizeof(DTNode) == 56 && sizeof(DGroup) == 64, "LDS copy of the instance level");
  S.curve_anyhit = (g_curve_anyhit && !getenv("FJGPU_NO_CURVE_ANYHIT")) ? 1 : 0;
  S.inst_lds = (g_inst_lds && !getenv("FJGPU_NO_INST_LDS")) ? 1 : 0;         // (each launcher checks the scene against its kernel's budget)
  e |= M.upload(hs.shaders.data(), hs.shaders.size(), &S.shaders);
  e |= M.upload(hs.xforms.data(), hs.xforms.size(), &S.xforms);
  S.cam_xform = nullptr;
  if (!hs.cam_static) e |= M.upload(&hs.cam_xform, 1, &S.cam_xform);
  S.has_motion = hs.xforms.empty() ? 0 : 1;
  for (const auto &ps : hs.primsets) if (!ps.tri_vel.empty() || !ps.curve_vel.empty()) S.has_motion = 1;   // vertex velocities need the ray's time too
  // shadow rays start at objects whose shader gathers light (SlIlluminance: plastic, hair) and go to
  // that object's shadow target group
  S.multi_shadow_groups = 0;
  for (const DInstance &I : hs.instances) {
    bool gathers = false;
    for (int k = 0; k < I.n_shaders; k++)
      if (I.shaders[k] >= 0 && (hs.shaders[I.shaders[k]].type == FJ_SHADER_PLASTIC || hs.shaders[I.shaders[k]].type == FJ_SHADER_HAIR)) gathers = true;
    if (gathers && I.shadow_target >= 0 && I.shadow_target < (int) hs.groups.size() && hs.groups[I.shadow_target].n_instances > 1)
      S.multi_shadow_groups = 1;
  }
  S.time_tab = nullptr; S.time_start = 0; S.time_end = 0;     // set per render call
  S.lrec_hair = nullptr;
  e |= M.upload(dtex.data(), dtex.size(), &S.textures);
  e |= M.upload(hs.light_samples.data(), hs.light_samples.size(), &S.light_samples);
  e |= M.upload(hs.area_lights.data(), hs.area_lights.size(), &S.area_lights);
  S.has_area = hs.area_lights.empty() ? 0 : 1;
  if (e) return fail(FJGPU_ENOMEM, "device allocation / upload failed while creating the scene");
  S.n_light_samples = (int) hs.light_samples.size();
  S.n_instances = (int) hs.instances.size();
  S.n_groups = (int) hs.groups.size();
  S.n_primsets = (int) hs.primsets.size();
  // FLAT groups: one world-space culling tree per group for the closest-hit walk of scenes with incoherent rays (fjgpu_dev_flat.h)
  int flat_stack_need = 0;
  S.flats = nullptr;
  {
    bool incoherent = false;         // (the rule of DScene.incoherent_rays below: two children per hit, or diffuse bounces)
    for (int i = 0; i < desc->n_shaders; i++) {
      const fj_shader_desc &sh = desc->shaders[i];
      auto lum = [](const float *c) { return .298912 * c[0] + .586611 * c[1] + .114478 * c[2] > 0.; };
      if (sh.type == FJ_SHADER_GLASS) incoherent = true;
      if (sh.type == FJ_SHADER_PATHTRACING && (lum(sh.diffuse) || (int) lum(sh.reflect) + (int) lum(sh.refract) >= 2)) incoherent = true;
    }
    std::vector<HostFlat> hf;
    // k_trace_closest_flat keeps M^-1 of every instance in its blocks' LDS (FJ_FLAT_LDS_INSTS of them; until round 5 it shared the phased walk's
    // copy of the whole instance level and its budget of 16 instances, while build_flat_group admits 32 per group: ADVICE round 4).  ONE decision,
    // made here: a scene beyond the budget, or with the inst_lds option off, builds no flat trees, S.flats stays null, and ray_sort_bits / the
    // closest_kernel query / the launcher all follow it.
    const bool flat_fits = S.inst_lds && S.n_instances <= FJ_FLAT_LDS_INSTS;
    // (option "flat_groups" 2 / FJGPU_FLAT_ALL: scenes with coherent closest-hit rays as well)
    const bool flat_wanted = incoherent || g_flat_groups >= 2 || getenv("FJGPU_FLAT_ALL");
    if (flat_wanted && flat_fits && g_flat_groups && !getenv("FJGPU_NO_FLAT") && FJ_CLOSEST_QNODES && build_flat_groups(hs, &hf)) {
      std::vector<DFlat> df(hf.size());
      for (size_t g = 0; g < hf.size() && !e; g++) {
        HostFlat &F = hf[g];
        DFlat &D = df[g];
        std::memset(&D, 0, sizeof(D));
        const DNode *d_nodes = nullptr;
        const size_t n_nodes = std::max<size_t>(1, F.tree.nodes.size());
        e |= M.upload(F.tree.nodes.data(), n_nodes, &d_nodes);
        DNodeQ *q = nullptr;
        if (!e && M.alloc(n_nodes, &q)) e = 1;
        if (!e && launch_quantize_nodes(nullptr, d_nodes, (uint32_t) n_nodes, &F.grid[0], &F.grid[3], q)) e = 1;
        D.nodes = q;
        e |= M.upload(F.refs.data(), F.refs.size(), &D.refs);
        e |= M.upload(F.refbox.data(), F.refbox.size(), &D.refbox);
        for (int a = 0; a < 3; a++) { D.qorigin[a] = F.grid[a]; D.qcell[a] = F.grid[3 + a]; }
        D.root = F.tree.root; D.n_inst = (int32_t) (F.refbox.size() / 6); D.n_prims = F.tree.n_prims;
        flat_stack_need = std::max(flat_stack_need, F.tree.stack_need);
      }
      if (!e) e |= M.upload(df.data(), df.size(), &S.flats);
      if (e) S.flats = nullptr;
      lap("flat groups");
    }
  }
  // traversal stack: entries beyond the LDS part live in a global overflow area sized for
  // the worst tree of the scene (usually none: stack_need <= FJ_STACK_LDS)
  {
    int need = S.flats ? flat_stack_need : 0;
    for (const auto &ps : hs.primsets) need = std::max(need, ps.stack_need);
    S.stack_overflow = nullptr;
    S.stack_overflow_shadow = nullptr;
    // (scenes with curve sets run the kernels that keep FJ_STACK_LDS_CURVES entries in LDS)
    bool any_curves = false;
    for (const auto &ps : hs.primsets) if (ps.type == FJ_PRIMSET_CURVE && ps.n_prims > 0) any_curves = true;
    (void) any_curves;
    const int lds_entries = FJ_STACK_LDS_MIN;      // the kernel with the fewest LDS entries decides
    if (need > lds_entries) {
      const size_t entries = (size_t) (need - lds_entries) * persistent_threads();
      if (M.alloc(entries, &S.stack_overflow) || M.alloc(entries, &S.stack_overflow_shadow)) return fail(FJGPU_ENOMEM, "device allocation failed for the traversal stack overflow area");
    }
    sc->stack_need = need;
    sc->tri_record_bytes = 36; sc->blas_nodes = 0;
    for (const auto &ps : hs.primsets) {
      sc->blas_nodes += ps.nodes.size();
      if (ps.type == FJ_PRIMSET_MESH && ps.n_prims > 0 && ps.tri_verts32.empty()) sc->tri_record_bytes = 72;
    }
    if (getenv("FJGPU_VERBOSE"))
      for (const auto &ps : hs.primsets)
        fprintf(stderr, "fjgpu: primset type %d prims %d nodes %zu binary depth %d stack need %d\n", ps.type, ps.n_prims,
            ps.nodes.size(), ps.max_depth, ps.stack_need);
  }
  S.has_curves = 0;
  S.all_opaque = 1;
  for (const auto &g : hs.groups) if (!g.all_opaque) S.all_opaque = 0;
  S.has_hair = 0;
  for (int i = 0; i < desc->n_shaders; i++) if (desc->shaders[i].type == FJ_SHADER_HAIR) S.has_hair = 1;
  for (const auto &ps : hs.primsets) if (ps.type == FJ_PRIMSET_CURVE && ps.n_prims > 0) S.has_curves = 1;
  S.target_group = hs.target_group;
  std::memcpy(S.cam_M, hs.cam_M, sizeof(S.cam_M));
  S.cam_znear = hs.cam_znear;
  S.cam_zfar = hs.cam_zfar;
  sc->cam_fov = hs.cam_fov;
  sc->n_light_samples = S.n_light_samples;
  sc->max_children = 0;
  sc->uses_sample_uid = S.has_motion || S.cam_xform != nullptr || S.has_area;
  sc->bounce_diffuse = sc->bounce_reflect = sc->bounce_refract = false;
  for (int i = 0; i < desc->n_shaders; i++) {
    const fj_shader_desc &sh = desc->shaders[i];
    if (sh.type == FJ_SHADER_PATHTRACING) sc->uses_sample_uid = true;
    if (sh.type == FJ_SHADER_GLASS || sh.type == FJ_SHADER_PATHTRACING) sc->has_filter_rays = true;     // (refraction children may carry a filter colour)
    auto lum = [](const float *c) { return .298912 * c[0] + .586611 * c[1] + .114478 * c[2] > 0.; };
    int k = 0;
    if (sh.type == FJ_SHADER_PLASTIC) { k = sh.do_reflect ? 1 : 0; if (k) sc->bounce_reflect = true; }
    else if (sh.type == FJ_SHADER_GLASS) { k = 2; sc->bounce_reflect = sc->bounce_refract = true; }
    else if (sh.type == FJ_SHADER_PATHTRACING) {
      k = (int) lum(sh.diffuse) + (int) lum(sh.reflect) + (int) lum(sh.refract);
      if (lum(sh.diffuse)) sc->bounce_diffuse = true;
      if (lum(sh.reflect)) sc->bounce_reflect = true;
      if (lum(sh.refract)) sc->bounce_refract = true;
    }
    sc->max_children = std::max(sc->max_children, k);
  }
  S.incoherent_rays = (sc->max_children >= 2 || sc->bounce_diffuse) ? 1 : 0;
  if (const char *e = getenv("FJGPU_PHASED_CLOSEST")) S.incoherent_rays = atoi(e) != 0;
  S.ray_perm = nullptr;
  S.trace_n_dev = nullptr;
  S.trace_ranges = nullptr;
  S.cam_uv = nullptr; S.cam_slot0 = 0; S.cam_tk = nullptr;
  S.shadow_join = nullptr;
  sc->split_shadow = S.multi_shadow_groups && S.all_opaque && !S.has_curves && !S.has_motion && S.blas_base && g_split_shadow;
  if (const char *e = getenv("FJGPU_SPLIT_SHADOW"))
    sc->split_shadow = atoi(e) != 0 && S.multi_shadow_groups && S.all_opaque && !S.has_curves && !S.has_motion && S.blas_base;
  {
    int kmax = 1;
    for (const auto &g : hs.groups) kmax = std::max(kmax, g.n_instances);
    sc->split_kfac = (uint32_t) std::min(kmax, 2);     // (more entries than that per (record, light) pair on average: overflow -> fallback;
                                                       //  C2 has 0.84; light-loop time at a bound of 2 / 3 / 4: 41 / 43 / 45 ms)
    if (const char *e = getenv("FJGPU_SPLIT_KFAC")) sc->split_kfac = (uint32_t) std::max(1, atoi(e));
  }
  for (int k = 0; k < 3; k++) { sc->scene_box[k] = DBL_MAX; sc->scene_box[3 + k] = -DBL_MAX; }
  for (const auto &I : hs.instances)
    for (int k = 0; k < 3; k++) { sc->scene_box[k] = std::min(sc->scene_box[k], I.wbounds[k]); sc->scene_box[3 + k] = std::max(sc->scene_box[3 + k], I.wbounds[3 + k]); }
  // default: scenes whose shaders scatter (diffuse bounces, two children per hit) sort their secondary rays
  // (not where the groups are flat: one walk per ray through one tree gains less from sorted rays than the sort costs -- C4 672 -> 653 ms without it)
  sc->ray_sort_bits = g_ray_sort >= 0 ? (int) g_ray_sort : (S.incoherent_rays && !S.has_curves && !S.has_motion && !S.flats ? FJ_RAY_SORT_BITS : 0);
  if (const char *e = getenv("FJGPU_RAY_SORT")) sc->ray_sort_bits = std::max(0, std::min(9, atoi(e)));
  HIP_TRY(hipDeviceSynchronize());
  lap("done");
  *out = sc.release();
  return 0;
}

static int build_host_scene(const fj_scene_desc *desc, fjgpu::HostScene *hs)
{
  std::string err;
  const auto t_build0 = std::chrono::steady_clock::now();
  // not set: the host's binned-SAH build where frames repeat (its tree traces 3-6 % faster), the GPU's clustering build for a scene that
  // renders one frame (0.03 s instead of 0.4 s at 7.2 M triangles; bin/scene on C3 end to end: 2.2 -> 1.5 s, profiles/r05_e2e_scene.txt)
  long device_mode = g_device_build >= 0 ? g_device_build : (g_single_frame ? 1 : 0);
  if (const char *e = getenv("FJGPU_DEVICE_BUILD")) device_mode = std::max(0, std::min(2, atoi(e)));
  const bool device_build = device_mode != 0;
  hs->device_build_quality = device_mode == 2 ? 0 : 1;
  const int be = fjgpu::BuildHostScene(desc, hs, &err, device_build);
  if (getenv("FJGPU_VERBOSE"))
    fprintf(stderr, "fjgpu: host scene build (BLAS, transforms, lights) %.3f s\n",
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t_build0).count());
  if (be) return fail(be, err);
  return 0;
}

extern "C" {

int fjgpu_scene_create(const fj_scene_desc *desc, int device, fjgpu_scene **out)
{
  if (!out || !desc) return fail(FJGPU_EINVAL, "null argument");
  *out = nullptr;
  fjgpu::HostScene hs;
  if (const int be = build_host_scene(desc, &hs)) return be;
  return upload_scene(desc, hs, device, out);
}

int fjgpu_scene_create_multi(const fj_scene_desc *desc, const int *devices, int n_devices, fjgpu_scene **out)
{
  if (!out || !desc || !devices || n_devices < 1) return fail(FJGPU_EINVAL, "bad argument");
  for (int k = 0; k < n_devices; k++) out[k] = nullptr;
  fjgpu::HostScene hs;                         // built once (the BLAS build is the expensive part), replicated
  if (const int be = build_host_scene(desc, &hs)) return be;
  for (int k = 0; k < n_devices; k++) {
    const int e = upload_scene(desc, hs, devices[k], &out[k]);
    if (e) { for (int j = 0; j < k; j++) { fjgpu_scene_destroy(out[j]); out[j] = nullptr; } return e; }
  }
  return 0;
}

void fjgpu_scene_destroy(fjgpu_scene *scene)
{
  if (!scene) return;
  (void) hipSetDevice(scene->device);
  (void) hipDeviceSynchronize();
  if (getenv("FJGPU_PHASE_STATS")) debug_phase_stats();
  if (scene->shadow_stream) (void) hipStreamDestroy(scene->shadow_stream);
  if (scene->read_stream) (void) hipStreamDestroy(scene->read_stream);
  if (scene->ev_shaded) (void) hipEventDestroy(scene->ev_shaded);
  for (hipEvent_t e : scene->ev_shadow_done) if (e) (void) hipEventDestroy(e);
  for (hipEvent_t e : scene->ev_frame) if (e) (void) hipEventDestroy(e);
  for (hipEvent_t e : scene->ev_pool) (void) hipEventDestroy(e);
  if (scene->d_frame) (void) hipFree(scene->d_frame);
  if (scene->d_slab) (void) hipFree(scene->d_slab);
  if (scene->d_rects) (void) hipFree(scene->d_rects);
  delete scene;
}

int fjgpu_host_instance_level(const fj_scene_desc *desc, int group, int32_t *out_inst, int32_t *out_skip, double *out_box, int cap)
{
  if (!desc || group < 0 || group >= desc->n_groups) return fail(FJGPU_EINVAL, "bad instance-level query");
  fjgpu::HostScene hs;
  std::string err;
  // (meshes are not needed for the instance level: their BLAS build is bypassed like in a device build)
  const int e = fjgpu::BuildHostScene(desc, &hs, &err, true);
  if (e) return fail(e, err);
  const DGroup &G = hs.groups[group];
  for (int k = 0; k < G.count && k < cap; k++) {
    const DTNode &nd = hs.group_nodes[G.first + k];
    if (out_inst) out_inst[k] = nd.inst;
    if (out_skip) out_skip[k] = nd.inst < 0 ? nd.skip - G.first : 0;
    if (out_box) std::memcpy(out_box + 6 * (size_t) k, nd.box, sizeof(nd.box));
  }
  return G.count;
}

int fjgpu_scene_query(const fjgpu_scene *scene, const char *name, double *value)
{
  if (!scene || !name || !value) return fail(FJGPU_EINVAL, "bad query call");
  const std::string n(name);
  if (n == "node_record_bytes") { *value = (double) sizeof(DNode); return 0; }
  if (n == "anyhit_node_record_bytes") { *value = (double) sizeof(DNodeQ); return 0; }
  if (n == "tri_record_bytes") { *value = scene->tri_record_bytes; return 0; }
  if (n == "scene_bytes") { *value = (double) scene->mem.bytes; return 0; }
  if (n == "work_bytes") { *value = (double) (scene->work ? scene->work->bytes : 0); return 0; }
  if (n == "stack_need") { *value = scene->stack_need; return 0; }
  if (n == "blas_nodes") { *value = (double) scene->blas_nodes; return 0; }
  // 1: shadow rays are walked by k_shadow_anyhit (every occluder opaque, no curves, no motion), 0: by k_shadow_trace
  // 1: shadow rays are walked by k_shadow_anyhit_curves (curve scene, every occluder opaque, no motion, instance level within the curve budget)
  if (n == "curve_anyhit") {
    const DScene &S = scene->S;
    *value = (S.has_curves && !S.has_motion && S.all_opaque && S.curve_anyhit && S.inst_lds && S.n_group_nodes <= FJ_INST_LDS_NODES_CURVES &&
              S.n_instances <= FJ_INST_LDS_INSTS_CURVES && S.n_groups <= FJ_INST_LDS_GROUPS_CURVES && FJ_CURVE_QNODES && FJ_CLOSEST_QNODES) ? 1 : 0;
    return 0;
  }
  if (n == "lean_anyhit") { *value = (scene->S.all_opaque && !scene->S.has_curves && !scene->S.has_motion && scene->S.blas_base) ? 1 : 0; return 0; }
  // which closest-hit kernel walks this scene (launch_trace_closest): 0 k_trace_closest<false, *, false>, 1 k_trace_closest_phased,
  // 2 k_trace_closest<true, *, false> (curve sets), 3 k_trace_closest<true, *, true> (time-sampled transforms / vertex velocities)
  // 4 k_trace_closest_flat (incoherent rays, every group flat, instance level within the phased walk's LDS budget)
  if (n == "closest_kernel") {
    const DScene &S = scene->S;
    const bool flat = !S.has_motion && !S.has_curves && S.flats;
    *value = S.has_motion ? 3 : (S.has_curves ? 2 : (flat ? 4 : (S.incoherent_rays ? 1 : 0)));
    return 0;
  }
  // ... and the node record it reads: the 64-byte quantised twin unless the ribbon test / motion instantiation runs
  if (n == "closest_node_record_bytes") { *value = (double) ((scene->S.has_motion || (scene->S.has_curves && !FJ_CURVE_QNODES) || !FJ_CLOSEST_QNODES) ? sizeof(DNode) : sizeof(DNodeQ)); return 0; }
  if (n == "has_curves") { *value = scene->S.has_curves; return 0; }
  if (n == "has_motion") { *value = scene->S.has_motion; return 0; }
  return fail(FJGPU_EINVAL, "unknown query " + n);
}

int fjgpu_set_batch_callback(fjgpu_scene *scene, fjgpu_batch_fn fn, void *user)
{
  if (!scene) return fail(FJGPU_EINVAL, "null scene");
  scene->batch_fn = fn; scene->batch_user = fn ? user : nullptr;
  return 0;
}

int fjgpu_dev_rccl_selftest(int device, int n_floats)
{
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || device < 0 || device >= nd) return fail(FJGPU_ENODEV, "no such HIP device");
  if (n_floats < 1) return fail(FJGPU_EINVAL, "bad size");
  HIP_TRY(hipSetDevice(device));
  std::string why;
  const std::vector<Rccl::comm_t> *comms = rccl_comms(std::vector<int>{device}, &why);
  if (!comms) return fail(FJGPU_ENODEV, "RCCL: " + why);
  Rccl &R = rccl();
  DeviceBuffers M;
  std::vector<float> src((size_t) n_floats), got((size_t) n_floats, -1.f);
  for (int i = 0; i < n_floats; i++) src[(size_t) i] = (float) (i % 1013) * .5f;
  const float *d_src = nullptr;
  float *d_dst = nullptr;
  if (M.upload(src.data(), src.size(), &d_src) || M.alloc(src.size(), &d_dst)) return fail(FJGPU_ENOMEM, "device allocation failed");
  HIP_TRY(hipMemset(d_dst, 0xff, sizeof(float) * src.size()));
  // the exchange's own calls on a communicator of one rank: a send to, and the matching receive from, rank 0 in one group
  int e = R.GroupStart();
  if (!e) e = R.Send(d_src, src.size(), Rccl::kFloat, 0, (*comms)[0], nullptr);
  if (!e) e = R.Recv(d_dst, src.size(), Rccl::kFloat, 0, (*comms)[0], nullptr);
  const int e2 = R.GroupEnd();
  if (e || e2) return fail(FJGPU_ENODEV, std::string("RCCL self exchange: ") + (R.GetErrorString ? R.GetErrorString(e ? e : e2) : "error"));
  HIP_TRY(hipStreamSynchronize(nullptr));
  HIP_TRY(hipMemcpy(got.data(), d_dst, sizeof(float) * got.size(), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < got.size(); i++) if (got[i] != src[i]) return fail(FJGPU_ENODEV, "RCCL self exchange returned other data");
  return 0;
}

int fjgpu_dev_sort_pairs(int device, const uint32_t *keys, int n, int key_bits, uint32_t *keys_out, uint32_t *perm, int repeats, double *sort_ms)
{
  if (n < 0 || key_bits < 1 || key_bits > 32 || (n > 0 && !keys)) return fail(FJGPU_EINVAL, "bad sort call");
  int nd = 0;
  if (hipGetDeviceCount(&nd) != hipSuccess || device < 0 || device >= nd) return fail(FJGPU_ENODEV, "no such HIP device");
  HIP_TRY(hipSetDevice(device));
  if (sort_ms) *sort_ms = 0;
  if (n == 0) return 0;
  DeviceBuffers M;
  const uint32_t *d_keys = nullptr;
  uint32_t *d_out = nullptr, *d_perm = nullptr;
  char *d_tmp = nullptr;
  const size_t tmp_bytes = ray_sort_pairs_temp_bytes((uint32_t) n, key_bits);
  if (M.upload(keys, (size_t) n, &d_keys) || M.alloc((size_t) n, &d_out) || M.alloc((size_t) n, &d_perm) || M.alloc(tmp_bytes, &d_tmp))
    return fail(FJGPU_ENOMEM, "device allocation failed for the sort");
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  double best = 1e30;
  for (int k = 0; k < std::max(1, repeats); k++) {
    (void) hipEventRecord(e0, nullptr);
    if (ray_sort_pairs(nullptr, d_keys, (uint32_t) n, key_bits, d_out, d_perm, d_tmp, tmp_bytes, keys_out != nullptr)) { (void) hipEventDestroy(e0); (void) hipEventDestroy(e1); return fail(FJGPU_ENODEV, "the sort's launches failed"); }
    (void) hipEventRecord(e1, nullptr);
    if (hipEventSynchronize(e1) != hipSuccess) { (void) hipEventDestroy(e0); (void) hipEventDestroy(e1); return fail(FJGPU_ENODEV, std::string("sort: ") + hipGetErrorString(hipGetLastError())); }
    float ms = 0.f;
    (void) hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, (double) ms);
  }
  (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
  if (sort_ms) *sort_ms = best;
  if (keys_out) HIP_TRY(hipMemcpy(keys_out, d_out, sizeof(uint32_t) * (size_t) n, hipMemcpyDeviceToHost));
  if (perm) HIP_TRY(hipMemcpy(perm, d_perm, sizeof(uint32_t) * (size_t) n, hipMemcpyDeviceToHost));
  return 0;
}

int fjgpu_set_option(fjgpu_scene *scene, const char *name, long value)
{
  if (!scene || !name) return fail(FJGPU_EINVAL, "bad option call");
  const std::string n(name);
  if (n == "batch_tiles") { scene->batch_tiles = value; return 0; }
  if (n == "batch_samples") { scene->batch_samples = value < 0 ? 0 : value; return 0; }
  if (n == "count_nodes") { scene->count_events = value != 0; return 0; }
  if (n == "count_all_shadow") { scene->count_all_shadow = value != 0; return 0; }
  if (n == "overlap_shadow") {
    if (value < 0 || value > 2) return fail(FJGPU_EINVAL, "overlap_shadow: 0 off, 1 on, 2 by the batch size");
    scene->overlap = value;
    return 0;
  }
  if (n == "release_work") {
    // the work arena back to the driver (the scene stays): a host that keeps its scene but renders nothing for a while -- or, like bench.py's
    // counter passes, wants another process to find the memory this one found.  The next render call allocates again.
    if (value && scene->work) {
      if (hipSetDevice(scene->device) != hipSuccess) return fail(FJGPU_ENODEV, "hipSetDevice");
      (void) hipDeviceSynchronize();
      scene->work.reset();
      scene->work_samples = scene->work_rays = 0; scene->tiles_cap = 0; scene->lrec_bufs = std::min(scene->lrec_bufs, 2); scene->squeue_rec_bytes = 0;
      scene->sort_cap = 0; scene->a_owner = nullptr; scene->a_samples = 0; scene->a_cell_bytes = 0;
      scene->d_aseen = scene->d_afinal = nullptr; scene->d_apstate = scene->d_acells = nullptr;
      for (auto &L : scene->levels) { L.rays = nullptr; L.paths = nullptr; L.cap = 0; L.keys = nullptr; }
    }
    return 0;
  }
  return fail(FJGPU_EINVAL, "unknown option " + n);
}

}  // extern "C"

namespace {

struct BatchTile { fjgpu::TileRect r; int nx, ny; uint32_t offset; };


// compact queue records (DShadowRayC, 56 bytes) where the consumers rebuild direction and distance: the lean and the curve any-hit walk,
// point / dome lights, no motion
bool scene_compact_squeue(const DScene &S)
{
  const bool lean = S.all_opaque && !S.has_curves && !S.has_motion && S.blas_base;
  const bool curve_walk = S.has_curves && !S.has_motion && S.all_opaque && S.curve_anyhit && S.inst_lds && S.n_group_nodes <= FJ_INST_LDS_NODES_CURVES &&
      S.n_instances <= FJ_INST_LDS_INSTS_CURVES && S.n_groups <= FJ_INST_LDS_GROUPS_CURVES && FJ_CURVE_QNODES && FJ_CLOSEST_QNODES;
  return (lean || curve_walk) && !S.has_area && g_compact_squeue && !getenv("FJGPU_NO_COMPACT_SQUEUE");
}

int ensure_work(fjgpu_scene *sc, size_t samples, size_t rays, int tiles, size_t tab_len)
{
  const int want_bufs = sc->overlap_now ? FJ_LREC_BUFS : 2;
  // the shadow queue is allocated at the record size the scene's walks read (56 or 80 bytes: 12 GB apart at the headline size); a scene
  // that goes back to the long record (option compact_squeue switched off between calls) gets a new arena
  const size_t rec_bytes = scene_compact_squeue(sc->S) ? sizeof(DShadowRayC) : sizeof(DShadowRay);
  if (samples > sc->work_samples || rays > sc->work_rays || tiles > sc->tiles_cap || want_bufs > sc->lrec_bufs || rec_bytes > sc->squeue_rec_bytes) {
    // the arena only GROWS: a reallocation for one reason (a small batch that wants the overlap's extra light-record buffers after a
    // whole frame, a longer tile list) keeps every other dimension at the largest size seen, or callers that alternate whole frames
    // with region / rank-share renders would free and reallocate tens of GB on every call
    samples = std::max(samples, sc->work_samples); rays = std::max(rays, sc->work_rays); tiles = std::max(tiles, sc->tiles_cap);
    const int old_bufs = sc->lrec_bufs;
    // (the extra light-record buffers of the overlap mode belong to SMALL batches: an arena that grows to a larger batch -- the second call after
    // a cold start -- takes what that batch wants, 2 x 80 B per ray, not the 4 the small one had: 21 GB at the headline size)
    sc->lrec_bufs = samples > sc->work_samples ? want_bufs : std::max(sc->lrec_bufs, want_bufs);
    sc->work.reset(new DeviceBuffers());
    // the adaptive sampler's buffers lived in the old arena (a new arena may be allocated at the
    // old one's address, so the owner pointer alone does not tell)
    sc->sort_cap = 0; sc->a_owner = nullptr; sc->a_samples = 0; sc->a_cell_bytes = 0;
    sc->d_aseen = sc->d_afinal = nullptr; sc->d_apstate = sc->d_acells = nullptr;
    DeviceBuffers &W = *sc->work;
    int e = 0;
    e |= W.alloc(samples * 2, &sc->d_suv);
    sc->d_stk = nullptr;
    if (sc->uses_sample_uid && !sc->S.has_motion && !sc->S.cam_xform) e |= W.alloc(samples * 2, &sc->d_stk);      // (tile, index) per sample: implicit camera rays
    e |= W.alloc(samples * 4, &sc->d_accum);
    for (auto &L : sc->levels) { L.rays = nullptr; L.paths = nullptr; L.cap = 0; L.keys = nullptr; }
    e |= W.alloc(rays, &sc->d_hits);
    for (int k = 0; k < sc->lrec_bufs; k++) {
      e |= W.alloc(rays, &sc->d_lrecs[k]);
      sc->d_lhair[k] = nullptr;
      if (sc->S.has_hair) e |= W.alloc(rays, &sc->d_lhair[k]);
    }
    // (rays queued once per candidate instance: room for twice the entries, or the light loop runs in many short launches)
    // (sized for the light loop's WORST case -- every (record, light) pair survives the cull -- where the memory is there: the bound decides
    // into how many launches the light loop of a level is cut, and each cut is a host round trip: a rank's share of C3 took three)
    const size_t per_ray = std::max<size_t>(sc->split_shadow ? 16 : 8, std::min<size_t>(64, (size_t) std::max(1, sc->n_light_samples) * (sc->split_shadow ? 2 : 1)));
    // (from the batch's own samples, not from the 8 M-entry floor small frames get for their RAY queues: a 64 x 64 render under a dome
    // light of 64 samples allocated 40 GB here -- ADVICE round 4.  The bound only decides into how many launches the shadow walk is cut.)
    const size_t lrec_bound = std::min<size_t>(rays, std::max<size_t>(samples * 4, (size_t) 1 << 16));
    sc->squeue_cap = std::min<size_t>(lrec_bound * per_ray, sc->squeue_max) + 4096 * 1024;   // + one chunk per resident wave
    { char *q = nullptr; e |= W.alloc(sc->squeue_cap * rec_bytes, &q); sc->d_squeue = (DShadowRay *) q; sc->squeue_rec_bytes = e ? 0 : rec_bytes; }
    // join slots of shadow rays queued once per candidate instance (DScene.shadow_join): a ray that has one
    // owns at least two queue entries
    sc->d_join = nullptr; sc->join_cap = 0;
    // (a slot per ray with >= 2 entries -- at most squeue_cap / 2 of them -- but the light loop's waves reserve slots JQ_CHUNK = 64 at a
    // time and a chunk roll-over discards up to 63 of them, i.e. up to about half of what is reserved over a launch: sized for that
    // worst case plus one chunk per resident wave and launch; 4 bytes per slot.  Beyond it the join area overflows like the queue
    // does and the frame is rendered again without the split)
    if (sc->split_shadow) { sc->join_cap = sc->squeue_cap + 1 + 32 * (persistent_threads() / 64) * 64; e |= W.alloc(sc->join_cap, &sc->d_join); }
    e |= W.alloc(1, &sc->d_cnt);
    e |= W.alloc((size_t) tiles, &sc->d_tiles);
    if (e) { sc->work.reset(); sc->work_samples = sc->work_rays = 0; sc->tiles_cap = 0; sc->lrec_bufs = std::min(old_bufs, 2); sc->squeue_rec_bytes = 0; return -1; }
    sc->work_samples = samples; sc->work_rays = rays; sc->tiles_cap = tiles;
  }
  if (tab_len > sc->tab_len) {
    // the tables live with the scene (small): 3 draws per sample of the largest tile
    std::vector<double> draws;
    fjgpu::XorShiftTable(2 * tab_len, &draws);
    if (sc->mem.upload(draws.data(), 2 * tab_len, const_cast<const double **>(&sc->d_jit))) return -1;
    if (sc->mem.upload(draws.data(), tab_len, const_cast<const double **>(&sc->d_tim))) return -1;
    sc->tab_len = tab_len;
  }
  return 0;
}

// AdaptiveGridSampler buffers: per sample the value its users see when it holds an earlier
// leaf's interpolation, the final (filtered) value, a state byte; per lattice cell a verdict
int ensure_adaptive(fjgpu_scene *sc, size_t samples, size_t cell_bytes)
{
  // (the work buffers were just sized by ensure_work: these live in the same arena, which is
  // replaced as a whole when it grows)
  if (sc->a_owner == sc->work.get() && samples <= sc->a_samples && cell_bytes <= sc->a_cell_bytes) return 0;
  DeviceBuffers &W = *sc->work;
  if (W.alloc(samples * 4, &sc->d_aseen) || W.alloc(samples * 4, &sc->d_afinal) || W.alloc(samples, &sc->d_apstate) ||
      W.alloc(cell_bytes, &sc->d_acells)) { sc->a_owner = nullptr; return -1; }
  sc->a_owner = sc->work.get(); sc->a_samples = samples; sc->a_cell_bytes = cell_bytes;
  return 0;
}

int ensure_level(fjgpu_scene *sc, int level, size_t cap)
{
  fjgpu_scene::Level &L = sc->levels[level];
  if (L.cap >= cap) return 0;
  DeviceBuffers &W = *sc->work;
  if (W.alloc(cap, &L.rays) || W.alloc(cap, &L.paths)) return -1;   // older, smaller buffers stay owned by `work`
  L.keys = nullptr;
  L.fc = nullptr;
  if (sc->has_filter_rays && level >= 1 && W.alloc(cap * 3, &L.fc)) return -1;
  if (sc->ray_sort_bits > 0 && level >= 1 && W.alloc(cap, &L.keys)) return -1;
  L.cap = cap;
  return 0;
}

// scratch of the ray-queue sort for up to `cap` rays (lives in the work arena like the queues)
int ensure_sort(fjgpu_scene *sc, size_t cap)
{
  if (sc->sort_cap >= cap) return 0;
  DeviceBuffers &W = *sc->work;
  sc->d_sort[0] = sc->d_sort[2] = nullptr;       // (keys come from the shading kernel; the values are the indices, implicit in the first pass)
  if (W.alloc(cap, &sc->d_sort[1]) || W.alloc(cap, &sc->d_sort[3])) return -1;
  sc->sort_tmp_bytes = ray_sort_temp_bytes((uint32_t) cap, sc->ray_sort_bits);
  char *tmp = nullptr;
  if (W.alloc(sc->sort_tmp_bytes, &tmp)) return -1;
  sc->d_sort_tmp = tmp;
  sc->sort_cap = cap;
  return 0;
}


}  // namespace

extern "C" {

static int render_tiles_once(fjgpu_scene *sc, const fj_render_desc *r, const int32_t *tile_ids, int n_tiles,
    float *d_fb, void *hip_stream, fjgpu_stats *stats);

int fjgpu_render_tiles(fjgpu_scene *sc, const fj_render_desc *r, const int32_t *tile_ids, int n_tiles,
    float *d_fb, void *hip_stream, fjgpu_stats *stats)
{
  if (!sc || !r || !d_fb) return fail(FJGPU_EINVAL, "null argument");
  sc->split_overflowed = false;
  int e = render_tiles_once(sc, r, tile_ids, n_tiles, d_fb, hip_stream, stats);
  if (e == FJGPU_ENOMEM && sc->split_overflowed) {
    // rays queued once per candidate instance outgrew the host's bound on the shadow queue (a scene whose
    // instance boxes overlap many times over): this scene goes back to whole rays and the tiles are rendered again
    // (said once per scene, verbose or not: from here on the scene's frames take the slower path)
    fprintf(stderr, "fjgpu: shadow queue overflow with rays split per candidate instance: rendering the tiles again without the split, "
        "which stays off for this scene (option split_shadow)\n");
    sc->split_shadow = false;
    e = render_tiles_once(sc, r, tile_ids, n_tiles, d_fb, hip_stream, stats);
  }
  if (!e) sc->render_calls++;
  return e;
}

static int render_tiles_once(fjgpu_scene *sc, const fj_render_desc *r, const int32_t *tile_ids, int n_tiles,
    float *d_fb, void *hip_stream, fjgpu_stats *stats)
{
  const bool adaptive = r->sampler_type == 1;          // Renderer::SetSamplerType, src/fj_renderer.cc:487-499
  if (r->sampler_type != 0 && !adaptive) return fail(FJGPU_EINVAL, "unknown sampler_type");
  if (adaptive && (r->adaptive_max_subdivision < 0 || r->adaptive_max_subdivision > 8 || !(r->adaptive_subdivision_threshold >= 0)))
    return fail(FJGPU_EINVAL, "adaptive_max_subdivision must be in [0, 8] and adaptive_subdivision_threshold >= 0");
  if (const char *why = bad_render(r)) return fail(FJGPU_EINVAL, why);
  HIP_TRY(hipSetDevice(sc->device));
  hipStream_t st = static_cast<hipStream_t>(hip_stream);

  std::vector<fjgpu::TileRect> all;
  fjgpu::GenerateTiles(*r, &all);
  std::vector<int> ids;
  if (tile_ids) ids.assign(tile_ids, tile_ids + n_tiles);
  else for (size_t i = 0; i < all.size(); i++) ids.push_back((int) i);
  for (int id : ids) if (id < 0 || id >= (int) all.size()) return fail(FJGPU_EINVAL, "tile id out of range");

  // samples of a tile: rate x rate per pixel plus the filter margin (fixed grid), or the
  // corners of a lattice of 2^max_subdivision cells per pixel, margin in whole pixels (adaptive
  // grid: count_samples_in_region / count_samples_in_margin, src/fj_adaptive_grid_sampler.cc:195-216)
  int margin[2];
  fjgpu::SamplerMargin(*r, margin);
  const int a_div = adaptive ? 1 << r->adaptive_max_subdivision : 1;
  if (adaptive) {
    margin[0] = (int) std::ceil((double) r->filter_w - 1);
    margin[1] = (int) std::ceil((double) r->filter_h - 1);
  }
  const size_t full_tile_samples = adaptive
      ? (size_t) (a_div * (r->tile_w + 2 * margin[0]) + 1) * (size_t) (a_div * (r->tile_h + 2 * margin[1]) + 1)
      : (size_t) (r->rate_x * r->tile_w + 2 * margin[0]) * (r->rate_y * r->tile_h + 2 * margin[1]);

  // Batch size.  The persistent traversal kernels pay a tail per launch (ray costs are
  // heavy tailed: the last waves finish long after the average one), so launches should be
  // few and large: up to 160 M samples per batch -- a whole 1080p / 64 spp frame, ~80 GB of
  // queues plus the shadow queue -- bounded by 40 % of the free HBM, unless told otherwise.
  // Measured on C3: 4 M samples per batch 473 ms/frame, 80 M 392 ms, whole frame 390 ms; later,
  // with a 196 ms frame: two batches 195.8 ms, one 192.9 ms.
  // (asked once per arena: the query is a driver round trip, and a rank's share of a frame is 20 ms; an allocation that fails because
  // somebody else took the memory meanwhile halves the batch below)
  auto query_free = [&]() {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t) 16 << 30;
    sc->free_cached = free_b + (sc->work ? sc->work->bytes : 0);   // our own work buffers are re-usable
  };
  if (!sc->work || sc->free_cached == 0) query_free();
  size_t free_now = sc->free_cached;
  // recursion levels this scene can reach: one queue per level, level = bounces so far, and a
  // bounce type only occurs if some shader of the scene emits it
  const int deepest = (sc->bounce_diffuse ? std::max(0, r->max_diffuse_depth) : 0) +
      (sc->bounce_reflect ? std::max(0, r->max_reflect_depth) : 0) + (sc->bounce_refract ? std::max(0, r->max_refract_depth) : 0);
  if (sc->levels.size() < (size_t) deepest + 1) sc->levels.resize((size_t) deepest + 1, fjgpu_scene::Level{nullptr, nullptr, 0, nullptr});
  long bt = sc->batch_tiles;
  // (a caller that renders ONE frame per scene -- SiRenderScene -- asks for batches of a few M samples: the work arena is then a few GB
  // instead of ~110 GB at the headline size, and a cold frame does not wait for the driver to hand out, and clear, that much memory)
  if (bt <= 0 && sc->batch_samples > 0) bt = std::max<long>(1, (long) ((size_t) sc->batch_samples / full_tile_samples));
  // COLD START: a scene's FIRST call renders in batches of FJ_COLD_BATCH_SAMPLES samples whatever the memory would hold -- a work arena of ~14 GB
  // instead of ~110 GB at the headline size: the first image is there after 0.15 s instead of the 2-5 s hipMalloc takes to hand out (and, after a process
  // that has just ended, clear) the large arena (profiles/r05_e2e_scene.txt).  The second call sizes its batches by memory as before and pays for the
  // growth once; a caller that renders one frame per scene never does.  ("cold_start" 0 / FJGPU_COLD_START=0: the first call already sizes by memory.)
  if (bt <= 0 && sc->render_calls == 0 && g_cold_start && !adaptive) {
    static const int on = [] { const char *e = getenv("FJGPU_COLD_START"); return e ? atoi(e) : 1; }();
    if (on) bt = std::max<long>(1, (long) ((size_t) g_cold_batch_samples / full_tile_samples));
  }
  if (bt <= 0) {
    const size_t per_sample = 32 + (size_t) (deepest + 1) * (sizeof(DRay) + sizeof(DPath) + (sc->has_filter_rays ? 12 : 0)) + sizeof(DHit) + 2 * sizeof(DLightRec) + (adaptive ? 72 : 0) +
        ((sc->ray_sort_bits > 0 && deepest >= 1) ? 24 : 0);       // (the ray sort's keys, slots, permutation and scratch)
    // (a scene without lights queues no shadow rays: the fifth of the HBM that queue may take goes to the ray queues --
    // C4, nine recursion levels: 1010 -> 994 ms per frame.  Walking a whole level in one launch with per-level hit
    // buffers and chunking only the shading was measured too: 130 -> ~50 closest-hit launches per frame, 1005 ms: dropped)
    const double share = sc->n_light_samples == 0 ? .6 : .4;
    const size_t target = std::min<size_t>((size_t) 160 << 20, (size_t) (share * (double) free_now) / per_sample);
    bt = std::max<long>(1, (long) (target / full_tile_samples));
  }
  bt = std::min<long>(bt, (long) ids.size());
  bt = std::min<long>(bt, (long) (((size_t) 1 << 31) / full_tile_samples));   // sample slots are 32-bit
  // (a batch that did not fit last time -- a batch_tiles option beyond the memory -- is not tried again frame after frame: each failed attempt
  // allocates and releases tens of GB, 15-20 s per frame measured with batch_tiles = 1020 on C4)
  if (sc->batch_fit_samples && full_tile_samples * (size_t) bt > sc->batch_fit_samples) bt = std::max<long>(1, (long) (sc->batch_fit_samples / full_tile_samples));
  if (bt < 1) bt = 1;
  // light loops beside the next levels' closest-hit walks (option overlap_shadow): on, or by the size of the batch
  sc->overlap_now = !adaptive && sc->n_light_samples > 0 && (sc->overlap == 1 || (sc->overlap == 2 && deepest >= 1 && full_tile_samples * (size_t) bt <= FJ_OVERLAP_MAX_SAMPLES));
  if (sc->overlap_now && enable_overlap(sc)) sc->overlap_now = false;
  if (!sc->read_stream) {      // (the speculative walk's counter read-back; without them the loop reads on the main stream as before)
    if (hipStreamCreateWithFlags(&sc->read_stream, hipStreamNonBlocking) != hipSuccess) sc->read_stream = nullptr;
    else if (hipEventCreateWithFlags(&sc->ev_shaded, hipEventDisableTiming) != hipSuccess) { (void) hipStreamDestroy(sc->read_stream); sc->read_stream = nullptr; sc->ev_shaded = nullptr; }
  }
  sc->squeue_max = std::max<size_t>((size_t) 4 << 20, std::min<size_t>((size_t) 512 << 20, (size_t) (.2 * (double) free_now) / sizeof(DShadowRay)));
  if (sc->n_light_samples == 0) sc->squeue_max = (size_t) 4 << 20;
  if (const char *e = getenv("FJGPU_SQUEUE_M")) sc->squeue_max = (size_t) std::max(1, atoi(e)) << 20;
  size_t cap_samples = 0, cap_rays = 0;
  bool requeried = false;
  for (;;) {
    cap_samples = full_tile_samples * (size_t) bt;
    // one level holds at most the rays its parent chunk can emit (the scheduler chunks by
    // max_children), so a queue never needs more than one entry per sample
    // (small frames get queues of at least 8 M entries: with room for the worst-case fan-out the
    // scheduler does not have to cut their levels into chunks of a few thousand rays)
    cap_rays = std::max<size_t>(cap_samples, std::min<size_t>((size_t) 8 << 20, cap_samples * 16)) + 1024;
    // the arena is about to be replaced: the shadow queue's bound follows what is free NOW, not what was free when the scene's first
    // frame asked (other scenes or processes may have allocated since)
    if (!requeried && sc->work && (cap_samples > sc->work_samples || cap_rays > sc->work_rays || (int) bt > sc->tiles_cap ||
                                   (sc->overlap_now ? FJ_LREC_BUFS : 2) > sc->lrec_bufs)) {
      requeried = true;
      query_free();
      free_now = sc->free_cached;
      if (!getenv("FJGPU_SQUEUE_M") && sc->n_light_samples != 0)
        sc->squeue_max = std::max<size_t>((size_t) 4 << 20, std::min<size_t>((size_t) 512 << 20, (size_t) (.2 * (double) free_now) / sizeof(DShadowRay)));
    }
    bool ok = ensure_work(sc, cap_samples, cap_rays, (int) bt, full_tile_samples) == 0;
    // every reachable level now, so that a failure shrinks the batch instead of ending the frame
    // (level 0 only where camera rays are explicit records: with implicit camera rays -- the rule of `implicit_cam` below -- nothing reads
    // its 112 bytes per sample, 15 GB of a cold C3 frame's allocations)
    const bool level0_implicit = !adaptive && sc->S.cam_xform == nullptr && (!sc->uses_sample_uid || !sc->S.has_motion) && !getenv("FJGPU_EXPLICIT_CAMERA_RAYS");
    for (int l = level0_implicit ? 1 : 0; ok && l <= deepest; l++) ok = ensure_level(sc, l, cap_rays) == 0;
    // ... and the ray sort's scratch (a failure here also halves the batch)
    if (ok && sc->ray_sort_bits > 0 && deepest >= 1) ok = ensure_sort(sc, cap_rays) == 0;
    if (ok) break;
    (void) hipGetLastError();
    if (bt == 1) return fail(FJGPU_ENOMEM, "device allocation failed for the wavefront work buffers");
    // another process holds part of the HBM: half the batch (the old arena is released first)
    sc->work.reset(); sc->work_samples = sc->work_rays = 0; sc->tiles_cap = 0; sc->free_cached = 0;
    for (auto &L : sc->levels) { L.rays = nullptr; L.paths = nullptr; L.cap = 0; L.keys = nullptr; }
    sc->a_owner = nullptr; sc->a_samples = 0; sc->a_cell_bytes = 0;
    sc->sort_cap = 0;
    bt = std::max<long>(1, bt / 2);
    sc->batch_fit_samples = full_tile_samples * (size_t) bt;      // (at most this from now on; a smaller one that fails lowers it again)
  }
  // adaptive grid: split / leaf byte per lattice cell of every level (4/3 of the finest level)
  const size_t a_cells0_cap = adaptive ? (size_t) bt * (size_t) (r->tile_w + 2 * margin[0]) * (size_t) (r->tile_h + 2 * margin[1]) : 0;
  const size_t a_cell_bytes = a_cells0_cap * (((((size_t) 1) << (2 * (r->adaptive_max_subdivision + 1))) - 1) / 3);
  if (adaptive && a_cells0_cap >= ((size_t) 1 << 32)) return fail(FJGPU_EINVAL, "too many lattice cells per batch: lower the batch_tiles option");
  if (adaptive && ensure_adaptive(sc, cap_samples, a_cell_bytes))
    return fail(FJGPU_ENOMEM, "device allocation failed for the adaptive sampler's buffers");

  // camera (Renderer::preprocess_camera + Camera::compute_uv_size)
  DScene S = sc->S;
  const double aspect = r->xres / (double) r->yres;
  S.cam_uv_size[1] = fjgpu::CameraUvSizeY(sc->cam_fov);
  S.cam_uv_size[0] = S.cam_uv_size[1] * aspect;
  S.time_tab = sc->d_tim; S.time_start = r->time_start; S.time_end = r->time_end;
  S.lrec_hair = nullptr;      // set per launch (double buffered)

  GenParams gp;
  gp.rate_x = r->rate_x; gp.rate_y = r->rate_y; gp.margin_x = margin[0]; gp.margin_y = margin[1];
  gp.udelta = 1. / (r->rate_x * r->xres);
  gp.vdelta = 1. / (r->rate_y * r->yres);
  gp.jitter = r->jitter;
  gp.jittered = r->jitter > 0 ? 1 : 0;
  gp.pad = 0;
  ShadeParams shp;
  shp.max_diffuse_depth = r->max_diffuse_depth; shp.max_reflect_depth = r->max_reflect_depth; shp.max_refract_depth = r->max_refract_depth;
  shp.count_all_shadow = (int) sc->count_all_shadow;
  shp.ray_capacity = (uint32_t) cap_rays; shp.light_capacity = (uint32_t) cap_rays;
  shp.fc_in = nullptr; shp.fc_out = nullptr;
  shp.next_keys = nullptr; shp.sort_bits = std::max(1, std::min(9, sc->ray_sort_bits)); shp.pad_ = 0;
  ray_sort_grid(sc->scene_box, shp.sort_bits, shp.sort_lo, shp.sort_scale);
  ShadowParams swp;
  swp.cos_half_pi = std::cos(3.14159265358979323846 / 2.);
  swp.cos_pi = std::cos(3.14159265358979323846);
  swp.pre_resolve = (S.all_opaque && !S.has_curves && !S.has_motion && S.blas_base) ? 1 : 0;   // the lean any-hit walk consumes the queue
  swp.cast_shadow = r->cast_shadow;
  // compact queue records (DShadowRayC) where the consumers rebuild direction and distance: the lean and the curve any-hit walk, point / dome lights, no motion
  S.compact_squeue = scene_compact_squeue(S) ? 1 : 0;
  S.pad_cs_ = 0;
  swp.compact = S.compact_squeue; swp.pad_ = 0;
  swp.queue_capacity = (uint32_t) sc->squeue_cap;
  swp.join_capacity = (swp.pre_resolve && sc->split_shadow && sc->d_join) ? (uint32_t) sc->join_cap : 0u;
  S.shadow_join = swp.join_capacity ? sc->d_join : nullptr;
  S.shadow_queue_cap = (uint32_t) sc->squeue_cap;
  ResolveParams rp;
  rp.xres = r->xres; rp.yres = r->yres; rp.rate_x = r->rate_x; rp.rate_y = r->rate_y;
  rp.npx_x = r->rate_x + 2 * margin[0]; rp.npx_y = r->rate_y + 2 * margin[1];
  AdaptiveParams ap;
  std::memset(&ap, 0, sizeof(ap));
  if (adaptive) {
    ap.D = r->adaptive_max_subdivision; ap.div = a_div;
    ap.margin_x = margin[0]; ap.margin_y = margin[1];
    ap.udelta = 1. / (a_div * r->xres);
    ap.vdelta = 1. / (a_div * r->yres);
    ap.jitter = r->jitter;
    ap.jittered = r->jitter > 0 ? 1 : 0;
    ap.threshold = (double) r->adaptive_subdivision_threshold;
    ap.ray_capacity = (uint32_t) cap_rays;
    // get_sampleset_in_pixel (src/fj_adaptive_grid_sampler.cc:172-193): a pixel's window starts
    // div samples after its neighbour's and is div * (1 + 2 margin) + 1 samples wide
    rp.rate_x = rp.rate_y = a_div;
    rp.npx_x = a_div * (1 + 2 * margin[0]) + 1; rp.npx_y = a_div * (1 + 2 * margin[1]) + 1;
  }
  rp.fw = (double) r->filter_w; rp.fh = (double) r->filter_h;

  fjgpu_stats acc;
  std::memset(&acc, 0, sizeof(acc));
  float ms = 0;
  // Per-launch timing without a host round trip per launch: event pairs are recorded around
  // every launch on its stream and turned into durations once, after the frame's last sync.
  struct Span { size_t a, b; double *bucket; };
  std::vector<Span> spans;
  size_t ev_used = 0;
  auto take_event = [&]() -> size_t {
    if (ev_used == sc->ev_pool.size()) { hipEvent_t e = nullptr; if (hipEventCreate(&e) != hipSuccess) return (size_t) -1; sc->ev_pool.push_back(e); }
    return ev_used++;
  };
  auto timed = [&](hipStream_t stream, double *bucket, auto &&launch) -> int {
    const size_t a = take_event(), b = take_event();
    if (a == (size_t) -1 || b == (size_t) -1) return -1;
    (void) hipEventRecord(sc->ev_pool[a], stream);
    const int le = launch();
    (void) hipEventRecord(sc->ev_pool[b], stream);
    if (le) return le;
    spans.push_back(Span{a, b, bucket});
    return 0;
  };
  // The light loop and the shadow traversal of level L are independent of the closest-hit
  // work of level L+1 (both consume what shading L produced), so they run on their own
  // stream and fill each other's tails; light records are double buffered between them.
  hipStream_t sst = (sc->overlap_now && sc->shadow_stream) ? sc->shadow_stream : st;
  unsigned shade_seq = 0;
  bool shadow_pending[FJ_LREC_BUFS] = {false, false, false, false};
  const unsigned lrec_ring = (sst != st && sc->lrec_bufs >= FJ_LREC_BUFS) ? FJ_LREC_BUFS : 2;
  int rc = 0;
  hipEvent_t *ev_all = sc->ev_frame;      // (created once per scene: two driver calls less per frame)
  if (!ev_all[0]) { HIP_TRY(hipEventCreate(&ev_all[0])); HIP_TRY(hipEventCreate(&ev_all[1])); }
  (void) hipEventRecord(ev_all[0], st);

  for (size_t b0 = 0; b0 < ids.size() && rc == 0; b0 += (size_t) bt) {
    const int nb = (int) std::min<size_t>((size_t) bt, ids.size() - b0);
    std::vector<TileDesc> td(nb);
    uint32_t off = 0, max_ts = 0, cell_off = 0;
    int max_px = 0, max_w0 = 0, max_h0 = 0;
    for (int k = 0; k < nb; k++) {
      const fjgpu::TileRect &t = all[ids[b0 + k]];
      TileDesc &d = td[k];
      d.xmin = t.xmin; d.ymin = t.ymin; d.xmax = t.xmax; d.ymax = t.ymax; d.id = t.id;
      d.nx = r->rate_x * (t.xmax - t.xmin) + 2 * margin[0];
      d.ny = r->rate_y * (t.ymax - t.ymin) + 2 * margin[1];
      d.cell_offset = cell_off; d.pad = 0;
      if (adaptive) {
        const int w0 = t.xmax - t.xmin + 2 * margin[0], h0 = t.ymax - t.ymin + 2 * margin[1];
        d.nx = a_div * w0 + 1; d.ny = a_div * h0 + 1;
        cell_off += (uint32_t) w0 * (uint32_t) h0;
        max_w0 = std::max(max_w0, w0); max_h0 = std::max(max_h0, h0);
      }
      d.sample_offset = off;
      off += (uint32_t) d.nx * (uint32_t) d.ny;
      max_ts = std::max(max_ts, (uint32_t) d.nx * (uint32_t) d.ny);
      max_px = std::max(max_px, (t.xmax - t.xmin) * (t.ymax - t.ymin));
    }
    const uint32_t n_samples = off;
    if ((hipMemcpyAsync(sc->d_tiles, td.data(), sizeof(TileDesc) * nb, hipMemcpyHostToDevice, st)) != hipSuccess) { rc = -1; break; }
    (void) hipMemsetAsync(sc->d_accum, 0, sizeof(float) * 4 * (size_t) n_samples, st);
    (void) hipMemsetAsync(sc->d_cnt, 0, sizeof(DCounters), st);

    // implicit camera rays (fjgpu_dev_shade.h): level 0 is rebuilt from the (u, v) table where it is needed
    // (scenes whose random streams are keyed by the sample's uid -- pathtracing shader, area lights -- get a (tile, index) table of 8 bytes per
    // sample beside it; time-sampled motion keeps explicit rays: the walks read the sample's time from the path record)
    const bool tk_cam = sc->uses_sample_uid && !S.has_motion && sc->d_stk != nullptr;
    const bool implicit_cam = !adaptive && S.cam_xform == nullptr && (!sc->uses_sample_uid || tk_cam) && !getenv("FJGPU_EXPLICIT_CAMERA_RAYS");
    if (!adaptive) {
      rc = timed(st, &acc.gen_ms, [&]() {
        return launch_gen_camera(st, S, gp, sc->d_tiles, nb, max_ts, sc->d_jit, sc->d_tim, sc->d_suv,
            implicit_cam ? nullptr : sc->levels[0].rays, implicit_cam ? nullptr : sc->levels[0].paths, (implicit_cam && tk_cam) ? sc->d_stk : nullptr);
      });
      if (rc) break;
      acc.rays.camera += n_samples;
    }

    // Depth-first wavefront schedule: the rays of one recursion level live in that
    // level's queue; a level is consumed in chunks small enough that the children a
    // chunk can emit (max_children per ray) fit the next level's queue, and that
    // queue is drained completely before the next chunk is taken.  Memory is bounded
    // by levels x capacity whatever the branching of the shaders (glass: 2 children,
    // pathtracing: up to 3), while plastic-only scenes run whole levels at once.
    // shadow-ray queue of this batch: the light loops of every level append, one traversal
    // launch consumes it (flush) at the end of the batch or when it could overflow
    const uint64_t sq_cap = sc->squeue_cap, sq_pad = shadow_queue_padding();
    const uint64_t nl_q = (uint64_t) std::max(1, sc->n_light_samples) * (S.shadow_join ? sc->split_kfac : 1u);
    uint64_t sq_bound = 0;             // upper bound of the entries reserved so far
    bool sq_dirty = false;
    shadow_queue_reset(sst, sc->d_cnt);
    auto flush_shadow = [&](const DScene &Sx) -> int {
      if (sq_dirty) {
        const int fe = timed(sst, &acc.shadow_walk_ms, [&]() {
          return launch_shadow_trace(sst, Sx, sc->d_squeue, sc->d_accum, sc->d_cnt, (int) sc->count_events);
        });
        if (fe) return fe;
        acc.trace_launches++; acc.shadow_walk_launches++;
        shadow_queue_reset(sst, sc->d_cnt);
      }
      sq_bound = 0; sq_dirty = false;
      return 0;
    };
    // SPECULATIVE WALK (round 5): after the shading launch of a level the host needs two counters back -- how many children, how many light
    // records -- before it can go on, and the device used to wait 40-145 us for that round trip at every level.  The children's closest-hit walk does
    // not need the host to know: it is enqueued right behind the shading launch and reads its ray count from the counter itself (DScene.trace_n_dev,
    // bounded by the chunk the host would have cut), while the counters come back on a stream of their own.  `walk_launched`: process() was called
    // for rays whose first chunk is already being walked.
    bool walk_launched = false;
    static const bool spec_env = [] { const char *e = getenv("FJGPU_SPEC_WALK"); return e ? atoi(e) != 0 : true; }();
    const bool speculate = g_spec_walk && spec_env && sc->ray_sort_bits <= 0 && sc->read_stream && sc->ev_shaded;
    std::function<int(int, uint32_t)> process = [&](int level, uint32_t count) -> int {
      // rays of the deepest reachable level cannot emit children (has_reached_bounce_limit):
      // there is no next queue
      const bool can_emit = level < deepest;
      bool skip_walk = walk_launched;      // (the first chunk's walk)
      walk_launched = false;
      if (can_emit && ensure_level(sc, level + 1, cap_rays)) return fail(FJGPU_ENOMEM, "device allocation failed for a ray queue level");
      const uint32_t kids = (uint32_t) std::max(1, sc->max_children);
      const uint32_t chunk_max = kids <= 1 ? count : std::max<uint32_t>(1u, (uint32_t) (cap_rays / kids));
      for (uint32_t off = 0; off < count; off += chunk_max) {
        const uint32_t n = std::min(chunk_max, count - off);
        const bool implicit = implicit_cam && level == 0;
        const DRay *rays = implicit ? nullptr : sc->levels[level].rays + off;
        const DPath *paths = implicit ? nullptr : sc->levels[level].paths + off;
        (void) hipMemsetAsync(&sc->d_cnt->next_count, 0, sizeof(uint32_t) * 2, st);   // next_count + light_count
        // secondary rays leave the shading kernel in emission order: walk them in (octant, cell) order
        DScene St = S;
        if (implicit) { St.cam_uv = sc->d_suv + 2 * (size_t) off; St.cam_slot0 = off; St.cam_tk = tk_cam ? sc->d_stk + 2 * (size_t) off : nullptr; }
        int e = 0;
        if (sc->ray_sort_bits > 0 && level >= 1 && (long) n >= g_ray_sort_min) {
          if (ensure_sort(sc, cap_rays)) return fail(FJGPU_ENOMEM, "device allocation failed for the ray sort");   // (sized with the queues: cannot fail here)
          // (the keys were written by the shading kernel that emitted these rays: ShadeParams.next_keys)
          e = timed(st, &acc.sort_ms, [&]() {
            return launch_ray_sort_keyed(st, sc->levels[level].keys + off, n, sc->ray_sort_bits, sc->d_sort[1], sc->d_sort[3],
                sc->d_sort_tmp, sc->sort_tmp_bytes);
          });
          if (e) return e;
          St.ray_perm = sc->d_sort[3];
          acc.rays_sorted += n;
        }
        if (!skip_walk) {
          e = timed(st, &acc.closest_ms, [&]() {
            return launch_trace_closest(st, St, rays, paths, sc->d_hits, n, sc->d_cnt, (int) sc->count_events);
          });
          if (e) return e;
          acc.trace_launches++; acc.closest_launches++;
        }
        skip_walk = false;
        const unsigned lb = shade_seq++ % lrec_ring;   // light-record buffer of this shading call
        if (shadow_pending[lb] && sst != st) (void) hipStreamWaitEvent(st, sc->ev_shadow_done[lb], 0);
        shadow_pending[lb] = false;
        DScene Sl = S;
        if (implicit) { Sl.cam_uv = St.cam_uv; Sl.cam_slot0 = off; Sl.cam_tk = St.cam_tk; }
        Sl.lrec_hair = sc->d_lhair[lb];
        ShadeParams shl = shp;
        shl.next_keys = (can_emit && sc->ray_sort_bits > 0) ? sc->levels[level + 1].keys : nullptr;
        shl.fc_in = (!implicit && sc->levels[level].fc) ? sc->levels[level].fc + 3 * (size_t) off : nullptr;
        shl.fc_out = can_emit ? sc->levels[level + 1].fc : nullptr;
        e = timed(st, &acc.shade_ms, [&]() {
          return launch_shade(st, Sl, shl, rays, paths, sc->d_hits, n, sc->d_accum,
              can_emit ? sc->levels[level + 1].rays : nullptr, can_emit ? sc->levels[level + 1].paths : nullptr, sc->d_lrecs[lb], sc->d_cnt);
        });
        if (e) return e;
        DCounters hc;
        bool launched_next = false;
        // (with the light loops on their own stream the ORDER of the launches decides who is resident first: level 0's light loop -- the big one -- must
        // reach the device before level 1's walk fills it, or the two take turns instead of overlapping: C2 93.8 -> 97.3 ms with the walk first; from
        // level 1 on the light loops are small and the round trip is what costs)
        if (speculate && can_emit && (sst == st || level >= 1)) {
          // the children's walk behind the shading launch, the counters back on the side stream (which waits for the shading launch only)
          if (hipEventRecord(sc->ev_shaded, st) != hipSuccess) return -1;
          DScene Sn = S;
          Sn.trace_n_dev = &sc->d_cnt->next_count;
          // (sized by what THIS launch can emit -- n rays, at most `kids` children each -- not by the queue's capacity: a deep level of a few
          //  hundred rays no longer starts a full persistent grid whose waves read the count and leave)
          const uint32_t chunk_cap = kids <= 1 ? (uint32_t) std::min<size_t>(cap_rays, 0xffffffffu) : std::max<uint32_t>(1u, (uint32_t) (cap_rays / kids));
          const uint32_t next_chunk = (uint32_t) std::min<uint64_t>(chunk_cap, (uint64_t) n * kids);
          e = timed(st, &acc.closest_ms, [&]() {
            return launch_trace_closest(st, Sn, sc->levels[level + 1].rays, sc->levels[level + 1].paths, sc->d_hits, next_chunk, sc->d_cnt, (int) sc->count_events);
          });
          if (e) return e;
          launched_next = true;
          if (hipStreamWaitEvent(sc->read_stream, sc->ev_shaded, 0) != hipSuccess ||
              hipMemcpyAsync(&hc, sc->d_cnt, sizeof(hc), hipMemcpyDeviceToHost, sc->read_stream) != hipSuccess || hipStreamSynchronize(sc->read_stream) != hipSuccess) return -1;
        }
        else if (hipMemcpyAsync(&hc, sc->d_cnt, sizeof(hc), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
        if (hc.overflow) { sc->split_overflowed = S.shadow_join != nullptr; return fail(FJGPU_ENOMEM, "ray queue overflow: lower the batch_tiles option"); }
        if (launched_next && hc.next_count) { acc.trace_launches++; acc.closest_launches++; }       // (a walk of no rays is not counted ...
        else if (launched_next) spans.pop_back();                                                   //  ... and not timed: its span was the last one recorded)
        if (hc.light_count) {
          // shading has completed (the host just synchronised with it): no event needed.
          // The queue holds hc.shadow_count entries so far (exact when the shadow work runs on
          // this stream, a lower bound otherwise: then the host-side bound is kept).
          if (sst == st) sq_bound = hc.shadow_count;
          uint32_t b = 0;
          bool bound_is_exact = sst == st;
          while (b < hc.light_count) {
            const uint64_t room = sq_cap > (uint64_t) sq_bound + sq_pad ? sq_cap - sq_bound - sq_pad : 0;
            uint32_t can = (uint32_t) std::min<uint64_t>(room / nl_q, hc.light_count - b);
            if (can < hc.light_count - b && !bound_is_exact && sst == st) {
              // the bound assumed that every (record, light) pair survived the cull: ask the
              // device how full the queue really is before paying for a traversal launch
              uint32_t used = 0;
              if (hipMemcpyAsync(&used, &sc->d_cnt->shadow_count, sizeof(used), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1;
              sq_bound = used;
              bound_is_exact = true;
              continue;
            }
            if (can == 0 || (can < hc.light_count - b && can < (1u << 16))) {
              // not worth a light-loop launch of its own: empty the queue first (an empty queue
              // that still cannot take one record's rays takes them one record at a time)
              if (sq_bound > 0) { e = flush_shadow(Sl); if (e) return e; continue; }
              if (can == 0) can = 1;
            }
            acc.light_loop_launches++;
            e = timed(sst, &acc.light_loop_ms, [&]() {
              return launch_shadow_cull(sst, Sl, swp, sc->d_lrecs[lb], b, b + can, sc->d_accum, sc->d_squeue, sc->d_cnt, (int) sc->count_events,
                  sst != st ? FJ_OVERLAP_CULL_BLOCKS : 0);
            });
            if (e) return e;
            sq_bound += (uint64_t) can * nl_q + sq_pad;      // worst case: every pair survives, every wave leaves a padded chunk
            bound_is_exact = false;
            sq_dirty = true;
            b += can;
          }
          if (sst != st) { (void) hipEventRecord(sc->ev_shadow_done[lb], sst); shadow_pending[lb] = true; }
          // overlap: the shadow rays of level 0 -- most of a frame's -- are walked NOW, on the shadow stream, while the main stream runs the
          // deeper levels' closest-hit walks, which are a few long rays each (a launch of its own per level, latency bound): the big walk hides them
          // (early_shadow = k: the queue is flushed after level k - 1's light loops -- 2: levels 0 and 1, nearly all of a frame's shadow rays, are walked
          //  beside the closest-hit walks of levels >= 2, which are a few long rays each)
          if (sst != st && sc->early_shadow > 0 && level == (int) sc->early_shadow - 1 && can_emit && hc.next_count) { e = flush_shadow(Sl); if (e) return e; }
        }
        if (hc.next_count) {
          if (!can_emit) return fail(FJGPU_EINVAL, "ray recursion deeper than the depth limits allow");
          walk_launched = launched_next;
          e = process(level + 1, hc.next_count);
          if (e) return e;
        }
      }
      return 0;
    };
    // the filter (and the adaptive sampler's decisions) need every shadow contribution so far
    auto finish_shadow = [&]() -> int {
      DScene Sf = S; Sf.lrec_hair = nullptr;
      const int fe = flush_shadow(Sf);
      if (fe) return fe;
      if (sst != st) { (void) hipEventRecord(sc->ev_shadow_done[0], sst); shadow_pending[0] = true; }
      for (int k = 0; k < FJ_LREC_BUFS; k++) if (shadow_pending[k] && sst != st) { (void) hipStreamWaitEvent(st, sc->ev_shadow_done[k], 0); shadow_pending[k] = false; }
      return 0;
    };
    if (!adaptive) {
      rc = process(0, n_samples);
      if (rc) break;
      rc = finish_shadow();
      if (rc) break;
    } else {
      // AdaptiveGridSampler level by level (fjgpu_dev_adaptive.h): points -> wavefront -> decisions
      ap.cells0 = cell_off;
      (void) hipMemsetAsync(sc->d_apstate, 0, (size_t) n_samples, st);
      (void) hipMemsetAsync(sc->d_acells, 0, (size_t) cell_off * (((((size_t) 1) << (2 * (ap.D + 1))) - 1) / 3), st);
      rc = timed(st, &acc.gen_ms, [&]() { return launch_adaptive_uv(st, ap, sc->d_tiles, nb, max_ts, sc->d_jit, sc->d_suv); });
      for (int level = 0; level <= ap.D && rc == 0; level++) {
        ap.level = level;
        (void) hipMemsetAsync(&sc->d_cnt->cam_count, 0, sizeof(uint32_t), st);
        const uint32_t max_pts = (uint32_t) ((max_w0 << level) + 1) * (uint32_t) ((max_h0 << level) + 1);
        rc = timed(st, &acc.gen_ms, [&]() {
          return launch_adaptive_points(st, S, ap, sc->d_tiles, nb, max_pts, sc->d_acells, sc->d_suv, sc->d_accum, sc->d_apstate,
              sc->d_aseen, sc->levels[0].rays, sc->levels[0].paths, sc->d_cnt);
        });
        if (rc) break;
        DCounters hc;
        if (hipMemcpyAsync(&hc, sc->d_cnt, sizeof(hc), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rc = -1; break; }
        if (hc.overflow) { rc = fail(FJGPU_ENOMEM, "camera ray queue overflow in the adaptive sampler"); break; }
        acc.rays.camera += hc.cam_count;
        if (hc.cam_count) {
          rc = process(0, hc.cam_count);
          if (rc) break;
          rc = finish_shadow();
          if (rc) break;
        }
        rc = timed(st, &acc.resolve_ms, [&]() {
          return launch_adaptive_decide(st, ap, sc->d_tiles, nb, (uint32_t) (max_w0 << level) * (uint32_t) (max_h0 << level),
              sc->d_acells, sc->d_accum, sc->d_apstate, sc->d_aseen);
        });
      }
      if (rc) break;
      rc = timed(st, &acc.resolve_ms, [&]() {
        return launch_adaptive_fill(st, ap, sc->d_tiles, nb, max_ts, sc->d_acells, sc->d_accum, sc->d_apstate, sc->d_aseen, sc->d_afinal);
      });
      if (rc) break;
    }
    rc = timed(st, &acc.resolve_ms, [&]() {
      return launch_resolve(st, rp, sc->d_tiles, nb, max_px, sc->d_suv, sc->d_accum, adaptive ? sc->d_afinal : nullptr, d_fb);
    });
    if (rc) break;
    DCounters hc;
    if (hipMemcpyAsync(&hc, sc->d_cnt, sizeof(hc), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { rc = -1; break; }
    if (hc.overflow) { sc->split_overflowed = S.shadow_join != nullptr; rc = fail(FJGPU_ENOMEM, "wavefront queue overflow: lower the batch_tiles option"); break; }
    acc.rays.shadow += hc.rays[CXT_SHADOW_RAY];
    acc.rays.diffuse += hc.rays[CXT_DIFFUSE_RAY];
    acc.rays.reflect += hc.rays[CXT_REFLECT_RAY];
    acc.rays.refract += hc.rays[CXT_REFRACT_RAY];
    acc.nodes_visited += hc.nodes;
    acc.prims_tested += hc.prims;
    acc.insts_tested += hc.insts;
    acc.rays_traced += hc.traced;
    acc.shadow_traversed += hc.squeued;
    acc.shadow_nodes += hc.sh_nodes; acc.shadow_prims += hc.sh_prims; acc.shadow_insts += hc.sh_insts;
    acc.batches++;
    if (sc->batch_fn) {
      // (the stream was synchronised above: the batch's pixels are final in d_fb)
      const std::vector<int32_t> bids(ids.begin() + (long) b0, ids.begin() + (long) b0 + nb);
      if (sc->batch_fn(sc->batch_user, sc->device, bids.data(), nb)) { acc.interrupted = 1; break; }
    }
  }
  (void) hipEventRecord(ev_all[1], st);
  hipError_t se = hipStreamSynchronize(st);
  if (sst != st) { const hipError_t s2 = hipStreamSynchronize(sst); if (se == hipSuccess) se = s2; }
  if (rc == 0 && se == hipSuccess) {
    (void) hipEventElapsedTime(&ms, ev_all[0], ev_all[1]);
    acc.total_ms = ms;
    for (const Span &sp : spans)
      if (hipEventElapsedTime(&ms, sc->ev_pool[sp.a], sc->ev_pool[sp.b]) == hipSuccess) *sp.bucket += ms;
    acc.closest_ms += acc.sort_ms;       // the sort is part of what the closest-hit side costs
    acc.trace_ms = acc.closest_ms + acc.light_loop_ms + acc.shadow_walk_ms;
  }
  if (rc > 0 || rc == -1) return fail(FJGPU_ENODEV, std::string("HIP failure in the wavefront loop: ") + hipGetErrorString(hipGetLastError()));
  if (rc) return rc;
  if (se != hipSuccess) return fail(FJGPU_ENODEV, std::string("stream synchronize: ") + hipGetErrorString(se));
  if (stats) *stats = acc;
  return 0;
}

// One frame on G devices of this process: the reference's worker pool (execute_rendering,
// src/fj_renderer.cc:747-791; MtRunParallelLoop, src/fj_multi_thread.cc:86-132) with GPUs for
// workers.  Tile k of the list goes to scene k % G (interleaved deal: neighbouring tiles cost
// alike), every device renders its share on its own host thread, packs the finished tiles into
// one slab, and the slab crosses xGMI as one peer copy into the first device, which scatters it
// into the frame; the frame goes to the host once.
int fjgpu_render_frame_multi(fjgpu_scene *const *scenes, int n_scenes, const fj_render_desc *r,
    const int32_t *tile_ids, int n_tiles, float *h_fb, fjgpu_stats *stats)
{
  if (!scenes || n_scenes < 1 || !r || !h_fb) return fail(FJGPU_EINVAL, "null argument");
  for (int k = 0; k < n_scenes; k++) if (!scenes[k]) return fail(FJGPU_EINVAL, "null scene");
  if (const char *why = bad_render(r)) return fail(FJGPU_EINVAL, why);
  std::vector<fjgpu::TileRect> all;
  fjgpu::GenerateTiles(*r, &all);
  std::vector<int32_t> ids;
  if (tile_ids) ids.assign(tile_ids, tile_ids + std::max(0, n_tiles));
  else for (size_t i = 0; i < all.size(); i++) ids.push_back((int32_t) i);
  for (int32_t id : ids) if (id < 0 || id >= (int32_t) all.size()) return fail(FJGPU_EINVAL, "tile id out of range");
  const int G = n_scenes;
  const size_t npx = (size_t) r->xres * r->yres;
  const int tile_px = r->tile_w * r->tile_h;
  std::vector<std::vector<int32_t>> mine(G);
  for (size_t k = 0; k < ids.size(); k++) mine[k % G].push_back(ids[k]);
  fjgpu_scene *first = scenes[0];

  // the first device's frame and the staging area for the other devices' slabs
  HIP_TRY(hipSetDevice(first->device));
  if (grow((void **) &first->d_frame, &first->d_frame_n, npx * 4, sizeof(float))) return fail(FJGPU_ENOMEM, "device allocation failed for the frame");
  HIP_TRY(hipMemset(first->d_frame, 0, npx * 4 * sizeof(float)));
  std::vector<size_t> stage_off(G, 0);
  size_t stage_px = 0, rects_n = 0;
  for (int d = 1; d < G; d++) { stage_off[d] = stage_px; stage_px += mine[d].size() * (size_t) tile_px; rects_n += mine[d].size(); }
  if (G > 1) {
    if (grow((void **) &first->d_slab, &first->d_slab_n, stage_px * 4, sizeof(float)) ||
        grow((void **) &first->d_rects, &first->d_rects_n, rects_n * 4, sizeof(int32_t)))
      return fail(FJGPU_ENOMEM, "device allocation failed for the tile slabs");
  }

  // The exchange: one peer copy per device (default: nothing to bring up for a frame), or -- option "multi_exchange" 1 / FJGPU_MULTI_RCCL=1, devices all
  // distinct -- RCCL's grouped point-to-point calls: every device's thread sends its slab to rank 0, whose thread posts all receives in one group
  // (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd over xGMI).  A communicator set that cannot be created falls back to the peer copies.
  const std::vector<Rccl::comm_t> *comms = nullptr;
  {
    bool want = g_multi_exchange == 1;
    if (const char *e = getenv("FJGPU_MULTI_RCCL")) want = atoi(e) != 0;
    std::vector<int> devs(G);
    for (int d = 0; d < G; d++) devs[d] = scenes[d]->device;
    std::vector<int> uniq = devs;
    std::sort(uniq.begin(), uniq.end());
    const bool distinct = std::adjacent_find(uniq.begin(), uniq.end()) == uniq.end();
    if (want && G > 1 && distinct) {
      std::string why;
      comms = rccl_comms(devs, &why);
      if (!comms && getenv("FJGPU_VERBOSE")) fprintf(stderr, "fjgpu: RCCL exchange not available (%s): peer copies\n", why.c_str());
    }
  }
  std::vector<int> rcs(G, 0);
  std::vector<std::string> errs(G);
  std::vector<fjgpu_stats> sts(G);
  auto work = [&](int d) {
    fjgpu_scene *sc = scenes[d];
    fjgpu_stats &st = sts[d];
    std::memset(&st, 0, sizeof(st));
    auto bad = [&](int code, const std::string &m) { rcs[d] = code; errs[d] = m; };
    if (hipSetDevice(sc->device) != hipSuccess) return bad(FJGPU_ENODEV, "hipSetDevice failed");
    if (mine[d].empty()) return;
    float *fb = first->d_frame;
    if (d > 0) {
      // (a second scene on the first device -- a test set-up -- still renders into its own frame)
      if (grow((void **) &sc->d_frame, &sc->d_frame_n, npx * 4, sizeof(float)) ||
          grow((void **) &sc->d_slab, &sc->d_slab_n, mine[d].size() * (size_t) tile_px * 4, sizeof(float)) ||
          grow((void **) &sc->d_rects, &sc->d_rects_n, mine[d].size() * 4, sizeof(int32_t)))
        return bad(FJGPU_ENOMEM, "device allocation failed for a device's frame");
      fb = sc->d_frame;
      // (tiles a batch callback keeps from being rendered are packed and sent all the same: they read 0)
      if (sc->batch_fn && hipMemset(fb, 0, npx * 4 * sizeof(float)) != hipSuccess) return bad(FJGPU_ENODEV, "frame clear failed");
    }
    const int e = fjgpu_render_tiles(sc, r, mine[d].data(), (int) mine[d].size(), fb, nullptr, &st);
    if (e) return bad(e, fjgpu_last_error());
    if (d > 0) {
      std::vector<int32_t> rects(mine[d].size() * 4);
      for (size_t k = 0; k < mine[d].size(); k++) {
        const fjgpu::TileRect &t = all[mine[d][k]];
        rects[4 * k] = t.xmin; rects[4 * k + 1] = t.ymin; rects[4 * k + 2] = t.xmax; rects[4 * k + 3] = t.ymax;
      }
      if (hipMemcpy(sc->d_rects, rects.data(), rects.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess)
        return bad(FJGPU_ENODEV, "rect upload failed");
      if (launch_move_tiles(nullptr, false, fb, r->xres, sc->d_rects, (int) mine[d].size(), tile_px, sc->d_slab)) return bad(FJGPU_ENODEV, "pack launch failed");
      // one peer copy per device: over xGMI when peer access is available, staged by the runtime otherwise.  (The RCCL exchange is NOT posted here:
      // a device whose frame failed would leave its partner waiting in the group for ever -- it runs below, after the join, once every device succeeded.)
      const size_t bytes = mine[d].size() * (size_t) tile_px * 4 * sizeof(float);
      if (comms) { if (hipStreamSynchronize(nullptr) != hipSuccess) return bad(FJGPU_ENODEV, "tile pack failed"); }
      else if (hipMemcpyPeer(first->d_slab + stage_off[d] * 4, first->device, sc->d_slab, sc->device, bytes) != hipSuccess)
        return bad(FJGPU_ENODEV, std::string("peer copy of a tile slab: ") + hipGetErrorString(hipGetLastError()));
    }
  };
  if (G == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int d = 0; d < G; d++) th.emplace_back(work, d);
    for (auto &t : th) t.join();
  }
  for (int d = 0; d < G; d++) if (rcs[d]) return fail(rcs[d], "device " + std::to_string(scenes[d]->device) + ": " + errs[d]);

  if (comms) {
    // The slab exchange in RCCL's spelling, collective over SUCCESS: it is entered only when every device rendered and packed its share (checked above),
    // from this one thread, as ONE group -- every non-empty device's send to rank 0 on its own communicator and rank 0's receives, also when rank 0
    // itself holds no tile -- so no rank can be left waiting for a partner that returned early.
    Rccl &R = rccl();
    int e = R.GroupStart();
    for (int q = 1; q < G && !e; q++) {
      if (mine[q].empty()) continue;
      const size_t count = mine[q].size() * (size_t) tile_px * 4;
      e = R.Send(scenes[q]->d_slab, count, Rccl::kFloat, 0, (*comms)[q], nullptr);
      if (!e) e = R.Recv(first->d_slab + stage_off[q] * 4, count, Rccl::kFloat, q, (*comms)[0], nullptr);
    }
    const int e2 = R.GroupEnd();
    if (e || e2) return fail(FJGPU_ENODEV, std::string("RCCL exchange of the tile slabs: ") + (R.GetErrorString ? R.GetErrorString(e ? e : e2) : "error"));
    for (int d = 0; d < G; d++) {
      HIP_TRY(hipSetDevice(scenes[d]->device));
      HIP_TRY(hipStreamSynchronize(nullptr));
    }
  }
  HIP_TRY(hipSetDevice(first->device));
  if (G > 1) {
    std::vector<int32_t> rects;
    for (int d = 1; d < G; d++)
      for (int32_t id : mine[d]) { const fjgpu::TileRect &t = all[id]; rects.insert(rects.end(), {t.xmin, t.ymin, t.xmax, t.ymax}); }
    if (!rects.empty()) {
      HIP_TRY(hipMemcpy(first->d_rects, rects.data(), rects.size() * sizeof(int32_t), hipMemcpyHostToDevice));
      if (launch_move_tiles(nullptr, true, first->d_frame, r->xres, first->d_rects, (int) (rects.size() / 4), tile_px, first->d_slab))
        return fail(FJGPU_ENODEV, "unpack launch failed");
    }
  }
  HIP_TRY(hipMemcpy(h_fb, first->d_frame, npx * 4 * sizeof(float), hipMemcpyDeviceToHost));
  if (stats) for (int d = 0; d < G; d++) stats[d] = sts[d];
  return 0;
}

int fjgpu_pack_tiles(const float *d_fb, int xres, const int32_t *d_rects, int n_tiles, int tile_px, float *d_slab, void *hip_stream)
{
  if (!d_fb || !d_rects || !d_slab || xres <= 0 || tile_px <= 0 || n_tiles < 0) return fail(FJGPU_EINVAL, "bad tile slab arguments");
  if (launch_move_tiles(static_cast<hipStream_t>(hip_stream), false, const_cast<float *>(d_fb), xres, d_rects, n_tiles, tile_px, d_slab))
    return fail(FJGPU_ENODEV, "tile pack launch failed");
  return 0;
}

int fjgpu_unpack_tiles(float *d_fb, int xres, const int32_t *d_rects, int n_tiles, int tile_px, const float *d_slab, void *hip_stream)
{
  if (!d_fb || !d_rects || !d_slab || xres <= 0 || tile_px <= 0 || n_tiles < 0) return fail(FJGPU_EINVAL, "bad tile slab arguments");
  if (launch_move_tiles(static_cast<hipStream_t>(hip_stream), true, d_fb, xres, d_rects, n_tiles, tile_px, const_cast<float *>(d_slab)))
    return fail(FJGPU_ENODEV, "tile unpack launch failed");
  return 0;
}

int fjgpu_render_frame(fjgpu_scene *sc, const fj_render_desc *r, float *h_fb, fjgpu_stats *stats)
{
  fjgpu_scene *one[1] = {sc};
  if (!sc) return fail(FJGPU_EINVAL, "null argument");
  return fjgpu_render_frame_multi(one, 1, r, nullptr, 0, h_fb, stats);
}

int fjgpu_trace(fjgpu_scene *sc, int group, int n, const double *rays, double *out_t, int32_t *out_ids,
    double *out_uv, fjgpu_stats *stats)
{
  if (!sc || !rays || !out_t || !out_ids || n < 0) return fail(FJGPU_EINVAL, "null argument");
  if (group < 0 || group >= sc->S.n_groups) return fail(FJGPU_EINVAL, "group out of range");
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(sc->device));
  DeviceBuffers W;
  DRay *d_rays; DHit *d_hits; DCounters *d_cnt;
  double *d_ranges;
  if (W.alloc((size_t) n, &d_rays) || W.alloc((size_t) n * 2, &d_ranges) || W.alloc((size_t) n, &d_hits) || W.alloc(1, &d_cnt))
    return fail(FJGPU_ENOMEM, "device allocation failed");
  {
    // the caller's rays are Ray records of 8 doubles (o, d, tmin, tmax: include/fjgpu.h); the device keeps origin / direction and the ranges apart
    std::vector<DRay> hr((size_t) n);
    std::vector<double> hg((size_t) n * 2);
    for (int i = 0; i < n; i++) {
      for (int k = 0; k < 3; k++) { hr[i].o[k] = rays[8 * (size_t) i + k]; hr[i].d[k] = rays[8 * (size_t) i + 3 + k]; }
      hg[2 * (size_t) i] = rays[8 * (size_t) i + 6]; hg[2 * (size_t) i + 1] = rays[8 * (size_t) i + 7];
    }
    HIP_TRY(hipMemcpy(d_rays, hr.data(), sizeof(DRay) * (size_t) n, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_ranges, hg.data(), sizeof(double) * 2 * (size_t) n, hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMemset(d_cnt, 0, sizeof(DCounters)));
  DScene S = sc->S;
  S.target_group = group;
  S.trace_ranges = d_ranges;
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  (void) hipEventRecord(e0, nullptr);
  const int le = launch_trace_closest(nullptr, S, d_rays, nullptr, d_hits, (uint32_t) n, d_cnt, (int) sc->count_events);
  (void) hipEventRecord(e1, nullptr);
  if (le) return fail(FJGPU_ENODEV, std::string("trace launch: ") + hipGetErrorString((hipError_t) le));
  HIP_TRY(hipDeviceSynchronize());
  float ms = 0;
  (void) hipEventElapsedTime(&ms, e0, e1);
  (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
  std::vector<DHit> h((size_t) n);
  HIP_TRY(hipMemcpy(h.data(), d_hits, sizeof(DHit) * (size_t) n, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) {
    out_t[i] = h[i].inst >= 0 ? h[i].t : DBL_MAX;
    out_ids[2 * i] = h[i].inst;
    out_ids[2 * i + 1] = h[i].inst >= 0 ? h[i].prim : -1;
    if (out_uv) { out_uv[2 * i] = h[i].inst >= 0 ? h[i].u : 0; out_uv[2 * i + 1] = h[i].inst >= 0 ? h[i].v : 0; }
  }
  if (stats) {
    std::memset(stats, 0, sizeof(*stats));
    DCounters hc;
    HIP_TRY(hipMemcpy(&hc, d_cnt, sizeof(hc), hipMemcpyDeviceToHost));
    stats->nodes_visited = hc.nodes; stats->prims_tested = hc.prims; stats->insts_tested = hc.insts; stats->rays_traced = hc.traced;
    stats->trace_ms = ms; stats->total_ms = ms; stats->trace_launches = 1;
  }
  return 0;
}

}  // extern "C"
