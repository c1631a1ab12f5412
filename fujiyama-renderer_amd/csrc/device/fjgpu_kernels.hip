// fjgpu_kernels.hip -- hand-written HIP kernels of the MI355X (gfx950, wave64)
// ray-intersection + integrator core.  Compiled with -ffp-contract=off: every
// FP64 expression that decides a hit or feeds geometry keeps the reference's
// operation order (citations per function); culling arithmetic (slab tests on
// widened boxes) is free to differ because it can only reject provable misses.
//
// Kernel map -- one translation unit, the device code lives in headers:
//   fjgpu_dev_math.h      vectors, slab / exact box tests, transforms at a time, tri_ray
//   fjgpu_dev_curve.h     curve_ray (Bezier ribbons) + the reference grid's cell listing
//   fjgpu_dev_traverse.h  traverse_persistent, k_trace_closest   (Accelerator::Intersect:
//                         instance loop -> 4-wide BLAS, LDS stack -> FP64 Moller-Trumbore)
//   fjgpu_dev_shade.h     k_gen_camera (sampler + Camera::GetRay), k_shade (trace_surface's
//                         attribute setup + the shader plugins; light records and child rays)
//   fjgpu_dev_shadow.h    k_shadow_cull (SlIlluminance light loop), k_shadow_trace
//   fjgpu_dev_anyhit.h    k_shadow_anyhit (lean any-hit walk: phase-scheduled, f32 slabs)
//   fjgpu_dev_anyhit_curves.h  k_shadow_anyhit_curves (the same scheduling for scenes with curve sets: + a ribbon phase)
//   fjgpu_dev_flat.h      k_trace_closest_flat (closest-hit walk of FLAT groups: one world-space culling tree per group, exact tests in object space)
//   here                  k_resolve (reconstruct_image / apply_pixel_filter), host launchers
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>

#include "fjgpu_types.h"
#include "fjgpu_kernels.h"
#include "fjgpu_xform_math.h"

#define BLOCK 256

#include "fjgpu_dev_math.h"
#include "fjgpu_tri_filter.h"
#include "fjgpu_dev_curve.h"
#include "fjgpu_dev_traverse.h"
#include "fjgpu_dev_flat.h"
#include "fjgpu_dev_shade.h"
#include "fjgpu_dev_shadow.h"
#include "fjgpu_dev_anyhit.h"
#include "fjgpu_dev_anyhit_curves.h"
#include "fjgpu_dev_adaptive.h"

// ------------------------------------------------------------------ k_resolve
// reconstruct_image + apply_pixel_filter (src/fj_renderer.cc:939-995) with
// eval_gaussian (src/fj_filter.cc:49-58).  One wave per pixel: the lanes take the
// (rate + 2 margin)^2 samples of the pixel's window in row-major order, so a load
// instruction reads whole window rows (npx_x * 16 B contiguous each) instead of 64
// addresses 128 B apart (the one-lane-per-pixel version spent 17.6 ms per C3 frame on L2
// requests, this one is bandwidth bound).  The weighted sums are reduced in f64 by a
// butterfly and rounded to f32 once; the reference adds the same f64 products into f32
// accumulators one by one -- the difference is the accumulators' rounding (~1e-7).
template <typename TData4>       // float4: samples as traced; double4: the adaptive sampler's interpolated samples
__global__ void __launch_bounds__(BLOCK) k_resolve(ResolveParams rp, const TileDesc *tiles,
    const double *s_uv, const TData4 *s_data, float *fb)
{
  const TileDesc T = tiles[blockIdx.y];
  const int tw = T.xmax - T.xmin, th = T.ymax - T.ymin;
  const int k = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);     // pixel of this wave
  if (k >= tw * th) return;
  const unsigned lane = __lane_id();
  const int px = T.xmin + k % tw, py = T.ymin + k / tw;
  const size_t base = (size_t) T.sample_offset + (size_t) (py - T.ymin) * rp.rate_y * T.nx + (size_t) (px - T.xmin) * rp.rate_x;
  const int nwin = rp.npx_x * rp.npx_y;
  double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0, wsum = 0;
  for (int w = (int) lane; w < nwin; w += 64) {
    const int sy = w / rp.npx_x, sx = w - sy * rp.npx_x;
    const size_t s = base + (size_t) sy * T.nx + sx;
    const double2 uv = reinterpret_cast<const double2 *>(s_uv)[s];
    const TData4 d = s_data[s];
    const double filtx = rp.xres * uv.x - (px + .5);
    const double filty = rp.yres * (1 - uv.y) - (py + .5);
    const double xx = 2 * filtx / rp.fw;
    const double yy = 2 * filty / rp.fh;
    const double wgt = exp(-2 * (xx * xx + yy * yy));
    acc0 += wgt * (double) d.x;
    acc1 += wgt * (double) d.y;
    acc2 += wgt * (double) d.z;
    acc3 += wgt * (double) d.w;
    wsum += wgt;
  }
  for (int off = 32; off > 0; off >>= 1) {
    acc0 += __shfl_xor(acc0, off);
    acc1 += __shfl_xor(acc1, off);
    acc2 += __shfl_xor(acc2, off);
    acc3 += __shfl_xor(acc3, off);
    wsum += __shfl_xor(wsum, off);
  }
  if (lane == 0) {
    const float inv_sum = 1.f / (float) wsum;
    float4 out;
    out.x = (float) acc0 * inv_sum; out.y = (float) acc1 * inv_sum; out.z = (float) acc2 * inv_sum; out.w = (float) acc3 * inv_sum;
    reinterpret_cast<float4 *>(fb)[(size_t) py * rp.xres + px] = out;
  }
}

// ---------------------------------------------------------- k_move_tiles
// Multi-GPU frame (fjgpu_render_frame_multi): a device's finished tiles are packed into one
// contiguous slab (tile k at k * tile_px pixels), the slab crosses xGMI as ONE peer copy, and
// the first device scatters it into its framebuffer.  kUnpack = false: framebuffer -> slab.
template <bool kUnpack>
__global__ void __launch_bounds__(BLOCK) k_move_tiles(float4 *fb, int xres, const int4 *rects, int tile_px, float4 *slab)
{
  const int4 r = rects[blockIdx.y];
  const int w = r.z - r.x, h = r.w - r.y;
  for (int p = blockIdx.x * BLOCK + threadIdx.x; p < w * h; p += gridDim.x * BLOCK) {
    const size_t at = (size_t) (r.y + p / w) * xres + (r.x + p % w);
    const size_t to = (size_t) blockIdx.y * tile_px + p;
    if (kUnpack) fb[at] = slab[to]; else slab[to] = fb[at];
  }
}

// ------------------------------------------------------- k_quantize_nodes
// DNode -> DNodeQ (fjgpu_types.h): child boxes outward onto the 65536^3 grid origin + q * cell.
// floor / ceil in f64, then checked against the decoded plane and stepped outward if a rounding
// went the wrong way: the decoded box contains the f32 box wherever that box lies inside the
// grid (coordinates clamp to [0, 65535]; the grid spans the primitive set's PADDED bounds, and a
// node box is the f64 primitive bounds rounded outward by <= 2 ulp of f32, which stays inside
// the 1e-4 padding for |coordinate| < 512 -- beyond that a clamped plane can sit inside the f32
// box by those ulps, but never inside the f64 bounds of the geometry: still conservative).
__global__ void __launch_bounds__(BLOCK) k_quantize_nodes(const DNode *nodes, uint32_t n, double ox, double oy, double oz,
    double cx, double cy, double cz, DNodeQ *out)
{
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  if (i >= n) return;
  const DNode nd = nodes[i];
  DNodeQ q;
  const double o[3] = {ox, oy, oz}, c[3] = {cx, cy, cz};
  for (int k = 0; k < 4; k++) {
    q.child[k] = nd.child[k];
    for (int a = 0; a < 3; a++) {
      const double lo = (double) nd.box[k][2 * a], hi = (double) nd.box[k][2 * a + 1];
      double ql = floor((lo - o[a]) / c[a]), qh = ceil((hi - o[a]) / c[a]);
      if (!(ql >= 0)) ql = 0;                    // (also the empty slots' +-FLT_MAX / NaN)
      if (!(qh <= 65535)) qh = 65535;
      if (!(qh >= 0)) qh = 0;
      if (!(ql <= 65535)) ql = 65535;
      if (o[a] + ql * c[a] > lo && ql > 0) ql -= 1;
      if (o[a] + qh * c[a] < hi && qh < 65535) qh += 1;
      q.q[k][2 * a] = (uint16_t) ql;
      q.q[k][2 * a + 1] = (uint16_t) qh;
    }
  }
  out[i] = q;
}

// ----------------------------------------------------------- host launchers
// persistent launches: at most PERSIST_BLOCKS_PER_CU resident blocks per CU
#define PERSIST_BLOCKS_PER_CU 4
#define PERSIST_BLOCKS_PER_CU_MAX 8      // no kernel launches more (sizes per-thread scratch)
static unsigned persistent_grid(unsigned long long blocks_needed, int blocks_per_cu = PERSIST_BLOCKS_PER_CU)
{
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  const unsigned long long cap = (unsigned long long) cus * blocks_per_cu;
  return (unsigned) (blocks_needed < cap ? (blocks_needed ? blocks_needed : 1) : cap);
}

size_t persistent_threads() { return (size_t) persistent_grid(~0ull, PERSIST_BLOCKS_PER_CU_MAX) * BLOCK; }

// (experiment knobs: fewer resident blocks of the light loop / the closest-hit walk leave room for the other stream's kernels, option overlap_shadow)
static int cull_blocks_per_cu(int asked) { static int b = -1; if (b < 0) { const char *e = getenv("FJGPU_CULL_BLOCKS"); b = e ? atoi(e) : 0; } return b > 0 ? b : (asked > 0 ? asked : PERSIST_BLOCKS_PER_CU); }
static int closest_blocks_per_cu() { static int b = 0; if (!b) { const char *e = getenv("FJGPU_CLOSEST_BLOCKS"); b = e ? atoi(e) : PERSIST_BLOCKS_PER_CU; if (b < 1) b = 1; } return b; }

// blocks per CU of the lean any-hit walk: what its registers and LDS stack allow
static int anyhit_blocks_per_cu(bool multi)
{
  int b = multi ? FJ_ANYHIT_MINB_MULTI : FJ_ANYHIT_MINB;
  if (const char *e = getenv("FJGPU_ANYHIT_BLOCKS")) b = atoi(e);
  if (b < 1) b = 1;
  if (b > PERSIST_BLOCKS_PER_CU_MAX) b = PERSIST_BLOCKS_PER_CU_MAX;
  return b;
}

static int canyhit_blocks_per_cu()
{
  int b = FJ_CANYHIT_MINB;
  if (const char *e = getenv("FJGPU_CANYHIT_BLOCKS")) b = atoi(e);
  return b < 1 ? 1 : (b > PERSIST_BLOCKS_PER_CU_MAX ? PERSIST_BLOCKS_PER_CU_MAX : b);
}

static long g_anyhit_filter_off = 0;     // diagnostics ("anyhit_filter_off"): every triangle test of the lean any-hit walk is left to its exact phase
void set_anyhit_filter_off(long v) { g_anyhit_filter_off = v != 0; }

static TravTune trav_tune()
{
  static TravTune t = {};
  if (t.grab == 0) {
    auto env = [](const char *name, uint32_t dflt) { const char *v = getenv(name); return v ? (uint32_t) atoi(v) : dflt; };
    t.refill = env("FJGPU_TRAV_REFILL", 40);       // (24 until the phase-scheduled any-hit walk; C3: closest 33.4 -> 30.5 ms, any-hit 83.7 -> 76.2)
    t.refill_curves = env("FJGPU_TRAV_REFILL_CURVES", 16);   // the ribbon-test instantiations (C5: 16 / 24 / 32 / 40 -> 2.59 / 2.63 / 2.81 / ~3 s)
    t.steps_curves = env("FJGPU_TRAV_STEPS_CURVES", 6);      // ... and their inner steps per iteration (2 / 3 / 4 -> 2.77 / 2.63 / 2.56 s; at 3 waves 4 / 6 -> 2.19 / 2.15 s)
    t.steps = env("FJGPU_TRAV_STEPS", 3);
    t.grab = env("FJGPU_TRAV_GRAB", 256);
    // lean any-hit walk: up to `anyhit_steps` inner steps per iteration, the 2nd and later ones only
    // while at least `min_inner` lanes are at inner nodes (C3 -4.5 ms, C6 -30 ms; the general walk
    // keeps the fixed 3: incoherent rays (C4) and curve leaves (C5) lost 2-7 % with it)
    // (round 4, six waves and the cheaper step: 4 / 24 -> 8 / 16: C3 walk 58.1 -> 56.8 ms; 6 / 16 56.9, 6 / 12 57.3, 6 / 8 57.9, 4 / 32 59.2)
    t.anyhit_steps = env("FJGPU_TRAV_ANYHIT_STEPS", 8);
    t.min_inner = env("FJGPU_TRAV_MININNER", 16);
    // the any-hit walk's vote between an inner step and a leaf step: inner while n_inner * leaf_bias8 >= n_leaf * 8 (8: whichever more lanes wait for)
    // (round 5: lanes held at a leaf, not idle lanes, are what an inner step of the any-hit walk runs without -- 35.4 of 64 at inner nodes, 15.8 held,
    // 12.8 idle on C3 --, so the leaf step runs a little before the held lanes are the majority; C3 walk at 10 / 8 / 6 / 5 / 4 / 3: 55.7 / 54.7 / 54.1 / 54.2 / 54.1 / 54.5 ms)
    t.leaf_bias8 = env("FJGPU_TRAV_LEAF_BIAS", 5);
    // ... and the lanes frozen behind an undecided triangle (f32 filter) that make the wave run its exact phase
    // (C3: 4 / 8 / 16 waiting lanes -> 53.7 / 53.9 / 53.9 ms at six waves, 51.3 / 51.3 at seven; a lane that can do nothing else runs it at once)
    t.exact_min = env("FJGPU_TRAV_EXACT_MIN", 8);
    t.leaf_bias8_flat = env("FJGPU_TRAV_LEAF_BIAS_FLAT", 8);
    t.leaf_bias8_phased = env("FJGPU_TRAV_LEAF_BIAS_PHASED", 8);
    t.leaf_bias8_canyhit = env("FJGPU_TRAV_LEAF_BIAS_CANYHIT", 8);
    // the phase-scheduled closest-hit walk (incoherent rays): C4 closest-hit side 758 / 748 ms at 3 / 5 steps, 766 / 758 / 739 at
    // min_inner 32 / 24 / 16; with 5 steps 727 / 716 at 16 / 12
    t.steps_phased = env("FJGPU_TRAV_STEPS_PHASED", 5);
    t.min_inner_phased = env("FJGPU_TRAV_MININNER_PHASED", 12);
    // the any-hit walk of curve scenes (fjgpu_dev_anyhit_curves.h)
    t.refill_canyhit = env("FJGPU_TRAV_REFILL_CANYHIT", 20);     // (C5 walk, refill / leaf_wait: 32 / 40 1252 ms, 24 / 48 1147, 16 / 48 1140, 20 / 56 1130, 12 / 52 1134)
    t.steps_canyhit = env("FJGPU_TRAV_STEPS_CANYHIT", 8);
    t.min_inner_canyhit = env("FJGPU_TRAV_MININNER_CANYHIT", 16);
    // the walk of flat groups (fjgpu_dev_flat.h); C4 frame, refill / steps / min: 40 / 5 / 12 672 ms, 32 / 8 / 12 and no ray sort 638
    t.refill_flat = env("FJGPU_TRAV_REFILL_FLAT", 32);
    t.steps_flat = env("FJGPU_TRAV_STEPS_FLAT", 12);
    t.min_inner_flat = env("FJGPU_TRAV_MININNER_FLAT", 12);
    t.leaf_wait_canyhit = env("FJGPU_TRAV_LEAFWAIT_CANYHIT", 56);
    t.leaf_wait = env("FJGPU_TRAV_LEAFWAIT", 40);   // curve scenes: lanes awaiting the second stage of the ribbon test before it runs
    if (t.refill < 1) t.refill = 1;
    if (t.refill > 64) t.refill = 64;
    if (t.steps < 1) t.steps = 1;
    if (t.grab < 64) t.grab = 64;
  }
  t.filter_off = (uint32_t) g_anyhit_filter_off;
  return t;
}

#ifdef FJ_WAVE_TIMELINE
// debug builds: the per-wave clock table of the walk just launched (WaveTimeline, fjgpu_dev_traverse.h) appended to $FJGPU_TIMELINE
static void timeline_zero(hipStream_t st)
{
  void *p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_timeline)) == hipSuccess) (void) hipMemsetAsync(p, 0, sizeof(unsigned long long) * FJ_TL_WAVES * 6, st);
}
static void timeline_dump(hipStream_t st, const char *kernel, unsigned waves)
{
  const char *path = getenv("FJGPU_TIMELINE");
  if (!path) return;
  static unsigned long long h[FJ_TL_WAVES * 6];
  void *p = nullptr;
  if (hipStreamSynchronize(st) != hipSuccess || hipGetSymbolAddress(&p, HIP_SYMBOL(g_timeline)) != hipSuccess) return;
  if (hipMemcpy(h, p, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return;
  FILE *f = fopen(path, "a");
  if (!f) return;
  if (waves > FJ_TL_WAVES) waves = FJ_TL_WAVES;
  fprintf(f, "# %s waves %u\n", kernel, waves);
  {
    unsigned long long rh[64];
    void *q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(g_rayhist)) == hipSuccess && hipMemcpy(rh, q, sizeof(rh), hipMemcpyDeviceToHost) == hipSuccess) {
      fprintf(f, "#hist");
      for (int b = 0; b < 20; b++) fprintf(f, " %llu:%llu", rh[b], rh[32 + b]);
      fprintf(f, "\n");
      (void) hipMemset(q, 0, sizeof(rh));
    }
  }
  for (unsigned w = 0; w < waves; w++) if (h[6 * w + 2]) fprintf(f, "%u %llu %llu %llu %llu %llu %llu %llu %llu\n", w, h[6 * w], h[6 * w + 1], h[6 * w + 2], h[6 * w + 3] >> 32, h[6 * w + 3] & 0xffffffffull,
      h[6 * w + 4] >> 32, h[6 * w + 4] & 0xffffffffull, h[6 * w + 5]);
  fclose(f);
}
#define TL_ZERO(st) timeline_zero(st)
#define TL_DUMP(st, name, grid) timeline_dump(st, name, (grid) * (BLOCK / 64))
#else
#define TL_ZERO(st) do { } while (0)
#define TL_DUMP(st, name, grid) do { } while (0)
#endif

#define LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int) e_; } while (0)

int launch_gen_camera(hipStream_t st, const DScene &S, const GenParams &gp, const TileDesc *d_tiles, int n_tiles,
    uint32_t max_tile_samples, const double *jit, const double *tim, double *s_uv, DRay *rays, DPath *paths, uint32_t *s_tk)
{
  dim3 grid((max_tile_samples + BLOCK - 1) / BLOCK, n_tiles);
  if (S.cam_xform) hipLaunchKernelGGL(k_gen_camera<true>, grid, dim3(BLOCK), 0, st, S, gp, d_tiles, jit, tim, s_uv, rays, paths, s_tk);
  else hipLaunchKernelGGL(k_gen_camera<false>, grid, dim3(BLOCK), 0, st, S, gp, d_tiles, jit, tim, s_uv, rays, paths, s_tk);
  LAUNCH_CHECK();
  return 0;
}

int launch_trace_closest(hipStream_t st, const DScene &S, const DRay *rays, const DPath *paths, DHit *hits,
    uint32_t n, DCounters *cnt, int count_events)
{
  if (n == 0) return 0;
  (void) hipMemsetAsync(&cnt->trace_xcd_head[0][0], 0, sizeof(cnt->trace_xcd_head), st);
  // scenes without curve sets run the lean instantiation (the ribbon test costs registers)
  // (the event counters cost registers and issue slots: counting is its own instantiation)
  const dim3 grid(persistent_grid((n + BLOCK - 1) / BLOCK, closest_blocks_per_cu()));
  TL_ZERO(st);
#define FJ_LAUNCH_CLOSEST(CURVES, COUNT, MOTION) hipLaunchKernelGGL((k_trace_closest<CURVES, COUNT, MOTION>), grid, dim3(BLOCK), 0, st, S, rays, paths, hits, n, cnt, trav_tune())
  if (S.has_motion) {      // time-sampled instance transforms: one general instantiation
    if (count_events) FJ_LAUNCH_CLOSEST(true, true, true); else FJ_LAUNCH_CLOSEST(true, false, true);
  } else if (S.has_curves) {
    if (InstLdsCurves::fits(S)) {      // the instance level in the blocks' LDS
      if (count_events) hipLaunchKernelGGL((k_trace_closest<true, true, false, true>), grid, dim3(BLOCK), 0, st, S, rays, paths, hits, n, cnt, trav_tune());
      else hipLaunchKernelGGL((k_trace_closest<true, false, false, true>), grid, dim3(BLOCK), 0, st, S, rays, paths, hits, n, cnt, trav_tune());
    }
    else if (count_events) FJ_LAUNCH_CLOSEST(true, true, false); else FJ_LAUNCH_CLOSEST(true, false, false);
  }
  else if (S.flats) {      // scenes whose groups are flat: one world-space tree per group (fjgpu_dev_flat.h; built only where the scene fits the kernel)
    if (S.n_groups == 1) {      // (one group: its arrays in scalar registers)
      if (count_events) hipLaunchKernelGGL((k_trace_closest_flat<true, true>), grid, dim3(BLOCK), 0, st, S, rays, paths, hits, n, cnt, trav_tune());
      else hipLaunchKernelGGL((k_trace_closest_flat<false, true>), grid, dim3(BLOCK), 0, st, S, rays, paths, hits, n, cnt, trav_tune());
    }
    else if (count_events) hipLaunchKernelGGL((k_trace_closest_flat<true, false>), grid, dim3(BLOCK), 0, st, S, rays, paths, hits, n, cnt, trav_tune());
    else hipLaunchKernelGGL((k_trace_closest_flat<false, false>), grid, dim3(BLOCK), 0, st, S, rays, paths, hits, n, cnt, trav_tune());
  }
  else if (S.incoherent_rays) {   // glass / pathtracing scenes: the phase-scheduled walk (see k_trace_closest_phased)
    if (InstLds::fits(S)) {     // the instance level in the blocks' LDS
      if (count_events) hipLaunchKernelGGL((k_trace_closest_phased<true, true>), grid, dim3(BLOCK), 0, st, S, rays, paths, hits, n, cnt, trav_tune());
      else hipLaunchKernelGGL((k_trace_closest_phased<false, true>), grid, dim3(BLOCK), 0, st, S, rays, paths, hits, n, cnt, trav_tune());
    } else {
      if (count_events) hipLaunchKernelGGL((k_trace_closest_phased<true, false>), grid, dim3(BLOCK), 0, st, S, rays, paths, hits, n, cnt, trav_tune());
      else hipLaunchKernelGGL((k_trace_closest_phased<false, false>), grid, dim3(BLOCK), 0, st, S, rays, paths, hits, n, cnt, trav_tune());
    }
  }
  else if (InstLdsBig::fits(S)) {
    if (count_events) hipLaunchKernelGGL((k_trace_closest<false, true, false, true>), grid, dim3(BLOCK), 0, st, S, rays, paths, hits, n, cnt, trav_tune());
    else hipLaunchKernelGGL((k_trace_closest<false, false, false, true>), grid, dim3(BLOCK), 0, st, S, rays, paths, hits, n, cnt, trav_tune());
  }
  else { if (count_events) FJ_LAUNCH_CLOSEST(false, true, false); else FJ_LAUNCH_CLOSEST(false, false, false); }
#undef FJ_LAUNCH_CLOSEST
  LAUNCH_CHECK();
  TL_DUMP(st, "closest", grid.x);
  return 0;
}

int launch_shade(hipStream_t st, const DScene &S, const ShadeParams &sp, const DRay *rays, const DPath *paths,
    const DHit *hits, uint32_t n, float *s_accum, DRay *next_rays, DPath *next_paths, DLightRec *lrecs, DCounters *cnt)
{
  if (n == 0) return 0;
#define FJ_LAUNCH_SHADE(MOTION) hipLaunchKernelGGL((k_shade<MOTION>), dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, S, sp, rays, paths, hits, n, \
    s_accum, next_rays, next_paths, lrecs, cnt)
  if (S.has_motion) FJ_LAUNCH_SHADE(true); else FJ_LAUNCH_SHADE(false);
#undef FJ_LAUNCH_SHADE
  LAUNCH_CHECK();
  return 0;
}

// Shadow work in two steps.  The light loop APPENDS the surviving shadow rays of records
// [b, e) to the queue (cnt->shadow_count keeps growing); the traversal consumes the whole
// queue in ONE launch per batch of tiles (or when the queue would overflow) -- a persistent
// kernel ends with a tail of a few slow rays (~1 ms on C3), so it pays to launch it once for
// the rays of every recursion level instead of once per level.
uint32_t shadow_queue_padding()          // every resident wave may leave one partially filled chunk
{
  return persistent_grid(1ull << 30, PERSIST_BLOCKS_PER_CU_MAX) * (BLOCK / 64) * SQ_CHUNK;
}

void shadow_queue_reset(hipStream_t st, DCounters *cnt)
{
  (void) hipMemsetAsync(&cnt->shadow_count, 0, 2 * sizeof(uint32_t), st);   // shadow_count + shadow_head
  (void) hipMemsetAsync(&cnt->join_count, 0, sizeof(uint32_t), st);
  (void) hipMemsetAsync(&cnt->shadow_xcd_head[0][0], 0, sizeof(cnt->shadow_xcd_head), st);
}

int launch_shadow_cull(hipStream_t st, const DScene &S, const ShadowParams &sp, const DLightRec *lrecs, uint32_t b, uint32_t e,
    float *s_accum, DShadowRay *squeue, DCounters *cnt, int count_events, int blocks_per_cu)
{
  if (e <= b) return 0;
  const unsigned long long threads = (unsigned long long) (e - b);       // one light record per lane
  (void) hipMemsetAsync(&cnt->cull_head, 0, sizeof(uint32_t), st);
#define FJ_LAUNCH_CULL(HAIR, AREA, SPLIT) hipLaunchKernelGGL((k_shadow_cull<HAIR, AREA, SPLIT>), dim3(persistent_grid((threads + BLOCK - 1) / BLOCK, cull_blocks_per_cu(blocks_per_cu))), dim3(BLOCK), 0, st, S, sp, lrecs, b, e, s_accum, squeue, cnt, count_events)
  // (SPLIT: rays into groups of several instances are queued once per candidate instance, DScene.shadow_join)
  const bool split = sp.join_capacity != 0 && S.shadow_join != nullptr;
  if (S.has_area) { if (split) FJ_LAUNCH_CULL(true, true, true); else FJ_LAUNCH_CULL(true, true, false); }          // general instantiation
  else if (S.has_hair) { if (split) FJ_LAUNCH_CULL(true, false, true); else FJ_LAUNCH_CULL(true, false, false); }
  else if (S.inst_lds && S.multi_shadow_groups && S.n_group_nodes <= FJ_CULL_LDS_NODES) {      // instance nodes in the blocks' LDS
    if (split) hipLaunchKernelGGL((k_shadow_cull<false, false, true, true>), dim3(persistent_grid((threads + BLOCK - 1) / BLOCK, cull_blocks_per_cu(blocks_per_cu))), dim3(BLOCK), 0, st, S, sp, lrecs, b, e, s_accum, squeue, cnt, count_events);
    else hipLaunchKernelGGL((k_shadow_cull<false, false, false, true>), dim3(persistent_grid((threads + BLOCK - 1) / BLOCK, cull_blocks_per_cu(blocks_per_cu))), dim3(BLOCK), 0, st, S, sp, lrecs, b, e, s_accum, squeue, cnt, count_events);
  }
  else { if (split) FJ_LAUNCH_CULL(false, false, true); else FJ_LAUNCH_CULL(false, false, false); }
#undef FJ_LAUNCH_CULL
  LAUNCH_CHECK();
  return 0;
}

int launch_shadow_trace(hipStream_t st, const DScene &S, const DShadowRay *squeue, float *s_accum, DCounters *cnt, int count_events)
{
  TL_ZERO(st);
  if (S.all_opaque && !S.has_curves && !S.has_motion && S.blas_base) {
#define FJ_LAUNCH_ANYHIT(COUNT, MULTI) hipLaunchKernelGGL((k_shadow_anyhit<COUNT, MULTI>), dim3(persistent_grid(1ull << 30, anyhit_blocks_per_cu(MULTI))), dim3(BLOCK), 0, st, S, squeue, s_accum, cnt, trav_tune())
    // (with DScene.shadow_join every queue entry names its instance: the instantiation without the instance-level walk)
    if (S.multi_shadow_groups && !S.shadow_join) { if (count_events) FJ_LAUNCH_ANYHIT(true, true); else FJ_LAUNCH_ANYHIT(false, true); }
    else { if (count_events) FJ_LAUNCH_ANYHIT(true, false); else FJ_LAUNCH_ANYHIT(false, false); }
#undef FJ_LAUNCH_ANYHIT
  } else {
#define FJ_LAUNCH_SHADOW(CURVES, COUNT, MOTION) hipLaunchKernelGGL((k_shadow_trace<CURVES, COUNT, MOTION>), dim3(persistent_grid(1ull << 30)), dim3(BLOCK), 0, st, S, squeue, s_accum, cnt, trav_tune())
    if (S.has_motion) { if (count_events) FJ_LAUNCH_SHADOW(true, true, true); else FJ_LAUNCH_SHADOW(true, false, true); }
    else if (S.has_curves) {
      if (InstLdsCurves::fits(S) && S.all_opaque && S.curve_anyhit && FJ_CURVE_QNODES && FJ_CLOSEST_QNODES) {      // the phase-scheduled any-hit walk with a ribbon phase
        const dim3 cgrid(persistent_grid(1ull << 30, canyhit_blocks_per_cu()));
        if (count_events) hipLaunchKernelGGL((k_shadow_anyhit_curves<true>), cgrid, dim3(BLOCK), 0, st, S, squeue, s_accum, cnt, trav_tune());
        else hipLaunchKernelGGL((k_shadow_anyhit_curves<false>), cgrid, dim3(BLOCK), 0, st, S, squeue, s_accum, cnt, trav_tune());
      }
      else if (InstLdsCurves::fits(S) && S.all_opaque) {      // ... and no hit records: every ray any-hit, an occluded one adds nothing
        if (count_events) hipLaunchKernelGGL((k_shadow_trace<true, true, false, true, true>), dim3(persistent_grid(1ull << 30)), dim3(BLOCK), 0, st, S, squeue, s_accum, cnt, trav_tune());
        else hipLaunchKernelGGL((k_shadow_trace<true, false, false, true, true>), dim3(persistent_grid(1ull << 30)), dim3(BLOCK), 0, st, S, squeue, s_accum, cnt, trav_tune());
      }
      else if (InstLdsCurves::fits(S)) {      // the instance level in the blocks' LDS
        if (count_events) hipLaunchKernelGGL((k_shadow_trace<true, true, false, true>), dim3(persistent_grid(1ull << 30)), dim3(BLOCK), 0, st, S, squeue, s_accum, cnt, trav_tune());
        else hipLaunchKernelGGL((k_shadow_trace<true, false, false, true>), dim3(persistent_grid(1ull << 30)), dim3(BLOCK), 0, st, S, squeue, s_accum, cnt, trav_tune());
      }
      else if (count_events) FJ_LAUNCH_SHADOW(true, true, false); else FJ_LAUNCH_SHADOW(true, false, false);
    }
    else if (InstLdsBig::fits(S)) {      // the instance level in the blocks' LDS
      if (count_events) hipLaunchKernelGGL((k_shadow_trace<false, true, false, true>), dim3(persistent_grid(1ull << 30)), dim3(BLOCK), 0, st, S, squeue, s_accum, cnt, trav_tune());
      else hipLaunchKernelGGL((k_shadow_trace<false, false, false, true>), dim3(persistent_grid(1ull << 30)), dim3(BLOCK), 0, st, S, squeue, s_accum, cnt, trav_tune());
    }
    else { if (count_events) FJ_LAUNCH_SHADOW(false, true, false); else FJ_LAUNCH_SHADOW(false, false, false); }
#undef FJ_LAUNCH_SHADOW
  }
  LAUNCH_CHECK();
  TL_DUMP(st, "shadow", persistent_grid(1ull << 30, PERSIST_BLOCKS_PER_CU_MAX));
  return 0;
}

int launch_quantize_nodes(hipStream_t st, const DNode *nodes, uint32_t n, const double *origin, const double *cell, DNodeQ *out)
{
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_quantize_nodes, dim3((n + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, nodes, n, origin[0], origin[1], origin[2],
      cell[0], cell[1], cell[2], out);
  LAUNCH_CHECK();
  return 0;
}

int launch_move_tiles(hipStream_t st, bool unpack, float *fb, int xres, const int32_t *d_rects, int n_tiles, int tile_px, float *slab)
{
  // (the tile index is grid.y, at most 65535 per launch: more tiles go in several launches)
  for (int t0 = 0; t0 < n_tiles; t0 += 65535) {
    const int nt = n_tiles - t0 < 65535 ? n_tiles - t0 : 65535;
    const dim3 grid((tile_px + BLOCK - 1) / BLOCK, nt);
    float4 *sl = (float4 *) slab + (size_t) t0 * tile_px;
    if (unpack) hipLaunchKernelGGL(k_move_tiles<true>, grid, dim3(BLOCK), 0, st, (float4 *) fb, xres, (const int4 *) d_rects + t0, tile_px, sl);
    else hipLaunchKernelGGL(k_move_tiles<false>, grid, dim3(BLOCK), 0, st, (float4 *) fb, xres, (const int4 *) d_rects + t0, tile_px, sl);
    LAUNCH_CHECK();
  }
  return 0;
}

void debug_phase_stats()
{
#ifdef FJ_TRI_FILTER_VALIDATE
  {
    unsigned long long c[8];
    if (hipMemcpyFromSymbol(c, HIP_SYMBOL(g_trifilter), sizeof(c)) == hipSuccess && (c[0] | c[1] | c[2])) {
      fprintf(stderr, "fjgpu tri filter: miss %llu hit %llu maybe %llu | exact hits %llu | CONTRADICTIONS %llu\n", c[0], c[1], c[2], c[4], c[3]);
      unsigned long long z[8] = {0};
      (void) hipMemcpyToSymbol(HIP_SYMBOL(g_trifilter), z, sizeof(z));
    }
  }
#endif
#ifdef FJ_PHASE_STATS
  {
    unsigned long long c[8];
    if (hipMemcpyFromSymbol(c, HIP_SYMBOL(g_curve_stat), sizeof(c)) == hipSuccess && c[2]) {
      fprintf(stderr, "fjgpu phase curves: second-stage execs %llu lanes %llu tests-hit %llu leaf-walks %llu nodes %llu\n", c[2], c[3], c[4], c[0], c[1]);
      unsigned long long z[8] = {0};
      (void) hipMemcpyToSymbol(HIP_SYMBOL(g_curve_stat), z, sizeof(z));
    }
  }
  {
    unsigned long long c[16];
    if (hipMemcpyFromSymbol(c, HIP_SYMBOL(g_cphase), sizeof(c)) == hipSuccess && c[12]) {
      static const char *pn[4] = {"turnover", "inner", "leaf", "ribbon"};
      const double all = (double) (c[0] + c[1] + c[2] + c[3]);
      fprintf(stderr, "fjgpu phase curve-anyhit: %llu iterations", c[12]);
      for (int k = 0; k < 4; k++) fprintf(stderr, " | %s %.1f %% of ticks, %llu execs, %.1f lanes, %.0f ticks each", pn[k], 100. * c[k] / all, c[4 + k], (double) c[8 + k] / (c[4 + k] ? c[4 + k] : 1), (double) c[k] / (c[4 + k] ? c[4 + k] : 1));
      fprintf(stderr, "\n");
      unsigned long long z[16] = {0};
      (void) hipMemcpyToSymbol(HIP_SYMBOL(g_cphase), z, sizeof(z));
    }
  }
  unsigned long long h[16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof(h)) != hipSuccess) return;
  // (the general walk, traverse_persistent, reports clock ticks per phase in slots 1 / 3 / 5 / 7: refill + instance entry, inner steps,
  //  leaf phase, second stage of the ribbon test; the lean any-hit walk reports executions / lanes there)
  static const char *names[16] = {"iters", "entry_execs|entry_ticks", "entry_lanes", "inner_execs|inner_ticks", "inner_lanes", "leaf_execs|leaf_ticks", "leaf_lanes",
      "tri_execs|stage2_ticks", "tri_lanes", "", "hits", "refills", "tail_iters_max", "tail_iters_sum", "walk_iters_sum", "walk_waves"};
  for (int i = 0; i < 16; i++) if (names[i][0]) fprintf(stderr, "fjgpu phase %-24s %llu\n", names[i], h[i]);
  // (the lean any-hit walk, tallies of its own: where the lanes are that take no part in a step -- idle = ray finished, waiting for the turnover)
  {
    unsigned long long a[16];
    if (hipMemcpyFromSymbol(a, HIP_SYMBOL(g_ahphase), sizeof(a)) == hipSuccess && a[3] && a[5]) {
      fprintf(stderr, "fjgpu phase anyhit-lanes: %llu iterations | inner steps %llu with %.1f lanes at inner nodes, %.1f idle, %.1f held at a leaf | leaf steps %llu with %.1f lanes, %.1f idle, "
          "%.1f at inner nodes | turnovers %llu with %.1f lanes | %llu rays occluded\n", a[0], a[3], (double) a[4] / a[3], (double) a[7] / a[3], (double) a[8] / a[3], a[5], (double) a[6] / a[5],
          (double) a[9] / a[5], (double) a[11] / a[5], a[1], (double) a[2] / (a[1] ? a[1] : 1), a[10]);
      if (a[12]) fprintf(stderr, "fjgpu phase anyhit-exact: %llu exact phases with %.1f lanes, %llu of them hits\n", a[12], (double) a[13] / a[12], a[14]);
      unsigned long long z[16] = {0};
      (void) hipMemcpyToSymbol(HIP_SYMBOL(g_ahphase), z, sizeof(z));
    }
  }
  // (the phase-scheduled closest-hit walk, traverse_phased: ticks / lanes of turnover, inner, leaf in 1..6, their executions in 7..9,
  //  trips of the instance loop summed over lanes in 10 and as the wave's maximum per turnover in 11)
  if (h[7] && h[11]) fprintf(stderr, "fjgpu phase phased-walk: turnover %llu execs %.1f lanes %.0f ticks | inner %llu execs %.1f lanes %.0f ticks | leaf %llu execs %.1f lanes %.0f ticks | "
      "instance loop %.2f trips per turnover lane, %.2f wave trips per turnover\n", h[7], (double) h[2] / h[7], (double) h[1] / h[7], h[8], (double) h[4] / (h[8] ? h[8] : 1), (double) h[3] / (h[8] ? h[8] : 1),
      h[9], (double) h[6] / (h[9] ? h[9] : 1), (double) h[5] / (h[9] ? h[9] : 1), (double) h[10] / (h[2] ? h[2] : 1), (double) h[11] / h[7]);
  unsigned long long z[16] = {0};
  (void) hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z));
#endif
}

int launch_resolve(hipStream_t st, const ResolveParams &rp, const TileDesc *d_tiles, int n_tiles, int max_tile_pixels,
    const double *s_uv, const float *s_accum, const double *s_data64, float *fb)
{
  const dim3 grid((max_tile_pixels + BLOCK / 64 - 1) / (BLOCK / 64), n_tiles);
  if (s_data64)
    hipLaunchKernelGGL(k_resolve<double4>, grid, dim3(BLOCK), 0, st, rp, d_tiles, s_uv, reinterpret_cast<const double4 *>(s_data64), fb);
  else
    hipLaunchKernelGGL(k_resolve<float4>, grid, dim3(BLOCK), 0, st, rp, d_tiles, s_uv, reinterpret_cast<const float4 *>(s_accum), fb);
  LAUNCH_CHECK();
  return 0;
}

int launch_adaptive_uv(hipStream_t st, const AdaptiveParams &ap, const TileDesc *d_tiles, int n_tiles, uint32_t max_tile_samples,
    const double *jit, double *s_uv)
{
  hipLaunchKernelGGL(k_adaptive_uv, dim3((max_tile_samples + BLOCK - 1) / BLOCK, n_tiles), dim3(BLOCK), 0, st, ap, d_tiles, jit, s_uv);
  LAUNCH_CHECK();
  return 0;
}

int launch_adaptive_points(hipStream_t st, const DScene &S, const AdaptiveParams &ap, const TileDesc *d_tiles, int n_tiles,
    uint32_t max_tile_points, const uint8_t *cells, const double *s_uv, const float *s_accum, uint8_t *pstate, double *seen,
    DRay *rays, DPath *paths, DCounters *cnt)
{
  const dim3 grid((max_tile_points + BLOCK - 1) / BLOCK, n_tiles);
  if (S.cam_xform)
    hipLaunchKernelGGL(k_adaptive_points<true>, grid, dim3(BLOCK), 0, st, S, ap, d_tiles, cells, s_uv, s_accum, pstate, seen, rays, paths, cnt);
  else
    hipLaunchKernelGGL(k_adaptive_points<false>, grid, dim3(BLOCK), 0, st, S, ap, d_tiles, cells, s_uv, s_accum, pstate, seen, rays, paths, cnt);
  LAUNCH_CHECK();
  return 0;
}

int launch_adaptive_decide(hipStream_t st, const AdaptiveParams &ap, const TileDesc *d_tiles, int n_tiles, uint32_t max_tile_cells,
    uint8_t *cells, const float *s_accum, const uint8_t *pstate, const double *seen)
{
  hipLaunchKernelGGL(k_adaptive_decide, dim3((max_tile_cells + BLOCK - 1) / BLOCK, n_tiles), dim3(BLOCK), 0, st, ap, d_tiles, cells,
      s_accum, pstate, seen);
  LAUNCH_CHECK();
  return 0;
}

int launch_adaptive_fill(hipStream_t st, const AdaptiveParams &ap, const TileDesc *d_tiles, int n_tiles, uint32_t max_tile_samples,
    const uint8_t *cells, const float *s_accum, const uint8_t *pstate, const double *seen, double *final_data)
{
  hipLaunchKernelGGL(k_adaptive_fill, dim3((max_tile_samples + BLOCK - 1) / BLOCK, n_tiles), dim3(BLOCK), 0, st, ap, d_tiles, cells,
      s_accum, pstate, seen, final_data);
  LAUNCH_CHECK();
  return 0;
}
