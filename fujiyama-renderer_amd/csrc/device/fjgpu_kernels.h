// fjgpu_kernels.h -- launch parameters and host launchers of fjgpu_kernels.hip
#ifndef FJGPU_KERNELS_H
#define FJGPU_KERNELS_H

#include <hip/hip_runtime.h>
#include "fjgpu_types.h"

struct TileDesc {              // one tile of the current batch
  int32_t xmin, ymin, xmax, ymax;
  int32_t nx, ny;              // samples incl. filter margin: rate * size + 2 * margin
  uint32_t sample_offset;      // first sample slot of this tile in the batch arrays
  int32_t id;
  uint32_t cell_offset;        // adaptive sampler: first level-0 lattice cell of this tile in the batch
  int32_t pad;
};

struct GenParams {
  int32_t rate_x, rate_y, margin_x, margin_y;
  double udelta, vdelta, jitter;
  int32_t jittered, pad;
};

struct AdaptiveParams {       // AdaptiveGridSampler, src/fj_adaptive_grid_sampler.cc
  int32_t D, div;              // adaptive_max_subdivision, 2^D lattice cells per pixel
  int32_t margin_x, margin_y;  // filter margin in PIXELS: ceil(filterwidth - 1)
  double udelta, vdelta, jitter, threshold;
  int32_t jittered, level;
  uint32_t cells0;             // level-0 lattice cells of the whole batch
  uint32_t ray_capacity;
};

struct ShadeParams {
  int32_t max_diffuse_depth, max_reflect_depth, max_refract_depth;
  int32_t count_all_shadow;    // 1: trace zero-weight light records too (reference ray counts)
  uint32_t ray_capacity, light_capacity;
  // ray-queue sort (fjgpu_raysort.hip): the key of a child ray is computed where the ray is emitted -- origin and direction
  // are in registers there -- instead of by a pass of its own over the queue (C4: 1.6 G rays x 48 B less read per frame)
  uint32_t *next_keys;         // key of the child in slot k of the next queue, or null (no sort in this scene / for this level)
  // filter colour of refraction children (DPath.flags bit 0), 3 floats per queue slot: of the rays being shaded / of the children emitted; null in
  // scenes without a glass or pathtracing shader (no ray ever sets the bit there)
  const float *fc_in;
  float *fc_out;
  double sort_lo[3], sort_scale[3];
  int32_t sort_bits, pad_;
};

struct ShadowParams {
  double cos_half_pi, cos_pi;  // cos(PI/2.), cos(PI) evaluated by the host libm
  uint32_t pre_resolve;        // 1: queue entries of single-instance groups carry ~instance (lean any-hit walk)
  uint32_t queue_capacity;     // entries of the shadow-ray queue
  uint32_t join_capacity;      // slots of DScene.shadow_join (0: shadow rays into groups of several instances stay whole)
  int32_t cast_shadow;
  int32_t compact;             // 1: the queue holds DShadowRayC records (DScene.compact_squeue)
  int32_t pad_;
};

struct ResolveParams {
  int32_t xres, yres, rate_x, rate_y, npx_x, npx_y;
  double fw, fh;
};

int launch_gen_camera(hipStream_t st, const DScene &S, const GenParams &gp, const TileDesc *d_tiles, int n_tiles,
    uint32_t max_tile_samples, const double *jit, const double *tim, double *s_uv, DRay *rays, DPath *paths, uint32_t *s_tk = nullptr);
int launch_trace_closest(hipStream_t st, const DScene &S, const DRay *rays, const DPath *paths, DHit *hits,
    uint32_t n, DCounters *cnt, int count_events);
int launch_shade(hipStream_t st, const DScene &S, const ShadeParams &sp, const DRay *rays, const DPath *paths,
    const DHit *hits, uint32_t n, float *s_accum, DRay *next_rays, DPath *next_paths, DLightRec *lrecs, DCounters *cnt);
uint32_t shadow_queue_padding();
void shadow_queue_reset(hipStream_t st, DCounters *cnt);
int launch_shadow_cull(hipStream_t st, const DScene &S, const ShadowParams &sp, const DLightRec *lrecs, uint32_t b, uint32_t e,
    float *s_accum, DShadowRay *squeue, DCounters *cnt, int count_events, int blocks_per_cu = 0);      // (0: what fits)
int launch_shadow_trace(hipStream_t st, const DScene &S, const DShadowRay *squeue, float *s_accum, DCounters *cnt, int count_events);
// the quantised node array of the lean any-hit walk from the f32 one (same indices)
int launch_quantize_nodes(hipStream_t st, const DNode *nodes, uint32_t n, const double *origin, const double *cell, DNodeQ *out);
// multi-GPU frame: pack a device's tiles (d_rects [n][4] = xmin ymin xmax ymax) into a slab of
// tile_px pixels per tile, or scatter such a slab into the framebuffer (unpack)
int launch_move_tiles(hipStream_t st, bool unpack, float *fb, int xres, const int32_t *d_rects, int n_tiles, int tile_px, float *slab);
void set_anyhit_filter_off(long v);     // diagnostics: fjgpu_global_option("anyhit_filter_off")
void debug_phase_stats();     // FJ_PHASE_STATS builds: print and reset the any-hit walk's phase tallies (stderr)
// threads of the largest persistent grid (sizes per-thread scratch such as the stack overflow area)
size_t persistent_threads();

// sample values: f32 RGBA as accumulated (fixed grid) or f64 RGBA (adaptive grid: interpolated samples)
int launch_resolve(hipStream_t st, const ResolveParams &rp, const TileDesc *d_tiles, int n_tiles, int max_tile_pixels,
    const double *s_uv, const float *s_accum, const double *s_data64, float *fb);

// adaptive grid sampler (fjgpu_dev_adaptive.h)
int launch_adaptive_uv(hipStream_t st, const AdaptiveParams &ap, const TileDesc *d_tiles, int n_tiles, uint32_t max_tile_samples,
    const double *jit, double *s_uv);
int launch_adaptive_points(hipStream_t st, const DScene &S, const AdaptiveParams &ap, const TileDesc *d_tiles, int n_tiles,
    uint32_t max_tile_points, const uint8_t *cells, const double *s_uv, const float *s_accum, uint8_t *pstate, double *seen,
    DRay *rays, DPath *paths, DCounters *cnt);
int launch_adaptive_decide(hipStream_t st, const AdaptiveParams &ap, const TileDesc *d_tiles, int n_tiles, uint32_t max_tile_cells,
    uint8_t *cells, const float *s_accum, const uint8_t *pstate, const double *seen);
int launch_adaptive_fill(hipStream_t st, const AdaptiveParams &ap, const TileDesc *d_tiles, int n_tiles, uint32_t max_tile_samples,
    const uint8_t *cells, const float *s_accum, const uint8_t *pstate, const double *seen, double *final_data);

#endif
