/* fjgpu.h -- C ABI of the MI355X ray-intersection + integrator core
 * (libfjgpu.so, hand-written HIP for gfx950).
 *
 * This is the compute half of the drop-in boundary.  In the reference the
 * whole path lives behind one call,
 *     Renderer::RenderScene()            src/fj_renderer.cc:667-684
 * reached from SiRenderScene()           src/fj_scene_interface.cc:247-278
 * after prepare_render() has computed bounds, created the implicit groups and
 * built the accelerators (same file :1204-1222).  A maintainer of the
 * reference binds these entry points at exactly that spot:
 *
 *   fjgpu_scene_create   replaces build_accelerators()       src/fj_scene_interface.cc:1161-1202
 *                        (GridAccelerator::build             src/fj_grid_accelerator.cc:69-160,
 *                         BVHAccelerator::build              src/fj_bvh_accelerator.cc:79-107)
 *   fjgpu_render_tiles   replaces execute_rendering()'s      src/fj_renderer.cc:747-791
 *                        MtRunParallelLoop(render_tile)      src/fj_renderer.cc:1098-1121
 *                        i.e. FixedGridSampler::generate_samples, Camera::GetRay,
 *                        SlTrace / SlIlluminance / the shader plugins,
 *                        reconstruct_image + apply_pixel_filter, FrameBuffer::SetColor
 *   fjgpu_trace          exposes Accelerator::Intersect      src/fj_accelerator.cc:94-113
 *                        for ray batches (parity tests, SlSurfaceRayIntersect users)
 *
 * Plain pointers and sizes only; no C++ or torch types; no globals besides the
 * last-error string; every function returns 0 or a negative FJGPU_E* code.
 * There is no CPU fallback: without a GPU every compute call fails loudly.
 */
#ifndef FJGPU_H
#define FJGPU_H

#include <stdint.h>
#include "fj_scene_desc.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
  FJGPU_OK = 0,
  FJGPU_ENODEV = -1,        /* no HIP device / HIP runtime error */
  FJGPU_EINVAL = -2,        /* malformed description */
  FJGPU_EUNSUPPORTED = -3,  /* feature outside the device path (named in the error string) */
  FJGPU_ENOMEM = -4
};

typedef struct fjgpu_scene fjgpu_scene;

/* Per-launch device counters, summed over the call (all optional to read) */
typedef struct fjgpu_stats {
  fj_ray_counts rays;          /* same events the reference would count */
  uint64_t nodes_visited;      /* BLAS nodes fetched (128 B each: a 4-wide node) */
  uint64_t prims_tested;       /* triangle tests (36 B each: f32 vertices, or 72 B as f64) / curve tests */
  uint64_t insts_tested;       /* instance boxes tested (48 B each) */
  uint64_t rays_traced;        /* rays entering a trace kernel (closest + shadow) */
  uint64_t shadow_traversed;   /* shadow rays that survived the instance-box cull and walked a BLAS */
  double   trace_ms;           /* HIP-event time of the trace kernels (closest + shadow) */
  double   shade_ms;           /* shading / queue kernels */
  double   gen_ms;             /* sample + camera-ray generation */
  double   resolve_ms;         /* pixel filter */
  double   total_ms;           /* first launch -> last kernel done */
  uint32_t trace_launches;
  uint32_t batches;
  /* the traversal side per kernel (trace_ms is their sum), so that one kernel's HIP-event time
   * can be set against its own algorithmic bytes and against rocprofv3's average for it */
  double   closest_ms;         /* k_trace_closest launches */
  double   light_loop_ms;      /* k_shadow_cull launches (SlIlluminance light loop + instance-box cull) */
  double   shadow_walk_ms;     /* k_shadow_anyhit / k_shadow_trace launches */
  uint64_t shadow_nodes;       /* BLAS nodes fetched by the shadow walk alone (counting instantiation) */
  uint64_t shadow_prims;       /* triangle / curve tests of the shadow walk alone */
  uint64_t shadow_insts;       /* instance boxes tested by the shadow walk alone */
  uint32_t closest_launches, light_loop_launches, shadow_walk_launches;
  uint32_t interrupted;        /* 1: the batch callback asked to stop; the batches after that one were not rendered */
  double   sort_ms;            /* ray-queue sort in front of the closest-hit walk (option "ray_sort"; part of closest_ms) */
  uint64_t rays_sorted;        /* rays that went through it */
} fjgpu_stats;

/* Number of visible HIP devices (0 when there is none). */
int fjgpu_device_count(void);

/* Build the device scene on HIP device `device`: BLAS per mesh / curve set,
 * instance table with precomputed matrices, groups, lights, shaders and
 * textures resident in HBM.  The description is copied. */
int fjgpu_scene_create(const fj_scene_desc *desc, int device, fjgpu_scene **out);
void fjgpu_scene_destroy(fjgpu_scene *scene);

/* Tiling of a frame exactly as Tiler::GenerateTiles (src/fj_tiler.cc:56-113). */
int fjgpu_tile_count(const fj_render_desc *render);
int fjgpu_tile_rect(const fj_render_desc *render, int tile_id, int32_t rect[4]);

/* Render the listed tiles (NULL = all tiles of the render region) into a
 * DEVICE framebuffer `d_framebuffer` of xres*yres*4 float32 (RGBA, row-major,
 * the layout of FrameBuffer, src/fj_framebuffer.cc:130-133); pixels of tiles
 * not listed are left untouched.  Work is enqueued on `hip_stream`
 * (a hipStream_t passed as void*, NULL = default stream) and the call returns
 * after the stream has been synchronised. */
int fjgpu_render_tiles(fjgpu_scene *scene, const fj_render_desc *render,
    const int32_t *tile_ids, int n_tiles,
    float *d_framebuffer, void *hip_stream, fjgpu_stats *stats);

/* Batch callback.  fjgpu_render_tiles cuts its tile list into BATCHES (as many tiles as the wavefront
 * queues hold: a whole 1080p / 64 spp frame is one batch, option "batch_tiles"); `fn` is called on the
 * calling host thread each time a batch is complete on the device -- the tiles' pixels are final in the
 * device framebuffer -- with the ids of its tiles.  This is where the host side fires the reference's
 * per-sample `sample_done` hook, once per batch (CbReportSampleDone in integrate_samples,
 * src/fj_renderer.cc:1061-1096; a host call per sample is infeasible, SURVEY 8b "Threading").  A nonzero
 * return stops the call after this batch, like the reference's worker pool after a CALLBACK_INTERRUPT
 * (render_tile -> LoopStatus::Cancel, src/fj_renderer.cc:1098-1121): the remaining tiles are not rendered,
 * the call returns 0 and stats->interrupted is 1.  fn = NULL removes it.  (If a call has to be repeated
 * because split shadow rays overflowed their queue -- see "split_shadow" -- its batches are reported again.) */
typedef int (*fjgpu_batch_fn)(void *user, int device, const int32_t *tile_ids, int n_tiles);
int fjgpu_set_batch_callback(fjgpu_scene *scene, fjgpu_batch_fn fn, void *user);

/* Convenience: all tiles, result copied to a HOST buffer (xres*yres*4 floats). */
int fjgpu_render_frame(fjgpu_scene *scene, const fj_render_desc *render,
    float *h_framebuffer, fjgpu_stats *stats);

/* The same scene resident on several devices of this process: the host-side build (BLAS,
 * matrices, light samples) runs once, the result is uploaded to every device in `devices`.
 * out[n_devices]; on failure nothing is left allocated. */
int fjgpu_scene_create_multi(const fj_scene_desc *desc, const int *devices, int n_devices, fjgpu_scene **out);

/* One frame on the devices of `scenes` (replicas of one scene, fjgpu_scene_create_multi) -- the
 * reference's worker pool with GPUs for workers (execute_rendering, src/fj_renderer.cc:747-791;
 * MtRunParallelLoop, src/fj_multi_thread.cc:86-132).  Tile k of `tile_ids` (NULL = all tiles of
 * the render region) is rendered by scenes[k % n_scenes] on that device's own host thread; each
 * device packs its finished tiles into one slab, which crosses xGMI as one peer copy into
 * scenes[0]'s device; the assembled frame is copied to the HOST buffer once.  Pixels of tiles not
 * listed are 0.  stats: NULL or [n_scenes], one entry per device. */
int fjgpu_render_frame_multi(fjgpu_scene *const *scenes, int n_scenes, const fj_render_desc *render,
    const int32_t *tile_ids, int n_tiles, float *h_framebuffer, fjgpu_stats *stats);

/* Tile slabs of the multi-process split (one process per GPU, bench.py / distributed.py: tile t belongs to rank t % N and
 * the finished tiles travel to rank 0 in ONE exchange): pack copies the rectangles rects[k] = (xmin, ymin, xmax, ymax)
 * (DEVICE array, n_tiles x 4 int32) of the DEVICE framebuffer d_fb (xres pixels per row, RGBA f32) into d_slab, tile k at
 * pixel k * tile_px, rows of the rectangle's own width; unpack is the inverse (a rectangle of zero area is skipped: padding
 * entries of the shorter ranks).  One kernel launch each on `hip_stream`; no synchronisation. */
int fjgpu_pack_tiles(const float *d_fb, int xres, const int32_t *d_rects, int n_tiles, int tile_px, float *d_slab, void *hip_stream);
int fjgpu_unpack_tiles(float *d_fb, int xres, const int32_t *d_rects, int n_tiles, int tile_px, const float *d_slab, void *hip_stream);

/* Closest hit of n rays against group `group` (HOST arrays; copied in/out).
 * rays [n][8] = orig xyz, dir xyz, tmin, tmax.  out_t [n] (DBL_MAX on miss),
 * out_ids [n][2] = instance, primitive (-1 on miss), out_uv [n][2] = barycentric
 * u, v of the hit (may be NULL). */
int fjgpu_trace(fjgpu_scene *scene, int group, int n, const double *rays,
    double *out_t, int32_t *out_ids, double *out_uv, fjgpu_stats *stats);

/* Tunables (all have defaults): "batch_tiles" tiles per wavefront batch, "batch_samples" samples per batch where batch_tiles is 0
 * (0, the default: as many as the memory budget holds -- fastest where frames repeat; a caller that renders one frame per scene
 * sets a few M: the work arena shrinks from ~110 GB to a few GB at the headline size and the cold frame starts sooner),
 * "count_nodes" 0/1 enable traversal event counters, "overlap_shadow" 0/1 run the shadow work of
 * a recursion level on its own stream, concurrent with the next level; "release_work" 1: hand the work arena back to the driver
 * now (the scene stays resident; the next render call allocates again). Returns 0 or FJGPU_EINVAL. */
int fjgpu_set_option(fjgpu_scene *scene, const char *name, long value);

/* Process-wide options read by fjgpu_scene_create (which stands for the reference's
 * build_accelerators(), src/fj_scene_interface.cc:1161-1202): "device_build" -1/0/1/2 -- 0: the host's binned-SAH
 * build; 1: build the BLAS of meshes on the GPU by locally-ordered clustering (surface-area agglomeration over the
 * Morton order); 2: on the GPU as the radix tree of the Morton codes (fastest build, slowest tree); -1 (default): not
 * set -- the host build, or, for scenes created while "single_frame_build" is 1, the GPU's clustering build.
 * "multi_exchange" 0/1: how fjgpu_render_frame_multi moves the devices' tile slabs to the first device -- 0 (default): one hipMemcpyPeer per
 * device (over xGMI where peer access exists; nothing to bring up, which is what a one-frame process wants); 1: RCCL -- once EVERY device has
 * rendered and packed its share, one ncclGroupStart / ncclGroupEnd holds each device's ncclSend and the first device's ncclRecv for it (the
 * exchange is collective over success: a device that failed returns its error and no rank is left waiting; librccl.so is loaded at first use;
 * devices must be distinct; a communicator that cannot be created falls back to the peer copies).  Same bytes over the same links.
 * "cold_start" 1/0 (default 1): a scene's FIRST fjgpu_render_tiles call renders in batches of 16 M samples whatever the memory would hold (a work
 * arena of ~14 GB instead of ~110 GB at 1080p / 64 spp: the first image after 0.15 s instead of the seconds the large allocation can take);
 * the second call sizes its batches by memory and pays for the growth once.  0: the first call already does.
 * "anyhit_filter_off" 0/1 (diagnostics, default 0): the lean any-hit walk's conservative f32 triangle filter decides nothing -- every leaf test goes
 * through the walk's exact phase (the reference's FP64 statements on the ray rebuilt from the queue entry).  Same image, several times the
 * walk's time: what tests use to put that phase under load.
 * "speculative_walk" 1/0 (default 1): the closest-hit walk of the next recursion level is enqueued behind a level's shading launch, before the
 * host has read how many rays that launch emitted (the walk reads the count from device memory): no host round trip between the levels.
 * "cold_batch_samples" n: ... in batches of n samples instead (0 = back to 16 M).
 * "single_frame_build" 0/1 (per calling THREAD): the caller renders ONE frame per scene it creates (SiRenderScene switches it on around its
 * scene creation; scenes created on other threads meanwhile are unaffected): the 0.4 s of host build the tree's 3-6 % faster frames would need many frames to earn back are not spent.
 * "device_tlas" 1/0 (default 1): the instance level of every group (the reference's BVHAccelerator over
 * ObjectInstances, src/fj_bvh_accelerator.cc:253-334) is built on the GPU; 0: by the host's builder (the
 * same list).  "tlas_verify" 0/1: scene creation also builds it on the host and fails on any byte that differs.
 * "ray_sort" -1..9: grid bits per axis of the ray-queue sort in front of the closest-hit walk (rays of
 * recursion level >= 1 in direction-octant / origin-cell order); 0 = off, -1 (default) = 7 bits in scenes
 * whose shaders scatter (glass, pathtracing), off elsewhere.  "ray_sort_min": launches of fewer rays keep
 * queue order (default 65536).  "split_shadow" 1/0 (default 1): in scenes served by the lean any-hit walk, a shadow
 * ray into a group of several instances is queued once per instance whose box it passes (joined by a counter)
 * instead of walking the group's instance level in the traversal kernel (C2: 134 -> 121 ms per frame).
 * "compact_squeue" 1/0 (default 1): shadow-queue records of 56 instead of 80 bytes (no direction / distance: the walk rebuilds them from the origin and
 * the light sample, same statements, same bits) where the lean or the curve any-hit walk consumes the queue and the lights are point / dome lights.
 * "flat_groups" 0/1/2 (default 1): scenes created from now on whose closest-hit rays are incoherent (glass, pathtracing shaders) and whose groups hold only
 * small static meshes built on the host (at most 2 M triangles and 32 instances per group, 80 instances in the scene: the walk keeps every instance's
 * M^-1 in LDS) get ONE world-space culling tree per group over the triangles of all its instances; the exact tests stay in object space
 * (k_trace_closest_flat).  0: the instance loop of k_trace_closest_phased.  2: scenes with coherent rays as well (an experiment switch: their walks lose
 * with it -- C2's closest-hit side 21 -> 29 ms, C3's 24 -> 104 with FJGPU_FLAT_MAX_TRIS_LOG2=25 -- profiles/r05_flat_for_coherent_scenes.txt).
 * "curve_anyhit" 1/0 (default 1): scenes created from now on that hold curve sets, no time-sampled motion and only opaque occluders walk their
 * shadow rays with k_shadow_anyhit_curves (phase-scheduled, ribbon tests as a phase of their own); 0: with the general k_shadow_trace.
 * "inst_lds" 1/0 (default 1): scenes created from now on whose instance level is small (79 threaded nodes / 33 instances /
 * 24 groups for the closest-hit and general shadow walks of mesh scenes, 39 / 16 / 12 for the phase-scheduled walk (the flat-group
 * walk keeps only M^-1 of up to 80 instances), 15 / 6 / 12 for the walks of curve scenes, 292 nodes for the light loop; one instance is a 304-byte DInstEntry) have it
 * copied to LDS by every block of those walks; 0, or a scene beyond its kernel's budget: it is read from global memory (same
 * results; option 0, or a scene of more than 80 instances, also builds no flat groups).
 * "batch_tiles" n: scenes created from now on start with the per-scene option of that name set to n (0 = sized by
 * memory; how a host that never sees the scene handle -- SiRenderScene -- cuts a frame into batches).  0 or FJGPU_EINVAL. */
int fjgpu_global_option(const char *name, long value);

/* Inspection, no device needed: the instance level of `group` as the HOST builder lays it out (the device builds
 * the same list, option "tlas_verify") -- a depth-first node list, node k: out_inst[k] = instance index or -1 for an
 * inner node, out_skip[k] = index (within this list) of the node after an inner node's subtree, out_box[6 k..] = the
 * instance's reference box / the inner node's widened union box.  The leaves are in the depth-first order of the
 * reference's BVHAccelerator over the group's instances (src/fj_bvh_accelerator.cc:253-334).  Writes at most `cap`
 * nodes (any out pointer may be NULL); returns the node count, or a negative error. */
int fjgpu_host_instance_level(const fj_scene_desc *desc, int group, int32_t *out_inst, int32_t *out_skip, double *out_box, int cap);

/* Facts about the built device scene (for measurement: record sizes of the actual layout).
 * "node_record_bytes" (128; "anyhit_node_record_bytes" 64: the lean any-hit walk reads the quantised
 * twin of a node), "tri_record_bytes" (36 when every mesh is stored as exact f32
 * triangles, else 72), "blas_nodes", "stack_need", "lean_anyhit" (1: shadow rays are walked by
 * k_shadow_anyhit, 0: not), "curve_anyhit" (1: by k_shadow_anyhit_curves -- curve scene, every occluder opaque, no motion; both 0: by the
 * general k_shadow_trace), "closest_kernel" (0 k_trace_closest, 1 k_trace_closest_phased,
 * 2 its curve instantiation, 3 its motion instantiation, 4 k_trace_closest_flat: one world-space tree per group), "closest_node_record_bytes" (64: the closest-hit walk reads the
 * quantised nodes too; 128 in scenes with curve sets or motion), "has_curves", "has_motion", "scene_bytes" (device memory of the
 * resident scene: BLAS, instance level, lights, textures), "work_bytes" (the wavefront work arena -- queues, accumulators -- as the
 * largest call so far sized it; scene_bytes + work_bytes = the HBM this scene holds).  Returns 0 or FJGPU_EINVAL. */
int fjgpu_scene_query(const fjgpu_scene *scene, const char *name, double *value);

/* Diagnostics: the host-side math that feeds geometry to the device (matrices,
 * RNG tables, sampler margins), exported so it can be pinned against the
 * reference's golden vectors on a machine without a GPU. */
void fjgpu_host_make_transform(int transform_order, int rotate_order, const double *trs9, double *M16, double *Minv16);
void fjgpu_host_xorshift_f01(int n, double *out);
void fjgpu_host_sampler_margin(const fj_render_desc *render, int32_t *margin2);
double fjgpu_host_camera_uv_size_y(double fov);

/* Diagnostics: RCCL bring-up on one device.  fjgpu_render_frame_multi can move the devices' tile slabs with RCCL's grouped point-to-point
 * calls (global option "multi_exchange" 1); this runs the same calls -- ncclCommInitAll, ncclGroupStart, ncclSend + ncclRecv, ncclGroupEnd --
 * on a communicator of ONE rank (`device` sends n_floats to itself) and checks the data: librccl.so is loadable, its symbols bind, a
 * collective-library kernel runs.  0, or FJGPU_ENODEV with the reason. */
int fjgpu_dev_rccl_selftest(int device, int n_floats);

/* Diagnostics: the ray-queue sort on its own (fjgpu_raysort.hip: the hand-written wave64 LSD radix sort that orders the rays
 * of recursion level >= 1 in front of the closest-hit walk).  Sorts the pairs (keys[i], i) of HOST arrays on `device`,
 * stable, over the low key_bits bits (1 .. 32): keys_out ascending, perm[k] = index of the k-th pair (either may be NULL; without
 * keys_out the last pass writes the permutation only, as the product's sort does).
 * sort_ms (may be NULL): the fastest of `repeats` device-side runs, HIP events around the sort's launches alone. */
int fjgpu_dev_sort_pairs(int device, const uint32_t *keys, int n, int key_bits, uint32_t *keys_out, uint32_t *perm, int repeats, double *sort_ms);

/* Human-readable message for the last error on this thread. */
const char *fjgpu_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* FJGPU_H */
