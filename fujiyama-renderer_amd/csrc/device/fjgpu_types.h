// fjgpu_types.h -- device-resident data layout of the MI355X core.
// Shared by the host-side scene builder (fjgpu_build.cc) and the HIP kernels.
#ifndef FJGPU_TYPES_H
#define FJGPU_TYPES_H

#include <stdint.h>
#include "fj_scene_desc.h"

// ---- BLAS node: a 4-wide BVH node, 128 B = eight 16-byte loads.  Child boxes are f32,
// rounded OUTWARD from the f64 primitive bounds (>= 1 ulp), so a box miss proves that no
// f64 primitive test below it would report a hit.
// child ref: 0xffffffff -> no child; bit 31 set -> leaf, payload = (first_prim << 3) |
//            (count - 1); else index of the child DNode.
struct DNode {
  float box[4][6];             // child k: (min, max) pairs of x, y, z -- one packed fma per axis (slab32_test)
  uint32_t child[4];
  uint32_t pad[4];
};
static_assert(sizeof(DNode) == 128, "DNode must be 128 bytes");

// ---- the same node for the lean any-hit walk: 64 B = four 16-byte loads instead of seven (the
// walk is bound by the L1's request rate: one request per lane and load instruction, because every
// lane reads another node).  Child boxes are quantised to a 65536^3 grid over the primitive set's
// padded bounds (DAnyInst.qorigin / qcell), outward: min = floor, max = ceil, so the decoded box
// contains the f32 box of the DNode with the same index.  Same child refs, same node indices.
struct DNodeQ {
  uint16_t q[4][6];            // child k: (min, max) pairs of x, y, z in grid units
  uint32_t child[4];
};
static_assert(sizeof(DNodeQ) == 64, "DNodeQ must be 64 bytes");

#ifndef FJ_CURVE_QNODES
#define FJ_CURVE_QNODES 1                // 1: the curve instantiations (scenes with curve sets, no motion) read quantised 64-byte nodes too
#endif
#ifndef FJ_CLOSEST_QNODES
#define FJ_CLOSEST_QNODES 1              // 0: the closest-hit walk reads the 128-byte f32 nodes in every instantiation
#endif

#define FJ_NO_CHILD 0xffffffffu
#define FJ_LEAF_FLAG 0x80000000u
#ifndef FJ_MAX_LEAF_PRIMS
#define FJ_MAX_LEAF_PRIMS 4
#endif
#define FJ_BVH_MAX_DEPTH 40          // depth bound of the binary tree the builder collapses
#ifndef FJ_STACK_LDS
#define FJ_STACK_LDS 32              // traversal stack entries per lane kept in LDS; deeper
                                     // entries (rare) go to a global overflow area
#endif
// kernels instantiated with the ribbon test keep fewer stack entries in LDS: they also hold the
// ray-space frame and one cached node of the curve subdivision per lane (fjgpu_dev_curve.h)
#ifndef FJ_STACK_LDS_CURVES
#define FJ_STACK_LDS_CURVES 20          // (measured in round 3: 20 / 24 entries and 40 without the subdivision cache give the same C5 walk, 1740 / 1743 / 1767 ms)
#endif
// the lean any-hit walk runs SIX blocks per CU: 12 entries x 1 KB per block + 12 KB for the object-space rays
// (FJ_ANYHIT_RAY_LDS, fjgpu_dev_anyhit.h) = 24 KB per block; 14 entries: the same time
#ifndef FJ_STACK_LDS_ANYHIT
#define FJ_STACK_LDS_ANYHIT 12
#endif
#ifndef FJ_STACK_LDS_MIN
#define FJ_STACK_LDS_MIN 12          // smallest of the three (sizes the global overflow area)
#endif

// ---- primitive set (one mesh or one curve set): its BLAS + attribute arrays
struct DPrimSet {
  const DNode *nodes;
  const DNodeQ *qnodes;        // the same tree with quantised boxes (meshes; the lean any-hit walk) or null
  const double *tri_verts;     // [n_prims][9]  pre-gathered v0 v1 v2 in BLAS leaf order (72 B / tri), or null:
  const float *tri_verts32;    // [n_prims][9]  the same values as f32 (36 B / tri) when EVERY coordinate of the
                               //               mesh is exactly representable in f32 (PLY data is); widened to
                               //               f64 on load, so the test sees identical operands
  const double *tri_vel;       // [n_prims][9]  per-vertex velocities in leaf order (Mesh::ray_intersect
                               //               moves each vertex by time * velocity) or null = static mesh
  const double *velocity;      // [n_points][3] the same per point (derivatives in the shading kernel) or null
  const uint32_t *prim_ids;    // [n_prims]     original primitive id of leaf slot k
  const double *P;             // [n_points][3] object-space positions (attribute fetch)
  const double *N;             // [n_points][3] or null
  const double *vN;            // [n_faces][3][3] per-corner normals (fj_mesh_desc.vertex_N) or null: they win over N
  const float *uv;             // [n_points][2] or null
  const int32_t *indices;      // [n_faces][3]
  const int32_t *face_group;   // [n_faces] or null
  // curves
  const double *curve_cp;      // [n_curves][12] pre-gathered control points (BLAS order)
  const double *curve_width;   // [n_curves][2]  end widths (BLAS order)
  const float *curve_Cd;       // [n_curves][6]  end colours (BLAS order)
  const int8_t *curve_depth;   // [n_curves]     cached split depth (BLAS order)
  const float *curve_capsule;  // [n_curves][8]  A xyz, B xyz, reach, the bits of prim_ids[slot]: the PIECE of the curve a BLAS slot stands for lies
                               //                within `reach` (ribbon radius included) of the segment AB; or null
  const double *curve_vel;     // [n_curves][12] control-point velocities (BLAS order) or null: Curve::ray_intersect
                               //                moves each control point by time * velocity
  double bounds[6];            // Accelerator::bounds_ = primset bounds + 1e-4 (object space)
  // curves only: geometry of the reference's uniform grid (origin = bounds min, cell size,
  // cell counts); needed to reproduce its "hit point must lie in a cell that lists the
  // primitive" acceptance rule for ribbons
  double grid_cell[3];
  int32_t grid_n[3];
  int32_t pad2;
  uint32_t root;               // child ref of the root
  int32_t type;                // FJ_PRIMSET_*
  int32_t n_prims;
  int32_t pad;
};

// ---- object instance: matrices precomputed on the host with the reference's
// arithmetic (static transforms; time-sampled TRS is a "next" row)
struct DInstance {
  double M[12];                // rows 0..2 of the 4x4 (row 3 is 0 0 0 1)
  double Minv[12];
  double wbounds[6];           // the REFERENCE's instance bounds (merge_sampled_bounds: |scale|!), exact
  int32_t primset;             // index into DPrimSet table
  int32_t n_shaders;
  int32_t shaders[FJ_MAX_SHADING_GROUPS];
  int32_t reflect_target, refract_target, shadow_target;
  int32_t xform;               // time-sampled transform: index into DScene.xforms (M / Minv above hold
                               // the time-0 matrices); -1 = static
  int32_t pn_prims;            // copies of the primitive set's entry data (DPrimSet bounds / nodes / root / n_prims,
  uint32_t proot;              // filled at upload): entering an instance costs ONE dependent load after the
  double pbounds[6];           // instance-level node instead of instance -> primitive set
  const DNode *pnodes;
  const DNodeQ *pqnodes;       // meshes: the quantised twin of the node array and its grid (plane = qorigin + q * qcell):
  double qorigin[3], qcell[3]; // the closest-hit walk of scenes without curve sets reads these 64-byte nodes too
  // ... and of what the SHADING kernel reads of a mesh (DPrimSet indices / N / uv / face_group and the type): a hit's attribute fetch is
  // hit -> instance -> indices -> normals instead of hit -> instance -> primitive set -> indices -> normals, one dependent gather less in a
  // kernel that waits for nothing else (k_shade)
  const int32_t *sh_indices;
  const double *sh_N;
  const float *sh_uv;
  const int32_t *sh_face_group;
  const double *sh_vN;         // per-corner normals of the mesh or null
  int32_t sh_type;             // FJ_PRIMSET_*
  int32_t sh_pad;
};

// What the closest-hit walks and the general shadow walk read when a ray enters an instance, packed.  A scene whose whole
// instance level (every DTNode, one of these per instance, every DGroup) fits the kernel's budget is copied to LDS by each
// block of those (persistent) walks, so the dependent loads of the instance loop (group -> node -> instance -> root) cost
// LDS instead of L1 / L2 round trips.
struct DInstEntry {
  double Minv[12];             // world -> object (time 0; DInstance.xform >= 0: evaluated per ray instead)
  double tbounds[6];           // a TIGHT world-space box of the instance's geometry (the image of pbounds under M, slightly widened; +-inf for
                               // time-sampled transforms): pure culling, tested before anything else of the instance is computed.  The reference's
                               // own instance box (DTNode.box, merge_sampled_bounds) is the bounding-sphere CUBE of a rotated object -- the walls
                               // of a Cornell box each "contain" the whole room -- and decides results only where a ray misses it; a ray that
                               // misses THIS box cannot hit the instance whatever that one says
  double pbounds[6];
  double qorigin[3], qcell[3];
  const void *nodes;           // the node array the scene's walks read: DNodeQ (meshes, scenes without curve sets and motion) or DNode
  const double *tri_verts;     // the primitive set's leaf-order arrays (DPrimSet): the leaf phase reads them through this record
  const float *tri_verts32;    // (curve sets: DPrimSet.curve_capsule -- they have no triangles; k_shadow_anyhit_curves reads the capsules through this record)
  const double *tri_vel;
  const uint32_t *prim_ids;
  int32_t ptype;               // FJ_PRIMSET_*
  int32_t pad;
  uint32_t proot;
  int32_t pn_prims;
  int32_t primset;
  int32_t xform;
};
#define FJ_INST_LDS_ENTRY_WORDS 38      // sizeof(DInstEntry) / 8
// budgets: DTNodes (56 B), DInstEntry records (304 B = 38 words), DGroups (64 B)
#define FJ_INST_LDS_NODES 39            // 8 072 bytes next to the 32 KB of stacks of a block: the phased walk still has 4 blocks per CU.  (The curve
                                        // instantiations have a budget of their own below; the motion kernels read global memory.)
#define FJ_INST_LDS_INSTS 16
#define FJ_INST_LDS_GROUPS 12
#define FJ_INST_LDS_NODES_BIG 79        // the walks with 3 blocks per CU (k_trace_closest, k_shadow_trace of mesh scenes): 16 200 bytes
#define FJ_INST_LDS_INSTS_BIG 33
#define FJ_INST_LDS_GROUPS_BIG 24
#define FJ_INST_LDS_NODES_CURVES 15     // the curve instantiations (3 blocks per CU, 20 KB of stacks + 28 KB of ray space per block): 3 656 bytes
#define FJ_INST_LDS_INSTS_CURVES 6
#define FJ_INST_LDS_GROUPS_CURVES 12
// k_trace_closest_flat needs M^-1 of a candidate's instance and nothing else of the instance level: a table of 12 doubles per instance in LDS
#define FJ_FLAT_LDS_INSTS 80            // 7 680 bytes (what the phased walk's instance level takes: 4 blocks per CU either way)
#define FJ_INST_LDS_BYTES (FJ_INST_LDS_NODES * 56 + FJ_INST_LDS_INSTS * 8 * FJ_INST_LDS_ENTRY_WORDS + FJ_INST_LDS_GROUPS * 64)

// Everything the lean any-hit walk needs to enter an instance, in one record (one dependent
// load after the queue entry instead of instance -> primitive set -> pointers)
struct DAnyInst {
  double Minv[12];             // world -> object
  double bounds[6];            // the primitive set's padded box (object space)
  double qorigin[3], qcell[3]; // grid of the quantised nodes: plane = qorigin + q * qcell
  // node and triangle arrays as 32-bit offsets from DScene.blas_base (two registers per lane
  // instead of four), both in units of 128 B
  uint32_t node_base;          // DNodeQ array
  uint32_t tri_base;           // pre-gathered triangles: f32 (36 B) or f64 (72 B) records
  uint32_t root;
  uint32_t tris_f32;           // 1: f32 records
  int32_t n_prims;
  float fbound;                // >= |any coordinate| of `bounds` (the f32 triangle filter's magnitude guard, fjgpu_tri_filter.h)
  uint32_t pad[2];
};
static_assert(sizeof(DAnyInst) == 224, "DAnyInst layout");

// Instance level of a group: a THREADED bounding-volume hierarchy (depth-first node list with
// skip links, no stack).  Walk: i = first; a leaf (inst >= 0) is a candidate instance, go to
// i + 1; an inner node whose box the ray misses jumps to `skip`, else go to i + 1.  The leaves are
// in the depth-first order of the REFERENCE's instance BVH (src/fj_bvh_accelerator.cc:253-334: the
// first instance visited keeps an exactly equal t); subtrees of up to FJ_TLAS_FLAT leaves have no
// inner node (they are scanned).  Inner boxes are unions of the instances' reference boxes,
// slightly widened.  Built on the device (fjgpu_tlas.hip) or, identically, on the host.
#define FJ_TLAS_FLAT 4
struct DTNode {
  double box[6];               // inner node: union box (widened); leaf: the instance's reference box (DInstance.wbounds:
                               // a ray that misses it never touches the instance record)
  int32_t inst;                // instance index, or -1 for an inner node
  int32_t skip;                // inner nodes: index of the node after this subtree
};

struct DGroup {
  int32_t first, count;        // slice of the DTNode array (count = nodes, not instances)
  int32_t all_opaque;          // every shader reachable in the group has Os == 1 -> any-hit shadows
  int32_t n_instances;
  double sbounds[6];           // single-instance group: its instance's bounds + 1e-4 (the group accelerator's box)
};

// ---- FLAT groups (scenes whose closest-hit rays are incoherent: glass, pathtracing; fjgpu_dev_flat.h).  A group whose instances are all small
// static meshes (the walls and objects of a Cornell box) gets ONE world-space culling tree over the triangles of all of them: a ray walks that
// tree once instead of looping over the instances, and a leaf names (instance, triangle) -- the exact tests stay what they are, in the
// instance's object space with its M^-1, after the reference's own instance box has been passed (once per ray and instance).
// one leaf slot of the flat tree, 48 bytes = three 16-byte loads: the triangle in its instance's OBJECT space (f32, exact: only sets whose
// coordinates are all representable in f32 are flattened), (instance << 8) | position of the instance in the group's visiting order, primitive id
struct DFlatRef { float v[9]; uint32_t inst_ord; uint32_t pid; uint32_t pad; };
static_assert(sizeof(DFlatRef) == 48, "DFlatRef must be 48 bytes");
struct DFlat {
  const DNodeQ *nodes;         // quantised 4-wide tree over the world-space boxes of the triangles
  const DFlatRef *refs;        // [n_prims], leaf order of that tree
  const double *refbox;        // [n_inst][6]: the REFERENCE's box of the instance at each position (DInstance.wbounds; a single-instance group: DGroup.sbounds)
  double qorigin[3], qcell[3]; // grid of the quantised nodes
  uint32_t root;
  int32_t n_inst, n_prims, pad;
};

struct DTexture {
  const float *tiles;
  int32_t width, height, nchannels, tilesize;
};

struct DLightSample {
  double P[3];
  float Cl[3];                 // Light::Illuminate result (position independent for point / dome)
  int32_t light;
  int32_t type;                // FJ_POINT_LIGHT / FJ_DOME_LIGHT: P and Cl above are final.  FJ_GRID_LIGHT /
                               // FJ_SPHERE_LIGHT: the position is drawn per shading event (DAreaLight)
  int32_t ordinal;             // index of this sample inside its light (0 restarts the light's stream)
};

// RectangleLight / SphereLight (src/fj_rectangle_light.cc, src/fj_sphere_light.cc) at time 0
struct DAreaLight {
  double M[12];                // light transform
  double N[3];                 // rectangle: normalize(M * (0,1,0))
  float color[3];
  float sample_intensity;      // intensity / sample_count
  int32_t double_sided;
  int32_t pad;
};

struct DLightHair;

struct DScene {
  const DPrimSet *primsets;
  const DInstance *instances;
  const DGroup *groups;
  const DTNode *group_nodes;
  const DInstEntry *inst_entries;  // [n_instances]
  int32_t n_group_nodes;
  int32_t inst_lds;            // 0: the walks read the instance level from global memory (option / FJGPU_NO_INST_LDS); else where it fits
  const fj_shader_desc *shaders;
  const DTexture *textures;
  const DLightSample *light_samples;
  DLightHair *lrec_hair;       // work buffer (set per render call) or null
  uint32_t shadow_queue_cap;   // entries of the shadow-ray queue (a walk never reads past it, whatever the slot counter says)
  uint32_t cam_slot0;          // implicit camera rays: sample slot of ray 0 of the launch (a level walked in chunks)
  uint32_t *shadow_join;       // work buffer or null.  Lean any-hit walk with shadow groups of several instances: a ray
                               // whose world-space test passes k >= 2 instance boxes is queued k times, once per
                               // instance, and all k entries name one slot here (DShadowRay.tindex = slot + 1; 0 = a
                               // single entry): (k << 16) | number of entries that reached the light so far.  The
                               // entry that completes the count adds the light; an occluded entry adds nothing.
  const DFlat *flats;              // [n_groups] when EVERY group of the scene is flat (k_trace_closest_flat walks them), else null
  const DAreaLight *area_lights;   // [n_lights] (entries of other light types unused) or null
  const DAnyInst *any_insts;       // [n_instances] (static mesh instances; the lean any-hit walk)
  const char *blas_base;           // lowest address of any BLAS node / triangle array (DAnyInst offsets); null: the
                                   // arrays do not fit 32-bit offsets and the general shadow walk is used
  int32_t n_light_samples;
  int32_t n_instances, n_groups, n_primsets;
  uint32_t *stack_overflow;    // [stack_need - FJ_STACK_LDS][persistent threads] or null (see TravStack)
  uint32_t *stack_overflow_shadow;   // the same for the shadow kernels (they run concurrently on their own stream)
  int32_t all_opaque;          // every group is all_opaque: shadow rays run the lean any-hit kernel
  int32_t has_area;            // any rectangle / sphere light: the light loop draws positions per event
  int32_t has_hair;            // any HairShader: selects the light-loop instantiation with its illuminance term
  int32_t has_curves;          // any curve primset: selects the traversal instantiation with the ribbon test
  int32_t target_group;
  int32_t curve_anyhit;        // option "curve_anyhit": shadow rays of curve scenes whose occluders are all opaque run k_shadow_anyhit_curves (else the general walk)
  int32_t incoherent_rays;     // some shader emits two children per hit or diffuse bounces (glass, pathtracing): the
                               // closest-hit walk of such mesh scenes is the phase-scheduled one
  const double *cam_uv;        // implicit camera rays: the (u, v) table of the batch's samples (sample slot = ray index of
                               // level 0; fjgpu_dev_shade.h) while level 0 is walked and shaded, else null
  int32_t compact_squeue;      // 1: the shadow-ray queue holds DShadowRayC records (set per render call)
  int32_t pad_cs_;
  const uint32_t *cam_tk;      // ... in scenes whose random streams are keyed by the sample's uid (pathtracing shader, area lights): (tile id, index of
                               // the sample in its tile), two words per sample slot of the batch: 8 bytes instead of the 48-byte path record; else null
  const uint32_t *ray_perm;    // closest-hit launch over a SORTED ray queue: entry k of the launch is ray ray_perm[k]
  const double *trace_ranges;  // fjgpu_trace: (tmin, tmax) per ray, given by the caller (render calls: null, the range follows from the ray's class)
  const uint32_t *trace_n_dev; // closest-hit launch enqueued BEFORE the host knows its ray count: the walk takes min(n, *trace_n_dev) rays (null: n)
                               // (hits are written to the ray's own slot); null = queue order
  // time-sampled transforms (motion blur): evaluated per ray at the sample's time
  const fj_xform_desc *xforms; // instances with DInstance.xform >= 0
  const fj_xform_desc *cam_xform;   // null = static camera
  const double *time_tab;      // draw k of the per-tile time stream (sample index in the tile -> [0,1])
  double time_start, time_end; // Renderer sample_time_range
  int32_t has_motion;          // any time-sampled instance transform: traversal / shading evaluate them
  int32_t multi_shadow_groups; // a shadow target group of more than one instance can receive shadow rays (else every
                               // shadow-queue entry names its instance: the leaner any-hit instantiation)
  // camera (static case): eye, matrix rows, uv_size
  double cam_M[12];
  double cam_uv_size[2];
  double cam_znear, cam_zfar;
};

// ---- wavefront records
struct DRay {                  // 48 B: origin and direction of Ray (src/fj_ray.h:11-22).  Its RANGE is not stored (round 6): a queued ray's [tmin, tmax]
  double o[3], d[3];           // follows from its class -- camera rays: (znear, zfar); every SlTrace child: tmax 1000 and tmin .001 or .0001 (DPath.cxt bit 7) --,
};                             // and fjgpu_trace's caller-given ranges travel in DScene.trace_ranges
static_assert(sizeof(DRay) == 48, "DRay must be 48 bytes");
#define FJ_CXT_TMIN_1E4 0x80u   // DPath.cxt bit 7: the ray's tmin is .0001 (else .001); the low bits are the context

enum { CXT_CAMERA_RAY = 0, CXT_SHADOW_RAY, CXT_DIFFUSE_RAY, CXT_REFLECT_RAY, CXT_REFRACT_RAY };

struct DPath {                 // 36 B per-ray path state
  uint32_t sample;             // destination sample slot in the batch
  float T[3];                  // RGB throughput down to this ray
  uint8_t cxt, ddepth, rdepth, tdepth;   // ray context (+ FJ_CXT_TMIN_1E4) + diffuse / reflect / refract depths
  int32_t group;               // trace target group
  uint32_t flags;              // bit0: apply pow(fc, t_hit) at this ray's hit -- fc, the filter colour a refraction child carries (glass / pathtracing), sits in
                               // a side array of its own (3 floats per slot, ShadeParams.fc_in / fc_out: only scenes with such shaders have it); bits 1..31: index of the sample in its tile
                               // (its time is time_tab[flags >> 1]; every ray of the sample's path tree carries it)
  uint32_t rng;                // pathtracing RNG contract: path key (child k of key p = 4 p + k)
  uint32_t uid;                // RNG contract: sample_uid(tile id, sample index in the tile) (fjgpu_dev_shade.h)
};
static_assert(sizeof(DPath) == 36, "DPath must be 36 bytes");

struct DHit {                  // 32 B
  double t, u, v;
  int32_t inst, prim;          // inst < 0: miss
};

struct DLightRec {             // 80 B: one shading event that gathers direct light
  double P[3];
  double N[3];                 // illuminance axis (Nf for plastic, N for hair)
  float W[3];                  // throughput * diffuse * diffuse_map (plastic) or throughput (hair)
  uint32_t sample;
  int32_t group;               // shadow target of the shaded object
  int32_t kind;                // bit 0: 0 lambert (plastic), 1 kajiya-kay (hair: + DLightHair of the same slot); bits 1..31: the
                               // sample's index in its tile (time of the shadow rays: DShadowRay.tindex)
  uint32_t uid;                // DPath.uid of the shading ray
  uint32_t key;                // path key of the shading ray (area-light stream, with uid)
};

struct DLightHair {            // 64 B, only in scenes with a HairShader: same slot as its DLightRec
  double aux[6];               // tangent (3), I (3)
  float Cd[3];                 // Cd * diffuse
  uint32_t pad;
};

struct DShadowRay {            // 80 B: a shadow ray that survived the instance-box cull
  double o[3], d[3], tmax;
  float c[3];                  // W * Kd * Cl: added to the sample, scaled by (1 - occluder Os)
  uint32_t sample;
  int32_t group;               // shadow target group; or ~instance when the light loop already settled that this
                               // one instance is the only candidate (single-instance group, lean any-hit walk)
  uint32_t tindex;             // sample index in its tile: the ray's time is time_tab[tindex] (motion blur)
};

// The COMPACT form of the same record, 56 B (DScene.compact_squeue: scenes whose shadow rays run the lean or the curve any-hit walk and whose lights are
// point / dome lights without motion): direction and distance are not stored -- the walk rebuilds them from the origin and the light sample's position with the
// light loop's own statements (Ln = Pl - Ps; distance = sqrt(dot(Ln, Ln)); Ln *= 1 / distance: the same bits).  The origin stays: the light records are
// recycled (ring of two or three buffers per batch) before the walk runs.
struct DShadowRayC {
  double o[3];
  float c[3];
  uint32_t sample;
  int32_t group;
  uint32_t tindex;
  uint32_t light;              // index into DScene.light_samples
  uint32_t pad;
};
static_assert(sizeof(DShadowRayC) == 56, "DShadowRayC must be 56 bytes");

struct DCounters {
  // Counters that many waves add to at the same time sit on their own 128-byte lines:
  // same-line atomics serialise in L2 (and a queue's slot counter must not wait behind tallies).
  unsigned long long rays[5];  // per context (fj_ray_counts order: camera shadow diffuse reflect refract)
  unsigned long long nodes, prims, insts, traced, squeued;
  unsigned long long sh_nodes, sh_prims, sh_insts;   // the shadow walk's share of nodes / prims / insts
  unsigned long long pad0_[3];
  uint32_t next_count;         // entries appended to the next ray queue                      (line 1)
  uint32_t light_count;        // entries appended to the light-record queue
  uint32_t overflow;
  uint32_t cam_count;          // adaptive sampler: camera rays queued by the current level
  uint32_t pad1_[28];
  uint32_t shadow_count;       // slots reserved in the shadow-ray queue                      (line 2)
  uint32_t shadow_head;        // persistent shadow traversal: next unclaimed queue index (shadow stream)
  uint32_t pad2_[30];
  uint32_t trace_head;         // persistent closest-hit traversal: next unclaimed queue index (line 3)
  uint32_t cull_head;          // light loop: next unclaimed light record of the current launch
  uint32_t pad3_[30];
  // the persistent walks claim per XCD (QueueClaim, fjgpu_dev_traverse.h): the shadow queue is cut into 8 regions, the waves of XCD x
  // start in region x (their L2 then holds the nodes of ONE stretch of the queue, not of eight) and
  // move on to the next region when theirs is empty; one head per 128-byte line
  uint32_t shadow_xcd_head[8][32];
  uint32_t trace_xcd_head[8][32];     // the same for the closest-hit walk (the two may run on streams of their own)
  uint32_t join_count;         // join slots handed out to shadow rays with several candidate instances (DScene.shadow_join);
  uint32_t pad4_[31];          // on a line of its own: one atomic per wave and light
};

#endif
