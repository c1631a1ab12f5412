// fjgpu_dev_traverse.h -- persistent per-lane traversal engine and the closest-hit kernel.
// Part of the kernels translation unit: included by fjgpu_kernels.hip only (device code,
// compiled with -ffp-contract=off; see the header of that file).
#ifndef FJGPU_DEV_TRAVERSE_H
#define FJGPU_DEV_TRAVERSE_H

// ----------------------------------------------------- persistent traversal
// One traversal engine for closest-hit and any-hit rays, written as a per-lane
// state machine so that a lane that finishes its ray is refilled from the
// wave's slice of the queue instead of idling until the slowest lane of the
// wave is done (ray costs are heavy tailed: most shadow rays leave the BLAS
// after a few nodes, a few walk hundreds).  Each wave owns a contiguous slice
// of the ray queue (static split, no global work counter) and hands indices to
// its idle lanes with ballot + prefix popcount.
//
// Semantics reproduced: a hit counts iff tmin <= t <= tmax with
// the ORIGINAL ray range (RayInRange, src/fj_ray.h:29-32); the closest one wins
// with strict '<' (src/fj_bvh_accelerator.cc:183, src/fj_grid_accelerator.cc:263);
// at exactly equal t inside one mesh the larger primitive id wins (the grid's
// LIFO cell lists test it first).  Instances of the group are visited in group
// order (ObjectInstance::RayIntersect, src/fj_object_instance.cc:213-243: the ray
// goes to object space with M^-1 and dir is NOT renormalised, so t is preserved).
#define TRAV_DONE 0xffffffffu
#ifndef FJ_TINY_PRIMS
#define FJ_TINY_PRIMS 4           // primitive sets of at most this many triangles are tested in the instance loop of the phase-scheduled walk (0: walked like any other)
#endif
// tunables (defaults measured on C3; overridable with FJGPU_TRAV_{REFILL,STEPS,GRAB})
#ifdef FJ_PHASE_STATS
// debug builds only: wave-level phase executions / tail lengths (printed by debug_phase_stats)
__device__ unsigned long long g_phase[16];
#endif
#ifdef FJ_WAVE_TIMELINE
// debug builds only (scripts/wave_timeline.py): per wave of a persistent walk, the 100 MHz wall clock at its start, at the first
// iteration after the queue ran dry for it, and at its end, plus its iteration counts -- the SHAPE of a launch's tail.
// The launchers of such a build synchronise after every walk and append the table to the file FJGPU_TIMELINE names.
#define FJ_TL_WAVES 8192
__device__ unsigned long long g_timeline[FJ_TL_WAVES * 6];
struct WaveTimeline {
  unsigned long long t0, tdry, tclaim;
  uint32_t iters, tail, iclaim, worst_ticks, worst_iters, claims;    // the wave's LONGEST claim: its duration and iterations
  __device__ __forceinline__ void begin() { t0 = tclaim = wall_clock64(); tdry = 0; iters = tail = iclaim = worst_ticks = worst_iters = claims = 0; }
  __device__ __forceinline__ void iter(bool dry) { iters++; if (dry) { if (!tdry) tdry = wall_clock64(); tail++; } }
  __device__ __forceinline__ void claim()          // a new slice of the queue (or the end): the previous one is over
  {
    const unsigned long long now = wall_clock64();
    const uint32_t d = (uint32_t) (now - tclaim);
    if (claims && d > worst_ticks) { worst_ticks = d; worst_iters = iters - iclaim; }
    tclaim = now; iclaim = iters; claims++;
  }
  __device__ __forceinline__ void end()
  {
    claim();
    const uint32_t w = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (__lane_id() == 0 && w < FJ_TL_WAVES) {
      const unsigned long long t1 = wall_clock64();
      g_timeline[6 * w + 0] = t0; g_timeline[6 * w + 1] = tdry ? tdry : t1; g_timeline[6 * w + 2] = t1;
      g_timeline[6 * w + 3] = ((unsigned long long) iters << 32) | tail;
      g_timeline[6 * w + 4] = ((unsigned long long) worst_ticks << 32) | worst_iters;
      g_timeline[6 * w + 5] = claims;
    }
  }
};
// ... and (-DFJ_RAY_HIST: two atomics per ray, the launch's times mean nothing then) a histogram of the rays' costs: inner steps per
// ray in power-of-two bins (rays, and the steps they took)
__device__ unsigned long long g_rayhist[64];
#ifdef FJ_RAY_HIST
#define FJ_TL_RAY(steps) do { const uint32_t s_ = (steps); const int b_ = 31 - __clz((int) (s_ | 1u)); atomicAdd(&g_rayhist[b_], 1ull); atomicAdd(&g_rayhist[32 + b_], (unsigned long long) s_); } while (0)
#else
#define FJ_TL_RAY(steps) do { } while (0)
#endif
#define FJ_TL_CLAIM() tl_.claim()
#define FJ_TL_DECL() WaveTimeline tl_; tl_.begin()
#define FJ_TL_ITER(dry) tl_.iter(dry)
#define FJ_TL_END() tl_.end()
#else
#define FJ_TL_RAY(steps) do { } while (0)
#define FJ_TL_CLAIM() do { } while (0)
#define FJ_TL_DECL() do { } while (0)
#define FJ_TL_ITER(dry) do { } while (0)
#define FJ_TL_END() do { } while (0)
#endif
struct TravTune { uint32_t refill, steps, grab, leaf_wait, min_inner, anyhit_steps, refill_curves, steps_curves, steps_phased, min_inner_phased, refill_canyhit, steps_canyhit, min_inner_canyhit, leaf_wait_canyhit, refill_flat, steps_flat, min_inner_flat, leaf_bias8, leaf_bias8_flat, leaf_bias8_phased, leaf_bias8_canyhit, exact_min, filter_off; };
#define TRAV_REFILL tune.refill   // refill when at least this many lanes are idle
#define TRAV_STEPS (int) tune.steps   // inner-node steps between leaf / refill checks
#define TRAV_GRAB tune.grab       // queue entries a wave claims per global atomic

// Claim size.  A wave takes `grab` consecutive rays per atomic; with few rays per launch (a
// rank of an 8-GPU job, a deep recursion level) whole claims decide the load balance -- 4
// claims per wave leave the slowest wave ~25 % behind -- so the claim shrinks until every
// wave gets at least ~16 of them (never below 16 rays: a wave has 64 lanes to fill).
__device__ __forceinline__ uint32_t adaptive_grab(uint32_t grab, uint32_t n)
{
  const uint32_t waves = gridDim.x * (BLOCK / 64);
  const uint32_t want = n / (waves * 16u);
  return want >= grab ? grab : (want < 16u ? 16u : want);
}

// Claims of a persistent walk, per XCD.  The queue is cut into 8 regions with a head each
// (DCounters.*_xcd_head, one per 128-byte line); the waves of XCD x start in region x -- their L2 then
// holds the nodes that ONE stretch of the queue walks, not those of eight, and a head has an eighth of
// the pullers -- and move on to the next region when theirs is empty.  Everything here is wave-uniform.
#ifndef FJ_XCD_HEADS
#define FJ_XCD_HEADS 1                  // 0: one region
#endif
__device__ __forceinline__ uint32_t xcc_id()      // the XCD this wave runs on (0..7); speed only: any value gives the same result
{
  uint32_t x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 0xfu;
}
struct QueueClaim {
  uint32_t *heads;
  uint32_t n, per, grab, region, left;
  __device__ __forceinline__ void init(uint32_t *heads_, uint32_t n_, uint32_t grab_)
  {
    heads = heads_; n = n_; grab = grab_;
    const uint32_t parts = FJ_XCD_HEADS ? 8u : 1u;
    per = ((n / parts + grab) / grab) * grab;       // a multiple of the claim; parts * per >= n
    region = FJ_XCD_HEADS ? (xcc_id() & 7u) : 0u;
    left = parts;
  }
  // the next slice [*next, *range_end) of the queue; false: the queue is empty
  __device__ __forceinline__ bool claim(unsigned lane, uint32_t *next, uint32_t *range_end)
  {
    while (left) {
      const uint32_t lo = region * per;
      if (lo < n) {
        const uint32_t hi = (n - lo < per) ? n : lo + per;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&heads[region * 32u], grab);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base < hi - lo) { *next = lo + base; *range_end = (hi - *next < grab) ? hi : *next + grab; return true; }
      }
      left--;
      region = (region + 1u) & 7u;
    }
    return false;
  }
};

__device__ __forceinline__ DRay implicit_camera_ray(const DScene &S, uint32_t i);      // fjgpu_dev_shade.h

struct RayIn { V3 o, d; double tmin, tmax, time; int group; bool anyhit; };

// Per-lane traversal stack: the first FJ_STACK_LDS entries in LDS ([depth][thread], lane
// consecutive, conflict free), deeper ones -- the builder reports the worst case of the
// scene's trees -- in a global overflow area ([depth][global thread]).
// (lds / ovf carry their address spaces: through generic pointers the pop `sp < lds_n ? lds[..] : ovf[..]` became a choice between two
// addresses and ONE FLAT load -- issued down the vector-memory path even when it reads LDS, and waited for with both counters -- on the
// critical path of every step that ends without a child)
typedef __attribute__((address_space(3))) uint32_t trav_lds_u32;
struct TravStack {
  trav_lds_u32 *lds;    // s_stack + threadIdx.x
  FJ_GLOBAL uint32_t *ovf;   // overflow base + global thread id (null when no tree needs it)
  uint32_t ovf_stride;  // threads in the grid
  double *rayspace;     // curve scenes: s_rayspace + threadIdx.x (RaySpace), else null
  int lds_n;            // entries kept in LDS (FJ_STACK_LDS, or FJ_STACK_LDS_CURVES)
  __device__ __forceinline__ void push(int &sp, uint32_t v) const
  {
    if (sp < lds_n) lds[sp * BLOCK] = v;
    else ovf[(size_t) (sp - lds_n) * ovf_stride] = v;
    sp++;
  }
  __device__ __forceinline__ uint32_t pop(int &sp) const
  {
    --sp;
    uint32_t v;
    if (sp < lds_n) v = lds[sp * BLOCK];
    else v = ovf[(size_t) (sp - lds_n) * ovf_stride];
    return v;
  }
};
__device__ __forceinline__ TravStack make_stack(uint32_t *s_stack, uint32_t *ovf, double *s_rayspace = nullptr, int lds_entries = 0)
{
  TravStack st;
  st.lds = (trav_lds_u32 *) (s_stack + threadIdx.x);
  st.lds_n = lds_entries ? lds_entries : (s_rayspace ? FJ_STACK_LDS_CURVES : FJ_STACK_LDS);
  st.rayspace = s_rayspace ? s_rayspace + threadIdx.x : nullptr;
  st.ovf_stride = gridDim.x * BLOCK;
  st.ovf = ovf ? (FJ_GLOBAL uint32_t *) (ovf + (size_t) blockIdx.x * BLOCK + threadIdx.x) : nullptr;
  return st;
}

// ---- the instance level in LDS (DInstEntry, fjgpu_types.h): [NODES DTNodes][INSTS DInstEntry][GROUPS DGroup], verbatim copies
template <int NODES_, int INSTS_, int GROUPS_> struct InstLdsT {
  static constexpr int NODES = NODES_, INSTS = INSTS_, GROUPS = GROUPS_;
  static constexpr int ENTRIES_AT = NODES * 7, GROUPS_AT = ENTRIES_AT + INSTS * FJ_INST_LDS_ENTRY_WORDS;      // in 8-byte words
  static constexpr int WORDS = GROUPS_AT + GROUPS * 8;
  static bool fits(const DScene &S)        // (host: the launchers pick the kInstLds instantiations by this)
  {
    return S.inst_lds && S.n_group_nodes <= NODES && S.n_instances <= INSTS && S.n_groups <= GROUPS;
  }
  static __device__ __forceinline__ void fill(const DScene &S, double *s_inst)      // every thread of the block
  {
    const unsigned long long *src_n = (const unsigned long long *) S.group_nodes, *src_e = (const unsigned long long *) S.inst_entries;
    const unsigned long long *src_g = (const unsigned long long *) S.groups;
    unsigned long long *dst = (unsigned long long *) s_inst;
    for (uint32_t w = threadIdx.x; w < (uint32_t) S.n_group_nodes * 7u; w += BLOCK) dst[w] = src_n[w];
    for (uint32_t w = threadIdx.x; w < (uint32_t) S.n_instances * FJ_INST_LDS_ENTRY_WORDS; w += BLOCK) dst[ENTRIES_AT + w] = src_e[w];
    for (uint32_t w = threadIdx.x; w < (uint32_t) S.n_groups * 8u; w += BLOCK) dst[GROUPS_AT + w] = src_g[w];
    __syncthreads();
  }
};
typedef InstLdsT<FJ_INST_LDS_NODES, FJ_INST_LDS_INSTS, FJ_INST_LDS_GROUPS> InstLds;                    // the phased walk: 4 blocks per CU
typedef InstLdsT<FJ_INST_LDS_NODES_BIG, FJ_INST_LDS_INSTS_BIG, FJ_INST_LDS_GROUPS_BIG> InstLdsBig;     // 3 blocks per CU: k_trace_closest, k_shadow_trace
typedef InstLdsT<FJ_INST_LDS_NODES_CURVES, FJ_INST_LDS_INSTS_CURVES, FJ_INST_LDS_GROUPS_CURVES> InstLdsCurves;   // ... of scenes with curve sets
template <bool kCurves> struct InstLdsOf { typedef InstLdsBig T; };
template <> struct InstLdsOf<true> { typedef InstLdsCurves T; };

// kInstLds: the scene's instance level sits in LDS at s_inst (filled by the kernel: InstLds); otherwise the same records are read
// from DScene.group_nodes / inst_entries / groups.  (A run-time choice through generic pointers was measured: the flat loads
// cost the fallback 8 % -- C2's closest-hit walk 20.1 -> 21.9 ms.)
// kAnyOnly: every ray of the launch is an any-hit ray and an occluded one adds nothing (shadow rays of scenes whose occluders are all
// opaque): no hit record is kept at all -- a hit retires the ray on the spot, a ray that runs out of instances reaches the light.
template <bool kCurves, bool kCount, bool kMotion, bool kInstLds, bool kAnyOnly, class Policy>
__device__ void traverse_persistent(const DScene &S, Policy &pol, TravTune tune, uint32_t n, uint32_t *head, TravStack stk, LocalCounters *lc, const double *s_inst)
{
  const DTNode *gnodes = kInstLds ? (const DTNode *) s_inst : S.group_nodes;
  typedef typename InstLdsOf<kCurves>::T IL;
  const DInstEntry *gents = kInstLds ? (const DInstEntry *) (s_inst + IL::ENTRIES_AT) : S.inst_entries;
  const DGroup *ggroups = kInstLds ? (const DGroup *) (s_inst + IL::GROUPS_AT) : S.groups;
  const unsigned lane = __lane_id();
  bool head_live = true;                   // wave-uniform: the global head still has entries
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  // work distribution: waves claim TRAV_GRAB consecutive queue entries at a time
  // from a global head (one atomic per 1024 rays).  Static per-wave slices were
  // measurably worse: neighbouring rays have correlated cost, so whole slices
  // end up cheap or expensive and the slowest wave sets the kernel time.
  uint32_t next = 0, range_end = 0;        // wave-uniform
  tune.grab = adaptive_grab(tune.grab, n);
  QueueClaim qc;
  qc.init(head, n, tune.grab);
  if (kCurves) { tune.refill = tune.refill_curves; tune.steps = tune.steps_curves; }
  bool have = false;
  uint32_t idx = 0;
  V3 o = mk(0, 0, 0), oo = o, od = o, d = o, winv = o;
  Slab32 s32 = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};        // conservative f32 slab constants of (ray, instance)
  double tmin = kAnyOnly ? .0001 : 0, tmax = 0, rtime = 0;    // (kAnyOnly = shadow rays: tmin is the constant ShadowPolicy hands out)
  Best best;
  best.t = DBL_MAX; best.u = best.v = 0; best.inst = -1; best.prim = -1;
  int ti = 0, tend = 0, ii = -1;           // cursor in the group's threaded instance BVH
  int group = 0;
  bool single = false;                     // the group has one instance (its padded box is the test)
  const double *gsb = nullptr;
  bool anyhit = kAnyOnly, dead_ray = false, plain = false;
  bool deep = false;                       // holding a curve whose ribbon test awaits its second stage
  const DPrimSet *P = nullptr;
  const DNode *nodes = nullptr;            // P->nodes, kept in registers: re-reading it through P put a
                                           // dependent load in front of every node fetch
  uint32_t cur = TRAV_DONE;
  uint32_t last_curve = 0xffffffffu;      // curve tested last for this (ray, instance)
  uint32_t pend = 0xffffffffu;            // BLAS slot of a curve whose second-stage test is deferred (the lane walks on)
                                           // (a SECOND deferred slot, so that a lane waits only at its third candidate, was measured on C5: shadow
                                           // walk 1692 -> 1695 ms with 3 % more nodes visited -- lanes waiting at a candidate are not what it lacks)
                                           // (POSTPONED leaves for any-hit rays as in the lean any-hit walk -- a lane at a leaf with a non-empty stack
                                           // sets the leaf aside and stays in the inner steps -- parity green, 1515 -> 1570 ms: every piece of added
                                           // state has cost this 168-VGPR kernel more in spills than it saved in iterations)
  int sp = 0;
#ifdef FJ_PHASE_STATS
  unsigned long long it_all = 0, it_tail = 0;   // wave iterations; ... after the queue ran dry for this wave
  // wave-level clock ticks per phase (s_memtime): refill+entry, inner steps, leaf phase, second stage of the ribbon test
  unsigned long long cyc[4] = {0, 0, 0, 0}, c_prev = __builtin_readcyclecounter();
#define FJ_CYC(k) do { const unsigned long long c_now = __builtin_readcyclecounter(); cyc[k] += c_now - c_prev; c_prev = c_now; } while (0)
#else
#define FJ_CYC(k) do { } while (0)
#endif

  FJ_TL_DECL();
  for (;;) {
    FJ_TL_ITER(!head_live && next >= range_end);
#ifdef FJ_PHASE_STATS
    it_all++;
    if (!head_live && next >= range_end) it_tail++;
#endif
    // ---- refill idle lanes from the wave's slice
    const unsigned long long idle = __ballot(!have);
    if (next >= range_end && head_live && (idle == ~0ull || (unsigned) __popcll(idle) >= TRAV_REFILL))
      { head_live = qc.claim(lane, &next, &range_end); FJ_TL_CLAIM(); }
    if (idle == ~0ull || ((unsigned) __popcll(idle) >= TRAV_REFILL && next < range_end)) {
      if (!have) {
        const uint32_t my = next + (uint32_t) __popcll(idle & lt_mask);
        if (my < range_end) {
          RayIn r;
          r.o = r.d = mk(0, 0, 1); r.tmin = r.tmax = r.time = 0; r.group = 0; r.anyhit = false;
          have = pol.fetch(my, &r);
          idx = my;
          tmin = kAnyOnly ? .0001 : r.tmin; tmax = r.tmax; anyhit = kAnyOnly ? true : r.anyhit;
          if (kMotion) rtime = r.time;
          group = r.group;
          const DGroup *G = &ggroups[r.group];
          ti = G->first; tend = ti + G->count;
          best.t = DBL_MAX; best.u = best.v = 0; best.inst = -1; best.prim = -1;
          cur = TRAV_DONE; sp = 0;
          if (!kCurves) {
            o = r.o; d = r.d;
            single = G->n_instances == 1;
            gsb = G->sbounds;
            dead_ray = has_negative_zero(d);   // every box test of the reference fails (see above)
            winv = mk(filter_rcp(d.x), filter_rcp(d.y), filter_rcp(d.z));
            plain = plain_dir(d);
          }
        }
      }
      next += (uint32_t) __popcll(idle);
      if (__ballot(have) == 0ull) {
        if (next >= range_end && !head_live) {
#ifdef FJ_PHASE_STATS
          if (lane == 0) { atomicMax(&g_phase[12], it_tail); atomicAdd(&g_phase[13], it_tail); atomicAdd(&g_phase[14], it_all); atomicAdd(&g_phase[15], 1ull);
            atomicAdd(&g_phase[1], cyc[0]); atomicAdd(&g_phase[3], cyc[1]); atomicAdd(&g_phase[5], cyc[2]); atomicAdd(&g_phase[7], cyc[3]); }
#endif
          FJ_TL_END();
          break;
        }
        continue;               // only padding slots were fetched / slice exhausted: claim more
      }
    }

    // (Round 3 measured the phases of this walk on C5 in clock ticks -- debug build: instance entry 24 %, inner steps 38 % with 24
    // of 64 lanes, leaf phase 18 %, second stage of the ribbon test 20 % -- and tried GATING them like the lean any-hit walk does,
    // a block running only with at least 8..64 lanes in it or when it is the fullest: 1733 -> 1755 .. 1864 ms, monotonically worse
    // with the thresholds.  Here lanes waiting for "their" phase cost more than running every phase with whoever is there.)
    // ---- lanes between instances: enter the next instance or retire the ray
    if (have && cur == TRAV_DONE && (!kCurves || pend == 0xffffffffu)) {
      bool found = false;
      uint32_t root = TRAV_DONE;
      if (kCurves) {
        // the curve instantiation lives on registers (the ribbon test): what only this block needs
        // of the world-space ray is read again from the queue instead of being carried through it
        RayIn r;
        r.o = r.d = mk(0, 0, 1); r.tmin = r.tmax = r.time = 0; r.group = 0; r.anyhit = false;
        pol.fetch(idx, &r);
        o = r.o; d = r.d;
        single = ggroups[group].n_instances == 1;
        gsb = ggroups[group].sbounds;
        dead_ray = has_negative_zero(d);
        winv = mk(filter_rcp(d.x), filter_rcp(d.y), filter_rcp(d.z));
        plain = plain_dir(d);
      }
      while (!dead_ray && ti < tend) {
        const DTNode *tn_ = &gnodes[ti];
        if (tn_->inst < 0) {             // inner node of the instance BVH: conservative box, skip link
          double tq;
          ti = slab(tn_->box, tn_->box + 3, o, winv, tmin, tmax, &tq) ? ti + 1 : tn_->skip;
          continue;
        }
        ii = tn_->inst;
        ti++;
        const DInstEntry *I = &gents[ii];
        if (kCount) lc->insts++;
        double tn;
        const double tfar = anyhit ? tmax : fmin(tmax, best.t);
        // a tight box of the geometry first (pure culling, DInstEntry.tbounds): nothing of this instance can be hit nearer than tfar
        if (!slab(I->tbounds, I->tbounds + 3, o, winv, tmin, tfar, &tn)) continue;
        // the reference's own (possibly non-enclosing) instance box, full ray range
        if (!box_ray_ref_fast(single ? gsb : tn_->box, o, d, winv, plain, tmin, tmax)) continue;
        if (kMotion && I->xform >= 0) {
          // ObjectInstance::RayIntersect evaluates a time-sampled transform at the ray's time
          double tm[12], tmi[12];
          xform_at(&S.xforms[I->xform], rtime, tm, tmi);
          oo = xpoint(tmi, o);
          od = xvector(tmi, d);
        } else {
          oo = xpoint(I->Minv, o);
          od = xvector(I->Minv, d);
        }
        if (has_negative_zero(od)) continue;
        const V3 inv = mk(filter_rcp(od.x), filter_rcp(od.y), filter_rcp(od.z));
        P = &S.primsets[I->primset];
        nodes = (const DNode *) I->nodes;
        if (I->pn_prims == 0) continue;
        if (!slab(I->pbounds, I->pbounds + 3, oo, inv, tmin, tfar, &tn)) continue;
        if (FJ_CLOSEST_QNODES && (!kCurves || (FJ_CURVE_QNODES && !kMotion))) s32 = slab32q_setup(oo, inv, I->qorigin, I->qcell);
        else s32 = slab32_setup(oo, inv, I->pbounds);
        root = I->proot;
        found = true;
        break;
      }
      if (found) {
        cur = root; sp = 0; last_curve = 0xffffffffu;
      }
      else { pol.finish(idx, best); have = false; }
    }

    FJ_CYC(0);
    // ---- inner nodes: a few steps for every lane that holds one
    for (int step = 0; step < TRAV_STEPS; step++) {
      const bool inner = have && !(cur & FJ_LEAF_FLAG);
      if (__ballot(inner) == 0ull) break;
        if (inner) {
        if (kCount) lc->nodes++;
        const double tf2 = anyhit ? tmax : fmin(tmax, best.t);
        const float tmin32 = f32_below(tmin), tmax32 = f32_above(tf2);
        float t0, t1, t2, t3;
        bool h0, h1, h2, h3;
        fj_v4u e;
        if (FJ_CLOSEST_QNODES && (!kCurves || (FJ_CURVE_QNODES && !kMotion))) {
          // 64-byte quantised node (DNodeQ): four 16-byte loads, sign-aware packed slab tests (slab32q_test)
          const FJ_GLOBAL fj_v4u *nq = (const FJ_GLOBAL fj_v4u *) ((const DNodeQ *) nodes + cur);
          const fj_v4u w0 = nq[0], w1 = nq[1], w2 = nq[2];
          e = nq[3];
          const uint32_t shx = slab32_shift(s32.x.i), shy = slab32_shift(s32.y.i), shz = slab32_shift(s32.z.i);
          h0 = slab32q_test(w0.x, w0.y, w0.z, s32, shx, shy, shz, tmin32, tmax32, &t0);
          h1 = slab32q_test(w0.w, w1.x, w1.y, s32, shx, shy, shz, tmin32, tmax32, &t1);
          h2 = slab32q_test(w1.z, w1.w, w2.x, s32, shx, shy, shz, tmin32, tmax32, &t2) && e.z != FJ_NO_CHILD;
          h3 = slab32q_test(w2.y, w2.z, w2.w, s32, shx, shy, shz, tmin32, tmax32, &t3) && e.w != FJ_NO_CHILD;
        } else {
        const FJ_GLOBAL fj_v4f *nd = (const FJ_GLOBAL fj_v4f *) (nodes + cur);
        // 128-byte node: seven 16-byte loads (4 child boxes as (min, max) pairs + 4 child refs)
        const fj_v4f q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3], q4 = nd[4], q5 = nd[5];
        e = ((const FJ_GLOBAL fj_v4u *) nd)[6];
        h0 = slab32_test(q0.xy, q0.zw, q1.xy, s32, tmin32, tmax32, &t0);                          // slot 0 always exists
        h1 = slab32_test(q1.zw, q2.xy, q2.zw, s32, tmin32, tmax32, &t1);                          // slot 1 always exists
        h2 = e.z != FJ_NO_CHILD && slab32_test(q3.xy, q3.zw, q4.xy, s32, tmin32, tmax32, &t2);
        h3 = e.w != FJ_NO_CHILD && slab32_test(q4.zw, q5.xy, q5.zw, s32, tmin32, tmax32, &t3);
        }
        // near-to-far order is a heuristic only: f32 keys, misses sort last.  (Any-hit rays do not need an order, but the network of
        // five compare-exchanges is cheaper than what replaces it: children in slot order visit 7 % more nodes, the nearest child first
        // and the rest in slot order costs four data-dependent pushes instead of three nested ones -- C5 shadow walk 1408 -> 1488 ms.)
        float k0 = h0 ? t0 : INFINITY, k1 = h1 ? t1 : INFINITY;
        float k2 = h2 ? t2 : INFINITY, k3 = h3 ? t3 : INFINITY;
        uint32_t r0 = e.x, r1 = e.y, r2 = e.z, r3 = e.w;
#define FJ_CSWAP(ka, ra, kb, rb) { const bool sw = kb < ka; const float tk = sw ? ka : kb; const uint32_t tr = sw ? ra : rb; ka = sw ? kb : ka; ra = sw ? rb : ra; kb = tk; rb = tr; }
        FJ_CSWAP(k0, r0, k1, r1) FJ_CSWAP(k2, r2, k3, r3) FJ_CSWAP(k0, r0, k2, r2) FJ_CSWAP(k1, r1, k3, r3) FJ_CSWAP(k1, r1, k2, r2)
#undef FJ_CSWAP
        const int nh = (int) h0 + (int) h1 + (int) h2 + (int) h3;
        if (nh == 0) cur = (sp == 0) ? TRAV_DONE : stk.pop(sp);
        else {
          cur = r0;
          if (nh > 3) stk.push(sp, r3);
          if (nh > 2) stk.push(sp, r2);
          if (nh > 1) stk.push(sp, r1);
        }
      }
    }

    FJ_CYC(1);
    // ---- leaves: FP64 Moller-Trumbore on the pre-gathered triangles; curve leaves (always a
    // single curve, fjgpu_curve_build.cc) only take the first stage of the ribbon test here --
    // does the curve's ray-space box reach the ray? -- and wait for the second stage below
    if (have && !deep && (cur & FJ_LEAF_FLAG) && cur != TRAV_DONE) {
      const uint32_t first = (cur & 0x7fffffffu) >> 3;
      const uint32_t cnt = (cur & 7u) + 1;
      bool stop = false;
      if (kCurves && P->type == FJ_PRIMSET_CURVE) {
        // BLAS entries are sub-segments of curves: the ribbon test of a curve runs once, not
        // once per piece entered (same ray, same instance: same result)
        const size_t sl = first;
        const uint32_t cid = FJ_G(uint32_t, P->prim_ids)[sl];
        if (cid != last_curve && (!P->curve_capsule || capsule_may_hit(FJ_G(float, P->curve_capsule) + sl * 8, oo, od))) {
          last_curve = cid;
          if (kCount) lc->prims++;
          const FJ_GLOBAL double *cvel = (kMotion && P->curve_vel) ? FJ_G(double, P->curve_vel) + sl * 12 : nullptr;
          // (a slot that passes its capsule goes to the second stage at once: that stage starts with the whole curve's ray-space box)
          deep = true;
          // the second stage is deferred: the lane remembers the curve and walks on (its result
          // only shortens the ray or ends it -- the walk stays correct without it); a lane that
          // already carries a deferred curve waits here instead
          if (deep && pend == 0xffffffffu) { pend = (uint32_t) sl; deep = false; }
        }
      } else {
        for (uint32_t k = 0; k < cnt; k++) {
          double t, u = 0, v = 0;
          if (kCount) lc->prims++;
          V3 v0, v1, v2;
          load_tri(P->tri_verts, P->tri_verts32, first + k, &v0, &v1, &v2);
          if (kMotion && P->tri_vel) {       // Mesh::ray_intersect: P + time * velocity (src/fj_mesh.cc:252-259)
            const FJ_GLOBAL double *w = FJ_G(double, P->tri_vel) + (size_t) (first + k) * 9;
            v0 = v0 + rtime * ld3(w); v1 = v1 + rtime * ld3(w + 3); v2 = v2 + rtime * ld3(w + 6);
          }
          if (!tri_ray(v0, v1, v2, oo, od, &t, &u, &v)) continue;
          if (!(tmin <= t && t <= tmax)) continue;
          if (kAnyOnly) { stop = true; break; }
          const int pid = (int) FJ_G(uint32_t, P->prim_ids)[first + k];
          if (t < best.t || (t == best.t && best.inst == ii && pid > best.prim)) {
            best.t = t; best.u = u; best.v = v; best.inst = ii; best.prim = pid;
            if (anyhit) { stop = true; break; }
          }
        }
      }
      if (stop) { if (!kAnyOnly) pol.finish(idx, best); have = false; cur = TRAV_DONE; }
      else if (!deep) cur = (sp == 0) ? TRAV_DONE : stk.pop(sp);
    }

    FJ_CYC(2);
    // ---- second stage of the ribbon test (the recursive subdivision, long and divergent:
    // PMC on C5 showed 8.8 of 64 lanes active per VALU instruction when every lane ran it as
    // soon as it reached a curve): it waits until enough lanes need it, or nobody can walk on
    if (kCurves) {
      const bool carrying = have && pend != 0xffffffffu;
      const unsigned long long pendm = __ballot(carrying);
      if (pendm) {
        // lanes that can make progress without a second-stage result: at an inner node or a fresh leaf
        const unsigned long long busym = __ballot(have && !deep && cur != TRAV_DONE);
        if ((unsigned) __popcll(pendm) >= tune.leaf_wait || busym == 0ull) {
          if (lane == (unsigned) __ffsll((long long) pendm) - 1u) { FJ_CURVE_STAT(2, 1); FJ_CURVE_STAT(3, __popcll(pendm)); }   // second-stage execs / lanes
          double t = 0, u = 0;
          bool hitc = false;
          if (carrying) {
            bool stop = false;
            const size_t sl = pend;
            pend = 0xffffffffu;
            const FJ_GLOBAL double *cvel = (kMotion && P->curve_vel) ? FJ_G(double, P->curve_vel) + sl * 12 : nullptr;
            // hit record of a curve: u = curve parameter v_hit, v = BLAS slot (attribute fetch)
            hitc = curve_ray(FJ_G(double, P->curve_cp) + sl * 12, cvel, rtime, FJ_G(double, P->curve_width)[2 * sl], FJ_G(double, P->curve_width)[2 * sl + 1],
                          (int) FJ_G(int8_t, P->curve_depth)[sl], RaySpace{stk.rayspace, oo, od}, &t, &u);
            if (hitc &&
                (cvel ? curve_listed_in_cell_of_moving(P, FJ_G(double, P->curve_cp) + sl * 12, cvel, oo + t * od)
                      : curve_listed_in_cell_of(P, FJ_G(double, P->curve_cp) + sl * 12, oo + t * od)) &&
                (tmin <= t && t <= tmax)) {
              FJ_CURVE_STAT(4, 1);          // second-stage tests that hit
              if (kAnyOnly) stop = true;
              else {
                const int pid = (int) FJ_G(uint32_t, P->prim_ids)[sl];
                if (t < best.t || (t == best.t && best.inst == ii && pid > best.prim)) {
                  best.t = t; best.u = u; best.v = (double) sl; best.inst = ii; best.prim = pid;
                  stop = anyhit;
                }
              }
            }
            if (stop) { if (!kAnyOnly) pol.finish(idx, best); have = false; deep = false; cur = TRAV_DONE; }
            else if (deep) {
              // the curve this lane was waiting at becomes the deferred one; walk on
              pend = (cur & 0x7fffffffu) >> 3;
              deep = false;
              cur = (sp == 0) ? TRAV_DONE : stk.pop(sp);
            }
          }
        }
      }
    }
    FJ_CYC(3);
  }
}

// ---------------------------------------------------- phase-scheduled walk (meshes)
// The same walk for scenes without curve sets, scheduled like the lean any-hit walk
// (fjgpu_dev_anyhit.h): a lane is in TURNOVER (no BLAS walk in progress: idle, finished, or between
// the instances of its group), at an INNER node or at a LEAF, and every iteration the wave runs ONE
// phase -- the one most lanes wait for; turnover once tune.refill lanes collected there or nothing
// else can run -- instead of every phase in turn with whoever happens to be there.  The leaf phase
// tests one triangle per lane and execution.  Same tests, same order per ray (instances in the
// group's order, children near to far, the leaf's triangles in slot order), same tie rules.
// kInstLds: the scene's instance level (DTNodes, then one DInstEntry per instance) sits in LDS at s_inst (DScene.inst_lds;
// filled by the kernel); otherwise the same records are read from DScene.group_nodes / inst_entries.
template <bool kCount, bool kMotion, bool kInstLds, class Policy>
__device__ void traverse_phased(const DScene &S, Policy &pol, TravTune tune, uint32_t n, uint32_t *head, TravStack stk, LocalCounters *lc, const double *s_inst)
{
  const DTNode *gnodes = kInstLds ? (const DTNode *) s_inst : S.group_nodes;
  const DInstEntry *gents = kInstLds ? (const DInstEntry *) (s_inst + InstLds::ENTRIES_AT) : S.inst_entries;
  const DGroup *ggroups = kInstLds ? (const DGroup *) (s_inst + InstLds::GROUPS_AT) : S.groups;
  const unsigned lane = __lane_id();
  bool head_live = true;
  uint32_t next = 0, range_end = 0;        // wave-uniform: the wave's claimed slice of the queue
  tune.grab = adaptive_grab(tune.grab, n);
  QueueClaim qc;
  qc.init(head, n, tune.grab);
  bool have = false;
  uint32_t idx = 0;
  V3 o = mk(0, 0, 0), oo = o, od = o, d = o;
  Slab32 s32 = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  double tmin = 0, tmax = 0, rtime = 0;
  Best best;
  best.t = DBL_MAX; best.u = best.v = 0; best.inst = -1; best.prim = -1;
  int ti = 0, tend = 0, ii = -1;           // cursor in the group's threaded instance BVH
  int group = 0;
  bool anyhit = false;
  const DNode *nodes = nullptr;
  uint32_t cur = TRAV_DONE;
  int sp = 0;
#ifdef FJ_PHASE_STATS
  // wave-level clock ticks, executions and lanes per phase; the instance loop's trips summed over lanes / as the wave's maximum
  unsigned long long pcyc[3] = {0, 0, 0}, pex[3] = {0, 0, 0}, pln[3] = {0, 0, 0}, trips_lane = 0, trips_wave = 0, pc_prev = __builtin_readcyclecounter();
#define FJ_PCYC(k, mask) do { const unsigned long long c_now = __builtin_readcyclecounter(); pcyc[k] += c_now - pc_prev; pc_prev = c_now; pex[k]++; pln[k] += __popcll(mask); } while (0)
#else
#define FJ_PCYC(k, mask) do { } while (0)
#endif

  for (;;) {
    const bool fin = cur == TRAV_DONE;
    const bool at_leaf = !fin && (cur & FJ_LEAF_FLAG);
    const bool at_inner = !fin && !at_leaf;
    const unsigned n_leaf = (unsigned) __popcll(__ballot(at_leaf)), n_inner = (unsigned) __popcll(__ballot(at_inner));
    const bool can_fetch = head_live || next < range_end;
    const unsigned long long m_turn = __ballot(fin && (have || can_fetch));

    if ((unsigned) __popcll(m_turn) >= TRAV_REFILL || (n_inner == 0 && n_leaf == 0)) {
      if (m_turn == 0ull) {
#ifdef FJ_PHASE_STATS
        if (lane == 0) {
          for (int k = 0; k < 3; k++) { atomicAdd(&g_phase[1 + 2 * k], pcyc[k]); atomicAdd(&g_phase[2 + 2 * k], pln[k]); atomicAdd(&g_phase[7 + k], pex[k]); }
          atomicAdd(&g_phase[10], trips_lane); atomicAdd(&g_phase[11], trips_wave);
        }
#endif
        break;
      }
      // ---- turnover: fetch, enter the next instance, or retire
      if (next >= range_end && head_live) head_live = qc.claim(lane, &next, &range_end);
      const bool fetch = fin && !have;
      const unsigned long long m_fetch = __ballot(fetch);
      if (fetch) {
        const uint32_t my = next + __builtin_amdgcn_mbcnt_hi((uint32_t) (m_fetch >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m_fetch, 0u));
        if (my < range_end) {
          RayIn r;
          r.o = r.d = mk(0, 0, 1); r.tmin = r.tmax = r.time = 0; r.group = 0; r.anyhit = false;
          have = pol.fetch(my, &r);
          idx = my;
          o = r.o; d = r.d; tmin = r.tmin; tmax = r.tmax; anyhit = r.anyhit;
          if (kMotion) rtime = r.time;
          group = r.group;
          ti = ggroups[group].first; tend = ti + ggroups[group].count;
          best.t = DBL_MAX; best.u = best.v = 0; best.inst = -1; best.prim = -1;
          sp = 0;
        }
      }
      next += (uint32_t) __popcll(m_fetch);
      if (next > range_end) next = range_end;
#ifdef FJ_PHASE_STATS
      unsigned my_trips = 0;
#endif
      if (fin && have) {
        bool found = false;
        uint32_t root = TRAV_DONE;
        const bool dead_ray = has_negative_zero(d);   // every box test of the reference fails (BoxRayIntersect's -0.0 quirk)
        const V3 winv = mk(filter_rcp(d.x), filter_rcp(d.y), filter_rcp(d.z));
        const bool plain = plain_dir(d);
        const bool single = ggroups[group].n_instances == 1;
        const double *gsb = ggroups[group].sbounds;
        while (!dead_ray && ti < tend) {
#ifdef FJ_PHASE_STATS
          my_trips++;
#endif
          // inner nodes of the instance BVH (conservative box, skip link) are passed in a loop of their own: a trip of the
          // outer loop is one INSTANCE for every lane
          const DTNode *tn_ = &gnodes[ti];
          int inst_ = tn_->inst;
          while (inst_ < 0) {
            double tq;
            ti = slab(tn_->box, tn_->box + 3, o, winv, tmin, tmax, &tq) ? ti + 1 : tn_->skip;
            if (ti >= tend) break;
            tn_ = &gnodes[ti];
            inst_ = tn_->inst;
          }
          if (inst_ < 0) break;            // (the list ended on a missed inner node)
          ii = inst_;
          ti++;
          const DInstEntry *I = &gents[ii];
          if (kCount) lc->insts++;
          double tn;
          const double tfar = anyhit ? tmax : fmin(tmax, best.t);
          // a tight box of the geometry first (pure culling, DInstEntry.tbounds): nothing of this instance can be hit nearer than tfar
          if (!slab(I->tbounds, I->tbounds + 3, o, winv, tmin, tfar, &tn)) continue;
          // the reference's own (possibly non-enclosing) instance box, full ray range
          if (!box_ray_ref_fast(single ? gsb : tn_->box, o, d, winv, plain, tmin, tmax)) continue;
          if (kMotion && I->xform >= 0) {
            double tm[12], tmi[12];
            xform_at(&S.xforms[I->xform], rtime, tm, tmi);
            oo = xpoint(tmi, o);
            od = xvector(tmi, d);
          } else {
            oo = xpoint(I->Minv, o);
            od = xvector(I->Minv, d);
          }
          if (has_negative_zero(od)) continue;
          if (I->pn_prims == 0) continue;
          if (FJ_TINY_PRIMS > 0 && !kMotion && I->pn_prims <= FJ_TINY_PRIMS) {
            // a handful of triangles (the walls of a box): tested HERE, one after the other, instead of being walked -- the walk of such a set is one
            // root node whose child boxes cull nothing the instance boxes have not culled, and costs the slab set-up, an inner step, a leaf
            // execution per triangle and a turnover more.  Same tests in leaf order, same tie rules; node boxes only ever rejected provable misses.
            bool stop = false;
            for (int k = 0; k < I->pn_prims; k++) {
              double t, u = 0, v = 0;
              if (kCount) lc->prims++;
              V3 v0, v1, v2;
              load_tri(I->tri_verts, I->tri_verts32, (uint32_t) k, &v0, &v1, &v2);
              if (tri_ray(v0, v1, v2, oo, od, &t, &u, &v) && tmin <= t && t <= tmax) {
                const int pid = (int) FJ_G(uint32_t, I->prim_ids)[k];
                if (t < best.t || (t == best.t && best.inst == ii && pid > best.prim)) {
                  best.t = t; best.u = u; best.v = v; best.inst = ii; best.prim = pid;
                  if (anyhit) { stop = true; break; }
                }
              }
            }
            if (stop) break;               // (an any-hit ray ends here: retired below with its hit)
            continue;
          }
          const V3 inv = mk(filter_rcp(od.x), filter_rcp(od.y), filter_rcp(od.z));
          nodes = (const DNode *) I->nodes;
          if (!slab(I->pbounds, I->pbounds + 3, oo, inv, tmin, tfar, &tn)) continue;
          if (FJ_CLOSEST_QNODES) s32 = slab32q_setup(oo, inv, I->qorigin, I->qcell);
          else s32 = slab32_setup(oo, inv, I->pbounds);
          root = I->proot;
          found = true;
          break;
        }
        if (found) { cur = root; sp = 0; }
        else { pol.finish(idx, best); have = false; }
      }
#ifdef FJ_PHASE_STATS
      {
        unsigned mx = my_trips, sum = my_trips;     // what the loop cost the wave (its slowest lane) / what the lanes needed
        for (int off = 32; off; off >>= 1) {
          const unsigned other = (unsigned) __shfl_xor((int) mx, off);
          mx = other > mx ? other : mx;
          sum += (unsigned) __shfl_xor((int) sum, off);
        }
        trips_wave += mx;
        trips_lane += sum;
      }
#endif
      FJ_PCYC(0, m_turn);
      continue;
    }

    if (n_inner * tune.leaf_bias8_phased >= n_leaf * 8u) {
      // ---- inner nodes; further steps without a new vote while enough lanes stay at inner nodes
      for (uint32_t step = 0;; step++) {
        const bool in_now = step == 0 ? at_inner : (cur != TRAV_DONE && !(cur & FJ_LEAF_FLAG));
        if (step > 0 && (step >= tune.steps_phased || (unsigned) __popcll(__ballot(in_now)) < tune.min_inner_phased)) break;
        if (in_now) {
          if (kCount) lc->nodes++;
          const double tf2 = anyhit ? tmax : fmin(tmax, best.t);
          const float tmin32 = f32_below(tmin), tmax32 = f32_above(tf2);
          float t0, t1, t2, t3;
          bool h0, h1, h2, h3;
          fj_v4u e;
          if (FJ_CLOSEST_QNODES) {
            const FJ_GLOBAL fj_v4u *nq = (const FJ_GLOBAL fj_v4u *) ((const DNodeQ *) nodes + cur);
            const fj_v4u w0 = nq[0], w1 = nq[1], w2 = nq[2];
            e = nq[3];
            const uint32_t shx = slab32_shift(s32.x.i), shy = slab32_shift(s32.y.i), shz = slab32_shift(s32.z.i);
            h0 = slab32q_test(w0.x, w0.y, w0.z, s32, shx, shy, shz, tmin32, tmax32, &t0);
            h1 = slab32q_test(w0.w, w1.x, w1.y, s32, shx, shy, shz, tmin32, tmax32, &t1);
            h2 = slab32q_test(w1.z, w1.w, w2.x, s32, shx, shy, shz, tmin32, tmax32, &t2) && e.z != FJ_NO_CHILD;
            h3 = slab32q_test(w2.y, w2.z, w2.w, s32, shx, shy, shz, tmin32, tmax32, &t3) && e.w != FJ_NO_CHILD;
          } else {
          const FJ_GLOBAL fj_v4f *nd = (const FJ_GLOBAL fj_v4f *) (nodes + cur);
          const fj_v4f q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3], q4 = nd[4], q5 = nd[5];
          e = ((const FJ_GLOBAL fj_v4u *) nd)[6];
          h0 = slab32_test(q0.xy, q0.zw, q1.xy, s32, tmin32, tmax32, &t0);
          h1 = slab32_test(q1.zw, q2.xy, q2.zw, s32, tmin32, tmax32, &t1);
          h2 = e.z != FJ_NO_CHILD && slab32_test(q3.xy, q3.zw, q4.xy, s32, tmin32, tmax32, &t2);
          h3 = e.w != FJ_NO_CHILD && slab32_test(q4.zw, q5.xy, q5.zw, s32, tmin32, tmax32, &t3);
          }
          // near-to-far order is a heuristic only: f32 keys, misses sort last
          float k0 = h0 ? t0 : INFINITY, k1 = h1 ? t1 : INFINITY;
          float k2 = h2 ? t2 : INFINITY, k3 = h3 ? t3 : INFINITY;
          uint32_t r0 = e.x, r1 = e.y, r2 = e.z, r3 = e.w;
#define FJ_CSWAP(ka, ra, kb, rb) { const bool sw = kb < ka; const float tk = sw ? ka : kb; const uint32_t tr = sw ? ra : rb; ka = sw ? kb : ka; ra = sw ? rb : ra; kb = tk; rb = tr; }
          FJ_CSWAP(k0, r0, k1, r1) FJ_CSWAP(k2, r2, k3, r3) FJ_CSWAP(k0, r0, k2, r2) FJ_CSWAP(k1, r1, k3, r3) FJ_CSWAP(k1, r1, k2, r2)
#undef FJ_CSWAP
          const int nh = (int) h0 + (int) h1 + (int) h2 + (int) h3;
          if (nh == 0) cur = (sp == 0) ? TRAV_DONE : stk.pop(sp);
          else {
            cur = r0;
            if (nh > 3) stk.push(sp, r3);
            if (nh > 2) stk.push(sp, r2);
            if (nh > 1) stk.push(sp, r1);
          }
        }
        FJ_PCYC(1, __ballot(in_now));
      }
    } else {
      // ---- leaves: ONE triangle per lane (FP64 Moller-Trumbore on the pre-gathered vertices)
      if (at_leaf) {
        const uint32_t first = (cur & 0x7fffffffu) >> 3;
        const uint32_t more = cur & 7u;
        bool stop = false;
        double t, u = 0, v = 0;
        if (kCount) lc->prims++;
        V3 v0, v1, v2;
        const DInstEntry *E = &gents[ii];      // (the instance this lane is in)
        load_tri(E->tri_verts, E->tri_verts32, first, &v0, &v1, &v2);
        if (kMotion && E->tri_vel) {       // Mesh::ray_intersect: P + time * velocity (src/fj_mesh.cc:252-259)
          const FJ_GLOBAL double *w = FJ_G(double, E->tri_vel) + (size_t) first * 9;
          v0 = v0 + rtime * ld3(w); v1 = v1 + rtime * ld3(w + 3); v2 = v2 + rtime * ld3(w + 6);
        }
        if (tri_ray(v0, v1, v2, oo, od, &t, &u, &v) && tmin <= t && t <= tmax) {
          const int pid = (int) FJ_G(uint32_t, E->prim_ids)[first];
          if (t < best.t || (t == best.t && best.inst == ii && pid > best.prim)) {
            best.t = t; best.u = u; best.v = v; best.inst = ii; best.prim = pid;
            stop = anyhit;
          }
        }
        if (stop) { pol.finish(idx, best); have = false; cur = TRAV_DONE; }
        else if (more) cur = FJ_LEAF_FLAG | ((first + 1u) << 3) | (more - 1u);
        else cur = (sp == 0) ? TRAV_DONE : stk.pop(sp);
      }
      FJ_PCYC(2, __ballot(at_leaf));
    }
  }
}

// ------------------------------------------------------------------ k_trace
struct ClosestPolicy {
  const DScene *S;
  const DRay *rays;
  const DPath *paths;
  DHit *hits;
  int default_group;
  // launch entry -> ray slot (DScene.ray_perm: the launch walks the queue in sorted order)
  __device__ __forceinline__ uint32_t slot(uint32_t i) const { return S->ray_perm ? S->ray_perm[i] : i; }
  __device__ bool fetch(uint32_t k, RayIn *r) const
  {
    const uint32_t i = slot(k);
    const DRay q = rays ? rays[i] : implicit_camera_ray(*S, i);     // (DScene.cam_uv: level 0 of a static camera)
    r->o = mk(q.o[0], q.o[1], q.o[2]); r->d = mk(q.d[0], q.d[1], q.d[2]);
    // the range is a function of the ray's class (DRay): camera rays (znear, zfar) -- Camera::GetRay, src/fj_camera.cc:105-106 --, SlTrace children
    // tmax 1000 and the tmin their shader passes (.001 or .0001: DPath.cxt bit 7); fjgpu_trace hands ranges over per ray
    if (S->trace_ranges) { r->tmin = S->trace_ranges[2 * (size_t) i]; r->tmax = S->trace_ranges[2 * (size_t) i + 1]; }
    else {
      const uint32_t cx = (rays && paths) ? paths[i].cxt : (uint32_t) CXT_CAMERA_RAY;
      if ((cx & 0x7fu) == CXT_CAMERA_RAY) { r->tmin = S->cam_znear; r->tmax = S->cam_zfar; }
      else { r->tmin = (cx & FJ_CXT_TMIN_1E4) ? .0001 : .001; r->tmax = 1000; }
    }
    r->time = (S->has_motion && paths) ? sample_time(*S, paths[i].flags >> 1) : 0.;   // fjgpu_trace: time 0
    r->group = paths ? paths[i].group : default_group;
    r->anyhit = false;
    return true;
  }
  __device__ void finish(uint32_t k, const Best &b) const
  {
    DHit h;
    h.t = b.t; h.u = b.u; h.v = b.v; h.inst = b.inst; h.prim = b.prim;
    hits[slot(k)] = h;
  }
  __device__ void finish_miss(uint32_t k) const      // (walks that write a ray's record whenever a candidate wins: nothing won)
  {
    DHit h;
    h.t = DBL_MAX; h.u = h.v = 0; h.inst = -1; h.prim = -1;
    hits[slot(k)] = h;
  }
};

#ifndef FJ_CURVE_MINB
#define FJ_CURVE_MINB 3          // curve scenes: 168 VGPRs (with spills) and 52 KB of LDS per block, 3 waves per SIMD
#endif
#ifndef FJ_MOTION_MINB
#define FJ_MOTION_MINB 2         // the general (time-sampled) instantiation
#endif
#ifndef FJ_CLOSEST_MINB
#define FJ_CLOSEST_MINB 3
#endif
#ifndef FJ_SHADOW_MINB
#define FJ_SHADOW_MINB 3          // the general shadow walk of mesh scenes (translucent occluders): 166 VGPRs, 3 waves
#endif
template <bool kCurves, bool kCount, bool kMotion, bool kInstLds = false>
__global__ void __launch_bounds__(BLOCK, kMotion ? FJ_MOTION_MINB : (kCurves ? FJ_CURVE_MINB : FJ_CLOSEST_MINB)) k_trace_closest(DScene S, const DRay *rays, const DPath *paths,
    DHit *hits, uint32_t n, DCounters *cnt, TravTune tune)
{
  if (S.trace_n_dev) { const uint32_t nd_ = *S.trace_n_dev; n = nd_ < n ? nd_ : n; }       // (a speculative launch: fjgpu_api.hip)
  __shared__ uint32_t s_stack[(kCurves ? FJ_STACK_LDS_CURVES : FJ_STACK_LDS) * BLOCK];
  __shared__ double s_rayspace[kCurves ? FJ_RAYSPACE_DOUBLES * BLOCK : 1];
  __shared__ double s_inst[kInstLds ? InstLdsOf<kCurves>::T::WORDS : 1];
  if (kInstLds) InstLdsOf<kCurves>::T::fill(S, s_inst);
  ClosestPolicy pol;
  pol.S = &S; pol.rays = rays; pol.paths = paths; pol.hits = hits; pol.default_group = S.target_group;
  LocalCounters lc = {0, 0, 0};
  traverse_persistent<kCurves, kCount, kMotion, kInstLds, false>(S, pol, tune, n, &cnt->trace_xcd_head[0][0], make_stack(s_stack, S.stack_overflow, kCurves ? s_rayspace : nullptr), &lc, s_inst);
  if (kCount) {
    flush_counters(cnt, lc.nodes, lc.prims, lc.insts, 0, 0);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&cnt->traced, (unsigned long long) n);
  }
}

// The phase-scheduled walk (traverse_phased) as the closest-hit kernel of mesh scenes whose secondary
// rays are INCOHERENT (glass: two children per hit, pathtracing: diffuse bounces).  Measured, closest-hit
// time per frame, this kernel (128 VGPRs, 4 waves) against k_trace_closest<false> (168, 3 waves): C4
// 903 vs 1167 ms, C2 20.2 vs 21.3, C3 (coherent camera and mirror rays into one dense mesh) 31.3 vs 30.4
// -- so the launcher picks by the scene's shaders.
#ifndef FJ_PHASED_MINB
#define FJ_PHASED_MINB 4
#endif
template <bool kCount, bool kInstLds>
__global__ void __launch_bounds__(BLOCK, FJ_PHASED_MINB) k_trace_closest_phased(DScene S, const DRay *rays, const DPath *paths,
    DHit *hits, uint32_t n, DCounters *cnt, TravTune tune)
{
  if (S.trace_n_dev) { const uint32_t nd_ = *S.trace_n_dev; n = nd_ < n ? nd_ : n; }
  __shared__ uint32_t s_stack[FJ_STACK_LDS * BLOCK];
  __shared__ double s_inst[kInstLds ? InstLds::WORDS : 1];
  if (kInstLds) InstLds::fill(S, s_inst);         // (the launcher picked this instantiation because the scene fits)
  ClosestPolicy pol;
  pol.S = &S; pol.rays = rays; pol.paths = paths; pol.hits = hits; pol.default_group = S.target_group;
  LocalCounters lc = {0, 0, 0};
  traverse_phased<kCount, false, kInstLds>(S, pol, tune, n, &cnt->trace_xcd_head[0][0], make_stack(s_stack, S.stack_overflow, nullptr, FJ_STACK_LDS), &lc, s_inst);
  if (kCount) {
    flush_counters(cnt, lc.nodes, lc.prims, lc.insts, 0, 0);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&cnt->traced, (unsigned long long) n);
  }
}

#endif
