// fjgpu_dev_anyhit.h -- the lean any-hit walk: shadow rays of scenes in which every possible
// occluder is opaque (Os = 1) and no curve set / time-sampled transform exists -- the common
// case and the dominant kernel of C1-C3.
// Part of the kernels translation unit: included by fjgpu_kernels.hip only (device code,
// compiled with -ffp-contract=off; see the header of that file).
#ifndef FJGPU_DEV_ANYHIT_H
#define FJGPU_DEV_ANYHIT_H

// Same tests, same instances, same result (occluded or not) as traverse_persistent with
// any-hit rays (SlIlluminance's shadow SlTrace, src/fj_shading.cc:338-355, through
// Accelerator::Intersect -> TriRayIntersect); what is gone is the closest-hit bookkeeping.
//
// Scheduling.  A lane is in one of three states: TURNOVER (no BLAS walk in progress: idle,
// just finished, or between the instances of a group), INNER (at a 4-wide node) or LEAF
// (holding 1..4 triangles).  Every iteration the WAVE executes exactly one phase -- the one
// most lanes wait for (turnover once enough lanes collected there, or when nothing else can
// run) -- instead of all phases in sequence with whoever happens to be there: lanes pile up
// where the wave is not, so a phase runs with most of the wave (measured before: turnover ran
// with 17, inner steps with 42, triangle tests with 16 of 64 lanes).  The leaf phase tests ONE
// triangle per lane and execution, so lanes with 1 and 4 triangles do not wait for each other.
//
// Box tests are the conservative sign-aware f32 slab test on the quantised 64-byte nodes
// (slab32q_test, fjgpu_dev_math.h).
//
// Entry.  The light loop has already tested the (single) instance box of single-instance
// shadow groups and writes ~instance into the queue entry: the walk reads one flat record
// (DAnyInst) and starts.  Groups with several instances walk the threaded instance BVH here.
//
// Measured and dropped, with their numbers (the code is kept as a patch: profiles/r04_anyhit_dropped_experiments.patch):
// an 8-wide twin of the tree (29 % fewer node visits, 15 % more VALU instructions: C3 walk 64.4 -> 72.7 ms), cache-warming
// touches of the node / triangle a lane will come back to (63.3 -> 68.3 .. 75.6 ms), the unsigned slab variant, ALU / load
// padding and the f64 validation of the slab test (profiles/r03_anyhit_bound_experiments.txt, r03_anyhit_wide8_and_perm.txt).

// Leaf phase (round 6).  FJ_ANYHIT_F32 1: a triangle is first put to the conservative packed-f32 filter of fjgpu_tri_filter.h
// (same determinants, explicit error bounds): FJ_TRI_MISS / FJ_TRI_HIT settle the test as the reference's FP64 statements
// would; FJ_TRI_MAYBE (a fraction of a percent: rays grazing an edge, a hit at t ~ tmin / tmax) parks the triangle in the
// lane (`pex`) and the lane walks on as after a miss (any-hit order is free).  A fourth phase -- EXACT, run once
// tune.exact_min lanes hold such a triangle, or as soon as one of them can do nothing else (its walk is over, or the filter
// left a second triangle undecided) -- rebuilds the FP64 object-space ray from the queue entry with the entry code's own
// statements and runs tri_ray_anyhit.  The FP64 ray therefore no longer lives in LDS: a lane keeps ten floats there (fl32(o), o - fl32(o),
// fl32(d), tfl) instead of seven doubles.  FJ_ANYHIT_F32 0 is round 5's leaf phase (FP64 test on every candidate).
#ifndef FJ_ANYHIT_F32
#define FJ_ANYHIT_F32 1
#endif
#if FJ_ANYHIT_F32
#define FJ_ANYHIT_RAY_WORDS 10          // 32-bit words of LDS per lane behind the stack
#else
#define FJ_ANYHIT_RAY_WORDS 14
#endif

// lane id the compiler cannot treat as a loop invariant
__device__ __forceinline__ uint32_t opaque_lane_id()
{
  uint32_t t;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(t));
  return t;
}
#ifndef FJ_NO_SCHED_FENCE
#define FJ_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define FJ_SCHED_FENCE() do { } while (0)
#endif
// TriRayIntersect (src/fj_triangle.cc:81-153, non-culling branch; tri_ray of fjgpu_dev_math.h) with scheduling fences between its steps:
// left to itself the scheduler overlaps them and the leaf phase needs a dozen registers more than the walk has.
// The same decision for rays that only ask WHETHER they hit (the any-hit walk: 7 of 8 rays hit nothing, and of the triangles tested
// nearly all are missed): u = U / det and v = V / det are compared with 0 and 1, and most tests are settled by the SIGNS and the SIZES of
// U, V and det -- without the division (a dozen instructions, one of them at a quarter of the rate).  Rejected early only where the
// reference's own arithmetic provably rejects: u = U * fl(1 / det) < 0 when U and det differ in sign and the product cannot underflow
// (|U| > 1e-100 |det|); u > 1 when |U| > |det| (1 + 1e-10) (fl(1 / det) and the product are each good to 2^-53); the same for v.
// Whatever is left takes the reference's statements as they are.
__device__ __forceinline__ bool tri_ray_anyhit(V3 v0, V3 v1, V3 v2, V3 orig, V3 dir, double tmin, double tmax)
{
  const V3 edge1 = v1 - v0;
  const V3 edge2 = v2 - v0;
  FJ_SCHED_FENCE();
  const V3 pvec = cross(dir, edge2);
  const double det = dot(edge1, pvec);
  FJ_SCHED_FENCE();
  if (det > -1e-6 && det < 1e-6) return false;
  const double adet = fabs(det);
  const V3 tvec = orig - v0;
  const double U = dot(tvec, pvec);
  const double aU = fabs(U);
  if (((U < 0.0) != (det < 0.0)) ? aU > 1e-100 * adet : aU > adet * (1.0 + 1e-10)) return false;
  FJ_SCHED_FENCE();
  const V3 qvec = cross(tvec, edge1);
  const double V = dot(dir, qvec);
  const double aV = fabs(V);
  if (((V < 0.0) != (det < 0.0)) ? aV > 1e-100 * adet : aV > adet * (1.0 + 1e-10)) return false;
  FJ_SCHED_FENCE();
  const double inv_det = 1.0 / det;
  const double uu = U * inv_det;
  if (uu < 0.0 || uu > 1.0) return false;
  const double vv = V * inv_det;
  if (vv < 0.0 || uu + vv > 1.0) return false;
  const double t = dot(edge2, qvec) * inv_det;
  return tmin <= t && t <= tmax;
}

#ifdef FJ_TRI_FILTER_VALIDATE
__device__ unsigned long long g_trifilter[8];      // verdicts (miss, hit, maybe), contradictions, exact hits
#endif
#ifdef FJ_PHASE_STATS
__device__ unsigned long long g_ahphase[16];
// debug build only: wave-level phase executions and the lanes active in them
#define PH(i, v) do { ph[i] += (unsigned long long) (v); } while (0)
#else
#define PH(i, v) do { } while (0)
#endif

// kMulti = false: every shadow group that can receive shadow rays has one instance, so every queue
// entry names its instance (the instance-BVH walk and its registers are compiled out).
template <bool kCount, bool kMulti>
__device__ void traverse_anyhit(const DScene &S, const DShadowRay *squeue, float *s_accum, TravTune tune,
    uint32_t n, uint32_t *xheads, uint32_t *s_stack, LocalCounters *lc)
{
  const unsigned lane = __lane_id();
  // Per-lane traversal stack: FJ_STACK_LDS_ANYHIT entries in LDS ([depth][thread]), deeper ones
  // in the global overflow area ([depth][global thread]; the builder reports the worst case).
  // Addresses are rebuilt from the thread index where they are needed: loop-invariant pointers
  // held in registers were what the compiler spilled (and reloaded before every push).
  // (the lane id comes from a volatile asm: a plain threadIdx.x expression is hoisted out of the
  // loop as an invariant and then spilled all the same)
  const uint32_t wave_first = __builtin_amdgcn_readfirstlane(threadIdx.x) & ~63u;     // SGPR
#define AH_TID() (wave_first + opaque_lane_id())
  // (the block's LDS through a pointer the compiler KNOWS to be LDS: through the generic parameter the stack's pop was a flat load --
  // the choice between the LDS part and the overflow area became a choice between two generic pointers -- which waits for both counters)
  typedef __attribute__((address_space(3))) uint32_t ah_lds_u32;
  typedef __attribute__((address_space(3))) double ah_lds_f64;
  ah_lds_u32 *const ls_stack = (ah_lds_u32 *) s_stack;
  // The stack pointer IS an LDS address: spa = the byte address of the lane's next free slot = stack base + (depth x BLOCK + thread) x 4.
  // A push / pop is one LDS instruction with an immediate offset and one add; with depth and thread index kept apart every access rebuilt
  // its address (two v_mbcnt, two shifts, an add3: a tenth of an inner step's instructions).  The thread's 4 x thread < 4 x BLOCK = one row:
  // the stack is empty while spa lies in the first row, and beyond row FJ_STACK_LDS_ANYHIT the entries live in the global overflow area.
  const uint32_t ah_base = (uint32_t) (uintptr_t) ls_stack;
  const uint32_t ah_row1 = ah_base + BLOCK * 4u, ah_ovf0 = ah_base + (uint32_t) FJ_STACK_LDS_ANYHIT * BLOCK * 4u;
#define AH_AT(addr) (*(ah_lds_u32 *) (uintptr_t) (addr))
#define AH_OVF(addr) S.stack_overflow_shadow[(size_t) ((((addr) - ah_base) >> 10) - FJ_STACK_LDS_ANYHIT) * (gridDim.x * BLOCK) + (size_t) blockIdx.x * BLOCK + AH_TID()]
  auto push = [&](uint32_t &spa_, uint32_t v) { if (spa_ < ah_ovf0) AH_AT(spa_) = v; else AH_OVF(spa_) = v; spa_ += BLOCK * 4u; };
  auto pop = [&](uint32_t &spa_) -> uint32_t { spa_ -= BLOCK * 4u; uint32_t v_; if (spa_ < ah_ovf0) v_ = AH_AT(spa_); else v_ = AH_OVF(spa_); return v_; };
  bool head_live = true;                   // wave-uniform: the global head still has entries
  uint32_t next = 0, range_end = 0;        // wave-uniform: the wave's claimed slice of the queue
  tune.grab = adaptive_grab(tune.grab, n);
  QueueClaim qc;                           // queue regions by XCD (DCounters.shadow_xcd_head)
  qc.init(xheads, n, tune.grab);
  bool have = false;                       // the lane holds a ray whose fate is open
  uint32_t idx = 0;
  // object-space ray (f64: the triangle test's operands): it lives in LDS ([k][thread]) instead of in 12 registers -- only
  // the leaf phase reads it -- which is what lets the walk run a sixth wave per SIMD (80 VGPRs)
#if FJ_ANYHIT_F32
  typedef __attribute__((address_space(3))) float ah_lds_f32;
  ah_lds_f32 *const s_ray = (ah_lds_f32 *) (ls_stack + FJ_STACK_LDS_ANYHIT * BLOCK);     // [10][thread]: oh xyz, ol xyz, d xyz, tfl (TriFilterRay)
  uint32_t pex = TRAV_DONE;                // triangle (leaf slot) the filter left undecided, awaiting the exact phase; bit 31: a SECOND one came up (the lane's
                                           // leaf cursor stays on it until the exact phase has run)
#define AH_PEX_STALL 0x80000000u
  int winst = 0;                           // kMulti: the instance being walked (the exact phase needs its M^-1 again)
#else
  ah_lds_f64 *const s_ray = (ah_lds_f64 *) (ls_stack + FJ_STACK_LDS_ANYHIT * BLOCK);
#endif
  Slab32P s32;                             // conservative slab constants of (ray, instance), in the packed fma's layout
  s32.ix = s32.iy = s32.iz = 0.f; s32.lhx = s32.lhy = s32.lhz = (fj_v2f) (0.f);
  float tmax32 = 0.f;                      // >= the ray's tmax (the exact f64 value sits in LDS behind the ray, for the triangle test)
  const float tmin32 = 9.9999e-5f;         // <= .0001
  int gi = 0, gend = 0;                    // cursor in the group's instance BVH; gi < 0: ~instance, settled by the light loop
  uint32_t node_base = 0, tri_base = 0;    // DAnyInst: offsets from S.blas_base (triangles: f32 records, see fjgpu_api.hip)
  uint32_t cur = TRAV_DONE;
  // a POSTPONED leaf: a lane that reaches a leaf while its stack is not empty sets the leaf aside and
  // walks on, so it has work in whichever phase the wave runs next (any hit ends the ray and 7 of 8
  // rays reach the light: the order of the tests is free, nothing is wasted but the inner steps an
  // occluded ray takes before its postponed leaf is tested)
  uint32_t pleaf = TRAV_DONE;
#define AH_POSTPONE() do { if (cur != TRAV_DONE && (cur & FJ_LEAF_FLAG) && pleaf == TRAV_DONE && spa >= ah_row1) { pleaf = cur; cur = pop(spa); } } while (0)
#define AH_PLEAF_NEXT() do { pleaf = TRAV_DONE; } while (0)
#define AH_PLEAF_CLEAR() do { pleaf = TRAV_DONE; } while (0)
  uint32_t spa = ah_base;                  // (set to the lane's own column when a ray enters an instance)
  const double tmin = .0001;
#ifdef FJ_PHASE_STATS
  unsigned long long ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

#ifdef FJ_WAVE_TIMELINE
  uint32_t ray_steps = 0;
#endif
  FJ_TL_DECL();
  for (;;) {
    PH(0, 1);
    FJ_TL_ITER(!head_live && next >= range_end);
    AH_POSTPONE();
    const bool walk_over = cur == TRAV_DONE && pleaf == TRAV_DONE;
#if FJ_ANYHIT_F32
    const bool has_pex = pex != TRAV_DONE;
    const bool stalled = has_pex && (pex & AH_PEX_STALL);       // its leaf cursor waits for the exact phase
#else
    const bool has_pex = false, stalled = false;
#endif
    const bool fin = walk_over && !has_pex;
    const bool at_inner = cur != TRAV_DONE && !(cur & FJ_LEAF_FLAG);
    const bool at_leaf = (pleaf != TRAV_DONE || (cur != TRAV_DONE && (cur & FJ_LEAF_FLAG))) && !stalled;     // a lane may be both
    const unsigned long long m_leaf = __ballot(at_leaf), m_inner = __ballot(at_inner);
    const unsigned n_leaf = (unsigned) __popcll(m_leaf), n_inner = (unsigned) __popcll(m_inner);
    // lanes for which a turnover does something: a ray to retire / move on, or a new one to fetch
    const bool can_fetch = head_live || next < range_end;
    const unsigned long long m_turn = __ballot(fin && (have || can_fetch));
    const unsigned n_turn = (unsigned) __popcll(m_turn);

#if FJ_ANYHIT_F32
    {
      // ---- exact: the reference's FP64 statements for the triangles the filter left undecided
      const unsigned long long m_ex = __ballot(has_pex);
      if (m_ex != 0ull && ((unsigned) __popcll(m_ex) >= tune.exact_min || __ballot(has_pex && (walk_over || stalled)) != 0ull)) {
        PH(12, 1); PH(13, __popcll(m_ex));
        if (has_pex) {
          V3 o, d;
          double tmax;
          sq_ray(S, squeue, idx, &o, &d, &tmax);
          const DAnyInst *A = &S.any_insts[kMulti ? winst : ~SQ_FIELD(S, squeue, idx, group)];
          // (the entry code's statements on the entry code's operands: the same object-space ray, bit for bit)
          const V3 oo_ = xpoint(A->Minv, o), od_ = xvector(A->Minv, d);
          FJ_SCHED_FENCE();
          V3 v0, v1, v2;
          load_tri(nullptr, (const float *) (S.blas_base + ((size_t) tri_base << 7)), pex & ~AH_PEX_STALL, &v0, &v1, &v2);
          if (tri_ray_anyhit(v0, v1, v2, oo_, od_, tmin, tmax)) {
            have = false; cur = TRAV_DONE; AH_PLEAF_CLEAR();     // occluded: nothing to add
#ifdef FJ_WAVE_TIMELINE
            FJ_TL_RAY(ray_steps);
#endif
            PH(10, 1); PH(14, 1);
          }
          pex = TRAV_DONE;
        }
        continue;
      }
    }
#endif
    if (n_turn >= TRAV_REFILL || (n_inner == 0 && n_leaf == 0)) {
      if (m_turn == 0ull) { FJ_TL_END(); break; }           // nothing in flight, nothing left to fetch
      PH(1, 1); PH(2, n_turn);
      // ---- turnover: retire, fetch, enter
      if (next >= range_end && head_live) { head_live = qc.claim(lane, &next, &range_end); FJ_TL_CLAIM(); }
      bool fetch = false;
      if (fin) {
        fetch = true;
        if (have) {
          if (kMulti && gi < gend) fetch = false;    // the group has more instances: the same ray goes on
          else {
            // reached the light: add c (an opaque occluder would have added c * (1 - Os) = 0)
            // one of k entries of a ray with k candidate instances: the light is added by the entry that completes
            // the count of those that reached it (DScene.shadow_join)
            bool add = true;
            if (!kMulti && S.shadow_join) {
              const uint32_t slot1 = SQ_FIELD(S, squeue, idx, tindex);
              if (slot1) { const uint32_t old = atomicAdd(&S.shadow_join[slot1 - 1u], 1u); add = ((old & 0xffffu) + 1u) == (old >> 16); }
            }
            if (add) {
              float *acc = s_accum + 4 * (size_t) SQ_FIELD(S, squeue, idx, sample);
              const float *qc_ = sq_colour(S, squeue, idx);
              const float r0 = qc_[0], r1 = qc_[1], r2 = qc_[2];
              if (r0 != 0.f) atomicAdd(acc + 0, r0);
              if (r1 != 0.f) atomicAdd(acc + 1, r1);
              if (r2 != 0.f) atomicAdd(acc + 2, r2);
            }
            have = false;
#ifdef FJ_WAVE_TIMELINE
            FJ_TL_RAY(ray_steps);
#endif
          }
        }
      }
      const unsigned long long m_fetch = __ballot(fetch);
      if (fetch) {
        const uint32_t my = next + __builtin_amdgcn_mbcnt_hi((uint32_t) (m_fetch >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m_fetch, 0u));   // set bits below this lane
        if (my < range_end) {
          const int g = SQ_FIELD(S, squeue, my, group);
          if (SQ_FIELD(S, squeue, my, sample) != SQ_INVALID) {         // (padding slot of a partially filled chunk)
#ifdef FJ_WAVE_TIMELINE
            ray_steps = 0;
#endif
            have = true;
            idx = my;
            if (!kMulti || g < 0) { gi = g; gend = 0; }
            else { gi = S.groups[g].first; gend = gi + S.groups[g].count; }
          }
        }
      }
      next += (uint32_t) __popcll(m_fetch);
      if (next > range_end) next = range_end;
      if (fin && have) {
        V3 o, d;
        double tmax;
        sq_ray(S, squeue, idx, &o, &d, &tmax);
        V3 winv = o;
        bool plain = false, single = false, dead = false;
        if (kMulti && gi >= 0) {
          // BoxRayIntersect's -0.0 quirk: such a ray fails every box test in the reference
          dead = has_negative_zero(d);
          winv = mk(filter_rcp(d.x), filter_rcp(d.y), filter_rcp(d.z));
          plain = plain_dir(d);
          single = S.groups[SQ_FIELD(S, squeue, idx, group)].n_instances == 1;
          if (dead) gi = gend;
        }
        for (;;) {
          int inst = -1;
          if (gi < 0) { inst = ~gi; gi = gend = 0; if (kCount) lc->insts++; }     // its box test passed in the light loop
          else if (!kMulti) break;
          else {
            while (gi < gend) {
              const DTNode *tn_ = &S.group_nodes[gi];
              if (tn_->inst < 0) {         // inner node of the instance BVH
                double tq;
                gi = slab(tn_->box, tn_->box + 3, o, winv, tmin, tmax, &tq) ? gi + 1 : tn_->skip;
                continue;
              }
              gi++;
              if (kCount) lc->insts++;
              // (a single-instance group's box: the light loop queued this ray BECAUSE that test passed)
              if (!single && !box_ray_ref_fast(tn_->box, o, d, winv, plain, tmin, tmax)) continue;
              inst = tn_->inst;
              break;
            }
            if (inst < 0) break;           // no instance left: the ray reaches the light (next turnover)
          }
          const DAnyInst *A = &S.any_insts[inst];
          if (A->n_prims == 0) continue;
          const V3 oo_ = xpoint(A->Minv, o), od_ = xvector(A->Minv, d);
          if (has_negative_zero(od_)) continue;
#if FJ_ANYHIT_F32
          {
            // the filter's ray (TriFilterRay, fjgpu_tri_filter.h): fl32(o), o - fl32(o), fl32(d), tfl; NaN where the error analysis does not hold
            const float ohx = (float) oo_.x, ohy = (float) oo_.y, ohz = (float) oo_.z;
            float dx32 = (float) od_.x, dy32 = (float) od_.y, dz32 = (float) od_.z;
            const float om = fmaxf(fmaxf(fabsf(ohx), fabsf(ohy)), fabsf(ohz)) * 1.0000002f;
            const float dm = fmaxf(fmaxf(fabsf(dx32), fabsf(dy32)), fabsf(dz32));
            float tfl = om * 2.3841864e-7f;
            // (magnitudes beyond the error analysis of the filter -- or a NaN anywhere: the comparison fails -- poison the ray: every verdict MAYBE.  The
            //  ray's tmax needs no guard: tmax32 x D may overflow to +inf, which makes "t > tmax" unprovable and "t < tmax" true, as they are.)
            if (!(dm <= 1073741824.f && om + A->fbound <= 536870912.f) || tune.filter_off) { dx32 = dy32 = dz32 = tfl = __builtin_nanf(""); }
            ah_lds_f32 *const wp_ = &s_ray[AH_TID()];
            wp_[0] = ohx; wp_[BLOCK] = ohy; wp_[2 * BLOCK] = ohz;
            wp_[3 * BLOCK] = (float) (oo_.x - (double) ohx); wp_[4 * BLOCK] = (float) (oo_.y - (double) ohy); wp_[5 * BLOCK] = (float) (oo_.z - (double) ohz);
            wp_[6 * BLOCK] = dx32; wp_[7 * BLOCK] = dy32; wp_[8 * BLOCK] = dz32; wp_[9 * BLOCK] = tfl;
            if (kMulti) winst = inst;
          }
#else
          { ah_lds_f64 *const wp_ = &s_ray[AH_TID()]; wp_[0] = oo_.x; wp_[BLOCK] = oo_.y; wp_[2 * BLOCK] = oo_.z; wp_[3 * BLOCK] = od_.x; wp_[4 * BLOCK] = od_.y; wp_[5 * BLOCK] = od_.z; wp_[6 * BLOCK] = tmax; }
#endif
          const V3 inv = mk(filter_rcp(od_.x), filter_rcp(od_.y), filter_rcp(od_.z));
          // the primitive set's own box: only where several instances are tried (a ray that misses
          // it finds no child box at the root either; in the single-instance walk the 12 registers
          // of the box cost more than that one node)
          if (kMulti) {
            double tn;
            if (!slab(A->bounds, A->bounds + 3, oo_, inv, tmin, tmax, &tn)) continue;
          }
          s32 = slab32p(slab32q_setup(oo_, inv, A->qorigin, A->qcell));
          tmax32 = f32_above(tmax);
          node_base = A->node_base; tri_base = A->tri_base;
          cur = A->root; spa = ah_base + AH_TID() * 4u;
          break;
        }
      }
      continue;
    }

    if (n_inner * tune.leaf_bias8 >= n_leaf * 8u) {
      // ---- inner nodes: one 64-byte node per lane; further steps without a new vote while at
      // least tune.min_inner lanes stay at inner nodes
      for (uint32_t step = 0;; step++) {
      const bool in_now = step == 0 ? at_inner : (cur != TRAV_DONE && !(cur & FJ_LEAF_FLAG));
      if (step > 0) {
        const unsigned n_now = (unsigned) __popcll(__ballot(in_now));
        if (step >= tune.anyhit_steps || n_now < tune.min_inner) break;
        PH(3, 1); PH(4, n_now);
      } else { PH(3, 1); PH(4, n_inner); }
#ifdef FJ_PHASE_STATS
      // where the lanes that take no part in this inner step are: idle (ray finished, waiting for the turnover) or held at a leaf
      { const bool in_ = step == 0 ? at_inner : (cur != TRAV_DONE && !(cur & FJ_LEAF_FLAG));
        const bool fin_ = cur == TRAV_DONE && pleaf == TRAV_DONE && !has_pex;
        PH(7, __popcll(__ballot(fin_))); PH(8, __popcll(__ballot(!in_ && !fin_))); }
#endif
      // (rare) a lane close to the end of its LDS stack: this step pushes through the overflow path
      const bool deep = __ballot(in_now && spa + 3u * BLOCK * 4u > ah_ovf0) != 0ull;
      if (in_now) {
        // 64-byte quantised node: four 16-byte loads (12 words of (min, max) pairs + 4 child refs)
        const FJ_GLOBAL fj_v4u *nd = (const FJ_GLOBAL fj_v4u *) (S.blas_base + ((size_t) node_base << 7) + ((size_t) cur << 6));
        if (kCount) lc->nodes++;
#ifdef FJ_WAVE_TIMELINE
        ray_steps++;
#endif
        const fj_v4u w0 = nd[0], w1 = nd[1], w2 = nd[2], e = nd[3];
        // (one box after the other: interleaved by the scheduler, the four tests held 48 temporaries)
        const uint32_t shx = slab32_shift(s32.ix), shy = slab32_shift(s32.iy), shz = slab32_shift(s32.iz);
        const bool h0 = slab32q_test(w0.x, w0.y, w0.z, s32, shx, shy, shz, tmin32, tmax32);
        FJ_SCHED_FENCE();
        const bool h1 = slab32q_test(w0.w, w1.x, w1.y, s32, shx, shy, shz, tmin32, tmax32);
        FJ_SCHED_FENCE();
        const bool h2 = slab32q_test(w1.z, w1.w, w2.x, s32, shx, shy, shz, tmin32, tmax32) && e.z != FJ_NO_CHILD;
        FJ_SCHED_FENCE();
        const bool h3 = slab32q_test(w2.y, w2.z, w2.w, s32, shx, shy, shz, tmin32, tmax32) && e.w != FJ_NO_CHILD;
        FJ_SCHED_FENCE();
        // any hit ends the ray and 7 of 8 rays reach the light, so the visiting order is free:
        // no distance sort (children are stored by decreasing surface area)
        uint32_t r0 = e.x, r1 = e.y, r2 = e.z, r3 = e.w;
        if (!h2) { r2 = r3; }
        if (!h1) { r1 = r2; r2 = r3; }
        if (!h0) { r0 = r1; r1 = r2; r2 = r3; }
        const int nh = (int) h0 + (int) h1 + (int) h2 + (int) h3;
        if (nh == 0) cur = (spa < ah_row1) ? TRAV_DONE : pop(spa);
        else {
          cur = r0;
          if (!deep) {
            // three unconditional stores (whatever lies above the new top is dead) instead of
            // three predicated ones
            ah_lds_u32 *top = &AH_AT(spa);
            top[0] = r1; top[BLOCK] = r2; top[2 * BLOCK] = r3;
            spa += (uint32_t) (nh - 1) * (BLOCK * 4u);
          } else {
            if (nh > 1) push(spa, r1);
            if (nh > 2) push(spa, r2);
            if (nh > 3) push(spa, r3);
          }
        }
        AH_POSTPONE();
      }
      }
    } else {
      // ---- leaves: ONE triangle per lane; the first hit inside [tmin, tmax] ends the ray
      PH(5, 1); PH(6, n_leaf);
      PH(9, __popcll(__ballot(fin))); PH(11, __popcll(__ballot(!at_leaf && !fin)));
      if (at_leaf) {
        const bool from_p = pleaf != TRAV_DONE;        // the postponed leaf first: its slot frees
        const uint32_t lf = from_p ? pleaf : cur;
        const uint32_t first = (lf & 0x7fffffffu) >> 3;
        const uint32_t more = lf & 7u;
        if (kCount) lc->prims++;
#if FJ_ANYHIT_F32
        TriFilterRay fr;
        {
          ah_lds_f32 *const rp_ = &s_ray[AH_TID()];
          fr.ohx = rp_[0]; fr.ohy = rp_[BLOCK]; fr.ohz = rp_[2 * BLOCK]; fr.olx = rp_[3 * BLOCK]; fr.oly = rp_[4 * BLOCK]; fr.olz = rp_[5 * BLOCK];
          fr.dx = rp_[6 * BLOCK]; fr.dy = rp_[7 * BLOCK]; fr.dz = rp_[8 * BLOCK]; fr.tfl = rp_[9 * BLOCK];
          // tmin = .0001; tmax32 in [tmax, tmax (1 + 2^-23)]: factors 1 -+ 2^-18
          fr.tmin_lo = 9.9999800e-05f; fr.tmin_hi = 1.0000020e-04f; fr.tmax_lo = tmax32 * 0.99999619f; fr.tmax_hi = tmax32 * 1.0000039f;
        }
        const int verdict = tri_filter32<true>((const FJ_GLOBAL float *) (S.blas_base + ((size_t) tri_base << 7)) + (size_t) first * 9, fr);
#ifdef FJ_TRI_FILTER_VALIDATE
        {
          // debug build: the exact test behind every decision of the filter
          V3 o, d, v0, v1, v2;
          double tmax;
          sq_ray(S, squeue, idx, &o, &d, &tmax);
          const DAnyInst *A = &S.any_insts[kMulti ? winst : ~SQ_FIELD(S, squeue, idx, group)];
          load_tri(nullptr, (const float *) (S.blas_base + ((size_t) tri_base << 7)), first, &v0, &v1, &v2);
          const bool ex_ = tri_ray_anyhit(v0, v1, v2, xpoint(A->Minv, o), xvector(A->Minv, d), tmin, tmax);
          atomicAdd(&g_trifilter[verdict], 1ull);
          if ((verdict == FJ_TRI_MISS && ex_) || (verdict == FJ_TRI_HIT && !ex_)) atomicAdd(&g_trifilter[3], 1ull);
          if (ex_) atomicAdd(&g_trifilter[4], 1ull);
        }
#endif
        // undecided: the triangle waits in `pex` for the exact phase and the cursor moves on as for a miss; with `pex` taken the cursor
        // stays (the exact phase runs at the next vote, the triangle is put to the filter again after it)
        const bool stall_now = verdict == FJ_TRI_MAYBE && has_pex;
        if (verdict == FJ_TRI_MAYBE) pex = has_pex ? (pex | AH_PEX_STALL) : first;
        const bool occluded = verdict == FJ_TRI_HIT;
        if (occluded) pex = TRAV_DONE;
#else
        V3 v0, v1, v2;
        load_tri(nullptr, (const float *) (S.blas_base + ((size_t) tri_base << 7)), first, &v0, &v1, &v2);
        FJ_SCHED_FENCE();
        ah_lds_f64 *const rp_ = &s_ray[AH_TID()];       // (one address for the six loads)
        const bool occluded = tri_ray_anyhit(v0, v1, v2, mk(rp_[0], rp_[BLOCK], rp_[2 * BLOCK]), mk(rp_[3 * BLOCK], rp_[4 * BLOCK], rp_[5 * BLOCK]), tmin, rp_[6 * BLOCK]);
        const bool stall_now = false;
#endif
        if (occluded) {
          have = false; cur = TRAV_DONE;     // occluded: nothing to add
          AH_PLEAF_CLEAR();
#ifdef FJ_WAVE_TIMELINE
          FJ_TL_RAY(ray_steps);
#endif
          PH(10, 1);
        }
        else if (stall_now) { }
        else if (from_p) { if (more) pleaf = FJ_LEAF_FLAG | ((first + 1u) << 3) | (more - 1u); else AH_PLEAF_NEXT(); }
        else if (more) cur = FJ_LEAF_FLAG | ((first + 1u) << 3) | (more - 1u);
        else cur = (spa < ah_row1) ? TRAV_DONE : pop(spa);
      }
    }
  }
#undef AH_AT
#undef AH_OVF
#undef AH_POSTPONE
#undef AH_PLEAF_NEXT
#undef AH_PLEAF_CLEAR
#undef AH_TID
#ifdef FJ_PHASE_STATS
  // 0-6 are wave-uniform tallies (lane 0 speaks for the wave); 10 was counted by single lanes
  for (int i = 0; i < 16; i++) {
    const unsigned long long v = (i == 10 || i == 14) ? wave_sum(ph[i]) : ph[i];
    if (lane == 0 && v) atomicAdd(&g_ahphase[i], v);      // (tallies of its own: g_phase also takes the closest-hit walks' ticks)
  }
#endif
}

// blocks per CU (= waves per SIMD): what the registers allow WITHOUT a spill (any spill in the loop
// doubled the frame time).  The walk is bound by node-fetch latency x occupancy (profiles/r03_anyhit_bound_experiments.txt:
// 3 / 4 / 5 waves 85 / 71 / 63.5 ms), so round 3 bought a SIXTH wave: the object-space ray -- 12 registers that only the
// leaf phase reads -- lives in LDS next to a stack of 12 instead of 24 entries (deeper ones in the global
// overflow area), which brings the instantiation without the instance-BVH walk to 80 VGPRs, no spill: C3 63.3 -> 60.4 ms, C6
// 294.5 -> 281.9, C2 53.8 -> 52.0 (profiles/r03_exp13_six_waves.txt).  A seventh wave (72 VGPRs: 10 spills, 8 stack entries)
// loses again: 67.4 ms.  The general instantiation needs 98 = 4 waves.
#ifndef FJ_ANYHIT_MINB
#if FJ_ANYHIT_F32
#define FJ_ANYHIT_MINB 7                // (round 6: 88 bytes of LDS per lane and 72 VGPRs; what the compiler spills at 72 it spills inside the rare exact phase)
#else
#define FJ_ANYHIT_MINB 6
#endif
#endif
#ifndef FJ_ANYHIT_MINB_MULTI
#define FJ_ANYHIT_MINB_MULTI 5          // (96 VGPRs, no spill: C2 without the split 134.5 -> 124.1 ms; with the split 117.9)
#endif
static_assert((FJ_STACK_LDS_ANYHIT * BLOCK * 4) % 8 == 0, "the object-space rays behind the LDS stack are doubles");
template <bool kCount, bool kMulti>
__global__ void __launch_bounds__(BLOCK, kMulti ? FJ_ANYHIT_MINB_MULTI : FJ_ANYHIT_MINB) k_shadow_anyhit(DScene S, const DShadowRay *squeue, float *s_accum,
    DCounters *cnt, TravTune tune)
{
  __shared__ alignas(16) uint32_t s_stack[FJ_STACK_LDS_ANYHIT * BLOCK + FJ_ANYHIT_RAY_WORDS * BLOCK];     // (+ the rays: TriFilterRay, or 6 doubles per thread and tmax)
  const uint32_t n = cnt->shadow_count < S.shadow_queue_cap ? cnt->shadow_count : S.shadow_queue_cap;
  LocalCounters lc = {0, 0, 0};
  traverse_anyhit<kCount, kMulti>(S, squeue, s_accum, tune, n, &cnt->shadow_xcd_head[0][0], s_stack, &lc);
  if (kCount) {
    flush_counters(cnt, lc.nodes, lc.prims, lc.insts, 0, 0);
    flush_shadow_walk_counters(cnt, lc.nodes, lc.prims, lc.insts);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&cnt->squeued, (unsigned long long) n);
  }
}

#endif
