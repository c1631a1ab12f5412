// fjgpu_dev_anyhit_curves.h -- the any-hit walk of scenes WITH curve sets: shadow rays of scenes in which every possible
// occluder is opaque (Os = 1) and no time-sampled transform exists, at least one primitive set being Bezier ribbons (C5).
// Part of the kernels translation unit: included by fjgpu_kernels.hip only (device code,
// compiled with -ffp-contract=off; see the header of that file).
#ifndef FJGPU_DEV_ANYHIT_CURVES_H
#define FJGPU_DEV_ANYHIT_CURVES_H

// Same tests, same instances, same result (occluded or not) as traverse_persistent<kCurves, .., kAnyOnly> (SlIlluminance's shadow
// SlTrace, src/fj_shading.cc:338-355, through Accelerator::Intersect -> Curve::ray_intersect, src/fj_curve.cc:187-232, or
// TriRayIntersect): the instance loop with the reference's instance box, the object-space ray through M^-1, the capsule of the
// piece entered, the ribbon test of the whole curve with the reference grid's cell rule (fjgpu_dev_curve.h), the inclusive range.
// What differs is the scheduling, which is the lean any-hit walk's (fjgpu_dev_anyhit.h): the general walk runs every block of its
// loop body in every iteration with whoever is there -- on C5 the instance entry with 14, the inner steps with 24, the leaf block
// with 21 of 64 lanes, 168 VGPRs with 22 spilled -- while here a lane is in ONE of four states and the wave executes ONE phase per
// iteration, the one most lanes wait for:
//   TURNOVER  no BLAS walk in progress and no ribbon test owed: retire the ray (it reached the light), fetch the next queue entry,
//             enter the next instance of the group (instance level in LDS, InstLdsCurves)
//   INNER     at a 4-wide quantised node (64 bytes, sign-aware packed-f32 slab tests, no distance sort: any hit ends the ray)
//   LEAF      at a fresh leaf: ONE triangle (mesh instance), or the capsule test of ONE curve piece (curve instance); a piece that
//             passes becomes the lane's PENDING curve and the lane walks on -- the ribbon test's result only ends the ray
//   RIBBON    the ribbon test (recursive subdivision in ray space, long and divergent) of every lane that carries a pending curve:
//             runs when tune.leaf_wait_canyhit lanes carry one, or when more lanes are stuck behind theirs (a second candidate, or the end
//             of the instance) than any other phase could serve
// State per lane: 9 slab constants, the node array, five words of cursors -- the object-space ray sits in LDS behind the stack (only
// the leaf and ribbon phases read it), so the ribbon phase has the register file to itself: no spill at 4 waves per SIMD.
#ifndef FJ_STACK_LDS_CANYHIT
#define FJ_STACK_LDS_CANYHIT 16           // stack entries per lane in LDS (deeper ones: the global overflow area); >= FJ_STACK_LDS_MIN
#endif
#ifndef FJ_CANYHIT_MINB
#define FJ_CANYHIT_MINB 4
#endif
// Measured and dropped (profiles/r04_curve_anyhit_walk.txt): the ribbon tests split off into a kernel of their own (the walk at 6 waves queues
// (ray, curve) pairs: without the early end of an occluded ray 7.1 G pairs instead of 1.8 G tests, walk + ribbon kernel 2240 ms against 1130);
// a ray visiting the MESH instances of its group before the curve sets (1074 -> 1161 ms: the rays through the fur are not the ones the body
// occludes); sweeps cut into parts of <= 4 / 8 / 12 leaf walks per ribbon phase (1465 / 1176 / 1162 ms against 1074: the root is rebuilt per part and
// the sweeps are not long enough to pay for it); 3 waves without a spill (1417 against 1319 at 4); pieces per curve and SAH costs (no change).
#ifndef FJ_CANYHIT_POSTPONE
#define FJ_CANYHIT_POSTPONE 1
#endif
static_assert(FJ_STACK_LDS_CANYHIT >= FJ_STACK_LDS_MIN, "the overflow area is sized for FJ_STACK_LDS_MIN entries in LDS");
static_assert((FJ_STACK_LDS_CANYHIT * BLOCK * 4) % 8 == 0, "the object-space rays behind the LDS stack are doubles");

#ifdef FJ_PHASE_STATS
// debug builds only: wave-level clock ticks [0..3], executions [4..7] and lanes [8..11] of turnover / inner / leaf / ribbon; [12] iterations
__device__ unsigned long long g_cphase[16];
#define CA_PH(k, mask) do { const unsigned long long c_now = __builtin_readcyclecounter(); cph[k] += c_now - c_prev; c_prev = c_now; cph[4 + k]++; cph[8 + k] += __popcll(mask); } while (0)
#else
#define CA_PH(k, mask) do { } while (0)
#endif

template <bool kCount>
__device__ void traverse_anyhit_curves(const DScene &S, const DShadowRay *squeue, float *s_accum, TravTune tune,
    uint32_t n, uint32_t *xheads, uint32_t *s_stack, LocalCounters *lc, const double *s_inst)
{
  typedef InstLdsCurves IL;
  const DTNode *gnodes = (const DTNode *) s_inst;
  const DInstEntry *gents = (const DInstEntry *) (s_inst + IL::ENTRIES_AT);
  const DGroup *ggroups = (const DGroup *) (s_inst + IL::GROUPS_AT);
  const unsigned lane = __lane_id();
  const uint32_t wave_first = __builtin_amdgcn_readfirstlane(threadIdx.x) & ~63u;     // SGPR
#define CA_TID() (wave_first + opaque_lane_id())
  typedef __attribute__((address_space(3))) uint32_t ca_lds_u32;
  typedef __attribute__((address_space(3))) double ca_lds_f64;
  ca_lds_u32 *const ls_stack = (ca_lds_u32 *) s_stack;
  // the stack pointer is an LDS byte address (see traverse_anyhit): base + (depth x BLOCK + thread) x 4; empty while in row 0
  const uint32_t ca_base = (uint32_t) (uintptr_t) ls_stack;
  const uint32_t ca_row1 = ca_base + BLOCK * 4u, ca_ovf0 = ca_base + (uint32_t) FJ_STACK_LDS_CANYHIT * BLOCK * 4u;
#define CA_AT(addr) (*(ca_lds_u32 *) (uintptr_t) (addr))
#define CA_OVF(addr) S.stack_overflow_shadow[(size_t) ((((addr) - ca_base) >> 10) - FJ_STACK_LDS_CANYHIT) * (gridDim.x * BLOCK) + (size_t) blockIdx.x * BLOCK + CA_TID()]
  auto push = [&](uint32_t &spa_, uint32_t v) { if (spa_ < ca_ovf0) CA_AT(spa_) = v; else CA_OVF(spa_) = v; spa_ += BLOCK * 4u; };
  auto pop = [&](uint32_t &spa_) -> uint32_t { spa_ -= BLOCK * 4u; uint32_t v_; if (spa_ < ca_ovf0) v_ = CA_AT(spa_); else v_ = CA_OVF(spa_); return v_; };
  // object-space ray + tmax (f64: operands of the triangle, capsule and ribbon tests) in LDS, [k][thread]
  ca_lds_f64 *const s_ray = (ca_lds_f64 *) (ls_stack + FJ_STACK_LDS_CANYHIT * BLOCK);

  bool head_live = true;                   // wave-uniform: the global head still has entries
  uint32_t next = 0, range_end = 0;        // wave-uniform: the wave's claimed slice of the queue
  tune.grab = adaptive_grab(tune.grab, n);
  QueueClaim qc;
  qc.init(xheads, n, tune.grab);
  bool have = false;                       // the lane holds a ray whose fate is open
  uint32_t idx = 0;
  Slab32P s32;
  s32.ix = s32.iy = s32.iz = 0.f; s32.lhx = s32.lhy = s32.lhz = (fj_v2f) (0.f);
  float tmax32 = 0.f;
  const float tmin32 = 9.9999e-5f;         // <= .0001
  const double tmin = .0001;
  int ti = 0, tend = 0, ii = 0;            // cursor in the group's threaded instance level; the instance being walked
  const FJ_GLOBAL char *nodes = nullptr;   // its DNodeQ array
  bool in_curves = false;                  // ... is a curve set
  uint32_t cur = TRAV_DONE;
  uint32_t spa = ca_base;
  const uint32_t NONE = 0xffffffffu;
  uint32_t pend = NONE;                    // BLAS slot of the curve whose ribbon test is owed (same instance)
  uint32_t last_curve = NONE;              // curve that became a candidate last for this (ray, instance): its other pieces are skipped
  // a POSTPONED leaf (FJ_CANYHIT_POSTPONE, as in the lean any-hit walk): a lane that reaches a leaf while its stack is not empty sets the
  // leaf aside and walks on, so it has work in whichever of the two phases the wave runs next; the leaf phase serves the postponed leaf first
  uint32_t pleaf = TRAV_DONE;
  bool deep = false;                       // the leaf at hand (pleaf if there is one, else cur) is a curve piece that passed its capsule while
                                           // another curve is pending: it waits for the ribbon phase

#ifdef FJ_PHASE_STATS
  unsigned long long cph[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, c_prev = __builtin_readcyclecounter();
#endif
  for (;;) {
#ifdef FJ_PHASE_STATS
    cph[12]++;
#endif
    if (FJ_CANYHIT_POSTPONE && cur != TRAV_DONE && (cur & FJ_LEAF_FLAG) && pleaf == TRAV_DONE && spa >= ca_row1 && !deep) { pleaf = cur; cur = pop(spa); }
    const bool fin = cur == TRAV_DONE && pleaf == TRAV_DONE;
    const bool carrying = pend != NONE;
    const bool at_inner = cur != TRAV_DONE && !(cur & FJ_LEAF_FLAG);
    const bool at_leaf = !deep && (pleaf != TRAV_DONE || (cur != TRAV_DONE && (cur & FJ_LEAF_FLAG)));     // a lane may be both
    const bool can_fetch = head_live || next < range_end;
    const bool turn = fin && !carrying && (have || can_fetch);
    const unsigned long long m_turn = __ballot(turn), m_pend = __ballot(carrying);
    const unsigned n_inner = (unsigned) __popcll(__ballot(at_inner)), n_leaf = (unsigned) __popcll(__ballot(at_leaf));
    const unsigned n_turn = (unsigned) __popcll(m_turn), n_pend = (unsigned) __popcll(m_pend);
    const unsigned n_stuck = (unsigned) __popcll(__ballot((deep && !at_inner) || (fin && carrying)));

    if (n_turn >= tune.refill_canyhit || (n_inner == 0 && n_leaf == 0 && n_pend == 0)) {
      if (m_turn == 0ull) break;           // nothing in flight, nothing left to fetch
      // ---- turnover: retire, fetch, enter
      if (next >= range_end && head_live) head_live = qc.claim(lane, &next, &range_end);
      bool fetch = false;
      if (turn) {
        fetch = true;
        if (have) {
          if (ti < tend) fetch = false;    // the group has more instances: the same ray goes on
          else {
            // ran out of instances: the ray reaches the light (an opaque occluder would have added c * (1 - Os) = 0)
            float *acc = s_accum + 4 * (size_t) SQ_FIELD(S, squeue, idx, sample);
            const float *qc_ = sq_colour(S, squeue, idx);
            const float r0 = qc_[0], r1 = qc_[1], r2 = qc_[2];
            if (r0 != 0.f) atomicAdd(acc + 0, r0);
            if (r1 != 0.f) atomicAdd(acc + 1, r1);
            if (r2 != 0.f) atomicAdd(acc + 2, r2);
            have = false;
          }
        }
      }
      const unsigned long long m_fetch = __ballot(fetch);
      if (fetch) {
        const uint32_t my = next + __builtin_amdgcn_mbcnt_hi((uint32_t) (m_fetch >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m_fetch, 0u));
        if (my < range_end) {
          const int g = SQ_FIELD(S, squeue, my, group);
          if (SQ_FIELD(S, squeue, my, sample) != SQ_INVALID) {         // (padding slot of a partially filled chunk)
            have = true;
            idx = my;
            ti = ggroups[g].first; tend = ti + ggroups[g].count;
          }
        }
      }
      next += (uint32_t) __popcll(m_fetch);
      if (next > range_end) next = range_end;
      if (turn && have) {
        V3 o, d;
        double tmax;
        sq_ray(S, squeue, idx, &o, &d, &tmax);
        const DGroup *G = &ggroups[SQ_FIELD(S, squeue, idx, group)];
        const bool single = G->n_instances == 1;
        // BoxRayIntersect's -0.0 quirk: such a ray fails every box test in the reference
        if (has_negative_zero(d)) ti = tend;
        const V3 winv = mk(filter_rcp(d.x), filter_rcp(d.y), filter_rcp(d.z));
        const bool plain = plain_dir(d);
        while (ti < tend) {
          const DTNode *tn_ = &gnodes[ti];
          if (tn_->inst < 0) {             // inner node of the instance level: conservative box, skip link
            double tq;
            ti = slab(tn_->box, tn_->box + 3, o, winv, tmin, tmax, &tq) ? ti + 1 : tn_->skip;
            continue;
          }
          ii = tn_->inst;
          ti++;
          const DInstEntry *I = &gents[ii];
          if (kCount) lc->insts++;
          double tn;
          // a tight box of the geometry first (pure culling, DInstEntry.tbounds)
          if (!slab(I->tbounds, I->tbounds + 3, o, winv, tmin, tmax, &tn)) continue;
          // the reference's own (possibly non-enclosing) instance box
          if (!box_ray_ref_fast(single ? G->sbounds : tn_->box, o, d, winv, plain, tmin, tmax)) continue;
          const V3 oo_ = xpoint(I->Minv, o), od_ = xvector(I->Minv, d);
          if (has_negative_zero(od_)) continue;
          if (I->pn_prims == 0) continue;
          const V3 inv = mk(filter_rcp(od_.x), filter_rcp(od_.y), filter_rcp(od_.z));
          if (!slab(I->pbounds, I->pbounds + 3, oo_, inv, tmin, tmax, &tn)) continue;
          { ca_lds_f64 *const wp_ = &s_ray[CA_TID()]; wp_[0] = oo_.x; wp_[BLOCK] = oo_.y; wp_[2 * BLOCK] = oo_.z; wp_[3 * BLOCK] = od_.x; wp_[4 * BLOCK] = od_.y; wp_[5 * BLOCK] = od_.z; wp_[6 * BLOCK] = tmax; }
          s32 = slab32p(slab32q_setup(oo_, inv, I->qorigin, I->qcell));
          tmax32 = f32_above(tmax);
          nodes = (const FJ_GLOBAL char *) I->nodes;
          in_curves = I->ptype == FJ_PRIMSET_CURVE;
          cur = I->proot; spa = ca_base + CA_TID() * 4u;
          last_curve = NONE;
          break;
        }
      }
      CA_PH(0, m_turn);
      continue;
    }

    if (n_pend >= tune.leaf_wait_canyhit || (n_inner == 0 && n_leaf == 0) || n_stuck > (n_inner > n_leaf ? n_inner : n_leaf)) {
      // ---- ribbon tests of the pending curves (n_pend > 0 here: with nothing else to run and no pending curve the turnover ran)
      if (carrying) {
        const uint32_t sl = pend;
        pend = NONE;
        const DInstEntry *E = &gents[ii];
        const DPrimSet *P = &S.primsets[E->primset];
        ca_lds_f64 *const rp_ = &s_ray[CA_TID()];
        const V3 oo = mk(rp_[0], rp_[BLOCK], rp_[2 * BLOCK]), od = mk(rp_[3 * BLOCK], rp_[4 * BLOCK], rp_[5 * BLOCK]);
        const FJ_GLOBAL double *cp = FJ_G(double, P->curve_cp) + (size_t) sl * 12;
        double t = 0, u = 0;
        const bool hitc = curve_ray<false>(cp, nullptr, 0., FJ_G(double, P->curve_width)[2 * (size_t) sl], FJ_G(double, P->curve_width)[2 * (size_t) sl + 1],
            (int) FJ_G(int8_t, P->curve_depth)[sl], RaySpace{nullptr, oo, od}, &t, &u);
        if (hitc && curve_listed_in_cell_of(P, cp, oo + t * od) && (tmin <= t && t <= rp_[6 * BLOCK])) {
          have = false; cur = TRAV_DONE; pleaf = TRAV_DONE; deep = false;     // occluded: nothing to add
        } else if (deep) {
          // the curve this lane was waiting at becomes the pending one; walk on
          deep = false;
          if (pleaf != TRAV_DONE) { pend = (pleaf & 0x7fffffffu) >> 3; pleaf = TRAV_DONE; }
          else { pend = (cur & 0x7fffffffu) >> 3; cur = (spa < ca_row1) ? TRAV_DONE : pop(spa); }
        }
      }
      CA_PH(3, m_pend);
      continue;
    }

    if (n_inner * tune.leaf_bias8_canyhit >= n_leaf * 8u) {
      // ---- inner nodes: one 64-byte node per lane; further steps without a new vote while enough lanes stay at inner nodes
      for (uint32_t step = 0;; step++) {
        const bool in_now = step == 0 ? at_inner : (cur != TRAV_DONE && !(cur & FJ_LEAF_FLAG));
        if (step > 0) {
          const unsigned n_now = (unsigned) __popcll(__ballot(in_now));
          if (step >= tune.steps_canyhit || n_now < tune.min_inner_canyhit) break;
        }
        const bool deepstk = __ballot(in_now && spa + 3u * BLOCK * 4u > ca_ovf0) != 0ull;
        if (in_now) {
          const FJ_GLOBAL fj_v4u *nd = (const FJ_GLOBAL fj_v4u *) (nodes + ((size_t) cur << 6));
          if (kCount) lc->nodes++;
          const fj_v4u w0 = nd[0], w1 = nd[1], w2 = nd[2], e = nd[3];
          const uint32_t shx = slab32_shift(s32.ix), shy = slab32_shift(s32.iy), shz = slab32_shift(s32.iz);
          const bool h0 = slab32q_test(w0.x, w0.y, w0.z, s32, shx, shy, shz, tmin32, tmax32);
          FJ_SCHED_FENCE();
          const bool h1 = slab32q_test(w0.w, w1.x, w1.y, s32, shx, shy, shz, tmin32, tmax32);
          FJ_SCHED_FENCE();
          const bool h2 = slab32q_test(w1.z, w1.w, w2.x, s32, shx, shy, shz, tmin32, tmax32) && e.z != FJ_NO_CHILD;
          FJ_SCHED_FENCE();
          const bool h3 = slab32q_test(w2.y, w2.z, w2.w, s32, shx, shy, shz, tmin32, tmax32) && e.w != FJ_NO_CHILD;
          FJ_SCHED_FENCE();
          uint32_t r0 = e.x, r1 = e.y, r2 = e.z, r3 = e.w;
          if (!h2) { r2 = r3; }
          if (!h1) { r1 = r2; r2 = r3; }
          if (!h0) { r0 = r1; r1 = r2; r2 = r3; }
          const int nh = (int) h0 + (int) h1 + (int) h2 + (int) h3;
          if (nh == 0) cur = (spa < ca_row1) ? TRAV_DONE : pop(spa);
          else {
            cur = r0;
            if (!deepstk) {
              ca_lds_u32 *top = &CA_AT(spa);
              top[0] = r1; top[BLOCK] = r2; top[2 * BLOCK] = r3;
              spa += (uint32_t) (nh - 1) * (BLOCK * 4u);
            } else {
              if (nh > 1) push(spa, r1);
              if (nh > 2) push(spa, r2);
              if (nh > 3) push(spa, r3);
            }
          }
          if (FJ_CANYHIT_POSTPONE && cur != TRAV_DONE && (cur & FJ_LEAF_FLAG) && pleaf == TRAV_DONE && spa >= ca_row1 && !deep) { pleaf = cur; cur = pop(spa); }
        }
        CA_PH(1, __ballot(in_now));
      }
    } else {
      // ---- leaves: ONE triangle, or the capsule of ONE curve piece, per lane
      if (at_leaf) {
        const bool from_p = pleaf != TRAV_DONE;        // the postponed leaf first: its slot frees
        const uint32_t lf = from_p ? pleaf : cur;
        const uint32_t first = (lf & 0x7fffffffu) >> 3;
        const uint32_t more = lf & 7u;
        const DInstEntry *E = &gents[ii];
        ca_lds_f64 *const rp_ = &s_ray[CA_TID()];
        const V3 oo = mk(rp_[0], rp_[BLOCK], rp_[2 * BLOCK]), od = mk(rp_[3 * BLOCK], rp_[4 * BLOCK], rp_[5 * BLOCK]);
        bool consumed = true;                          // the leaf at hand is done with (else: it waits, deep)
        if (in_curves) {
          // BLAS entries are pieces of curves (one per leaf, fjgpu_curve_build.cc): the ribbon test of a curve runs once per
          // (ray, instance) while its pieces are entered one after the other, whichever piece was entered.  The capsule record of
          // the slot (two 16-byte loads through the pointer in the instance's LDS entry) names the curve in its last word.
          const FJ_GLOBAL fj_v4f *caps = (const FJ_GLOBAL fj_v4f *) E->tri_verts32 + (size_t) first * 2;
          const fj_v4f c0 = caps[0], c1 = caps[1];
          const uint32_t cid = __float_as_uint(c1.w);
          if (cid != last_curve && capsule_may_hit_v(c0, c1, oo, od)) {
            last_curve = cid;
            if (kCount) lc->prims++;
            if (pend == NONE) pend = first;
            else { deep = true; consumed = false; }
          }
          if (consumed) { if (from_p) pleaf = TRAV_DONE; else cur = (spa < ca_row1) ? TRAV_DONE : pop(spa); }
        } else {
          if (kCount) lc->prims++;
          V3 v0, v1, v2;
          load_tri(E->tri_verts, E->tri_verts32, first, &v0, &v1, &v2);
          FJ_SCHED_FENCE();
          if (tri_ray_anyhit(v0, v1, v2, oo, od, tmin, rp_[6 * BLOCK])) { have = false; cur = TRAV_DONE; pleaf = TRAV_DONE; }     // occluded: nothing to add
          else if (from_p) pleaf = more ? (FJ_LEAF_FLAG | ((first + 1u) << 3) | (more - 1u)) : TRAV_DONE;
          else if (more) cur = FJ_LEAF_FLAG | ((first + 1u) << 3) | (more - 1u);
          else cur = (spa < ca_row1) ? TRAV_DONE : pop(spa);
        }
      }
      CA_PH(2, __ballot(at_leaf));
    }
  }
#ifdef FJ_PHASE_STATS
  if (lane == 0) for (int i = 0; i < 13; i++) if (cph[i]) atomicAdd(&g_cphase[i], cph[i]);
#endif
#undef CA_AT
#undef CA_OVF
#undef CA_TID
}

template <bool kCount>
__global__ void __launch_bounds__(BLOCK, FJ_CANYHIT_MINB) k_shadow_anyhit_curves(DScene S, const DShadowRay *squeue, float *s_accum,
    DCounters *cnt, TravTune tune)
{
  __shared__ alignas(16) uint32_t s_stack[FJ_STACK_LDS_CANYHIT * BLOCK + 14 * BLOCK];     // (+ the rays: 6 doubles per thread, and tmax)
  __shared__ double s_inst[InstLdsCurves::WORDS];
  InstLdsCurves::fill(S, s_inst);          // (the launcher picked this kernel because the scene fits)
  const uint32_t n = cnt->shadow_count < S.shadow_queue_cap ? cnt->shadow_count : S.shadow_queue_cap;
  LocalCounters lc = {0, 0, 0};
  traverse_anyhit_curves<kCount>(S, squeue, s_accum, tune, n, &cnt->shadow_xcd_head[0][0], s_stack, &lc, s_inst);
  if (kCount) {
    flush_counters(cnt, lc.nodes, lc.prims, lc.insts, 0, 0);
    flush_shadow_walk_counters(cnt, lc.nodes, lc.prims, lc.insts);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&cnt->squeued, (unsigned long long) n);
  }
}

#endif
