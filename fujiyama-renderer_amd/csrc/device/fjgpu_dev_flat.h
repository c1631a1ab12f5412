// fjgpu_dev_flat.h -- the closest-hit walk of scenes whose groups are FLAT (DFlat, fjgpu_types.h): one world-space culling tree per group
// over the triangles of all its instances.
// Part of the kernels translation unit: included by fjgpu_kernels.hip only (device code,
// compiled with -ffp-contract=off; see the header of that file).
#ifndef FJGPU_DEV_FLAT_H
#define FJGPU_DEV_FLAT_H
#define FJ_FLAT_NO_HIT 0xffffffffu     // "no candidate has won yet" in the walk's best_ord

// What the reference does for a ray into a group (BVHAccelerator over ObjectInstances, src/fj_bvh_accelerator.cc:164-241; ObjectInstance::
// RayIntersect, src/fj_object_instance.cc:213-243): visit the instances whose box the ray passes, in the BVH's order; in each, transform the ray
// with M^-1 (no renormalising: t is preserved), intersect the set's accelerator, keep the hit if it is strictly nearer.  The phase-scheduled
// walk (traverse_phased) does exactly that, and on C4 -- nine small instances, 256 spp of incoherent rays -- 41 % of its wave ticks are the
// TURNOVER between instances: a ray enters 1.24 instances and rejects 6.8 more by their boxes, in a per-lane loop the whole wave waits for.
// Here the group has ONE tree, built over the world-space boxes of every triangle of every instance (fjgpu_api.hip, build_flat_groups): a ray
// is fetched, walks that tree once (boxes are pure culling: a world-space box contains its triangle's image under M, padded), and a leaf names
// (instance, triangle).  The DECISIONS are the reference's, evaluated per candidate:
//   * the instance's own box (BoxRayIntersect on DInstance.wbounds, full ray range) -- once per ray and instance, remembered in two bit masks;
//   * the -0.0 rule on the object-space direction (every box test of the set's accelerator fails for such a ray);
//   * TriRayIntersect in OBJECT space, with o' = M^-1 o, d' = M^-1 d as the reference computes them, the inclusive range on the original t range;
//   * nearer wins; at exactly equal t the instance EARLIER in the group's order wins (the reference visits it first and later ones need a strictly
//     smaller t), within an instance the larger primitive id (the grid's LIFO cell lists).
// A turnover is retire + fetch + slab set-up: one per ray.  Same phases and votes as traverse_phased otherwise.
template <bool kCount, bool kOne, class Policy>
__device__ void traverse_flat(const DScene &S, Policy &pol, TravTune tune, uint32_t n, uint32_t *head, TravStack stk, LocalCounters *lc, const double *s_inst)
{
  const double *s_minv = s_inst;          // [instance][12]: M^-1 of every instance (k_trace_closest_flat fills it)
  const unsigned lane = __lane_id();
  bool head_live = true;
  uint32_t next = 0, range_end = 0;        // wave-uniform: the wave's claimed slice of the queue
  tune.grab = adaptive_grab(tune.grab, n);
  QueueClaim qc;
  qc.init(head, n, tune.grab);
  bool have = false;
  uint32_t idx = 0;
  V3 o = mk(0, 0, 0), d = o;
  Slab32 s32 = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  double tmin = 0, tmax = 0;
  // the nearest hit so far: t, the position of its instance in the group and its primitive id are what later candidates are compared with; the
  // record itself (u, v, instance) goes to the ray's hit slot whenever a candidate wins (the walk is near to far: 1.4 winners per C4 ray, and
  // the line is still in L2 when the second one comes) instead of riding along in five registers -- the loop spilled sixteen
  double best_t = DBL_MAX;
  int best_prim = -1;
  uint32_t best_ord = FJ_FLAT_NO_HIT;
  bool anyhit = false;
  // (kOne: the scene has ONE group -- node and leaf arrays are the same for every ray: scalar registers)
  const FJ_GLOBAL char *nodes = kOne ? (const FJ_GLOBAL char *) S.flats[0].nodes : nullptr;
  const FJ_GLOBAL fj_v4u *refs = kOne ? (const FJ_GLOBAL fj_v4u *) S.flats[0].refs : nullptr;      // DFlatRef records: three 16-byte words each
  const double *refbox = kOne ? S.flats[0].refbox : nullptr;
  uint32_t fl_pass = 0, fl_fail = 0;       // instances (by position in the group) whose own box and -0.0 rule this ray passed / failed
  uint32_t cur = TRAV_DONE;
  int sp = 0;
#ifdef FJ_PHASE_STATS
  // wave-level clock ticks, executions and lanes per phase (the slots of traverse_phased: debug_phase_stats prints them)
  unsigned long long pcyc[3] = {0, 0, 0}, pex[3] = {0, 0, 0}, pln[3] = {0, 0, 0}, pc_prev = __builtin_readcyclecounter();
#define FJ_FCYC(k, mask) do { const unsigned long long c_now = __builtin_readcyclecounter(); pcyc[k] += c_now - pc_prev; pc_prev = c_now; pex[k]++; pln[k] += __popcll(mask); } while (0)
#else
#define FJ_FCYC(k, mask) do { } while (0)
#endif

  for (;;) {
    const bool fin = cur == TRAV_DONE;
    const bool at_leaf = !fin && (cur & FJ_LEAF_FLAG);
    const bool at_inner = !fin && !at_leaf;
    const unsigned n_leaf = (unsigned) __popcll(__ballot(at_leaf)), n_inner = (unsigned) __popcll(__ballot(at_inner));
    const bool can_fetch = head_live || next < range_end;
    const unsigned long long m_turn = __ballot(fin && (have || can_fetch));

    if ((unsigned) __popcll(m_turn) >= tune.refill_flat || (n_inner == 0 && n_leaf == 0)) {
      if (m_turn == 0ull) {
#ifdef FJ_PHASE_STATS
        if (lane == 0) {
          for (int k = 0; k < 3; k++) { atomicAdd(&g_phase[1 + 2 * k], pcyc[k]); atomicAdd(&g_phase[2 + 2 * k], pln[k]); atomicAdd(&g_phase[7 + k], pex[k]); }
          atomicAdd(&g_phase[10], pex[0]); atomicAdd(&g_phase[11], pex[0]);
        }
#endif
        break;
      }
      // ---- turnover: retire, fetch, set up the walk
      if (next >= range_end && head_live) head_live = qc.claim(lane, &next, &range_end);
      if (fin && have) { if (best_ord == FJ_FLAT_NO_HIT) pol.finish_miss(idx); have = false; }
      const bool fetch = fin;
      const unsigned long long m_fetch = __ballot(fetch);
      bool fresh = false;
      if (fetch) {
        const uint32_t my = next + __builtin_amdgcn_mbcnt_hi((uint32_t) (m_fetch >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m_fetch, 0u));
        if (my < range_end) {
          RayIn r;
          r.o = r.d = mk(0, 0, 1); r.tmin = r.tmax = r.time = 0; r.group = 0; r.anyhit = false;
          have = pol.fetch(my, &r);
          idx = my;
          o = r.o; d = r.d; tmin = r.tmin; tmax = r.tmax; anyhit = r.anyhit;
          best_t = DBL_MAX; best_prim = -1; best_ord = FJ_FLAT_NO_HIT;
          fl_pass = fl_fail = 0;
          sp = 0;
          fresh = have;
          if (fresh) {
            const DFlat *F = &S.flats[kOne ? 0 : r.group];
            if (!kOne) { nodes = (const FJ_GLOBAL char *) F->nodes; refs = (const FJ_GLOBAL fj_v4u *) F->refs; refbox = F->refbox; }
            // BoxRayIntersect's -0.0 quirk: every box test of the reference fails for such a ray -- it hits nothing
            if (!has_negative_zero(d) && F->n_prims > 0) {
              const V3 winv = mk(filter_rcp(d.x), filter_rcp(d.y), filter_rcp(d.z));
              s32 = slab32q_setup(o, winv, F->qorigin, F->qcell);
              cur = F->root;
            }
          }
        }
      }
      next += (uint32_t) __popcll(m_fetch);
      if (next > range_end) next = range_end;
      FJ_FCYC(0, m_turn);
      continue;
    }

    if (n_inner * tune.leaf_bias8_flat >= n_leaf * 8u) {
      // ---- inner nodes; further steps without a new vote while enough lanes stay at inner nodes
      for (uint32_t step = 0;; step++) {
        const bool in_now = step == 0 ? at_inner : (cur != TRAV_DONE && !(cur & FJ_LEAF_FLAG));
        if (step > 0 && (step >= tune.steps_flat || (unsigned) __popcll(__ballot(in_now)) < tune.min_inner_flat)) break;
        if (in_now) {
          if (kCount) lc->nodes++;
          const double tf2 = anyhit ? tmax : fmin(tmax, best_t);
          const float tmin32 = f32_below(tmin), tmax32 = f32_above(tf2);
          float t0, t1, t2, t3;
          const FJ_GLOBAL fj_v4u *nq = (const FJ_GLOBAL fj_v4u *) (nodes + ((size_t) cur << 6));
          const fj_v4u w0 = nq[0], w1 = nq[1], w2 = nq[2], e = nq[3];
          const uint32_t shx = slab32_shift(s32.x.i), shy = slab32_shift(s32.y.i), shz = slab32_shift(s32.z.i);
          const bool h0 = slab32q_test(w0.x, w0.y, w0.z, s32, shx, shy, shz, tmin32, tmax32, &t0);
          const bool h1 = slab32q_test(w0.w, w1.x, w1.y, s32, shx, shy, shz, tmin32, tmax32, &t1);
          const bool h2 = slab32q_test(w1.z, w1.w, w2.x, s32, shx, shy, shz, tmin32, tmax32, &t2) && e.z != FJ_NO_CHILD;
          const bool h3 = slab32q_test(w2.y, w2.z, w2.w, s32, shx, shy, shz, tmin32, tmax32, &t3) && e.w != FJ_NO_CHILD;
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
        FJ_FCYC(1, __ballot(in_now));
      }
    } else {
      // ---- leaves: ONE candidate (instance, triangle) per lane
      if (at_leaf) {
        const uint32_t first = (cur & 0x7fffffffu) >> 3;
        const uint32_t more = cur & 7u;
        bool stop = false;
        const fj_v4u q0 = refs[3 * (size_t) first], q1 = refs[3 * (size_t) first + 1], q2 = refs[3 * (size_t) first + 2];
        const uint32_t ref_io = q2.y;
        const uint32_t inst = ref_io >> 8, ord = ref_io & 255u, bit = 1u << ord;
        if (!(fl_fail & bit)) {
          const double *Mi = s_minv + 12 * inst;
          const V3 oo = xpoint(Mi, o), od = xvector(Mi, d);
          bool ok = true;
          if (!(fl_pass & bit)) {
            // first candidate of this instance for this ray: the tests that decide whether the ray reaches the instance at all
            if (kCount) lc->insts++;
            const V3 winv = mk(filter_rcp(d.x), filter_rcp(d.y), filter_rcp(d.z));
// (`refbox` comes out of a record: generic to the compiler, FLAT loads waited for with both counters; here through the global address space: C4 walk 367 -> 364 ms)
            ok = box_ray_ref_fast(FJ_G(double, refbox) + 6 * ord, o, d, winv, plain_dir(d), tmin, tmax) && !has_negative_zero(od);
            if (ok) fl_pass |= bit; else fl_fail |= bit;
          }
          if (ok) {
            double t, u = 0, v = 0;
            if (kCount) lc->prims++;
            const V3 v0 = mk((double) __uint_as_float(q0.x), (double) __uint_as_float(q0.y), (double) __uint_as_float(q0.z));
            const V3 v1 = mk((double) __uint_as_float(q0.w), (double) __uint_as_float(q1.x), (double) __uint_as_float(q1.y));
            const V3 v2 = mk((double) __uint_as_float(q1.z), (double) __uint_as_float(q1.w), (double) __uint_as_float(q2.x));
            if (tri_ray_early(v0, v1, v2, oo, od, &t, &u, &v) && tmin <= t && t <= tmax) {
              const int pid = (int) q2.z;
              if (t < best_t || (t == best_t && best_ord != FJ_FLAT_NO_HIT && (ord < best_ord || (ord == best_ord && pid > best_prim)))) {
                best_t = t; best_prim = pid; best_ord = ord;
                Best b;
                b.t = t; b.u = u; b.v = v; b.inst = (int) inst; b.prim = pid;
                pol.finish(idx, b);
                stop = anyhit;
              }
            }
          }
        }
        if (stop) cur = TRAV_DONE;           // (an any-hit ray: retired with its hit at the next turnover)
        else if (more) cur = FJ_LEAF_FLAG | ((first + 1u) << 3) | (more - 1u);
        else cur = (sp == 0) ? TRAV_DONE : stk.pop(sp);
      }
      FJ_FCYC(2, __ballot(at_leaf));
    }
  }
#undef FJ_FCYC
}

#ifndef FJ_FLAT_MINB
#define FJ_FLAT_MINB 4
#endif
#ifndef FJ_STACK_LDS_FLAT
#define FJ_STACK_LDS_FLAT FJ_STACK_LDS
#endif
static_assert(FJ_STACK_LDS_FLAT >= FJ_STACK_LDS_MIN, "the overflow area is sized for FJ_STACK_LDS_MIN entries in LDS");
template <bool kCount, bool kOne>
__global__ void __launch_bounds__(BLOCK, FJ_FLAT_MINB) k_trace_closest_flat(DScene S, const DRay *rays, const DPath *paths,
    DHit *hits, uint32_t n, DCounters *cnt, TravTune tune)
{
  if (S.trace_n_dev) { const uint32_t nd_ = *S.trace_n_dev; n = nd_ < n ? nd_ : n; }       // (a speculative launch: fjgpu_api.hip)
  __shared__ uint32_t s_stack[FJ_STACK_LDS_FLAT * BLOCK];
  __shared__ double s_inst[FJ_FLAT_LDS_INSTS * 12];
  // M^-1 of every instance (the launcher picked this kernel because the scene has at most FJ_FLAT_LDS_INSTS of them)
  for (uint32_t w = threadIdx.x; w < (uint32_t) S.n_instances * 12u; w += BLOCK) s_inst[w] = S.inst_entries[w / 12u].Minv[w % 12u];
  __syncthreads();
  ClosestPolicy pol;
  pol.S = &S; pol.rays = rays; pol.paths = paths; pol.hits = hits; pol.default_group = S.target_group;
  LocalCounters lc = {0, 0, 0};
  traverse_flat<kCount, kOne>(S, pol, tune, n, &cnt->trace_xcd_head[0][0], make_stack(s_stack, S.stack_overflow, nullptr, FJ_STACK_LDS_FLAT), &lc, s_inst);
  if (kCount) {
    flush_counters(cnt, lc.nodes, lc.prims, lc.insts, 0, 0);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&cnt->traced, (unsigned long long) n);
  }
}

#endif
