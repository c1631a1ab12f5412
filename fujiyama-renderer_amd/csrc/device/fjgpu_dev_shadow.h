// fjgpu_dev_shadow.h -- SlIlluminance: light loop (cull), general shadow traversal, lean any-hit walk.
// Part of the kernels translation unit: included by fjgpu_kernels.hip only (device code,
// compiled with -ffp-contract=off; see the header of that file).
#ifndef FJGPU_DEV_SHADOW_H
#define FJGPU_DEV_SHADOW_H

// ------------------------------------------------------------------- k_shadow
// SlIlluminance (src/fj_shading.cc:296-359) in two wavefront stages.
//
// k_shadow_cull: every (light record, light sample) pair.  One record per lane,
// every lane loops over the light samples (wave-uniform index).  A pair is tested
// against the world AABBs of the shadow group's instances: if it misses all of them
// the light is unoccluded and Kd * Cl goes into the lane's per-record sum (W * sum is
// added to the sample at the end); otherwise the ray is appended -- ballot + prefix
// count into a chunk the wave reserved -- to the compact shadow-ray queue.
//
// k_shadow_trace: the compact queue only, so every lane of a wave is
// traversing (no lanes idling while a neighbour walks the BLAS).  Adds
// c * (1 - Os_occluder) to the sample, or c when the ray reaches the light.
#ifndef SQ_CHUNK
#define SQ_CHUNK 512u          // shadow-queue slots a wave reserves per global atomic
#endif
#ifndef JQ_CHUNK
#define JQ_CHUNK 64u           // join slots a wave reserves per global atomic (split shadow rays; >= 64: one round may need a slot per lane); fjgpu_api.hip sizes the slack
#endif
#define SQ_INVALID 0xffffffffu // DShadowRay.sample of a padding slot
// queue entry i in either record format (DScene.compact_squeue is wave-uniform: a scalar branch)
__device__ __forceinline__ const DShadowRayC *sq_c(const DShadowRay *squeue, uint32_t i) { return (const DShadowRayC *) squeue + i; }
#define SQ_FIELD(S, squeue, i, f) ((S).compact_squeue ? sq_c(squeue, i)->f : (squeue)[i].f)
__device__ __forceinline__ const float *sq_colour(const DScene &S, const DShadowRay *squeue, uint32_t i) { return S.compact_squeue ? sq_c(squeue, i)->c : squeue[i].c; }
// the ray of entry i: stored, or rebuilt with the light loop's statements (k_shadow_cull: Ln, distance)
__device__ __forceinline__ void sq_ray(const DScene &S, const DShadowRay *squeue, uint32_t i, V3 *o, V3 *d, double *tmax)
{
  if (!S.compact_squeue) {
    const DShadowRay *q = &squeue[i];
    *o = mk(q->o[0], q->o[1], q->o[2]); *d = mk(q->d[0], q->d[1], q->d[2]); *tmax = q->tmax;
    return;
  }
  const DShadowRayC *q = sq_c(squeue, i);
  const V3 Ps = mk(q->o[0], q->o[1], q->o[2]);
  const DLightSample *LS = &S.light_samples[q->light];
  const V3 Pl = mk(LS->P[0], LS->P[1], LS->P[2]);
  V3 Ln = mk(Pl.x - Ps.x, Pl.y - Ps.y, Pl.z - Ps.z);
  const double distance = sqrt(dot(Ln, Ln));
  if (distance > 0) {
    const double inv = 1. / distance;
    Ln = mk(Ln.x * inv, Ln.y * inv, Ln.z * inv);
  }
  *o = Ps; *d = Ln; *tmax = distance;
}

// a light sample through the constant address space (wave-uniform address: scalar loads)
__device__ __forceinline__ DLightSample ld_light_sample_k(const DLightSample *p)
{
  typedef const __attribute__((address_space(4))) double kd;
  typedef const __attribute__((address_space(4))) float kf;
  typedef const __attribute__((address_space(4))) int32_t ki;
  kd *q = (kd *) (uintptr_t) p;
  kf *f = (kf *) (uintptr_t) (p->Cl);
  ki *i = (ki *) (uintptr_t) (&p->light);
  DLightSample r;
  r.P[0] = q[0]; r.P[1] = q[1]; r.P[2] = q[2];
  r.Cl[0] = f[0]; r.Cl[1] = f[1]; r.Cl[2] = f[2];
  r.light = i[0]; r.type = i[1]; r.ordinal = i[2];
  return r;
}

#ifndef FJ_CULL_MINB
#define FJ_CULL_MINB 1
#endif
#ifndef FJ_CULL_CLAIMS_PER_WAVE
#define FJ_CULL_CLAIMS_PER_WAVE 16u  // the light loop's record claims: a launch's records over (waves x this), in [64, 1024].  (4 until round 5: the waves of a launch
                                     // lived 69-87 % of it, scripts/kernel_pmc.py; 4 / 8 / 16 / 32: C3 113.9 / 113.5 / 113.8 / 117.4 ms, C2 98.0 -> 92.4 at 16, a rank's
                                     // share of C3 18.8 -> 18.5, C6 unchanged)
#endif
#ifndef FJ_CULL_MINB_PLAIN
#define FJ_CULL_MINB_PLAIN 3      // point lights, no hair, rays split per candidate instance: 164 VGPRs as written (capped to 128 it spills 21-24)
#endif
#ifndef FJ_CULL_MINB_WHOLE
#define FJ_CULL_MINB_WHOLE 4      // ... whole rays (single-instance shadow groups: C3, C6): 137 VGPRs as written, 127 without a spill when capped: C3 light loop 18.9 -> 18.5 ms
#endif
// kNodesLds: every block keeps the scene's threaded instance nodes (DScene.group_nodes, at most FJ_CULL_LDS_NODES) in LDS -- the
// candidate search of a (point, light) pair is a chain of dependent node reads (launch_shadow_cull picks it where the scene fits)
#ifndef FJ_CULL_LDS_NODES
#define FJ_CULL_LDS_NODES 292      // 16 352 bytes
#endif
template <bool kHair, bool kArea, bool kSplit, bool kNodesLds = false>
__global__ void __launch_bounds__(BLOCK, (kHair || kArea) ? FJ_CULL_MINB : (kSplit ? FJ_CULL_MINB_PLAIN : FJ_CULL_MINB_WHOLE)) k_shadow_cull(DScene S, ShadowParams sp, const DLightRec *lrecs,
    uint32_t rec_begin, uint32_t rec_end, float *s_accum, DShadowRay *squeue, DCounters *cnt, int count_events)
{
  __shared__ double s_nodes[kNodesLds ? FJ_CULL_LDS_NODES * 7 : 1];
  if (kNodesLds) {
    const unsigned long long *src = (const unsigned long long *) S.group_nodes;
    for (uint32_t w = threadIdx.x; w < (uint32_t) S.n_group_nodes * 7u; w += BLOCK) ((unsigned long long *) s_nodes)[w] = src[w];
    __syncthreads();
  }
  const DTNode *gnodes = kNodesLds ? (const DTNode *) s_nodes : S.group_nodes;
  unsigned long long c_insts = 0, c_shadow = 0;
  const unsigned lane = __lane_id();
  const uint32_t n = rec_end - rec_begin;
  // waves claim contiguous runs of records (so the rays a wave emits -- and the shadow-queue
  // chunks it fills -- stay spatially coherent) from a global head: equal static slices left
  // the kernel waiting for the waves whose records face the lights, and assumed that every
  // block of the grid is resident
  const uint32_t n_waves = (gridDim.x * BLOCK) >> 6;
  uint32_t claim = (n / (n_waves * FJ_CULL_CLAIMS_PER_WAVE)) & ~63u;
  claim = claim < 64u ? 64u : (claim > 1024u ? 1024u : claim);
  // queue space is reserved SQ_CHUNK slots at a time: one atomic per 512 rays
  // instead of one per wave iteration (a single-address atomic per iteration
  // serialised the whole kernel in L2)
  uint32_t chunk_base = 0, chunk_used = SQ_CHUNK;   // wave-uniform; "used == CHUNK" = no chunk yet
  uint32_t jchunk_base = 0, jchunk_used = JQ_CHUNK; // the same for the join slots of split rays

  uint32_t slice_end = 0, r0 = 0;
  for (;; r0 += 64u) {
    if (r0 >= slice_end) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&cnt->cull_head, claim);
      r0 = __shfl(base, 0);
      if (r0 >= n) break;
      slice_end = r0 + claim < n ? r0 + claim : n;
    }
    const uint32_t rec = rec_begin + r0 + lane;           // one record per lane
    const bool active = (r0 + lane) < slice_end;

    float sum[3] = {0.f, 0.f, 0.f};
    uint32_t r_sample = 0;
    float W[3] = {0.f, 0.f, 0.f};
    const uint32_t nl = (uint32_t) S.n_light_samples;
    DLightRec R;
    DLightHair H;
    for (int q = 0; q < 6; q++) H.aux[q] = 0;
    H.Cd[0] = H.Cd[1] = H.Cd[2] = 0.f;
    R.uid = R.key = 0; R.kind = 0;
    XS xs = {0, 0, 0, 0};
    V3 Ps = mk(0, 0, 0), axis = Ps, nml_axis = Ps;
    int g_first = 0, g_count = 0;
    bool g_single = false;
    int g_inst = 0;                        // the instance of a single-instance group
    const double *g_sbounds = nullptr;     // stays a global-memory pointer (a by-value DGroup lands in scratch)
    double sb[6] = {0, 0, 0, 0, 0, 0};
    double cos_limit = 0;
    if (active) {
      R = lrecs[rec];
      if (kHair && (R.kind & 1)) H = S.lrec_hair[rec];
      Ps = mk(R.P[0], R.P[1], R.P[2]);
      axis = mk(R.N[0], R.N[1], R.N[2]);
      nml_axis = normalize(axis);
      r_sample = R.sample;
      W[0] = R.W[0]; W[1] = R.W[1]; W[2] = R.W[2];
      cos_limit = (!kHair || (R.kind & 1) == 0) ? sp.cos_half_pi : sp.cos_pi;
      g_first = S.groups[R.group].first; g_count = S.groups[R.group].count;
      g_single = S.groups[R.group].n_instances == 1;
      g_sbounds = S.groups[R.group].sbounds;
      // a single-instance group's box is the same for every light of the record: read it once
      // (per pair it cost two dependent loads -- node, then box -- before any arithmetic)
      if (g_single) { for (int q = 0; q < 6; q++) sb[q] = g_sbounds[q]; g_inst = gnodes[g_first].inst; }
    }
    // A surface point well inside that box (every point of the occluder itself, but for its outermost
    // 2e-4) needs no box test per light: the unit-length ray leaves the box at t >= 2e-4 > tmin and
    // entered it at t <= 0 < distance, which is all BoxRayIntersect asks for (src/fj_box.cc:73-138).
    const bool deep_inside = active && g_single &&
        Ps.x > sb[0] + 2e-4 && Ps.x < sb[3] - 2e-4 && Ps.y > sb[1] + 2e-4 && Ps.y < sb[4] - 2e-4 && Ps.z > sb[2] + 2e-4 && Ps.z < sb[5] - 2e-4;
    // (Tried in round 3 and dropped: per-lane LISTS of the lights that are not surely behind the surface -- a cheap first
    // pass with the light index wave-uniform, then every lane walks its own list, so that the expensive part runs
    // max-list-length times instead of once per light.  Only 36 % of C3's pairs are lit, yet the loop got slower, 22.8 ->
    // 27.0 ms (C6 56.7 -> 59.2): per-lane light records are vector loads, the lists of a wave's lanes differ enough that the
    // longest is ~3/4 of all lights, and both loops in one kernel cost 4 spilled registers.)
    // the light index is wave-uniform: the sample's 72 bytes come through the scalar cache into
    // SGPRs instead of 64 identical vector loads
    for (uint32_t l = 0; l < nl; l++) {
      bool emit = false;
      DShadowRay q;
      // split mode (DScene.shadow_join): a ray into a group of several instances is queued once per instance
      // whose box it passes; the lane enumerates them over the rounds below
      bool pending = false;
      int tcur = 0;
      uint32_t ncand = 0, jslot1 = 0, nb = 0, be = 0;
      int cb0 = 0, cb1 = 0, cb2 = 0, cb3 = 0;
      V3 winv_s = mk(0, 0, 0), Ln_s = winv_s;
      bool plain_s = false;
      double dist_s = 0;
      float k_s[3] = {0.f, 0.f, 0.f};
      if (active) {
        // (the table is written once at scene creation: read through the constant address space the wave-uniform record comes through the scalar
        // cache into SGPRs -- as a generic pointer it was three vector loads of 64 identical lanes, waited for at the head of every iteration:
        // C3 light loop 19.8 -> 18.9 ms, C6 47.9 -> 46.5, C5 57.7 -> 55.3)
        // (loaded one iteration ahead: 18.8 -> 19.8 ms, dropped)
        const DLightSample LS = ld_light_sample_k(S.light_samples + l);
        V3 Pl = mk(LS.P[0], LS.P[1], LS.P[2]);
        float Cl[3] = {LS.Cl[0], LS.Cl[1], LS.Cl[2]};
        if (kArea && (LS.type == FJ_GRID_LIGHT || LS.type == FJ_SPHERE_LIGHT)) {
          // RectangleLight / SphereLight::get_samples + illuminate with the per-event stream
          const DAreaLight *A = &S.area_lights[LS.light];
          if (LS.ordinal == 0) xs = area_stream(R.uid, R.key, LS.light);
          V3 Nl;
          if (LS.type == FJ_GRID_LIGHT) {
            const double px = xs.f01() - .5;
            const double pz = xs.f01() - .5;
            Pl = xpoint(A->M, mk(px, 0, pz));
            Nl = mk(A->N[0], A->N[1], A->N[2]);
          } else {
            V3 o;
            double dd;
            for (;;) {                                      // XorShift::HollowSphereRand
              o.x = 2 * xs.f01() - 1;
              o.y = 2 * xs.f01() - 1;
              o.z = 2 * xs.f01() - 1;
              dd = dot(o, o);
              if (dd > 0 && dd <= 1) break;
            }
            const double inv = 1. / sqrt(dd);
            const V3 p = mk(o.x * inv, o.y * inv, o.z * inv);
            Pl = xpoint(A->M, p);
            Nl = normalize(xvector(A->M, p));
          }
          const V3 Lq = normalize(mk(Ps.x - Pl.x, Ps.y - Pl.y, Ps.z - Pl.z));
          double dl = dot(Lq, Nl);
          float k;
          if (LS.type == FJ_GRID_LIGHT) {
            dl = A->double_sided ? fabs(dl) : (dl > 0. ? dl : 0.);
            k = (float) (dl * (double) A->sample_intensity);
          } else k = dl > 0 ? A->sample_intensity : 0.f;
          Cl[0] = k * A->color[0]; Cl[1] = k * A->color[1]; Cl[2] = k * A->color[2];
        }
        V3 Ln = mk(Pl.x - Ps.x, Pl.y - Ps.y, Pl.z - Ps.z);
        // lights clearly behind the surface (half of all pairs) skip the f64 sqrt and division:
        // when the unnormalised dot product is negative by 1e-12 of its terms' magnitude, the
        // normalised one (each factor rounded to 2^-53) is negative too, hence below cos(PI/2.)
        const double behind = nml_axis.x * Ln.x + nml_axis.y * Ln.y + nml_axis.z * Ln.z;
        const double terms = fabs(nml_axis.x * Ln.x) + fabs(nml_axis.y * Ln.y) + fabs(nml_axis.z * Ln.z);
        const bool surely_behind = cos_limit >= 0 && behind < -1e-12 * terms;
        double distance = 0, cosangle = -1;
        if (!surely_behind) {
          distance = sqrt(dot(Ln, Ln));
          if (distance > 0) {
            const double inv = 1. / distance;
            Ln = mk(Ln.x * inv, Ln.y * inv, Ln.z * inv);
          }
          cosangle = dot(nml_axis, Ln);
        }
        const bool lit = !surely_behind && !(cosangle < cos_limit) && !(Cl[0] < .0001 && Cl[1] < .0001 && Cl[2] < .0001);
        if (lit) {
          float k[3] = {0.f, 0.f, 0.f};
          if (!kHair || (R.kind & 1) == 0) {   // plastic_shader.cc:131-137
            float Kd = (float) dot(axis, Ln);
            Kd = (float) (Kd > 0 ? (double) Kd : 0.);
            k[0] = Kd * Cl[0]; k[1] = Kd * Cl[1]; k[2] = Kd * Cl[2];
          } else {                       // hair_shader.cc:184-206 (the plugin's sqrt / pow are the C
                                         // library's double versions on float arguments)
            const V3 tangent = mk(H.aux[0], H.aux[1], H.aux[2]);
            const V3 Iv = mk(H.aux[3], H.aux[4], H.aux[5]);
            const float TL = (float) dot(tangent, Ln);
            const float diff = (float) sqrt((double) (1 - TL * TL));
            const float roughness = .05f;
            const float TI = (float) dot(tangent, Iv);
            float spec = (float) (sqrt((double) (1 - TL * TL)) * sqrt((double) (1 - TI * TI)) + (double) (TL * TI));
            // pow(spec, 1 / roughness) with the plugin's constant roughness .05f: 1 / .05f is 20 exactly, and the twentieth power by five
            // multiplications differs from the C library's pow by a few ulp of a double before the result is rounded to float (the library
            // routine was most of this instantiation's registers and a hundred instructions per (point, light) pair)
            static_assert(1 / .05f == 20.f, "the exponent of the hair shader's specular term");
            { const double s1 = (double) spec, s2 = s1 * s1, s4 = s2 * s2, s5 = s4 * s1, s10 = s5 * s5; spec = (float) (s10 * s10); }
            k[0] = (H.Cd[0] * diff + spec) * Cl[0];
            k[1] = (H.Cd[1] * diff + spec) * Cl[1];
            k[2] = (H.Cd[2] * diff + spec) * Cl[2];
          }
          bool maybe_occluded = false;
          if (sp.cast_shadow) {
            c_shadow++;
            // group bounds test + leaf bounds of the instance BVH, as culling
            if (deep_inside) maybe_occluded = !has_negative_zero(Ln);
            else if (!has_negative_zero(Ln)) {
              const V3 winv = mk(filter_rcp(Ln.x), filter_rcp(Ln.y), filter_rcp(Ln.z));
              const bool plain = plain_dir(Ln);
              if (g_single) {
                maybe_occluded = box_ray_ref_fast(sb, Ps, Ln, winv, plain, .0001, distance);
                if (!maybe_occluded) c_insts++;
              }
              if (kSplit && !g_single && sp.join_capacity) {
                pending = true; tcur = g_first; winv_s = winv; plain_s = plain; Ln_s = Ln; dist_s = distance;
                k_s[0] = k[0]; k_s[1] = k[1]; k_s[2] = k[2];
              }
              for (int ti = g_first; !g_single && !pending && ti < g_first + g_count;) {      // threaded instance BVH (DTNode)
                const DTNode *tn_ = &gnodes[ti];
                if (tn_->inst < 0) {
                  double tq;
                  ti = slab(tn_->box, tn_->box + 3, Ps, winv, .0001, distance, &tq) ? ti + 1 : tn_->skip;
                  continue;
                }
                ti++;
                if (box_ray_ref_fast(tn_->box, Ps, Ln, winv, plain, .0001, distance)) { maybe_occluded = true; break; }
                c_insts++;
              }
            }
          }
          if (maybe_occluded) {
            emit = true;
            q.o[0] = Ps.x; q.o[1] = Ps.y; q.o[2] = Ps.z;
            q.d[0] = Ln.x; q.d[1] = Ln.y; q.d[2] = Ln.z;
            q.tmax = distance;
            q.c[0] = W[0] * k[0]; q.c[1] = W[1] * k[1]; q.c[2] = W[2] * k[2];
            // (lean any-hit walk: a single-instance group's only candidate is settled here)
            // (tindex: the sample's slot in the time table for the general walk; in split mode the join slot, 0 = none)
            q.sample = r_sample; q.group = (sp.pre_resolve && g_single) ? ~g_inst : R.group; q.tindex = (kSplit && sp.join_capacity) ? 0u : ((uint32_t) R.kind >> 1);
          } else if (!kSplit || !pending) {
            sum[0] += k[0]; sum[1] += k[1]; sum[2] += k[2];
          }
        }
      }
      for (;;) {
      // ---- split mode: candidates are collected four at a time in ONE pass over the group's instance level (every
      // lane walks the whole level, so the walk is uniform across the wave), then queued one per round
      bool need_slot = false;
      if (kSplit && pending && be == nb) {
        nb = 0; be = 0;
        while (tcur < g_first + g_count && nb < 4u) {        // threaded instance BVH (DTNode), resumed where it stopped
          const DTNode *tn_ = &gnodes[tcur];
          if (tn_->inst < 0) {
            double tq;
            tcur = slab(tn_->box, tn_->box + 3, Ps, winv_s, .0001, dist_s, &tq) ? tcur + 1 : tn_->skip;
            continue;
          }
          tcur++;
          if (box_ray_ref_fast(tn_->box, Ps, Ln_s, winv_s, plain_s, .0001, dist_s)) {
            if (nb == 0u) cb0 = tn_->inst; else if (nb == 1u) cb1 = tn_->inst; else if (nb == 2u) cb2 = tn_->inst; else cb3 = tn_->inst;
            nb++;
          } else c_insts++;
        }
        if (nb == 0u) {
          pending = false;
          if (ncand == 0u) { sum[0] += k_s[0]; sum[1] += k_s[1]; sum[2] += k_s[2]; }     // no instance box in the way
        } else {
          need_slot = jslot1 == 0u && ncand + nb >= 2u;
          ncand += nb;
        }
      }
      // a second candidate: the ray gets a join slot (one atomic per wave) before any of its entries is written
      const unsigned long long m_slot = kSplit ? __ballot(need_slot) : 0ull;
      if (kSplit && m_slot) {
        // join slots are reserved JQ_CHUNK at a time like the queue's: one atomic per wave and ROUND on the one counter
        // serialised in L2 (slots a wave reserves and does not use are never referenced)
        const uint32_t n_slot = (uint32_t) __popcll(m_slot);
        if (jchunk_used + n_slot > JQ_CHUNK) {
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(&cnt->join_count, JQ_CHUNK);
          jchunk_base = __shfl(base, 0);
          jchunk_used = 0;
        }
        if (need_slot) {
          jslot1 = jchunk_base + jchunk_used + (uint32_t) __popcll(m_slot & ((1ull << lane) - 1ull)) + 1u;
          if (jslot1 > sp.join_capacity) { cnt->overflow = 1; jslot1 = 1; }
        }
        jchunk_used += n_slot;
      }
      if (kSplit && pending && be < nb) {
        const int cand = be == 0u ? cb0 : (be == 1u ? cb1 : (be == 2u ? cb2 : cb3));
        be++;
        emit = true;
        q.o[0] = Ps.x; q.o[1] = Ps.y; q.o[2] = Ps.z;
        q.d[0] = Ln_s.x; q.d[1] = Ln_s.y; q.d[2] = Ln_s.z;
        q.tmax = dist_s;
        q.c[0] = W[0] * k_s[0]; q.c[1] = W[1] * k_s[1]; q.c[2] = W[2] * k_s[2];
        q.sample = r_sample; q.group = ~cand; q.tindex = jslot1;
        if (be == nb && tcur >= g_first + g_count) pending = false;
      }
      // ---- compaction into the wave's current chunk (ballot + prefix popcount)
      const unsigned long long mask = __ballot(emit);
      const uint32_t need = (uint32_t) __popcll(mask);
      if (need) {
        if (chunk_used + need > SQ_CHUNK) {
          // retire the chunk: mark its unused tail as padding, reserve a new one
          if (chunk_used < SQ_CHUNK)
            for (uint32_t k = chunk_used + lane; k < SQ_CHUNK; k += 64)
              if (chunk_base + k < sp.queue_capacity) { if (sp.compact) ((DShadowRayC *) squeue)[chunk_base + k].sample = SQ_INVALID; else squeue[chunk_base + k].sample = SQ_INVALID; }
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(&cnt->shadow_count, SQ_CHUNK);
          chunk_base = __shfl(base, 0);
          chunk_used = 0;
        }
        if (emit) {
          const uint32_t slot = chunk_base + chunk_used + (uint32_t) __popcll(mask & ((1ull << lane) - 1ull));
          if (slot >= sp.queue_capacity) cnt->overflow = 1;
          else if (sp.compact) {
            DShadowRayC qc;
            qc.o[0] = q.o[0]; qc.o[1] = q.o[1]; qc.o[2] = q.o[2];
            qc.c[0] = q.c[0]; qc.c[1] = q.c[1]; qc.c[2] = q.c[2];
            qc.sample = q.sample; qc.group = q.group; qc.tindex = q.tindex; qc.light = l; qc.pad = 0;
            ((DShadowRayC *) squeue)[slot] = qc;
          }
          else squeue[slot] = q;
        }
        chunk_used += need;
      }
      emit = false;
      if (!kSplit || __ballot(pending) == 0ull) break;
      }
      if (kSplit && ncand >= 2) S.shadow_join[jslot1 - 1u] = ncand << 16;
    }
    if (active) {
      float *acc = s_accum + 4 * (size_t) r_sample;
      const float r0v = W[0] * sum[0], r1v = W[1] * sum[1], r2v = W[2] * sum[2];
      if (r0v != 0.f) atomicAdd(acc + 0, r0v);
      if (r1v != 0.f) atomicAdd(acc + 1, r1v);
      if (r2v != 0.f) atomicAdd(acc + 2, r2v);
    }
  }
  // pad the tail of the last chunk
  if (chunk_used < SQ_CHUNK)
    for (uint32_t k = chunk_used + lane; k < SQ_CHUNK; k += 64)
      if (chunk_base + k < sp.queue_capacity) { if (sp.compact) ((DShadowRayC *) squeue)[chunk_base + k].sample = SQ_INVALID; else squeue[chunk_base + k].sample = SQ_INVALID; }
  flush_counters(cnt, 0, 0, count_events ? c_insts : 0, count_events ? c_shadow : 0, c_shadow);
}

struct ShadowPolicy {
  const DScene *S;
  const DShadowRay *squeue;
  float *s_accum;
  __device__ bool fetch(uint32_t i, RayIn *r) const
  {
    const DShadowRay q = squeue[i];
    if (q.sample == SQ_INVALID) return false;      // padding slot of a partially filled chunk
    r->o = mk(q.o[0], q.o[1], q.o[2]); r->d = mk(q.d[0], q.d[1], q.d[2]);
    r->tmin = .0001; r->tmax = q.tmax;
    r->time = S->has_motion ? sample_time(*S, q.tindex) : 0.;
    r->group = q.group;
    r->anyhit = S->groups[q.group].all_opaque != 0;
    return true;
  }
  __device__ void finish(uint32_t i, const Best &b) const
  {
    float ac = 1.f;
    if (b.inst >= 0) {
      // the occluder's shader runs in shadow context and only its Os is used
      // (src/fj_shading.cc:338-355,548-569): opacity for plastic, 1 otherwise
      float Os = 1.f;
      const DInstance *I = &S->instances[b.inst];
      const DPrimSet *P = &S->primsets[I->primset];
      const int sg = (P->face_group && b.prim >= 0) ? P->face_group[b.prim] : 0;
      int sid;
      if (sg < 0 || sg >= I->n_shaders) sid = I->shaders[0];
      else { sid = I->shaders[sg]; if (sid < 0) sid = I->shaders[0]; }
      if (sid >= 0 && S->shaders[sid].type == FJ_SHADER_PLASTIC) Os = S->shaders[sid].opacity;
      Os = (float) clampd(Os, 0, 1);
      ac = 1 - Os;
    }
    if (ac == 0.f) return;
    const DShadowRay *q = &squeue[i];
    const float r0 = q->c[0] * ac, r1 = q->c[1] * ac, r2 = q->c[2] * ac;
    float *acc = s_accum + 4 * (size_t) q->sample;
    if (r0 != 0.f) atomicAdd(acc + 0, r0);
    if (r1 != 0.f) atomicAdd(acc + 1, r1);
    if (r2 != 0.f) atomicAdd(acc + 2, r2);
  }
};

// kAnyOnly (the launcher picks it for curve scenes whose occluders are all opaque: C5): see traverse_persistent
template <bool kCurves, bool kCount, bool kMotion, bool kInstLds = false, bool kAnyOnly = false>
__global__ void __launch_bounds__(BLOCK, kMotion ? FJ_MOTION_MINB : (kCurves ? FJ_CURVE_MINB : FJ_SHADOW_MINB)) k_shadow_trace(DScene S, const DShadowRay *squeue, float *s_accum,
    DCounters *cnt, TravTune tune)
{
  __shared__ uint32_t s_stack[(kCurves ? FJ_STACK_LDS_CURVES : FJ_STACK_LDS) * BLOCK];
  __shared__ double s_rayspace[kCurves ? FJ_RAYSPACE_DOUBLES * BLOCK : 1];
  __shared__ double s_inst[kInstLds ? InstLdsOf<kCurves>::T::WORDS : 1];
  if (kInstLds) InstLdsOf<kCurves>::T::fill(S, s_inst);
  const uint32_t n = cnt->shadow_count;         // written by k_shadow_cull earlier on this stream
  ShadowPolicy pol;
  pol.S = &S; pol.squeue = squeue; pol.s_accum = s_accum;
  LocalCounters lc = {0, 0, 0};
  traverse_persistent<kCurves, kCount, kMotion, kInstLds, kAnyOnly>(S, pol, tune, n, &cnt->shadow_xcd_head[0][0], make_stack(s_stack, S.stack_overflow_shadow, kCurves ? s_rayspace : nullptr), &lc, s_inst);
  if (kCount) {
    flush_counters(cnt, lc.nodes, lc.prims, lc.insts, 0, 0);
    flush_shadow_walk_counters(cnt, lc.nodes, lc.prims, lc.insts);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&cnt->squeued, (unsigned long long) n);
  }
}

#endif
