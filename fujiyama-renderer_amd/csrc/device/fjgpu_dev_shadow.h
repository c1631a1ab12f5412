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
#define SQ_INVALID 0xffffffffu // DShadowRay.sample of a padding slot

#ifndef FJ_CULL_MINB
#define FJ_CULL_MINB 1
#endif
template <bool kHair, bool kArea>
__global__ void __launch_bounds__(BLOCK, FJ_CULL_MINB) k_shadow_cull(DScene S, ShadowParams sp, const DLightRec *lrecs,
    uint32_t rec_begin, uint32_t rec_end, float *s_accum, DShadowRay *squeue, DCounters *cnt, int count_events)
{
  unsigned long long c_insts = 0, c_shadow = 0;
  const unsigned lane = __lane_id();
  const uint32_t n = rec_end - rec_begin;
  // waves claim contiguous runs of records (so the rays a wave emits -- and the shadow-queue
  // chunks it fills -- stay spatially coherent) from a global head: equal static slices left
  // the kernel waiting for the waves whose records face the lights, and assumed that every
  // block of the grid is resident
  const uint32_t n_waves = (gridDim.x * BLOCK) >> 6;
  uint32_t claim = (n / (n_waves * 4u)) & ~63u;
  claim = claim < 64u ? 64u : (claim > 1024u ? 1024u : claim);
  // queue space is reserved SQ_CHUNK slots at a time: one atomic per 512 rays
  // instead of one per wave iteration (a single-address atomic per iteration
  // serialised the whole kernel in L2)
  uint32_t chunk_base = 0, chunk_used = SQ_CHUNK;   // wave-uniform; "used == CHUNK" = no chunk yet

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
    const double *g_sbounds = nullptr;     // stays a global-memory pointer (a by-value DGroup lands in scratch)
    double sb[6] = {0, 0, 0, 0, 0, 0};
    double cos_limit = 0;
    if (active) {
      R = lrecs[rec];
      if (kHair && R.kind == 1) H = S.lrec_hair[rec];
      Ps = mk(R.P[0], R.P[1], R.P[2]);
      axis = mk(R.N[0], R.N[1], R.N[2]);
      nml_axis = normalize(axis);
      r_sample = R.sample;
      W[0] = R.W[0]; W[1] = R.W[1]; W[2] = R.W[2];
      cos_limit = (!kHair || R.kind == 0) ? sp.cos_half_pi : sp.cos_pi;
      g_first = S.groups[R.group].first; g_count = S.groups[R.group].count;
      g_single = S.groups[R.group].n_instances == 1;
      g_sbounds = S.groups[R.group].sbounds;
      // a single-instance group's box is the same for every light of the record: read it once
      // (per pair it cost two dependent loads -- node, then box -- before any arithmetic)
      if (g_single) for (int q = 0; q < 6; q++) sb[q] = g_sbounds[q];
    }
    // the light index is wave-uniform: the sample's 72 bytes come through the scalar cache into
    // SGPRs instead of 64 identical vector loads
    for (uint32_t l = 0; l < nl; l++) {
      bool emit = false;
      DShadowRay q;
      if (active) {
        const DLightSample LS = S.light_samples[l];
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
          if (!kHair || R.kind == 0) {   // plastic_shader.cc:131-137
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
            spec = (float) pow((double) spec, (double) (1 / roughness));
            k[0] = (H.Cd[0] * diff + spec) * Cl[0];
            k[1] = (H.Cd[1] * diff + spec) * Cl[1];
            k[2] = (H.Cd[2] * diff + spec) * Cl[2];
          }
          bool maybe_occluded = false;
          if (sp.cast_shadow) {
            c_shadow++;
            // group bounds test + leaf bounds of the instance BVH, as culling
            if (!has_negative_zero(Ln)) {
              const V3 winv = mk(filter_rcp(Ln.x), filter_rcp(Ln.y), filter_rcp(Ln.z));
              const bool plain = plain_dir(Ln);
              if (g_single) {
                maybe_occluded = box_ray_ref_fast(sb, Ps, Ln, winv, plain, .0001, distance);
                if (!maybe_occluded) c_insts++;
              }
              for (int ti = g_first; !g_single && ti < g_first + g_count;) {      // threaded instance BVH (DTNode)
                const DTNode *tn_ = &S.group_nodes[ti];
                if (tn_->inst < 0) {
                  double tq;
                  ti = slab(tn_->box, tn_->box + 3, Ps, winv, .0001, distance, &tq) ? ti + 1 : tn_->skip;
                  continue;
                }
                ti++;
                const DInstance *I = &S.instances[tn_->inst];
                if (box_ray_ref_fast(I->wbounds, Ps, Ln, winv, plain, .0001, distance)) { maybe_occluded = true; break; }
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
            q.sample = r_sample; q.group = R.group; q.tindex = R.uid & 0xfffffu;
          } else {
            sum[0] += k[0]; sum[1] += k[1]; sum[2] += k[2];
          }
        }
      }
      // ---- compaction into the wave's current chunk (ballot + prefix popcount)
      const unsigned long long mask = __ballot(emit);
      const uint32_t need = (uint32_t) __popcll(mask);
      if (need) {
        if (chunk_used + need > SQ_CHUNK) {
          // retire the chunk: mark its unused tail as padding, reserve a new one
          if (chunk_used < SQ_CHUNK)
            for (uint32_t k = chunk_used + lane; k < SQ_CHUNK; k += 64)
              if (chunk_base + k < sp.queue_capacity) squeue[chunk_base + k].sample = SQ_INVALID;
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(&cnt->shadow_count, SQ_CHUNK);
          chunk_base = __shfl(base, 0);
          chunk_used = 0;
        }
        if (emit) {
          const uint32_t slot = chunk_base + chunk_used + (uint32_t) __popcll(mask & ((1ull << lane) - 1ull));
          if (slot < sp.queue_capacity) squeue[slot] = q;
          else cnt->overflow = 1;
        }
        chunk_used += need;
      }
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
      if (chunk_base + k < sp.queue_capacity) squeue[chunk_base + k].sample = SQ_INVALID;
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

template <bool kCurves, bool kCount, bool kMotion>
__global__ void __launch_bounds__(BLOCK, (kCurves || kMotion) ? FJ_CURVE_MINB : FJ_SHADOW_MINB) k_shadow_trace(DScene S, const DShadowRay *squeue, float *s_accum,
    DCounters *cnt, TravTune tune)
{
  __shared__ uint32_t s_stack[(kCurves ? FJ_STACK_LDS_CURVES : FJ_STACK_LDS) * BLOCK];
  __shared__ double s_rayspace[kCurves ? FJ_RAYSPACE_DOUBLES * BLOCK : 1];
  const uint32_t n = cnt->shadow_count;         // written by k_shadow_cull earlier on this stream
  ShadowPolicy pol;
  pol.S = &S; pol.squeue = squeue; pol.s_accum = s_accum;
  LocalCounters lc = {0, 0, 0};
  traverse_persistent<kCurves, kCount, kMotion>(S, pol, tune, n, &cnt->shadow_head, make_stack(s_stack, S.stack_overflow_shadow, kCurves ? s_rayspace : nullptr), &lc);
  if (kCount) {
    flush_counters(cnt, lc.nodes, lc.prims, lc.insts, 0, 0);
    flush_shadow_walk_counters(cnt, lc.nodes, lc.prims, lc.insts);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&cnt->squeued, (unsigned long long) n);
  }
}

// ---- conservative f32 slab test for the any-hit walk.
// Per (ray, instance) and axis a the entry code keeps  i32 = (float)(1/od_a),
// o32 = (float)(oo_a/od_a)  and a margin  E = 1.5 * 2^-22 * (Bmax_a * |i32| + |o32|),
// Bmax_a >= |any box coordinate| of the primitive set.  For a box plane b (f32):
//   t~ = fmaf(b, i32, -o32)  differs from the exact (b - oo_a)/od_a by at most
//   |b/od| 2^-24 (i32 rounding) + |oo/od| 2^-24 (o32 rounding) + |t~| 2^-24 (fma rounding)
//   <= 2^-23 (Bmax |i32| + |o32|) (1 + 2^-22)  <  E,
// so [min(t~0, t~1) - E, max(t~0, t~1) + E] contains the exact slab interval and the f64
// interval of slab_f32box (whose own error is ~2^-52 relative): whatever the f64 test
// accepts this one accepts -- it can only cull less.  An axis whose E is not a finite
// number below 1e30 (direction component zero or denormal: 1/od = inf, 0 * inf = NaN) is
// given i32 = o32 = 0, E = 1e30: t~ = 0, interval [-1e30, 1e30], it never culls.  With
// E < 1e30 every product is below 2.8e36, so no inf and no NaN can arise in the test.
struct Slab32 { float ix, iy, iz, ox, oy, oz, ex, ey, ez; };
#ifdef FJ_EXP_SLAB_VALIDATE
__device__ unsigned long long g_slab_lost, g_slab_extra, g_slab_tests;
#endif

__device__ __forceinline__ void slab32_axis(double inv, double oo, double bmax_abs, float *i32, float *o32, float *e32)
{
  float i = (float) inv, o = (float) (oo * inv);
  float e = 3.6e-7f * ((float) bmax_abs * 1.0000002f * fabsf(i) + fabsf(o));
  if (!(e < 1e30f)) { i = 0.f; o = 0.f; e = 1e30f; }
  *i32 = i; *o32 = o; *e32 = e;
}

__device__ __forceinline__ Slab32 slab32_setup(V3 oo, V3 inv, const double *bounds)
{
  Slab32 s;
  slab32_axis(inv.x, oo.x, fmax(fabs(bounds[0]), fabs(bounds[3])), &s.ix, &s.ox, &s.ex);
  slab32_axis(inv.y, oo.y, fmax(fabs(bounds[1]), fabs(bounds[4])), &s.iy, &s.oy, &s.ey);
  slab32_axis(inv.z, oo.z, fmax(fabs(bounds[2]), fabs(bounds[5])), &s.iz, &s.oz, &s.ez);
  return s;
}

// box = {min xyz, max xyz}; tmin32 <= tmin and tmax32 >= tmax of the ray
__device__ __forceinline__ bool slab32_test(const float *b, const Slab32 &s, float tmin32, float tmax32)
{
  const float x0 = fmaf(b[0], s.ix, -s.ox), x1 = fmaf(b[3], s.ix, -s.ox);
  const float y0 = fmaf(b[1], s.iy, -s.oy), y1 = fmaf(b[4], s.iy, -s.oy);
  const float z0 = fmaf(b[2], s.iz, -s.oz), z1 = fmaf(b[5], s.iz, -s.oz);
  const float lx = fminf(x0, x1) - s.ex, hx = fmaxf(x0, x1) + s.ex;
  const float ly = fminf(y0, y1) - s.ey, hy = fmaxf(y0, y1) + s.ey;
  const float lz = fminf(z0, z1) - s.ez, hz = fmaxf(z0, z1) + s.ez;
  const float tn = fmaxf(fmaxf(lx, ly), fmaxf(lz, tmin32));
  const float tf = fminf(fminf(hx, hy), fminf(hz, tmax32));
  return tn <= tf;
}

// ---- lean any-hit traversal: shadow rays of scenes in which every possible occluder is
// opaque (Os = 1) and no curve set exists -- the common case and the dominant kernel of
// C1-C3.  Same tests, same order of instances, same result (occluded or not) as
// traverse_persistent with anyhit rays; what is gone is the closest-hit bookkeeping
// (best t/u/v/ids, tie rule, range shrinking) and the world-space ray, which is re-read
// from the queue entry on the rare instance switches.  The point is registers: occupancy
// decides throughput on this latency-bound walk.
template <bool kCount>
__device__ void traverse_anyhit(const DScene &S, const DShadowRay *squeue, float *s_accum, TravTune tune,
    uint32_t n, uint32_t *head, TravStack stk, LocalCounters *lc)
{
  const unsigned lane = __lane_id();
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  bool head_live = true;
  uint32_t next = 0, range_end = 0;
  tune.grab = adaptive_grab(tune.grab, n);
  bool have = false, hit = false;
  uint32_t idx = 0;
  V3 oo = mk(0, 0, 0), od = oo;
#if !defined(FJ_EXP_ANYHIT_F32SLAB) || defined(FJ_EXP_SLAB_VALIDATE)
  V3 inv_keep = oo;
#endif
#ifdef FJ_EXP_ANYHIT_F32SLAB
  Slab32 s32 = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  float tmax32 = 0.f;
  const float tmin32 = 9.9999e-5f;      // <= .0001
#endif
  double tmax = 0;
  int gi = 0, gend = 0;
  const DNode *nodes = nullptr;
  const double *tris = nullptr;
  const float *tris32 = nullptr;
  uint32_t cur = TRAV_DONE;
  int sp = 0;
  const double tmin = .0001;

  for (;;) {
    // ---- refill idle lanes (see traverse_persistent)
    const unsigned long long idle = __ballot(!have);
    if (next >= range_end && head_live && (idle == ~0ull || (unsigned) __popcll(idle) >= TRAV_REFILL)) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(head, (uint32_t) TRAV_GRAB);
      base = __shfl(base, 0);
      if (base >= n) head_live = false;
      else { next = base; range_end = (n - base < TRAV_GRAB) ? n : base + TRAV_GRAB; }
    }
    if (idle == ~0ull || ((unsigned) __popcll(idle) >= TRAV_REFILL && next < range_end)) {
      if (!have) {
        const uint32_t my = next + (uint32_t) __popcll(idle & lt_mask);
        if (my < range_end && squeue[my].sample != SQ_INVALID) {
          have = true; hit = false;
          idx = my;
          const int g = squeue[my].group;
          gi = S.groups[g].first;
          gend = gi + S.groups[g].count;
          cur = TRAV_DONE; sp = 0;
        }
      }
      next += (uint32_t) __popcll(idle);
      if (__ballot(have) == 0ull) {
        if (next >= range_end && !head_live) break;
        continue;
      }
    }

    // ---- between instances: retire the ray or enter the next instance
    if (have && cur == TRAV_DONE) {
      const DShadowRay *q = &squeue[idx];
      bool found = false;
      if (!hit) {
        const V3 o = mk(q->o[0], q->o[1], q->o[2]), d = mk(q->d[0], q->d[1], q->d[2]);
        tmax = q->tmax;
        if (!has_negative_zero(d)) {
          const V3 winv = mk(filter_rcp(d.x), filter_rcp(d.y), filter_rcp(d.z));
          const bool plain = plain_dir(d);
          const DGroup *G = &S.groups[q->group];
          const bool single = G->n_instances == 1;
          while (gi < gend) {
            const DTNode *tn_ = &S.group_nodes[gi];
            if (tn_->inst < 0) {           // inner node of the instance BVH
              double tq;
              gi = slab(tn_->box, tn_->box + 3, o, winv, tmin, tmax, &tq) ? gi + 1 : tn_->skip;
              continue;
            }
            gi++;
            const DInstance *I = &S.instances[tn_->inst];
            if (kCount) lc->insts++;
            // (a single-instance group's box: the light loop queued this ray BECAUSE this very test
            // -- same box, same ray, same range -- passed there)
            if (!single && !box_ray_ref_fast(I->wbounds, o, d, winv, plain, tmin, tmax)) continue;
            oo = xpoint(I->Minv, o);
            od = xvector(I->Minv, d);
            if (has_negative_zero(od)) continue;
            const V3 inv = mk(1. / od.x, 1. / od.y, 1. / od.z);
            const DPrimSet *P = &S.primsets[I->primset];
            if (P->n_prims == 0) continue;
            double tn;
            if (!slab(P->bounds, P->bounds + 3, oo, inv, tmin, tmax, &tn)) continue;
#if !defined(FJ_EXP_ANYHIT_F32SLAB) || defined(FJ_EXP_SLAB_VALIDATE)
            inv_keep = inv;
#endif
#ifdef FJ_EXP_ANYHIT_F32SLAB
            s32 = slab32_setup(oo, inv, P->bounds);
            tmax32 = nextafterf((float) tmax, INFINITY);
#endif
            nodes = P->nodes; tris = P->tri_verts; tris32 = P->tri_verts32;
            cur = P->root; sp = 0;
            found = true;
            break;
          }
        }
      }
      if (!found) {
        if (!hit) {      // reached the light: add c (an opaque occluder adds c * (1 - Os) = 0)
          float *acc = s_accum + 4 * (size_t) q->sample;
          const float r0 = q->c[0], r1 = q->c[1], r2 = q->c[2];
          if (r0 != 0.f) atomicAdd(acc + 0, r0);
          if (r1 != 0.f) atomicAdd(acc + 1, r1);
          if (r2 != 0.f) atomicAdd(acc + 2, r2);
        }
        have = false;
      }
    }

    // ---- inner nodes
    for (int step = 0; step < (int) tune.anyhit_steps; step++) {
      const bool inner = have && !(cur & FJ_LEAF_FLAG);
      const unsigned long long im = __ballot(inner);
      if (im == 0ull || (step > 0 && (uint32_t) __popcll(im) < tune.min_inner)) break;
      if (inner) {
        const FJ_GLOBAL fj_v4f *nd = (const FJ_GLOBAL fj_v4f *) (nodes + cur);
        if (kCount) lc->nodes++;
        const fj_v4f q0 = nd[0], q1 = nd[1], q2 = nd[2], q3 = nd[3], q4 = nd[4], q5 = nd[5];
        const fj_v4u e = ((const FJ_GLOBAL fj_v4u *) nd)[6];
        const float b0[6] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y};
        const float b1[6] = {q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
        const float b2[6] = {q3.x, q3.y, q3.z, q3.w, q4.x, q4.y};
        const float b3[6] = {q4.z, q4.w, q5.x, q5.y, q5.z, q5.w};
#ifndef FJ_EXP_ANYHIT_F32SLAB
        double t0, t1, t2, t3;
        const bool h0 = slab_f32box(b0, b0 + 3, oo, inv_keep, tmin, tmax, &t0);
        const bool h1 = slab_f32box(b1, b1 + 3, oo, inv_keep, tmin, tmax, &t1);
        const bool h2 = e.z != FJ_NO_CHILD && slab_f32box(b2, b2 + 3, oo, inv_keep, tmin, tmax, &t2);
        const bool h3 = e.w != FJ_NO_CHILD && slab_f32box(b3, b3 + 3, oo, inv_keep, tmin, tmax, &t3);
#else
        const bool h0 = slab32_test(b0, s32, tmin32, tmax32);
        const bool h1 = slab32_test(b1, s32, tmin32, tmax32);
        const bool h2 = e.z != FJ_NO_CHILD && slab32_test(b2, s32, tmin32, tmax32);
        const bool h3 = e.w != FJ_NO_CHILD && slab32_test(b3, s32, tmin32, tmax32);
#ifdef FJ_EXP_SLAB_VALIDATE
        {   // every box the f64 test accepts must be accepted by the f32 test
          double tq;
          const bool g0 = slab_f32box(b0, b0 + 3, oo, inv_keep, tmin, tmax, &tq);
          const bool g1 = slab_f32box(b1, b1 + 3, oo, inv_keep, tmin, tmax, &tq);
          const bool g2 = e.z != FJ_NO_CHILD && slab_f32box(b2, b2 + 3, oo, inv_keep, tmin, tmax, &tq);
          const bool g3 = e.w != FJ_NO_CHILD && slab_f32box(b3, b3 + 3, oo, inv_keep, tmin, tmax, &tq);
          const int lost = (int) (g0 && !h0) + (int) (g1 && !h1) + (int) (g2 && !h2) + (int) (g3 && !h3);
          const int extra = (int) (h0 && !g0) + (int) (h1 && !g1) + (int) (h2 && !g2) + (int) (h3 && !g3);
          if (lost) atomicAdd(&g_slab_lost, (unsigned long long) lost);
          if (extra) atomicAdd(&g_slab_extra, (unsigned long long) extra);
          atomicAdd(&g_slab_tests, (unsigned long long) (2 + (e.z != FJ_NO_CHILD) + (e.w != FJ_NO_CHILD)));
        }
#endif
#endif
        // any hit ends the ray, so the visiting order is free: no distance sort; children
        // are stored by decreasing surface area (the builder), larger ones first
        uint32_t r0 = e.x, r1 = e.y, r2 = e.z, r3 = e.w;
        if (!h2) { r2 = r3; }
        if (!h1) { r1 = r2; r2 = r3; }
        if (!h0) { r0 = r1; r1 = r2; r2 = r3; }
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

    // ---- leaves: the first triangle hit inside [tmin, tmax] ends the ray
    if (have && (cur & FJ_LEAF_FLAG) && cur != TRAV_DONE) {
      const uint32_t first = (cur & 0x7fffffffu) >> 3;
      const uint32_t cnt = (cur & 7u) + 1;
      for (uint32_t k = 0; k < cnt; k++) {
        double t, u, v;
        if (kCount) lc->prims++;
        V3 v0, v1, v2;
        load_tri(tris, tris32, first + k, &v0, &v1, &v2);
        if (!tri_ray(v0, v1, v2, oo, od, &t, &u, &v)) continue;
        if (!(tmin <= t && t <= tmax)) continue;
        hit = true;
        break;
      }
      if (hit) { have = false; cur = TRAV_DONE; }
      else cur = (sp == 0) ? TRAV_DONE : stk.pop(sp);
    }
  }
}

#ifndef FJ_ANYHIT_MINB
#define FJ_ANYHIT_MINB 4
#endif
template <bool kCount>
__global__ void __launch_bounds__(BLOCK, FJ_ANYHIT_MINB) k_shadow_anyhit(DScene S, const DShadowRay *squeue, float *s_accum,
    DCounters *cnt, TravTune tune)
{
  __shared__ uint32_t s_stack[FJ_STACK_LDS * BLOCK];
  const uint32_t n = cnt->shadow_count;
  LocalCounters lc = {0, 0, 0};
  traverse_anyhit<kCount>(S, squeue, s_accum, tune, n, &cnt->shadow_head, make_stack(s_stack, S.stack_overflow_shadow), &lc);
  if (kCount) {
    flush_counters(cnt, lc.nodes, lc.prims, lc.insts, 0, 0);
    flush_shadow_walk_counters(cnt, lc.nodes, lc.prims, lc.insts);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&cnt->squeued, (unsigned long long) n);
  }
}

#endif
