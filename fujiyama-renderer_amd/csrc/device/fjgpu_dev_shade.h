// fjgpu_dev_shade.h -- camera-ray generation and the shading kernel (attributes + the five shader plugins).
// Part of the kernels translation unit: included by fjgpu_kernels.hip only (device code,
// compiled with -ffp-contract=off; see the header of that file).
#ifndef FJGPU_DEV_SHADE_H
#define FJGPU_DEV_SHADE_H

// Sample uid of the random-stream contract (the oracle computes the same): tile id * 2^20 + sample index
// in the tile as a 64-bit number, its high word folded into the low one -- the plain 32-bit sum for frames of up to 4096
// tiles of up to 2^20 samples, distinct streams beyond.  (The sample's TIME index travels separately: DPath.flags >> 1.)
__device__ __forceinline__ uint32_t sample_uid(int32_t tile_id, uint32_t k)
{
  const unsigned long long u = ((unsigned long long) (uint32_t) tile_id << 20) + k;
  return (uint32_t) u ^ ((uint32_t) (u >> 32) * 0x9E3779B1u);
}

// Camera::GetRay (src/fj_camera.cc:79-110): a time-sampled camera is evaluated at the
// sample's time, a static one uses the host-built matrix.  k = index of the sample in its
// tile (time stream, sample uid), slot = its place in the batch's sample arrays.
// (static camera: the ray of sample coordinates (u, v), from the host-built matrix)
__device__ __forceinline__ DRay static_camera_ray(const DScene &S, double u, double v)
{
  const double *cam = S.cam_M;
  const V3 target = mk((u - .5) * S.cam_uv_size[0], (v - .5) * S.cam_uv_size[1], -1);
  const V3 tw = xpoint(cam, target);
  const V3 eye = mk(cam[3], cam[7], cam[11]);
  const V3 dir = normalize(tw - eye);
  DRay r;
  r.o[0] = eye.x; r.o[1] = eye.y; r.o[2] = eye.z;
  r.d[0] = dir.x; r.d[1] = dir.y; r.d[2] = dir.z;
  return r;                    // (range: S.cam_znear .. S.cam_zfar, ray_range)
}
__device__ __forceinline__ DPath camera_path(const DScene &S, uint32_t slot, uint32_t k, uint32_t uid)
{
  DPath p;
  p.sample = slot;
  p.T[0] = p.T[1] = p.T[2] = 1.f;
  p.cxt = CXT_CAMERA_RAY; p.ddepth = p.rdepth = p.tdepth = 0;
  p.group = S.target_group;
  p.flags = k << 1; p.rng = 0; p.uid = uid;
  return p;
}
// IMPLICIT camera rays (DScene.cam_uv): with a static camera, and no random stream or sample time keyed by the
// sample's uid anywhere in the scene, a camera ray is a function of its sample's (u, v) alone, and its path state is a
// constant but for the sample slot.  k_gen_camera then writes only the (u, v) table the pixel filter needs anyway, and
// the closest-hit walk and the shading kernel rebuild ray i of level 0 from cam_uv[i] -- ~50 instructions -- instead of
// reading the ray and path records (64 + 48 bytes per sample then; 48 + 36 since round 6) that k_gen_camera would have written (C3: 141 M samples, 15.8 GB less written
// and 25 GB less read per frame).  The same arithmetic in the same order as camera_ray: the same rays bit for bit.
__device__ __forceinline__ DRay implicit_camera_ray(const DScene &S, uint32_t i)
{
  const double2 uv = reinterpret_cast<const double2 *>(S.cam_uv)[i];
  return static_camera_ray(S, uv.x, uv.y);
}

template <bool kMovingCamera>
__device__ __forceinline__ void camera_ray(const DScene &S, double u, double v, int32_t tile_id, uint32_t k, uint32_t slot,
    DRay *ray_out, DPath *path_out)
{
  if (!kMovingCamera) *ray_out = static_camera_ray(S, u, v);
  else {
    double cm[12], cmi[12];
    xform_at(S.cam_xform, sample_time(S, k), cm, cmi);
    const double *cam = cm;
    const V3 target = mk((u - .5) * S.cam_uv_size[0], (v - .5) * S.cam_uv_size[1], -1);
    const V3 tw = xpoint(cam, target);
    const V3 eye = mk(cam[3], cam[7], cam[11]);
    const V3 dir = normalize(tw - eye);
    DRay r;
    r.o[0] = eye.x; r.o[1] = eye.y; r.o[2] = eye.z;
    r.d[0] = dir.x; r.d[1] = dir.y; r.d[2] = dir.z;
    *ray_out = r;
  }
  *path_out = camera_path(S, slot, k, sample_uid(tile_id, k));
}

// --------------------------------------------------------------- k_gen_camera
// FixedGridSampler::generate_samples (src/fj_fixed_grid_sampler.cc:33-84) with the
// per-tile XorShift streams read from host-built tables (the stream restarts
// for every tile, so draw k is the same number in every tile), then
// Camera::GetRay (src/fj_camera.cc:79-110) with the host-built camera matrix.
template <bool kMovingCamera>
__global__ void __launch_bounds__(BLOCK) k_gen_camera(DScene S, GenParams gp, const TileDesc *tiles,
    const double *jitter_tab, const double *time_tab, double *s_uv, DRay *rays, DPath *paths, uint32_t *s_tk)
{
  const TileDesc T = tiles[blockIdx.y];
  const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
  const uint32_t ns = (uint32_t) T.nx * (uint32_t) T.ny;
  if (k >= ns) return;
  const int x = (int) (k % (uint32_t) T.nx), y = (int) (k / (uint32_t) T.nx);
  const int xoffset = T.xmin * gp.rate_x - gp.margin_x;
  const int yoffset = T.ymin * gp.rate_y - gp.margin_y;

  double u = (.5 + x + xoffset) * gp.udelta;
  double v = 1 - (.5 + y + yoffset) * gp.vdelta;
  if (gp.jittered) {
    const double u_jitter = jitter_tab[2 * (size_t) k] * gp.jitter;
    const double v_jitter = jitter_tab[2 * (size_t) k + 1] * gp.jitter;
    u += gp.udelta * (u_jitter - .5);
    v += gp.vdelta * (v_jitter - .5);
  }
  const uint32_t slot = T.sample_offset + k;
  s_uv[2 * (size_t) slot] = u;
  s_uv[2 * (size_t) slot + 1] = v;
  (void) time_tab;   // (the same table as S.time_tab)

  if (!kMovingCamera && rays == nullptr) {            // implicit camera rays: the (u, v) table is all that is needed ...
    if (s_tk) { s_tk[2 * (size_t) slot] = (uint32_t) T.id; s_tk[2 * (size_t) slot + 1] = k; }      // ... and who the sample is, where a random stream asks
    return;
  }
  camera_ray<kMovingCamera>(S, u, v, T.id, k, slot, rays + slot, paths + slot);
}

// -------------------------------------------------------------------- shading

// Texture::Lookup, src/fj_texture.cc:51-78 + MipInput::ReadTile clamp (src/fj_mipmap.cc:153-170)
__device__ void tex_lookup(const DTexture &tex, float u, float v, float out[4])
{
  if (tex.width == 0 || tex.tiles == nullptr) { out[0] = 1.f; out[1] = .63f; out[2] = .63f; out[3] = 1.f; return; }
  const int ts = tex.tilesize;
  const int xnt = tex.width / ts, ynt = tex.height / ts;
  const float tu = u - floorf(u);
  const float tv = v - floorf(v);
  const float su = tu * xnt;
  const float sv = (1 - tv) * ynt;
  int xt = (int) floorf(su), yt = (int) floorf(sv);
  xt = xt < 0 ? 0 : (xt > xnt - 1 ? xnt - 1 : xt);
  yt = yt < 0 ? 0 : (yt > ynt - 1 ? ynt - 1 : yt);
  const int xp = (int) ((su - floorf(su)) * 64);
  const int yp = (int) ((sv - floorf(sv)) * 64);
  if (xp < 0 || xp >= ts || yp < 0 || yp >= ts) { out[0] = out[1] = out[2] = out[3] = 0.f; return; }
  const FJ_GLOBAL float *p = FJ_G(float, tex.tiles) + ((size_t) (yt * xnt + xt) * ts * ts + (size_t) (yp * ts + xp)) * tex.nchannels;
  switch (tex.nchannels) {
  case 1: out[0] = out[1] = out[2] = p[0]; out[3] = 1.f; break;
  case 3: out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; out[3] = 1.f; break;
  case 4: out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; out[3] = p[3]; break;
  default: out[0] = out[1] = out[2] = out[3] = 0.f; break;
  }
}

__device__ __forceinline__ V3 faceforward(V3 I, V3 N) { return (dot(I, N) < 0) ? N : mk(-N.x, -N.y, -N.z); }   // src/fj_shading.cc:42-51

__device__ double fresnel(V3 I, V3 N, double ior)   // SlFresnel, src/fj_shading.cc:53-73
{
  double c = -1 * dot(I, N);
  double eta;
  if (c > 0) eta = ior;
  else { eta = 1. / ior; c *= -1; }
  const double k2 = .0;
  const double F0 = ((1. - eta) * (1. - eta) + k2) / ((1. + eta) * (1. + eta) + k2);
  return F0 + (1. - F0) * pow(1. - c, 5.);
}

__device__ __forceinline__ V3 reflect(V3 I, V3 N)   // SlReflect, :90-98
{
  const double c = -1 * dot(I, N);
  return mk(I.x + 2 * c * N.x, I.y + 2 * c * N.y, I.z + 2 * c * N.z);
}

__device__ V3 refract(V3 I, V3 N, double ior)        // SlRefract, :100-138
{
  V3 n;
  double eta;
  double c1 = -1 * dot(I, N);
  if (c1 < 0) { c1 *= -1; eta = 1 / ior; n = mk(-N.x, -N.y, -N.z); }
  else { eta = ior; n = N; }
  const double radicand = 1 - eta * eta * (1 - c1 * c1);
  if (radicand < 0.) return reflect(I, N);
  const double nc = eta * c1 - sqrt(radicand);
  return mk(eta * I.x + nc * n.x, eta * I.y + nc * n.y, eta * I.z + nc * n.z);
}

__device__ __forceinline__ float luminance4(const float c[4]) { return (float) (.298912 * c[0] + .586611 * c[1] + .114478 * c[2]); }

__device__ __forceinline__ float luminance3(const float c[3]) { return (float) (.298912 * c[0] + .586611 * c[1] + .114478 * c[2]); }

// Counter-based RNG contract of the pathtracing path: the reference's
// seeded XorShift (src/fj_random.cc:18-43) with seed = mix(sample uid, path key), four
// warm-up draws, then the two numbers of the diffuse bounce.
__device__ __forceinline__ uint32_t pt_mix(uint32_t uid, uint32_t key)
{
  uint32_t h = uid * 0x9E3779B1u ^ (key + 0x7F4A7C15u) * 0x85EBCA77u;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}

// XorShift of src/fj_random.cc:10-43 (state in registers)
struct XS {
  uint32_t a, b, c, d;
  __device__ __forceinline__ uint32_t next()
  {
    const uint32_t t = a ^ (a << 11);
    a = b; b = c; c = d;
    d = (d ^ (d >> 19)) ^ (t ^ (t >> 8));
    return d;
  }
  __device__ __forceinline__ double f01() { return (double) next() / 4294967295u; }
};

// stream of one (shading event, area light): RNG contract of DESIGN.md 4
__device__ __forceinline__ XS area_stream(uint32_t uid, uint32_t key, int light)
{
  uint32_t seed = pt_mix(pt_mix(uid, key) ^ 0x51ED270Bu, (uint32_t) light);
  XS r;
  r.a = seed = 1812433253U * (seed ^ (seed >> 30)) + 0u;
  r.b = seed = 1812433253U * (seed ^ (seed >> 30)) + 1u;
  r.c = seed = 1812433253U * (seed ^ (seed >> 30)) + 2u;
  r.d = seed = 1812433253U * (seed ^ (seed >> 30)) + 3u;
  for (int i = 0; i < 4; i++) r.next();
  return r;
}

__device__ void pt_draw2(uint32_t uid, uint32_t key, double *x1, double *x2)
{
  uint32_t h = uid * 0x9E3779B1u ^ (key + 0x7F4A7C15u) * 0x85EBCA77u;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  uint32_t st[4];
  uint32_t seed = h;
  for (uint32_t i = 0; i < 4; i++) st[i] = seed = 1812433253U * (seed ^ (seed >> 30)) + i;
  double out[2] = {0, 0};
  for (int i = 0; i < 6; i++) {
    const uint32_t t = st[0] ^ (st[0] << 11);
    st[0] = st[1]; st[1] = st[2]; st[2] = st[3];
    st[3] = (st[3] ^ (st[3] >> 19)) ^ (t ^ (t >> 8));
    if (i >= 4) out[i - 4] = (double) st[3] / 4294967295u;
  }
  *x1 = out[0];
  *x2 = out[1];
}

// SlBumpMapping, src/fj_shading.cc:418-464
__device__ V3 bump_mapping(const DTexture &bump, V3 dPdu, V3 dPdv, float tu, float tv, double amplitude, V3 N)
{
  if (bump.width == 0 || bump.height == 0) return N;
  const float du = (float) (1. / bump.width);
  const float dv = (float) (1. / bump.height);
  float c0[4], c1[4];
  tex_lookup(bump, tu - du, tv, c0);
  tex_lookup(bump, tu + du, tv, c1);
  const float Bu = (luminance4(c0) - luminance4(c1)) / (2 * du);
  tex_lookup(bump, tu, tv - dv, c0);
  tex_lookup(bump, tu, tv + dv, c1);
  const float Bv = (luminance4(c0) - luminance4(c1)) / (2 * dv);
  V3 a = cross(N, dPdu), b = cross(N, dPdv);
  a = mk(a.x * du, a.y * du, a.z * du);
  b = mk(b.x * du, b.y * du, b.z * du);
  const V3 nb = mk(N.x + amplitude * (Bv * a.x - Bu * b.x),
                   N.y + amplitude * (Bv * a.y - Bu * b.y),
                   N.z + amplitude * (Bv * a.z - Bu * b.z));
  return normalize(nb);
}

// Block-aggregated append of the shading kernel's four outputs -- the light record and the diffuse / reflect / refract
// children, the three children into ONE queue -- in a single pass: slots by ballot prefix count inside the wave, wave
// offsets through LDS, ONE global atomic per block and queue, the two issued side by side by two lanes of the same wave
// instruction.  (Every wave of a frame adding to the same counters -- which share one cache line -- serialises in L2:
// 10 M same-line atomics per C3 frame were most of the shading kernel's time.  Four appends one after the other -- each two
// barriers and a returning atomic the whole block waits for -- were a third of a block's life on C4.)
// Every thread of the block must call it.  A block's children sit in the queue as [diffuse | reflect | refract].
struct AppendSlots { uint32_t light, c2, c0, c1; };
__device__ __forceinline__ AppendSlots block_append4(bool want_light, bool want2, bool want0, bool want1, DCounters *cnt, uint32_t *s_tmp)
{
  constexpr int W = BLOCK / 64;
  const unsigned long long mL = __ballot(want_light), m2 = __ballot(want2), m0 = __ballot(want0), m1 = __ballot(want1);
  const unsigned lane = __lane_id(), w = threadIdx.x >> 6;
  if (lane == 0) {
    s_tmp[w] = (uint32_t) __popcll(mL); s_tmp[W + w] = (uint32_t) __popcll(m2);
    s_tmp[2 * W + w] = (uint32_t) __popcll(m0); s_tmp[3 * W + w] = (uint32_t) __popcll(m1);
  }
  __syncthreads();
  uint32_t tot[4], before[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    uint32_t t = 0, bf = 0;
#pragma unroll
    for (int k = 0; k < W; k++) { const uint32_t c = s_tmp[q * W + k]; bf += (unsigned) k < w ? c : 0u; t += c; }
    tot[q] = t; before[q] = bf;
  }
  if (threadIdx.x < 2) {       // lane 0: the light queue; lane 1: the ray queue
    const uint32_t total = threadIdx.x == 0 ? tot[0] : tot[1] + tot[2] + tot[3];
    uint32_t base = 0;
    if (total) base = atomicAdd(threadIdx.x == 0 ? &cnt->light_count : &cnt->next_count, total);
    s_tmp[4 * W + threadIdx.x] = base;
  }
  if (threadIdx.x >= 64 && threadIdx.x < 67) {     // (another wave: the per-context ray counts; nobody waits for these)
    const int q = (int) threadIdx.x - 63;
    static_assert(CXT_DIFFUSE_RAY == 2 && CXT_REFLECT_RAY == 3 && CXT_REFRACT_RAY == 4, "contexts of the three children");
    if (tot[q]) atomicAdd(&cnt->rays[q + 1], (unsigned long long) tot[q]);
  }
  __syncthreads();
  const unsigned long long lt = (1ull << lane) - 1ull;
  const uint32_t baseL = s_tmp[4 * W], baseR = s_tmp[4 * W + 1];
  AppendSlots a;
  a.light = want_light ? baseL + before[0] + (uint32_t) __popcll(mL & lt) : 0xffffffffu;
  a.c2 = want2 ? baseR + before[1] + (uint32_t) __popcll(m2 & lt) : 0xffffffffu;
  a.c0 = want0 ? baseR + tot[1] + before[2] + (uint32_t) __popcll(m0 & lt) : 0xffffffffu;
  a.c1 = want1 ? baseR + tot[1] + tot[2] + before[3] + (uint32_t) __popcll(m1 & lt) : 0xffffffffu;
  return a;
}

// key of the ray-queue sort: direction octant (3 bits) over the Morton code of the origin's cell in a 2^bits grid per axis
// over the scene box (fjgpu_raysort.hip sorts (key, index) pairs over exactly these 3 + 3 bits bits)
__device__ __forceinline__ uint32_t sort_spread3(uint32_t v)      // 10 bits -> every third bit
{
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
__device__ __forceinline__ uint32_t ray_sort_key(V3 o, V3 d, const double *lo, const double *scale, int bits)
{
  const double cells = (double) (1u << bits);
  const double p[3] = {o.x, o.y, o.z};
  uint32_t c[3];
  for (int a = 0; a < 3; a++) {
    double x = (p[a] - lo[a]) * scale[a];
    x = x < 0. ? 0. : (x > cells - 1. ? cells - 1. : x);        // (NaN compares false twice: cell 0 below)
    c[a] = x == x ? (uint32_t) x : 0u;
  }
  const uint32_t oct = (d.x < 0. ? 1u : 0u) | (d.y < 0. ? 2u : 0u) | (d.z < 0. ? 4u : 0u);
  return (oct << (3 * bits)) | sort_spread3(c[0]) | (sort_spread3(c[1]) << 1) | (sort_spread3(c[2]) << 2);
}

// A child ray as the shaders leave it: what differs between the three children of a hit.  Their origin is the hit point,
// tmax is 1000 (src/shaders/*: SlTrace's TraceContext), the depth counters follow from the context, and only a refraction child
// carries a filter colour -- kept out of the record so that three of them do not hold 72 registers to the end of the kernel.
struct ChildRay {
  bool want;
  V3 d;
  float T[3];
  float tmin;                  // .001f or .0001f (widened to the same double the reference's literal gives: see emit_child)
  int group;
  uint32_t flags;
};

// writes child `c` to its slot of the next level's queue (block_append4)
__device__ __forceinline__ void emit_child(const ChildRay &c, uint32_t slot, int cxt, V3 o, const DPath &parent, const float *fc,
    uint32_t sample, uint32_t uid, uint32_t tbits, uint32_t key,
    DRay *next_rays, DPath *next_paths, DCounters *cnt, uint32_t capacity, const ShadeParams &sp)
{
  if (!c.want) return;
  if (slot >= capacity) { cnt->overflow = 1; return; }
  DRay r;
  r.o[0] = o.x; r.o[1] = o.y; r.o[2] = o.z;
  r.d[0] = c.d.x; r.d[1] = c.d.y; r.d[2] = c.d.z;
  next_rays[slot] = r;           // (range: tmin by class below, tmax 1000)
  if (sp.next_keys) sp.next_keys[slot] = ray_sort_key(o, c.d, sp.sort_lo, sp.sort_scale, sp.sort_bits);
  DPath p;
  p.sample = sample;
  p.T[0] = c.T[0]; p.T[1] = c.T[1]; p.T[2] = c.T[2];
  p.cxt = (uint8_t) (cxt | (c.tmin < .0005f ? FJ_CXT_TMIN_1E4 : 0u));
  p.ddepth = parent.ddepth + (cxt == CXT_DIFFUSE_RAY ? 1 : 0);
  p.rdepth = parent.rdepth + (cxt == CXT_REFLECT_RAY ? 1 : 0);
  p.tdepth = parent.tdepth + (cxt == CXT_REFRACT_RAY ? 1 : 0);
  p.group = c.group;
  if (fc && (c.flags & 1u) && sp.fc_out) { float *fo = sp.fc_out + 3 * (size_t) slot; fo[0] = fc[0]; fo[1] = fc[1]; fo[2] = fc[2]; }
  p.flags = c.flags | tbits; p.rng = key; p.uid = uid;     // tbits: the sample's time index << 1
  next_paths[slot] = p;
}

// trace_surface's SurfaceInput setup + Shader::Evaluate for the device shaders.
// Radiance is accumulated as throughput-weighted terms: every shader term of
// the reference is linear in the radiance returned by its child SlTrace calls,
// so `Cs = local + sum_k w_k * C_child_k` unrolls into per-path products
// (DESIGN.md 6).
#ifndef FJ_SHADE_MINB
#define FJ_SHADE_MINB 3           // resident blocks per CU the register budget is cut for (169 VGPRs as written = 2 waves; 164 = 3: C3 15.8 -> 14.8 ms, C4 269 -> 258)
#endif
// (Round 3 tried a lean instantiation without the glass / hair / pathtracing paths for scenes of ConstantShader /
// PlasticShader only: 138 VGPRs instead of 168, or 128 with 12 spills for a fourth wave -- C3 14.5 ms either way.  The
// kernel is bound by the attribute gathers themselves, not by occupancy.)
template <bool kMotion>
__global__ void __launch_bounds__(BLOCK, FJ_SHADE_MINB) k_shade(DScene S, ShadeParams sp, const DRay *rays, const DPath *paths,
    const DHit *hits, uint32_t n, float *s_accum, DRay *next_rays, DPath *next_paths,
    DLightRec *lrecs, DCounters *cnt)
{
  const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
  const bool active = i < n;
  DHit h;
  h.inst = -1;
  if (active) h = hits[i];
  const bool hit = active && h.inst >= 0;

  ChildRay c0, c1, c2;   // reflect, refract, diffuse children (glass: c0 + c1, plastic: c0, pathtracing: any)
  c0.want = c1.want = c2.want = false;
  V3 Pw = mk(0, 0, 0);   // the hit point: origin of every child
  float fc1[3] = {1.f, 1.f, 1.f};      // filter colour a refraction child carries to its hit
  DPath p;
  p.ddepth = p.rdepth = p.tdepth = 0;
  bool want_light = false;
  DLightRec lr;
  DLightHair lh;
  lr.kind = 0;
  uint32_t sample = 0, rng = 0, uid = 0, tbits = 0;

  if (hit) {
    // (level 0 with implicit camera rays: nothing was written, ray and path state follow from the sample slot)
    const DRay r = rays ? rays[i] : implicit_camera_ray(S, i);
    if (rays) p = paths[i];
    else if (S.cam_tk) { const uint32_t tid_ = S.cam_tk[2 * (size_t) i], k_ = S.cam_tk[2 * (size_t) i + 1]; p = camera_path(S, S.cam_slot0 + i, k_, sample_uid((int32_t) tid_, k_)); }
    else p = camera_path(S, S.cam_slot0 + i, 0u, 0u);
    sample = p.sample;
    rng = p.rng;
    uid = p.uid;
    tbits = p.flags & ~1u;
    const DInstance *I = &S.instances[h.inst];
    const DPrimSet *P = &S.primsets[I->primset];
    const V3 ro = mk(r.o[0], r.o[1], r.o[2]), rd = mk(r.d[0], r.d[1], r.d[2]);

    // instance matrices: host-built for static instances, evaluated at the ray's time for
    // time-sampled ones (the traversal did the same, so P and N belong to the same pose)
    const double *IM = I->M, *IMinv = I->Minv;
    double tm[12], tmi[12];
    if (kMotion && I->xform >= 0) {
      xform_at(&S.xforms[I->xform], sample_time(S, p.flags >> 1), tm, tmi);
      IM = tm; IMinv = tmi;
    }
    const V3 oo = xpoint(IMinv, ro);
    const V3 od = xvector(IMinv, rd);
    V3 N = mk(0, 0, 0);
    float tu = 0.f, tv = 0.f;
    V3 dPdu = mk(0, 0, 0), dPdv = mk(0, 0, 0);
    float Cd[3] = {1.f, 1.f, 1.f};                                // Intersection default
    bool has_uv = false;
    float t0u = 0, t0v = 0, t1u = 0, t1v = 0, t2u = 0, t2v = 0;
    int i0 = 0, i1 = 0, i2 = 0;
    int sg = 0;
    if (I->sh_type == FJ_PRIMSET_CURVE) {
      // --- Curve::ray_intersect attribute part (src/fj_curve.cc:211-229): dPdv = curve
      // derivative at v_hit, Cd = lerp of the end colours; N / uv / dPdu stay zero
      const size_t sl = (size_t) h.v;
      const double vhit = h.u;
      const FJ_GLOBAL double *cp = FJ_G(double, P->curve_cp) + sl * 12;
      V3 c0 = ld3(cp), c1 = ld3(cp + 3), c2 = ld3(cp + 6), c3 = ld3(cp + 9);
      if (kMotion && P->curve_vel) {          // time_sample (src/fj_curve.cc:392-397): cp += time * velocity
        const double tm_ = sample_time(S, p.flags >> 1);
        const FJ_GLOBAL double *w = FJ_G(double, P->curve_vel) + sl * 12;
        c0 = c0 + tm_ * ld3(w); c1 = c1 + tm_ * ld3(w + 3); c2 = c2 + tm_ * ld3(w + 6); c3 = c3 + tm_ * ld3(w + 9);
      }
      const double uu = 1 - vhit;
      const double da = 2 * uu * uu, db = 4 * uu * vhit, dc = 2 * vhit * vhit;
      dPdv = da * (c1 - c0) + db * (c2 - c1) + dc * (c3 - c2);   // derivative_bezier3, :474-486
      const float tl = (float) vhit;
      const FJ_GLOBAL float *cd = FJ_G(float, P->curve_Cd) + sl * 6;
      Cd[0] = (1 - tl) * cd[0] + tl * cd[3];
      Cd[1] = (1 - tl) * cd[1] + tl * cd[4];
      Cd[2] = (1 - tl) * cd[2] + tl * cd[5];
    } else {
      // --- Mesh::ray_intersect attribute part (src/fj_mesh.cc:267-305) in object space
      // (the face's point indices: only where something is read through them -- a mesh whose normals sit per corner (sh_vN) and that has no
      //  texture coordinates needs none of them here)
      if (!I->sh_vN || I->sh_uv) { const FJ_GLOBAL int32_t *ix = FJ_G(int32_t, I->sh_indices) + 3 * (size_t) h.prim; i0 = ix[0]; i1 = ix[1]; i2 = ix[2]; }
      V3 n0 = mk(0, 0, 0), n1 = n0, n2 = n0;
      // compute_shading_normal, src/fj_mesh.cc:108-120: the mesh's per-corner ("vertex") normals where it has them, else its point normals
      if (I->sh_vN) { const FJ_GLOBAL double *vn = FJ_G(double, I->sh_vN) + 9 * (size_t) h.prim; n0 = ld3(vn); n1 = ld3(vn + 3); n2 = ld3(vn + 6); }
      else if (I->sh_N) { n0 = ld3(FJ_G(double, I->sh_N) + 3 * (size_t) i0); n1 = ld3(FJ_G(double, I->sh_N) + 3 * (size_t) i1); n2 = ld3(FJ_G(double, I->sh_N) + 3 * (size_t) i2); }
      N = (1 - h.u - h.v) * n0 + h.u * n1 + h.v * n2;            // TriComputeNormal, src/fj_triangle.cc:44-49
      has_uv = I->sh_uv != nullptr;
      if (has_uv) {
        t0u = FJ_G(float, I->sh_uv)[2 * (size_t) i0]; t0v = FJ_G(float, I->sh_uv)[2 * (size_t) i0 + 1];
        t1u = FJ_G(float, I->sh_uv)[2 * (size_t) i1]; t1v = FJ_G(float, I->sh_uv)[2 * (size_t) i1 + 1];
        t2u = FJ_G(float, I->sh_uv)[2 * (size_t) i2]; t2v = FJ_G(float, I->sh_uv)[2 * (size_t) i2 + 1];
        const float tt = (float) (1 - h.u - h.v);                  // f32 barycentric, src/fj_mesh.cc:285
        tu = (float) (tt * t0u + h.u * t1u + h.v * t2u);
        tv = (float) (tt * t0v + h.u * t1v + h.v * t2v);
      }
      sg = I->sh_face_group ? FJ_G(int32_t, I->sh_face_group)[h.prim] : 0;
    }
    Pw = oo + h.t * od;                                        // RayPointAt in object space
    // --- ObjectInstance::RayIntersect back-transform (src/fj_object_instance.cc:231-240)
    Pw = xpoint(IM, Pw);
    N = normalize(xvector(IM, N));
    dPdv = xvector(IM, dPdv);

    // --- shader lookup: ObjectInstance::GetShader (src/fj_object_instance.cc:177-191)
    int sid;
    if (sg < 0 || sg >= I->n_shaders) sid = I->shaders[0];
    else { sid = I->shaders[sg]; if (sid < 0) sid = I->shaders[0]; }

    // pending pow(filter, t_hit) of a refraction child (glass_shader.cc:117-121)
    if (p.flags & 1u) {
      const float *pfc = sp.fc_in + 3 * (size_t) i;
      p.T[0] = (float) (p.T[0] * pow((double) pfc[0], h.t));
      p.T[1] = (float) (p.T[1] * pow((double) pfc[1], h.t));
      p.T[2] = (float) (p.T[2] * pow((double) pfc[2], h.t));
    }

    float Cs[3] = {.5f, 1.f, 0.f};   // NO_SHADER_COLOR, src/fj_shading.cc:24
    float Os = 1.f;
    bool add_cs = true;
    const V3 Iw = rd;
    if (sid >= 0) {
      const fj_shader_desc *sh = &S.shaders[sid];
      switch (sh->type) {
      case FJ_SHADER_CONSTANT: {   // constant_shader.cc:72-96
        if (sh->texture >= 0) {
          float ct[4];
          tex_lookup(S.textures[sh->texture], tu, tv, ct);
          Cs[0] = ct[0] * sh->diffuse[0]; Cs[1] = ct[1] * sh->diffuse[1]; Cs[2] = ct[2] * sh->diffuse[2];
        } else { Cs[0] = sh->diffuse[0]; Cs[1] = sh->diffuse[1]; Cs[2] = sh->diffuse[2]; }
        Os = 1.f;
        break;
      }
      case FJ_SHADER_PLASTIC: {    // plastic_shader.cc:101-179
        V3 Nf = faceforward(Iw, N);
        if (sh->bump_map >= 0) {
          if (has_uv) {
            // TriComputeDerivatives (src/fj_triangle.cc:51-74) on the object-space
            // vertices, then the instance's M as a vector transform
            V3 p0 = ld3(FJ_G(double, P->P) + 3 * (size_t) i0), p1 = ld3(FJ_G(double, P->P) + 3 * (size_t) i1), p2 = ld3(FJ_G(double, P->P) + 3 * (size_t) i2);
            if (kMotion && P->velocity) {    // the time-sampled vertices of Mesh::ray_intersect
              const double tm_ = sample_time(S, p.flags >> 1);
              p0 = p0 + tm_ * ld3(FJ_G(double, P->velocity) + 3 * (size_t) i0);
              p1 = p1 + tm_ * ld3(FJ_G(double, P->velocity) + 3 * (size_t) i1);
              p2 = p2 + tm_ * ld3(FJ_G(double, P->velocity) + 3 * (size_t) i2);
            }
            const V3 dP1 = p1 - p0, dP2 = p2 - p0;
            const float du1 = t1u - t0u, du2 = t2u - t0u, dv1 = t1v - t0v, dv2 = t2v - t0v;
            const float determinant = du1 * dv2 - dv1 * du2;
            if (determinant != 0) {
              const float invdet = (float) (1. / determinant);
              dPdu = ((double) dv2 * dP1 - (double) dv1 * dP2) * (double) invdet;
              dPdv = ((double) (-du2) * dP1 + (double) du1 * dP2) * (double) invdet;
            }
            dPdu = xvector(IM, dPdu);
            dPdv = xvector(IM, dPdv);   // (mesh dPdv is zero until here)
          }
          Nf = bump_mapping(S.textures[sh->bump_map], dPdu, dPdv, tu, tv, (double) sh->bump_amplitude, Nf);
        }
        add_cs = false;
        if (S.n_light_samples > 0) {
          float dm[4] = {1.f, 1.f, 1.f, 1.f};
          if (sh->diffuse_map >= 0) tex_lookup(S.textures[sh->diffuse_map], tu, tv, dm);
          want_light = true;
          lr.P[0] = Pw.x; lr.P[1] = Pw.y; lr.P[2] = Pw.z;
          lr.N[0] = Nf.x; lr.N[1] = Nf.y; lr.N[2] = Nf.z;
          lr.W[0] = p.T[0] * (sh->diffuse[0] * dm[0]);
          lr.W[1] = p.T[1] * (sh->diffuse[1] * dm[1]);
          lr.W[2] = p.T[2] * (sh->diffuse[2] * dm[2]);
          lr.sample = sample;
          lr.group = I->shadow_target;
          lr.kind = 0;
          lr.uid = p.uid; lr.key = p.rng; lr.kind |= (int32_t) (p.flags & ~1u);     // (kind bit 0; the time index above it)
          if (lr.W[0] == 0.f && lr.W[1] == 0.f && lr.W[2] == 0.f && !sp.count_all_shadow) want_light = false;
        }
        if (sh->do_reflect && (int) p.rdepth + 1 <= sp.max_reflect_depth) {
          const V3 R = normalize(reflect(Iw, Nf));
          const double Kr = fresnel(Iw, Nf, (double) (1.f / sh->ior));
          c0.want = true;
          c0.d = R; c0.tmin = .001f;
          c0.T[0] = (float) (Kr * sh->reflect[0]) * p.T[0];
          c0.T[1] = (float) (Kr * sh->reflect[1]) * p.T[1];
          c0.T[2] = (float) (Kr * sh->reflect[2]) * p.T[2];
          c0.group = I->reflect_target;
          c0.flags = 0;
        }
        Os = sh->opacity;
        break;
      }
      case FJ_SHADER_GLASS: {      // glass_shader.cc:88-130 (N is not face-forwarded)
        add_cs = false;
        const double Kr = fresnel(Iw, N, (double) (1.f / sh->ior));
        const double Kt = 1 - Kr;
        if ((int) p.rdepth + 1 <= sp.max_reflect_depth) {
          c0.want = true;
          c0.d = normalize(reflect(Iw, N)); c0.tmin = .0001f;
          c0.T[0] = (float) Kr * p.T[0]; c0.T[1] = (float) Kr * p.T[1]; c0.T[2] = (float) Kr * p.T[2];
          c0.group = I->reflect_target;
          c0.flags = 0;
        }
        if ((int) p.tdepth + 1 <= sp.max_refract_depth) {
          c1.want = true;
          c1.d = normalize(refract(Iw, N, (double) (1.f / sh->ior))); c1.tmin = .0001f;
          c1.T[0] = (float) Kt * p.T[0]; c1.T[1] = (float) Kt * p.T[1]; c1.T[2] = (float) Kt * p.T[2];
          c1.group = I->refract_target;
          const bool filt = sh->do_color_filter && dot(Iw, N) < 0;
          fc1[0] = sh->filter_color[0]; fc1[1] = sh->filter_color[1]; fc1[2] = sh->filter_color[2];
          c1.flags = filt ? 1u : 0u;
        }
        Os = 1.f;
        break;
      }
      case FJ_SHADER_HAIR: {       // hair_shader.cc:87-117: Kajiya-Kay over all light samples
        add_cs = false;
        if (S.n_light_samples > 0) {
          const V3 tangent = normalize(dPdv);
          want_light = true;
          lr.P[0] = Pw.x; lr.P[1] = Pw.y; lr.P[2] = Pw.z;
          lr.N[0] = N.x; lr.N[1] = N.y; lr.N[2] = N.z;     // illuminance axis = in.N (zero for curves)
          lh.aux[0] = tangent.x; lh.aux[1] = tangent.y; lh.aux[2] = tangent.z;
          lh.aux[3] = Iw.x; lh.aux[4] = Iw.y; lh.aux[5] = Iw.z;
          lr.W[0] = p.T[0]; lr.W[1] = p.T[1]; lr.W[2] = p.T[2];
          lh.Cd[0] = Cd[0] * sh->diffuse[0]; lh.Cd[1] = Cd[1] * sh->diffuse[1]; lh.Cd[2] = Cd[2] * sh->diffuse[2]; lh.pad = 0;
          lr.sample = sample;
          lr.group = I->shadow_target;
          lr.kind = 1;
          lr.uid = p.uid; lr.key = p.rng; lr.kind |= (int32_t) (p.flags & ~1u);     // (kind bit 0; the time index above it)
        }
        Os = 1.f;
        break;
      }
      case FJ_SHADER_PATHTRACING: {   // pathtracing_shader.cc:125-257 with the counter-based RNG contract
        float Cdm[3] = {Cd[0], Cd[1], Cd[2]};
        V3 Np = N;
        if (sh->diffuse_map >= 0) {
          float dm[4];
          tex_lookup(S.textures[sh->diffuse_map], tu, tv, dm);
          Cdm[0] *= dm[0]; Cdm[1] *= dm[1]; Cdm[2] *= dm[2];
        }
        if (sh->bump_map >= 0) {
          if (has_uv) {
            V3 p0 = ld3(FJ_G(double, P->P) + 3 * (size_t) i0), p1 = ld3(FJ_G(double, P->P) + 3 * (size_t) i1), p2 = ld3(FJ_G(double, P->P) + 3 * (size_t) i2);
            if (kMotion && P->velocity) {    // the time-sampled vertices of Mesh::ray_intersect
              const double tm_ = sample_time(S, p.flags >> 1);
              p0 = p0 + tm_ * ld3(FJ_G(double, P->velocity) + 3 * (size_t) i0);
              p1 = p1 + tm_ * ld3(FJ_G(double, P->velocity) + 3 * (size_t) i1);
              p2 = p2 + tm_ * ld3(FJ_G(double, P->velocity) + 3 * (size_t) i2);
            }
            const V3 dP1 = p1 - p0, dP2 = p2 - p0;
            const float du1 = t1u - t0u, du2 = t2u - t0u, dv1 = t1v - t0v, dv2 = t2v - t0v;
            const float determinant = du1 * dv2 - dv1 * du2;
            if (determinant != 0) {
              const float invdet = (float) (1. / determinant);
              dPdu = ((double) dv2 * dP1 - (double) dv1 * dP2) * (double) invdet;
              dPdv = ((double) (-du2) * dP1 + (double) du1 * dP2) * (double) invdet;
            }
            dPdu = xvector(IM, dPdu);
            dPdv = xvector(IM, dPdv);
          }
          Np = bump_mapping(S.textures[sh->bump_map], dPdu, dPdv, tu, tv, (double) sh->bump_amplitude, N);
        }
        Cs[0] = sh->emission[0]; Cs[1] = sh->emission[1]; Cs[2] = sh->emission[2];   // Le, added below
        if (luminance3(sh->diffuse) > 0.f && (int) p.ddepth + 1 <= sp.max_diffuse_depth) {   // integrate_diffuse
          const V3 w = Np;
          V3 u = fabs(w.x) > .001 ? mk(0, 1, 0) : mk(1, 0, 0);
          u = normalize(cross(u, w));
          const V3 v = cross(w, u);
          double x1, x2;
          pt_draw2(uid, rng, &x1, &x2);
          const double r1 = 2. * 3.14159265358979323846 * x1;
          const double r2 = x2;
          const double r2sqrt = sqrt(r2);
          double sin_r1, cos_r1;
          sincos(r1, &sin_r1, &cos_r1);       // (one argument reduction for the two; the same values as sin() and cos())
          const V3 D = normalize(u * cos_r1 * r2sqrt + v * sin_r1 * r2sqrt + w * sqrt(1. - r2));
          const float kd = (float) dot(Np, D);
          c2.want = true;
          c2.d = D; c2.tmin = .001f;
          c2.T[0] = p.T[0] * (Cdm[0] * kd * sh->diffuse[0]);
          c2.T[1] = p.T[1] * (Cdm[1] * kd * sh->diffuse[1]);
          c2.T[2] = p.T[2] * (Cdm[2] * kd * sh->diffuse[2]);
          c2.group = I->reflect_target;                              // SlDiffuseContext uses the REFLECT target
          c2.flags = 0;
        }
        if (luminance3(sh->reflect) > 0.f && (int) p.rdepth + 1 <= sp.max_reflect_depth) {   // integrate_reflect
          const float kr = (float) fresnel(Iw, Np, 1. / (double) sh->ior);
          c0.want = true;
          c0.d = normalize(reflect(Iw, Np)); c0.tmin = .001f;
          c0.T[0] = p.T[0] * (kr * sh->reflect[0]); c0.T[1] = p.T[1] * (kr * sh->reflect[1]); c0.T[2] = p.T[2] * (kr * sh->reflect[2]);
          c0.group = I->reflect_target;
          c0.flags = 0;
        }
        if (luminance3(sh->refract) > 0.f && (int) p.tdepth + 1 <= sp.max_refract_depth) {   // integrate_refract
          const double Kr = fresnel(Iw, Np, (double) (1 / sh->ior));      // 1/ior in f32, as in the plugin
          const float kt = (float) (1 - Kr);
          c1.want = true;
          c1.d = normalize(refract(Iw, Np, 1. / (double) sh->ior)); c1.tmin = .0001f;
          c1.T[0] = p.T[0] * (kt * sh->refract[0]); c1.T[1] = p.T[1] * (kt * sh->refract[1]); c1.T[2] = p.T[2] * (kt * sh->refract[2]);
          c1.group = I->refract_target;
          const bool filt = sh->do_color_filter && dot(Iw, Np) < 0;
          fc1[0] = sh->filter_color[0]; fc1[1] = sh->filter_color[1]; fc1[2] = sh->filter_color[2];
          c1.flags = filt ? 1u : 0u;
        }
        Os = 1.f;
        break;
      }
      default:
        add_cs = false;
        break;
      }
    }
    Os = (float) clampd(Os, 0, 1);
    float *acc = s_accum + 4 * (size_t) sample;
    if (add_cs) {
      const float r0 = p.T[0] * Cs[0], r1 = p.T[1] * Cs[1], r2 = p.T[2] * Cs[2];
      if (r0 != 0.f) atomicAdd(acc + 0, r0);
      if (r1 != 0.f) atomicAdd(acc + 1, r1);
      if (r2 != 0.f) atomicAdd(acc + 2, r2);
    }
    if ((p.cxt & 0x7fu) == CXT_CAMERA_RAY) acc[3] = Os;   // one camera ray per sample
  }

  // ---- compaction: ballot + prefix count, one atomic per block and queue
  __shared__ uint32_t s_tmp[4 * (BLOCK / 64) + 2];
  const AppendSlots slots = block_append4(want_light, c2.want, c0.want, c1.want, cnt, s_tmp);
  const uint32_t lslot = slots.light;
  if (want_light) {
    if (lslot < sp.light_capacity) {
      lrecs[lslot] = lr;
      if ((lr.kind & 1) && S.lrec_hair) S.lrec_hair[lslot] = lh;
    }
    else cnt->overflow = 1;
  }
  emit_child(c2, slots.c2, CXT_DIFFUSE_RAY, Pw, p, nullptr, sample, uid, tbits, 4 * rng + 1, next_rays, next_paths, cnt, sp.ray_capacity, sp);
  emit_child(c0, slots.c0, CXT_REFLECT_RAY, Pw, p, nullptr, sample, uid, tbits, 4 * rng + 2, next_rays, next_paths, cnt, sp.ray_capacity, sp);
  emit_child(c1, slots.c1, CXT_REFRACT_RAY, Pw, p, fc1, sample, uid, tbits, 4 * rng + 3, next_rays, next_paths, cnt, sp.ray_capacity, sp);
}

#endif
