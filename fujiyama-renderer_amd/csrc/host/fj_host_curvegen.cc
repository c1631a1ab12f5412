// fj_host_curvegen.cc -- built-in CurveGeneratorProcedure + Curve bounds.
//
// Host-side geometry producer of BASELINE config 5 (fur on a mesh), with the
// behaviour of the reference's
//   procedures/curve_generator_procedure/curve_generator_procedure.cc:133-270
//   (generate_curve: curves per face = int(100000 * area), libc srand/rand seeded
//    per (face, curve) and per (curve, vertex), gravity-bent normals, Perlin-noise
//    colour between a dark and a light fur tone)
//   src/fj_noise.cc:34-128  (Ken Perlin's "improved noise", two octaves)
//   src/fj_curve.cc:124-146 (Curve::ComputeBounds)
// It runs before RenderScene and only writes Curve attributes.
#include "fj_host.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <thread>

namespace fjhost {

namespace {

struct V { double x, y, z; };
inline V operator+(V a, V b) { return V{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V operator-(V a, V b) { return V{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V operator*(V a, double s) { return V{a.x * s, a.y * s, a.z * s}; }
inline V operator*(double s, V a) { return a * s; }
inline V mulv(V a, V b) { return V{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline double dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V cross(V a, V b) { return V{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V normalize(V a)
{
  const double len = std::sqrt(dot(a, a));
  if (len == 0) return a;
  const double inv = 1. / len;
  return a * inv;
}
inline V at(const std::vector<double> &a, int i) { return V{a[3 * i], a[3 * i + 1], a[3 * i + 2]}; }

// Ken Perlin's reference permutation (Improved Noise, 2002), repeated twice
const unsigned char kPerm[256] = {
  151,160,137,91,90,15,131,13,201,95,96,53,194,233,7,225,140,36,103,30,69,142,8,99,37,240,21,10,23,190,6,148,
  247,120,234,75,0,26,197,62,94,252,219,203,117,35,11,32,57,177,33,88,237,149,56,87,174,20,125,136,171,168,
  68,175,74,165,71,134,139,48,27,166,77,146,158,231,83,111,229,122,60,211,133,230,220,105,92,41,55,46,245,40,
  244,102,143,54,65,25,63,161,1,216,80,73,209,76,132,187,208,89,18,169,200,196,135,130,116,188,159,86,164,100,
  109,198,173,186,3,64,52,217,226,250,124,123,5,202,38,147,118,126,255,82,85,212,207,206,59,227,47,16,58,17,
  182,189,28,42,223,183,170,213,119,248,152,2,44,154,163,70,221,153,101,155,167,43,172,9,129,22,39,253,19,98,
  108,110,79,113,224,232,178,185,112,104,218,246,97,228,251,34,242,193,238,210,144,12,191,179,162,241,81,51,
  145,235,249,14,239,107,49,192,214,31,181,199,106,157,184,84,204,176,115,121,50,45,127,4,150,254,138,236,205,
  93,222,114,67,29,24,72,243,141,128,195,78,66,215,61,156,180};
inline int perm(int i) { return kPerm[i & 255]; }

inline double fade(double t) { return t * t * t * (t * (t * 6 - 15) + 10); }
inline double nlerp(double t, double a, double b) { return a + t * (b - a); }
inline double grad(int hash, double x, double y, double z)
{
  const int h = hash & 15;
  const double u = h < 8 ? x : y;
  const double v = h < 4 ? y : (h == 12 || h == 14) ? x : z;
  return ((h & 1) == 0 ? u : -u) + ((h & 2) == 0 ? v : -v);
}

double periodic_noise3d(double x, double y, double z)   // src/fj_noise.cc:85-128
{
  const int X = (int) std::floor(x) & 255, Y = (int) std::floor(y) & 255, Z = (int) std::floor(z) & 255;
  const double xx = x - std::floor(x), yy = y - std::floor(y), zz = z - std::floor(z);
  const double u = fade(xx), v = fade(yy), w = fade(zz);
  const int A = perm(X) + Y, AA = perm(A) + Z, AB = perm(A + 1) + Z;
  const int B = perm(X + 1) + Y, BA = perm(B) + Z, BB = perm(B + 1) + Z;
  return nlerp(w,
      nlerp(v, nlerp(u, grad(perm(AA), xx, yy, zz), grad(perm(BA), xx - 1, yy, zz)),
               nlerp(u, grad(perm(AB), xx, yy - 1, zz), grad(perm(BB), xx - 1, yy - 1, zz))),
      nlerp(v, nlerp(u, grad(perm(AA + 1), xx, yy, zz - 1), grad(perm(BA + 1), xx - 1, yy, zz - 1)),
               nlerp(u, grad(perm(AB + 1), xx, yy - 1, zz - 1), grad(perm(BB + 1), xx - 1, yy - 1, zz - 1))));
}

double perlin_noise(V P, double lacunarity, double persistence, int octaves)   // :34-49
{
  double value = 0, amp = 1;
  for (int i = 0; i < octaves; i++) {
    value += amp * periodic_noise3d(P.x, P.y, P.z);
    amp *= persistence;
    P = P * lacunarity;
  }
  return value;
}

double smooth_step(double a, double b, double x)        // src/fj_numeric.h:76-88
{
  const double t = (x - a) / (b - a);
  if (t <= 0) return 0;
  if (t >= 1) return 1;
  return t * t * (3 - 2 * t);
}

// PerlinNoise3d, src/fj_noise.cc:50-66: three decorrelated scalar fields
V perlin_noise3d(V P, double lacunarity, double persistence, int octaves)
{
  V out;
  out.x = perlin_noise(P, lacunarity, persistence, octaves);
  out.y = perlin_noise(P + V{131.977, 21.1823, 71.0231}, lacunarity, persistence, octaves);
  out.z = perlin_noise(P + V{237.492, 11.1312, 133.129}, lacunarity, persistence, octaves);
  return out;
}

}  // namespace

// VelocityGeneratorProcedure (procedures/velocity_generator_procedure/
// velocity_generator_procedure.cc:98-121): a noise velocity field that fades out along z,
// then ComputeNormals + ComputeBounds (the bounds now include the end-of-shutter positions)
int RunVelocityGenerator(Scene *sc, Procedure *proc, std::string *err)
{
  if (proc->mesh < 0) { *err = "VelocityGeneratorProcedure: no mesh assigned"; return -1; }
  Mesh &mesh = *sc->meshes[proc->mesh];
  const int n = mesh.point_count();
  const double zmin = mesh.bounds[2], zmax = mesh.bounds[5];
  mesh.velocity.assign((size_t) n * 3, 0.);
  for (int i = 0; i < n; i++) {
    const V pos = at(mesh.P, i);
    const double znml = (pos.z - zmin) / (zmax - zmin);
    const double vscale = .2 * (1 - smooth_step(.2, .7, znml));
    const V noise_vec = perlin_noise3d(.2 * pos, 2, .5, 1);
    const V vel = vscale * noise_vec;
    mesh.velocity[3 * i] = vel.x; mesh.velocity[3 * i + 1] = vel.y; mesh.velocity[3 * i + 2] = vel.z;
  }
  mesh.ComputeNormals();
  mesh.ComputeBounds();
  return 0;
}

// Curve::ComputeBounds, src/fj_curve.cc:124-146: union of the per-curve bounds (control
// points +- that curve's max radius, at shutter open and close), expanded once more by
// the largest radius of the set
void Curve::ComputeBounds()
{
  const double big = 1.7976931348623157e308;
  double mn[3] = {big, big, big}, mx[3] = {-big, -big, -big};
  double max_radius = 0;
  for (size_t c = 0; c < indices.size(); c++) {
    const int i0 = indices[c];
    const double w0 = width[i0], w1 = width[i0 + 3];
    const double radius = .5 * (w0 > w1 ? w0 : w1);
    for (int pass = 0; pass < 2; pass++) {
      double bmn[3] = {big, big, big}, bmx[3] = {-big, -big, -big};
      for (int k = 0; k < 4; k++)
        for (int a = 0; a < 3; a++) {
          double p = P[3 * (i0 + k) + a];
          if (pass == 1 && !velocity.empty()) p = p + 1 * velocity[3 * (i0 + k) + a];
          if (p < bmn[a]) bmn[a] = p;
          if (p > bmx[a]) bmx[a] = p;
        }
      for (int a = 0; a < 3; a++) {
        const double lo = bmn[a] - radius, hi = bmx[a] + radius;
        if (lo < mn[a]) mn[a] = lo;
        if (hi > mx[a]) mx[a] = hi;
      }
    }
    if (radius > max_radius) max_radius = radius;
  }
  for (int a = 0; a < 3; a++) { bounds[a] = mn[a] - max_radius; bounds[3 + a] = mx[a] + max_radius; }
}

// ---------------------------------------------------------------- CurveGeneratorProcedure
// What the reference's procedure DOES (procedures/curve_generator_procedure/curve_generator_procedure.cc): fur mode (:133-270) grows
// int(100000 * area) one-cubic strands per face -- root at random barycentrics, direction = the interpolated normal pulled down by a random
// "gravity", three inner control points jittered, width .003 -> .0001, colour between two fur tones by two octaves of Perlin noise --, every
// random number being the FIRST values of libc's rand() after an srand() with a seed computed from (face, strand) or (strand, control point);
// hair mode (:280-452) grows strands of five chained cubics from the upper front of the mesh, positions drawn from one default-seeded
// XorShift, bent by Perlin noise and a downward pull, with per-control-point velocities.  The fur has to be the REFERENCE's fur (C5 parity),
// so the arithmetic per strand is the reference's; the organisation is this library's own:
//   * a strand table (first strand of every face by prefix sum), so every strand is an independent job on the host threads;
//   * SeededRand: glibc's rand() stream after srand(seed) computed locally (TYPE_3 additive-feedback generator seeded by the Park-Miller
//     LCG, 310 outputs discarded -- stdlib/random_r.c), thread-safe and the same on any libc: the reference's goldens were made with glibc;
//   * results written straight into the Curve's flat arrays (P, width, Cd, velocity, indices).
// glibc: srand(seed) then rand(), rand(), ... (TYPE_3: r[i] = r[i-3] + r[i-31], output r[i] >> 1)
class SeededRand {
 public:
  explicit SeededRand(uint32_t seed)
  {
    int32_t word = seed ? (int32_t) seed : 1;
    r_[0] = (uint32_t) word;
    for (int i = 1; i < 31; i++) {
      const long hi = word / 127773, lo = word % 127773;
      long w = 16807 * lo - 2836 * hi;
      if (w < 0) w += 2147483647;
      word = (int32_t) w;
      r_[i] = (uint32_t) word;
    }
    f_ = 3; b_ = 0;
    for (int i = 0; i < 310; i++) (void) next();
  }
  uint32_t next()
  {
    r_[f_] += r_[b_];
    const uint32_t out = r_[f_] >> 1;
    f_ = f_ == 30 ? 0 : f_ + 1;
    b_ = b_ == 30 ? 0 : b_ + 1;
    return out;
  }
  double unit() { return (double) next() / 2147483647.0; }       // ((double) rand()) / RAND_MAX
 private:
  uint32_t r_[31];
  int f_, b_;
};

namespace {

// srand()'s argument is `unsigned`: the reference passes doubles and ints, converted the C way
inline uint32_t seed_of(double x) { return (uint32_t) x; }
inline uint32_t seed_of(long x) { return (uint32_t) x; }

struct XorShiftHost {              // src/fj_random.cc:10-43, default seed
  uint32_t s[4] = {123456789u, 362436069u, 521288629u, 88675123u};
  double f01()
  {
    const uint32_t t = s[0] ^ (s[0] << 11);
    s[0] = s[1]; s[1] = s[2]; s[2] = s[3];
    s[3] = (s[3] ^ (s[3] >> 19)) ^ (t ^ (t >> 8));
    return (double) s[3] / 4294967295u;
  }
};

struct Corner3 { V p[3], n[3]; };
inline Corner3 face_corners(const Mesh &mesh, int face)
{
  Corner3 c;
  const int32_t *ix = &mesh.indices[3 * (size_t) face];
  const bool has_n = !mesh.N.empty();
  for (int k = 0; k < 3; k++) { c.p[k] = at(mesh.P, ix[k]); c.n[k] = has_n ? at(mesh.N, ix[k]) : V{0, 0, 0}; }
  return c;
}
inline double face_area(const Corner3 &c)                     // TriComputeArea
{
  const V x = cross(c.p[1] - c.p[0], c.p[2] - c.p[0]);
  return .5 * std::sqrt(dot(x, x));
}
inline void put3(std::vector<double> &a, size_t i, V v) { a[3 * i] = v.x; a[3 * i + 1] = v.y; a[3 * i + 2] = v.z; }

// strands per face -> first strand of every face; returns the total (or -1: too many)
long strand_table(const std::vector<int> &per_face, std::vector<long> *first)
{
  first->assign(per_face.size() + 1, 0);
  for (size_t f = 0; f < per_face.size(); f++) (*first)[f + 1] = (*first)[f] + per_face[f];
  return first->back();
}

// jobs [0, n) on the host threads (strands are independent of each other)
template <class F> void for_each_job(long n, F body)
{
  const unsigned hw = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
  const unsigned nt = (unsigned) std::max<long>(1, std::min<long>(hw, n / 2048 + 1));
  if (nt == 1) { for (long i = 0; i < n; i++) body(i); return; }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++) th.emplace_back([=]() { for (long i = n * t / nt; i < n * (t + 1) / nt; i++) body(i); });
  for (auto &x : th) x.join();
}

// the face a strand belongs to (binary search in the table)
inline int face_of(const std::vector<long> &first, long strand)
{
  return (int) (std::upper_bound(first.begin(), first.end(), strand) - first.begin()) - 1;
}

}  // namespace

// hair mode: strands of `kLinks` chained cubics; the (u, v) of every root comes from ONE serial XorShift stream (drawn here in strand order,
// two numbers per strand), everything after that is per strand
static int GenerateHair(const Mesh &mesh, Curve &curve, std::string *err)
{
  const int kLinks = 5;
  const double ylo = mesh.bounds[1], yhi = mesh.bounds[4], zlo = mesh.bounds[2], zhi = mesh.bounds[5];
  const int n_faces = mesh.face_count();
  std::vector<int> per_face(n_faces);
  for (int f = 0; f < n_faces; f++) {
    const Corner3 c = face_corners(mesh, f);
    const double yn = ((c.p[0].y + c.p[1].y + c.p[2].y) / 3. - ylo) / (yhi - ylo);
    const double zn = ((c.p[0].z + c.p[1].z + c.p[2].z) / 3. - zlo) / (zhi - zlo);
    per_face[f] = 100000 * face_area(c);
    if (yn < .5 || zn > .78) per_face[f] = 0;             // the upper, front part of the mesh only
  }
  std::vector<long> first;
  const long n_strands = strand_table(per_face, &first);
  if (n_strands * kLinks > 50000000) { *err = "CurveGeneratorProcedure: more than 5e7 curves requested"; return -1; }
  const size_t n_cubics = (size_t) n_strands * kLinks, n_cp = 4 * n_cubics;
  curve.P.assign(n_cp * 3, 0.);
  curve.width.assign(n_cp, 0.);
  curve.Cd.assign(n_cp * 3, 0.f);
  curve.velocity.assign(n_cp * 3, 0.);
  curve.indices.assign(n_cubics, 0);
  curve.uv.clear();

  std::vector<double> uv(2 * (size_t) n_strands);
  { XorShiftHost rng; for (long s = 0; s < n_strands; s++) { const double u = rng.f01(); uv[2 * s] = u; uv[2 * s + 1] = (1 - u) * rng.f01(); } }

  for_each_job(n_strands, [&](long s) {
    const Corner3 c = face_corners(mesh, face_of(first, s));
    const double u = uv[2 * s], v = uv[2 * s + 1], t = 1 - u - v;
    V at_p = t * c.p[0] + u * c.p[1] + v * c.p[2];
    V dir = normalize(t * c.n[0] + u * c.n[1] + v * c.n[2]);
    dir.y = dir.y < .1 ? dir.y : .1;
    if (dir.x < .1 && dir.z < .1) {            // (the reference divides a component by itself here: 1, or NaN for 0)
      dir.x /= dir.x; dir.z /= dir.z;
      dir.x *= .5; dir.z *= .5;
    }
    dir = normalize(dir);
    size_t cp = 4 * (size_t) s * kLinks;
    for (int link = 0; link < kLinks; link++) {
      curve.indices[(size_t) s * kLinks + link] = (int32_t) cp;
      for (int k = 0; k < 4; k++, cp++) {
        static const double taper[4] = {1, .5, .2, .05};
        put3(curve.P, cp, at_p);
        curve.Cd[3 * cp] = .9f; curve.Cd[3 * cp + 1] = .8f; curve.Cd[3 * cp + 2] = .5f;
        curve.width[cp] = (link == kLinks - 1) ? .0005 * taper[k] : .0005;
        const V here = at_p;
        if (k != 3) {
          // the next control point: a step along the strand plus noise in x and z, then the direction is pulled down
          const V wob = perlin_noise3d(mulv(here, V{100, 2, 100}), 2, .5, 2);
          at_p = at_p + (.01 * dir + mulv(V{.002 * .1, 0, .002 * .1}, wob));
          dir = normalize(at_p - here);
          dir.y += -.5;
          dir = normalize(dir);
        }
        const V drift = perlin_noise3d(1 * here + V{0, 5, 0}, 2, .5, 2);
        put3(curve.velocity, cp, (smooth_step(1, kLinks, link) * .01) * drift);
      }
    }
  });
  curve.ComputeBounds();
  return 0;
}

// fur mode
static int GenerateFur(const Mesh &mesh, Curve &curve, std::string *err)
{
  const int n_faces = mesh.face_count();
  std::vector<int> per_face(n_faces);
  for (int f = 0; f < n_faces; f++) per_face[f] = 100000 * face_area(face_corners(mesh, f));
  std::vector<long> first;
  const long n_strands = strand_table(per_face, &first);
  if (n_strands > 50000000) { *err = "CurveGeneratorProcedure: more than 5e7 curves requested"; return -1; }
  const size_t n_cp = 4 * (size_t) n_strands;
  curve.P.assign(n_cp * 3, 0.);
  curve.width.assign(n_cp, 0.);
  curve.Cd.assign(n_cp * 3, 0.f);
  curve.indices.assign((size_t) n_strands, 0);
  curve.uv.clear();
  curve.velocity.clear();

  for_each_job(n_strands, [&](long s) {
    const int f = face_of(first, s);
    const long j = s - first[f];                                   // the strand's number on its face
    const Corner3 c = face_corners(mesh, f);
    // root and direction: three seeds per (face, strand), one draw each
    const double u = SeededRand(seed_of(12.34 * f + 1232 * j)).unit();
    const double v = (1 - u) * SeededRand(seed_of(21.43 * f + 213 * j)).unit();
    const double t = 1 - u - v;
    const V root = t * c.p[0] + u * c.p[1] + v * c.p[2];
    V lean = normalize(t * c.n[0] + u * c.n[1] + v * c.n[2]);
    lean.y -= .5 + .5 * SeededRand(seed_of((long) f + j)).unit();  // "gravity"
    lean = normalize(lean);
    // colour: one value per strand (the reference evaluates it per control point from the same root)
    double tone = 1 * perlin_noise(mulv(root, V{3, 3, 3}) + V{0, 1, 0}, 2, .5, 2);
    tone = smooth_step(.55, .75, tone);
    const float w = (float) tone;                                   // Lerp(Color, Color, float)
    static const float dark[3] = {.8f, .5f, .3f}, light[3] = {.9f, .88f, .85f};
    static const double width[4] = {.003, .002, .001, .0001};
    const double len = .02;
    for (int k = 0; k < 4; k++) {
      const size_t cp = 4 * (size_t) s + k;
      V jitter{0, 0, 0};
      if (k > 0) { SeededRand r(seed_of(12 * s + 49 * (long) k)); jitter.x = r.unit(); jitter.y = r.unit(); jitter.z = r.unit(); }
      put3(curve.P, cp, root + (.75 * len) * jitter + k * len / 3. * lean);
      curve.width[cp] = width[k];
      for (int a = 0; a < 3; a++) curve.Cd[3 * cp + a] = (1 - w) * dark[a] + w * light[a];
    }
    curve.indices[(size_t) s] = (int32_t) (4 * s);
  });
  curve.ComputeBounds();
  return 0;
}

}  // namespace fjhost

// Diagnostics (include/fj_scene_interface.h): the first n values of the rand() stream the fur generator draws after srand(seed), so that a test
// can hold them against the C library's own (tests/test_host_boundary.py)
extern "C" void fj_dev_seeded_rand(uint32_t seed, int n, uint32_t *out)
{
  fjhost::SeededRand r(seed);
  for (int i = 0; i < n; i++) out[i] = r.next();
}

namespace fjhost {

int RunCurveGenerator(Scene *sc, Procedure *proc, std::string *err)
{
  if (proc->mesh < 0 || proc->curve < 0) { *err = "CurveGeneratorProcedure: mesh and curve must be assigned"; return -1; }
  const Mesh &mesh = *sc->meshes[proc->mesh];
  Curve &curve = *sc->curves[proc->curve];
  auto hair = proc->numbers.find("is_hair");
  const bool is_hair = hair != proc->numbers.end() && !hair->second.empty() && hair->second[0] > 0;
  return is_hair ? GenerateHair(mesh, curve, err) : GenerateFur(mesh, curve, err);
}

}  // namespace fjhost
