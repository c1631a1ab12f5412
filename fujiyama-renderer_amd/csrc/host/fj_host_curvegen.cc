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

#include <cmath>
#include <cstdlib>

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

inline double unit_rand() { return ((double) std::rand()) / RAND_MAX; }

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

// generate_hair (procedures/curve_generator_procedure/curve_generator_procedure.cc:280-452):
// strands of five chained cubics grown from the upper, front part of the mesh, bent by
// Perlin noise and a downward pull, with per-vertex velocities (hair_velocity_blur.py).
// Positions come from ONE default-seeded XorShift (deterministic), unlike the fur mode's rand().
namespace {
struct XorShiftHost {
  uint32_t s[4] = {123456789u, 362436069u, 521288629u, 88675123u};
  double f01()
  {
    const uint32_t t = s[0] ^ (s[0] << 11);
    s[0] = s[1]; s[1] = s[2]; s[2] = s[3];
    s[3] = (s[3] ^ (s[3] >> 19)) ^ (t ^ (t >> 8));
    return (double) s[3] / 4294967295u;
  }
};
}  // namespace

static int GenerateHair(const Mesh &mesh, Curve &curve, std::string *err)
{
  const double ymin = mesh.bounds[1], ymax = mesh.bounds[4], zmin = mesh.bounds[2], zmax = mesh.bounds[5];
  const int FACE_COUNT = mesh.face_count();
  const int N_CURVES_PER_HAIR = 5;
  std::vector<int> ncurves_on_face(FACE_COUNT);
  long total = 0;
  for (int i = 0; i < FACE_COUNT; i++) {
    const int32_t *ix = &mesh.indices[3 * i];
    const V P0 = at(mesh.P, ix[0]), P1 = at(mesh.P, ix[1]), P2 = at(mesh.P, ix[2]);
    const V c = cross(P1 - P0, P2 - P0);
    const double area = .5 * std::sqrt(dot(c, c));           // TriComputeArea
    const double ycenter = (P0.y + P1.y + P2.y) / 3.;
    const double ynml = (ycenter - ymin) / (ymax - ymin);
    const double zcenter = (P0.z + P1.z + P2.z) / 3.;
    const double znml = (zcenter - zmin) / (zmax - zmin);
    ncurves_on_face[i] = 100000 * area;
    if (ynml < .5 || znml > .78) ncurves_on_face[i] = 0;
    total += (long) ncurves_on_face[i] * N_CURVES_PER_HAIR;
  }
  if (total > 50000000) { *err = "CurveGeneratorProcedure: more than 5e7 curves requested"; return -1; }
  const int total_ncurves = (int) total;
  const int total_ncps = 4 * total_ncurves;
  curve.P.assign((size_t) total_ncps * 3, 0.);
  curve.width.assign(total_ncps, 0.);
  curve.Cd.assign((size_t) total_ncps * 3, 0.f);
  curve.velocity.assign((size_t) total_ncps * 3, 0.);
  curve.indices.assign(total_ncurves, 0);
  curve.uv.clear();

  XorShiftHost rng;
  const bool has_N = !mesh.N.empty();
  int curve_id = 0, cp_id = 0;
  for (int i = 0; i < FACE_COUNT; i++) {
    const int32_t *ix = &mesh.indices[3 * i];
    const V P0 = at(mesh.P, ix[0]), P1 = at(mesh.P, ix[1]), P2 = at(mesh.P, ix[2]);
    const V zero{0, 0, 0};
    const V N0 = has_N ? at(mesh.N, ix[0]) : zero, N1 = has_N ? at(mesh.N, ix[1]) : zero, N2 = has_N ? at(mesh.N, ix[2]) : zero;
    for (int j = 0; j < ncurves_on_face[i]; j++) {
      const double u = rng.f01();
      const double v = (1 - u) * rng.f01();
      const double t = 1 - u - v;
      const V src_P = t * P0 + u * P1 + v * P2;
      V src_N = normalize(t * N0 + u * N1 + v * N2);
      src_N.y = src_N.y < .1 ? src_N.y : .1;                 // Min(src_N.y, .1)
      if (src_N.x < .1 && src_N.z < .1) {
        src_N.x /= src_N.x;                                   // (sic) 1, or NaN for 0
        src_N.z /= src_N.z;
        src_N.x *= .5;
        src_N.z *= .5;
      }
      src_N = normalize(src_N);
      V next_P = src_P, next_N = src_N;
      for (int k = 0; k < N_CURVES_PER_HAIR; k++) {
        curve.indices[curve_id] = cp_id;
        for (int vtx = 0; vtx < 4; vtx++) {
          const double w[4] = {1, .5, .2, .05};
          curve.P[3 * cp_id] = next_P.x; curve.P[3 * cp_id + 1] = next_P.y; curve.P[3 * cp_id + 2] = next_P.z;
          curve.Cd[3 * cp_id] = .9f; curve.Cd[3 * cp_id + 1] = .8f; curve.Cd[3 * cp_id + 2] = .5f;
          curve.width[cp_id] = (k == N_CURVES_PER_HAIR - 1) ? .0005 * w[vtx] : .0005;
          const V curr_P = next_P;
          if (vtx != 3) {
            const double amp = .002 * .1, freq = 100, segment_len = .01;
            const V Q = mulv(curr_P, V{freq, 2, freq});
            const V noise_vec = perlin_noise3d(Q, 2, .5, 2);
            next_P = next_P + (segment_len * next_N + mulv(V{amp, 0, amp}, noise_vec));
            next_N = normalize(next_P - curr_P);
            next_N.y += -.5;
            next_N = normalize(next_N);
          }
          {
            const double amp = .01, freq = 1;
            const V Q = freq * curr_P + V{0, 5, 0};
            const V noise_vec = perlin_noise3d(Q, 2, .5, 2);
            const double vmult = smooth_step(1, N_CURVES_PER_HAIR, k);
            const V curr_v = (vmult * amp) * noise_vec;
            curve.velocity[3 * cp_id] = curr_v.x; curve.velocity[3 * cp_id + 1] = curr_v.y; curve.velocity[3 * cp_id + 2] = curr_v.z;
          }
          cp_id++;
        }
        curve_id++;
      }
    }
  }
  curve.ComputeBounds();
  return 0;
}

int RunCurveGenerator(Scene *sc, Procedure *proc, std::string *err)
{
  if (proc->mesh < 0 || proc->curve < 0) { *err = "CurveGeneratorProcedure: mesh and curve must be assigned"; return -1; }
  auto hair = proc->numbers.find("is_hair");
  if (hair != proc->numbers.end() && !hair->second.empty() && hair->second[0] > 0)
    return GenerateHair(*sc->meshes[proc->mesh], *sc->curves[proc->curve], err);
  const Mesh &mesh = *sc->meshes[proc->mesh];
  Curve &curve = *sc->curves[proc->curve];
  const int FACE_COUNT = mesh.face_count();

  std::vector<int> ncurves_on_face(FACE_COUNT);
  long total = 0;
  for (int i = 0; i < FACE_COUNT; i++) {
    const int32_t *ix = &mesh.indices[3 * i];
    const V P0 = at(mesh.P, ix[0]), P1 = at(mesh.P, ix[1]), P2 = at(mesh.P, ix[2]);
    const V c = cross(P1 - P0, P2 - P0);
    const double area = .5 * std::sqrt(dot(c, c));           // TriComputeArea
    ncurves_on_face[i] = 100000 * area;
    total += ncurves_on_face[i];
  }
  if (total > 50000000) { *err = "CurveGeneratorProcedure: more than 5e7 curves requested"; return -1; }
  const int total_ncurves = (int) total;
  const int total_ncps = 4 * total_ncurves;

  std::vector<V> sourceP(total_ncurves), sourceN(total_ncurves);
  int curve_id = 0;
  const bool has_N = !mesh.N.empty();
  for (int i = 0; i < FACE_COUNT; i++) {
    const int32_t *ix = &mesh.indices[3 * i];
    const V P0 = at(mesh.P, ix[0]), P1 = at(mesh.P, ix[1]), P2 = at(mesh.P, ix[2]);
    const V zero{0, 0, 0};
    const V N0 = has_N ? at(mesh.N, ix[0]) : zero, N1 = has_N ? at(mesh.N, ix[1]) : zero, N2 = has_N ? at(mesh.N, ix[2]) : zero;
    for (int j = 0; j < ncurves_on_face[i]; j++) {
      std::srand(12.34 * i + 1232 * j);
      const double u = unit_rand();
      std::srand(21.43 * i + 213 * j);
      const double v = (1 - u) * unit_rand();
      const double t = 1 - u - v;
      const V src_P = t * P0 + u * P1 + v * P2;
      V src_N = normalize(t * N0 + u * N1 + v * N2);
      std::srand(i + j);
      const double gravity = .5 + .5 * unit_rand();
      src_N.y -= gravity;
      src_N = normalize(src_N);
      sourceP[curve_id] = src_P;
      sourceN[curve_id] = src_N;
      curve_id++;
    }
  }

  curve.P.assign((size_t) total_ncps * 3, 0.);
  curve.width.assign(total_ncps, 0.);
  curve.Cd.assign((size_t) total_ncps * 3, 0.f);
  curve.indices.assign(total_ncurves, 0);
  curve.uv.clear();
  curve.velocity.clear();
  int cp_id = 0;
  for (int i = 0; i < total_ncurves; i++) {
    for (int vtx = 0; vtx < 4; vtx++) {
      V noisevec{0, 0, 0};
      std::srand(12 * i + 49 * vtx);
      if (vtx > 0) {
        noisevec.x = unit_rand();
        noisevec.y = unit_rand();
        noisevec.z = unit_rand();
      }
      const V src_P = sourceP[i], src_N = sourceN[i];
      const double LENGTH = .02;
      const double noiseamp = .75 * LENGTH;
      const V dst_P = src_P + noiseamp * noisevec + vtx * LENGTH / 3. * src_N;
      curve.P[3 * cp_id] = dst_P.x; curve.P[3 * cp_id + 1] = dst_P.y; curve.P[3 * cp_id + 2] = dst_P.z;
      if (vtx == 0) {
        curve.width[cp_id] = .003; curve.width[cp_id + 1] = .002; curve.width[cp_id + 2] = .001; curve.width[cp_id + 3] = .0001;
      }
      const double amp = 1;
      const float dark[3] = {.8f, .5f, .3f}, light[3] = {.9f, .88f, .85f};
      const V src_Q = mulv(src_P, V{3, 3, 3}) + V{0, 1, 0};
      double C_noise = amp * perlin_noise(src_Q, 2, .5, 2);
      C_noise = smooth_step(.55, .75, C_noise);
      const float tt = (float) C_noise;                       // Lerp(Color, Color, float)
      for (int k = 0; k < 3; k++) curve.Cd[3 * cp_id + k] = (1 - tt) * dark[k] + tt * light[k];
      cp_id++;
    }
    curve.indices[i] = 4 * i;
  }
  curve.ComputeBounds();
  return 0;
}

}  // namespace fjhost
