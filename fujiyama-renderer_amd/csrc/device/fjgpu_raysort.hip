// fjgpu_raysort.hip -- order a ray queue for the closest-hit walk (SURVEY 7 K5 "ray sorting").
//
// The rays of recursion level >= 1 (mirror, refraction, diffuse bounces) arrive in the order the
// shading kernel emitted them: neighbours in the queue start at neighbouring surface points but
// leave in unrelated directions, so the 64 rays a wave walks together enter different instances
// and different subtrees.  This stage gives every ray a key -- direction octant (3 bits) above the
// Morton code of its origin's cell in a 2^b grid over the traced group's box (3 b bits) -- and
// sorts (key, slot) pairs with hipcub's radix sort over exactly those 3 + 3 b bits.  The records
// themselves do not move: the walk reads launch entry k through DScene.ray_perm[k] and writes the
// hit to the ray's own slot, so shading (ray k, hit k) sees nothing of it.
//
// The reference has no counterpart (its rays are traced one at a time where they are spawned,
// src/fj_shading.cc:248-300); the order in which rays are traced changes no result.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "fjgpu_raysort.h"

namespace {

constexpr int RS_BLOCK = 256;

__device__ __forceinline__ uint32_t spread3(uint32_t v)      // 10 bits -> every third bit
{
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

struct SortGrid { double lo[3], scale[3]; int bits; };

__global__ void __launch_bounds__(RS_BLOCK) k_ray_sort_keys(const DRay *rays, uint32_t n, SortGrid g, uint32_t *keys, uint32_t *slots)
{
  const uint32_t i = blockIdx.x * RS_BLOCK + threadIdx.x;
  if (i >= n) return;
  const DRay *r = &rays[i];
  const double cells = (double) (1u << g.bits);
  uint32_t c[3];
  for (int a = 0; a < 3; a++) {
    double x = (r->o[a] - g.lo[a]) * g.scale[a];
    x = x < 0. ? 0. : (x > cells - 1. ? cells - 1. : x);        // (NaN compares false twice: cell 0 after the cast below)
    c[a] = x == x ? (uint32_t) x : 0u;
  }
  const uint32_t oct = (r->d[0] < 0. ? 1u : 0u) | (r->d[1] < 0. ? 2u : 0u) | (r->d[2] < 0. ? 4u : 0u);
  keys[i] = (oct << (3 * g.bits)) | spread3(c[0]) | (spread3(c[1]) << 1) | (spread3(c[2]) << 2);
  slots[i] = i;
}

__global__ void __launch_bounds__(RS_BLOCK) k_iota(uint32_t *out, uint32_t n)
{
  const uint32_t i = blockIdx.x * RS_BLOCK + threadIdx.x;
  if (i < n) out[i] = i;
}

}  // namespace

void ray_sort_grid(const double box[6], int bits, double lo[3], double scale[3])
{
  for (int a = 0; a < 3; a++) {
    const double w = box[3 + a] - box[a];
    lo[a] = box[a];
    scale[a] = w > 0. ? (double) (1u << bits) / w : 0.;
  }
}

int ray_sort_fill_iota(hipStream_t st, uint32_t *iota, uint32_t n)
{
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_iota, dim3((n + RS_BLOCK - 1) / RS_BLOCK), dim3(RS_BLOCK), 0, st, iota, n);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_ray_sort_keyed(hipStream_t st, const uint32_t *keys, uint32_t n, int bits, uint32_t *keys_alt, const uint32_t *iota, uint32_t *perm,
    void *temp, size_t temp_bytes)
{
  if (n == 0) return 0;
  if (bits < 1) bits = 1;
  if (bits > 9) bits = 9;
  if (hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, keys_alt, iota, perm, (int) n, 0, 3 + 3 * bits, st) != hipSuccess) return -1;
  return 0;
}

size_t ray_sort_temp_bytes(uint32_t n, int bits)
{
  size_t bytes = 0;
  uint32_t *nil = nullptr;
  (void) hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, nil, nil, nil, nil, (int) n, 0, 3 + 3 * bits, 0);
  return bytes;
}

int launch_ray_sort(hipStream_t st, const DRay *rays, uint32_t n, const double box[6], int bits,
    uint32_t *keys, uint32_t *keys_alt, uint32_t *slots, uint32_t *perm, void *temp, size_t temp_bytes)
{
  if (n == 0) return 0;
  if (bits < 1) bits = 1;
  if (bits > 9) bits = 9;
  SortGrid g;
  g.bits = bits;
  for (int a = 0; a < 3; a++) {
    const double w = box[3 + a] - box[a];
    g.lo[a] = box[a];
    g.scale[a] = w > 0. ? (double) (1u << bits) / w : 0.;
  }
  hipLaunchKernelGGL(k_ray_sort_keys, dim3((n + RS_BLOCK - 1) / RS_BLOCK), dim3(RS_BLOCK), 0, st, rays, n, g, keys, slots);
  if (hipGetLastError() != hipSuccess) return -1;
  if (hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, keys_alt, slots, perm, (int) n, 0, 3 + 3 * bits, st) != hipSuccess) return -1;
  return 0;
}
