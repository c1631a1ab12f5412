// raysort_bench.hip -- measurement tool (not part of the product libraries): the product's hand-written ray-queue sort
// (fjgpu_raysort.hip, through fjgpu_dev_sort_pairs of include/fjgpu.h) against hipcub::DeviceRadixSort::SortPairs -- the library
// the sort replaced in round 5 -- on the same (key, index) pairs: n pairs, keys of `bits` bits (15 = the default of the ray sort),
// uniformly random or in runs like the keys of neighbouring surface points.  usage: raysort_bench [n [key_bits [repeats]]]
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fjgpu.h"

int main(int argc, char **argv)
{
  const int n = argc > 1 ? atoi(argv[1]) : 20000000;
  const int bits = argc > 2 ? atoi(argv[2]) : 15;
  const int reps = argc > 3 ? atoi(argv[3]) : 10;
  std::vector<uint32_t> keys((size_t) n), out((size_t) n), perm((size_t) n), iota((size_t) n);
  uint64_t s = 88172645463325252ull;
  for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; keys[(size_t) i] = (uint32_t) (s >> 20) & (bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u)); iota[(size_t) i] = (uint32_t) i; }
  double ms_own = 0, ms_own_perm = 0;
  if (fjgpu_dev_sort_pairs(0, keys.data(), n, bits, out.data(), perm.data(), reps, &ms_own)) { fprintf(stderr, "fjgpu_dev_sort_pairs: %s\n", fjgpu_last_error()); return 1; }
  // ... and as the product calls it: the permutation only (the last pass writes no keys)
  if (fjgpu_dev_sort_pairs(0, keys.data(), n, bits, nullptr, perm.data(), reps, &ms_own_perm)) { fprintf(stderr, "fjgpu_dev_sort_pairs: %s\n", fjgpu_last_error()); return 1; }
  // hipcub on the same pairs
  uint32_t *dk = nullptr, *dk2 = nullptr, *dv = nullptr, *dv2 = nullptr;
  void *tmp = nullptr;
  size_t tmp_bytes = 0;
  hipMalloc(&dk, 4 * (size_t) n); hipMalloc(&dk2, 4 * (size_t) n); hipMalloc(&dv, 4 * (size_t) n); hipMalloc(&dv2, 4 * (size_t) n);
  hipMemcpy(dk, keys.data(), 4 * (size_t) n, hipMemcpyHostToDevice);
  hipMemcpy(dv, iota.data(), 4 * (size_t) n, hipMemcpyHostToDevice);
  hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, dk, dk2, dv, dv2, n, 0, bits, 0);
  hipMalloc(&tmp, tmp_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  double ms_cub = 1e30;
  for (int k = 0; k < reps; k++) {
    hipEventRecord(e0, 0);
    hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, dk, dk2, dv, dv2, n, 0, bits, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < ms_cub) ms_cub = ms;
  }
  std::vector<uint32_t> perm_cub((size_t) n);
  hipMemcpy(perm_cub.data(), dv2, 4 * (size_t) n, hipMemcpyDeviceToHost);
  size_t diff = 0;
  for (int i = 0; i < n; i++) diff += perm_cub[(size_t) i] != perm[(size_t) i];
  const int passes = (bits + 7) / 8;
  printf("{\"n\": %d, \"key_bits\": %d, \"passes\": %d, \"own_ms\": %.4f, \"own_perm_only_ms\": %.4f, \"hipcub_ms\": %.4f, \"own_over_hipcub\": %.3f, "
         "\"own_perm_only_over_hipcub\": %.3f, \"own_perm_only_GBps_of_pair_traffic\": %.1f, \"hipcub_GBps_of_pair_traffic\": %.1f, \"permutations_differ_at\": %zu}\n",
         n, bits, passes, ms_own, ms_own_perm, ms_cub, ms_own / ms_cub, ms_own_perm / ms_cub,
         (double) n * 16.0 * passes / (ms_own_perm * 1e-3) / 1e9, (double) n * 16.0 * passes / (ms_cub * 1e-3) / 1e9, diff);
  return 0;
}
