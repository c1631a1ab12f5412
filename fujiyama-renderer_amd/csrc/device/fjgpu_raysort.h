// fjgpu_raysort.h -- ray-queue sort in front of the closest-hit walk (fjgpu_raysort.hip)
#ifndef FJGPU_RAYSORT_H
#define FJGPU_RAYSORT_H

#include <hip/hip_runtime.h>

#include "fjgpu_types.h"

#ifndef FJ_RAY_SORT_BITS
#define FJ_RAY_SORT_BITS 4           // default grid of the sort: 2^4 cells per axis (15-bit keys, two radix passes: C4 sort 41.9 -> 28.7 ms per frame, the walk 668 -> 678, frame 1002 -> 1000; 3 / 5 / 7 bits: 1006 / 1008 / 1002)
#endif
#ifndef FJ_RAY_SORT_MIN
#define FJ_RAY_SORT_MIN (1u << 16)   // smaller launches are walked in queue order (the sort's launches would cost more)
#endif

// bytes of scratch the radix sort of n (key, slot) pairs over 3 + 3 * bits key bits needs (the other pair buffer of the passes' ping-pong,
// the tile table and the digit totals)
size_t ray_sort_temp_bytes(uint32_t n, int bits);

// perm[k] = slot of the k-th ray in (direction octant, Morton cell of the origin in a 2^bits grid over
// `box`) order.  keys / keys_alt / perm: n words each; temp: ray_sort_temp_bytes(n, bits).
int launch_ray_sort(hipStream_t st, const DRay *rays, uint32_t n, const double box[6], int bits,
    uint32_t *keys, uint32_t *keys_alt, uint32_t *perm, void *temp, size_t temp_bytes);

// the same with the keys already in place (written by the shading kernel where the rays were emitted, ShadeParams.next_keys)
int launch_ray_sort_keyed(hipStream_t st, const uint32_t *keys, uint32_t n, int bits, uint32_t *keys_alt, uint32_t *perm,
    void *temp, size_t temp_bytes);
// the sort itself: (keys[i], i) pairs, stable, over the low key_bits bits (1 .. 32) -> perm, and keys_out ascending if want_keys (keys_out is a
// buffer of n words either way: the passes alternate between it and `temp`) (fjgpu_dev_sort_pairs: tests, tools/raysort_bench)
int ray_sort_pairs(hipStream_t st, const uint32_t *keys, uint32_t n, int key_bits, uint32_t *keys_out, uint32_t *perm, void *temp, size_t temp_bytes, bool want_keys);
size_t ray_sort_pairs_temp_bytes(uint32_t n, int key_bits);
int ray_sort_fill_iota(hipStream_t st, uint32_t *iota, uint32_t n);
void ray_sort_grid(const double box[6], int bits, double lo[3], double scale[3]);

#endif
