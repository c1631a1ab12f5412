// fjgpu_raysort.hip -- order a ray queue for the closest-hit walk (SURVEY 7 K5 "ray sorting").
//
// The rays of recursion level >= 1 (mirror, refraction, diffuse bounces) arrive in the order the
// shading kernel emitted them: neighbours in the queue start at neighbouring surface points but
// leave in unrelated directions, so the 64 rays a wave walks together enter different instances
// and different subtrees.  This stage gives every ray a key -- direction octant (3 bits) above the
// Morton code of its origin's cell in a 2^b grid over the traced group's box (3 b bits) -- and
// sorts (key, slot) pairs over exactly those 3 + 3 b bits.  The records themselves do not move:
// the walk reads launch entry k through DScene.ray_perm[k] and writes the hit to the ray's own
// slot, so shading (ray k, hit k) sees nothing of it.
//
// The sort is hand-written for wave64 (round 5; until then hipcub::DeviceRadixSort): a stable LSD
// radix sort with digits of at most 8 bits, ceil((3 + 3 b) / 8) passes (two for the default 15-bit
// keys), four launches per pass:
//   k_rs_hist     one block per TILE of 4096 pairs: digit histogram in LDS (ds_add), written as row
//                 `tile` of the table T[tile][digit] (1 KB, coalesced)
//   k_rs_cols     exclusive scan of every column of T along the tiles, in two levels: a block scans a
//                 SEGMENT of 128 tiles for 64 digits at a time (a wave reads 64 consecutive digits of one
//                 tile: coalesced), 16 tiles per step through LDS with a carry, and leaves the segment's
//                 totals in S[segment][digit]; k_rs_segs then scans S along the segments (one thread per
//                 digit) and writes the digit totals G[digit]
//   k_rs_scatter  one block per tile: every wave ranks its quarter of the tile 64 pairs at a time --
//                 the lanes that hold the same digit find each other with eight ballots (match-any),
//                 rank = prefix popcount within the peers, the group's lowest lane moves the wave's
//                 running digit offset in LDS by the group's size and hands the old value round with a
//                 cross-lane read --, the tile is SORTED IN LDS (32 KB), and then written out in sorted
//                 order, so that the lanes of a wave write runs of consecutive addresses (16 pairs per
//                 digit and tile on average) instead of 4-byte scatters.  Stable by construction:
//                 lane order within a round, rounds in order, waves by position, tiles by T + S.
// The first pass takes the values implicitly (value = index).  Traffic per pass: keys read twice, pairs
// read and written once = 20 bytes per pair (+ 1 KB of table per tile).
//
// The reference has no counterpart (its rays are traced one at a time where they are spawned,
// src/fj_shading.cc:248-300); the order in which rays are traced changes no result.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "fjgpu_raysort.h"

namespace {

constexpr int RS_BLOCK = 256;
constexpr int RS_ITEMS = 16;                       // pairs per thread
constexpr int RS_TILE = RS_BLOCK * RS_ITEMS;       // 4096 pairs per block
constexpr int RS_WAVES = RS_BLOCK / 64;
constexpr int RS_WAVE_ITEMS = RS_TILE / RS_WAVES;  // a wave's quarter of the tile: 1024 pairs, 16 rounds of 64
constexpr int RS_BINS = 256;

__device__ __forceinline__ uint32_t spread3(uint32_t v)      // 10 bits -> every third bit
{
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

struct SortGrid { double lo[3], scale[3]; int bits; };

__global__ void __launch_bounds__(RS_BLOCK) k_ray_sort_keys(const DRay *rays, uint32_t n, SortGrid g, uint32_t *keys)
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
}

__global__ void __launch_bounds__(RS_BLOCK) k_iota(uint32_t *out, uint32_t n)
{
  const uint32_t i = blockIdx.x * RS_BLOCK + threadIdx.x;
  if (i < n) out[i] = i;
}

// ---- pass, launch 1: T[tile][digit] = pairs of the tile with that digit
__global__ void __launch_bounds__(RS_BLOCK) k_rs_hist(const uint32_t *keys, uint32_t n, uint32_t shift, uint32_t mask, uint32_t *T)
{
  __shared__ uint32_t h[RS_BINS];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * (uint32_t) RS_TILE;
  for (int r = 0; r < RS_ITEMS; r++) {
    const uint32_t e = base + (uint32_t) r * RS_BLOCK + threadIdx.x;
    if (e < n) atomicAdd(&h[(keys[e] >> shift) & mask], 1u);
  }
  __syncthreads();
  T[(size_t) blockIdx.x * RS_BINS + threadIdx.x] = h[threadIdx.x];
}

// exclusive scan over the block's values (one per thread): wave prefix by cross-lane shifts, wave totals through LDS
template <int kThreads>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t *s_wave /* [kThreads / 64] */, uint32_t *total)
{
  const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  uint32_t incl = v;
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t up = __shfl_up(incl, d);
    if ((int) lane >= d) incl += up;
  }
  if (lane == 63u) s_wave[w] = incl;
  __syncthreads();
  uint32_t before = 0, all = 0;
  for (int k = 0; k < kThreads / 64; k++) { const uint32_t t = s_wave[k]; if (k < (int) w) before += t; all += t; }
  __syncthreads();              // (s_wave may be reused by the caller's next scan)
  *total = all;
  return before + incl - v;
}

// ---- pass, launch 2a: columns of T scanned along the tiles WITHIN segments of RS_SEG tiles; S[segment][digit] = the segment's total.
// grid (segments, 4 digit groups), block = 64 digits x 16 tiles
constexpr int RS_SEG = 128, RS_COLS_ROWS = 16;
__global__ void __launch_bounds__(64 * RS_COLS_ROWS) k_rs_cols(uint32_t *T, uint32_t n_tiles, uint32_t *S)
{
  __shared__ uint32_t s_v[RS_COLS_ROWS][64];
  const unsigned dx = threadIdx.x & 63u, ty = threadIdx.x >> 6;
  const uint32_t d = blockIdx.y * 64u + dx;
  const uint32_t t0 = blockIdx.x * (uint32_t) RS_SEG, t1 = t0 + RS_SEG < n_tiles ? t0 + RS_SEG : n_tiles;
  uint32_t carry = 0;
  for (uint32_t t = t0; t < t1; t += RS_COLS_ROWS) {
    const uint32_t tile = t + ty;
    const uint32_t v = tile < t1 ? T[(size_t) tile * RS_BINS + d] : 0u;
    s_v[ty][dx] = v;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (int k = 0; k < RS_COLS_ROWS; k++) { const uint32_t x = s_v[k][dx]; if (k < (int) ty) before += x; all += x; }
    __syncthreads();
    if (tile < t1) T[(size_t) tile * RS_BINS + d] = carry + before;
    carry += all;
  }
  if (ty == 0) S[(size_t) blockIdx.x * RS_BINS + d] = carry;
}

// ---- pass, launch 2b: S scanned along the segments (thread d: column d), the digit totals into G
__global__ void __launch_bounds__(RS_BINS) k_rs_segs(uint32_t *S, uint32_t n_segs, uint32_t *G)
{
  uint32_t carry = 0;
  for (uint32_t g = 0; g < n_segs; g++) {
    const uint32_t v = S[(size_t) g * RS_BINS + threadIdx.x];
    S[(size_t) g * RS_BINS + threadIdx.x] = carry;
    carry += v;
  }
  G[threadIdx.x] = carry;
}

// ---- pass, launch 3: rank the tile's pairs (stable), sort the tile in LDS, write it out in sorted order
__global__ void __launch_bounds__(RS_BLOCK, 4) k_rs_scatter(const uint32_t *keys_in, const uint32_t *vals_in, uint32_t n, uint32_t shift, uint32_t mask,
    const uint32_t *T, const uint32_t *S, const uint32_t *G, uint32_t *keys_out, uint32_t *vals_out)
{
  __shared__ uint32_t s_key[RS_TILE], s_val[RS_TILE];
  __shared__ uint32_t s_cnt[RS_WAVES][RS_BINS];       // per wave: pairs with digit d, then the wave's running position for d in the sorted tile
  __shared__ uint32_t s_start[RS_BINS];               // position of the tile's first pair with digit d in the sorted tile
  __shared__ uint32_t s_gbase[RS_BINS];               // ... and in the output array
  __shared__ uint32_t s_wave[RS_WAVES];
  const unsigned tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
  const uint32_t tile = blockIdx.x, base = tile * (uint32_t) RS_TILE;
  const uint32_t cnt_tile = n - base < (uint32_t) RS_TILE ? n - base : (uint32_t) RS_TILE;
  for (int k = 0; k < RS_WAVES; k++) s_cnt[k][tid] = 0;
  __syncthreads();
  // wave w owns pairs [w * 1024, (w + 1) * 1024) of the tile, round r = its pairs r * 64 .. r * 64 + 63 (coalesced: 256 B per round)
  const uint32_t e0 = w * (uint32_t) RS_WAVE_ITEMS + lane;
#pragma unroll 4
  for (int r = 0; r < RS_ITEMS; r++) {
    const uint32_t e = e0 + (uint32_t) r * 64u;
    if (e < cnt_tile) atomicAdd(&s_cnt[w][(keys_in[base + e] >> shift) & mask], 1u);
  }
  __syncthreads();
  {
    // thread d: where digit d starts in the sorted tile (exclusive scan of the tile's digit totals), where each wave's share of it
    // starts, and where the tile's share starts in the output (digit base from the row totals G, + the scanned table)
    const uint32_t c0 = s_cnt[0][tid], c1 = s_cnt[1][tid], c2 = s_cnt[2][tid], c3 = s_cnt[3][tid];
    uint32_t total;
    const uint32_t start = block_exclusive_scan<RS_BLOCK>(c0 + c1 + c2 + c3, s_wave, &total);
    const uint32_t gx = block_exclusive_scan<RS_BLOCK>(G[tid], s_wave, &total);
    s_start[tid] = start;
    s_gbase[tid] = gx + S[(size_t) (tile / (uint32_t) RS_SEG) * RS_BINS + tid] + T[(size_t) tile * RS_BINS + tid];
    s_cnt[0][tid] = start; s_cnt[1][tid] = start + c0; s_cnt[2][tid] = start + c0 + c1; s_cnt[3][tid] = start + c0 + c1 + c2;
  }
  __syncthreads();
  // the ranking sweep re-reads the tile (its 32 KB are in the L2 / L1 from the sweep above), one round ahead of the round it ranks:
  // a register array of the wave's 16 rounds made the scheduler hoist the ballots of all of them (428 VGPRs, one wave per SIMD)
  uint32_t kc = 0, vc = 0;
  if (e0 < cnt_tile) { kc = keys_in[base + e0]; vc = vals_in ? vals_in[base + e0] : base + e0; }
#ifndef RS_RANK_UNROLL
#define RS_RANK_UNROLL 4        // (1 / 2 / 4: 20 M pairs of 15-bit keys in 0.337 / 0.328 / 0.322 ms, 160 M in 3.01 / 2.64 / 2.69; hipcub 0.355 and 2.36)
#endif
#pragma unroll RS_RANK_UNROLL
  for (int r = 0; r < RS_ITEMS; r++) {
    const uint32_t e = e0 + (uint32_t) r * 64u, en = e + 64u;
    uint32_t kn = 0, vn = 0;
    if (r + 1 < RS_ITEMS && en < cnt_tile) { kn = keys_in[base + en]; vn = vals_in ? vals_in[base + en] : base + en; }      // (first pass: the value IS the index)
    const bool valid = e < cnt_tile;
    const uint32_t digit = (kc >> shift) & mask;
    // match-any: the lanes of this round that hold the same digit (eight ballots)
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
      const bool bit = (digit >> b) & 1u;
      const unsigned long long m = __ballot(valid && bit);
      peers &= bit ? m : ~m;
    }
    if (valid) {
      const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) peers, 0u));   // peers below this lane
      const int leader = __ffsll((long long) peers) - 1;
      uint32_t off = 0;
      if ((int) lane == leader) off = atomicAdd(&s_cnt[w][digit], (uint32_t) __popcll(peers));
      off = __shfl(off, leader);
      s_key[off + rank] = kc;
      s_val[off + rank] = vc;
    }
    kc = kn; vc = vn;
  }
  __syncthreads();
  for (uint32_t i = tid; i < cnt_tile; i += RS_BLOCK) {
    const uint32_t key = s_key[i];
    const uint32_t d = (key >> shift) & mask;
    const uint32_t g = s_gbase[d] + (i - s_start[d]);
    if (keys_out) keys_out[g] = key;          // (the last pass of the product's sort: only the permutation is read)
    vals_out[g] = s_val[i];
  }
}

struct RsPlan { int passes, bits_per_pass; uint32_t n_tiles, n_segs; size_t off_vals, off_T, off_S, off_G, bytes; };

RsPlan rs_plan(uint32_t n, int key_bits)
{
  RsPlan p;
  if (key_bits < 1) key_bits = 1;
  if (key_bits > 32) key_bits = 32;
  p.passes = (key_bits + 7) / 8;
  p.bits_per_pass = (key_bits + p.passes - 1) / p.passes;
  p.n_tiles = (n + RS_TILE - 1) / RS_TILE;
  auto up = [](size_t b) { return (b + 255) & ~(size_t) 255; };
  const size_t words = up((size_t) n * 4);
  p.off_vals = words;                                  // [0, words): the other key buffer; then the other value buffer
  p.off_T = 2 * words;
  p.n_segs = (p.n_tiles + RS_SEG - 1) / RS_SEG;
  p.off_S = p.off_T + up((size_t) RS_BINS * std::max<uint32_t>(1u, p.n_tiles) * 4);
  p.off_G = p.off_S + up((size_t) RS_BINS * std::max<uint32_t>(1u, p.n_segs) * 4);
  p.bytes = p.off_G + up(RS_BINS * 4);
  return p;
}

// (keys_in, implicit index) -> (keys_out, perm) over key_bits bits; temp: rs_plan(n, key_bits).bytes
int rs_sort(hipStream_t st, const uint32_t *keys_in, uint32_t n, int key_bits, uint32_t *keys_out, uint32_t *perm, void *temp, size_t temp_bytes, bool want_keys)
{
  if (n == 0) return 0;
  const RsPlan p = rs_plan(n, key_bits);
  if (temp == nullptr || temp_bytes < p.bytes) return -1;
  char *t = (char *) temp;
  uint32_t *kA = (uint32_t *) t, *vA = (uint32_t *) (t + p.off_vals), *T = (uint32_t *) (t + p.off_T), *S = (uint32_t *) (t + p.off_S), *G = (uint32_t *) (t + p.off_G);
  const uint32_t *kin = keys_in, *vin = nullptr;
  for (int q = 0; q < p.passes; q++) {
    const uint32_t shift = (uint32_t) (q * p.bits_per_pass);
    const int width = std::min(p.bits_per_pass, std::max(1, key_bits - q * p.bits_per_pass));
    const uint32_t mask = (1u << width) - 1u;
    // the last pass writes (keys_out, perm); the passes before it alternate so that it does
    const bool to_out = ((p.passes - 1 - q) & 1) == 0;
    uint32_t *kout = to_out ? keys_out : kA, *vout = to_out ? perm : vA;
    if (q == p.passes - 1 && !want_keys) kout = nullptr;
    hipLaunchKernelGGL(k_rs_hist, dim3(p.n_tiles), dim3(RS_BLOCK), 0, st, kin, n, shift, mask, T);
    hipLaunchKernelGGL(k_rs_cols, dim3(p.n_segs, RS_BINS / 64), dim3(64 * RS_COLS_ROWS), 0, st, T, p.n_tiles, S);
    hipLaunchKernelGGL(k_rs_segs, dim3(1), dim3(RS_BINS), 0, st, S, p.n_segs, G);
    hipLaunchKernelGGL(k_rs_scatter, dim3(p.n_tiles), dim3(RS_BLOCK), 0, st, kin, vin, n, shift, mask, T, S, G, kout, vout);
    kin = kout; vin = vout;          // (kout is null only after the last pass)
  }
  return hipGetLastError() == hipSuccess ? 0 : -1;
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

int launch_ray_sort_keyed(hipStream_t st, const uint32_t *keys, uint32_t n, int bits, uint32_t *keys_alt, uint32_t *perm, void *temp, size_t temp_bytes)
{
  if (bits < 1) bits = 1;
  if (bits > 9) bits = 9;
  return rs_sort(st, keys, n, 3 + 3 * bits, keys_alt, perm, temp, temp_bytes, false);
}

size_t ray_sort_temp_bytes(uint32_t n, int bits)
{
  if (bits < 1) bits = 1;
  if (bits > 9) bits = 9;
  return rs_plan(n, 3 + 3 * bits).bytes;
}

int ray_sort_pairs(hipStream_t st, const uint32_t *keys, uint32_t n, int key_bits, uint32_t *keys_out, uint32_t *perm, void *temp, size_t temp_bytes, bool want_keys)
{
  return rs_sort(st, keys, n, key_bits, keys_out, perm, temp, temp_bytes, want_keys);
}

size_t ray_sort_pairs_temp_bytes(uint32_t n, int key_bits) { return rs_plan(n, key_bits).bytes; }

int launch_ray_sort(hipStream_t st, const DRay *rays, uint32_t n, const double box[6], int bits,
    uint32_t *keys, uint32_t *keys_alt, uint32_t *perm, void *temp, size_t temp_bytes)
{
  if (n == 0) return 0;
  if (bits < 1) bits = 1;
  if (bits > 9) bits = 9;
  SortGrid g;
  g.bits = bits;
  ray_sort_grid(box, bits, g.lo, g.scale);
  hipLaunchKernelGGL(k_ray_sort_keys, dim3((n + RS_BLOCK - 1) / RS_BLOCK), dim3(RS_BLOCK), 0, st, rays, n, g, keys);
  if (hipGetLastError() != hipSuccess) return -1;
  return rs_sort(st, keys, n, 3 + 3 * bits, keys_alt, perm, temp, temp_bytes, false);
}
