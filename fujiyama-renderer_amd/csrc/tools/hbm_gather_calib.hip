// hbm_gather_calib.hip -- calibration of rocprofv3's FETCH_SIZE for the access pattern of the BVH walks.
//
// MI355X_MICROARCH.md ("HBM"): on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced
// streaming read (TCC_EA0_RDREQ x 64 B with 128-byte requests tallied at 64 B) and "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern".  The
// traversal kernels (fjgpu_dev_anyhit.h, fjgpu_dev_traverse.h) gather 64-byte node records -- four
// 16-byte loads per lane, every lane another record -- so this tool runs exactly that pattern on a
// KNOWN byte count, over an array far larger than L2 + Infinity Cache (default 4 GiB against 32 MiB
// + 256 MiB), next to the streaming read the guide calibrated and a 128-byte-record gather:
//
//   k_calib_stream     every lane 16 B, consecutive lanes consecutive addresses, the whole array once
//   k_calib_gather64   every lane `reps` random 64-byte records   (4 x global_load_dwordx4, 64-B aligned)
//   k_calib_gather128  every lane `reps` random 128-byte records  (8 x global_load_dwordx4, 128-B aligned)
//   k_calib_gather36   every lane `reps` random 36-byte triangle records (9 x global_load_dword)
//
// It prints one JSON line with the bytes each kernel must read and its HIP-event time; run it under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE                       (pass 1)
//   rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum   (pass 2)
// and scripts/fetch_calibration.py divides: factor = bytes needed / (FETCH_SIZE x 1024).  bench.py multiplies
// the walk's FETCH_SIZE by the factor measured for k_calib_gather64 instead of the guide's streaming constant.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__global__ void __launch_bounds__(256) k_calib_fill(v4u *a, size_t n16)
{
  for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t) gridDim.x * 256) {
    v4u v; v.x = (uint32_t) i; v.y = mix32((uint32_t) i); v.z = ~(uint32_t) i; v.w = 1u;
    a[i] = v;
  }
}

__global__ void __launch_bounds__(256) k_calib_stream(const v4u *a, size_t n16, uint32_t *out)
{
  uint32_t acc = 0;
  for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t) gridDim.x * 256) {
    const v4u v = a[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  out[(size_t) blockIdx.x * 256 + threadIdx.x] = acc;
}

// kLoads x 16 B per record, records aligned to their size
template <int kLoads>
__device__ __forceinline__ uint32_t gather(const v4u *a, uint32_t n_rec_mask, uint32_t reps, uint32_t seed)
{
  const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0, h = mix32(tid * 0x9E3779B1u + seed);
  for (uint32_t r = 0; r < reps; r++) {
    h = mix32(h + 0x632BE5ABu);
    const v4u *p = a + (size_t) (h & n_rec_mask) * kLoads;
#pragma unroll
    for (int k = 0; k < kLoads; k++) { const v4u v = p[k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  }
  return acc;
}
__global__ void __launch_bounds__(256) k_calib_gather64(const v4u *a, uint32_t mask, uint32_t reps, uint32_t *out)
{
  out[(size_t) blockIdx.x * 256 + threadIdx.x] = gather<4>(a, mask, reps, 1u);
}
__global__ void __launch_bounds__(256) k_calib_gather128(const v4u *a, uint32_t mask, uint32_t reps, uint32_t *out)
{
  out[(size_t) blockIdx.x * 256 + threadIdx.x] = gather<8>(a, mask, reps, 2u);
}
// 36-byte records (the f32 triangles of the walks: 9 dword loads, 4-byte aligned, packed back to back)
__global__ void __launch_bounds__(256) k_calib_gather36(const uint32_t *a, uint32_t n_rec, uint32_t reps, uint32_t *out)
{
  const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0, h = mix32(tid * 0x9E3779B1u + 3u);
  for (uint32_t r = 0; r < reps; r++) {
    h = mix32(h + 0x632BE5ABu);
    const uint32_t *p = a + (size_t) (h % n_rec) * 9;
#pragma unroll
    for (int k = 0; k < 9; k++) acc ^= p[k];
  }
  out[tid] = acc;
}

int main(int argc, char **argv)
{
  const double gib = argc > 1 ? atof(argv[1]) : 4.0;          // array size
  const uint32_t reps = argc > 2 ? (uint32_t) atoi(argv[2]) : 64u;
  int log2_bytes = 12;      // (arrays from 4 KiB: a footprint inside one CU's 32 KB L1 measures the L1's own request rate)
  while (log2_bytes < 37 && (double) (1ull << (log2_bytes + 1)) <= gib * 1073741824.0) log2_bytes++;
  const size_t bytes = (size_t) 1 << log2_bytes;
  const size_t n16 = bytes / 16;
  int dev = 0;
  hipDeviceProp_t prop;
  CK(hipGetDevice(&dev));
  CK(hipGetDeviceProperties(&prop, dev));
  const unsigned grid = (unsigned) prop.multiProcessorCount * 8u;     // 8 blocks of 4 waves per CU
  const size_t threads = (size_t) grid * 256;
  v4u *a = nullptr;
  uint32_t *out = nullptr;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&out, threads * sizeof(uint32_t)));
  hipLaunchKernelGGL(k_calib_fill, dim3(grid), dim3(256), 0, 0, a, n16);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms[4] = {0, 0, 0, 0};
  const uint32_t n36 = (uint32_t) (bytes / 36);
  for (int pass = 0; pass < 2; pass++) {        // pass 0 warms (clocks, TLB); the counters of both are reported per launch
    for (int k = 0; k < 4; k++) {
      CK(hipEventRecord(e0, 0));
      if (k == 0) hipLaunchKernelGGL(k_calib_stream, dim3(grid), dim3(256), 0, 0, a, n16, out);
      if (k == 1) hipLaunchKernelGGL(k_calib_gather64, dim3(grid), dim3(256), 0, 0, a, (uint32_t) (bytes / 64 - 1), reps, out);
      if (k == 2) hipLaunchKernelGGL(k_calib_gather128, dim3(grid), dim3(256), 0, 0, a, (uint32_t) (bytes / 128 - 1), reps, out);
      if (k == 3) hipLaunchKernelGGL(k_calib_gather36, dim3(grid), dim3(256), 0, 0, (const uint32_t *) a, n36, reps, out);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms[k], e0, e1));
    }
  }
  const double need[4] = {(double) bytes, (double) threads * reps * 64.0, (double) threads * reps * 128.0, (double) threads * reps * 36.0};
  const char *names[4] = {"k_calib_stream", "k_calib_gather64", "k_calib_gather128", "k_calib_gather36"};
  printf("{\"device\": \"%s\", \"array_bytes\": %zu, \"threads\": %zu, \"reps\": %u, \"launches_per_kernel\": 2, \"kernels\": {", prop.name, bytes, threads, reps);
  for (int k = 0; k < 4; k++)
    printf("%s\"%s\": {\"bytes_needed_per_launch\": %.0f, \"ms\": %.4f, \"GBps_needed\": %.1f}", k ? ", " : "", names[k], need[k], ms[k],
        need[k] / (ms[k] * 1e-3) / 1e9);
  printf("}}\n");
  (void) hipFree(a); (void) hipFree(out);
  return 0;
}
