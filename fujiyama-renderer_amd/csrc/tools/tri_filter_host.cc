// tri_filter_host.cc -- validation tool: the f32 triangle filter of the walks (device/fjgpu_tri_filter.h, the SAME source the kernels
// compile) built for the host, so that tests/test_tri_filter.py can hold its decisions against the exact test on hundreds of millions of
// (ray, triangle) pairs without a GPU.  Built with the ROCm clang++ (ext_vector_type, __builtin_elementwise_fma -> fmaf: one rounding,
// like v_pk_fma_f32).  Not part of the product path.
#include <stdint.h>
#include "fjgpu_tri_filter.h"

extern "C" {

// tris [n][9] f32; rays [n][8] f64: o xyz, d xyz, tmin, tmax; bound_abs [n] (>= |any vertex coordinate| of the "primitive set");
// out [n]: FJ_TRI_MISS / FJ_TRI_HIT / FJ_TRI_MAYBE (want_hit = 0: hits stay undecided)
void fj_tri_filter_batch(int64_t n, const float *tris, const double *rays, const float *bound_abs, int want_hit, int8_t *out)
{
  for (int64_t i = 0; i < n; i++) {
    const double *r = rays + 8 * i;
    const TriFilterRay fr = tri_filter_ray(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], bound_abs[i]);
    out[i] = (int8_t) (want_hit ? tri_filter32<true>(tris + 9 * i, fr) : tri_filter32<false>(tris + 9 * i, fr));
  }
}

}
