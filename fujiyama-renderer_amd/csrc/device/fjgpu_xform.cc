// fjgpu_xform.cc -- host math that feeds geometry to the device.
//
// Every transcendental that influences a hit/miss decision is evaluated HERE,
// on the host, with the reference's own operation order (libm sin/cos/tan,
// no FP contraction), so the matrices and tables the kernels consume are
// bit-identical to what the reference computes per ray:
//   make_transform_matrix   reference src/fj_transform.cc:335-391
//   MatRotateX/Y/Z, MatMultiply, MatInverse (Cramer)   src/fj_matrix.cc:66-206
//   PropLerpSamples         src/fj_property.cc:317-345
//   Tiler::GenerateTiles    src/fj_tiler.cc:56-113
//   XorShift                src/fj_random.cc:10-43
// Compiled with -ffp-contract=off.
#include "fjgpu_build.h"
#include "fjgpu_xform_math.h"

#include <cmath>
#include <cstring>

namespace fjgpu {

void MakeTransform(const fj_xform_desc &x, double time, double M[16], double Minv[16])
{
  fjx::make_transform(x, time, M, Minv);
}

// MatTransformBounds, src/fj_matrix.cc:224-249
void TransformBounds(const double M[16], const double in[6], double out[6])
{
  const double big = 1.7976931348623157e308;
  double mn[3] = {big, big, big}, mx[3] = {-big, -big, -big};
  for (int c = 0; c < 8; c++) {
    const double p[3] = {(c & 1) ? in[3] : in[0], (c & 2) ? in[4] : in[1], (c & 4) ? in[5] : in[2]};
    for (int k = 0; k < 3; k++) {
      const double q = M[4 * k] * p[0] + M[4 * k + 1] * p[1] + M[4 * k + 2] * p[2] + M[4 * k + 3];
      if (q < mn[k]) mn[k] = q;
      if (q > mx[k]) mx[k] = q;
    }
  }
  for (int k = 0; k < 3; k++) { out[k] = mn[k]; out[3 + k] = mx[k]; }
}

void GenerateTiles(const fj_render_desc &r, std::vector<TileRect> *tiles)
{
  auto imax = [](int a, int b) { return a > b ? a : b; };
  auto imin = [](int a, int b) { return a < b ? a : b; };
  const int xmin = r.region[0], ymin = r.region[1], xmax = r.region[2], ymax = r.region[3];
  const int X0 = (int) std::floor(imax(0, xmin) / (double) r.tile_w);
  const int Y0 = (int) std::floor(imax(0, ymin) / (double) r.tile_h);
  const int X1 = (int) std::ceil(imin(r.xres, xmax) / (double) r.tile_w);
  const int Y1 = (int) std::ceil(imin(r.yres, ymax) / (double) r.tile_h);
  tiles->clear();
  for (int y = Y0; y < Y1; y++)
    for (int x = X0; x < X1; x++) {
      TileRect t;
      t.id = (int) tiles->size();
      t.xmin = imax(x * r.tile_w, xmin);
      t.ymin = imax(y * r.tile_h, ymin);
      t.xmax = imin((x + 1) * r.tile_w, xmax);
      t.ymax = imin((y + 1) * r.tile_h, ymax);
      tiles->push_back(t);
    }
}

// count_samples_in_margin, src/fj_fixed_grid_sampler.cc:131-136
void SamplerMargin(const fj_render_desc &r, int margin[2])
{
  margin[0] = (int) std::ceil((((double) r.filter_w - 1) * r.rate_x) * .5);
  margin[1] = (int) std::ceil((((double) r.filter_h - 1) * r.rate_y) * .5);
}

void XorShiftTable(size_t n, std::vector<double> *out)
{
  uint32_t s0 = 123456789u, s1 = 362436069u, s2 = 521288629u, s3 = 88675123u;
  out->resize(n);
  for (size_t i = 0; i < n; i++) {
    const uint32_t t = s0 ^ (s0 << 11);
    s0 = s1; s1 = s2; s2 = s3;
    s3 = (s3 ^ (s3 >> 19)) ^ (t ^ (t >> 8));
    (*out)[i] = static_cast<double>(s3) / 4294967295u;   // / UINT32_MAX
  }
}

double CameraUvSizeY(double fov)   // src/fj_camera.cc:98-102
{
  return 2 * std::tan((fov / 2.) * FJX_PI / 180.);
}

}  // namespace fjgpu

// ---- diagnostics (include/fjgpu.h): let the CPU test-suite pin the host math
// against the reference's golden vectors without a GPU
extern "C" {

void fjgpu_host_make_transform(int transform_order, int rotate_order, const double *trs, double *M, double *Minv)
{
  fj_xform_desc x;
  std::memset(&x, 0, sizeof(x));
  x.transform_order = transform_order;
  x.rotate_order = rotate_order;
  x.n_translate = x.n_rotate = x.n_scale = 1;
  for (int k = 0; k < 3; k++) { x.translate[0].v[k] = trs[k]; x.rotate[0].v[k] = trs[3 + k]; x.scale[0].v[k] = trs[6 + k]; }
  fjgpu::MakeTransform(x, 0, M, Minv);
}

void fjgpu_host_xorshift_f01(int n, double *out)
{
  std::vector<double> t;
  fjgpu::XorShiftTable((size_t) (n > 0 ? n : 0), &t);
  for (int i = 0; i < n; i++) out[i] = t[i];
}

void fjgpu_host_sampler_margin(const fj_render_desc *r, int32_t *margin)
{
  int m[2];
  fjgpu::SamplerMargin(*r, m);
  margin[0] = m[0]; margin[1] = m[1];
}

double fjgpu_host_camera_uv_size_y(double fov) { return fjgpu::CameraUvSizeY(fov); }

}  // extern "C"
