// oracle/restate/fjo_render.h -- TEST INFRASTRUCTURE ONLY (CPU oracle).
#ifndef FJO_RENDER_H
#define FJO_RENDER_H

#include "fjo_scene.h"

#include <vector>

namespace fjo {

struct Tile { int id, xmin, ymin, xmax, ymax; };               // src/fj_tiler.h:13-20
struct Sample { double uv[2]; double data[4]; double time; };  // src/fj_pixel_sample.h:13-22

struct CameraState {
  const fj_camera_desc *d;
  double uv_size[2];
  bool is_static;
  Xfm xfm_static;
};

void GenerateTiles(const fj_render_desc &r, std::vector<Tile> *tiles);
void SamplerMargin(const fj_render_desc &r, int margin[2]);
void GenerateSamples(const fj_render_desc &r, const Tile &tile, std::vector<Sample> *samples, int nsamples[2]);
void CameraInit(const fj_camera_desc *d, int xres, int yres, CameraState *cam);
void CameraGetRay(const CameraState &cam, const double uv[2], double time, Ray *ray);
double GaussianFilter(double xwidth, double ywidth, double x, double y);
Col4 TextureLookup(const fj_texture_desc &tex, float u, float v);

// serial_rng != 0: one worker and the reference's own random streams in its draw order (the
// restatement then reproduces a `thread_count 1` reference render of PathtracingShader / area
// light scenes bit for bit); 0: the counter-based streams shared with the device
int RenderTiles(Scene *sc, const fj_render_desc &r, const int32_t *tile_ids, int n_tiles,
    float *fb, int nthreads, fj_ray_counts *counts, int serial_rng = 0);

}  // namespace fjo
#endif
