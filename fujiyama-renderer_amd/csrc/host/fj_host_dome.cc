// fj_host_dome.cc -- DomeLight::preprocess (host, once per RenderScene).
//
// The dome light's sample directions / colours are an INPUT of the hot path
// (importance sampling is a one-time host preprocess).  Behaviour of
// the reference's src/fj_dome_light.cc:58-97 -> StratifiedImportanceSampling
// (src/fj_importance_sampling.cc:102-150, make_histgram :222-239, lookup_histgram
// :241-251, index_to_uv :253-260, uv_to_dir :268-277) on the environment map sampled
// at 1/8 resolution, seed 0; sin / cos are the host libm's.
#include "fj_host.h"

#include <cmath>

namespace fjhost {

namespace {

const double kPi = 3.14159265358979323846;

// Texture::Lookup (src/fj_texture.cc:51-78) on the resident .mip tiles
void tex_lookup(const Texture &tex, float u, float v, float out[4])
{
  if (tex.width == 0 || tex.tiles.empty()) { out[0] = 1.f; out[1] = .63f; out[2] = .63f; out[3] = 1.f; return; }
  const int ts = tex.tilesize;
  const int xnt = tex.width / ts, ynt = tex.height / ts;
  const float tu = u - std::floor(u);
  const float tv = v - std::floor(v);
  const float su = tu * xnt;
  const float sv = (1 - tv) * ynt;
  int xt = (int) std::floor(su), yt = (int) std::floor(sv);
  xt = xt < 0 ? 0 : (xt > xnt - 1 ? xnt - 1 : xt);
  yt = yt < 0 ? 0 : (yt > ynt - 1 ? ynt - 1 : yt);
  const int xp = (int) ((su - std::floor(su)) * 64);
  const int yp = (int) ((sv - std::floor(sv)) * 64);
  if (xp < 0 || xp >= ts || yp < 0 || yp >= ts) { out[0] = out[1] = out[2] = out[3] = 0.f; return; }
  const float *p = &tex.tiles[((size_t) (yt * xnt + xt) * ts * ts + (size_t) (yp * ts + xp)) * tex.nchannels];
  switch (tex.nchannels) {
  case 1: out[0] = out[1] = out[2] = p[0]; out[3] = 1.f; break;
  case 3: out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; out[3] = 1.f; break;
  case 4: out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; out[3] = p[3]; break;
  default: out[0] = out[1] = out[2] = out[3] = 0.f; break;
  }
}

inline float luminance4(const float c[4]) { return (float) (.298912 * c[0] + .586611 * c[1] + .114478 * c[2]); }

struct XorShift128 {
  uint32_t s[4] = {123456789u, 362436069u, 521288629u, 88675123u};
  double next01()
  {
    const uint32_t t = s[0] ^ (s[0] << 11);
    s[0] = s[1]; s[1] = s[2]; s[2] = s[3];
    s[3] = (s[3] ^ (s[3] >> 19)) ^ (t ^ (t >> 8));
    return static_cast<double>(s[3]) / 4294967295u;
  }
};

void index_to_uv(int xres, int yres, int index, float *u, float *v)
{
  const int x = index % xres, y = index / xres;
  *u = (float) ((.5 + x) / xres);
  *v = (float) (1. - ((.5 + y) / yres));
}

}  // namespace

int PreprocessDomeLight(Scene *sc, Light *light)
{
  const int NSAMPLES = light->d.sample_count;
  fj_dome_sample init;
  init.uv[0] = init.uv[1] = (float) (1. / NSAMPLES);
  init.color[0] = 1.f; init.color[1] = .63f; init.color[2] = .63f;
  {
    double d[3] = {1. / NSAMPLES, 1, 1. / NSAMPLES};
    const double len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const double inv = 1. / len;
    for (int k = 0; k < 3; k++) init.dir[k] = d[k] * inv;
  }
  light->dome_samples.assign(NSAMPLES, init);
  const int ti = light->d.environment_map;
  if (ti < 0) return 0;
  const Texture &tex = *sc->textures[ti];
  const int XRES = tex.width / 8, YRES = tex.height / 8;
  const int NPIXELS = XRES * YRES;
  if (NPIXELS <= 0) return 0;

  std::vector<double> hist(NPIXELS);
  double sum = 0;
  for (int i = 0; i < NPIXELS; i++) {
    float u, v, c[4];
    index_to_uv(XRES, YRES, i, &u, &v);
    tex_lookup(tex, u, v, c);
    sum += luminance4(c);
    hist[i] = sum;
  }
  sum = hist[NPIXELS - 1];
  XorShift128 rng;       // seed 0: no warm-up draws
  for (int i = 0; i < NSAMPLES; i++) {
    const double key = sum * ((i + rng.next01()) / NSAMPLES);
    int index = -1;
    for (int k = 0; k < NPIXELS; k++) if (key < hist[k]) { index = k; break; }
    if (index < 0) { g_last_error = "DomeLight: environment map has no luminance"; return -1; }
    fj_dome_sample s;
    index_to_uv(XRES, YRES, index, &s.uv[0], &s.uv[1]);
    const double phi = 2 * kPi * s.uv[0];
    const double theta = kPi * (s.uv[1] - .5);
    const double r = std::cos(theta);
    s.dir[0] = r * std::sin(phi);
    s.dir[1] = std::sin(theta);
    s.dir[2] = r * std::cos(phi);
    float c[4];
    tex_lookup(tex, s.uv[0], s.uv[1], c);
    s.color[0] = c[0]; s.color[1] = c[1]; s.color[2] = c[2];
    light->dome_samples[i] = s;
  }
  return 0;
}

}  // namespace fjhost
