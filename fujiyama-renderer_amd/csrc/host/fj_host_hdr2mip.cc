// fj_host_hdr2mip.cc -- texture conversion either side of the path: Radiance .hdr -> tiled .mip
// (the reference's tools/hdr2mip: RGBE decode, then MipOutput::GenerateFromSourceData + WriteFile,
// src/fj_mipmap.cc:246-300,317-390).
//
//   * RGBE: "#?RADIANCE" header with FORMAT=32-bit_rle_rgbe, the resolution line "-Y h +X w", then
//     scanlines either flat (4 bytes per pixel) or new-style run-length encoded per channel
//     (2 2 hi lo, then for each of the four channels runs: count > 128 -> count - 128 copies of
//     the next byte, else `count` literal bytes).  A pixel is mantissa * 2^(e - 136), 0 when e = 0.
//   * .mip: the image is resampled to the next powers of two with the reference's 2 x 2 box
//     weights around the rounded source position, cut into tiles of min(64, w, h) pixels, and
//     written as "MIPM", version 1, width, height, channels, tilesize, tiles row-major (each tile
//     row-major, float32).
#include "fj_host.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace fjhost {

static bool read_line(FILE *fp, std::string *line)
{
  line->clear();
  for (int c; (c = fgetc(fp)) != EOF;) {
    if (c == '\n') return true;
    line->push_back((char) c);
    if (line->size() > 4096) return false;
  }
  return !line->empty();
}

static inline void rgbe_to_float(const unsigned char *p, float *dst)
{
  if (p[3]) {
    const float f = (float) std::ldexp(1.0, (int) p[3] - (128 + 8));
    dst[0] = p[0] * f; dst[1] = p[1] * f; dst[2] = p[2] * f;
  } else dst[0] = dst[1] = dst[2] = 0.f;
}

// pixels: [h][w][3] float32, top scanline first
int ReadRgbeFile(const std::string &path, int *w, int *h, std::vector<float> *pixels, std::string *err)
{
  FILE *fp = fopen(path.c_str(), "rb");
  if (!fp) { *err = "cannot open " + path; return -1; }
  std::string line;
  bool format_ok = false;
  *w = *h = 0;
  // header: "#?..." first, key=value lines, an empty line, then the resolution
  if (!read_line(fp, &line) || line.compare(0, 2, "#?") != 0) { fclose(fp); *err = path + ": not a Radiance picture"; return -1; }
  while (read_line(fp, &line)) {
    if (line.empty()) break;
    if (line == "FORMAT=32-bit_rle_rgbe") format_ok = true;
  }
  if (!format_ok) { fclose(fp); *err = path + ": no FORMAT=32-bit_rle_rgbe line"; return -1; }
  if (!read_line(fp, &line) || sscanf(line.c_str(), "-Y %d +X %d", h, w) != 2 || *w <= 0 || *h <= 0) {
    fclose(fp); *err = path + ": missing -Y h +X w"; return -1;
  }
  const int W = *w, H = *h;
  pixels->assign((size_t) W * H * 3, 0.f);
  std::vector<unsigned char> scan((size_t) W * 4);
  for (int y = 0; y < H; y++) {
    unsigned char head[4];
    if (fread(head, 1, 4, fp) != 4) { fclose(fp); *err = path + ": truncated"; return -1; }
    const bool rle = W >= 8 && W <= 0x7fff && head[0] == 2 && head[1] == 2 && !(head[2] & 0x80);
    if (!rle) {
      // flat: this scanline's first pixel is `head`, the rest follows
      std::memcpy(scan.data(), head, 4);
      if (W > 1 && fread(scan.data() + 4, 1, (size_t) (W - 1) * 4, fp) != (size_t) (W - 1) * 4) { fclose(fp); *err = path + ": truncated"; return -1; }
      for (int x = 0; x < W; x++) rgbe_to_float(&scan[(size_t) x * 4], &(*pixels)[((size_t) y * W + x) * 3]);
      continue;
    }
    if (((int) head[2] << 8 | head[3]) != W) { fclose(fp); *err = path + ": scanline width mismatch"; return -1; }
    std::vector<unsigned char> chan((size_t) W);
    for (int c = 0; c < 4; c++) {
      int x = 0;
      while (x < W) {
        const int n0 = fgetc(fp);
        if (n0 == EOF) { fclose(fp); *err = path + ": truncated"; return -1; }
        if (n0 > 128) {
          const int n = n0 - 128, v = fgetc(fp);
          if (v == EOF || x + n > W) { fclose(fp); *err = path + ": bad run"; return -1; }
          std::memset(&chan[x], v, (size_t) n);
          x += n;
        } else {
          if (n0 == 0 || x + n0 > W || fread(&chan[x], 1, (size_t) n0, fp) != (size_t) n0) { fclose(fp); *err = path + ": bad run"; return -1; }
          x += n0;
        }
      }
      for (int k = 0; k < W; k++) scan[(size_t) k * 4 + c] = chan[k];
    }
    for (int x = 0; x < W; x++) rgbe_to_float(&scan[(size_t) x * 4], &(*pixels)[((size_t) y * W + x) * 3]);
  }
  fclose(fp);
  return 0;
}

static int next_pow2(int v) { int p = 1; while (p < v && p < 32768) p *= 2; return p; }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// MipOutput::GenerateFromSourceData + WriteFile
int WriteMipFile(const std::string &path, const float *src, int sw, int sh, int nch, std::string *err)
{
  if (nch != 1 && nch != 3) { *err = "mip: 1 or 3 channels"; return -1; }
  const int dw = next_pow2(sw), dh = next_pow2(sh);
  std::vector<float> dst((size_t) dw * dh * nch);
  if (dw == sw && dh == sh) std::memcpy(dst.data(), src, sizeof(float) * dst.size());
  else {
    // scale_and_copy_image, src/fj_mipmap.cc:333-390: float positions, weights .5 -+ (pos - round(pos))
    const float xscale = dw / (float) sw, yscale = dh / (float) sh;
    for (int y = 0; y < dh; y++) {
      const float py = (float) ((y + .5) / yscale);
      for (int x = 0; x < dw; x++) {
        const float px = (float) ((x + .5) / xscale);
        const float cx = (float) std::round(px), cy = (float) std::round(py);
        const int xs[2] = {clampi((int) cx - 1, 0, sw - 1), clampi((int) cx, 0, sw - 1)};
        const int ys[2] = {clampi((int) cy - 1, 0, sh - 1), clampi((int) cy, 0, sh - 1)};
        const float xw0 = (float) (.5 - (px - cx)), xw1 = 1 - xw0;
        const float yw0 = (float) (.5 - (py - cy)), yw1 = 1 - yw0;
        const float wgt[4] = {xw0 * yw0, xw1 * yw0, xw0 * yw1, xw1 * yw1};
        const size_t idx[4] = {((size_t) ys[0] * sw + xs[0]) * nch, ((size_t) ys[0] * sw + xs[1]) * nch,
                               ((size_t) ys[1] * sw + xs[0]) * nch, ((size_t) ys[1] * sw + xs[1]) * nch};
        float *d = &dst[((size_t) y * dw + x) * nch];
        for (int c = 0; c < nch; c++) {
          d[c] = 0;
          for (int k = 0; k < 4; k++) d[c] += wgt[k] * src[idx[k] + c];
        }
      }
    }
  }
  int ts = 64 < dw ? 64 : dw;
  ts = ts < dh ? ts : dh;
  FILE *fp = fopen(path.c_str(), "wb");
  if (!fp) { *err = "cannot write " + path; return -1; }
  const int head[5] = {1, dw, dh, nch, ts};
  fwrite("MIPM", 1, 4, fp);
  fwrite(head, sizeof(int), 5, fp);
  std::vector<float> tile((size_t) ts * ts * nch);
  for (int ty = 0; ty < dh / ts; ty++)
    for (int tx = 0; tx < dw / ts; tx++) {
      for (int y = 0; y < ts; y++)
        std::memcpy(&tile[(size_t) y * ts * nch], &dst[(((size_t) ty * ts + y) * dw + (size_t) tx * ts) * nch], sizeof(float) * ts * nch);
      fwrite(tile.data(), sizeof(float), tile.size(), fp);
    }
  fclose(fp);
  return 0;
}

}  // namespace fjhost

extern "C" int fj_hdr2mip(const char *hdr_path, const char *mip_path)
{
  if (!hdr_path || !mip_path) return -1;
  int w = 0, h = 0;
  std::vector<float> px;
  std::string err;
  if (fjhost::ReadRgbeFile(hdr_path, &w, &h, &px, &err) || fjhost::WriteMipFile(mip_path, px.data(), w, h, 3, &err)) {
    fprintf(stderr, "hdr2mip: %s\n", err.c_str());
    return -1;
  }
  return 0;
}
