// oracle/ref_render.cc -- TEST INFRASTRUCTURE ONLY.
//
// Driver around the UNMODIFIED reference (oracle/_ref/libscene.so + the
// reference's tools/scene_parser Parser class, compiled from /root/reference
// by oracle/Makefile).  It feeds a .scn file line by line to the reference
// parser exactly like the reference's tools/scene_parser/main.cc:9-47 does,
// with one addition: just before a `RenderScene` line it installs frame / tile
// report callbacks through the reference's own public hooks
// (SiSetFrameReportCallback / SiSetTileReportCallback,
// src/fj_scene_interface.cc:1007-1045) so that
//   * the fbview socket code is out of the timed path,
//   * RenderScene is timed with steady_clock (frame start -> frame done),
//   * the finished framebuffer is dumped as raw float32 (the .fb text writer
//     only keeps 6 significant digits, src/fj_framebuffer_io.cc:59-64).
//
// usage: ref_render scene.scn out.fjfb
// out.fjfb: "FJFB" int32 xres yres nchannels, float64 render_seconds,
//           float32[yres*xres*nchannels]
#include "parser.h"
#include "fj_scene_interface.h"
#include "fj_framebuffer.h"
#include "fj_callback.h"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>

using namespace fj;

struct Capture {
  std::chrono::steady_clock::time_point t0;
  double seconds;
  const char *out_path;
};

static Interrupt on_frame_start(void *data, const FrameInfo *info)
{
  Capture *cap = static_cast<Capture *>(data);
  cap->t0 = std::chrono::steady_clock::now();
  return CALLBACK_CONTINUE;
}

static Interrupt on_frame_done(void *data, const FrameInfo *info)
{
  Capture *cap = static_cast<Capture *>(data);
  const auto t1 = std::chrono::steady_clock::now();
  cap->seconds = std::chrono::duration<double>(t1 - cap->t0).count();

  const FrameBuffer *fb = info->framebuffer;
  const int32_t hdr[3] = {fb->GetWidth(), fb->GetHeight(), fb->GetChannelCount()};
  FILE *fp = fopen(cap->out_path, "wb");
  if (fp == NULL) {
    fprintf(stderr, "ref_render: cannot write %s\n", cap->out_path);
    return CALLBACK_CONTINUE;
  }
  fwrite("FJFB", 1, 4, fp);
  fwrite(hdr, sizeof(int32_t), 3, fp);
  fwrite(&cap->seconds, sizeof(double), 1, fp);
  fwrite(fb->GetReadOnly(0, 0, 0), sizeof(float), fb->GetSize(), fp);
  fclose(fp);
  return CALLBACK_CONTINUE;
}

static Interrupt on_tile(void *data, const TileInfo *info) { return CALLBACK_CONTINUE; }
static Interrupt on_sample(void *data) { return CALLBACK_CONTINUE; }

int main(int argc, const char **argv)
{
  if (argc != 3) {
    fprintf(stderr, "usage: ref_render scene.scn out.fjfb\n");
    return 2;
  }
  std::ifstream file(argv[1]);
  if (!file) {
    fprintf(stderr, "ref_render: cannot open %s\n", argv[1]);
    return 2;
  }

  Capture cap;
  cap.seconds = -1;
  cap.out_path = argv[2];

  Parser parser;
  std::string line;
  int n_renderers = 0;
  while (getline(file, line)) {
    std::istringstream iss(line);
    std::string head;
    iss >> head;
    if (head == "RenderScene") {
      // ID = type * 10^7 + index (src/fj_scene_interface.cc:44,1058-1075);
      // Type_Renderer == 8 (enum EntryType, same file :46-64). The scenes we
      // drive create exactly one renderer, so index 0.
      const ID ren = 8L * 10000000L + (n_renderers - 1);
      SiSetFrameReportCallback(ren, &cap, on_frame_start, NULL, on_frame_done);
      SiSetTileReportCallback(ren, &cap, on_tile, on_sample, on_tile);
    }
    const int err = parser.ParseLine(line);
    if (err) {
      std::cerr << "error: " << parser.GetErrorMessage() << ": "
                << parser.GetLineNumber() << ": " << line << std::endl;
      return 1;
    }
    if (head == "NewRenderer") {
      n_renderers++;
    }
  }
  printf("{\"ref_render_seconds\": %.6f}\n", cap.seconds);
  return 0;
}
