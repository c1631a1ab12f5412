// scene_main.cc -- `scene [path]`: feed a scene-description file (or stdin) to
// libfjscene.so, like the reference's bin/scene (tools/scene_parser/main.cc:9-47).
#include "fj_scene_interface.h"

#include <chrono>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

int main(int argc, const char **argv)
{
  std::stringstream text;
  if (argc == 2) {
    std::ifstream file(argv[1]);
    if (!file) { std::cerr << "error: Could not open file: " << argv[1] << std::endl; return -1; }
    text << file.rdbuf();
  } else if (argc == 1) {
    text << std::cin.rdbuf();
  } else {
    std::cerr << "usage: scene [path]" << std::endl;
    return 0;
  }
  const auto t0 = std::chrono::steady_clock::now();
  const int err = fj_scene_run_text(text.str().c_str(), 1);
  const auto t1 = std::chrono::steady_clock::now();
  if (err) { std::cerr << fj_scene_last_error() << std::endl; return -1; }
  fj_render_stats st;
  if (fj_scene_last_stats(&st) == 0 && st.render_seconds > 0)
    std::cout << "# RenderScene " << st.render_seconds << " s (prepare " << st.prepare_seconds << " s)\n";
  fj::SiCloseScene();
  if (getenv("FJ_SCENE_TIMING"))
    std::cout << "# commands " << std::chrono::duration<double>(t1 - t0).count() << " s, close "
              << std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() << " s\n";
  return 0;
}
