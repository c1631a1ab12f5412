// fj_host_procedures.cc -- built-in geometry procedures of libfjscene.so.
//
// StanfordPlyProcedure: PLY -> Mesh with the behaviour of the reference's
// procedures/stanfordply_procedure/ply2mesh.cc:51-171 (x y z as Real, optional
// uv1/uv2 as float, polygons fan-triangulated (v0, v[k+1], v[k+2]), then
// Mesh::ComputeNormals and Mesh::ComputeBounds).  The PLY reader itself is our
// own (the reference vendors plyfile.c); it accepts ascii and both binary
// byte orders with arbitrary extra properties.
#include <algorithm>
#include <string>
#include <vector>
#include <thread>
#include "fj_host.h"

#include <chrono>
#include <cstdint>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <functional>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

namespace fjhost {

namespace {

enum PlyType { T_NONE, T_I8, T_U8, T_I16, T_U16, T_I32, T_U32, T_F32, T_F64 };

PlyType parse_type(const std::string &s)
{
  if (s == "char" || s == "int8") return T_I8;
  if (s == "uchar" || s == "uint8") return T_U8;
  if (s == "short" || s == "int16") return T_I16;
  if (s == "ushort" || s == "uint16") return T_U16;
  if (s == "int" || s == "int32") return T_I32;
  if (s == "uint" || s == "uint32") return T_U32;
  if (s == "float" || s == "float32") return T_F32;
  if (s == "double" || s == "float64") return T_F64;
  return T_NONE;
}

int type_size(PlyType t)
{
  switch (t) {
  case T_I8: case T_U8: return 1;
  case T_I16: case T_U16: return 2;
  case T_I32: case T_U32: case T_F32: return 4;
  case T_F64: return 8;
  default: return 0;
  }
}

struct PlyProp { std::string name; bool is_list; PlyType count_type, type; };
struct PlyElem { std::string name; long count; std::vector<PlyProp> props; };

struct Reader {
  std::ifstream f;
  int format;   // 0 ascii, 1 little, 2 big
  bool ok;
  // binary files: everything after the header in one piece (a stream read per number cost 0.8 s of the 1.5 s a 7.2 M-triangle scene took to
  // assemble) -- the file is mapped, not copied (a 104 MB read into a zero-filled vector was a third of what was left of the procedure)
  struct View { const char *p = nullptr; size_t n = 0; const char *data() const { return p; } size_t size() const { return n; } } body;
  void *map = nullptr; size_t map_len = 0;
  std::vector<char> copy;
  size_t pos = 0;
  ~Reader() { if (map) munmap(map, map_len); }
  void slurp(const std::string &path)
  {
    const std::streampos here = f.tellg();
    f.seekg(0, std::ios::end);
    const std::streampos end = f.tellg();
    f.seekg(here);
    pos = 0;
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd >= 0) {
      void *m = mmap(nullptr, (size_t) end, PROT_READ, MAP_PRIVATE, fd, 0);
      ::close(fd);
      if (m != MAP_FAILED) { map = m; map_len = (size_t) end; body.p = (const char *) m + (size_t) here; body.n = (size_t) (end - here); return; }
    }
    copy.resize((size_t) (end - here));
    if (!copy.empty()) f.read(copy.data(), (std::streamsize) copy.size());
    if (!f) ok = false;
    body.p = copy.data(); body.n = copy.size();
  }
  double read_number(PlyType t)
  {
    if (format == 0) {
      double v = 0;
      if (!(f >> v)) ok = false;
      return v;
    }
    unsigned char b[8];
    const int n = type_size(t);
    if (pos + (size_t) n > body.size()) { ok = false; return 0; }
    std::memcpy(b, body.data() + pos, (size_t) n);
    pos += (size_t) n;
    if (format == 2) for (int i = 0; i < n / 2; i++) std::swap(b[i], b[n - 1 - i]);
    switch (t) {
    case T_I8: { int8_t v; std::memcpy(&v, b, 1); return v; }
    case T_U8: { uint8_t v; std::memcpy(&v, b, 1); return v; }
    case T_I16: { int16_t v; std::memcpy(&v, b, 2); return v; }
    case T_U16: { uint16_t v; std::memcpy(&v, b, 2); return v; }
    case T_I32: { int32_t v; std::memcpy(&v, b, 4); return v; }
    case T_U32: { uint32_t v; std::memcpy(&v, b, 4); return v; }
    case T_F32: { float v; std::memcpy(&v, b, 4); return v; }
    case T_F64: { double v; std::memcpy(&v, b, 8); return v; }
    default: return 0;
    }
  }
};

}  // namespace

int ReadPlyFile(const std::string &path, Mesh *mesh, std::string *err)
{
  const auto t0_ = std::chrono::steady_clock::now();
  Reader rd;
  rd.ok = true;
  rd.format = -1;
  rd.f.open(path.c_str(), std::ios::binary);
  if (!rd.f) { *err = "couldn't open input file: " + path; return -1; }

  std::string line;
  std::getline(rd.f, line);
  if (line.substr(0, 3) != "ply") { *err = "not a PLY file: " + path; return -1; }
  std::vector<PlyElem> elems;
  while (std::getline(rd.f, line)) {
    if (!line.empty() && line[line.size() - 1] == '\r') line.erase(line.size() - 1);
    std::istringstream iss(line);
    std::string key;
    iss >> key;
    if (key == "format") {
      std::string fmt;
      iss >> fmt;
      rd.format = fmt == "ascii" ? 0 : (fmt == "binary_little_endian" ? 1 : (fmt == "binary_big_endian" ? 2 : -1));
    } else if (key == "element") {
      PlyElem e;
      iss >> e.name >> e.count;
      elems.push_back(e);
    } else if (key == "property" && !elems.empty()) {
      PlyProp p;
      std::string t;
      iss >> t;
      if (t == "list") {
        std::string ct, vt;
        iss >> ct >> vt >> p.name;
        p.is_list = true; p.count_type = parse_type(ct); p.type = parse_type(vt);
      } else {
        iss >> p.name;
        p.is_list = false; p.count_type = T_NONE; p.type = parse_type(t);
      }
      if (p.type == T_NONE) { *err = "bad PLY property type in " + path; return -1; }
      elems.back().props.push_back(p);
    } else if (key == "end_header") {
      break;
    }
  }
  if (rd.format < 0) { *err = "unknown PLY format in " + path; return -1; }
  if (rd.format != 0) rd.slurp(path);

  std::vector<double> P;
  std::vector<float> uv;
  std::vector<int32_t> indices;
  bool has_uv = false;
  long nverts = 0;
  std::vector<double> list_vals;
  for (const PlyElem &e : elems) {
    const bool is_vertex = e.name == "vertex", is_face = e.name == "face";
    int ix = -1, iy = -1, iz = -1, iu = -1, iv = -1, ilist = -1;
    for (size_t k = 0; k < e.props.size(); k++) {
      const std::string &n = e.props[k].name;
      if (is_vertex) {
        if (n == "x") ix = (int) k; else if (n == "y") iy = (int) k; else if (n == "z") iz = (int) k;
        else if (n == "uv1") iu = (int) k; else if (n == "uv2") iv = (int) k;
      } else if (is_face && e.props[k].is_list && (n == "vertex_indices" || n == "vertex_index")) ilist = (int) k;
    }
    if (is_vertex) {
      nverts = e.count;
      P.assign(3 * (size_t) nverts, 0.);
      has_uv = iu >= 0 || iv >= 0;
      if (has_uv) uv.assign(2 * (size_t) nverts, 0.f);
    }
    // Fast paths for little-endian binary files (what the converters write): fixed-size vertex records are decoded on the host threads; so are
    // faces that all have the same number of corners (one list property, verified before anything is written).  Same values as the loop below.
    if (rd.format == 1 && e.count > 0) {
      const unsigned hc = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
      const unsigned nt = (unsigned) std::max<long>(1, std::min<long>((long) hc, e.count / 65536 + 1));
      auto run = [&](const std::function<void(long, long)> &fn) {
        if (nt == 1) { fn(0, e.count); return; }
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++) th.emplace_back(fn, e.count * t / nt, e.count * (t + 1) / nt);
        for (auto &x : th) x.join();
      };
      auto num = [](const char *p, PlyType t) -> double {
        switch (t) {
        case T_I8: { int8_t v; std::memcpy(&v, p, 1); return v; }
        case T_U8: { uint8_t v; std::memcpy(&v, p, 1); return v; }
        case T_I16: { int16_t v; std::memcpy(&v, p, 2); return v; }
        case T_U16: { uint16_t v; std::memcpy(&v, p, 2); return v; }
        case T_I32: { int32_t v; std::memcpy(&v, p, 4); return v; }
        case T_U32: { uint32_t v; std::memcpy(&v, p, 4); return v; }
        case T_F32: { float v; std::memcpy(&v, p, 4); return v; }
        case T_F64: { double v; std::memcpy(&v, p, 8); return v; }
        default: return 0;
        }
      };
      bool fixed = true;
      size_t stride = 0;
      std::vector<size_t> off(e.props.size(), 0);
      for (size_t k = 0; k < e.props.size(); k++) { if (e.props[k].is_list) fixed = false; off[k] = stride; stride += (size_t) type_size(e.props[k].type); }
      if (fixed && stride > 0) {
        if (rd.pos + stride * (size_t) e.count > rd.body.size()) { rd.ok = false; break; }
        if (is_vertex) {
          const char *base = rd.body.data() + rd.pos;
          run([&](long i0, long i1) {
            for (long i = i0; i < i1; i++) {
              const char *r = base + stride * (size_t) i;
              if (ix >= 0) P[3 * i] = num(r + off[ix], e.props[ix].type);
              if (iy >= 0) P[3 * i + 1] = num(r + off[iy], e.props[iy].type);
              if (iz >= 0) P[3 * i + 2] = num(r + off[iz], e.props[iz].type);
              if (iu >= 0) uv[2 * i] = (float) num(r + off[iu], e.props[iu].type);
              if (iv >= 0) uv[2 * i + 1] = (float) num(r + off[iv], e.props[iv].type);
            }
          });
        }
        rd.pos += stride * (size_t) e.count;
        continue;
      }
      if (is_face && e.props.size() == 1 && ilist == 0) {
        const PlyProp &p = e.props[0];
        const size_t cs = (size_t) type_size(p.count_type), vs = (size_t) type_size(p.type);
        if (rd.pos + cs <= rd.body.size()) {
          const int n = (int) num(rd.body.data() + rd.pos, p.count_type);
          const size_t rec = cs + vs * (size_t) (n > 0 ? n : 0);
          if (n >= 3 && n <= 255 && rd.pos + rec * (size_t) e.count <= rd.body.size()) {
            const char *base = rd.body.data() + rd.pos;
            std::vector<char> same(nt, 1);
            if (nt == 1) { for (long i = 0; i < e.count; i++) if ((int) num(base + rec * (size_t) i, p.count_type) != n) { same[0] = 0; break; } }
            else {
              std::vector<std::thread> th;
              for (unsigned t = 0; t < nt; t++) th.emplace_back([&, t]() {
                for (long i = e.count * t / nt; i < e.count * (t + 1) / nt; i++) if ((int) num(base + rec * (size_t) i, p.count_type) != n) { same[t] = 0; break; }
              });
              for (auto &x : th) x.join();
            }
            bool all_same = true;
            for (char c : same) all_same = all_same && c;
            if (all_same) {
              const size_t tri0 = indices.size();
              indices.resize(tri0 + (size_t) e.count * 3 * (size_t) (n - 2));
              run([&](long i0, long i1) {
                for (long i = i0; i < i1; i++) {
                  const char *r = base + rec * (size_t) i + cs;
                  const int32_t a0 = (int32_t) num(r, p.type);
                  int32_t *out = &indices[tri0 + (size_t) i * 3 * (size_t) (n - 2)];
                  for (int j = 0; j < n - 2; j++) {     // n triangles in a polygon is (n vertices - 2)
                    out[3 * j] = a0;
                    out[3 * j + 1] = (int32_t) num(r + vs * (size_t) (j + 1), p.type);
                    out[3 * j + 2] = (int32_t) num(r + vs * (size_t) (j + 2), p.type);
                  }
                }
              });
              rd.pos += rec * (size_t) e.count;
              continue;
            }
          }
        }
      }
    }
    for (long i = 0; i < e.count && rd.ok; i++) {
      for (size_t k = 0; k < e.props.size(); k++) {
        const PlyProp &p = e.props[k];
        if (!p.is_list) {
          const double v = rd.read_number(p.type);
          if (is_vertex) {
            if ((int) k == ix) P[3 * i] = v; else if ((int) k == iy) P[3 * i + 1] = v; else if ((int) k == iz) P[3 * i + 2] = v;
            else if ((int) k == iu) uv[2 * i] = (float) v; else if ((int) k == iv) uv[2 * i + 1] = (float) v;
          }
        } else {
          const int n = (int) rd.read_number(p.count_type);
          list_vals.resize(n > 0 ? n : 0);
          for (int j = 0; j < n; j++) list_vals[j] = rd.read_number(p.type);
          if (is_face && (int) k == ilist)
            for (int j = 0; j < n - 2; j++) {     // n triangles in a polygon is (n vertices - 2)
              indices.push_back((int32_t) list_vals[0]);
              indices.push_back((int32_t) list_vals[j + 1]);
              indices.push_back((int32_t) list_vals[j + 2]);
            }
        }
      }
    }
  }
  if (!rd.ok) { *err = "truncated PLY file: " + path; return -1; }
  for (int32_t i : indices)
    if (i < 0 || i >= nverts) { *err = "PLY face index out of range: " + path; return -1; }

  const auto tA_ = std::chrono::steady_clock::now();
  mesh->P.swap(P);
  mesh->indices.swap(indices);
  mesh->uv.swap(uv);
  mesh->N.clear();
  mesh->velocity.clear();
  mesh->face_group.clear();
  mesh->ComputeNormals();
  const auto tB_ = std::chrono::steady_clock::now();
  mesh->ComputeBounds();
  if (getenv("FJ_SCENE_TIMING"))
    fprintf(stderr, "fjhost: ply %s: read %.3f s, normals %.3f s, bounds %.3f s\n", path.c_str(), std::chrono::duration<double>(tA_ - t0_).count(),
        std::chrono::duration<double>(tB_ - tA_).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - tB_).count());
  return 0;
}

int RunCurveGenerator(Scene *sc, Procedure *proc, std::string *err);

int RunProcedure(Scene *sc, Procedure *proc, std::string *err)
{
  if (proc->plugin->name == "StanfordPlyProcedure") {
    // stanfordply_procedure.cc: properties "mesh", "filepath", "io_mode" ("r" reads)
    if (proc->mesh < 0) { *err = "StanfordPlyProcedure: no mesh assigned"; return -1; }
    auto fp = proc->strings.find("filepath");
    if (fp == proc->strings.end()) { *err = "StanfordPlyProcedure: no filepath"; return -1; }
    auto mode = proc->strings.find("io_mode");
    if (mode != proc->strings.end() && mode->second != "r") { *err = "StanfordPlyProcedure: only io_mode r is supported"; return -1; }
    return ReadPlyFile(fp->second, sc->meshes[proc->mesh].get(), err);
  }
  if (proc->plugin->name == "CurveGeneratorProcedure") return RunCurveGenerator(sc, proc, err);
  if (proc->plugin->name == "VelocityGeneratorProcedure") return RunVelocityGenerator(sc, proc, err);
  *err = "procedure " + proc->plugin->name + " is not built in";
  return -1;
}

// .fb writer: the reference's plain-text PTO format, src/fj_framebuffer_io.cc:46-68
int WriteFrameBuffer(const std::string &filename, const fj::FrameBuffer &fb)
{
  // The reference's .fb is TEXT (FbSaveCroppedData -> WritePto*, src/fj_framebuffer_io.cc: one line of four numbers per pixel, written with
  // ostream's default float format = printf's "%g"): 8.3 M numbers at 1080p.  Through one ostream that was 0.8 s -- half of bin/scene's 1.5 s on
  // the headline scene, five times the frame it saves --, so the rows are formatted on the host threads (snprintf "%g": the same characters, pinned
  // byte for byte against the reference's writer in tests/test_oracle_golden.py) and written in order.
  FILE *fp = std::fopen(filename.c_str(), "wb");
  if (!fp) return -1;
  const int W = fb.GetWidth(), H = fb.GetHeight(), nc = fb.GetChannelCount();
  std::string head = "#PTO Plain Text Object\n#Fujiyama Renderer FrameBuffer\n";     // WritePtoHeader, src/fj_pto.h:15-21
  head += "resolution " + std::to_string(W) + " " + std::to_string(H) + "\n";
  head += "channel_count " + std::to_string(nc) + "\n";
  head += "begin pixels\n";
  const unsigned hc = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
  const unsigned nt = (unsigned) std::max(1, std::min<int>((int) hc, H));
  std::vector<std::string> part(nt);
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; t++) th.emplace_back([&, t]() {
    const int y0 = (int) ((long long) H * t / nt), y1 = (int) ((long long) H * (t + 1) / nt);
    std::string &out = part[t];
    out.reserve((size_t) (y1 - y0) * (size_t) W * 40);
    char buf[128];
    for (int y = y0; y < y1; y++)
      for (int x = 0; x < W; x++) {
        const float *p = fb.GetReadOnly(x, y, 0);
        float c[4] = {0, 0, 0, 0};
        if (nc == 1) { c[0] = c[1] = c[2] = p[0]; c[3] = 1; }
        else if (nc == 3) { c[0] = p[0]; c[1] = p[1]; c[2] = p[2]; c[3] = 1; }
        else if (nc == 4) { c[0] = p[0]; c[1] = p[1]; c[2] = p[2]; c[3] = p[3]; }
        const int n = std::snprintf(buf, sizeof(buf), "%g %g %g %g\n", (double) c[0], (double) c[1], (double) c[2], (double) c[3]);
        out.append(buf, (size_t) n);
      }
  });
  for (auto &t : th) t.join();
  bool ok = std::fwrite(head.data(), 1, head.size(), fp) == head.size();
  for (unsigned t = 0; t < nt && ok; t++) ok = std::fwrite(part[t].data(), 1, part[t].size(), fp) == part[t].size();
  static const char tail[] = "end pixels\n";
  ok = ok && std::fwrite(tail, 1, sizeof(tail) - 1, fp) == sizeof(tail) - 1;
  return (std::fclose(fp) == 0 && ok) ? 0 : -1;
}

}  // namespace fjhost
