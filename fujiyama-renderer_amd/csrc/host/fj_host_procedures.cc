// fj_host_procedures.cc -- built-in geometry procedures of libfjscene.so.
//
// StanfordPlyProcedure: PLY -> Mesh with the behaviour of the reference's
// procedures/stanfordply_procedure/ply2mesh.cc:51-171 (x y z as Real, optional
// uv1/uv2 as float, polygons fan-triangulated (v0, v[k+1], v[k+2]), then
// Mesh::ComputeNormals and Mesh::ComputeBounds).  The PLY reader itself is our
// own (the reference vendors plyfile.c); it accepts ascii and both binary
// byte orders with arbitrary extra properties.
#include <algorithm>
#include <cmath>
#include <map>
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
  mesh->vertex_N.clear();
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

// ---------------------------------------------------------------- WavefrontObjProcedure
// OBJ -> Mesh with the behaviour of the reference's procedures/wavefrontobj_procedure (ObjParser.cc:158-236 line grammar,
// ObjBuffer.h:55-146 callbacks, ObjBuffer.cc:6-116 mesh assembly): the one producer of FACE GROUPS (`g name`) and of
// per-CORNER normals (`vn` + `f v//vn`) among the reference's procedures.
//   v x y z [w]        position (doubles, as written -- such a mesh need not be f32-exact)
//   vn x y z           normal value;  vt is parsed and dropped (ObjBufferToMesh sets no texture coordinates)
//   f a b c d ...      polygon, fan-triangulated (a, k+1, k+2); a corner is v | v/vt | v//vn | v/vt/vn; an index i > 0 names
//                      element i - 1, i < 0 counts back from the elements read SO FAR, 0 stays 0 (ObjParser.cc:128-137)
//   g name ...         faces from here on belong to group `name` (first word; ids in order of first appearance, "" = 0)
// No `vn` anywhere: point normals = sum of the unit face normals of the faces at a point, in face order, normalised
// (ObjBufferComputeNormals -- a corner repeated within a face counts twice there, unlike Mesh::ComputeNormals).  With `vn`:
// Mesh.vertex_N[3 f + k] = the value the corner's vn index names; every face must carry vn indices then (the reference
// reads past its index array otherwise).
namespace {
struct ObjScan {
  const char *p, *end;
  void blanks() { while (p < end && (*p == ' ' || *p == '\t' || *p == '\f' || *p == '\v' || *p == '\r')) p++; }
  bool at_eol() const { return p >= end || *p == '\n'; }
  void next_line() { while (p < end && *p != '\n') p++; if (p < end) p++; }
  std::string word() { blanks(); const char *b = p; while (p < end && !(*p == ' ' || *p == '\t' || *p == '\f' || *p == '\v' || *p == '\r' || *p == '\n')) p++; return std::string(b, p); }
  bool number(double *out) { blanks(); if (at_eol()) return false; char *e = nullptr; const double v = std::strtod(p, &e); if (e == p) return false; p = e; *out = v; return true; }
  bool integer(long *out) { if (at_eol()) return false; char *e = nullptr; const long v = std::strtol(p, &e, 10); if (e == p) return false; p = e; *out = v; return true; }
};
inline long obj_index(long count, long i) { return i > 0 ? i - 1 : (i < 0 ? i + count : 0); }
}

int ReadObjFile(const std::string &path, Mesh *mesh, std::string *err)
{
  std::vector<char> text;
  {
    FILE *fp = std::fopen(path.c_str(), "rb");
    if (!fp) { *err = "cannot open OBJ file: " + path; return -1; }
    std::fseek(fp, 0, SEEK_END);
    const long n = std::ftell(fp);
    std::fseek(fp, 0, SEEK_SET);
    text.resize(n > 0 ? (size_t) n + 1 : 1);
    const size_t got = n > 0 ? std::fread(text.data(), 1, (size_t) n, fp) : 0;
    std::fclose(fp);
    text[got] = '\n';
    text.resize(got + 1);
  }
  std::vector<double> P, Nv;
  std::vector<int32_t> tri, tri_n, group_of_face;
  std::map<std::string, int> group_id;
  group_id[""] = 0;
  long n_v = 0, n_vt = 0, n_vn = 0;
  int current_group = 0;
  std::vector<long> cv, cn;            // the corners of one `f` line
  ObjScan sc{text.data(), text.data() + text.size()};
  for (; sc.p < sc.end; sc.next_line()) {
    const std::string tag = sc.word();
    if (tag == "v" || tag == "vn" || tag == "vt") {
      double c[4] = {0, 0, 0, 0};
      for (int k = 0; k < 4 && sc.number(&c[k]); k++) {}
      if (tag == "v") { P.insert(P.end(), c, c + 3); n_v++; }
      else if (tag == "vn") { Nv.insert(Nv.end(), c, c + 3); n_vn++; }
      else n_vt++;
    } else if (tag == "f") {
      cv.clear(); cn.clear();
      for (;;) {
        sc.blanks();
        long v = 0, vt = 0, vn = 0;
        bool has_vn = false;
        if (!sc.integer(&v)) break;
        if (sc.p < sc.end && *sc.p == '/') {
          sc.p++;
          if (sc.p < sc.end && *sc.p == '/') { sc.p++; has_vn = sc.integer(&vn); }
          else { (void) sc.integer(&vt); if (sc.p < sc.end && *sc.p == '/') { sc.p++; has_vn = sc.integer(&vn); } }
        }
        cv.push_back(obj_index(n_v, v));
        if (has_vn) cn.push_back(obj_index(n_vn, vn));
      }
      (void) n_vt;
      for (size_t k = 0; k + 2 < cv.size(); k++) {
        tri.push_back((int32_t) cv[0]); tri.push_back((int32_t) cv[k + 1]); tri.push_back((int32_t) cv[k + 2]);
        if (!cn.empty()) {
          if (cn.size() != cv.size()) { *err = "OBJ face with normals on some corners only: " + path; return -1; }
          tri_n.push_back((int32_t) cn[0]); tri_n.push_back((int32_t) cn[k + 1]); tri_n.push_back((int32_t) cn[k + 2]);
        }
        group_of_face.push_back(current_group);
      }
    } else if (tag == "g") {
      const std::string name = sc.word();
      if (name.empty()) { *err = "OBJ `g` without a name: " + path; return -1; }     // (the reference indexes an empty list there)
      auto it = group_id.find(name);
      if (it == group_id.end()) { const int id = (int) group_id.size(); group_id[name] = id; current_group = id; }
      else current_group = it->second;
    }
  }
  const size_t nf = tri.size() / 3;
  if (P.empty() || nf == 0) { *err = "OBJ file without vertices or faces: " + path; return -1; }
  for (int32_t i : tri) if (i < 0 || i >= n_v) { *err = "OBJ face index out of range: " + path; return -1; }
  for (int32_t i : tri_n) if (i < 0 || i >= n_vn) { *err = "OBJ normal index out of range: " + path; return -1; }
  if (!Nv.empty() && tri_n.size() != tri.size()) { *err = "OBJ file with `vn` values but faces without normal indices: " + path; return -1; }

  *mesh = Mesh();                                     // Mesh::Clear, src/fj_mesh.cc:127-135
  mesh->P.swap(P);
  mesh->indices.swap(tri);
  if (Nv.empty()) {
    // ObjBufferComputeNormals, ObjBuffer.cc:80-116: TriComputeFaceNormal = Normalize(Cross(P1 - P0, P2 - P0)), added to the three
    // corners' points in face order (a repeated corner adds twice), then Normalize (a zero vector stays)
    std::vector<double> &N = mesh->N;
    N.assign(mesh->P.size(), 0.);
    auto normalize = [](double *v) { const double len = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); if (len == 0) return; const double inv = 1. / len; v[0] *= inv; v[1] *= inv; v[2] *= inv; };
    for (size_t f = 0; f < nf; f++) {
      const int32_t *ix = &mesh->indices[3 * f];
      const double *p0 = &mesh->P[3 * (size_t) ix[0]], *p1 = &mesh->P[3 * (size_t) ix[1]], *p2 = &mesh->P[3 * (size_t) ix[2]];
      const double a[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, b[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
      double ng[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
      normalize(ng);
      for (int k = 0; k < 3; k++) for (int c = 0; c < 3; c++) N[3 * (size_t) ix[k] + c] += ng[c];
    }
    for (size_t i = 0; i < N.size(); i += 3) normalize(&N[i]);
  } else {
    mesh->vertex_N.resize(nf * 9);
    for (size_t c = 0; c < nf * 3; c++) for (int k = 0; k < 3; k++) mesh->vertex_N[3 * c + k] = Nv[3 * (size_t) tri_n[c] + k];
  }
  mesh->face_group.assign(group_of_face.begin(), group_of_face.end());
  mesh->face_group_name = group_id;
  mesh->ComputeBounds();
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
  if (proc->plugin->name == "WavefrontObjProcedure") {
    // wavefrontobj_procedure.cc:34-39,77-99: properties "mesh", "filepath", "io_mode" ("r" reads, anything else fails)
    if (proc->mesh < 0) { *err = "WavefrontObjProcedure: no mesh assigned"; return -1; }
    auto fp = proc->strings.find("filepath");
    if (fp == proc->strings.end() || fp->second.empty()) { *err = "WavefrontObjProcedure: no filepath"; return -1; }
    auto mode = proc->strings.find("io_mode");
    if (mode != proc->strings.end() && mode->second != "r") { *err = "WavefrontObjProcedure: only io_mode r is supported"; return -1; }
    return ReadObjFile(fp->second, sc->meshes[proc->mesh].get(), err);
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
