// fj_host_parser.cc -- the `scene` command language.
//
// Own line parser for the text the reference's bin/scene speaks (reference
// tools/scene_parser/parser.cc:45-96, command.cc:40-541): one command per line,
// whitespace tokens, '#' comments, exact arity, unique new-entry names, numbers
// via strtod plus the ORDER_* / sampler symbols, first failure aborts.
// Plus the extern "C" face of the library.
#include "fj_host.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>

using namespace fj;
using namespace fjhost;

namespace {

// argument kinds
enum Kind { NEW, REF, NUM, STR, LIGHT, GROUPNAME };

struct Cmd { const char *name; int nargs; Kind kinds[6]; };

const Cmd commands[] = {
  {"OpenPlugin", 2, {NEW, STR}},
  {"RenderScene", 1, {REF}},
  {"RunProcedure", 1, {REF}},
  {"SaveFrameBuffer", 2, {REF, STR}},
  {"AddObjectToGroup", 2, {REF, REF}},
  {"NewObjectInstance", 2, {NEW, REF}},
  {"NewFrameBuffer", 2, {NEW, STR}},
  {"NewObjectGroup", 1, {NEW}},
  {"NewPointCloud", 1, {NEW}},
  {"NewTurbulence", 1, {NEW}},
  {"NewProcedure", 2, {NEW, REF}},
  {"NewRenderer", 1, {NEW}},
  {"NewTexture", 2, {NEW, STR}},
  {"NewCamera", 2, {NEW, STR}},
  {"NewShader", 2, {NEW, REF}},
  {"NewVolume", 1, {NEW}},
  {"NewCurve", 1, {NEW}},
  {"NewLight", 2, {NEW, LIGHT}},
  {"NewMesh", 1, {NEW}},
  {"AssignFrameBuffer", 2, {REF, REF}},
  {"AssignObjectGroup", 3, {REF, STR, REF}},
  {"AssignPointCloud", 3, {REF, STR, REF}},
  {"AssignTurbulence", 3, {REF, STR, REF}},
  {"AssignTexture", 3, {REF, STR, REF}},
  {"AssignVolume", 3, {REF, STR, REF}},
  {"AssignCamera", 2, {REF, REF}},
  {"AssignShader", 3, {REF, GROUPNAME, REF}},
  {"AssignCurve", 3, {REF, STR, REF}},
  {"AssignMesh", 3, {REF, STR, REF}},
  {"SetProperty1", 3, {REF, STR, NUM}},
  {"SetProperty2", 4, {REF, STR, NUM, NUM}},
  {"SetProperty3", 5, {REF, STR, NUM, NUM, NUM}},
  {"SetProperty4", 6, {REF, STR, NUM, NUM, NUM, NUM}},
  {"SetStringProperty", 3, {REF, STR, STR}},
  {"SetSampleProperty3", 6, {REF, STR, NUM, NUM, NUM, NUM}},
  {"ShowPropertyList", 1, {STR}},
};

bool symbol_number(const std::string &s, double *out)
{
  static const char *orders[] = {"ORDER_SRT", "ORDER_STR", "ORDER_RST", "ORDER_RTS", "ORDER_TRS", "ORDER_TSR",
                                 "ORDER_XYZ", "ORDER_XZY", "ORDER_YXZ", "ORDER_YZX", "ORDER_ZXY", "ORDER_ZYX"};
  for (int i = 0; i < 12; i++) if (s == orders[i]) { *out = i; return true; }
  if (s == "FIXED_GRID_SAMPER") { *out = SI_FIXED_GRID_SAMPLER; return true; }   // (sic) the reference's spelling
  if (s == "ADAPTIVE_GRID_SAMPLER") { *out = SI_ADAPTIVE_GRID_SAMPLER; return true; }
  return false;
}

struct Parser {
  std::map<std::string, ID> names;
  std::string error;

  int parse_line(const std::string &line, bool echo)
  {
    std::vector<std::string> tok;
    std::istringstream iss(line);
    std::string s;
    while (iss >> s) tok.push_back(s);
    if (tok.empty() || tok[0][0] == '#') return 0;

    const Cmd *cmd = nullptr;
    for (const Cmd &c : commands) if (tok[0] == c.name) { cmd = &c; break; }
    if (!cmd) { error = "unknown command"; return -1; }
    const int nargs = (int) tok.size() - 1;
    if (nargs < cmd->nargs) { error = "too few arguments"; return -1; }
    if (nargs > cmd->nargs) { error = "too many arguments"; return -1; }

    ID ids[6] = {0, 0, 0, 0, 0, 0};
    double num[6] = {0, 0, 0, 0, 0, 0};
    const char *str[6] = {0, 0, 0, 0, 0, 0};
    std::string grp;
    for (int i = 0; i < cmd->nargs; i++) {
      const std::string &a = tok[i + 1];
      str[i] = a.c_str();
      switch (cmd->kinds[i]) {
      case NEW:
        if (names.count(a)) { error = "entry name already exists"; return -1; }
        break;
      case REF: {
        auto it = names.find(a);
        if (it == names.end()) { error = "entry name not found"; return -1; }
        ids[i] = it->second;
        break;
      }
      case NUM: {
        if (symbol_number(a, &num[i])) break;
        char *end = nullptr;
        num[i] = std::strtod(a.c_str(), &end);
        if (*end != '\0') { error = "bad number"; return -1; }
        break;
      }
      case LIGHT:
        if (a == "PointLight") num[i] = SI_POINT_LIGHT;
        else if (a == "GridLight") num[i] = SI_GRID_LIGHT;
        else if (a == "SphereLight") num[i] = SI_SPHERE_LIGHT;
        else if (a == "DomeLight") num[i] = SI_DOME_LIGHT;
        else { error = "bad light type"; return -1; }
        break;
      case GROUPNAME:
        grp = (a == "DEFAULT_SHADING_GROUP") ? "" : a;
        str[i] = grp.c_str();
        break;
      case STR:
        break;
      }
    }
    if (echo) {
      std::printf("-- %s: ", tok[0].c_str());
      for (int i = 1; i <= cmd->nargs; i++) std::printf("[%s]%s", tok[i].c_str(), i == cmd->nargs ? "\n" : " ");
    }

    const std::string &c = tok[0];
    ID new_id = 0;
    Status st = SI_SUCCESS;
    bool makes_entry = cmd->kinds[0] == NEW;
    if (c == "OpenPlugin") new_id = SiOpenPlugin(str[1]);
    else if (c == "RenderScene") st = SiRenderScene(ids[0]);
    else if (c == "RunProcedure") st = SiRunProcedure(ids[0]);
    else if (c == "SaveFrameBuffer") st = SiSaveFrameBuffer(ids[0], str[1]);
    else if (c == "AddObjectToGroup") st = SiAddObjectToGroup(ids[0], ids[1]);
    else if (c == "NewObjectInstance") new_id = SiNewObjectInstance(ids[1]);
    else if (c == "NewFrameBuffer") new_id = SiNewFrameBuffer(str[1]);
    else if (c == "NewObjectGroup") new_id = SiNewObjectGroup();
    else if (c == "NewPointCloud") new_id = SiNewPointCloud();
    else if (c == "NewTurbulence") new_id = SiNewTurbulence();
    else if (c == "NewProcedure") new_id = SiNewProcedure(ids[1]);
    else if (c == "NewRenderer") new_id = SiNewRenderer();
    else if (c == "NewTexture") new_id = SiNewTexture(str[1]);
    else if (c == "NewCamera") new_id = SiNewCamera(str[1]);
    else if (c == "NewShader") new_id = SiNewShader(ids[1]);
    else if (c == "NewVolume") new_id = SiNewVolume();
    else if (c == "NewCurve") new_id = SiNewCurve();
    else if (c == "NewLight") new_id = SiNewLight((int) num[1]);
    else if (c == "NewMesh") new_id = SiNewMesh();
    else if (c == "AssignFrameBuffer") st = SiAssignFrameBuffer(ids[0], ids[1]);
    else if (c == "AssignObjectGroup") st = SiAssignObjectGroup(ids[0], str[1], ids[2]);
    else if (c == "AssignPointCloud") st = SiAssignPointCloud(ids[0], str[1], ids[2]);
    else if (c == "AssignTurbulence") st = SiAssignTurbulence(ids[0], str[1], ids[2]);
    else if (c == "AssignTexture") st = SiAssignTexture(ids[0], str[1], ids[2]);
    else if (c == "AssignVolume") st = SiAssignVolume(ids[0], str[1], ids[2]);
    else if (c == "AssignCamera") st = SiAssignCamera(ids[0], ids[1]);
    else if (c == "AssignShader") st = SiAssignShader(ids[0], str[1], ids[2]);
    else if (c == "AssignCurve") st = SiAssignCurve(ids[0], str[1], ids[2]);
    else if (c == "AssignMesh") st = SiAssignMesh(ids[0], str[1], ids[2]);
    else if (c == "SetProperty1") st = SiSetProperty1(ids[0], str[1], num[2]);
    else if (c == "SetProperty2") st = SiSetProperty2(ids[0], str[1], num[2], num[3]);
    else if (c == "SetProperty3") st = SiSetProperty3(ids[0], str[1], num[2], num[3], num[4]);
    else if (c == "SetProperty4") st = SiSetProperty4(ids[0], str[1], num[2], num[3], num[4], num[5]);
    else if (c == "SetStringProperty") st = SiSetStringProperty(ids[0], str[1], str[2]);
    else if (c == "SetSampleProperty3") st = SiSetSampleProperty3(ids[0], str[1], num[2], num[3], num[4], num[5]);
    else if (c == "ShowPropertyList") {
      // "(Type) : (Name) : (Default)" like print_property_list, tools/scene_parser/command.cc:584-650
      const Property *p = SiGetPropertyList(str[0]);
      if (!p) std::printf("#   No property is available for %s\n", str[0]);
      for (; p && p->IsValid(); p++) {
        const Vector4 &d = p->GetDefaultValue();
        std::printf("#   %-15.15s : %-20.20s : (%g, %g, %g, %g)\n", p->GetTypeString(), p->GetName(), d.x, d.y, d.z, d.w);
      }
    }

    if (makes_entry) {
      if (new_id == SI_BADID) { error = g_last_error.empty() ? "command failed" : g_last_error; return -1; }
      names[tok[1]] = new_id;
    } else if (st == SI_FAIL) {
      // CommandResult::IsFail (tools/scene_parser/command.cc:731-734): status SI_FAIL and entry id SI_BADID -- and a
      // command that makes no entry never sets the id (CommandResult's constructor leaves SI_BADID, :687-690), so a
      // failed SetProperty* / Assign* aborts the reference's `scene` too (checked against the compiled reference:
      // tests/test_host_boundary.py::test_failed_commands_abort_like_the_reference).
      error = g_last_error.empty() ? "command failed" : g_last_error;
      return -1;
    }
    return 0;
  }
};

std::string last_parse_error;

}  // namespace

extern "C" {

int fj_scene_run_text(const char *text, int echo)
{
  if (!text) return -1;
  Parser parser;
  SiOpenScene();                      // Parser ctor, reference tools/scene_parser/parser.cc:30-38
  std::istringstream in(text);
  std::string line;
  int line_no = 0;
  last_parse_error.clear();
  g_last_error.clear();
  while (std::getline(in, line)) {
    line_no++;
    if (parser.parse_line(line, echo != 0)) {
      std::ostringstream os;
      os << "error: " << parser.error << ": " << line_no << ": " << line;
      last_parse_error = os.str();
      return line_no;
    }
  }
  return 0;
}

const char *fj_scene_last_error(void) { return last_parse_error.empty() ? g_last_error.c_str() : last_parse_error.c_str(); }

void fj_scene_set_deferred_render(int on) { g_deferred_render = on != 0; }

int fj_scene_get_desc(const fj_scene_desc **scene, const fj_render_desc **render)
{
  Scene *sc = get_scene();
  if (!sc || !sc->has_desc) return -1;
  if (scene) *scene = &sc->desc;
  if (render) *render = &sc->render;
  return 0;
}

const float *fj_framebuffer_data(long framebuffer, int *width, int *height, int *nchannels)
{
  Scene *sc = get_scene();
  if (!sc) return nullptr;
  const long type = framebuffer / TYPE_ID_OFFSET;
  const long idx = framebuffer - type * TYPE_ID_OFFSET;
  if (type != Type_FrameBuffer || idx < 0 || idx >= (long) sc->framebuffers.size()) return nullptr;
  const FrameBuffer *fb = sc->framebuffers[idx].get();
  if (width) *width = fb->GetWidth();
  if (height) *height = fb->GetHeight();
  if (nchannels) *nchannels = fb->GetChannelCount();
  return fb->GetReadOnly(0, 0, 0);
}

int fj_write_fb_file(const char *filename, int width, int height, int nchannels, const float *pixels)
{
  if (!filename || !pixels || width <= 0 || height <= 0 || nchannels <= 0) return -1;
  FrameBuffer fb;
  fb.Resize(width, height, nchannels);
  std::memcpy(fb.GetWritable(0, 0, 0), pixels, sizeof(float) * (size_t) width * height * nchannels);
  return WriteFrameBuffer(filename, fb);
}

int fj_scene_property_table(const char *type_name, char *out, int out_size)
{
  const Property *p = SiGetPropertyList(type_name);
  if (!p || !out || out_size <= 0) return -1;
  std::string text;
  char line[256];
  for (; p->IsValid(); p++) {
    const Vector4 &d = p->GetDefaultValue();
    std::snprintf(line, sizeof(line), "%s %s %.17g %.17g %.17g %.17g\n", p->GetTypeString(), p->GetName(), d.x, d.y, d.z, d.w);
    text += line;
  }
  if ((int) text.size() + 1 > out_size) return -1;
  std::memcpy(out, text.c_str(), text.size() + 1);
  return (int) text.size();
}

int fj_scene_last_stats(fj_render_stats *out)
{
  if (!out) return -1;
  *out = g_last_stats;
  return 0;
}

// SiGetPropertyList for C callers: the table is opaque there, its entries are read through accessors
const void *fj_SiGetPropertyList(const char *type_name) { return (const void *) SiGetPropertyList(type_name); }
int fj_property_is_valid(const void *table, int k) { return (table && k >= 0 && ((const Property *) table)[k].IsValid()) ? 1 : 0; }
const char *fj_property_name(const void *table, int k) { return fj_property_is_valid(table, k) ? ((const Property *) table)[k].GetName() : nullptr; }
const char *fj_property_type_string(const void *table, int k) { return fj_property_is_valid(table, k) ? ((const Property *) table)[k].GetTypeString() : nullptr; }
int fj_property_default(const void *table, int k, double out4[4])
{
  if (!fj_property_is_valid(table, k) || !out4) return -1;
  const Vector4 &d = ((const Property *) table)[k].GetDefaultValue();
  out4[0] = d.x; out4[1] = d.y; out4[2] = d.z; out4[3] = d.w;
  return 0;
}

// ---- 1:1 C spellings of the Si* functions
int  fj_SiGetErrorNo(void) { return SiGetErrorNo(); }
long fj_SiOpenPlugin(const char *f) { return SiOpenPlugin(f); }
int  fj_SiOpenScene(void) { return SiOpenScene(); }
int  fj_SiCloseScene(void) { return SiCloseScene(); }
int  fj_SiRenderScene(long r) { return SiRenderScene(r); }
int  fj_SiSaveFrameBuffer(long fb, const char *f) { return SiSaveFrameBuffer(fb, f); }
int  fj_SiRunProcedure(long p) { return SiRunProcedure(p); }
int  fj_SiAddObjectToGroup(long g, long o) { return SiAddObjectToGroup(g, o); }
long fj_SiNewObjectInstance(long p) { return SiNewObjectInstance(p); }
long fj_SiNewFrameBuffer(const char *a) { return SiNewFrameBuffer(a); }
long fj_SiNewObjectGroup(void) { return SiNewObjectGroup(); }
long fj_SiNewPointCloud(void) { return SiNewPointCloud(); }
long fj_SiNewTurbulence(void) { return SiNewTurbulence(); }
long fj_SiNewProcedure(long p) { return SiNewProcedure(p); }
long fj_SiNewRenderer(void) { return SiNewRenderer(); }
long fj_SiNewTexture(const char *f) { return SiNewTexture(f); }
long fj_SiNewCamera(const char *a) { return SiNewCamera(a); }
long fj_SiNewShader(long p) { return SiNewShader(p); }
long fj_SiNewVolume(void) { return SiNewVolume(); }
long fj_SiNewCurve(void) { return SiNewCurve(); }
long fj_SiNewLight(int t) { return SiNewLight(t); }
long fj_SiNewMesh(void) { return SiNewMesh(); }
int  fj_SiAssignFrameBuffer(long r, long f) { return SiAssignFrameBuffer(r, f); }
int  fj_SiAssignObjectGroup(long i, const char *n, long g) { return SiAssignObjectGroup(i, n, g); }
int  fj_SiAssignPointCloud(long i, const char *n, long p) { return SiAssignPointCloud(i, n, p); }
int  fj_SiAssignTurbulence(long i, const char *n, long t) { return SiAssignTurbulence(i, n, t); }
int  fj_SiAssignTexture(long i, const char *n, long t) { return SiAssignTexture(i, n, t); }
int  fj_SiAssignVolume(long i, const char *n, long v) { return SiAssignVolume(i, n, v); }
int  fj_SiAssignCamera(long r, long c) { return SiAssignCamera(r, c); }
int  fj_SiAssignShader(long o, const char *g, long s) { return SiAssignShader(o, g, s); }
int  fj_SiAssignCurve(long i, const char *n, long c) { return SiAssignCurve(i, n, c); }
int  fj_SiAssignMesh(long i, const char *n, long m) { return SiAssignMesh(i, n, m); }
int  fj_SiSetProperty1(long i, const char *n, double a) { return SiSetProperty1(i, n, a); }
int  fj_SiSetProperty2(long i, const char *n, double a, double b) { return SiSetProperty2(i, n, a, b); }
int  fj_SiSetProperty3(long i, const char *n, double a, double b, double c) { return SiSetProperty3(i, n, a, b, c); }
int  fj_SiSetProperty4(long i, const char *n, double a, double b, double c, double d) { return SiSetProperty4(i, n, a, b, c, d); }
int  fj_SiSetStringProperty(long i, const char *n, const char *s) { return SiSetStringProperty(i, n, s); }
int  fj_SiSetSampleProperty3(long i, const char *n, double a, double b, double c, double t) { return SiSetSampleProperty3(i, n, a, b, c, t); }
int  fj_SiSetFrameReportCallback(long r, void *data, fj_frame_callback start, fj_frame_callback abort_, fj_frame_callback done)
{
  return SiSetFrameReportCallback(r, data, (FrameStartCallback) start, (FrameAbortCallback) abort_, (FrameDoneCallback) done);
}
int  fj_SiSetTileReportCallback(long r, void *data, fj_tile_callback start, fj_sample_callback sample_done, fj_tile_callback done)
{
  return SiSetTileReportCallback(r, data, (TileStartCallback) start, (SampleDoneCallback) sample_done, (TileDoneCallback) done);
}

}  // extern "C"
