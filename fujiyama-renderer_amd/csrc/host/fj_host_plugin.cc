// fj_host_plugin.cc -- the Shader plugin ABI of libfjscene.so (include/fj_plugin_abi.h).
//
// Two halves:
//   * the symbols a reference shader DSO imports from libscene.so, defined here with the same
//     mangled names so that the DSO links against libfjscene.so unchanged (Prop*, Property,
//     PlgSetupInfo, Shader, XorShift, MtGetThreadID, Texture::Lookup and the Sl* shading library);
//   * the loader: dlopen + Initialize(PluginInfo *) + validation, as reference src/fj_plugin.cc:28-69.
//
// The GPU build never calls a DSO's evaluate() (the device twin selected by PluginInfo.plugin_name
// shades); the Sl* functions and Texture::Lookup a DSO's evaluate() would call exist so that the DSO
// links, and abort loudly if something does call them: there is no host shading path.
#include "fj_host.h"
#include "fj_plugin_abi.h"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace fj {

// ------------------------------------------------------------ property values
static PropertyValue make_value(int type, Real a, Real b, Real c, Real d)
{
  PropertyValue v;
  v.type = type;
  v.vector = Vector4(a, b, c, d);
  return v;
}
PropertyValue PropNull() { return PropertyValue(); }
PropertyValue PropScalar(Real v0) { return make_value(PROP_SCALAR, v0, 0, 0, 0); }
PropertyValue PropVector2(Real v0, Real v1) { return make_value(PROP_VECTOR2, v0, v1, 0, 0); }
PropertyValue PropVector3(Real v0, Real v1, Real v2) { return make_value(PROP_VECTOR3, v0, v1, v2, 0); }
PropertyValue PropVector4(Real v0, Real v1, Real v2, Real v3) { return make_value(PROP_VECTOR4, v0, v1, v2, v3); }
PropertyValue PropString(const char *string) { PropertyValue v; v.type = PROP_STRING; v.string = string; return v; }
PropertyValue PropObjectGroup(ObjectGroup *p) { PropertyValue v; v.type = PROP_OBJECTGROUP; v.object_group = p; return v; }
PropertyValue PropPointCloud(PointCloud *p) { PropertyValue v; v.type = PROP_POINTCLOUD; v.pointcloud = p; return v; }
PropertyValue PropTurbulence(Turbulence *p) { PropertyValue v; v.type = PROP_TURBULENCE; v.turbulence = p; return v; }
PropertyValue PropTexture(Texture *p) { PropertyValue v; v.type = PROP_TEXTURE; v.texture = p; return v; }
PropertyValue PropVolume(Volume *p) { PropertyValue v; v.type = PROP_VOLUME; v.volume = p; return v; }
PropertyValue PropCurve(Curve *p) { PropertyValue v; v.type = PROP_CURVE; v.curve = p; return v; }
PropertyValue PropMesh(Mesh *p) { PropertyValue v; v.type = PROP_MESH; v.mesh = p; return v; }

// ------------------------------------------------------------ Property (src/fj_property.cc:153-240)
Property::Property() : type_(PROP_NONE), name_(NULL), default_value_(), set_value_fn_(NULL) {}
Property::Property(const char *name, const PropertyValue &value, SetValueFn set_value_fn)
    : type_(value.type), name_(name), default_value_(value.vector), set_value_fn_(set_value_fn) {}
Property::~Property() {}
bool Property::IsValid() const { return type_ != PROP_NONE; }
int Property::GetType() const { return type_; }
const char *Property::GetName() const { return name_; }
const Vector4 &Property::GetDefaultValue() const { return default_value_; }
const char *Property::GetTypeString() const
{
  static const char *names[] = {"none", "scalar", "vector2", "vector3", "vector4", "string", "ObjectGroup", "PointCloud",
      "Turbulence", "Texture", "Shader", "Volume", "Curve", "Mesh"};
  return (type_ >= 0 && type_ <= PROP_MESH) ? names[type_] : "none";
}
int Property::SetValue(void *self, const PropertyValue &value) const
{
  if (self == NULL || set_value_fn_ == NULL) return -1;
  return set_value_fn_(self, value);
}

// first entry of the table with this type and name (src/fj_property.cc:242-257)
const Property *PropFind(const Property *list, int type, const char *name)
{
  if (list == NULL || name == NULL) return NULL;
  for (const Property *p = list; p->IsValid(); p++)
    if (p->GetType() == type && std::strcmp(p->GetName(), name) == 0) return p;
  return NULL;
}

// every vector-typed property of the table set to its default through its own setter; object
// typed ones (texture ...) to null (src/fj_property.cc:259-315)
int PropSetAllDefaultValues(void *self, const Property *list)
{
  int err_count = 0;
  if (list == NULL) return -1;
  for (const Property *p = list; p->IsValid(); p++) {
    PropertyValue v;
    v.type = p->GetType();
    switch (p->GetType()) {
    case PROP_SCALAR: case PROP_VECTOR2: case PROP_VECTOR3: case PROP_VECTOR4:
      v.vector = p->GetDefaultValue();
      break;
    default:
      break;              // string / object properties default to null
    }
    if (p->SetValue(self, v)) err_count++;
  }
  return err_count ? -1 : 0;
}

// ------------------------------------------------------------ plugin description
static int plg_errno = PLG_ERR_NONE;

static bool valid_info(const PluginInfo *info)
{
  return info->api_version == PLUGIN_API_VERSION && info->plugin_type && info->plugin_name && info->create_instance &&
      info->delete_instance && info->property_list && info->meta;
}

int PlgSetupInfo(PluginInfo *info, int api_version, const char *plugin_type, const char *plugin_name,
    PlgCreateInstanceFn create_instance, PlgDeleteInstanceFn delete_instance, const Property *property_list, const MetaInfo *meta)
{
  info->api_version = api_version;
  info->plugin_type = plugin_type;
  info->plugin_name = plugin_name;
  info->create_instance = create_instance;
  info->delete_instance = delete_instance;
  info->property_list = property_list;
  info->meta = meta;
  return valid_info(info) ? 0 : -1;
}

int PlgGetErrorNo(void) { return plg_errno; }

// ------------------------------------------------------------ Shader / XorShift / thread id
Shader::Shader() {}
Shader::~Shader() {}
void Shader::Evaluate(const TraceContext &cxt, const SurfaceInput &in, SurfaceOutput *out) const { evaluate(cxt, in, out); }

XorShift::XorShift() { state[0] = 123456789; state[1] = 362436069; state[2] = 521288629; state[3] = 88675123; }   // src/fj_random.cc:10-16
XorShift::XorShift(unsigned int seed)
{
  for (uint32_t i = 0; i < 4; i++) state[i] = seed = 1812433253U * (seed ^ (seed >> 30)) + i;                    // :18-24
}
uint32_t XorShift::NextInteger()                                                                                    // :26-38
{
  const uint32_t t = state[0] ^ (state[0] << 11);
  state[0] = state[1]; state[1] = state[2]; state[2] = state[3];
  state[3] = (state[3] ^ (state[3] >> 19)) ^ (t ^ (t >> 8));
  return state[3];
}
Real XorShift::NextFloat01() { return static_cast<Real>(NextInteger()) / 4294967295u; }                           // :40-43

int MtGetThreadID() { return 0; }

// ------------------------------------------------------------ host shading library: not in this build
[[noreturn]] static void no_host_shading(const char *what)
{
  std::fprintf(stderr, "libfjscene: %s called: the GPU build has no host shading path -- shader plugins run as device "
      "code selected by PluginInfo.plugin_name, their evaluate() is never invoked\n", what);
  std::abort();
}
Color4 Texture::Lookup(float, float) const { no_host_shading("Texture::Lookup"); }
void SlFaceforward(const Vector *, const Vector *, Vector *) { no_host_shading("SlFaceforward"); }
double SlFresnel(const Vector *, const Vector *, double) { no_host_shading("SlFresnel"); }
double SlPhong(const Vector *, const Vector *, const Vector *, double) { no_host_shading("SlPhong"); }
void SlReflect(const Vector *, const Vector *, Vector *) { no_host_shading("SlReflect"); }
void SlRefract(const Vector *, const Vector *, double, Vector *) { no_host_shading("SlRefract"); }
int SlTrace(const TraceContext *, const Vector *, const Vector *, double, double, Color4 *, double *) { no_host_shading("SlTrace"); }
TraceContext SlDiffuseContext(const TraceContext *, const ObjectInstance *) { no_host_shading("SlDiffuseContext"); }
TraceContext SlReflectContext(const TraceContext *, const ObjectInstance *) { no_host_shading("SlReflectContext"); }
TraceContext SlRefractContext(const TraceContext *, const ObjectInstance *) { no_host_shading("SlRefractContext"); }
TraceContext SlShadowContext(const TraceContext *, const ObjectInstance *) { no_host_shading("SlShadowContext"); }
int SlIlluminance(const TraceContext *, const LightSample *, const Vector *, const Vector *, double, const SurfaceInput *, LightOutput *)
{
  no_host_shading("SlIlluminance");
}
int SlGetLightCount(const SurfaceInput *) { no_host_shading("SlGetLightCount"); }
int SlGetLightSampleCount(const SurfaceInput *) { no_host_shading("SlGetLightSampleCount"); }
LightSample *SlNewLightSamples(const SurfaceInput *) { no_host_shading("SlNewLightSamples"); }
void SlFreeLightSamples(LightSample *) { no_host_shading("SlFreeLightSamples"); }
void SlBumpMapping(const Texture *, const Vector *, const Vector *, const TexCoord *, double, const Vector *, Vector *)
{
  no_host_shading("SlBumpMapping");
}

}  // namespace fj

// ================================================================= loader
namespace fjhost {

// dlopen(name + ".so" unless it ends in ".so") like OsDlopen (src/internal/fj_os_unix.cc:10-24),
// Initialize(PluginInfo *), validation: Plugin::Open, src/fj_plugin.cc:28-69.  Returns 0 and fills
// *out, or the PlgErrorNo of the failing step.
int OpenPluginDso(const char *filename, LoadedPlugin *out)
{
  fj::plg_errno = fj::PLG_ERR_NONE;
  std::string path(filename);
  if (path.size() < 3 || path.compare(path.size() - 3, 3, ".so") != 0) path += ".so";
  void *dso = dlopen(path.c_str(), RTLD_LAZY);
  if (!dso) return fj::plg_errno = fj::PLG_ERR_PLUGIN_NOT_FOUND;
  fj::PlgInitializeFn init = reinterpret_cast<fj::PlgInitializeFn>(dlsym(dso, "Initialize"));
  if (!init) { dlclose(dso); return fj::plg_errno = fj::PLG_ERR_INIT_PLUGIN_FUNC_NOT_EXIST; }
  fj::PluginInfo info;
  if (init(&info)) { dlclose(dso); return fj::plg_errno = fj::PLG_ERR_INIT_PLUGIN_FUNC_FAIL; }
  if (!fj::valid_info(&info)) { dlclose(dso); return fj::plg_errno = fj::PLG_ERR_BAD_PLUGIN_INFO; }
  out->dso = dso;
  out->info = info;
  return 0;
}

void ClosePluginDso(LoadedPlugin *p)
{
  if (!p || !p->dso) return;
  for (void *inst : p->instances) p->info.delete_instance(inst);     // Plugin::Close, src/fj_plugin.cc:71-77
  p->instances.clear();
  dlclose(p->dso);
  p->dso = nullptr;
}

}  // namespace fjhost
