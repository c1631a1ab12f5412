/* fj_plugin_abi.h -- the Shader plugin ABI of libfjscene.so.
 *
 * The reference loads a shader as a DSO: SiOpenPlugin dlopen()s it, calls its
 *     extern "C" int Initialize(fj::PluginInfo *)
 * and from then on talks to it through PluginInfo (reference src/fj_plugin.h:34-54,
 * src/fj_plugin.cc:28-69): plugin type and name, create / delete instance, and a
 * Property table whose entries carry a name, a typed default and a setter
 * (src/fj_property.h:39-111).  libfjscene.so speaks the same protocol, so a shader DSO
 * built for the reference loads unchanged: this header declares -- with the reference's
 * names, signatures and object layouts, because those ARE the binary interface -- every
 * type and function such a DSO is compiled against (what `#include "fj_shader.h"` gives
 * it there), and libfjscene.so exports every symbol it imports
 * (`nm -D --undefined-only PlasticShader.so`).
 *
 * What the GPU build does with a loaded shader DSO: PluginInfo.plugin_name selects the
 * device implementation (PlasticShader, ConstantShader, GlassShader, HairShader,
 * PathtracingShader run as HIP code); the DSO's own Property table supplies the property
 * names, types and defaults; its setters are called as the reference calls them.  Its
 * evaluate() is never called: there is no host shading path.  The Sl* functions a DSO
 * imports for evaluate() therefore exist only so that the DSO links; called, they report
 * the error and abort -- a shader without a device twin is refused at SiOpenPlugin.
 *
 * Stated limit (SURVEY 8b "How parameters reach the GPU" asks that ANY shader DSO still load and run on a host path
 * whose Sl* exports are the CPU restatement): this build has NO host shading path by design -- the product must fail
 * loudly where the HIP path cannot serve, and the CPU restatement is test infrastructure (oracle/) that nothing under
 * fujiyama-renderer_amd/ may link.  So a foreign shader DSO (the reference tree's own examples: MaterialShader,
 * SSSShader, VolumeShader -- all SURVEY section 2 out of scope) passes the whole loading protocol -- dlopen, Initialize,
 * PluginInfo validation with the reference's PlgErrorNo -> SiErrorNo mapping -- and is then refused by SiOpenPlugin with
 * SI_ERR_FAILLOAD and a message naming the plugin, instead of rendering on the CPU.  A maintainer who wants such a
 * shader on the device adds its evaluate() to fjgpu_dev_shade.h's switch and its name to kKnownPlugins
 * (csrc/host/fj_host_scene.cc); tests/test_host_boundary.py pins both behaviours.
 */
#ifndef FJ_PLUGIN_ABI_H
#define FJ_PLUGIN_ABI_H

#ifdef __cplusplus
#include <cmath>
#include <cstddef>
#include <stdint.h>

#define PLUGIN_API_VERSION 1                 /* src/fj_plugin.h:12 */
#define SHADER_PLUGIN_TYPE "Shader"          /* src/fj_shader.h:19 */
#define FJ_PLUGIN_API __attribute__((visibility("default")))
#define FJ_API __attribute__((visibility("default")))

namespace fj {

typedef double Real;
#define REAL_MAX 1.7976931348623157e+308     /* src/fj_types.h: DBL_MAX */
const Real PI = 3.14159265358979323846;

/* ---- scalars (src/fj_numeric.h) */
inline Real Abs(Real x) { return x < 0 ? -x : x; }
inline Real Sqrt(Real x) { return std::sqrt(x); }
inline Real Pow(Real x, Real e) { return std::pow(x, e); }
inline Real Min(Real a, Real b) { return a < b ? a : b; }
inline Real Max(Real a, Real b) { return a > b ? a : b; }
inline Real Clamp(Real x, Real lo, Real hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* The small value types below declare an (empty) destructor like the reference's: under the
 * Itanium C++ ABI that makes them non-trivial for calls -- returned through a hidden pointer
 * (Texture::Lookup's Color4) -- so it is part of the binary interface. */
/* ---- Vector: 3 x f64 (src/fj_vector.h) */
struct Vector {
  Real x, y, z;
  Vector() : x(0), y(0), z(0) {}
  Vector(Real xx, Real yy, Real zz) : x(xx), y(yy), z(zz) {}
  ~Vector() {}
  Real operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  Real &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
  const Vector &operator+=(const Vector &a) { x += a.x; y += a.y; z += a.z; return *this; }
  const Vector &operator-=(const Vector &a) { x -= a.x; y -= a.y; z -= a.z; return *this; }
  const Vector &operator*=(Real s) { x *= s; y *= s; z *= s; return *this; }
  const Vector &operator/=(Real s) { const Real inv = 1. / s; x *= inv; y *= inv; z *= inv; return *this; }
};
inline Vector operator+(const Vector &a, const Vector &b) { return Vector(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vector operator-(const Vector &a, const Vector &b) { return Vector(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vector operator-(const Vector &a) { return Vector(-a.x, -a.y, -a.z); }
inline Vector operator*(const Vector &a, Real s) { return Vector(a.x * s, a.y * s, a.z * s); }
inline Vector operator*(Real s, const Vector &a) { return a * s; }
inline Vector operator/(const Vector &a, Real s) { const Real inv = 1. / s; return Vector(a.x * inv, a.y * inv, a.z * inv); }
inline Real Dot(const Vector &a, const Vector &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vector Cross(const Vector &a, const Vector &b) { return Vector(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline Real Length(const Vector &a) { return Sqrt(Dot(a, a)); }
inline Vector Normalize(const Vector &a) { const Real len = Length(a); if (len == 0) return a; return a * (1. / len); }

struct Vector4 {
  Real x, y, z, w;
  Vector4() : x(0), y(0), z(0), w(0) {}
  Vector4(Real xx, Real yy, Real zz, Real ww) : x(xx), y(yy), z(zz), w(ww) {}
  ~Vector4() {}
  Real operator[](int i) const { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : w)); }
};

/* ---- Color: 3 x f32, Color4: 4 x f32 (src/fj_color.h) */
struct Color {
  float r, g, b;
  Color() : r(0), g(0), b(0) {}
  Color(float rr, float gg, float bb) : r(rr), g(gg), b(bb) {}
  ~Color() {}
  float operator[](int i) const { return i == 0 ? r : (i == 1 ? g : b); }
  float &operator[](int i) { return i == 0 ? r : (i == 1 ? g : b); }
  const Color &operator+=(const Color &a) { r += a.r; g += a.g; b += a.b; return *this; }
  const Color &operator-=(const Color &a) { r -= a.r; g -= a.g; b -= a.b; return *this; }
  const Color &operator*=(const Color &a) { r *= a.r; g *= a.g; b *= a.b; return *this; }
  const Color &operator*=(float s) { r *= s; g *= s; b *= s; return *this; }
  const Color &operator/=(float s) { const float inv = 1.f / s; r *= inv; g *= inv; b *= inv; return *this; }
};
inline Color operator+(const Color &a, const Color &b) { return Color(a.r + b.r, a.g + b.g, a.b + b.b); }
inline Color operator-(const Color &a, const Color &b) { return Color(a.r - b.r, a.g - b.g, a.b - b.b); }
inline Color operator*(const Color &a, const Color &b) { return Color(a.r * b.r, a.g * b.g, a.b * b.b); }
inline Color operator*(const Color &a, float s) { return Color(a.r * s, a.g * s, a.b * s); }
inline Color operator*(float s, const Color &a) { return a * s; }
inline Color operator/(const Color &a, float s) { const float inv = 1.f / s; return Color(a.r * inv, a.g * inv, a.b * inv); }
inline Color operator-(const Color &a) { return Color(-a.r, -a.g, -a.b); }
inline float Luminance(const Color &a) { return .298912 * a.r + .586611 * a.g + .114478 * a.b; }

struct Color4 {
  float r, g, b, a;
  Color4() : r(0), g(0), b(0), a(0) {}
  Color4(float rr, float gg, float bb, float aa) : r(rr), g(gg), b(bb), a(aa) {}
  ~Color4() {}
  float operator[](int i) const { return i == 0 ? r : (i == 1 ? g : (i == 2 ? b : a)); }
};
inline float Luminance4(const Color4 &c) { return .298912 * c.r + .586611 * c.g + .114478 * c.b; }
inline Color ToColor(const Color4 &c) { return Color(c.r, c.g, c.b); }
inline Color4 ToColor4(const Color &c, float alpha = 1.f) { return Color4(c.r, c.g, c.b, alpha); }

struct TexCoord {
  float u, v;
  TexCoord() : u(0), v(0) {}
  TexCoord(float uu, float vv) : u(uu), v(vv) {}
  ~TexCoord() {}
};

/* ---- objects a shader only holds pointers to */
class ObjectGroup;
class ObjectInstance;
class PointCloud;
class Turbulence;
class Volume;
class Curve;
class Mesh;
class Light;
class Shader;

/* Texture::Lookup(u, v): nearest texel of the tiled .mip file, (1, .63, .63, 1) without one
 * (src/fj_texture.cc:111-121).  The object behind the pointer is libfjscene's. */
class FJ_API Texture {
public:
  Color4 Lookup(float u, float v) const;
private:
  Texture();
};

/* ---- properties (src/fj_property.h:13-111): layouts are ABI, 120 B / 56 B */
enum PropertyType {
  PROP_NONE = 0, PROP_SCALAR, PROP_VECTOR2, PROP_VECTOR3, PROP_VECTOR4, PROP_STRING, PROP_OBJECTGROUP,
  PROP_POINTCLOUD, PROP_TURBULENCE, PROP_TEXTURE, PROP_SHADER, PROP_VOLUME, PROP_CURVE, PROP_MESH
};

class FJ_API PropertyValue {
public:
  PropertyValue() : type(PROP_NONE), vector(), string(NULL), object_group(NULL), pointcloud(NULL), turbulence(NULL),
      texture(NULL), shader(NULL), volume(NULL), curve(NULL), mesh(NULL), time(0) {}
  ~PropertyValue() {}
  int type;
  Vector4 vector;
  const char *string;
  ObjectGroup *object_group;
  PointCloud *pointcloud;
  Turbulence *turbulence;
  Texture *texture;
  Shader *shader;
  Volume *volume;
  Curve *curve;
  Mesh *mesh;
  Real time;
};

FJ_API PropertyValue PropNull();
FJ_API PropertyValue PropScalar(Real v0);
FJ_API PropertyValue PropVector2(Real v0, Real v1);
FJ_API PropertyValue PropVector3(Real v0, Real v1, Real v2);
FJ_API PropertyValue PropVector4(Real v0, Real v1, Real v2, Real v3);
FJ_API PropertyValue PropString(const char *string);
FJ_API PropertyValue PropObjectGroup(ObjectGroup *group);
FJ_API PropertyValue PropPointCloud(PointCloud *pointcloud);
FJ_API PropertyValue PropTurbulence(Turbulence *turbulence);
FJ_API PropertyValue PropTexture(Texture *texture);
FJ_API PropertyValue PropVolume(Volume *volume);
FJ_API PropertyValue PropCurve(Curve *curve);
FJ_API PropertyValue PropMesh(Mesh *mesh);

class FJ_API Property {
public:
  typedef int (*SetValueFn)(void *self, const PropertyValue &value);
  Property();                                                      /* the terminator of a property table: PROP_NONE */
  Property(const char *name, const PropertyValue &value, SetValueFn set_value_fn);
  ~Property();
  bool IsValid() const;
  int GetType() const;
  const char *GetName() const;
  const Vector4 &GetDefaultValue() const;
  const char *GetTypeString() const;
  int SetValue(void *self, const PropertyValue &value) const;
private:
  int type_;
  const char *name_;
  Vector4 default_value_;
  SetValueFn set_value_fn_;
};

FJ_API const Property *PropFind(const Property *list, int type, const char *name);
FJ_API int PropSetAllDefaultValues(void *self, const Property *list);

/* ---- plugin description (src/fj_plugin.h:18-67), 56 B */
class PluginInfo;
typedef int (*PlgInitializeFn)(PluginInfo *info);
typedef void *(*PlgCreateInstanceFn)(void);
typedef void (*PlgDeleteInstanceFn)(void *obj);

enum PlgErrorNo {
  PLG_ERR_NONE = 0, PLG_ERR_PLUGIN_NOT_FOUND, PLG_ERR_INIT_PLUGIN_FUNC_NOT_EXIST, PLG_ERR_INIT_PLUGIN_FUNC_FAIL,
  PLG_ERR_BAD_PLUGIN_INFO, PLG_ERR_CLOSE_PLUGIN_FAIL, PLG_ERR_NO_MEMORY
};

class MetaInfo {
public:
  const char *name;
  const char *data;
};

class PluginInfo {
public:
  PluginInfo() : api_version(0), plugin_type(NULL), plugin_name(NULL), create_instance(NULL), delete_instance(NULL),
      property_list(NULL), meta(NULL) {}
  ~PluginInfo() {}
  int api_version;
  const char *plugin_type;
  const char *plugin_name;
  PlgCreateInstanceFn create_instance;
  PlgDeleteInstanceFn delete_instance;
  const Property *property_list;
  const MetaInfo *meta;
};

/* fills *info and validates it (api version, no null member): 0 or -1 (src/fj_plugin.cc:129-149) */
FJ_API int PlgSetupInfo(PluginInfo *info, int api_version, const char *plugin_type, const char *plugin_name,
    PlgCreateInstanceFn create_instance, PlgDeleteInstanceFn delete_instance, const Property *property_list,
    const MetaInfo *meta);
FJ_API int PlgGetErrorNo(void);

/* ---- shading state handed to evaluate() (src/fj_shading.h:18-80): 96 / 176 / 16 / 56 B */
enum RayContext { CXT_CAMERA_RAY = 0, CXT_SHADOW_RAY, CXT_DIFFUSE_RAY, CXT_REFLECT_RAY, CXT_REFRACT_RAY };

class FJ_API TraceContext {
public:
  int ray_context;
  int diffuse_depth, reflect_depth, refract_depth;
  int max_diffuse_depth, max_reflect_depth, max_refract_depth;
  int cast_shadow;
  double time;
  float opacity_threshold;
  double raymarch_step, raymarch_shadow_step, raymarch_diffuse_step, raymarch_reflect_step, raymarch_refract_step;
  const ObjectGroup *trace_target;
};

class FJ_API SurfaceInput {
public:
  Vector P, N;
  Color Cd;
  TexCoord uv;
  float Alpha;
  Vector Ng, I;
  Vector dPdu, dPdv;
  const ObjectInstance *shaded_object;
};

class FJ_API SurfaceOutput {
public:
  Color Cs;
  float Os;
};

class FJ_API LightOutput {
public:
  Color Cl, Ol;
  Vector Ln;
  double distance;
};

class FJ_API LightSample {              /* src/fj_light.h:20-29, 72 B */
public:
  const Light *light;
  Vector P, N;
  Color color;
};

/* host-side shading library a DSO's evaluate() links against (src/fj_shading.h:82-137).  The GPU
 * build never calls evaluate(); these abort with a message (see the header comment). */
FJ_API void SlFaceforward(const Vector *I, const Vector *N, Vector *Nf);
FJ_API double SlFresnel(const Vector *I, const Vector *N, double ior);
FJ_API double SlPhong(const Vector *I, const Vector *N, const Vector *L, double roughness);
FJ_API void SlReflect(const Vector *I, const Vector *N, Vector *R);
FJ_API void SlRefract(const Vector *I, const Vector *N, double ior, Vector *T);
FJ_API int SlTrace(const TraceContext *cxt, const Vector *ray_orig, const Vector *ray_dir, double ray_tmin, double ray_tmax,
    Color4 *out_color, double *t_hit);
FJ_API TraceContext SlDiffuseContext(const TraceContext *cxt, const ObjectInstance *obj);
FJ_API TraceContext SlReflectContext(const TraceContext *cxt, const ObjectInstance *obj);
FJ_API TraceContext SlRefractContext(const TraceContext *cxt, const ObjectInstance *obj);
FJ_API TraceContext SlShadowContext(const TraceContext *cxt, const ObjectInstance *obj);
FJ_API int SlIlluminance(const TraceContext *cxt, const LightSample *sample, const Vector *Ps, const Vector *axis, double angle,
    const SurfaceInput *in, LightOutput *out);
FJ_API int SlGetLightCount(const SurfaceInput *in);
FJ_API int SlGetLightSampleCount(const SurfaceInput *in);
FJ_API LightSample *SlNewLightSamples(const SurfaceInput *in);
FJ_API void SlFreeLightSamples(LightSample *samples);
FJ_API void SlBumpMapping(const Texture *bump_map, const Vector *dPdu, const Vector *dPdv, const TexCoord *texcoord,
    double amplitude, const Vector *N, Vector *N_bump);

/* src/fj_random.h: the xorshift128 generator (state 123456789, 362436069, 521288629, 88675123) */
class FJ_API XorShift {
public:
  XorShift();
  XorShift(unsigned int seed);
  ~XorShift() {}
  uint32_t NextInteger();
  Real NextFloat01();
  uint32_t state[4];
};

FJ_API int MtGetThreadID();                  /* src/fj_multi_thread.h: 0 here (no host workers shade) */

/* ---- the plugin's instance type (src/fj_shader.h:23-34): vtable = {dtor, dtor, evaluate} */
class FJ_API Shader {
public:
  Shader();
  virtual ~Shader();
  void Evaluate(const TraceContext &cxt, const SurfaceInput &in, SurfaceOutput *out) const;
private:
  virtual void evaluate(const TraceContext &cxt, const SurfaceInput &in, SurfaceOutput *out) const = 0;
};

}  /* namespace fj */
#endif /* __cplusplus */
#endif /* FJ_PLUGIN_ABI_H */
