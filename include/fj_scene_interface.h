/* fj_scene_interface.h -- host-side drop-in boundary of libfjscene.so.
 *
 * Two spellings of the same 41-function scene API:
 *
 *   1. C++ linkage, `namespace fj`, the names / argument meaning / return
 *      conventions of the reference's public scene interface
 *      (reference src/fj_scene_interface.h:64-127): ID = long handles encoded
 *      as type * 10^7 + index (src/fj_scene_interface.cc:44,1058-1075),
 *      Status 0 / -1, SI_BADID -1, a global error number read by
 *      SiGetErrorNo().  A C++ host written against the reference (e.g. its
 *      scenes/cube.cc, tools/scene_parser) recompiles against this header.
 *
 *   2. extern "C" `fj_Si*` with plain C types: the thin FFI any other
 *      language binds (ctypes stub in INTEGRATION.md), plus a few helpers to
 *      run scene-description text and read back the float framebuffer.
 *
 * Behind SiRenderScene() the frame is rendered by the HIP core of
 * include/fjgpu.h; there is no CPU fallback.
 */
#ifndef FJ_SCENE_INTERFACE_H
#define FJ_SCENE_INTERFACE_H

#include <stddef.h>
#include <stdint.h>
#include "fj_scene_desc.h"

#ifdef __cplusplus
#include <vector>
#include "fj_plugin_abi.h"       /* Property (SiGetPropertyList), the Shader plugin ABI */

namespace fj {

typedef long int ID;
typedef int Status;
enum { SI_BADID = -1 };
enum { SI_FAIL = -1, SI_SUCCESS = 0 };

enum SiErrorNo {
  SI_ERR_NONE = 0,
  SI_ERR_NO_MEMORY,
  SI_ERR_BADTYPE,
  SI_ERR_FAILLOAD,
  SI_ERR_FAILNEW,
  SI_ERR_PLUGIN_NOT_FOUND,
  SI_ERR_INIT_PLUGIN_FUNC_NOT_EXIST,
  SI_ERR_INIT_PLUGIN_FUNC_FAIL,
  SI_ERR_BAD_PLUGIN_INFO,
  SI_ERR_CLOSE_PLUGIN_FAIL,
  SI_ERR_UNDEFINED
};

enum SiTransformOrder {
  SI_ORDER_SRT = 0, SI_ORDER_STR, SI_ORDER_RST, SI_ORDER_RTS, SI_ORDER_TRS, SI_ORDER_TSR,
  SI_ORDER_XYZ, SI_ORDER_XZY, SI_ORDER_YXZ, SI_ORDER_YZX, SI_ORDER_ZXY, SI_ORDER_ZYX
};

enum SiLightType { SI_POINT_LIGHT = 0, SI_GRID_LIGHT, SI_SPHERE_LIGHT, SI_DOME_LIGHT };
enum SiSamplerType { SI_FIXED_GRID_SAMPLER = 0, SI_ADAPTIVE_GRID_SAMPLER = 1 };

/* FrameBuffer: W x H x C interleaved float (reference src/fj_framebuffer.h) */
class FrameBuffer {
public:
  FrameBuffer() : width_(0), height_(0), nchannels_(0) {}
  int GetWidth() const { return width_; }
  int GetHeight() const { return height_; }
  int GetChannelCount() const { return nchannels_; }
  int GetSize() const { return width_ * height_ * nchannels_; }
  void Resize(int w, int h, int c) { width_ = w; height_ = h; nchannels_ = c; buf_.assign((size_t) w * h * c, 0.f); }
  bool IsEmpty() const { return buf_.empty(); }
  float *GetWritable(int x, int y, int z) { return inside(x, y, z) ? &buf_[index(x, y, z)] : 0; }
  const float *GetReadOnly(int x, int y, int z) const { return inside(x, y, z) ? &buf_[index(x, y, z)] : 0; }
private:
  size_t index(int x, int y, int z) const { return ((size_t) y * width_ + x) * nchannels_ + z; }
  bool inside(int x, int y, int z) const { return x >= 0 && x < width_ && y >= 0 && y < height_ && z >= 0 && z < nchannels_; }
  std::vector<float> buf_;
  int width_, height_, nchannels_;
};

struct Int2 { int x, y; int operator[](int i) const { return i ? y : x; } };
struct Rectangle { Int2 min, max; };

/* render callbacks (reference src/fj_callback.h:15-98) */
struct FrameInfo {
  int32_t frame_id;
  int worker_count;
  int tile_count;
  int xres, yres;
  Rectangle frame_region;
  const FrameBuffer *framebuffer;
};
struct TileInfo {
  int32_t frame_id;
  int worker_id;
  int region_id;
  int total_region_count;
  Rectangle tile_region;
  const FrameBuffer *framebuffer;
};
enum { CALLBACK_CONTINUE = 0, CALLBACK_INTERRUPT = -1 };
typedef int Interrupt;
typedef Interrupt (*FrameStartCallback)(void *data, const FrameInfo *info);
typedef Interrupt (*FrameAbortCallback)(void *data, const FrameInfo *info);
typedef Interrupt (*FrameDoneCallback)(void *data, const FrameInfo *info);
typedef Interrupt (*TileStartCallback)(void *data, const TileInfo *info);
typedef Interrupt (*TileDoneCallback)(void *data, const TileInfo *info);
typedef Interrupt (*SampleDoneCallback)(void *data);

int SiGetErrorNo(void);

ID SiOpenPlugin(const char *filename);

Status SiOpenScene(void);
Status SiCloseScene(void);
Status SiRenderScene(ID renderer);
Status SiSaveFrameBuffer(ID framebuffer, const char *filename);
Status SiRunProcedure(ID procedure);

Status SiAddObjectToGroup(ID group, ID object);

ID SiNewObjectInstance(ID primset);
ID SiNewFrameBuffer(const char *arg);
ID SiNewObjectGroup(void);
ID SiNewPointCloud(void);
ID SiNewTurbulence(void);
ID SiNewProcedure(ID plugin);
ID SiNewRenderer(void);
ID SiNewTexture(const char *filename);
ID SiNewCamera(const char *arg);
ID SiNewShader(ID plugin);
ID SiNewVolume(void);
ID SiNewCurve(void);
ID SiNewLight(int light_type);
ID SiNewMesh(void);

Status SiAssignFrameBuffer(ID renderer, ID framebuffer);
Status SiAssignObjectGroup(ID id, const char *name, ID group);
Status SiAssignPointCloud(ID id, const char *name, ID pointcloud);
Status SiAssignTurbulence(ID id, const char *name, ID turbulence);
Status SiAssignTexture(ID id, const char *name, ID texture);
Status SiAssignVolume(ID id, const char *name, ID volume);
Status SiAssignCamera(ID renderer, ID camera);
Status SiAssignShader(ID object, const char *shading_group, ID shader);
Status SiAssignCurve(ID id, const char *name, ID curve);
Status SiAssignMesh(ID id, const char *name, ID mesh);

Status SiSetProperty1(ID id, const char *name, double v0);
Status SiSetProperty2(ID id, const char *name, double v0, double v1);
Status SiSetProperty3(ID id, const char *name, double v0, double v1, double v2);
Status SiSetProperty4(ID id, const char *name, double v0, double v1, double v2, double v3);
Status SiSetStringProperty(ID id, const char *name, const char *string);
Status SiSetSampleProperty3(ID id, const char *name, double v0, double v1, double v2, double time);

/* property table of a built-in type ("Renderer", "ObjectInstance", "Camera", "Light") or of an
 * opened plugin by its plugin name ("PlasticShader" ...), terminated by an invalid Property
 * (reference src/fj_scene_interface.cc:1047-1051); NULL for an unknown name */
const Property *SiGetPropertyList(const char *type_name);

Status SiSetFrameReportCallback(ID id, void *data,
    FrameStartCallback frame_start, FrameAbortCallback frame_abort, FrameDoneCallback frame_done);
Status SiSetTileReportCallback(ID id, void *data,
    TileStartCallback tile_start, SampleDoneCallback sample_done, TileDoneCallback tile_done);

}  /* namespace fj */
#endif /* __cplusplus */

#ifdef __cplusplus
extern "C" {
#endif

/* ---- thin C FFI: one entry per Si* function, same argument meaning ---- */
int  fj_SiGetErrorNo(void);
long fj_SiOpenPlugin(const char *filename);
int  fj_SiOpenScene(void);
int  fj_SiCloseScene(void);
int  fj_SiRenderScene(long renderer);
int  fj_SiSaveFrameBuffer(long framebuffer, const char *filename);
int  fj_SiRunProcedure(long procedure);
int  fj_SiAddObjectToGroup(long group, long object);
long fj_SiNewObjectInstance(long primset);
long fj_SiNewFrameBuffer(const char *arg);
long fj_SiNewObjectGroup(void);
long fj_SiNewPointCloud(void);
long fj_SiNewTurbulence(void);
long fj_SiNewProcedure(long plugin);
long fj_SiNewRenderer(void);
long fj_SiNewTexture(const char *filename);
long fj_SiNewCamera(const char *arg);
long fj_SiNewShader(long plugin);
long fj_SiNewVolume(void);
long fj_SiNewCurve(void);
long fj_SiNewLight(int light_type);
long fj_SiNewMesh(void);
int  fj_SiAssignFrameBuffer(long renderer, long framebuffer);
int  fj_SiAssignObjectGroup(long id, const char *name, long group);
int  fj_SiAssignPointCloud(long id, const char *name, long pointcloud);
int  fj_SiAssignTurbulence(long id, const char *name, long turbulence);
int  fj_SiAssignTexture(long id, const char *name, long texture);
int  fj_SiAssignVolume(long id, const char *name, long volume);
int  fj_SiAssignCamera(long renderer, long camera);
int  fj_SiAssignShader(long object, const char *shading_group, long shader);
int  fj_SiAssignCurve(long id, const char *name, long curve);
int  fj_SiAssignMesh(long id, const char *name, long mesh);
int  fj_SiSetProperty1(long id, const char *name, double v0);
int  fj_SiSetProperty2(long id, const char *name, double v0, double v1);
int  fj_SiSetProperty3(long id, const char *name, double v0, double v1, double v2);
int  fj_SiSetProperty4(long id, const char *name, double v0, double v1, double v2, double v3);
int  fj_SiSetStringProperty(long id, const char *name, const char *string);
int  fj_SiSetSampleProperty3(long id, const char *name, double v0, double v1, double v2, double time);
/* SiGetPropertyList (reference src/fj_scene_interface.h:116): the table as an opaque pointer, NULL for an unknown name; C has no Property
 * class, so entry k (0, 1, ... up to the first invalid one, the table's terminator) is read through the accessors */
const void *fj_SiGetPropertyList(const char *type_name);
int  fj_property_is_valid(const void *table, int k);             /* 0: the terminator */
const char *fj_property_name(const void *table, int k);
const char *fj_property_type_string(const void *table, int k);   /* Property::GetTypeString */
int  fj_property_default(const void *table, int k, double out4[4]);    /* Property::GetDefaultValue: 0 or -1 */
/* callbacks: (void *data, const FrameInfo * / const TileInfo *) -> CALLBACK_CONTINUE (0) or
 * CALLBACK_INTERRUPT (-1); layouts of src/fj_callback.h:15-98; NULL = no hook */
typedef int (*fj_frame_callback)(void *data, const void *frame_info);
typedef int (*fj_tile_callback)(void *data, const void *tile_info);
typedef int (*fj_sample_callback)(void *data);
int  fj_SiSetFrameReportCallback(long renderer, void *data, fj_frame_callback frame_start, fj_frame_callback frame_abort,
    fj_frame_callback frame_done);
int  fj_SiSetTileReportCallback(long renderer, void *data, fj_tile_callback tile_start, fj_sample_callback sample_done,
    fj_tile_callback tile_done);

/* ---- helpers on top of the Si API ---- */

/* Feed scene-description text (the `scene` command language)
 * line by line to the built-in parser; stops at the first failing command like
 * the reference's bin/scene (tools/scene_parser/main.cc:34-43).
 * Returns 0, or the 1-based number of the failing line (message via
 * fj_scene_last_error()).  `echo` != 0 prints the "-- Name: [arg]" echo. */
int fj_scene_run_text(const char *text, int echo);
const char *fj_scene_last_error(void);

/* When set, `RenderScene` only prepares the scene (bounds, implicit groups,
 * dome-light samples) and does NOT render: the caller fetches the flat
 * description with fj_scene_get_desc() and drives include/fjgpu.h itself
 * (tests, bench.py and the multi-GPU tile sharding do this). */
void fj_scene_set_deferred_render(int on);

/* Flat description of the current scene as of the last RenderScene; pointers
 * stay valid until the next RenderScene / SiCloseScene. Returns 0 on success. */
int fj_scene_get_desc(const fj_scene_desc **scene, const fj_render_desc **render);

/* SiGetPropertyList(type_name) as text, one "type name v0 v1 v2 v3" line per property (the C
 * spelling of the C++ Property table): bytes written, or -1 when no such table exists */
int fj_scene_property_table(const char *type_name, char *out, int out_size);

/* The .fb writer of SiSaveFrameBuffer (plain-text PTO, reference src/fj_framebuffer_io.cc:46-68) on a
 * caller's pixel array [height][width][nchannels] float32: 0 or -1 */
int fj_write_fb_file(const char *filename, int width, int height, int nchannels, const float *pixels);

/* Radiance .hdr (RGBE, flat or run-length encoded) -> the tiled .mip a Texture reads: the reference's
 * tools/hdr2mip (resampled to powers of two, tiles of 64; src/fj_mipmap.cc:246-300).  0 or -1. */
int fj_hdr2mip(const char *hdr_path, const char *mip_path);

/* Diagnostics: the first n values of the rand() stream CurveGeneratorProcedure draws after srand(seed) -- glibc's generator computed by the
 * library itself (thread-safe, the same on any C library; the reference's fur depends on it: procedures/curve_generator_procedure/
 * curve_generator_procedure.cc:171-186,209-215) */
void fj_dev_seeded_rand(uint32_t seed, int n, uint32_t *out);

/* Float framebuffer of a FrameBuffer ID: returns pointer (W*H*C floats) or NULL */
const float *fj_framebuffer_data(long framebuffer, int *width, int *height, int *nchannels);

/* statistics of the last SiRenderScene that rendered on the GPU */
typedef struct fj_render_stats {
  double render_seconds;     /* frame start -> frame done (framebuffer on host) */
  double prepare_seconds;    /* bounds + groups + flatten + BVH build + upload */
  fj_ray_counts rays;
} fj_render_stats;
int fj_scene_last_stats(fj_render_stats *out);

#ifdef __cplusplus
}
#endif
#endif /* FJ_SCENE_INTERFACE_H */
