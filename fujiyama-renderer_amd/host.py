"""ctypes face of lib/libfjscene.so (include/fj_scene_interface.h)."""
import ctypes as C

import numpy as np

from . import ffi

_lib = None

TYPE_ID_OFFSET = 10000000
TYPE_FRAMEBUFFER = 3
TYPE_RENDERER = 8


class FrameInfo(C.Structure):          # fj::FrameInfo, src/fj_callback.h:15-37
    _fields_ = [("frame_id", C.c_int32), ("worker_count", C.c_int), ("tile_count", C.c_int), ("xres", C.c_int),
                ("yres", C.c_int), ("frame_region", C.c_int * 4), ("framebuffer", C.c_void_p)]


class TileInfo(C.Structure):           # fj::TileInfo, src/fj_callback.h:39-59
    _fields_ = [("frame_id", C.c_int32), ("worker_id", C.c_int), ("region_id", C.c_int), ("total_region_count", C.c_int),
                ("tile_region", C.c_int * 4), ("framebuffer", C.c_void_p)]


FRAME_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(FrameInfo))
TILE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(TileInfo))
SAMPLE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p)
CALLBACK_CONTINUE, CALLBACK_INTERRUPT = 0, -1


def lib():
    global _lib
    if _lib is None:
        ffi.load("libfjgpu.so")
        L = ffi.load("libfjscene.so")
        L.fj_scene_run_text.argtypes = [C.c_char_p, C.c_int]
        L.fj_scene_run_text.restype = C.c_int
        L.fj_scene_last_error.restype = C.c_char_p
        L.fj_scene_set_deferred_render.argtypes = [C.c_int]
        L.fj_scene_get_desc.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.POINTER(ffi.RenderDesc))]
        L.fj_scene_get_desc.restype = C.c_int
        L.fj_framebuffer_data.argtypes = [C.c_long, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.fj_framebuffer_data.restype = C.POINTER(C.c_float)
        L.fj_scene_last_stats.argtypes = [C.POINTER(ffi.RenderStats)]
        L.fj_SiCloseScene.restype = C.c_int
        L.fj_SiRenderScene.argtypes = [C.c_long]
        L.fj_SiRenderScene.restype = C.c_int
        L.fj_SiSetFrameReportCallback.argtypes = [C.c_long, C.c_void_p, FRAME_CB, FRAME_CB, FRAME_CB]
        L.fj_SiSetFrameReportCallback.restype = C.c_int
        L.fj_SiSetTileReportCallback.argtypes = [C.c_long, C.c_void_p, TILE_CB, SAMPLE_CB, TILE_CB]
        L.fj_SiSetTileReportCallback.restype = C.c_int
        _lib = L
    return _lib


class SceneError(RuntimeError):
    pass


def run_scene_text(text, deferred=False, echo=False):
    """Run scene-description text through the built-in parser.

    deferred=True: RenderScene only prepares the flat description (no GPU work).
    Raises SceneError with the parser's message on the first failing command.
    """
    L = lib()
    L.fj_scene_set_deferred_render(1 if deferred else 0)
    rc = L.fj_scene_run_text(text.encode("utf-8"), 1 if echo else 0)
    if rc != 0:
        raise SceneError(L.fj_scene_last_error().decode("utf-8", "replace"))
    return 0


def render_with_callbacks(renderer_index=0, frame_start=None, frame_done=None, tile_start=None, tile_done=None,
                          sample_done=None, frame_abort=None):
    """SiSetFrameReportCallback / SiSetTileReportCallback + SiRenderScene on renderer
    #renderer_index of the current scene (built with run_scene_text(..., deferred=True)).
    Callbacks are Python callables (info) -> CALLBACK_CONTINUE / CALLBACK_INTERRUPT.
    Returns SiRenderScene's status (0 = SI_SUCCESS, -1 = SI_FAIL)."""
    L = lib()
    rid = TYPE_ID_OFFSET * TYPE_RENDERER + renderer_index

    def wrap(kind, fn):
        if fn is None:
            return kind()
        if kind is SAMPLE_CB:
            return kind(lambda data: int(fn() or 0))
        return kind(lambda data, info: int(fn(info.contents) or 0))

    keep = [wrap(FRAME_CB, frame_start), wrap(FRAME_CB, frame_abort), wrap(FRAME_CB, frame_done),
            wrap(TILE_CB, tile_start), wrap(SAMPLE_CB, sample_done), wrap(TILE_CB, tile_done)]
    if L.fj_SiSetFrameReportCallback(rid, None, keep[0], keep[1], keep[2]) != 0:
        raise SceneError("SiSetFrameReportCallback failed")
    if L.fj_SiSetTileReportCallback(rid, None, keep[3], keep[4], keep[5]) != 0:
        raise SceneError("SiSetTileReportCallback failed")
    L.fj_scene_set_deferred_render(0)
    rc = L.fj_SiRenderScene(rid)
    # unhook before the ctypes thunks go away
    L.fj_SiSetFrameReportCallback(rid, None, FRAME_CB(), FRAME_CB(), FRAME_CB())
    L.fj_SiSetTileReportCallback(rid, None, TILE_CB(), SAMPLE_CB(), TILE_CB())
    return rc


def get_desc():
    """(scene_desc void*, RenderDesc copy) of the last RenderScene."""
    L = lib()
    sp = C.c_void_p()
    rp = C.POINTER(ffi.RenderDesc)()
    if L.fj_scene_get_desc(C.byref(sp), C.byref(rp)) != 0:
        raise SceneError("no scene description available (RenderScene not run)")
    return sp, rp.contents.copy()


def framebuffer(index=0):
    """numpy copy [H, W, C] float32 of framebuffer #index."""
    L = lib()
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    p = L.fj_framebuffer_data(TYPE_ID_OFFSET * TYPE_FRAMEBUFFER + index, C.byref(w), C.byref(h), C.byref(c))
    if not p or w.value * h.value * c.value == 0:
        raise SceneError("framebuffer is empty")
    n = w.value * h.value * c.value
    return np.ctypeslib.as_array(p, shape=(n,)).reshape(h.value, w.value, c.value).copy()


def last_stats():
    st = ffi.RenderStats()
    lib().fj_scene_last_stats(C.byref(st))
    return st


def close_scene():
    lib().fj_SiCloseScene()
