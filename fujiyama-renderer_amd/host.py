"""ctypes face of lib/libfjscene.so (include/fj_scene_interface.h)."""
import ctypes as C

import numpy as np

from . import ffi

_lib = None

TYPE_ID_OFFSET = 10000000
TYPE_FRAMEBUFFER = 3
TYPE_RENDERER = 8


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
