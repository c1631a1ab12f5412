"""ctypes loader for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_DIR = os.path.join(ROOT, "oracle", "_ref")

import sys
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from fujiyama_renderer_amd import ffi  # noqa: E402

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            raise RuntimeError("oracle/liboracle.so missing: run make -C oracle restate")
        L = C.CDLL(ORACLE_SO)
        L.fjo_scene_create.restype = C.c_void_p
        L.fjo_scene_create.argtypes = [C.c_void_p]
        L.fjo_scene_destroy.argtypes = [C.c_void_p]
        L.fjo_scene_destroy.restype = None
        L.fjo_scene_render.argtypes = [C.c_void_p, C.POINTER(ffi.RenderDesc), C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_int, C.POINTER(ffi.RayCounts)]
        L.fjo_scene_render_serial.argtypes = [C.c_void_p, C.POINTER(ffi.RenderDesc), C.c_void_p, C.c_int, C.c_void_p,
                                              C.POINTER(ffi.RayCounts)]
        L.fjo_scene_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


class OracleScene(object):
    def __init__(self, scene_desc_ptr):
        self._h = C.c_void_p(lib().fjo_scene_create(scene_desc_ptr))

    def close(self):
        if self._h:
            lib().fjo_scene_destroy(self._h)
            self._h = None

    def render(self, render, tile_ids=None, threads=None):
        fb = np.zeros((render.yres, render.xres, 4), dtype=np.float32)
        rc = ffi.RayCounts()
        if tile_ids is None:
            ids_p, n = None, 0
        else:
            ids = np.ascontiguousarray(tile_ids, dtype=np.int32)
            ids_p, n = ids.ctypes.data_as(C.c_void_p), len(ids)
        threads = threads or min(os.cpu_count() or 1, 64)
        e = lib().fjo_scene_render(self._h, C.byref(render), ids_p, n, fb.ctypes.data_as(C.c_void_p), threads, C.byref(rc))
        if e:
            raise RuntimeError("oracle render failed: %d" % e)
        return fb, rc

    def render_serial(self, render, tile_ids=None):
        """one worker, the reference's own random streams in its draw order (fjo_render.h:
        serial_rng): reproduces a `thread_count 1` reference render of PathtracingShader /
        area-light scenes bit for bit"""
        fb = np.zeros((render.yres, render.xres, 4), dtype=np.float32)
        rc = ffi.RayCounts()
        if tile_ids is None:
            ids_p, n = None, 0
        else:
            ids = np.ascontiguousarray(tile_ids, dtype=np.int32)
            ids_p, n = ids.ctypes.data_as(C.c_void_p), len(ids)
        e = lib().fjo_scene_render_serial(self._h, C.byref(render), ids_p, n, fb.ctypes.data_as(C.c_void_p), C.byref(rc))
        if e:
            raise RuntimeError("oracle render failed: %d" % e)
        return fb, rc

    def trace(self, group, rays, time=0.0):
        rays = np.ascontiguousarray(rays, dtype=np.float64)
        n = rays.shape[0]
        t = np.empty(n)
        ids = np.empty((n, 2), dtype=np.int32)
        attr = np.empty((n, 8))
        e = lib().fjo_scene_trace(self._h, group, n, rays.ctypes.data_as(C.c_void_p), time,
                                  t.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p), attr.ctypes.data_as(C.c_void_p))
        if e:
            raise RuntimeError("oracle trace failed: %d" % e)
        return t, ids, attr
