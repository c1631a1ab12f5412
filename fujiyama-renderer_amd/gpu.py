"""ctypes face of lib/libfjgpu.so (include/fjgpu.h): the HIP core."""
import ctypes as C

import numpy as np

from . import ffi

_lib = None


class GpuError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        L = ffi.load("libfjgpu.so")
        L.fjgpu_last_error.restype = C.c_char_p
        L.fjgpu_device_count.restype = C.c_int
        L.fjgpu_scene_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        L.fjgpu_scene_destroy.argtypes = [C.c_void_p]
        L.fjgpu_scene_destroy.restype = None
        L.fjgpu_tile_count.argtypes = [C.POINTER(ffi.RenderDesc)]
        L.fjgpu_tile_rect.argtypes = [C.POINTER(ffi.RenderDesc), C.c_int, C.POINTER(C.c_int32)]
        L.fjgpu_render_tiles.argtypes = [C.c_void_p, C.POINTER(ffi.RenderDesc), C.c_void_p, C.c_int,
                                         C.c_void_p, C.c_void_p, C.POINTER(ffi.GpuStats)]
        L.fjgpu_render_frame.argtypes = [C.c_void_p, C.POINTER(ffi.RenderDesc), C.c_void_p, C.POINTER(ffi.GpuStats)]
        L.fjgpu_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.POINTER(ffi.GpuStats)]
        L.fjgpu_scene_create_multi.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
        L.fjgpu_render_frame_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(ffi.RenderDesc), C.c_void_p, C.c_int,
                                               C.c_void_p, C.POINTER(ffi.GpuStats)]
        L.fjgpu_pack_tiles.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.fjgpu_unpack_tiles.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.fjgpu_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_long]
        L.fjgpu_global_option.argtypes = [C.c_char_p, C.c_long]
        L.fjgpu_scene_query.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double)]
        L.fjgpu_dev_rccl_selftest.argtypes = [C.c_int, C.c_int]
        L.fjgpu_dev_sort_pairs.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        _lib = L
    return _lib


def _check(rc):
    if rc != 0:
        raise GpuError("fjgpu error %d: %s" % (rc, lib().fjgpu_last_error().decode("utf-8", "replace")))


def device_count():
    return lib().fjgpu_device_count()


class Scene(object):
    """Device-resident scene (BLAS, instances, lights, shaders, textures)."""

    def __init__(self, scene_desc_ptr, device=0):
        self._h = C.c_void_p()
        _check(lib().fjgpu_scene_create(scene_desc_ptr, device, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().fjgpu_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def query(self, name):
        v = C.c_double(0)
        _check(lib().fjgpu_scene_query(self._h, name.encode(), C.byref(v)))
        return v.value

    def set_option(self, name, value):
        _check(lib().fjgpu_set_option(self._h, name.encode(), int(value)))

    def render_frame(self, render):
        """All tiles -> numpy [H, W, 4] float32 plus GpuStats."""
        fb = np.zeros((render.yres, render.xres, 4), dtype=np.float32)
        st = ffi.GpuStats()
        _check(lib().fjgpu_render_frame(self._h, C.byref(render), fb.ctypes.data_as(C.c_void_p), C.byref(st)))
        return fb, st

    def render_tiles(self, render, tile_ids, d_framebuffer_ptr, stream=None):
        """Listed tiles into a DEVICE framebuffer (raw pointer, e.g. torch data_ptr())."""
        st = ffi.GpuStats()
        if tile_ids is None:
            ids_p, n = None, 0
        else:
            ids = np.ascontiguousarray(tile_ids, dtype=np.int32)
            ids_p, n = ids.ctypes.data_as(C.c_void_p), len(ids)
        _check(lib().fjgpu_render_tiles(self._h, C.byref(render), ids_p, n, C.c_void_p(d_framebuffer_ptr),
                                        C.c_void_p(stream or 0), C.byref(st)))
        return st

    def trace(self, group, rays):
        """Closest hits of rays [n, 8] (orig, dir, tmin, tmax) -> t [n], ids [n, 2], uv [n, 2], stats."""
        rays = np.ascontiguousarray(rays, dtype=np.float64)
        n = rays.shape[0]
        t = np.empty(n, dtype=np.float64)
        ids = np.empty((n, 2), dtype=np.int32)
        uv = np.empty((n, 2), dtype=np.float64)
        st = ffi.GpuStats()
        _check(lib().fjgpu_trace(self._h, group, n, rays.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p),
                                 ids.ctypes.data_as(C.c_void_p), uv.ctypes.data_as(C.c_void_p), C.byref(st)))
        return t, ids, uv, st


class MultiScene(object):
    """The same scene resident on several devices (fjgpu_scene_create_multi): one host-side
    build, one upload per entry of `devices` (an index may repeat: two replicas on one device
    exercise the multi-device path on a single GPU)."""

    def __init__(self, scene_desc_ptr, devices):
        self.n = len(devices)
        self._h = (C.c_void_p * self.n)()
        devs = (C.c_int * self.n)(*devices)
        _check(lib().fjgpu_scene_create_multi(scene_desc_ptr, devs, self.n, self._h))

    def close(self):
        for k in range(self.n):
            if self._h[k]:
                lib().fjgpu_scene_destroy(self._h[k])
                self._h[k] = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render_frame(self, render, tile_ids=None):
        """Tile k of the list on replica k % n, slabs gathered on the first device ->
        numpy [H, W, 4] float32 plus one GpuStats per replica."""
        fb = np.zeros((render.yres, render.xres, 4), dtype=np.float32)
        st = (ffi.GpuStats * self.n)()
        if tile_ids is None:
            ids_p, n = None, 0
        else:
            ids = np.ascontiguousarray(tile_ids, dtype=np.int32)
            ids_p, n = ids.ctypes.data_as(C.c_void_p), len(ids)
        _check(lib().fjgpu_render_frame_multi(self._h, self.n, C.byref(render), ids_p, n, fb.ctypes.data_as(C.c_void_p), st))
        return fb, list(st)


def host_instance_level(scene_desc_ptr, group):
    """(inst [n], skip [n], box [n, 6]) of the group's instance level as the host builder lays it out; no device needed"""
    L = lib()
    L.fjgpu_host_instance_level.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.fjgpu_host_instance_level.restype = C.c_int
    n = L.fjgpu_host_instance_level(scene_desc_ptr, group, None, None, None, 0)
    if n < 0:
        _check(n)
    inst = np.zeros(n, dtype=np.int32)
    skip = np.zeros(n, dtype=np.int32)
    box = np.zeros((n, 6), dtype=np.float64)
    _check(min(0, L.fjgpu_host_instance_level(scene_desc_ptr, group, inst.ctypes.data_as(C.c_void_p), skip.ctypes.data_as(C.c_void_p),
                                              box.ctypes.data_as(C.c_void_p), n)))
    return inst, skip, box


def global_option(name, value):
    """process-wide option of the core, e.g. global_option("device_build", 1)"""
    _check(lib().fjgpu_global_option(name.encode(), int(value)))


def tile_count(render):
    return lib().fjgpu_tile_count(C.byref(render))


def rccl_selftest(device=0, n_floats=1 << 20):
    """RCCL bring-up on one device through the core's own loader (include/fjgpu.h: fjgpu_dev_rccl_selftest)"""
    _check(lib().fjgpu_dev_rccl_selftest(device, n_floats))


def sort_pairs(keys, key_bits, device=0, repeats=1):
    """the ray-queue sort on its own (include/fjgpu.h: fjgpu_dev_sort_pairs): (keys[i], i) pairs, stable, over the low key_bits
    bits -> (sorted keys, perm, milliseconds of the fastest of `repeats` device-side runs)"""
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    out, perm = np.empty_like(keys), np.empty_like(keys)
    ms = C.c_double(0)
    _check(lib().fjgpu_dev_sort_pairs(device, keys.ctypes.data_as(C.c_void_p), int(keys.size), int(key_bits),
                                      out.ctypes.data_as(C.c_void_p), perm.ctypes.data_as(C.c_void_p), int(repeats), C.byref(ms)))
    return out, perm, ms.value


def tile_rect(render, tile_id):
    r = (C.c_int32 * 4)()
    _check(lib().fjgpu_tile_rect(C.byref(render), tile_id, r))
    return tuple(r)


def pack_tiles(fb_ptr, xres, rects_ptr, n_tiles, tile_px, slab_ptr, stream=None):
    """device framebuffer -> tile slab (include/fjgpu.h: fjgpu_pack_tiles); raw device pointers"""
    _check(lib().fjgpu_pack_tiles(fb_ptr, xres, rects_ptr, n_tiles, tile_px, slab_ptr, stream))


def unpack_tiles(fb_ptr, xres, rects_ptr, n_tiles, tile_px, slab_ptr, stream=None):
    _check(lib().fjgpu_unpack_tiles(fb_ptr, xres, rects_ptr, n_tiles, tile_px, slab_ptr, stream))
