"""ctypes bindings of the CPU oracle (oracle/librt_oracle.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  The product (ray-tracing-practice_amd/) never does.
"""
import ctypes as C
import os
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(_ROOT, "ray-tracing-practice_amd"))
import rtp_bindings as rb  # noqa: E402  (struct layouts only)


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("rays", C.c_uint64), ("node_visits", C.c_uint64), ("box_hits", C.c_uint64),
                ("sphere_tests", C.c_uint64), ("plane_tests", C.c_uint64), ("material_fetches", C.c_uint64),
                ("texture_fetches", C.c_uint64), ("max_stack", C.c_uint32)]

    def bytes_per_sample(self, spp):
        """rays*(visits*36 + sphere*32 + plane*80 + 64 per material fetch) + 12/spp."""
        if not self.samples:
            return 0.0
        # 64 B of material per ray is the survey's stated upper bound (1 hit per ray)
        b = self.node_visits * 36 + self.sphere_tests * 32 + self.plane_tests * 80 + self.rays * 64 \
            + self.texture_fetches * 64
        return b / self.samples + 12.0 / spp


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_ROOT, "oracle", "librt_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run __graft_entry__.build() or make -C oracle")
        l = C.CDLL(path)
        l.orc_wang_hash.restype = C.c_uint32
        l.orc_wang_hash.argtypes = [C.c_uint32]
        l.orc_random_float.restype = C.c_float
        l.orc_random_float.argtypes = [C.POINTER(C.c_uint32)]
        l.orc_get_ray.argtypes = [C.POINTER(rb.CameraData), C.c_int, C.c_int, C.POINTER(C.c_uint32), C.c_void_p, C.c_void_p]
        l.orc_tex2d.argtypes = [C.POINTER(rb.Texture), C.c_float, C.c_float, C.c_void_p]
        l.orc_closest_hit.argtypes = [C.POINTER(rb.SceneDesc), C.c_void_p, C.c_void_p, C.POINTER(C.c_float),
                                      C.POINTER(C.c_int), C.POINTER(C.c_int)]
        l.orc_closest_hit_bruteforce.argtypes = l.orc_closest_hit.argtypes
        l.orc_trace_sample.argtypes = [C.POINTER(rb.SceneDesc), C.POINTER(rb.CameraData), C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
        l.orc_render.argtypes = [C.POINTER(rb.SceneDesc), C.POINTER(rb.CameraData), C.c_int, C.c_int, C.c_void_p,
                                 C.c_int, C.POINTER(Stats)]
        l.orc_write_color.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _lib = l
    return _lib


def render(host_scene, cam, row0=0, row1=None, threads=1, want_stats=False):
    row1 = cam.image_height if row1 is None else row1
    fb = np.zeros((row1 - row0, cam.image_width, 3), dtype=np.float32)
    st = Stats()
    lib().orc_render(C.byref(host_scene.desc), C.byref(cam), row0, row1, fb.ctypes.data, threads,
                     C.byref(st) if want_stats else None)
    return (fb, st) if want_stats else fb


def trace_samples(host_scene, cam, ijs):
    ijs = np.asarray(ijs, dtype=np.int32).reshape(-1, 3)
    n = ijs.shape[0]
    rad = np.empty((n, 3), dtype=np.float32)
    rays = np.empty(n, dtype=np.int32)
    seeds = np.empty(n, dtype=np.uint32)
    r = C.c_int32()
    s = C.c_uint32()
    l = lib()
    for k in range(n):
        l.orc_trace_sample(C.byref(host_scene.desc), C.byref(cam), int(ijs[k, 0]), int(ijs[k, 1]), int(ijs[k, 2]),
                           rad[k].ctypes.data, C.byref(r), C.byref(s))
        rays[k] = r.value
        seeds[k] = s.value
    return rad, rays, seeds


def write_color_bytes(fb_sum, divisor):
    fb = np.ascontiguousarray(fb_sum, dtype=np.float32).reshape(-1, 3)
    out = np.empty((fb.shape[0], 3), dtype=np.uint8)
    l = lib()
    for k in range(fb.shape[0]):
        l.orc_write_color(fb[k].ctypes.data, divisor, out[k].ctypes.data)
    return out
