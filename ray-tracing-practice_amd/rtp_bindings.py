"""ctypes bindings of the product libraries (no oracle in here).

librtp_host.so  — pure host code: config parser, scene/BVH builders, camera, saver arithmetic
                  (ray-tracing-practice_amd/host/rtp_host.h).  Loads without a GPU.
librtp_amd.so   — the MI355X render library behind the C ABI of include/rtp_amd.h.  Loading it
                  needs the HIP runtime; every compute call needs a GPU and fails loudly without
                  one (there is no CPU fallback in the product).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class Vec3(C.Structure):
    _fields_ = [("e", C.c_float * 3)]


class Sphere(C.Structure):
    _fields_ = [("center", Vec3), ("radius", C.c_float), ("material_idx", C.c_int32), ("_pad", C.c_int32 * 3)]


class Plane(C.Structure):
    _fields_ = [("type", C.c_int32), ("D", C.c_float), ("material_idx", C.c_int32), ("w", Vec3), ("u", Vec3),
                ("v", Vec3), ("base", Vec3), ("normal", Vec3), ("_pad", C.c_int32 * 2)]


class Material(C.Structure):
    _fields_ = [("type", C.c_int32), ("fuzz", C.c_float), ("ir", C.c_float), ("absorption", Vec3), ("albedo", Vec3),
                ("emit", Vec3), ("texture_id", C.c_uint64), ("reserved", C.c_uint64)]


class BvhNode(C.Structure):
    _fields_ = [("box", C.c_float * 6), ("left", C.c_int32), ("right", C.c_int32), ("type", C.c_int32)]


class CameraData(C.Structure):
    _fields_ = [("origin", Vec3), ("pixel00_loc", Vec3), ("pixel_delta_u", Vec3), ("pixel_delta_v", Vec3),
                ("background", Vec3), ("image_width", C.c_int32), ("image_height", C.c_int32),
                ("samples_per_pixel", C.c_int32), ("max_depth", C.c_int32)]


class Texture(C.Structure):
    _fields_ = [("rgba", C.POINTER(C.c_float)), ("width", C.c_int32), ("height", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [("spheres", C.POINTER(Sphere)), ("num_spheres", C.c_int32),
                ("planes", C.POINTER(Plane)), ("num_planes", C.c_int32),
                ("materials", C.POINTER(Material)), ("num_materials", C.c_int32),
                ("nodes", C.POINTER(BvhNode)), ("num_nodes", C.c_int32),
                ("textures", C.POINTER(Texture)), ("num_textures", C.c_int32)]


class Shard(C.Structure):
    _fields_ = [("band_rows", C.c_int32), ("num_parts", C.c_int32), ("part", C.c_int32)]


class Timing(C.Structure):
    """rt_timing (include/rtp_amd.h): an out-structure of the caller's size — struct_bytes is set on construction."""
    _fields_ = [("struct_bytes", C.c_uint32), ("kernel_ms", C.c_float), ("num_workgroups", C.c_uint32), ("workgroup_size", C.c_uint32),
                ("lds_bytes", C.c_uint32), ("scene_in_lds", C.c_uint32), ("trace_launches", C.c_uint32),
                ("trace_ms", C.c_float), ("guarded", C.c_uint32), ("flagged_samples", C.c_uint64), ("rework_ms", C.c_float),
                ("guard_unproven", C.c_uint32), ("kernel", C.c_uint32), ("guard_dynamic", C.c_uint32), ("wide_nodes", C.c_uint32),
                ("sphere_only", C.c_uint32), ("primary_visibility", C.c_uint32), ("primary_ms", C.c_float),
                ("trace_vgprs", C.c_uint32), ("trace_scratch_bytes", C.c_uint32), ("abandoned_passes", C.c_uint32),
                ("traced_samples", C.c_uint64), ("guard_paused", C.c_uint32), ("front_primitives", C.c_uint32)]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_bytes = C.sizeof(Timing)


TRAVERSAL_AUTO, TRAVERSAL_EXACT, TRAVERSAL_GUARDED = 0, 1, 2
BUILD_HOST_SAH, BUILD_DEVICE_LBVH = 0, 1
KERNEL_AUTO, KERNEL_MEGA, KERNEL_WAVEFRONT = 0, 1, 2


class Config(C.Structure):
    """rt_config (include/rtp_amd.h)."""
    _fields_ = [("struct_bytes", C.c_uint32), ("tree_build", C.c_int32), ("guard_gamma_ulps", C.c_float),
                ("guard_exact_leaf_table", C.c_int32), ("traversal", C.c_int32), ("guard_min_primitives", C.c_int32),
                ("guard_keep", C.c_int32), ("guard_repack", C.c_int32), ("kernel", C.c_int32), ("workspace_bytes", C.c_uint64),
                ("pass_spp", C.c_int32), ("stack_levels", C.c_int32), ("flag_capacity", C.c_uint32), ("scene_in_lds", C.c_int32),
                ("lds_treelet", C.c_int32), ("workgroups_per_cu", C.c_int32), ("k_inner", C.c_int32), ("k_shade", C.c_int32),
                ("reserve_chunk", C.c_int32), ("reserve_taper", C.c_int32), ("wavefront_paths", C.c_int32),
                ("wavefront_exchange", C.c_int32), ("wide_nodes", C.c_int32), ("guard_dynamic_margins", C.c_int32),
                ("sphere_only_kernel", C.c_int32), ("overlap_rework", C.c_int32), ("primary_visibility", C.c_int32),
                ("guard_bail_share", C.c_int32), ("guard_front_primitives", C.c_int32), ("reuse_view_lists", C.c_int32), ("resume_flagged", C.c_int32)]


def new_config():
    """rt_config with the library's defaults (what the header's rt_config_init macro does: the CALLER's struct size travels along)."""
    cfg = Config()
    lib = amd_lib()
    if hasattr(lib, "rt_config_init_sized"):
        lib.rt_config_init_sized(C.byref(cfg), C.sizeof(Config))
        assert cfg.struct_bytes == C.sizeof(Config) or os.environ.get("RTP_AMD_LIB"), (cfg.struct_bytes, C.sizeof(Config))
    else:       # an OLDER build loaded through RTP_AMD_LIB for an A/B run (developer tools): it fills the fields it has
        assert os.environ.get("RTP_AMD_LIB")
        lib.rt_config_init(C.byref(cfg))
    return cfg


class ConfigInfo(C.Structure):
    _fields_ = [("num_frames", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("max_depth", C.c_int32),
                ("sqrt_spp", C.c_int32), ("fov_degrees", C.c_float)]


assert C.sizeof(Sphere) == 32 and C.sizeof(Plane) == 80 and C.sizeof(Material) == 64
assert C.sizeof(BvhNode) == 36 and C.sizeof(CameraData) == 76

# Every symbol include/rtp_amd.h declares (tests check that the library exports all of them).
RTP_AMD_SYMBOLS = [
    "rt_set_device", "rt_scene_create", "rt_scene_create_ex", "rt_config_init", "rt_config_init_sized", "rt_config_from_env", "rt_scene_set_config",
    "rt_scene_get_config", "rt_scene_destroy", "rt_scene_guard_reason", "rt_shard_rows", "rt_render", "rt_render_tile", "rt_last_kernel_ms",
    "rt_last_timing", "rt_timing_init",
    "rt_render_to_host", "rt_trace_samples", "rt_closest_hits", "rt_device_alloc", "rt_device_free", "rt_copy_to_host", "rt_tonemap",
    "rt_get_last_error_string", "rt_version_string",
    "rt_context_create", "rt_context_destroy", "rt_context_num_devices", "rt_context_transport", "rt_context_scene_create",
    "rt_render_sharded", "rt_gather",
]

_host = None
_amd = None


def host_lib():
    """librtp_host.so (pure host)."""
    global _host
    if _host is None:
        path = os.path.join(_HERE, "librtp_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run __graft_entry__.build() (or make -C ray-tracing-practice_amd)")
        lib = C.CDLL(path)
        lib.rtp_host_scene_from_config.restype = C.c_void_p
        lib.rtp_host_scene_from_config.argtypes = [C.c_char_p, C.c_char_p]
        lib.rtp_host_scene_rtiow.restype = C.c_void_p
        lib.rtp_host_scene_rtiow.argtypes = [C.c_uint32, C.c_int32, C.c_int32, C.c_int32]
        lib.rtp_host_scene_from_arrays.restype = C.c_void_p
        lib.rtp_host_scene_from_arrays.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(Material), C.c_int32]
        lib.rtp_host_scene_free.argtypes = [C.c_void_p]
        lib.rtp_host_scene_desc.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        lib.rtp_host_scene_config.argtypes = [C.c_void_p, C.POINTER(ConfigInfo)]
        lib.rtp_host_frame_camera.argtypes = [C.c_void_p, C.c_int32, C.POINTER(CameraData)]
        lib.rtp_host_make_camera.argtypes = [C.c_int32, C.c_int32, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                             C.POINTER(C.c_float), C.c_int32, C.c_int32, C.POINTER(CameraData)]
        lib.rtp_host_quantize.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
        lib.rtp_host_write_binary_image.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        lib.rtp_host_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        lib.rtp_host_default_config.restype = C.c_char_p
        _host = lib
    return _host


def amd_lib():
    """librtp_amd.so (HIP).  Raises if the library is not built — there is no fallback."""
    global _amd
    if _amd is None:
        # RTP_AMD_LIB: developer override used by tools/ to A/B differently compiled kernels
        path = os.environ.get("RTP_AMD_LIB") or os.path.join(_HERE, "librtp_amd.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: the HIP render library must be built (no CPU fallback exists)")
        lib = C.CDLL(path)
        lib.rt_set_device.argtypes = [C.c_int32]
        lib.rt_scene_create.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
        lib.rt_scene_destroy.argtypes = [C.c_void_p]
        lib.rt_config_init.argtypes = [C.POINTER(Config)]
        lib.rt_config_init.restype = None
        if hasattr(lib, "rt_config_init_sized"):
            lib.rt_config_init_sized.argtypes = [C.POINTER(Config), C.c_uint32]
            lib.rt_config_init_sized.restype = None
        lib.rt_config_from_env.argtypes = [C.POINTER(Config)]
        lib.rt_config_from_env.restype = None
        lib.rt_scene_create_ex.argtypes = [C.POINTER(SceneDesc), C.POINTER(Config), C.POINTER(C.c_void_p)]
        lib.rt_scene_set_config.argtypes = [C.c_void_p, C.POINTER(Config)]
        lib.rt_scene_get_config.argtypes = [C.c_void_p, C.POINTER(Config)]
        lib.rt_scene_guard_reason.argtypes = [C.c_void_p]
        lib.rt_scene_guard_reason.restype = C.c_char_p
        lib.rt_shard_rows.argtypes = [C.c_int32, C.POINTER(Shard)]
        lib.rt_shard_rows.restype = C.c_int32
        lib.rt_render.argtypes = [C.c_void_p, C.POINTER(CameraData), C.POINTER(Shard), C.c_void_p, C.c_void_p,
                                  C.c_int32, C.POINTER(Timing)]
        lib.rt_render_tile.argtypes = [C.c_void_p, C.POINTER(CameraData), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_int32, C.POINTER(Timing)]
        lib.rt_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        lib.rt_last_timing.argtypes = [C.c_void_p, C.POINTER(Timing)]
        lib.rt_timing_init.argtypes = [C.POINTER(Timing)]
        lib.rt_timing_init.restype = None
        lib.rt_render_to_host.argtypes = [C.c_void_p, C.POINTER(CameraData), C.POINTER(Shard), C.c_void_p,
                                          C.POINTER(Timing)]
        lib.rt_trace_samples.argtypes = [C.c_void_p, C.POINTER(CameraData), C.c_int32, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]
        lib.rt_closest_hits.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.rt_device_alloc.argtypes = [C.c_uint64, C.POINTER(C.c_void_p)]
        lib.rt_device_free.argtypes = [C.c_void_p]
        lib.rt_copy_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        lib.rt_tonemap.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
        lib.rt_get_last_error_string.restype = C.c_char_p
        lib.rt_version_string.restype = C.c_char_p
        lib.rt_context_create.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]
        lib.rt_context_destroy.argtypes = [C.c_void_p]
        lib.rt_context_num_devices.argtypes = [C.c_void_p]
        lib.rt_context_transport.argtypes = [C.c_void_p]
        lib.rt_context_transport.restype = C.c_char_p
        lib.rt_context_scene_create.argtypes = [C.c_void_p, C.POINTER(SceneDesc), C.POINTER(Config)]
        lib.rt_render_sharded.argtypes = [C.c_void_p, C.POINTER(CameraData), C.c_int32, C.c_void_p, C.POINTER(Timing)]
        lib.rt_gather.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        _amd = lib
    return _amd


class RtError(RuntimeError):
    pass


def _check(status, what):
    if status != 0:
        msg = amd_lib().rt_get_last_error_string().decode()
        raise RtError(f"{what} failed with rt_status {status}: {msg}")


class HostScene:
    """Host-side scene (arrays in the reference's layouts) built by librtp_host.so."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("host scene construction failed")
        self._h = C.c_void_p(handle)
        self.desc = SceneDesc()
        host_lib().rtp_host_scene_desc(self._h, C.byref(self.desc))
        self.info = ConfigInfo()
        host_lib().rtp_host_scene_config(self._h, C.byref(self.info))

    @classmethod
    def from_config(cls, text, texture_dir=""):
        return cls(host_lib().rtp_host_scene_from_config(text.encode(), (texture_dir or "").encode()))

    @classmethod
    def rtiow(cls, seed=12345, half_extent=11, textured_quad=False, texture_size=1024):
        return cls(host_lib().rtp_host_scene_rtiow(seed, half_extent, int(textured_quad), texture_size))

    @classmethod
    def from_arrays(cls, spheres, planes, materials):
        """spheres: [n,5] (cx,cy,cz,radius,material); planes: [n,11] (base,u,v,material,type);
        materials: list of Material."""
        sp = np.ascontiguousarray(np.asarray(spheres, dtype=np.float32).reshape(-1, 5))
        pl = np.ascontiguousarray(np.asarray(planes, dtype=np.float32).reshape(-1, 11))
        mats = (Material * max(len(materials), 1))(*materials)
        return cls(host_lib().rtp_host_scene_from_arrays(sp.ctypes.data, sp.shape[0], pl.ctypes.data, pl.shape[0], mats,
                                                         len(materials)))

    def frame_camera(self, frame=0):
        cam = CameraData()
        host_lib().rtp_host_frame_camera(self._h, frame, C.byref(cam))
        return cam

    def nodes_array(self):
        n = self.desc.num_nodes
        return np.ctypeslib.as_array(C.cast(self.desc.nodes, C.POINTER(C.c_int32)), shape=(n, 9)).copy()

    def close(self):
        if self._h:
            host_lib().rtp_host_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_camera(width, height, vfov, eye, target, background=(0, 0, 0), spp=1, max_depth=50):
    cam = CameraData()
    f3 = C.c_float * 3
    host_lib().rtp_host_make_camera(width, height, vfov, f3(*eye), f3(*target), f3(*background), spp, max_depth,
                                    C.byref(cam))
    return cam


def rtiow_camera(width, height, spp, max_depth=50):
    """Benchmark camera of SURVEY.md §8(d): from (13,3,2) at the origin, vfov 20, sky (0.7,0.8,1.0)."""
    return make_camera(width, height, 20.0, (13, 3, 2), (0, 0, 0), (0.7, 0.8, 1.0), spp, max_depth)


def quantize(fb_sum, divisor):
    fb = np.ascontiguousarray(fb_sum, dtype=np.float32)
    out = np.empty(fb.size, dtype=np.uint8)
    host_lib().rtp_host_quantize(fb.ctypes.data, fb.size // 3, divisor, out.ctypes.data)
    return out.reshape(fb.shape)


def binary_image_bytes(fb_sum, width, height, divisor):
    """Bytes of the file BinarySaver writes (src/camera.cu:128-153)."""
    return np.array([width, height], dtype=np.int32).tobytes() + quantize(fb_sum, divisor).tobytes()


# Developer tools (tools/*.py) set HONOUR_ENV = True: handles made without an explicit honour_env then overlay the RTP_*
# variables through rt_config_from_env().  Tests and bench.py leave it off: they configure through rt_config fields only.
HONOUR_ENV = False
# rt_config fields every DeviceScene made without them starts from (a test fixture's way to say "guarded walk throughout")
DEFAULTS = {}


class DeviceScene:
    """rt_scene handle (device-resident repacked scene).

    Configuration: keyword arguments are rt_config fields (e.g. traversal=rb.TRAVERSAL_EXACT, pass_spp=64,
    tree_build=rb.BUILD_DEVICE_LBVH); configure(**fields) changes the render-time ones later.  With
    honour_env=True (what developer tools ask for through rb.HONOUR_ENV; the library itself never reads the
    environment) the RTP_* developer variables are overlaid through rt_config_from_env() at creation and
    before every render, so a harness can flip a knob around a single call."""

    def __init__(self, host_scene, device=None, honour_env=None, **config):
        lib = amd_lib()
        if device is not None:
            _check(lib.rt_set_device(device), "rt_set_device")
        self._h = C.c_void_p()
        self._honour_env = HONOUR_ENV if honour_env is None else honour_env
        self._explicit = dict(config)
        cfg = self._make_config()
        _check(lib.rt_scene_create_ex(C.byref(host_scene.desc), C.byref(cfg), C.byref(self._h)), "rt_scene_create_ex")
        self._keep = host_scene

    def _make_config(self):
        cfg = new_config()
        for k, v in {**DEFAULTS, **self._explicit}.items():
            setattr(cfg, k, v)
        if self._honour_env:
            amd_lib().rt_config_from_env(C.byref(cfg))
        return cfg

    def configure(self, **fields):
        """Change render-time rt_config fields of this handle."""
        self._explicit.update(fields)
        self._apply_config()

    def _apply_config(self):
        cfg = self._make_config()
        _check(amd_lib().rt_scene_set_config(self._h, C.byref(cfg)), "rt_scene_set_config")

    def config(self):
        cfg = Config()
        cfg.struct_bytes = C.sizeof(Config)
        _check(amd_lib().rt_scene_get_config(self._h, C.byref(cfg)), "rt_scene_get_config")
        return cfg

    def render_to_host(self, cam, shard=None):
        lib = amd_lib()
        rows = lib.rt_shard_rows(cam.image_height, C.byref(shard) if shard else None)
        fb = np.empty((rows, cam.image_width, 3), dtype=np.float32)
        t = Timing()
        self._apply_config()
        _check(lib.rt_render_to_host(self._h, C.byref(cam), C.byref(shard) if shard else None, fb.ctypes.data,
                                     C.byref(t)), "rt_render_to_host")
        return fb, t

    def render(self, cam, d_fb_ptr, shard=None, stream=None, sync=True):
        """d_fb_ptr: integer device address (e.g. torch tensor.data_ptr())."""
        t = Timing()
        self._apply_config()
        _check(amd_lib().rt_render(self._h, C.byref(cam), C.byref(shard) if shard else None, C.c_void_p(d_fb_ptr),
                                   C.c_void_p(stream or 0), 1 if sync else 0, C.byref(t)), "rt_render")
        return t

    def render_tile_to_host(self, cam, x0, y0, w, h):
        """rt_render_tile into a fresh device buffer, copied to the host: (h, w, 3) float32 sums and the rt_timing."""
        lib = amd_lib()
        d = C.c_void_p()
        _check(lib.rt_device_alloc(max(w, 0) * max(h, 0) * 12 or 12, C.byref(d)), "rt_device_alloc")
        t = Timing()
        self._apply_config()
        try:
            _check(lib.rt_render_tile(self._h, C.byref(cam), x0, y0, w, h, d, C.c_void_p(0), 1, C.byref(t)), "rt_render_tile")
            fb = np.empty((h, w, 3), dtype=np.float32)
            _check(lib.rt_copy_to_host(fb.ctypes.data, d, fb.nbytes), "rt_copy_to_host")
        finally:
            lib.rt_device_free(d)
        return fb, t

    def last_kernel_ms(self):
        ms = C.c_float()
        _check(amd_lib().rt_last_kernel_ms(self._h, C.byref(ms)), "rt_last_kernel_ms")
        return ms.value

    def last_timing(self):
        """Full rt_timing of the most recent rt_render of this scene (waits for it)."""
        t = Timing()
        _check(amd_lib().rt_last_timing(self._h, C.byref(t)), "rt_last_timing")
        return t

    def trace_kernel_name(self):
        """Name (as rocprofv3 prints it) of the dominant kernel of the most recent rt_render."""
        t = self.last_timing()
        lds = "true" if t.scene_in_lds else "false"
        if t.kernel == KERNEL_WAVEFRONT:
            return f"void rtk::render_kernel_wf<{lds}>(rtk::KParams)"
        return (f"void rtk::render_kernel<{lds}, {'false' if t.guarded else 'true'}, {'true' if t.guard_dynamic else 'false'}, "
                f"{'true' if t.wide_nodes else 'false'}, {'true' if t.sphere_only else 'false'}, "
                f"{'true' if t.primary_visibility else 'false'}>(rtk::KParams)")

    def trace_samples(self, cam, ijs):
        ijs = np.ascontiguousarray(ijs, dtype=np.int32).reshape(-1, 3)
        n = ijs.shape[0]
        rad = np.empty((n, 3), dtype=np.float32)
        rays = np.empty(n, dtype=np.int32)
        seeds = np.empty(n, dtype=np.uint32)
        _check(amd_lib().rt_trace_samples(self._h, C.byref(cam), n, ijs.ctypes.data, rad.ctypes.data, rays.ctypes.data,
                                          seeds.ctypes.data), "rt_trace_samples")
        return rad, rays, seeds

    def guard_reason(self):
        """'' when rt_render may use the guarded near-first walk, else why not."""
        return amd_lib().rt_scene_guard_reason(self._h).decode()

    def closest_hits(self, origins, directions):
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        hit = np.zeros(n, dtype=np.int32)
        t = np.zeros(n, dtype=np.float32)
        prim = np.zeros(n, dtype=np.int32)
        _check(amd_lib().rt_closest_hits(self._h, n, o.ctypes.data, d.ctypes.data, hit.ctypes.data, t.ctypes.data,
                                         prim.ctypes.data), "rt_closest_hits")
        return hit, t, prim

    def close(self):
        if self._h:
            amd_lib().rt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """rt_context: one frame sharded over the GPUs of the node through the C ABI (rt_render_sharded + rt_gather)."""

    def __init__(self, num_devices=0, ordinals=None):
        self._h = C.c_void_p()
        arr = (C.c_int32 * len(ordinals))(*ordinals) if ordinals else None
        _check(amd_lib().rt_context_create(num_devices, arr, C.byref(self._h)), "rt_context_create")
        self._keep = None

    @property
    def num_devices(self):
        return amd_lib().rt_context_num_devices(self._h)

    @property
    def transport(self):
        return amd_lib().rt_context_transport(self._h).decode()

    def scene(self, host_scene, **config):
        cfg = new_config()
        for k, v in config.items():
            setattr(cfg, k, v)
        _check(amd_lib().rt_context_scene_create(self._h, C.byref(host_scene.desc), C.byref(cfg)), "rt_context_scene_create")
        self._keep = host_scene

    def render(self, cam, d_fb_ptr, band_rows=8):
        """d_fb_ptr: device address on the root device of image_height*image_width*3 floats.  Returns the per-device timings."""
        t = (Timing * self.num_devices)()
        for k in range(self.num_devices):
            t[k].struct_bytes = C.sizeof(Timing)
        _check(amd_lib().rt_render_sharded(self._h, C.byref(cam), band_rows, C.c_void_p(d_fb_ptr), t), "rt_render_sharded")
        return list(t)

    def close(self):
        if self._h:
            amd_lib().rt_context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
