"""Host-side mirror (config loader, scene/BVH builders, camera, savers) and C-ABI loading."""
import ctypes as C
import io
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import rtp_bindings as rb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_library_exports_every_declared_symbol():
    lib = rb.amd_lib()
    header = open(os.path.join(ROOT, "include", "rtp_amd.h")).read()
    import re
    declared = sorted(set(re.findall(r"\b(rt_[a-z_]+)\s*\(", header)))
    assert declared == sorted(rb.RTP_AMD_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.rt_version_string()


def test_timing_struct_is_sized_by_the_caller():
    """rt_timing is an out-structure of the CALLER's size: rt_timing_init writes sizeof(rt_timing) as the library was compiled,
    which is what the ctypes mirror holds; a call with struct_bytes unset is refused (no GPU needed for either)."""
    lib = rb.amd_lib()
    t = rb.Timing()
    t.struct_bytes = 0
    lib.rt_timing_init(C.byref(t))
    assert t.struct_bytes == C.sizeof(rb.Timing)
    t.struct_bytes = 0
    assert lib.rt_last_timing(None, C.byref(t)) == 1          # RT_ERR_INVALID_ARG (null scene)


def test_shard_rows_partition_the_image():
    lib = rb.amd_lib()
    for height in (1, 7, 64, 225, 1080):
        for band in (1, 4, 16):
            for parts in (1, 2, 3, 8):
                rows = [lib.rt_shard_rows(height, C.byref(rb.Shard(band, parts, p))) for p in range(parts)]
                want = [sum(1 for j in range(height) if (j // band) % parts == p) for p in range(parts)]
                assert rows == want
        assert lib.rt_shard_rows(height, None) == height


def test_invalid_arguments_are_reported_not_fatal():
    lib = rb.amd_lib()
    assert lib.rt_scene_create(None, None) == 1
    assert lib.rt_render(None, None, None, None, None, 1, None) == 1
    assert b"null" in lib.rt_get_last_error_string()


def test_default_config_round_trip():
    text = rb.host_lib().rtp_host_default_config().decode()
    hs = rb.HostScene.from_config(text)
    i = hs.info
    assert (i.num_frames, i.width, i.height, i.max_depth, i.sqrt_spp) == (100, 1080, 720, 50, 50)
    assert abs(i.fov_degrees - 50.0) < 1e-6
    assert hs.desc.num_spheres == 94 and hs.desc.num_planes == 105 and hs.desc.num_nodes == 397
    # floor is METAL with fuzz = reflection coefficient; beads emit lights[0].col * 0.1
    m0, m1 = hs.desc.materials[0], hs.desc.materials[1]
    assert m0.type == 1 and abs(m0.fuzz - 0.3) < 1e-7 and m0.texture_id == 0     # ../floor2.jpg does not exist
    assert m1.type == 3 and list(m1.emit.e) == [np.float32(10.0) * np.float32(0.1)] * 3


def test_light_count_is_clamped_without_consuming_lines(test_config_text):
    lines = test_config_text.strip().split("\n")
    k = lines.index("4")     # the light count line
    lines[k] = "6"
    hs = rb.HostScene.from_config("\n".join(lines) + "\n")
    assert hs.desc.num_materials == 12        # still 4 lights → same material count


def test_bvh_structure(test_config_text):
    for hs in (rb.HostScene.from_config(test_config_text), rb.HostScene.rtiow()):
        n_prims = hs.desc.num_spheres + hs.desc.num_planes
        nodes = hs.nodes_array()
        assert nodes.shape[0] == 2 * n_prims - 1
        boxes = nodes[:, :6].copy().view(np.float32)
        left, right, typ = nodes[:, 6], nodes[:, 7], nodes[:, 8]
        leaves = left < 0
        assert leaves.sum() == n_prims and set(typ[~leaves]) == {-1}
        assert sorted(right[leaves & (typ == 0)]) == list(range(hs.desc.num_spheres))
        assert sorted(right[leaves & (typ == 1)]) == list(range(hs.desc.num_planes))
        inner = np.where(~leaves)[0]
        assert (left[inner] == inner + 1).all() and (right[inner] > left[inner]).all()     # pre-order
        for k in inner:
            for c in (left[k], right[k]):
                assert (boxes[k, 0::2] <= boxes[c, 0::2]).all() and (boxes[k, 1::2] >= boxes[c, 1::2]).all()
        assert ((boxes[:, 1::2] - boxes[:, 0::2]) >= np.float32(9.9e-5)).all()       # never thinner than ~1e-4
        for k in np.where(leaves & (typ == 0))[0][:50]:
            s = hs.desc.spheres[right[k]]
            c = np.array(list(s.center.e), dtype=np.float32)
            assert np.array_equal(boxes[k, 0::2], c - np.float32(s.radius)) and np.array_equal(boxes[k, 1::2], c + np.float32(s.radius))


def test_plane_precomputation(test_config_text):
    hs = rb.HostScene.from_config(test_config_text)
    for k in range(0, hs.desc.num_planes, 7):
        p = hs.desc.planes[k]
        u, v, base = (np.array(list(x.e), dtype=np.float64) for x in (p.u, p.v, p.base))
        n = np.cross(u, v)
        assert np.allclose(np.array(list(p.normal.e)), n / np.linalg.norm(n), atol=1e-6)
        assert np.isclose(p.D, np.dot(n / np.linalg.norm(n), base), atol=1e-4)
        assert np.allclose(np.array(list(p.w.e)), n / np.dot(n, n), rtol=1e-5, atol=1e-7)


def test_camera_orbit_and_saver(test_config_text, tmp_path):
    text = rb.host_lib().rtp_host_default_config().decode()
    hs = rb.HostScene.from_config(text)
    c0, c25 = hs.frame_camera(0), hs.frame_camera(25)
    # radius-15 orbit: frame 25 of 100 is a quarter turn later
    assert np.isclose(np.hypot(c0.origin.e[0], c0.origin.e[1]), 15.0, atol=1e-4)
    assert np.isclose(np.hypot(c25.origin.e[0], c25.origin.e[1]), 15.0, atol=1e-4)
    assert abs(c25.origin.e[0]) < 1e-3 and c25.origin.e[1] < -14.9
    assert c0.samples_per_pixel == 2500 and c0.max_depth == 50 and list(c0.background.e) == [0, 0, 0]

    rng = np.random.default_rng(0)
    fb = rng.uniform(0, 6, (5, 7, 3)).astype(np.float32)
    fb[0, 0] = [0.0, 1e6, -3.0]
    want = np.floor(256 * np.clip(np.sqrt(np.maximum(fb, 0) * np.float32(0.5)), 0, np.float32(0.999))).astype(np.uint8)
    got = rb.quantize(fb, 2)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1 and (got == want).mean() > 0.97
    path = str(tmp_path / "frame_0.png")
    rb.host_lib().rtp_host_write_binary_image(path.encode(), fb.ctypes.data, 7, 5, 2)
    data = open(path, "rb").read()
    assert struct.unpack("<ii", data[:8]) == (7, 5) and data[8:] == got.tobytes()
    png = str(tmp_path / "frame.png")
    rb.host_lib().rtp_host_write_png(png.encode(), fb.ctypes.data, 7, 5, 2)
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(png)), got)


def test_cli_default_and_cpu_modes():
    exe = os.path.join(ROOT, "ray-tracing-practice_amd", "rtp_main")
    out = subprocess.run([exe, "--default"], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout == rb.host_lib().rtp_host_default_config().decode()
    cpu = subprocess.run([exe, "--cpu"], capture_output=True, text=True, input="")
    assert cpu.returncode == 2 and "not available" in cpu.stderr      # no CPU render path in the product


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "ray-tracing-practice_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".so", ".o")) or f == "rtp_main":
                continue
            text = open(os.path.join(dirpath, f), errors="ignore").read()
            assert "rt_oracle" not in text and "oracle_bindings" not in text and "librt_oracle" not in text, f


def test_lambertian_and_metal_diffuse_share_one_routine():
    """include/materials.h spells the uniform-hemisphere scatter twice (LAMBERTIAN :74-78, METAL's 20 % branch
    :91-95).  The reference's own scenes never instantiate LAMBERTIAN, so the reference images (sha256 pins) only
    run the METAL copy.  Kernel and oracle each have ONE routine that both materials reach, so those pins cover
    LAMBERTIAN's arithmetic by construction: one definition, one call site in shade() / two calls of the same
    routine in the oracle, and no second copy of the hemisphere flip anywhere."""
    import re
    k = open(os.path.join(ROOT, "ray-tracing-practice_amd", "csrc", "rt_kernel.hip.inc")).read()
    code = "\n".join(line.split("//")[0] for line in k.splitlines())
    assert len(re.findall(r"\bf3 scatter_diffuse_dir\(", code)) == 1
    assert len(re.findall(r"=\s*scatter_diffuse_dir\(", code)) == 1
    shade = code[code.index("bool shade("):code.index("void start_sample(")]
    assert "scatter_diffuse_dir(in_sphere, normal)" in shade
    # the one draw that feeds it is shared by both materials, and nothing else flips a vector into the hemisphere
    assert "if (is_lamb || is_metal) in_sphere = random_in_unit_sphere(L.seed);" in shade
    assert len(re.findall(r"dot\(u, normal\) > 0\.0f \? u : neg\(u\)", code)) == 1
    o = open(os.path.join(ROOT, "oracle", "rt_oracle.c")).read()
    assert len(re.findall(r"static int scatter_diffuse\(", o)) == 1
    assert len(re.findall(r"return scatter_diffuse\(rec, attenuation, scattered, seed, albedo\);", o)) == 2
    assert len(re.findall(r"random_in_hemisphere\(rec->normal, seed\)", o)) == 1


def test_config_defaults_and_env_overlay():
    """rt_config: defaults, forward compatibility of struct_bytes, and the explicit environment overlay (the library
    itself never reads the environment — only rt_config_from_env does)."""
    lib = rb.amd_lib()
    cfg = rb.Config()
    lib.rt_config_init(C.byref(cfg))
    assert cfg.struct_bytes == C.sizeof(rb.Config)
    assert cfg.traversal == rb.TRAVERSAL_AUTO and cfg.guard_gamma_ulps == 0.0 and cfg.guard_min_primitives == 64
    assert cfg.guard_repack == 1 and cfg.scene_in_lds == 1 and cfg.lds_treelet == 1 and cfg.reserve_taper == 1
    assert cfg.workspace_bytes == 0          # auto: a sixteenth of the device's memory
    saved = {k: os.environ.get(k) for k in ("RTP_TRAVERSAL", "RTP_PASS_SPP", "RTP_GUARD_GAMMA_ULPS", "RTP_SLAB_GIB")}
    try:
        os.environ.update(RTP_TRAVERSAL="threaded", RTP_PASS_SPP="64", RTP_GUARD_GAMMA_ULPS="8", RTP_SLAB_GIB="2")
        lib.rt_config_from_env(C.byref(cfg))
        assert cfg.traversal == rb.TRAVERSAL_EXACT and cfg.pass_spp == 64 and cfg.guard_gamma_ulps == 8.0
        assert cfg.workspace_bytes == 2 << 30
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    # a caller compiled against an OLDER, shorter rt_config (what the header's rt_config_init macro hands over is ITS sizeof):
    # the library writes that many bytes and not one more — defaults, the environment overlay and rt_scene_get_config alike
    short = C.sizeof(rb.Config) - 8
    buf = (C.c_uint8 * (C.sizeof(rb.Config) + 16))(*([0xAB] * (C.sizeof(rb.Config) + 16)))
    old = C.cast(buf, C.POINTER(rb.Config))
    lib.rt_config_init_sized(old, short)
    assert old.contents.struct_bytes == short and old.contents.guard_min_primitives == 64
    assert all(b == 0xAB for b in bytes(buf)[short:]), "rt_config_init_sized wrote past the caller's struct"
    os.environ["RTP_NO_FRONT"] = "1"
    try:
        lib.rt_config_from_env(old)
        full = rb.new_config()
        lib.rt_config_from_env(C.byref(full))
    finally:
        os.environ.pop("RTP_NO_FRONT")
    assert all(b == 0xAB for b in bytes(buf)[short:]), "rt_config_from_env wrote past the caller's struct"
    assert full.guard_front_primitives == -1 and full.struct_bytes == C.sizeof(rb.Config)
    src = open(os.path.join(ROOT, "ray-tracing-practice_amd", "csrc", "rt_capi.hip")).read()
    render = src[src.index("rt_status render_impl(rt_scene *sc"):src.index("rt_status rt_last_timing(")]
    shipped = re_strip_dev(render)
    assert "getenv" not in shipped and "env_int" not in shipped


def re_strip_dev(text):
    """Drop the #ifdef RTP_DEV_QUEUE_KERNEL … #endif blocks (developer build only)."""
    import re
    return re.sub(r"#ifdef RTP_DEV_QUEUE_KERNEL.*?#endif", "", text, flags=re.S)
