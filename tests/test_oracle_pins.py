"""Pins the CPU oracle (and the host mirror that feeds it) to outputs of the reference itself.

The reference cannot be compiled in this image, so the pins are the known answers SURVEY.md §4/§6
recorded from its CPU path: RNG vectors, the CameraData of the create_test_config.py scene, scene
element counts, traversal statistics and — the strong one — the sha256 of the BinarySaver file the
reference's `main --cpu` writes for that scene, which only matches if the config parser, the
polyhedra/scene builder, the BVH, the camera, every intersection/material routine, the RNG, the
sample summation order and the saver arithmetic all agree bit for bit.
"""
import ctypes as C
import hashlib

import numpy as np

import oracle_bindings as ob
import rtp_bindings as rb


def test_wang_hash_known_answers(golden):
    lib = ob.lib()
    for k, v in golden["wang_hash"].items():
        assert lib.orc_wang_hash(int(k)) == v


def test_random_float_known_answers(golden):
    lib = ob.lib()
    seed = C.c_uint32(lib.orc_wang_hash((lib.orc_wang_hash(0) + 0) & 0xFFFFFFFF))   # pixel (0,0), sample 0
    assert seed.value == golden["pixel00_sample0_seed"]
    got = [lib.orc_random_float(C.byref(seed)) for _ in range(4)]
    want = np.array(golden["pixel00_first_random_floats"], dtype=np.float32)
    assert np.array_equal(np.array(got, dtype=np.float32), want)


def test_random_float_can_return_one():
    # seeds >= 0xFFFFFF80 round up to 2^32 in float
    lib = ob.lib()
    # find a state whose hash lands in the top 128 values is impractical; check the arithmetic
    assert np.float32(np.uint32(0xFFFFFF80)) / np.float32(4294967296.0) == np.float32(1.0)
    assert lib.orc_wang_hash(61) == 0


def test_config_scene_counts(test_config_text, golden):
    hs = rb.HostScene.from_config(test_config_text)
    c = golden["config_scene_counts"]
    assert hs.desc.num_spheres == c["spheres"]
    assert hs.desc.num_planes == c["planes"]
    assert hs.desc.num_materials == c["materials"]
    assert hs.desc.num_nodes == c["bvh_nodes"]
    types = [hs.desc.planes[i].type for i in range(hs.desc.num_planes)]
    assert types.count(0) == c["quads"] and types.count(2) == c["triangles"]


def test_camera_data_known_answer(test_config_text, golden):
    hs = rb.HostScene.from_config(test_config_text)
    cam = hs.frame_camera(0)
    for field, want in golden["test_config_camera"].items():
        got = np.array(list(getattr(cam, field).e), dtype=np.float32)
        assert np.array_equal(got, np.array(want, dtype=np.float32)), field
    assert (cam.image_width, cam.image_height, cam.samples_per_pixel, cam.max_depth) == (200, 100, 4, 5)


def test_reference_image_sha256(test_config_text, golden):
    """oracle render + BinarySaver bytes == the file the reference's `main --cpu` wrote."""
    hs = rb.HostScene.from_config(test_config_text)
    cam = hs.frame_camera(0)
    fb = ob.render(hs, cam)
    data = rb.binary_image_bytes(fb, cam.image_width, cam.image_height, hs.info.sqrt_spp)
    assert len(data) == golden["test_config_binary_saver_bytes"]
    assert hashlib.sha256(data).hexdigest() == golden["test_config_binary_saver_sha256"]
    # the oracle's own restatement of the saver arithmetic agrees with the host mirror's
    assert np.array_equal(ob.write_color_bytes(fb[:5], hs.info.sqrt_spp).reshape(5, -1, 3),
                          rb.quantize(fb[:5], hs.info.sqrt_spp))


def test_textured_reference_image_sha256(golden, tmp_path):
    """Second full-image pin: the reference's config.txt shrunk to 400x225, depth
    10, 4^2 spp with floor.jpg as the floor texture.  Covers the host JPEG decoder, tex2D_cpu, the
    textured METAL floor and the polyhedra at a second resolution.  Needs /root/reference (for
    floor.jpg and config.txt), so it runs in the build container only."""
    import os
    import pytest
    ref = "/root/reference"
    if not os.path.exists(os.path.join(ref, "floor.jpg")):
        pytest.skip("reference tree not present (GPU box)")
    lines = open(os.path.join(ref, "config.txt")).read().split("\n")
    while not lines[-1].strip():
        lines.pop()
    lines[0] = "1"
    lines[2] = "400 225 50"
    lines[8] = lines[8].replace("../floor2.jpg", os.path.join(ref, "floor.jpg"))
    lines[-1] = "10 4"
    hs = rb.HostScene.from_config("\n".join(lines) + "\n")       # the host mirror's own JPEG decoder
    assert hs.desc.num_textures == 1 and hs.desc.textures[0].width == 2000 and hs.desc.textures[0].height == 1330
    cam = hs.frame_camera(0)
    fb = ob.render(hs, cam, threads=8)
    data = rb.binary_image_bytes(fb, 400, 225, hs.info.sqrt_spp)
    assert hashlib.sha256(data).hexdigest() == golden["config_txt_400x225_d10_spp16_floor_jpg_sha256"]


def test_jpeg_decoder_matches_reference_decoder(tmp_path):
    """Every texel of floor.jpg decoded by host/jpeg_decoder.cpp equals, bit for bit, what the
    reference's own vendored stb_image.h produces (compiled where it lies by `make -C oracle _ref`;
    build container only)."""
    import ctypes as C
    import os
    import subprocess
    import pytest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    jpg = "/root/reference/floor.jpg"
    if not os.path.exists(jpg):
        pytest.skip("reference tree not present (GPU box)")
    subprocess.run(["make", "-C", os.path.join(root, "oracle"), "_ref"], check=True, capture_output=True)
    pfm = str(tmp_path / "floor.pfm")
    subprocess.run([os.path.join(root, "oracle", "_ref", "stb_decode"), jpg, pfm], check=True)
    lib = rb.host_lib()
    lib.rtp_host_load_texture.argtypes = [C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p]
    w, h = C.c_int32(), C.c_int32()
    assert lib.rtp_host_load_texture(jpg.encode(), C.byref(w), C.byref(h), None) == 0
    mine = np.empty((h.value, w.value, 4), dtype=np.float32)
    assert lib.rtp_host_load_texture(jpg.encode(), C.byref(w), C.byref(h), mine.ctypes.data) == 0
    raw = open(pfm, "rb").read()
    start = raw.index(b"-1.0\n") + 5
    ref = np.frombuffer(raw[start:], dtype="<f4").reshape(h.value, w.value, 3)[::-1]
    assert (w.value, h.value) == (2000, 1330)
    assert np.array_equal(mine[:, :, :3].view(np.uint32), np.ascontiguousarray(ref).view(np.uint32))
    assert (mine[:, :, 3] == 1.0).all()


def test_threaded_render_equals_serial(test_config_text):
    hs = rb.HostScene.from_config(test_config_text)
    cam = hs.frame_camera(0)
    a = ob.render(hs, cam, threads=1)
    b = ob.render(hs, cam, threads=5)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    part = ob.render(hs, cam, row0=37, row1=53, threads=3)
    assert np.array_equal(part.view(np.uint32), a[37:53].view(np.uint32))


def test_benchmark_scene_matches_survey_statistics(golden):
    g = golden["rtiow"]
    hs = rb.HostScene.rtiow()
    assert hs.desc.num_spheres == g["spheres"] and hs.desc.num_nodes == g["nodes"]
    cam = rb.rtiow_camera(300, 200, 4, 50)       # survey: 1200x800x4; ratios are resolution independent to ~1 %
    _, st = ob.render(hs, cam, threads=4, want_stats=True)
    assert abs(st.rays / st.samples - g["rays_per_sample_1200x800x4"]) < 0.05
    assert abs(st.node_visits / st.rays - g["node_visits_per_ray"]) < 0.6
    assert abs(st.sphere_tests / st.rays - g["sphere_tests_per_ray"]) < 0.05
    assert st.max_stack == g["max_stack"]


def test_stress_scene_counts(golden):
    hs = rb.HostScene.rtiow(half_extent=158)
    assert hs.desc.num_spheres == golden["stress"]["spheres"]
    assert hs.desc.num_nodes == golden["stress"]["nodes"]


def test_golden_probe_files_still_match_oracle(test_config_text):
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    g = np.load(os.path.join(here, "golden", "config_probe.npz"))
    hs = rb.HostScene.from_config(test_config_text)
    cam = hs.frame_camera(0)
    rad, rays, seeds = ob.trace_samples(hs, cam, g["ijs"])
    assert np.array_equal(rad.view(np.uint32), g["rad_bits"]) and np.array_equal(rays, g["rays"]) and np.array_equal(seeds, g["seeds"])
    assert np.array_equal(ob.render(hs, cam).view(np.uint32), g["fb_bits"])
    g = np.load(os.path.join(here, "golden", "rtiow_probe.npz"))
    hs = rb.HostScene.rtiow()
    cam = rb.rtiow_camera(1920, 1080, 500, 50)
    rad, rays, seeds = ob.trace_samples(hs, cam, g["ijs"][:600])
    assert np.array_equal(rad.view(np.uint32), g["rad_bits"][:600]) and np.array_equal(seeds, g["seeds"][:600])
    assert np.array_equal(ob.render(hs, rb.rtiow_camera(96, 64, 4, 50)).view(np.uint32), g["fb_bits"])


def test_closest_hit_is_traversal_order_independent():
    """Brute force over all leaves (same box gate, same primitive tests) finds the same hit as
    the BVH walk: the property that lets the GPU use its own tree and visit order."""
    hs = rb.HostScene.rtiow()
    rng = np.random.default_rng(7)
    lib = ob.lib()
    n_hit = 0
    for _ in range(1500):
        o = rng.uniform(-8, 8, 3).astype(np.float32)
        o[2] = abs(o[2]) * 0.3 + 0.05
        d = rng.normal(size=3).astype(np.float32)
        t1, t2 = C.c_float(), C.c_float()
        ty1, ty2, i1, i2 = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        h1 = lib.orc_closest_hit(C.byref(hs.desc), o.ctypes.data, d.ctypes.data, C.byref(t1), C.byref(ty1), C.byref(i1))
        h2 = lib.orc_closest_hit_bruteforce(C.byref(hs.desc), o.ctypes.data, d.ctypes.data, C.byref(t2), C.byref(ty2), C.byref(i2))
        assert h1 == h2
        if h1:
            n_hit += 1
            assert t1.value == t2.value and (ty1.value, i1.value) == (ty2.value, i2.value)
    assert n_hit > 500
