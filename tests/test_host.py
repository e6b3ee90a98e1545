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
