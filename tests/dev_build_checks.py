"""Checks that need the DEVELOPER build of the render library (librtp_amd_dev.so, `make -C ray-tracing-practice_amd dev`):
the experimental wavefront kernel and the rt_debug_* entry points.  Not collected by the normal test run (the file name does
not match test_*.py): tests/test_gpu_parity.py::test_developer_build_checks runs it in one child process with RTP_AMD_LIB
pointing at the developer library."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_bindings as ob
import rtp_bindings as rb

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_same_frame(got, want, what):
    same = (bits(got) == bits(want)).all(axis=-1)
    assert same.all(), f"{what}: {(~same).sum()} of {same.size} pixels differ, max abs diff {np.abs(got - want).max()}"


@pytest.fixture(scope="module")
def config_scene(test_config_text):
    host = rb.HostScene.from_config(test_config_text)
    return host, rb.DeviceScene(host, device=0)


def test_this_is_the_developer_library():
    v = rb.amd_lib().rt_version_string()
    assert b"dev=1" in v and b"parity=1" in v


def test_wavefront_kernel_gives_the_same_frames():
    """rt_config.kernel = RT_KERNEL_WAVEFRONT (rt_kernel_wf.hip.inc: a wave owns a pool of paths in wave-private L2-resident
    stacks and alternates dense SHADE / GENERATE / EXCHANGE / TRACE steps) — same bits as the oracle: S-rtiow at several
    pool sizes and exchange thresholds, a frame smaller than one wave's pool, the config scene (planes, lights,
    absorbing glass), a far camera (far-origin flags) and a 2-entry traversal stack (many flagged samples)."""
    host = rb.HostScene.rtiow()
    cam = rb.rtiow_camera(200, 120, 12, 50)
    want = ob.render(host, cam, threads=8)
    for paths, exch in ((0, 0), (128, 4), (512, 32), (192, 64)):
        dev = rb.DeviceScene(host, device=0, honour_env=False, kernel=rb.KERNEL_WAVEFRONT, wavefront_paths=paths, wavefront_exchange=exch)
        fb, t = dev.render_to_host(cam)
        assert t.kernel == rb.KERNEL_WAVEFRONT and t.guarded == 1
        assert_same_frame(fb, want, f"wavefront kernel, pool {paths}, exchange {exch}")
    dev = rb.DeviceScene(host, device=0, honour_env=False, kernel=rb.KERNEL_WAVEFRONT)
    tiny = rb.rtiow_camera(7, 5, 3, 50)
    fb, t = dev.render_to_host(tiny)
    assert_same_frame(fb, ob.render(host, tiny, threads=4), "wavefront kernel, 105 samples in all")
    far = rb.make_camera(160, 90, 3.0, (400.0, 90.0, 60.0), (0, 0, 0), (0.7, 0.8, 1.0), 4, 50)
    dev.configure(guard_repack=0, stack_levels=2, guard_keep=1)
    fb, t = dev.render_to_host(far)
    assert t.kernel == rb.KERNEL_WAVEFRONT and t.flagged_samples > 1000
    assert_same_frame(fb, ob.render(host, far, threads=8), "wavefront kernel, far camera, 2-entry stack")


def test_wavefront_kernel_on_the_config_scene(config_scene):
    host, _ = config_scene
    dev = rb.DeviceScene(host, device=0, honour_env=False, kernel=rb.KERNEL_WAVEFRONT, traversal=rb.TRAVERSAL_GUARDED, guard_keep=1)
    cam = host.frame_camera(0)
    fb, t = dev.render_to_host(cam)
    assert t.kernel == rb.KERNEL_WAVEFRONT and t.guarded == 1
    assert_same_frame(fb, ob.render(host, cam, threads=8), "wavefront kernel, config scene")


def test_fast_reciprocal_and_sqrt_match_ieee_for_every_float():
    """rt_device_math.h recip() / sqrt_cr(): a hardware estimate plus one fused correction inside an exponent fence, the
    compiler's correctly rounded sequence outside it.  Proof by exhaustion on the device that renders: all 2^32 binary32
    inputs, every result bit compared with 1.0f / x and sqrtf(x) (the reference's own operations, include/vec3.h:97,105)."""
    import ctypes as C
    lib = rb.amd_lib()
    out = (C.c_uint64 * 3)()
    lib.rt_debug_check_fast_math.argtypes = [C.POINTER(C.c_uint64)]
    lib.rt_debug_check_fast_math.restype = C.c_int
    assert lib.rt_debug_check_fast_math(out) == 0
    assert out[2] == 2 ** 32
    assert out[0] == 0, "recip() differs from 1.0f / x for %d inputs" % out[0]
    assert out[1] == 0, "sqrt_cr() differs from sqrtf(x) for %d inputs" % out[1]


def test_sphere_roots_from_one_reciprocal_match_the_plain_divisions():
    """test_sphere's root selection (both fp64 quotients from one v_rcp_f64 + Newton steps, no scaling instructions) against
    the reference's form with the compiler's correctly rounded divisions, on 2^32 SAMPLED operand sets (not an exhaustive proof: four
    operands span 2^128 combinations): raw random bit patterns
    (all exponents, inf, NaN, denormals) and scene-scale operands alike — same acceptance, same accepted root, bit for bit."""
    import ctypes as C
    lib = rb.amd_lib()
    out = (C.c_uint64 * 3)()
    lib.rt_debug_check_sphere_roots.argtypes = [C.c_uint64, C.POINTER(C.c_uint64)]
    lib.rt_debug_check_sphere_roots.restype = C.c_int
    assert lib.rt_debug_check_sphere_roots(2 ** 32, out) == 0
    assert out[2] == 2 ** 32
    assert out[1] > 2 ** 26          # accepted roots are really being produced and compared
    assert out[0] == 0, "%d operand sets differ" % out[0]


def test_tripwire_turns_a_scheduling_fault_into_an_error():
    """The developer build checks, at every step-kind vote, that each live lane is walking, stuck at a leaf or finished, and counts
    main-loop rounds without progress (rt_kernel.hip.inc, RTP_TRIPWIRE).  rt_debug_trip_test injects the fault round 3's hang
    came from — a lane at a leaf with its park slot empty: the launch ENDS, rt_last_timing reports RT_ERR_HIP with the tripwire's
    code, and the next frame of the same handle is the oracle's again."""
    lib = rb.amd_lib()
    lib.rt_debug_trip_test.argtypes = [C.c_void_p, C.c_uint32]
    host = rb.HostScene.rtiow()
    dev = rb.DeviceScene(host, device=0, honour_env=False, traversal=rb.TRAVERSAL_GUARDED, guard_keep=1)
    cam = rb.rtiow_camera(160, 90, 8, 50)
    want = ob.render(host, cam, threads=8)
    fb, t = dev.render_to_host(cam)
    assert t.guarded == 1
    assert_same_frame(fb, want, "developer build, tripwire armed, no fault")
    assert lib.rt_debug_trip_test(dev._h, 1) == 0
    with pytest.raises(rb.RtError, match="aborted.*1414678785"):          # 0x54524901: kTripPartition
        dev.render_to_host(cam)
    assert lib.rt_debug_trip_test(dev._h, 0) == 0
    fb, t = dev.render_to_host(cam)
    assert_same_frame(fb, want, "frame after the tripped one")


def test_wide_nodes_give_the_same_frames(config_scene):
    """rt_config.wide_nodes = 1: the guarded walk on the 4-wide collapse of its tree (step_wide; SURVEY.md §8(f3) "wide
    nodes") — LDS-resident tables (S-rtiow, the config scene with its planes), tables read through L1/L2 with a treelet
    in LDS (6 500 spheres), with distance-aware margins and a 3-entry stack (overflow flags): the oracle's frames."""
    host = rb.HostScene.rtiow()
    cam = rb.rtiow_camera(240, 135, 8, 50)
    want = ob.render(host, cam, threads=8)
    dev = rb.DeviceScene(host, device=0, honour_env=False, wide_nodes=1, traversal=rb.TRAVERSAL_GUARDED, guard_keep=1)
    fb, t = dev.render_to_host(cam)
    assert t.guarded == 1 and t.wide_nodes == 1 and t.scene_in_lds == 1
    assert_same_frame(fb, want, "S-rtiow, wide nodes")
    dev.configure(stack_levels=3, guard_keep=1)
    fb, t3 = dev.render_to_host(cam)
    assert t3.wide_nodes == 1 and t3.flagged_samples > t.flagged_samples
    assert_same_frame(fb, want, "S-rtiow, wide nodes, 3-entry stack")
    dev = rb.DeviceScene(host, device=0, honour_env=False, wide_nodes=1, guard_dynamic_margins=2, traversal=rb.TRAVERSAL_GUARDED, guard_keep=1)
    fb, t = dev.render_to_host(cam)
    assert t.wide_nodes == 1 and t.guard_dynamic == 1
    assert_same_frame(fb, want, "S-rtiow, wide nodes + distance-aware margins")
    chost, _ = config_scene
    dev = rb.DeviceScene(chost, device=0, honour_env=False, wide_nodes=1, traversal=rb.TRAVERSAL_GUARDED, guard_keep=1)
    ccam = chost.frame_camera(0)
    fb, t = dev.render_to_host(ccam)
    assert t.guarded == 1 and t.wide_nodes == 1
    assert_same_frame(fb, ob.render(chost, ccam, threads=8), "config scene, wide nodes")
    big = rb.HostScene.rtiow(half_extent=40)
    dev = rb.DeviceScene(big, device=0, honour_env=False, wide_nodes=1, traversal=rb.TRAVERSAL_GUARDED, guard_keep=1)
    bcam = rb.rtiow_camera(320, 180, 4, 50)
    fb, t = dev.render_to_host(bcam)
    assert t.guarded == 1 and t.wide_nodes == 1 and t.scene_in_lds == 0
    assert_same_frame(fb, ob.render(big, bcam, threads=8), "6 500 spheres through L1/L2, wide nodes")
