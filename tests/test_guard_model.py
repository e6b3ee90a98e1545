"""CPU model of the guarded near-first walk (docs/LOG.md §3b) against the oracle's hit_bvh.

tools/nearfirst_study.c restates the scheme in plain C on top of the oracle's own primitive tests:
SAH tree over leaf boxes inflated by the per-sphere margin, near-first walk with the fused box
test, the far-origin test, and the three flags (inconsistent final hit, exact tie, full stack).
For every ray the oracle's path tracer casts it compares the walk's result with hit_bvh's:
a ray whose results differ must have been flagged.  (The GPU kernel itself is checked against the
oracle in test_gpu_parity.py; this keeps the ARGUMENT under test where there is no GPU.)
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ray-tracing-practice_amd")


@pytest.fixture(scope="module")
def study_binary(tmp_path_factory):
    subprocess.run(["make", "-C", PKG, "librtp_host.so"], check=True, capture_output=True)
    out = str(tmp_path_factory.mktemp("guard_model") / "nf_study")
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-std=gnu11", "-w", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tools", "nearfirst_study.c"), "-L" + PKG, "-lrtp_host", "-Wl,-rpath," + PKG, "-lm", "-lpthread", "-o", out]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return out


@pytest.mark.parametrize("half_extent,width,height,spp,levels,dyn", [(11, 192, 108, 6, 5, False), (40, 200, 112, 4, 8, False),
                                                                     (40, 200, 112, 4, 8, True), (100, 192, 108, 3, 12, True)])
def test_every_order_sensitive_ray_is_flagged(study_binary, half_extent, width, height, spp, levels, dyn):
    """dyn: distance-aware margins (leaf boxes carry the rounding floor only; every box test grows its box by
    k x (distance from the ray origin to the farthest corner)^2) with the PROVEN discriminant budget gamma = 24 ulp."""
    env = dict(os.environ, FUSED="1", GAMMA_ULPS="24")
    if dyn:
        env["DYN"] = "1"
    res = subprocess.run([study_binary, str(half_extent), str(width), str(height), str(spp), "9.5e-7", str(levels)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    text = res.stdout
    rays = int(re.search(r"rays (\d+)", text).group(1))
    m = re.search(r"flagged ([0-9.]+)% \(ties (\d+), inconsistent final (\d+), overflow (\d+)\)\s+mismatches (\d+) \(unflagged (\d+)\)", text)
    assert m, text[-2000:]
    flagged_pct, mismatches, unflagged = float(m.group(1)), int(m.group(5)), int(m.group(6))
    dep = float(re.search(r"max departure/eps ([0-9.eE+-]+)", text).group(1))
    steps = float(re.search(r"pair steps ([0-9.]+)", text).group(1))
    visits = float(re.search(r"visits/ray ref ([0-9.]+)", text).group(1))
    assert rays > 200000
    assert unflagged == 0, f"{unflagged} rays differ from hit_bvh without having been flagged"
    assert flagged_pct < 1.0                     # the exact re-walk stays a small share
    if not dyn:
        assert dep < 0.5                         # computed hits stay well inside the (statically) inflated boxes
    assert steps * 2 < visits                    # and the walk does pay: under half the box tests
    assert "UNFLAGGED MISMATCH" not in res.stderr


@pytest.mark.parametrize("half_extent,width,height,spp,dyn", [(11, 192, 108, 6, False), (40, 200, 112, 4, True), (100, 192, 108, 3, True)])
def test_front_primitives_and_parametric_growth(study_binary, half_extent, width, height, spp, dyn):
    """Round 4, the model of what the kernels do now (MODEL=1): the scene-spanning primitive (the ground sphere) is tested
    before the walk and is not a leaf of the tree (TOPBIG); with distance-aware margins the boxes — binary16, rounded outward —
    grow by the PARAMETRIC rule of step_pair_par (GROW=4): from the exit parameter of the box a node was entered through.  Every
    ray the oracle casts: a result that differs from hit_bvh's must have been flagged, and on every accepted hit of a small sphere
    the growth its leaf's box was tested with covers k |o - c|^2 — the induction hypothesis the rule rests on."""
    env = dict(os.environ, GAMMA_ULPS="24", MODEL="1", TOPBIG="1", GROW="4" if dyn else "0")
    if dyn:
        env.update(DYN="1", HALF16="1")
    res = subprocess.run([study_binary, str(half_extent), str(width), str(height), str(spp)], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    m = re.search(r"MODEL: pair steps/ray ([0-9.]+)\s+leaf tests/ray ([0-9.]+) \(\+ ([0-9.]+) before the walk\).*mismatches (\d+) \(unflagged (\d+)\)\s+"
                  r"growth bound checked on (\d+) hits, broken (\d+)", res.stdout)
    assert m, res.stdout[-2000:]
    pairs, front, unflagged, checked, broken = float(m.group(1)), float(m.group(3)), int(m.group(5)), int(m.group(6)), int(m.group(7))
    assert front == 1.0 and unflagged == 0 and broken == 0
    assert (checked > 50000) == dyn
    old = float(re.search(r"SAH tree, near-first.*\(pair steps ([0-9.]+)\)", res.stdout).group(1))      # the walk with every primitive a leaf, corner growth
    assert pairs < old
