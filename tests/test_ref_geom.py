"""The oracle's geometry against the REFERENCE'S OWN HEADERS, compiled from where they lie (oracle/ref_geom.cpp →
oracle/_ref/libref_geom.so, build container only): vec3 / ray / interval / aabb / hittable_object / sphere / plane / bvh /
bvh_builder — everything of the hot path that compiles in this image without a stand-in.  Bit for bit, on crafted extremes
(zeros, ±inf reciprocals, NaN planes, denormals, exact interval ends) and random values.

Where the library is absent (the GPU box, if it did not travel) the same comparison runs against tests/golden/ref_geom.npz,
outputs of that library recorded by tests/golden/make_ref_geom_golden.py.
Not covered, because it does not compile here (<curand_kernel.h>, <cuda_runtime.h>): random_utils.h, materials.h, camera.cuh —
the RNG, the materials and the camera stay pinned by SURVEY-session records only (tests/test_oracle_pins.py).
"""
import os

import numpy as np
import pytest

import ref_geom_cases as rg

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden", "ref_geom.npz")
have_ref = os.path.exists(rg.REF_LIB)


def compare(ref, orc, what):
    for key in ref:
        a, b = ref[key], orc[key]
        if key.startswith("rec_") or key in ("rec",) or key.startswith("code"):
            hit = ref["hit" + key[key.index("_"):]] if "_" in key else ref["hit"]
            a, b = a[hit != 0], b[hit != 0]           # records exist for hits only
        d = rg.differing(a, b)
        assert not d.any(), f"{what}: '{key}' differs in {int(d.sum())} of {d.size} items, first at {int(np.argmax(d))}: {a[np.argmax(d)]} vs {b[np.argmax(d)]}"


@pytest.mark.skipif(not have_ref, reason="oracle/_ref/libref_geom.so is built only where /root/reference exists")
def test_struct_sizes_match_the_survey():
    assert list(rg.Ref().sizes()) == [12, 24, 8, 24, 44, 32, 80, 36]          # vec3 Ray Interval AABB HitRecord SphereData PlaneData BVHNode


@pytest.mark.skipif(not have_ref, reason="oracle/_ref/libref_geom.so is built only where /root/reference exists")
def test_primitives_against_the_reference_headers():
    """A million inputs through AABB::hit, operator/, unit_vector, reflect, refract, near_zero, dot, cross, len, contains, Ray::at,
    set_face_normal; a quarter of a million rays against a sphere / a plane each (hit_sphere + get_sphere_uv, the PlaneData
    constructor, hit_plane + is_interior_*): the oracle's restatements give the reference's bits."""
    rng = np.random.default_rng(20260)
    ref, orc = rg.Ref(), rg.Orc()
    c = rg.primitive_cases(rng, 250_000)
    r, o = ref.primitives(c), orc.primitives(c)
    assert r["aabb_hit"].sum() > 1000 and r["hit_sphere"].sum() > 50_000 and r["hit_plane"].sum() > 10_000       # the cases do hit things
    assert r["near_zero"].sum() > 1000 and (r["contains"] == 0).sum() > 1000
    compare(r, o, "primitives")
    # the cheap operators on three more batches: a million items in all
    for seed in (1, 2, 3):
        c = rg.primitive_cases(np.random.default_rng(seed), 250_000)
        keep = ("aabb_hit", "div", "unit", "reflect", "refract", "near_zero", "dot", "cross", "len", "contains", "ray_at", "face_normal", "front")
        r, o = ref.primitives(c), orc.primitives(c)
        compare({k: r[k] for k in keep}, o, f"batch {seed}")


@pytest.mark.skipif(not have_ref, reason="oracle/_ref/libref_geom.so is built only where /root/reference exists")
def test_bvh_build_and_traversal_against_the_reference_headers():
    """build_bvh (include/bvh_builder.h) against the host mirror's builder — node for node, boxes bit for bit — and hit_bvh
    (include/bvh.h) against the oracle's on 60 000 rays per scene: sphere scenes, mixed scenes with thin axis-aligned quads, one
    primitive, an empty scene.  (The reference's child order comes from an out-of-bounds read; it only matters on exact ties,
    which random scenes do not have: the comparison is strict.)"""
    rng = np.random.default_rng(4711)
    ref, orc = rg.Ref(), rg.Orc()
    total_hits = 0
    for ns, npl in ((1, 0), (0, 1), (2, 0), (37, 0), (500, 0), (0, 40), (120, 60), (1500, 200), (5000, 0)):
        sph, pl, types, o, d = rg.scene_cases(rng, ns, npl, 60_000)
        r, q = ref.scene(sph, pl, types, o, d), orc.scene(sph, pl, types, o, d)
        assert r["nodes"].shape == (2 * (ns + npl) - 1, 9)
        compare(r, q, f"{ns} spheres, {npl} planes")
        total_hits += int(r["hit"].sum())
    assert total_hits > 100_000


def test_oracle_against_the_recorded_reference_outputs():
    """The same comparison against outputs of the reference's headers recorded in tests/golden/ref_geom.npz (1 024 primitive cases,
    two scenes): runs everywhere, also where the reference itself is not."""
    g = np.load(GOLDEN)
    orc = rg.Orc()
    c = {k[3:]: g[k] for k in g.files if k.startswith("in_")}
    o = orc.primitives(c)
    compare({k[4:]: g[k] for k in g.files if k.startswith("out_")}, o, "recorded primitives")
    for tag in ("s0", "s1"):
        q = orc.scene(g[f"{tag}_sph"], g[f"{tag}_pl"], g[f"{tag}_types"], g[f"{tag}_o"], g[f"{tag}_d"])
        compare({k: g[f"{tag}_{k}"] for k in ("nodes", "hit", "rec", "code")}, q, f"recorded scene {tag}")
