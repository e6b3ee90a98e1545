"""Inputs and drivers shared by tests/test_ref_geom.py and tests/golden/make_ref_geom_golden.py.

`Ref` wraps oracle/_ref/libref_geom.so — the reference's OWN headers (vec3, ray, interval, aabb, hittable_object, sphere, plane,
bvh, bvh_builder) compiled from where they lie, build container only — and `Orc` the oracle's batched views of its restatements
(oracle/rt_oracle.c, orc_geom_*).  Both take the same arrays and return dictionaries of numpy arrays with the same keys, so a
comparison is `same_bits(ref[k], orc[k])` key by key.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libref_geom.so")

SPECIALS = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 2.0, 1e-4, 1e-8, 9.9e-9, 1e-30, 1e-38, 1.4e-45, -1.4e-45, 1e30, 1e32, 3.4e38, np.inf, -np.inf, np.nan,
                     0.001, 0.0010000001, 0.25, 0.999, 1.0000001], dtype=np.float32)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def differing(a, b):
    """Per item: does any element differ?  Floats are compared bit for bit (so -0 is not +0), except that a NaN equals a NaN: which
    payload and sign a NaN carries depends on the operand order a compiler picks for a commutative operation, and nothing in the
    renderer can tell NaNs apart (every comparison with one is false)."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.shape != b.shape:
        return np.ones(max(a.shape[0], 1), bool)
    if a.dtype == np.float32:
        d = (a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))
    else:
        d = a != b
    return d.reshape(a.shape[0], -1).any(axis=1) if a.ndim else np.array([bool(d)])


def same_bits(a, b):
    return not differing(a, b).any()


def vectors(rng, n, scale=10.0, special_share=0.15):
    """n x 3 floats: scene-scale values with zeros, infinities, NaNs, denormals and huge values mixed in."""
    v = rng.normal(0.0, scale, (n, 3)).astype(np.float32)
    v *= (10.0 ** rng.uniform(-3, 3, (n, 1))).astype(np.float32)
    mask = rng.random((n, 3)) < special_share
    v[mask] = rng.choice(SPECIALS, int(mask.sum()))
    return v


def scalars(rng, n, special_share=0.2):
    s = (rng.normal(0.0, 1.0, n) * 10.0 ** rng.uniform(-4, 4, n)).astype(np.float32)
    mask = rng.random(n) < special_share
    s[mask] = rng.choice(SPECIALS, int(mask.sum()))
    return s


def primitive_cases(rng, n):
    """Inputs of the per-primitive functions: crafted extremes + random values (the same arrays for both libraries)."""
    c = {}
    lo = vectors(rng, n, 5.0, 0.05)
    ext = np.abs(vectors(rng, n, 3.0, 0.05))
    ext[rng.random(n) < 0.1] = 0.0                          # flat boxes
    hi = lo + ext
    boxes = np.stack([lo[:, 0], hi[:, 0], lo[:, 1], hi[:, 1], lo[:, 2], hi[:, 2]], axis=1)
    boxes[rng.random(n) < 0.03] = np.nan                    # NaN planes
    c["boxes"] = f32(boxes)
    c["origins"] = vectors(rng, n, 8.0, 0.05)
    d = vectors(rng, n, 1.0, 0.0)
    axis = rng.random((n, 3)) < 0.15                        # rays parallel to an axis: 1 / 0 = +-inf reciprocals
    d[axis] = rng.choice(np.array([0.0, -0.0], np.float32), int(axis.sum()))
    c["dirs"] = f32(d)
    c["tmin"] = np.where(rng.random(n) < 0.8, np.float32(0.001), scalars(rng, n)).astype(np.float32)
    c["tmax"] = np.where(rng.random(n) < 0.5, np.float32(1e30), np.abs(scalars(rng, n))).astype(np.float32)
    c["a"] = vectors(rng, n)
    c["b"] = vectors(rng, n)
    c["t"] = scalars(rng, n)
    nrm = vectors(rng, n, 1.0, 0.02)
    c["unit_n"] = f32(nrm / np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30))
    c["eta"] = np.where(rng.random(n) < 0.5, np.float32(1.0 / 1.5), np.abs(scalars(rng, n, 0.05))).astype(np.float32)
    c["tiny"] = (vectors(rng, n, 1.0, 0.3) * np.float32(1e-8)).astype(np.float32)
    c["lo"], c["hi"], c["x"] = scalars(rng, n), scalars(rng, n), scalars(rng, n)
    tie = rng.random(n) < 0.2                               # value exactly on an end of the interval
    c["x"][tie] = np.where(rng.random(int(tie.sum())) < 0.5, c["lo"][tie], c["hi"][tie])
    # spheres / planes placed so that a good share of the rays hits them
    centre = c["origins"] + c["dirs"] * np.abs(rng.normal(3.0, 2.0, (n, 1))).astype(np.float32) + rng.normal(0, 0.3, (n, 3)).astype(np.float32)
    radius = np.abs(rng.normal(0.5, 0.5, n)).astype(np.float32) + np.float32(0.01)
    inside = rng.random(n) < 0.1                            # ray origin inside the sphere
    centre[inside] = c["origins"][inside]
    radius[inside] = np.float32(5.0)
    c["spheres"] = f32(np.concatenate([centre, radius[:, None]], axis=1))
    pu, pv = vectors(rng, n, 2.0, 0.0), vectors(rng, n, 2.0, 0.0)
    c["plane_base"] = f32(centre - 0.5 * (pu + pv))
    c["plane_u"], c["plane_v"] = f32(pu), f32(pv)
    c["plane_type"] = i32(rng.integers(0, 3, n))
    return c


def scene_cases(rng, n_spheres, n_planes, n_rays, spread=6.0):
    sph = np.zeros((n_spheres, 4), np.float32)
    sph[:, :3] = rng.uniform(-spread, spread, (n_spheres, 3))
    sph[:, 3] = 10.0 ** rng.uniform(-1.5, 0.2, n_spheres)
    pl = np.zeros((n_planes, 9), np.float32)
    pl[:, :3] = rng.uniform(-spread, spread, (n_planes, 3))
    pl[:, 3:9] = rng.uniform(-2.5, 2.5, (n_planes, 6))
    axis = rng.random(n_planes) < 0.3                       # axis-aligned quads: thin boxes, widened by expand_to_min / pad
    for k in np.nonzero(axis)[0]:
        a = int(rng.integers(0, 3))
        pl[k, 3:9] = 0
        pl[k, 3 + (a + 1) % 3] = rng.uniform(1, 3)
        pl[k, 6 + (a + 2) % 3] = rng.uniform(1, 3)
    types = i32(rng.integers(0, 3, n_planes))
    o = rng.uniform(-1.5 * spread, 1.5 * spread, (n_rays, 3)).astype(np.float32)
    target = rng.uniform(-spread, spread, (n_rays, 3)).astype(np.float32)
    d = (target - o).astype(np.float32)
    return f32(sph), f32(pl), types, f32(o), f32(d)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Ref:
    """The reference's own headers (oracle/_ref/libref_geom.so)."""

    def __init__(self, path=REF_LIB):
        self.lib = C.CDLL(path)
        self.lib.ref_build_bvh.restype = C.c_int32

    def sizes(self):
        out = np.zeros(8, np.int32)
        self.lib.ref_sizes(_p(out))
        return out

    def primitives(self, c):
        n = c["a"].shape[0]
        L, N, r = self.lib, C.c_int64(c["a"].shape[0]), {}
        r["aabb_hit"] = np.zeros(n, np.int32)
        L.ref_aabb_hit(N, _p(c["boxes"]), _p(c["origins"]), _p(c["dirs"]), _p(c["tmin"]), _p(c["tmax"]), _p(r["aabb_hit"]))
        r["div"] = np.zeros((n, 3), np.float32); L.ref_vec3_div(N, _p(c["a"]), _p(c["t"]), _p(r["div"]))
        r["unit"] = np.zeros((n, 3), np.float32); L.ref_unit_vector(N, _p(c["a"]), _p(r["unit"]))
        r["reflect"] = np.zeros((n, 3), np.float32); L.ref_reflect(N, _p(c["a"]), _p(c["unit_n"]), _p(r["reflect"]))
        r["refract"] = np.zeros((n, 3), np.float32); L.ref_refract(N, _p(c["unit_n"]), _p(c["b"]), _p(c["eta"]), _p(r["refract"]))
        r["near_zero"] = np.zeros(n, np.int32); L.ref_near_zero(N, _p(c["tiny"]), _p(r["near_zero"]))
        r["dot"], r["cross"], r["len"] = np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
        L.ref_dot_cross_len(N, _p(c["a"]), _p(c["b"]), _p(r["dot"]), _p(r["cross"]), _p(r["len"]))
        flags, clamp, expand = np.zeros(n, np.int32), np.zeros(n, np.float32), np.zeros((n, 2), np.float32)
        L.ref_interval(N, _p(c["lo"]), _p(c["hi"]), _p(c["x"]), _p(flags), _p(clamp), _p(expand))
        r["contains"] = flags & 1
        r["ray_at"] = np.zeros((n, 3), np.float32); L.ref_ray_at(N, _p(c["origins"]), _p(c["dirs"]), _p(c["t"]), _p(r["ray_at"]))
        r["face_normal"], r["front"] = np.zeros((n, 3), np.float32), np.zeros(n, np.int32)
        L.ref_set_face_normal(N, _p(c["dirs"]), _p(c["a"]), _p(r["face_normal"]), _p(r["front"]))
        for kind in ("sphere", "plane"):
            hit, rec, code = np.zeros(n, np.int32), np.zeros((n, 9), np.float32), np.zeros(n, np.int32)
            if kind == "sphere":
                L.ref_hit_sphere(N, _p(c["origins"]), _p(c["dirs"]), _p(c["tmin"]), _p(c["tmax"]), _p(c["spheres"]), _p(hit), _p(rec), _p(code))
            else:
                L.ref_hit_plane(N, _p(c["origins"]), _p(c["dirs"]), _p(c["tmin"]), _p(c["tmax"]), _p(c["plane_base"]), _p(c["plane_u"]), _p(c["plane_v"]),
                                _p(c["plane_type"]), _p(hit), _p(rec), _p(code))
            r[f"hit_{kind}"], r[f"rec_{kind}"], r[f"code_{kind}"] = hit, rec, code
        r["plane_data"] = np.zeros((n, 72), np.uint8)
        L.ref_plane_make(N, _p(c["plane_base"]), _p(c["plane_u"]), _p(c["plane_v"]), _p(c["plane_type"]), _p(r["plane_data"]))
        return r

    def boxes(self, p, q, a, b):
        n = p.shape[0]
        r = {"from_points": np.zeros((n, 6), np.float32), "surround": np.zeros((n, 6), np.float32)}
        self.lib.ref_aabb_from_points(C.c_int64(n), _p(p), _p(q), _p(r["from_points"]))
        self.lib.ref_aabb_surround(C.c_int64(n), _p(a), _p(b), _p(r["surround"]))
        return r

    def scene(self, sph, pl, types, o, d):
        ns, npl, n = sph.shape[0], pl.shape[0], o.shape[0]
        nodes = np.zeros((max(2 * (ns + npl), 1), 9), np.int32)
        count = self.lib.ref_build_bvh(ns, _p(sph), npl, _p(pl), _p(types), _p(nodes), nodes.shape[0])
        hit, rec, code = np.zeros(n, np.int32), np.zeros((n, 9), np.float32), np.zeros(n, np.int32)
        self.lib.ref_hit_bvh(ns, _p(sph), npl, _p(pl), _p(types), C.c_int64(n), _p(o), _p(d), C.c_float(0.001), C.c_float(1e30), _p(hit), _p(rec), _p(code))
        return {"nodes": nodes[:count].copy(), "hit": hit, "rec": rec, "code": code}


class Orc:
    """The oracle's restatements (oracle/librt_oracle.so) + the host mirror's scene / BVH builders (librtp_host.so)."""

    def __init__(self):
        import oracle_bindings as ob
        import rtp_bindings as rb
        self.lib, self.rb = ob.lib(), rb

    def _host_scene(self, spheres4, base, u, v, types):
        """Host mirror arrays (rt_sphere / rt_plane records) for item-wise spheres and planes; material index = k & 0xffff."""
        rb = self.rb
        ns, npl = spheres4.shape[0], base.shape[0]
        sp = np.zeros((ns, 5), np.float32)
        sp[:, :4] = spheres4
        sp[:, 4] = 0
        pl = np.zeros((npl, 11), np.float32)
        pl[:, 0:3], pl[:, 3:6], pl[:, 6:9] = base, u, v
        pl[:, 9] = 0
        pl[:, 10] = types
        return rb.HostScene.from_arrays(sp, pl, [rb.Material()])

    def primitives(self, c):
        n = c["a"].shape[0]
        L, N, r = self.lib, C.c_int64(c["a"].shape[0]), {}
        r["aabb_hit"] = np.zeros(n, np.int32)
        L.orc_geom_aabb_hit(N, _p(c["boxes"]), _p(c["origins"]), _p(c["dirs"]), _p(c["tmin"]), _p(c["tmax"]), _p(r["aabb_hit"]))
        r["div"] = np.zeros((n, 3), np.float32); L.orc_geom_vec3_div(N, _p(c["a"]), _p(c["t"]), _p(r["div"]))
        r["unit"] = np.zeros((n, 3), np.float32); L.orc_geom_unit_vector(N, _p(c["a"]), _p(r["unit"]))
        r["reflect"] = np.zeros((n, 3), np.float32); L.orc_geom_reflect(N, _p(c["a"]), _p(c["unit_n"]), _p(r["reflect"]))
        r["refract"] = np.zeros((n, 3), np.float32); L.orc_geom_refract(N, _p(c["unit_n"]), _p(c["b"]), _p(c["eta"]), _p(r["refract"]))
        r["near_zero"] = np.zeros(n, np.int32); L.orc_geom_near_zero(N, _p(c["tiny"]), _p(r["near_zero"]))
        r["dot"], r["cross"], r["len"] = np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
        L.orc_geom_dot_cross_len(N, _p(c["a"]), _p(c["b"]), _p(r["dot"]), _p(r["cross"]), _p(r["len"]))
        r["contains"] = np.zeros(n, np.int32); L.orc_geom_contains(N, _p(c["lo"]), _p(c["hi"]), _p(c["x"]), _p(r["contains"]))
        r["ray_at"] = np.zeros((n, 3), np.float32); L.orc_geom_ray_at(N, _p(c["origins"]), _p(c["dirs"]), _p(c["t"]), _p(r["ray_at"]))
        r["face_normal"], r["front"] = np.zeros((n, 3), np.float32), np.zeros(n, np.int32)
        L.orc_geom_set_face_normal(N, _p(c["dirs"]), _p(c["a"]), _p(r["face_normal"]), _p(r["front"]))
        host = self._host_scene(c["spheres"], c["plane_base"], c["plane_u"], c["plane_v"], c["plane_type"])
        # the records the host mirror made: material index = position, as the reference harness numbers them
        sph = np.ctypeslib.as_array(C.cast(host.desc.spheres, C.POINTER(C.c_int32)), shape=(n, 8))
        sph[:, 4] = np.arange(n) & 0xffff
        pln = np.ctypeslib.as_array(C.cast(host.desc.planes, C.POINTER(C.c_int32)), shape=(n, 20))
        pln[:, 2] = np.arange(n) & 0xffff
        for kind, recs in (("sphere", host.desc.spheres), ("plane", host.desc.planes)):
            hit, rec, code = np.zeros(n, np.int32), np.zeros((n, 9), np.float32), np.zeros(n, np.int32)
            getattr(L, f"orc_geom_hit_{kind}")(N, _p(c["origins"]), _p(c["dirs"]), _p(c["tmin"]), _p(c["tmax"]), recs, _p(hit), _p(rec), _p(code))
            r[f"hit_{kind}"], r[f"rec_{kind}"], r[f"code_{kind}"] = hit, rec, code
        r["plane_data"] = np.ctypeslib.as_array(C.cast(host.desc.planes, C.POINTER(C.c_uint8)), shape=(n, 80))[:, :72].copy()
        self._keep = host
        return r

    def scene(self, sph, pl, types, o, d):
        rb = self.rb
        ns, npl, n = sph.shape[0], pl.shape[0], o.shape[0]
        host = self._host_scene(sph, pl[:, 0:3], pl[:, 3:6], pl[:, 6:9], types)          # host mirror: PlaneData ctor + build_bvh
        nodes = host.nodes_array()
        hit, rec, code = np.zeros(n, np.int32), np.zeros((n, 9), np.float32), np.zeros(n, np.int32)
        self.lib.orc_geom_hit_bvh(C.byref(host.desc), C.c_int64(n), _p(o), _p(d), C.c_float(0.001), C.c_float(1e30), _p(hit), _p(rec), _p(code))
        return {"nodes": nodes, "hit": hit, "rec": rec, "code": code}
