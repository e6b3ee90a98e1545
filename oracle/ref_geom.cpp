/*
 * ref_geom.cpp — TEST INFRASTRUCTURE.  Harness around the part of the reference's OWN hot-path headers that compiles in this
 * image without any stand-in: include/vec3.h, ray.h, interval.h, aabb.h, hittable_object.h, sphere.h, plane.h, bvh.h and
 * bvh_builder.h, #included from where they lie under /root/reference (never copied into this repo) and compiled by
 * `hipcc -x hip --cuda-host-only` (the headers say __host__ __device__; only the host side is built and run).  The one thing
 * they need beyond the C++ library — `__align__(16)` on SphereData / PlaneData, which nvcc's implicit cuda_runtime.h supplies —
 * comes from ROCm's own <hip/hip_runtime.h> (amd_hip_runtime.h defines it), included below: a header this image HAS, nothing
 * written for the purpose.  random_utils.h and materials.h (#include <curand_kernel.h>) and camera.cuh (<cuda_runtime.h>) do not
 * compile here and are NOT part of this: the RNG, the materials and the camera stay pinned by SURVEY-session records only
 * (DESIGN.md §2).
 *
 * What it exposes, in batches (n items per call, plain C arrays): AABB::hit (include/aabb.h:42-65), the AABB constructors with
 * expand_to_min (:14-33, :92-97), vec3 operator/ and unit_vector (include/vec3.h:97,105), reflect / refract / near_zero / dot /
 * cross / len (:55-70,99-103), Interval::contains / surrounds / clamp / expand (include/interval.h:16-29), Ray::at
 * (include/ray.h:12), HitRecord::set_face_normal (include/hittable_object.h:17-20), hit_sphere + get_sphere_uv
 * (include/sphere.h:16-53), the PlaneData constructor and hit_plane + is_interior_* (include/plane.h:19-96), build_bvh
 * (include/bvh_builder.h:17-120) and hit_bvh (include/bvh.h:19-65 — compiled with its out-of-bounds `direction()[-1]` as it stands:
 * whatever child order that yields here, results differ from any other order only on exact ties).
 * tests/test_ref_geom.py drives the same inputs through this library and through the oracle's restatements, bit for bit, and
 * tests/golden/make_ref_geom_golden.py records a sample of its outputs as a fixture that travels to the GPU box.
 * Built only where /root/reference exists (oracle/Makefile, target `_ref`); output under oracle/_ref/ (git-ignored).
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <ostream>
#include <vector>

#include "vec3.h"
#include "ray.h"
#include "interval.h"
#include "aabb.h"
#include "hittable_object.h"
#include "sphere.h"
#include "plane.h"
#include "bvh.h"
#include "bvh_builder.h"

namespace {
inline vec3 v(const float *p) { return vec3(p[0], p[1], p[2]); }
inline void put(float *p, const vec3 &a) { p[0] = a[0]; p[1] = a[1]; p[2] = a[2]; }
inline void put_box(float *p, const AABB &b) {
    p[0] = b.x_interval.min; p[1] = b.x_interval.max; p[2] = b.y_interval.min; p[3] = b.y_interval.max;
    p[4] = b.z_interval.min; p[5] = b.z_interval.max;
}
// a box with exactly these planes (no constructor: those widen thin axes)
inline AABB raw_box(const float *p) {
    AABB b;
    b.x_interval = Interval(p[0], p[1]); b.y_interval = Interval(p[2], p[3]); b.z_interval = Interval(p[4], p[5]);
    return b;
}
}  // namespace

extern "C" {

// AABB::hit(const Ray &, Interval): boxes 6 floats each (x.min x.max y.min y.max z.min z.max), rays as origin / direction
void ref_aabb_hit(int64_t n, const float *boxes, const float *origins, const float *dirs, const float *tmin, const float *tmax, int32_t *out) {
    for (int64_t k = 0; k < n; ++k)
        out[k] = raw_box(boxes + 6 * k).hit(Ray(v(origins + 3 * k), v(dirs + 3 * k)), Interval(tmin[k], tmax[k])) ? 1 : 0;
}
// AABB(const point3 &, const point3 &) and AABB(const AABB &, const AABB &), both with expand_to_min; then pad() once more
void ref_aabb_from_points(int64_t n, const float *p, const float *q, float *out_boxes) {
    for (int64_t k = 0; k < n; ++k) { AABB b(v(p + 3 * k), v(q + 3 * k)); b.pad(); put_box(out_boxes + 6 * k, b); }
}
void ref_aabb_surround(int64_t n, const float *a, const float *b, float *out_boxes) {
    for (int64_t k = 0; k < n; ++k) put_box(out_boxes + 6 * k, surround(raw_box(a + 6 * k), raw_box(b + 6 * k)));
}
void ref_vec3_div(int64_t n, const float *a, const float *t, float *out) {
    for (int64_t k = 0; k < n; ++k) put(out + 3 * k, v(a + 3 * k) / t[k]);
}
void ref_unit_vector(int64_t n, const float *a, float *out) {
    for (int64_t k = 0; k < n; ++k) put(out + 3 * k, unit_vector(v(a + 3 * k)));
}
void ref_reflect(int64_t n, const float *a, const float *nrm, float *out) {
    for (int64_t k = 0; k < n; ++k) put(out + 3 * k, v(a + 3 * k).reflect(v(nrm + 3 * k)));
}
void ref_refract(int64_t n, const float *a, const float *nrm, const float *eta, float *out) {
    for (int64_t k = 0; k < n; ++k) put(out + 3 * k, v(a + 3 * k).refract(v(nrm + 3 * k), eta[k]));
}
void ref_near_zero(int64_t n, const float *a, int32_t *out) {
    for (int64_t k = 0; k < n; ++k) out[k] = v(a + 3 * k).near_zero() ? 1 : 0;
}
void ref_dot_cross_len(int64_t n, const float *a, const float *b, float *out_dot, float *out_cross, float *out_len) {
    for (int64_t k = 0; k < n; ++k) {
        out_dot[k] = dot(v(a + 3 * k), v(b + 3 * k));
        put(out_cross + 3 * k, cross(v(a + 3 * k), v(b + 3 * k)));
        out_len[k] = v(a + 3 * k).len();
    }
}
// Interval(lo, hi): contains | surrounds << 1, clamp(x), expand(delta) → (min, max)
void ref_interval(int64_t n, const float *lo, const float *hi, const float *x, int32_t *out_flags, float *out_clamp, float *out_expand) {
    for (int64_t k = 0; k < n; ++k) {
        const Interval i(lo[k], hi[k]);
        out_flags[k] = (i.contains(x[k]) ? 1 : 0) | (i.surrounds(x[k]) ? 2 : 0);
        out_clamp[k] = i.clamp(x[k]);
        const Interval e = i.expand(x[k]);
        out_expand[2 * k] = e.min; out_expand[2 * k + 1] = e.max;
    }
}
void ref_ray_at(int64_t n, const float *origins, const float *dirs, const float *t, float *out) {
    for (int64_t k = 0; k < n; ++k) put(out + 3 * k, Ray(v(origins + 3 * k), v(dirs + 3 * k)).at(t[k]));
}
void ref_set_face_normal(int64_t n, const float *dirs, const float *outward, float *out_normal, int32_t *out_front) {
    for (int64_t k = 0; k < n; ++k) {
        HitRecord rec;
        rec.set_face_normal(Ray(vec3(0, 0, 0), v(dirs + 3 * k)), v(outward + 3 * k));
        put(out_normal + 3 * k, rec.normal);
        out_front[k] = rec.front_face ? 1 : 0;
    }
}
// HitRecord → 9 floats (t, point, normal, u, v) + front_face | material_idx << 1 (u, v only where the reference sets them)
static void put_rec(float *f, int32_t *code, const HitRecord &rec, bool has_uv) {
    f[0] = rec.t; put(f + 1, rec.point); put(f + 4, rec.normal);
    f[7] = has_uv ? rec.u : 0.0f; f[8] = has_uv ? rec.v : 0.0f;
    *code = (rec.front_face ? 1 : 0) | (rec.material_idx << 1);
}
// hit_sphere: spheres as (cx, cy, cz, radius), material_idx = k
void ref_hit_sphere(int64_t n, const float *origins, const float *dirs, const float *tmin, const float *tmax, const float *spheres,
                    int32_t *out_hit, float *out_rec9, int32_t *out_code) {
    for (int64_t k = 0; k < n; ++k) {
        const SphereData s(v(spheres + 4 * k), spheres[4 * k + 3], (int)(k & 0xffff));
        HitRecord rec;
        std::memset(&rec, 0, sizeof(rec));
        out_hit[k] = hit_sphere(Ray(v(origins + 3 * k), v(dirs + 3 * k)), Interval(tmin[k], tmax[k]), rec, s) ? 1 : 0;
        if (out_hit[k]) put_rec(out_rec9 + 9 * k, out_code + k, rec, true);
    }
}
// PlaneData(base, u, v, material, type) as the 72 bytes the reference lays out (type, D, material_idx, w, u, v, base, normal)
void ref_plane_make(int64_t n, const float *base, const float *u, const float *vv, const int32_t *type, uint8_t *out72) {
    for (int64_t k = 0; k < n; ++k) {
        const PlaneData p(v(base + 3 * k), v(u + 3 * k), v(vv + 3 * k), (int)(k & 0xffff), (PlaneType)type[k]);
        std::memcpy(out72 + 72 * k, &p, 72);
    }
}
void ref_hit_plane(int64_t n, const float *origins, const float *dirs, const float *tmin, const float *tmax, const float *base, const float *u,
                   const float *vv, const int32_t *type, int32_t *out_hit, float *out_rec9, int32_t *out_code) {
    for (int64_t k = 0; k < n; ++k) {
        const PlaneData p(v(base + 3 * k), v(u + 3 * k), v(vv + 3 * k), (int)(k & 0xffff), (PlaneType)type[k]);
        HitRecord rec;
        std::memset(&rec, 0, sizeof(rec));
        out_hit[k] = hit_plane(Ray(v(origins + 3 * k), v(dirs + 3 * k)), Interval(tmin[k], tmax[k]), rec, p) ? 1 : 0;
        if (out_hit[k]) put_rec(out_rec9 + 9 * k, out_code + k, rec, true);
    }
}
// A scene: spheres (cx, cy, cz, r) and planes (base, u, v: 9 floats) + types; material_idx of a primitive = 2 * index + type, so
// that a HitRecord names the primitive it came from.
namespace {
struct RefScene {
    std::vector<SphereData> spheres;
    std::vector<PlaneData> planes;
    std::vector<BVHNode> nodes;
};
RefScene make_scene(int32_t ns, const float *spheres, int32_t np, const float *planes, const int32_t *types) {
    RefScene sc;
    for (int32_t k = 0; k < ns; ++k) sc.spheres.emplace_back(v(spheres + 4 * k), spheres[4 * k + 3], 2 * k);
    for (int32_t k = 0; k < np; ++k) sc.planes.emplace_back(v(planes + 9 * k), v(planes + 9 * k + 3), v(planes + 9 * k + 6), 2 * k + 1, (PlaneType)types[k]);
    sc.nodes = build_bvh(sc.spheres, sc.planes);
    return sc;
}
}  // namespace
// build_bvh → 9 words per node (box as x.min x.max y.min y.max z.min z.max, left, right, type); returns the node count
int32_t ref_build_bvh(int32_t ns, const float *spheres, int32_t np, const float *planes, const int32_t *types, int32_t *out_nodes9, int32_t capacity) {
    const RefScene sc = make_scene(ns, spheres, np, planes, types);
    const int32_t count = (int32_t)sc.nodes.size();
    for (int32_t k = 0; k < count && k < capacity; ++k) {
        float box[6];
        put_box(box, sc.nodes[(size_t)k].box);
        std::memcpy(out_nodes9 + 9 * k, box, 24);
        out_nodes9[9 * k + 6] = sc.nodes[(size_t)k].left; out_nodes9[9 * k + 7] = sc.nodes[(size_t)k].right; out_nodes9[9 * k + 8] = sc.nodes[(size_t)k].type;
    }
    return count;
}
// hit_bvh over the tree build_bvh makes of the scene, for n rays
void ref_hit_bvh(int32_t ns, const float *spheres, int32_t np, const float *planes, const int32_t *types, int64_t n, const float *origins,
                 const float *dirs, float tmin, float tmax, int32_t *out_hit, float *out_rec9, int32_t *out_code) {
    RefScene sc = make_scene(ns, spheres, np, planes, types);
    for (int64_t k = 0; k < n; ++k) {
        HitRecord rec;
        std::memset(&rec, 0, sizeof(rec));
        out_hit[k] = hit_bvh(Ray(v(origins + 3 * k), v(dirs + 3 * k)), Interval(tmin, tmax), rec, sc.nodes.data(), (int)sc.nodes.size(),
                             sc.spheres.data(), sc.planes.data()) ? 1 : 0;
        if (out_hit[k]) put_rec(out_rec9 + 9 * k, out_code + k, rec, true);
    }
}
// sizes the survey measured on these headers: vec3, Ray, Interval, AABB, HitRecord, SphereData, PlaneData, BVHNode
void ref_sizes(int32_t out[8]) {
    out[0] = (int32_t)sizeof(vec3); out[1] = (int32_t)sizeof(Ray); out[2] = (int32_t)sizeof(Interval); out[3] = (int32_t)sizeof(AABB);
    out[4] = (int32_t)sizeof(HitRecord); out[5] = (int32_t)sizeof(SphereData); out[6] = (int32_t)sizeof(PlaneData); out[7] = (int32_t)sizeof(BVHNode);
}

}  // extern "C"
