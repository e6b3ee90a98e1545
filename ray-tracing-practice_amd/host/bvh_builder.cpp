#include "bvh_builder.h"

#include <algorithm>
#include <cmath>

#include "vec_math.h"

namespace rtp {
namespace {

// Interval::expand(delta) applied when size() < 1e-4 (include/aabb.h:92-97, include/interval.h:26-29).
// The comparison is float-vs-double (delta is a double literal there); the padding itself is
// float(0.0001) / 2 because expand() takes a float.
inline void widen_thin_axis(float &lo, float &hi) {
    const double delta = 0.0001;
    if (static_cast<double>(hi - lo) < delta) {
        const float padding = static_cast<float>(delta) / 2;
        lo = lo - padding;
        hi = hi + padding;
    }
}

inline Box finish(Box b) {
    for (int a = 0; a < 3; ++a) widen_thin_axis(b.lo[a], b.hi[a]);
    return b;
}

// AABB(point, point): include/aabb.h:21-26
inline Box box_from_corners(Vec3 p, Vec3 q) {
    Box b;
    for (int a = 0; a < 3; ++a) {
        const float pa = p.axis(a), qa = q.axis(a);
        if (pa <= qa) { b.lo[a] = pa; b.hi[a] = qa; } else { b.lo[a] = qa; b.hi[a] = pa; }
    }
    return finish(b);
}

// AABB(AABB, AABB): include/aabb.h:28-33 with Interval(a,b) = fminf/fmaxf (include/interval.h:13-14)
inline Box box_union(const Box &p, const Box &q) {
    Box b;
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = fminf(p.lo[a], q.lo[a]);
        b.hi[a] = fmaxf(p.hi[a], q.hi[a]);
    }
    return finish(b);
}

struct BuildPrim {
    Box box;
    int type;
    int index;
    float centroid[3];
};

struct Builder {
    std::vector<BuildPrim> prims;
    std::vector<rt_bvh_node> nodes;

    // include/bvh_builder.h:52-97
    int emit(int first, int last) {
        const int me = static_cast<int>(nodes.size());
        nodes.push_back(rt_bvh_node{});

        Box bounds = prims[first].box;
        for (int k = first + 1; k < last; ++k) bounds = box_union(bounds, prims[k].box);
        for (int a = 0; a < 3; ++a) {
            nodes[me].box[2 * a] = bounds.lo[a];
            nodes[me].box[2 * a + 1] = bounds.hi[a];
        }

        if (last - first == 1) {
            nodes[me].left = -1;
            nodes[me].right = prims[first].index;
            nodes[me].type = prims[first].type;
            return me;
        }

        // Box of the centroids; every point box is itself widened to 1e-4 (AABB(c,c) constructor).
        auto point_box = [](const float c[3]) { return box_from_corners(Vec3(c[0], c[1], c[2]), Vec3(c[0], c[1], c[2])); };
        Box cb = point_box(prims[first].centroid);
        for (int k = first + 1; k < last; ++k) cb = box_union(cb, point_box(prims[k].centroid));

        int axis = 0;
        float widest = cb.hi[0] - cb.lo[0];
        if (cb.hi[1] - cb.lo[1] > widest) { axis = 1; widest = cb.hi[1] - cb.lo[1]; }
        if (cb.hi[2] - cb.lo[2] > widest) { axis = 2; }

        const int mid = (first + last) / 2;
        std::nth_element(prims.begin() + first, prims.begin() + mid, prims.begin() + last,
                         [axis](const BuildPrim &a, const BuildPrim &b) { return a.centroid[axis] < b.centroid[axis]; });

        const int l = emit(first, mid);
        const int r = emit(mid, last);
        nodes[me].left = l;
        nodes[me].right = r;
        nodes[me].type = -1;
        return me;
    }
};

}  // namespace

Box sphere_bounds(const rt_sphere &s) {
    const Vec3 c(s.center), r(s.radius, s.radius, s.radius);
    return box_from_corners(c - r, c + r);
}

Box plane_bounds(const rt_plane &p) {
    const Vec3 p0(p.base);
    const Vec3 p1 = p0 + Vec3(p.u);
    const Vec3 p2 = p0 + Vec3(p.v);
    const Vec3 p3 = p0 + Vec3(p.u) + Vec3(p.v);
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = fminf(p0.axis(a), fminf(p1.axis(a), p2.axis(a)));
        hi[a] = fmaxf(p0.axis(a), fmaxf(p1.axis(a), p2.axis(a)));
        if (p.type == RT_PLANE_QUAD || p.type == RT_PLANE_ELLIPSE) {  // triangles stop at p2
            lo[a] = fminf(lo[a], p3.axis(a));
            hi[a] = fmaxf(hi[a], p3.axis(a));
        }
    }
    // AABB(min,max) then pad(): both apply the same widening; the second is a no-op.
    return finish(box_from_corners(Vec3(lo[0], lo[1], lo[2]), Vec3(hi[0], hi[1], hi[2])));
}

std::vector<rt_bvh_node> build_bvh(const std::vector<rt_sphere> &spheres, const std::vector<rt_plane> &planes) {
    Builder b;
    b.prims.reserve(spheres.size() + planes.size());
    for (size_t i = 0; i < spheres.size(); ++i) {
        BuildPrim bp;
        bp.box = sphere_bounds(spheres[i]);
        bp.type = 0;
        bp.index = static_cast<int>(i);
        for (int a = 0; a < 3; ++a) bp.centroid[a] = spheres[i].center.e[a];
        b.prims.push_back(bp);
    }
    for (size_t i = 0; i < planes.size(); ++i) {
        BuildPrim bp;
        bp.box = plane_bounds(planes[i]);
        bp.type = 1;
        bp.index = static_cast<int>(i);
        // "approx centroid": base + (u + v) * 0.5f  (include/bvh_builder.h:112)
        const Vec3 c = Vec3(planes[i].base) + (Vec3(planes[i].u) + Vec3(planes[i].v)) * 0.5f;
        bp.centroid[0] = c.x; bp.centroid[1] = c.y; bp.centroid[2] = c.z;
        b.prims.push_back(bp);
    }
    if (b.prims.empty()) return {};
    b.nodes.reserve(2 * b.prims.size());
    b.emit(0, static_cast<int>(b.prims.size()));
    return std::move(b.nodes);
}

}  // namespace rtp
