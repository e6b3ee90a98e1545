// bvh_builder.h — host-side BVH construction, kept on the host as in the reference
// (include/bvh_builder.h:10-120): median split on the widest centroid axis, one primitive per
// leaf, nodes emitted in pre-order, 2N-1 nodes.  The output is the reference's 36-byte node
// array (rt_bvh_node), which the C ABI takes as is.
#pragma once
#include <vector>
#include "../../include/rtp_amd.h"

namespace rtp {

// Axis-aligned box with the reference's "never thinner than 1e-4" rule (include/aabb.h:92-97).
struct Box {
    float lo[3], hi[3];
};

Box sphere_bounds(const rt_sphere &s);   // include/bvh_builder.h:17-20
Box plane_bounds(const rt_plane &p);     // include/bvh_builder.h:22-50

// include/bvh_builder.h:99-120.  Spheres first (type 0), then planes (type 1).
std::vector<rt_bvh_node> build_bvh(const std::vector<rt_sphere> &spheres, const std::vector<rt_plane> &planes);

}  // namespace rtp
