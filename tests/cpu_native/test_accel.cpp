// CPU-only structural test of csrc/rt_accel.cpp (the host-side packer of the device tables), built
// with AddressSanitizer + UBSan by tests/test_accel_native.py.  It links the host mirror's scene
// builders to get real inputs, so the sanitizers also sweep the config parser, the polyhedra
// builders, the BVH builder and the JPEG/PPM loaders' callers.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <sstream>
#include <vector>

#include "../../ray-tracing-practice_amd/csrc/rt_accel.h"
#include "../../ray-tracing-practice_amd/host/bvh_builder.h"
#include "../../ray-tracing-practice_amd/host/scene_builder.h"
#include "../../ray-tracing-practice_amd/host/scene_params.h"

static int failures = 0;
#define CHECK(cond)                                                         \
    do {                                                                    \
        if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

static int32_t ibits(float f) { int32_t v; std::memcpy(&v, &f, 4); return v; }

// Walk a link table pretending every box is hit (hit_all) or only boxes on the way to nothing (miss):
// returns the visited record order.
static std::vector<int32_t> walk_threaded(const std::vector<float> &t, int32_t count, bool hit_all) {
    std::vector<int32_t> order;
    int32_t k = 0;
    while (k >= 0 && k < count && order.size() < (size_t)count * 2 + 8) {
        order.push_back(k);
        k = hit_all ? k + 1 : ibits(t[(size_t)k * 8 + 6]);
    }
    return order;
}
static std::vector<int32_t> walk_explicit(const std::vector<float> &x, int32_t count, bool hit_all) {
    std::vector<int32_t> order;
    int32_t k = 0;
    while (k >= 0 && k < count && order.size() < (size_t)count * 2 + 8) {
        order.push_back(k);
        const int32_t z = ibits(x[(size_t)k * 8 + 6]), w = ibits(x[(size_t)k * 8 + 7]);
        k = (hit_all && w >= 0) ? w : z;      // inner records: hit link in w; leaves: next link in z
    }
    return order;
}

static void check_scene(rtp::HostScene &hs, const char *name) {
    const rt_scene_desc d = hs.desc();
    for (int mode = 0; mode < 2; ++mode) {
        rtaccel::Packed pk;
        const std::string err = rtaccel::pack_scene(d, mode ? rtaccel::TreeMode::Sah : rtaccel::TreeMode::Reference, pk);
        CHECK(err.empty());
        const int32_t n = pk.num_tnodes;
        CHECK(n == d.num_nodes);
        CHECK((int32_t)pk.tnodes.size() == (n + 1) * 8 && (int32_t)pk.xnodes.size() == (n + 1) * 8);
        // threaded table: "hit everything" visits every record once, in order; every leaf word names a
        // distinct primitive; miss links point forward and never past the sentinel
        const std::vector<int32_t> all = walk_threaded(pk.tnodes, n, true);
        CHECK((int32_t)all.size() == n);
        std::vector<char> seen_s(d.num_spheres, 0), seen_p(d.num_planes, 0);
        int leaves = 0;
        for (int32_t k = 0; k < n; ++k) {
            const int32_t miss = ibits(pk.tnodes[(size_t)k * 8 + 6]), leaf = ibits(pk.tnodes[(size_t)k * 8 + 7]);
            const int32_t m = miss & INT32_MAX;
            CHECK(m > k && m <= n);
            CHECK((miss < 0) == (m == n));
            if (leaf != 0) {
                CHECK(leaf < 0);
                const int32_t code = (leaf & INT32_MAX) - 1;
                CHECK(m == k + 1);
                if (code & 1) { CHECK(!seen_p[code >> 1]); seen_p[code >> 1] = 1; } else { CHECK(!seen_s[code >> 1]); seen_s[code >> 1] = 1; }
                ++leaves;
            }
            CHECK(pk.tnodes[(size_t)k * 8 + 0] <= pk.tnodes[(size_t)k * 8 + 1]);
        }
        CHECK(leaves == d.num_spheres + d.num_planes);
        CHECK(walk_threaded(pk.tnodes, n, false).size() == 1);          // root missed → walk over
        CHECK(ibits(pk.tnodes[(size_t)n * 8 + 6]) < 0);                  // sentinel ends the walk
        // explicit-link copy: same set of records, same walk length, treelet first, parents before children
        const std::vector<int32_t> xall = walk_explicit(pk.xnodes, n, true);
        CHECK((int32_t)xall.size() == n);
        std::vector<char> visited(n, 0);
        for (int32_t k : xall) { CHECK(!visited[k]); visited[k] = 1; }
        CHECK(pk.num_top <= 2048 && pk.num_top <= n && (n < 2048 ? pk.num_top == n || pk.num_top > 0 : pk.num_top == 2048));
        CHECK(walk_explicit(pk.xnodes, n, false).size() == 1);
        // child-pair tree: every primitive appears exactly once as a leaf code
        std::vector<char> s2(d.num_spheres, 0), p2(d.num_planes, 0);
        for (int32_t k = 0; k < pk.num_internal; ++k)
            for (int c = 0; c < 2; ++c) {
                const int32_t code = ibits(pk.nodes[(size_t)k * 16 + 12 + c]);
                if (code < 0 && code != rtaccel::kTraversalDone) {
                    const int32_t pc = -(code + 1);
                    if (pc & 1) { CHECK(!p2[pc >> 1]); p2[pc >> 1] = 1; } else { CHECK(!s2[pc >> 1]); s2[pc >> 1] = 1; }
                } else if (code >= 0) {
                    CHECK(code > k && code < pk.num_internal);
                }
            }
        if (pk.num_internal == 0) {       // one primitive: the root itself is its leaf code
            CHECK(d.num_spheres + d.num_planes == 1 && pk.root < 0 && pk.root != rtaccel::kTraversalDone && pk.max_depth == 0);
        } else {
            for (char c : s2) CHECK(c);
            for (char c : p2) CHECK(c);
            CHECK(pk.root == 0 && pk.max_depth >= 1 && pk.max_depth < 64);
        }
    }
    std::printf("%s: ok (%d nodes)\n", name, d.num_nodes);
}

int main() {
    {
        std::istringstream in(rtp::default_config_text());
        rtp::SceneParams p = rtp::read_scene_params(in);
        rtp::HostScene hs;
        rtp::build_config_scene(p, "", hs);
        check_scene(hs, "config scene");
        // validation: broken inputs are reported, not trusted
        rt_scene_desc d = hs.desc();
        rtaccel::Packed pk;
        std::vector<rt_bvh_node> bad(hs.nodes);
        bad[0].left = 0;
        d.nodes = bad.data();
        CHECK(!rtaccel::pack_scene(d, rtaccel::TreeMode::Reference, pk).empty());
        bad = hs.nodes;
        bad[5].right = 1 << 20;
        d.nodes = bad.data();
        CHECK(!rtaccel::pack_scene(d, rtaccel::TreeMode::Reference, pk).empty());
    }
    for (int ext : {1, 11, 40}) {
        rtp::RtiowOptions o;
        o.half_extent = ext;
        o.textured_floor_quad = ext == 40;
        o.texture_size = 32;
        rtp::HostScene hs;
        rtp::build_rtiow_scene(o, hs);
        check_scene(hs, "rtiow scene");
    }
    {   // one primitive, and none
        rtp::HostScene hs;
        rt_material m{};
        hs.materials.push_back(m);
        hs.spheres.push_back(rtp::make_sphere(rtp::Vec3(0, 0, 0), 1.0f, 0));
        hs.nodes = rtp::build_bvh(hs.spheres, hs.planes);
        check_scene(hs, "single sphere");
        rtp::HostScene empty;
        rtaccel::Packed pk;
        CHECK(rtaccel::pack_scene(empty.desc(), rtaccel::TreeMode::Sah, pk).empty());
        CHECK(pk.num_tnodes == 0 && pk.root == rtaccel::kTraversalDone);
    }
    {   // guarded-walk parameters (docs/LOG.md §3b)
        rtp::RtiowOptions o;
        o.half_extent = 11;
        rtp::HostScene hs;
        rtp::build_rtiow_scene(o, hs);
        rtaccel::Packed pk;
        CHECK(rtaccel::pack_scene(hs.desc(), rtaccel::TreeMode::Guarded, pk).empty());
        CHECK(pk.guard.ok && pk.guard.reason.empty());
        CHECK(pk.guard.num_small == 485 && pk.guard.num_large == 1);           // the ground sphere is the one "large" sphere
        CHECK(pk.leaf_boxes.empty());                                          // leaf boxes are fl(c -/+ r): derived in the kernel
        CHECK(pk.guard.d0_sq > 4 * pk.guard.cluster_radius * pk.guard.cluster_radius * 0.99f);
        // every child box of the walk's tree holds the sphere below it with the documented margin
        size_t leaves_seen = 0;
        for (int32_t k = 0; k < pk.num_internal; ++k)
            for (int c = 0; c < 2; ++c) {
                const float *n = &pk.nodes[(size_t)k * 16];
                const int32_t code = ibits(n[12 + c]);
                if (code >= 0) continue;
                const int32_t idx = (-(code + 1)) >> 1;
                const rt_sphere &sp = hs.spheres[(size_t)idx];
                const float *lo = n + 6 * c, *hi = n + 6 * c + 3;
                const double r = sp.radius;
                // small spheres: gamma * reach^2 / (2 r) with reach = d0 + cluster radius (4 % of the smallest radius)
                const double reach = std::sqrt((double)pk.guard.d0_sq) + pk.guard.cluster_radius;
                const double margin = idx == 0 ? 1e-4 : 0.99 * rtaccel::kGuardGammaBound * reach * reach / (2 * r);   // a small scene: the bound
                if (idx != 0 && r < 0.25) CHECK(margin > 0.035 * r);
                for (int a = 0; a < 3; ++a) {
                    CHECK(lo[a] <= sp.center.e[a] - r - margin && hi[a] >= sp.center.e[a] + r + margin);
                    CHECK(lo[a] >= sp.center.e[a] - r - 0.1 * r && hi[a] <= sp.center.e[a] + r + 0.1 * r);   // and not absurdly more
                }
                ++leaves_seen;
            }
        // the ground sphere spans the scene: a FRONT primitive (tested as a ray is armed), not a leaf — with its inflated box kept
        CHECK(pk.guard.num_front == 1 && pk.guard.front_code[0] == 0 && hs.spheres[0].radius == 1000.0f);
        for (int a = 0; a < 3; ++a)
            CHECK(pk.guard.front_box[0][2 * a] <= hs.spheres[0].center.e[a] - 1000.0 - 1e-4 && pk.guard.front_box[0][2 * a + 1] >= hs.spheres[0].center.e[a] + 1000.0 + 1e-4);
        CHECK(leaves_seen + 1 == hs.spheres.size());
        {   // … unless the caller says otherwise
            rtaccel::PackOptions all_leaves;
            all_leaves.front_max = 0;
            rtaccel::Packed pk0;
            CHECK(rtaccel::pack_scene(hs.desc(), rtaccel::TreeMode::Guarded, pk0, all_leaves).empty());
            CHECK(pk0.guard.ok && pk0.guard.num_front == 0 && pk0.num_internal == pk.num_internal + 1);
        }
        // the table the kernel reads: the same boxes as binary16, each plane rounded outward, same child codes
        CHECK(pk.hnodes.size() == (size_t)pk.num_internal * 8);
        for (int32_t k = 0; k < pk.num_internal; ++k) {
            const float *f = &pk.nodes[(size_t)k * 16];        // lo0.xyz hi0.xyz lo1.xyz hi1.xyz codes
            uint16_t h[12];
            std::memcpy(h, &pk.hnodes[(size_t)k * 8], sizeof(h));
            for (int c = 0; c < 2; ++c)
                for (int a = 0; a < 3; ++a) {
                    const float lo = rtaccel::half_to_float(h[6 * c + 2 * a]), hi = rtaccel::half_to_float(h[6 * c + 2 * a + 1]);
                    CHECK(lo <= f[6 * c + a] && hi >= f[6 * c + 3 + a]);
                    CHECK(f[6 * c + a] - lo <= 1.0f + 1e-3f * std::fabs(f[6 * c + a]));       // and by less than one binary16 step
                    CHECK(hi - f[6 * c + 3 + a] <= 1.0f + 1e-3f * std::fabs(f[6 * c + 3 + a]));
                }
            CHECK(ibits(pk.hnodes[(size_t)k * 8 + 6]) == ibits(f[12]) && ibits(pk.hnodes[(size_t)k * 8 + 7]) == ibits(f[13]));
        }
        for (uint32_t b = 0; b < 0xffffff00u; b += 65521u) {       // directed rounding: a bracket, and a tight one
            float x;
            std::memcpy(&x, &b, 4);
            if (std::isnan(x)) continue;
            const float dn = rtaccel::half_to_float(rtaccel::float_to_half_dir(x, true)), up = rtaccel::half_to_float(rtaccel::float_to_half_dir(x, false));
            CHECK(dn <= x && x <= up);
            if (std::fabs(x) <= 65504.0f && std::fabs(x) >= 6.2e-5f) CHECK(up - dn <= std::fabs(x) * (1.0f / 1024.0f));
        }
        // planes are eligible (their exact leaf boxes always travel as a table); a plane of unknown type is not
        rtp::HostScene with_plane;
        o.textured_floor_quad = true;
        o.texture_size = 8;
        rtp::build_rtiow_scene(o, with_plane);
        CHECK(rtaccel::pack_scene(with_plane.desc(), rtaccel::TreeMode::Guarded, pk).empty());
        CHECK(pk.guard.ok && pk.plane_leaf_boxes.size() == with_plane.planes.size() * 8 && pk.num_tnodes > 0);
        // the ground sphere, then the floor quad under the field of small spheres (half of what is left once the ground is out)
        CHECK(pk.guard.num_front == 2 && pk.guard.front_code[0] == 1 && pk.guard.front_code[1] == 0);       // (the plane first)
        {
            std::vector<rt_plane> planes(with_plane.planes);
            rt_scene_desc dp = with_plane.desc();
            planes[0].type = 7;
            dp.planes = planes.data();
            CHECK(rtaccel::pack_scene(dp, rtaccel::TreeMode::Guarded, pk).empty());
            CHECK(!pk.guard.ok && pk.guard.reason == "plane of unknown type");
        }
        // not eligible: a sphere in two leaves; a leaf box that does not hold its sphere
        std::vector<rt_bvh_node> nodes(hs.nodes);
        rt_scene_desc d = hs.desc();
        d.nodes = nodes.data();
        int first_leaf = -1, second_leaf = -1;
        for (size_t k = 0; k < nodes.size(); ++k)
            if (nodes[k].left < 0) { if (first_leaf < 0) first_leaf = (int)k; else if (second_leaf < 0) second_leaf = (int)k; }
        const int32_t keep = nodes[(size_t)second_leaf].right;
        nodes[(size_t)second_leaf].right = nodes[(size_t)first_leaf].right;
        CHECK(rtaccel::pack_scene(d, rtaccel::TreeMode::Guarded, pk).empty() && !pk.guard.ok);
        nodes[(size_t)second_leaf].right = keep;
        nodes[(size_t)first_leaf].box[1] -= 0.05f;
        CHECK(rtaccel::pack_scene(d, rtaccel::TreeMode::Guarded, pk).empty() && !pk.guard.ok);
        // a caller's tree with padded leaf boxes is eligible, but then the exact boxes travel as a table
        nodes = hs.nodes;
        for (rt_bvh_node &n : nodes) { for (int a = 0; a < 3; ++a) { n.box[2 * a] -= 0.01f; n.box[2 * a + 1] += 0.01f; } }
        CHECK(rtaccel::pack_scene(d, rtaccel::TreeMode::Guarded, pk).empty());
        CHECK(pk.guard.ok && pk.leaf_boxes.size() == hs.spheres.size() * 8);
    }
    std::printf(failures ? "FAILED (%d)\n" : "all ok\n", failures);
    return failures ? 1 : 0;
}
