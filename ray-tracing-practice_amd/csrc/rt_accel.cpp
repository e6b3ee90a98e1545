#include "rt_accel.h"
#include "rt_device_math.h"      // schlick_r0sq: the packer evaluates it with the arithmetic the kernel would use

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>

namespace rtaccel {
namespace {

inline float bits_as_float(int32_t v) { float f; std::memcpy(&f, &v, 4); return f; }

struct LeafRef {
    float box[6];   // x.min x.max y.min y.max z.min z.max — the reference layout
    int32_t code;
};

struct BuildNode {
    float box[6];
    int32_t child[2];   // codes
};

void box_union(const float a[6], const float b[6], float out[6]) {
    for (int k = 0; k < 3; ++k) {
        out[2 * k] = fminf(a[2 * k], b[2 * k]);
        out[2 * k + 1] = fmaxf(a[2 * k + 1], b[2 * k + 1]);
    }
}

float half_area(const float b[6]) {
    const float dx = b[1] - b[0], dy = b[3] - b[2], dz = b[5] - b[4];
    return dx * dy + dy * dz + dz * dx;
}

// Top-down SAH build over the leaf boxes (sweep on all three axes; N log^2 N).  Only the
// topology differs from the caller's tree: boxes are exact unions of the same leaf boxes.
struct SahBuilder {
    std::vector<LeafRef> &leaves;
    std::vector<BuildNode> nodes;
    std::vector<float> right_area;

    explicit SahBuilder(std::vector<LeafRef> &l) : leaves(l) {}

    static constexpr int kBinnedAbove = 512, kBins = 32;

    // Partitions [first, last) by the cheapest of 3 x 31 bin boundaries; returns the split position.
    int binned_split(int first, int last) {
        const int n = last - first;
        float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = first; k < last; ++k)
            for (int a = 0; a < 3; ++a) {
                const float c = leaves[static_cast<size_t>(k)].box[2 * a] + leaves[static_cast<size_t>(k)].box[2 * a + 1];   // 2 x centroid
                clo[a] = fminf(clo[a], c);
                chi[a] = fmaxf(chi[a], c);
            }
        int best_axis = -1, best_bin = -1;
        float best_cost = INFINITY;
        for (int a = 0; a < 3; ++a) {
            if (!(chi[a] > clo[a])) continue;
            const float scale = kBins / (chi[a] - clo[a]);
            float bbox[kBins][6];
            int count[kBins];
            for (int b = 0; b < kBins; ++b) {
                count[b] = 0;
                for (int k = 0; k < 3; ++k) { bbox[b][2 * k] = INFINITY; bbox[b][2 * k + 1] = -INFINITY; }
            }
            for (int k = first; k < last; ++k) {
                const LeafRef &l = leaves[static_cast<size_t>(k)];
                int b = static_cast<int>((l.box[2 * a] + l.box[2 * a + 1] - clo[a]) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                count[b]++;
                box_union(bbox[b], l.box, bbox[b]);
            }
            float right_area_bin[kBins];
            int right_count[kBins];
            float acc[6] = {INFINITY, -INFINITY, INFINITY, -INFINITY, INFINITY, -INFINITY};
            int cnt = 0;
            for (int b = kBins - 1; b >= 1; --b) {
                if (count[b]) box_union(acc, bbox[b], acc);
                cnt += count[b];
                right_area_bin[b] = cnt ? half_area(acc) : 0.0f;
                right_count[b] = cnt;
            }
            float lacc[6] = {INFINITY, -INFINITY, INFINITY, -INFINITY, INFINITY, -INFINITY};
            int lcnt = 0;
            for (int b = 1; b < kBins; ++b) {       // boundary between bin b-1 and b
                if (count[b - 1]) box_union(lacc, bbox[b - 1], lacc);
                lcnt += count[b - 1];
                if (lcnt == 0 || right_count[b] == 0) continue;
                const float cost = half_area(lacc) * lcnt + right_area_bin[b] * right_count[b];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
            }
        }
        if (best_axis < 0) {      // all centroids coincide: any split will do
            return first + n / 2;
        }
        const int a = best_axis;
        const float scale = kBins / (chi[a] - clo[a]);
        const float lo = clo[a];
        const int bb = best_bin;
        auto mid = std::partition(leaves.begin() + first, leaves.begin() + last, [a, scale, lo, bb](const LeafRef &l) {
            int b = static_cast<int>((l.box[2 * a] + l.box[2 * a + 1] - lo) * scale);
            b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
            return b < bb;
        });
        const int pos = static_cast<int>(mid - leaves.begin());
        return (pos == first || pos == last) ? first + n / 2 : pos;
    }

    int32_t build(int first, int last, float out_box[6]) {
        if (last - first == 1) {
            std::memcpy(out_box, leaves[first].box, sizeof(float) * 6);
            return leaves[first].code;
        }
        const int n = last - first;
        if (n > kBinnedAbove) {
            // big ranges: binned SAH (32 bins per axis over the centroid bounds, O(n) per level) — the full
            // sweep below sorts the range three times per node, 0.6 s for 100 k primitives
            const int mid = binned_split(first, last);
            const int32_t me = static_cast<int32_t>(nodes.size());
            nodes.push_back(BuildNode{});
            float lb[6], rb[6];
            const int32_t l = build(first, mid, lb);
            const int32_t r = build(mid, last, rb);
            nodes[static_cast<size_t>(me)].child[0] = l;
            nodes[static_cast<size_t>(me)].child[1] = r;
            box_union(lb, rb, out_box);
            std::memcpy(nodes[static_cast<size_t>(me)].box, out_box, sizeof(float) * 6);
            return me;
        }
        int best_axis = -1, best_split = -1;
        float best_cost = INFINITY;
        for (int axis = 0; axis < 3; ++axis) {
            std::sort(leaves.begin() + first, leaves.begin() + last, [axis](const LeafRef &a, const LeafRef &b) {
                return a.box[2 * axis] + a.box[2 * axis + 1] < b.box[2 * axis] + b.box[2 * axis + 1];
            });
            right_area.resize(static_cast<size_t>(n));
            float acc[6];
            std::memcpy(acc, leaves[last - 1].box, sizeof(acc));
            for (int k = n - 1; k >= 1; --k) {
                box_union(acc, leaves[first + k].box, acc);
                right_area[static_cast<size_t>(k)] = half_area(acc);
            }
            std::memcpy(acc, leaves[first].box, sizeof(acc));
            for (int k = 1; k < n; ++k) {
                box_union(acc, leaves[first + k - 1].box, acc);
                const float cost = half_area(acc) * k + right_area[static_cast<size_t>(k)] * (n - k);
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = k; }
            }
        }
        if (best_axis < 0) { best_axis = 0; best_split = n / 2; }
        if (best_axis != 2) {
            const int axis = best_axis;
            std::sort(leaves.begin() + first, leaves.begin() + last, [axis](const LeafRef &a, const LeafRef &b) {
                return a.box[2 * axis] + a.box[2 * axis + 1] < b.box[2 * axis] + b.box[2 * axis + 1];
            });
        }
        const int32_t me = static_cast<int32_t>(nodes.size());
        nodes.push_back(BuildNode{});
        float lb[6], rb[6];
        const int32_t l = build(first, first + best_split, lb);
        const int32_t r = build(first + best_split, last, rb);
        nodes[static_cast<size_t>(me)].child[0] = l;
        nodes[static_cast<size_t>(me)].child[1] = r;
        box_union(lb, rb, out_box);
        std::memcpy(nodes[static_cast<size_t>(me)].box, out_box, sizeof(float) * 6);
        return me;
    }
};

}  // namespace

// ---- binary16 with directed rounding (the walk's boxes may only grow) ------------------------------
float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
    if (exp == 0x1f) return bits_as_float((int32_t)(sign | 0x7f800000u | (man << 13)));
    if (exp == 0) {
        const float v = std::ldexp((float)man, -24);
        return sign ? -v : v;
    }
    return bits_as_float((int32_t)(sign | ((exp + 112u) << 23) | (man << 13)));
}
// largest half <= f (toward_minus_inf) or smallest half >= f; finite results where a finite one exists
uint16_t float_to_half_dir(float f, bool toward_minus_inf) {
    if (std::isnan(f)) return toward_minus_inf ? 0xfc00u : 0x7c00u;      // -inf / +inf: the conservative end
    if (f > 65504.0f) return toward_minus_inf ? 0x7bffu : 0x7c00u;
    if (f < -65504.0f) return toward_minus_inf ? 0xfc00u : 0xfbffu;
    // positive halves are ordered like their bit patterns: binary search on the magnitude
    const float a = std::fabs(f);
    uint32_t lo = 0, hi = 0x7bff;                    // largest magnitude pattern with value <= a
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (half_to_float((uint16_t)mid) <= a) lo = mid; else hi = mid - 1;
    }
    const bool exact = half_to_float((uint16_t)lo) == a;
    const bool negative = std::signbit(f);
    // rounding a positive value down / a negative value up keeps the smaller magnitude; otherwise the next one
    const bool shrink = negative != toward_minus_inf;
    uint32_t mag = lo;
    if (!exact && !shrink) mag = lo + 1;              // <= 0x7c00 (inf) by the range checks above: 0x7bff + 1
    return (uint16_t)((negative ? 0x8000u : 0u) | mag);
}

namespace {

// pack_scene in steps that share the caller's arrays and the leaves found in them.
struct Packer {
    const rt_scene_desc &d;
    TreeMode mode;
    Packed &out;
    PackOptions opt;
    const float *camera_hint;
    std::vector<LeafRef> leaves;          // typed leaves of the caller's tree (boxes get inflated by prepare_guard)
    std::vector<int32_t> depth;           // per caller node
    std::vector<char> reachable;          // per caller node: reached from the root
    std::vector<BuildNode> bnodes;        // the traversal tree the pair tables are emitted from

    std::string validate() const;         // counts, indices, textures
    std::string collect_leaves();         // + the caller's tree is a tree in pre-order
    void thread_tables();                 // reference-order walk: threaded table and explicit-link table
    void prepare_guard();                 // guarded walk: eligibility, margins, exact leaf boxes
    void build_tree();                    // SAH / caller topology / leaves for the device builder
    void pair_tables();                   // child-pair records (fp32 and binary16), tree depth
    std::string primitive_tables();       // spheres, planes, materials, textures
};

std::string Packer::validate() const {
    if (d.num_spheres < 0 || d.num_planes < 0 || d.num_materials < 0 || d.num_nodes < 0 || d.num_textures < 0)
        return "negative element count";
    if ((d.num_spheres && !d.spheres) || (d.num_planes && !d.planes) || (d.num_materials && !d.materials) ||
        (d.num_nodes && !d.nodes) || (d.num_textures && !d.textures))
        return "null array with non-zero count";
    if (d.num_spheres > (1 << 29) || d.num_planes > (1 << 29)) return "too many primitives";

    for (int i = 0; i < d.num_spheres; ++i)
        if (d.spheres[i].material_idx < 0 || d.spheres[i].material_idx >= d.num_materials)
            return "sphere material index out of range";
    for (int i = 0; i < d.num_planes; ++i) {
        if (d.planes[i].material_idx < 0 || d.planes[i].material_idx >= d.num_materials)
            return "plane material index out of range";
    }
    for (int i = 0; i < d.num_materials; ++i) {
        if (d.materials[i].texture_id > static_cast<uint64_t>(d.num_textures) || d.materials[i].texture_id >= (1u << 28)) return "material texture id out of range";
        if (d.materials[i].type < 0 || d.materials[i].type > 3) return "unknown material type";
    }
    for (int i = 0; i < d.num_textures; ++i)
        if (!d.textures[i].rgba || d.textures[i].width <= 0 || d.textures[i].height <= 0) return "bad texture";

    return "";
}

std::string Packer::collect_leaves() {
    // ---- leaves of the caller's tree, validated to be a tree in pre-order (children after parent)
    leaves.clear();
    depth.assign(static_cast<size_t>(d.num_nodes), 0);
    reachable.assign(static_cast<size_t>(d.num_nodes), 0);
    if (d.num_nodes > 0) reachable[0] = 1;
    for (int k = 0; k < d.num_nodes; ++k) {
        const rt_bvh_node &n = d.nodes[k];
        if (!reachable[static_cast<size_t>(k)]) continue;   // unreachable entries are never visited by hit_bvh
        if (n.left < 0) {
            if (n.type == 0) {
                if (n.right < 0 || n.right >= d.num_spheres) return "leaf sphere index out of range";
            } else if (n.type == 1) {
                if (n.right < 0 || n.right >= d.num_planes) return "leaf plane index out of range";
            } else {
                continue;   // leaf of unknown type: the reference tests nothing there (include/bvh.h:39-43)
            }
            LeafRef lr;
            std::memcpy(lr.box, n.box, sizeof(lr.box));
            lr.code = leaf_code(n.right, n.type);
            leaves.push_back(lr);
        } else {
            if (n.left <= k || n.left >= d.num_nodes || n.right <= k || n.right >= d.num_nodes)
                return "BVH nodes must be in pre-order (children after their parent, inside the array)";
            if (reachable[static_cast<size_t>(n.left)] || reachable[static_cast<size_t>(n.right)] || n.left == n.right)
                return "BVH node referenced twice";
            reachable[static_cast<size_t>(n.left)] = reachable[static_cast<size_t>(n.right)] = 1;
            depth[static_cast<size_t>(n.left)] = depth[static_cast<size_t>(n.right)] = depth[static_cast<size_t>(k)] + 1;
        }
    }

    return "";
}

void Packer::thread_tables() {
    // ---- threaded copy in the reference's own visit order (iterative DFS, left first)
    {
        struct Item { int32_t node; int32_t stack_ptr; };
        std::vector<Item> todo;
        std::vector<int32_t> order, subtree_end_slot;
        std::vector<int32_t> visit_sp;
        if (d.num_nodes > 0) todo.push_back({0, 0});
        // first pass: visit order (exactly the pops of hit_bvh, including its silent pruning
        // when fewer than two stack slots are left, include/bvh.h:51)
        std::vector<int32_t> pos_of(static_cast<size_t>(d.num_nodes), -1);
        while (!todo.empty()) {
            const Item it = todo.back();
            todo.pop_back();
            pos_of[static_cast<size_t>(it.node)] = static_cast<int32_t>(order.size());
            order.push_back(it.node);
            visit_sp.push_back(it.stack_ptr);
            const rt_bvh_node &n = d.nodes[it.node];
            if (n.left >= 0 && it.stack_ptr + 2 <= 32) {
                todo.push_back({n.right, it.stack_ptr});       // popped after the whole left subtree
                todo.push_back({n.left, it.stack_ptr + 1});
            }
        }
        const int32_t count = static_cast<int32_t>(order.size());
        // skip pointer = position just after the node's subtree: computed backwards
        std::vector<int32_t> skip(static_cast<size_t>(count), count);
        for (int32_t p = count - 1; p >= 0; --p) {
            const rt_bvh_node &n = d.nodes[order[static_cast<size_t>(p)]];
            const bool expanded = n.left >= 0 && visit_sp[static_cast<size_t>(p)] + 2 <= 32;
            if (!expanded) skip[static_cast<size_t>(p)] = p + 1;
            else skip[static_cast<size_t>(p)] = skip[static_cast<size_t>(pos_of[static_cast<size_t>(n.right)])];
        }
        out.num_tnodes = count;
        out.tnodes.resize(static_cast<size_t>(count + 1) * 8);
        const int32_t blocked = INT32_MIN;      // sign bit: "no box test wanted" (see step_threaded)
        for (int32_t p = 0; p < count; ++p) {
            const rt_bvh_node &n = d.nodes[order[static_cast<size_t>(p)]];
            float *o = &out.tnodes[static_cast<size_t>(p) * 8];
            for (int k = 0; k < 6; ++k) o[k] = n.box[k];      // x.min x.max y.min y.max z.min z.max, as the caller has them
            const int32_t miss = skip[static_cast<size_t>(p)];
            o[6] = bits_as_float(miss == count ? (count | blocked) : miss);     // the link that ends the walk is negative
            int32_t leaf = 0;
            if (n.left < 0 && (n.type == 0 || n.type == 1)) leaf = (2 * n.right + n.type + 1) | blocked;
            o[7] = bits_as_float(leaf);
        }
        {   // end sentinel at index count: reached when the last node's box was hit; never hit itself
            float *o = &out.tnodes[static_cast<size_t>(count) * 8];
            o[0] = 1e30f; o[1] = -1e30f; o[2] = 1e30f; o[3] = -1e30f; o[4] = 1e30f; o[5] = -1e30f;
            o[6] = bits_as_float(count | blocked);
            o[7] = bits_as_float(0);
        }

        // ---- the same walk with EXPLICIT links, for scenes too big for LDS: the records of the top
        // levels (breadth-first from the root) come first so the kernel can keep that "treelet" in
        // LDS and read only deeper nodes through L1/L2; deeper records stay in depth-first order.
        //   inner record: z = miss link, w = hit link (>= 0)
        //   leaf record : z = next link (hit or miss), w = sign | (2*index+type+1)   (sign alone: a leaf
        //                 with nothing to test — untyped, or an inner node hit_bvh prunes)
        //   links are record indices; the end of the walk is the sentinel record at index count,
        //   whose box is never hit and whose miss link is negative.
        {
            const int32_t want_top = 2048;                     // records kept in LDS at most (64 KB)
            std::vector<int32_t> new_of(static_cast<size_t>(count), -1);       // DFS position → record index
            std::vector<int32_t> bfs;
            bfs.reserve(static_cast<size_t>(want_top));
            if (count > 0) bfs.push_back(0);
            for (size_t head = 0; head < bfs.size() && static_cast<int32_t>(bfs.size()) < want_top; ++head) {
                const int32_t p = bfs[head];
                const rt_bvh_node &n = d.nodes[order[static_cast<size_t>(p)]];
                const bool expanded = n.left >= 0 && visit_sp[static_cast<size_t>(p)] + 2 <= 32;
                if (!expanded) continue;
                bfs.push_back(pos_of[static_cast<size_t>(n.left)]);
                if (static_cast<int32_t>(bfs.size()) < want_top) bfs.push_back(pos_of[static_cast<size_t>(n.right)]);
            }
            int32_t next_index = 0;
            for (int32_t p : bfs) new_of[static_cast<size_t>(p)] = next_index++;
            out.num_top = next_index;
            for (int32_t p = 0; p < count; ++p)
                if (new_of[static_cast<size_t>(p)] < 0) new_of[static_cast<size_t>(p)] = next_index++;
            auto link = [&](int32_t dfs_pos) { return dfs_pos == count ? count : new_of[static_cast<size_t>(dfs_pos)]; };
            out.xnodes.resize(static_cast<size_t>(count + 1) * 8);
            for (int32_t p = 0; p < count; ++p) {
                const rt_bvh_node &n = d.nodes[order[static_cast<size_t>(p)]];
                float *o = &out.xnodes[static_cast<size_t>(new_of[static_cast<size_t>(p)]) * 8];
                for (int k = 0; k < 6; ++k) o[k] = n.box[k];
                const bool expanded = n.left >= 0 && visit_sp[static_cast<size_t>(p)] + 2 <= 32;
                o[6] = bits_as_float(link(skip[static_cast<size_t>(p)]));
                if (expanded) {
                    o[7] = bits_as_float(link(p + 1));
                } else {
                    int32_t leaf = blocked;
                    if (n.left < 0 && (n.type == 0 || n.type == 1)) leaf |= 2 * n.right + n.type + 1;
                    o[7] = bits_as_float(leaf);
                }
            }
            float *o = &out.xnodes[static_cast<size_t>(count) * 8];
            o[0] = 1e30f; o[1] = -1e30f; o[2] = 1e30f; o[3] = -1e30f; o[4] = 1e30f; o[5] = -1e30f;
            o[6] = bits_as_float(count | blocked);
            o[7] = bits_as_float(0);
            out.xroot = count > 0 ? new_of[0] : count;
        }
    }

}

void Packer::prepare_guard() {
    // ---- guarded mode: eligibility, per-primitive inflation, exact leaf boxes for the final check
    if (mode == TreeMode::Guarded || mode == TreeMode::GuardedLeaves) {
        Packed::Guard &g = out.guard;
        std::string why;
        const double u = 5.9604645e-8;       // 2^-24
        // error budget of hit_sphere's discriminant, in units of |oc|^2 |d|^2: the term-by-term worst-case bound
        // (kGuardGammaBound) unless the caller opted into a smaller, unproven margin (rt_config.guard_gamma_ulps)
        const double gamma = opt.gamma;
        std::vector<int32_t> leaf_of_sphere(static_cast<size_t>(d.num_spheres), -1), leaf_of_plane(static_cast<size_t>(d.num_planes), -1);
        if (leaves.empty()) why = "no primitives";
        else if (d.num_spheres >= (1 << 24) || d.num_planes >= (1 << 24)) why = "too many primitives";
        for (int k = 0; k < d.num_nodes && why.empty(); ++k) {
            if (!reachable[static_cast<size_t>(k)]) continue;
            const rt_bvh_node &n = d.nodes[k];
            if (depth[static_cast<size_t>(k)] + 2 > 32) why = "tree deeper than hit_bvh's stack";       // the reference prunes silently there
            for (int a = 0; a < 6 && why.empty(); ++a) if (!std::isfinite(n.box[a])) why = "non-finite box";
            if (n.left < 0 && n.type == 0) {
                if (leaf_of_sphere[static_cast<size_t>(n.right)] >= 0) { why = "sphere referenced by two leaves"; break; }
                leaf_of_sphere[static_cast<size_t>(n.right)] = k;
                const rt_sphere &s = d.spheres[n.right];
                if (!(s.radius > 0) || !std::isfinite(s.radius)) { why = "sphere radius not positive"; break; }
                for (int a = 0; a < 3; ++a) {
                    const float c = s.center.e[a];
                    if (!std::isfinite(c)) { why = "non-finite sphere centre"; break; }
                    // the leaf box must hold the sphere (up to the rounding of c -/+ r)
                    const float tol = 4.0f * 5.9604645e-8f * (fabsf(c) + s.radius);
                    if (n.box[2 * a] > c - s.radius + tol || n.box[2 * a + 1] < c + s.radius - tol) why = "leaf box does not contain its sphere";
                }
            } else if (n.left < 0 && n.type == 1) {
                if (leaf_of_plane[static_cast<size_t>(n.right)] >= 0) { why = "plane referenced by two leaves"; break; }
                leaf_of_plane[static_cast<size_t>(n.right)] = k;
                const rt_plane &pl = d.planes[n.right];
                if (pl.type != RT_PLANE_QUAD && pl.type != RT_PLANE_ELLIPSE && pl.type != RT_PLANE_TRIANGLE) { why = "plane of unknown type"; break; }
                // accepted hits lie (up to rounding) inside base + a u + b v, a, b in [0,1] (a + b <= 1 for a
                // triangle): the leaf box must hold those corners
                const int corners = pl.type == RT_PLANE_TRIANGLE ? 3 : 4;
                for (int c = 0; c < corners && why.empty(); ++c)
                    for (int a = 0; a < 3; ++a) {
                        const double x = double(pl.base.e[a]) + ((c & 1) ? double(pl.u.e[a]) : 0.0) + ((c & 2) ? double(pl.v.e[a]) : 0.0);
                        if (!std::isfinite(x)) { why = "non-finite plane"; break; }
                        const double tol = 8.0 * u * (std::fabs(double(pl.base.e[a])) + std::fabs(double(pl.u.e[a])) + std::fabs(double(pl.v.e[a])));
                        if (n.box[2 * a] > x + tol || n.box[2 * a + 1] < x - tol) { why = "leaf box does not contain its plane"; break; }
                    }
            } else if (n.left < 0) {
                why = "leaf of unknown type";
            } else {
                for (int child : {n.left, n.right}) {
                    const rt_bvh_node &c = d.nodes[child];
                    for (int a = 0; a < 3; ++a)
                        if (c.box[2 * a] < n.box[2 * a] || c.box[2 * a + 1] > n.box[2 * a + 1]) why = "child box not inside its parent's";
                }
            }
        }
        if (why.empty()) {
            // bounding sphere (C, r_all) of everything a path can start from: the surfaces of all primitives
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (const LeafRef &l : leaves)
                for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], l.box[2 * a]); hi[a] = fmaxf(hi[a], l.box[2 * a + 1]); }
            const double C[3] = {0.5 * (double(lo[0]) + hi[0]), 0.5 * (double(lo[1]) + hi[1]), 0.5 * (double(lo[2]) + hi[2])};
            auto dist = [](const double a[3], const float b[3]) {
                return std::sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
            };
            double r_all = 0, coord_max = 0;
            for (const LeafRef &l : leaves) {
                double far2 = 0;
                for (int a = 0; a < 3; ++a) {
                    const double e = std::max(std::fabs(l.box[2 * a] - C[a]), std::fabs(l.box[2 * a + 1] - C[a]));
                    far2 += e * e;
                    coord_max = std::max(coord_max, std::max(std::fabs(double(l.box[2 * a])), std::fabs(double(l.box[2 * a + 1]))));
                }
                r_all = std::max(r_all, std::sqrt(far2));
            }
            // ray origins: any point of a scene surface is within r_all of C; 25 % on top for the camera (checked per render)
            double origin_radius = 1.25 * r_all;
            if (camera_hint) {      // a camera farther out than that: size the margins for it (rt_render re-packs on demand)
                const float cam[3] = {camera_hint[0], camera_hint[1], camera_hint[2]};
                origin_radius = std::max(origin_radius, 1.25 * dist(C, cam));
            }
            for (int a = 0; a < 3; ++a) g.origin_center[a] = static_cast<float>(C[a]);
            g.origin_radius = static_cast<float>(origin_radius);
            // floor of every margin: the rounding of the walk's own box tests and of o + t d (a few ulps of the
            // coordinates of the box and of any ray origin)
            const double eps_floor = 32.0 * u * (coord_max + std::fabs(C[0]) + std::fabs(C[1]) + std::fabs(C[2]) + origin_radius);
            // planes: an accepted hit passed the interior test on the computed point itself, so it is in
            // the primitive up to rounding — the floor, doubled, is the whole margin
            const double eps_plane = 2.0 * eps_floor;
            // spheres, class L ("large"): the margin for ANY admissible origin stays below 5 % of the radius
            std::vector<char> large(static_cast<size_t>(d.num_spheres), 0);
            std::vector<double> eps(static_cast<size_t>(d.num_spheres), 0.0);
            float slo[3] = {INFINITY, INFINITY, INFINITY}, shi[3] = {-INFINITY, -INFINITY, -INFINITY};
            double r_min_small = INFINITY;
            for (int i = 0; i < d.num_spheres; ++i) {
                if (leaf_of_sphere[static_cast<size_t>(i)] < 0) continue;
                const rt_sphere &s = d.spheres[i];
                const double reach = dist(C, s.center.e) + origin_radius;
                const double e = gamma * reach * reach / (2.0 * s.radius);
                if (e <= 0.05 * s.radius) {
                    large[static_cast<size_t>(i)] = 1;
                    eps[static_cast<size_t>(i)] = e;
                    g.num_large++;
                } else {
                    g.num_small++;
                    r_min_small = std::min(r_min_small, double(s.radius));
                    for (int a = 0; a < 3; ++a) { slo[a] = fminf(slo[a], s.center.e[a]); shi[a] = fmaxf(shi[a], s.center.e[a]); }
                }
            }
            if (g.num_small > 0) {
                const double sc[3] = {0.5 * (double(slo[0]) + shi[0]), 0.5 * (double(slo[1]) + shi[1]), 0.5 * (double(slo[2]) + shi[2])};
                double rs = 0;
                for (int i = 0; i < d.num_spheres; ++i)
                    if (leaf_of_sphere[static_cast<size_t>(i)] >= 0 && !large[static_cast<size_t>(i)])
                        rs = std::max(rs, dist(sc, d.spheres[i].center.e) + d.spheres[i].radius);
                // origins within d0 of the centre: margin 4 % of the smallest radius, but never less than the
                // cluster itself (paths start on its surfaces)
                // … and up to 25 % of it where that buys origins out to 3 cluster radii from the centre (a camera
                // orbiting a compact scene of tiny spheres must not make every primary ray a far-origin ray)
                double reach = std::sqrt(0.08 * r_min_small * r_min_small / gamma);
                const double reach_25 = std::sqrt(0.50 * r_min_small * r_min_small / gamma);
                reach = std::max(reach, std::min(4.0 * rs, reach_25));
                if (reach < 2.0 * rs) reach = 2.0 * rs;
                if (camera_hint) {  // primary rays from a far camera: without this every one that heads for the cluster is flagged
                    const float cam[3] = {camera_hint[0], camera_hint[1], camera_hint[2]};
                    reach = std::max(reach, 1.25 * (dist(sc, cam) + rs));
                }
                // One margin per sphere has to cover the farthest admissible origin: for tiny spheres spread over a wide
                // volume it (∝ reach^2 / r) approaches or exceeds the radius and swallows the tree (S-100k with the proven
                // gamma: 3.6 radii — the walk then does more work than the reference's).  Such scenes get DISTANCE-AWARE
                // margins instead: the leaves keep only the rounding floor and the walk grows every box it tests by
                // dyn_k * (distance from the ray's own origin to the box's farthest corner)^2, which bounds
                // gamma |o - c_q|^2 / (2 r_q) for every small sphere q below the box, wherever the ray starts.
                const double eps_static = gamma * reach * reach / (2.0 * r_min_small);
                // … automatically only where the small spheres are of one size class (largest / smallest radius <= 8): the
                // growth factor dyn_k follows the SMALLEST of them, so with radii spread over decades every box of the tree
                // would grow for the sake of a few specks and the walk loses to the reference's (tools/guard_stress.py:
                // 2-3 x slower on such scenes); those keep the static rule, i.e. mostly the exact walk
                double r_max_small = 0;
                for (int i = 0; i < d.num_spheres; ++i)
                    if (leaf_of_sphere[static_cast<size_t>(i)] >= 0 && !large[static_cast<size_t>(i)]) r_max_small = std::max(r_max_small, double(d.spheres[i].radius));
                // … and where the tree is too big for the LDS-resident walk anyway and the static margins are small: the growth is
                // smaller than they are everywhere within reach, and the walk through L1 / L2 (step_wide_par) does not care how big the
                // tree is (S-rtiow scaled to 785 / 1 298 spheres: 6.8 / 5.2 Gsamples/s LDS-resident at one workgroup per CU, ≈ 7.5
                // through L1 / L2; tools/size_sweep.py)
                const bool too_big_for_lds = opt.lds_pair_budget > 0 && static_cast<int64_t>(leaves.size()) - 1 > opt.lds_pair_budget;
                const bool dynamic = opt.dynamic == 2 || (opt.dynamic == 0 && ((eps_static > 0.25 * r_min_small && r_max_small <= 8.0 * r_min_small) ||
                                                                               (too_big_for_lds && eps_static <= 0.25 * r_min_small)));
                if (!dynamic && eps_static > 64.0 * r_min_small) why = "margins exceed 64 radii for the smallest spheres";
                const double d0 = reach - rs;
                for (int i = 0; i < d.num_spheres; ++i)
                    if (leaf_of_sphere[static_cast<size_t>(i)] >= 0 && !large[static_cast<size_t>(i)])
                        eps[static_cast<size_t>(i)] = dynamic ? 0.0 : gamma * reach * reach / (2.0 * d.spheres[i].radius);
                for (int a = 0; a < 3; ++a) g.center[a] = static_cast<float>(sc[a]);
                g.d0_sq = dynamic ? INFINITY : static_cast<float>(d0 * d0 * (1.0 - 1e-6));      // dynamic: no far-origin test needed
                g.cluster_radius = static_cast<float>(rs * (1.0 + 1e-6));
                g.far_k = static_cast<float>(gamma / (2.0 * r_min_small) * (1.0 + 1e-6));
                g.dyn_k = dynamic ? static_cast<float>(gamma / (2.0 * r_min_small) * (1.0 + 4e-6)) : 0.0f;
                g.dyn_rmax = dynamic ? static_cast<float>(r_max_small * (1.0 + 1e-6)) : 0.0f;
                for (int i = 0; i < d.num_spheres; ++i) {
                    if (leaf_of_sphere[static_cast<size_t>(i)] < 0 || large[static_cast<size_t>(i)]) continue;
                    const float *b = d.nodes[leaf_of_sphere[static_cast<size_t>(i)]].box;
                    for (int a = 0; a < 3; ++a) { slo[a] = fminf(slo[a], b[2 * a]); shi[a] = fmaxf(shi[a], b[2 * a + 1]); }
                }
                for (int a = 0; a < 3; ++a) { g.box[2 * a] = slo[a]; g.box[2 * a + 1] = shi[a]; }
            } else {
                g.d0_sq = INFINITY;       // no small spheres: no far-origin test
            }
            // inflate the leaf boxes the traversal tree is built from; keep the exact ones for the final check
            out.leaf_boxes.assign(static_cast<size_t>(d.num_spheres) * 8, 0.0f);
            out.plane_leaf_boxes.assign(static_cast<size_t>(d.num_planes) * 8, 0.0f);
            for (LeafRef &l : leaves) {
                const int32_t code = -(l.code + 1);
                const int32_t i = code >> 1;
                float *lb = (code & 1) ? &out.plane_leaf_boxes[static_cast<size_t>(i) * 8] : &out.leaf_boxes[static_cast<size_t>(i) * 8];
                for (int a = 0; a < 6; ++a) lb[a] = l.box[a];
                const double e = (code & 1) ? eps_plane : std::max(eps[static_cast<size_t>(i)], eps_floor);
                const float ef = static_cast<float>(e * (1.0 + 1e-6));
                for (int a = 0; a < 3; ++a) {
                    l.box[2 * a] = std::nextafterf(l.box[2 * a] - ef, -INFINITY);
                    l.box[2 * a + 1] = std::nextafterf(l.box[2 * a + 1] + ef, INFINITY);
                }
            }
            // sphere leaf boxes that are exactly fl(c - r), fl(c + r) (what the reference's builder makes) can be
            // recomputed in the kernel from the sphere record: no table
            bool derivable = true;
            for (int i = 0; i < d.num_spheres && derivable; ++i) {
                if (leaf_of_sphere[static_cast<size_t>(i)] < 0) continue;
                const rt_sphere &s = d.spheres[i];
                const float *lb = &out.leaf_boxes[static_cast<size_t>(i) * 8];
                for (int a = 0; a < 3; ++a)
                    if (lb[2 * a] != s.center.e[a] - s.radius || lb[2 * a + 1] != s.center.e[a] + s.radius) derivable = false;
            }
            if (derivable && !opt.leaf_table) out.leaf_boxes.clear();
            g.ok = why.empty();
            // ---- front primitives (rt_accel.h): while the largest leaf box spans at least half of the surface of what is left,
            // it leaves the tree (which keeps at least two leaves: its root stays an inner node)
            g.num_front = 0;
            const int front_max = std::min(opt.front_max, kMaxFront);
            while (g.ok && g.num_front < front_max && leaves.size() > 2) {
                float all[6] = {INFINITY, -INFINITY, INFINITY, -INFINITY, INFINITY, -INFINITY};
                size_t big = 0;
                for (size_t k = 0; k < leaves.size(); ++k) {
                    box_union(all, leaves[k].box, all);
                    if (half_area(leaves[k].box) > half_area(leaves[big].box)) big = k;
                }
                if (!(half_area(leaves[big].box) >= 0.5f * half_area(all))) break;
                g.front_code[g.num_front] = -(leaves[big].code + 1);
                std::memcpy(g.front_box[g.num_front], leaves[big].box, sizeof(float) * 6);
                g.num_front++;
                leaves.erase(leaves.begin() + static_cast<std::ptrdiff_t>(big));
            }
            // planes first (the cheaper test)
            for (int a = 1; a < g.num_front; ++a)
                for (int b = a; b > 0 && (g.front_code[b] & 1) && !(g.front_code[b - 1] & 1); --b) {
                    std::swap(g.front_code[b], g.front_code[b - 1]);
                    for (int k = 0; k < 6; ++k) std::swap(g.front_box[b][k], g.front_box[b - 1][k]);
                }
        }
        g.reason = why;
    }

}

void Packer::build_tree() {
    // ---- traversal tree
    bnodes.clear();
    if (mode == TreeMode::GuardedLeaves) {
        // the caller builds the tree itself (device builder, rt_build.hip) from the inflated leaves
        out.root = kTraversalDone;
        if (out.guard.ok) {
            out.guard_leaf_boxes.reserve(leaves.size() * 6);
            out.guard_leaf_codes.reserve(leaves.size());
            for (const LeafRef &l : leaves) {
                out.guard_leaf_boxes.insert(out.guard_leaf_boxes.end(), l.box, l.box + 6);
                out.guard_leaf_codes.push_back(l.code);
            }
        }
    } else if (leaves.empty()) {
        out.root = kTraversalDone;
    } else if (mode == TreeMode::Sah || mode == TreeMode::Guarded || d.nodes[0].left < 0) {
        SahBuilder b(leaves);
        float rb[6];
        out.root = b.build(0, static_cast<int>(leaves.size()), rb);
        bnodes = std::move(b.nodes);
        if (mode == TreeMode::Guarded && out.root >= 0) {
            // top levels first (breadth-first from the root): a scene too big for LDS keeps records
            // [0, num_top_pairs) there as a "treelet" and reads the deeper ones through L1/L2
            const int32_t count = static_cast<int32_t>(bnodes.size());
            const int32_t want_top = std::min<int32_t>(count, 2048);
            std::vector<int32_t> new_of(static_cast<size_t>(count), -1), bfs;
            bfs.reserve(static_cast<size_t>(want_top));
            bfs.push_back(out.root);
            for (size_t head = 0; head < bfs.size(); ++head)
                for (int c = 0; c < 2; ++c) {
                    const int32_t ch = bnodes[static_cast<size_t>(bfs[head])].child[c];
                    if (ch >= 0 && static_cast<int32_t>(bfs.size()) < want_top) bfs.push_back(ch);
                }
            int32_t next_index = 0;
            for (int32_t k : bfs) new_of[static_cast<size_t>(k)] = next_index++;
            out.num_top_pairs = next_index;
            for (int32_t k = 0; k < count; ++k)
                if (new_of[static_cast<size_t>(k)] < 0) new_of[static_cast<size_t>(k)] = next_index++;
            std::vector<BuildNode> moved(static_cast<size_t>(count));
            for (int32_t k = 0; k < count; ++k) {
                BuildNode n = bnodes[static_cast<size_t>(k)];
                for (int c = 0; c < 2; ++c)
                    if (n.child[c] >= 0) n.child[c] = new_of[static_cast<size_t>(n.child[c])];
                moved[static_cast<size_t>(new_of[static_cast<size_t>(k)])] = n;
            }
            bnodes = std::move(moved);
            out.root = new_of[static_cast<size_t>(out.root)];
        }
    } else {
        // same topology as the caller's tree: internal node k → dense index, in array order
        std::vector<int32_t> dense(static_cast<size_t>(d.num_nodes), -1);
        int32_t count = 0;
        for (int k = 0; k < d.num_nodes; ++k)
            if (reachable[static_cast<size_t>(k)] && d.nodes[k].left >= 0) dense[static_cast<size_t>(k)] = count++;
        bnodes.resize(static_cast<size_t>(count));
        auto code_of = [&](int32_t k) -> int32_t {
            const rt_bvh_node &c = d.nodes[k];
            if (c.left >= 0) return dense[static_cast<size_t>(k)];
            if (c.type == 0 || c.type == 1) return leaf_code(c.right, c.type);
            return kTraversalDone;   // untyped leaf: never hit
        };
        for (int k = 0; k < d.num_nodes; ++k) {
            if (dense[static_cast<size_t>(k)] < 0) continue;
            BuildNode &bn = bnodes[static_cast<size_t>(dense[static_cast<size_t>(k)])];
            std::memcpy(bn.box, d.nodes[k].box, sizeof(bn.box));
            bn.child[0] = code_of(d.nodes[k].left);
            bn.child[1] = code_of(d.nodes[k].right);
        }
        out.root = 0;
    }

}

void Packer::pair_tables() {
    // ---- child-pair node table
    // box of a child code: leaf → its exact leaf box; internal → that node's box
    std::vector<const float *> sphere_box(static_cast<size_t>(d.num_spheres), nullptr), plane_box(static_cast<size_t>(d.num_planes), nullptr);
    for (const LeafRef &l : leaves) {
        const int32_t c = -(l.code + 1);
        if (c & 1) plane_box[static_cast<size_t>(c >> 1)] = l.box; else sphere_box[static_cast<size_t>(c >> 1)] = l.box;
    }
    const float empty_box[6] = {1e30f, -1e30f, 1e30f, -1e30f, 1e30f, -1e30f};   // never hit
    auto box_of = [&](int32_t code) -> const float * {
        if (code == kTraversalDone) return empty_box;
        if (code >= 0) return bnodes[static_cast<size_t>(code)].box;
        const int32_t c = -(code + 1);
        return (c & 1) ? plane_box[static_cast<size_t>(c >> 1)] : sphere_box[static_cast<size_t>(c >> 1)];
    };
    out.num_internal = static_cast<int32_t>(bnodes.size());
    // (every inner node of a tree built here has two children — step_pair_par relies on it; the caller-topology tree of
    // TreeMode::Reference may carry untyped leaves as empty slots, and no guarded kernel walks that one)
    out.full_pairs = true;
    for (const BuildNode &b : bnodes)
        if (b.child[0] == kTraversalDone || b.child[1] == kTraversalDone) out.full_pairs = false;
    out.nodes.resize(bnodes.size() * 16);
    for (size_t k = 0; k < bnodes.size(); ++k) {
        float *o = &out.nodes[k * 16];
        const float *b0 = box_of(bnodes[k].child[0]);
        const float *b1 = box_of(bnodes[k].child[1]);
        // lo0.xyz hi0.xyz lo1.xyz hi1.xyz
        o[0] = b0[0]; o[1] = b0[2]; o[2] = b0[4]; o[3] = b0[1]; o[4] = b0[3]; o[5] = b0[5];
        o[6] = b1[0]; o[7] = b1[2]; o[8] = b1[4]; o[9] = b1[1]; o[10] = b1[3]; o[11] = b1[5];
        o[12] = bits_as_float(bnodes[k].child[0]);
        o[13] = bits_as_float(bnodes[k].child[1]);
        o[14] = 0; o[15] = 0;
    }
    // the guarded walk reads the pair boxes as binary16, rounded OUTWARD (its boxes only have to contain the
    // inflated leaf boxes): 32 B per node instead of 64 — half the LDS footprint and half the read traffic.
    //   8 halves: lo0.x hi0.x lo0.y hi0.y lo0.z hi0.z lo1.x hi1.x | 4 halves: lo1.y hi1.y lo1.z hi1.z, code0, code1
    if (mode == TreeMode::Guarded) {
        out.hnodes.assign(bnodes.size() * 8, 0.0f);
        for (size_t k = 0; k < bnodes.size(); ++k) {
            const float *o = &out.nodes[k * 16];        // lo0.xyz hi0.xyz lo1.xyz hi1.xyz
            uint16_t h[12];
            for (int a = 0; a < 3; ++a) {
                h[2 * a] = float_to_half_dir(o[a], true);
                h[2 * a + 1] = float_to_half_dir(o[3 + a], false);
                h[6 + 2 * a] = float_to_half_dir(o[6 + a], true);
                h[6 + 2 * a + 1] = float_to_half_dir(o[9 + a], false);
            }
            std::memcpy(&out.hnodes[k * 8], h, sizeof(h));          // 24 bytes = floats 0..5
            out.hnodes[k * 8 + 6] = o[12];
            out.hnodes[k * 8 + 7] = o[13];
        }
    }
    // ---- 4-wide nodes (Guarded, host-built tree): the binary tree collapsed — an inner child is replaced by its own
    // two children, largest box first, until the node has four children or only leaves.  Half the steps per ray for
    // slightly FEWER box tests (tools/nearfirst_study.c WIDE=1: S-rtiow 5.4 steps / 20.1 box tests per ray instead of
    // 10.8 / 22.6; S-100k 10.1 / 39.5 instead of 20.0 / 41.1).  Same boxes, same leaves: everything docs/LOG.md §3b says
    // about the guarded walk holds unchanged.  Nodes are numbered breadth-first (top of the tree first, for the LDS
    // treelet of big scenes).
    //   wnodes  (fp32, 7 x float4): lo.x[4] hi.x[4] lo.y[4] hi.y[4] lo.z[4] hi.z[4] code[4]
    //   whnodes (binary16 rounded outward, 4 x float4): per axis the (lo, hi) pairs of the four children — one dword each: x[4] y[4] z[4] —, then code[4]
    //   unused child slots: code kTraversalDone (the kernel tells them by the code, not by the box).
    if (mode == TreeMode::Guarded && out.root >= 0) {
        struct Wide { int32_t child[4]; int n; };
        std::vector<Wide> wide;
        std::vector<int32_t> wide_of(bnodes.size(), -1), order;       // binary inner node → wide node; BFS queue of binary nodes
        order.push_back(out.root);
        wide_of[static_cast<size_t>(out.root)] = 0;
        for (size_t head = 0; head < order.size(); ++head) {
            const BuildNode &b = bnodes[static_cast<size_t>(order[head])];
            Wide w{{b.child[0], b.child[1], kTraversalDone, kTraversalDone}, 2};
            while (w.n < 4) {
                int best = -1;
                float best_area = -1.0f;
                for (int k = 0; k < w.n; ++k)
                    if (w.child[k] >= 0) {
                        const float a = half_area(bnodes[static_cast<size_t>(w.child[k])].box);
                        if (a > best_area) { best_area = a; best = k; }
                    }
                if (best < 0) break;
                const BuildNode &c = bnodes[static_cast<size_t>(w.child[best])];
                w.child[best] = c.child[0];
                w.child[w.n++] = c.child[1];
            }
            for (int k = 0; k < w.n; ++k)
                if (w.child[k] >= 0) {
                    wide_of[static_cast<size_t>(w.child[k])] = static_cast<int32_t>(order.size());
                    order.push_back(w.child[k]);
                }
            wide.push_back(w);
        }
        out.num_wide = static_cast<int32_t>(wide.size());
        out.num_top_wide = std::min<int32_t>(out.num_wide, 1024);
        out.wroot = 0;
        out.wnodes.assign(wide.size() * 28, 0.0f);
        out.whnodes.assign(wide.size() * 16, 0.0f);
        for (size_t i = 0; i < wide.size(); ++i) {
            float *o = &out.wnodes[i * 28];
            uint16_t h[24];
            for (int k = 0; k < 4; ++k) {
                const int32_t code = wide[i].child[k];
                const float *b = box_of(code);
                for (int a = 0; a < 3; ++a) {
                    o[(2 * a) * 4 + k] = b[2 * a];
                    o[(2 * a + 1) * 4 + k] = b[2 * a + 1];
                    h[a * 8 + 2 * k] = float_to_half_dir(b[2 * a], true);            // per axis: the (lo, hi) pairs of the four children
                    h[a * 8 + 2 * k + 1] = float_to_half_dir(b[2 * a + 1], false);
                }
                const int32_t mapped = code >= 0 ? wide_of[static_cast<size_t>(code)] : code;
                o[24 + k] = bits_as_float(mapped);
                out.whnodes[i * 16 + 12 + static_cast<size_t>(k)] = bits_as_float(mapped);
            }
            std::memcpy(&out.whnodes[i * 16], h, sizeof(h));       // 48 bytes = floats 0..11
        }
        // stack bound of the wide walk: up to three entries per level
        int32_t maxd = 0;
        std::function<void(int32_t, int32_t)> wwalk = [&](int32_t w, int32_t dep) {
            maxd = std::max(maxd, dep);
            for (int k = 0; k < wide[static_cast<size_t>(w)].n; ++k) {
                const int32_t c = wide[static_cast<size_t>(w)].child[k];
                if (c >= 0) wwalk(wide_of[static_cast<size_t>(c)], dep + 1);
            }
        };
        wwalk(0, 1);
        out.wide_depth = maxd;
    }
    // depth of the traversal tree (stack bound: one entry per level at most)
    {
        int32_t maxd = 0;
        std::function<void(int32_t, int32_t)> walk = [&](int32_t code, int32_t dep) {
            if (code < 0) { maxd = std::max(maxd, dep); return; }
            walk(bnodes[static_cast<size_t>(code)].child[0], dep + 1);
            walk(bnodes[static_cast<size_t>(code)].child[1], dep + 1);
        };
        if (out.root != kTraversalDone) walk(out.root, 0);
        out.max_depth = maxd;
    }

}

std::string Packer::primitive_tables() {
    // ---- primitive and material tables
    out.spheres.resize(static_cast<size_t>(d.num_spheres) * 4);
    out.sphere_mat.resize(static_cast<size_t>(d.num_spheres));
    for (int i = 0; i < d.num_spheres; ++i) {
        float *o = &out.spheres[static_cast<size_t>(i) * 4];
        o[0] = d.spheres[i].center.e[0]; o[1] = d.spheres[i].center.e[1]; o[2] = d.spheres[i].center.e[2];
        o[3] = d.spheres[i].radius;
        out.sphere_mat[static_cast<size_t>(i)] = d.spheres[i].material_idx;
    }
    out.planes.resize(static_cast<size_t>(d.num_planes) * 20);
    for (int i = 0; i < d.num_planes; ++i) {
        const rt_plane &p = d.planes[i];
        float *o = &out.planes[static_cast<size_t>(i) * 20];
        o[0] = p.normal.e[0]; o[1] = p.normal.e[1]; o[2] = p.normal.e[2]; o[3] = p.D;
        o[4] = p.w.e[0]; o[5] = p.w.e[1]; o[6] = p.w.e[2]; o[7] = bits_as_float(p.type);
        o[8] = p.u.e[0]; o[9] = p.u.e[1]; o[10] = p.u.e[2]; o[11] = bits_as_float(p.material_idx);
        o[12] = p.v.e[0]; o[13] = p.v.e[1]; o[14] = p.v.e[2]; o[15] = 0;
        o[16] = p.base.e[0]; o[17] = p.base.e[1]; o[18] = p.base.e[2]; o[19] = 0;
    }
    out.materials.resize(static_cast<size_t>(d.num_materials) * 12);
    for (int i = 0; i < d.num_materials; ++i) {
        const rt_material &m = d.materials[i];
        float *o = &out.materials[static_cast<size_t>(i) * 12];
        const int32_t tag = m.type | (static_cast<int32_t>(m.texture_id) << 2);
        o[0] = m.albedo.e[0]; o[1] = m.albedo.e[1]; o[2] = m.albedo.e[2]; o[3] = bits_as_float(tag);
        o[4] = m.emit.e[0]; o[5] = m.emit.e[1]; o[6] = m.emit.e[2]; o[7] = m.fuzz;
        // DIELECTRIC never reads fuzz: its slot carries the refraction ratio of a front-face hit, (float)(1.0 / ir)
        // (include/materials.h:100), so the kernel's glass branch has one division less
        if (m.type == RT_MAT_DIELECTRIC) {
            o[7] = static_cast<float>(1.0 / static_cast<double>(m.ir));
            // … and never reads albedo either (its attenuation starts from 1): slots 0 and 1 carry Schlick's r0^2
            // (include/materials.h:65-66) for the two refraction ratios a hit can have, front (1/ir) and back (ir)
            o[0] = rtd::schlick_r0sq(o[7]);
            o[1] = rtd::schlick_r0sq(m.ir);
        }
        o[8] = m.absorption.e[0]; o[9] = m.absorption.e[1]; o[10] = m.absorption.e[2]; o[11] = m.ir;
    }
    size_t texels = 0;
    for (int i = 0; i < d.num_textures; ++i) {
        const rt_texture &t = d.textures[i];
        out.tex_info.push_back(static_cast<int32_t>(texels));
        out.tex_info.push_back(t.width);
        out.tex_info.push_back(t.height);
        out.tex_info.push_back(0);
        const size_t n = static_cast<size_t>(t.width) * t.height;
        if (texels + n > (1u << 30)) return "textures too large";
        out.tex_data.insert(out.tex_data.end(), t.rgba, t.rgba + n * 4);
        texels += n;
    }
    return "";
}

}  // namespace

std::string pack_scene(const rt_scene_desc &d, TreeMode mode, Packed &out, const PackOptions &opt, const float *camera_hint) {
    out = Packed{};
    Packer pk{d, mode, out, opt, camera_hint, {}, {}, {}, {}};
    std::string err = pk.validate();
    if (!err.empty()) return err;
    err = pk.collect_leaves();
    if (!err.empty()) return err;
    pk.thread_tables();
    pk.prepare_guard();
    pk.build_tree();
    pk.pair_tables();
    if (mode == TreeMode::Guarded && out.guard.ok && !out.full_pairs) {      // (cannot happen: the SAH build splits every range in two)
        out.guard.ok = false;
        out.guard.reason = "traversal tree with an empty child slot";
    }
    return pk.primitive_tables();
}

}  // namespace rtaccel
