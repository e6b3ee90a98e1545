// rt_accel.h — host-side repacking of the reference's scene arrays into the device layout.
//
// Input: the arrays of rt_scene_desc (reference layouts).  Output: flat float4 tables the kernel
// stages into LDS:
//   nodes      4 x float4 per INTERNAL node ("child-pair" form): the two child boxes and two child
//              codes, so one traversal step reads 64 B and tests both children;
//              child code >= 0 → internal node index, < 0 → leaf, -(2*prim_index + prim_type) - 1.
//   spheres    1 x float4: center.xyz, radius;  sphere_mat: int per sphere
//   planes     5 x float4: (normal, D) (w, type) (u, material) (v, 0) (base, 0)
//   materials  3 x float4: (albedo, type | texture_id << 2) (emit, fuzz) (absorption, ir)
// TreeMode::Reference / Sah: leaf boxes are copied bit for bit from the caller's BVH leaves; inner
// boxes are exact fmin/fmax unions of leaf boxes.  TreeMode::Guarded (what rt_scene_create uses):
// the same SAH build over leaf boxes inflated by a per-primitive margin, see Packed::Guard.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "../../include/rtp_amd.h"

namespace rtaccel {

constexpr int32_t kTraversalDone = INT32_MIN;   // "no more nodes" sentinel of the traversal

inline int32_t leaf_code(int32_t prim_index, int32_t prim_type) { return -(2 * prim_index + prim_type) - 1; }

constexpr int kMaxFront = 4;         // front primitives of the guarded walk (Packed::Guard)

struct Packed {
    std::vector<float> nodes;      // 16 floats per internal node
    std::vector<float> hnodes;     // Guarded: the same nodes in 8 floats — boxes as binary16 rounded outward + 2 codes
                                   // (what the kernel reads; layout in rt_accel.cpp)
    std::vector<float> spheres;    // 4 per sphere
    std::vector<int32_t> sphere_mat;
    std::vector<float> planes;     // 20 per plane
    std::vector<float> materials;  // 12 per material
    std::vector<float> tex_data;   // RGBA floats of all textures, concatenated
    std::vector<int32_t> tex_info; // 4 per texture: float offset (in float4 units), width, height, 0
    // "Threaded" copy of the caller's tree for reference-order traversal: 8 floats per node, nodes
    // in the order hit_bvh pops them (left child first, include/bvh.h:52-59): x.min x.max y.min
    // y.max | z.min z.max, miss link (index of the next node when this subtree is skipped; negative
    // when that ends the walk), leaf word ((2*index+type+1) | sign bit for a typed leaf, 0
    // otherwise).  Visiting node k: box hit → next is k+1, miss → miss link.  One extra record at
    // index num_tnodes is an end sentinel whose box is never hit.
    std::vector<float> tnodes;
    int32_t num_tnodes = 0;
    // The same walk with explicit hit/miss links and the top levels first (see rt_accel.cpp): used
    // when the tables do not fit in LDS — records [0, num_top) are the LDS "treelet".
    std::vector<float> xnodes;
    int32_t num_top = 0, xroot = 0;
    int32_t root = kTraversalDone; // node code of the root (leaf code when the scene has one primitive)
    int32_t num_internal = 0;
    bool full_pairs = true;        // no inner node has an empty child slot (kTraversalDone)
    int32_t num_top_pairs = 0;     // Guarded: nodes [0, num_top_pairs) are the top of the tree, breadth-first
    int32_t max_depth = 0;         // longest root→leaf path in internal nodes = traversal stack bound
    // Guarded, host-built tree: the same tree collapsed to 4-wide nodes (layout in rt_accel.cpp); empty when the root is a leaf
    std::vector<float> wnodes;     // 28 floats per wide node (fp32 boxes)
    std::vector<float> whnodes;    // 16 floats per wide node (binary16 boxes rounded outward)
    int32_t num_wide = 0, num_top_wide = 0, wroot = kTraversalDone, wide_depth = 0;
    // Guarded near-first walk (TreeMode::Guarded, docs/LOG.md §3b): `nodes` then is an SAH tree over
    // leaf boxes INFLATED by a per-sphere margin, and the kernel sends every sample whose result
    // could depend on the visit order to the exact reference-order walk.  guard.ok == false (with
    // a reason) → the scene is not eligible and only the reference-order walk may be used.
    struct Guard {
        bool ok = false;
        std::string reason;
        float center[3] = {0, 0, 0};   // centre of the "small sphere" cluster (class S)
        float d0_sq = 0;               // origins farther than this from the centre get the far-origin test
        float cluster_radius = 0;      // class S spheres lie within this distance of the centre
        float box[6] = {0, 0, 0, 0, 0, 0};   // their bounding box (x.min x.max y.min y.max z.min z.max)
        float far_k = 0;               // far-origin inflation of that box: far_k * (|o - centre| + cluster_radius)^2
        // Distance-aware margins (0 = off): the small spheres' leaf boxes carry only the rounding floor and every box
        // test of the walk grows its box by dyn_k * (distance from the ray origin to the box's farthest corner)^2
        // >= gamma |o - c_q|^2 / (2 r_q) for every small sphere q below — no assumption about where rays start.
        float dyn_k = 0;
        float dyn_rmax = 0;            // … and the largest radius among those small spheres (the parametric form of the growth, step_pair_par)
        int32_t num_small = 0, num_large = 0;
        // every margin assumes ray origins within origin_radius of origin_center (all scene surfaces +
        // 25 %): the camera position is checked against it per render
        float origin_center[3] = {0, 0, 0};
        float origin_radius = 0;
        // Front primitives (PackOptions::front_max): primitives whose inflated leaf box spans at least half of the surface of
        // everything that is left (the ground sphere of S-rtiow, a floor quad under a field of spheres) are NOT leaves of
        // the walk's tree.  Nearly every ray would reach them anyway — through a root step that prunes nothing and a
        // primitive test in a divergent leaf step; the kernel tests them when it arms a ray instead, all lanes of the
        // wave together, and the walk starts with their hit as its `closest`.  Codes are 2 * index + type; boxes are the inflated
        // leaf boxes (x.min x.max y.min y.max z.min z.max), which the per-pixel candidate lists still need (rt_beam.h).
        int32_t num_front = 0;
        int32_t front_code[kMaxFront] = {0, 0, 0, 0};
        float front_box[kMaxFront][6] = {};
    } guard;
    std::vector<float> leaf_boxes;        // 8 floats per sphere: the caller's exact leaf box (+2 pad), for the final check
                                          // (empty when every box is exactly fl(c -/+ r): the kernel recomputes it)
    std::vector<float> plane_leaf_boxes;  // 8 floats per plane, same purpose
    std::vector<float> guard_leaf_boxes;  // GuardedLeaves: 6 floats per leaf, inflated
    std::vector<int32_t> guard_leaf_codes;
};

// GuardedLeaves: everything of Guarded except the tree itself — the inflated leaves are returned in
// guard_leaf_boxes / guard_leaf_codes for a device-side builder (rt_build.h).
enum class TreeMode { Reference, Sah, Guarded, GuardedLeaves };

// Rounding-error budget of hit_sphere's discriminant in units of |oc|^2 |d|^2 (see docs/LOG.md §3b).
// kGuardGammaBound: the sum of every rounding's worst case is ~21 x 2^-24; 24 is what every scene gets by default.
// kGuardGammaObserved: 16x the largest error observed in ~10^8 sphere tests — NOT a bound; only used when the caller
// opts in through rt_config.guard_gamma_ulps (reported as rt_timing.guard_unproven).
constexpr float kGuardGammaBound = 24.0f * 5.9604645e-8f;
constexpr float kGuardGammaObserved = 8.0f * 5.9604645e-8f;

struct PackOptions {
    double gamma = kGuardGammaBound;   // discriminant error budget the leaf margins cover
    bool leaf_table = false;           // always emit the exact sphere leaf boxes as a table (developer)
    int dynamic = 0;                   // distance-aware margins for the small spheres: 0 = where static ones would exceed a
                                       // quarter of the smallest radius, 1 = never, 2 = always
    int front_max = kMaxFront;         // at most this many front primitives (Packed::Guard::num_front); 0: every primitive is a leaf of the tree
    int lds_pair_budget = 0;           // > 0: how many pair nodes the LDS-resident walk can hold at full occupancy.  A scene with more, whose
                                       // static margins would be small (under a quarter of the smallest radius: the distance-aware growth is
                                       // smaller still), gets distance-aware margins and with them the walk through L1 / L2 — instead of an
                                       // LDS-resident walk at one workgroup per CU or with a stack of four (dynamic == 0 only)
};

// binary16 helpers of the half-precision node table (exposed for the native test)
float half_to_float(uint16_t h);
uint16_t float_to_half_dir(float f, bool toward_minus_inf);

// Returns "" on success, else a message (→ RT_ERR_INVALID_ARG).
// camera_hint (3 floats, optional): the guarded walk's margins are sized so that this ray origin is covered
// too (a camera far outside the scene).
std::string pack_scene(const rt_scene_desc &desc, TreeMode mode, Packed &out, const PackOptions &opt = PackOptions(),
                       const float *camera_hint = nullptr);

}  // namespace rtaccel
