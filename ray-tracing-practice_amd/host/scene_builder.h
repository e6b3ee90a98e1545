// scene_builder.h — host-side scene construction (kept on the host, as in the reference).
//
//  * build_config_scene(): the reference's create_scene (src/main.cu:346-497) — three polyhedral
//    dielectric bodies with metal edge strips and emissive edge spheres, a (textured) metal floor
//    quad and radius-1 light spheres — producing the same arrays in the same order.
//  * build_rtiow_scene(): the benchmark scenes of BASELINE.json (SURVEY.md §8(d) "S-rtiow",
//    "S-100k"): the reference has no such generator, so this one is the build's own, driven by
//    the reference's RNG so the scene is reproducible everywhere.
#pragma once
#include <string>
#include <vector>

#include "../../include/rtp_amd.h"
#include "scene_params.h"
#include "texture_io.h"

namespace rtp {

// Owner of everything the reference keeps in host_spheres/host_planes/host_materials/
// host_bvh_nodes (src/main.cu:573-578) plus decoded textures.
struct HostScene {
    std::vector<rt_sphere> spheres;
    std::vector<rt_plane> planes;
    std::vector<rt_material> materials;
    std::vector<rt_bvh_node> nodes;
    std::vector<TextureImage> textures;
    std::vector<rt_texture> texture_views;  // rebuilt by desc()

    // View in the layout rt_scene_create() takes.  Valid while *this is alive and unmodified.
    rt_scene_desc desc();
};

// PlaneData's host constructor (include/plane.h:19-28): normal = unit(cross(u,v)),
// D = dot(normal, base), w = n / dot(n, n).
rt_plane make_plane(Vec3 base, Vec3 u, Vec3 v, int material_idx, int type);
rt_sphere make_sphere(Vec3 center, float radius, int material_idx);

// src/main.cu:346-497 (CPU branch semantics: textures are sampled like tex2D_cpu).
// texture_dir: if non-empty, a relative texture path is resolved against it.
void build_config_scene(const SceneParams &params, const std::string &texture_dir, HostScene &out);

struct RtiowOptions {
    unsigned seed = 12345u;
    int half_extent = 11;          // a, b in [-half_extent, half_extent): 11 → 486 spheres
    bool textured_floor_quad = false;  // S-100k adds one textured METAL quad under the field
    int texture_size = 1024;       // procedural checker texture edge (only if textured)
};
void build_rtiow_scene(const RtiowOptions &opt, HostScene &out);

}  // namespace rtp
