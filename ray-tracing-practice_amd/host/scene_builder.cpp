#include "scene_builder.h"

#include <cmath>
#include <cstring>

#include "bvh_builder.h"
#include "host_rng.h"

namespace rtp {

rt_scene_desc HostScene::desc() {
    texture_views.clear();
    for (const TextureImage &t : textures) texture_views.push_back(rt_texture{t.rgba.data(), t.width, t.height});
    rt_scene_desc d;
    d.spheres = spheres.data();      d.num_spheres = static_cast<int32_t>(spheres.size());
    d.planes = planes.data();        d.num_planes = static_cast<int32_t>(planes.size());
    d.materials = materials.data();  d.num_materials = static_cast<int32_t>(materials.size());
    d.nodes = nodes.data();          d.num_nodes = static_cast<int32_t>(nodes.size());
    d.textures = texture_views.data(); d.num_textures = static_cast<int32_t>(texture_views.size());
    return d;
}

rt_plane make_plane(Vec3 base, Vec3 u, Vec3 v, int material_idx, int type) {
    rt_plane p;
    std::memset(&p, 0, sizeof(p));
    const Vec3 n = cross(u, v);
    const Vec3 unit_n = normalized(n);
    p.type = type;
    p.material_idx = material_idx;
    p.base = base.pod();
    p.u = u.pod();
    p.v = v.pod();
    p.normal = unit_n.pod();
    p.D = dot(unit_n, base);
    p.w = div_scalar(n, dot(n, n)).pod();
    return p;
}

rt_sphere make_sphere(Vec3 center, float radius, int material_idx) {
    rt_sphere s;
    std::memset(&s, 0, sizeof(s));
    s.center = center.pod();
    s.radius = radius;
    s.material_idx = material_idx;
    return s;
}

namespace {

rt_material blank_material(int type) {
    rt_material m;
    std::memset(&m, 0, sizeof(m));  // the reference leaves unused fields uninitialised; they are never read
    m.type = type;
    return m;
}

// One polyhedral body = shared vertex directions + a face list + an edge list.  The reference has
// three hand-unrolled generators (add_octahedron src/main.cu:248-308, add_cube :62-133,
// add_dodecahedron :138-233); they differ only in the tables, the face primitive, the
// face-distance constant and whether faces or edge strips are emitted first, so one table-driven
// emitter reproduces all three (same primitive order, same float expressions).
struct BodyEmitter {
    HostScene &scene;
    Vec3 center;
    float r;
    int body_mat, border_mat, edge_light_mat, lights_per_edge;
    float bead_radius;  // r / 100 * 2

    // Metal strip along one (shrunken) edge plus its emissive beads (src/main.cu:104-124).
    void edge_strip(Vec3 start, Vec3 end) {
        const Vec3 along = end - start;
        const Vec3 mid = (start + end) * 0.5f;
        const Vec3 radial = normalized(mid - center);
        const Vec3 tangent = normalized(cross(along, radial));
        const float width = r * 0.05f;
        const Vec3 base = start - tangent * (width * 0.5f);
        scene.planes.push_back(make_plane(base, along, tangent * width, border_mat, RT_PLANE_QUAD));
        for (int k = 0; k < lights_per_edge; ++k) {
            const float t = (k + 0.5f) / lights_per_edge;
            const Vec3 pos = (1.0f - t) * start + t * end;
            scene.spheres.push_back(make_sphere(pos, bead_radius, edge_light_mat));
        }
    }
    void triangle(Vec3 a, Vec3 b, Vec3 c) {
        scene.planes.push_back(make_plane(a, b - a, c - a, body_mat, RT_PLANE_TRIANGLE));
    }
    void quad(Vec3 a, Vec3 b, Vec3 d) {
        scene.planes.push_back(make_plane(a, b - a, d - a, body_mat, RT_PLANE_QUAD));
    }
};

// Scale of the inner "light" skeleton so the beads sit just under the faces.
float skeleton_scale(float dist_to_face, float bead_radius, bool zero_when_too_small) {
    if (dist_to_face > bead_radius) return (dist_to_face - bead_radius) / dist_to_face;
    return zero_when_too_small ? 0.0f : 1.0f;
}

void emit_octahedron(BodyEmitter &e) {  // src/main.cu:248-308
    static const float dirs[6][3] = {{0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}, {1, 0, 0}, {-1, 0, 0}};
    static const int faces[8][3] = {{0, 2, 4}, {0, 4, 3}, {0, 3, 5}, {0, 5, 2}, {1, 4, 2}, {1, 3, 4}, {1, 5, 3}, {1, 2, 5}};
    static const int edges[12][2] = {{0, 2}, {0, 4}, {0, 3}, {0, 5}, {1, 2}, {1, 4}, {1, 3}, {1, 5}, {2, 4}, {4, 3}, {3, 5}, {5, 2}};
    const float scale = skeleton_scale(e.r * 0.57735026919f, e.bead_radius, false);
    Vec3 outer[6], inner[6];
    for (int i = 0; i < 6; ++i) {
        const Vec3 d = normalized(Vec3(dirs[i][0], dirs[i][1], dirs[i][2]));
        outer[i] = e.center + d * e.r;
        inner[i] = e.center + d * (e.r * scale);
    }
    for (const auto &f : faces) e.triangle(outer[f[0]], outer[f[1]], outer[f[2]]);
    for (const auto &ed : edges) e.edge_strip(inner[ed[0]], inner[ed[1]]);
}

void emit_cube(BodyEmitter &e) {  // src/main.cu:62-133
    static const float dirs[8][3] = {{-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1},
                                     {-1, -1, 1},  {1, -1, 1},  {1, 1, 1},  {-1, 1, 1}};
    static const int edges[12][2] = {{0, 1}, {1, 5}, {5, 4}, {4, 0}, {3, 2}, {2, 6}, {6, 7}, {7, 3}, {0, 3}, {1, 2}, {5, 6}, {4, 7}};
    static const int faces[6][4] = {{4, 5, 6, 7}, {1, 0, 3, 2}, {5, 1, 2, 6}, {4, 7, 3, 0}, {7, 6, 2, 3}, {0, 1, 5, 4}};
    const float scale = skeleton_scale(e.r / std::sqrt(3.0f), e.bead_radius, true);
    Vec3 outer[8], inner[8];
    for (int i = 0; i < 8; ++i) {
        const Vec3 d = normalized(Vec3(dirs[i][0], dirs[i][1], dirs[i][2]));
        outer[i] = e.center + d * e.r;
        inner[i] = e.center + d * (e.r * scale);
    }
    for (const auto &ed : edges) e.edge_strip(inner[ed[0]], inner[ed[1]]);  // strips first for the cube
    for (const auto &f : faces) e.quad(outer[f[0]], outer[f[1]], outer[f[3]]);
}

void emit_dodecahedron(BodyEmitter &e) {  // src/main.cu:138-233
    const float phi = 1.61803398875f;
    const float inv = 1.0f / phi;
    const float dirs[20][3] = {{1, 1, 1},      {1, 1, -1},     {1, -1, 1},      {1, -1, -1},     {-1, 1, 1},
                               {-1, 1, -1},    {-1, -1, 1},    {-1, -1, -1},    {0, phi, inv},   {0, phi, -inv},
                               {0, -phi, inv}, {0, -phi, -inv}, {inv, 0, phi},  {inv, 0, -phi},  {-inv, 0, phi},
                               {-inv, 0, -phi}, {phi, inv, 0},  {phi, -inv, 0}, {-phi, inv, 0},  {-phi, -inv, 0}};
    static const int faces[12][5] = {{12, 2, 17, 16, 0}, {8, 4, 14, 12, 0},  {16, 1, 9, 8, 0},   {17, 3, 13, 1, 16},
                                     {13, 15, 5, 9, 1},  {14, 6, 10, 2, 12}, {10, 11, 3, 17, 2}, {3, 11, 7, 15, 13},
                                     {18, 19, 6, 14, 4}, {9, 5, 18, 4, 8},   {7, 11, 10, 6, 19}, {5, 15, 7, 19, 18}};
    const float scale = skeleton_scale(e.r * 0.79465447229f, e.bead_radius, false);
    Vec3 outer[20], inner[20];
    for (int i = 0; i < 20; ++i) {
        const Vec3 d = normalized(Vec3(dirs[i][0], dirs[i][1], dirs[i][2]));
        outer[i] = e.center + d * e.r;
        inner[i] = e.center + d * (e.r * scale);
    }
    bool seen[20][20] = {};
    for (const auto &f : faces) {
        // pentagon as a fan of three triangles, then the not-yet-seen edges of this face
        e.triangle(outer[f[0]], outer[f[1]], outer[f[2]]);
        e.triangle(outer[f[0]], outer[f[2]], outer[f[3]]);
        e.triangle(outer[f[0]], outer[f[3]], outer[f[4]]);
        for (int k = 0; k < 5; ++k) {
            const int a = f[k], b = f[(k + 1) % 5];
            const int lo = a < b ? a : b, hi = a < b ? b : a;
            if (seen[lo][hi]) continue;
            seen[lo][hi] = true;
            e.edge_strip(inner[lo], inner[hi]);
        }
    }
}

}  // namespace

void build_config_scene(const SceneParams &params, const std::string &texture_dir, HostScene &out) {
    out = HostScene{};

    // material 0: floor — METAL, albedo = tint, fuzz = reflection coefficient (src/main.cu:349-360)
    rt_material floor_mat = blank_material(RT_MAT_METAL);
    floor_mat.albedo = params.floor.tint.pod();
    floor_mat.fuzz = params.floor.reflection;
    if (!params.floor.texture_path.empty()) {
        std::string path = params.floor.texture_path;
        if (!texture_dir.empty() && path[0] != '/') path = texture_dir + "/" + path;
        TextureImage img;
        if (load_texture(path, img)) {
            out.textures.push_back(std::move(img));
            floor_mat.texture_id = out.textures.size();  // 1-based
        }
    }
    out.materials.push_back(floor_mat);
    const int floor_idx = 0;

    // material 1: edge beads — emit = lights[0].col * 0.1 (src/main.cu:362-367).  With zero lights
    // the reference indexes an empty vector; here the beads are simply black.
    rt_material bead = blank_material(RT_MAT_DIFFUSE_LIGHT);
    if (!params.lights.empty()) bead.emit = (params.lights[0].colour * static_cast<float>(0.1)).pod();
    out.materials.push_back(bead);
    const int bead_idx = 1;

    for (size_t i = 0; i < params.bodies.size(); ++i) {
        const BodyParams &b = params.bodies[i];
        rt_material glass = blank_material(RT_MAT_DIELECTRIC);  // src/main.cu:376-383
        glass.ir = 1.0f + b.reflection;
        const float strength = (1.0f - b.transparency) * 0.5f;
        glass.absorption = Vec3(strength * (1.0f - b.colour.x), strength * (1.0f - b.colour.y), strength * (1.0f - b.colour.z)).pod();
        out.materials.push_back(glass);
        const int glass_idx = static_cast<int>(out.materials.size()) - 1;

        rt_material border = blank_material(RT_MAT_METAL);  // grey 0.5, fuzz 0.6
        border.albedo = Vec3(0.5f, 0.5f, 0.5f).pod();
        border.fuzz = 0.6f;
        out.materials.push_back(border);
        const int border_idx = static_cast<int>(out.materials.size()) - 1;

        BodyEmitter e{out, b.center, b.radius, glass_idx, border_idx, bead_idx, b.lights_per_edge, b.radius / 100 * 2};
        if (i == 0) emit_octahedron(e);
        else if (i == 1) emit_cube(e);
        else emit_dodecahedron(e);
    }

    // floor quad (src/main.cu:413-415)
    const Vec3 *c = params.floor.corners;
    out.planes.push_back(make_plane(c[0], c[1] - c[0], c[3] - c[0], floor_idx, RT_PLANE_QUAD));

    // radius-1 emissive spheres (src/main.cu:417-426)
    for (const LightParams &l : params.lights) {
        rt_material lm = blank_material(RT_MAT_DIFFUSE_LIGHT);
        lm.emit = l.colour.pod();
        out.materials.push_back(lm);
        out.spheres.push_back(make_sphere(l.position, 1.0f, static_cast<int>(out.materials.size()) - 1));
    }

    out.nodes = build_bvh(out.spheres, out.planes);
}

void build_rtiow_scene(const RtiowOptions &opt, HostScene &out) {
    out = HostScene{};
    unsigned seed = opt.seed;
    auto rf = [&seed]() { return random_float(seed); };
    auto add = [&out](Vec3 center, float radius, const rt_material &m) {
        out.materials.push_back(m);
        out.spheres.push_back(make_sphere(center, radius, static_cast<int>(out.materials.size()) - 1));
    };

    rt_material ground = blank_material(RT_MAT_LAMBERTIAN);
    ground.albedo = Vec3(0.5f, 0.5f, 0.5f).pod();
    add(Vec3(0, 0, -1000), 1000, ground);

    for (int a = -opt.half_extent; a < opt.half_extent; ++a) {
        for (int b = -opt.half_extent; b < opt.half_extent; ++b) {
            const float choose = rf();
            const float cx = a + 0.9f * rf();
            const float cy = b + 0.9f * rf();
            const Vec3 center(cx, cy, 0.2f);
            if (!(length(center - Vec3(4, 0, 0.2f)) > 0.9f)) continue;
            if (choose < 0.8f) {
                rt_material m = blank_material(RT_MAT_LAMBERTIAN);
                const float r0 = rf() * rf(), g0 = rf() * rf(), b0 = rf() * rf();
                m.albedo = Vec3(r0, g0, b0).pod();
                add(center, 0.2f, m);
            } else if (choose < 0.95f) {
                rt_material m = blank_material(RT_MAT_METAL);
                const float r0 = 0.5f + 0.5f * rf(), g0 = 0.5f + 0.5f * rf(), b0 = 0.5f + 0.5f * rf();
                m.albedo = Vec3(r0, g0, b0).pod();
                m.fuzz = 0.5f * rf();
                add(center, 0.2f, m);
            } else {
                rt_material m = blank_material(RT_MAT_DIELECTRIC);
                m.ir = 1.5f;
                add(center, 0.2f, m);
            }
        }
    }
    rt_material glass = blank_material(RT_MAT_DIELECTRIC);
    glass.ir = 1.5f;
    add(Vec3(0, 0, 1), 1.0f, glass);
    rt_material brown = blank_material(RT_MAT_LAMBERTIAN);
    brown.albedo = Vec3(0.4f, 0.2f, 0.1f).pod();
    add(Vec3(-4, 0, 1), 1.0f, brown);
    rt_material mirror = blank_material(RT_MAT_METAL);
    mirror.albedo = Vec3(0.7f, 0.6f, 0.5f).pod();
    mirror.fuzz = 0.0f;
    add(Vec3(4, 0, 1), 1.0f, mirror);

    if (opt.textured_floor_quad) {
        TextureImage img;
        make_checker_texture(opt.texture_size, img);
        out.textures.push_back(std::move(img));
        rt_material m = blank_material(RT_MAT_METAL);
        m.albedo = Vec3(1, 1, 1).pod();
        m.fuzz = 0.3f;
        m.texture_id = out.textures.size();
        out.materials.push_back(m);
        const float h = static_cast<float>(opt.half_extent) + 2.0f;
        // a quad floating just above the ground sphere's top, under the small spheres
        out.planes.push_back(make_plane(Vec3(-h, -h, 0.0005f), Vec3(2 * h, 0, 0), Vec3(0, 2 * h, 0),
                                        static_cast<int>(out.materials.size()) - 1, RT_PLANE_QUAD));
    }
    out.nodes = build_bvh(out.spheres, out.planes);
}

}  // namespace rtp
