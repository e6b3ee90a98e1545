// scene_params.h — the stdin scene description of the reference CLI
// (include/scene_params.h:8-58, parser src/main.cu:499-550, --default text src/main.cu:552-570).
#pragma once
#include <iosfwd>
#include <string>
#include <vector>
#include "vec_math.h"

namespace rtp {

// r(t) = r0 + A_r sin(w_r t + p_r), z(t) = z0 + A_z sin(w_z t + p_z), phi(t) = phi0 + w t
// (src/camera.cu:301-315); one set for the eye, one for the look-at point.
struct OrbitParams {
    float r0 = 0, z0 = 0, phi0 = 0;
    float amp_r = 0, amp_z = 0;
    float w_r = 0, w_z = 0, w_phi = 0;
    float phase_r = 0, phase_z = 0;
};

struct BodyParams {
    Vec3 center, colour;
    float radius = 0, reflection = 0, transparency = 0;
    int lights_per_edge = 0;
};

struct FloorParams {
    Vec3 corners[4];
    std::string texture_path;
    Vec3 tint;
    float reflection = 0;
};

struct LightParams {
    Vec3 position, colour;
};

struct SceneParams {
    int num_frames = 0;
    std::string output_pattern;  // printf pattern with one %d
    int width = 0, height = 0;
    float fov_degrees = 0;
    OrbitParams eye, target;
    std::vector<BodyParams> bodies;  // always 3: octahedron, cube, dodecahedron
    FloorParams floor;
    std::vector<LightParams> lights;  // at most 4
    int max_depth = 0;
    int sqrt_spp = 0;
};

// Whitespace-token parser, same token order as src/main.cu:499-550.  A light count above 4 is
// clamped WITHOUT consuming the surplus light lines (src/main.cu:538-540), as the reference does.
SceneParams read_scene_params(std::istream &in);

// The text `main --default` prints (src/main.cu:552-570).
std::string default_config_text();

}  // namespace rtp
