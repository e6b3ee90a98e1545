#include "scene_params.h"

#include <istream>
#include <sstream>

namespace rtp {
namespace {
void read_vec(std::istream &in, Vec3 &v) { in >> v.x >> v.y >> v.z; }
void read_orbit(std::istream &in, OrbitParams &o) {
    in >> o.r0 >> o.z0 >> o.phi0;
    in >> o.amp_r >> o.amp_z;
    in >> o.w_r >> o.w_z >> o.w_phi;
    in >> o.phase_r >> o.phase_z;
}
}  // namespace

SceneParams read_scene_params(std::istream &in) {
    SceneParams p;
    in >> p.num_frames >> p.output_pattern;
    in >> p.width >> p.height >> p.fov_degrees;
    read_orbit(in, p.eye);
    read_orbit(in, p.target);

    p.bodies.resize(3);
    for (BodyParams &b : p.bodies) {
        read_vec(in, b.center);
        read_vec(in, b.colour);
        in >> b.radius >> b.reflection >> b.transparency >> b.lights_per_edge;
    }
    for (Vec3 &c : p.floor.corners) read_vec(in, c);
    in >> p.floor.texture_path;
    read_vec(in, p.floor.tint);
    in >> p.floor.reflection;

    int num_lights = 0;
    in >> num_lights;
    if (num_lights > 4) num_lights = 4;
    if (num_lights < 0) num_lights = 0;
    p.lights.resize(static_cast<size_t>(num_lights));
    for (LightParams &l : p.lights) {
        read_vec(in, l.position);
        read_vec(in, l.colour);
    }
    in >> p.max_depth >> p.sqrt_spp;
    return p;
}

std::string default_config_text() {
    std::ostringstream o;
    o << 100 << "\n"
      << "/home/zloyaloha/development/ray-tracing-practice/images/render_%d.png\n"
      << "1080 720 50\n"
      << "15.0 4.5 3.14159    0.0 4.5    0.0 1.0 1.0    0.0 -1.57\n"
      << "0.0 4.5 0.0    0.0 4.5    0.0 1.0 0.0    0.0 -1.57\n"
      << "0.0 0.0 3.0     0.3 0.0 0.0     3.0     1.5     0.1     3\n"
      << "4 0.0 6.0     0.0 0.3 0.0     3.0     1.2     0.1     2\n"
      << "8 0.0 9.0     0.0 0.0 0.3     3.0     1     0.1     1\n"
      << "-15.0 -15.0 -1.0      -15.0 15.0 -1.0       15.0 15.0 -1.0        15.0 -15.0 -1.0 ../floor2.jpg\n"
      << "1.0 1.0 1.0\n"
      << "0.3\n"
      << "4\n"
      << "-15.0 -15.0 1  10.0 10.0 10.0\n"
      << "-15.0 15.0 1   10.0 10.0 10.0\n"
      << "15.0 15.0 1    10.0 10.0 10.0\n"
      << "15.0 -15.0 1   10.0 10.0 10.0\n"
      << "50 50\n";
    return o.str();
}

}  // namespace rtp
