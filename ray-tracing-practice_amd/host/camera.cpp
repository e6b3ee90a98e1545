#include "camera.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "png_writer.h"
#include "scene_params.h"

namespace rtp {

// ---- savers ---------------------------------------------------------------------------------

void Saver::quantize(Vec3 pixel_sum, int divisor, uint8_t rgb[3]) {
    const Vec3 mean = div_scalar(pixel_sum, static_cast<float>(divisor));  // pixel_color / samplesPerPixel
    const float ch[3] = {linear_to_gamma(mean.x), linear_to_gamma(mean.y), linear_to_gamma(mean.z)};
    for (int k = 0; k < 3; ++k) {
        float c = ch[k];
        if (ch[k] < 0.0f) c = 0.0f;      // Interval(0.0, 0.999).clamp (include/interval.h:18-22)
        if (ch[k] > 0.999f) c = 0.999f;
        rgb[k] = static_cast<uint8_t>(256 * c);
    }
}

PpmFileSaver::PpmFileSaver(int spp, const std::string &filename) : Saver(spp), out_(filename) {}
void PpmFileSaver::set_format(int w, int h) {
    width_ = w; height_ = h;
    out_ << "P3\n" << w << ' ' << h << "\n255\n";
}
void PpmFileSaver::write_color(Vec3 s) {
    uint8_t c[3];
    quantize(s, divisor_, c);
    out_ << int(c[0]) << ' ' << int(c[1]) << ' ' << int(c[2]) << '\n';
}

void StdoutSaver::set_format(int w, int h) {
    width_ = w; height_ = h;
    std::cout << "P3\n" << w << ' ' << h << "\n255\n";
}
void StdoutSaver::write_color(Vec3 s) {
    uint8_t c[3];
    quantize(s, divisor_, c);
    std::cout << int(c[0]) << ' ' << int(c[1]) << ' ' << int(c[2]) << '\n';
}

PngSaver::PngSaver(int spp, const std::string &filepath) : Saver(spp), path_(filepath) {}
void PngSaver::set_format(int w, int h) {
    width_ = w; height_ = h;
    pixels_.assign(static_cast<size_t>(w) * h * 3, 0);
    count_ = 0;
}
void PngSaver::write_color(Vec3 s) {
    if ((count_ + 1) * 3 > pixels_.size()) return;
    quantize(s, divisor_, &pixels_[count_ * 3]);
    ++count_;
}
PngSaver::~PngSaver() {
    if (!pixels_.empty()) write_png_rgb8(path_, width_, height_, pixels_.data());
}

BinarySaver::BinarySaver(int spp, const std::string &filepath) : Saver(spp), out_(filepath, std::ios::binary) {}
void BinarySaver::set_format(int w, int h) {
    width_ = w; height_ = h;
    const int32_t hdr[2] = {w, h};
    out_.write(reinterpret_cast<const char *>(hdr), sizeof(hdr));
}
void BinarySaver::write_color(Vec3 s) {
    uint8_t c[3];
    quantize(s, divisor_, c);
    out_.write(reinterpret_cast<const char *>(c), 3);
}

// ---- camera ------------------------------------------------------------------------------------

Camera::Camera(int height, int width, std::unique_ptr<Saver> image_saver, Vec3 camera_pos, Vec3 look_at_point)
    : image_width(width), image_height(height), aspect_ratio(static_cast<float>(width) / height),
      saver(std::move(image_saver)), origin(camera_pos), look_at(look_at_point) {
    if (saver) saver->set_format(image_width, image_height);
}

rt_camera_data Camera::build_camera_data() const {
    const float pi = 3.1415926535897932385;                            // src/camera.cu:12 (a float there)
    const float theta = static_cast<float>(vfov * pi / 180.0);          // float*float, then double divide
    const float h = tanf(theta / 2);
    const float viewport_height = static_cast<float>(2.0 * h);
    const float viewport_width = viewport_height * (static_cast<float>(image_width) / image_height);

    const Vec3 w = normalized(origin - look_at);
    const Vec3 u = normalized(cross(vup_, w));
    const Vec3 v = cross(w, u);
    const Vec3 horizontal = viewport_width * u;
    const Vec3 vertical = viewport_height * v;

    rt_camera_data d;
    const Vec3 du = div_scalar(horizontal, static_cast<float>(image_width));
    const Vec3 dv = div_scalar(-vertical, static_cast<float>(image_height));
    const Vec3 upper_left = origin - w - div_scalar(horizontal, 2.0f) + div_scalar(vertical, 2.0f);
    d.origin = origin.pod();
    d.pixel_delta_u = du.pod();
    d.pixel_delta_v = dv.pod();
    d.pixel00_loc = (upper_left + 0.5f * (du + dv)).pod();
    d.background = background_color.pod();
    d.image_width = image_width;
    d.image_height = image_height;
    d.samples_per_pixel = samples_per_pixel;
    d.max_depth = max_depth;
    return d;
}

// ---- frame driver ------------------------------------------------------------------------------

void orbit_pose(const SceneParams &p, int frame, Vec3 &eye, Vec3 &target) {
    const float t = static_cast<float>((static_cast<float>(frame) / p.num_frames) * 2.0f * M_PI);
    auto pose = [t](const OrbitParams &o) {
        const float r = o.r0 + o.amp_r * sinf(o.w_r * t + o.phase_r);
        const float z = o.z0 + o.amp_z * sinf(o.w_z * t + o.phase_z);
        const float phi = o.phi0 + o.w_phi * t;
        return Vec3(r * cosf(phi), r * sinf(phi), z);
    };
    eye = pose(p.eye);
    target = pose(p.target);
}

}  // namespace rtp
