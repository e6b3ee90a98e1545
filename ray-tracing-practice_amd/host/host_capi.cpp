#include "rtp_host.h"

#include <algorithm>
#include <sstream>
#include <string>

#include "camera.h"
#include "png_writer.h"
#include "bvh_builder.h"
#include "scene_builder.h"
#include "scene_params.h"

struct rtp_host_scene {
    rtp::HostScene scene;
    rtp::SceneParams params;
    bool from_config = false;
};

extern "C" {

rtp_host_scene *rtp_host_scene_from_config(const char *config_text, const char *texture_dir) {
    if (!config_text) return nullptr;
    auto *s = new rtp_host_scene;
    std::istringstream in(config_text);
    s->params = rtp::read_scene_params(in);
    s->from_config = true;
    rtp::build_config_scene(s->params, texture_dir ? texture_dir : "", s->scene);
    return s;
}

rtp_host_scene *rtp_host_scene_rtiow(uint32_t seed, int32_t half_extent, int32_t textured_quad, int32_t texture_size) {
    auto *s = new rtp_host_scene;
    rtp::RtiowOptions o;
    o.seed = seed;
    o.half_extent = half_extent;
    o.textured_floor_quad = textured_quad != 0;
    o.texture_size = texture_size > 0 ? texture_size : 1024;
    rtp::build_rtiow_scene(o, s->scene);
    return s;
}

rtp_host_scene *rtp_host_scene_from_arrays(const float *spheres, int32_t num_spheres, const float *planes, int32_t num_planes,
                                           const rt_material *materials, int32_t num_materials) {
    auto *s = new rtp_host_scene;
    for (int32_t i = 0; i < num_spheres; ++i) {
        const float *p = spheres + 5 * i;
        s->scene.spheres.push_back(rtp::make_sphere(rtp::Vec3(p[0], p[1], p[2]), p[3], static_cast<int>(p[4])));
    }
    for (int32_t i = 0; i < num_planes; ++i) {
        const float *p = planes + 11 * i;
        s->scene.planes.push_back(rtp::make_plane(rtp::Vec3(p[0], p[1], p[2]), rtp::Vec3(p[3], p[4], p[5]), rtp::Vec3(p[6], p[7], p[8]),
                                                  static_cast<int>(p[9]), static_cast<int>(p[10])));
    }
    s->scene.materials.assign(materials, materials + num_materials);
    s->scene.nodes = rtp::build_bvh(s->scene.spheres, s->scene.planes);
    return s;
}

void rtp_host_scene_free(rtp_host_scene *s) { delete s; }

void rtp_host_scene_desc(rtp_host_scene *s, rt_scene_desc *out) { *out = s->scene.desc(); }

void rtp_host_scene_config(const rtp_host_scene *s, rtp_config_info *out) {
    *out = rtp_config_info{};
    if (!s->from_config) return;
    out->num_frames = s->params.num_frames;
    out->width = s->params.width;
    out->height = s->params.height;
    out->max_depth = s->params.max_depth;
    out->sqrt_spp = s->params.sqrt_spp;
    out->fov_degrees = s->params.fov_degrees;
}

void rtp_host_frame_camera(const rtp_host_scene *s, int32_t frame, rt_camera_data *out) {
    rtp::Vec3 eye, target;
    rtp::orbit_pose(s->params, frame, eye, target);
    rtp::Camera cam(s->params.height, s->params.width, nullptr, eye, target);
    cam.vfov = s->params.fov_degrees;
    cam.samples_per_pixel = s->params.sqrt_spp * s->params.sqrt_spp;
    cam.max_depth = s->params.max_depth;
    cam.background_color = rtp::Vec3(0, 0, 0);
    *out = cam.build_camera_data();
}

void rtp_host_make_camera(int32_t width, int32_t height, float vfov_degrees, const float eye[3], const float target[3],
                          const float background[3], int32_t samples_per_pixel, int32_t max_depth, rt_camera_data *out) {
    rtp::Camera cam(height, width, nullptr, rtp::Vec3(eye[0], eye[1], eye[2]), rtp::Vec3(target[0], target[1], target[2]));
    cam.vfov = vfov_degrees;
    cam.samples_per_pixel = samples_per_pixel;
    cam.max_depth = max_depth;
    cam.background_color = rtp::Vec3(background[0], background[1], background[2]);
    *out = cam.build_camera_data();
}

void rtp_host_quantize(const float *fb_sum, int64_t num_pixels, int32_t divisor, uint8_t *rgb8) {
    for (int64_t p = 0; p < num_pixels; ++p)
        rtp::Saver::quantize(rtp::Vec3(fb_sum[3 * p], fb_sum[3 * p + 1], fb_sum[3 * p + 2]), divisor, rgb8 + 3 * p);
}

int32_t rtp_host_write_binary_image(const char *path, const float *fb_sum, int32_t width, int32_t height, int32_t divisor) {
    rtp::BinarySaver saver(divisor, path);
    saver.set_format(width, height);
    for (int64_t p = 0; p < static_cast<int64_t>(width) * height; ++p)
        saver.write_color(rtp::Vec3(fb_sum[3 * p], fb_sum[3 * p + 1], fb_sum[3 * p + 2]));
    return 0;
}

int32_t rtp_host_write_png(const char *path, const float *fb_sum, int32_t width, int32_t height, int32_t divisor) {
    rtp::PngSaver saver(divisor, path);
    saver.set_format(width, height);
    for (int64_t p = 0; p < static_cast<int64_t>(width) * height; ++p)
        saver.write_color(rtp::Vec3(fb_sum[3 * p], fb_sum[3 * p + 1], fb_sum[3 * p + 2]));
    return 0;
}

int32_t rtp_host_load_texture(const char *path, int32_t *width, int32_t *height, float *rgba) {
    static thread_local std::string cached_path;
    static thread_local rtp::TextureImage cached;
    if (!path || !width || !height) return 1;
    if (cached_path != path) {
        rtp::TextureImage img;
        if (!rtp::load_texture(path, img)) return 1;
        cached = std::move(img);
        cached_path = path;
    }
    *width = cached.width;
    *height = cached.height;
    if (rgba) std::copy(cached.rgba.begin(), cached.rgba.end(), rgba);
    return 0;
}

const char *rtp_host_default_config(void) {
    static const std::string text = rtp::default_config_text();
    return text.c_str();
}

}  // extern "C"
