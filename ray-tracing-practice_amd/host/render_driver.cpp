// render_driver.cpp — the part of the host mirror that talks to the GPU library through the C ABI:
// Camera::render (src/camera.cu:198-216), gpu_render (src/camera.cu:290-349) and the
// checkCudaErrors equivalent.  Kept apart from camera.cpp so the pure-host library
// (librtp_host.so) has no dependency on librtp_amd.so.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "camera.h"
#include "scene_params.h"

namespace rtp {

// ---- error handling ---------------------------------------------------------------------------

void check_rt(rt_status st, const char *expr, const char *file, int line) {
    if (st == RT_OK) return;
    std::cerr << "RT error = " << static_cast<unsigned>(st) << " at " << file << ":" << line << " '" << expr << "' "
              << rt_get_last_error_string() << "\n";
    std::exit(99);
}

namespace { thread_local rt_scene *g_bound_scene = nullptr; }
void bind_scene(rt_scene *scene) { g_bound_scene = scene; }
rt_scene *bound_scene() { return g_bound_scene; }

void Camera::render(float *d_fb) const {
    const size_t num_pixels = static_cast<size_t>(image_width) * image_height;
    const rt_camera_data cam = build_camera_data();  // cudaMemcpyToSymbol(d_cam_data_const), src/camera.cu:324-325
    RTP_CHECK(rt_render(bound_scene(), &cam, nullptr, d_fb, nullptr, 1, &last_timing));

    std::vector<float> host_fb(num_pixels * 3);
    RTP_CHECK(rt_copy_to_host(host_fb.data(), d_fb, num_pixels * 3 * sizeof(float)));
    if (!saver) return;
    for (size_t p = 0; p < num_pixels; ++p)
        saver->write_color(Vec3(host_fb[3 * p], host_fb[3 * p + 1], host_fb[3 * p + 2]));
}


void gpu_render(const SceneParams &params) {
    float *d_fb = nullptr;
    const size_t num_pixels = static_cast<size_t>(params.width) * params.height;
    RTP_CHECK(rt_device_alloc(num_pixels * 3 * sizeof(float), reinterpret_cast<void **>(&d_fb)));

    for (int n = 0; n < params.num_frames; ++n) {
        char filename[256];
        snprintf(filename, sizeof(filename), params.output_pattern.c_str(), n);
        auto saver = std::make_unique<BinarySaver>(params.sqrt_spp, filename);
        Vec3 eye, target;
        orbit_pose(params, n, eye, target);

        Camera camera(params.height, params.width, std::move(saver), eye, target);
        camera.vfov = params.fov_degrees;
        camera.samples_per_pixel = params.sqrt_spp * params.sqrt_spp;
        camera.max_depth = params.max_depth;
        camera.background_color = Vec3(0, 0, 0);

        // the reference brackets Camera::render — kernel, D2H and saver I/O — with events
        const auto t0 = std::chrono::steady_clock::now();
        camera.render(d_fb);
        const auto t1 = std::chrono::steady_clock::now();
        const float ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
        const long long total_rays = static_cast<long long>(params.width) * params.height * params.sqrt_spp * params.sqrt_spp;
        std::cout << n << "\t" << ms << "\t" << total_rays << "\n";
    }
    rt_device_free(d_fb);  // unchecked in the reference too (src/camera.cu:348)
}

}  // namespace rtp
