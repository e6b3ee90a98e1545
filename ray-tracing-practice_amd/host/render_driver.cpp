// render_driver.cpp — the part of the host mirror that talks to the GPU library through the C ABI:
// Camera::render (src/camera.cu:198-216), gpu_render (src/camera.cu:290-349) and the
// checkCudaErrors equivalent.  Kept apart from camera.cpp so the pure-host library
// (librtp_host.so) has no dependency on librtp_amd.so.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <mutex>
#include <thread>

#include <unistd.h>

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
    rt_timing_init(&last_timing);
    RTP_CHECK(rt_render(bound_scene(), &cam, nullptr, d_fb, nullptr, 1, &last_timing));

    std::vector<float> host_fb(num_pixels * 3);
    RTP_CHECK(rt_copy_to_host(host_fb.data(), d_fb, num_pixels * 3 * sizeof(float)));
    if (!saver) return;
    for (size_t p = 0; p < num_pixels; ++p)
        saver->write_color(Vec3(host_fb[3 * p], host_fb[3 * p + 1], host_fb[3 * p + 2]));
}


// Frame file name.  The reference hands the pattern it read from the config straight to
// snprintf(filename, 256, pattern, n) (src/camera.cu:298-299): a pattern with anything but one integer conversion
// is undefined behaviour there.  Same result for the patterns that are defined — "%d", "%5d", "%03d", "%%", the
// 255-character truncation — and a clean failure (message + exit 99, like every other fatal error of the driver)
// for the rest, instead of passing untrusted text to printf.
std::string frame_filename(const std::string &pattern, int n) {
    std::string out;
    int conversions = 0;
    for (size_t k = 0; k < pattern.size(); ++k) {
        if (pattern[k] != '%') { out.push_back(pattern[k]); continue; }
        if (k + 1 < pattern.size() && pattern[k + 1] == '%') { out.push_back('%'); ++k; continue; }
        size_t e = k + 1;
        std::string spec = "%";
        if (e < pattern.size() && (pattern[e] == '0' || pattern[e] == '-')) spec.push_back(pattern[e++]);
        int digits = 0;
        while (e < pattern.size() && pattern[e] >= '0' && pattern[e] <= '9' && digits < 3) { spec.push_back(pattern[e++]); ++digits; }
        if (e >= pattern.size() || (pattern[e] != 'd' && pattern[e] != 'i') || ++conversions > 1) {
            std::fprintf(stderr, "output path pattern '%s': only one %%d (optionally %%0Nd / %%Nd) is supported\n", pattern.c_str());
            std::exit(99);
        }
        spec.push_back('d');
        char buf[32];
        std::snprintf(buf, sizeof(buf), spec.c_str(), n);
        out += buf;
        k = e;
    }
    if (out.size() > 255) out.resize(255);
    return out;
}

void gpu_render(const SceneParams &params) {
    float *d_fb = nullptr;
    const size_t num_pixels = static_cast<size_t>(params.width) * params.height;
    RTP_CHECK(rt_device_alloc(num_pixels * 3 * sizeof(float), reinterpret_cast<void **>(&d_fb)));

    for (int n = 0; n < params.num_frames; ++n) {
        const std::string filename = frame_filename(params.output_pattern, n);
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

// ---- animation driver ("next" rows f1 + f2 of SURVEY.md §8) ---------------------------------------
// The reference renders its frames one after the other on one GPU and spends most of each frame's
// wall time pushing pixels through three 1-byte ofstream writes (src/camera.cu:211-215,148-152).
// Frames are independent (each has its own camera pose and output file), so here
//   * frames are dealt round-robin to `num_devices` GPUs, one host thread and one rt_scene per GPU;
//   * the saver arithmetic runs on the device (rt_tonemap: byte-exact ISaver::writeColor), so
//     the D2H copy is RGB8 (4x smaller than the float sums);
//   * the file of frame n is written by a writer thread while frame n+1 renders.
// The files are byte-identical to what gpu_render() above (and the reference) writes.
namespace {

struct PendingFile {
    std::string path;
    int width = 0, height = 0;
    std::vector<uint8_t> rgb;
};

void write_binary_frame(const PendingFile &f) {
    std::ofstream out(f.path, std::ios::binary);
    const int32_t hdr[2] = {f.width, f.height};     // BinarySaver::setFormat (src/camera.cu:131-136)
    out.write(reinterpret_cast<const char *>(hdr), sizeof(hdr));
    out.write(reinterpret_cast<const char *>(f.rgb.data()), static_cast<std::streamsize>(f.rgb.size()));
}

}  // namespace

void gpu_render_pipelined(const SceneParams &params, const rt_scene_desc &desc, int num_devices) {
    if (num_devices < 1) num_devices = 1;
    std::mutex print_mutex;
    auto worker = [&](int dev) {
        RTP_CHECK(rt_set_device(dev));
        rt_scene *scene = nullptr;
        RTP_CHECK(rt_scene_create(&desc, &scene));
        const size_t num_pixels = static_cast<size_t>(params.width) * params.height;
        float *d_fb = nullptr;
        uint8_t *d_rgb = nullptr;
        RTP_CHECK(rt_device_alloc(num_pixels * 3 * sizeof(float), reinterpret_cast<void **>(&d_fb)));
        RTP_CHECK(rt_device_alloc(num_pixels * 3, reinterpret_cast<void **>(&d_rgb)));
        std::thread writer;
        for (int n = dev; n < params.num_frames; n += num_devices) {
            const std::string filename = frame_filename(params.output_pattern, n);
            Vec3 eye, target;
            orbit_pose(params, n, eye, target);
            Camera camera(params.height, params.width, nullptr, eye, target);
            camera.vfov = params.fov_degrees;
            camera.samples_per_pixel = params.sqrt_spp * params.sqrt_spp;
            camera.max_depth = params.max_depth;
            camera.background_color = Vec3(0, 0, 0);
            const rt_camera_data cam = camera.build_camera_data();

            const auto t0 = std::chrono::steady_clock::now();
            rt_timing timing;
            rt_timing_init(&timing);
            RTP_CHECK(rt_render(scene, &cam, nullptr, d_fb, nullptr, 1, &timing));
            RTP_CHECK(rt_tonemap(d_fb, d_rgb, static_cast<int64_t>(num_pixels) * 3, params.sqrt_spp, nullptr));
            auto file = std::make_shared<PendingFile>();
            file->path = filename;
            file->width = params.width;
            file->height = params.height;
            file->rgb.resize(num_pixels * 3);
            RTP_CHECK(rt_copy_to_host(file->rgb.data(), d_rgb, num_pixels * 3));
            if (writer.joinable()) writer.join();           // at most one file in flight per GPU
            writer = std::thread([file]() { write_binary_frame(*file); });
            const auto t1 = std::chrono::steady_clock::now();
            const float ms = std::chrono::duration<float, std::milli>(t1 - t0).count();
            const long long total_rays = static_cast<long long>(params.width) * params.height * params.sqrt_spp * params.sqrt_spp;
            std::lock_guard<std::mutex> lock(print_mutex);
            std::cout << n << "\t" << ms << "\t" << total_rays << "\n";
        }
        if (writer.joinable()) writer.join();
        rt_device_free(d_fb);
        rt_device_free(d_rgb);
        RTP_CHECK(rt_scene_destroy(scene));
    };
    std::vector<std::thread> threads;
    for (int d = 1; d < num_devices; ++d) threads.emplace_back(worker, d);
    worker(0);
    for (std::thread &t : threads) t.join();
}

// ---- one frame over all GPUs (BASELINE configs[3]) ------------------------------------------
// The other way to use a node: every frame is split into interleaved 8-row bands over `num_devices` GPUs
// (rt_context / rt_render_sharded: scene replicated, one RCCL gather per frame to the root GPU), then the root runs
// the saver arithmetic on the device and the file is written while the next frame renders.  Same files, byte for byte.
void gpu_render_sharded(const SceneParams &params, const rt_scene_desc &desc, int num_devices) {
    rt_context *ctx = nullptr;
    {
        // RCCL prints a version banner on stdout when the first communicator is made; stdout is this program's data channel
        // (the per-frame TSV, src/camera.cu:346).  No other thread of this process exists yet, so fd 1 can point at fd 2 for
        // the duration of the call — the application's business, not the library's.
        std::cout.flush();
        fflush(stdout);
        const int saved_stdout = dup(1);
        if (saved_stdout >= 0) (void)dup2(2, 1);
        const rt_status st = rt_context_create(num_devices, nullptr, &ctx);
        fflush(stdout);
        if (saved_stdout >= 0) { (void)dup2(saved_stdout, 1); (void)close(saved_stdout); }
        RTP_CHECK(st);
    }
    RTP_CHECK(rt_context_scene_create(ctx, &desc, nullptr));
    const int n = rt_context_num_devices(ctx);
    const size_t num_pixels = static_cast<size_t>(params.width) * params.height;
    float *d_fb = nullptr;
    uint8_t *d_rgb = nullptr;
    RTP_CHECK(rt_set_device(0));
    RTP_CHECK(rt_device_alloc(num_pixels * 3 * sizeof(float), reinterpret_cast<void **>(&d_fb)));
    RTP_CHECK(rt_device_alloc(num_pixels * 3, reinterpret_cast<void **>(&d_rgb)));
    std::thread writer;
    std::vector<rt_timing> timings(static_cast<size_t>(n));
    for (rt_timing &t : timings) rt_timing_init(&t);
    for (int f = 0; f < params.num_frames; ++f) {
        const std::string filename = frame_filename(params.output_pattern, f);
        Vec3 eye, target;
        orbit_pose(params, f, eye, target);
        Camera camera(params.height, params.width, nullptr, eye, target);
        camera.vfov = params.fov_degrees;
        camera.samples_per_pixel = params.sqrt_spp * params.sqrt_spp;
        camera.max_depth = params.max_depth;
        camera.background_color = Vec3(0, 0, 0);
        const rt_camera_data cam = camera.build_camera_data();
        const auto t0 = std::chrono::steady_clock::now();
        RTP_CHECK(rt_render_sharded(ctx, &cam, 8, d_fb, timings.data()));
        RTP_CHECK(rt_tonemap(d_fb, d_rgb, static_cast<int64_t>(num_pixels) * 3, params.sqrt_spp, nullptr));
        auto file = std::make_shared<PendingFile>();
        file->path = filename;
        file->width = params.width;
        file->height = params.height;
        file->rgb.resize(num_pixels * 3);
        RTP_CHECK(rt_copy_to_host(file->rgb.data(), d_rgb, num_pixels * 3));
        if (writer.joinable()) writer.join();
        writer = std::thread([file]() { write_binary_frame(*file); });
        const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        const long long total_rays = static_cast<long long>(params.width) * params.height * params.sqrt_spp * params.sqrt_spp;
        std::cout << f << "\t" << ms << "\t" << total_rays << "\n";
    }
    if (writer.joinable()) writer.join();
    rt_device_free(d_fb);
    rt_device_free(d_rgb);
    RTP_CHECK(rt_context_destroy(ctx));
}

}  // namespace rtp
