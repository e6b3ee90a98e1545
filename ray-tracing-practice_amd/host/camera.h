// camera.h — host-side mirror of the reference's Camera / ISaver interface
// (include/camera.cuh:31-84,117-139; src/camera.cu:52-216).  Camera::render() is the drop-in
// point: it calls the MI355X render library through the C ABI (include/rtp_amd.h) where the
// reference launches render_kernel.
#pragma once
#include <cstdint>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "../../include/rtp_amd.h"
#include "vec_math.h"

namespace rtp {

// ISaver (include/camera.cuh:31-43).  write_color receives the per-pixel SUM of sample
// radiances and divides by the constructor argument — which the reference's drivers set to
// sqrt_rays_per_pixel, not to the sample count (src/camera.cu:300,357): images are sqrt_spp
// times brighter than a mean.  Kept as is for byte parity.
class Saver {
public:
    explicit Saver(int samples_per_pixel) : divisor_(samples_per_pixel) {}
    virtual ~Saver() = default;
    virtual void write_color(Vec3 pixel_sum) = 0;
    virtual void set_format(int width, int height) = 0;
    static float linear_to_gamma(float linear) { return std::sqrt(linear); }  // src/camera.cu:54
    // ÷n → sqrt → clamp[0,0.999] → ×256 → u8 (src/camera.cu:138-147)
    static void quantize(Vec3 pixel_sum, int divisor, uint8_t rgb[3]);

protected:
    int divisor_;
    int width_ = 0, height_ = 0;
};

// FileSaver: ASCII PPM "P3" (src/camera.cu:56-73).
class PpmFileSaver : public Saver {
public:
    PpmFileSaver(int samples_per_pixel, const std::string &filename);
    void write_color(Vec3 pixel_sum) override;
    void set_format(int width, int height) override;
private:
    std::ofstream out_;
};

// OutStreamSaver: the same P3 text on stdout (src/camera.cu:75-92).
class StdoutSaver : public Saver {
public:
    explicit StdoutSaver(int samples_per_pixel) : Saver(samples_per_pixel) {}
    void write_color(Vec3 pixel_sum) override;
    void set_format(int width, int height) override;
};

// PNGSaver (src/camera.cu:94-126): collects RGB8 and writes a PNG when destroyed.  The reference
// compiles it but never instantiates it; its encoder is the third-party stb_image_write, which is
// not part of this repository — png_writer.cpp is a small encoder of its own (same pixels,
// different compressed bytes).
class PngSaver : public Saver {
public:
    PngSaver(int samples_per_pixel, const std::string &filepath);
    ~PngSaver() override;
    void write_color(Vec3 pixel_sum) override;
    void set_format(int width, int height) override;
private:
    std::vector<uint8_t> pixels_;
    std::string path_;
    size_t count_ = 0;
};

// BinarySaver (src/camera.cu:128-153): int32 width, int32 height, then RGB8 rows, top row first.
// This is the saver both reference drivers use (into files NAMED *.png).
class BinarySaver : public Saver {
public:
    BinarySaver(int samples_per_pixel, const std::string &filepath);
    void write_color(Vec3 pixel_sum) override;
    void set_format(int width, int height) override;
private:
    std::ofstream out_;
};

// cudaMemcpyToSymbol(d_scene_data_const, ...) of gpu_render (src/camera.cu:291): the scene every
// later Camera::render() of this thread traces.
void bind_scene(rt_scene *scene);
rt_scene *bound_scene();

// checkCudaErrors (include/camera.cuh:20-29): message on stderr, then exit(99).
void check_rt(rt_status st, const char *expr, const char *file, int line);
#define RTP_CHECK(expr) ::rtp::check_rt((expr), #expr, __FILE__, __LINE__)

class Camera {
public:
    // Argument order (height, width, ...) as in the reference (include/camera.cuh:119-120).
    Camera(int height, int width, std::unique_ptr<Saver> image_saver, Vec3 camera_pos, Vec3 look_at_point = Vec3(0, 0, 0));

    // src/camera.cu:198-216: render the frame into the DEVICE buffer d_fb (width*height*3 floats,
    // per-pixel sums), copy it back and stream every pixel through the saver, row-major.
    void render(float *d_fb) const;
    // src/camera.cu:171-196
    rt_camera_data build_camera_data() const;

    int image_width, image_height;
    float aspect_ratio;
    int samples_per_pixel = 300;
    int max_depth = 50;
    Vec3 background_color{0, 0, 0};
    float vfov = 60.0f;
    std::unique_ptr<Saver> saver;
    Vec3 origin, look_at;
    mutable rt_timing last_timing{};  // kernel time of the last render() (not in the reference)

private:
    Vec3 vup_{0, 0, 1};  // z-up, fixed (src/camera.cu:164)
};

struct SceneParams;
// Eye / look-at for frame n of num_frames (src/camera.cu:301-315).
void orbit_pose(const SceneParams &p, int frame, Vec3 &eye, Vec3 &target);

// gpu_render (src/camera.cu:290-349): per frame BinarySaver + orbit camera + render, printing
// "n \t ms \t W*H*sqrt_spp^2".  The scene must already be bound.
void gpu_render(const SceneParams &params);

// Animation driver beyond the reference: frames dealt round-robin to num_devices GPUs, saver
// arithmetic on the device, file output overlapped with the next frame.  Same files, byte for byte.
void gpu_render_pipelined(const SceneParams &params, const rt_scene_desc &desc, int num_devices);

// … and the other split: every frame sharded in row bands over num_devices GPUs (<= 0: all) with one RCCL gather per
// frame (rt_context, rt_render_sharded).  Same files, byte for byte.
void gpu_render_sharded(const SceneParams &params, const rt_scene_desc &desc, int num_devices);

}  // namespace rtp
