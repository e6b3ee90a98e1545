/*
 * rtp_host.h — C view of the host-side pieces (librtp_host.so) for non-C++ callers
 * (tests and bench.py bind it with ctypes).  Pure host code: no GPU, no HIP.
 * Everything here mirrors reference host code that STAYS on the host:
 *   config parser        src/main.cu:499-550
 *   scene builder        src/main.cu:62-497
 *   BVH builder          include/bvh_builder.h:10-120
 *   camera / orbit       src/camera.cu:171-196, 301-315
 *   saver arithmetic     src/camera.cu:138-153
 */
#ifndef RTP_HOST_H
#define RTP_HOST_H
#include <stdint.h>
#include "../../include/rtp_amd.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rtp_host_scene rtp_host_scene;

typedef struct rtp_config_info {
    int32_t num_frames, width, height, max_depth, sqrt_spp;
    float fov_degrees;
} rtp_config_info;

/* read_scene_params + create_scene on a config text.  texture_dir may be NULL/"". */
rtp_host_scene *rtp_host_scene_from_config(const char *config_text, const char *texture_dir);
/* Benchmark scenes: half_extent 11 → S-rtiow (486 spheres); 158 → S-100k. */
rtp_host_scene *rtp_host_scene_rtiow(uint32_t seed, int32_t half_extent, int32_t textured_quad, int32_t texture_size);
/* A scene from caller-made primitives: spheres = n x (cx,cy,cz,radius,material), planes = n x
 * (base xyz, u xyz, v xyz, material, type) — normal/D/w are computed like the reference's PlaneData
 * constructor — and materials in the rt_material layout; the BVH is built by build_bvh(). */
rtp_host_scene *rtp_host_scene_from_arrays(const float *spheres, int32_t num_spheres, const float *planes, int32_t num_planes,
                                           const rt_material *materials, int32_t num_materials);
void rtp_host_scene_free(rtp_host_scene *s);

/* Arrays in the layouts rt_scene_create() takes; valid until the scene is freed. */
void rtp_host_scene_desc(rtp_host_scene *s, rt_scene_desc *out);
/* Only meaningful for config scenes (zeros otherwise). */
void rtp_host_scene_config(const rtp_host_scene *s, rtp_config_info *out);
/* CameraData of frame n as gpu_render/cpu_render set it up (orbit pose, spp = sqrt_spp^2,
 * black background). */
void rtp_host_frame_camera(const rtp_host_scene *s, int32_t frame, rt_camera_data *out);
/* Camera::build_camera_data for an explicit pose (z-up). */
void rtp_host_make_camera(int32_t width, int32_t height, float vfov_degrees, const float eye[3], const float target[3],
                          const float background[3], int32_t samples_per_pixel, int32_t max_depth, rt_camera_data *out);
/* ISaver::writeColor arithmetic over num_pixels pixel sums → 3*num_pixels bytes. */
void rtp_host_quantize(const float *fb_sum, int64_t num_pixels, int32_t divisor, uint8_t *rgb8);
/* BinarySaver file image: 8-byte header + RGB8.  Returns 0 on success. */
int32_t rtp_host_write_binary_image(const char *path, const float *fb_sum, int32_t width, int32_t height, int32_t divisor);
int32_t rtp_host_write_png(const char *path, const float *fb_sum, int32_t width, int32_t height, int32_t divisor);
/* Host texture loader (JPEG / PPM / PFM → float RGBA, stbi_loadf rule).  Returns 0 on success and
 * fills width/height; rgba (may be NULL to query the size) receives width*height*4 floats. */
int32_t rtp_host_load_texture(const char *path, int32_t *width, int32_t *height, float *rgba);
/* Text of `main --default`. */
const char *rtp_host_default_config(void);

#ifdef __cplusplus
}
#endif
#endif
