/*
 * rt_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's per-pixel hot path, used only as the checker in
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
 * ray-tracing-practice_amd/ may include, link or call it.
 *
 * Pinning: the reference itself cannot be built in this image without
 * writing stand-ins for cuda_runtime.h / curand_kernel.h, which is not allowed, so this oracle is
 * pinned by the known answers SURVEY.md §4 recorded from the reference's own CPU path
 * (wang_hash / random_float vectors, the CameraData of the create_test_config.py scene, and the
 * sha256 of the 60 008-byte BinarySaver file of that scene) — tests/test_oracle_pins.py.
 * Not covered by any pin: tex2D_cpu (no reference output with a texture is reproducible here
 * without the reference's vendored JPEG decoder) — "parity unpinned" for the textured branch.
 *
 * Data layouts are the reference's own (include/rtp_amd.h documents offsets and cites them).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>
#include "../include/rtp_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Work counters of the instrumented traversal (feed the algorithmic
 * bytes-per-sample figure of the roofline). */
typedef struct orc_stats {
    uint64_t samples;
    uint64_t rays;          /* hit_scene calls */
    uint64_t node_visits;   /* nodes popped and loaded in hit_bvh (include/bvh.h:34) */
    uint64_t box_hits;      /* AABB::hit returned true */
    uint64_t sphere_tests;
    uint64_t plane_tests;
    uint64_t material_fetches;
    uint64_t texture_fetches;
    uint32_t max_stack;
} orc_stats;

uint32_t orc_wang_hash(uint32_t seed);                 /* include/random_utils.h:7-14 */
float    orc_random_float(uint32_t *seed);             /* include/random_utils.h:16-19 */

/* CameraData::get_ray, include/camera.cuh:97-109. */
void orc_get_ray(const rt_camera_data *cam, int i, int j, uint32_t *seed, float origin[3], float dir[3]);

/* tex2D_cpu, include/materials.h:20-51 (x0,y0 additionally wrapped: the reference reads out of
 * bounds there). */
void orc_tex2d(const rt_texture *tex, float u, float v, float rgb[3]);

/* hit_scene → hit_bvh (include/scene.h:23-35, include/bvh.h:19-65).  Returns 1 on hit and fills
 * t, primitive type (0 sphere / 1 plane) and index. */
int orc_closest_hit(const rt_scene_desc *scene, const float origin[3], const float dir[3],
                    float *t, int *prim_type, int *prim_index);

/* ray_color_host for one (i,j,s) sample (src/camera.cu:41-45,254-288). */
void orc_trace_sample(const rt_scene_desc *scene, const rt_camera_data *cam, int i, int j, int s,
                      float radiance[3], int32_t *rays, uint32_t *final_seed);

/* Camera::render_cpu (src/camera.cu:36-50) over rows [row0,row1); fb_sum holds
 * (row1-row0)*width*3 floats.  num_threads > 1 splits rows between pthreads (each pixel is still
 * computed exactly as the serial loop does).  stats may be NULL. */
void orc_render(const rt_scene_desc *scene, const rt_camera_data *cam, int row0, int row1,
                float *fb_sum, int num_threads, orc_stats *stats);

/* ISaver::writeColor arithmetic (src/camera.cu:138-153): 3 floats → 3 bytes. */
void orc_write_color(const float rgb_sum[3], int divisor, uint8_t out[3]);

/* Brute-force closest hit over every primitive with the same primitive tests and the same
 * leaf-box gate — used to show that the result of hit_bvh does not depend on traversal order. */
int orc_closest_hit_bruteforce(const rt_scene_desc *scene, const float origin[3], const float dir[3],
                               float *t, int *prim_type, int *prim_index);

#ifdef __cplusplus
}
#endif
#endif
