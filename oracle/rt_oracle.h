/*
 * rt_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference's per-pixel hot path, used only as the checker in
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under
 * ray-tracing-practice_amd/ may include, link or call it.
 *
 * Pinning.  What of the reference compiles here from its own sources without a stand-in is compiled
 * (oracle/Makefile target `_ref`) and compared with this restatement bit for bit: its vec3 / ray / interval / aabb /
 * hittable_object / sphere / plane / bvh / bvh_builder headers behind oracle/ref_geom.cpp (tests/test_ref_geom.py: AABB::hit,
 * the vector algebra, hit_sphere, hit_plane, hit_bvh, build_bvh — a million crafted and random inputs, plus a recorded fixture
 * that runs everywhere), and its vendored stb_image.h behind oracle/ref_stb_decode.c.  random_utils.h, materials.h
 * (<curand_kernel.h>) and camera.cuh (<cuda_runtime.h>) do not compile in this image and stand-ins are not allowed: the RNG,
 * material_scatter, get_ray / build_camera_data and the frame loop are pinned only by the known answers SURVEY.md §4 recorded
 * from the reference's own CPU path (wang_hash / random_float vectors, the CameraData of the create_test_config.py scene, the
 * sha256 of two whole BinarySaver files) — tests/test_oracle_pins.py.  By the rule that makes the oracle as a whole
 * "parity unpinned": those records are not fixtures the reference holds.
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

/* Batched views of the primitive restatements (AABB::hit, vec3 operator/ / unit_vector / reflect / refract / near_zero / dot /
 * cross / len, Interval::contains, Ray::at, set_face_normal, hit_sphere, hit_plane, hit_bvh) with the argument lists of
 * oracle/ref_geom.cpp — the same calls made on the reference's own headers; tests/test_ref_geom.py compares the two bit for bit. */
void orc_geom_aabb_hit(int64_t n, const float *boxes, const float *origins, const float *dirs, const float *tmin, const float *tmax, int32_t *out);
void orc_geom_vec3_div(int64_t n, const float *a, const float *t, float *out);
void orc_geom_unit_vector(int64_t n, const float *a, float *out);
void orc_geom_reflect(int64_t n, const float *a, const float *nrm, float *out);
void orc_geom_refract(int64_t n, const float *a, const float *nrm, const float *eta, float *out);
void orc_geom_near_zero(int64_t n, const float *a, int32_t *out);
void orc_geom_dot_cross_len(int64_t n, const float *a, const float *b, float *out_dot, float *out_cross, float *out_len);
void orc_geom_contains(int64_t n, const float *lo, const float *hi, const float *x, int32_t *out);
void orc_geom_ray_at(int64_t n, const float *origins, const float *dirs, const float *t, float *out);
void orc_geom_set_face_normal(int64_t n, const float *dirs, const float *outward, float *out_normal, int32_t *out_front);
void orc_geom_hit_sphere(int64_t n, const float *origins, const float *dirs, const float *tmin, const float *tmax, const rt_sphere *spheres,
                         int32_t *out_hit, float *out_rec9, int32_t *out_code);
void orc_geom_hit_plane(int64_t n, const float *origins, const float *dirs, const float *tmin, const float *tmax, const rt_plane *planes,
                        int32_t *out_hit, float *out_rec9, int32_t *out_code);
void orc_geom_hit_bvh(const rt_scene_desc *scene, int64_t n, const float *origins, const float *dirs, float tmin, float tmax,
                      int32_t *out_hit, float *out_rec9, int32_t *out_code);

#ifdef __cplusplus
}
#endif
#endif
