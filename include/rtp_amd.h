/*
 * rtp_amd.h — C ABI of the MI355X-native path-tracing render library (librtp_amd.so).
 *
 * Drop-in boundary: this library replaces the reference's device side of
 *     void Camera::render(color *d_fb) const        (reference include/camera.cuh:122, src/camera.cu:198-216)
 * i.e. the launch of render_kernel (src/camera.cu:17-34) and everything it calls
 * (ray_color src/camera.cu:218-252, hit_scene include/scene.h:23, hit_bvh include/bvh.h:19,
 * hit_sphere include/sphere.h:24, hit_plane include/plane.h:57, material_scatter
 * include/materials.h:70, the RNG of include/random_utils.h:7-42).
 *
 * The reference passes its inputs through two __constant__ symbols
 * (d_cam_data_const / d_scene_data_const, src/camera.cu:14-15, written at :291 and :325) whose
 * scene pointers were cudaMalloc'd by create_scene (src/main.cu:429-474).  Here the same data are
 * passed explicitly: rt_scene_create() takes the host arrays create_scene builds, in the reference's
 * own struct layouts, and rt_render() takes the 76-byte CameraData the reference uploads per frame.
 *
 * Plain C types only; caller owns every buffer; no function exits the process
 * (the reference's checkCudaErrors calls exit(99), include/camera.cuh:20-29 — the host-side
 * mirror in ray-tracing-practice_amd/host reproduces that behaviour on top of these status codes).
 */
#ifndef RTP_AMD_H
#define RTP_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------------- */
typedef enum rt_status {
    RT_OK = 0,
    RT_ERR_INVALID_ARG = 1,   /* null pointer, negative count, index out of range in the scene */
    RT_ERR_NO_DEVICE = 2,     /* no HIP device / device ordinal out of range */
    RT_ERR_HIP = 3,           /* a HIP runtime call failed; see rt_get_last_error_string() */
    RT_ERR_UNSUPPORTED = 4,   /* scene exceeds a documented limit of this build */
    RT_ERR_OUT_OF_MEMORY = 5
} rt_status;

/* ---- reference data layouts (sizes/offsets measured on the reference's headers, x86-64) --- */

/* vec3 / point3 / color: include/vec3.h:18 — three packed floats, 12 bytes. */
typedef struct rt_vec3 { float e[3]; } rt_vec3;

/* SphereData: include/sphere.h:8-14 — __align__(16), 32 bytes. */
typedef struct rt_sphere {
    rt_vec3 center;        /* +0  */
    float   radius;        /* +12 */
    int32_t material_idx;  /* +16 */
    int32_t _pad[3];       /* +20 (alignment tail of the reference struct) */
} rt_sphere;

/* PlaneType: include/plane.h:7. */
enum { RT_PLANE_QUAD = 0, RT_PLANE_ELLIPSE = 1, RT_PLANE_TRIANGLE = 2 };

/* PlaneData: include/plane.h:9-17 — __align__(16), 80 bytes.  normal, D and w are the values
 * the reference's host-only constructor precomputes (include/plane.h:19-28). */
typedef struct rt_plane {
    int32_t type;          /* +0  */
    float   D;             /* +4  dot(normal, base) */
    int32_t material_idx;  /* +8  */
    rt_vec3 w;             /* +12 n / dot(n,n), n = cross(u,v) */
    rt_vec3 u;             /* +24 */
    rt_vec3 v;             /* +36 */
    rt_vec3 base;          /* +48 */
    rt_vec3 normal;        /* +60 unit_vector(n) */
    int32_t _pad[2];       /* +72 */
} rt_plane;

/* MaterialType: include/materials.h:12. */
enum { RT_MAT_LAMBERTIAN = 0, RT_MAT_METAL = 1, RT_MAT_DIELECTRIC = 2, RT_MAT_DIFFUSE_LIGHT = 3 };

/* MaterialData: include/materials.h:53-62 — 64 bytes.  The reference's two trailing handles
 * (cudaTextureObject_t tex_obj at +48, CpuTexture* cpu_tex at +56) are process-local; here +48
 * holds a 1-based index into rt_scene_desc.textures (0 = untextured) and +56 is reserved (0). */
typedef struct rt_material {
    int32_t  type;         /* +0  */
    float    fuzz;         /* +4  */
    float    ir;           /* +8  */
    rt_vec3  absorption;   /* +12 */
    rt_vec3  albedo;       /* +24 */
    rt_vec3  emit;         /* +36 */
    uint64_t texture_id;   /* +48 */
    uint64_t reserved;     /* +56 */
} rt_material;

/* BVHNode: include/bvh.h:7-12 — 36 bytes: AABB as x.min,x.max,y.min,y.max,z.min,z.max
 * (include/aabb.h:8, include/interval.h:7-8) then left,right,type.  Leaf <=> left < 0, then
 * right = primitive index and type = 0 sphere / 1 plane; internal nodes carry type = -1
 * (include/bvh_builder.h:61-64,92-94).  Pre-order array, root = node 0. */
typedef struct rt_bvh_node {
    float   box[6];
    int32_t left, right, type;
} rt_bvh_node;

/* CameraData: include/camera.cuh:86-95 — 76 bytes. */
typedef struct rt_camera_data {
    rt_vec3 origin;
    rt_vec3 pixel00_loc;
    rt_vec3 pixel_delta_u;
    rt_vec3 pixel_delta_v;
    rt_vec3 background;
    int32_t image_width;
    int32_t image_height;
    int32_t samples_per_pixel;
    int32_t max_depth;
} rt_camera_data;

/* CpuTexture: include/materials.h:14-18 — float RGBA rows, top row first (stbi_loadf(...,4),
 * src/main.cu:52-60).  Sampling follows tex2D_cpu (include/materials.h:20-51): gfx950 has no
 * texture units, so the device does the same software bilinear fetch from a linear buffer. */
typedef struct rt_texture {
    const float *rgba;
    int32_t width, height;
} rt_texture;

/* What the reference keeps in SceneData + BVHTree (include/scene.h:9-21, include/bvh.h:14-17). */
typedef struct rt_scene_desc {
    const rt_sphere   *spheres;   int32_t num_spheres;
    const rt_plane    *planes;    int32_t num_planes;
    const rt_material *materials; int32_t num_materials;
    const rt_bvh_node *nodes;     int32_t num_nodes;     /* the one BVH tree of the scene */
    const rt_texture  *textures;  int32_t num_textures;
} rt_scene_desc;

/* Which rows of the image one call renders.  Rows are grouped into bands of band_rows rows;
 * band b belongs to part (b % num_parts).  The call renders the rows of `part` and writes them
 * compacted, in increasing row order, into the output buffer (rt_shard_rows() rows of width
 * image_width).  {0,1,0} or a null pointer = the whole image. */
typedef struct rt_shard {
    int32_t band_rows;
    int32_t num_parts;
    int32_t part;
} rt_shard;

/* What a render call did and what it cost.  An OUT structure that grows with the library: the caller sets struct_bytes to the
 * sizeof(rt_timing) it was compiled with (rt_timing_init does) and the library writes at most that many bytes — a caller built
 * against an older, shorter header keeps working.  struct_bytes smaller than the first two fields is RT_ERR_INVALID_ARG. */
typedef struct rt_timing {
    uint32_t struct_bytes;    /* in: sizeof(rt_timing) as the caller compiled it */
    float    kernel_ms;       /* hipEvent time from the first to after the last kernel of the call, on the given stream */
    uint32_t num_workgroups;
    uint32_t workgroup_size;
    uint32_t lds_bytes;
    uint32_t scene_in_lds;    /* 1 when the whole traversal structure is LDS-resident */
    uint32_t trace_launches;  /* passes: launches of the path-tracing kernel (one per pass of samples per pixel) */
    float    trace_ms;        /* sum of the trace launches' hipEvent durations (the dominant kernel) */
    uint32_t guarded;         /* 1: guarded near-first walk + exact re-walk of flagged samples; 0: exact walk only */
    uint64_t flagged_samples; /* samples the guarded walk handed to the exact walk (0 when not guarded) */
    float    rework_ms;       /* sum of the exact re-walk launches' durations (0 when not guarded) */
    uint32_t guard_unproven;  /* 1: the guarded walk ran with rt_config.guard_gamma_ulps below the proven bound */
    uint32_t kernel;          /* RT_KERNEL_* actually used */
    uint32_t guard_dynamic;   /* 1: the guarded walk ran with distance-aware margins (rt_config.guard_dynamic_margins) */
    uint32_t wide_nodes;      /* 1: the guarded walk ran on 4-wide nodes */
    uint32_t sphere_only;     /* 1: the sphere-only build of the guarded kernel ran (rt_config.sphere_only_kernel) */
    uint32_t primary_visibility; /* 1: camera rays were resolved by the per-pixel candidate pass (rt_config.primary_visibility) */
    float    primary_ms;      /* … its launches' hipEvent durations (candidate lists + one pass per trace launch) */
    uint32_t trace_vgprs;     /* vector registers per lane of the trace kernel that ran, as the loaded code object reports them */
    uint32_t trace_scratch_bytes; /* … and its scratch (spill) bytes per lane */
    uint32_t abandoned_passes; /* guarded passes that gave up part-way because too many of their samples were being flagged
                                  (rt_config.guard_bail_share): the exact walk rendered those passes whole */
    uint64_t traced_samples;  /* samples the trace kernel worked on: all of them, or with the primary-visibility pass those of the pixels
                                  some leaf can be hit through (the others got the background without any per-sample work) */
    uint32_t guard_paused;    /* 1: this handle has stepped aside to the exact walk for its next frames (a frame abandoned a pass or
                                  flagged more than the bail share of its samples; rt_config.guard_keep = 1 prevents it) */
    uint32_t front_primitives; /* primitives the guarded walk tested at the start of every ray instead of keeping them in its tree
                                  (rt_config.guard_front_primitives); 0 when the exact walk ran */
} rt_timing;
/* *t = zeros with struct_bytes = sizeof(rt_timing). */
void rt_timing_init(rt_timing *t);

typedef struct rt_scene rt_scene;   /* opaque: device-resident repacked scene */

/* Library configuration.  The reference has no equivalent (its launch shape is hard-coded,
 * src/camera.cu:200-204); everything that changes how THIS library renders travels through this struct,
 * never through the environment.  rt_config_init() fills the defaults; 0 in a field marked "0 = auto"
 * means the library decides.  Results are the same bits for every setting except where a field says
 * otherwise. */
enum { RT_TRAVERSAL_AUTO = 0, RT_TRAVERSAL_EXACT = 1, RT_TRAVERSAL_GUARDED = 2 };
enum { RT_BUILD_HOST_SAH = 0, RT_BUILD_DEVICE_LBVH = 1 };
enum { RT_KERNEL_AUTO = 0, RT_KERNEL_MEGA = 1, RT_KERNEL_WAVEFRONT = 2 };
typedef struct rt_config {
    uint32_t struct_bytes;        /* sizeof(rt_config) as the caller compiled it */
    /* --- fixed at rt_scene_create_ex ------------------------------------------------------------ */
    int32_t  tree_build;          /* RT_BUILD_*: who builds the guarded walk's own tree */
    float    guard_gamma_ulps;    /* rounding budget of hit_sphere's discriminant, in units of 2^-24 |oc|^2 |d|^2, that
                                     the guarded walk's leaf margins cover.  0 = the proven bound (24).  A smaller
                                     positive value is an UNPROVEN margin: opt-in, reported in rt_timing.guard_unproven */
    int32_t  guard_exact_leaf_table; /* 1: always upload the exact leaf boxes as a table (developer) */
    /* --- may be changed between frames with rt_scene_set_config ----------------------------------- */
    int32_t  traversal;           /* RT_TRAVERSAL_*: AUTO = guarded near-first walk where the scene is eligible and
                                     has at least guard_min_primitives primitives, else the reference-order walk.  AUTO also
                                     MEASURES: a guarded frame that flagged more than 0.4 % of its samples is followed by one
                                     frame on the exact walk, and the handle keeps whichever cost less per sample (same bits
                                     either way).  GUARDED / EXACT: that walk, no measuring */
    int32_t  guard_min_primitives;/* default 64: below that the exact walk's tree is a few levels deep and the second launch the
                                     guarded walk needs costs more than it saves (random scenes of 20-60 spheres: 1.3-1.4 x slower) */
    int32_t  guard_keep;          /* 1: keep the guarded walk even after a frame that flagged more than the bail share of its samples */
    int32_t  guard_repack;        /* 1 (default): re-pack the guarded tree for a camera outside the reach it was sized for */
    int32_t  kernel;              /* RT_KERNEL_*: AUTO picks per scene */
    uint64_t workspace_bytes;     /* budget of the per-pass sample workspace the scene handle owns (12 bytes per sample of a
                                     pass, allocated on demand).  0 = default: a sixteenth of the device's memory.  A smaller
                                     budget means more, shorter passes: 4 GiB costs 1.9 % at 1920x1080x500 spp.  Footprint: rows are
                                     padded to 32 samples (128-byte lines; to 4 samples for passes shorter than 32), so a pass of S
                                     samples per pixel takes pixels x round_up(S, 32) x 12 bytes */
    int32_t  pass_spp;            /* samples per pixel per pass (0 = auto: what the workspace admits) */
    int32_t  stack_levels;        /* cap on the guarded walk's per-lane stack entries (0 = auto) */
    uint32_t flag_capacity;       /* cap on the flagged-sample list (0 = auto; overflow = "re-walk everything") */
    int32_t  scene_in_lds;        /* 1 (default): stage tables in LDS when they fit; 0: read them through L1/L2 */
    int32_t  lds_treelet;         /* 1 (default): scenes too big for LDS keep the top of their tree there */
    int32_t  workgroups_per_cu;   /* 0 = auto */
    int32_t  k_inner, k_shade;    /* wave scheduling thresholds in lanes (0 = defaults: 24 / 48; 32 / 52 for the LDS-resident guarded walk, 48 / 52 for big scenes with distance-aware margins) */
    int32_t  reserve_chunk;       /* work indices per queue reservation in units of 64 (0 = auto) */
    int32_t  reserve_taper;       /* 1 (default): reservations shrink towards the end of a pass */
    int32_t  wavefront_paths;     /* RT_KERNEL_WAVEFRONT: paths in flight per wave, >= 128 (0 = auto) */
    int32_t  wavefront_exchange;  /* … lanes that must be free before a wave exchanges results for new rays (0 = auto) */
    int32_t  wide_nodes;          /* 0 (default): scenes with distance-aware margins (guard_dynamic_margins) walk the 4-wide form of their tree —
                                     half the dependent record loads per ray, and with the growth of the boxes in parametric form fewer
                                     instructions as well (BASELINE configs[4] +1.7 %) —, every other scene child-pair nodes; -1: pair nodes
                                     always; 1: 4-wide nodes for every guarded walk — developer build only (RT_ERR_UNSUPPORTED in the
                                     shipped library): 6 % slower than the octant pair walk on S-rtiow */
    int32_t  guard_dynamic_margins; /* (fixed at create) margins of the guarded walk's small spheres: 0 = auto (distance-aware
                                     where one margin per sphere would exceed a quarter of the smallest radius), 1 = always one
                                     margin per sphere, 2 = always distance-aware */
    int32_t  sphere_only_kernel;  /* 0 (default): scenes without planes, textures and absorbing dielectrics are rendered by the sphere-only
                                     build of the guarded kernel (64 registers per lane, 8 waves per SIMD instead of 6; the same frame bit
                                     for bit) — the LDS-resident octant walk where the tables fit, the pair walk through L1 / L2 with
                                     distance-aware margins beyond; -1: always the general kernel */
    int32_t  overlap_rework;      /* 0 (default): the exact re-walk of flagged samples and the accumulation of their pixels run on a
                                     second stream of the handle beside the accumulation of all other pixels; -1: one after the other */
    int32_t  primary_visibility;  /* 0 (default): where the guarded walk's tables are LDS-resident, the first hit of every camera ray
                                     comes from a per-pixel candidate list (the leaves the pixel's cone of rays can reach, made once per
                                     frame; 68 bytes per pixel of device memory, taken on demand — without it the frame is rendered the
                                     other way) instead of a walk per sample; pixels no leaf can be hit through get the background
                                     without any per-sample work, the others are traced expensive ones first — the same frame bit for
                                     bit; -1: camera rays walk the tree */
    int32_t  guard_bail_share;    /* what the guarded walk may cost before it steps aside, as the share of flagged samples in 1/256ths
                                     (0 = default: 64, i.e. 25 % — measured break-even is 21-26 % of a frame's samples; -1: never).
                                     Inside a pass: once the flagged share of the samples handed out so far exceeds it AND 9.4 % of the
                                     whole pass is on the list AND the pass is still in its first quarter, the waves stop fetching and
                                     the exact walk renders the WHOLE pass (bounded loss: what the guarded launch had done).  Between frames: a frame that abandoned a
                                     pass, or flagged more than this share overall, makes the handle use the exact walk from then on —
                                     found at the next render call from what the previous one left in host memory, no rt_last_timing
                                     needed.  Below it RT_TRAVERSAL_AUTO decides by measurement (see traversal) */
    int32_t  guard_front_primitives; /* (fixed at create) 0 (default): up to four primitives that span the scene — a leaf box of at least half
                                     the surface of everything that is left: a ground sphere, a floor quad — are not leaves of the guarded
                                     walk's tree; every ray tests them when it is armed, all lanes of a wave together, and walks with their
                                     hit as its closest so far (the same frame bit for bit: the guarded walk's result does not depend on
                                     the order of its tests); -1: every primitive is a leaf of the tree */
    int32_t  reuse_view_lists;    /* 0 (default): the per-pixel candidate lists of the primary-visibility pass and the fetch order made from
                                     them are kept with the handle, and a call with the same camera, image, shard and tree on the same
                                     stream (the next batch of a progressive render, the next frame of a still) does not make them again
                                     (0.4 ms at 1920x1080); -1: every call makes them (bench.py: every timed frame does all of a frame's work) */
    int32_t  resume_flagged;      /* 0 (default): the exact re-walk of a sample the guarded walk flagged goes on from the ray that was flagged —
                                     the path's state at that point is left in a 17 MB table of the handle (a sample whose slot is taken, or
                                     that was flagged inside a shade step, is redone from its camera ray as before); -1: every flagged
                                     sample is redone from the camera.  The same frame bit for bit: up to the flagged ray both walks agree */
} rt_config;

/* ---- entry points -------------------------------------------------------------------------- */

/* Select the HIP device this thread's subsequent calls use (hipSetDevice). */
rt_status rt_set_device(int32_t device_ordinal);

/* Replaces create_scene's upload block (src/main.cu:429-474) and gpu_render's
 * cudaMemcpyToSymbol(d_scene_data_const) (src/camera.cu:291): validates the arrays, repacks them
 * to the device layout and uploads them once. */
rt_status rt_scene_create(const rt_scene_desc *desc, rt_scene **out_scene);

/* Defaults into *cfg.  rt_config grows with the library, like rt_timing: the macro hands over the size the CALLER was compiled
 * with and the library writes at most that many bytes (struct_bytes = that size) — a caller built against an older, shorter
 * header keeps working.  A binding that calls the exported function rt_config_init itself (ctypes, cgo: no macro) gets the
 * library's own sizeof(rt_config) and has to mirror the struct of the library it loads. */
void rt_config_init_sized(rt_config *cfg, uint32_t struct_bytes);
void rt_config_init(rt_config *cfg);
#define rt_config_init(cfg) rt_config_init_sized((cfg), (uint32_t)sizeof(rt_config))
/* Developer convenience for test harnesses and tools: overlays the RTP_* environment variables (RTP_TRAVERSAL,
 * RTP_BUILD, RTP_SLAB_GIB, RTP_PASS_SPP, …; list in INTEGRATION.md) onto *cfg.  The library itself never reads
 * the environment: a host that wants this behaviour calls it explicitly. */
void rt_config_from_env(rt_config *cfg);
/* rt_scene_create with a configuration (NULL = defaults). */
rt_status rt_scene_create_ex(const rt_scene_desc *desc, const rt_config *cfg, rt_scene **out_scene);
/* Replace the render-time fields of the scene's configuration (the create-time fields are ignored). */
rt_status rt_scene_set_config(rt_scene *scene, const rt_config *cfg);
/* The scene's configuration into *cfg: at most cfg->struct_bytes bytes (set by rt_config_init; below 8: RT_ERR_INVALID_ARG). */
rt_status rt_scene_get_config(const rt_scene *scene, rt_config *cfg);

/* Replaces destroy_scene_arrays / destroy_texture_resources (src/main.cu:235-246,322-344). */
rt_status rt_scene_destroy(rt_scene *scene);

/* Which closest-hit walk rt_render uses for this scene.  "" = the guarded near-first walk with an
 * exact re-walk of flagged samples (sphere-only scenes whose tables fit LDS); otherwise the reason
 * the scene only gets the reference-order walk (e.g. "scene has planes").  Results are the same
 * bits either way; this is a diagnostic.  The string lives as long as the scene. */
const char *rt_scene_guard_reason(const rt_scene *scene);

/* Number of rows rt_render writes for (image_height, shard). */
int32_t rt_shard_rows(int32_t image_height, const rt_shard *shard);

/* Replaces cudaMemcpyToSymbol(d_cam_data_const) + render_kernel<<<>>> + cudaDeviceSynchronize
 * (src/camera.cu:325,204-206).  d_fb_sum is DEVICE memory, rt_shard_rows()*image_width*3 floats,
 * row-major; like the reference's framebuffer it receives the SUM over samples_per_pixel of the
 * per-sample radiance, added in sample order (src/camera.cu:27-33).  hip_stream is a hipStream_t
 * (NULL = default stream).  With sync != 0 the call waits for the kernel and fills `timing`
 * (may be NULL); with sync == 0 it only enqueues (timing->kernel_ms is then read later with
 * rt_last_kernel_ms()).
 * Limits: at most 2^24 pixels per call (rt_shard_rows() x image_width; 4K = 2^23) and samples_per_pixel <= 65536 —
 * RT_ERR_UNSUPPORTED beyond.  The scene must have been created on the calling thread's current device
 * (RT_ERR_INVALID_ARG otherwise).  One handle renders one frame at a time: calls on the same handle must be
 * issued to the same stream or separated by a synchronisation.  (With rt_config.overlap_rework the handle runs part of a
 * pass on a second, non-blocking stream of its own; that stream is forked from and joined to hip_stream with events inside the
 * call, so to the caller everything still happens in hip_stream's order.) */
rt_status rt_render(rt_scene *scene, const rt_camera_data *cam, const rt_shard *shard,
                    float *d_fb_sum, void *hip_stream, int32_t sync, rt_timing *timing);

/* The same for a RECTANGLE of the image (tile_x0, tile_y0, w, h): d_fb_sum is tile_h rows of tile_w pixels, row-major;
 * pixel (x, y) of the tile is pixel (tile_x0 + x, tile_y0 + y) of the image — the same seeds, the same camera rays, the same
 * sums, so tiles of any shape assemble to the bits of the whole frame.  RT_ERR_INVALID_ARG for a tile that leaves the image. */
rt_status rt_render_tile(rt_scene *scene, const rt_camera_data *cam, int32_t tile_x0, int32_t tile_y0, int32_t tile_w, int32_t tile_h,
                         float *d_fb_sum, void *hip_stream, int32_t sync, rt_timing *timing);

/* Milliseconds of the most recent rt_render kernel of this scene (waits for it). */
rt_status rt_last_kernel_ms(rt_scene *scene, float *ms);
/* The whole rt_timing of the most recent rt_render of this scene (waits for it). */
rt_status rt_last_timing(rt_scene *scene, rt_timing *timing);

/* Convenience for hosts without their own device allocator: the whole of Camera::render up to
 * and including its cudaMemcpy D2H (src/camera.cu:198-209) into a HOST buffer. */
rt_status rt_render_to_host(rt_scene *scene, const rt_camera_data *cam, const rt_shard *shard,
                            float *h_fb_sum, rt_timing *timing);

/* Per-sample probe used by the parity tests: for n (i, j, s) triples (ijs = 3*n int32) returns the
 * radiance ray_color returns for that sample (3*n floats), the number of rays traced (n int32)
 * and the RNG state after the path (n uint32).  All pointers are HOST memory. */
rt_status rt_trace_samples(rt_scene *scene, const rt_camera_data *cam, int32_t n, const int32_t *ijs,
                           float *radiance, int32_t *rays, uint32_t *final_seed);

/* Ray-level probe used by the parity tests: hit_scene (include/scene.h:23-35) for n arbitrary rays
 * (origins/directions = 3*n floats each, HOST memory) with the interval (0.001, 1e30) ray_color
 * uses.  hit[k] = 1/0; t[k] and prim[k] (2*index + type, type 0 sphere / 1 plane) are written for
 * hits only. */
rt_status rt_closest_hits(rt_scene *scene, int32_t n, const float *origins, const float *directions,
                          int32_t *hit, float *t, int32_t *prim);

/* Device buffer management for hosts that do not link the HIP runtime themselves: replace
 * gpu_render's cudaMalloc / cudaFree of the framebuffer (src/camera.cu:295,348) and
 * Camera::render's cudaMemcpy device→host (src/camera.cu:209). */
rt_status rt_device_alloc(uint64_t bytes, void **out_device_ptr);
rt_status rt_device_free(void *device_ptr);
rt_status rt_copy_to_host(void *host_dst, const void *device_src, uint64_t bytes);

/* Device-side saver arithmetic ("next" row f1; ISaver::writeColor, src/camera.cu:138-153):
 * d_rgb8[k] = u8(256*clamp(sqrt(d_fb_sum[k] * (1/divisor)),0,0.999)).  Device pointers. */
rt_status rt_tonemap(const float *d_fb_sum, uint8_t *d_rgb8, int64_t num_floats, int32_t divisor,
                     void *hip_stream);

/* ---- multi-GPU: one frame sharded over the GPUs of one node ------------------------------------------
 * The reference has no multi-GPU path (src/camera.cu:290-349 renders on the current device).  A context owns, per
 * device, a stream, a replica of the scene, that device's rows of the frame and an RCCL communicator (single
 * process; RCCL is bound at run time and only needed for more than one device).  rt_render_sharded() renders the
 * interleaved row bands (band b → device b % N, as rt_shard) on all devices concurrently and rt_gather()s them:
 * ONE grouped ncclSend per device / ncclRecv on the root over xGMI, then the root puts the bands at their image rows.
 * The assembled frame is bit-identical to a one-device rt_render.  One host thread drives a context. */
typedef struct rt_context rt_context;

/* num_devices <= 0: every GPU of the node.  device_ordinals NULL: 0 … num_devices-1.  The first device is the root. */
rt_status rt_context_create(int32_t num_devices, const int32_t *device_ordinals, rt_context **out_ctx);
rt_status rt_context_destroy(rt_context *ctx);
int32_t rt_context_num_devices(const rt_context *ctx);
/* "rccl" (ncclSend/ncclRecv, also for a one-device context when librccl is present), "local" (one device, no RCCL) or
 * "copy" (device_ordinals lists a device more than once — a rehearsal of an N-way shard on fewer GPUs: the rows move
 * with device copies, RCCL does not admit one GPU twice). */
const char *rt_context_transport(const rt_context *ctx);
/* rt_scene_create_ex on every device of the context (replaces the context's previous scene). */
rt_status rt_context_scene_create(rt_context *ctx, const rt_scene_desc *desc, const rt_config *cfg);
/* Camera::render for the whole node: d_fb_sum_root is image_height*image_width*3 floats on the ROOT device; returns
 * when the assembled frame is there.  band_rows <= 0: 8.  timings: NULL or num_devices entries (per-device rt_timing; every entry initialised with rt_timing_init — the first entry's struct_bytes is the array stride).
 * Stream contract: the context works on non-blocking streams of its own.  rt_render_sharded and rt_gather drain the root
 * device (hipDeviceSynchronize) before they write d_fb_sum_root, so work the caller queued on that buffer earlier, on any
 * stream, is complete by then; they return after their own writes are complete.  The caller must not use the buffer from
 * another thread during the call.  With more than one device the RCCL transport has run on real hardware only as a
 * one-device self-gather so far. */
rt_status rt_render_sharded(rt_context *ctx, const rt_camera_data *cam, int32_t band_rows, float *d_fb_sum_root,
                            rt_timing *timings);
/* The collective alone: assembles the rows the devices hold from the last rt_render_sharded of this geometry. */
rt_status rt_gather(rt_context *ctx, int32_t image_width, int32_t image_height, int32_t band_rows, float *d_fb_sum_root);

/* Thread-local text of the last failing call ("" if none). */
const char *rt_get_last_error_string(void);

/* Build identification: "rtp_amd <version> gfx950 parity=<0|1>". */
const char *rt_version_string(void);

#ifdef __cplusplus
}
#endif
#endif /* RTP_AMD_H */
