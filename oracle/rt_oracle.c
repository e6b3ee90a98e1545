/*
 * rt_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see rt_oracle.h).
 *
 * Plain-C restatement of the reference's CPU render path.  Every function cites the reference
 * lines it follows.  Arithmetic notes that decide bit parity (SURVEY.md §5 "numerics"):
 *   - vec3 / float is "(1.0 / t) * v": the reciprocal is taken in double, narrowed to float,
 *     then three float multiplies (include/vec3.h:53,97);
 *   - sphere roots are (-half_b -+ sqrtf(D)) / a evaluated in double, compared/narrowed as float
 *     (include/sphere.h:35-45); plane denom/root are double (include/plane.h:58-63);
 *   - float expressions are evaluated left to right with no fused multiply-add
 *     (build with -ffp-contract=off; x86-64 SSE2 has no excess precision).
 * Must be compiled WITHOUT -ffast-math and WITHOUT -march flags that enable FMA.
 */
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float e[3]; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {{x, y, z}}; return r; }
static inline v3 from_rt(rt_vec3 a) { return V(a.e[0], a.e[1], a.e[2]); }
/* include/vec3.h:76-86,93 */
static inline v3 add(v3 a, v3 b) { return V(a.e[0] + b.e[0], a.e[1] + b.e[1], a.e[2] + b.e[2]); }
static inline v3 sub(v3 a, v3 b) { return V(a.e[0] - b.e[0], a.e[1] - b.e[1], a.e[2] - b.e[2]); }
static inline v3 mulv(v3 a, v3 b) { return V(a.e[0] * b.e[0], a.e[1] * b.e[1], a.e[2] * b.e[2]); }
static inline v3 scale(float t, v3 v) { return V(t * v.e[0], t * v.e[1], t * v.e[2]); }
static inline v3 neg(v3 a) { return V(-a.e[0], -a.e[1], -a.e[2]); }
/* include/vec3.h:97: (1.0 / t) * v — double reciprocal narrowed to the float parameter. */
static inline v3 divf(v3 v, float t) { float inv = (float)(1.0 / (double)t); return scale(inv, v); }
/* include/vec3.h:99 */
static inline float dot(v3 a, v3 b) { return a.e[0] * b.e[0] + a.e[1] * b.e[1] + a.e[2] * b.e[2]; }
/* include/vec3.h:101-103 */
static inline v3 cross(v3 a, v3 b) {
    return V(a.e[1] * b.e[2] - a.e[2] * b.e[1], a.e[2] * b.e[0] - a.e[0] * b.e[2], a.e[0] * b.e[1] - a.e[1] * b.e[0]);
}
/* include/vec3.h:55-56 */
static inline float lensq(v3 a) { return a.e[0] * a.e[0] + a.e[1] * a.e[1] + a.e[2] * a.e[2]; }
static inline float len(v3 a) { return sqrtf(lensq(a)); }
/* include/vec3.h:105 */
static inline v3 unit(v3 a) { return divf(a, len(a)); }
/* include/vec3.h:58-61 */
static inline int near_zero(v3 a) {
    const float s = 1e-8;
    return (fabsf(a.e[0]) < s) && (fabsf(a.e[1]) < s) && (fabsf(a.e[2]) < s);
}
/* include/vec3.h:63: v - 2*dot(v,n)*n */
static inline v3 reflect(v3 v, v3 n) { return sub(v, scale(2 * dot(v, n), n)); }
/* include/vec3.h:65-70 */
static inline v3 refract(v3 v, v3 n, float etai_over_etat) {
    float cos_theta = fminf(dot(neg(v), n), 1.0);
    v3 r_out_perp = scale(etai_over_etat, add(v, scale(cos_theta, n)));
    v3 r_out_parallel = scale(-sqrtf(fabsf(1.0 - lensq(r_out_perp))), n);
    return add(r_out_perp, r_out_parallel);
}

typedef struct { v3 o, d; } ray;
static inline v3 ray_at(const ray *r, float t) { return add(r->o, scale(t, r->d)); } /* include/ray.h:12 */

/* include/interval.h:16 (inclusive) */
static inline int contains(float lo, float hi, float x) { return lo <= x && x <= hi; }

/* include/hittable_object.h:8-21 */
typedef struct {
    v3 point, normal;
    float t;
    int front_face;
    int material_idx;
    float u, v;
} hitrec;

static inline void set_face_normal(hitrec *rec, const ray *r, v3 outward) {
    rec->front_face = dot(r->d, outward) < 0;
    rec->normal = rec->front_face ? outward : neg(outward);
}

/* ---- RNG: include/random_utils.h ------------------------------------------------------------ */
uint32_t orc_wang_hash(uint32_t seed) { /* :7-14 */
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return seed;
}
float orc_random_float(uint32_t *seed) { /* :16-19; can return exactly 1.0f */
    *seed = orc_wang_hash(*seed);
    return (float)(*seed) / 4294967296.0f;
}
static inline float random_range(uint32_t *seed, float lo, float hi) { /* :21-23 */
    return lo + (hi - lo) * orc_random_float(seed);
}
static v3 random_in_unit_sphere(uint32_t *seed) { /* :25-32: x, y, z drawn in that order */
    for (;;) {
        float x = random_range(seed, -1.0, 1.0);
        float y = random_range(seed, -1.0, 1.0);
        float z = random_range(seed, -1.0, 1.0);
        v3 c = V(x, y, z);
        if (lensq(c) < 1.0) return c;
    }
}
static v3 random_unit_vector(uint32_t *seed) { return unit(random_in_unit_sphere(seed)); } /* :34 */
static v3 random_in_hemisphere(v3 normal, uint32_t *seed) { /* :36-42 */
    v3 s = random_unit_vector(seed);
    if (dot(s, normal) > 0.0) return s;
    return neg(s);
}

/* ---- camera: include/camera.cuh:97-109 ------------------------------------------------------- */
static ray get_ray(const rt_camera_data *cam, int i, int j, uint32_t *seed) {
    v3 du = from_rt(cam->pixel_delta_u), dv = from_rt(cam->pixel_delta_v);
    v3 pixel_center = add(add(from_rt(cam->pixel00_loc), scale((float)i, du)), scale((float)j, dv));
    float offset_x = orc_random_float(seed) - 0.5f;
    float offset_y = orc_random_float(seed) - 0.5f;
    v3 pixel_sample = add(add(pixel_center, scale(offset_x, du)), scale(offset_y, dv));
    ray r;
    r.o = from_rt(cam->origin);
    r.d = sub(pixel_sample, r.o);
    return r;
}
void orc_get_ray(const rt_camera_data *cam, int i, int j, uint32_t *seed, float origin[3], float dir[3]) {
    ray r = get_ray(cam, i, j, seed);
    memcpy(origin, r.o.e, 12);
    memcpy(dir, r.d.e, 12);
}

/* ---- AABB::hit: include/aabb.h:42-65 ---------------------------------------------------------- */
static inline int aabb_hit(const float box[6], const ray *r, float tmin, float tmax) {
    for (int a = 0; a < 3; a++) {
        float invD = 1 / r->d.e[a];
        float orig = r->o.e[a];
        float t1 = (box[2 * a] - orig) * invD;
        float t2 = (box[2 * a + 1] - orig) * invD;
        if (invD < 0) { float tmp = t1; t1 = t2; t2 = tmp; }
        if (t1 > tmin) tmin = t1;
        if (t2 < tmax) tmax = t2;
        if (tmax <= tmin) return 0;
    }
    return 1;
}

/* ---- sphere: include/sphere.h:16-53 ----------------------------------------------------------- */
static inline void sphere_uv(v3 p, float *u, float *v) { /* :16-22 */
    float theta = acosf(p.e[1]);
    float phi = atan2f(-p.e[2], p.e[0]) + M_PI;
    *u = phi / (2 * M_PI);
    *v = theta / M_PI;
}
static int hit_sphere(const ray *r, float tmin, float tmax, hitrec *rec, const rt_sphere *s) {
    v3 center = from_rt(s->center);
    v3 oc = sub(r->o, center);
    float a = lensq(r->d);
    float half_b = dot(oc, r->d);
    float c = lensq(oc) - s->radius * s->radius;
    float D = half_b * half_b - a * c;
    if (D < 0) return 0;
    double sqrtD = sqrtf(D);
    double root = (-half_b - sqrtD) / a;
    if (!contains(tmin, tmax, (float)root)) {
        root = (-half_b + sqrtD) / a;
        if (!contains(tmin, tmax, (float)root)) return 0;
    }
    rec->t = (float)root;
    rec->point = ray_at(r, rec->t);
    v3 outward = divf(sub(rec->point, center), s->radius);
    set_face_normal(rec, r, outward);
    rec->material_idx = s->material_idx;
    sphere_uv(outward, &rec->u, &rec->v);
    return 1;
}

/* ---- plane: include/plane.h:30-96 -------------------------------------------------------------- */
static int hit_plane(const ray *r, float tmin, float tmax, hitrec *rec, const rt_plane *p) {
    v3 normal = from_rt(p->normal);
    double denom = dot(normal, r->d);
    if (fabsf((float)denom) < 1e-8) return 0;
    double root = (p->D - dot(normal, r->o)) / denom;
    if (!contains(tmin, tmax, (float)root)) return 0;

    v3 base = from_rt(p->base);
    v3 intersection = ray_at(r, (float)root);
    v3 ph = sub(intersection, base);
    v3 w = from_rt(p->w);
    float alpha = dot(w, cross(ph, from_rt(p->v)));
    float beta = dot(w, cross(from_rt(p->u), ph));
    double a = alpha, b = beta; /* is_interior_* take doubles (:30,40,48) */
    switch (p->type) {
        case RT_PLANE_QUAD:
            if (!contains(0, 1, (float)a) || !contains(0, 1, (float)b)) return 0;
            break;
        case RT_PLANE_ELLIPSE:
            if (powf(a - 0.5, 2) + powf(b - 0.5, 2) > 0.25) return 0;
            break;
        case RT_PLANE_TRIANGLE:
            if (a < 0 || b < 0 || (a + b) > 1) return 0;
            break;
        default:
            break; /* unknown type: u,v left unset by the reference too */
    }
    if (p->type == RT_PLANE_QUAD || p->type == RT_PLANE_ELLIPSE || p->type == RT_PLANE_TRIANGLE) {
        rec->u = (float)a;
        rec->v = (float)b;
    }
    rec->t = (float)root;
    rec->point = ray_at(r, (float)root);
    set_face_normal(rec, r, normal);
    rec->material_idx = p->material_idx;
    return 1;
}

/* ---- BVH traversal: include/bvh.h:19-65, include/scene.h:23-35 ------------------------------- */
/* The reference picks the child visit order from r.direction()[node.type] with node.type == -1
 * for internal nodes (include/bvh.h:52-53) — an out-of-bounds read.  Visit order only matters for
 * exact-t ties, so this restatement always visits the left child first. */
static int hit_bvh(const rt_scene_desc *sc, const ray *r, float tmin, float tmax, hitrec *rec,
                   int *prim_type, int *prim_index, orc_stats *st) {
#ifdef ORC_STUDY_HOOK      /* developer studies (tools/) see every ray the path loop casts */
    ORC_STUDY_HOOK(sc, r);
#endif
    int stack[32];
    int sp = 0;
    stack[sp++] = 0;
    int hit_anything = 0;
    float closest = tmax;
    const rt_bvh_node *nodes = sc->nodes;
    const int num_nodes = sc->num_nodes;
    while (sp > 0) {
        int node_idx = stack[--sp];
        if (node_idx >= num_nodes || node_idx < 0) continue;
        const rt_bvh_node *node = &nodes[node_idx];
        if (st) st->node_visits++;
        if (aabb_hit(node->box, r, tmin, closest)) {
            if (st) st->box_hits++;
            if (node->left < 0) {
                hitrec tmp;
                int hit = 0;
                if (node->type == 0) {
                    if (st) st->sphere_tests++;
                    hit = hit_sphere(r, tmin, closest, &tmp, &sc->spheres[node->right]);
                } else if (node->type == 1) {
                    if (st) st->plane_tests++;
                    hit = hit_plane(r, tmin, closest, &tmp, &sc->planes[node->right]);
                }
                if (hit) {
                    hit_anything = 1;
                    closest = tmp.t;
                    *rec = tmp;
                    *prim_type = node->type;
                    *prim_index = node->right;
                }
            } else if (sp + 2 <= 32) {
                stack[sp++] = node->right;
                stack[sp++] = node->left;
                if (st && (uint32_t)sp > st->max_stack) st->max_stack = (uint32_t)sp;
            }
        }
    }
    return hit_anything;
}

int orc_closest_hit(const rt_scene_desc *scene, const float origin[3], const float dir[3],
                    float *t, int *prim_type, int *prim_index) {
    ray r;
    memcpy(r.o.e, origin, 12);
    memcpy(r.d.e, dir, 12);
    hitrec rec;
    int pt = -1, pi = -1;
    if (scene->num_nodes <= 0) return 0;
    int h = hit_bvh(scene, &r, 0.001f, 1e30f, &rec, &pt, &pi, NULL);
    if (h) { *t = rec.t; *prim_type = pt; *prim_index = pi; }
    return h;
}

int orc_closest_hit_bruteforce(const rt_scene_desc *sc, const float origin[3], const float dir[3],
                               float *t, int *prim_type, int *prim_index) {
    ray r;
    memcpy(r.o.e, origin, 12);
    memcpy(r.d.e, dir, 12);
    float closest = 1e30f;
    int found = 0;
    for (int n = 0; n < sc->num_nodes; n++) {
        const rt_bvh_node *node = &sc->nodes[n];
        if (node->left >= 0) continue;
        if (!aabb_hit(node->box, &r, 0.001f, closest)) continue;
        hitrec tmp;
        int hit = 0;
        if (node->type == 0) hit = hit_sphere(&r, 0.001f, closest, &tmp, &sc->spheres[node->right]);
        else if (node->type == 1) hit = hit_plane(&r, 0.001f, closest, &tmp, &sc->planes[node->right]);
        if (hit) { found = 1; closest = tmp.t; *t = tmp.t; *prim_type = node->type; *prim_index = node->right; }
    }
    return found;
}

/* ---- texture: include/materials.h:20-51 ------------------------------------------------------ */
void orc_tex2d(const rt_texture *tex, float u, float v, float rgb[3]) {
    if (!tex || !tex->rgba) { rgb[0] = rgb[1] = rgb[2] = 1; return; }
    u = u - floorf(u);
    v = v - floorf(v);
    float px = u * tex->width;
    float py = (1.0f - v) * tex->height;
    int x0 = (int)px;
    int y0 = (int)py;
    int x1 = (x0 + 1) % tex->width;
    int y1 = (y0 + 1) % tex->height;
    float dx = px - x0;
    float dy = py - y0;
    /* deviation: the reference indexes x0,y0 unwrapped and reads past the row/image when
     * u or v lands exactly on 1.0; wrap them instead of reading out of bounds. */
    int x0w = x0 % tex->width, y0w = y0 % tex->height;
    const float *d = tex->rgba;
    int W = tex->width;
#define PX(x, y) V(d[((y) * W + (x)) * 4], d[((y) * W + (x)) * 4 + 1], d[((y) * W + (x)) * 4 + 2])
    v3 c00 = PX(x0w, y0w), c10 = PX(x1, y0w), c01 = PX(x0w, y1), c11 = PX(x1, y1);
#undef PX
    v3 top = add(scale(1.0f - dx, c00), scale(dx, c10));
    v3 bot = add(scale(1.0f - dx, c01), scale(dx, c11));
    v3 res = add(scale(1.0f - dy, top), scale(dy, bot));
    memcpy(rgb, res.e, 12);
}

/* ---- materials: include/materials.h:64-142 ---------------------------------------------------- */
static inline float reflectance(float cosine, float ref_idx) { /* :64-68 */
    float r0 = (1 - ref_idx) / (1 + ref_idx);
    r0 = r0 * r0;
    return r0 + (1 - r0) * powf((1 - cosine), 5);
}

/* The uniform-hemisphere scatter.  include/materials.h writes these five lines twice, token for token: as the whole
 * LAMBERTIAN case (:74-78) and as METAL's 20 % branch (:91-95).  One routine here, so that the reference images
 * that pin METAL (tests/test_oracle_pins.py) exercise the very code LAMBERTIAN runs — the reference's own scenes
 * never instantiate LAMBERTIAN. */
static int scatter_diffuse(const hitrec *rec, v3 *attenuation, ray *scattered, uint32_t *seed, v3 albedo) {
    v3 dir = random_in_hemisphere(rec->normal, seed);
    if (near_zero(dir)) dir = rec->normal;
    scattered->o = rec->point;
    scattered->d = dir;
    *attenuation = albedo;
    return 1;
}

/* mat->albedo already texture-modulated by the caller (src/camera.cu:268-270). */
static int material_scatter(const ray *r_in, const hitrec *rec, v3 *attenuation, ray *scattered,
                            uint32_t *seed, const rt_material *mat, v3 albedo) {
    switch (mat->type) {
        case RT_MAT_LAMBERTIAN: /* :73-79 */
            return scatter_diffuse(rec, attenuation, scattered, seed, albedo);
        case RT_MAT_METAL: { /* :81-96 */
            float p_metal = 0.8f;
            if (orc_random_float(seed) < p_metal) {
                v3 reflected = reflect(unit(r_in->d), rec->normal);
                scattered->o = rec->point;
                scattered->d = add(reflected, scale(mat->fuzz, random_in_unit_sphere(seed)));
                *attenuation = albedo;
                return dot(scattered->d, rec->normal) > 0;
            }
            return scatter_diffuse(rec, attenuation, scattered, seed, albedo);      /* :90-95 */
        }
        case RT_MAT_DIELECTRIC: { /* :98-133 */
            v3 att = V(1.0, 1.0, 1.0);
            float refraction_ratio = rec->front_face ? (float)(1.0 / mat->ir) : mat->ir;
            v3 unit_direction = unit(r_in->d);
            float cos_theta = fminf(dot(neg(unit_direction), rec->normal), 1.0);
            float sin_theta = sqrtf(1.0 - cos_theta * cos_theta);
            int cannot_refract = refraction_ratio * sin_theta > 1.0;
            v3 direction;
            /* short-circuit ||: the Schlick draw happens only when refraction is possible (:108) */
            if (cannot_refract || reflectance(cos_theta, refraction_ratio) > orc_random_float(seed)) {
                direction = reflect(unit_direction, rec->normal);
            } else {
                direction = refract(unit_direction, rec->normal, refraction_ratio);
            }
            float distance = len(sub(rec->point, r_in->o));
            v3 ab = from_rt(mat->absorption);
            /* exp(float) resolves to the float overload in the reference's translation unit */
            v3 transmission = V(expf(-ab.e[0] * distance), expf(-ab.e[1] * distance), expf(-ab.e[2] * distance));
            if (!rec->front_face) att = mulv(att, transmission);
            float p = fmaxf(att.e[0], fmaxf(att.e[1], att.e[2]));
            if (orc_random_float(seed) > p) return 0;
            att = scale((float)(1.0 / (double)p), att); /* attenuation /= p (include/vec3.h:53) */
            float offset = 1e-4f;
            v3 origin = add(rec->point, scale((dot(direction, rec->normal) > 0 ? 1.0f : -1.0f), scale(offset, rec->normal)));
            scattered->o = origin;
            scattered->d = direction;
            *attenuation = att;
            return 1;
        }
        case RT_MAT_DIFFUSE_LIGHT:
        default:
            return 0;
    }
}

/* ---- ray_color_host: src/camera.cu:254-288 --------------------------------------------------- */
static v3 ray_color(ray r, uint32_t *seed, const rt_scene_desc *sc, const rt_camera_data *cam,
                    int32_t *rays_out, orc_stats *st) {
    v3 final_color = V(0.0f, 0.0f, 0.0f);
    v3 beta = V(1.0f, 1.0f, 1.0f);
    ray cur = r;
    int32_t nrays = 0;
    for (int depth = 0; depth < cam->max_depth; depth++) {
        hitrec rec;
        int pt = -1, pi = -1;
        nrays++;
        if (st) st->rays++;
        int h = sc->num_nodes > 0 ? hit_bvh(sc, &cur, 0.001f, 1e30f, &rec, &pt, &pi, st) : 0;
        if (!h) {
            final_color = add(final_color, mulv(beta, from_rt(cam->background)));
            break;
        }
        const rt_material *mat = &sc->materials[rec.material_idx];
        if (st) st->material_fetches++;
        v3 albedo = from_rt(mat->albedo);
        if (mat->texture_id != 0) {
            float tc[3];
            orc_tex2d(&sc->textures[mat->texture_id - 1], rec.u, rec.v, tc);
            albedo = mulv(albedo, V(tc[0], tc[1], tc[2])); /* :269 albedo * tex */
            if (st) st->texture_fetches++;
        }
        final_color = add(final_color, mulv(beta, from_rt(mat->emit)));
        ray scattered;
        v3 attenuation;
        if (!material_scatter(&cur, &rec, &attenuation, &scattered, seed, mat, albedo)) break;
        beta = mulv(beta, attenuation);
        cur = scattered;
    }
    if (rays_out) *rays_out = nrays;
    return final_color;
}

void orc_trace_sample(const rt_scene_desc *scene, const rt_camera_data *cam, int i, int j, int s,
                      float radiance[3], int32_t *rays, uint32_t *final_seed) {
    /* src/camera.cu:41-45: base = wang_hash(i*W + j) with i = column */
    uint32_t base = orc_wang_hash((uint32_t)i * (uint32_t)cam->image_width + (uint32_t)j);
    uint32_t seed = orc_wang_hash(base + (uint32_t)s);
    ray r = get_ray(cam, i, j, &seed);
    v3 c = ray_color(r, &seed, scene, cam, rays, NULL);
    memcpy(radiance, c.e, 12);
    if (final_seed) *final_seed = seed;
}

/* ---- Camera::render_cpu: src/camera.cu:36-50 -------------------------------------------------- */
typedef struct {
    const rt_scene_desc *scene;
    const rt_camera_data *cam;
    int row0, row1, tid, nthreads;
    float *fb;
    orc_stats stats;
    int want_stats;
} job;

static void *render_rows(void *arg) {
    job *jb = (job *)arg;
    const rt_camera_data *cam = jb->cam;
    const int W = cam->image_width;
    orc_stats *st = jb->want_stats ? &jb->stats : NULL;
    for (int j = jb->row0 + jb->tid; j < jb->row1; j += jb->nthreads) {
        for (int i = 0; i < W; i++) {
            v3 pixel = V(0, 0, 0);
            uint32_t base = orc_wang_hash((uint32_t)i * (uint32_t)W + (uint32_t)j);
            for (int s = 0; s < cam->samples_per_pixel; s++) {
                uint32_t seed = orc_wang_hash(base + (uint32_t)s);
                ray r = get_ray(cam, i, j, &seed);
                v3 c = ray_color(r, &seed, jb->scene, cam, NULL, st);
                pixel = add(pixel, c);
                if (st) st->samples++;
            }
            memcpy(jb->fb + ((size_t)(j - jb->row0) * W + i) * 3, pixel.e, 12);
        }
    }
    return NULL;
}

void orc_render(const rt_scene_desc *scene, const rt_camera_data *cam, int row0, int row1,
                float *fb_sum, int num_threads, orc_stats *stats) {
    if (num_threads < 1) num_threads = 1;
    if (num_threads > 256) num_threads = 256;
    job *jobs = (job *)calloc((size_t)num_threads, sizeof(job));
    pthread_t *th = (pthread_t *)calloc((size_t)num_threads, sizeof(pthread_t));
    for (int t = 0; t < num_threads; t++) {
        jobs[t].scene = scene; jobs[t].cam = cam; jobs[t].row0 = row0; jobs[t].row1 = row1;
        jobs[t].tid = t; jobs[t].nthreads = num_threads; jobs[t].fb = fb_sum; jobs[t].want_stats = stats != NULL;
    }
    if (num_threads == 1) {
        render_rows(&jobs[0]);
    } else {
        for (int t = 0; t < num_threads; t++) pthread_create(&th[t], NULL, render_rows, &jobs[t]);
        for (int t = 0; t < num_threads; t++) pthread_join(th[t], NULL);
    }
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        for (int t = 0; t < num_threads; t++) {
            const orc_stats *s = &jobs[t].stats;
            stats->samples += s->samples; stats->rays += s->rays; stats->node_visits += s->node_visits;
            stats->box_hits += s->box_hits; stats->sphere_tests += s->sphere_tests; stats->plane_tests += s->plane_tests;
            stats->material_fetches += s->material_fetches; stats->texture_fetches += s->texture_fetches;
            if (s->max_stack > stats->max_stack) stats->max_stack = s->max_stack;
        }
    }
    free(jobs);
    free(th);
}

/* ---- ISaver::writeColor: src/camera.cu:138-153 ------------------------------------------------ */
void orc_write_color(const float rgb_sum[3], int divisor, uint8_t out[3]) {
    /* pixel_color / samplesPerPixel → (1.0 / float(n)) narrowed to float, times each channel */
    float inv = (float)(1.0 / (double)(float)divisor);
    for (int k = 0; k < 3; k++) {
        float lin = inv * rgb_sum[k];
        float g = sqrtf(lin); /* linearToGamma, src/camera.cu:54 */
        float c = g;
        if (g < 0.0f) c = 0.0f;          /* Interval(0.0, 0.999).clamp, include/interval.h:18-22 */
        if (g > 0.999f) c = 0.999f;
        out[k] = (uint8_t)(256 * c);
    }
}

/* ---- batched views of the primitives above, for tests/test_ref_geom.py --------------------------------------------------------
 * The same calls, on the same arrays, as oracle/ref_geom.cpp makes on the reference's own headers (compiled from where they lie):
 * the test compares the two bit for bit.  Nothing here computes anything new. */
static ray ray_of(const float *o, const float *d) { ray r; memcpy(r.o.e, o, 12); memcpy(r.d.e, d, 12); return r; }
static void put_rec(float *f, int32_t *code, const hitrec *rec, int id) {
    f[0] = rec->t; memcpy(f + 1, rec->point.e, 12); memcpy(f + 4, rec->normal.e, 12); f[7] = rec->u; f[8] = rec->v;
    *code = (rec->front_face ? 1 : 0) | (id << 1);
}
void orc_geom_aabb_hit(int64_t n, const float *boxes, const float *origins, const float *dirs, const float *tmin, const float *tmax, int32_t *out) {
    for (int64_t k = 0; k < n; ++k) { ray r = ray_of(origins + 3 * k, dirs + 3 * k); out[k] = aabb_hit(boxes + 6 * k, &r, tmin[k], tmax[k]); }
}
void orc_geom_vec3_div(int64_t n, const float *a, const float *t, float *out) {
    for (int64_t k = 0; k < n; ++k) { v3 r = divf(V(a[3 * k], a[3 * k + 1], a[3 * k + 2]), t[k]); memcpy(out + 3 * k, r.e, 12); }
}
void orc_geom_unit_vector(int64_t n, const float *a, float *out) {
    for (int64_t k = 0; k < n; ++k) { v3 r = unit(V(a[3 * k], a[3 * k + 1], a[3 * k + 2])); memcpy(out + 3 * k, r.e, 12); }
}
void orc_geom_reflect(int64_t n, const float *a, const float *nrm, float *out) {
    for (int64_t k = 0; k < n; ++k) { v3 r = reflect(V(a[3 * k], a[3 * k + 1], a[3 * k + 2]), V(nrm[3 * k], nrm[3 * k + 1], nrm[3 * k + 2])); memcpy(out + 3 * k, r.e, 12); }
}
void orc_geom_refract(int64_t n, const float *a, const float *nrm, const float *eta, float *out) {
    for (int64_t k = 0; k < n; ++k) {
        v3 r = refract(V(a[3 * k], a[3 * k + 1], a[3 * k + 2]), V(nrm[3 * k], nrm[3 * k + 1], nrm[3 * k + 2]), eta[k]);
        memcpy(out + 3 * k, r.e, 12);
    }
}
void orc_geom_near_zero(int64_t n, const float *a, int32_t *out) {
    for (int64_t k = 0; k < n; ++k) out[k] = near_zero(V(a[3 * k], a[3 * k + 1], a[3 * k + 2]));
}
void orc_geom_dot_cross_len(int64_t n, const float *a, const float *b, float *out_dot, float *out_cross, float *out_len) {
    for (int64_t k = 0; k < n; ++k) {
        v3 x = V(a[3 * k], a[3 * k + 1], a[3 * k + 2]), y = V(b[3 * k], b[3 * k + 1], b[3 * k + 2]), c = cross(x, y);
        out_dot[k] = dot(x, y); memcpy(out_cross + 3 * k, c.e, 12); out_len[k] = len(x);
    }
}
void orc_geom_contains(int64_t n, const float *lo, const float *hi, const float *x, int32_t *out) {
    for (int64_t k = 0; k < n; ++k) out[k] = contains(lo[k], hi[k], x[k]);
}
void orc_geom_ray_at(int64_t n, const float *origins, const float *dirs, const float *t, float *out) {
    for (int64_t k = 0; k < n; ++k) { ray r = ray_of(origins + 3 * k, dirs + 3 * k); v3 p = ray_at(&r, t[k]); memcpy(out + 3 * k, p.e, 12); }
}
void orc_geom_set_face_normal(int64_t n, const float *dirs, const float *outward, float *out_normal, int32_t *out_front) {
    const float zero[3] = {0, 0, 0};
    for (int64_t k = 0; k < n; ++k) {
        ray r = ray_of(zero, dirs + 3 * k);
        hitrec rec;
        set_face_normal(&rec, &r, V(outward[3 * k], outward[3 * k + 1], outward[3 * k + 2]));
        memcpy(out_normal + 3 * k, rec.normal.e, 12);
        out_front[k] = rec.front_face;
    }
}
/* item k: ray k against sphere k / plane k (records in the ABI's layouts); code = front_face | (k & 0xffff) << 1 */
void orc_geom_hit_sphere(int64_t n, const float *origins, const float *dirs, const float *tmin, const float *tmax, const rt_sphere *spheres,
                         int32_t *out_hit, float *out_rec9, int32_t *out_code) {
    for (int64_t k = 0; k < n; ++k) {
        ray r = ray_of(origins + 3 * k, dirs + 3 * k);
        hitrec rec;
        memset(&rec, 0, sizeof(rec));
        out_hit[k] = hit_sphere(&r, tmin[k], tmax[k], &rec, &spheres[k]);
        if (out_hit[k]) put_rec(out_rec9 + 9 * k, out_code + k, &rec, (int)(k & 0xffff));
    }
}
void orc_geom_hit_plane(int64_t n, const float *origins, const float *dirs, const float *tmin, const float *tmax, const rt_plane *planes,
                        int32_t *out_hit, float *out_rec9, int32_t *out_code) {
    for (int64_t k = 0; k < n; ++k) {
        ray r = ray_of(origins + 3 * k, dirs + 3 * k);
        hitrec rec;
        memset(&rec, 0, sizeof(rec));
        out_hit[k] = hit_plane(&r, tmin[k], tmax[k], &rec, &planes[k]);
        if (out_hit[k]) put_rec(out_rec9 + 9 * k, out_code + k, &rec, (int)(k & 0xffff));
    }
}
/* hit_bvh for n rays; code = front_face | (2 * primitive index + type) << 1 */
void orc_geom_hit_bvh(const rt_scene_desc *scene, int64_t n, const float *origins, const float *dirs, float tmin, float tmax,
                      int32_t *out_hit, float *out_rec9, int32_t *out_code) {
    for (int64_t k = 0; k < n; ++k) {
        ray r = ray_of(origins + 3 * k, dirs + 3 * k);
        hitrec rec;
        memset(&rec, 0, sizeof(rec));
        int pt = -1, pi = -1;
        out_hit[k] = scene->num_nodes > 0 ? hit_bvh(scene, &r, tmin, tmax, &rec, &pt, &pi, NULL) : 0;
        if (out_hit[k]) put_rec(out_rec9 + 9 * k, out_code + k, &rec, 2 * pi + pt);
    }
}
