/* Developer study (CPU): how many node visits would a per-octant near-first order of the CALLER'S
 * tree save over the reference's left-first order, how often would the conservative pruning band
 * flag a ray for an exact re-walk, and is the result identical for the unflagged rays?
 * Build: gcc -O2 -ffp-contract=off -std=gnu11 -Iinclude tools/nearfirst_study.c \
 *            -Lray-tracing-practice_amd -lrtp_host -Wl,-rpath,$PWD/ray-tracing-practice_amd -lm -lpthread -o /tmp/nf_study
 */
#include <stdio.h>
#include <stdlib.h>
struct rt_scene_desc;
struct ray_s;
static void study_ray(const void *sc, const void *r);
#define ORC_STUDY_HOOK(sc, r) study_ray(sc, r)
#include "../oracle/rt_oracle.c"
#include "../ray-tracing-practice_amd/host/rtp_host.h"

static unsigned long long n_rays, v_ref, v_oct, v_oct_band, n_flag, n_flag_prune, n_flag_incons, n_mismatch, n_mismatch_unflagged, lt_ref, lt_oct;
static float g_delta = 1.0f / 1024;
static int *g_axis;      /* split axis guessed per inner node */

/* box test returning the entry distance too */
static inline int aabb_hit_e(const float box[6], const ray *r, float tmin, float tmax, float *enter) {
    for (int a = 0; a < 3; a++) {
        float invD = 1 / r->d.e[a];
        float orig = r->o.e[a];
        float t1 = (box[2 * a] - orig) * invD;
        float t2 = (box[2 * a + 1] - orig) * invD;
        if (invD < 0) { float tmp = t1; t1 = t2; t2 = tmp; }
        if (t1 > tmin) tmin = t1;
        if (t2 < tmax) tmax = t2;
        if (tmax <= tmin) { *enter = tmin; return 0; }
    }
    *enter = tmin;
    return 1;
}

struct snode_s; static void *g_s_any; static void sah_ray(const ray *r, float c_ref, int prim_ref);
static void *g_w_any; static void wide_ray(const ray *r, float c_ref, int prim_ref);
static int g_defer_on; static void defer_ray(const ray *r, float c_ref, int prim_ref);
static int g_leafk_on; static void leafk_ray(const ray *r, float c_ref, int prim_ref);
static int g_model_on; static void model_ray(const ray *r, float c_ref, int prim_ref);
static int g_fused = 0;
static inline int aabb_hit_fused(const float box[6], const ray *r, float tmin0, float tmax0, float *enter) {
    float nr[3], fr[3];
    for (int a = 0; a < 3; a++) {
        float inv = 1 / r->d.e[a];
        inv = fminf(fmaxf(inv, -1e18f), 1e18f);
        const float noi = -(r->o.e[a] * inv);
        const float ta = fmaf(box[2 * a], inv, noi), tb = fmaf(box[2 * a + 1], inv, noi);
        nr[a] = fminf(ta, tb); fr[a] = fmaxf(ta, tb);
    }
    const float tmin = fmaxf(fmaxf(nr[0], nr[1]), fmaxf(nr[2], tmin0));
    const float tmax = fminf(fminf(fr[0], fr[1]), fminf(fr[2], tmax0));
    *enter = tmin;
    return tmax > tmin;
}
static void study_ray(const void *scv, const void *rv) {
    const rt_scene_desc *sc = (const rt_scene_desc *)scv;
    const ray *r = (const ray *)rv;
    const rt_bvh_node *nodes = sc->nodes;
    n_rays++;
    /* reference walk */
    float c_ref = 1e30f; int p_ref = -1;
    {
        int stack[64], sp = 0; stack[sp++] = 0;
        while (sp > 0) {
            const rt_bvh_node *n = &nodes[stack[--sp]];
            v_ref++;
            if (!aabb_hit(n->box, r, 0.001f, c_ref)) continue;
            if (n->left < 0) {
                hitrec tmp; lt_ref++;
                if (hit_sphere(r, 0.001f, c_ref, &tmp, &sc->spheres[n->right])) { c_ref = tmp.t; p_ref = (int)(n - nodes); }
            } else { stack[sp++] = n->right; stack[sp++] = n->left; }
        }
    }
    /* octant order walk with pruning band */
    float c = 1e30f; int p = -1; int flagged_prune = 0, flagged_incons = 0;
    {
        int stack[64], sp = 0; stack[sp++] = 0;
        while (sp > 0) {
            const int idx = stack[--sp];
            const rt_bvh_node *n = &nodes[idx];
            v_oct++;
            float enter;
            /* exact test with the running closest decides the walk */
            const int hit = aabb_hit_e(n->box, r, 0.001f, c, &enter);
            if (!hit) {
                /* would the box have been entered with a slightly larger closest?  then a primitive whose computed
                 * root lies in front of its own box could hide below it */
                if (c < 1e30f) {
                    float e2;
                    if (aabb_hit_e(n->box, r, 0.001f, c * (1.0f + g_delta), &e2)) flagged_prune = 1;
                }
                continue;
            }
            if (n->left < 0) {
                hitrec tmp; lt_oct++;
                /* contains(): root <= closest is accepted; on an exact tie the reference keeps the LAST visited = larger index */
                if (hit_sphere(r, 0.001f, c, &tmp, &sc->spheres[n->right])) {
                    if (tmp.t < c || idx > p) { if (tmp.t <= enter) flagged_incons = 1; c = tmp.t; p = idx; }
                }
            } else {
                const int a = g_axis[idx];
                const int left_first = r->d.e[a] >= 0;
                if (left_first) { stack[sp++] = n->right; stack[sp++] = n->left; }
                else { stack[sp++] = n->left; stack[sp++] = n->right; }
            }
        }
    }
    if (g_s_any) sah_ray(r, c_ref, p_ref >= 0 ? nodes[p_ref].right : -1);
    if (g_w_any) wide_ray(r, c_ref, p_ref >= 0 ? nodes[p_ref].right : -1);
    if (g_defer_on) defer_ray(r, c_ref, p_ref >= 0 ? nodes[p_ref].right : -1);
    if (g_leafk_on) leafk_ray(r, c_ref, p_ref >= 0 ? nodes[p_ref].right : -1);
    if (g_model_on) model_ray(r, c_ref, p_ref >= 0 ? nodes[p_ref].right : -1);
    const int flagged = flagged_prune | flagged_incons;
    n_flag += flagged; n_flag_prune += flagged_prune; n_flag_incons += flagged_incons;
    const int mismatch = (p != p_ref) || (p >= 0 && c != c_ref);
    n_mismatch += mismatch;
    if (mismatch && !flagged) n_mismatch_unflagged++;
}

/* ---- own SAH tree over the same leaf boxes, walked near-first by entry distance ---- */
typedef struct { float box[6]; int left, right, prim; } snode;
static snode *g_s; static int g_sn;
static const rt_scene_desc *g_scn;
static float *g_pbox;  /* per sphere leaf box from the caller's tree, inflated by eps_q */
static float *g_rbox;  /* the caller's exact leaf boxes */
static float g_sc[3], g_d0 = 64.0f, g_gamma = 8 * 5.96e-8f;
static int g_cmp_axis;
static int cmp_prim(const void *a, const void *b) {
    const float *A = g_pbox + 6 * *(const int *)a, *B = g_pbox + 6 * *(const int *)b;
    const float ca = A[2 * g_cmp_axis] + A[2 * g_cmp_axis + 1], cb = B[2 * g_cmp_axis] + B[2 * g_cmp_axis + 1];
    return ca < cb ? -1 : ca > cb;
}
static float area(const float *b) { const float x = b[1] - b[0], y = b[3] - b[2], z = b[5] - b[4]; return x * y + y * z + z * x; }
static void grow(float *b, const float *q) { for (int a = 0; a < 3; a++) { if (q[2*a] < b[2*a]) b[2*a] = q[2*a]; if (q[2*a+1] > b[2*a+1]) b[2*a+1] = q[2*a+1]; } }
static int sah_build(int *ids, int n) {
    const int me = g_sn++;
    float bb[6] = {1e30f, -1e30f, 1e30f, -1e30f, 1e30f, -1e30f};
    for (int i = 0; i < n; i++) grow(bb, g_pbox + 6 * ids[i]);
    memcpy(g_s[me].box, bb, sizeof bb);
    if (n == 1) { g_s[me].left = -1; g_s[me].prim = ids[0]; return me; }
    float best = 1e30f; int best_axis = 0, best_k = n / 2;
    int *tmp = malloc(n * sizeof(int)); float *ra = malloc(n * sizeof(float));
    for (int a = 0; a < 3; a++) {
        memcpy(tmp, ids, n * sizeof(int)); g_cmp_axis = a; qsort(tmp, n, sizeof(int), cmp_prim);
        float rb[6] = {1e30f, -1e30f, 1e30f, -1e30f, 1e30f, -1e30f};
        for (int i = n - 1; i > 0; i--) { grow(rb, g_pbox + 6 * tmp[i]); ra[i] = area(rb); }
        float lb[6] = {1e30f, -1e30f, 1e30f, -1e30f, 1e30f, -1e30f};
        for (int k = 1; k < n; k++) {
            grow(lb, g_pbox + 6 * tmp[k - 1]);
            const float cost = area(lb) * k + ra[k] * (n - k);
            if (cost < best) { best = cost; best_axis = a; best_k = k; }
        }
    }
    g_cmp_axis = best_axis; qsort(ids, n, sizeof(int), cmp_prim);
    free(tmp); free(ra);
    const int l = sah_build(ids, best_k), r = sah_build(ids + best_k, n - best_k);
    g_s[me].left = l; g_s[me].right = r;
    return me;
}
static unsigned long long v_sah, lt_sah, n_sah_flag, n_sah_mismatch, n_sah_mismatch_unflagged, v_sah_pairs, n_tie, n_incons_final, n_overflow;
static unsigned long long depth_hist[40];
static double g_max_ratio;       /* max observed (enter' - t) / bound over accepted hits */
static float g_beta = 16 * 5.96e-8f;
static unsigned long long n_far, n_far_flag;
static float g_bs[6], g_rs, g_fark;
static int g_levels = 64;
static float *g_rmax;            /* per SAH node: largest radius below */
/* DYN=1: distance-aware margins (docs/LOG.md §3b "dynamic margins"): the small spheres' leaf boxes carry only the
 * rounding floor, and every box test of the walk grows the box by g_dynk * (distance from the ray origin to the
 * box's farthest corner)^2 — an upper bound of gamma |o - c_q|^2 / (2 r_q) for every small sphere q below. */
static int g_dyn = 0;
static float g_dynk = 0;
static inline int box_test(const float box[6], const ray *r, float tmin, float tmax, float *enter) {
    if (!g_dyn) return g_fused ? aabb_hit_fused(box, r, tmin, tmax, enter) : aabb_hit_e(box, r, tmin, tmax, enter);
    float d2 = 0;
    for (int a = 0; a < 3; a++) { const float m = fmaxf(fabsf(r->o.e[a] - box[2 * a]), fabsf(box[2 * a + 1] - r->o.e[a])); d2 = fmaf(m, m, d2); }
    const float E = g_dynk * d2;
    float g[6];
    for (int a = 0; a < 3; a++) { g[2 * a] = box[2 * a] - E; g[2 * a + 1] = box[2 * a + 1] + E; }
    return aabb_hit_fused(g, r, tmin, tmax, enter);
}
static void sah_ray(const ray *r, float c_ref, int prim_ref) {
    float c = 1e30f; int p = -1; int flag = 0; float p_enter = 0;
    struct { int idx; float enter; } stack[64]; int sp = 0, maxsp = 0;
    const float inv_len = 1.0f / sqrtf(lensq(r->d));
    const float bd = g_beta * inv_len;
    if (!g_dyn) {   /* far-origin test, as guard_origin() in the kernel */
        float dd = 0; for (int a = 0; a < 3; a++) dd += (r->o.e[a] - g_sc[a]) * (r->o.e[a] - g_sc[a]);
        if (dd > g_d0 * g_d0) {
            n_far++;
            const float reach = sqrtf(dd) + g_rs, grow = g_fark * reach * reach;
            float gb[6], e;
            for (int a = 0; a < 3; a++) { gb[2 * a] = g_bs[2 * a] - grow; gb[2 * a + 1] = g_bs[2 * a + 1] + grow; }
            if (aabb_hit_e(gb, r, 0.001f, 1e30f, &e) || !(grow < 1e30f)) { flag = 1; n_far_flag++; }
        }
    }
    int cur = 0; float cur_enter = 0.001f; int have = 1;
    v_sah++;
    { float e0; if (!aabb_hit_e(g_s[0].box, r, 0.001f, 1e30f, &e0)) have = 0; cur_enter = e0; }
    while (have) {
        const snode *n = &g_s[cur];
        if (n->left < 0) {
            hitrec tmp; lt_sah++; const float c_before = c;
            if (hit_sphere(r, 0.001f, c, &tmp, &g_scn->spheres[n->prim])) {
                if (tmp.t == c && p >= 0) { flag = 1; n_tie++; }
                c = tmp.t; p = n->prim; p_enter = cur_enter;
                double dep = 0; { const float *rb = g_rbox + 6 * n->prim; for (int a = 0; a < 3; a++) { const double x = (double)r->o.e[a] + (double)tmp.t * r->d.e[a]; if (rb[2*a] - x > dep) dep = rb[2*a] - x; if (x - rb[2*a+1] > dep) dep = x - rb[2*a+1]; } }
                const double bound = (double)(g_rbox[6 * n->prim] - g_pbox[6 * n->prim]);
                if (dep > 0 && !flag) { const double ratio = dep / bound; if (ratio > g_max_ratio) { g_max_ratio = ratio; fprintf(stderr, "dep %.3g eps %.3g t %.5g r %.3g\n", dep, bound, tmp.t, g_scn->spheres[n->prim].radius); } }
                if (0) { const double ratio = ((double)cur_enter - tmp.t) / bound; if (ratio > g_max_ratio) { g_max_ratio = ratio; fprintf(stderr, "worst: t %.6g enter %.6g c_before %.6g r %.4g |d| %.4g o (%.3f %.3f %.3f) d (%.4f %.4f %.4f) center (%.3f %.3f %.3f)\n", tmp.t, cur_enter, c_before, g_scn->spheres[n->prim].radius, 1.0f / inv_len, r->o.e[0], r->o.e[1], r->o.e[2], r->d.e[0], r->d.e[1], r->d.e[2], g_scn->spheres[n->prim].center.e[0], g_scn->spheres[n->prim].center.e[1], g_scn->spheres[n->prim].center.e[2]); } }
            }
        } else {
            float el, er; v_sah += 2; v_sah_pairs++;
            const float cl = c + g_beta * c, cr = cl; (void)bd;
            const int hl = box_test(g_s[n->left].box, r, 0.001f, cl, &el), hr = box_test(g_s[n->right].box, r, 0.001f, cr, &er);
            if (hl && hr) {
                const int lf = el <= er;
                if (sp >= g_levels) { flag = 1; n_overflow++; }
                else { stack[sp].idx = lf ? n->right : n->left; stack[sp++].enter = lf ? er : el; if (sp > maxsp) maxsp = sp; }
                cur = lf ? n->left : n->right; cur_enter = lf ? el : er; continue;
            } else if (hl) { cur = n->left; cur_enter = el; continue; }
            else if (hr) { cur = n->right; cur_enter = er; continue; }
        }
        if (sp == 0) break;
        sp--; cur = stack[sp].idx; cur_enter = stack[sp].enter;
    }
    depth_hist[maxsp]++;
    if (p >= 0) { float e; const int h = aabb_hit_e(g_rbox + 6 * p, r, 0.001f, 1e30f, &e); if (!h || c <= e) { flag = 1; n_incons_final++; } }
    (void)p_enter;
    n_sah_flag += flag;
    const int mm = (p != prim_ref) || (p >= 0 && c != c_ref);
    n_sah_mismatch += mm; if (mm && !flag) { n_sah_mismatch_unflagged++; fprintf(stderr, "UNFLAGGED MISMATCH: ref prim %d t %.9g  got prim %d t %.9g  o (%.9g %.9g %.9g) d (%.9g %.9g %.9g)\n", prim_ref, c_ref, p, c, r->o.e[0], r->o.e[1], r->o.e[2], r->d.e[0], r->d.e[1], r->d.e[2]); }
}
/* ---- WIDE=1: the same tree collapsed to 4-wide nodes (a child that is an inner node is replaced by its two children,
 * largest box first, until the node has four children or only leaves); near-first walk: all children tested in one
 * step, the nearest hit child is entered, the others are pushed far-to-near. */
typedef struct { int child[4]; int n; } wnode;      /* child: index into g_s (leaf iff g_s[c].left < 0) */
static wnode *g_w; static int *g_wof;               /* g_wof[binary inner node] = its wide node */
static unsigned long long w_steps, w_boxes, w_leaf, w_mismatch, w_push, w_maxsp_hist[64];
static void wide_build(int i) {
    wnode *w = &g_w[i];
    w->n = 2; w->child[0] = g_s[i].left; w->child[1] = g_s[i].right;
    for (;;) {
        if (w->n == 4) break;
        int best = -1; float ba = -1;
        for (int k = 0; k < w->n; k++) if (g_s[w->child[k]].left >= 0) { const float a = area(g_s[w->child[k]].box); if (a > ba) { ba = a; best = k; } }
        if (best < 0) break;
        const int c = w->child[best];
        w->child[best] = g_s[c].left; w->child[w->n++] = g_s[c].right;
    }
    for (int k = 0; k < w->n; k++) if (g_s[w->child[k]].left >= 0) wide_build(w->child[k]);
}
static void wide_ray(const ray *r, float c_ref, int prim_ref) {
    float c = 1e30f; int p = -1;
    struct { int idx; float enter; } stack[128]; int sp = 0, maxsp = 0;
    int cur = 0; float e0;
    if (!box_test(g_s[0].box, r, 0.001f, 1e30f, &e0)) cur = -1;
    while (cur >= 0) {
        if (g_s[cur].left < 0) {
            hitrec tmp; w_leaf++;
            if (hit_sphere(r, 0.001f, c, &tmp, &g_scn->spheres[g_s[cur].prim])) { c = tmp.t; p = g_s[cur].prim; }
        } else {
            const wnode *w = &g_w[cur];
            w_steps++; w_boxes += w->n;
            int hi[4]; float he[4]; int nh = 0;
            const float cl = c + g_beta * c;
            for (int k = 0; k < w->n; k++) { float e; if (box_test(g_s[w->child[k]].box, r, 0.001f, cl, &e)) { hi[nh] = w->child[k]; he[nh++] = e; } }
            for (int a = 1; a < nh; a++) for (int b = a; b > 0 && he[b] < he[b - 1]; b--) { const float te = he[b]; he[b] = he[b - 1]; he[b - 1] = te; const int ti = hi[b]; hi[b] = hi[b - 1]; hi[b - 1] = ti; }
            for (int k = nh - 1; k >= 1; k--) { stack[sp].idx = hi[k]; stack[sp++].enter = he[k]; w_push++; }
            if (sp > maxsp) maxsp = sp;
            if (nh > 0) { cur = hi[0]; continue; }
        }
        if (sp == 0) break;
        cur = stack[--sp].idx;
    }
    w_maxsp_hist[maxsp < 63 ? maxsp : 63]++;
    if ((p != prim_ref) || (p >= 0 && c != c_ref)) w_mismatch++;
}
/* ---- DEFER=k: the guarded walk with a one-entry "pending primitive": a lane that reaches a leaf does not stop for the
 * primitive test, it parks the leaf and walks on; the parked leaf is tested when a second leaf is reached, after k more
 * pair steps (the wave's next leaf step), or at the end.  Counts the extra box tests the later pruning costs. */
static int g_defer = 0;
static unsigned long long df_pairs, df_leaf, df_mismatch;
static void defer_ray(const ray *r, float c_ref, int prim_ref) {
    float c = 1e30f; int p = -1;
    int stack[128]; int sp = 0;
    int pend = -1, pend_age = 0;
    int cur = 0; float e0;
    if (!box_test(g_s[0].box, r, 0.001f, 1e30f, &e0)) cur = -1;
    int done = cur < 0;
    for (;;) {
        if (pend >= 0 && (pend_age >= g_defer || done || g_s[cur].left < 0)) {      /* test the parked leaf */
            hitrec tmp; df_leaf++;
            if (hit_sphere(r, 0.001f, c, &tmp, &g_scn->spheres[g_s[pend].prim])) { c = tmp.t; p = g_s[pend].prim; }
            pend = -1;
        }
        if (done) break;
        const snode *n = &g_s[cur];
        int next = -1;
        if (n->left < 0) { pend = cur; pend_age = 0; }
        else {
            float el, er; df_pairs++; pend_age++;
            const float cl = c + g_beta * c;
            const int hl = box_test(g_s[n->left].box, r, 0.001f, cl, &el), hr = box_test(g_s[n->right].box, r, 0.001f, cl, &er);
            if (hl && hr) { const int lf = el <= er; stack[sp++] = lf ? n->right : n->left; next = lf ? n->left : n->right; }
            else if (hl) next = n->left;
            else if (hr) next = n->right;
        }
        if (next < 0) { if (sp == 0) done = 1; else next = stack[--sp]; }
        cur = next;
    }
    if ((p != prim_ref) || (p >= 0 && c != c_ref)) df_mismatch++;
}

/* ---- LEAFK=k: leaves of up to k spheres (no per-sphere boxes): a leaf visit runs k discriminant tests; counts pair steps,
 * leaf visits, discriminant tests and tests that reach the fp64 roots. */
typedef struct { float box[6]; int left, right, first, count; } knode;
static knode *g_k; static int g_kn, g_leafk, *g_kids;
static unsigned long long k_pairs, k_visits, k_disc, k_roots, k_mismatch, k_bottom;
static int leafk_build(int *ids, int n) {
    const int me = g_kn++;
    float bb[6] = {1e30f, -1e30f, 1e30f, -1e30f, 1e30f, -1e30f};
    for (int i = 0; i < n; i++) grow(bb, g_pbox + 6 * ids[i]);
    memcpy(g_k[me].box, bb, sizeof bb);
    if (n <= g_leafk) { g_k[me].left = -1; g_k[me].first = (int)(ids - g_kids); g_k[me].count = n; return me; }
    float best = 1e30f; int best_axis = 0, best_k = n / 2;
    int *tmp = malloc(n * sizeof(int)); float *ra = malloc(n * sizeof(float));
    for (int a = 0; a < 3; a++) {
        memcpy(tmp, ids, n * sizeof(int)); g_cmp_axis = a; qsort(tmp, n, sizeof(int), cmp_prim);
        float rb[6] = {1e30f, -1e30f, 1e30f, -1e30f, 1e30f, -1e30f};
        for (int i = n - 1; i > 0; i--) { grow(rb, g_pbox + 6 * tmp[i]); ra[i] = area(rb); }
        float lb[6] = {1e30f, -1e30f, 1e30f, -1e30f, 1e30f, -1e30f};
        for (int k = 1; k < n; k++) {
            grow(lb, g_pbox + 6 * tmp[k - 1]);
            const float cost = area(lb) * k + ra[k] * (n - k);
            if (cost < best) { best = cost; best_axis = a; best_k = k; }
        }
    }
    g_cmp_axis = best_axis; qsort(ids, n, sizeof(int), cmp_prim);
    free(tmp); free(ra);
    const int l = leafk_build(ids, best_k), r = leafk_build(ids + best_k, n - best_k);
    g_k[me].left = l; g_k[me].right = r;
    return me;
}
static void leafk_ray(const ray *r, float c_ref, int prim_ref) {
    float c = 1e30f; int p = -1;
    int stack[128]; int sp = 0;
    int cur = 0; float e0;
    if (!box_test(g_k[0].box, r, 0.001f, 1e30f, &e0)) cur = -1;
    while (cur >= 0) {
        const knode *n = &g_k[cur];
        int next = -1;
        if (n->left < 0) {
            k_visits++;
            for (int q = 0; q < n->count; q++) {
                const int prim = g_kids[n->first + q];
                const rt_sphere *s = &g_scn->spheres[prim];
                k_disc++;
                { v3 oc = sub(r->o, from_rt(s->center)); float a = lensq(r->d), hb = dot(oc, r->d), cc = lensq(oc) - s->radius * s->radius; if (hb * hb - a * cc >= 0) k_roots++; }
                hitrec tmp;
                if (hit_sphere(r, 0.001f, c, &tmp, s)) { c = tmp.t; p = prim; }
            }
        } else {
            float el, er; k_pairs++;
            if (g_k[n->left].left < 0 && g_k[n->right].left < 0) k_bottom++;
            const float cl = c + g_beta * c;
            const int hl = box_test(g_k[n->left].box, r, 0.001f, cl, &el), hr = box_test(g_k[n->right].box, r, 0.001f, cl, &er);
            if (hl && hr) { const int lf = el <= er; stack[sp++] = lf ? n->right : n->left; next = lf ? n->left : n->right; }
            else if (hl) next = n->left;
            else if (hr) next = n->right;
        }
        if (next < 0) { if (sp == 0) break; next = stack[--sp]; }
        cur = next;
    }
    if ((p != prim_ref) || (p >= 0 && c != c_ref)) k_mismatch++;
}

/* ---- MODEL=1: the kernel's pair-step walk (round 4 study): what would tighter boxes and primitives tested at the start of
 * the ray buy?  Its own SAH tree (TOPBIG=1: without the spheres whose leaf box covers more than half of the root's surface —
 * those are tested before the walk, so the walk starts with their hit as its `closest`), node boxes as the kernel holds them
 * (HALF16=1: rounded outward to binary16), and the distance-aware growth by one of three rules:
 *   GROW=1  dyn_k D^2, D = distance from the origin to the farthest corner of the UNION of the two children (round 3's walk, since removed)
 *   GROW=4  the kernel's rule (step_pair_par): k (T |d| + sqrt(3) (r_max + e))^2, T = the exit parameter of the (grown, clipped) box the
 *           node was entered through, e the growth that box was tested with — every accepted hit below the node lies on the ray inside
 *           that box, so T |d| bounds its distance; a node off the stack gets the ray's own bound (corner of the tree / the hit in hand).
 *           Every accepted hit of a small sphere is checked against the induction hypothesis (growth of its box >= k |o - c|^2).
 *   GROW=2  the same with the parent's growth kept on the stack for a popped node
 *   GROW=3  per child, farthest corner of the child's own box (box_test above)
 *   GROW=0  none (the leaf boxes carry static margins) */
static snode *g_m; static int g_mn; static float *g_mbox;      /* node boxes as walked */
static int g_grow, g_half16, g_topbig, *g_top, g_ntop;
static float g_rho;
static unsigned long long m_pairs, m_leaf, m_top, m_mismatch, m_flag, m_mismatch_unflagged, m_pops;
static float down16(float x) { if (x == 0 || !isfinite(x)) return x; int e; frexpf(x, &e); const float q = ldexpf(1.0f, e - 11); return floorf(x / q) * q; }
static float up16(float x) { if (x == 0 || !isfinite(x)) return x; int e; frexpf(x, &e); const float q = ldexpf(1.0f, e - 11); return ceilf(x / q) * q; }
static void planes_of(const float *box, const ray *r, float nr[3], float fr[3], float ainv[3]) {
    for (int a = 0; a < 3; a++) {
        float inv = 1 / r->d.e[a];
        inv = fminf(fmaxf(inv, -1e18f), 1e18f);
        const float noi = -(r->o.e[a] * inv);
        const float ta = fmaf(box[2 * a], inv, noi), tb = fmaf(box[2 * a + 1], inv, noi);
        nr[a] = fminf(ta, tb); fr[a] = fmaxf(ta, tb); ainv[a] = fabsf(inv);
    }
}
static unsigned long long m_inv_checked, m_inv_broken;
/* the kernel's par_growth (rt_kernel.hip.inc): E(T, e) = (sqrt(k) |d| T + sqrt(k) sqrt(3) r_max + sqrt(k) sqrt(3) e)^2 */
static float g_sqrtk, g_parb, g_parc3;
static inline float par_growth(float T, float ea, float e) { const float s = fmaf(e, g_parc3, fmaf(T, ea, g_parb)); return s * s; }
static void model_ray(const ray *r, float c_ref, int prim_ref) {
    float c = 1e30f; int p = -1, flag = 0;
    for (int k = 0; k < g_ntop; k++) {
        hitrec tmp; m_top++;
        if (hit_sphere(r, 0.001f, c, &tmp, &g_scn->spheres[g_top[k]])) { if (tmp.t == c && p >= 0) flag = 1; c = tmp.t; p = g_top[k]; }
    }
    struct { int idx; float e; } stack[128]; int sp = 0;
    const int par = g_grow == 2 || g_grow == 4;
    const float ea = g_sqrtk * 1.000002f * sqrtf(lensq(r->d));
    int cur = g_mn > 0 ? 0 : -1;
    float e_cur = 0, er = 0, tested_e = 1e30f;      /* tested_e: the growth the current node's own box was tested with */
    if (cur >= 0 && par) {      /* arm_ray: the corner bound of the whole tree, or what a hit already in hand allows */
        float d2 = 0; const float *b = g_mbox;
        for (int a = 0; a < 3; a++) { const float m = fmaxf(fabsf(r->o.e[a] - b[2 * a]), fabsf(b[2 * a + 1] - r->o.e[a])); d2 = fmaf(m, m, d2); }
        er = g_dynk * 1.0009765625f * d2;
        er = fminf(er, par_growth(c, ea, er));
        e_cur = er;
    }
    if (cur >= 0 && g_m[0].left < 0) { hitrec tmp; m_leaf++; if (hit_sphere(r, 0.001f, c, &tmp, &g_scn->spheres[g_m[0].prim])) { c = tmp.t; p = g_m[0].prim; } cur = -1; }
    while (cur >= 0) {
        const snode *n = &g_m[cur];
        int next = -1; float e_next = 0, tested_next = 1e30f;
        if (n->left < 0) {
            hitrec tmp; m_leaf++;
            if (hit_sphere(r, 0.001f, c, &tmp, &g_scn->spheres[n->prim])) {
                if (tmp.t == c && p >= 0) flag = 1;
                c = tmp.t; p = n->prim;
                if (par) {
                    /* the induction hypothesis of step_pair_par, on this very hit: the growth the leaf's box was tested with
                     * covers k |o - c_q|^2 (small spheres: the ones the growth exists for) */
                    const rt_sphere *q = &g_scn->spheres[n->prim];
                    if (q->radius < 100) {
                        double x2 = 0; for (int a = 0; a < 3; a++) x2 += ((double)r->o.e[a] - q->center.e[a]) * ((double)r->o.e[a] - q->center.e[a]);
                        m_inv_checked++;
                        if ((double)g_dynk * x2 > (double)tested_e * (1.0 + 1e-5)) m_inv_broken++;
                    }
                    er = fminf(er, par_growth(c, ea, er));      /* the leaf step: the ray's own bound comes down with the hit */
                }
            }
        } else {
            m_pairs++;
            const float t_far = c * (1.0f + 9.5367431640625e-7f);
            const float *b0 = g_mbox + 6 * n->left, *b1 = g_mbox + 6 * n->right;
            float n0[3], f0[3], n1[3], f1[3], ai[3];
            planes_of(b0, r, n0, f0, ai); planes_of(b1, r, n1, f1, ai);
            float g0 = 0, g1 = 0;
            if (g_grow == 1) {
                float d2 = 0;
                for (int a = 0; a < 3; a++) { const float m = fmaxf(fmaxf(fabsf(n0[a]), fabsf(n1[a])), fmaxf(fabsf(f0[a]), fabsf(f1[a]))) * fmaxf(fabsf(r->d.e[a]), 1e-18f); d2 = fmaf(m, m, d2); }
                g0 = g1 = g_dynk * 1.0009765625f * d2;
            } else if (g_grow == 3) {
                for (int k = 0; k < 2; k++) { const float *b = k ? b1 : b0; float d2 = 0;
                    for (int a = 0; a < 3; a++) { const float m = fmaxf(fabsf(r->o.e[a] - b[2 * a]), fabsf(b[2 * a + 1] - r->o.e[a])); d2 = fmaf(m, m, d2); }
                    if (k) g1 = g_dynk * d2; else g0 = g_dynk * d2; }
            } else if (par) g0 = g1 = e_cur;
            float e0 = 0.001f, q0 = t_far, e1 = 0.001f, q1 = t_far;
            for (int a = 0; a < 3; a++) {
                e0 = fmaxf(e0, n0[a] - g0 * ai[a]); q0 = fminf(q0, f0[a] + g0 * ai[a]);
                e1 = fmaxf(e1, n1[a] - g1 * ai[a]); q1 = fminf(q1, f1[a] + g1 * ai[a]);
            }
            const int h0 = q0 > e0, h1 = q1 > e1;
            const int second_first = h1 && !(h0 && e0 <= e1);
            const int near_c = second_first ? n->right : n->left, far_c = second_first ? n->left : n->right;
            const float q_near = second_first ? q1 : q0;
            if (h0 && h1) { stack[sp].idx = far_c; stack[sp++].e = e_cur; }       /* (e: what its box was tested with — for the check above) */
            if (h0 || h1) {
                next = near_c; tested_next = e_cur;
                if (par) e_next = fminf(e_cur, par_growth(q_near, ea, e_cur));
            }
        }
        if (next < 0) { if (sp == 0) break; --sp; next = stack[sp].idx; tested_next = stack[sp].e; m_pops++;
            e_next = g_grow == 2 ? fminf(stack[sp].e, er) : er; }      /* GROW=4 (the kernel): nothing but the node is kept on the stack */
        cur = next; e_cur = e_next; tested_e = tested_next;
    }
    if (p >= 0) { float e; const int h = aabb_hit_e(g_rbox + 6 * p, r, 0.001f, 1e30f, &e); if (!h || c <= e) flag = 1; }
    m_flag += flag;
    const int mm = (p != prim_ref) || (p >= 0 && c != c_ref);
    m_mismatch += mm;
    if (mm && !flag) { m_mismatch_unflagged++; if (m_mismatch_unflagged < 10) fprintf(stderr, "MODEL UNFLAGGED MISMATCH: ref prim %d t %.9g  got prim %d t %.9g\n", prim_ref, c_ref, p, c); }
}
static void model_setup(const rt_scene_desc *sc) {
    g_grow = getenv("GROW") ? atoi(getenv("GROW")) : 1;
    g_half16 = getenv("HALF16") != NULL; g_topbig = getenv("TOPBIG") != NULL;
    /* root box of everything */
    float rb[6] = {1e30f, -1e30f, 1e30f, -1e30f, 1e30f, -1e30f};
    for (int i = 0; i < sc->num_spheres; i++) grow(rb, g_pbox + 6 * i);
    int *ids = malloc(sizeof(int) * sc->num_spheres), n = 0;
    g_top = malloc(sizeof(int) * sc->num_spheres);
    float rmax_small = 0;
    for (int i = 0; i < sc->num_spheres; i++) {
        if (g_topbig && area(g_pbox + 6 * i) > 0.5f * area(rb)) g_top[g_ntop++] = i; else ids[n++] = i;
        if (sc->spheres[i].radius < 100 && sc->spheres[i].radius > rmax_small) rmax_small = sc->spheres[i].radius;
    }
    g_rho = rmax_small * 1.001f;
    g_sqrtk = sqrtf(g_dynk * 1.0009765625f) * 1.000001f; g_parb = g_sqrtk * 1.7320509f * rmax_small * 1.000001f; g_parc3 = g_sqrtk * 1.7320509f;
    snode *save = g_s; const int save_n = g_sn;
    g_m = malloc(sizeof(snode) * 2 * (n + 1)); g_s = g_m; g_sn = 0;
    if (n > 0) sah_build(ids, n);
    g_mn = g_sn; g_s = save; g_sn = save_n;
    g_mbox = malloc(sizeof(float) * 6 * (g_mn + 1));
    for (int i = 0; i < g_mn; i++) for (int a = 0; a < 3; a++) {
        g_mbox[6 * i + 2 * a] = g_half16 ? down16(g_m[i].box[2 * a]) : g_m[i].box[2 * a];
        g_mbox[6 * i + 2 * a + 1] = g_half16 ? up16(g_m[i].box[2 * a + 1]) : g_m[i].box[2 * a + 1];
    }
    g_model_on = 1;
    printf("MODEL: grow %d half16 %d topbig %d (%d primitives tested before the walk), tree of %d nodes, rho %.4g dyn_k %.4g\n", g_grow, g_half16, g_topbig, g_ntop, g_mn, g_rho, g_dynk);
}
static float fill_rmax(int i) {
    if (g_s[i].left < 0) return g_rmax[i] = g_scn->spheres[g_s[i].prim].radius;
    const float a = fill_rmax(g_s[i].left), b = fill_rmax(g_s[i].right);
    return g_rmax[i] = a > b ? a : b;
}

int main(int argc, char **argv) {
    const int half = argc > 1 ? atoi(argv[1]) : 11;
    const int W = argc > 2 ? atoi(argv[2]) : 192, H = argc > 3 ? atoi(argv[3]) : 108, spp = argc > 4 ? atoi(argv[4]) : 8;
    if (argc > 5) g_beta = (float)atof(argv[5]);
    if (argc > 6) g_levels = atoi(argv[6]);
    if (getenv("FUSED")) g_fused = 1;
    if (getenv("DYN")) { g_dyn = 1; g_fused = 1; }
    if (getenv("GAMMA_ULPS")) g_gamma = (float)atof(getenv("GAMMA_ULPS")) * 5.9604645e-8f;
    rtp_host_scene *hs = rtp_host_scene_rtiow(12345u, half, 0, 0);
    rt_scene_desc sc; rtp_host_scene_desc(hs, &sc);
    rt_camera_data cam;
    const float eye[3] = {13, 3, 2}, at[3] = {0, 0, 0}, bg[3] = {0.7f, 0.8f, 1.0f};
    rtp_host_make_camera(W, H, 20.0f, eye, at, bg, spp, 50, &cam);
    g_axis = calloc(sc.num_nodes, sizeof(int));
    for (int i = 0; i < sc.num_nodes; i++) {
        const rt_bvh_node *n = &sc.nodes[i];
        if (n->left < 0) continue;
        const float *L = sc.nodes[n->left].box, *R = sc.nodes[n->right].box;
        float best = -1e30f; int ax = 0;
        for (int a = 0; a < 3; a++) {
            const float sep = (R[2 * a] + R[2 * a + 1]) - (L[2 * a] + L[2 * a + 1]);
            const float ext = n->box[2 * a + 1] - n->box[2 * a];
            const float s = sep / (ext > 0 ? ext : 1);
            if (s > best) { best = s; ax = a; }
        }
        g_axis[i] = ax;
    }
    g_scn = &sc;
    g_pbox = malloc(sizeof(float) * 6 * sc.num_spheres);
    for (int i = 0; i < sc.num_nodes; i++) if (sc.nodes[i].left < 0) memcpy(g_pbox + 6 * sc.nodes[i].right, sc.nodes[i].box, 24);
    g_rbox = malloc(sizeof(float) * 6 * sc.num_spheres);
    memcpy(g_rbox, g_pbox, sizeof(float) * 6 * sc.num_spheres);
    {   /* guard centre: centroid box of the spheres with radius below 10x the median-ish (here: < 100) */
        float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
        for (int i = 0; i < sc.num_spheres; i++) if (sc.spheres[i].radius < 100) for (int a = 0; a < 3; a++) {
            if (sc.spheres[i].center.e[a] < lo[a]) lo[a] = sc.spheres[i].center.e[a];
            if (sc.spheres[i].center.e[a] > hi[a]) hi[a] = sc.spheres[i].center.e[a]; }
        for (int a = 0; a < 3; a++) g_sc[a] = 0.5f * (lo[a] + hi[a]);
        float rmin = 1e30f; g_rs = 0;
        for (int a = 0; a < 3; a++) { g_bs[2 * a] = 1e30f; g_bs[2 * a + 1] = -1e30f; }
        for (int i = 0; i < sc.num_spheres; i++) if (sc.spheres[i].radius < 100) {
            float dc = 0; for (int a = 0; a < 3; a++) dc += (sc.spheres[i].center.e[a] - g_sc[a]) * (sc.spheres[i].center.e[a] - g_sc[a]);
            if (sqrtf(dc) + sc.spheres[i].radius > g_rs) g_rs = sqrtf(dc) + sc.spheres[i].radius;
            if (sc.spheres[i].radius < rmin) rmin = sc.spheres[i].radius;
            for (int a = 0; a < 3; a++) { if (g_rbox[6 * i + 2 * a] < g_bs[2 * a]) g_bs[2 * a] = g_rbox[6 * i + 2 * a]; if (g_rbox[6 * i + 2 * a + 1] > g_bs[2 * a + 1]) g_bs[2 * a + 1] = g_rbox[6 * i + 2 * a + 1]; }
        }
        g_fark = g_gamma / (2 * rmin);
        g_dynk = g_gamma / (2 * rmin) * 1.000001f;
        { float reach = sqrtf(0.08f * rmin * rmin / g_gamma); if (reach < 2 * g_rs) reach = 2 * g_rs; g_d0 = reach - g_rs; }
        if (argc > 7) g_d0 = (float)atof(argv[7]);
        for (int i = 0; i < sc.num_spheres; i++) {
            float dc = 0; for (int a = 0; a < 3; a++) dc += (sc.spheres[i].center.e[a] - g_sc[a]) * (sc.spheres[i].center.e[a] - g_sc[a]);
            const float reach = sc.spheres[i].radius < 100 ? g_d0 + g_rs : 1.25f * 2002.0f;
            float eps = g_gamma * reach * reach / (2 * sc.spheres[i].radius);
            if (g_dyn && sc.spheres[i].radius < 100) eps = 32 * 5.96e-8f * (2 * g_rs + 2002.0f);      /* the rounding floor only */
            for (int a = 0; a < 3; a++) { g_pbox[6 * i + 2 * a] -= eps; g_pbox[6 * i + 2 * a + 1] += eps; }
            if (i < 3 || i == sc.num_spheres - 1) printf("sphere %d r %.3g eps %.3g\n", i, sc.spheres[i].radius, eps);
        }
    }
    int *ids = malloc(sizeof(int) * sc.num_spheres);
    for (int i = 0; i < sc.num_spheres; i++) ids[i] = i;
    snode *sn = malloc(sizeof(snode) * 2 * sc.num_spheres);
    g_s = sn; g_sn = 0; sah_build(ids, sc.num_spheres); g_rmax = malloc(sizeof(float) * g_sn); fill_rmax(0); g_s_any = sn;
    if (getenv("DEFER")) { g_defer = atoi(getenv("DEFER")); g_defer_on = 1; }
    if (getenv("LEAFK")) { g_leafk = atoi(getenv("LEAFK")); g_kids = malloc(sizeof(int) * sc.num_spheres); for (int i = 0; i < sc.num_spheres; i++) g_kids[i] = i;
        g_k = malloc(sizeof(knode) * 2 * sc.num_spheres); g_kn = 0; leafk_build(g_kids, sc.num_spheres); g_leafk_on = 1; printf("LEAFK tree: %d nodes\n", g_kn); }
    if (getenv("MODEL")) model_setup(&sc);
    if (getenv("WIDE")) { g_w = calloc(g_sn, sizeof(wnode)); wide_build(0); g_w_any = g_w; }
    float *fb = malloc((size_t)W * H * 3 * sizeof(float));
    orc_render(&sc, &cam, 0, H, fb, 1, NULL);
    printf("scene half=%d nodes=%d  %dx%d spp=%d delta=%g\n", half, sc.num_nodes, W, H, spp, g_delta);
    printf("rays %llu  visits/ray ref %.2f  octant-order %.2f (%.1f%%)  leaf tests/ray ref %.2f oct %.2f\n", n_rays, (double)v_ref / n_rays,
           (double)v_oct / n_rays, 100.0 * v_oct / v_ref, (double)lt_ref / n_rays, (double)lt_oct / n_rays);
    printf("flagged %.4f%% (prune band %.4f%%, inconsistent accept %.6f%%)  mismatches %llu, of which unflagged %llu\n", 100.0 * n_flag / n_rays,
           100.0 * n_flag_prune / n_rays, 100.0 * n_flag_incons / n_rays, n_mismatch, n_mismatch_unflagged);
    printf("SAH tree, near-first, inflated pruning beta=%g levels=%d: box tests/ray %.2f (pair steps %.2f)  leaf tests/ray %.2f\n", g_beta, g_levels,
           (double)v_sah / n_rays, (double)v_sah_pairs / n_rays, (double)lt_sah / n_rays);
    printf("  flagged %.4f%% (ties %llu, inconsistent final %llu, overflow %llu)  mismatches %llu (unflagged %llu)  far origins %llu (flagged %llu)  max departure/eps %.3g\n",
           100.0 * n_sah_flag / n_rays, n_tie, n_incons_final, n_overflow, n_sah_mismatch, n_sah_mismatch_unflagged, n_far, n_far_flag, g_max_ratio);
    if (g_w) {
        printf("WIDE (4-wide collapse): steps/ray %.2f  box tests/ray %.2f  leaf tests/ray %.2f  pushes/ray %.2f  mismatches (unguarded) %llu  max stack:", (double)w_steps / n_rays,
               (double)w_boxes / n_rays, (double)w_leaf / n_rays, (double)w_push / n_rays, w_mismatch);
        for (int i = 0; i < 64; i++) if (w_maxsp_hist[i]) printf(" %d:%.3f%%", i, 100.0 * w_maxsp_hist[i] / n_rays);
        printf("\n");
    }
    if (g_defer_on) printf("DEFER (parked leaf, tested after %d pair steps): pair steps/ray %.2f  leaf tests/ray %.2f  mismatches (unguarded) %llu\n", g_defer,
                           (double)df_pairs / n_rays, (double)df_leaf / n_rays, df_mismatch);
    if (g_leafk_on) printf("LEAFK=%d: pair steps/ray %.2f (bottom pairs %.2f)  leaf visits/ray %.2f  discriminants/ray %.2f  reaching roots/ray %.2f  mismatches (unguarded) %llu\n", g_leafk,
                           (double)k_pairs / n_rays, (double)k_bottom / n_rays, (double)k_visits / n_rays, (double)k_disc / n_rays, (double)k_roots / n_rays, k_mismatch);
    if (g_model_on) printf("MODEL: pair steps/ray %.2f  leaf tests/ray %.2f (+ %.2f before the walk)  pops/ray %.2f  flagged %.4f%%  mismatches %llu (unflagged %llu)  growth bound checked on %llu hits, broken %llu\n", (double)m_pairs / n_rays,
                           (double)m_leaf / n_rays, (double)m_top / n_rays, (double)m_pops / n_rays, 100.0 * m_flag / n_rays, m_mismatch, m_mismatch_unflagged, m_inv_checked, m_inv_broken);
    printf("  max pending-stack depth per ray:");
    for (int i = 0; i < 24; i++) if (depth_hist[i]) printf(" %d:%.4f%%", i, 100.0 * depth_hist[i] / n_rays);
    printf("\n");
    return 0;
}
