// rt_capi.hip — implementation of the C ABI in include/rtp_amd.h on top of the gfx950 kernels.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/rtp_amd.h"
#include "rt_accel.h"
#include "rt_build.h"
#include "rt_device_math.h"
#include "rt_kernel.hip.inc"
#include "rt_primary.hip.inc"
// Developer build only (make dev → librtp_amd_dev.so, -DRTP_DEV_BUILD): the two experimental kernels that lost to render_kernel
// (rt_kernel_wf.hip.inc: wave-owned path pools in L2, −37 %; rt_kernel_queue.hip.inc: T-wave/S-wave LDS queues, −30 %; docs/LOG.md) and
// the rt_debug_* entry points (exhaustive on-device checks of recip / sqrt_cr / sphere_root, the RTP_STATS counters).  The shipped
// library contains none of them.
#ifdef RTP_DEV_BUILD
#define RTP_DEV_QUEUE_KERNEL 1
#include "rt_kernel_wf.hip.inc"
#include "rt_kernel_queue.hip.inc"
#else
namespace rtk { constexpr int kWfBlock = 256, kWfRayRows = 0, kWfHitRows = 0; }
#define RTP_WF_MIN_WAVES 4
#endif

namespace {

thread_local std::string g_last_error;

rt_status fail(rt_status st, const std::string &msg) {
    g_last_error = msg;
    return st;
}

#define HIP_TRY(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return fail(e_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP,                   \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                              \
    } while (0)

// Only rt_config_from_env() (an explicit call of the host) reads the environment.
int env_int(const char *name, int fallback) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : fallback;
}

// rt_config.workspace_bytes == 0: a sixteenth of the device's memory (18 GB of an MI355X's 288 GB).  The sample slab of a
// pass is the only large allocation of a scene handle and is grown on demand to what the frames actually need: 12.4 GB for
// one 1920x1080x500-spp pass.  A smaller budget only cuts the frame into more passes: 4 GiB → 3 passes, 1.9 % slower
// (each extra pass costs the drain of a trace launch and one more re-walk launch, ≈1.7 ms).
#ifndef RTP_BY_PIXEL_MIN
#define RTP_BY_PIXEL_MIN 96         /* samples per pixel and pass from which the primary-visibility pass takes a wave per PIXEL (configs[4], passes of 125: 10.6 instead of 11.9 ms; S-rtiow at 100 spp: 1.84 vs 1.91) */
#endif
constexpr uint64_t kWorkspaceShareOfDevice = 16;
constexpr uint64_t kSampleBytes = 12;        // one radiance record of the slab

constexpr uint32_t kLdsLimit = 160 * 1024;
constexpr int kMaxPasses = 1024;        // >= 64 samples per pass
constexpr int kTimedPasses = 64;        // trace launches individually timed per call

// Counter block of a scene handle (uint32 words): work counters of the trace launches, of the exact
// re-walk launches, the flagged-sample counts (one each per pass), then 16 developer words.
#ifndef RTP_TAPER_FACTOR
#define RTP_TAPER_FACTOR 2          /* reservations shrink to remaining / (this x waves) */
#endif
constexpr int kQueueWork = 0, kQueueRework = kMaxPasses, kQueueFlag = 2 * kMaxPasses, kQueueStats = 3 * kMaxPasses;
constexpr int kQueueDirty = 3 * kMaxPasses + 16;      // per pass: pixels with a flagged sample (overlapped re-walk)
constexpr int kQueueAbandon = 4 * kMaxPasses + 16;    // per pass: the guarded launch gave up part-way (render_kernel, flag_reserve)
constexpr int kQueueHoles = 5 * kMaxPasses + 16;      // per pass: slots of the flagged-sample list reserved and never filled (flag_chunk_drain)
constexpr int kQueueWords = 6 * kMaxPasses + 16;
static_assert(kQueueHoles - kQueueFlag == (int)rtk::kFlagHolesWords, "the holes word sits where the kernels look for it");
// LDS of a guarded kernel that is neither tables nor stacks nor work ranges: its constants block and, behind it, two words per wave
// (of at most sixteen) for the wave's chunk of the flagged-sample list (rt_kernel.hip.inc: fill_consts, flag_collect)
constexpr uint32_t kGuardBlockBytes = 16u * (uint32_t)rtk::kConstRows + (uint32_t)(rtk::kSimpleBlock / rtk::kWave) * 8u;
// What a render call leaves for the NEXT one to read (rt_scene::feedback): per pass the flagged count and the abandon word, copied
// to pinned host memory at the end of the call, with an event — the handle's decision to step aside from the guarded walk needs no
// rt_last_timing and no synchronisation of the caller's.
constexpr int kFeedbackSlots = 4;
constexpr uint32_t kDefaultBailShare = 64;            // of 256: 25 % (rt_config.guard_bail_share)
constexpr uint32_t kDefaultBailLatest = 2;            // of 8: a pass is given up in its first quarter or not at all
constexpr uint32_t kDefaultBailFloor = 96;            // of 1024: 9.4 % of the pass on the list before it is given up (swept with kDefaultBailLatest: docs/LOG.md round 4)
constexpr uint32_t kExploreShare = 1;                 // of 256: a guarded frame that flagged more than 0.4 % is timed against an exact one …
constexpr float kExploreOverhead = 0.15f;             // … and so is one that spent more than this share of its time outside the trace launch

// Traversal (rt_config.traversal).  EXACT ("threaded"): the caller's tree in the reference's own visit order — the
// result is the reference's by construction.  GUARDED (AUTO's choice where the scene is eligible): near-first walk
// of an SAH tree over inflated leaf boxes; every sample whose result could depend on the visit order
// is flagged and re-walked in threaded mode, so the frame is the same (docs/LOG.md §3b).
// Trees of a few dozen primitives have nothing to gain from a second walk (random scenes of 20-60 spheres: the guarded frame
// 1.3-1.4 x the exact one — its re-walk launch is a fixed half millisecond); everything else eligible gets the
// guarded one (the reference's default scene, ~200 primitives: 12.4 vs 9.0 Gsamples/s).
bool guarded_wanted(const rt_config &cfg, int64_t primitives) {
    if (cfg.traversal == RT_TRAVERSAL_GUARDED) return true;
    return primitives >= cfg.guard_min_primitives;
}

void config_defaults(rt_config &c) {
    std::memset(&c, 0, sizeof(c));
    c.struct_bytes = (uint32_t)sizeof(rt_config);
    c.tree_build = RT_BUILD_HOST_SAH;
    c.guard_gamma_ulps = 0.0f;
    c.traversal = RT_TRAVERSAL_AUTO;
    c.guard_min_primitives = 64;
    c.guard_repack = 1;
    c.kernel = RT_KERNEL_AUTO;
    c.workspace_bytes = 0;
    c.scene_in_lds = 1;
    c.lds_treelet = 1;
    c.reserve_taper = 1;
    c.wide_nodes = 0;
    c.guard_bail_share = 0;
    c.guard_front_primitives = 0;
    c.reuse_view_lists = 0;
    c.resume_flagged = 0;
}

// A caller compiled against an older, shorter rt_config: its fields, defaults for the rest.
rt_config config_from_caller(const rt_config *in) {
    rt_config c;
    config_defaults(c);
    if (in && in->struct_bytes >= 8) {
        const size_t n = in->struct_bytes < sizeof(rt_config) ? in->struct_bytes : sizeof(rt_config);
        std::memcpy(&c, in, n);
        c.struct_bytes = (uint32_t)sizeof(rt_config);
    }
    if (c.guard_min_primitives < 0) c.guard_min_primitives = 0;
    return c;
}

// Pair nodes the LDS-resident guarded walk can hold at two workgroups per CU with a stack of seven entries (what render_impl's own
// fit test comes to for the sphere-only build and the general one) — PackOptions::lds_pair_budget: scenes beyond it are packed for
// the walk through L1 / L2 where that costs nothing.  An estimate is enough: both walks render the same frame.
int32_t lds_pair_budget(const rt_scene_desc &d) {
    bool plain = d.num_planes == 0;          // the sphere-only build: no planes, no textures (absorbing glass aside: the estimate may be a little generous)
    for (int i = 0; plain && i < d.num_materials; ++i) plain = d.materials[i].texture_id == 0;
    const int64_t budget = (int64_t)kLdsLimit / 2;
    const int64_t lanes = plain ? rtk::kSimpleBlock : rtk::kBlock;
    const int64_t fixed = (int64_t)d.num_spheres * 16 + ((int64_t)d.num_spheres + 3) / 4 * 16 + (int64_t)d.num_planes * 80 +
                          (plain ? 0 : (int64_t)d.num_materials * 16 * RTP_LDS_MAT_ROWS) + 16 * rtk::kConstRows + (lanes / 64) * (8 + 32 * 4) + 7 * lanes * 4;
    const int64_t nodes = (budget - fixed) / rtk::kOctNodeBytes;
    return (int32_t)(nodes < 1 ? 1 : nodes);
}

rtaccel::PackOptions pack_options(const rt_config &cfg, const rt_scene_desc *d = nullptr) {
    rtaccel::PackOptions o;
    if (d && cfg.scene_in_lds != 0) o.lds_pair_budget = lds_pair_budget(*d);
    o.dynamic = cfg.guard_dynamic_margins;
    if (cfg.guard_gamma_ulps > 0.0f) o.gamma = (double)cfg.guard_gamma_ulps * 5.9604644775390625e-8;
    o.leaf_table = cfg.guard_exact_leaf_table != 0;
    // (the developer build's wavefront kernel arms its rays itself: everything stays in the tree)
    o.front_max = (cfg.guard_front_primitives < 0 || cfg.kernel == RT_KERNEL_WAVEFRONT) ? 0 : rtaccel::kMaxFront;
    return o;
}
uint32_t bail_share_of(const rt_config &cfg) {      // in 1/256ths; 0 = never
    if (cfg.guard_bail_share < 0) return 0u;
    const uint32_t s = cfg.guard_bail_share == 0 ? kDefaultBailShare : (uint32_t)cfg.guard_bail_share;
    return s > 255u ? 255u : s;
}
// … and what must be on the list before a pass is given up, in 1/1024ths of the pass (RTP_BAIL_FLOOR: for measurements)
uint32_t bail_floor() {
    static const uint32_t f = [] { const int v = env_int("RTP_BAIL_FLOOR", (int)kDefaultBailFloor); return (uint32_t)(v < 0 ? 0 : v > 255 ? 255 : v); }();
    return f;
}
uint32_t bail_latest() {          // … and until when, in eighths of the pass (RTP_BAIL_LATEST)
    static const uint32_t f = [] { const int v = env_int("RTP_BAIL_LATEST", (int)kDefaultBailLatest); return (uint32_t)(v < 1 ? 1 : v > 8 ? 8 : v); }();
    return f;
}
// rt_timing is an out-structure of the caller's size (include/rtp_amd.h)
rt_status timing_check(const rt_timing *t) {
    if (t && t->struct_bytes < 8) { g_last_error = "rt_timing.struct_bytes is not set (rt_timing_init)"; return RT_ERR_INVALID_ARG; }
    return RT_OK;
}
void timing_out(const rt_timing &src, rt_timing *dst) {
    if (!dst) return;
    const uint32_t n = dst->struct_bytes < sizeof(rt_timing) ? dst->struct_bytes : (uint32_t)sizeof(rt_timing);
    std::memcpy(dst, &src, n);
    dst->struct_bytes = n;
}
bool gamma_unproven(const rt_config &cfg) {
    return cfg.guard_gamma_ulps > 0.0f && (double)cfg.guard_gamma_ulps * 5.9604644775390625e-8 < (double)rtaccel::kGuardGammaBound;
}

}  // namespace

struct rt_scene {
    int device = 0;
    rt_config cfg{};
    float4 *tnodes = nullptr, *xnodes = nullptr;
    int32_t num_tnodes = 0, num_top = 0, num_top_pairs = 0;
    float4 *hnodes = nullptr;       // pair records with binary16 planes (guarded walk from global memory)
    float4 *wnodes = nullptr, *whnodes = nullptr;     // the same tree as 4-wide nodes (fp32 / binary16 boxes); null: pair nodes only
    int32_t num_wide = 0, num_top_wide = 0, wroot = rtk::kDone, wide_depth = 0;
    float4 *nodes = nullptr, *spheres = nullptr, *planes = nullptr, *materials = nullptr, *tex_data = nullptr;
    int32_t *sphere_mat = nullptr;
    int4 *tex_info = nullptr;
    uint32_t *queue = nullptr;      // counter block, see kQueue*
    float *slab = nullptr;          // per-sample radiance workspace of one pass (3 floats per sample), grown on demand
    size_t slab_floats = 0;
    rtaccel::Packed::Guard guard;   // guarded-walk eligibility and parameters
    // host copies of what the guard depends on: a camera outside the reach the margins were sized for makes
    // rt_render re-pack the guarded walk's tree for it (reach only ever grows)
    std::vector<rt_sphere> host_spheres;
    std::vector<rt_plane> host_planes;
    std::vector<rt_bvh_node> host_nodes;
    bool device_built = false;
    int repacks = 0;
    bool repack_refused = false;    // a re-pack for a far camera was not eligible: do not try again
    bool guard_paused = false;      // a frame abandoned a guarded pass or flagged more than the bail share of its samples: later frames use the exact walk
    // feedback of the last few render calls (kFeedbackSlots): [0, passes) flagged counts, [kMaxPasses, kMaxPasses + passes) abandon words
    struct Feedback {
        uint32_t *host = nullptr; hipEvent_t start = nullptr, done = nullptr;
        bool pending = false, guarded = false, exploring = false;
        int passes = 0; uint64_t samples = 0, serial = 0;
    };
    Feedback feedback[kFeedbackSlots];
    int feedback_next = 0;
    // RT_TRAVERSAL_AUTO's cost model is a measurement: a guarded frame that flagged more than kExploreShare of its samples makes the
    // handle render ONE frame with the exact walk and keep whichever was faster per sample (judge_frame)
    double guarded_ns_per_sample = 0.0, exact_ns_per_sample = 0.0;
    bool explore_exact = false;
    uint64_t frame_serial = 0;      // render calls so far (a feedback slot knows which one it belongs to)
    uint32_t trip_test = 0;         // developer build: rt_debug_trip_test
    const uint32_t *last_traced_pixels = nullptr;     // device word of the last call's primary-visibility pass (null: every pixel was traced)
    int32_t last_spp = 0;
    float4 *leaf_boxes = nullptr, *plane_leaf_boxes = nullptr;   // exact leaf boxes (final check of the guarded walk)
    uint32_t *flag_list = nullptr;  // work indices of flagged samples, grown on demand
    // resume table (rt_kernel.hip.inc): where a flagged sample's path stood when the flagged ray was armed — 2^18 tags + 64-byte states
    // (17 MB), made with the first guarded frame; a device short of memory renders without it (flagged samples restart from the camera)
    uint32_t *resume_tag = nullptr; float4 *resume_state = nullptr;
    size_t flag_cap = 0;
    // overlapped re-walk: the exact re-walk + the accumulation of the pixels it touches run on aux_stream beside the
    // accumulation of all other pixels on the caller's stream
    uint32_t *dirty = nullptr, *dirty_list = nullptr;      // per local pixel: mark, list of marked pixels
    size_t dirty_cap = 0;
    hipStream_t aux_stream = nullptr, list_stream = nullptr;      // (list_stream: the marked pixels are listed beside the re-walk)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_listed = nullptr;
    uint32_t *cand = nullptr;       // primary visibility: per local pixel rtk::kCandWords words (candidate leaves), grown on demand
    size_t cand_pixels = 0;
    // what the lists in `cand` (and the fetch order behind them) were made for: a call with the same view of the same tree on the
    // same stream — the next sample batch of a progressive render, the next frame of a still — reuses them (0.4 ms at 1080p)
    struct CandKey { float view[12]; int32_t dims[9]; int repacks; hipStream_t stream; bool valid = false; } cand_key;
    float4 *wf_pool = nullptr;      // render_kernel_wf: ray/hit stacks of every resident wave, grown on demand
    size_t wf_pool_float4s = 0;
    int32_t num_internal = 0, num_spheres = 0, num_planes = 0, num_materials = 0, root = rtk::kDone, tree_depth = 0;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    std::vector<hipEvent_t> pass_events;   // per pass: before the trace launch, after it, after the exact re-walk (first kTimedPasses passes)
    int timed_passes = 0;
    rt_timing last{};
    int last_passes = 0;
    uint64_t last_samples = 0;
    bool timed = false;
    int num_cus = 0;
    uint64_t device_bytes = 0;      // total memory of the device (workspace default: a sixteenth of it)
    float build_ms = 0.0f;          // device BVH build time (RTP_BUILD=device), else 0
    bool absorbing_glass = false;   // some DIELECTRIC material has a non-zero absorption (Beer-Lambert code needed)
};

namespace {

template <class T>
rt_status upload(const std::vector<T> &host, void **dev) {
    *dev = nullptr;
    if (host.empty()) return RT_OK;
    HIP_TRY(hipMalloc(dev, host.size() * sizeof(T)));
    HIP_TRY(hipMemcpy(*dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}

// Reciprocal for div_magic(): exact quotients for every n <= n_max (checked, not assumed).
bool make_magic(uint32_t d, uint64_t n_max, rtk::Magic &g) {
    if (d == 0) return false;
    if (d == 1) { g.m = 0; g.s = 0; return true; }
    uint32_t lg = 0;
    while ((2u << lg) <= d) ++lg;                     // floor(log2 d)
    g.s = 31 + lg;
    const uint64_t mm = ((uint64_t)1 << g.s) / d + 1;
    if (mm >> 32) return false;
    g.m = (uint32_t)mm;
    // n*m/2^s - n/d = n * (m*d - 2^s) / (d * 2^s) <= n * d / (d * 2^s): the floor is exact while n * d < 2^s
    return n_max * (uint64_t)d < ((uint64_t)1 << g.s);
}

void normalise_shard(const rt_shard *in, int32_t height, rt_shard &out) {
    if (!in || in->num_parts <= 1 || in->band_rows <= 0) {
        out.band_rows = height > 0 ? height : 1;
        out.num_parts = 1;
        out.part = 0;
    } else {
        out = *in;
    }
}

// A scene's tables live on the device it was created on: calls from a thread whose current device is another one
// would launch there with this device's pointers.
rt_status check_device(const rt_scene *sc) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return fail(RT_ERR_HIP, "hipGetDevice failed");
    if (cur != sc->device)
        return fail(RT_ERR_INVALID_ARG, "scene was created on device " + std::to_string(sc->device) + " but the calling thread's current device is " +
                                            std::to_string(cur) + " (call rt_set_device first)");
    return RT_OK;
}

struct Tile { int32_t x0, y0, w, h; };
rt_status fill_params(const rt_scene *sc, const rt_camera_data *cam, const rt_shard *shard, rtk::KParams &P, const Tile *tile = nullptr) {
    std::memset(&P, 0, sizeof(P));
    if (!sc || !cam) return fail(RT_ERR_INVALID_ARG, "null scene or camera");
    if (cam->image_width <= 0 || cam->image_height <= 0) return fail(RT_ERR_INVALID_ARG, "image size must be positive");
    // (tile offsets travel in 24 bits of the guarded kernels' constants block: fill_consts)
    if (cam->image_width >= (1 << 24) || cam->image_height >= (1 << 24)) return fail(RT_ERR_UNSUPPORTED, "image width or height of 2^24 or more");
    if (shard && shard->num_parts > 1 && (shard->part < 0 || shard->part >= shard->num_parts || shard->band_rows <= 0))
        return fail(RT_ERR_INVALID_ARG, "bad shard");
    std::memcpy(P.origin, cam->origin.e, 12);
    std::memcpy(P.p00, cam->pixel00_loc.e, 12);
    std::memcpy(P.du, cam->pixel_delta_u.e, 12);
    std::memcpy(P.dv, cam->pixel_delta_v.e, 12);
    std::memcpy(P.bg, cam->background.e, 12);
    P.width = cam->image_width;
    P.height = cam->image_height;
    P.spp = cam->samples_per_pixel;
    P.max_depth = cam->max_depth;
    rt_shard s;
    normalise_shard(shard, cam->image_height, s);
    P.band_rows = s.band_rows;
    P.num_parts = s.num_parts;
    P.part = s.part;
    P.local_rows = rt_shard_rows(cam->image_height, shard);
    P.row_w = P.width;
    if (tile) {             // a rectangle of the image instead of row bands
        if (shard && shard->num_parts > 1) return fail(RT_ERR_INVALID_ARG, "a tile and a shard in one call");
        if (tile->w <= 0 || tile->h <= 0 || tile->x0 < 0 || tile->y0 < 0 || (int64_t)tile->x0 + tile->w > cam->image_width ||
            (int64_t)tile->y0 + tile->h > cam->image_height)
            return fail(RT_ERR_INVALID_ARG, "tile outside the image");
        P.local_rows = tile->h;
        P.row_w = tile->w;
        P.tile_x0 = tile->x0;
        P.tile_y0 = tile->y0;
    }
    P.nodes = sc->nodes; P.hnodes = sc->hnodes; P.num_internal = sc->num_internal; P.root = sc->root;
    P.wnodes = sc->wnodes; P.whnodes = sc->whnodes; P.num_wide = sc->num_wide; P.wroot = sc->wroot;
    P.tnodes = sc->tnodes; P.num_tnodes = sc->num_tnodes;
    P.xnodes = sc->xnodes; P.num_top = sc->num_top;
    P.spheres = sc->spheres; P.num_spheres = sc->num_spheres;
    P.planes = sc->planes; P.num_planes = sc->num_planes;
    P.materials = sc->materials; P.num_materials = sc->num_materials;
    P.sphere_mat = sc->sphere_mat;
    P.tex_data = sc->tex_data; P.tex_info = sc->tex_info;
    P.queue = sc->queue;
    if ((uint64_t)P.local_rows * (uint64_t)P.row_w > (1u << 24)) return fail(RT_ERR_UNSUPPORTED, "more than 2^24 pixels per call");
    P.total_work = 0;      // set per pass in rt_render
    P.stats = sc->queue + kQueueStats;
    P.trip_test = sc->trip_test;
    {
        const uint64_t pixels = (uint64_t)P.local_rows * (uint64_t)P.row_w;
        if (!make_magic((uint32_t)P.row_w, pixels + 1, P.magic_width) ||
            !make_magic((uint32_t)P.band_rows, (uint64_t)P.local_rows + 1, P.magic_band))
            return fail(RT_ERR_UNSUPPORTED, "image too large for the work index arithmetic");
    }
    P.stack_levels = 0;
    P.leaf_boxes = sc->leaf_boxes;
    P.plane_leaf_boxes = sc->plane_leaf_boxes;
    std::memcpy(P.g_center, sc->guard.center, 12);
    P.g_d0sq = sc->guard.d0_sq;
    P.g_rs = sc->guard.cluster_radius;
    P.g_fark = sc->guard.far_k;
    P.g_dynk = sc->guard.dyn_k;
    {   // step_pair_par: sqrt(k) with the walk's slack, and sqrt(k) sqrt(3) r_max — both rounded up
        // (sqrt(k) also multiplies the kernel's APPROXIMATE sqrt of |d|^2 — one ulp — and rounds in a float product: 4e-6 on top)
        const double sk = std::sqrt((double)sc->guard.dyn_k * (double)rtk::kDynSlack) * (1.0 + 4e-6);
        P.g_dyn_sqrtk = std::nextafterf((float)sk, INFINITY);
        P.g_dyn_b = std::nextafterf((float)(sk * 1.7320508075688772 * (double)sc->guard.dyn_rmax * (1.0 + 1e-6)), INFINITY);
        P.g_dyn_c3 = std::nextafterf((float)(sk * 1.7320508075688772), INFINITY);
        P.g_dyn_kslack = std::nextafterf((float)((double)sc->guard.dyn_k * (double)rtk::kDynSlack), INFINITY);
    }
    P.num_front = sc->guard.num_front;
    std::memcpy(P.front_code, sc->guard.front_code, sizeof(P.front_code));
    std::memcpy(P.front_box, sc->guard.front_box, sizeof(P.front_box));
    std::memcpy(P.g_box, sc->guard.box, 24);
    P.k_inner = sc->cfg.k_inner > 0 ? sc->cfg.k_inner : 24;
    P.k_shade = sc->cfg.k_shade > 0 ? sc->cfg.k_shade : 48;
    P.chunk = 64u;      // set per pass in rt_render
    return RT_OK;
}

// Re-pack the guarded walk's tree with margins that cover ray origins at `cam` (see rt_render).
rt_status repack_for_camera(rt_scene *sc, const float cam[3], hipStream_t stream) {
    rt_scene_desc d{};
    d.spheres = sc->host_spheres.data(); d.num_spheres = (int32_t)sc->host_spheres.size();
    d.planes = sc->host_planes.data(); d.num_planes = (int32_t)sc->host_planes.size();
    d.nodes = sc->host_nodes.data(); d.num_nodes = (int32_t)sc->host_nodes.size();
    // materials and textures do not enter the tree: one dummy material satisfies the index validation
    std::vector<rt_sphere> spheres(sc->host_spheres);
    std::vector<rt_plane> planes(sc->host_planes);
    for (rt_sphere &s : spheres) s.material_idx = 0;
    for (rt_plane &p : planes) p.material_idx = 0;
    rt_material dummy{};
    d.spheres = spheres.data();
    d.planes = planes.data();
    d.materials = &dummy; d.num_materials = 1;
    rtaccel::Packed pk;
    const std::string err = rtaccel::pack_scene(d, rtaccel::TreeMode::Guarded, pk, pack_options(sc->cfg, &d), cam);
    if (!err.empty()) return fail(RT_ERR_INVALID_ARG, "re-pack for a far camera: " + err);
    if (!pk.guard.ok) {          // margins for that distance would swallow the tree: such cameras get the exact walk
        sc->repack_refused = true;
        return RT_OK;
    }
    HIP_TRY(hipStreamSynchronize(stream));          // the old tables may still be in use on this stream
    float4 *nodes = nullptr, *hnodes = nullptr, *wnodes = nullptr, *whnodes = nullptr, *leaf_boxes = nullptr, *plane_leaf_boxes = nullptr;
    rt_status st = RT_OK;
    if ((st = upload(pk.nodes, (void **)&nodes)) != RT_OK || (st = upload(pk.hnodes, (void **)&hnodes)) != RT_OK ||
        (st = upload(pk.wnodes, (void **)&wnodes)) != RT_OK || (st = upload(pk.whnodes, (void **)&whnodes)) != RT_OK ||
        (st = upload(pk.leaf_boxes, (void **)&leaf_boxes)) != RT_OK || (st = upload(pk.plane_leaf_boxes, (void **)&plane_leaf_boxes)) != RT_OK) {
        (void)hipFree(nodes); (void)hipFree(hnodes); (void)hipFree(wnodes); (void)hipFree(whnodes); (void)hipFree(leaf_boxes); (void)hipFree(plane_leaf_boxes);
        return st;
    }
    (void)hipFree(sc->nodes); (void)hipFree(sc->hnodes); (void)hipFree(sc->wnodes); (void)hipFree(sc->whnodes);
    (void)hipFree(sc->leaf_boxes); (void)hipFree(sc->plane_leaf_boxes);
    sc->nodes = nodes; sc->hnodes = hnodes; sc->wnodes = wnodes; sc->whnodes = whnodes; sc->leaf_boxes = leaf_boxes; sc->plane_leaf_boxes = plane_leaf_boxes;
    sc->num_wide = pk.num_wide; sc->num_top_wide = pk.num_top_wide; sc->wroot = pk.wroot; sc->wide_depth = pk.wide_depth;
    sc->num_internal = pk.num_internal;
    sc->num_top_pairs = pk.num_top_pairs;
    sc->root = pk.root;
    sc->tree_depth = pk.max_depth;
    sc->guard = pk.guard;
    sc->repacks++;
    return RT_OK;
}

}  // namespace

// for rt_multi.hip (same library, not exported)
__attribute__((visibility("hidden"))) void rt_internal_set_error(const std::string &msg) { g_last_error = msg; }

extern "C" {

const char *rt_get_last_error_string(void) { return g_last_error.c_str(); }

// parity: the arithmetic flags the bit-for-bit claims rest on (no contraction, IEEE divide / sqrt, denormals kept, no fast-math) —
// the Makefile passes -DRTP_PARITY_FLAGS=1 together with them, and only with them; dev: the developer build (see above)
#if defined(RTP_PARITY_FLAGS) && !defined(__FAST_MATH__)
#define RTP_VERSION_PARITY "1"
#else
#define RTP_VERSION_PARITY "0"
#endif
#ifdef RTP_DEV_BUILD
#define RTP_VERSION_DEV "1"
#else
#define RTP_VERSION_DEV "0"
#endif
const char *rt_version_string(void) { return "rtp_amd 0.4 gfx950 parity=" RTP_VERSION_PARITY " dev=" RTP_VERSION_DEV; }

rt_status rt_set_device(int32_t device_ordinal) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device");
    if (device_ordinal < 0 || device_ordinal >= n) return fail(RT_ERR_NO_DEVICE, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device_ordinal));
    return RT_OK;
}

// (the header's macro of the same name calls rt_config_init_sized with the caller's sizeof; this is the symbol FFI bindings reach)
void (rt_config_init)(rt_config *cfg) {
    if (cfg) config_defaults(*cfg);
}
void rt_config_init_sized(rt_config *cfg, uint32_t struct_bytes) {
    if (!cfg || struct_bytes < 8) return;
    rt_config c;
    config_defaults(c);
    const uint32_t n = struct_bytes < sizeof(rt_config) ? struct_bytes : (uint32_t)sizeof(rt_config);
    std::memcpy(cfg, &c, n);
    cfg->struct_bytes = n;
}

void rt_config_from_env(rt_config *user) {
    if (!user || user->struct_bytes < 8) return;
    // (on a copy of the library's size: a caller compiled against a shorter rt_config gets back only the fields it has)
    rt_config full = config_from_caller(user);
    rt_config *cfg = &full;
    auto str_is = [](const char *name, const char *value) { const char *v = getenv(name); return v && std::string(v) == value; };
    if (str_is("RTP_TRAVERSAL", "threaded") || str_is("RTP_TRAVERSAL", "exact")) cfg->traversal = RT_TRAVERSAL_EXACT;
    if (str_is("RTP_TRAVERSAL", "guarded")) cfg->traversal = RT_TRAVERSAL_GUARDED;
    if (str_is("RTP_BUILD", "device")) cfg->tree_build = RT_BUILD_DEVICE_LBVH;
    if (str_is("RTP_KERNEL", "mega")) cfg->kernel = RT_KERNEL_MEGA;
    if (str_is("RTP_KERNEL", "wavefront")) cfg->kernel = RT_KERNEL_WAVEFRONT;
    if (const char *v = getenv("RTP_GUARD_GAMMA_ULPS")) cfg->guard_gamma_ulps = (float)atof(v);
    cfg->guard_exact_leaf_table = env_int("RTP_GUARD_TABLE", cfg->guard_exact_leaf_table);
    cfg->guard_dynamic_margins = env_int("RTP_GUARD_DYNAMIC", cfg->guard_dynamic_margins);
    cfg->guard_min_primitives = env_int("RTP_GUARD_MIN_PRIMS", cfg->guard_min_primitives);
    cfg->guard_keep = env_int("RTP_GUARD_KEEP", cfg->guard_keep);
    if (env_int("RTP_NO_REPACK", 0)) cfg->guard_repack = 0;
    if (const char *v = getenv("RTP_SLAB_GIB")) { const double g = atof(v); if (g > 0) cfg->workspace_bytes = (uint64_t)(g * 1073741824.0); }
    cfg->pass_spp = env_int("RTP_PASS_SPP", cfg->pass_spp);
    cfg->stack_levels = env_int("RTP_STACK_LEVELS", cfg->stack_levels);
    cfg->flag_capacity = (uint32_t)env_int("RTP_FLAG_CAP", (int)cfg->flag_capacity);
    if (env_int("RTP_NO_LDS_SCENE", 0)) cfg->scene_in_lds = 0;
    if (env_int("RTP_NO_TREELET", 0)) cfg->lds_treelet = 0;
    cfg->workgroups_per_cu = env_int("RTP_WGS_PER_CU", cfg->workgroups_per_cu);
    cfg->k_inner = env_int("RTP_K_INNER", cfg->k_inner);
    cfg->k_shade = env_int("RTP_K_SHADE", cfg->k_shade);
    cfg->reserve_chunk = env_int("RTP_CHUNK", cfg->reserve_chunk);
    cfg->wavefront_paths = env_int("RTP_WF_PATHS", cfg->wavefront_paths);
    cfg->wavefront_exchange = env_int("RTP_WF_EXCHANGE", cfg->wavefront_exchange);
    if (env_int("RTP_NO_TAPER", 0)) cfg->reserve_taper = 0;
    cfg->wide_nodes = env_int("RTP_WIDE", cfg->wide_nodes);
    if (env_int("RTP_NO_SIMPLE", 0)) cfg->sphere_only_kernel = -1;
    if (env_int("RTP_NO_OVERLAP", 0)) cfg->overlap_rework = -1;
    if (env_int("RTP_NO_PRIMARY", 0)) cfg->primary_visibility = -1;
    cfg->guard_bail_share = env_int("RTP_BAIL_SHARE", cfg->guard_bail_share);
    if (env_int("RTP_NO_FRONT", 0)) cfg->guard_front_primitives = -1;
    if (env_int("RTP_NO_VIEW_CACHE", 0)) cfg->reuse_view_lists = -1;
    if (env_int("RTP_NO_RESUME", 0)) cfg->resume_flagged = -1;
    const uint32_t n = user->struct_bytes < sizeof(rt_config) ? user->struct_bytes : (uint32_t)sizeof(rt_config);
    std::memcpy(user, &full, n);
    user->struct_bytes = n;
}

rt_status rt_scene_create(const rt_scene_desc *desc, rt_scene **out_scene) { return rt_scene_create_ex(desc, nullptr, out_scene); }

rt_status rt_scene_set_config(rt_scene *sc, const rt_config *cfg) {
    if (!sc || !cfg) return fail(RT_ERR_INVALID_ARG, "null argument");
    rt_config c = config_from_caller(cfg);
    // create-time fields keep the values the tables were built with
    c.tree_build = sc->cfg.tree_build;
    c.guard_gamma_ulps = sc->cfg.guard_gamma_ulps;
    c.guard_exact_leaf_table = sc->cfg.guard_exact_leaf_table;
    c.guard_dynamic_margins = sc->cfg.guard_dynamic_margins;
    c.guard_front_primitives = sc->cfg.guard_front_primitives;
    sc->cfg = c;
    return RT_OK;
}

rt_status rt_scene_get_config(const rt_scene *sc, rt_config *cfg) {
    if (!sc || !cfg) return fail(RT_ERR_INVALID_ARG, "null argument");
    if (cfg->struct_bytes < 8) return fail(RT_ERR_INVALID_ARG, "rt_config.struct_bytes is not set (rt_config_init)");
    const uint32_t n = cfg->struct_bytes < sizeof(rt_config) ? cfg->struct_bytes : (uint32_t)sizeof(rt_config);
    std::memcpy(cfg, &sc->cfg, n);
    cfg->struct_bytes = n;
    return RT_OK;
}

rt_status rt_scene_create_ex(const rt_scene_desc *desc, const rt_config *user_cfg, rt_scene **out_scene) {
    if (!desc || !out_scene) return fail(RT_ERR_INVALID_ARG, "null argument");
    *out_scene = nullptr;
    const rt_config cfg = config_from_caller(user_cfg);
    if (cfg.guard_gamma_ulps < 0.0f || !(cfg.guard_gamma_ulps == cfg.guard_gamma_ulps)) return fail(RT_ERR_INVALID_ARG, "guard_gamma_ulps must be >= 0");
    rtaccel::Packed pk;
    const rtaccel::PackOptions popt = pack_options(cfg, desc);
    // RT_BUILD_DEVICE_LBVH: the guarded walk's tree is built on the GPU (LBVH, rt_build.hip) instead of the host's SAH
    // builder — any tree over the inflated leaves gives the same image (docs/LOG.md §3b)
    const bool device_build = cfg.tree_build == RT_BUILD_DEVICE_LBVH;
    std::string err = rtaccel::pack_scene(*desc, device_build ? rtaccel::TreeMode::GuardedLeaves : rtaccel::TreeMode::Guarded, pk, popt);
    if (!err.empty()) return fail(RT_ERR_INVALID_ARG, err);
    if (device_build && !pk.guard.ok) {       // not eligible for the guarded walk: nothing to build, exact walk only
        err = rtaccel::pack_scene(*desc, rtaccel::TreeMode::Guarded, pk, popt);
        if (!err.empty()) return fail(RT_ERR_INVALID_ARG, err);
    }

    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(RT_ERR_NO_DEVICE, "no HIP device");
    rt_scene *sc = new (std::nothrow) rt_scene;
    if (!sc) return fail(RT_ERR_OUT_OF_MEMORY, "host allocation failed");
    sc->cfg = cfg;
    rt_status st = RT_OK;
    auto bail = [&](rt_status s) { rt_scene_destroy(sc); return s; };
    if (hipGetDevice(&sc->device) != hipSuccess) return bail(fail(RT_ERR_HIP, "hipGetDevice failed"));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, sc->device) != hipSuccess) return bail(fail(RT_ERR_HIP, "hipGetDeviceProperties failed"));
    sc->num_cus = prop.multiProcessorCount;
    sc->device_bytes = (uint64_t)prop.totalGlobalMem;
    if (device_build && pk.guard.ok) {
        rtbuild::DeviceTree tree;
        const std::string berr = rtbuild::build_lbvh(pk.guard_leaf_boxes.data(), pk.guard_leaf_codes.data(), (int32_t)pk.guard_leaf_codes.size(), tree);
        if (!berr.empty()) return bail(fail(RT_ERR_HIP, "device BVH build: " + berr));
        sc->nodes = (float4 *)tree.nodes;
        sc->hnodes = (float4 *)tree.hnodes;
        pk.num_internal = tree.num_internal;
        pk.root = tree.root;
        pk.max_depth = tree.depth;
        pk.num_top_pairs = 0;
        sc->build_ms = tree.build_ms;
    } else {
        if ((st = upload(pk.nodes, (void **)&sc->nodes)) != RT_OK) return bail(st);
        if ((st = upload(pk.hnodes, (void **)&sc->hnodes)) != RT_OK) return bail(st);
        if ((st = upload(pk.wnodes, (void **)&sc->wnodes)) != RT_OK) return bail(st);
        if ((st = upload(pk.whnodes, (void **)&sc->whnodes)) != RT_OK) return bail(st);
        sc->num_wide = pk.num_wide; sc->num_top_wide = pk.num_top_wide; sc->wroot = pk.wroot; sc->wide_depth = pk.wide_depth;
    }
    if ((st = upload(pk.tnodes, (void **)&sc->tnodes)) != RT_OK) return bail(st);
    sc->num_tnodes = pk.num_tnodes;
    if ((st = upload(pk.xnodes, (void **)&sc->xnodes)) != RT_OK) return bail(st);
    sc->num_top = pk.num_top;
    if ((st = upload(pk.spheres, (void **)&sc->spheres)) != RT_OK) return bail(st);
    if ((st = upload(pk.planes, (void **)&sc->planes)) != RT_OK) return bail(st);
    if ((st = upload(pk.materials, (void **)&sc->materials)) != RT_OK) return bail(st);
    if ((st = upload(pk.sphere_mat, (void **)&sc->sphere_mat)) != RT_OK) return bail(st);
    if ((st = upload(pk.tex_data, (void **)&sc->tex_data)) != RT_OK) return bail(st);
    if ((st = upload(pk.tex_info, (void **)&sc->tex_info)) != RT_OK) return bail(st);
    if ((st = upload(pk.leaf_boxes, (void **)&sc->leaf_boxes)) != RT_OK) return bail(st);
    if ((st = upload(pk.plane_leaf_boxes, (void **)&sc->plane_leaf_boxes)) != RT_OK) return bail(st);
    sc->guard = pk.guard;
    for (int32_t m = 0; m < desc->num_materials; ++m) {
        const rt_material &mat = desc->materials[m];
        if (mat.type == RT_MAT_DIELECTRIC && !(mat.absorption.e[0] == 0.0f && mat.absorption.e[1] == 0.0f && mat.absorption.e[2] == 0.0f)) sc->absorbing_glass = true;
    }
    if (pk.guard.ok) {
        sc->host_spheres.assign(desc->spheres, desc->spheres + desc->num_spheres);
        sc->host_planes.assign(desc->planes, desc->planes + desc->num_planes);
        sc->host_nodes.assign(desc->nodes, desc->nodes + desc->num_nodes);
        sc->device_built = device_build;
    }
    if (hipMalloc((void **)&sc->queue, kQueueWords * 4) != hipSuccess) return bail(fail(RT_ERR_OUT_OF_MEMORY, "hipMalloc(queue) failed"));
    if (hipEventCreate(&sc->ev_start) != hipSuccess || hipEventCreate(&sc->ev_stop) != hipSuccess)
        return bail(fail(RT_ERR_HIP, "hipEventCreate failed"));
    sc->num_internal = pk.num_internal;
    sc->num_top_pairs = pk.num_top_pairs;
    sc->num_spheres = (int32_t)pk.sphere_mat.size();
    sc->num_planes = (int32_t)(pk.planes.size() / 20);
    sc->num_materials = (int32_t)(pk.materials.size() / 12);
    sc->root = pk.root;
    sc->tree_depth = pk.max_depth;
    *out_scene = sc;
    return RT_OK;
}

rt_status rt_scene_destroy(rt_scene *sc) {
    if (!sc) return RT_OK;
    (void)hipFree(sc->tnodes);
    (void)hipFree(sc->xnodes);
    (void)hipFree(sc->nodes); (void)hipFree(sc->hnodes); (void)hipFree(sc->wnodes); (void)hipFree(sc->whnodes);
    (void)hipFree(sc->spheres); (void)hipFree(sc->planes); (void)hipFree(sc->materials);
    (void)hipFree(sc->sphere_mat); (void)hipFree(sc->tex_data); (void)hipFree(sc->tex_info); (void)hipFree(sc->queue); (void)hipFree(sc->slab);
    (void)hipFree(sc->leaf_boxes); (void)hipFree(sc->plane_leaf_boxes); (void)hipFree(sc->flag_list); (void)hipFree(sc->wf_pool);
    (void)hipFree(sc->dirty); (void)hipFree(sc->dirty_list); (void)hipFree(sc->cand);
    (void)hipFree(sc->resume_tag); (void)hipFree(sc->resume_state);
    if (sc->aux_stream) (void)hipStreamDestroy(sc->aux_stream);
    if (sc->list_stream) (void)hipStreamDestroy(sc->list_stream);
    if (sc->ev_listed) (void)hipEventDestroy(sc->ev_listed);
    if (sc->ev_fork) (void)hipEventDestroy(sc->ev_fork);
    if (sc->ev_join) (void)hipEventDestroy(sc->ev_join);
    for (hipEvent_t e : sc->pass_events) (void)hipEventDestroy(e);
    if (sc->ev_start) (void)hipEventDestroy(sc->ev_start);
    if (sc->ev_stop) (void)hipEventDestroy(sc->ev_stop);
    for (rt_scene::Feedback &f : sc->feedback) {
        if (f.done) { if (f.pending) (void)hipEventSynchronize(f.done); (void)hipEventDestroy(f.done); }
        if (f.start) (void)hipEventDestroy(f.start);
        if (f.host) (void)hipHostFree(f.host);
    }
    delete sc;
    return RT_OK;
}

const char *rt_scene_guard_reason(const rt_scene *sc) {
    if (!sc) return "null scene";
    return sc->guard.ok ? "" : sc->guard.reason.c_str();
}

int32_t rt_shard_rows(int32_t image_height, const rt_shard *shard) {
    if (image_height <= 0) return 0;
    if (!shard || shard->num_parts <= 1 || shard->band_rows <= 0) return image_height;
    const int64_t band = shard->band_rows, parts = shard->num_parts, part = shard->part;
    if (part < 0 || part >= parts) return 0;
    const int64_t cycle = band * parts;
    const int64_t full = image_height / cycle, rem = image_height % cycle;
    int64_t rows = full * band;
    const int64_t start = part * band;
    if (rem > start) rows += (rem - start < band) ? rem - start : band;
    return (int32_t)rows;
}

}  // extern "C"

namespace {
// The handle's own judgement of its guarded walk, from what a finished render call left behind (its flagged counts, abandon words
// and event times): a pass that gave up, or more flagged samples overall than the bail share, and the following frames go to the
// exact walk; a frame that flagged more than kExploreShare makes the next AUTO frame an exact one, and the faster of the two per
// sample stays (rt_config.guard_keep: the guarded walk stays whatever happens).
// trace / re-walk / primary-pass time of the handle's most recent frame, from its per-pass events (the frame must be done)
rt_status frame_parts(rt_scene *sc, float &trace, float &rework, float &primary) {
    trace = rework = primary = 0.0f;
    for (int p = 0; p < sc->timed_passes; ++p) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, sc->pass_events[4 * p], sc->pass_events[4 * p + 1]));
        primary += ms;
        HIP_TRY(hipEventElapsedTime(&ms, sc->pass_events[4 * p + 1], sc->pass_events[4 * p + 2]));
        trace += ms;
        HIP_TRY(hipEventElapsedTime(&ms, sc->pass_events[4 * p + 2], sc->pass_events[4 * p + 3]));
        rework += ms;
    }
    // passes beyond the individually timed ones are priced at the mean of the timed ones
    if (sc->timed_passes > 0) {
        const float scale = (float)sc->last.trace_launches / (float)sc->timed_passes;
        trace *= scale; rework *= scale; primary *= scale;
        float ms = 0.0f;          // + the per-pixel candidate lists, made once per call before the first pass
        HIP_TRY(hipEventElapsedTime(&ms, sc->ev_start, sc->pass_events[0]));
        primary += ms;
    }
    return RT_OK;
}
rt_status judge_frame(rt_scene *sc, rt_scene::Feedback &f) {
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, f.start, f.done));
    const double ns = f.samples ? (double)ms * 1e6 / (double)f.samples : 0.0;
    const uint32_t share = bail_share_of(sc->cfg);
    if (f.guarded) {
        uint64_t total = 0;
        uint32_t gave_up = 0;
        // (list slots handed out less the ones nobody filled: rt_kernel.hip.inc, flag_chunk_drain)
        for (int p = 0; p < f.passes; ++p) { total += f.host[p] - f.host[2 * kMaxPasses + p]; gave_up += f.host[kMaxPasses + p] != 0u ? 1u : 0u; }
        if (share != 0u && (gave_up != 0u || total * 256u > (uint64_t)share * f.samples)) sc->guard_paused = true;
        if (gave_up == 0u) sc->guarded_ns_per_sample = ns;
        // worth a measurement?  many flagged samples, or much time spent on what the exact walk does not need (the primary pass,
        // the re-walk launch) — the guarded trace launch would have to be faster by that much just to draw level
        bool doubt = total * 256u > (uint64_t)kExploreShare * f.samples;
        if (!doubt && f.serial == sc->frame_serial && sc->timed) {       // (the per-pass events are this frame's: nothing was enqueued after it)
            float trace = 0.0f, rework = 0.0f, primary = 0.0f;
            if (const rt_status st = frame_parts(sc, trace, rework, primary)) return st;
            doubt = rework + primary > kExploreOverhead * ms;
        }
        if (share != 0u && !sc->guard_paused && sc->exact_ns_per_sample == 0.0 && doubt) sc->explore_exact = true;
    } else if (f.exploring) {
        sc->exact_ns_per_sample = ns;
        sc->explore_exact = false;
        if (sc->guarded_ns_per_sample > 0.0 && ns < 0.95 * sc->guarded_ns_per_sample) sc->guard_paused = true;
    }
    return RT_OK;
}
// feedback slots whose frames have finished are read (wait_all: every pending one is waited for)
rt_status poll_feedback(rt_scene *sc, bool wait_all) {
    for (int k = 0; k < kFeedbackSlots; ++k) {
        rt_scene::Feedback &f = sc->feedback[(sc->feedback_next + k) % kFeedbackSlots];      // oldest first
        if (!f.pending) continue;
        if (wait_all) HIP_TRY(hipEventSynchronize(f.done));
        const hipError_t e = hipEventQuery(f.done);
        if (e == hipErrorNotReady) continue;
        HIP_TRY(e);
        f.pending = false;
        if (const rt_status st = judge_frame(sc, f)) return st;
    }
    return RT_OK;
}
// the slot of the frame about to be enqueued
rt_status acquire_feedback(rt_scene *sc, rt_scene::Feedback **out) {
    rt_scene::Feedback &f = sc->feedback[sc->feedback_next];
    sc->feedback_next = (sc->feedback_next + 1) % kFeedbackSlots;
    if (f.pending) {                   // the caller is kFeedbackSlots frames ahead of the device: wait for the oldest
        HIP_TRY(hipEventSynchronize(f.done));
        f.pending = false;
        if (const rt_status st = judge_frame(sc, f)) return st;
    }
    if (!f.host) {
        HIP_TRY(hipHostMalloc((void **)&f.host, 3 * kMaxPasses * sizeof(uint32_t), hipHostMallocDefault));
        HIP_TRY(hipEventCreate(&f.start));
        HIP_TRY(hipEventCreate(&f.done));
    }
    *out = &f;
    return RT_OK;
}
// rt_render and rt_render_tile: whole rows of a shard, or a rectangle
rt_status render_impl(rt_scene *sc, const rt_camera_data *cam, const rt_shard *shard, const Tile *tile, float *d_fb_sum, void *hip_stream,
                      int32_t sync, rt_timing *timing) {
    rtk::KParams P;
    rt_status st = fill_params(sc, cam, shard, P, tile);
    if (st != RT_OK) return st;
    if ((st = check_device(sc)) != RT_OK) return st;
    if (!d_fb_sum) return fail(RT_ERR_INVALID_ARG, "null framebuffer");
    if ((st = timing_check(timing)) != RT_OK) return st;
    const rt_config &cfg = sc->cfg;
    hipStream_t stream = (hipStream_t)hip_stream;
    P.fb = d_fb_sum;
    timing_out(rt_timing{}, timing);
    // what earlier frames of this handle reported about their guarded walk (no wait: whatever has landed by now)
    if ((st = poll_feedback(sc, false)) != RT_OK) return st;
    const size_t fb_bytes = (size_t)P.local_rows * P.row_w * 3 * sizeof(float);
    if (P.local_rows == 0) return RT_OK;
    if (P.spp <= 0 || P.max_depth <= 0) {     // the reference's loops add nothing: all-zero sums
        HIP_TRY(hipMemsetAsync(d_fb_sum, 0, fb_bytes, stream));
        if (sync) HIP_TRY(hipStreamSynchronize(stream));
        sc->timed = false;
        return RT_OK;
    }

    const uint32_t waves = rtk::kBlock / rtk::kWave;
    const uint32_t pool_bytes = waves * 8u;                  // per-wave reserved work range
    const uint64_t prim_f4 = (uint64_t)P.num_spheres + (uint64_t)P.num_planes * 5 + (uint64_t)P.num_materials * RTP_LDS_MAT_ROWS +      // LDS keeps the first RTP_LDS_MAT_ROWS of the 3 material rows
                             ((uint64_t)P.num_spheres + 3) / 4;

    // ---- exact (threaded) walk: launch shape
    struct Shape { bool in_lds; uint32_t lds_bytes; int wgs_per_cu; int32_t stack_levels; int32_t num_top; };
    Shape exact{};
    {
        const uint64_t scene_bytes = (((uint64_t)P.num_tnodes + 1) * 2 + prim_f4) * 16;
        exact.in_lds = cfg.scene_in_lds && scene_bytes + pool_bytes <= kLdsLimit;
        // big scene: the top of the tree (explicit-link records [0, num_top)) is staged in LDS
        exact.num_top = (exact.in_lds || !cfg.lds_treelet) ? 0 : P.num_top;
        const uint64_t per_wg = exact.in_lds ? scene_bytes + pool_bytes : pool_bytes + (uint64_t)exact.num_top * 32u;
        // workgroups per CU: what the register budget admits (RTP_MIN_WAVES waves per SIMD), unless that
        // many LDS-resident scene copies do not fit next to each other
        exact.wgs_per_cu = cfg.workgroups_per_cu;
        if (exact.wgs_per_cu <= 0) {
            exact.wgs_per_cu = RTP_MIN_WAVES * 256 / rtk::kBlock;
            while (exact.wgs_per_cu > 1 && (uint64_t)exact.wgs_per_cu * per_wg > kLdsLimit) --exact.wgs_per_cu;
        }
        if (exact.wgs_per_cu < 1) exact.wgs_per_cu = 1;
        exact.lds_bytes = (uint32_t)per_wg;
    }
    // The sphere-only build of the exact walk (render_kernel<true, true, …, kSimple>: no plane, texture or absorption code, material
    // rows from global memory, node records in the 64-byte octant layout of step_threaded_oct; 64 registers, 1024-thread
    // workgroups, 8 waves per SIMD) for WHOLE passes of scenes that have none of those: S-rtiow 5 210 against 4 855 Msamples/s.
    // The re-walk of a list keeps the general build: it is a few long paths, and those run slower in the tighter kernel
    // (headline frame: re-walk 3.1 ms instead of 1.7).
    Shape exact_s{};
    bool exact_simple = exact.in_lds && cfg.sphere_only_kernel >= 0 && P.num_planes == 0 && sc->tex_data == nullptr && !sc->absorbing_glass &&
                        cfg.workgroups_per_cu == 0 && P.num_spheres > 0;
    if (exact_simple) {
        const uint64_t simple_bytes = (((uint64_t)P.num_tnodes + 1) * 4 + (uint64_t)P.num_spheres + ((uint64_t)P.num_spheres + 3) / 4) * 16;
        const uint64_t simple_pool = (uint64_t)(rtk::kSimpleBlock / rtk::kWave) * 8u + 16u * rtk::kConstRows;      // work ranges + the constants block
        exact_s.in_lds = true;
        exact_s.wgs_per_cu = rtk::kSimpleWaves * 256 / rtk::kSimpleBlock;
        exact_s.lds_bytes = (uint32_t)(simple_bytes + simple_pool);
        if ((uint64_t)exact_s.wgs_per_cu * exact_s.lds_bytes > kLdsLimit) exact_simple = false;
    }

    // ---- guarded near-first walk: only for eligible scenes that fit LDS with a useful stack
    bool guarded = sc->guard.ok && cfg.traversal != RT_TRAVERSAL_EXACT && P.root >= 0 && guarded_wanted(cfg, (int64_t)P.num_spheres + P.num_planes) &&
                   !(sc->guard_paused && !cfg.guard_keep);
    // (AUTO only: one exact frame to time the guarded walk against — judge_frame)
    const bool exploring = guarded && sc->explore_exact && cfg.traversal == RT_TRAVERSAL_AUTO && !cfg.guard_keep;
    if (exploring) guarded = false;
    Shape fast{};
    // kernel form of the guarded pass: render_kernel (a lane owns a path) or render_kernel_wf (a wave owns a pool of paths)
    const bool want_wavefront = cfg.kernel == RT_KERNEL_WAVEFRONT;
#ifndef RTP_DEV_BUILD
    if (want_wavefront) return fail(RT_ERR_UNSUPPORTED, "RT_KERNEL_WAVEFRONT is an experiment of the developer build (make dev): not in this library");
#endif
    // 4-wide nodes (where the scene has them: a host-built tree with at least one inner node).  Scenes with distance-aware margins
    // walk them by default (step_wide_par: half the dependent record loads of the pair walk and, with the growth in parametric
    // form, fewer instructions per ray — configs[4] +1.7 %).  For every other walk they stay an experiment of the developer build
    // (rt_config.wide_nodes = 1: measured 6 % slower on S-rtiow; docs/LOG.md); the wavefront kernel walks pairs
#ifndef RTP_DEV_BUILD
    if (cfg.wide_nodes > 0) return fail(RT_ERR_UNSUPPORTED, "rt_config.wide_nodes = 1 is an experiment of the developer build (make dev): not in this library");
#endif
    // (it arms its rays itself: a tree without its front primitives — rt_accel.h — is not its to walk)
    if (want_wavefront && sc->guard.num_front > 0)
        return fail(RT_ERR_UNSUPPORTED, "the wavefront kernel needs a scene handle created with it selected (its tree must hold every primitive)");
    // (a function of the handle's tables: evaluated again after a re-pack for a far camera — which may turn distance-aware margins
    // on or off — below)
    auto wide_nodes_wanted = [&]() {
        // (step_wide_par tells an empty child slot by its box — the finite inverted (65504, -65504) of the binary16 table — which holds
        // while the growth stays below 65504: it never exceeds dyn_k x (twice the radius every ray origin lies within)^2)
        const double growth_max = (double)sc->guard.dyn_k * 4.0 * (double)sc->guard.origin_radius * (double)sc->guard.origin_radius;
        const bool wide_par_scene = sc->guard.dyn_k > 0.0f && growth_max < 16384.0;
        return sc->whnodes != nullptr && sc->num_wide > 0 && !want_wavefront && (cfg.wide_nodes > 0 || (cfg.wide_nodes == 0 && wide_par_scene));
    };
    bool wide = wide_nodes_wanted();
    uint32_t gblock = want_wavefront ? (uint32_t)rtk::kWfBlock : (uint32_t)rtk::kBlock;     // threads per workgroup of the guarded pass
    int gwgs_per_cu = (want_wavefront ? RTP_WF_MIN_WAVES : RTP_MIN_WAVES) * 256 / (int)gblock;
    // the sphere-only build of the octant walk (render_kernel<…, kSimple>): 1024-thread workgroups, 8 waves per SIMD
    bool simple = false;
    uint32_t flag_chunk_words = 0;       // LDS words per wave for its chunk of the flagged-sample list (rt_kernel.hip.inc, flag_collect): 2, or 0
    if (guarded) {
        // The margins were sized for ray origins within origin_radius of origin_center, and those of the small
        // spheres for origins within sqrt(d0_sq) of their cluster: a camera outside either gets the tree re-packed
        // with margins for where it is (once per growth of the reach; the exact walk's tables do not change).
        auto outside = [&](const float *c, double radius_sq) {
            const double dx = (double)cam->origin.e[0] - c[0], dy = (double)cam->origin.e[1] - c[1], dz = (double)cam->origin.e[2] - c[2];
            return !(dx * dx + dy * dy + dz * dz <= radius_sq);
        };
        const bool far_cam = outside(sc->guard.origin_center, (double)sc->guard.origin_radius * sc->guard.origin_radius) ||
                             (sc->guard.num_small > 0 && outside(sc->guard.center, (double)sc->guard.d0_sq));
        if (far_cam && !sc->repack_refused && std::isfinite(cam->origin.e[0]) && std::isfinite(cam->origin.e[1]) &&
            std::isfinite(cam->origin.e[2]) && cfg.guard_repack) {
            st = repack_for_camera(sc, cam->origin.e, stream);
            if (st != RT_OK) return st;
            st = fill_params(sc, cam, shard, P, tile);       // table pointers and guard parameters changed
            if (st != RT_OK) return st;
            P.fb = d_fb_sum;
            wide = wide_nodes_wanted();
        }
        if (outside(sc->guard.origin_center, (double)sc->guard.origin_radius * sc->guard.origin_radius)) guarded = false;
        if (sc->repack_refused && sc->guard.num_small > 0 && outside(sc->guard.center, (double)sc->guard.d0_sq)) guarded = false;
    }
    if (guarded) {
        // fp32 records when LDS-resident: 7 float4 per wide node, 4 per pair node — 5 in the octant layout the kernel stages
        // for the pair walk with static margins (rt_kernel.hip.inc, step_octant)
        const bool octant = (RTP_OCTANT != 0) && !wide && !want_wavefront && !(sc->guard.dyn_k > 0.0f);
        uint64_t table_bytes = ((wide ? (uint64_t)P.num_wide * 7 : (uint64_t)P.num_internal * (octant ? 5 : 4)) + prim_f4) * 16 +
                               (!want_wavefront ? kGuardBlockBytes : 0u);      // + the guarded kernels' constants block
        const int32_t deepest = wide ? 3 * sc->wide_depth : sc->tree_depth;          // a wide node leaves up to three children waiting
        const int32_t want = deepest + 1 > 2 ? deepest + 1 : 2;       // never overflows
        uint32_t per_level = gblock * 4u;
        uint32_t pool_extra = 0;       // (the sphere-only builds have four more waves' work ranges than pool_bytes counts)
        // sphere-only build: no planes, no textures, leaf boxes recomputed from the spheres; its LDS holds no material rows
        // (all three come from global memory), which pays for the wider stack rows of 1024 lanes
        simple = octant && cfg.scene_in_lds != 0 && cfg.sphere_only_kernel >= 0 && P.num_planes == 0 && sc->tex_data == nullptr &&
                 P.leaf_boxes == nullptr && cfg.workgroups_per_cu == 0 && !sc->absorbing_glass;
        if (simple) {
            const uint64_t simple_bytes = ((uint64_t)P.num_internal * 5 + (uint64_t)P.num_spheres + ((uint64_t)P.num_spheres + 3) / 4) * 16;
            const uint64_t budget = kLdsLimit / 2;
            const int64_t fit = simple_bytes + pool_bytes + 256 < budget ? (int64_t)((budget - simple_bytes - pool_bytes - 256) / ((uint64_t)rtk::kSimpleBlock * 4u)) : 0;
            if (fit >= 6 || fit >= want) {          // at least the sentinel + 5 levels: below that the re-walk launch eats the gain
                gblock = (uint32_t)rtk::kSimpleBlock;
                gwgs_per_cu = rtk::kSimpleWaves * 256 / rtk::kSimpleBlock;
                table_bytes = simple_bytes + ((uint64_t)(rtk::kSimpleBlock / rtk::kWave) * 8u - pool_bytes) + kGuardBlockBytes;      // + the four extra waves' work ranges + the constants block
                per_level = gblock * 4u;
            } else {
                simple = false;
            }
        }
        // … and of the walk through L1 / L2 for scenes with distance-aware margins (step_pair_par on pair nodes: the 4-wide step does
        // not fit 64 registers): sphere-only scenes beyond what LDS holds — S-rtiow x 785 … 99 857 spheres: +7 … +10 % over the general
        // build on 4-wide nodes (8.1 / 7.7 / 6.7 / 3.9 against 7.5 / 7.2 / 6.2 / 3.6 Gsamples/s; tools/size_sweep.py)
        const bool dyn_global_scene = sc->guard.dyn_k > 0.0f && !want_wavefront && cfg.wide_nodes <= 0;
        if (!simple && dyn_global_scene && cfg.sphere_only_kernel >= 0 && P.num_planes == 0 && sc->tex_data == nullptr && P.leaf_boxes == nullptr &&
            cfg.workgroups_per_cu == 0 && !sc->absorbing_glass) {
            simple = true;
            wide = false;
            gblock = (uint32_t)rtk::kSimpleBlock;
            gwgs_per_cu = rtk::kSimpleWaves * 256 / rtk::kSimpleBlock;
            per_level = gblock * 4u;
            pool_extra = (uint32_t)(rtk::kSimpleBlock / rtk::kWave) * 8u - pool_bytes;
        }
        // tables in LDS when they leave room for a useful stack at full occupancy; else they are read
        // through L1/L2 and LDS holds only the stacks
        auto levels_for = [&](uint64_t scene_bytes, int wgs_per_cu) -> int32_t {
            const uint64_t budget = kLdsLimit / (uint64_t)wgs_per_cu;
            if (scene_bytes + pool_bytes >= budget) return 0;
            const int64_t fit = (int64_t)((budget - scene_bytes - pool_bytes) / per_level);
            return (int32_t)(fit < want ? fit : want);
        };
        const int32_t min_levels = want < 4 ? want : 4;          // a shorter stack flags too many rays
        fast.wgs_per_cu = gwgs_per_cu;
        // Scenes with distance-aware margins always take their records through L1 / L2 (step_pair_par: 32-byte binary16 records, a
        // full stack, two workgroups per CU whatever the size of the tree).  The LDS-resident form of that walk lost on every
        // random scene it was tried on — tables of a thousand nodes leave room for one workgroup per CU or for a stack of four,
        // and the rays a short stack hands to the exact walk cost more than L1 does: tools/dyn_probe.py, docs/LOG.md round 4.
        const bool dyn_global = sc->guard.dyn_k > 0.0f && !want_wavefront;
        fast.in_lds = cfg.scene_in_lds != 0 && !dyn_global;
        if (fast.in_lds) {
            fast.stack_levels = levels_for(table_bytes, fast.wgs_per_cu);
            while (fast.wgs_per_cu > 1 && fast.stack_levels < min_levels) fast.stack_levels = levels_for(table_bytes, --fast.wgs_per_cu);
            if (fast.stack_levels < min_levels) fast.in_lds = false;
        }
        // (step_pair_par / step_wide_par, the walks of scenes with distance-aware margins, keep a sentinel in level 0 — one level
        // more for the same twelve entries — and two rows of per-ray values behind the stack)
        const bool dyn_pair = !fast.in_lds && dyn_global;       // (pair nodes, or their 4-wide form)
        const int32_t extra_rows = dyn_pair ? 2 : 0;
        if (!fast.in_lds) {
            // tables through L1/L2: a 12-entry stack per lane (deeper ones are rare enough to flag), the rest of
            // the workgroup's LDS share holds the top of the tree
            fast.wgs_per_cu = gwgs_per_cu;
            const uint64_t budget = kLdsLimit / (uint64_t)fast.wgs_per_cu;
            const int64_t fit = budget > pool_bytes + pool_extra + kGuardBlockBytes ? (int64_t)((budget - pool_bytes - pool_extra - kGuardBlockBytes) / per_level) : 0;
            // (the 4-wide walk leaves up to three children of a node waiting, and LDS holds nothing but the stacks: twenty entries)
            const int32_t cap = (dyn_pair ? (wide ? 21 : 13) : 12) + extra_rows;
            fast.stack_levels = (int32_t)std::min<int64_t>(std::min<int64_t>(fit, want + extra_rows), cap);
        }
        if (const int forced = cfg.stack_levels) fast.stack_levels = forced + extra_rows < fast.stack_levels ? forced + extra_rows : fast.stack_levels;
        if (fast.stack_levels - extra_rows < (want < 2 ? want : 2)) guarded = false;
        // (the walks of scenes with distance-aware margins read every record through L1 / L2: no treelet)
        if (!fast.in_lds && cfg.lds_treelet && !dyn_pair) {
            const uint64_t budget = kLdsLimit / (uint64_t)fast.wgs_per_cu;
            const uint64_t used = pool_bytes + (uint64_t)fast.stack_levels * per_level + (!want_wavefront ? kGuardBlockBytes : 0u);
            const int64_t fit = budget > used ? (int64_t)((budget - used) / (wide ? 64 : 32)) : 0;
            const int32_t top_have = wide ? sc->num_top_wide : sc->num_top_pairs;
            fast.num_top = (int32_t)(fit < top_have ? fit : top_have);
        }
        if ((!fast.in_lds && !dyn_global_scene) || fast.wgs_per_cu != gwgs_per_cu) simple = false;      // (cannot happen after the fit test above; the general kernel is always right)
        fast.lds_bytes = (uint32_t)((fast.in_lds ? table_bytes : (uint64_t)fast.num_top * (wide ? 64 : 32) + pool_extra + (!want_wavefront ? kGuardBlockBytes : 0u)) + pool_bytes +
                                    (uint64_t)fast.stack_levels * per_level);
        if (const int w = cfg.workgroups_per_cu) { if ((uint64_t)w * fast.lds_bytes <= kLdsLimit) fast.wgs_per_cu = w; }
        if (!want_wavefront) flag_chunk_words = 2u;          // (the wave's chunk of the flagged-sample list: inside kGuardBlockBytes)
    }
    bool use_queue = false;
#ifdef RTP_DEV_QUEUE_KERNEL
    if (const char *kq = getenv("RTP_KERNEL")) use_queue = std::string(kq) == "queue";      // developer build only
#endif
    if (use_queue) guarded = false;

    // Samples per pass: as many as the workspace budget admits (rt_config.workspace_bytes, default just under 4 GiB;
    // 12 bytes per sample), at least 64, and few enough for the 32-bit work index and its reciprocal-multiply
    // division; the passes of a frame are made equally long.
    // Fewer, larger launches amortise the end-of-launch tail — on a row shard of an N-GPU frame
    // the pass grows N-fold, so a launch keeps the size it has on one GPU.
    const uint32_t num_pixels = (uint32_t)P.local_rows * (uint32_t)P.row_w;
    int pass_size = P.spp;
    // rows of the slab start on 128-byte lines (32 slots x 12 B = 3 lines) — except for passes shorter than that, whose rows are
    // only padded to the 16 bytes the accumulate kernel's row reads need (a 4K frame at 1 spp: 0.4 GB instead of 3.2 GB)
    auto pitch_of = [](int pass) { return pass < 32 ? (uint32_t)((pass + 3) & ~3) : (uint32_t)((pass + 31) & ~31); };
    {
        const uint64_t budget = cfg.workspace_bytes ? cfg.workspace_bytes : sc->device_bytes / kWorkspaceShareOfDevice;
        uint64_t fit = budget / ((uint64_t)num_pixels * kSampleBytes);
        fit &= ~(uint64_t)31;
        const uint64_t index_fit = (((uint64_t)1 << 30) - 64) / num_pixels;     // total_work + 64 <= 2^30
        if (fit > index_fit) fit = index_fit;
        if (fit < 64) fit = 64;
        if ((uint64_t)pass_size > fit) {
            const int passes_wanted = (int)((P.spp + fit - 1) / fit);
            pass_size = (P.spp + passes_wanted - 1) / passes_wanted;
        }
        if (const int forced = cfg.pass_spp) pass_size = forced < P.spp ? forced : P.spp;
        rtk::Magic probe;
        while (pass_size > 64 && !make_magic((uint32_t)pass_size, (uint64_t)num_pixels * pass_size + 64, probe)) --pass_size;
    }
    // workspace: three floats per (local pixel, slot)
    size_t need = (size_t)num_pixels * (size_t)pitch_of(pass_size) * 3;
    if (sc->slab_floats < need) {
        HIP_TRY(hipStreamSynchronize(stream));
        (void)hipFree(sc->slab);
        sc->slab = nullptr;
        sc->slab_floats = 0;
        for (;;) {      // a device short of memory gets shorter passes, not an error
            const hipError_t e = hipMalloc((void **)&sc->slab, need * sizeof(float));
            if (e == hipSuccess) break;
            (void)hipGetLastError();
            if (e != hipErrorOutOfMemory || pass_size <= 64) HIP_TRY(e);
            pass_size = pass_size / 2 < 64 ? 64 : pass_size / 2;
            need = (size_t)num_pixels * (size_t)pitch_of(pass_size) * 3;
        }
        sc->slab_floats = need;
    }
    P.slab = sc->slab;
    P.num_pixels = num_pixels;
    const int passes = (P.spp + pass_size - 1) / pass_size;
    if (passes > kMaxPasses) return fail(RT_ERR_UNSUPPORTED, "samples_per_pixel above 65536");
    if (guarded) {
        // flagged-sample list: a quarter of a pass's samples (a fuller list means "re-walk everything")
        const size_t cap = (size_t)num_pixels * (size_t)pass_size / 4 + 65536;
        if (sc->flag_cap < cap) {
            HIP_TRY(hipStreamSynchronize(stream));
            (void)hipFree(sc->flag_list);
            sc->flag_list = nullptr;
            sc->flag_cap = 0;
            HIP_TRY(hipMalloc((void **)&sc->flag_list, cap * sizeof(uint32_t)));
            sc->flag_cap = cap;
        }
        if (cfg.overlap_rework >= 0 && sc->dirty_cap < (size_t)num_pixels) {
            HIP_TRY(hipStreamSynchronize(stream));
            (void)hipFree(sc->dirty); (void)hipFree(sc->dirty_list);
            sc->dirty = sc->dirty_list = nullptr;
            sc->dirty_cap = 0;
            HIP_TRY(hipMalloc((void **)&sc->dirty, (size_t)num_pixels * sizeof(uint32_t)));
            HIP_TRY(hipMalloc((void **)&sc->dirty_list, (size_t)num_pixels * sizeof(uint32_t)));
            sc->dirty_cap = (size_t)num_pixels;
        }
        if (cfg.overlap_rework >= 0 && !sc->aux_stream) {
            HIP_TRY(hipStreamCreateWithFlags(&sc->aux_stream, hipStreamNonBlocking));
            HIP_TRY(hipStreamCreateWithFlags(&sc->list_stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&sc->ev_listed, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&sc->ev_fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&sc->ev_join, hipEventDisableTiming));
        }
    }

    // kernel form of the guarded pass: render_kernel (a lane owns a path) or render_kernel_wf (a wave owns a pool of paths)
    const bool dyn = guarded && sc->guard.dyn_k > 0.0f;                     // distance-aware margins (big or widely spread scenes)
    const bool wavefront = guarded && want_wavefront && !use_queue && !dyn;
    uint32_t wf_target = 0;
    if (wavefront) {
        wf_target = (uint32_t)(cfg.wavefront_paths > 0 ? cfg.wavefront_paths : 192);
        if (wf_target < 128u) wf_target = 128u;              // below that a wave can run dry (rt_kernel_wf.hip.inc)
        if (wf_target > 4096u) wf_target = 4096u;
        wf_target = (wf_target + 63u) & ~63u;
        const size_t waves_total = (size_t)sc->num_cus * fast.wgs_per_cu * (rtk::kWfBlock / rtk::kWave);
        const size_t need_pool = waves_total * (size_t)(rtk::kWfRayRows + rtk::kWfHitRows) * wf_target;
        if (sc->wf_pool_float4s < need_pool) {
            HIP_TRY(hipStreamSynchronize(stream));
            (void)hipFree(sc->wf_pool);
            sc->wf_pool = nullptr;
            sc->wf_pool_float4s = 0;
            HIP_TRY(hipMalloc((void **)&sc->wf_pool, need_pool * sizeof(float4)));
            sc->wf_pool_float4s = need_pool;
        }
    }
    const Shape &main_shape = guarded ? fast : (exact_simple ? exact_s : exact);
    const uint32_t max_wgs = (uint32_t)(((uint64_t)num_pixels * (P.spp < 64 ? P.spp : 64) + rtk::kBlock - 1) / rtk::kBlock);
    auto grid_for = [&](const Shape &sh) {
        int wgs = sc->num_cus * sh.wgs_per_cu;
        if ((uint32_t)wgs > max_wgs) wgs = (int)max_wgs;
        return wgs < 1 ? 1 : wgs;
    };
    int wgs = grid_for(main_shape);

    HIP_TRY(hipMemsetAsync(sc->queue, 0, kQueueWords * 4, stream));
    // (not with a capped flag list or an unproven margin: a list that overflowed makes the re-walk rewrite EVERY slab entry
    // while the other stream sums the rows of unflagged pixels — harmless only where both walks give the same bits)
    const bool overlap = guarded && !wavefront && !use_queue && cfg.overlap_rework >= 0 && sc->aux_stream != nullptr && sc->dirty != nullptr &&
                         cfg.flag_capacity == 0 && !gamma_unproven(cfg);
    // an early return between the fork to the second stream and the join must not leave that stream running unobserved
    struct JoinOnExit {
        rt_scene *sc; hipStream_t stream; bool forked = false, listing = false;
        ~JoinOnExit() {
            if (forked && hipEventRecord(sc->ev_join, sc->aux_stream) == hipSuccess) (void)hipStreamWaitEvent(stream, sc->ev_join, 0);
            if (listing) (void)hipStreamWaitEvent(stream, sc->ev_listed, 0);
        }
    } join_guard{sc, stream};
    // Primary visibility without a walk (rt_primary.hip.inc): the pair-node walks of render_kernel — the LDS-resident octant walk
    // with static margins, and the walk with distance-aware margins (LDS-resident or through L1/L2) — and a camera inside the
    // distance static margins were sized for (so that the far-origin test can never fire for a camera ray; the device compares
    // in float: a hair of slack)
    bool prim = guarded && !wavefront && !use_queue && (!wide || (dyn && !fast.in_lds)) && (dyn || ((RTP_OCTANT != 0) && fast.in_lds)) && cfg.primary_visibility >= 0 &&
                sc->nodes != nullptr && P.max_depth < rtk::kMaxPrimDepth;
    if (prim && sc->guard.num_small > 0) {
        const double dx = (double)cam->origin.e[0] - sc->guard.center[0], dy = (double)cam->origin.e[1] - sc->guard.center[1], dz = (double)cam->origin.e[2] - sc->guard.center[2];
        if (!(dx * dx + dy * dy + dz * dz <= (double)sc->guard.d0_sq * (1.0 - 1e-5))) prim = false;
    }
    if (prim && sc->cand_pixels < (size_t)num_pixels) {
        HIP_TRY(hipStreamSynchronize(stream));
        (void)hipFree(sc->cand);
        sc->cand = nullptr;
        sc->cand_pixels = 0;
        sc->cand_key.valid = false;
        // (64 bytes per pixel; a device short of memory renders without the pass rather than not at all)
        // + 4 bytes per pixel for the fetch order and 12 per 256 pixels for its counting sort (order_* kernels)
        const size_t cand_words = (size_t)num_pixels * (rtk::kCandWords + 1) + 3 * (((size_t)num_pixels + rtk::kOrderBlock - 1) / rtk::kOrderBlock);
        if (hipMalloc((void **)&sc->cand, cand_words * sizeof(uint32_t)) == hipSuccess) {
            sc->cand_pixels = (size_t)num_pixels;
        } else {
            (void)hipGetLastError();
            sc->cand = nullptr;
            prim = false;
        }
    }
    P.cand = prim ? sc->cand : nullptr;
    P.order = prim ? sc->cand + sc->cand_pixels * rtk::kCandWords : nullptr;
    P.traced_pixels = prim ? P.order + sc->cand_pixels + 2 * (((size_t)num_pixels + rtk::kOrderBlock - 1) / rtk::kOrderBlock) : nullptr;      // counts[2 * blocks] after the scan
    rt_scene::Feedback *feedback = nullptr;
    if ((st = acquire_feedback(sc, &feedback)) != RT_OK) return st;
    HIP_TRY(hipEventRecord(feedback->start, stream));
    HIP_TRY(hipEventRecord(sc->ev_start, stream));
    rt_scene::CandKey key{};
    if (prim) {
        std::memcpy(key.view + 0, cam->origin.e, 12); std::memcpy(key.view + 3, cam->pixel00_loc.e, 12);
        std::memcpy(key.view + 6, cam->pixel_delta_u.e, 12); std::memcpy(key.view + 9, cam->pixel_delta_v.e, 12);
        const int32_t dims[9] = {P.width, P.height, P.local_rows, P.band_rows, P.num_parts, P.part, P.row_w, P.tile_x0, P.tile_y0};
        std::memcpy(key.dims, dims, sizeof(dims));
        key.repacks = sc->repacks; key.stream = stream; key.valid = true;
    }
    const bool cand_cached = prim && cfg.reuse_view_lists >= 0 && sc->cand_key.valid && std::memcmp(key.view, sc->cand_key.view, sizeof(key.view)) == 0 &&
                             std::memcmp(key.dims, sc->cand_key.dims, sizeof(key.dims)) == 0 && key.repacks == sc->cand_key.repacks && key.stream == sc->cand_key.stream;
    if (!cand_cached) sc->cand_key.valid = false;          // (valid again once the launches below are queued)
    if (prim && !cand_cached) {
        const double coord_max = rtbeam::coord_bound(cam->origin.e, cam->pixel00_loc.e, cam->pixel_delta_u.e, cam->pixel_delta_v.e, cam->image_width, cam->image_height);
        hipLaunchKernelGGL(rtk::cand_kernel, dim3((num_pixels + 255u) / 256u), dim3(256), 0, stream, P, sc->cand, coord_max);
        HIP_TRY(hipGetLastError());
        // the order the trace kernel fetches the pixels in: expensive ones first (rt_primary.hip.inc)
        uint32_t *order = sc->cand + sc->cand_pixels * rtk::kCandWords, *counts = order + sc->cand_pixels;
        const uint32_t order_blocks = (num_pixels + (uint32_t)rtk::kOrderBlock - 1u) / (uint32_t)rtk::kOrderBlock;
        hipLaunchKernelGGL(rtk::order_count_kernel, dim3(order_blocks), dim3(rtk::kOrderBlock), 0, stream, (const uint32_t *)sc->cand, num_pixels, order_blocks, counts);
        hipLaunchKernelGGL(rtk::order_scan_kernel, dim3(1), dim3(1024), 0, stream, counts, 3u * order_blocks);
        hipLaunchKernelGGL(rtk::order_scatter_kernel, dim3(order_blocks), dim3(rtk::kOrderBlock), 0, stream, (const uint32_t *)sc->cand, num_pixels, order_blocks,
                           (const uint32_t *)counts, order);
        HIP_TRY(hipGetLastError());
        sc->cand_key = key;
    }
    hipStream_t launch_stream = stream;       // the exact re-walk may go to the handle's second stream (overlap_rework)
    // registers and scratch of the dominant (trace) kernel as the loaded code object reports them → rt_timing
    uint32_t trace_vgprs = 0, trace_scratch = 0;
    auto note_resources = [&](auto kernel) {
        hipFuncAttributes attr;
        if (trace_vgprs == 0 && hipFuncGetAttributes(&attr, (const void *)kernel) == hipSuccess) {
            trace_vgprs = (uint32_t)attr.numRegs;
            trace_scratch = (uint32_t)attr.localSizeBytes;
        }
    };
    auto launch = [&](auto kernel, const rtk::KParams &KP, int grid, uint32_t lds) -> hipError_t {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        note_resources(kernel);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(rtk::kBlock), lds, launch_stream, KP);
        return hipGetLastError();
    };
    auto launch_simple = [&](auto kernel, const rtk::KParams &KP, int grid, uint32_t lds) -> hipError_t {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        note_resources(kernel);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(rtk::kSimpleBlock), lds, stream, KP);
        return hipGetLastError();
    };
    auto launch_simple_on = [&](auto kernel, const rtk::KParams &KP, int grid, uint32_t lds) -> hipError_t {      // (on launch_stream: the re-walk may run on the second stream)
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        note_resources(kernel);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(rtk::kSimpleBlock), lds, launch_stream, KP);
        return hipGetLastError();
    };
    auto launch_wf = [&](auto kernel, const rtk::KParams &KP, int grid, uint32_t lds) -> hipError_t {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        note_resources(kernel);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(rtk::kWfBlock), lds, stream, KP);
        return hipGetLastError();
    };
    auto launch_exact = [&](rtk::KParams &KP, int grid, bool whole_pass) -> hipError_t {
        KP.stack_levels = 0;
        KP.num_top = exact.num_top;
        if (whole_pass && exact_simple) {
            KP.flag_chunk_words = 0;
            KP.bail_share = 0;
            rtk::fill_consts(KP);          // (its launch constants come from the LDS block, like the guarded sphere-only build's)
            return launch_simple_on(rtk::render_kernel<true, true, false, false, true>, KP, grid, exact_s.lds_bytes);
        }
        if (exact.in_lds) return launch(rtk::render_kernel<true, true>, KP, grid, exact.lds_bytes);
        return launch(rtk::render_kernel<false, true>, KP, grid, exact.lds_bytes);
    };
    while ((int)sc->pass_events.size() < 4 * (passes < kTimedPasses ? passes : kTimedPasses)) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        sc->pass_events.push_back(e);
    }
    sc->timed_passes = 0;
    uint32_t q_lds = 0;
#ifdef RTP_DEV_QUEUE_KERNEL
    // RTP_KERNEL=queue: T-waves/S-waves with LDS queues (rt_kernel_queue.hip.inc), exact walk + LDS-resident scenes only
    if (use_queue && exact.in_lds) {
        P.q_s_waves = env_int("RTP_Q_SWAVES", 4);
        P.q_k_refill = env_int("RTP_Q_KREFILL", 16);
        P.q_k_busy = env_int("RTP_Q_KBUSY", 0);
        P.q_k_leaf = env_int("RTP_Q_KLEAF", 24);
        const uint64_t base_f4 = ((uint64_t)P.num_tnodes + 1) * 2 + (uint64_t)P.num_spheres + (uint64_t)P.num_planes * 5 + ((uint64_t)P.num_spheres + 3) / 4;
        const uint64_t mat_bytes = (uint64_t)P.num_materials * 48;
        uint32_t slots = (uint32_t)env_int("RTP_Q_SLOTS", 1152);
        auto lds_for = [&](uint32_t n, bool mats) {
            uint32_t cap = 64;
            while (cap < n) cap <<= 1;
            return base_f4 * 16 + (mats ? mat_bytes : 0) + (uint64_t)n * rtk::kSlotDwords * 4 + 2ull * cap * 4 + (rtk::QC_WORDS + 2 * (rtk::kQBlock / rtk::kWave)) * 4 + 16;
        };
        P.q_mats_in_lds = env_int("RTP_Q_MATS_LDS", 1);
        if (P.q_mats_in_lds && lds_for(slots, true) > kLdsLimit) P.q_mats_in_lds = 0;
        while (slots > 128 && lds_for(slots, P.q_mats_in_lds != 0) > kLdsLimit) slots -= 64;
        if (lds_for(slots, P.q_mats_in_lds != 0) > kLdsLimit) use_queue = false;
        P.q_slots = slots;
        P.q_ring_cap = 64;
        while (P.q_ring_cap < slots) P.q_ring_cap <<= 1;
        q_lds = (uint32_t)lds_for(slots, P.q_mats_in_lds != 0);
    } else {
        use_queue = false;
    }
    if (use_queue) {
        wgs = sc->num_cus;
        if ((uint32_t)wgs > max_wgs) wgs = (int)max_wgs;
        if (wgs < 1) wgs = 1;
    }
#endif
    for (int pass = 0; pass < passes; ++pass) {
        // samples [pass_first, pass_first + pass_count) of every pixel, traced in any order into the slab …
        const bool timed_pass = pass < kTimedPasses;
        if (timed_pass) HIP_TRY(hipEventRecord(sc->pass_events[4 * pass], stream));
        P.pass_first = pass * pass_size;
        P.pass_count = P.spp - P.pass_first < pass_size ? P.spp - P.pass_first : pass_size;
        P.total_work = num_pixels * (uint32_t)P.pass_count;      // work index = pixel * pass_count + slot
        P.slab_pitch = pitch_of(pass_size);
        if ((uint64_t)num_pixels * (uint64_t)P.pass_count >= (1ull << 31) - 4096 || !make_magic((uint32_t)P.pass_count, (uint64_t)P.total_work + 64, P.magic_count))
            return fail(RT_ERR_UNSUPPORTED, "image too large for the work index arithmetic");
        P.queue = sc->queue + kQueueWork + pass;
        P.work_list = nullptr;
        // Work indices a wave reserves per atomicAdd on the pass's counter.  6144 waves hammering ONE address
        // with an atomic per 64 samples was the bottleneck of the whole kernel (5.05 -> 6.25 Gsamples/s with
        // 512 per atomic); small frames keep at least 16 reservations per wave so the tail stays balanced.
        {
            const uint64_t waves_total = (uint64_t)wgs * ((wavefront ? rtk::kWfBlock : ((guarded ? simple : exact_simple) ? rtk::kSimpleBlock : rtk::kBlock)) / rtk::kWave);
            uint64_t per = (uint64_t)P.total_work / (waves_total * 16u * 64u);
            per = per < 1 ? 1 : (per > 16 ? 16 : per);      // 1024 per atomic with 8192 waves: +1 % over 512 (32: the same, 64: the tail shows)
            if (const int forced = cfg.reserve_chunk) per = (uint64_t)(forced > 0 ? forced : 1);
            P.chunk = (uint32_t)(64u * per);
            P.taper_shift = 1;                      // remaining / (2 x waves), rounded to a power of two
            while (((uint64_t)1 << P.taper_shift) < RTP_TAPER_FACTOR * waves_total) ++P.taper_shift;
            if (!cfg.reserve_taper) P.taper_shift = 0;
        }
        if (prim) {
            // primary visibility of this pass's samples: (hit distance, primitive) into each sample's slot of the slab
            int pgrid = sc->num_cus * 8;                            // 256-thread workgroups: 8 waves per SIMD
            const bool by_pixel = P.pass_count >= RTP_BY_PIXEL_MIN;  // a wave per pixel once a pixel (nearly) fills it twice
            const uint32_t units = by_pixel ? (num_pixels + 3u) / 4u : (P.total_work + 255u) / 256u;
            if ((uint32_t)pgrid > units) pgrid = (int)units;
            if (by_pixel) {
                if (P.num_planes > 0) hipLaunchKernelGGL(rtk::primary_pixel_kernel<true>, dim3(pgrid), dim3(256), 0, stream, P);
                else hipLaunchKernelGGL(rtk::primary_pixel_kernel<false>, dim3(pgrid), dim3(256), 0, stream, P);
            } else if (P.num_planes > 0) hipLaunchKernelGGL(rtk::primary_kernel<true>, dim3(pgrid), dim3(256), 0, stream, P);
            else hipLaunchKernelGGL(rtk::primary_kernel<false>, dim3(pgrid), dim3(256), 0, stream, P);
            HIP_TRY(hipGetLastError());
        }
        if (timed_pass) HIP_TRY(hipEventRecord(sc->pass_events[4 * pass + 1], stream));
#ifdef RTP_DEV_QUEUE_KERNEL
        if (use_queue) {
            P.stack_levels = 0;
            HIP_TRY(hipFuncSetAttribute((const void *)rtk::render_kernel_q<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)q_lds));
            hipLaunchKernelGGL(rtk::render_kernel_q<true>, dim3(wgs), dim3(rtk::kQBlock), q_lds, stream, P);
            HIP_TRY(hipGetLastError());
        } else
#endif
        if (guarded) {
            // near-first walk; samples it cannot vouch for go to the list …
            P.stack_levels = fast.stack_levels;
            P.num_top = fast.num_top;
            // vote thresholds of the octant walk (its pair steps are cheaper, so a block of them is worth starting for fewer
            // idle lanes and a shade step is worth waiting for a few more): re-swept after step_octant, +1.0 %
            const bool octant_launch = (RTP_OCTANT != 0) && fast.in_lds && !wide && !wavefront && !dyn;
            if (octant_launch && sc->cfg.k_inner <= 0) P.k_inner = 32;
            if (octant_launch && sc->cfg.k_shade <= 0) P.k_shade = 52;
            // scene with distance-aware margins (records through L1/L2, step_pair_par / step_wide_par): a block of steps costs memory round
            // trips on top of its instructions — worth starting only for a nearly full wave (S-100k, swept 16-52 x 44-56: +4 %)
            const bool big_dyn_launch = dyn && !fast.in_lds && !wavefront;
            if (big_dyn_launch && sc->cfg.k_inner <= 0) P.k_inner = 48;
            if (big_dyn_launch && sc->cfg.k_shade <= 0) P.k_shade = 52;
            P.flag_list = sc->flag_list;
            // resume table: cleared per pass (1 MB of tags); not for the experimental kernels, nor for paths longer than the depth field
            constexpr uint32_t kResumeBits = 18;
            if (!wavefront && cfg.resume_flagged >= 0 && P.max_depth < rtk::kMaxPrimDepth) {
                if (sc->resume_tag == nullptr) {
                    if (hipMalloc((void **)&sc->resume_tag, sizeof(uint32_t) << kResumeBits) != hipSuccess ||
                        hipMalloc((void **)&sc->resume_state, (size_t)64 << kResumeBits) != hipSuccess) {
                        (void)hipGetLastError();
                        (void)hipFree(sc->resume_tag); (void)hipFree(sc->resume_state);
                        sc->resume_tag = nullptr; sc->resume_state = nullptr;
                    }
                }
                if (sc->resume_tag != nullptr) HIP_TRY(hipMemsetAsync(sc->resume_tag, 0, sizeof(uint32_t) << kResumeBits, stream));
                P.resume_tag = sc->resume_tag; P.resume_state = sc->resume_state; P.resume_shift = 32u - kResumeBits;
            } else {
                P.resume_tag = nullptr; P.resume_state = nullptr;
            }
            P.flag_count = sc->queue + kQueueFlag + pass;
            P.dirty = overlap ? sc->dirty : nullptr;
            if (overlap) HIP_TRY(hipMemsetAsync(sc->dirty, 0, (size_t)num_pixels * sizeof(uint32_t), stream));      // this pass's marks (8 MB at 1080p: microseconds)
            P.dirty_list = overlap ? sc->dirty_list : nullptr;
            P.dirty_count = sc->queue + kQueueDirty + pass;
            P.flag_cap = (uint32_t)(sc->flag_cap < 0xffffffffu ? sc->flag_cap : 0xffffffffu);
            if (const uint32_t tiny = cfg.flag_capacity) P.flag_cap = tiny < P.flag_cap ? tiny : P.flag_cap;   // test hook: overflow path
            P.flag_chunk_words = flag_chunk_words;
            // in-launch bail-out (not for a caller who insists on the guarded walk, nor for the experimental kernels)
            P.bail_share = (cfg.guard_keep || wavefront) ? 0u : bail_share_of(cfg);
            P.bail_floor = bail_floor();
            P.bail_latest = bail_latest();
            P.abandon = P.bail_share != 0u ? sc->queue + kQueueAbandon + pass : nullptr;
            rtk::fill_consts(P);          // (everything the constants block copies is final now)
            if (wavefront) {
                P.wf_pool = sc->wf_pool;
                P.wf_cap = P.wf_target = wf_target;
                P.wf_k_exchange = cfg.wavefront_exchange > 0 ? (cfg.wavefront_exchange > 64 ? 64 : cfg.wavefront_exchange) : 16;
#ifdef RTP_DEV_BUILD
                if (fast.in_lds) HIP_TRY(launch_wf(rtk::render_kernel_wf<true>, P, wgs, fast.lds_bytes));
                else HIP_TRY(launch_wf(rtk::render_kernel_wf<false>, P, wgs, fast.lds_bytes));
#else
                (void)launch_wf;
#endif
            } else if (wide && dyn && !fast.in_lds) {      // distance-aware margins on 4-wide nodes (step_wide_par)
                if (prim) HIP_TRY(launch(rtk::render_kernel<false, false, true, true, false, true>, P, wgs, fast.lds_bytes));
                else HIP_TRY(launch(rtk::render_kernel<false, false, true, true>, P, wgs, fast.lds_bytes));
#ifdef RTP_DEV_BUILD
            } else if (wide) {
                if (fast.in_lds) HIP_TRY(launch(rtk::render_kernel<true, false, false, true>, P, wgs, fast.lds_bytes));
                else HIP_TRY(launch(rtk::render_kernel<false, false, false, true>, P, wgs, fast.lds_bytes));
#endif
            } else if (dyn && simple) {      // sphere-only scenes beyond what LDS holds: step_pair_par at 64 registers, 8 waves per SIMD
                if (prim) HIP_TRY(launch_simple(rtk::render_kernel<false, false, true, false, true, true>, P, wgs, fast.lds_bytes));
                else HIP_TRY(launch_simple(rtk::render_kernel<false, false, true, false, true>, P, wgs, fast.lds_bytes));
            } else if (dyn) {
                if (prim) HIP_TRY(launch(rtk::render_kernel<false, false, true, false, false, true>, P, wgs, fast.lds_bytes));
                else HIP_TRY(launch(rtk::render_kernel<false, false, true>, P, wgs, fast.lds_bytes));
            } else if (fast.in_lds && simple && prim) HIP_TRY(launch_simple(rtk::render_kernel<true, false, false, false, true, true>, P, wgs, fast.lds_bytes));
            else if (fast.in_lds && simple) HIP_TRY(launch_simple(rtk::render_kernel<true, false, false, false, true>, P, wgs, fast.lds_bytes));
            else if (fast.in_lds && prim) HIP_TRY(launch(rtk::render_kernel<true, false, false, false, false, true>, P, wgs, fast.lds_bytes));
            else if (fast.in_lds) HIP_TRY(launch(rtk::render_kernel<true, false>, P, wgs, fast.lds_bytes));
            else HIP_TRY(launch(rtk::render_kernel<false, false>, P, wgs, fast.lds_bytes));
            if (timed_pass) HIP_TRY(hipEventRecord(sc->pass_events[4 * pass + 2], stream));
            // … and are walked again in the reference's order, overwriting their slab entries
            rtk::KParams R = P;
            R.k_inner = sc->cfg.k_inner > 0 ? sc->cfg.k_inner : 24;
            R.k_shade = sc->cfg.k_shade > 0 ? sc->cfg.k_shade : 48;
            R.queue = sc->queue + kQueueRework + pass;
            R.work_list = sc->flag_list;
            R.work_count = sc->queue + kQueueFlag + pass;
            R.work_cap = P.flag_cap;
            R.chunk = 64u;                     // a short list: finest granularity
            R.taper_shift = 0;
            {       // … unless it turns out to be the whole pass (overflow, abandoned guarded pass): a trace launch's reservations
                const uint64_t waves_total = (uint64_t)grid_for(exact) * (rtk::kBlock / rtk::kWave);
                uint64_t per = (uint64_t)P.total_work / (waves_total * 16u * 64u);
                per = per < 1 ? 1 : (per > 16 ? 16 : per);
                R.full_chunk = (uint32_t)(64u * per);
                R.full_taper = 1;
                while (((uint64_t)1 << R.full_taper) < RTP_TAPER_FACTOR * waves_total) ++R.full_taper;
            }
            R.dirty = nullptr;
            if (overlap) {
                // the re-walk and the accumulation of the pixels it touches on the second stream …
                HIP_TRY(hipEventRecord(sc->ev_fork, stream));
                HIP_TRY(hipStreamWaitEvent(sc->aux_stream, sc->ev_fork, 0));
                join_guard.forked = true;
                launch_stream = sc->aux_stream;
                // the pixels the trace launch marked, as a list for the second accumulate launch — made on a third stream beside the
                // re-walk (0.2 ms at 1080p that would otherwise lengthen the re-walk's chain past the other pixels' accumulation)
                HIP_TRY(hipStreamWaitEvent(sc->list_stream, sc->ev_fork, 0));
                hipLaunchKernelGGL(rtk::dirty_compact_kernel, dim3((num_pixels + rtk::kDirtyPixels - 1) / rtk::kDirtyPixels), dim3(rtk::kDirtyBlock), 0, sc->list_stream,
                                   (const uint32_t *)sc->dirty, num_pixels, sc->dirty_list, sc->queue + kQueueDirty + pass);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventRecord(sc->ev_listed, sc->list_stream));
                join_guard.listing = true;
            }
            HIP_TRY(launch_exact(R, grid_for(exact), false));
            launch_stream = stream;
        } else {
            HIP_TRY(launch_exact(P, wgs, true));
        }
        const bool overlapped = overlap && guarded && !wavefront;
        if (timed_pass) {
            if (!guarded) HIP_TRY(hipEventRecord(sc->pass_events[4 * pass + 2], stream));
            HIP_TRY(hipEventRecord(sc->pass_events[4 * pass + 3], overlapped ? sc->aux_stream : stream));
            sc->timed_passes = pass + 1;
        }
        // … then added to the pixel sums strictly in sample order
        const dim3 acc_grid((num_pixels + 64 * rtk::kAccWaves - 1) / (64 * rtk::kAccWaves)), acc_block(64 * rtk::kAccWaves);
        if (overlapped) {
            // (a pass the guarded launch gave up has rows nobody traced yet: both launches stand down — P.abandon — and a third one,
            // after the re-walk of everything, sums every pixel)
            HIP_TRY(hipStreamWaitEvent(sc->aux_stream, sc->ev_listed, 0));
            join_guard.listing = false;
            hipLaunchKernelGGL(rtk::accumulate_kernel<true>, acc_grid, acc_block, 0, sc->aux_stream, d_fb_sum, (const float *)sc->slab, num_pixels, P.slab_pitch,
                               P.pass_count, pass == 0 ? 1 : 0, sc->dirty, (const uint32_t *)sc->dirty_list, (const uint32_t *)(sc->queue + kQueueDirty + pass),
                               (const uint32_t *)nullptr, 0u, 0.0f, 0.0f, 0.0f, (const uint32_t *)P.abandon, 0u);
            HIP_TRY(hipEventRecord(sc->ev_join, sc->aux_stream));
            // … while every other pixel is accumulated here
            hipLaunchKernelGGL(rtk::accumulate_kernel<false>, acc_grid, acc_block, 0, stream, d_fb_sum, (const float *)sc->slab, num_pixels, P.slab_pitch,
                               P.pass_count, pass == 0 ? 1 : 0, sc->dirty, (const uint32_t *)nullptr, (const uint32_t *)nullptr, P.cand, (uint32_t)rtk::kCandWords, P.bg[0], P.bg[1], P.bg[2],
                               (const uint32_t *)P.abandon, 0u);
            HIP_TRY(hipStreamWaitEvent(stream, sc->ev_join, 0));
            join_guard.forked = false;
            if (P.abandon)
                hipLaunchKernelGGL(rtk::accumulate_kernel<false>, acc_grid, acc_block, 0, stream, d_fb_sum, (const float *)sc->slab, num_pixels, P.slab_pitch,
                                   P.pass_count, pass == 0 ? 1 : 0, (uint32_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr, P.cand, (uint32_t)rtk::kCandWords,
                                   P.bg[0], P.bg[1], P.bg[2], (const uint32_t *)P.abandon, 1u);
        } else {
            hipLaunchKernelGGL(rtk::accumulate_kernel<false>, acc_grid, acc_block, 0, stream, d_fb_sum, (const float *)sc->slab, num_pixels, P.slab_pitch,
                               P.pass_count, pass == 0 ? 1 : 0, (uint32_t *)nullptr, (const uint32_t *)nullptr, (const uint32_t *)nullptr, P.cand, (uint32_t)rtk::kCandWords, P.bg[0], P.bg[1], P.bg[2]);
        }
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(sc->ev_stop, stream));
    {
        // flagged counts and abandon words of this call → pinned host memory, and the frame's end, for a later call's poll_feedback
        rt_scene::Feedback &f = *feedback;
        if (guarded) {
            HIP_TRY(hipMemcpyAsync(f.host, sc->queue + kQueueFlag, (size_t)passes * 4, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(f.host + kMaxPasses, sc->queue + kQueueAbandon, (size_t)passes * 4, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(f.host + 2 * kMaxPasses, sc->queue + kQueueHoles, (size_t)passes * 4, hipMemcpyDeviceToHost, stream));
        }
        HIP_TRY(hipEventRecord(f.done, stream));
        f.pending = true;
        f.guarded = guarded;
        f.exploring = exploring;
        f.passes = passes;
        f.samples = (uint64_t)num_pixels * (uint64_t)P.spp;
        f.serial = ++sc->frame_serial;
    }
    sc->timed = true;
    sc->last = rt_timing{};
    sc->last.num_workgroups = (uint32_t)wgs;
    sc->last.workgroup_size = wavefront ? (uint32_t)rtk::kWfBlock : ((guarded ? simple : exact_simple) ? (uint32_t)rtk::kSimpleBlock : (uint32_t)rtk::kBlock);
    sc->last.lds_bytes = main_shape.lds_bytes;
#ifdef RTP_DEV_QUEUE_KERNEL
    if (use_queue) { sc->last.workgroup_size = rtk::kQBlock; sc->last.lds_bytes = q_lds; }
#endif
    (void)q_lds;
    sc->last.scene_in_lds = main_shape.in_lds ? 1u : 0u;
    sc->last.trace_launches = (uint32_t)passes;
    sc->last.guarded = guarded ? 1u : 0u;
    sc->last.guard_unproven = (guarded && gamma_unproven(cfg)) ? 1u : 0u;
    sc->last.kernel = wavefront ? RT_KERNEL_WAVEFRONT : RT_KERNEL_MEGA;
    sc->last.guard_dynamic = dyn ? 1u : 0u;
    sc->last.front_primitives = guarded ? (uint32_t)sc->guard.num_front : 0u;
    sc->last.wide_nodes = (guarded && wide) ? 1u : 0u;
    sc->last.sphere_only = ((guarded && simple && !wavefront && !wide) || (!guarded && exact_simple)) ? 1u : 0u;
    sc->last.primary_visibility = prim ? 1u : 0u;
    sc->last.trace_vgprs = trace_vgprs;
    sc->last.trace_scratch_bytes = trace_scratch;
    sc->last_passes = passes;
    sc->last_samples = (uint64_t)num_pixels * (uint64_t)P.spp;
    sc->last_traced_pixels = prim ? P.traced_pixels : nullptr;
    sc->last_spp = P.spp;
    if (sync) return rt_last_timing(sc, timing);
    timing_out(sc->last, timing);
    return RT_OK;
}
}  // namespace

extern "C" {

rt_status rt_render(rt_scene *sc, const rt_camera_data *cam, const rt_shard *shard, float *d_fb_sum, void *hip_stream,
                    int32_t sync, rt_timing *timing) {
    return render_impl(sc, cam, shard, nullptr, d_fb_sum, hip_stream, sync, timing);
}

rt_status rt_render_tile(rt_scene *sc, const rt_camera_data *cam, int32_t tile_x0, int32_t tile_y0, int32_t tile_w, int32_t tile_h,
                         float *d_fb_sum, void *hip_stream, int32_t sync, rt_timing *timing) {
    const Tile tile{tile_x0, tile_y0, tile_w, tile_h};
    return render_impl(sc, cam, nullptr, &tile, d_fb_sum, hip_stream, sync, timing);
}

void rt_timing_init(rt_timing *t) {
    if (!t) return;
    std::memset(t, 0, sizeof(*t));
    t->struct_bytes = (uint32_t)sizeof(*t);
}

rt_status rt_last_timing(rt_scene *sc, rt_timing *timing) {
    if (!sc) return fail(RT_ERR_INVALID_ARG, "null argument");
    if (const rt_status ts = timing_check(timing)) return ts;
    if (sc->timed) {
        HIP_TRY(hipEventSynchronize(sc->ev_stop));
        HIP_TRY(hipEventElapsedTime(&sc->last.kernel_ms, sc->ev_start, sc->ev_stop));
        float sum = 0.0f, rework = 0.0f, primary = 0.0f;
        if (const rt_status ps = frame_parts(sc, sum, rework, primary)) return ps;
        sc->last.trace_ms = sum;
        sc->last.rework_ms = rework;
        sc->last.primary_ms = sc->last.primary_visibility ? primary : 0.0f;
        if (sc->last.guarded) {
            std::vector<uint32_t> counts((size_t)sc->last_passes), gave_up((size_t)sc->last_passes), holes((size_t)sc->last_passes);
            HIP_TRY(hipMemcpy(counts.data(), sc->queue + kQueueFlag, counts.size() * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(holes.data(), sc->queue + kQueueHoles, holes.size() * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(gave_up.data(), sc->queue + kQueueAbandon, gave_up.size() * 4, hipMemcpyDeviceToHost));
            uint64_t total = 0;
            uint32_t abandoned = 0;
            for (size_t p = 0; p < counts.size(); ++p) { total += counts[p] - holes[p]; abandoned += gave_up[p] != 0u ? 1u : 0u; }
            sc->last.flagged_samples = total;
            sc->last.abandoned_passes = abandoned;
        }
        // (the frame is done, so what it left for the handle's judgement has landed as well: a scene the guarded walk keeps handing
        // back — dense overlaps, a camera inside a sphere, … — is cheaper on the exact walk alone; later frames of this handle use it)
        if (const rt_status ps = poll_feedback(sc, true)) return ps;
        sc->last.guard_paused = sc->guard_paused ? 1u : 0u;
        sc->last.traced_samples = sc->last_samples;
        if (sc->last_traced_pixels) {
            uint32_t traced = 0;
            HIP_TRY(hipMemcpy(&traced, sc->last_traced_pixels, 4, hipMemcpyDeviceToHost));
            sc->last.traced_samples = (uint64_t)traced * (uint64_t)sc->last_spp;
        }
        uint32_t abort_code = 0;
        HIP_TRY(hipMemcpy(&abort_code, sc->queue + kQueueStats + 15, 4, hipMemcpyDeviceToHost));
        if (abort_code != 0) return fail(RT_ERR_HIP, "render kernel aborted (protocol timeout, code " + std::to_string(abort_code) + ")");
    }
    timing_out(sc->last, timing);
    return RT_OK;
}

#ifdef RTP_DEV_BUILD
// Developer hook (not part of the ABI header): the proof by exhaustion behind rt_device_math.h's recip() and sqrt_cr().
// Runs every one of the 2^32 binary32 bit patterns through them on the current device and counts the inputs whose result
// differs in any bit from the compiler's correctly rounded 1.0f / x and sqrtf(x) (NaN results count as equal to NaN).
// out[0]: mismatches of recip, out[1]: of sqrt_cr, out[2]: inputs compared.  Both must be 0.
namespace rtk {
__global__ void __launch_bounds__(256) fast_math_check_kernel(unsigned long long *out) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    unsigned long long bad_r = 0, bad_s = 0, seen = 0;
    for (uint64_t b = tid; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const float wr = 1.0f / x, gr = rtd::recip(x);
        const float ws = sqrtf(x), gs = rtd::sqrt_cr(x);
        bad_r += (__float_as_uint(wr) != __float_as_uint(gr)) && !(wr != wr && gr != gr);
        bad_s += (__float_as_uint(ws) != __float_as_uint(gs)) && !(ws != ws && gs != gs);
        ++seen;
    }
    atomicAdd(&out[0], bad_r); atomicAdd(&out[1], bad_s); atomicAdd(&out[2], seen);
}
}  // namespace rtk
// The same for test_sphere's root selection (rt_kernel.hip.inc, sphere_root: both quotients from one fp64 reciprocal, no
// scaling): n operand sets (half_b, D, a, closest) — half of them raw random bit patterns (every exponent, infinities, NaN,
// denormals, zeros), half with exponents near the scene's scale — through sphere_root and through sphere_root_plain, the
// reference's form with the compiler's division.  out[0]: sets where "accepted" differs or the accepted t differs in any
// bit, out[1]: sets with an accepted root, out[2]: sets compared.
namespace rtk {
__global__ void __launch_bounds__(256) sphere_root_check_kernel(unsigned long long *out, unsigned long long n) {
    const uint64_t tid = blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0, hits = 0, seen = 0;
    for (uint64_t i = tid; i < n; i += stride) {
        uint32_t h = rtd::wang_hash((uint32_t)i ^ 0x9e3779b9u) + (uint32_t)(i >> 32) * 0x85ebca6bu;
        uint32_t w[4];
        for (int k = 0; k < 4; ++k) { h = rtd::wang_hash(h + 0x632be5abu * (k + 1)); w[k] = h; }
        if (i & 1) {            // exponents within 2^-20 .. 2^20 of 1, random mantissas and signs
            for (int k = 0; k < 4; ++k) w[k] = (w[k] & 0x807fffffu) | ((107u + ((w[k] >> 23) & 255u) % 41u) << 23);
        }
        const float half_b = __uint_as_float(w[0]);
        const float disc = __uint_as_float(w[1] & 0x7fffffffu);          // test_sphere has returned for D < 0
        const float a = __uint_as_float(w[2] & 0x7fffffffu);             // a = |d|^2
        const float closest = __uint_as_float(w[3] & 0x7fffffffu);
        const double sq = (double)rtd::sqrt_cr(disc), nb = (double)(-half_b), da = (double)a;
        float tf = 0.0f, tp = 0.0f;
        const bool of = sphere_root(nb, sq, da, closest, tf), op = sphere_root_plain(nb, sq, da, closest, tp);
        bad += (of != op) || (of && __float_as_uint(tf) != __float_as_uint(tp));
        hits += op;
        ++seen;
    }
    atomicAdd(&out[0], bad); atomicAdd(&out[1], hits); atomicAdd(&out[2], seen);
}
}  // namespace rtk
rt_status rt_debug_check_sphere_roots(uint64_t n, uint64_t out[3]) {
    if (!out) return fail(RT_ERR_INVALID_ARG, "null argument");
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc(&d, 24));
    hipError_t e = hipMemset(d, 0, 24);
    if (e == hipSuccess) { hipLaunchKernelGGL(rtk::sphere_root_check_kernel, dim3(4096), dim3(256), 0, 0, d, (unsigned long long)n); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpy(out, d, 24, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    HIP_TRY(e);
    return RT_OK;
}

rt_status rt_debug_check_fast_math(uint64_t out[3]) {
    if (!out) return fail(RT_ERR_INVALID_ARG, "null argument");
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc(&d, 24));
    hipError_t e = hipMemset(d, 0, 24);
    if (e == hipSuccess) { hipLaunchKernelGGL(rtk::fast_math_check_kernel, dim3(4096), dim3(256), 0, 0, d); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpy(out, d, 24, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    HIP_TRY(e);
    return RT_OK;
}

// Developer hook (not part of the ABI header): the next render calls of this scene inject the fault the tripwire exists for
// (rt_kernel.hip.inc, RTP_TRIPWIRE) — rt_last_timing must then fail with RT_ERR_HIP and the tripwire's code instead of the
// launch hanging.  0 switches it off again.
rt_status rt_debug_trip_test(rt_scene *sc, uint32_t on) {
    if (!sc) return fail(RT_ERR_INVALID_ARG, "null argument");
    sc->trip_test = on;
    return RT_OK;
}

// Developer hook (not part of the ABI header): raw counters of an RTP_STATS build.
rt_status rt_debug_read_stats(rt_scene *sc, uint32_t out[16]) {
    if (!sc || !out) return fail(RT_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, sc->queue + kQueueStats, 60, hipMemcpyDeviceToHost));
    return RT_OK;
}

#endif      // RTP_DEV_BUILD

rt_status rt_last_kernel_ms(rt_scene *sc, float *ms) {
    if (!sc || !ms) return fail(RT_ERR_INVALID_ARG, "null argument");
    *ms = 0.0f;
    if (!sc->timed) return RT_OK;
    HIP_TRY(hipEventSynchronize(sc->ev_stop));
    HIP_TRY(hipEventElapsedTime(ms, sc->ev_start, sc->ev_stop));
    return RT_OK;
}

rt_status rt_render_to_host(rt_scene *sc, const rt_camera_data *cam, const rt_shard *shard, float *h_fb_sum, rt_timing *timing) {
    if (!sc || !cam || !h_fb_sum) return fail(RT_ERR_INVALID_ARG, "null argument");
    const int32_t rows = rt_shard_rows(cam->image_height, shard);
    const size_t bytes = (size_t)rows * (size_t)(cam->image_width > 0 ? cam->image_width : 0) * 3 * sizeof(float);
    if (bytes == 0) return RT_OK;
    float *d_fb = nullptr;
    HIP_TRY(hipMalloc((void **)&d_fb, bytes));
    rt_status st = rt_render(sc, cam, shard, d_fb, nullptr, 1, timing);
    if (st == RT_OK && hipMemcpy(h_fb_sum, d_fb, bytes, hipMemcpyDeviceToHost) != hipSuccess) st = fail(RT_ERR_HIP, "hipMemcpy D2H failed");
    (void)hipFree(d_fb);
    return st;
}

rt_status rt_trace_samples(rt_scene *sc, const rt_camera_data *cam, int32_t n, const int32_t *ijs, float *radiance,
                           int32_t *rays, uint32_t *final_seed) {
    if (n < 0 || (n > 0 && (!ijs || !radiance || !rays || !final_seed))) return fail(RT_ERR_INVALID_ARG, "null argument");
    rtk::KParams P;
    rt_status st = fill_params(sc, cam, nullptr, P);
    if (st != RT_OK) return st;
    if ((st = check_device(sc)) != RT_OK) return st;
    if (n == 0) return RT_OK;
    for (int32_t k = 0; k < n; ++k)
        if (ijs[3 * k] < 0 || ijs[3 * k] >= cam->image_width || ijs[3 * k + 1] < 0 || ijs[3 * k + 1] >= cam->image_height || ijs[3 * k + 2] < 0)
            return fail(RT_ERR_INVALID_ARG, "sample coordinate out of range");
    int32_t *d_ijs = nullptr, *d_rays = nullptr;
    float *d_rad = nullptr;
    uint32_t *d_seed = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_ijs); (void)hipFree(d_rays); (void)hipFree(d_rad); (void)hipFree(d_seed); };
    if (hipMalloc((void **)&d_ijs, (size_t)n * 12) != hipSuccess || hipMalloc((void **)&d_rad, (size_t)n * 12) != hipSuccess ||
        hipMalloc((void **)&d_rays, (size_t)n * 4) != hipSuccess || hipMalloc((void **)&d_seed, (size_t)n * 4) != hipSuccess) {
        cleanup();
        return fail(RT_ERR_OUT_OF_MEMORY, "hipMalloc failed");
    }
    if (hipMemcpy(d_ijs, ijs, (size_t)n * 12, hipMemcpyHostToDevice) != hipSuccess) { cleanup(); return fail(RT_ERR_HIP, "hipMemcpy H2D failed"); }
    P.probe_ijs = d_ijs; P.probe_rad = d_rad; P.probe_rays = d_rays; P.probe_seed = d_seed; P.probe_n = n;
    hipLaunchKernelGGL(rtk::probe_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, P);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(radiance, d_rad, (size_t)n * 12, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(rays, d_rays, (size_t)n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(final_seed, d_seed, (size_t)n * 4, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fail(RT_ERR_HIP, std::string("probe kernel: ") + hipGetErrorString(e));
    return RT_OK;
}

rt_status rt_closest_hits(rt_scene *sc, int32_t n, const float *origins, const float *directions, int32_t *hit, float *t, int32_t *prim) {
    if (!sc || n < 0 || (n > 0 && (!origins || !directions || !hit || !t || !prim))) return fail(RT_ERR_INVALID_ARG, "null argument");
    if (n == 0) return RT_OK;
    rt_camera_data cam{};
    cam.image_width = cam.image_height = 1;
    cam.samples_per_pixel = cam.max_depth = 1;
    rtk::KParams P;
    rt_status st = fill_params(sc, &cam, nullptr, P);
    if (st != RT_OK) return st;
    if ((st = check_device(sc)) != RT_OK) return st;
    float *d_o = nullptr, *d_d = nullptr, *d_t = nullptr;
    int32_t *d_hit = nullptr, *d_prim = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_o); (void)hipFree(d_d); (void)hipFree(d_t); (void)hipFree(d_hit); (void)hipFree(d_prim); };
    const size_t n3 = (size_t)n * 12, n1 = (size_t)n * 4;
    if (hipMalloc((void **)&d_o, n3) != hipSuccess || hipMalloc((void **)&d_d, n3) != hipSuccess || hipMalloc((void **)&d_t, n1) != hipSuccess ||
        hipMalloc((void **)&d_hit, n1) != hipSuccess || hipMalloc((void **)&d_prim, n1) != hipSuccess) {
        cleanup();
        return fail(RT_ERR_OUT_OF_MEMORY, "hipMalloc failed");
    }
    hipError_t e = hipMemcpy(d_o, origins, n3, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_d, directions, n3, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_t, 0, n1);
    if (e == hipSuccess) e = hipMemset(d_prim, 0xff, n1);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rtk::closest_hit_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, P, d_o, d_d, n, d_hit, d_t, d_prim);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(hit, d_hit, n1, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(t, d_t, n1, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(prim, d_prim, n1, hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fail(RT_ERR_HIP, std::string("closest-hit probe: ") + hipGetErrorString(e));
    return RT_OK;
}

rt_status rt_device_alloc(uint64_t bytes, void **out) {
    if (!out) return fail(RT_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (bytes == 0) return RT_OK;
    HIP_TRY(hipMalloc(out, bytes));
    return RT_OK;
}

rt_status rt_device_free(void *p) {
    if (p) HIP_TRY(hipFree(p));
    return RT_OK;
}

rt_status rt_copy_to_host(void *dst, const void *src, uint64_t bytes) {
    if (bytes == 0) return RT_OK;
    if (!dst || !src) return fail(RT_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return RT_OK;
}

rt_status rt_tonemap(const float *d_fb_sum, uint8_t *d_rgb8, int64_t num_floats, int32_t divisor, void *hip_stream) {
    if (num_floats <= 0) return RT_OK;
    if (!d_fb_sum || !d_rgb8) return fail(RT_ERR_INVALID_ARG, "null argument");
    const float inv = (float)(1.0 / (double)(float)divisor);     // pixel_color / samplesPerPixel, include/vec3.h:97
    int64_t blocks = (num_floats + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(rtk::tonemap_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)hip_stream, d_fb_sum, d_rgb8, num_floats, inv);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

}  // extern "C"
