// rt_multi.hip — one frame sharded over the GPUs of one node, behind the C ABI (include/rtp_amd.h, "multi-GPU").
//
// The reference renders on one GPU (src/camera.cu:290-349 has no device selection at all); BASELINE.json's
// configs[3] asks for the frame's tiles spread over the node's GPUs with ONE gather over xGMI at frame end
// (SURVEY.md §8(e)).  A context owns, per device: a stream, a replica of the scene, the device's rows of the frame,
// and an RCCL communicator (single process, ncclCommInitAll).  rt_render_sharded():
//     every device renders its interleaved row bands (rt_render with an rt_shard, asynchronously, each on its own
//     stream) → rt_gather(): one grouped ncclSend per device / ncclRecv per peer on the root over the direct
//     xGMI links → one kernel on the root puts the bands at their image rows.
// No other data-path collective; the assembled frame is bit-identical to a one-GPU frame because pixels are
// independent and the RNG is a pure function of (column, row, sample) (src/camera.cu:25-28).
//
// RCCL is bound at run time (dlopen "librccl.so.1", the functions by name): the render library itself does not
// depend on it, a process that already carries an RCCL (PyTorch) gets that same copy, and a one-device context
// works without it (its gather is a device-local copy).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/rtp_amd.h"

__attribute__((visibility("hidden"))) void rt_internal_set_error(const std::string &msg);      // rt_capi.hip

namespace {

// ---- the handful of RCCL entry points used, resolved by name (rccl/rccl.h: ncclGetUniqueId … ncclGroupEnd) -----
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;                    // ncclSuccess == 0
constexpr int kNcclFloat32 = 7;              // ncclDataType_t: ncclFloat32 (rccl.h)
struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl &rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
    }
    if (!r.handle) return r;
    auto sym = [&](const char *n) { return dlsym(r.handle, n); };
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.ok = r.CommInitAll && r.CommDestroy && r.Send && r.Recv && r.GroupStart && r.GroupEnd;
    return r;
}

// local row lr of `part` → image row (the inverse of rt_render's compaction, include/rtp_amd.h rt_shard)
__global__ void unshard_kernel(float *frame, const float *rows, int32_t width, int32_t local_rows, int32_t band_rows, int32_t num_parts,
                               int32_t part) {
    const int64_t n = (int64_t)local_rows * width * 3;
    const int64_t row_floats = (int64_t)width * 3;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t lr = k / row_floats, x = k - lr * row_floats;
        const int64_t band = lr / band_rows;
        const int64_t row = (band * num_parts + part) * band_rows + (lr - band * band_rows);
        frame[row * row_floats + x] = rows[k];
    }
}

}  // namespace

struct rt_context {
    std::vector<int> devices;
    std::vector<hipStream_t> streams;
    std::vector<ncclComm_t> comms;            // empty when the transport is "local"
    std::vector<rt_scene *> scenes;
    std::vector<float *> local_rows;          // per device: its rows of the current frame
    std::vector<size_t> local_floats;
    std::vector<float *> staging;             // on the root: what each peer sent (peer 0 = the root's own rows, not staged)
    std::vector<size_t> staging_floats;
    std::vector<hipEvent_t> done;             // per device: its rows are rendered
    std::string transport = "local";
};

namespace {

rt_status mfail(rt_status st, const std::string &msg) {
    rt_internal_set_error(msg);
    return st;
}

#define MHIP(expr)                                                                                          \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return mfail(e_ == hipErrorOutOfMemory ? RT_ERR_OUT_OF_MEMORY : RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

#define MNCCL(expr)                                                                                         \
    do {                                                                                                    \
        ncclResult_t r_ = (expr);                                                                           \
        if (r_ != 0)                                                                                        \
            return mfail(RT_ERR_HIP, std::string(#expr) + ": " + (rccl().GetErrorString ? rccl().GetErrorString(r_) : "RCCL error")); \
    } while (0)

rt_status grow(float *&p, size_t &have, size_t need) {
    if (have >= need) return RT_OK;
    (void)hipFree(p);
    p = nullptr;
    have = 0;
    MHIP(hipMalloc((void **)&p, need * sizeof(float)));
    have = need;
    return RT_OK;
}

}  // namespace

extern "C" {

rt_status rt_context_create(int32_t num_devices, const int32_t *device_ordinals, rt_context **out_ctx) {
    if (!out_ctx) return mfail(RT_ERR_INVALID_ARG, "null argument");
    *out_ctx = nullptr;
    int present = 0;
    if (hipGetDeviceCount(&present) != hipSuccess || present <= 0) return mfail(RT_ERR_NO_DEVICE, "no HIP device");
    if (num_devices <= 0) num_devices = present;           // 0 = every GPU of the node
    if (num_devices > 64) return mfail(RT_ERR_INVALID_ARG, "more than 64 devices");
    rt_context *ctx = new (std::nothrow) rt_context;
    if (!ctx) return mfail(RT_ERR_OUT_OF_MEMORY, "host allocation failed");
    // A device listed more than once (explicit ordinals only): a rehearsal of an N-way shard on fewer GPUs.  RCCL does
    // not admit one GPU twice in a communicator, so such a context moves the rows with device copies ("copy").
    bool duplicates = false;
    if (!device_ordinals && num_devices > present) { delete ctx; return mfail(RT_ERR_NO_DEVICE, "more devices requested than present"); }
    for (int i = 0; i < num_devices; ++i) {
        const int d = device_ordinals ? device_ordinals[i] : i;
        if (d < 0 || d >= present) { delete ctx; return mfail(RT_ERR_NO_DEVICE, "device ordinal out of range"); }
        for (int seen : ctx->devices)
            if (seen == d) duplicates = true;
        ctx->devices.push_back(d);
    }
    const size_t n = ctx->devices.size();
    ctx->streams.assign(n, nullptr);
    ctx->scenes.assign(n, nullptr);
    ctx->local_rows.assign(n, nullptr);
    ctx->local_floats.assign(n, 0);
    ctx->staging.assign(n, nullptr);
    ctx->staging_floats.assign(n, 0);
    ctx->done.assign(n, nullptr);
    auto bail = [&](rt_status st) { rt_context_destroy(ctx); return st; };
    for (size_t i = 0; i < n; ++i) {
        if (hipSetDevice(ctx->devices[i]) != hipSuccess || hipStreamCreateWithFlags(&ctx->streams[i], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->done[i], hipEventDisableTiming) != hipSuccess)
            return bail(mfail(RT_ERR_HIP, "stream/event creation failed"));
    }
    // RCCL communicators: required for more than one device; with one device they are created when RCCL is there (the
    // gather then runs through ncclSend/ncclRecv to itself — the same code path a node with 8 GPUs takes)
    Rccl &r = rccl();
    if (duplicates) {
        ctx->transport = "copy";
    } else if (r.ok) {
        ctx->comms.assign(n, nullptr);
        // (This RCCL build prints a version banner on stdout when the first communicator is made.  A library does not touch
        // the process's file descriptors: a host whose stdout is a data channel — the CLI mirror prints the reference's
        // per-frame TSV there — redirects around this call itself, before it starts any thread: host/render_driver.cpp.)
        const ncclResult_t rc = r.CommInitAll(ctx->comms.data(), (int)n, ctx->devices.data());
        if (rc != 0) {
            ctx->comms.clear();
            if (n > 1) return bail(mfail(RT_ERR_HIP, std::string("ncclCommInitAll: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error")));
        } else {
            ctx->transport = "rccl";
        }
    } else if (n > 1) {
        return bail(mfail(RT_ERR_UNSUPPORTED, "librccl.so.1 not found: a context of more than one device needs RCCL"));
    }
    (void)hipSetDevice(ctx->devices[0]);
    *out_ctx = ctx;
    return RT_OK;
}

rt_status rt_context_destroy(rt_context *ctx) {
    if (!ctx) return RT_OK;
    for (size_t i = 0; i < ctx->devices.size(); ++i) {
        (void)hipSetDevice(ctx->devices[i]);
        if (ctx->streams[i]) (void)hipStreamSynchronize(ctx->streams[i]);
        if (i < ctx->comms.size() && ctx->comms[i]) (void)rccl().CommDestroy(ctx->comms[i]);
        if (ctx->scenes[i]) (void)rt_scene_destroy(ctx->scenes[i]);
        (void)hipFree(ctx->local_rows[i]);
        (void)hipFree(ctx->staging[i]);
        if (ctx->done[i]) (void)hipEventDestroy(ctx->done[i]);
        if (ctx->streams[i]) (void)hipStreamDestroy(ctx->streams[i]);
    }
    if (!ctx->devices.empty()) (void)hipSetDevice(ctx->devices[0]);
    delete ctx;
    return RT_OK;
}

int32_t rt_context_num_devices(const rt_context *ctx) { return ctx ? (int32_t)ctx->devices.size() : 0; }

const char *rt_context_transport(const rt_context *ctx) { return ctx ? ctx->transport.c_str() : ""; }

rt_status rt_context_scene_create(rt_context *ctx, const rt_scene_desc *desc, const rt_config *cfg) {
    if (!ctx || !desc) return mfail(RT_ERR_INVALID_ARG, "null argument");
    for (size_t i = 0; i < ctx->devices.size(); ++i) {
        MHIP(hipSetDevice(ctx->devices[i]));
        if (ctx->scenes[i]) { (void)rt_scene_destroy(ctx->scenes[i]); ctx->scenes[i] = nullptr; }
        const rt_status st = rt_scene_create_ex(desc, cfg, &ctx->scenes[i]);
        if (st != RT_OK) return mfail(st, std::string("device ") + std::to_string(ctx->devices[i]) + ": " + rt_get_last_error_string());
    }
    MHIP(hipSetDevice(ctx->devices[0]));
    return RT_OK;
}

rt_status rt_gather(rt_context *ctx, int32_t image_width, int32_t image_height, int32_t band_rows, float *d_fb_sum_root) {
    if (!ctx || !d_fb_sum_root) return mfail(RT_ERR_INVALID_ARG, "null argument");
    if (image_width <= 0 || image_height <= 0 || band_rows <= 0) return mfail(RT_ERR_INVALID_ARG, "bad frame geometry");
    const int n = (int)ctx->devices.size();
    std::vector<int32_t> rows((size_t)n);
    for (int r = 0; r < n; ++r) {
        const rt_shard sh{band_rows, n, r};
        rows[(size_t)r] = rt_shard_rows(image_height, n > 1 ? &sh : nullptr);
        if ((size_t)rows[(size_t)r] * image_width * 3 > ctx->local_floats[(size_t)r]) return mfail(RT_ERR_INVALID_ARG, "rt_gather before rt_render_sharded of this geometry");
    }
    // Ordering against the caller's own work on d_fb_sum_root: the context's streams are non-blocking streams of its own, so
    // nothing the caller queued on that buffer (a memset on its stream, say) is ordered before the writes below by itself.
    // The root device is drained first — the frame takes tens of milliseconds, this costs microseconds.  (Contract, see
    // include/rtp_amd.h: the caller must not touch d_fb_sum_root from another thread during the call.)
    MHIP(hipSetDevice(ctx->devices[0]));
    MHIP(hipDeviceSynchronize());
    // the root's receive buffers
    const bool via_rccl = !ctx->comms.empty();
    const bool via_copy = !via_rccl && n > 1;
    for (int r = (via_rccl || via_copy) ? 0 : 1; r < n; ++r) {
        const rt_status st = grow(ctx->staging[(size_t)r], ctx->staging_floats[(size_t)r], (size_t)rows[(size_t)r] * image_width * 3);
        if (st != RT_OK) return st;
    }
    // the root's stream waits for every device's rows (events are visible across devices)
    for (int r = 0; r < n; ++r) MHIP(hipStreamWaitEvent(ctx->streams[0], ctx->done[(size_t)r], 0));
    if (via_rccl) {
        Rccl &rc = rccl();
        MNCCL(rc.GroupStart());
        // (a failing call inside the group must not leave the group open: it is closed before the error is returned)
        ncclResult_t bad = 0;
        const char *what = "";
        for (int r = 0; r < n && bad == 0; ++r) {
            const size_t count = (size_t)rows[(size_t)r] * image_width * 3;
            if (count == 0) continue;
            bad = rc.Recv(ctx->staging[(size_t)r], count, kNcclFloat32, r, ctx->comms[0], ctx->streams[0]);
            what = "ncclRecv";
            if (bad == 0) {
                bad = rc.Send(ctx->local_rows[(size_t)r], count, kNcclFloat32, 0, ctx->comms[(size_t)r], ctx->streams[(size_t)r]);
                what = "ncclSend";
            }
        }
        const ncclResult_t ended = rc.GroupEnd();
        if (bad != 0) return mfail(RT_ERR_HIP, std::string(what) + ": " + (rc.GetErrorString ? rc.GetErrorString(bad) : "RCCL error"));
        MNCCL(ended);
    }
    if (via_copy)
        for (int r = 0; r < n; ++r) {
            const size_t bytes = (size_t)rows[(size_t)r] * image_width * 3 * sizeof(float);
            if (bytes) MHIP(hipMemcpyPeerAsync(ctx->staging[(size_t)r], ctx->devices[0], ctx->local_rows[(size_t)r], ctx->devices[(size_t)r], bytes, ctx->streams[0]));
        }
    // bands → image rows, on the root
    MHIP(hipSetDevice(ctx->devices[0]));
    for (int r = 0; r < n; ++r) {
        if (rows[(size_t)r] == 0) continue;
        const float *src = (via_rccl || via_copy) ? ctx->staging[(size_t)r] : ctx->local_rows[(size_t)r];      // ("local": n == 1)
        const int64_t total = (int64_t)rows[(size_t)r] * image_width * 3;
        int64_t blocks = (total + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(unshard_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->streams[0], d_fb_sum_root, src, image_width, rows[(size_t)r],
                           n > 1 ? band_rows : image_height, n, r);
        MHIP(hipGetLastError());
    }
    MHIP(hipStreamSynchronize(ctx->streams[0]));
    if (via_rccl)
        for (int r = 1; r < n; ++r) {       // the senders' streams: their send has completed once the root's receive has
            MHIP(hipSetDevice(ctx->devices[(size_t)r]));
            MHIP(hipStreamSynchronize(ctx->streams[(size_t)r]));
        }
    MHIP(hipSetDevice(ctx->devices[0]));
    return RT_OK;
}

rt_status rt_render_sharded(rt_context *ctx, const rt_camera_data *cam, int32_t band_rows, float *d_fb_sum_root, rt_timing *timings) {
    if (!ctx || !cam || !d_fb_sum_root) return mfail(RT_ERR_INVALID_ARG, "null argument");
    if (band_rows <= 0) band_rows = 8;
    const int n = (int)ctx->devices.size();
    for (int r = 0; r < n; ++r)
        if (!ctx->scenes[(size_t)r]) return mfail(RT_ERR_INVALID_ARG, "rt_context_scene_create first");
    // enqueue every device's shard without waiting: the devices render concurrently
    for (int r = 0; r < n; ++r) {
        MHIP(hipSetDevice(ctx->devices[(size_t)r]));
        const rt_shard sh{band_rows, n, r};
        const int32_t rows = rt_shard_rows(cam->image_height, n > 1 ? &sh : nullptr);
        const size_t floats = (size_t)rows * (size_t)(cam->image_width > 0 ? cam->image_width : 0) * 3;
        rt_status st = grow(ctx->local_rows[(size_t)r], ctx->local_floats[(size_t)r], floats ? floats : 1);
        if (st != RT_OK) return st;
        st = rt_render(ctx->scenes[(size_t)r], cam, n > 1 ? &sh : nullptr, ctx->local_rows[(size_t)r], ctx->streams[(size_t)r], 0, nullptr);
        if (st != RT_OK) return mfail(st, std::string("device ") + std::to_string(ctx->devices[(size_t)r]) + ": " + rt_get_last_error_string());
        MHIP(hipEventRecord(ctx->done[(size_t)r], ctx->streams[(size_t)r]));
    }
    const rt_status st = rt_gather(ctx, cam->image_width, cam->image_height, band_rows, d_fb_sum_root);
    if (st != RT_OK) return st;
    if (timings) {
        // (an array of the CALLER's rt_timing: its first element says how far apart the elements are)
        const uint32_t stride = timings[0].struct_bytes;
        if (stride < 8) return mfail(RT_ERR_INVALID_ARG, "rt_timing.struct_bytes is not set (rt_timing_init)");
        for (int r = 0; r < n; ++r) {
            MHIP(hipSetDevice(ctx->devices[(size_t)r]));
            rt_timing *slot = reinterpret_cast<rt_timing *>(reinterpret_cast<char *>(timings) + (size_t)r * stride);
            slot->struct_bytes = stride;
            const rt_status ts = rt_last_timing(ctx->scenes[(size_t)r], slot);
            if (ts != RT_OK) return mfail(ts, rt_get_last_error_string());
        }
    }
    MHIP(hipSetDevice(ctx->devices[0]));
    return RT_OK;
}

}  // extern "C"
