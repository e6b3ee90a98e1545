// rt_build.hip — device LBVH builder for the guarded walk's tree (see rt_build.h).
#include "rt_build.h"

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

namespace rtbuild {
namespace {

constexpr int kMaxLarge = 16;      // primitives chained above the LBVH root at most

struct Box { float v[6]; };       // x.min x.max y.min y.max z.min z.max

__device__ __forceinline__ Box box_union(const Box &a, const Box &b) {
    Box r;
    for (int k = 0; k < 3; ++k) {
        r.v[2 * k] = fminf(a.v[2 * k], b.v[2 * k]);
        r.v[2 * k + 1] = fmaxf(a.v[2 * k + 1], b.v[2 * k + 1]);
    }
    return r;
}

// 21 bits → every third bit of a 63-bit word
__device__ __forceinline__ uint64_t spread3(uint32_t x) {
    uint64_t v = x & 0x1fffffu;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

struct MortonFrame { float lo[3]; float scale[3]; };

__global__ void morton_kernel(const Box *boxes, int32_t first, int32_t m, MortonFrame f, uint64_t *keys, uint32_t *vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const Box b = boxes[first + i];
    uint32_t q[3];
    for (int a = 0; a < 3; ++a) {
        const float c = 0.5f * (b.v[2 * a] + b.v[2 * a + 1]);
        float t = (c - f.lo[a]) * f.scale[a];
        t = fminf(fmaxf(t, 0.0f), 2097151.0f);
        q[a] = (uint32_t)t;
    }
    keys[i] = (spread3(q[0]) << 2) | (spread3(q[1]) << 1) | spread3(q[2]);
    vals[i] = (uint32_t)i;
}

// common-prefix length of the keys at sorted positions i and j; equal keys are told apart by position
__device__ __forceinline__ int delta(const uint64_t *keys, int m, int i, int j) {
    if (j < 0 || j >= m) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned)(i ^ j));
    return __clzll((long long)(a ^ b));
}

// Karras 2012: internal node i of m - 1; children >= 0 → internal index, < 0 → ~(sorted leaf position)
__global__ void hierarchy_kernel(const uint64_t *keys, int32_t m, int32_t *left, int32_t *right, int32_t *parent_internal,
                                 int32_t *parent_leaf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m - 1) return;
    const int d = (delta(keys, m, i, i + 1) - delta(keys, m, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, m, i, i - d);
    int lmax = 2;
    while (delta(keys, m, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, m, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, m, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) / 2;
        if (delta(keys, m, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int32_t lc = (lo == gamma) ? ~gamma : gamma;
    const int32_t rc = (hi == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    left[i] = lc;
    right[i] = rc;
    if (lc >= 0) parent_internal[lc] = i; else parent_leaf[~lc] = i;
    if (rc >= 0) parent_internal[rc] = i; else parent_leaf[~rc] = i;
    if (i == 0) parent_internal[0] = -1;
}

// bottom-up: the second thread to reach a node computes its box and height and goes on
__global__ void refit_kernel(const Box *boxes, int32_t first, const uint32_t *vals, int32_t m, const int32_t *left, const int32_t *right,
                             const int32_t *parent_internal, const int32_t *parent_leaf, Box *node_box, int32_t *height,
                             uint32_t *arrived) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    int32_t node = parent_leaf[p];
    while (node >= 0) {
        __threadfence();
        if (atomicAdd(&arrived[node], 1u) == 0u) return;      // the sibling subtree is not done yet
        __threadfence();
        const int32_t lc = left[node], rc = right[node];
        const Box lb = lc >= 0 ? node_box[lc] : boxes[first + vals[~lc]];
        const Box rb = rc >= 0 ? node_box[rc] : boxes[first + vals[~rc]];
        const int32_t lh = lc >= 0 ? height[lc] : 0, rh = rc >= 0 ? height[rc] : 0;
        node_box[node] = box_union(lb, rb);
        height[node] = 1 + (lh > rh ? lh : rh);
        node = parent_internal[node];
    }
}

// the child-pair records the guarded walk reads (rt_accel.cpp): fp32 planes (64 B, LDS-resident scenes) and the
// same with 12 binary16 planes rounded OUTWARD (32 B, scenes read through L1/L2), two child codes each
__device__ __forceinline__ void write_pair(float4 *nodes, float4 *hnodes, int32_t k, const Box &b0, int32_t c0, const Box &b1, int32_t c1) {
    nodes[4 * k + 0] = make_float4(b0.v[0], b0.v[2], b0.v[4], b0.v[1]);      // lo0.xyz hi0.x
    nodes[4 * k + 1] = make_float4(b0.v[3], b0.v[5], b1.v[0], b1.v[2]);      // hi0.yz lo1.xy
    nodes[4 * k + 2] = make_float4(b1.v[4], b1.v[1], b1.v[3], b1.v[5]);      // lo1.z hi1.xyz
    nodes[4 * k + 3] = make_float4(__int_as_float(c0), __int_as_float(c1), 0.0f, 0.0f);
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    typedef _Float16 half4 __attribute__((ext_vector_type(4)));
    half8 a;
    half4 b;
    const __half h[12] = {__float2half_rd(b0.v[0]), __float2half_ru(b0.v[1]), __float2half_rd(b0.v[2]), __float2half_ru(b0.v[3]),
                          __float2half_rd(b0.v[4]), __float2half_ru(b0.v[5]), __float2half_rd(b1.v[0]), __float2half_ru(b1.v[1]),
                          __float2half_rd(b1.v[2]), __float2half_ru(b1.v[3]), __float2half_rd(b1.v[4]), __float2half_ru(b1.v[5])};
    for (int i = 0; i < 8; ++i) a[i] = __builtin_bit_cast(_Float16, h[i]);
    for (int i = 0; i < 4; ++i) b[i] = __builtin_bit_cast(_Float16, h[8 + i]);
    const float2 bxy = __builtin_bit_cast(float2, b);
    hnodes[2 * k + 0] = __builtin_bit_cast(float4, a);
    hnodes[2 * k + 1] = make_float4(bxy.x, bxy.y, __int_as_float(c0), __int_as_float(c1));
}

__global__ void emit_kernel(const Box *boxes, const int32_t *codes, int32_t first, const uint32_t *vals, int32_t m, const int32_t *left,
                            const int32_t *right, const Box *node_box, float4 *nodes, float4 *hnodes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m - 1) return;
    const int32_t lc = left[i], rc = right[i];
    const Box lb = lc >= 0 ? node_box[lc] : boxes[first + vals[~lc]];
    const Box rb = rc >= 0 ? node_box[rc] : boxes[first + vals[~rc]];
    const int32_t c0 = lc >= 0 ? lc : codes[first + vals[~lc]];
    const int32_t c1 = rc >= 0 ? rc : codes[first + vals[~rc]];
    write_pair(nodes, hnodes, i, lb, c0, rb, c1);
}

struct Summary { int32_t root, num_internal, depth, pad; };

// one thread: the large primitives [0, num_large) become a chain above the LBVH root
__global__ void chain_kernel(const Box *boxes, const int32_t *codes, int32_t num_large, int32_t m, const Box *node_box,
                             const int32_t *height, float4 *nodes, float4 *hnodes, Summary *out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    bool have = false;
    int32_t sub_code = 0, depth = 0, next = m >= 2 ? m - 1 : 0;
    Box sub_box;
    if (m >= 2) { have = true; sub_code = 0; sub_box = node_box[0]; depth = height[0]; }
    else if (m == 1) { have = true; sub_code = codes[num_large]; sub_box = boxes[num_large]; depth = 0; }
    for (int32_t j = 0; j < num_large; ++j) {
        if (!have) { have = true; sub_code = codes[j]; sub_box = boxes[j]; depth = 0; continue; }
        write_pair(nodes, hnodes, next, boxes[j], codes[j], sub_box, sub_code);
        sub_box = box_union(boxes[j], sub_box);
        sub_code = next++;
        depth++;
    }
    out->root = sub_code;
    out->num_internal = next;
    out->depth = depth;
}

#define BUILD_TRY(expr)                                                                         \
    do {                                                                                        \
        const hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); goto done; } \
    } while (0)

}  // namespace

std::string build_lbvh(const float *leaf_boxes, const int32_t *leaf_codes, int32_t n, DeviceTree &out) {
    out = DeviceTree{};
    if (n <= 0) return "no primitives";
    // ---- host: which primitives are "large" (an extent above a quarter of the scene's), at most kMaxLarge,
    // moved to the front; Morton frame over the centres of the others
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int32_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], leaf_boxes[6 * (size_t)i + 2 * a]);
            hi[a] = std::max(hi[a], leaf_boxes[6 * (size_t)i + 2 * a + 1]);
        }
    const float scene_extent = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
    auto extent_of = [&](int32_t i) {
        const float *b = leaf_boxes + 6 * (size_t)i;
        return std::max(b[1] - b[0], std::max(b[3] - b[2], b[5] - b[4]));
    };
    std::vector<int32_t> order((size_t)n);
    std::iota(order.begin(), order.end(), 0);
    int32_t num_large = 0;
    if (n > 2) {
        std::vector<int32_t> big;
        for (int32_t i = 0; i < n; ++i)
            if (extent_of(i) > 0.25f * scene_extent) big.push_back(i);
        std::sort(big.begin(), big.end(), [&](int32_t a, int32_t b) { return extent_of(a) > extent_of(b); });
        if ((int)big.size() > kMaxLarge) big.resize(kMaxLarge);
        if ((int32_t)big.size() < n) {
            std::vector<char> is_big((size_t)n, 0);
            for (int32_t i : big) is_big[(size_t)i] = 1;
            order.clear();
            for (int32_t i : big) order.push_back(i);
            for (int32_t i = 0; i < n; ++i) if (!is_big[(size_t)i]) order.push_back(i);
            num_large = (int32_t)big.size();
        }
    }
    const int32_t m = n - num_large;
    std::vector<Box> h_boxes((size_t)n);
    std::vector<int32_t> h_codes((size_t)n);
    MortonFrame frame;
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int32_t k = 0; k < n; ++k) {
        const int32_t i = order[(size_t)k];
        std::memcpy(h_boxes[(size_t)k].v, leaf_boxes + 6 * (size_t)i, sizeof(float) * 6);
        h_codes[(size_t)k] = leaf_codes[i];
        if (k >= num_large)
            for (int a = 0; a < 3; ++a) {
                const float c = 0.5f * (h_boxes[(size_t)k].v[2 * a] + h_boxes[(size_t)k].v[2 * a + 1]);
                clo[a] = std::min(clo[a], c);
                chi[a] = std::max(chi[a], c);
            }
    }
    for (int a = 0; a < 3; ++a) {
        frame.lo[a] = m > 0 ? clo[a] : 0.0f;
        const float ext = m > 0 ? chi[a] - clo[a] : 0.0f;
        frame.scale[a] = ext > 0.0f ? 2097151.0f / ext : 0.0f;
    }

    std::string err;
    Box *d_boxes = nullptr, *d_node_box = nullptr;
    int32_t *d_codes = nullptr, *d_left = nullptr, *d_right = nullptr, *d_parent_i = nullptr, *d_parent_l = nullptr, *d_height = nullptr;
    uint64_t *d_keys = nullptr, *d_keys_sorted = nullptr;
    uint32_t *d_vals = nullptr, *d_vals_sorted = nullptr, *d_arrived = nullptr;
    void *d_temp = nullptr;
    size_t temp_bytes = 0;
    float4 *d_nodes = nullptr, *d_hnodes = nullptr;
    Summary *d_summary = nullptr;
    Summary summary{};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const int32_t max_internal = (m >= 2 ? m - 1 : 0) + num_large;
    const int32_t mm = m > 0 ? m : 1;
    const int threads = 256;

    BUILD_TRY(hipMalloc((void **)&d_boxes, sizeof(Box) * (size_t)n));
    BUILD_TRY(hipMalloc((void **)&d_codes, sizeof(int32_t) * (size_t)n));
    BUILD_TRY(hipMemcpy(d_boxes, h_boxes.data(), sizeof(Box) * (size_t)n, hipMemcpyHostToDevice));
    BUILD_TRY(hipMemcpy(d_codes, h_codes.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
    BUILD_TRY(hipMalloc((void **)&d_keys, 8 * (size_t)mm));
    BUILD_TRY(hipMalloc((void **)&d_keys_sorted, 8 * (size_t)mm));
    BUILD_TRY(hipMalloc((void **)&d_vals, 4 * (size_t)mm));
    BUILD_TRY(hipMalloc((void **)&d_vals_sorted, 4 * (size_t)mm));
    BUILD_TRY(hipMalloc((void **)&d_left, 4 * (size_t)mm));
    BUILD_TRY(hipMalloc((void **)&d_right, 4 * (size_t)mm));
    BUILD_TRY(hipMalloc((void **)&d_parent_i, 4 * (size_t)mm));
    BUILD_TRY(hipMalloc((void **)&d_parent_l, 4 * (size_t)mm));
    BUILD_TRY(hipMalloc((void **)&d_height, 4 * (size_t)mm));
    BUILD_TRY(hipMalloc((void **)&d_arrived, 4 * (size_t)mm));
    BUILD_TRY(hipMalloc((void **)&d_node_box, sizeof(Box) * (size_t)mm));
    BUILD_TRY(hipMalloc((void **)&d_nodes, 64 * (size_t)(max_internal > 0 ? max_internal : 1)));
    BUILD_TRY(hipMalloc((void **)&d_hnodes, 32 * (size_t)(max_internal > 0 ? max_internal : 1)));
    BUILD_TRY(hipMalloc((void **)&d_summary, sizeof(Summary)));
    if (m > 1) {
        BUILD_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, d_keys, d_keys_sorted, d_vals, d_vals_sorted, m, 0, 63));
        BUILD_TRY(hipMalloc(&d_temp, temp_bytes > 0 ? temp_bytes : 16));
    }
    BUILD_TRY(hipEventCreate(&ev0));
    BUILD_TRY(hipEventCreate(&ev1));
    BUILD_TRY(hipEventRecord(ev0, 0));
    if (m > 1) {
        hipLaunchKernelGGL(morton_kernel, dim3((m + threads - 1) / threads), dim3(threads), 0, 0, d_boxes, num_large, m, frame, d_keys, d_vals);
        BUILD_TRY(hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, d_keys, d_keys_sorted, d_vals, d_vals_sorted, m, 0, 63));
        BUILD_TRY(hipMemsetAsync(d_arrived, 0, 4 * (size_t)m, 0));
        hipLaunchKernelGGL(hierarchy_kernel, dim3((m + threads - 1) / threads), dim3(threads), 0, 0, d_keys_sorted, m, d_left, d_right,
                           d_parent_i, d_parent_l);
        hipLaunchKernelGGL(refit_kernel, dim3((m + threads - 1) / threads), dim3(threads), 0, 0, d_boxes, num_large, d_vals_sorted, m, d_left,
                           d_right, d_parent_i, d_parent_l, d_node_box, d_height, d_arrived);
        hipLaunchKernelGGL(emit_kernel, dim3((m + threads - 1) / threads), dim3(threads), 0, 0, d_boxes, d_codes, num_large, d_vals_sorted, m,
                           d_left, d_right, d_node_box, d_nodes, d_hnodes);
    }
    hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, 0, d_boxes, d_codes, num_large, m, d_node_box, d_height, d_nodes, d_hnodes, d_summary);
    BUILD_TRY(hipGetLastError());
    BUILD_TRY(hipEventRecord(ev1, 0));
    BUILD_TRY(hipEventSynchronize(ev1));
    BUILD_TRY(hipEventElapsedTime(&out.build_ms, ev0, ev1));
    BUILD_TRY(hipMemcpy(&summary, d_summary, sizeof(summary), hipMemcpyDeviceToHost));
    out.nodes = d_nodes;
    d_nodes = nullptr;
    out.hnodes = d_hnodes;
    d_hnodes = nullptr;
    out.num_internal = summary.num_internal;
    out.root = summary.root;
    out.depth = summary.depth;

done:
    (void)hipFree(d_boxes); (void)hipFree(d_codes); (void)hipFree(d_keys); (void)hipFree(d_keys_sorted); (void)hipFree(d_vals);
    (void)hipFree(d_vals_sorted); (void)hipFree(d_left); (void)hipFree(d_right); (void)hipFree(d_parent_i); (void)hipFree(d_parent_l);
    (void)hipFree(d_height); (void)hipFree(d_arrived); (void)hipFree(d_node_box); (void)hipFree(d_temp); (void)hipFree(d_nodes); (void)hipFree(d_hnodes);
    (void)hipFree(d_summary);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    return err;
}

}  // namespace rtbuild
