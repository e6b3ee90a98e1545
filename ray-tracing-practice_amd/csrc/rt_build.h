// rt_build.h — device-side builder of the guarded walk's traversal tree.
//
// The guarded walk (docs/LOG.md §3b) re-walks every order-sensitive sample on the CALLER's tree, so
// the tree it walks first may be ANY tree whose boxes contain the inflated leaf boxes: the image is
// the same bits.  That frees the choice of builder: this one is an LBVH (63-bit Morton codes of the
// box centres, device radix sort, Karras' parallel hierarchy, bottom-up refit), with the few
// primitives that span a large part of the scene (a ground sphere, a floor quad) kept out of the
// Morton order and chained above the root instead.  Output: the child-pair node table of
// rt_accel.h, resident in device memory.
#pragma once
#include <cstdint>
#include <string>

namespace rtbuild {

struct DeviceTree {
    void *nodes = nullptr;      // float4[4 * num_internal]: fp32 pair records; hipMalloc'ed, the caller frees it
    void *hnodes = nullptr;     // float4[2 * num_internal]: the same records with binary16 planes; likewise
    int32_t num_internal = 0;
    int32_t root = 0;           // node code: >= 0 internal node, < 0 leaf code (single primitive)
    int32_t depth = 0;          // longest root-to-leaf path in internal nodes
    float build_ms = 0.0f;      // device time, hipEvents around sort + kernels
};

// leaf_boxes: n x 6 floats (x.min x.max y.min y.max z.min z.max), leaf_codes: n leaf codes
// (rtaccel::leaf_code); host arrays.  Returns "" on success.
std::string build_lbvh(const float *leaf_boxes, const int32_t *leaf_codes, int32_t n, DeviceTree &out);

}  // namespace rtbuild
