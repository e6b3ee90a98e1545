// host_rng.h — the reference's hash RNG (include/random_utils.h:7-19), host copy used only by the
// benchmark-scene generator so generated scenes are identical on every machine.
#pragma once
#include <cstdint>

namespace rtp {

inline uint32_t wang_hash(uint32_t s) {
    s = (s ^ 61u) ^ (s >> 16);
    s *= 9u;
    s ^= s >> 4;
    s *= 0x27d4eb2du;
    s ^= s >> 15;
    return s;
}

inline float random_float(unsigned &state) {
    state = wang_hash(state);
    return static_cast<float>(state) / 4294967296.0f;
}

}  // namespace rtp
