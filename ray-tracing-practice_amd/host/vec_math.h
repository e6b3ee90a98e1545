// vec_math.h — host-side float vector algebra with the reference's rounding behaviour.
//
// The host mirror must reproduce the reference's geometry bit for bit (same spheres, planes,
// BVH boxes and CameraData), so each helper documents the evaluation order it keeps:
//   * products/sums are plain float, left to right (reference include/vec3.h:76-103);
//   * division by a scalar is "(1.0 / t) * v": reciprocal in double, narrowed to float, then
//     float multiplies (include/vec3.h:53,97) — NOT three float divisions.
// Build with -ffp-contract=off and no -ffast-math.
#pragma once
#include <cmath>
#include "../../include/rtp_amd.h"

namespace rtp {

struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    explicit Vec3(const rt_vec3 &r) : x(r.e[0]), y(r.e[1]), z(r.e[2]) {}
    rt_vec3 pod() const { return rt_vec3{{x, y, z}}; }
    float axis(int a) const { return a == 0 ? x : (a == 1 ? y : z); }
};

inline Vec3 operator+(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator-(Vec3 a) { return {-a.x, -a.y, -a.z}; }
inline Vec3 operator*(float t, Vec3 v) { return {t * v.x, t * v.y, t * v.z}; }
inline Vec3 operator*(Vec3 v, float t) { return t * v; }
inline Vec3 hadamard(Vec3 a, Vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
// (1.0 / t) * v, include/vec3.h:97
inline Vec3 div_scalar(Vec3 v, float t) { return static_cast<float>(1.0 / static_cast<double>(t)) * v; }
inline float dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross(Vec3 a, Vec3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float length_squared(Vec3 v) { return v.x * v.x + v.y * v.y + v.z * v.z; }
inline float length(Vec3 v) { return sqrtf(length_squared(v)); }
inline Vec3 normalized(Vec3 v) { return div_scalar(v, length(v)); }

}  // namespace rtp
