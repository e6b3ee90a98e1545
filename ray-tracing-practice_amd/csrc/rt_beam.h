// rt_beam.h — which leaves can the PRIMARY rays of one pixel reach?  (host + device; double precision)
//
// Every sample of pixel (i, j) starts at the camera origin O and heads for a point of the pixel's footprint,
//     d = p00 + (i + ox) du + (j + oy) dv - O,   |ox|, |oy| <= 0.5          (CameraData::get_ray, include/camera.cuh:97-109)
// so all rays of the pixel lie in a thin cone around the axis a = p00 + i du + j dv - O.  With e = d - a,
// |e| <= h = (|du| + |dv|) / 2 + (rounding of the float evaluation of d), the angle phi between a ray and the axis has
//     sin(phi) = |a x e| / (|a| |d|) <= |e| / |d| <= h / (|a| - h) =: k.
// A point p of such a ray that lies inside a box B is at most |p - O| sin(phi) <= D k away from the axis, D = distance from
// O to the farthest corner of B, and the nearest axis point is in front of O.  Hence: if any ray of the pixel meets B, the
// AXIS RAY meets B grown by rho = D k on every side.  beam_hits_box() tests exactly that (slab test in double, rho rounded up
// by a relative 1e-6 and an absolute term of a few hundred float ulps of the coordinates involved).
//
// What it is for (rt_primary.hip.inc): the guarded near-first walk (docs/LOG.md §3b) rests on ONE geometric fact — a primitive
// for which hit_sphere / hit_plane can return a hit for a ray has its computed hit point inside the primitive's INFLATED leaf
// box (the box the walk's tree is built from).  A primitive whose inflated leaf box no ray of the pixel can meet can therefore
// not be hit by any sample of the pixel; the others are the pixel's candidates, and testing all of them gives the closest
// hit the walk would have found — without walking, once per pixel instead of once per sample.  Boxes of the distance-aware
// mode (Packed::Guard::dyn_k) are grown by dyn_k D^2 like the walk grows them, D being the same distance.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define RT_BEAM_HD __host__ __device__ inline
#else
#define RT_BEAM_HD static inline
#endif

namespace rtbeam {

constexpr int kMaxStack = 64;           // inner nodes waiting during the per-pixel descent; deeper → the pixel has no list
constexpr int32_t kDoneCode = INT32_MIN;

struct Beam {
    double o[3];       // camera origin
    double a[3];       // axis direction (not normalised)
    double inv[3];     // 1 / a (inf where a == 0)
    double k;          // bound on sin(angle between any ray of the pixel and the axis); < 0: no usable cone (everything is a candidate)
    double o_max;      // max |o| component
};

// Magnitude bound M of every intermediate of the float evaluation of d for an image of `width` x `height` pixels.
RT_BEAM_HD double coord_bound(const float org[3], const float p00[3], const float du[3], const float dv[3], int width, int height) {
    double m = 0;
    for (int c = 0; c < 3; ++c) {
        const double v = fabs((double)p00[c]) + (double)width * fabs((double)du[c]) + (double)height * fabs((double)dv[c]) + fabs((double)org[c]);
        m = v > m ? v : m;
    }
    return m;
}

RT_BEAM_HD Beam make_beam(const float org[3], const float p00[3], const float du[3], const float dv[3], int i, int j, double coord_max) {
    Beam b;
    double len2 = 0, ldu = 0, ldv = 0, om = 0;
    for (int c = 0; c < 3; ++c) {
        b.o[c] = (double)org[c];
        b.a[c] = ((double)p00[c] + (double)i * (double)du[c] + (double)j * (double)dv[c]) - (double)org[c];
        b.inv[c] = 1.0 / b.a[c];
        len2 += b.a[c] * b.a[c];
        ldu += (double)du[c] * (double)du[c];
        ldv += (double)dv[c] * (double)dv[c];
        om = fabs(b.o[c]) > om ? fabs(b.o[c]) : om;
    }
    b.o_max = om;
    // the float evaluation: centre = (p00 + i du) + j dv, sample = (centre + ox du) + oy dv, d = sample - O — ten roundings of
    // values below coord_max per component, each at most 2^-24 of it: 64 ulps per component (x sqrt 3 for the length) is generous
    const double h = 0.5 * (sqrt(ldu) + sqrt(ldv)) * (1.0 + 1e-9) + 64.0 * 5.9604644775390625e-8 * 1.7320508075688772 * coord_max;
    const double len = sqrt(len2);
    b.k = (len > 4.0 * h && len < 1e300) ? h / (len - h) : -1.0;       // NaN / inf / a camera sitting on the image plane: no cone
    return b;
}

// May a ray of the pixel meet the box [lo, hi] (grown by grow_k * D^2, D = distance from O to its farthest corner)?
// false only if certainly not.
RT_BEAM_HD bool beam_hits_box(const Beam &b, const float lo[3], const float hi[3], double grow_k) {
    if (!(b.k >= 0.0)) return true;
    double d2 = 0, cm = b.o_max;
    for (int c = 0; c < 3; ++c) {
        const double e0 = fabs((double)lo[c] - b.o[c]), e1 = fabs((double)hi[c] - b.o[c]);
        const double e = e0 > e1 ? e0 : e1;
        d2 += e * e;
        const double m0 = fabs((double)lo[c]), m1 = fabs((double)hi[c]);
        cm = m0 > cm ? m0 : cm;
        cm = m1 > cm ? m1 : cm;
    }
    if (!(d2 < 1e300)) return true;
    const double D = sqrt(d2);
    const double rho = (D * b.k + grow_k * d2) * (1.0 + 1e-6) + 256.0 * 5.9604644775390625e-8 * cm;
    double tmin = 0.0, tmax = 1e308;
    for (int c = 0; c < 3; ++c) {
        const double l = (double)lo[c] - rho, h = (double)hi[c] + rho;
        if (b.a[c] == 0.0) {
            if (b.o[c] < l || b.o[c] > h) return false;
        } else {
            const double t1 = (l - b.o[c]) * b.inv[c], t2 = (h - b.o[c]) * b.inv[c];
            const double tn = t1 < t2 ? t1 : t2, tf = t1 < t2 ? t2 : t1;
            tmin = tn > tmin ? tn : tmin;
            tmax = tf < tmax ? tf : tmax;
        }
    }
    // (1/a carries a relative 2^-53; rho's relative 1e-6 dwarfs it)
    return !(tmax < tmin);
}

// The same question for a SPHERE leaf, asked of the sphere instead of its box.  hit_sphere reports a hit only if its float
// discriminant is >= 0, and then the true distance from the centre to the ray's LINE is at most r + e, e the leaf's margin
// (docs/LOG.md §3b: disc >= -gamma |oc|^2 |d|^2 gives dist^2 <= r^2 + gamma |oc|^2) — the inflated leaf box [c -/+ (r + e)]
// is that ball's bounding cube, so the ball is (centre, half-width) of the box.  Let p be the point of such a ray's line
// nearest to c: |p - c| <= r + e and |p - O| <= |c - O| + r + e =: L; p is at most L sin(phi) <= L k away from the AXIS line.
// Hence: a sphere whose test can report a hit for any ray of the pixel has its centre within (r + e) + L k of the axis line.
// (A cube's corners stick out of its ball by a factor sqrt 3: the strip of pixels between the horizon and the silhouette of a
// huge ground sphere passes the box test and fails this one.)  false only if certainly not.
RT_BEAM_HD bool beam_hits_ball(const Beam &b, const float lo[3], const float hi[3], double grow_k) {
    if (!(b.k >= 0.0)) return true;
    double w[3], r0 = 0.0, w2 = 0.0, a2 = 0.0, wa = 0.0, cm = b.o_max;
    for (int c = 0; c < 3; ++c) {
        const double m = 0.5 * ((double)lo[c] + (double)hi[c]), h = 0.5 * ((double)hi[c] - (double)lo[c]);
        r0 = h > r0 ? h : r0;
        w[c] = m - b.o[c];
        w2 += w[c] * w[c];
        a2 += b.a[c] * b.a[c];
        wa += w[c] * b.a[c];
        const double m0 = fabs((double)lo[c]), m1 = fabs((double)hi[c]);
        cm = m0 > cm ? m0 : cm;
        cm = m1 > cm ? m1 : cm;
    }
    if (!(w2 < 1e300) || !(a2 > 0.0) || !(a2 < 1e300)) return true;
    const double far = sqrt(w2) + 1.7320508075688772 * r0;           // the farthest corner of the cube, as beam_hits_box measures D
    const double ball = r0 + grow_k * far * far;
    const double rho = (ball + (sqrt(w2) + ball) * b.k) * (1.0 + 1e-6) + 256.0 * 5.9604644775390625e-8 * cm;
    // squared distance from the centre to the axis line: |w|^2 - (w . a)^2 / |a|^2, compared without the division
    const double lhs = (w2 * a2 - wa * wa) * (1.0 - 1e-9);           // (cancellation: the subtraction loses relative 1e-16 of w2 a2)
    return !(lhs > rho * rho * a2 + 1e-12 * w2 * a2);
}

// Descent of the guarded walk's child-pair table (rt_accel.h: 16 floats per inner node — lo0.xyz hi0.xyz lo1.xyz hi1.xyz,
// code0, code1; code >= 0 inner node, < 0 leaf = -(2 index + type) - 1) with the pixel's cone: writes the leaf codes of the
// candidates (as 2 index + type) to out[0 .. max_out) and returns their number, or -1 when there are more than max_out or
// the descent needs more than kMaxStack waiting nodes — the pixel then has no list and its samples walk the tree themselves.
template <class NodeTab>
RT_BEAM_HD int beam_candidates(const Beam &b, NodeTab nodes, int32_t root, double grow_k, uint32_t *out, int max_out) {
    if (root == kDoneCode) return 0;
    int n = 0;
    if (root < 0) {                 // a tree of one primitive
        if (max_out < 1) return -1;
        out[0] = (uint32_t)(-(root + 1));
        return 1;
    }
    int32_t stack[kMaxStack];
    int sp = 0;
    int32_t node = root;
    for (;;) {
        const float *r = nodes(node);
        int32_t next = kDoneCode;
        for (int c = 0; c < 2; ++c) {
            union { float f; int32_t i; } code;
            code.f = r[12 + c];
            if (code.i == kDoneCode) continue;
            const float lo[3] = {r[6 * c + 0], r[6 * c + 1], r[6 * c + 2]}, hi[3] = {r[6 * c + 3], r[6 * c + 4], r[6 * c + 5]};
            if (!beam_hits_box(b, lo, hi, grow_k)) continue;
            if (code.i < 0) {
                const uint32_t leaf = (uint32_t)(-(code.i + 1));
                if (!(leaf & 1u) && !beam_hits_ball(b, lo, hi, grow_k)) continue;      // a sphere the cone passes by
                if (n >= max_out) return -1;
                out[n++] = leaf;
            } else if (next == kDoneCode) {
                next = code.i;
            } else {
                if (sp >= kMaxStack) return -1;
                stack[sp++] = code.i;
            }
        }
        if (next != kDoneCode) node = next;
        else if (sp > 0) node = stack[--sp];
        else break;
    }
    return n;
}

// The front primitives of the guarded walk (rt_accel.h, Packed::Guard: scene-spanning primitives that are not leaves of the
// tree) join a pixel's list by the same two tests a leaf passes in the descent above.  codes: 2 index + type; boxes: 6 floats
// each, x.min x.max y.min y.max z.min z.max (the inflated leaf boxes).  n: candidates already in out (-1 stays -1).
RT_BEAM_HD int beam_front_candidates(const Beam &b, int num_front, const int32_t *codes, const float *boxes, double grow_k, uint32_t *out, int n, int max_out) {
    for (int f = 0; f < num_front && n >= 0; ++f) {
        const float *x = boxes + 6 * f;
        const float lo[3] = {x[0], x[2], x[4]}, hi[3] = {x[1], x[3], x[5]};
        if (!beam_hits_box(b, lo, hi, grow_k)) continue;
        if (!(codes[f] & 1) && !beam_hits_ball(b, lo, hi, grow_k)) continue;
        if (n >= max_out) return -1;
        out[n++] = (uint32_t)codes[f];
    }
    return n;
}

}  // namespace rtbeam
