/*
 * orc_math.h — float helpers of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, loaded by
 * or called from the shipped HIP path; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may use it.
 *
 * Numeric contract shared (by specification, not by code) with the HIP
 * kernels: IEEE-754 binary32, round-to-nearest-even, no fused multiply-add
 * except where fmaf is written out, sums evaluated left to right.  Build with
 * -ffp-contract=off and without -ffast-math.
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float v3_len(v3 a) { return sqrtf(v3_dot(a, a)); }
static inline v3 v3_cross(v3 a, v3 b) {
    return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* GLSL normalize(): component-wise division by the length (0/0 -> NaN is kept). */
static inline v3 v3_normalize(v3 a) {
    float l = v3_len(a);
    return v3_make(a.x / l, a.y / l, a.z / l);
}

/* The shading pass's dot product and matrix products, with fused multiply-adds (fmaf is exact: one rounding per
 * term).  GLSL leaves contraction open; the GPU path evaluates exactly these expressions.  The geometry stage
 * keeps the unfused forms below, which the visibility keys are defined with. */
static inline float v3_dot_fma(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v4 m4_mul_v4_fma(const float* m, v4 v) {
    v4 r;
    r.x = fmaf(m[12], v.w, fmaf(m[8], v.z, fmaf(m[4], v.y, m[0] * v.x)));
    r.y = fmaf(m[13], v.w, fmaf(m[9], v.z, fmaf(m[5], v.y, m[1] * v.x)));
    r.z = fmaf(m[14], v.w, fmaf(m[10], v.z, fmaf(m[6], v.y, m[2] * v.x)));
    r.w = fmaf(m[15], v.w, fmaf(m[11], v.z, fmaf(m[7], v.y, m[3] * v.x)));
    return r;
}
static inline v3 m4_mul_dir_fma(const float* m, v3 v) {
    return v3_make(fmaf(m[8], v.z, fmaf(m[4], v.y, m[0] * v.x)), fmaf(m[9], v.z, fmaf(m[5], v.y, m[1] * v.x)),
                   fmaf(m[10], v.z, fmaf(m[6], v.y, m[2] * v.x)));
}

/* normalize() of the shading pass's per-pixel vectors: reciprocal of the length, then three products (GLSL
 * leaves the rounding of normalize() open; the GPU path does exactly this).  0 * inf -> NaN is kept. */
static inline v3 v3_normalize_rcp(v3 a) {
    float r = 1.0f / v3_len(a);
    return v3_make(a.x * r, a.y * r, a.z * r);
}

/* Column-major 4x4 times column vector: ((c0*x + c1*y) + c2*z) + c3*w per row. */
static inline v4 m4_mul_v4(const float* m, v4 v) {
    v4 r;
    r.x = ((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * v.w;
    r.y = ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * v.w;
    r.z = ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * v.w;
    r.w = ((m[3] * v.x + m[7] * v.y) + m[11] * v.z) + m[15] * v.w;
    return r;
}
/* mat3(m) * v: upper-left 3x3 of a column-major 4x4. */
static inline v3 m4_mul_dir(const float* m, v3 v) {
    v3 r;
    r.x = (m[0] * v.x + m[4] * v.y) + m[8] * v.z;
    r.y = (m[1] * v.x + m[5] * v.y) + m[9] * v.z;
    r.z = (m[2] * v.x + m[6] * v.y) + m[10] * v.z;
    return r;
}
static inline void m4_mul_m4(const float* a, const float* b, float* out) {
    for (int c = 0; c < 4; ++c) {
        v4 col = {b[c * 4 + 0], b[c * 4 + 1], b[c * 4 + 2], b[c * 4 + 3]};
        v4 r = m4_mul_v4(a, col);
        out[c * 4 + 0] = r.x; out[c * 4 + 1] = r.y; out[c * 4 + 2] = r.z; out[c * 4 + 3] = r.w;
    }
}

static inline uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline float maxf3(float a, float b, float c) { return fmaxf(a, fmaxf(b, c)); }

#endif
