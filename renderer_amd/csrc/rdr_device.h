// rdr_device.h — device-side math and the software texture sampler.
//
// Numeric contract of every kernel that decides which triangle owns a pixel:
// IEEE binary32, no fused multiply-add except where fmaf is spelled out, sums
// left to right, correctly rounded division and square root.  The library is
// compiled with -ffp-contract=off so that the expressions below mean exactly
// what they say.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rdr_internal.h"

#define RDR_DEV __device__ __forceinline__

struct vec3 {
    float x, y, z;
};
struct vec4 {
    float x, y, z, w;
};

RDR_DEV vec3 mk3(float x, float y, float z) { return vec3{x, y, z}; }
RDR_DEV vec4 mk4(float x, float y, float z, float w) { return vec4{x, y, z, w}; }
RDR_DEV vec3 operator+(vec3 a, vec3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RDR_DEV vec3 operator-(vec3 a, vec3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RDR_DEV vec3 operator*(vec3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RDR_DEV vec3 operator*(vec3 a, vec3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RDR_DEV float dot3(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RDR_DEV float len3(vec3 a) { return sqrtf(dot3(a, a)); }
RDR_DEV vec3 cross3(vec3 a, vec3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// GLSL normalize(): divide by the length; 0/0 stays NaN.
RDR_DEV vec3 normalize3(vec3 a) {
    float l = len3(a);
    return mk3(a.x / l, a.y / l, a.z / l);
}
// normalize() of the per-pixel shading vectors: one correctly rounded reciprocal of the length and three
// products.  GLSL leaves normalize()'s rounding to the implementation (drivers lower it to
// x * inversesqrt(dot(x, x))), and two of three divisions are a tenth of the shading pass's arithmetic.
// A zero vector still gives NaN (0 * inf).
RDR_DEV vec3 normalize3_rcp(vec3 a) {
    const float r = 1.0f / len3(a);
    return mk3(a.x * r, a.y * r, a.z * r);
}
RDR_DEV float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
RDR_DEV float max3f(float a, float b, float c) { return fmaxf(a, fmaxf(b, c)); }

// Column-major mat4 * vec4: ((c0*x + c1*y) + c2*z) + c3*w per row.
RDR_DEV vec4 mul_m4(const float* m, vec4 v) {
    vec4 r;
    r.x = ((m[0] * v.x + m[4] * v.y) + m[8] * v.z) + m[12] * v.w;
    r.y = ((m[1] * v.x + m[5] * v.y) + m[9] * v.z) + m[13] * v.w;
    r.z = ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * v.w;
    r.w = ((m[3] * v.x + m[7] * v.y) + m[11] * v.z) + m[15] * v.w;
    return r;
}
// mat3(m) * v
RDR_DEV vec3 mul_m3(const float* m, vec3 v) {
    return mk3((m[0] * v.x + m[4] * v.y) + m[8] * v.z, (m[1] * v.x + m[5] * v.y) + m[9] * v.z,
               (m[2] * v.x + m[6] * v.y) + m[10] * v.z);
}

// The list passes' forms of the three products above, with fused multiply-adds: one rounding per term instead of
// two, and four instructions per row instead of seven.  GLSL leaves contraction to the compiler; the rasterisers
// keep the unfused forms, which the bit-exact visibility keys are defined with.  The CPU checker evaluates the same
// expressions with fmaf().
RDR_DEV float dot3_fma(vec3 a, vec3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
RDR_DEV vec4 mul_m4_fma(const float* m, vec4 v) {
    vec4 r;
    r.x = fmaf(m[12], v.w, fmaf(m[8], v.z, fmaf(m[4], v.y, m[0] * v.x)));
    r.y = fmaf(m[13], v.w, fmaf(m[9], v.z, fmaf(m[5], v.y, m[1] * v.x)));
    r.z = fmaf(m[14], v.w, fmaf(m[10], v.z, fmaf(m[6], v.y, m[2] * v.x)));
    r.w = fmaf(m[15], v.w, fmaf(m[11], v.z, fmaf(m[7], v.y, m[3] * v.x)));
    return r;
}
RDR_DEV vec3 mul_m3_fma(const float* m, vec3 v) {
    return mk3(fmaf(m[8], v.z, fmaf(m[4], v.y, m[0] * v.x)), fmaf(m[9], v.z, fmaf(m[5], v.y, m[1] * v.x)),
               fmaf(m[10], v.z, fmaf(m[6], v.y, m[2] * v.x)));
}

// Load a 4x4 matrix whose address is the same in every lane of the wave.
RDR_DEV void load_m4(const float* p, float* m) {
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = p[i];
}

// A load through the constant address space, for data that no kernel of the running launch writes: at an address
// that is the same in every lane the compiler uses the scalar unit (result in scalar registers, a counter of its
// own, nothing on the vector memory path).
template <typename T>
RDR_DEV T load_uniform(const T* p) {
    return *(const __attribute__((address_space(4))) T*)(p);
}
// The same for a record that is read field by field: a reference through which every field the lanes agree on the
// address of is a scalar load (and an ordinary vector load where they do not).
template <typename T>
RDR_DEV const __attribute__((address_space(4))) T& uniform_ref(const T* p) {
    return *(const __attribute__((address_space(4))) T*)(p);
}
// body(k) once for every distinct value k of `key` among the wave's active lanes, with the lanes that hold k active
// and k in a scalar register: what body reads at addresses derived from k goes through the scalar unit (uniform_ref,
// load_uniform).  The loop itself is uniform (a scalar mask of the lanes still to serve; the branch is inside it):
// written as a loop that every lane leaves after its own turn, the compiler rotates it into "wait for your turn,
// then do the work once behind the loop", where all lanes are back and the scalar value is the first lane's.  The
// comparison goes through an opaque copy of k, because inside the branch the compiler would otherwise put the lane's
// own key, which it knows to be equal, in k's place, and every address would be a vector again.
template <typename F>
RDR_DEV void for_each_uniform(uint32_t key, F&& body) {
    unsigned long long pending = __ballot(true);
    while (pending) {
        const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, __ffsll((long long)pending) - 1);
        uint32_t k_compare = k;
        asm volatile("" : "+s"(k_compare));
        const bool mine = key == k_compare;
        if (mine) body(k);
        pending &= ~__ballot(mine);
    }
}

RDR_DEV void load_m4_uniform(const float* p, float* m) {
#pragma unroll
    for (int i = 0; i < 16; ++i) m[i] = load_uniform(p + i);
}
// Four bytes at any byte address as one load (global memory takes unaligned dword accesses).
RDR_DEV uint32_t load_u32_unaligned(const uint8_t* p) {
    uint32_t w;
    __builtin_memcpy(&w, p, 4);
    return w;
}

RDR_DEV float model_max_scale(const float* m) {
    float sx = len3(mk3(m[0], m[1], m[2]));
    float sy = len3(mk3(m[4], m[5], m[6]));
    float sz = len3(mk3(m[8], m[9], m[10]));
    return fmaxf(sx, fmaxf(sy, sz));
}

// ---- wave64 helpers ----------------------------------------------------------
RDR_DEV uint32_t lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
// Number of set bits of a wave ballot below this lane.
RDR_DEV uint32_t ballot_rank(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
RDR_DEV uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
RDR_DEV unsigned long long wave_sum64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Inclusive prefix sum across the wave.
RDR_DEV uint32_t wave_scan_inclusive(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o, 64);
        if (lane >= (uint32_t)o) v += t;
    }
    return v;
}

// ---- texture sampler -----------------------------------------------------------
// One sampler for every texture: LINEAR mag/min/mip, REPEAT, LOD 0..16
// (reference: renderer/src/renderer.rs:755-763).

RDR_DEV uint32_t mip_dim(uint32_t d, uint32_t level) {
    uint32_t v = d >> level;
    return v ? v : 1u;
}

// b / 255.0f for a byte value, bit-exact, without the IEEE division sequence: one
// Newton step on b * fl(1/255) is correctly rounded for all 256 inputs (rdr_create
// checks this on the host, whose fmaf rounds like the device's).
RDR_DEV float unorm8(uint32_t b) {
    const float c = 0x1.010102p-8f; // fl(1 / 255)
    const float x = (float)b, q = x * c;
    return fmaf(fmaf(-255.0f, q, x), c, q);
}

RDR_DEV vec4 decode_texel(const uint8_t* p, uint32_t format, const float* srgb_lut) {
    if (format == RDR_FORMAT_RGBA32F) {
        const float4 f = *reinterpret_cast<const float4*>(p);
        return mk4(f.x, f.y, f.z, f.w);
    }
    const uint32_t t = *reinterpret_cast<const uint32_t*>(p);
    const uint32_t r = t & 255u, g = (t >> 8) & 255u, b = (t >> 16) & 255u, a = t >> 24;
    if (format == RDR_FORMAT_RGBA8_SRGB)
        return mk4(srgb_lut[r], srgb_lut[g], srgb_lut[b], unorm8(a));
    return mk4(unorm8(r), unorm8(g), unorm8(b), unorm8(a));
}

// a + t (b - a), the product and the sum fused: two instructions per channel
RDR_DEV vec4 lerp4(vec4 a, vec4 b, float t) {
    return mk4(fmaf(t, b.x - a.x, a.x), fmaf(t, b.y - a.y, a.y), fmaf(t, b.z - a.z, a.z), fmaf(t, b.w - a.w, a.w));
}

RDR_DEV int wrap_repeat(int i, int n) {
    if ((n & (n - 1)) == 0) return i & (n - 1); // power-of-two extent: the mask is the non-negative remainder
    int m = i % n;
    return m < 0 ? m + n : m;
}

// `lut`: 256 sRGB -> linear values; with `lut_linear` 256 more behind them, unorm8() of every byte (a table in LDS
// then takes the place of three vector instructions per channel)
RDR_DEV vec4 decode_rgba8(uint32_t t, uint32_t format, const float* lut, bool lut_linear = false) {
    const uint32_t r = t & 255u, g = (t >> 8) & 255u, b = (t >> 16) & 255u, a = t >> 24;
    if (format == RDR_FORMAT_RGBA8_SRGB)
        return mk4(lut[r], lut[g], lut[b], lut_linear ? lut[256u + a] : unorm8(a));
    if (lut_linear) return mk4(lut[256u + r], lut[256u + g], lut[256u + b], lut[256u + a]);
    return mk4(unorm8(r), unorm8(g), unorm8(b), unorm8(a));
}

// Two horizontally adjacent 4-byte texels with one 8-byte load (the address is only 4-byte aligned, which the
// memory path accepts).  At the last column the second word belongs to the next row, or to the slack behind the
// texel blob: the caller replaces it.
struct TexelPair {
    uint32_t a, b;
};
RDR_DEV TexelPair load_texel_pair(const uint8_t* p) {
    typedef uint32_t __attribute__((ext_vector_type(2), aligned(4))) u32x2_a4;
    const u32x2_a4 v = *reinterpret_cast<const u32x2_a4*>(p);
    return TexelPair{v.x, v.y};
}

template <typename Image> // GpuImage, or GpuImage in the constant address space (uniform_ref)
RDR_DEV vec4 image_bilinear(const Image& im, const uint8_t* texels, const float* lut, uint32_t level,
                            float u, float v, bool lut_linear = false) {
    const int w = (int)mip_dim(im.width, level), h = (int)mip_dim(im.height, level);
    const uint8_t* base = texels + im.offset + im.level_offset[level];
    const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float ax = x - fx, ay = y - fy;
    const int x0 = wrap_repeat((int)fx, w), x1 = wrap_repeat((int)fx + 1, w);
    const int y0 = wrap_repeat((int)fy, h), y1 = wrap_repeat((int)fy + 1, h);
    vec4 t00, t10, t01, t11;
    if (im.format == RDR_FORMAT_RGBA32F) {
        // texel indices fit 32 bits (a level has at most 16 384 x 16 384 texels); only the byte offset needs 64
        t00 = decode_texel(base + ((size_t)(uint32_t)(y0 * w + x0) << 4), im.format, lut);
        t10 = decode_texel(base + ((size_t)(uint32_t)(y0 * w + x1) << 4), im.format, lut);
        t01 = decode_texel(base + ((size_t)(uint32_t)(y1 * w + x0) << 4), im.format, lut);
        t11 = decode_texel(base + ((size_t)(uint32_t)(y1 * w + x1) << 4), im.format, lut);
    } else {
        // a footprint's two texels of a row are neighbours in memory unless it wraps around the last column:
        // four loads per tap become two, and the address unit is what the textured passes keep busiest
        TexelPair r0 = load_texel_pair(base + ((size_t)(uint32_t)(y0 * w + x0) << 2));
        TexelPair r1 = load_texel_pair(base + ((size_t)(uint32_t)(y1 * w + x0) << 2));
        if (x1 != x0 + 1) {
            r0.b = *reinterpret_cast<const uint32_t*>(base + ((size_t)(uint32_t)(y0 * w + x1) << 2));
            r1.b = *reinterpret_cast<const uint32_t*>(base + ((size_t)(uint32_t)(y1 * w + x1) << 2));
        }
        t00 = decode_rgba8(r0.a, im.format, lut, lut_linear); t10 = decode_rgba8(r0.b, im.format, lut, lut_linear);
        t01 = decode_rgba8(r1.a, im.format, lut, lut_linear); t11 = decode_rgba8(r1.b, im.format, lut, lut_linear);
    }
    return lerp4(lerp4(t00, t10, ax), lerp4(t01, t11, ax), ay);
}

template <typename Image> // GpuImage, or GpuImage in the constant address space (uniform_ref)
RDR_DEV vec4 image_sample_lod(const Image& im, const uint8_t* texels, const float* lut, float u, float v,
                              float lod, bool lut_linear = false) {
    float maxl = (float)(im.mip_count - 1);
    if (maxl > 16.0f) maxl = 16.0f;
    if (!(lod > 0.0f)) lod = 0.0f;
    if (lod > maxl) lod = maxl;
    const float fl = floorf(lod);
    const uint32_t l0 = (uint32_t)fl;
    const uint32_t l1 = l0 + 1 < im.mip_count ? l0 + 1 : l0;
    const float t = lod - fl;
    const vec4 a = image_bilinear(im, texels, lut, l0, u, v, lut_linear);
    if (l1 == l0 || t == 0.0f) return a;
    const vec4 b = image_bilinear(im, texels, lut, l1, u, v, lut_linear);
    return lerp4(a, b, t);
}

template <typename Image> // GpuImage, or GpuImage in the constant address space (uniform_ref)
RDR_DEV vec4 image_sample_grad(const Image& im, const uint8_t* texels, const float* lut, float u, float v,
                               float dudx, float dvdx, float dudy, float dvdy, bool lut_linear = false) {
    const float w = (float)im.width, h = (float)im.height;
    const float ax = dudx * w, ay = dvdx * h, bx = dudy * w, by = dvdy * h;
    // log2 of the longer footprint axis = half the log2 of its squared length: no square root
    const float rho2 = fmaxf(ax * ax + ay * ay, bx * bx + by * by);
    return image_sample_lod(im, texels, lut, u, v, 0.5f * log2f(rho2), lut_linear);
}

// ---- cubemaps ------------------------------------------------------------------
RDR_DEV int cube_face(vec3 d, float& sc, float& tc, float& ma) {
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    if (ax >= ay && ax >= az) {
        ma = ax;
        if (d.x >= 0.0f) { sc = -d.z; tc = -d.y; return 0; }
        sc = d.z; tc = -d.y; return 1;
    }
    if (ay >= az) {
        ma = ay;
        if (d.y >= 0.0f) { sc = d.x; tc = d.z; return 2; }
        sc = d.x; tc = -d.z; return 3;
    }
    ma = az;
    if (d.z >= 0.0f) { sc = d.x; tc = -d.y; return 4; }
    sc = -d.x; tc = -d.y; return 5;
}

// Cubemaps are stored with a one-texel border around every face (rdr_pack_cubemap in
// rdr_internal.h): a bilinear footprint reaches at most one texel outside its face, and the border
// holds the texel that seamless filtering takes there, so all four taps are plain neighbours.
RDR_DEV vec4 cube_bilinear(const GpuCubemap& cm, const uint8_t* texels, uint32_t level, vec3 d) {
    float sc, tc, ma;
    const int face = cube_face(d, sc, tc, ma);
    const int n = (int)mip_dim(cm.size, level);
    const float inv_ma = 1.0f / ma; // one reciprocal for both face coordinates
    const float s = 0.5f * (sc * inv_ma + 1.0f), t = 0.5f * (tc * inv_ma + 1.0f);
    const float x = s * (float)n - 0.5f, y = t * (float)n - 0.5f;
    const float fx = floorf(x), fy = floorf(y);
    const float ax = x - fx, ay = y - fy;
    // -1 .. n-1 for any finite direction; the clamp keeps a non-finite one inside the allocation
    const int i0 = min(max((int)fx, -1), n - 1), j0 = min(max((int)fy, -1), n - 1);
    const uint32_t tb = cm.format == RDR_FORMAT_RGBA32F ? 16u : 4u;
    const uint32_t pitch = (uint32_t)(n + 2);
    // six faces of at most 16 386 x 16 386 bordered texels: the texel index fits 32 bits, the byte offset may not
    const uint32_t texel = ((uint32_t)face * pitch + (uint32_t)(j0 + 1)) * pitch + (uint32_t)(i0 + 1);
    const uint8_t* p = texels + cm.offset + cm.level_offset[level] + (size_t)texel * tb;
    const vec4 t00 = decode_texel(p, cm.format, nullptr);
    const vec4 t10 = decode_texel(p + tb, cm.format, nullptr);
    const vec4 t01 = decode_texel(p + (size_t)(pitch * tb), cm.format, nullptr);
    const vec4 t11 = decode_texel(p + (size_t)(pitch * tb) + tb, cm.format, nullptr);
    return lerp4(lerp4(t00, t10, ax), lerp4(t01, t11, ax), ay);
}

RDR_DEV vec4 cube_sample_lod(const GpuCubemap& cm, const uint8_t* texels, vec3 d, float lod) {
    const float maxl = (float)(cm.mip_count - 1);
    if (!(lod > 0.0f)) lod = 0.0f;
    if (lod > maxl) lod = maxl;
    const float fl = floorf(lod);
    const uint32_t l0 = (uint32_t)fl;
    const uint32_t l1 = l0 + 1 < cm.mip_count ? l0 + 1 : l0;
    const float t = lod - fl;
    const vec4 a = cube_bilinear(cm, texels, l0, d);
    if (l1 == l0 || t == 0.0f) return a;
    const vec4 b = cube_bilinear(cm, texels, l1, d);
    return lerp4(a, b, t);
}
