/*
 * orc_texture.h — software sampler of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY (see orc_math.h).
 *
 * Restates the one sampler the reference binds to every texture
 * (renderer/src/renderer.rs:755-763): LINEAR mag/min, LINEAR mipmap mode,
 * REPEAT addressing, LOD range 0..16, no anisotropy.  sRGB texels are decoded
 * to linear before filtering, as a *_SRGB Vulkan format does.
 */
#ifndef ORC_TEXTURE_H
#define ORC_TEXTURE_H

#include "orc_math.h"
#include "../include/rdr_abi.h"

static float g_srgb_lut[256];
static int g_srgb_lut_ready = 0;

static inline void orc_srgb_lut_init(void) {
    if (g_srgb_lut_ready) return;
    for (int i = 0; i < 256; ++i) {
        float c = (float)i / 255.0f;
        g_srgb_lut[i] = c <= 0.04045f ? c / 12.92f : powf((c + 0.055f) / 1.055f, 2.4f);
    }
    g_srgb_lut_ready = 1;
}

static inline uint32_t mip_dim(uint32_t d, uint32_t level) {
    uint32_t v = d >> level;
    return v ? v : 1u;
}

static inline size_t texel_bytes(uint32_t format) { return format == RDR_FORMAT_RGBA32F ? 16 : 4; }

static inline v4 orc_decode_texel(const uint8_t* p, uint32_t format) {
    v4 r;
    if (format == RDR_FORMAT_RGBA32F) {
        memcpy(&r, p, 16);
    } else if (format == RDR_FORMAT_RGBA8_SRGB) {
        r.x = g_srgb_lut[p[0]]; r.y = g_srgb_lut[p[1]]; r.z = g_srgb_lut[p[2]];
        r.w = (float)p[3] / 255.0f;
    } else {
        r.x = (float)p[0] / 255.0f; r.y = (float)p[1] / 255.0f;
        r.z = (float)p[2] / 255.0f; r.w = (float)p[3] / 255.0f;
    }
    return r;
}

/* a + t (b - a) with the product and the sum fused, as the GPU sampler evaluates it */
static inline v4 v4_lerp(v4 a, v4 b, float t) {
    v4 r = {fmaf(t, b.x - a.x, a.x), fmaf(t, b.y - a.y, a.y), fmaf(t, b.z - a.z, a.z), fmaf(t, b.w - a.w, a.w)};
    return r;
}

static inline int wrap_repeat(int i, int n) {
    int m = i % n;
    return m < 0 ? m + n : m;
}

/* Byte offset of a mip level inside a tightly packed 2D mip chain. */
static inline size_t image_level_offset(const rdr_image_desc* im, uint32_t level) {
    size_t off = 0;
    for (uint32_t l = 0; l < level; ++l)
        off += (size_t)mip_dim(im->width, l) * mip_dim(im->height, l) * texel_bytes(im->format);
    return off;
}

/* Bilinear tap at one level: texel centres at integer + 0.5, REPEAT. */
static inline v4 orc_image_bilinear(const rdr_image_desc* im, uint32_t level, float u, float v) {
    int w = (int)mip_dim(im->width, level), h = (int)mip_dim(im->height, level);
    const uint8_t* base = (const uint8_t*)im->data + image_level_offset(im, level);
    size_t tb = texel_bytes(im->format);
    float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    float ax = x - fx, ay = y - fy;
    int x0 = wrap_repeat((int)fx, w), x1 = wrap_repeat((int)fx + 1, w);
    int y0 = wrap_repeat((int)fy, h), y1 = wrap_repeat((int)fy + 1, h);
    v4 t00 = orc_decode_texel(base + ((size_t)y0 * w + x0) * tb, im->format);
    v4 t10 = orc_decode_texel(base + ((size_t)y0 * w + x1) * tb, im->format);
    v4 t01 = orc_decode_texel(base + ((size_t)y1 * w + x0) * tb, im->format);
    v4 t11 = orc_decode_texel(base + ((size_t)y1 * w + x1) * tb, im->format);
    return v4_lerp(v4_lerp(t00, t10, ax), v4_lerp(t01, t11, ax), ay);
}

/* textureLod on a 2D image. */
static inline v4 orc_image_sample_lod(const rdr_image_desc* im, float u, float v, float lod) {
    float maxl = (float)(im->mip_count - 1);
    if (maxl > 16.0f) maxl = 16.0f;
    if (!(lod > 0.0f)) lod = 0.0f; /* also catches NaN and -inf */
    if (lod > maxl) lod = maxl;
    float fl = floorf(lod);
    uint32_t l0 = (uint32_t)fl;
    uint32_t l1 = l0 + 1 < im->mip_count ? l0 + 1 : l0;
    float t = lod - fl;
    v4 a = orc_image_bilinear(im, l0, u, v);
    if (l1 == l0 || t == 0.0f) return a;
    v4 b = orc_image_bilinear(im, l1, u, v);
    return v4_lerp(a, b, t);
}

/* textureGrad: isotropic LOD from the longer of the two footprint axes. */
static inline v4 orc_image_sample_grad(const rdr_image_desc* im, float u, float v, float dudx,
                                       float dvdx, float dudy, float dvdy) {
    float w = (float)im->width, h = (float)im->height;
    float ax = dudx * w, ay = dvdx * h, bx = dudy * w, by = dvdy * h;
    /* log2 of the longer footprint axis, as half the log2 of its squared length */
    float rho2 = fmaxf(ax * ax + ay * ay, bx * bx + by * by);
    float lod = 0.5f * log2f(rho2);
    return orc_image_sample_lod(im, u, v, lod);
}

/* ---------------- cubemaps ---------------- */

static inline size_t cube_level_offset(const rdr_cubemap_desc* cm, uint32_t level) {
    size_t off = 0;
    for (uint32_t l = 0; l < level; ++l) {
        size_t n = mip_dim(cm->size, l);
        off += 6 * n * n * texel_bytes(cm->format);
    }
    return off;
}

/* Vulkan cube face selection: face index and (sc, tc, |ma|). */
static inline int cube_face(v3 d, float* sc, float* tc, float* ma) {
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    if (ax >= ay && ax >= az) {
        *ma = ax;
        if (d.x >= 0.0f) { *sc = -d.z; *tc = -d.y; return 0; }
        *sc = d.z; *tc = -d.y; return 1;
    }
    if (ay >= az) {
        *ma = ay;
        if (d.y >= 0.0f) { *sc = d.x; *tc = d.z; return 2; }
        *sc = d.x; *tc = -d.z; return 3;
    }
    *ma = az;
    if (d.z >= 0.0f) { *sc = d.x; *tc = -d.y; return 4; }
    *sc = -d.x; *tc = -d.y; return 5;
}

/* Seamless cube edges (Vulkan leaves the exact texel choice to the implementation): a bilinear
 * footprint reaches at most one texel outside its face.  Texel (i, j) of `face`, i and j in
 * [-1, n], is replaced by the cube texel that contains its centre once the centre is projected
 * back onto the cube.  In units of half a texel that centre is an integer vector (major component
 * +-n, the others 2i+1-n and 2j+1-n), so the face selection of cube_face() and the texel index are
 * evaluated exactly: texels inside the face map to themselves, edge texels to the neighbouring
 * face's border row, the corner texel to the border corner of the face that wins cube_face()'s
 * tie order. */
static inline void cube_wrap_texel(int* face, int* i, int* j, int n) {
    int a = 2 * *i + 1 - n, b = 2 * *j + 1 - n;
    int x, y, z;
    switch (*face) { /* same convention as the direction of face-space point (a, b) */
    case 0: x = n; y = -b; z = -a; break;
    case 1: x = -n; y = -b; z = a; break;
    case 2: x = a; y = n; z = b; break;
    case 3: x = a; y = -n; z = -b; break;
    case 4: x = a; y = -b; z = n; break;
    default: x = -a; y = -b; z = -n; break;
    }
    int ax = abs(x), ay = abs(y), az = abs(z), sc, tc;
    if (ax >= ay && ax >= az) {
        if (x >= 0) { *face = 0; sc = -z; tc = -y; } else { *face = 1; sc = z; tc = -y; }
    } else if (ay >= az) {
        if (y >= 0) { *face = 2; sc = x; tc = z; } else { *face = 3; sc = x; tc = -z; }
    } else {
        if (z >= 0) { *face = 4; sc = x; tc = -y; } else { *face = 5; sc = -x; tc = -y; }
    }
    /* floor((sc + n) / 2 - 1/2 + 1/2): sc + n - 1 is even for in-face centres; arithmetic shift floors */
    int ii = (sc + n - 1) >> 1, jj = (tc + n - 1) >> 1;
    *i = ii < 0 ? 0 : (ii > n - 1 ? n - 1 : ii);
    *j = jj < 0 ? 0 : (jj > n - 1 ? n - 1 : jj);
}

static inline v4 cube_fetch(const rdr_cubemap_desc* cm, uint32_t level, int face, int i, int j) {
    int n = (int)mip_dim(cm->size, level);
    size_t tb = texel_bytes(cm->format);
    cube_wrap_texel(&face, &i, &j, n);
    const uint8_t* base = (const uint8_t*)cm->data + cube_level_offset(cm, level) +
                          (size_t)face * n * n * tb;
    return orc_decode_texel(base + ((size_t)j * n + i) * tb, cm->format);
}

static inline v4 orc_cube_bilinear(const rdr_cubemap_desc* cm, uint32_t level, v3 d) {
    float sc, tc, ma;
    int face = cube_face(d, &sc, &tc, &ma);
    int n = (int)mip_dim(cm->size, level);
    float inv_ma = 1.0f / ma; /* one reciprocal for both face coordinates, as on the GPU */
    float s = 0.5f * (sc * inv_ma + 1.0f), t = 0.5f * (tc * inv_ma + 1.0f);
    float x = s * (float)n - 0.5f, y = t * (float)n - 0.5f;
    float fx = floorf(x), fy = floorf(y);
    float ax = x - fx, ay = y - fy;
    int i0 = (int)fx, j0 = (int)fy;
    v4 t00 = cube_fetch(cm, level, face, i0, j0);
    v4 t10 = cube_fetch(cm, level, face, i0 + 1, j0);
    v4 t01 = cube_fetch(cm, level, face, i0, j0 + 1);
    v4 t11 = cube_fetch(cm, level, face, i0 + 1, j0 + 1);
    return v4_lerp(v4_lerp(t00, t10, ax), v4_lerp(t01, t11, ax), ay);
}

/* textureLod on a cubemap. */
static inline v4 orc_cube_sample_lod(const rdr_cubemap_desc* cm, v3 d, float lod) {
    float maxl = (float)(cm->mip_count - 1);
    if (!(lod > 0.0f)) lod = 0.0f;
    if (lod > maxl) lod = maxl;
    float fl = floorf(lod);
    uint32_t l0 = (uint32_t)fl;
    uint32_t l1 = l0 + 1 < cm->mip_count ? l0 + 1 : l0;
    float t = lod - fl;
    v4 a = orc_cube_bilinear(cm, l0, d);
    if (l1 == l0 || t == 0.0f) return a;
    v4 b = orc_cube_bilinear(cm, l1, d);
    return v4_lerp(a, b, t);
}

#endif
