/*
 * orc_bake.c — CPU restatement of the reference's IBL bake: the BRDF look-up table and the
 * importance-sampled environment-map prefilter the shading pass samples.
 *
 * TEST INFRASTRUCTURE ONLY (see orc_math.h): the checker for the HIP bake kernels, never the
 * thing measured or shipped.  Parity unpinned: the reference holds no fixtures for the bake.
 *
 * Follows (paths relative to the reference tree):
 *   assets/shaders/brdf.comp.glsl:20-143            BRDF LUT, r/g GGX split sum, b Charlie
 *   assets/shaders/filter_envmap.frag.glsl:26-245   Lambertian / GGX / Charlie prefilter
 *   assets/shaders/math.inc.glsl:28-37              hammersley2d
 *   bake/src/brdf.rs:190-245                        one pass per face and mip, roughness =
 *                                                   mip / (mips - 1), flipped viewport
 *   assets/shaders/fullscreen_triangle.vert.glsl:7-9  fragment -> uv
 */
#include <stdlib.h>
#include "orc_texture.h"
#include "rdr_oracle.h"

#define ORC_PI 3.1415926535897932384626433832795f

static void hammersley2d(uint32_t i, uint32_t n, float* x, float* y) {
    uint32_t bits = (i << 16) | (i >> 16);
    bits = ((bits & 0x55555555u) << 1) | ((bits & 0xAAAAAAAAu) >> 1);
    bits = ((bits & 0x33333333u) << 2) | ((bits & 0xCCCCCCCCu) >> 2);
    bits = ((bits & 0x0F0F0F0Fu) << 4) | ((bits & 0xF0F0F0F0u) >> 4);
    bits = ((bits & 0x00FF00FFu) << 8) | ((bits & 0xFF00FF00u) >> 8);
    *x = (float)i / (float)n;
    *y = (float)bits * 2.3283064365386963e-10f;
}

/* GLSL mod(x, y) = x - y * floor(x / y), fract(x) = x - floor(x) */
static float glsl_mod(float x, float y) { return x - y * floorf(x / y); }
static float glsl_fract(float x) { return x - floorf(x); }

static float bake_random(float cx, float cy) { /* brdf.comp.glsl:20-27 */
    float dt = cx * 12.9898f + cy * 78.233f;
    float sn = glsl_mod(dt, 3.14f);
    return glsl_fract(sinf(sn) * 43758.5453f);
}

/* hemisphere sample (sin_theta cos phi, sin_theta sin phi, cos_theta) rotated into the frame of
 * `normal` the way brdf.comp.glsl:38-46,93-101 does */
static v3 to_normal_frame(v3 h, v3 normal) {
    v3 up = fabsf(normal.z) < 0.999f ? v3_make(0.0f, 0.0f, 1.0f) : v3_make(1.0f, 0.0f, 0.0f);
    v3 tx = v3_normalize(v3_cross(up, normal));
    v3 ty = v3_normalize(v3_cross(normal, tx));
    return v3_normalize(v3_add(v3_add(v3_scale(tx, h.x), v3_scale(ty, h.y)), v3_scale(normal, h.z)));
}

static float d_charlie(float sheen_roughness, float ndh) { /* brdf.comp.glsl:61-67, filter_envmap.frag.glsl:75-81 */
    sheen_roughness = fmaxf(sheen_roughness, 0.000001f);
    float inv_r = 1.0f / sheen_roughness;
    float sin2h = 1.0f - ndh * ndh;
    return (2.0f + inv_r) * powf(sin2h, inv_r * 0.5f) / (2.0f * ORC_PI);
}

int orc_bake_brdf_lut(uint32_t size, uint32_t samples, float* out_rgba32f) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int gy = 0; gy < (int)size; ++gy) {
        for (uint32_t gx = 0; gx < size; ++gx) {
            float ndv_in = ((float)gx + 0.5f) / (float)size;
            float roughness = 1.0f - ((float)gy + 0.5f) / (float)size;
            const v3 n = v3_make(0.0f, 0.0f, 1.0f);
            v3 v = v3_make(sqrtf(1.0f - ndv_in * ndv_in), 0.0f, ndv_in);
            float lr = 0.0f, lg = 0.0f, lb = 0.0f;
            for (uint32_t i = 0; i < samples; ++i) { /* GGX, brdf.comp.glsl:105-123 */
                float xi_x, xi_y;
                hammersley2d(i, samples, &xi_x, &xi_y);
                float alpha = roughness * roughness;
                float phi = 2.0f * ORC_PI * xi_x + bake_random(n.x, n.z) * 0.1f;
                float cos_t = sqrtf((1.0f - xi_y) / (1.0f + (alpha * alpha - 1.0f) * xi_y));
                float sin_t = sqrtf(1.0f - cos_t * cos_t);
                v3 h = to_normal_frame(v3_make(sin_t * cosf(phi), sin_t * sinf(phi), cos_t), n);
                float vdh_s = v3_dot(v, h);
                v3 l = v3_sub(v3_scale(h, 2.0f * vdh_s), v);
                float ndl = fmaxf(v3_dot(n, l), 0.0f), ndv = fmaxf(v3_dot(n, v), 0.0f);
                float vdh = fmaxf(vdh_s, 0.0f), hdn = fmaxf(v3_dot(h, n), 0.0f);
                if (ndl > 0.0f) {
                    float k = (roughness * roughness) / 2.0f;
                    float g = (ndl / (ndl * (1.0f - k) + k)) * (ndv / (ndv * (1.0f - k) + k));
                    float g_vis = (g * vdh) / (hdn * ndv);
                    float fc = powf(1.0f - vdh, 5.0f);
                    lr += (1.0f - fc) * g_vis;
                    lg += fc * g_vis;
                }
            }
            for (uint32_t i = 0; i < samples; ++i) { /* Charlie, brdf.comp.glsl:124-139 */
                float xi_x, xi_y;
                hammersley2d(i, samples, &xi_x, &xi_y);
                float alpha = roughness * roughness;
                float phi = 2.0f * ORC_PI * xi_x;
                float sin_t = powf(xi_y, alpha / (2.0f * alpha + 1.0f));
                float cos_t = sqrtf(1.0f - sin_t * sin_t);
                v3 h = to_normal_frame(v3_make(sin_t * cosf(phi), sin_t * sinf(phi), cos_t), n);
                float vdh_s = v3_dot(v, h);
                v3 l = v3_sub(v3_scale(h, 2.0f * vdh_s), v);
                float ndl = fmaxf(v3_dot(n, l), 0.0f), ndv = fmaxf(v3_dot(n, v), 0.0f);
                float vdh = fmaxf(vdh_s, 0.0f), hdn = fmaxf(v3_dot(h, n), 0.0f);
                if (ndl > 0.0f) {
                    float dist = d_charlie(roughness, hdn);
                    float vis = clampf(1.0f / (4.0f * (ndl + ndv - ndl * ndv)), 0.0f, 1.0f);
                    lb += vis * dist * ndl * vdh;
                }
            }
            float* o = out_rgba32f + ((size_t)gy * size + gx) * 4;
            o[0] = lr / (float)samples; o[1] = lg / (float)samples; o[2] = lb / (float)samples;
            o[3] = 0.0f; /* never written by the shader (brdf.comp.glsl:137-141) */
        }
    }
    return 0;
}

static v3 uv_to_xyz(uint32_t face, float u, float v) { /* filter_envmap.frag.glsl:26-34 */
    switch (face) {
    case 0: return v3_make(1.0f, v, -u);
    case 1: return v3_make(-1.0f, v, u);
    case 2: return v3_make(u, 1.0f, -v);
    case 3: return v3_make(u, -1.0f, v);
    case 4: return v3_make(u, v, 1.0f);
    default: return v3_make(-u, v, -1.0f);
    }
}

int orc_bake_env_map(const rdr_cubemap_desc* src, uint32_t distribution, uint32_t size, uint32_t mips,
                     uint32_t samples, float* out_rgba32f) {
    size_t level_base = 0;
    for (uint32_t level = 0; level < mips; ++level) {
        const uint32_t n_px = size >> level ? size >> level : 1u;
        const float roughness = mips > 1 ? (float)level / (float)(mips - 1) : 0.0f;
        for (uint32_t face = 0; face < 6; ++face) {
            float* dst = out_rgba32f + (level_base + (size_t)face * n_px * n_px) * 4;
#pragma omp parallel for schedule(dynamic, 1)
            for (int j = 0; j < (int)n_px; ++j) {
                for (uint32_t i = 0; i < n_px; ++i) {
                    /* fullscreen triangle under the flipped viewport: uv = (x, 1 - y) of the fragment centre */
                    float uvx = ((float)i + 0.5f) / (float)n_px, uvy = 1.0f - ((float)j + 0.5f) / (float)n_px;
                    v3 n = v3_normalize(uv_to_xyz(face, uvx * 2.0f - 1.0f, uvy * 2.0f - 1.0f));
                    /* generate_tbn, filter_envmap.frag.glsl:117-132 */
                    v3 bitangent = v3_make(0.0f, 1.0f, 0.0f);
                    float n_dot_up = n.y;
                    if (1.0f - fabsf(n_dot_up) <= 0.0000001f)
                        bitangent = n_dot_up > 0.0f ? v3_make(0.0f, 0.0f, 1.0f) : v3_make(0.0f, 0.0f, -1.0f);
                    v3 tangent = v3_normalize(v3_cross(bitangent, n));
                    bitangent = v3_cross(n, tangent);
                    float cr = 0.0f, cg = 0.0f, cb = 0.0f, weights = 0.0f;
                    for (uint32_t s = 0; s < samples; ++s) {
                        float xi_x, xi_y;
                        hammersley2d(s, samples, &xi_x, &xi_y);
                        float cos_t, sin_t, pdf;
                        const float phi = 2.0f * ORC_PI * xi_x;
                        if (distribution == 0) { /* cosine weighted, :102-114 */
                            cos_t = sqrtf(1.0f - xi_y);
                            sin_t = sqrtf(xi_y);
                            pdf = cos_t / ORC_PI;
                        } else if (distribution == 1) { /* GGX, :43-72 */
                            float alpha = roughness * roughness;
                            cos_t = clampf(sqrtf((1.0f - xi_y) / (1.0f + (alpha * alpha - 1.0f) * xi_y)), 0.0f, 1.0f);
                            sin_t = sqrtf(1.0f - cos_t * cos_t);
                            float a = cos_t * alpha;
                            float k = alpha / (1.0f - cos_t * cos_t + a * a);
                            pdf = k * k * (1.0f / ORC_PI) / 4.0f;
                        } else { /* Charlie, :83-100 */
                            float alpha = roughness * roughness;
                            sin_t = powf(xi_y, alpha / (2.0f * alpha + 1.0f));
                            cos_t = sqrtf(1.0f - sin_t * sin_t);
                            pdf = d_charlie(alpha, cos_t) / 4.0f;
                        }
                        v3 local = v3_normalize(v3_make(sin_t * cosf(phi), sin_t * sinf(phi), cos_t));
                        v3 h = v3_add(v3_add(v3_scale(tangent, local.x), v3_scale(bitangent, local.y)), v3_scale(n, local.z));
                        /* compute_lod, :190-199: width and height are the bake's base size */
                        float lod = 0.5f * log2f(6.0f * (float)size * (float)size / ((float)samples * pdf));
                        if (distribution == 0) {
                            v4 c = orc_cube_sample_lod(src, h, lod);
                            cr += c.x; cg += c.y; cb += c.z;
                        } else {
                            /* reflect(-v, h) with v = n */
                            float d = v3_dot(h, v3_scale(n, -1.0f));
                            v3 l = v3_normalize(v3_sub(v3_scale(n, -1.0f), v3_scale(h, 2.0f * d)));
                            float ndl = v3_dot(n, l);
                            if (ndl > 0.0f) {
                                if (roughness == 0.0f) lod = 0.0f;
                                v4 c = orc_cube_sample_lod(src, l, lod);
                                cr += c.x * ndl; cg += c.y * ndl; cb += c.z * ndl;
                                weights += ndl;
                            }
                        }
                    }
                    float div = weights != 0.0f ? weights : (float)samples;
                    float* o = dst + ((size_t)j * n_px + i) * 4;
                    o[0] = cr / div; o[1] = cg / div; o[2] = cb / div; o[3] = 1.0f;
                }
            }
        }
        level_base += (size_t)6 * n_px * n_px;
    }
    return 0;
}
