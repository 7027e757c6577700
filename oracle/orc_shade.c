/*
 * orc_shade.c — CPU restatement of the reference's deferred shading pass and
 * of the sRGB store its swapchain format performs.
 *
 * TEST INFRASTRUCTURE ONLY (see orc_math.h).  PARITY UNPINNED: no reference
 * test, fixture or golden image covers this path.
 *
 * Followed (paths relative to the reference tree):
 *   pixel -> uv, visbuffer decode        assets/shaders/fullscreen_triangle.vert.glsl:7-9,
 *                                        shading.frag.glsl:317-349
 *   analytic barycentrics + derivatives  shading.frag.glsl:81-171
 *   material taps and their quirks       shading.frag.glsl:373-414 (uv1 is a copy of uv0;
 *                                        emissive/specular taps ignore the texture transform)
 *   glTF PBR terms                       shading.frag.glsl:181-315,427-529
 *   sun shadow ray                       shading.frag.glsl:421-425,529 and the acceleration
 *                                        structures of renderer/src/renderer/
 *                                        acceleration_sturct_utils.rs:23-49,253-279 (LOD0
 *                                        triangles per mesh, one instance per draw, all opaque)
 *   sRGB store                           lava/src/swapchain.rs:100-115 (an *_SRGB image)
 *
 * Choices where the Vulkan implementation has latitude: texture(brdf_lut) and
 * texture(lambertian) use LOD 0 (their implicit LOD depends on quad
 * derivatives); pow(x, 5) and pow(x, 2) are exact products; the ray/triangle
 * test is watertight (as the ray query is), in object space with the instance
 * transform inverted in double precision; a ray hits when tmin < t < tmax.
 * GLSL fixes neither the rounding of normalize() nor contraction nor the order of
 * linear steps; the forms chosen (and evaluated identically by the GPU path, every
 * operation correctly rounded): normalize(x) = x * (1 / sqrt(x.x)); fmaf chains for
 * the matrix products and three-term dot products of this pass; mat3(model) applied
 * to the interpolated normal and tangent; texel blends a + t (b - a) with the product
 * fused; LOD = 0.5 * log2 of the squared footprint axis.
 */
#include <stdlib.h>
#include <stdio.h>
#include "orc_math.h"
#include "orc_texture.h"
#include "rdr_oracle.h"

#ifdef _OPENMP
#include <omp.h>
#endif

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

#define PI_F 3.1415926535897932384626433832795f
#define MIN_ROUGHNESS 0.04f

/* ------------------------------------------------------------------ */
/* Shadow-ray acceleration structure (oracle's own, median-split BVH)   */
/* ------------------------------------------------------------------ */

typedef struct {
    float lo[3], hi[3];
    int32_t left;  /* inner: index of left child (right = left + 1); leaf: first item */
    int32_t count; /* 0 = inner node */
} bvh_node;

typedef struct {
    bvh_node* nodes;
    int32_t* items;
    int n_nodes;
} bvh;

typedef struct { float lo[3], hi[3], c[3]; } item_box;

static void box_empty(float* lo, float* hi) {
    for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
}
static void box_grow(float* lo, float* hi, const float* blo, const float* bhi) {
    for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], blo[k]); hi[k] = fmaxf(hi[k], bhi[k]); }
}

static const item_box* g_sort_boxes;
static int g_sort_axis;
static int cmp_items(const void* a, const void* b) {
    float ca = g_sort_boxes[*(const int32_t*)a].c[g_sort_axis];
    float cb = g_sort_boxes[*(const int32_t*)b].c[g_sort_axis];
    return (ca > cb) - (ca < cb);
}

static void bvh_build_rec(bvh* b, const item_box* boxes, int node, int first, int count) {
    bvh_node* n = &b->nodes[node];
    box_empty(n->lo, n->hi);
    float clo[3], chi[3];
    box_empty(clo, chi);
    for (int i = 0; i < count; ++i) {
        const item_box* ib = &boxes[b->items[first + i]];
        box_grow(n->lo, n->hi, ib->lo, ib->hi);
        box_grow(clo, chi, ib->c, ib->c);
    }
    if (count <= 4) { n->left = first; n->count = count; return; }
    int axis = 0;
    float ext = chi[0] - clo[0];
    for (int k = 1; k < 3; ++k) if (chi[k] - clo[k] > ext) { ext = chi[k] - clo[k]; axis = k; }
    g_sort_boxes = boxes; g_sort_axis = axis;
    qsort(b->items + first, (size_t)count, sizeof(int32_t), cmp_items);
    int half = count / 2;
    int l = b->n_nodes; b->n_nodes += 2;
    n = &b->nodes[node];
    n->left = l; n->count = 0;
    bvh_build_rec(b, boxes, l, first, half);
    bvh_build_rec(b, boxes, l + 1, first + half, count - half);
}

static void bvh_build(bvh* b, const item_box* boxes, int count) {
    b->nodes = (bvh_node*)malloc(sizeof(bvh_node) * (size_t)(2 * count + 1));
    b->items = (int32_t*)malloc(sizeof(int32_t) * (size_t)(count > 0 ? count : 1));
    for (int i = 0; i < count; ++i) b->items[i] = i;
    b->n_nodes = 1;
    if (count == 0) { box_empty(b->nodes[0].lo, b->nodes[0].hi); b->nodes[0].left = 0; b->nodes[0].count = 0; return; }
    bvh_build_rec(b, boxes, 0, 0, count);
}

/* Conservative slab test (hit interval widened by a few ulps). */
static int ray_box(const float* lo, const float* hi, const float* o, const float* inv_d, float tmin,
                   float tmax) {
    float t0 = tmin, t1 = tmax;
    for (int k = 0; k < 3; ++k) {
        float tn = (lo[k] - o[k]) * inv_d[k], tf = (hi[k] - o[k]) * inv_d[k];
        if (tn > tf) { float t = tn; tn = tf; tf = t; }
        tn = tn - 4e-6f * fabsf(tn);
        tf = tf + 4e-6f * fabsf(tf);
        /* NaN (0 * inf) must not reject */
        if (tn > t0) t0 = tn;
        if (tf < t1) t1 = tf;
        if (t0 > t1) return 0;
    }
    return 1;
}

/* Ray / triangle, two-sided, open interval (tmin, tmax), watertight: the Vulkan ray query the reference uses
 * guarantees that a ray through an edge or a vertex shared by triangles hits at least one of them, so a closed
 * mesh has no cracks for a ray to slip through.  The test is the one of Woop, Benthin and Wald ("Watertight
 * Ray/Triangle Intersection", 2013): the triangle is moved to the ray's origin and sheared so that the ray runs
 * along the axis of its largest direction component; the three edge functions of the projected triangle are
 * then differences of products of the same two vertices whichever of the two triangles at an edge computes
 * them, hence agree in sign, and an edge function that comes out exactly zero is recomputed in double
 * precision.  t = T / det is compared without the division.  (Both faces count, so the paper's swap of
 * the two other axes for a negative direction component, which only keeps the winding, is left out.) */
static float v3_get(v3 a, int k) { return k == 0 ? a.x : (k == 1 ? a.y : a.z); }
static int ray_triangle(v3 o, v3 d, v3 p0, v3 p1, v3 p2, float tmin, float tmax) {
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    int kz = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
    int kx = kz == 2 ? 0 : kz + 1, ky = kx == 2 ? 0 : kx + 1;
    float dz = v3_get(d, kz);
    float Sx = v3_get(d, kx) / dz, Sy = v3_get(d, ky) / dz, Sz = 1.0f / dz;
    v3 A = v3_sub(p0, o), B = v3_sub(p1, o), C = v3_sub(p2, o);
    float Akz = v3_get(A, kz), Bkz = v3_get(B, kz), Ckz = v3_get(C, kz);
    float Ax = v3_get(A, kx) - Sx * Akz, Ay = v3_get(A, ky) - Sy * Akz;
    float Bx = v3_get(B, kx) - Sx * Bkz, By = v3_get(B, ky) - Sy * Bkz;
    float Cx = v3_get(C, kx) - Sx * Ckz, Cy = v3_get(C, ky) - Sy * Ckz;
    float U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) {
        U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
        V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
        W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
    }
    if ((U < 0.0f || V < 0.0f || W < 0.0f) && (U > 0.0f || V > 0.0f || W > 0.0f)) return 0;
    float det = (U + V) + W;
    if (det == 0.0f) return 0;
    float Az = Sz * Akz, Bz = Sz * Bkz, Cz = Sz * Ckz;
    float T = (U * Az + V * Bz) + W * Cz;
    float sgn = det < 0.0f ? -1.0f : 1.0f;
    float a = fabsf(det), Ts = T * sgn;
    return Ts > tmin * a && Ts < tmax * a;
}

typedef struct {
    const rdr_scene_desc* s;
    bvh* blas;       /* per mesh, over LOD0 triangles */
    bvh tlas;        /* over draws */
    float* inv;      /* per draw: 12 floats, rows of the inverse 3x4 */
} shadow_scene;

static uint32_t mesh_index_count(const rdr_scene_desc* s, uint32_t mi) {
    uint32_t end = mi + 1 < s->mesh_count ? s->meshes[mi + 1].index_offset : s->index_count;
    return end - s->meshes[mi].index_offset;
}

/* Inverse of the affine part of a column-major mat4, in double, rounded once. */
static void affine_inverse(const float* m, float* out12) {
    double a00 = m[0], a10 = m[1], a20 = m[2];
    double a01 = m[4], a11 = m[5], a21 = m[6];
    double a02 = m[8], a12 = m[9], a22 = m[10];
    double tx = m[12], ty = m[13], tz = m[14];
    double c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
    double det = a00 * c00 + a01 * c01 + a02 * c02;
    double id = 1.0 / det;
    double i00 = c00 * id, i01 = (a02 * a21 - a01 * a22) * id, i02 = (a01 * a12 - a02 * a11) * id;
    double i10 = c01 * id, i11 = (a00 * a22 - a02 * a20) * id, i12 = (a02 * a10 - a00 * a12) * id;
    double i20 = c02 * id, i21 = (a01 * a20 - a00 * a21) * id, i22 = (a00 * a11 - a01 * a10) * id;
    out12[0] = (float)i00; out12[1] = (float)i01; out12[2] = (float)i02;
    out12[3] = (float)(-(i00 * tx + i01 * ty + i02 * tz));
    out12[4] = (float)i10; out12[5] = (float)i11; out12[6] = (float)i12;
    out12[7] = (float)(-(i10 * tx + i11 * ty + i12 * tz));
    out12[8] = (float)i20; out12[9] = (float)i21; out12[10] = (float)i22;
    out12[11] = (float)(-(i20 * tx + i21 * ty + i22 * tz));
}

static void shadow_scene_build(shadow_scene* ss, const rdr_scene_desc* s) {
    ss->s = s;
    ss->blas = (bvh*)calloc(s->mesh_count ? s->mesh_count : 1, sizeof(bvh));
    for (uint32_t mi = 0; mi < s->mesh_count; ++mi) {
        const rdr_mesh* mesh = &s->meshes[mi];
        uint32_t ntri = mesh_index_count(s, mi) / 3;
        item_box* boxes = (item_box*)malloc(sizeof(item_box) * (ntri ? ntri : 1));
        for (uint32_t t = 0; t < ntri; ++t) {
            box_empty(boxes[t].lo, boxes[t].hi);
            for (int k = 0; k < 3; ++k) {
                uint32_t vi = s->indices[mesh->index_offset + 3 * t + k] + mesh->vertex_offset;
                box_grow(boxes[t].lo, boxes[t].hi, s->vertices[vi].p, s->vertices[vi].p);
            }
            for (int k = 0; k < 3; ++k) {
                float pad = 2e-6f * fmaxf(fabsf(boxes[t].lo[k]), fabsf(boxes[t].hi[k])) + 1e-30f;
                boxes[t].lo[k] -= pad; boxes[t].hi[k] += pad;
                boxes[t].c[k] = 0.5f * (boxes[t].lo[k] + boxes[t].hi[k]);
            }
        }
        bvh_build(&ss->blas[mi], boxes, (int)ntri);
        free(boxes);
    }
    item_box* ib = (item_box*)malloc(sizeof(item_box) * (s->draw_count ? s->draw_count : 1));
    ss->inv = (float*)malloc(sizeof(float) * 12 * (s->draw_count ? s->draw_count : 1));
    for (uint32_t di = 0; di < s->draw_count; ++di) {
        const rdr_mesh_draw* d = &s->mesh_draws[di];
        const float* model = s->global_transforms + (size_t)d->model_index * 16;
        affine_inverse(model, ss->inv + 12 * (size_t)di);
        const bvh_node* root = &ss->blas[d->mesh_index].nodes[0];
        box_empty(ib[di].lo, ib[di].hi);
        for (int corner = 0; corner < 8; ++corner) {
            v4 p = {(corner & 1) ? root->hi[0] : root->lo[0], (corner & 2) ? root->hi[1] : root->lo[1],
                    (corner & 4) ? root->hi[2] : root->lo[2], 1.0f};
            v4 w = m4_mul_v4(model, p);
            float wp[3] = {w.x, w.y, w.z};
            box_grow(ib[di].lo, ib[di].hi, wp, wp);
        }
        for (int k = 0; k < 3; ++k) {
            float pad = 1e-4f * (ib[di].hi[k] - ib[di].lo[k]) + 1e-6f;
            ib[di].lo[k] -= pad; ib[di].hi[k] += pad;
            ib[di].c[k] = 0.5f * (ib[di].lo[k] + ib[di].hi[k]);
        }
    }
    bvh_build(&ss->tlas, ib, (int)s->draw_count);
    free(ib);
}

static void shadow_scene_free(shadow_scene* ss) {
    for (uint32_t mi = 0; mi < ss->s->mesh_count; ++mi) { free(ss->blas[mi].nodes); free(ss->blas[mi].items); }
    free(ss->blas); free(ss->tlas.nodes); free(ss->tlas.items); free(ss->inv);
}

static void safe_inv_dir(v3 d, float* inv_d) {
    inv_d[0] = 1.0f / d.x; inv_d[1] = 1.0f / d.y; inv_d[2] = 1.0f / d.z;
}

static int blas_any_hit(const shadow_scene* ss, uint32_t mesh_index, v3 o, v3 d, float tmin, float tmax) {
    const rdr_scene_desc* s = ss->s;
    const rdr_mesh* mesh = &s->meshes[mesh_index];
    const bvh* b = &ss->blas[mesh_index];
    float of[3] = {o.x, o.y, o.z}, inv_d[3];
    safe_inv_dir(d, inv_d);
    int stack[64], sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const bvh_node* n = &b->nodes[stack[--sp]];
        if (!ray_box(n->lo, n->hi, of, inv_d, tmin, tmax)) continue;
        if (n->count == 0 && n->left != 0) { stack[sp++] = n->left; stack[sp++] = n->left + 1; continue; }
        for (int i = 0; i < n->count; ++i) {
            uint32_t t = (uint32_t)b->items[n->left + i];
            const uint32_t* idx = &s->indices[mesh->index_offset + 3 * t];
            const float* p0 = s->vertices[idx[0] + mesh->vertex_offset].p;
            const float* p1 = s->vertices[idx[1] + mesh->vertex_offset].p;
            const float* p2 = s->vertices[idx[2] + mesh->vertex_offset].p;
            if (ray_triangle(o, d, v3_make(p0[0], p0[1], p0[2]), v3_make(p1[0], p1[1], p1[2]),
                             v3_make(p2[0], p2[1], p2[2]), tmin, tmax))
                return 1;
        }
    }
    return 0;
}

static int shadow_any_hit(const shadow_scene* ss, v3 o, v3 d, float tmin, float tmax) {
    const rdr_scene_desc* s = ss->s;
    if (s->draw_count == 0) return 0;
    float of[3] = {o.x, o.y, o.z}, inv_d[3];
    safe_inv_dir(d, inv_d);
    int stack[64], sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const bvh_node* n = &ss->tlas.nodes[stack[--sp]];
        if (!ray_box(n->lo, n->hi, of, inv_d, tmin, tmax)) continue;
        if (n->count == 0 && n->left != 0) { stack[sp++] = n->left; stack[sp++] = n->left + 1; continue; }
        for (int i = 0; i < n->count; ++i) {
            uint32_t di = (uint32_t)ss->tlas.items[n->left + i];
            const float* iv = ss->inv + 12 * (size_t)di;
            v3 oo = v3_make(((iv[0] * o.x + iv[1] * o.y) + iv[2] * o.z) + iv[3],
                            ((iv[4] * o.x + iv[5] * o.y) + iv[6] * o.z) + iv[7],
                            ((iv[8] * o.x + iv[9] * o.y) + iv[10] * o.z) + iv[11]);
            v3 od = v3_make((iv[0] * d.x + iv[1] * d.y) + iv[2] * d.z,
                            (iv[4] * d.x + iv[5] * d.y) + iv[6] * d.z,
                            (iv[8] * d.x + iv[9] * d.y) + iv[10] * d.z);
            if (blas_any_hit(ss, s->mesh_draws[di].mesh_index, oo, od, tmin, tmax)) return 1;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* Shading                                                             */
/* ------------------------------------------------------------------ */

typedef struct { v3 lambda, ddx, ddy; } bary_deriv;

/* shading.frag.glsl:81-120 */
static bary_deriv calc_full_bary(v4 pt0, v4 pt1, v4 pt2, float ndcx, float ndcy, float winw, float winh) {
    bary_deriv r;
    v3 inv_w = v3_make(1.0f / pt0.w, 1.0f / pt1.w, 1.0f / pt2.w);
    float n0x = pt0.x * inv_w.x, n0y = pt0.y * inv_w.x;
    float n1x = pt1.x * inv_w.y, n1y = pt1.y * inv_w.y;
    float n2x = pt2.x * inv_w.z, n2y = pt2.y * inv_w.z;
    /* determinant(mat2(ndc2 - ndc1, ndc0 - ndc1)), columns */
    float ax = n2x - n1x, ay = n2y - n1y, bx = n0x - n1x, by = n0y - n1y;
    float inv_det = 1.0f / (ax * by - bx * ay);
    r.ddx = v3_mul(v3_scale(v3_make(n1y - n2y, n2y - n0y, n0y - n1y), inv_det), inv_w);
    r.ddy = v3_mul(v3_scale(v3_make(n2x - n1x, n0x - n2x, n1x - n0x), inv_det), inv_w);
    float ddx_sum = (r.ddx.x + r.ddx.y) + r.ddx.z;
    float ddy_sum = (r.ddy.x + r.ddy.y) + r.ddy.z;
    float dx = ndcx - n0x, dy = ndcy - n0y;
    float interp_inv_w = (inv_w.x + dx * ddx_sum) + dy * ddy_sum;
    float interp_w = 1.0f / interp_inv_w;
    r.lambda.x = interp_w * ((inv_w.x + dx * r.ddx.x) + dy * r.ddy.x);
    r.lambda.y = interp_w * ((0.0f + dx * r.ddx.y) + dy * r.ddy.y);
    r.lambda.z = interp_w * ((0.0f + dx * r.ddx.z) + dy * r.ddy.z);
    float sx = 2.0f / winw, sy = 2.0f / winh;
    r.ddx = v3_scale(r.ddx, sx); r.ddy = v3_scale(r.ddy, sy);
    ddx_sum *= sx; ddy_sum *= sy;
    r.ddy = v3_scale(r.ddy, -1.0f); ddy_sum *= -1.0f;
    float iwx = 1.0f / (interp_inv_w + ddx_sum), iwy = 1.0f / (interp_inv_w + ddy_sum);
    r.ddx = v3_sub(v3_scale(v3_add(v3_scale(r.lambda, interp_inv_w), r.ddx), iwx), r.lambda);
    r.ddy = v3_sub(v3_scale(v3_add(v3_scale(r.lambda, interp_inv_w), r.ddy), iwy), r.lambda);
    return r;
}

typedef struct { float v, dx, dy; } interp1;
static interp1 interp_scalar(const bary_deriv* b, float a0, float a1, float a2) {
    v3 m = v3_make(a0, a1, a2);
    interp1 r = {v3_dot_fma(m, b->lambda), v3_dot_fma(m, b->ddx), v3_dot_fma(m, b->ddy)};
    return r;
}
static v3 interp_v3(const bary_deriv* b, v3 a0, v3 a1, v3 a2) {
    return v3_make(interp_scalar(b, a0.x, a1.x, a2.x).v, interp_scalar(b, a0.y, a1.y, a2.y).v,
                   interp_scalar(b, a0.z, a1.z, a2.z).v);
}

typedef struct { float u, v, dudx, dvdx, dudy, dvdy; } interp_uv;

/* shading.frag.glsl:165-171 */
static interp_uv uv_transform(interp_uv in, float sx, float sy, float ox, float oy) {
    interp_uv r;
    r.dudx = in.dudx * sx; r.dvdx = in.dvdx * sy;
    r.dudy = in.dudy * sx; r.dvdy = in.dvdy * sy;
    r.u = in.u * sx + ox; r.v = in.v * sy + oy;
    return r;
}

static v4 sample_or_default(const rdr_scene_desc* s, uint32_t tex, interp_uv uv, v4 def) {
    if (tex == RDR_NO_TEXTURE) return def;
    if (tex >= s->image_count || s->images[tex].data == NULL) { v4 z = {0, 0, 0, 0}; return z; }
    return orc_image_sample_grad(&s->images[tex], uv.u, uv.v, uv.dudx, uv.dvdx, uv.dudy, uv.dvdy);
}

static inline float pow5(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }
static inline v3 v3_splat(float a) { return v3_make(a, a, a); }
static inline v3 v3_mix(v3 a, v3 b, float t) {
    float s = 1.0f - t;
    return v3_make(a.x * s + b.x * t, a.y * s + b.y * t, a.z * s + b.z * t);
}
static inline float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
static inline v3 v3_clamp01(v3 a) { return v3_make(clampf(a.x, 0, 1), clampf(a.y, 0, 1), clampf(a.z, 0, 1)); }

static float geometric_occlusion(float ndl, float ndv, float alpha) {
    float r2 = alpha * alpha;
    float al = 2.0f * ndl / (ndl + sqrtf(r2 + (1.0f - r2) * (ndl * ndl)));
    float av = 2.0f * ndv / (ndv + sqrtf(r2 + (1.0f - r2) * (ndv * ndv)));
    return al * av;
}
static float microfacet_distribution(float alpha, float ndh) {
    float r2 = alpha * alpha;
    float f = (ndh * r2 - ndh) * ndh + 1.0f;
    return r2 / (PI_F * f * f);
}
static v3 f_schlick(v3 f0, v3 f90, float vdh) {
    float p = pow5(clampf(1.0f - vdh, 0.0f, 1.0f));
    return v3_add(f0, v3_scale(v3_sub(f90, f0), p));
}
static v3 diffuse_burley(v3 diffuse, float ldh, float ndl, float ndv, float alpha) {
    float f90 = 2.0f * ldh * ldh * alpha - 0.5f;
    float a = 1.0f + f90 * pow5(1.0f - ndl);
    float b = 1.0f + f90 * pow5(1.0f - ndv);
    return v3_scale(v3_scale(v3_scale(diffuse, 1.0f / PI_F), a), b);
}
static v3 reflect_neg(v3 v, v3 n) { /* reflect(-v, n) */
    v3 i = v3_scale(v, -1.0f);
    float d = v3_dot(n, i);
    return v3_sub(i, v3_scale(n, 2.0f * d));
}

typedef struct {
    const rdr_scene_desc* s;
    rdr_image_desc lut;
    rdr_cubemap_desc lambertian, ggx, charlie;
} env_ctx;

/* Debug state of orc_shade_quad_lod (below): the level of detail to take the Lambertian tap at instead of 0, and
 * where shade_pixel leaves the direction it takes that tap with. */
static _Thread_local float dbg_lambert_lod = 0.0f;
static _Thread_local v3 dbg_lambert_dir;

static v4 brdf_lut_tap(const env_ctx* e, float x, float y) {
    return orc_image_sample_lod(&e->lut, clampf(x, 0.0f, 1.0f), clampf(y, 0.0f, 1.0f), 0.0f);
}

static v3 ibl_ggx(const env_ctx* e, v3 n, v3 v, float ndv, float rough, v3 refl0, float spec_w) {
    v3 r = v3_normalize_rcp(reflect_neg(v, n));
    float mips = (float)e->ggx.mip_count;
    float lod = rough * (mips - 1.0f);
    v4 brdf = brdf_lut_tap(e, ndv, rough);
    v4 sl = orc_cube_sample_lod(&e->ggx, r, lod);
    v3 fr = v3_sub(v3_make(fmaxf(1.0f - rough, refl0.x), fmaxf(1.0f - rough, refl0.y), fmaxf(1.0f - rough, refl0.z)), refl0);
    v3 ks = v3_add(refl0, v3_scale(fr, pow5(1.0f - ndv)));
    v3 fss = v3_add(v3_scale(ks, brdf.x), v3_splat(brdf.y));
    return v3_mul(v3_scale(v3_make(sl.x, sl.y, sl.z), spec_w), fss);
}

static v3 ibl_lambertian(const env_ctx* e, float ndv, v3 n, float rough, v3 diffuse, v3 f0, float spec_w) {
    v4 fab = brdf_lut_tap(e, ndv, rough);
    v4 ir = orc_cube_sample_lod(&e->lambertian, n, dbg_lambert_lod); /* 0 outside orc_shade_quad_lod */
    dbg_lambert_dir = n;
    v3 irr = v3_make(ir.x, ir.y, ir.z);
    v3 fr = v3_sub(v3_make(fmaxf(1.0f - rough, f0.x), fmaxf(1.0f - rough, f0.y), fmaxf(1.0f - rough, f0.z)), f0);
    v3 ks = v3_add(f0, v3_scale(fr, pow5(1.0f - ndv)));
    v3 fss = v3_add(v3_scale(v3_scale(ks, spec_w), fab.x), v3_splat(fab.y));
    float ems = 1.0f - (fab.x + fab.y);
    v3 favg = v3_scale(v3_add(f0, v3_scale(v3_sub(v3_splat(1.0f), f0), 1.0f / 21.0f)), spec_w);
    v3 fms;
    fms.x = ems * fss.x * favg.x / (1.0f - favg.x * ems);
    fms.y = ems * fss.y * favg.y / (1.0f - favg.y * ems);
    fms.z = ems * fss.z * favg.z / (1.0f - favg.z * ems);
    v3 kd = v3_mul(diffuse, v3_add(v3_sub(v3_splat(1.0f), fss), fms));
    return v3_mul(v3_add(fms, kd), irr);
}

static v3 ibl_charlie(const env_ctx* e, v3 n, v3 v, float ndv, float sr, v3 sc) {
    float mips = (float)e->charlie.mip_count;
    float lod = sr * (mips - 1.0f);
    v3 r = v3_normalize_rcp(reflect_neg(v, n));
    float brdf = brdf_lut_tap(e, ndv, sr).z;
    v4 smp = orc_cube_sample_lod(&e->charlie, r, lod);
    return v3_scale(v3_mul(v3_make(smp.x, smp.y, smp.z), sc), brdf);
}

static float d_charlie(float sr, float ndh) {
    sr = fmaxf(sr, 0.000001f);
    float ag = sr * sr, inv_r = 1.0f / ag;
    float sin2h = 1.0f - ndh * ndh;
    return (2.0f + inv_r) * powf(sin2h, inv_r * 0.5f) / (2.0f * PI_F);
}
static float lambda_sheen_helper(float x, float ag) {
    float om = (1.0f - ag) * (1.0f - ag);
    float a = mixf(21.5473f, 25.3245f, om), b = mixf(3.82987f, 3.32435f, om);
    float c = mixf(0.19823f, 0.16801f, om), d = mixf(-1.97760f, -1.27393f, om);
    float e = mixf(-4.32054f, -4.85967f, om);
    return a / (1.0f + b * powf(x, c)) + d * x + e;
}
static float lambda_sheen(float ct, float ag) {
    if (fabsf(ct) < 0.5f) return expf(lambda_sheen_helper(ct, ag));
    return expf(2.0f * lambda_sheen_helper(0.5f, ag) - lambda_sheen_helper(1.0f - ct, ag));
}
static float v_sheen(float ndl, float ndv, float sr) {
    sr = fmaxf(sr, 0.000001f);
    float ag = sr * sr;
    /* the reference evaluates lambda_sheen(n_dot_l) twice (shading.frag.glsl:291) */
    return clampf(1.0f / ((1.0f + lambda_sheen(ndl, ag) + lambda_sheen(ndl, ag)) * (4.0f * ndv * ndl)), 0.0f, 1.0f);
}

static v3 clearcoat_punctual(v3 cn, v3 v, v3 l, v3 h, float vdh, v3 f0, v3 f90, float cr) {
    float ndl = clampf(v3_dot(cn, l), 0.001f, 1.0f);
    float ndv = clampf(v3_dot(cn, v), 0.001f, 1.0f);
    float ndh = clampf(v3_dot(cn, h), 0.001f, 1.0f);
    v3 f = f_schlick(f0, f90, vdh);
    float g = geometric_occlusion(ndl, ndv, cr * cr);
    float d = microfacet_distribution(cr * cr, ndh);
    v3 spec = v3_scale(v3_scale(v3_scale(f, g), d), 1.0f / (4.0f * ndl * ndv));
    return v3_scale(spec, ndl);
}

static uint8_t srgb_encode(float c) {
    c = clampf(c, 0.0f, 1.0f);
    float e = c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
    return (uint8_t)floorf(e * 255.0f + 0.5f);
}

static const uint8_t k_black_rgba8[4] = {0, 0, 0, 0};
static const float k_zero_rgba32f[4] = {0, 0, 0, 0};

static void env_setup(env_ctx* e, const rdr_scene_desc* s) {
    e->s = s;
    rdr_cubemap_desc black = {k_black_rgba8, 1, 1, RDR_FORMAT_RGBA8_UNORM, 0};
    /* six faces of one texel each: reuse the same 4 zero bytes via a 24-byte buffer */
    static const uint8_t black6[24] = {0};
    black.data = black6;
    rdr_image_desc zero_lut = {k_zero_rgba32f, 1, 1, 1, RDR_FORMAT_RGBA32F};
    if (s->env) {
        e->lut = s->env->brdf_lut.data ? s->env->brdf_lut : zero_lut;
        e->lambertian = s->env->lambertian.data ? s->env->lambertian : black;
        e->ggx = s->env->ggx.data ? s->env->ggx : black;
        e->charlie = s->env->charlie.data ? s->env->charlie : black;
    } else {
        e->lut = zero_lut; e->lambertian = black; e->ggx = black; e->charlie = black;
    }
}

static v3 shade_pixel(const rdr_scene_desc* s, const rdr_ubo* u, const env_ctx* env,
                      const shadow_scene* ss, uint32_t draw_id, uint32_t meshlet_index,
                      uint32_t tri, float uvx, float uvy) {
    const rdr_mesh_draw* d = &s->mesh_draws[draw_id];
    const rdr_mesh* mesh = &s->meshes[d->mesh_index];
    const float* model = s->global_transforms + (size_t)d->model_index * 16;
    rdr_material mat;
    if (d->material_index < s->material_count) mat = s->materials[d->material_index];
    else memset(&mat, 0, sizeof(mat)); /* robust buffer access: zeros */
    const rdr_meshlet* ml = &s->meshlets[meshlet_index];

    uint32_t i0 = s->meshlet_triangles[ml->triangle_offset + tri * 3 + 0];
    uint32_t i1 = s->meshlet_triangles[ml->triangle_offset + tri * 3 + 1];
    uint32_t i2 = s->meshlet_triangles[ml->triangle_offset + tri * 3 + 2];
    const rdr_vertex* v1 = &s->vertices[mesh->vertex_offset + s->meshlet_vertices[ml->vertex_offset + i0]];
    const rdr_vertex* v2 = &s->vertices[mesh->vertex_offset + s->meshlet_vertices[ml->vertex_offset + i1]];
    const rdr_vertex* v3p = &s->vertices[mesh->vertex_offset + s->meshlet_vertices[ml->vertex_offset + i2]];

    v4 p1 = {v1->p[0], v1->p[1], v1->p[2], 1.0f}, p2 = {v2->p[0], v2->p[1], v2->p[2], 1.0f},
       p3 = {v3p->p[0], v3p->p[1], v3p->p[2], 1.0f};
    v4 w1 = m4_mul_v4_fma(model, p1), w2 = m4_mul_v4_fma(model, p2), w3 = m4_mul_v4_fma(model, p3);
    v4 c1 = m4_mul_v4_fma(u->view_proj, w1), c2 = m4_mul_v4_fma(u->view_proj, w2), c3 = m4_mul_v4_fma(u->view_proj, w3);

    bary_deriv bary = calc_full_bary(c1, c2, c3, uvx * 2.0f - 1.0f, uvy * 2.0f - 1.0f,
                                     u->screen_width, u->screen_height);

    v3 w_pos = interp_v3(&bary, v3_make(w1.x, w1.y, w1.z), v3_make(w2.x, w2.y, w2.z), v3_make(w3.x, w3.y, w3.z));
    /* mat3(model) applied to the interpolated vector (linear either way; the GPU path does the same) */
    v3 norm = m4_mul_dir_fma(model, interp_v3(&bary, v3_make(v1->n[0], v1->n[1], v1->n[2]), v3_make(v2->n[0], v2->n[1], v2->n[2]),
                                              v3_make(v3p->n[0], v3p->n[1], v3p->n[2])));
    v3 tangent = m4_mul_dir_fma(model, interp_v3(&bary, v3_make(v1->t[0], v1->t[1], v1->t[2]), v3_make(v2->t[0], v2->t[1], v2->t[2]),
                                                 v3_make(v3p->t[0], v3p->t[1], v3p->t[2])));
    interp1 iu = interp_scalar(&bary, v1->uv[0], v2->uv[0], v3p->uv[0]);
    interp1 iv = interp_scalar(&bary, v1->uv[1], v2->uv[1], v3p->uv[1]);
    /* uv1 is built from .xy as well (shading.frag.glsl:373-374), so uv_index never matters */
    interp_uv uv0 = {iu.v, iv.v, iu.dx, iv.dx, iu.dy, iv.dy};

    v4 one = {1, 1, 1, 1}, nmap_def = {0.5f, 0.5f, 1.0f, 0.0f};
    v4 albedo = sample_or_default(s, mat.albedo_texture,
        uv_transform(uv0, mat.albedo_texture_transform[0], mat.albedo_texture_transform[1],
                     mat.albedo_texture_transform[2], mat.albedo_texture_transform[3]), one);
    albedo.x *= mat.base_color_factor[0]; albedo.y *= mat.base_color_factor[1];
    albedo.z *= mat.base_color_factor[2]; albedo.w *= mat.base_color_factor[3];
    v4 nm4 = sample_or_default(s, mat.normal_texture,
        uv_transform(uv0, mat.normal_texture_transform[0], mat.normal_texture_transform[1],
                     mat.normal_texture_transform[2], mat.normal_texture_transform[3]), nmap_def);
    v3 nmap = v3_make(nm4.x * 2.0f - 1.0f, nm4.y * 2.0f - 1.0f, nm4.z * 2.0f - 1.0f);
    v4 mr = sample_or_default(s, mat.metallic_roughness_texture,
        uv_transform(uv0, mat.metallic_roughness_texture_transform[0], mat.metallic_roughness_texture_transform[1],
                     mat.metallic_roughness_texture_transform[2], mat.metallic_roughness_texture_transform[3]), one);
    v4 occ = sample_or_default(s, mat.occlusion_texture,
        uv_transform(uv0, mat.occlusion_texture_transform[0], mat.occlusion_texture_transform[1],
                     mat.occlusion_texture_transform[2], mat.occlusion_texture_transform[3]), one);
    interp_uv uv_id = uv_transform(uv0, 1.0f, 1.0f, 0.0f, 0.0f);
    v4 em4 = sample_or_default(s, mat.emissive_texture, uv_id, one);
    v3 emissive = v3_make(mat.emissive_factor_alpha_cutoff[0] * em4.x, mat.emissive_factor_alpha_cutoff[1] * em4.y,
                          mat.emissive_factor_alpha_cutoff[2] * em4.z);
    v4 sc4 = sample_or_default(s, mat.specular_color_texture, uv_id, one);
    v4 sp4 = sample_or_default(s, mat.specular_texture, uv_id, one);
    v3 spec_color_factor = v3_make(mat.specular_factor[0] * sc4.x, mat.specular_factor[1] * sc4.y, mat.specular_factor[2] * sc4.z);
    float spec_factor_w = mat.specular_factor[3] * sp4.x;

    v3 nnormal = v3_normalize_rcp(norm);
    v3 ntangent = v3_normalize_rcp(tangent);
    v3 bitangent = v3_scale(v3_cross(nnormal, ntangent), (v1->t[3] * v2->t[3]) * v3p->t[3]);
    v3 nrm = v3_normalize_rcp(v3_add(v3_add(v3_scale(ntangent, nmap.x), v3_scale(bitangent, nmap.y)), v3_scale(nnormal, nmap.z)));

    v3 L = v3_normalize(v3_make(u->sun_direction[0], u->sun_direction[1], u->sun_direction[2]));
    int shadow_hit = shadow_any_hit(ss, w_pos, L, 1e-2f, 1000.0f);

    float metallic = clampf(mat.metallic_roughness_normal_occlusion[0] * mr.z, 0.0f, 1.0f);
    float prough = mr.y * mat.metallic_roughness_normal_occlusion[1];
    if (mat.material_type & RDR_MAT_SPEC_GLOSS) prough = 1.0f - mr.w * mat.specular_glossiness_factor[3];
    else prough = clampf(prough, MIN_ROUGHNESS, 1.0f);
    float alpha = prough * prough;

    v3 base = v3_make(albedo.x, albedo.y, albedo.z);
    float fi = (mat.ior - 1.0f) / (mat.ior + 1.0f);
    v3 f0 = v3_splat(fi * fi);
    v3 diffuse = v3_mix(base, v3_splat(0.0f), metallic);
    v3 specular = v3_mix(f0, base, metallic);
    float spec_w = 1.0f;
    if (mat.material_type & RDR_MAT_SPEC) {
        v3 dsf0 = v3_make(fminf(f0.x * spec_color_factor.x, 1.0f), fminf(f0.y * spec_color_factor.y, 1.0f),
                          fminf(f0.z * spec_color_factor.z, 1.0f));
        f0 = v3_mix(dsf0, base, metallic);
        spec_w = spec_factor_w;
    }
    if (mat.material_type & RDR_MAT_SPEC_GLOSS) {
        f0 = v3_make(mat.specular_glossiness_factor[0] * mr.x, mat.specular_glossiness_factor[1] * mr.y,
                     mat.specular_glossiness_factor[2] * mr.z);
        diffuse = v3_scale(base, 1.0f - maxf3(f0.x, f0.y, f0.z));
        specular = f0;
    }
    float reflectance = maxf3(specular.x, specular.y, specular.z);
    float refl90 = clampf(reflectance * 25.0f, 0.0f, 1.0f);
    v3 r0 = specular, r90 = v3_splat(refl90);

    v3 cam = v3_make(u->camera_position[0], u->camera_position[1], u->camera_position[2]);
    v3 v = v3_scale(v3_normalize_rcp(v3_sub(w_pos, cam)), -1.0f);
    v3 h = v3_normalize_rcp(v3_add(L, v));
    float ndv = clampf(fabsf(v3_dot(nrm, v)), 0.001f, 1.0f);
    float ndl = clampf(v3_dot(nrm, L), 0.001f, 1.0f);
    float ndh = clampf(v3_dot(nrm, h), 0.0f, 1.0f);
    float ldh = clampf(v3_dot(L, h), 0.0f, 1.0f);
    float vdh = clampf(v3_dot(v, h), 0.0f, 1.0f);

    const float light = 5.0f;
    v3 f = f_schlick(r0, r90, vdh);
    float g = geometric_occlusion(ndl, ndv, alpha);
    float dd = microfacet_distribution(alpha, ndh);
    v3 diffuse_c = v3_mul(v3_sub(v3_splat(1.0f), f), diffuse_burley(diffuse, ldh, ndl, ndv, alpha));
    v3 spec_c = v3_scale(v3_scale(v3_scale(f, g), dd), 1.0f / (4.0f * ndl * ndv));
    v3 color = v3_scale(v3_add(diffuse_c, spec_c), ndl * light);

    v3 ibl_s = ibl_ggx(env, nrm, v, ndv, prough, r0, spec_w);
    v3 ibl_d = ibl_lambertian(env, ndv, nrm, prough, diffuse, r0, spec_w);
    color = v3_scale(v3_clamp01(v3_add(v3_add(v3_add(ibl_s, ibl_d), color), emissive)), occ.x);

    if (mat.material_type & RDR_MAT_CLEARCOAT) {
        v4 cn4 = sample_or_default(s, mat.clearcoat_texture, uv_id, nmap_def);
        v3 cm = v3_make(cn4.x * 2.0f - 1.0f, cn4.y * 2.0f - 1.0f, cn4.z * 2.0f - 1.0f);
        v3 cnrm = v3_normalize_rcp(v3_add(v3_add(v3_scale(ntangent, cm.x), v3_scale(bitangent, cm.y)), v3_scale(nnormal, cm.z)));
        float cndv = clampf(fabsf(v3_dot(cnrm, v)), 0.001f, 1.0f);
        float cr = mat.clearcoat_factor[1];
        v3 cf0 = v3_splat(fi * fi);
        v3 contrib = ibl_ggx(env, cnrm, v, cndv, cr, cf0, 1.0f);
        v3 fres = f_schlick(r0, r90, cndv);
        color = v3_add(v3_mul(color, v3_sub(v3_splat(1.0f), v3_scale(fres, mat.clearcoat_factor[0]))), contrib);
        color = v3_add(color, v3_scale(clearcoat_punctual(cnrm, v, L, h, vdh, v3_splat(MIN_ROUGHNESS), v3_splat(1.0f), cr), light));
    }
    if (mat.material_type & RDR_MAT_SHEEN) {
        v3 sc = v3_make(mat.sheen_color_roughness_factor[0], mat.sheen_color_roughness_factor[1], mat.sheen_color_roughness_factor[2]);
        float sr = mat.sheen_color_roughness_factor[3];
        color = v3_add(color, ibl_charlie(env, nrm, v, ndv, sr, sc));
        float sd = d_charlie(sr, ndh), sv = v_sheen(ndl, ndv, sr);
        color = v3_add(color, v3_scale(v3_scale(v3_scale(sc, sd), sv), ndl * light));
    }
    color = v3_scale(color, shadow_hit ? 0.35f : 1.0f);
    return color;
}

int orc_shade(const rdr_scene_desc* s, const rdr_ubo* u, const uint64_t* vis, const uint8_t* draw_lod,
              uint32_t row_lo, uint32_t row_hi, uint8_t* out_rgba8, float* out_linear) {
    orc_srgb_lut_init();
    int W = (int)u->screen_width, H = (int)u->screen_height;
    if (row_hi > (uint32_t)H) row_hi = (uint32_t)H;
    env_ctx env;
    env_setup(&env, s);
    shadow_scene ss;
    shadow_scene_build(&ss, s);

#pragma omp parallel for schedule(dynamic, 1)
    for (int iy = (int)row_lo; iy < (int)row_hi; ++iy) {
        for (int ix = 0; ix < W; ++ix) {
            size_t pi = (size_t)iy * W + ix;
            uint64_t key = vis[pi];
            uint8_t* o8 = out_rgba8 + pi * 4;
            float* of = out_linear ? out_linear + pi * 4 : NULL;
            if (key == RDR_VIS_CLEAR) {
                o8[0] = o8[1] = o8[2] = o8[3] = 0;
                if (of) { of[0] = of[1] = of[2] = of[3] = 0.0f; }
                continue;
            }
            uint32_t id = (uint32_t)key, instance = id >> 7, tri = id & 127u;
            uint32_t lo = 0, hi = s->draw_count;
            while (hi - lo > 1) {
                uint32_t mid = (lo + hi) / 2;
                if (s->mesh_draws[mid].meshlet_visibility_offset <= instance) lo = mid; else hi = mid;
            }
            const rdr_mesh_draw* d = &s->mesh_draws[lo];
            const rdr_mesh_lod* Ld = &s->meshes[d->mesh_index].lods[draw_lod[lo]];
            uint32_t meshlet = Ld->meshlet_offset + (instance - d->meshlet_visibility_offset);
            /* the full-screen triangle's interpolated uv at the pixel centre: a product with 1 / W, 1 / H */
            float uvx = ((float)ix + 0.5f) * (1.0f / (float)W);
            float uvy = 1.0f - ((float)iy + 0.5f) * (1.0f / (float)H);
            v3 c = shade_pixel(s, u, &env, &ss, lo, meshlet, tri, uvx, uvy);
            o8[0] = srgb_encode(c.x); o8[1] = srgb_encode(c.y); o8[2] = srgb_encode(c.z); o8[3] = 255;
            if (of) { of[0] = c.x; of[1] = c.y; of[2] = c.z; of[3] = 1.0f; }
        }
    }
    shadow_scene_free(&ss);
    return 0;
}

/* ------------------------------------------------------------------ */
/* How much does taking texture(lambertian, n) at level 0 cost?        */
/* ------------------------------------------------------------------ */
/* shading.frag.glsl:231 samples the irradiance map with an implicit level of detail, which a Vulkan
 * implementation derives from the differences of the coordinate inside each 2 x 2 pixel quad; this restatement
 * (and the kernels) take level 0.  The other implicit-LOD taps need no such question: the BRDF table has one
 * level (bake/src/brdf.rs:59-65 writes num_levels = 1), and the GGX / Charlie taps state their level.
 * This function shades the frame twice: once as orc_shade does, recording the tap's direction per pixel, and once
 * with the level a quad-difference implementation would use: per quad, d/dx from the two pixels of a row and
 * d/dy from the two pixels of a column, transformed to the selected face's coordinates as the specification's
 * "cube map derivative transformation" has it, rho = the longer of the two footprint axes in texels of level 0,
 * level = log2(rho) clamped to the chain.  A quad with an uncovered pixel has no defined differences (the shader
 * has returned there, shading.frag.glsl:324-326): such pixels keep level 0 and are counted in *undefined.
 * out_lod (W*H floats, may be NULL) receives the level used. */
static void cube_face_map(int face, v3 d, float* sc, float* tc, float* ma) {
    switch (face) {
    case 0: *sc = -d.z; *tc = -d.y; *ma = d.x; break;
    case 1: *sc = d.z; *tc = -d.y; *ma = -d.x; break;
    case 2: *sc = d.x; *tc = d.z; *ma = d.y; break;
    case 3: *sc = d.x; *tc = -d.z; *ma = -d.y; break;
    case 4: *sc = d.x; *tc = -d.y; *ma = d.z; break;
    default: *sc = -d.x; *tc = -d.y; *ma = -d.z; break;
    }
}

int orc_shade_quad_lod(const rdr_scene_desc* s, const rdr_ubo* u, const uint64_t* vis, const uint8_t* draw_lod,
                       uint8_t* out_rgba8, float* out_lod, uint64_t* undefined) {
    int W = (int)u->screen_width, H = (int)u->screen_height;
    size_t npx = (size_t)W * H;
    float* dirs = (float*)malloc(npx * 3 * sizeof(float));
    float* lods = (float*)calloc(npx, sizeof(float));
    if (!dirs || !lods) { free(dirs); free(lods); return 1; }
    orc_srgb_lut_init();
    env_ctx env;
    env_setup(&env, s);
    shadow_scene ss;
    shadow_scene_build(&ss, s);
    uint64_t undef = 0;
    for (int pass = 0; pass < 2; ++pass) {
#pragma omp parallel for schedule(dynamic, 1)
        for (int iy = 0; iy < H; ++iy) {
            for (int ix = 0; ix < W; ++ix) {
                size_t pi = (size_t)iy * W + ix;
                uint64_t key = vis[pi];
                uint8_t* o8 = out_rgba8 + pi * 4;
                if (key == RDR_VIS_CLEAR) {
                    o8[0] = o8[1] = o8[2] = o8[3] = 0;
                    continue;
                }
                uint32_t id = (uint32_t)key, instance = id >> 7, tri = id & 127u;
                uint32_t lo = 0, hi = s->draw_count;
                while (hi - lo > 1) {
                    uint32_t mid = (lo + hi) / 2;
                    if (s->mesh_draws[mid].meshlet_visibility_offset <= instance) lo = mid; else hi = mid;
                }
                const rdr_mesh_draw* d = &s->mesh_draws[lo];
                const rdr_mesh_lod* Ld = &s->meshes[d->mesh_index].lods[draw_lod[lo]];
                uint32_t meshlet = Ld->meshlet_offset + (instance - d->meshlet_visibility_offset);
                float uvx = ((float)ix + 0.5f) * (1.0f / (float)W);
                float uvy = 1.0f - ((float)iy + 0.5f) * (1.0f / (float)H);
                dbg_lambert_lod = pass ? lods[pi] : 0.0f;
                v3 c = shade_pixel(s, u, &env, &ss, lo, meshlet, tri, uvx, uvy);
                dbg_lambert_lod = 0.0f;
                if (!pass) { dirs[pi * 3] = dbg_lambert_dir.x; dirs[pi * 3 + 1] = dbg_lambert_dir.y; dirs[pi * 3 + 2] = dbg_lambert_dir.z; }
                o8[0] = srgb_encode(c.x); o8[1] = srgb_encode(c.y); o8[2] = srgb_encode(c.z); o8[3] = 255;
            }
        }
        if (pass) break;
        /* the level per 2 x 2 quad (quads are aligned to even pixel coordinates) */
        const float n0 = (float)env.lambertian.size, max_lod = (float)(env.lambertian.mip_count - 1);
        for (int qy = 0; qy < H; qy += 2) {
            for (int qx = 0; qx < W; qx += 2) {
                int whole = qx + 1 < W && qy + 1 < H;
                for (int k = 0; k < 4 && whole; ++k)
                    whole = vis[(size_t)(qy + (k >> 1)) * W + qx + (k & 1)] != RDR_VIS_CLEAR;
                if (!whole) {
                    for (int k = 0; k < 4; ++k) {
                        int x = qx + (k & 1), y = qy + (k >> 1);
                        if (x < W && y < H && vis[(size_t)y * W + x] != RDR_VIS_CLEAR) undef++;
                    }
                    continue;
                }
                for (int k = 0; k < 4; ++k) {
                    int x = qx + (k & 1), y = qy + (k >> 1);
                    size_t p = (size_t)y * W + x;
                    const float* a = dirs + p * 3;
                    const float* bx = dirs + ((size_t)y * W + (qx + 1)) * 3; /* the row's right pixel minus its left */
                    const float* ax = dirs + ((size_t)y * W + qx) * 3;
                    const float* by = dirs + ((size_t)(qy + 1) * W + x) * 3; /* the column's lower pixel minus its upper */
                    const float* ay = dirs + ((size_t)qy * W + x) * 3;
                    v3 dv = v3_make(a[0], a[1], a[2]);
                    v3 ddx = v3_make(bx[0] - ax[0], bx[1] - ax[1], bx[2] - ax[2]);
                    v3 ddy = v3_make(by[0] - ay[0], by[1] - ay[1], by[2] - ay[2]);
                    float sc, tc, ma;
                    int face = cube_face(dv, &sc, &tc, &ma);
                    float scx, tcx, max_, scy, tcy, may;
                    cube_face_map(face, dv, &sc, &tc, &ma);
                    cube_face_map(face, ddx, &scx, &tcx, &max_);
                    cube_face_map(face, ddy, &scy, &tcy, &may);
                    float inv = 1.0f / (ma * ma);
                    float dsdx = 0.5f * (scx * ma - sc * max_) * inv, dtdx = 0.5f * (tcx * ma - tc * max_) * inv;
                    float dsdy = 0.5f * (scy * ma - sc * may) * inv, dtdy = 0.5f * (tcy * ma - tc * may) * inv;
                    float rx = sqrtf(dsdx * dsdx + dtdx * dtdx) * n0, ry = sqrtf(dsdy * dsdy + dtdy * dtdy) * n0;
                    float rho = fmaxf(rx, ry);
                    float lod = rho > 0.0f ? log2f(rho) : 0.0f;
                    if (!(lod > 0.0f)) lod = 0.0f;
                    if (lod > max_lod) lod = max_lod;
                    lods[p] = lod;
                }
            }
        }
    }
    if (out_lod) memcpy(out_lod, lods, npx * sizeof(float));
    if (undefined) *undefined = undef;
    shadow_scene_free(&ss);
    free(dirs); free(lods);
    return 0;
}

/* Skybox pass on the pixels no geometry covers (renderer/src/renderer/pass.rs:397-445).  The
 * reference draws a cube of half-size 10 000 centred on the camera (cubemap.vert.glsl:11-46)
 * with the depth test GREATER against the frame's depth image (lava/src/pipelines.rs:189-191):
 * it survives exactly where the depth is still the clear value 0, and the interpolated `dir`
 * (cubemap.vert.glsl:46) is the pixel's view ray up to a positive scale, so
 * texture(cubemap, dir) (cubemap.frag.glsl:11) is the cubemap seen along that ray.  The
 * implicit LOD is taken as 0, like the other implicit-LOD fetches of this restatement.  The
 * colour goes through the swapchain's sRGB encode, alpha is stored as is. */
int orc_skybox(const rdr_scene_desc* s, const rdr_ubo* u, const uint64_t* vis, uint32_t row_lo, uint32_t row_hi,
               uint8_t* out_rgba8) {
    int W = (int)u->screen_width, H = (int)u->screen_height;
    if (row_hi > (uint32_t)H) row_hi = (uint32_t)H;
    if (!s->env || !s->env->skybox.data || s->env->skybox.mip_count == 0) return 0;
    const rdr_cubemap_desc* cm = &s->env->skybox;
#pragma omp parallel for schedule(static)
    for (int iy = (int)row_lo; iy < (int)row_hi; ++iy) {
        for (int ix = 0; ix < W; ++ix) {
            size_t pi = (size_t)iy * W + ix;
            if (vis[pi] != RDR_VIS_CLEAR) continue;
            float nx = (((float)ix + 0.5f) / (float)W) * 2.0f - 1.0f;
            float ny = (1.0f - ((float)iy + 0.5f) / (float)H) * 2.0f - 1.0f;
            /* view-space ray (x/z, y/z, 1) from the projection's scales, then the view rotation transposed */
            float vx = nx / u->proj[0], vy = ny / u->proj[5];
            v3 d;
            d.x = (u->view[0] * vx + u->view[1] * vy) + u->view[2];
            d.y = (u->view[4] * vx + u->view[5] * vy) + u->view[6];
            d.z = (u->view[8] * vx + u->view[9] * vy) + u->view[10];
            v4 c = orc_cube_sample_lod(cm, d, 0.0f);
            uint8_t* o8 = out_rgba8 + pi * 4;
            o8[0] = srgb_encode(c.x); o8[1] = srgb_encode(c.y); o8[2] = srgb_encode(c.z);
            o8[3] = (uint8_t)floorf(clampf(c.w, 0.0f, 1.0f) * 255.0f + 0.5f);
        }
    }
    return 0;
}
