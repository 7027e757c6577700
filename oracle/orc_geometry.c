/*
 * orc_geometry.c — CPU restatement of the reference's geometry passes:
 * per-draw cull, per-meshlet cull, vertex transform, per-triangle cull and
 * rasterisation into a packed 64-bit depth|ID visibility buffer.
 *
 * TEST INFRASTRUCTURE ONLY (see orc_math.h).  PARITY UNPINNED: the reference
 * ships no tests, golden vectors or fixtures for this path, and its
 * rasteriser/sampler are a Vulkan driver's; the checks that pin this file are
 * hand-derived known-answer scenes under tests/.
 *
 * Followed, function by function (paths relative to the reference tree):
 *   orc_build_ubo        renderer/src/renderer.rs:1148-1197, scenery/src/camera.rs:12-14,
 *                        renderer/src/math.rs:3-5,17-25 (+ glam 0.30.4 look_to_lh,
 *                        perspective_infinite_reverse_lh — not vendored, formulas restated)
 *   draw_cull            assets/shaders/cull.comp.glsl:54-130
 *   meshlet_cull         assets/shaders/shader.task.glsl:59-137
 *   transform + tri cull assets/shaders/shader.mesh.glsl:72-131
 *   fixed-function state lava/src/pipelines.rs:149-191 (flipped viewport, CLOCKWISE
 *                        front, cull FRONT / NONE, depth GREATER, clear 0)
 *   alpha test           assets/shaders/shader.frag.glsl:55-69
 *
 * The two-pass early/late occlusion split (cull.comp.glsl LATE, depth pyramid)
 * is a conservative optimisation that never changes the image, so one pass over
 * every draw is restated here.
 *
 * Rasterisation rules this oracle fixes where Vulkan leaves latitude:
 *   - the perspective division is a product with the correctly rounded reciprocal of w;
 *   - vertices snap to 1/256 pixel (round-to-nearest-even);
 *   - coverage is exact 64-bit integer edge functions at pixel centres with the
 *     top-left rule;
 *   - depth is z/w interpolated affinely in screen space from the integer
 *     barycentrics, clamped to [0,1];
 *   - triangles with a vertex outside the near plane (w < z) or outside a
 *     guard band of 8 NDC units are clipped in clip space (Sutherland-Hodgman,
 *     new vertices always interpolated from the inside vertex) and fanned;
 *   - round() in the small-triangle test is round-to-nearest-even;
 *   - depth ties resolve to the lower ID (64-bit min), not API order.
 */
#include <stdlib.h>
#include <stdio.h>
#include "orc_math.h"
#include "orc_texture.h"
#include "rdr_oracle.h"

#define GUARD_BAND 8.0f

/* ------------------------------------------------------------------ */
/* UBO                                                                 */
/* ------------------------------------------------------------------ */

static uint32_t power_2_floor(uint32_t x) {
    uint32_t r = 1;
    while (r * 2 < x) r *= 2;
    return r;
}

/* glam Vec3::normalize(): multiply by the reciprocal length. */
static v3 glam_normalize(v3 a) {
    float rl = 1.0f / v3_len(a);
    return v3_scale(a, rl);
}

void orc_build_ubo(uint32_t width, uint32_t height, const float cam_pos[3], const float cam_dir[3],
                   const float sun_dir[3], uint32_t draw_count, uint32_t lod_enabled,
                   uint32_t cull_enabled, rdr_ubo* u) {
    memset(u, 0, sizeof(*u));
    v3 eye = v3_make(cam_pos[0], cam_pos[1], cam_pos[2]);
    v3 dir = v3_make(cam_dir[0], cam_dir[1], cam_dir[2]);
    v3 up = v3_make(0.0f, 1.0f, 0.0f);
    /* look_to_lh(eye, -direction, Y) == look_to_rh(eye, direction, Y) */
    v3 f = glam_normalize(dir);
    v3 s = glam_normalize(v3_cross(f, up));
    v3 v = v3_cross(s, f);
    float* V = u->view;
    V[0] = s.x; V[1] = v.x; V[2] = -f.x; V[3] = 0.0f;
    V[4] = s.y; V[5] = v.y; V[6] = -f.y; V[7] = 0.0f;
    V[8] = s.z; V[9] = v.z; V[10] = -f.z; V[11] = 0.0f;
    V[12] = -v3_dot(eye, s); V[13] = -v3_dot(eye, v); V[14] = v3_dot(eye, f); V[15] = 1.0f;

    float near_plane = 0.01f;
    float fov = 45.0f * (3.14159265358979323846f / 180.0f);
    float aspect = (float)width / (float)height;
    float sn = sinf(0.5f * fov), cs = cosf(0.5f * fov);
    float h = cs / sn;
    float w = h / aspect;
    float* P = u->proj;
    P[0] = w; P[5] = h; P[11] = 1.0f; P[14] = near_plane; /* everything else 0 */
    P[0] *= -1.0f;

    m4_mul_m4(u->proj, u->view, u->view_proj);
    u->camera_position[0] = eye.x; u->camera_position[1] = eye.y; u->camera_position[2] = eye.z;

    /* row(3) + row(0), row(3) + row(1) of proj, normalised by the xyz length */
    float fxv[4], fyv[4];
    for (int c = 0; c < 4; ++c) {
        fxv[c] = P[c * 4 + 3] + P[c * 4 + 0];
        fyv[c] = P[c * 4 + 3] + P[c * 4 + 1];
    }
    float lx = v3_len(v3_make(fxv[0], fxv[1], fxv[2]));
    float ly = v3_len(v3_make(fyv[0], fyv[1], fyv[2]));
    u->frustum[0] = fxv[0] / lx;
    u->frustum[1] = fxv[2] / lx;
    u->frustum[2] = fyv[1] / ly;
    u->frustum[3] = fyv[2] / ly;

    u->near_plane = near_plane;
    u->far_plane = 500.0f;
    u->lod_enabled = lod_enabled ? 1u : 0u;
    u->cull_enabled = cull_enabled ? 1u : 0u;
    u->draw_count = draw_count;
    u->depth_pyramid_width = (float)power_2_floor(width);
    u->depth_pyramid_height = (float)power_2_floor(height);
    u->lod_target = (1.0f / P[5]) * (1.0f / (float)height);
    u->screen_width = (float)width;
    u->screen_height = (float)height;
    u->sun_direction[0] = sun_dir[0]; u->sun_direction[1] = sun_dir[1];
    u->sun_direction[2] = sun_dir[2]; u->sun_direction[3] = 0.0f;
}

/* ------------------------------------------------------------------ */
/* Shared per-frame context                                            */
/* ------------------------------------------------------------------ */

typedef struct {
    const rdr_scene_desc* s;
    const rdr_ubo* u;
    int W, H;
    uint64_t* vis;
    /* counters (per thread copies are summed by the caller) */
    uint64_t tris_tested, tris_raster, tris_big, fragments, vertices;
} geo_ctx;

static inline void vis_min(uint64_t* p, uint64_t key) {
    uint64_t cur = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (key < cur) {
        if (__atomic_compare_exchange_n(p, &cur, key, 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) break;
    }
}

typedef struct {
    float x, y, z, w; /* clip space */
    float uv[4];      /* vertex uv0.xy | uv1.zw */
} clip_vert;

typedef struct {
    int32_t X, Y; /* 24.8 fixed point framebuffer position, y down */
    float z;      /* z/w */
    float inv_w;
    float uv[4];
} snap_vert;

typedef struct {
    int enabled;  /* alpha-mask pass */
    int has_tex;
    const rdr_image_desc* tex;
    int uv_set;
    float sx, sy, ox, oy; /* albedo_texture_transform */
    float factor_a;       /* base_color_factor.a */
    float cutoff;
} alpha_test;

static snap_vert project_vertex(const geo_ctx* c, const clip_vert* v) {
    snap_vert r;
    /* the perspective division: one correctly rounded reciprocal of w, then products (GLSL does not fix the
       rounding of x / w; a division by a shared scalar is lowered this way by shader compilers) */
    float rw = 1.0f / v->w;
    float ndcx = v->x * rw, ndcy = v->y * rw;
    float hw = (float)c->W * 0.5f, hh = (float)c->H * 0.5f;
    float xf = hw * ndcx + hw;
    float yf = -hh * ndcy + hh; /* viewport {y = H, height = -H} */
    r.X = (int32_t)rintf(xf * 256.0f);
    r.Y = (int32_t)rintf(yf * 256.0f);
    r.z = v->z * rw;
    r.inv_w = rw;
    memcpy(r.uv, v->uv, sizeof(r.uv));
    return r;
}

static inline int edge_top_left(int32_t dx, int32_t dy) { return dy < 0 || (dy == 0 && dx > 0); }

/* Perspective-correct uv set at the sample whose three edge values are e[]. */
static void persp_uv(const snap_vert* v0, const snap_vert* v1, const snap_vert* v2, int64_t e0,
                     int64_t e1, int64_t e2, int set, float* u, float* v) {
    float b0 = (float)e0 * v0->inv_w, b1 = (float)e1 * v1->inv_w, b2 = (float)e2 * v2->inv_w;
    float sum = (b0 + b1) + b2;
    int o = set ? 2 : 0;
    *u = ((b0 * v0->uv[o] + b1 * v1->uv[o]) + b2 * v2->uv[o]) / sum;
    *v = ((b0 * v0->uv[o + 1] + b1 * v1->uv[o + 1]) + b2 * v2->uv[o + 1]) / sum;
}

/* Rasterise one snapped triangle. cull_front: opaque pipeline (cull FRONT). */
static void raster_triangle(geo_ctx* c, snap_vert v0, snap_vert v1, snap_vert v2, int cull_front,
                            uint32_t id, const alpha_test* at) {
    int64_t A = (int64_t)(v1.X - v0.X) * (v2.Y - v0.Y) - (int64_t)(v1.Y - v0.Y) * (v2.X - v0.X);
    if (A == 0) return;
    if (A > 0) {
        /* clockwise on a y-down screen == front-facing under frontFace = CLOCKWISE */
        if (cull_front) return;
    } else {
        snap_vert t = v1; v1 = v2; v2 = t;
        A = -A;
    }
    int32_t minX = v0.X < v1.X ? v0.X : v1.X; if (v2.X < minX) minX = v2.X;
    int32_t maxX = v0.X > v1.X ? v0.X : v1.X; if (v2.X > maxX) maxX = v2.X;
    int32_t minY = v0.Y < v1.Y ? v0.Y : v1.Y; if (v2.Y < minY) minY = v2.Y;
    int32_t maxY = v0.Y > v1.Y ? v0.Y : v1.Y; if (v2.Y > maxY) maxY = v2.Y;
    /* pixel px has its centre at px*256 + 128 */
    int px0 = (minX - 128 + 255) >> 8, px1 = (maxX - 128) >> 8;
    int py0 = (minY - 128 + 255) >> 8, py1 = (maxY - 128) >> 8;
    if (px0 < 0) px0 = 0; if (py0 < 0) py0 = 0;
    if (px1 > c->W - 1) px1 = c->W - 1; if (py1 > c->H - 1) py1 = c->H - 1;
    if (px0 > px1 || py0 > py1) return;

    int32_t dx01 = v1.X - v0.X, dy01 = v1.Y - v0.Y;
    int32_t dx12 = v2.X - v1.X, dy12 = v2.Y - v1.Y;
    int32_t dx20 = v0.X - v2.X, dy20 = v0.Y - v2.Y;
    int tl01 = edge_top_left(dx01, dy01), tl12 = edge_top_left(dx12, dy12),
        tl20 = edge_top_left(dx20, dy20);
    float inv_area = 1.0f / (float)A;
    float dz1 = v1.z - v0.z, dz2 = v2.z - v0.z;

    for (int py = py0; py <= py1; ++py) {
        int64_t Py = (int64_t)py * 256 + 128;
        for (int px = px0; px <= px1; ++px) {
            int64_t Px = (int64_t)px * 256 + 128;
            int64_t e01 = (int64_t)dx01 * (Py - v0.Y) - (int64_t)dy01 * (Px - v0.X); /* weight of v2 */
            int64_t e12 = (int64_t)dx12 * (Py - v1.Y) - (int64_t)dy12 * (Px - v1.X); /* weight of v0 */
            int64_t e20 = (int64_t)dx20 * (Py - v2.Y) - (int64_t)dy20 * (Px - v2.X); /* weight of v1 */
            if (!(e01 > 0 || (e01 == 0 && tl01))) continue;
            if (!(e12 > 0 || (e12 == 0 && tl12))) continue;
            if (!(e20 > 0 || (e20 == 0 && tl20))) continue;
            if (at->enabled) {
                float alpha = 1.0f;
                if (at->has_tex) {
                    float u, v, ux, vx, uy, vy;
                    persp_uv(&v0, &v1, &v2, e12, e20, e01, at->uv_set, &u, &v);
                    persp_uv(&v0, &v1, &v2, e12 - (int64_t)dy12 * 256, e20 - (int64_t)dy20 * 256,
                             e01 - (int64_t)dy01 * 256, at->uv_set, &ux, &vx);
                    persp_uv(&v0, &v1, &v2, e12 + (int64_t)dx12 * 256, e20 + (int64_t)dx20 * 256,
                             e01 + (int64_t)dx01 * 256, at->uv_set, &uy, &vy);
                    u = u * at->sx + at->ox; v = v * at->sy + at->oy;
                    ux = ux * at->sx + at->ox; vx = vx * at->sy + at->oy;
                    uy = uy * at->sx + at->ox; vy = vy * at->sy + at->oy;
                    alpha = orc_image_sample_grad(at->tex, u, v, ux - u, vx - v, uy - u, vy - v).w;
                }
                alpha = alpha * at->factor_a;
                if (alpha < at->cutoff) continue;
            }
            float l1 = (float)e20 * inv_area, l2 = (float)e01 * inv_area;
            float z = fmaf(l2, dz2, fmaf(l1, dz1, v0.z));
            z = clampf(z, 0.0f, 1.0f);
            uint64_t key = ((uint64_t)(0x7FFFFFFFu - f32_bits(z)) << 32) | id;
            vis_min(&c->vis[(size_t)py * c->W + px], key);
            c->fragments++;
        }
    }
}

static clip_vert clip_lerp(const clip_vert* in, const clip_vert* out, float din, float dout) {
    float t = din / (din - dout);
    clip_vert r;
    r.x = in->x + t * (out->x - in->x);
    r.y = in->y + t * (out->y - in->y);
    r.z = in->z + t * (out->z - in->z);
    r.w = in->w + t * (out->w - in->w);
    for (int k = 0; k < 4; ++k) r.uv[k] = in->uv[k] + t * (out->uv[k] - in->uv[k]);
    return r;
}

static inline float plane_dist(const clip_vert* v, int plane) {
    switch (plane) {
    case 0: return v->w - v->z;              /* near: z <= w */
    case 1: return GUARD_BAND * v->w - v->x; /* x <= G w    */
    case 2: return GUARD_BAND * v->w + v->x;
    case 3: return GUARD_BAND * v->w - v->y;
    default: return GUARD_BAND * v->w + v->y;
    }
}

static inline int vert_needs_clip(const clip_vert* v) {
    float g = GUARD_BAND * v->w;
    return !(v->w >= v->z) || !(fabsf(v->x) <= g) || !(fabsf(v->y) <= g);
}

/* The fixed-function half: clip, project, snap, face-cull, rasterise. */
static void draw_triangle(geo_ctx* c, const clip_vert* a, const clip_vert* b, const clip_vert* cc,
                          int cull_front, uint32_t id, const alpha_test* at) {
    if (!(vert_needs_clip(a) || vert_needs_clip(b) || vert_needs_clip(cc))) {
        raster_triangle(c, project_vertex(c, a), project_vertex(c, b), project_vertex(c, cc),
                        cull_front, id, at);
        return;
    }
    c->tris_big++;
    clip_vert poly[12], tmp[12];
    int n = 3;
    poly[0] = *a; poly[1] = *b; poly[2] = *cc;
    for (int plane = 0; plane < 5 && n > 0; ++plane) {
        int m = 0;
        for (int i = 0; i < n; ++i) {
            const clip_vert* p = &poly[i];
            const clip_vert* q = &poly[(i + 1) % n];
            float dp = plane_dist(p, plane), dq = plane_dist(q, plane);
            int pin = dp >= 0.0f, qin = dq >= 0.0f;
            if (pin) tmp[m++] = *p;
            if (pin && !qin) tmp[m++] = clip_lerp(p, q, dp, dq);
            else if (!pin && qin) tmp[m++] = clip_lerp(q, p, dq, dp);
        }
        memcpy(poly, tmp, sizeof(clip_vert) * (size_t)m);
        n = m;
    }
    if (n < 3) return;
    snap_vert s0 = project_vertex(c, &poly[0]);
    snap_vert sp = project_vertex(c, &poly[1]);
    for (int i = 2; i < n; ++i) {
        snap_vert sn = project_vertex(c, &poly[i]);
        raster_triangle(c, s0, sp, sn, cull_front, id, at);
        sp = sn;
    }
}

/* ------------------------------------------------------------------ */
/* Culling                                                             */
/* ------------------------------------------------------------------ */

static float model_max_scale(const float* m) {
    float sx = v3_len(v3_make(m[0], m[1], m[2]));
    float sy = v3_len(v3_make(m[4], m[5], m[6]));
    float sz = v3_len(v3_make(m[8], m[9], m[10]));
    return fmaxf(sx, fmaxf(sy, sz));
}

static v3 to_view(const rdr_ubo* u, const float* model, v3 p, float w) {
    v4 p4 = {p.x, p.y, p.z, w};
    v4 world = m4_mul_v4(model, p4);
    v4 view = m4_mul_v4(u->view, world);
    return v3_make(view.x, view.y, view.z);
}

/* cull.comp.glsl:72-109 — returns 1 when the draw emits task commands. */
static int draw_cull(const rdr_scene_desc* s, const rdr_ubo* u, uint32_t di, uint32_t* lod_out,
                     int* visible_out) {
    const rdr_mesh_draw* d = &s->mesh_draws[di];
    const rdr_mesh* mesh = &s->meshes[d->mesh_index];
    const float* model = s->global_transforms + (size_t)d->model_index * 16;
    v3 center = to_view(u, model, v3_make(mesh->center[0], mesh->center[1], mesh->center[2]), 1.0f);
    float max_scale = model_max_scale(model);
    float radius = max_scale * mesh->radius;

    int visible = 1;
    visible = visible && (center.z * -u->frustum[1] + fabsf(center.x) * u->frustum[0] < radius);
    visible = visible && (center.z * -u->frustum[3] + fabsf(center.y) * u->frustum[2] < radius);
    visible = visible && (center.z + radius > -u->near_plane) && (center.z + radius < u->far_plane);
    *visible_out = visible;
    if (!(visible || d->alpha_mode == 1)) return 0;

    uint32_t lod_index = 0;
    float threshold = fmaxf(v3_len(center) - radius, 0.0f) * u->lod_target / max_scale;
    for (uint32_t i = 1; i < mesh->lod_count; ++i)
        if (mesh->lods[i].error < threshold) lod_index = i;
    *lod_out = u->lod_enabled == 1 ? lod_index : 0;
    return 1;
}

/* shader.task.glsl:82-105 */
static int meshlet_cull(const rdr_ubo* u, const float* model, float max_scale, const rdr_meshlet* m) {
    v3 cc = to_view(u, model, v3_make(m->center[0], m->center[1], m->center[2]), 1.0f);
    v3 axis_in = v3_make((float)m->cone_axis[0] / 127.0f, (float)m->cone_axis[1] / 127.0f,
                         (float)m->cone_axis[2] / 127.0f);
    v3 axis = v3_normalize(to_view(u, model, axis_in, 0.0f));
    float radius = max_scale * m->radius;
    float cutoff = (float)m->cone_cutoff / 127.0f;
    int accept = 1;
    accept = accept && !(v3_dot(cc, axis) >= cutoff * v3_len(cc) + radius);
    accept = accept && (cc.z * -u->frustum[1] + cc.x * u->frustum[0] < radius);
    accept = accept && (cc.z * -u->frustum[3] + cc.y * u->frustum[2] < radius);
    accept = accept && (cc.z + radius > -u->near_plane) && (cc.z + radius < u->far_plane);
    return accept;
}

/* shader.mesh.glsl:72-131 followed by the fixed-function stage. */
static void draw_meshlet(geo_ctx* c, uint32_t draw_id, uint32_t meshlet_index, uint32_t instance,
                         const alpha_test* at) {
    const rdr_scene_desc* s = c->s;
    const rdr_ubo* u = c->u;
    const rdr_mesh_draw* d = &s->mesh_draws[draw_id];
    const rdr_mesh* mesh = &s->meshes[d->mesh_index];
    const rdr_meshlet* m = &s->meshlets[meshlet_index];
    const float* model = s->global_transforms + (size_t)d->model_index * 16;

    clip_vert cv[64];
    float scx[64], scy[64];
    c->vertices += m->vertex_count;
    for (uint32_t i = 0; i < m->vertex_count; ++i) {
        uint32_t vi = s->meshlet_vertices[m->vertex_offset + i] + mesh->vertex_offset;
        const rdr_vertex* vtx = &s->vertices[vi];
        v4 p = {vtx->p[0], vtx->p[1], vtx->p[2], 1.0f};
        v4 clip = m4_mul_v4(u->view_proj, m4_mul_v4(model, p));
        cv[i].x = clip.x; cv[i].y = clip.y; cv[i].z = clip.z; cv[i].w = clip.w;
        memcpy(cv[i].uv, vtx->uv, sizeof(float) * 4);
        float rw = 1.0f / clip.w; /* shader.mesh.glsl:87-88: clip.xy / clip.w, as a product with the reciprocal */
        scx[i] = (clip.x * rw * 0.5f + 0.5f) * u->screen_width;
        scy[i] = (clip.y * rw * 0.5f + 0.5f) * u->screen_height;
    }
    for (uint32_t t = 0; t < m->triangle_count; ++t) {
        uint32_t ia = s->meshlet_triangles[m->triangle_offset + t * 3 + 0];
        uint32_t ib = s->meshlet_triangles[m->triangle_offset + t * 3 + 1];
        uint32_t ic = s->meshlet_triangles[m->triangle_offset + t * 3 + 2];
        c->tris_tested++;
        float ebax = scx[ib] - scx[ia], ebay = scy[ib] - scy[ia];
        float ecax = scx[ic] - scx[ia], ecay = scy[ic] - scy[ia];
        int culled = ebax * ecay <= ebay * ecax;
        float bminx = fminf(scx[ia], fminf(scx[ib], scx[ic]));
        float bminy = fminf(scy[ia], fminf(scy[ib], scy[ic]));
        float bmaxx = fmaxf(scx[ia], fmaxf(scx[ib], scx[ic]));
        float bmaxy = fmaxf(scy[ia], fmaxf(scy[ib], scy[ic]));
        const float sbprec = 1.0f / 256.0f;
        culled = culled || (rintf(bminx - sbprec) == rintf(bmaxx) ||
                            rintf(bminy) == rintf(bmaxy + sbprec));
        culled = culled && (cv[ia].w > 0.0f && cv[ib].w > 0.0f && cv[ic].w > 0.0f);
        if (culled) continue;
        c->tris_raster++;
        uint32_t id = (instance << 7) | t;
        draw_triangle(c, &cv[ia], &cv[ib], &cv[ic], d->alpha_mode == 0, id, at);
    }
}

static void make_alpha_test(const rdr_scene_desc* s, const rdr_mesh_draw* d, alpha_test* at) {
    memset(at, 0, sizeof(*at));
    if (d->alpha_mode != 1) return;
    at->enabled = 1;
    /* materials[0xFFFF] past the end reads zeros under robustBufferAccess
       (lava/src/device.rs:167): alpha 0 against cutoff 0 never discards. */
    if (d->material_index >= s->material_count) return;
    const rdr_material* mat = &s->materials[d->material_index];
    at->factor_a = mat->base_color_factor[3];
    at->cutoff = mat->emissive_factor_alpha_cutoff[3];
    if (mat->albedo_texture != RDR_NO_TEXTURE && mat->albedo_texture < s->image_count &&
        s->images[mat->albedo_texture].data == NULL) {
        at->factor_a = 0.0f; /* an image without data samples as zero: alpha = 0 * factor */
    } else if (mat->albedo_texture != RDR_NO_TEXTURE && mat->albedo_texture < s->image_count) {
        at->has_tex = 1;
        at->tex = &s->images[mat->albedo_texture];
        at->uv_set = mat->albedo_texture_uv == 0 ? 0 : 1;
        at->sx = mat->albedo_texture_transform[0]; at->sy = mat->albedo_texture_transform[1];
        at->ox = mat->albedo_texture_transform[2]; at->oy = mat->albedo_texture_transform[3];
    }
}

int orc_render_visbuffer(const rdr_scene_desc* s, const rdr_ubo* u, uint32_t part_lo,
                         uint32_t part_hi, uint64_t* vis, uint8_t* draw_lod, rdr_stats* stats) {
    return orc_render_visbuffer_interleaved(s, u, part_lo, part_hi, 64, 0, 1, vis, draw_lod, stats);
}

/* The sort-last share of rdr_set_partition_interleaved on top of the range: chunk j of draw d
 * (chunks of `chunk` consecutive meshlets of the selected level of detail) belongs to rank
 * (j + d) % world. */
int orc_render_visbuffer_interleaved(const rdr_scene_desc* s, const rdr_ubo* u, uint32_t part_lo,
                                     uint32_t part_hi, uint32_t chunk, uint32_t rank, uint32_t world,
                                     uint64_t* vis, uint8_t* draw_lod, rdr_stats* stats) {
    if (chunk == 0 || world == 0 || rank >= world) return 1;
    orc_srgb_lut_init();
    int W = (int)u->screen_width, H = (int)u->screen_height;
    size_t npx = (size_t)W * H;
    for (size_t i = 0; i < npx; ++i) vis[i] = RDR_VIS_CLEAR;
    uint64_t draws_visible = 0, meshlets_tested = 0, meshlets_visible = 0, tris_submitted = 0;
    uint64_t tris_tested = 0, tris_raster = 0, tris_big = 0, fragments = 0, vertices = 0;

#pragma omp parallel for schedule(dynamic, 4) reduction(+ : draws_visible, meshlets_tested, \
    meshlets_visible, tris_submitted, tris_tested, tris_raster, tris_big, fragments, vertices)
    for (int64_t di = 0; di < (int64_t)s->draw_count; ++di) {
        uint32_t lod = 0;
        int visible = 0;
        if (draw_lod) draw_lod[di] = 0xFF;
        if (!draw_cull(s, u, (uint32_t)di, &lod, &visible)) continue;
        if (draw_lod) draw_lod[di] = (uint8_t)lod;
        draws_visible++;
        const rdr_mesh_draw* d = &s->mesh_draws[di];
        const rdr_mesh* mesh = &s->meshes[d->mesh_index];
        const rdr_mesh_lod* L = &mesh->lods[lod];
        const float* model = s->global_transforms + (size_t)d->model_index * 16;
        float max_scale = model_max_scale(model);
        alpha_test at;
        make_alpha_test(s, d, &at);
        geo_ctx c = {s, u, W, H, vis, 0, 0, 0, 0, 0};
        for (uint32_t k = 0; k < L->meshlet_count; ++k) {
            uint32_t instance = d->meshlet_visibility_offset + k;
            if (instance < part_lo || instance >= part_hi) continue;
            if (world > 1 && (k / chunk + (uint32_t)di) % world != rank) continue;
            uint32_t mi = L->meshlet_offset + k;
            tris_submitted += s->meshlets[mi].triangle_count;
            meshlets_tested++;
            if (!meshlet_cull(u, model, max_scale, &s->meshlets[mi])) continue;
            meshlets_visible++;
            draw_meshlet(&c, (uint32_t)di, mi, instance, &at);
        }
        tris_tested += c.tris_tested; tris_raster += c.tris_raster;
        tris_big += c.tris_big; fragments += c.fragments; vertices += c.vertices;
    }
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->draws_visible = draws_visible;
        stats->meshlets_tested = meshlets_tested;
        stats->meshlets_visible = meshlets_visible;
        stats->triangles_submitted = tris_submitted;
        stats->triangles_tested = tris_tested;
        stats->triangles_rasterised = tris_raster;
        stats->triangles_big = tris_big;
        stats->fragments = fragments;
        stats->vertices_transformed = vertices;
    }
    return 0;
}

/* Packed key -> the reference's visibility-buffer texel
 * (assets/shaders/shader.frag.glsl:71-74; clear value pass.rs:250). */
int orc_resolve(const rdr_scene_desc* s, uint32_t width, uint32_t height, const uint64_t* vis,
                const uint8_t* draw_lod, uint32_t* out_xy) {
    size_t npx = (size_t)width * height;
    for (size_t i = 0; i < npx; ++i) {
        uint64_t key = vis[i];
        if (key == RDR_VIS_CLEAR) {
            out_xy[2 * i] = 0xFFFFFFFFu; out_xy[2 * i + 1] = 0xFFFFFFFFu;
            continue;
        }
        uint32_t id = (uint32_t)key;
        uint32_t instance = id >> 7, tri = id & 127u;
        /* last draw whose meshlet_visibility_offset <= instance */
        uint32_t lo = 0, hi = s->draw_count;
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) / 2;
            if (s->mesh_draws[mid].meshlet_visibility_offset <= instance) lo = mid; else hi = mid;
        }
        const rdr_mesh_draw* d = &s->mesh_draws[lo];
        const rdr_mesh_lod* L = &s->meshes[d->mesh_index].lods[draw_lod[lo]];
        uint32_t meshlet = L->meshlet_offset + (instance - d->meshlet_visibility_offset);
        out_xy[2 * i] = (meshlet << 8) | tri;
        out_xy[2 * i + 1] = lo;
    }
    return 0;
}
