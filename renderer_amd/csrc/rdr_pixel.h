// rdr_pixel.h — what the shadow pass and the shading pass share: decoding a visibility
// key, the covered-pixel list entry, and the analytic barycentrics of a pixel inside its
// triangle (assets/shaders/shading.frag.glsl:81-171).
#pragma once
#include "rdr_device.h"

namespace {

// ---- barycentrics -----------------------------------------------------------------
struct BaryDeriv {
    vec3 lambda, ddx, ddy;
};

// Correctly rounded a / b, whatever the translation unit is built with.  The shadow pass is built with exact
// division throughout, and this is then a plain division; the shading pass is built with the approximate fp32
// division (RDR_FAST_DIVISION) and still has to divide exactly here: the barycentrics of a triangle many screens
// large cancel catastrophically and turn a last-bit difference in 1 / w into a few per cent of colour.  The
// sequence is the hardware's own IEEE one (scale, reciprocal, two Newton steps, residual, fix-up).
#ifdef RDR_FAST_DIVISION
RDR_DEV float div_exact(float a, float b) {
    bool scaled_d, scaled_n;
    const float d = __builtin_amdgcn_div_scalef(a, b, false, &scaled_d);
    const float n = __builtin_amdgcn_div_scalef(a, b, true, &scaled_n);
    float r = __builtin_amdgcn_rcpf(d);
    const float e0 = fmaf(-d, r, 1.0f);
    r = fmaf(e0, r, r);
    float q = n * r;
    const float e1 = fmaf(-d, q, n);
    q = fmaf(e1, r, q);
    const float e2 = fmaf(-d, q, n);
    return __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(e2, r, q, scaled_n), b, a);
}
#else
RDR_DEV float div_exact(float a, float b) { return a / b; }
#endif

// `sx`, `sy`: 2 / window width, 2 / window height (correctly rounded)
RDR_DEV BaryDeriv calc_full_bary(vec4 pt0, vec4 pt1, vec4 pt2, float ndcx, float ndcy, float sx, float sy) {
    BaryDeriv r;
    const vec3 inv_w = mk3(div_exact(1.0f, pt0.w), div_exact(1.0f, pt1.w), div_exact(1.0f, pt2.w));
    const float n0x = pt0.x * inv_w.x, n0y = pt0.y * inv_w.x;
    const float n1x = pt1.x * inv_w.y, n1y = pt1.y * inv_w.y;
    const float n2x = pt2.x * inv_w.z, n2y = pt2.y * inv_w.z;
    const float ax = n2x - n1x, ay = n2y - n1y, bx = n0x - n1x, by = n0y - n1y;
    const float inv_det = div_exact(1.0f, ax * by - bx * ay);
    r.ddx = (mk3(n1y - n2y, n2y - n0y, n0y - n1y) * inv_det) * inv_w;
    r.ddy = (mk3(n2x - n1x, n0x - n2x, n1x - n0x) * inv_det) * inv_w;
    float ddx_sum = (r.ddx.x + r.ddx.y) + r.ddx.z;
    float ddy_sum = (r.ddy.x + r.ddy.y) + r.ddy.z;
    const float dx = ndcx - n0x, dy = ndcy - n0y;
    const float interp_inv_w = (inv_w.x + dx * ddx_sum) + dy * ddy_sum;
    const float interp_w = div_exact(1.0f, interp_inv_w);
    r.lambda.x = interp_w * ((inv_w.x + dx * r.ddx.x) + dy * r.ddy.x);
    r.lambda.y = interp_w * ((0.0f + dx * r.ddx.y) + dy * r.ddy.y);
    r.lambda.z = interp_w * ((0.0f + dx * r.ddx.z) + dy * r.ddy.z);
    r.ddx = r.ddx * sx;
    r.ddy = r.ddy * sy;
    ddx_sum *= sx;
    ddy_sum *= sy;
    r.ddy = r.ddy * -1.0f;
    ddy_sum *= -1.0f;
    const float iwx = div_exact(1.0f, interp_inv_w + ddx_sum), iwy = div_exact(1.0f, interp_inv_w + ddy_sum);
    r.ddx = ((r.lambda * interp_inv_w + r.ddx) * iwx) - r.lambda;
    r.ddy = ((r.lambda * interp_inv_w + r.ddy) * iwy) - r.lambda;
    return r;
}

RDR_DEV vec3 interp3(const BaryDeriv& b, vec3 a0, vec3 a1, vec3 a2) {
    return mk3(dot3_fma(mk3(a0.x, a1.x, a2.x), b.lambda), dot3_fma(mk3(a0.y, a1.y, a2.y), b.lambda),
               dot3_fma(mk3(a0.z, a1.z, a2.z), b.lambda));
}

RDR_DEV vec3 vload3(const float* p) { return mk3(p[0], p[1], p[2]); }

// Decode a visibility key into (draw, meshlet, triangle).
RDR_DEV void decode_key(const SceneView& sv, unsigned long long key, uint32_t& draw_id, uint32_t& meshlet, uint32_t& tri) {
    const uint32_t id = (uint32_t)key, instance = id >> 7;
    tri = id & 127u;
    draw_id = sv.instance_draw[instance];
    const rdr_mesh_draw d = sv.draws[draw_id];
    const rdr_mesh_lod L = sv.meshes[d.mesh_index].lods[sv.draw_lod[draw_id]];
    meshlet = L.meshlet_offset + (instance - d.meshlet_visibility_offset);
}

// A covered-pixel list entry resolved to what the shading passes work on.
struct ListedPixel {
    uint32_t pi, draw_id, tri, lod;
    uint32_t i1, i2, i3; // the triangle's vertices (absolute)
    uint32_t model_index, material_index;
    float uvx, uvy;
};

RDR_DEV ListedPixel listed_pixel(const SceneView& sv, const FrameView& fv, uint32_t entry) {
    ListedPixel p;
    p.pi = fv.pixel_list[entry];
    const uint32_t id = fv.pixel_key[entry], instance = id >> 7;
    p.tri = id & 127u;
    p.draw_id = sv.instance_draw[instance];
    const DrawFrame df = sv.draw_frame[p.draw_id];
    const TriVertices tv = sv.tri_table[(size_t)(df.meshlet_base + instance) * kTriTableStride + p.tri];
    p.i1 = df.mesh_vertex_offset + tv.v[0];
    p.i2 = df.mesh_vertex_offset + tv.v[1];
    p.i3 = df.mesh_vertex_offset + tv.v[2];
    p.model_index = df.model_index;
    p.material_index = df.material_lod & 0xFFFFu;
    p.lod = df.material_lod >> 16;
    const uint32_t packed = p.pi, iy = packed >> 16, ix = packed & 0xFFFFu;
    p.pi = iy * fv.width + ix;
    // the full-screen triangle's interpolated uv at the pixel centre (fullscreen_triangle.vert.glsl:7-9)
    p.uvx = ((float)ix + 0.5f) * fv.inv_width;
    p.uvy = 1.0f - ((float)iy + 0.5f) * fv.inv_height;
    return p;
}

} // namespace
