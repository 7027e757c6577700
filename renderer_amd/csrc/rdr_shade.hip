// rdr_shade.hip — deferred glTF-PBR shading of the visibility buffer, sun shadow
// ray and sRGB store, one lane per pixel, 8x8 pixels per wave.
//
// What is computed follows the reference's shading pass (paths relative to the
// reference tree): assets/shaders/shading.frag.glsl:81-171 (analytic
// barycentrics and derivatives), :317-531 (material model, IBL, shadow ray),
// assets/shaders/fullscreen_triangle.vert.glsl:7-9 (pixel -> uv), and the sRGB
// encode an *_SRGB swapchain image performs (lava/src/swapchain.rs:100-115).
// The ray query against the Vulkan TLAS/BLAS
// (renderer/src/renderer/acceleration_sturct_utils.rs:23-49,253-279) becomes a
// software two-level BVH traversal.
#include "rdr_device.h"

namespace {

constexpr float kPi = 3.1415926535897932384626433832795f;
constexpr float kMinRoughness = 0.04f;

// ---- shadow ray -------------------------------------------------------------------
// Conservative ray / box slab test on one child of a wide node.  `w` holds one axis of
// the box as an fp16 pair (low half = box minimum, high half = box maximum).  Only "never
// miss a box the ray touches" matters for the result, so the planes are evaluated as
// t = plane * (1/d) + c with c = -o/d precomputed per ray: one mixed-precision FMA per
// plane, with the half -> float conversion folded into it.  The rounding of c, of the fast
// reciprocal and of the FMA is covered by a per-axis margin folded into the two constants
// (c_near = c - m, c_far = c + m).  The pairs are rotated by the ray's direction sign so
// that the low half is always the entry plane.  NaNs (0 * inf, inf - inf) fall out of
// min/max and never reject.
struct SlabRay {
    vec3 inv_d;  // 1 / d
    vec3 c_near; // -o / d - margin
    vec3 c_far;  // -o / d + margin
    uint32_t rx, ry, rz; // 16 where d < 0 on that axis, else 0: rotation that puts the entry plane low
};

RDR_DEV float half_of(uint32_t w, bool hi) {
    return (float)__builtin_bit_cast(_Float16, (uint16_t)(hi ? (w >> 16) : (w & 0xFFFFu)));
}

RDR_DEV bool ray_box(uint32_t wx, uint32_t wy, uint32_t wz, const SlabRay& r, float tmin, float tmax) {
    wx = __builtin_amdgcn_alignbit(wx, wx, r.rx);
    wy = __builtin_amdgcn_alignbit(wy, wy, r.ry);
    wz = __builtin_amdgcn_alignbit(wz, wz, r.rz);
    const float nx = fmaf(half_of(wx, false), r.inv_d.x, r.c_near.x), fx = fmaf(half_of(wx, true), r.inv_d.x, r.c_far.x);
    const float ny = fmaf(half_of(wy, false), r.inv_d.y, r.c_near.y), fy = fmaf(half_of(wy, true), r.inv_d.y, r.c_far.y);
    const float nz = fmaf(half_of(wz, false), r.inv_d.z, r.c_near.z), fz = fmaf(half_of(wz, true), r.inv_d.z, r.c_far.z);
    const float t_in = fmaxf(fmaxf(fmaxf(nx, ny), nz), tmin);
    const float t_out = fminf(fminf(fminf(fx, fy), fz), tmax);
    return t_in <= t_out;
}

// Moeller-Trumbore without the division, two-sided, open interval (tmin, tmax): with
// a = |det| and the numerators multiplied by sign(det), the ray hits when
// U >= 0, V >= 0, U + V <= a and tmin * a < T < tmax * a.  e1 = p1 - p0 and
// e2 = p2 - p0 are stored with the triangle.
RDR_DEV bool ray_triangle(vec3 o, vec3 d, vec3 p0, vec3 e1, vec3 e2, float tmin, float tmax) {
    const vec3 pv = cross3(d, e2);
    const float det = dot3(e1, pv);
    const vec3 tv = o - p0;
    const float un = dot3(tv, pv);
    const vec3 qv = cross3(tv, e1);
    const float vn = dot3(d, qv);
    const float tn = dot3(e2, qv);
    const uint32_t sign = __float_as_uint(det) & 0x80000000u;
    const float a = fabsf(det);
    const float U = __uint_as_float(__float_as_uint(un) ^ sign);
    const float V = __uint_as_float(__float_as_uint(vn) ^ sign);
    const float T = __uint_as_float(__float_as_uint(tn) ^ sign);
    return det != 0.0f && U >= 0.0f && V >= 0.0f && U + V <= a && T > tmin * a && T < tmax * a;
}

// Traversal stacks.  The fast one keeps kLdsStack entries per lane in LDS (one
// column per lane: conflict-free) and only records that it overflowed; a ray whose
// traversal overflowed without finding a hit is traced again with the deep,
// scratch-backed stack.  Reasonably balanced trees never reach the second path.
constexpr int kLdsStack = 24;
constexpr int kDeepStack = 160;

struct LdsStack {
    uint32_t* base; // &shared[thread], stride 256 entries, limit + 1 rows
    int sp, limit;
    bool overflow;
    // Branch-free conditional push: the store always happens, into row `limit` (a dump row
    // nobody reads) when there is nothing to push or no room.
    RDR_DEV void push_if(bool p, uint32_t v) {
        const bool go = p & (sp < limit);
        base[(go ? sp : limit) * 256] = v;
        overflow |= p & !go;
        sp += go ? 1 : 0;
    }
    RDR_DEV void push(uint32_t v) { push_if(true, v); }
    RDR_DEV uint32_t pop() { return base[--sp * 256]; }
};

struct DeepStack {
    uint32_t data[kDeepStack + 1];
    int sp;
    bool overflow;
    RDR_DEV void push_if(bool p, uint32_t v) {
        const bool go = p & (sp < kDeepStack);
        data[go ? sp : kDeepStack] = v;
        overflow |= p & !go;
        sp += go ? 1 : 0;
    }
    RDR_DEV void push(uint32_t v) { push_if(true, v); }
    RDR_DEV uint32_t pop() { return data[--sp]; }
};

struct Ray {
    vec3 o, d;
    SlabRay slab;
};

// Margin per axis: |c| * 2^-24 (rounding of c) plus (2^-23 + 2^-24) * |t| for the
// reciprocal and the FMA with |t| <= tmax where it can matter; (|c| + tmax) * 2^-21
// covers both four times over.
RDR_DEV Ray make_ray(vec3 o, vec3 d, float tmax) {
    Ray r;
    r.o = o; r.d = d;
    const vec3 inv = mk3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
    const vec3 c = mk3(-o.x * inv.x, -o.y * inv.y, -o.z * inv.z);
    const float k = 4.76837158203125e-7f; // 2^-21
    const vec3 m = mk3((fabsf(c.x) + tmax) * k, (fabsf(c.y) + tmax) * k, (fabsf(c.z) + tmax) * k);
    r.slab.inv_d = inv;
    r.slab.c_near = c - m;
    r.slab.c_far = c + m;
    r.slab.rx = d.x < 0.0f ? 16u : 0u;
    r.slab.ry = d.y < 0.0f ? 16u : 0u;
    r.slab.rz = d.z < 0.0f ? 16u : 0u;
    return r;
}

// The ray in an instance's object space: origin and (unnormalised) direction through
// the inverse model matrix, parameter t unchanged.
RDR_DEV Ray object_ray(const GpuInstance& inst, vec3 o, vec3 d, float tmax) {
    const float* iv = inst.inv;
    const vec3 oo = mk3(((iv[0] * o.x + iv[1] * o.y) + iv[2] * o.z) + iv[3], ((iv[4] * o.x + iv[5] * o.y) + iv[6] * o.z) + iv[7],
                        ((iv[8] * o.x + iv[9] * o.y) + iv[10] * o.z) + iv[11]);
    const vec3 od = mk3((iv[0] * d.x + iv[1] * d.y) + iv[2] * d.z, (iv[4] * d.x + iv[5] * d.y) + iv[6] * d.z,
                        (iv[8] * d.x + iv[9] * d.y) + iv[10] * d.z);
    return make_ray(oo, od, tmax);
}

// Is anything between the surface point and the sun?  One loop walks both levels: `cur` is
// the child code being processed, the stack holds the ones still to do, and `floor` is the
// stack height at which the current instance's BLAS is exhausted (-1 while in the top
// level).  The result is a plain "any triangle hit", so the visiting order is free: the
// pixel's own instance, by far the likeliest occluder, goes first (with the top level's
// root waiting at the bottom of the stack) and is skipped when the top level meets it
// again.  Every iteration handles one node and, if that left nothing to continue with,
// pops the next; state changes are in place so that the loop has one back edge.
template <typename Stack>
RDR_DEV bool shadow_any_hit(const SceneView& sv, uint32_t own_instance, vec3 o, vec3 d, float tmin, float tmax, Stack& st) {
    st.sp = 0;
    st.overflow = false;
    st.push(0u); // node 0 of a tree holds the children of its root
    int floor = 1;
    Ray ray = object_ray(sv.instances[own_instance], o, d, tmax);
    uint32_t node_base = sv.instances[own_instance].blas_root, tri_base = sv.instances[own_instance].tri_base;
    uint32_t cur = 0u;
    bool hit = false;
    for (;;) {
        if (!(cur & kBvhLeaf)) {
            // one wide node: all its loads are in flight together
            const uint4* q = reinterpret_cast<const uint4*>(sv.bvh + (node_base + cur));
            const uint4 codes = q[0], bx = q[1], by = q[2], bz = q[3];
            const uint32_t code[4] = {codes.x, codes.y, codes.z, codes.w};
            const uint32_t wx[4] = {bx.x, bx.y, bx.z, bx.w}, wy[4] = {by.x, by.y, by.z, by.w}, wz[4] = {bz.x, bz.y, bz.z, bz.w};
            cur = kBvhNone;
#pragma unroll
            for (int i = 0; i < kBvhWidth; ++i) {
                const bool h = (code[i] != kBvhNone) & ray_box(wx[i], wy[i], wz[i], ray.slab, tmin, tmax);
                st.push_if(h & (cur != kBvhNone), cur);
                cur = h ? code[i] : cur;
            }
        } else if (cur != kBvhNone) {
            const uint32_t count = (cur >> 24) & 127u, first = cur & 0xFFFFFFu;
            cur = kBvhNone;
            if (floor >= 0) {
                const float* tp = sv.blas_tris + (size_t)(tri_base + first) * 9;
#pragma clang loop vectorize(disable)
                for (uint32_t i = 0; i < count; ++i, tp += 9)
                    hit |= ray_triangle(ray.o, ray.d, mk3(tp[0], tp[1], tp[2]), mk3(tp[3], tp[4], tp[5]), mk3(tp[6], tp[7], tp[8]), tmin, tmax);
            } else {
                // top-level leaf: take its first instance now, leave the others on the stack
                const uint32_t instance = sv.tlas_items[first];
                if (count > 1u) st.push(kBvhLeaf | ((count - 1u) << 24) | (first + 1u));
                if (instance != own_instance) {
                    const GpuInstance& inst = sv.instances[instance];
                    ray = object_ray(inst, o, d, tmax);
                    node_base = inst.blas_root;
                    tri_base = inst.tri_base;
                    floor = st.sp;
                    cur = 0u;
                }
            }
        }
        if (hit) break;
        if (cur == kBvhNone) {
            if (st.sp == floor) { // this instance is done: back to the world ray
                floor = -1;
                ray = make_ray(o, d, tmax);
                node_base = sv.tlas_base;
            }
            if (st.sp == 0) break;
            cur = st.pop();
        }
    }
    return hit;
}

// ---- barycentrics -----------------------------------------------------------------
struct BaryDeriv {
    vec3 lambda, ddx, ddy;
};

RDR_DEV BaryDeriv calc_full_bary(vec4 pt0, vec4 pt1, vec4 pt2, float ndcx, float ndcy, float winw, float winh) {
    BaryDeriv r;
    const vec3 inv_w = mk3(1.0f / pt0.w, 1.0f / pt1.w, 1.0f / pt2.w);
    const float n0x = pt0.x * inv_w.x, n0y = pt0.y * inv_w.x;
    const float n1x = pt1.x * inv_w.y, n1y = pt1.y * inv_w.y;
    const float n2x = pt2.x * inv_w.z, n2y = pt2.y * inv_w.z;
    const float ax = n2x - n1x, ay = n2y - n1y, bx = n0x - n1x, by = n0y - n1y;
    const float inv_det = 1.0f / (ax * by - bx * ay);
    r.ddx = (mk3(n1y - n2y, n2y - n0y, n0y - n1y) * inv_det) * inv_w;
    r.ddy = (mk3(n2x - n1x, n0x - n2x, n1x - n0x) * inv_det) * inv_w;
    float ddx_sum = (r.ddx.x + r.ddx.y) + r.ddx.z;
    float ddy_sum = (r.ddy.x + r.ddy.y) + r.ddy.z;
    const float dx = ndcx - n0x, dy = ndcy - n0y;
    const float interp_inv_w = (inv_w.x + dx * ddx_sum) + dy * ddy_sum;
    const float interp_w = 1.0f / interp_inv_w;
    r.lambda.x = interp_w * ((inv_w.x + dx * r.ddx.x) + dy * r.ddy.x);
    r.lambda.y = interp_w * ((0.0f + dx * r.ddx.y) + dy * r.ddy.y);
    r.lambda.z = interp_w * ((0.0f + dx * r.ddx.z) + dy * r.ddy.z);
    const float sx = 2.0f / winw, sy = 2.0f / winh;
    r.ddx = r.ddx * sx;
    r.ddy = r.ddy * sy;
    ddx_sum *= sx;
    ddy_sum *= sy;
    r.ddy = r.ddy * -1.0f;
    ddy_sum *= -1.0f;
    const float iwx = 1.0f / (interp_inv_w + ddx_sum), iwy = 1.0f / (interp_inv_w + ddy_sum);
    r.ddx = ((r.lambda * interp_inv_w + r.ddx) * iwx) - r.lambda;
    r.ddy = ((r.lambda * interp_inv_w + r.ddy) * iwy) - r.lambda;
    return r;
}

RDR_DEV vec3 interp3(const BaryDeriv& b, vec3 a0, vec3 a1, vec3 a2) {
    return mk3(dot3(mk3(a0.x, a1.x, a2.x), b.lambda), dot3(mk3(a0.y, a1.y, a2.y), b.lambda),
               dot3(mk3(a0.z, a1.z, a2.z), b.lambda));
}

struct InterpUv {
    float u, v, dudx, dvdx, dudy, dvdy;
};

RDR_DEV InterpUv uv_transform(InterpUv in, float sx, float sy, float ox, float oy) {
    InterpUv r;
    r.dudx = in.dudx * sx; r.dvdx = in.dvdx * sy;
    r.dudy = in.dudy * sx; r.dvdy = in.dvdy * sy;
    r.u = in.u * sx + ox; r.v = in.v * sy + oy;
    return r;
}

RDR_DEV vec4 sample_or_default(const SceneView& sv, uint32_t tex, InterpUv uv, vec4 def) {
    if (tex == RDR_NO_TEXTURE) return def;
    if (tex >= sv.image_count || sv.images[tex].mip_count == 0u) return mk4(0.0f, 0.0f, 0.0f, 0.0f);
    return image_sample_grad(sv.images[tex], sv.texels, sv.srgb_lut, uv.u, uv.v, uv.dudx, uv.dvdx, uv.dudy, uv.dvdy);
}

// ---- BRDF terms -------------------------------------------------------------------
RDR_DEV float pow5(float x) {
    const float x2 = x * x, x4 = x2 * x2;
    return x4 * x;
}
RDR_DEV vec3 splat(float a) { return mk3(a, a, a); }
RDR_DEV vec3 mix3(vec3 a, vec3 b, float t) {
    const float s = 1.0f - t;
    return mk3(a.x * s + b.x * t, a.y * s + b.y * t, a.z * s + b.z * t);
}
RDR_DEV float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
RDR_DEV vec3 clamp01(vec3 a) { return mk3(clampf(a.x, 0.0f, 1.0f), clampf(a.y, 0.0f, 1.0f), clampf(a.z, 0.0f, 1.0f)); }

RDR_DEV float geometric_occlusion(float ndl, float ndv, float alpha) {
    const float r2 = alpha * alpha;
    const float al = 2.0f * ndl / (ndl + sqrtf(r2 + (1.0f - r2) * (ndl * ndl)));
    const float av = 2.0f * ndv / (ndv + sqrtf(r2 + (1.0f - r2) * (ndv * ndv)));
    return al * av;
}
RDR_DEV float microfacet_distribution(float alpha, float ndh) {
    const float r2 = alpha * alpha;
    const float f = (ndh * r2 - ndh) * ndh + 1.0f;
    return r2 / (kPi * f * f);
}
RDR_DEV vec3 f_schlick(vec3 f0, vec3 f90, float vdh) {
    const float p = pow5(clampf(1.0f - vdh, 0.0f, 1.0f));
    return f0 + (f90 - f0) * p;
}
RDR_DEV vec3 diffuse_burley(vec3 diffuse, float ldh, float ndl, float ndv, float alpha) {
    const float f90 = 2.0f * ldh * ldh * alpha - 0.5f;
    const float a = 1.0f + f90 * pow5(1.0f - ndl);
    const float b = 1.0f + f90 * pow5(1.0f - ndv);
    return ((diffuse * (1.0f / kPi)) * a) * b;
}
RDR_DEV vec3 reflect_neg(vec3 v, vec3 n) {
    const vec3 i = v * -1.0f;
    const float d = dot3(n, i);
    return i - n * (2.0f * d);
}

RDR_DEV vec4 brdf_lut_tap(const SceneView& sv, float x, float y) {
    return image_sample_lod(sv.brdf_lut, sv.texels, sv.srgb_lut, clampf(x, 0.0f, 1.0f), clampf(y, 0.0f, 1.0f), 0.0f);
}

RDR_DEV vec3 ibl_ggx(const SceneView& sv, vec3 n, vec3 v, float ndv, float rough, vec3 refl0, float spec_w) {
    const vec3 r = normalize3(reflect_neg(v, n));
    const float mips = (float)sv.ggx.mip_count;
    const float lod = rough * (mips - 1.0f);
    const vec4 brdf = brdf_lut_tap(sv, ndv, rough);
    const vec4 sl = cube_sample_lod(sv.ggx, sv.texels, r, lod);
    const vec3 fr = mk3(fmaxf(1.0f - rough, refl0.x), fmaxf(1.0f - rough, refl0.y), fmaxf(1.0f - rough, refl0.z)) - refl0;
    const vec3 ks = refl0 + fr * pow5(1.0f - ndv);
    const vec3 fss = ks * brdf.x + splat(brdf.y);
    return (mk3(sl.x, sl.y, sl.z) * spec_w) * fss;
}

RDR_DEV vec3 ibl_lambertian(const SceneView& sv, float ndv, vec3 n, float rough, vec3 diffuse, vec3 f0, float spec_w) {
    const vec4 fab = brdf_lut_tap(sv, ndv, rough);
    const vec4 ir = cube_sample_lod(sv.lambertian, sv.texels, n, 0.0f);
    const vec3 irr = mk3(ir.x, ir.y, ir.z);
    const vec3 fr = mk3(fmaxf(1.0f - rough, f0.x), fmaxf(1.0f - rough, f0.y), fmaxf(1.0f - rough, f0.z)) - f0;
    const vec3 ks = f0 + fr * pow5(1.0f - ndv);
    const vec3 fss = (ks * spec_w) * fab.x + splat(fab.y);
    const float ems = 1.0f - (fab.x + fab.y);
    const vec3 favg = (f0 + (splat(1.0f) - f0) * (1.0f / 21.0f)) * spec_w;
    vec3 fms;
    fms.x = ems * fss.x * favg.x / (1.0f - favg.x * ems);
    fms.y = ems * fss.y * favg.y / (1.0f - favg.y * ems);
    fms.z = ems * fss.z * favg.z / (1.0f - favg.z * ems);
    const vec3 kd = diffuse * ((splat(1.0f) - fss) + fms);
    return (fms + kd) * irr;
}

RDR_DEV vec3 ibl_charlie(const SceneView& sv, vec3 n, vec3 v, float ndv, float sr, vec3 sc) {
    const float mips = (float)sv.charlie.mip_count;
    const float lod = sr * (mips - 1.0f);
    const vec3 r = normalize3(reflect_neg(v, n));
    const float brdf = brdf_lut_tap(sv, ndv, sr).z;
    const vec4 smp = cube_sample_lod(sv.charlie, sv.texels, r, lod);
    return (mk3(smp.x, smp.y, smp.z) * sc) * brdf;
}

RDR_DEV float d_charlie(float sr, float ndh) {
    sr = fmaxf(sr, 0.000001f);
    const float ag = sr * sr, inv_r = 1.0f / ag;
    const float sin2h = 1.0f - ndh * ndh;
    return (2.0f + inv_r) * powf(sin2h, inv_r * 0.5f) / (2.0f * kPi);
}
RDR_DEV float lambda_sheen_helper(float x, float ag) {
    const float om = (1.0f - ag) * (1.0f - ag);
    const float a = mixf(21.5473f, 25.3245f, om), b = mixf(3.82987f, 3.32435f, om);
    const float c = mixf(0.19823f, 0.16801f, om), d = mixf(-1.97760f, -1.27393f, om);
    const float e = mixf(-4.32054f, -4.85967f, om);
    return a / (1.0f + b * powf(x, c)) + d * x + e;
}
RDR_DEV float lambda_sheen(float ct, float ag) {
    if (fabsf(ct) < 0.5f) return expf(lambda_sheen_helper(ct, ag));
    return expf(2.0f * lambda_sheen_helper(0.5f, ag) - lambda_sheen_helper(1.0f - ct, ag));
}
RDR_DEV float v_sheen(float ndl, float ndv, float sr) {
    sr = fmaxf(sr, 0.000001f);
    const float ag = sr * sr;
    // the reference evaluates lambda_sheen(n_dot_l) twice (shading.frag.glsl:291)
    return clampf(1.0f / ((1.0f + lambda_sheen(ndl, ag) + lambda_sheen(ndl, ag)) * (4.0f * ndv * ndl)), 0.0f, 1.0f);
}
RDR_DEV vec3 clearcoat_punctual(vec3 cn, vec3 v, vec3 l, vec3 h, float vdh, vec3 f0, vec3 f90, float cr) {
    const float ndl = clampf(dot3(cn, l), 0.001f, 1.0f);
    const float ndv = clampf(dot3(cn, v), 0.001f, 1.0f);
    const float ndh = clampf(dot3(cn, h), 0.001f, 1.0f);
    const vec3 f = f_schlick(f0, f90, vdh);
    const float g = geometric_occlusion(ndl, ndv, cr * cr);
    const float d = microfacet_distribution(cr * cr, ndh);
    const vec3 spec = ((f * g) * d) * (1.0f / (4.0f * ndl * ndv));
    return spec * ndl;
}

RDR_DEV uint32_t srgb_encode(float c) {
    c = clampf(c, 0.0f, 1.0f);
    const float e = c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
    return (uint32_t)floorf(e * 255.0f + 0.5f);
}

RDR_DEV vec3 vload3(const float* p) { return mk3(p[0], p[1], p[2]); }

// World-space position of the surface point seen through a pixel (the shadow ray's origin).
RDR_DEV vec3 pixel_world_position(const SceneView& sv, const rdr_ubo& u, uint32_t draw_id, uint32_t meshlet_index,
                                  uint32_t tri, float uvx, float uvy) {
    const rdr_mesh_draw d = sv.draws[draw_id];
    const uint32_t mesh_vertex_offset = sv.meshes[d.mesh_index].vertex_offset;
    float model[16];
    load_m4(sv.transforms + (size_t)d.model_index * 16, model);
    const rdr_meshlet* ml = &sv.meshlets[meshlet_index];
    const uint8_t* tp = sv.meshlet_triangles + ml->triangle_offset + tri * 3;
    const uint32_t* mv = sv.meshlet_vertices + ml->vertex_offset;
    const uint32_t i1 = mesh_vertex_offset + mv[tp[0]], i2 = mesh_vertex_offset + mv[tp[1]], i3 = mesh_vertex_offset + mv[tp[2]];
    const vec4 w1 = mul_m4(model, mk4(sv.pos_x[i1], sv.pos_y[i1], sv.pos_z[i1], 1.0f));
    const vec4 w2 = mul_m4(model, mk4(sv.pos_x[i2], sv.pos_y[i2], sv.pos_z[i2], 1.0f));
    const vec4 w3 = mul_m4(model, mk4(sv.pos_x[i3], sv.pos_y[i3], sv.pos_z[i3], 1.0f));
    const vec4 c1 = mul_m4(u.view_proj, w1), c2 = mul_m4(u.view_proj, w2), c3 = mul_m4(u.view_proj, w3);
    const BaryDeriv bary = calc_full_bary(c1, c2, c3, uvx * 2.0f - 1.0f, uvy * 2.0f - 1.0f, u.screen_width, u.screen_height);
    return interp3(bary, mk3(w1.x, w1.y, w1.z), mk3(w2.x, w2.y, w2.z), mk3(w3.x, w3.y, w3.z));
}

RDR_DEV vec3 shade_pixel(const SceneView& sv, const rdr_ubo& u, uint32_t draw_id, uint32_t meshlet_index, uint32_t tri,
                         float uvx, float uvy, bool shadow_hit) {
    const rdr_mesh_draw d = sv.draws[draw_id];
    const uint32_t mesh_vertex_offset = sv.meshes[d.mesh_index].vertex_offset;
    float model[16];
    load_m4(sv.transforms + (size_t)d.model_index * 16, model);
    rdr_material mat;
    if (d.material_index < sv.material_count) {
        mat = sv.materials[d.material_index];
    } else { // robust buffer access: zeros
        uint32_t* z = reinterpret_cast<uint32_t*>(&mat);
        for (uint32_t i = 0; i < sizeof(mat) / 4; ++i) z[i] = 0u;
    }
    const rdr_meshlet* ml = &sv.meshlets[meshlet_index];
    const uint8_t* tp = sv.meshlet_triangles + ml->triangle_offset + tri * 3;
    const uint32_t* mv = sv.meshlet_vertices + ml->vertex_offset;
    const rdr_vertex* v1 = &sv.vertices[mesh_vertex_offset + mv[tp[0]]];
    const rdr_vertex* v2 = &sv.vertices[mesh_vertex_offset + mv[tp[1]]];
    const rdr_vertex* v3 = &sv.vertices[mesh_vertex_offset + mv[tp[2]]];

    const vec4 w1 = mul_m4(model, mk4(v1->p[0], v1->p[1], v1->p[2], 1.0f));
    const vec4 w2 = mul_m4(model, mk4(v2->p[0], v2->p[1], v2->p[2], 1.0f));
    const vec4 w3 = mul_m4(model, mk4(v3->p[0], v3->p[1], v3->p[2], 1.0f));
    const vec4 c1 = mul_m4(u.view_proj, w1), c2 = mul_m4(u.view_proj, w2), c3 = mul_m4(u.view_proj, w3);
    const BaryDeriv bary = calc_full_bary(c1, c2, c3, uvx * 2.0f - 1.0f, uvy * 2.0f - 1.0f, u.screen_width, u.screen_height);

    const vec3 w_pos = interp3(bary, mk3(w1.x, w1.y, w1.z), mk3(w2.x, w2.y, w2.z), mk3(w3.x, w3.y, w3.z));
    const vec3 norm = interp3(bary, mul_m3(model, vload3(v1->n)), mul_m3(model, vload3(v2->n)), mul_m3(model, vload3(v3->n)));
    const vec3 tangent = interp3(bary, mul_m3(model, vload3(v1->t)), mul_m3(model, vload3(v2->t)), mul_m3(model, vload3(v3->t)));
    const vec3 us = mk3(v1->uv[0], v2->uv[0], v3->uv[0]), vs = mk3(v1->uv[1], v2->uv[1], v3->uv[1]);
    // uv1 is interpolated from .xy as well (shading.frag.glsl:373-374): the uv index never matters
    const InterpUv uv0 = {dot3(us, bary.lambda), dot3(vs, bary.lambda), dot3(us, bary.ddx), dot3(vs, bary.ddx),
                          dot3(us, bary.ddy), dot3(vs, bary.ddy)};

    const vec4 one = mk4(1.0f, 1.0f, 1.0f, 1.0f), nmap_def = mk4(0.5f, 0.5f, 1.0f, 0.0f);
    vec4 albedo = sample_or_default(sv, mat.albedo_texture,
                                    uv_transform(uv0, mat.albedo_texture_transform[0], mat.albedo_texture_transform[1],
                                                 mat.albedo_texture_transform[2], mat.albedo_texture_transform[3]), one);
    albedo.x *= mat.base_color_factor[0]; albedo.y *= mat.base_color_factor[1];
    albedo.z *= mat.base_color_factor[2]; albedo.w *= mat.base_color_factor[3];
    const vec4 nm4 = sample_or_default(sv, mat.normal_texture,
                                       uv_transform(uv0, mat.normal_texture_transform[0], mat.normal_texture_transform[1],
                                                    mat.normal_texture_transform[2], mat.normal_texture_transform[3]), nmap_def);
    const vec3 nmap = mk3(nm4.x * 2.0f - 1.0f, nm4.y * 2.0f - 1.0f, nm4.z * 2.0f - 1.0f);
    const vec4 mr = sample_or_default(sv, mat.metallic_roughness_texture,
                                      uv_transform(uv0, mat.metallic_roughness_texture_transform[0], mat.metallic_roughness_texture_transform[1],
                                                   mat.metallic_roughness_texture_transform[2], mat.metallic_roughness_texture_transform[3]), one);
    const vec4 occ = sample_or_default(sv, mat.occlusion_texture,
                                       uv_transform(uv0, mat.occlusion_texture_transform[0], mat.occlusion_texture_transform[1],
                                                    mat.occlusion_texture_transform[2], mat.occlusion_texture_transform[3]), one);
    const InterpUv uv_id = uv_transform(uv0, 1.0f, 1.0f, 0.0f, 0.0f);
    const vec4 em4 = sample_or_default(sv, mat.emissive_texture, uv_id, one);
    const vec3 emissive = mk3(mat.emissive_factor_alpha_cutoff[0] * em4.x, mat.emissive_factor_alpha_cutoff[1] * em4.y,
                              mat.emissive_factor_alpha_cutoff[2] * em4.z);
    const vec4 sc4 = sample_or_default(sv, mat.specular_color_texture, uv_id, one);
    const vec4 sp4 = sample_or_default(sv, mat.specular_texture, uv_id, one);
    const vec3 spec_color_factor = mk3(mat.specular_factor[0] * sc4.x, mat.specular_factor[1] * sc4.y, mat.specular_factor[2] * sc4.z);
    const float spec_factor_w = mat.specular_factor[3] * sp4.x;

    const vec3 nnormal = normalize3(norm);
    const vec3 ntangent = normalize3(tangent);
    const vec3 bitangent = cross3(nnormal, ntangent) * ((v1->t[3] * v2->t[3]) * v3->t[3]);
    const vec3 nrm = normalize3((ntangent * nmap.x + bitangent * nmap.y) + nnormal * nmap.z);

    const vec3 L = normalize3(mk3(u.sun_direction[0], u.sun_direction[1], u.sun_direction[2]));

    const float metallic = clampf(mat.metallic_roughness_normal_occlusion[0] * mr.z, 0.0f, 1.0f);
    float prough = mr.y * mat.metallic_roughness_normal_occlusion[1];
    if (mat.material_type & RDR_MAT_SPEC_GLOSS) prough = 1.0f - mr.w * mat.specular_glossiness_factor[3];
    else prough = clampf(prough, kMinRoughness, 1.0f);
    const float alpha = prough * prough;

    const vec3 base = mk3(albedo.x, albedo.y, albedo.z);
    const float fi = (mat.ior - 1.0f) / (mat.ior + 1.0f);
    vec3 f0 = splat(fi * fi);
    vec3 diffuse = mix3(base, splat(0.0f), metallic);
    vec3 specular = mix3(f0, base, metallic);
    float spec_w = 1.0f;
    if (mat.material_type & RDR_MAT_SPEC) {
        const vec3 dsf0 = mk3(fminf(f0.x * spec_color_factor.x, 1.0f), fminf(f0.y * spec_color_factor.y, 1.0f),
                              fminf(f0.z * spec_color_factor.z, 1.0f));
        f0 = mix3(dsf0, base, metallic);
        spec_w = spec_factor_w;
    }
    if (mat.material_type & RDR_MAT_SPEC_GLOSS) {
        f0 = mk3(mat.specular_glossiness_factor[0] * mr.x, mat.specular_glossiness_factor[1] * mr.y,
                 mat.specular_glossiness_factor[2] * mr.z);
        diffuse = base * (1.0f - max3f(f0.x, f0.y, f0.z));
        specular = f0;
    }
    const float reflectance = max3f(specular.x, specular.y, specular.z);
    const float refl90 = clampf(reflectance * 25.0f, 0.0f, 1.0f);
    const vec3 r0 = specular, r90 = splat(refl90);

    const vec3 cam = mk3(u.camera_position[0], u.camera_position[1], u.camera_position[2]);
    const vec3 v = normalize3(w_pos - cam) * -1.0f;
    const vec3 h = normalize3(L + v);
    const float ndv = clampf(fabsf(dot3(nrm, v)), 0.001f, 1.0f);
    const float ndl = clampf(dot3(nrm, L), 0.001f, 1.0f);
    const float ndh = clampf(dot3(nrm, h), 0.0f, 1.0f);
    const float ldh = clampf(dot3(L, h), 0.0f, 1.0f);
    const float vdh = clampf(dot3(v, h), 0.0f, 1.0f);

    const float light = 5.0f;
    const vec3 f = f_schlick(r0, r90, vdh);
    const float g = geometric_occlusion(ndl, ndv, alpha);
    const float dd = microfacet_distribution(alpha, ndh);
    const vec3 diffuse_c = (splat(1.0f) - f) * diffuse_burley(diffuse, ldh, ndl, ndv, alpha);
    const vec3 spec_c = ((f * g) * dd) * (1.0f / (4.0f * ndl * ndv));
    vec3 color = (diffuse_c + spec_c) * (ndl * light);

    const vec3 ibl_s = ibl_ggx(sv, nrm, v, ndv, prough, r0, spec_w);
    const vec3 ibl_d = ibl_lambertian(sv, ndv, nrm, prough, diffuse, r0, spec_w);
    color = clamp01(((ibl_s + ibl_d) + color) + emissive) * occ.x;

    if (mat.material_type & RDR_MAT_CLEARCOAT) {
        const vec4 cn4 = sample_or_default(sv, mat.clearcoat_texture, uv_id, nmap_def);
        const vec3 cm = mk3(cn4.x * 2.0f - 1.0f, cn4.y * 2.0f - 1.0f, cn4.z * 2.0f - 1.0f);
        const vec3 cnrm = normalize3((ntangent * cm.x + bitangent * cm.y) + nnormal * cm.z);
        const float cndv = clampf(fabsf(dot3(cnrm, v)), 0.001f, 1.0f);
        const float cr = mat.clearcoat_factor[1];
        const vec3 cf0 = splat(fi * fi);
        const vec3 contrib = ibl_ggx(sv, cnrm, v, cndv, cr, cf0, 1.0f);
        const vec3 fres = f_schlick(r0, r90, cndv);
        color = color * (splat(1.0f) - fres * mat.clearcoat_factor[0]) + contrib;
        color = color + clearcoat_punctual(cnrm, v, L, h, vdh, splat(kMinRoughness), splat(1.0f), cr) * light;
    }
    if (mat.material_type & RDR_MAT_SHEEN) {
        const vec3 sc = mk3(mat.sheen_color_roughness_factor[0], mat.sheen_color_roughness_factor[1], mat.sheen_color_roughness_factor[2]);
        const float sr = mat.sheen_color_roughness_factor[3];
        color = color + ibl_charlie(sv, nrm, v, ndv, sr, sc);
        const float sd = d_charlie(sr, ndh), svis = v_sheen(ndl, ndv, sr);
        color = color + ((sc * sd) * svis) * (ndl * light);
    }
    return color * (shadow_hit ? 0.35f : 1.0f);
}

// Decode a visibility key into (draw, meshlet, triangle).
RDR_DEV void decode_key(const SceneView& sv, unsigned long long key, uint32_t& draw_id, uint32_t& meshlet, uint32_t& tri) {
    const uint32_t id = (uint32_t)key, instance = id >> 7;
    tri = id & 127u;
    draw_id = sv.instance_draw[instance];
    const rdr_mesh_draw d = sv.draws[draw_id];
    const rdr_mesh_lod L = sv.meshes[d.mesh_index].lods[sv.draw_lod[draw_id]];
    meshlet = L.meshlet_offset + (instance - d.meshlet_visibility_offset);
}

// Pass 0: classification.  A workgroup owns a 64x64-pixel region as sixteen 16x16 blocks, each
// wave an 8x8 sub-block.  Pixels with no geometry get the shading pass's clear colour here;
// covered pixels are appended to a compact pixel list, which the two passes below run over
// with full waves.  One atomic per workgroup reserves the region's span of the list (same-
// address atomics serialise, so they are kept to a few thousand per frame); inside the span
// the order is block-, wave-, lane-major, so 64 consecutive entries stay spatially close.
constexpr uint32_t kClassifyRegion = 64;

__global__ __launch_bounds__(256) void k_classify(FrameView fv) {
    __shared__ uint32_t wave_count[16 * 4];
    __shared__ uint32_t wg_base;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t x0 = blockIdx.x * kClassifyRegion + (wave & 1u) * 8 + (lane & 7u);
    const uint32_t y0 = fv.row_lo + blockIdx.y * kClassifyRegion + (wave >> 1) * 8 + (lane >> 3);
    uint32_t covered_bits = 0;
    uint32_t* color = reinterpret_cast<uint32_t*>(fv.color);
#pragma unroll
    for (uint32_t b = 0; b < 16; ++b) {
        const uint32_t ix = x0 + (b & 3u) * 16, iy = y0 + (b >> 2) * 16;
        bool covered = false;
        if (ix < fv.width && iy < fv.row_hi) {
            const uint32_t pi = iy * fv.width + ix;
            covered = fv.vis[pi] != RDR_VIS_CLEAR;
            if (!covered) color[pi] = 0u; // the shading pass clears its target to (0,0,0,0)
        }
        const unsigned long long m = __ballot(covered);
        if (lane == 0) wave_count[b * 4 + wave] = (uint32_t)__popcll(m);
        covered_bits |= covered ? (1u << b) : 0u;
    }
    __syncthreads();
    if (threadIdx.x < 64) { // exclusive prefix sum of the 64 (block, wave) counts
        const uint32_t c = wave_count[threadIdx.x];
        const uint32_t inc = wave_scan_inclusive(c, lane);
        wave_count[threadIdx.x] = inc - c;
        if (threadIdx.x == 63) wg_base = inc ? atomicAdd(&fv.counters->pixel_count, inc) : 0u;
    }
    __syncthreads();
    const uint32_t base = wg_base;
#pragma unroll
    for (uint32_t b = 0; b < 16; ++b) {
        const bool covered = (covered_bits >> b) & 1u;
        const unsigned long long m = __ballot(covered);
        if (covered) {
            const uint32_t ix = x0 + (b & 3u) * 16, iy = y0 + (b >> 2) * 16;
            fv.pixel_list[base + wave_count[b * 4 + wave] + ballot_rank(m)] = iy * fv.width + ix;
        }
    }
}

struct ListedPixel {
    uint32_t pi, draw_id, meshlet, tri;
    float uvx, uvy;
};

RDR_DEV ListedPixel listed_pixel(const SceneView& sv, const FrameView& fv, uint32_t entry) {
    ListedPixel p;
    p.pi = fv.pixel_list[entry];
    decode_key(sv, fv.vis[p.pi], p.draw_id, p.meshlet, p.tri);
    const uint32_t iy = p.pi / fv.width, ix = p.pi - iy * fv.width;
    p.uvx = ((float)ix + 0.5f) / (float)fv.width;
    p.uvy = 1.0f - ((float)iy + 0.5f) / (float)fv.height;
    return p;
}

// Pass 1: the sun shadow ray of every covered pixel (shading.frag.glsl:421-425), kept
// apart from the material evaluation so that the pointer-chasing traversal runs at
// high occupancy with its stack in LDS.  The result is stored per list entry.
__global__ __launch_bounds__(256) void k_shadow(SceneView sv, FrameView fv) {
    __shared__ uint32_t stack_mem[(kLdsStack + 1) * 256];
    const uint32_t entry = blockIdx.x * 256u + threadIdx.x;
    if (entry >= fv.counters->pixel_count) return;
    const ListedPixel p = listed_pixel(sv, fv, entry);
    const vec3 w_pos = pixel_world_position(sv, fv.ubo, p.draw_id, p.meshlet, p.tri, p.uvx, p.uvy);
    const rdr_ubo& u = fv.ubo;
    const vec3 L = normalize3(mk3(u.sun_direction[0], u.sun_direction[1], u.sun_direction[2]));
    LdsStack st;
    st.base = stack_mem + threadIdx.x;
    st.limit = (int)fv.lds_stack_limit;
    const bool h = shadow_any_hit(sv, p.draw_id, w_pos, L, 1e-2f, 1000.0f, st);
    uint8_t hit = h ? 1 : 0;
    if (!h && st.overflow) { // a miss that dropped stack entries proves nothing: leave it to k_shadow_deep
        hit = 2;
        atomicAdd(&fv.counters->shadow_overflow, 1u);
    }
    fv.shadow[entry] = hit;
}

// Pass 1b: pixels whose traversal overflowed the LDS stack, traced again with the deep
// scratch-backed stack.  Exits at once when there are none (the normal case).
__global__ __launch_bounds__(256) void k_shadow_deep(SceneView sv, FrameView fv) {
    if (fv.counters->shadow_overflow == 0u) return;
    const uint32_t count = fv.counters->pixel_count;
    const rdr_ubo& u = fv.ubo;
    for (uint32_t entry = blockIdx.x * blockDim.x + threadIdx.x; entry < count; entry += gridDim.x * blockDim.x) {
        if (fv.shadow[entry] != 2) continue;
        const ListedPixel p = listed_pixel(sv, fv, entry);
        const vec3 w_pos = pixel_world_position(sv, u, p.draw_id, p.meshlet, p.tri, p.uvx, p.uvy);
        const vec3 L = normalize3(mk3(u.sun_direction[0], u.sun_direction[1], u.sun_direction[2]));
        DeepStack st;
        fv.shadow[entry] = shadow_any_hit(sv, p.draw_id, w_pos, L, 1e-2f, 1000.0f, st) ? 1 : 0;
    }
}

// Pass 2: material evaluation, lighting, sRGB store of every covered pixel.
__global__ __launch_bounds__(256) void k_shade(SceneView sv, FrameView fv) {
    const uint32_t entry = blockIdx.x * 256u + threadIdx.x;
    if (entry >= fv.counters->pixel_count) return;
    const ListedPixel p = listed_pixel(sv, fv, entry);
    const vec3 c = shade_pixel(sv, fv.ubo, p.draw_id, p.meshlet, p.tri, p.uvx, p.uvy, fv.shadow[entry] != 0);
    reinterpret_cast<uint32_t*>(fv.color)[p.pi] =
        srgb_encode(c.x) | (srgb_encode(c.y) << 8) | (srgb_encode(c.z) << 16) | (255u << 24);
}

} // namespace

int rdr_launch_shade(const SceneView& sv, const FrameView& fv, hipStream_t stream, hipEvent_t* ev) {
    if (fv.row_hi <= fv.row_lo) return 0;
    const dim3 tiles((fv.width + kClassifyRegion - 1u) / kClassifyRegion, (fv.row_hi - fv.row_lo + kClassifyRegion - 1u) / kClassifyRegion);
    // the list passes are launched for the worst case (every pixel covered); workgroups
    // beyond the list's length exit on their first instruction
    const uint32_t list_blocks = (uint32_t)(((size_t)(fv.row_hi - fv.row_lo) * fv.width + 255u) / 256u);
    if (ev) RDR_HIP(hipEventRecord(ev[6], stream));
    RDR_HIP(hipMemsetAsync(&fv.counters->pixel_count, 0, 2 * sizeof(uint32_t), stream)); // pixel_count, shadow_overflow
    hipLaunchKernelGGL(k_classify, tiles, dim3(256), 0, stream, fv);
    if (ev) RDR_HIP(hipEventRecord(ev[7], stream));
    hipLaunchKernelGGL(k_shadow, dim3(list_blocks), dim3(256), 0, stream, sv, fv);
    hipLaunchKernelGGL(k_shadow_deep, dim3(512), dim3(256), 0, stream, sv, fv);
    if (ev) RDR_HIP(hipEventRecord(ev[8], stream));
    hipLaunchKernelGGL(k_shade, dim3(list_blocks), dim3(256), 0, stream, sv, fv);
    if (ev) RDR_HIP(hipEventRecord(ev[9], stream));
    RDR_HIP(hipGetLastError());
    return 0;
}
