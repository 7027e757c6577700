// rdr_shade.hip — deferred glTF-PBR shading of the covered pixels and the sRGB store.
//
// What is computed follows the reference's shading pass (paths relative to the
// reference tree): assets/shaders/shading.frag.glsl:81-171 (analytic
// barycentrics and derivatives), :317-531 (material model, IBL),
// assets/shaders/fullscreen_triangle.vert.glsl:7-9 (pixel -> uv), and the sRGB
// encode an *_SRGB swapchain image performs (lava/src/swapchain.rs:100-115).
//
// Built with the hardware's fast division and square root (the image is held to 2/255, not to bit equality); the
// divisions of the analytic barycentrics alone stay correctly rounded (div_exact in rdr_pixel.h): a triangle many
// screens large cancels catastrophically there and turns a last-bit difference into per cents of colour.
#include "rdr_pixel.h"

namespace {

constexpr float kPi = 3.1415926535897932384626433832795f;
constexpr float kMinRoughness = 0.04f;

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

RDR_DEV vec4 sample_or_default(const SceneView& sv, const float* lut, uint32_t tex, InterpUv uv, vec4 def) {
    if (tex == RDR_NO_TEXTURE) return def;
    if (tex >= sv.image_count) return mk4(0.0f, 0.0f, 0.0f, 0.0f);
    const auto& im = uniform_ref(&sv.images[tex]);
    if (im.mip_count == 0u) return mk4(0.0f, 0.0f, 0.0f, 0.0f);
    return image_sample_grad(im, sv.texels, lut, uv.u, uv.v, uv.dudx, uv.dvdx, uv.dudy, uv.dvdy, true);
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

RDR_DEV vec3 ibl_ggx(const SceneView& sv, vec3 n, vec3 v, float ndv, float rough, vec3 refl0, float spec_w, vec4 brdf) {
    const vec3 r = normalize3_rcp(reflect_neg(v, n));
    const float mips = (float)sv.ggx.mip_count;
    const float lod = rough * (mips - 1.0f);
    const vec4 sl = cube_sample_lod(sv.ggx, sv.texels, r, lod);
    const vec3 fr = mk3(fmaxf(1.0f - rough, refl0.x), fmaxf(1.0f - rough, refl0.y), fmaxf(1.0f - rough, refl0.z)) - refl0;
    const vec3 ks = refl0 + fr * pow5(1.0f - ndv);
    const vec3 fss = ks * brdf.x + splat(brdf.y);
    return (mk3(sl.x, sl.y, sl.z) * spec_w) * fss;
}

RDR_DEV vec3 ibl_lambertian(const SceneView& sv, float ndv, vec3 n, float rough, vec3 diffuse, vec3 f0, float spec_w, vec4 fab) {
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
    const vec3 r = normalize3_rcp(reflect_neg(v, n));
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
    // c^(1/2.4) through the hardware log2 / exp2 (about 1e-6 relative; the result is quantised to 8 bits)
    const float e = c <= 0.0031308f ? 12.92f * c : 1.055f * __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(c) * (1.0f / 2.4f)) - 0.055f;
    return (uint32_t)floorf(e * 255.0f + 0.5f);
}

// `srgb`: the sRGB decode table (SceneView::srgb_lut or a copy of it)
// `material`: the pixel's material, or material_count for an index past the end; the same in every lane that is in here
// together (k_shade takes a wave's materials one at a time), so that everything read of the material, of its textures'
// descriptors, and every branch on its type goes through the scalar unit.
RDR_DEV vec3 shade_pixel(const SceneView& sv, const float* srgb, const FrameView& fv, const ListedPixel& px, uint32_t material, bool shadow_hit) {
    const rdr_ubo& u = fv.ubo;
    const float uvx = px.uvx, uvy = px.uvy;
    float model[16];
    load_m4(sv.transforms + (size_t)px.model_index * 16, model);
    // fields are read where they are used (the copy of all 272 bytes held some forty registers through the texture taps);
    // an index past the end lands on the zero material the upload appends
    const auto& mat = uniform_ref(&sv.materials[material]);
    const rdr_vertex* v1 = &sv.vertices[px.i1];
    const rdr_vertex* v2 = &sv.vertices[px.i2];
    const rdr_vertex* v3 = &sv.vertices[px.i3];

    const vec4 w1 = mul_m4_fma(model, mk4(v1->p[0], v1->p[1], v1->p[2], 1.0f));
    const vec4 w2 = mul_m4_fma(model, mk4(v2->p[0], v2->p[1], v2->p[2], 1.0f));
    const vec4 w3 = mul_m4_fma(model, mk4(v3->p[0], v3->p[1], v3->p[2], 1.0f));
    const vec4 c1 = mul_m4_fma(u.view_proj, w1), c2 = mul_m4_fma(u.view_proj, w2), c3 = mul_m4_fma(u.view_proj, w3);
    const BaryDeriv bary = calc_full_bary(c1, c2, c3, uvx * 2.0f - 1.0f, uvy * 2.0f - 1.0f, fv.two_over_width, fv.two_over_height);

    const vec3 w_pos = interp3(bary, mk3(w1.x, w1.y, w1.z), mk3(w2.x, w2.y, w2.z), mk3(w3.x, w3.y, w3.z));
    // mat3(model) is applied to the interpolated vector, not to the three it is interpolated from: the same
    // value but for rounding (both steps are linear), and a third of the products
    const vec3 norm = mul_m3_fma(model, interp3(bary, vload3(v1->n), vload3(v2->n), vload3(v3->n)));
    const vec3 tangent = mul_m3_fma(model, interp3(bary, vload3(v1->t), vload3(v2->t), vload3(v3->t)));
    const vec3 us = mk3(v1->uv[0], v2->uv[0], v3->uv[0]), vs = mk3(v1->uv[1], v2->uv[1], v3->uv[1]);
    // uv1 is interpolated from .xy as well (shading.frag.glsl:373-374): the uv index never matters
    const InterpUv uv0 = {dot3_fma(us, bary.lambda), dot3_fma(vs, bary.lambda), dot3_fma(us, bary.ddx), dot3_fma(vs, bary.ddx),
                          dot3_fma(us, bary.ddy), dot3_fma(vs, bary.ddy)};

    const vec4 one = mk4(1.0f, 1.0f, 1.0f, 1.0f), nmap_def = mk4(0.5f, 0.5f, 1.0f, 0.0f);
    vec4 albedo = sample_or_default(sv, srgb, mat.albedo_texture,
                                    uv_transform(uv0, mat.albedo_texture_transform[0], mat.albedo_texture_transform[1],
                                                 mat.albedo_texture_transform[2], mat.albedo_texture_transform[3]), one);
    albedo.x *= mat.base_color_factor[0]; albedo.y *= mat.base_color_factor[1];
    albedo.z *= mat.base_color_factor[2]; albedo.w *= mat.base_color_factor[3];
    const vec4 nm4 = sample_or_default(sv, srgb, mat.normal_texture,
                                       uv_transform(uv0, mat.normal_texture_transform[0], mat.normal_texture_transform[1],
                                                    mat.normal_texture_transform[2], mat.normal_texture_transform[3]), nmap_def);
    const vec3 nmap = mk3(nm4.x * 2.0f - 1.0f, nm4.y * 2.0f - 1.0f, nm4.z * 2.0f - 1.0f);
    const vec4 mr = sample_or_default(sv, srgb, mat.metallic_roughness_texture,
                                      uv_transform(uv0, mat.metallic_roughness_texture_transform[0], mat.metallic_roughness_texture_transform[1],
                                                   mat.metallic_roughness_texture_transform[2], mat.metallic_roughness_texture_transform[3]), one);
    const vec4 occ = sample_or_default(sv, srgb, mat.occlusion_texture,
                                       uv_transform(uv0, mat.occlusion_texture_transform[0], mat.occlusion_texture_transform[1],
                                                    mat.occlusion_texture_transform[2], mat.occlusion_texture_transform[3]), one);
    const InterpUv uv_id = uv_transform(uv0, 1.0f, 1.0f, 0.0f, 0.0f);
    const vec4 em4 = sample_or_default(sv, srgb, mat.emissive_texture, uv_id, one);
    const vec3 emissive = mk3(mat.emissive_factor_alpha_cutoff[0] * em4.x, mat.emissive_factor_alpha_cutoff[1] * em4.y,
                              mat.emissive_factor_alpha_cutoff[2] * em4.z);
    const vec4 sc4 = sample_or_default(sv, srgb, mat.specular_color_texture, uv_id, one);
    const vec4 sp4 = sample_or_default(sv, srgb, mat.specular_texture, uv_id, one);
    const vec3 spec_color_factor = mk3(mat.specular_factor[0] * sc4.x, mat.specular_factor[1] * sc4.y, mat.specular_factor[2] * sc4.z);
    const float spec_factor_w = mat.specular_factor[3] * sp4.x;

    const vec3 nnormal = normalize3_rcp(norm);
    const vec3 ntangent = normalize3_rcp(tangent);
    const vec3 bitangent = cross3(nnormal, ntangent) * ((v1->t[3] * v2->t[3]) * v3->t[3]);
    const vec3 nrm = normalize3_rcp((ntangent * nmap.x + bitangent * nmap.y) + nnormal * nmap.z);

    const vec3 L = mk3(fv.sun_unit[0], fv.sun_unit[1], fv.sun_unit[2]);

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
    const vec3 v = normalize3_rcp(w_pos - cam) * -1.0f;
    const vec3 h = normalize3_rcp(L + v);
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

    const vec4 lut = brdf_lut_tap(sv, ndv, prough); // the same tap serves the specular and the diffuse term
    const vec3 ibl_s = ibl_ggx(sv, nrm, v, ndv, prough, r0, spec_w, lut);
    const vec3 ibl_d = ibl_lambertian(sv, ndv, nrm, prough, diffuse, r0, spec_w, lut);
    color = clamp01(((ibl_s + ibl_d) + color) + emissive) * occ.x;

    if (mat.material_type & RDR_MAT_CLEARCOAT) {
        const vec4 cn4 = sample_or_default(sv, srgb, mat.clearcoat_texture, uv_id, nmap_def);
        const vec3 cm = mk3(cn4.x * 2.0f - 1.0f, cn4.y * 2.0f - 1.0f, cn4.z * 2.0f - 1.0f);
        const vec3 cnrm = normalize3_rcp((ntangent * cm.x + bitangent * cm.y) + nnormal * cm.z);
        const float cndv = clampf(fabsf(dot3(cnrm, v)), 0.001f, 1.0f);
        const float cr = mat.clearcoat_factor[1];
        const vec3 cf0 = splat(fi * fi);
        const vec3 contrib = ibl_ggx(sv, cnrm, v, cndv, cr, cf0, 1.0f, brdf_lut_tap(sv, cndv, cr));
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

// Pass 2: material evaluation, lighting, sRGB store of every covered pixel.  The grid is sized for the worst
// case (every pixel covered); workgroups beyond the list's end exit on their first load, which costs nothing
// measurable (a grid cut to the list's length ran in the same time).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_shade(SceneView sv, FrameView fv) {
    // the sRGB decode table in LDS: a trilinear tap of an sRGB texture looks up 24 values, each a dependent load
    // behind its texel if the table stays in memory
    __shared__ float lut[512]; // and unorm8() of every byte behind it
    const uint32_t entry = blockIdx.x * 256u + threadIdx.x;
    if (entry == 0u) // last kernel of the shading half that every frame runs: the frame's status for the host
        __hip_atomic_store(&fv.host_status[1], (fv.frame_seq << 8) | (fv.counters->deep_overflow ? RDR_FRAME_SHADOW_STACK : 0u),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint32_t count = fv.counters->pixel_count;
    if (blockIdx.x * 256u >= count) return; // the whole workgroup
    lut[threadIdx.x] = sv.srgb_lut[threadIdx.x];
    lut[256u + threadIdx.x] = unorm8(threadIdx.x);
    __syncthreads();
    if (entry >= count) return;
    const ListedPixel p = listed_pixel(sv, fv, entry);
    const uint32_t material = p.material_index < sv.material_count ? p.material_index : sv.material_count;
    const bool shadow_hit = fv.shadow[entry] != 0;
    vec3 c = mk3(0.0f, 0.0f, 0.0f);
    // one material of the wave at a time (usually there is one)
    for_each_uniform(material, [&](uint32_t m) { c = shade_pixel(sv, lut, fv, p, m, shadow_hit); });
    reinterpret_cast<uint32_t*>(fv.color)[p.pi] =
        srgb_encode(c.x) | (srgb_encode(c.y) << 8) | (srgb_encode(c.z) << 16) | (255u << 24);
}

// Skybox pass (assets/shaders/cubemap.vert.glsl:41-47, cubemap.frag.glsl:9-12,
// renderer/src/renderer/pass.rs:397-445): a camera-centred cube drawn with the depth test
// GREATER survives exactly on the pixels still at the clear depth, and its interpolated
// direction is the pixel's view ray, so every uncovered pixel shows the cubemap along its
// ray (implicit LOD taken as 0), sRGB-encoded by the store; alpha is stored as is.
__global__ __launch_bounds__(256) void k_skybox(SceneView sv, FrameView fv) {
    const uint32_t ix = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint32_t vrow = blockIdx.y * 4u + (threadIdx.x >> 6); // of the rows this renderer shades
    if (ix >= fv.width || vrow >= fv.shade_rows) return;
    const uint32_t iy = rdr_shade_row(fv, vrow);
    const size_t pi = (size_t)iy * fv.width + ix;
    if (fv.vis[pi] != RDR_VIS_CLEAR) return;
    const rdr_ubo& u = fv.ubo;
    const float nx = (((float)ix + 0.5f) / (float)fv.width) * 2.0f - 1.0f;
    const float ny = (1.0f - ((float)iy + 0.5f) / (float)fv.height) * 2.0f - 1.0f;
    // view-space ray (x/z, y/z, 1) from the projection's scales, then the view rotation transposed
    const float vx = nx / u.proj[0], vy = ny / u.proj[5];
    const vec3 d = mk3((u.view[0] * vx + u.view[1] * vy) + u.view[2], (u.view[4] * vx + u.view[5] * vy) + u.view[6],
                       (u.view[8] * vx + u.view[9] * vy) + u.view[10]);
    const vec4 c = cube_sample_lod(sv.skybox, sv.texels, d, 0.0f);
    const uint32_t a = (uint32_t)floorf(clampf(c.w, 0.0f, 1.0f) * 255.0f + 0.5f);
    reinterpret_cast<uint32_t*>(fv.color)[pi] = srgb_encode(c.x) | (srgb_encode(c.y) << 8) | (srgb_encode(c.z) << 16) | (a << 24);
}

} // namespace

int rdr_launch_shade(const SceneView& sv, const FrameView& fv, hipStream_t stream, hipEvent_t* ev) {
    if (fv.shade_rows == 0u) return 0;
    if (rdr_launch_shadow(sv, fv, stream, ev)) return 1;
    const uint32_t list_blocks = (uint32_t)(((size_t)fv.shade_rows * fv.width + 255u) / 256u);
    hipLaunchKernelGGL(k_shade, dim3(list_blocks), dim3(256), 0, stream, sv, fv);
    if (fv.skybox)
        hipLaunchKernelGGL(k_skybox, dim3((fv.width + 63u) / 64u, (fv.shade_rows + 3u) / 4u), dim3(256), 0, stream, sv, fv);
    if (ev) RDR_HIP(hipEventRecord(ev[9], stream));
    RDR_HIP(hipGetLastError());
    return 0;
}
