// rdr_bake.hip — the IBL inputs of the shading pass, baked on the GPU: the BRDF look-up
// table and the importance-sampled prefilter of an environment cubemap.
//
// What is computed follows the reference's bake tool (paths relative to the reference
// tree): assets/shaders/brdf.comp.glsl:20-143 (GGX split sum in r/g, Charlie in b),
// assets/shaders/filter_envmap.frag.glsl:26-245 (Lambertian / GGX / Charlie filter with
// mip-filtered samples), assets/shaders/math.inc.glsl:28-37 (Hammersley points),
// bake/src/brdf.rs:12-80,99-338 (sizes, one pass per face and mip, roughness = mip / (mips-1)).
// One wave per output texel: whatever depends on the sample index alone (the Hammersley point, the direction in
// the texel's tangent frame, the source level of detail) is worked out once per workgroup into LDS, the 64 lanes
// take every 64th sample, and the partial sums meet in a butterfly.  Even the smallest mip levels (24 texels) then
// fill waves, and the per-sample square roots, sines and logarithms are paid once per workgroup instead of once per
// texel.  The per-sample values are those of the one-thread-per-texel form; only the order of the sums differs.
#include <vector>

#include "rdr_device.h"

namespace {

constexpr float kPi = 3.1415926535897932384626433832795f;

RDR_DEV void hammersley2d(uint32_t i, uint32_t n, float& x, float& y) {
    x = (float)i / (float)n;
    y = (float)__brev(i) * 2.3283064365386963e-10f;
}

RDR_DEV float glsl_mod(float x, float y) { return x - y * floorf(x / y); }

RDR_DEV float bake_random(float cx, float cy) {
    const float dt = cx * 12.9898f + cy * 78.233f;
    const float sn = glsl_mod(dt, 3.14f);
    const float v = sinf(sn) * 43758.5453f;
    return v - floorf(v);
}

RDR_DEV vec3 to_normal_frame(vec3 h, vec3 normal) {
    const vec3 up = fabsf(normal.z) < 0.999f ? mk3(0.0f, 0.0f, 1.0f) : mk3(1.0f, 0.0f, 0.0f);
    const vec3 tx = normalize3(cross3(up, normal));
    const vec3 ty = normalize3(cross3(normal, tx));
    return normalize3((tx * h.x + ty * h.y) + normal * h.z);
}

RDR_DEV float d_charlie(float sheen_roughness, float ndh) {
    sheen_roughness = fmaxf(sheen_roughness, 0.000001f);
    const float inv_r = 1.0f / sheen_roughness;
    const float sin2h = 1.0f - ndh * ndh;
    return (2.0f + inv_r) * powf(sin2h, inv_r * 0.5f) / (2.0f * kPi);
}

RDR_DEV float wave_sum_f32(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

constexpr uint32_t kBakeWaves = 4;        // waves (texels in flight) per workgroup
constexpr uint32_t kBakeChunk = 2048;     // samples staged in LDS at a time

// Grid: one workgroup of kBakeWaves waves per kBakeWaves consecutive texels (row-major).
__global__ __launch_bounds__(kBakeWaves * 64) void k_bake_brdf_lut(uint32_t size, uint32_t samples, float4* out) {
    // per sample: cos and sin of its azimuth with the GGX loop's jitter (x, y) and without, for the Charlie loop (z, w)
    __shared__ float4 azimuth[kBakeChunk];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t texel = blockIdx.x * kBakeWaves + wave;
    const bool live = texel < size * size; // whole waves: a wave without a texel still helps to fill the table
    const uint32_t gx = live ? texel % size : 0u, gy = live ? texel / size : 0u;
    const float ndv_in = ((float)gx + 0.5f) / (float)size;
    const float roughness = 1.0f - ((float)gy + 0.5f) / (float)size;
    const vec3 n = mk3(0.0f, 0.0f, 1.0f);
    const vec3 v = mk3(sqrtf(1.0f - ndv_in * ndv_in), 0.0f, ndv_in);
    const float alpha = roughness * roughness;
    const float jitter = bake_random(n.x, n.z) * 0.1f;
    float lr = 0.0f, lg = 0.0f, lb = 0.0f;
    for (uint32_t s0 = 0; s0 < samples; s0 += kBakeChunk) {
        const uint32_t chunk = min(kBakeChunk, samples - s0);
        __syncthreads(); // the previous chunk's readers are done
        for (uint32_t k = threadIdx.x; k < chunk; k += kBakeWaves * 64u) {
            float xi_x, xi_y;
            hammersley2d(s0 + k, samples, xi_x, xi_y);
            const float phi_j = 2.0f * kPi * xi_x + jitter, phi = 2.0f * kPi * xi_x;
            azimuth[k] = make_float4(cosf(phi_j), sinf(phi_j), cosf(phi), sinf(phi));
        }
        __syncthreads();
        if (!live) continue;
        for (uint32_t k = lane; k < chunk; k += 64u) {
            const float4 az = azimuth[k];
            const float xi_y = (float)__brev(s0 + k) * 2.3283064365386963e-10f; // hammersley2d's second coordinate
            {   // GGX importance sample: scale and bias of the split sum
                const float cos_t = sqrtf((1.0f - xi_y) / (1.0f + (alpha * alpha - 1.0f) * xi_y));
                const float sin_t = sqrtf(1.0f - cos_t * cos_t);
                const vec3 h = to_normal_frame(mk3(sin_t * az.x, sin_t * az.y, cos_t), n);
                const float vdh_s = dot3(v, h);
                const vec3 l = h * (2.0f * vdh_s) - v;
                const float ndl = fmaxf(dot3(n, l), 0.0f), ndv = fmaxf(dot3(n, v), 0.0f);
                const float vdh = fmaxf(vdh_s, 0.0f), hdn = fmaxf(dot3(h, n), 0.0f);
                if (ndl > 0.0f) {
                    const float kk = (roughness * roughness) / 2.0f;
                    const float g = (ndl / (ndl * (1.0f - kk) + kk)) * (ndv / (ndv * (1.0f - kk) + kk));
                    const float g_vis = (g * vdh) / (hdn * ndv);
                    const float fc = powf(1.0f - vdh, 5.0f);
                    lr += (1.0f - fc) * g_vis;
                    lg += fc * g_vis;
                }
            }
            {   // Charlie importance sample: the sheen term
                const float sin_t = powf(xi_y, alpha / (2.0f * alpha + 1.0f));
                const float cos_t = sqrtf(1.0f - sin_t * sin_t);
                const vec3 h = to_normal_frame(mk3(sin_t * az.z, sin_t * az.w, cos_t), n);
                const float vdh_s = dot3(v, h);
                const vec3 l = h * (2.0f * vdh_s) - v;
                const float ndl = fmaxf(dot3(n, l), 0.0f), ndv = fmaxf(dot3(n, v), 0.0f);
                const float vdh = fmaxf(vdh_s, 0.0f), hdn = fmaxf(dot3(h, n), 0.0f);
                if (ndl > 0.0f) {
                    const float dist = d_charlie(roughness, hdn);
                    const float vis = clampf(1.0f / (4.0f * (ndl + ndv - ndl * ndv)), 0.0f, 1.0f);
                    lb += vis * dist * ndl * vdh;
                }
            }
        }
    }
    lr = wave_sum_f32(lr); lg = wave_sum_f32(lg); lb = wave_sum_f32(lb);
    if (live && lane == 0u) out[texel] = make_float4(lr / (float)samples, lg / (float)samples, lb / (float)samples, 0.0f);
}

RDR_DEV vec3 uv_to_xyz(uint32_t face, float u, float v) {
    switch (face) {
    case 0: return mk3(1.0f, v, -u);
    case 1: return mk3(-1.0f, v, u);
    case 2: return mk3(u, 1.0f, -v);
    case 3: return mk3(u, -1.0f, v);
    case 4: return mk3(u, v, 1.0f);
    default: return mk3(-u, v, -1.0f);
    }
}

// One launch per mip level; texel t = (face * n + j) * n + i, kBakeWaves consecutive texels per workgroup.
__global__ __launch_bounds__(kBakeWaves * 64) void k_bake_env_map(GpuCubemap src, const uint8_t* texels, uint32_t distribution, uint32_t base_size,
                                                                 uint32_t n_px, float roughness, uint32_t samples, float4* dst) {
    // per sample, the same for every texel of the level: the direction in the texel's frame (x, y, z) and the source
    // level of detail that matches the sample's solid angle (w)
    __shared__ float4 table[kBakeChunk];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t texel = blockIdx.x * kBakeWaves + wave;
    const bool live = texel < 6u * n_px * n_px;
    const uint32_t i = live ? texel % n_px : 0u, j = live ? (texel / n_px) % n_px : 0u, face = live ? texel / (n_px * n_px) : 0u;
    const float uvx = ((float)i + 0.5f) / (float)n_px, uvy = 1.0f - ((float)j + 0.5f) / (float)n_px;
    const vec3 n = normalize3(uv_to_xyz(face, uvx * 2.0f - 1.0f, uvy * 2.0f - 1.0f));
    vec3 bitangent = mk3(0.0f, 1.0f, 0.0f);
    if (1.0f - fabsf(n.y) <= 0.0000001f) bitangent = n.y > 0.0f ? mk3(0.0f, 0.0f, 1.0f) : mk3(0.0f, 0.0f, -1.0f);
    const vec3 tangent = normalize3(cross3(bitangent, n));
    bitangent = cross3(n, tangent);
    const float alpha = roughness * roughness;
    float cr = 0.0f, cg = 0.0f, cb = 0.0f, weights = 0.0f;
    for (uint32_t s0 = 0; s0 < samples; s0 += kBakeChunk) {
        const uint32_t chunk = min(kBakeChunk, samples - s0);
        __syncthreads(); // the previous chunk's readers are done
        for (uint32_t k = threadIdx.x; k < chunk; k += kBakeWaves * 64u) {
            float xi_x, xi_y;
            hammersley2d(s0 + k, samples, xi_x, xi_y);
            const float phi = 2.0f * kPi * xi_x;
            float cos_t, sin_t, pdf;
            if (distribution == 0u) {
                cos_t = sqrtf(1.0f - xi_y);
                sin_t = sqrtf(xi_y);
                pdf = cos_t / kPi;
            } else if (distribution == 1u) {
                cos_t = clampf(sqrtf((1.0f - xi_y) / (1.0f + (alpha * alpha - 1.0f) * xi_y)), 0.0f, 1.0f);
                sin_t = sqrtf(1.0f - cos_t * cos_t);
                const float a = cos_t * alpha;
                const float kk = alpha / (1.0f - cos_t * cos_t + a * a);
                pdf = kk * kk * (1.0f / kPi) / 4.0f;
            } else {
                sin_t = powf(xi_y, alpha / (2.0f * alpha + 1.0f));
                cos_t = sqrtf(1.0f - sin_t * sin_t);
                pdf = d_charlie(alpha, cos_t) / 4.0f;
            }
            const vec3 local = normalize3(mk3(sin_t * cosf(phi), sin_t * sinf(phi), cos_t));
            float lod = 0.5f * log2f(6.0f * (float)base_size * (float)base_size / ((float)samples * pdf));
            if (distribution != 0u && roughness == 0.0f) lod = 0.0f;
            table[k] = make_float4(local.x, local.y, local.z, lod);
        }
        __syncthreads();
        if (!live) continue;
        for (uint32_t k = lane; k < chunk; k += 64u) {
            const float4 e = table[k];
            const vec3 h = (tangent * e.x + bitangent * e.y) + n * e.z;
            if (distribution == 0u) {
                const vec4 c = cube_sample_lod(src, texels, h, e.w);
                cr += c.x; cg += c.y; cb += c.z;
            } else {
                const vec3 mn = n * -1.0f;
                const float d = dot3(h, mn);
                const vec3 l = normalize3(mn - h * (2.0f * d));
                const float ndl = dot3(n, l);
                if (ndl > 0.0f) {
                    const vec4 c = cube_sample_lod(src, texels, l, e.w);
                    cr += c.x * ndl; cg += c.y * ndl; cb += c.z * ndl;
                    weights += ndl;
                }
            }
        }
    }
    cr = wave_sum_f32(cr); cg = wave_sum_f32(cg); cb = wave_sum_f32(cb); weights = wave_sum_f32(weights);
    const float div = weights != 0.0f ? weights : (float)samples;
    if (live && lane == 0u) dst[texel] = make_float4(cr / div, cg / div, cb / div, 1.0f);
}

uint32_t mip_size(uint32_t d, uint32_t level) {
    const uint32_t v = d >> level;
    return v ? v : 1u;
}

} // namespace

extern "C" int rdr_bake_brdf_lut(int hip_device, uint32_t size, uint32_t samples, float* out_rgba32f) {
    RDR_CHECK(out_rgba32f && size >= 1 && size <= 4096 && samples >= 1, "rdr_bake_brdf_lut: bad argument");
    RDR_HIP(hipSetDevice(hip_device));
    float4* dev = nullptr;
    const size_t bytes = (size_t)size * size * sizeof(float4);
    RDR_HIP(hipMalloc(&dev, bytes));
    hipLaunchKernelGGL(k_bake_brdf_lut, dim3((size * size + kBakeWaves - 1u) / kBakeWaves), dim3(kBakeWaves * 64u), 0, 0, size, samples, dev);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out_rgba32f, dev, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    RDR_HIP(e);
    return 0;
}

extern "C" int rdr_bake_env_map(int hip_device, const rdr_cubemap_desc* skybox, uint32_t distribution, uint32_t size,
                                uint32_t mips, uint32_t samples, float* out_rgba32f) {
    RDR_CHECK(skybox && skybox->data && out_rgba32f, "rdr_bake_env_map: null argument");
    RDR_CHECK(skybox->size && skybox->mip_count >= 1 && skybox->mip_count <= 17 &&
                  (skybox->format == RDR_FORMAT_RGBA8_UNORM || skybox->format == RDR_FORMAT_RGBA32F),
              "rdr_bake_env_map: source cubemap invalid (size, 1..17 mips, RGBA8_UNORM or RGBA32F)");
    RDR_CHECK(distribution <= 2u && size >= 1 && size <= 8192 && mips >= 1 && mips <= 14 && samples >= 1,
              "rdr_bake_env_map: bad distribution / size / mips / samples");
    RDR_HIP(hipSetDevice(hip_device));
    GpuCubemap g{};
    std::vector<uint8_t> padded;
    rdr_pack_cubemap(*skybox, g, padded);
    const size_t src_bytes = padded.size();
    g.offset = 0;
    size_t out_texels = 0;
    for (uint32_t l = 0; l < mips; ++l) out_texels += (size_t)6 * mip_size(size, l) * mip_size(size, l);
    uint8_t* src_dev = nullptr;
    float4* out_dev = nullptr;
    RDR_HIP(hipMalloc(&src_dev, src_bytes));
    hipError_t e = hipMalloc(&out_dev, out_texels * sizeof(float4));
    if (e == hipSuccess) e = hipMemcpy(src_dev, padded.data(), src_bytes, hipMemcpyHostToDevice);
    size_t base = 0;
    for (uint32_t l = 0; l < mips && e == hipSuccess; ++l) {
        const uint32_t n = mip_size(size, l);
        const float roughness = mips > 1 ? (float)l / (float)(mips - 1) : 0.0f;
        hipLaunchKernelGGL(k_bake_env_map, dim3((6u * n * n + kBakeWaves - 1u) / kBakeWaves), dim3(kBakeWaves * 64u), 0, 0, g, src_dev, distribution,
                           size, n, roughness, samples, out_dev + base);
        e = hipGetLastError();
        base += (size_t)6 * n * n;
    }
    if (e == hipSuccess) e = hipMemcpy(out_rgba32f, out_dev, out_texels * sizeof(float4), hipMemcpyDeviceToHost);
    (void)hipFree(src_dev);
    (void)hipFree(out_dev);
    RDR_HIP(e);
    return 0;
}
