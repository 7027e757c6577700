// rdr_bc.hip — block-compressed textures (BC1, BC2, BC3, BC4 unorm, BC7) expanded to RGBA8 on the
// GPU when a scene is uploaded.
//
// The reference hands the blocks of a .dds file to Vulkan unchanged
// (renderer/src/renderer/image_loader.rs:63-135) and the sampler hardware decodes them; here the
// sampler is software over RGBA8 texels, so the blocks are decoded once, at upload, and filtering
// then sees exactly the texel values a block decoder produces.  The interpolated palette entries
// follow the usual integer forms, rounded to nearest: (2a + b + 1) / 3 for colours, and
// ((7 - i)a + ib + 3) / 7 or ((5 - i)a + ib + 2) / 5 for the single-channel ramp.  BC7 is specified
// bit-exactly: ((64 - w)a + wb + 32) >> 6 with the format's weight tables.
#include <hip/hip_runtime.h>

#include <mutex>

#include "rdr_internal.h"

#define BC7_TABLE static __device__ const
#include "rdr_bc7_tables.h"

// Staging memory of rdr_decode_bc_noalloc: the blocks of a mip chain in the first part, the decoded
// texels in the second.  It lives in the code object, so every GPU the library runs on holds its
// own copy, at its own address, until the process ends.  In host code this name is only a handle
// for hipGetSymbolAddress: device code and copies go through the per-device address kept below.
constexpr size_t kBcScratchBlocks = (size_t)3 << 19; // 1.5 MiB: blocks of a full 1024 x 1024 chain at 16 bytes per block
constexpr size_t kBcScratchTexels = (size_t)6 << 20; // 6 MiB: its decoded RGBA8 levels, each padded to 16 bytes
static __device__ __align__(256) uint8_t scratch[kBcScratchBlocks + kBcScratchTexels];

namespace {

uint32_t level_dim(uint32_t d, uint32_t l) {
    const uint32_t v = d >> l;
    return v ? v : 1u;
}

struct Scratch { // device allocation released on every return path
    void* ptr = nullptr;
    ~Scratch() {
        if (ptr) (void)hipFree(ptr);
    }
};

__device__ inline uint32_t rgb565(uint32_t c) {
    const uint32_t r = (c >> 11) & 31u, g = (c >> 5) & 63u, b = c & 31u;
    return ((r << 3) | (r >> 2)) | (((g << 2) | (g >> 4)) << 8) | (((b << 3) | (b >> 2)) << 16);
}

__device__ inline uint32_t mix3(uint32_t a, uint32_t b) { // per channel (2a + b + 1) / 3
    uint32_t out = 0;
#pragma unroll
    for (int s = 0; s < 24; s += 8) out |= ((2u * ((a >> s) & 255u) + ((b >> s) & 255u) + 1u) / 3u) << s;
    return out;
}

__device__ inline uint32_t mix2(uint32_t a, uint32_t b) { // per channel (a + b) / 2
    uint32_t out = 0;
#pragma unroll
    for (int s = 0; s < 24; s += 8) out |= ((((a >> s) & 255u) + ((b >> s) & 255u)) / 2u) << s;
    return out;
}

// The colour half of a block: two RGB565 end points and sixteen 2-bit selectors.  `punch`: the
// BC1 rule that c0 <= c1 means three colours plus transparent black.
__device__ inline void color_block(const uint8_t* b, bool punch, uint32_t* px) {
    const uint32_t c0 = b[0] | (b[1] << 8), c1 = b[2] | (b[3] << 8);
    const uint32_t sel = b[4] | (b[5] << 8) | (b[6] << 16) | ((uint32_t)b[7] << 24);
    uint32_t pal[4];
    pal[0] = rgb565(c0) | 0xFF000000u;
    pal[1] = rgb565(c1) | 0xFF000000u;
    if (!punch || c0 > c1) {
        pal[2] = mix3(pal[0], pal[1]) | 0xFF000000u;
        pal[3] = mix3(pal[1], pal[0]) | 0xFF000000u;
    } else {
        pal[2] = mix2(pal[0], pal[1]) | 0xFF000000u;
        pal[3] = 0u;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) px[i] = pal[(sel >> (2 * i)) & 3u];
}

// The single-channel ramp of BC3's alpha and of BC4: two 8-bit end points, sixteen 3-bit selectors.
__device__ inline void ramp_block(const uint8_t* b, uint32_t* v) {
    const uint32_t a0 = b[0], a1 = b[1];
    uint32_t pal[8];
    pal[0] = a0; pal[1] = a1;
    if (a0 > a1) {
#pragma unroll
        for (uint32_t i = 1; i < 7; ++i) pal[i + 1] = ((7u - i) * a0 + i * a1 + 3u) / 7u;
    } else {
#pragma unroll
        for (uint32_t i = 1; i < 5; ++i) pal[i + 1] = ((5u - i) * a0 + i * a1 + 2u) / 5u;
        pal[6] = 0u; pal[7] = 255u;
    }
    const unsigned long long sel = (unsigned long long)b[2] | ((unsigned long long)b[3] << 8) | ((unsigned long long)b[4] << 16) |
                                   ((unsigned long long)b[5] << 24) | ((unsigned long long)b[6] << 32) | ((unsigned long long)b[7] << 40);
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = pal[(sel >> (3 * i)) & 7u];
}

// ---- BC7 ----
// Per mode: subsets, partition bits, rotation bits, index-selector bit, colour bits, alpha bits,
// one p bit per end point, one p bit per subset, bits of the first and of the second index set.
struct Bc7Mode {
    uint8_t subsets, partition_bits, rotation_bits, selector_bit, color_bits, alpha_bits, endpoint_p, shared_p, index_bits, index2_bits;
};
__device__ const Bc7Mode kBc7Modes[8] = {
    {3, 4, 0, 0, 4, 0, 1, 0, 3, 0}, {2, 6, 0, 0, 6, 0, 0, 1, 3, 0}, {3, 6, 0, 0, 5, 0, 0, 0, 2, 0}, {2, 6, 0, 0, 7, 0, 1, 0, 2, 0},
    {1, 0, 2, 1, 5, 6, 0, 0, 2, 3}, {1, 0, 2, 0, 7, 8, 0, 0, 2, 2}, {1, 0, 0, 0, 7, 7, 1, 0, 4, 0}, {2, 6, 0, 0, 5, 5, 1, 0, 2, 0},
};
// interpolation weights of 2-, 3- and 4-bit indices, at offsets 0, 4 and 12
__device__ const uint8_t kBc7Weights[28] = {0, 21, 43, 64, 0, 9, 18, 27, 37, 46, 55, 64, 0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64};

// The 128 bits of a block as two words; `at` advances over the fields, lowest bit first.
struct Bc7Bits {
    unsigned long long lo, hi;
    uint32_t at;
    __device__ uint32_t take(uint32_t count) { // count <= 8
        unsigned long long v;
        if (at >= 64u) v = hi >> (at - 64u);
        else v = at ? (lo >> at) | (hi << (64u - at)) : lo;
        at += count;
        return (uint32_t)v & ((1u << count) - 1u);
    }
};

// Widens a `bits`-wide end-point value to eight bits by repeating its top bits below it.
__device__ inline uint32_t bc7_widen(uint32_t v, uint32_t bits) {
    v <<= 8u - bits;
    return v | (v >> bits);
}

__device__ inline uint32_t bc7_mix(uint32_t a, uint32_t b, uint32_t w) { return ((64u - w) * a + w * b + 32u) >> 6; }

// One BC7 block -> sixteen RGBA8 texels, row-major.
__device__ void bc7_block(const uint8_t* b, uint32_t* px) {
    Bc7Bits in{0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        in.lo |= (unsigned long long)b[i] << (8 * i);
        in.hi |= (unsigned long long)b[8 + i] << (8 * i);
    }
    const uint32_t low = (uint32_t)in.lo & 255u;
    if (low == 0u) { // the reserved mode: all zero
#pragma unroll
        for (int t = 0; t < 16; ++t) px[t] = 0u;
        return;
    }
    const uint32_t mode = (uint32_t)__ffs((int)low) - 1u;
    const Bc7Mode m = kBc7Modes[mode];
    in.at = mode + 1u;
    const uint32_t partition = in.take(m.partition_bits);
    const uint32_t rotation = in.take(m.rotation_bits);
    const uint32_t selector = in.take(m.selector_bit);
    const uint32_t ends = 2u * m.subsets;
    uint32_t e[6][4];
    for (uint32_t ch = 0; ch < 3u; ++ch)
        for (uint32_t k = 0; k < ends; ++k) e[k][ch] = in.take(m.color_bits);
    for (uint32_t k = 0; k < ends; ++k) e[k][3] = in.take(m.alpha_bits);
    const uint32_t extra = m.endpoint_p | m.shared_p;
    if (extra) {
        uint32_t p = 0;
        for (uint32_t k = 0; k < ends; ++k) {
            if (m.endpoint_p || !(k & 1u)) p = in.take(1u);
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) e[k][ch] = (e[k][ch] << 1) | p;
        }
    }
    const uint32_t cbits = m.color_bits + extra, abits = m.alpha_bits ? m.alpha_bits + extra : 0u;
    for (uint32_t k = 0; k < ends; ++k) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) e[k][ch] = bc7_widen(e[k][ch], cbits);
        e[k][3] = abits ? bc7_widen(e[k][3], abits) : 255u;
    }
    uint32_t shape = 0u, anchor1 = 16u, anchor2 = 16u; // 16: no such anchor
    if (m.subsets == 2) {
        shape = kBc7Shape2[partition];
        anchor1 = kBc7Anchor2[partition];
    } else if (m.subsets == 3) {
        shape = kBc7Shape3[partition];
        anchor1 = kBc7Anchor3a[partition];
        anchor2 = kBc7Anchor3b[partition];
    }
    // the second index set follows the first, whose length is 16 indices less one bit per subset
    Bc7Bits in2 = in;
    in2.at = in.at + 16u * m.index_bits - m.subsets;
    const uint32_t table1 = m.index_bits == 2 ? 0u : (m.index_bits == 3 ? 4u : 12u);
    const uint32_t table2 = m.index2_bits == 2 ? 0u : 4u;
    for (uint32_t t = 0; t < 16u; ++t) {
        const uint32_t subset = (shape >> (2u * t)) & 3u;
        const bool is_anchor = t == 0u || t == anchor1 || t == anchor2;
        uint32_t wc = kBc7Weights[table1 + in.take(m.index_bits - (is_anchor ? 1u : 0u))], wa = wc;
        if (m.index2_bits) {
            const uint32_t w2 = kBc7Weights[table2 + in2.take(m.index2_bits - (t == 0u ? 1u : 0u))];
            if (selector) wc = w2;
            else wa = w2;
        }
        const uint32_t* e0 = e[2u * subset];
        const uint32_t* e1 = e[2u * subset + 1u];
        uint32_t c[4];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) c[ch] = bc7_mix(e0[ch], e1[ch], wc);
        c[3] = bc7_mix(e0[3], e1[3], wa);
        if (rotation == 1u) { const uint32_t s = c[3]; c[3] = c[0]; c[0] = s; }
        else if (rotation == 2u) { const uint32_t s = c[3]; c[3] = c[1]; c[1] = s; }
        else if (rotation == 3u) { const uint32_t s = c[3]; c[3] = c[2]; c[2] = s; }
        px[t] = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
    }
}

// One thread per 4x4 block; texels outside the level (levels smaller than a block) are dropped.
__global__ __launch_bounds__(256) void k_decode_bc(uint32_t format, uint32_t width, uint32_t height, const uint8_t* blocks, uint32_t* out) {
    const uint32_t nbx = (width + 3u) / 4u, nby = (height + 3u) / 4u;
    const uint32_t bi = blockIdx.x * blockDim.x + threadIdx.x;
    if (bi >= nbx * nby) return;
    const uint32_t bx = bi % nbx, by = bi / nbx;
    const uint32_t block_bytes = (format == RDR_FORMAT_BC1 || format == RDR_FORMAT_BC4) ? 8u : 16u;
    const uint8_t* b = blocks + (size_t)bi * block_bytes;
    uint32_t px[16];
    if (format >= RDR_FORMAT_BC7) {
        bc7_block(b, px);
    } else if (format == RDR_FORMAT_BC1) {
        color_block(b, true, px);
    } else if (format == RDR_FORMAT_BC2) {
        color_block(b + 8, false, px);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t a4 = (b[i >> 1] >> ((i & 1) * 4)) & 15u;
            px[i] = (px[i] & 0x00FFFFFFu) | ((a4 * 17u) << 24);
        }
    } else if (format == RDR_FORMAT_BC3) {
        uint32_t a[16];
        color_block(b + 8, false, px);
        ramp_block(b, a);
#pragma unroll
        for (int i = 0; i < 16; ++i) px[i] = (px[i] & 0x00FFFFFFu) | (a[i] << 24);
    } else { // BC4: red from the ramp, green = blue = 0, alpha = 1
        uint32_t v[16];
        ramp_block(b, v);
#pragma unroll
        for (int i = 0; i < 16; ++i) px[i] = v[i] | 0xFF000000u;
    }
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
        const uint32_t y = by * 4u + j;
        if (y >= height) break;
        uint32_t* row = out + (size_t)y * width + bx * 4u;
        if (bx * 4u + 3u < width && (width & 3u) == 0u) {
            *reinterpret_cast<uint4*>(row) = make_uint4(px[j * 4], px[j * 4 + 1], px[j * 4 + 2], px[j * 4 + 3]);
        } else {
            for (uint32_t i = 0; i < 4 && bx * 4u + i < width; ++i) row[i] = px[j * 4 + i];
        }
    }
}

} // namespace

size_t rdr_bc_level_bytes(uint32_t format, uint32_t width, uint32_t height) {
    const size_t block_bytes = (format == RDR_FORMAT_BC1 || format == RDR_FORMAT_BC4) ? 8 : 16;
    return (size_t)((width + 3u) / 4u) * ((height + 3u) / 4u) * block_bytes;
}

// Decodes one level: `blocks` and `out` are device pointers, `out` width * height RGBA8 texels,
// 16-byte aligned when the width is a multiple of four.
int rdr_launch_decode_bc(uint32_t format, uint32_t width, uint32_t height, const uint8_t* blocks, uint32_t* out, hipStream_t stream) {
    RDR_CHECK(format >= RDR_FORMAT_BC1 && format <= RDR_FORMAT_BC7_SRGB, "decode_bc: format is not a block-compressed one");
    RDR_CHECK(width && height, "decode_bc: empty level");
    const uint32_t n = ((width + 3u) / 4u) * ((height + 3u) / 4u);
    hipLaunchKernelGGL(k_decode_bc, dim3((n + 255u) / 256u), dim3(256), 0, stream, format, width, height, blocks, out);
    RDR_HIP(hipGetLastError());
    return 0;
}

namespace {
// Address of `scratch` on each device, looked up once per device with that device current.
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
uint8_t* g_scratch_ptr[kMaxDevices] = {};

int scratch_on_device(int device, uint8_t** out) {
    RDR_CHECK(device >= 0 && device < kMaxDevices, "decode_bc_noalloc: device index out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        RDR_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        RDR_CHECK(p != nullptr, "decode_bc_noalloc: the staging array has no device address");
        g_scratch_ptr[device] = static_cast<uint8_t*>(p);
    }
    *out = g_scratch_ptr[device];
    return 0;
}
} // namespace

// The variant of rdr_decode_bc for callers that must not allocate: same arguments, same result, its
// two staging buffers replaced by the file-scope array above.  Calls no hipMalloc / hipFree.
extern "C" int rdr_decode_bc_noalloc(int device, uint32_t format, uint32_t width, uint32_t height, uint32_t mip_count,
                                     const void* blocks, size_t block_bytes, void* out_rgba8) {
    RDR_CHECK(format >= RDR_FORMAT_BC1 && format <= RDR_FORMAT_BC7_SRGB, "decode_bc_noalloc: format must be RDR_FORMAT_BC1 .. RDR_FORMAT_BC7_SRGB");
    RDR_CHECK(blocks && out_rgba8, "decode_bc_noalloc: null pointer");
    RDR_CHECK(width && height && mip_count >= 1 && mip_count <= 17, "decode_bc_noalloc: dimensions / mip count invalid");
    size_t need = 0, padded = 0;
    size_t at[17]; // every decoded level starts on a 16-byte boundary of its own
    for (uint32_t l = 0; l < mip_count; ++l) {
        const uint32_t w = level_dim(width, l), h = level_dim(height, l);
        need += rdr_bc_level_bytes(format, w, h);
        at[l] = padded;
        padded += ((size_t)w * h + 3) & ~(size_t)3;
    }
    RDR_CHECK(block_bytes >= need, "decode_bc_noalloc: block data shorter than the mip chain needs");
    RDR_CHECK(need <= kBcScratchBlocks && padded * 4 <= kBcScratchTexels,
              "decode_bc_noalloc: mip chain larger than the built-in staging memory (1.5 MiB of blocks, 6 MiB of "
              "decoded texels: up to a full 1024 x 1024 chain); use rdr_decode_bc");
    struct DeviceRestore { // the caller's current device is put back on every return path
        int previous = -1;
        ~DeviceRestore() {
            if (previous >= 0) (void)hipSetDevice(previous);
        }
    } restore;
    if (hipGetDevice(&restore.previous) != hipSuccess) restore.previous = -1;
    RDR_HIP(hipSetDevice(device));
    uint8_t* base = nullptr;
    if (scratch_on_device(device, &base)) return 1;
    uint8_t* src = base;
    uint32_t* dst = reinterpret_cast<uint32_t*>(base + kBcScratchBlocks);
    RDR_HIP(hipMemcpy(src, blocks, need, hipMemcpyHostToDevice));
    size_t off = 0;
    for (uint32_t l = 0; l < mip_count; ++l) {
        const uint32_t w = level_dim(width, l), h = level_dim(height, l);
        if (rdr_launch_decode_bc(format, w, h, src + off, dst + at[l], nullptr)) return 1;
        off += rdr_bc_level_bytes(format, w, h);
    }
    RDR_HIP(hipStreamSynchronize(nullptr));
    uint32_t* o = static_cast<uint32_t*>(out_rgba8);
    for (uint32_t l = 0; l < mip_count; ++l) {
        const size_t n = (size_t)level_dim(width, l) * level_dim(height, l);
        RDR_HIP(hipMemcpy(o, dst + at[l], n * 4, hipMemcpyDeviceToHost));
        o += n;
    }
    return 0;
}

// C-ABI: a whole mip chain from host memory to host memory (tools, tests); the scene upload uses
// the same kernel without the trip back.
extern "C" int rdr_decode_bc(int device, uint32_t format, uint32_t width, uint32_t height, uint32_t mip_count, const void* blocks,
                             size_t block_bytes, void* out_rgba8) {
    RDR_CHECK(format >= RDR_FORMAT_BC1 && format <= RDR_FORMAT_BC7_SRGB, "decode_bc: format must be RDR_FORMAT_BC1 .. RDR_FORMAT_BC7_SRGB");
    RDR_CHECK(blocks && out_rgba8, "decode_bc: null pointer");
    RDR_CHECK(width && height && mip_count >= 1 && mip_count <= 17, "decode_bc: dimensions / mip count invalid");
    size_t need = 0, padded = 0;
    std::vector<size_t> at(mip_count); // every decoded level starts on a 16-byte boundary of its own
    for (uint32_t l = 0; l < mip_count; ++l) {
        const uint32_t w = level_dim(width, l), h = level_dim(height, l);
        need += rdr_bc_level_bytes(format, w, h);
        at[l] = padded;
        padded += ((size_t)w * h + 3) & ~(size_t)3;
    }
    RDR_CHECK(block_bytes >= need, "decode_bc: block data shorter than the mip chain needs");
    RDR_HIP(hipSetDevice(device));
    Scratch src, dst;
    RDR_HIP(hipMalloc(&src.ptr, need));
    RDR_HIP(hipMalloc(&dst.ptr, padded * 4));
    RDR_HIP(hipMemcpy(src.ptr, blocks, need, hipMemcpyHostToDevice));
    size_t off = 0;
    for (uint32_t l = 0; l < mip_count; ++l) {
        const uint32_t w = level_dim(width, l), h = level_dim(height, l);
        if (rdr_launch_decode_bc(format, w, h, static_cast<const uint8_t*>(src.ptr) + off, static_cast<uint32_t*>(dst.ptr) + at[l], nullptr)) return 1;
        off += rdr_bc_level_bytes(format, w, h);
    }
    RDR_HIP(hipDeviceSynchronize());
    uint32_t* o = static_cast<uint32_t*>(out_rgba8);
    for (uint32_t l = 0; l < mip_count; ++l) {
        const size_t n = (size_t)level_dim(width, l) * level_dim(height, l);
        RDR_HIP(hipMemcpy(o, static_cast<uint32_t*>(dst.ptr) + at[l], n * 4, hipMemcpyDeviceToHost));
        o += n;
    }
    return 0;
}
