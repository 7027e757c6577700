/*
 * orc_bc.c — CPU decode of block-compressed texture levels (BC1, BC2, BC3, BC4 unorm, BC7) to RGBA8.
 *
 * TEST INFRASTRUCTURE ONLY (see rdr_oracle.h).  The reference uploads the blocks of a .dds file to a
 * Vulkan BC image (renderer/src/renderer/image_loader.rs:63-135) and the sampler hardware decodes
 * them; it ships no compressed texture and no decoded image to pin against.
 *   BC1-BC4: PARITY UNPINNED.  They follow the block layouts of the Khronos data format specification
 *   (S3TC / RGTC chapters) with palette entries rounded to nearest (the specification leaves the
 *   rounding of these formats to the implementation), checked against hand-computed blocks.
 *   BC7: the format is specified bit-exactly (BPTC chapter); pinned against an independent decoder
 *   (Pillow's DDS reader) on the blocks of tests/golden/bc7_blocks.npz, all eight modes.
 */
#include <string.h>

#include "orc_bc7_tables.h"
#include "rdr_oracle.h"

static unsigned expand5(unsigned v) { return (v << 3) | (v >> 2); }
static unsigned expand6(unsigned v) { return (v << 2) | (v >> 4); }

/* palette of the colour half: out[k][0..3] = r, g, b, a */
static void color_palette(const uint8_t* b, int punch_through, unsigned out[4][4]) {
    const unsigned c0 = (unsigned)b[0] | ((unsigned)b[1] << 8), c1 = (unsigned)b[2] | ((unsigned)b[3] << 8);
    const unsigned e[2] = {c0, c1};
    for (int k = 0; k < 2; ++k) {
        out[k][0] = expand5(e[k] >> 11);
        out[k][1] = expand6((e[k] >> 5) & 63u);
        out[k][2] = expand5(e[k] & 31u);
        out[k][3] = 255;
    }
    const int four = !punch_through || c0 > c1;
    for (int ch = 0; ch < 3; ++ch) {
        const unsigned a = out[0][ch], c = out[1][ch];
        out[2][ch] = four ? (2 * a + c + 1) / 3 : (a + c) / 2;
        out[3][ch] = four ? (a + 2 * c + 1) / 3 : 0;
    }
    out[2][3] = 255;
    out[3][3] = four ? 255 : 0;
}

/* palette of an 8-byte single-channel block */
static void ramp_palette(const uint8_t* b, unsigned out[8]) {
    const unsigned a0 = b[0], a1 = b[1];
    out[0] = a0;
    out[1] = a1;
    if (a0 > a1) {
        for (unsigned k = 2; k < 8; ++k) out[k] = ((8 - k) * a0 + (k - 1) * a1 + 3) / 7;
    } else {
        for (unsigned k = 2; k < 6; ++k) out[k] = ((6 - k) * a0 + (k - 1) * a1 + 2) / 5;
        out[6] = 0;
        out[7] = 255;
    }
}

static unsigned ramp_selector(const uint8_t* b, unsigned texel) { /* 3 bits per texel from byte 2 on */
    const unsigned bit = 3 * texel;
    const unsigned lo = b[2 + bit / 8], hi = bit / 8 + 1 < 6 ? b[3 + bit / 8] : 0;
    return ((lo | (hi << 8)) >> (bit % 8)) & 7u;
}

static void decode_texel(uint32_t format, const uint8_t* block, unsigned texel, uint8_t rgba[4]) {
    unsigned pal[4][4], ramp[8];
    const uint8_t* color = format == RDR_FORMAT_BC1 ? block : block + 8;
    if (format == RDR_FORMAT_BC4) {
        ramp_palette(block, ramp);
        rgba[0] = (uint8_t)ramp[ramp_selector(block, texel)];
        rgba[1] = rgba[2] = 0;
        rgba[3] = 255;
        return;
    }
    color_palette(color, format == RDR_FORMAT_BC1, pal);
    const unsigned sel = (color[4 + texel / 4] >> (2 * (texel % 4))) & 3u;
    for (int ch = 0; ch < 4; ++ch) rgba[ch] = (uint8_t)pal[sel][ch];
    if (format == RDR_FORMAT_BC2) {
        const unsigned a4 = (block[texel / 2] >> (4 * (texel % 2))) & 15u;
        rgba[3] = (uint8_t)(a4 * 17u);
    } else if (format == RDR_FORMAT_BC3) {
        ramp_palette(block, ramp);
        rgba[3] = (uint8_t)ramp[ramp_selector(block, texel)];
    }
}

/* ---- BC7: eight modes; fields are read from bit 0 of byte 0 upwards ---- */

typedef struct {
    int subsets, partition_bits, rotation_bits, selector_bit, color_bits, alpha_bits, endpoint_p, shared_p, index_bits, index2_bits;
} Bc7Mode;

static const Bc7Mode kBc7Modes[8] = {
    {3, 4, 0, 0, 4, 0, 1, 0, 3, 0}, {2, 6, 0, 0, 6, 0, 0, 1, 3, 0}, {3, 6, 0, 0, 5, 0, 0, 0, 2, 0}, {2, 6, 0, 0, 7, 0, 1, 0, 2, 0},
    {1, 0, 2, 1, 5, 6, 0, 0, 2, 3}, {1, 0, 2, 0, 7, 8, 0, 0, 2, 2}, {1, 0, 0, 0, 7, 7, 1, 0, 4, 0}, {2, 6, 0, 0, 5, 5, 1, 0, 2, 0},
};

static const unsigned kBc7Weights2[4] = {0, 21, 43, 64};
static const unsigned kBc7Weights3[8] = {0, 9, 18, 27, 37, 46, 55, 64};
static const unsigned kBc7Weights4[16] = {0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64};

static unsigned bc7_bits(const uint8_t* b, unsigned* at, unsigned count) {
    unsigned v = 0;
    for (unsigned i = 0; i < count; ++i, ++*at) v |= (unsigned)((b[*at >> 3] >> (*at & 7)) & 1u) << i;
    return v;
}

static unsigned bc7_weight(unsigned bits, unsigned index) {
    return bits == 2 ? kBc7Weights2[index] : bits == 3 ? kBc7Weights3[index] : kBc7Weights4[index];
}

static unsigned bc7_mix(unsigned e0, unsigned e1, unsigned w) { return ((64 - w) * e0 + w * e1 + 32) >> 6; }

static void bc7_block(const uint8_t* b, uint8_t out[16][4]) {
    unsigned mode = 0;
    while (mode < 8 && !((b[0] >> mode) & 1u)) ++mode;
    if (mode == 8) { /* reserved: every texel is zero */
        memset(out, 0, 64);
        return;
    }
    const Bc7Mode m = kBc7Modes[mode];
    unsigned at = mode + 1;
    const unsigned partition = bc7_bits(b, &at, (unsigned)m.partition_bits);
    const unsigned rotation = bc7_bits(b, &at, (unsigned)m.rotation_bits);
    const unsigned selector = bc7_bits(b, &at, (unsigned)m.selector_bit);
    const int ends = 2 * m.subsets;
    unsigned e[6][4];
    for (int ch = 0; ch < 3; ++ch)
        for (int k = 0; k < ends; ++k) e[k][ch] = bc7_bits(b, &at, (unsigned)m.color_bits);
    for (int k = 0; k < ends; ++k) e[k][3] = m.alpha_bits ? bc7_bits(b, &at, (unsigned)m.alpha_bits) : 0;
    int color_bits = m.color_bits, alpha_bits = m.alpha_bits;
    if (m.endpoint_p || m.shared_p) {
        unsigned p[6];
        if (m.endpoint_p)
            for (int k = 0; k < ends; ++k) p[k] = bc7_bits(b, &at, 1);
        else
            for (int s = 0; s < m.subsets; ++s) p[2 * s] = p[2 * s + 1] = bc7_bits(b, &at, 1);
        for (int k = 0; k < ends; ++k)
            for (int ch = 0; ch < 4; ++ch) e[k][ch] = (e[k][ch] << 1) | p[k];
        ++color_bits;
        if (alpha_bits) ++alpha_bits;
    }
    for (int k = 0; k < ends; ++k) {
        for (int ch = 0; ch < 3; ++ch) {
            const unsigned v = e[k][ch] << (8 - color_bits);
            e[k][ch] = v | (v >> color_bits);
        }
        if (alpha_bits) {
            const unsigned v = (e[k][3] << (8 - alpha_bits)) & 255u;
            e[k][3] = v | (v >> alpha_bits);
        } else {
            e[k][3] = 255;
        }
    }
    unsigned subset[16], first[16], second[16];
    for (unsigned t = 0; t < 16; ++t)
        subset[t] = m.subsets == 1 ? 0 : ((m.subsets == 2 ? kBc7Shape2 : kBc7Shape3)[partition] >> (2 * t)) & 3u;
    unsigned anchor[3] = {0, 0, 0};
    if (m.subsets == 2) anchor[1] = kBc7Anchor2[partition];
    if (m.subsets == 3) { anchor[1] = kBc7Anchor3a[partition]; anchor[2] = kBc7Anchor3b[partition]; }
    for (unsigned t = 0; t < 16; ++t) first[t] = bc7_bits(b, &at, (unsigned)m.index_bits - (anchor[subset[t]] == t));
    for (unsigned t = 0; t < 16; ++t) second[t] = m.index2_bits ? bc7_bits(b, &at, (unsigned)m.index2_bits - (t == 0)) : 0;
    for (unsigned t = 0; t < 16; ++t) {
        unsigned wc = bc7_weight((unsigned)m.index_bits, first[t]), wa = wc;
        if (m.index2_bits) {
            const unsigned w2 = bc7_weight((unsigned)m.index2_bits, second[t]);
            if (selector) wc = w2; else wa = w2;
        }
        const unsigned* e0 = e[2 * subset[t]];
        const unsigned* e1 = e[2 * subset[t] + 1];
        unsigned px[4];
        for (int ch = 0; ch < 3; ++ch) px[ch] = bc7_mix(e0[ch], e1[ch], wc);
        px[3] = bc7_mix(e0[3], e1[3], wa);
        if (rotation) { const unsigned s = px[3]; px[3] = px[rotation - 1]; px[rotation - 1] = s; }
        for (int ch = 0; ch < 4; ++ch) out[t][ch] = (uint8_t)px[ch];
    }
}

int orc_decode_bc(uint32_t format, uint32_t width, uint32_t height, uint32_t mip_count, const void* blocks, void* out_rgba8) {
    if (format < RDR_FORMAT_BC1 || format > RDR_FORMAT_BC7_SRGB || !width || !height || !mip_count) return 1;
    if (format >= RDR_FORMAT_BC7) { /* the sRGB code differs in how the texels are read later, not in the decode */
        const uint8_t* src = (const uint8_t*)blocks;
        uint8_t* dst = (uint8_t*)out_rgba8;
        for (uint32_t level = 0; level < mip_count; ++level) {
            const uint32_t w = (width >> level) ? (width >> level) : 1u, h = (height >> level) ? (height >> level) : 1u;
            const uint32_t across = (w + 3) / 4, down = (h + 3) / 4;
            for (uint32_t by = 0; by < down; ++by)
                for (uint32_t bx = 0; bx < across; ++bx) {
                    uint8_t px[16][4];
                    bc7_block(src + ((size_t)by * across + bx) * 16, px);
                    for (uint32_t t = 0; t < 16; ++t) {
                        const uint32_t x = bx * 4 + t % 4, y = by * 4 + t / 4;
                        if (x < w && y < h) memcpy(dst + ((size_t)y * w + x) * 4, px[t], 4);
                    }
                }
            src += (size_t)across * down * 16;
            dst += (size_t)w * h * 4;
        }
        return 0;
    }
    const size_t block_bytes = (format == RDR_FORMAT_BC1 || format == RDR_FORMAT_BC4) ? 8 : 16;
    const uint8_t* src = (const uint8_t*)blocks;
    uint8_t* dst = (uint8_t*)out_rgba8;
    for (uint32_t level = 0; level < mip_count; ++level) {
        const uint32_t w = (width >> level) ? (width >> level) : 1u, h = (height >> level) ? (height >> level) : 1u;
        const uint32_t across = (w + 3) / 4;
        for (uint32_t y = 0; y < h; ++y)
            for (uint32_t x = 0; x < w; ++x)
                decode_texel(format, src + ((size_t)(y / 4) * across + x / 4) * block_bytes, (y % 4) * 4 + x % 4, dst + ((size_t)y * w + x) * 4);
        src += (size_t)across * ((h + 3) / 4) * block_bytes;
        dst += (size_t)w * h * 4;
    }
    return 0;
}
