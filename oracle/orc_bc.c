/*
 * orc_bc.c — CPU decode of block-compressed texture levels (BC1, BC2, BC3, BC4 unorm) to RGBA8.
 *
 * TEST INFRASTRUCTURE ONLY (see rdr_oracle.h).  PARITY UNPINNED: the reference uploads the blocks of
 * a .dds file to a Vulkan BC image (renderer/src/renderer/image_loader.rs:63-135) and the sampler
 * hardware decodes them; the reference ships no compressed texture and no decoded image to pin
 * against, so this follows the block layouts of the Khronos data format specification (S3TC / RGTC
 * chapters) with palette entries rounded to nearest, and is checked against hand-computed blocks.
 */
#include <string.h>

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

int orc_decode_bc(uint32_t format, uint32_t width, uint32_t height, uint32_t mip_count, const void* blocks, void* out_rgba8) {
    if (format < RDR_FORMAT_BC1 || format > RDR_FORMAT_BC4 || !width || !height || !mip_count) return 1;
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
