/*
 * rdr_oracle.h — entry points of the CPU oracle (liborc.so).
 *
 * TEST INFRASTRUCTURE ONLY: a CPU restatement of the reference's per-frame
 * path used to check the HIP kernels and as the timed CPU baseline.  The
 * shipped library never links, loads or calls it.
 */
#ifndef RDR_ORACLE_H
#define RDR_ORACLE_H

#include "../include/rdr_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

void orc_build_ubo(uint32_t width, uint32_t height, const float cam_pos[3], const float cam_dir[3],
                   const float sun_dir[3], uint32_t draw_count, uint32_t lod_enabled,
                   uint32_t cull_enabled, rdr_ubo* out);

/* Cull + transform + rasterise meshlet instances [part_lo, part_hi) into
 * vis[W*H] (cleared here).  draw_lod[draw_count] receives the LOD each draw
 * selected, 0xFF for draws that emitted nothing. */
int orc_render_visbuffer(const rdr_scene_desc* scene, const rdr_ubo* ubo, uint32_t part_lo,
                         uint32_t part_hi, uint64_t* vis, uint8_t* draw_lod, rdr_stats* stats);
/* The same with the interleaved share of rdr_set_partition_interleaved on top of the range. */
int orc_render_visbuffer_interleaved(const rdr_scene_desc* scene, const rdr_ubo* ubo, uint32_t part_lo,
                                     uint32_t part_hi, uint32_t chunk, uint32_t rank, uint32_t world,
                                     uint64_t* vis, uint8_t* draw_lod, rdr_stats* stats);

int orc_resolve(const rdr_scene_desc* scene, uint32_t width, uint32_t height, const uint64_t* vis,
                const uint8_t* draw_lod, uint32_t* out_xy);

/* Deferred shading + sRGB store of rows [row_lo, row_hi).  out_rgba8 is W*H*4
 * (row 0 = top); out_linear, when not NULL, receives W*H*4 floats before the
 * sRGB encode. */
int orc_shade(const rdr_scene_desc* scene, const rdr_ubo* ubo, const uint64_t* vis,
              const uint8_t* draw_lod, uint32_t row_lo, uint32_t row_hi, uint8_t* out_rgba8,
              float* out_linear);

/* Block-compressed mip chain (RDR_FORMAT_BC1..BC7_SRGB) -> tightly packed RGBA8 levels. */
int orc_decode_bc(uint32_t format, uint32_t width, uint32_t height, uint32_t mip_count, const void* blocks, void* out_rgba8);

/* Debug: the frame shaded with texture(lambertian, n) taken at the level of detail a 2 x 2-quad-difference
 * implementation would use instead of level 0 (see orc_shade.c).  *undefined counts covered pixels whose quad has an
 * uncovered pixel (no defined differences: they keep level 0). */
int orc_shade_quad_lod(const rdr_scene_desc* scene, const rdr_ubo* ubo, const uint64_t* vis, const uint8_t* draw_lod,
                       uint8_t* out_rgba8, float* out_lod, uint64_t* undefined);

/* Skybox pass: fills the pixels of rows [row_lo, row_hi) whose key is the clear
 * value from env.skybox (no-op without one).  Run after orc_shade. */
int orc_skybox(const rdr_scene_desc* scene, const rdr_ubo* ubo, const uint64_t* vis,
               uint32_t row_lo, uint32_t row_hi, uint8_t* out_rgba8);

/* IBL bake (bake/src/brdf.rs, brdf.comp.glsl, filter_envmap.frag.glsl).  The LUT is size x size
 * RGBA32F, row y = roughness 1 - (y + 0.5) / size, column x = n.v (x + 0.5) / size.  The
 * prefiltered map is mip-major, face-major RGBA32F, base `size`, `mips` levels, roughness =
 * level / (mips - 1); distribution 0 = Lambertian, 1 = GGX, 2 = Charlie. */
int orc_bake_brdf_lut(uint32_t size, uint32_t samples, float* out_rgba32f);
int orc_bake_env_map(const rdr_cubemap_desc* skybox, uint32_t distribution, uint32_t size, uint32_t mips,
                     uint32_t samples, float* out_rgba32f);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
