/*
 * rdr_abi.h — C ABI of the MI355X visibility-buffer hot path.
 *
 * This is the drop-in boundary: the entry points below are what a foreign
 * function interface on the reference side binds instead of recording the
 * Vulkan passes through its `lava` backend.  Every struct that crosses the
 * boundary is byte-for-byte the reference's `#[repr(C)]` / std430 layout, so
 * a host that already holds the reference's `Scene` can pass pointers to its
 * vectors without conversion.  Citations are relative to the reference tree.
 *
 * Conventions (reference: anyhow::Result everywhere, nothing panics on the
 * frame path — renderer/src/renderer.rs:862-1002):
 *   - every fallible call returns int, 0 = success, non-zero = failure, and
 *     rdr_last_error() returns a thread-local message for the last failure;
 *   - the caller keeps ownership of every host pointer; rdr_set_scene copies
 *     everything to HBM and returns after the copy is complete
 *     (lava/src/device.rs:575-583);
 *   - one renderer = one GPU = one caller thread (lava's LDevice is
 *     !Send + !Sync, lava/src/device.rs:87-96).
 */
#ifndef RDR_ABI_H
#define RDR_ABI_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ */
/* Scene data: layouts of scenery/src/{vertex,mesh}.rs and            */
/* assets/shaders/mesh.inc.glsl                                        */
/* ------------------------------------------------------------------ */

/* scenery/src/vertex.rs:4-11, mesh.inc.glsl:8-14 — 64 B */
typedef struct rdr_vertex {
    float p[4];  /* position.xyz, w unused            */
    float n[4];  /* normal.xyz, w unused              */
    float uv[4]; /* uv0.xy | uv1.zw                   */
    float t[4];  /* tangent.xyz, w = bitangent sign   */
} rdr_vertex;

/* scenery/src/mesh.rs:17-28, mesh.inc.glsl:16-28 — 32 B */
typedef struct rdr_meshlet {
    float center[3];
    float radius;
    int8_t cone_axis[3];
    int8_t cone_cutoff;
    uint32_t vertex_offset;   /* into meshlet_vertices              */
    uint32_t triangle_offset; /* into meshlet_triangles (bytes)     */
    uint8_t triangle_count;   /* <= 124                             */
    uint8_t vertex_count;     /* <= 64                              */
    uint8_t _pad[2];
} rdr_meshlet;

/* scenery/src/mesh.rs:30-37, mesh.inc.glsl:30-35 — 16 B */
typedef struct rdr_mesh_lod {
    uint32_t meshlet_offset;
    uint32_t meshlet_count;
    float error;
    uint32_t _pad;
} rdr_mesh_lod;

/* scenery/src/mesh.rs:39-49, mesh.inc.glsl:37-45 — 160 B */
typedef struct rdr_mesh {
    float center[3];
    float radius;
    uint32_t vertex_offset;
    uint32_t vertex_count;
    uint32_t lod_count;
    uint32_t index_offset; /* LOD0 triangle list start in `indices` */
    rdr_mesh_lod lods[8];
} rdr_mesh;

/* scenery/src/mesh.rs:61-70, mesh.inc.glsl:47-53 — 16 B */
typedef struct rdr_mesh_draw {
    uint32_t alpha_mode; /* 0 = opaque pass, 1 = alpha-mask pass (mesh.rs:11-15) */
    uint32_t meshlet_visibility_offset;
    uint32_t model_index;
    uint16_t mesh_index;
    uint16_t material_index; /* 0xFFFF = primitive without material (scene.rs:649) */
} rdr_mesh_draw;

/* scenery/src/mesh.rs:83-123, mesh.inc.glsl:55-92 — 272 B */
typedef struct rdr_material {
    float base_color_factor[4];
    float metallic_roughness_normal_occlusion[4];
    float emissive_factor_alpha_cutoff[4];
    float specular_glossiness_factor[4];
    float clearcoat_factor[4];
    float sheen_color_roughness_factor[4];
    float albedo_texture_transform[4];
    float normal_texture_transform[4];
    float metallic_roughness_texture_transform[4];
    float emissive_texture_transform[4];
    float occlusion_texture_transform[4];
    float specular_factor[4];
    uint32_t albedo_texture, albedo_texture_uv;
    uint32_t normal_texture, normal_texture_uv;
    uint32_t metallic_roughness_texture, metallic_roughness_texture_uv;
    uint32_t emissive_texture, emissive_texture_uv;
    uint32_t occlusion_texture, occlusion_texture_uv;
    uint32_t clearcoat_texture, clearcoat_texture_uv;
    uint32_t specular_color_texture, specular_color_texture_uv;
    uint32_t specular_texture, specular_texture_uv;
    float ior;
    uint32_t alpha_mode;
    uint32_t material_type; /* RDR_MAT_* bits */
    uint32_t _pad;
} rdr_material;

/* scenery/src/mesh.rs:72-81, mesh.inc.glsl:129-133 */
#define RDR_MAT_METAL_ROUGHNESS 1u
#define RDR_MAT_SPEC_GLOSS (1u << 1)
#define RDR_MAT_CLEARCOAT (1u << 2)
#define RDR_MAT_SHEEN (1u << 3)
#define RDR_MAT_SPEC (1u << 4)

#define RDR_NO_TEXTURE 0xFFFFFFFFu

/* renderer/src/renderer.rs:56-76, mesh.inc.glsl:104-122 — 288 B.
 * Matrices are column-major (glam Mat4 == GLSL mat4): m[col*4 + row]. */
typedef struct rdr_ubo {
    float view_proj[16];
    float view[16];
    float proj[16];
    float camera_position[3];
    uint32_t _pad0;
    float near_plane;
    float far_plane;
    float depth_pyramid_width;
    float depth_pyramid_height;
    float frustum[4];
    uint32_t draw_count;
    uint32_t lod_enabled;
    uint32_t cull_enabled;
    float lod_target;
    float screen_width;
    float screen_height;
    uint32_t _pad1[2];
    float sun_direction[4];
} rdr_ubo;

/* ------------------------------------------------------------------ */
/* Textures.  The reference loads files (renderer/src/renderer/        */
/* image_loader.rs) into Vulkan images; the boundary takes the decoded */
/* texels instead, with the colour-space choice the loader made        */
/* (image_loader.rs:47-51,164-168).                                    */
/* ------------------------------------------------------------------ */
#define RDR_FORMAT_RGBA8_UNORM 0u
#define RDR_FORMAT_RGBA8_SRGB 1u
#define RDR_FORMAT_RGBA32F 2u
/* Block-compressed 2D images, as a .dds file stores them (image_loader.rs:63-135 hands the blocks
 * to Vulkan; the sRGB variants of BC1-BC3 are bound as UNORM there, :72-74, so those have one
 * code per block layout, while BC7 keeps its sRGB variant, :81-82).  Decoded to RGBA8 on the GPU
 * when the scene is uploaded: RGBA8_SRGB for RDR_FORMAT_BC7_SRGB, RGBA8_UNORM for the others.
 * Level i holds ceil(w_i / 4) x ceil(h_i / 4) blocks of 8 (BC1, BC4) or 16 bytes, row-major. */
#define RDR_FORMAT_BC1 3u /* RGB565 end points, 1-bit punch-through alpha */
#define RDR_FORMAT_BC2 4u /* explicit 4-bit alpha + BC1 colours */
#define RDR_FORMAT_BC3 5u /* interpolated alpha + BC1 colours */
#define RDR_FORMAT_BC4 6u /* one interpolated channel (red), unorm */
#define RDR_FORMAT_BC7 7u /* eight block modes, up to three end-point pairs, RGBA */
#define RDR_FORMAT_BC7_SRGB 8u /* the same blocks; red, green and blue hold sRGB-encoded values */

/* 2D image: mip levels tightly packed, level 0 first, rows top to bottom;
 * level i is max(1,width>>i) x max(1,height>>i). */
typedef struct rdr_image_desc {
    const void* data;
    uint32_t width, height;
    uint32_t mip_count;
    uint32_t format;
} rdr_image_desc;

/* Decode a block-compressed mip chain (RDR_FORMAT_BC1..BC7_SRGB) to tightly packed RGBA8 levels, host
 * memory to host memory, with the kernel rdr_set_scene uses at upload.  What the sampler hardware
 * does per fetch behind vkCmdCopyBufferToImage of a BC image (image_loader.rs:96-121). */
int rdr_decode_bc(int device, uint32_t format, uint32_t width, uint32_t height, uint32_t mip_count, const void* blocks,
                  size_t block_bytes, void* out_rgba8);

/* rdr_decode_bc for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments
 * and bit-identical output; the two staging buffers rdr_decode_bc allocates per call are replaced by
 * a fixed array inside the library's code object (7.5 MiB per GPU, held until the process ends), and
 * the call makes no hipMalloc / hipFree.  Limit: the chain's blocks must fit 1.5 MiB and its decoded
 * levels (each padded to 16 bytes) 6 MiB, i.e. up to a full 1024 x 1024 mip chain in any of the block
 * formats; a larger chain is refused with an error and nothing is written.  All calls on one GPU
 * share the array: calls that overlap on one GPU race, the caller serialises them. */
int rdr_decode_bc_noalloc(int device, uint32_t format, uint32_t width, uint32_t height, uint32_t mip_count,
                          const void* blocks, size_t block_bytes, void* out_rgba8);

/* Cubemap: mip-major; within a mip the six faces in Vulkan layer order
 * +X,-X,+Y,-Y,+Z,-Z; level i is max(1,size>>i) square. */
typedef struct rdr_cubemap_desc {
    const void* data;
    uint32_t size;
    uint32_t mip_count;
    uint32_t format;
    uint32_t _pad;
} rdr_cubemap_desc;

/* scenery/src/scene.rs:27-33 (EnvMap), bound at renderer/src/renderer/pass.rs:539-558 */
typedef struct rdr_env_desc {
    rdr_image_desc brdf_lut;
    rdr_cubemap_desc skybox; /* background, sampled when rdr_config.skybox = 1 */
    rdr_cubemap_desc lambertian;
    rdr_cubemap_desc ggx;
    rdr_cubemap_desc charlie;
} rdr_env_desc;

/* The arrays SceneResources::load_scene uploads (renderer/src/renderer.rs:426-586). */
typedef struct rdr_scene_desc {
    const rdr_vertex* vertices;
    const uint32_t* indices; /* all meshes' LOD0 triangle lists, mesh-local vertex ids */
    const uint32_t* meshlet_vertices;
    const uint8_t* meshlet_triangles;
    const rdr_meshlet* meshlets;
    const rdr_mesh* meshes;
    const rdr_mesh_draw* mesh_draws;
    const rdr_material* materials;
    const float* global_transforms; /* transform_count column-major mat4 */
    const rdr_image_desc* images;   /* images[i] is texture index i of the materials */
    const rdr_env_desc* env;        /* NULL: black environment, zero BRDF LUT */
    uint32_t vertex_count;
    uint32_t index_count;
    uint32_t meshlet_vertex_count;
    uint32_t meshlet_triangle_bytes;
    uint32_t meshlet_count;
    uint32_t mesh_count;
    uint32_t draw_count;
    uint32_t material_count;
    uint32_t transform_count;
    uint32_t image_count;
    uint32_t meshlet_visibility_count; /* scenery/src/scene.rs:648-651 */
    uint32_t _pad;
} rdr_scene_desc;

/* scenery/src/lib.rs:7-14 (RendererConfig); only the two fields that reach
 * the GPU (renderer/src/renderer.rs:1182-1183) plus this path's own switches. */
typedef struct rdr_config {
    uint32_t lod_enabled;
    uint32_t cull_enabled;   /* 1: two-phase occlusion culling (early phase, depth pyramid, late phase;
                              * cull.comp.glsl:60-95, shader.task.glsl:96-128); never changes the image */
    uint32_t collect_stats;  /* 1: count triangles/fragments per frame (small cost) */
    uint32_t time_kernels;   /* n >= 1: bracket the kernels of every n-th frame with events for rdr_get_stats
                              * (the brackets cost about 33 us per bracketed frame at 4K) */
    uint32_t skybox;         /* 1: pixels with no geometry show env.skybox (SkyBoxPass,
                              * renderer/src/renderer/pass.rs:397-445); 0: they stay (0,0,0,0) */
} rdr_config;

/* Packed visibility-buffer key written by the rasteriser with a 64-bit
 * atomic min:
 *   bits 62..32  0x7FFFFFFF - float_bits(depth)   depth = reverse-Z z/w in [0,1]
 *   bits 31..7   meshlet instance = draw.meshlet_visibility_offset + (meshlet - lod.meshlet_offset)
 *   bits  6..0   triangle index inside the meshlet (<= 123)
 * Bit 63 is never set, so signed and unsigned 64-bit min agree on every key. */
#define RDR_VIS_CLEAR 0x7FFFFFFFFFFFFFFFull

#define RDR_STAT_KERNELS 8
enum {
    RDR_K_CLEAR = 0,
    RDR_K_DRAW_CULL = 1,
    RDR_K_MESHLET_CULL = 2,
    RDR_K_RASTER = 3,
    RDR_K_SHADOW = 4,
    RDR_K_BIG_RASTER = 5,
    RDR_K_SHADE = 6,
    RDR_K_CLASSIFY = 7
};

typedef struct rdr_stats {
    float kernel_ms[RDR_STAT_KERNELS]; /* mean per frame since the previous rdr_get_stats (time_kernels = 1) */
    uint64_t draws_visible;
    uint64_t commands;            /* 64-meshlet cull commands emitted */
    uint64_t meshlets_tested;
    uint64_t meshlets_visible;
    uint64_t triangles_submitted; /* sum over visible draws of the selected LOD's triangles */
    uint64_t triangles_tested;    /* triangles of visible meshlets */
    uint64_t triangles_rasterised;/* survived backface + small-triangle tests */
    uint64_t triangles_big;       /* routed to the clip / tile path */
    uint64_t fragments;           /* 64-bit atomics issued to memory */
    uint64_t pixels_covered;      /* filled by rdr_shade */
    uint64_t vertices_transformed;/* vertices of visible meshlets */
    uint64_t frames_timed;        /* frames kernel_ms is averaged over (time_kernels = 1) */
    uint64_t bin_entries;         /* tile path: (triangle, 64x64 tile) pairs binned in the last frame */
    uint64_t bin_max;             /* tile path: entries of the fullest tile */
} rdr_stats;

typedef struct rdr_renderer rdr_renderer;

/* Replaces Renderer::new (renderer/src/renderer.rs:793).  hip_devices[0] is
 * the GPU this renderer owns; n_devices must be 1 (one process per GPU). */
int rdr_create(const rdr_config* cfg, uint32_t width, uint32_t height,
               const int* hip_devices, int n_devices, rdr_renderer** out);

/* Replaces Renderer::set_scene (renderer.rs:1004) / SceneResources::load_scene
 * (renderer.rs:426-698): validates, uploads, builds the shadow-ray BVH. */
int rdr_set_scene(rdr_renderer* r, const rdr_scene_desc* scene);

/* Replaces SceneResources::upload_transforms (renderer.rs:700-722): the whole array. */
int rdr_update_transforms(rdr_renderer* r, const float* mat4s, uint32_t count);

/* Replaces Renderer::render (renderer.rs:862): UBO update (renderer.rs:1148-1197)
 * then cull -> raster -> shade -> sRGB store, in the order of renderer.rs:803-843. */
int rdr_render(rdr_renderer* r, const float cam_pos[3], const float cam_dir[3],
               const float sun_dir[3]);

/* Work a frame had to drop because a list filled up.  The reference cannot drop work (its lists are
 * sized by the driver); here every list has a capacity fixed at rdr_set_scene, and a frame that
 * exceeds one is reported, never silently wrong: the kernels leave these flags in pinned host
 * memory, and the next rdr_render / rdr_render_visbuffer / rdr_shade on the renderer looks at
 * them first (no synchronisation) and fails with a message if the last finished frame dropped work,
 * as a Vulkan call of the reference's render() would fail through `?` (renderer.rs:862-1002);
 * readbacks and rdr_get_stats fail the same way.  rdr_frame_status waits for the submitted work and
 * returns the flags of the last frame in *flags (0 = complete) without raising an error. */
#define RDR_FRAME_DEFERRED_OVERFLOW 1u /* clip / tile path: deferred triangle list */
#define RDR_FRAME_SETUP_OVERFLOW 2u    /* tile path: screen-space records */
#define RDR_FRAME_BIN_OVERFLOW 4u      /* tile path: per-tile bins */
#define RDR_FRAME_MID_OVERFLOW 8u      /* mid-sized triangle queue */
#define RDR_FRAME_FRAG_OVERFLOW 16u    /* alpha-tested fragment queue */
#define RDR_FRAME_LIST_OVERFLOW 32u    /* cull command / visible meshlet lists */
#define RDR_FRAME_SHADOW_STACK 64u     /* a shadow ray's traversal stack overflowed without a hit: its pixel may be lit wrongly */
#define RDR_FRAME_PEER_TIMEOUT 128u    /* sort-last exchange: a peer rank did not arrive within 5 s (rdr_composite_*) */
int rdr_frame_status(rdr_renderer* r, uint32_t* flags);

/* The two halves of rdr_render, for sort-last rendering: geometry into the
 * 64-bit visibility buffer, (composite by the caller), then shading. */
int rdr_render_visbuffer(rdr_renderer* r, const float cam_pos[3], const float cam_dir[3],
                         const float sun_dir[3]);
int rdr_shade(rdr_renderer* r);

/* Sort-last share of this renderer: meshlet instances [lo, hi) of the scene's
 * meshlet_visibility_count.  Default is everything. */
int rdr_set_partition(rdr_renderer* r, uint32_t instance_lo, uint32_t instance_hi);

/* Shading restricted to rows [row_lo, row_hi) (after a reduce-scatter). Default all. */
int rdr_set_shade_rows(rdr_renderer* r, uint32_t row_lo, uint32_t row_hi);

/* The balanced forms of the two calls above (no counterpart in the reference, which is single-GPU;
 * the id order they deal out is the draw order of scenery/src/scene.rs:648-651, which for an
 * instanced scene is a spatial order, so contiguous N-ths give one rank the near instances and
 * another the far ones).
 * rdr_set_partition_interleaved: the meshlets of every draw are cut into chunks of `chunk`
 * consecutive meshlets (of the level of detail the draw selected; chunk = a multiple of 64, at most
 * 2^20), and chunk j of draw d belongs to rank (j + d) % world.  Every meshlet instance belongs to
 * exactly one rank, so the 64-bit minimum over the ranks' buffers is the single-GPU buffer.  The
 * range of rdr_set_partition still applies on top (default: everything); world = 1 switches the
 * interleaving off.
 * rdr_set_shade_bands: the frame's rows are cut into bands of `band` rows (a power of two, at
 * least 32: the exchange works on 32 x 32 tiles), and band b belongs to rank b % world.  Replaces
 * the row range of rdr_set_shade_rows, and rdr_set_shade_rows replaces it. */
int rdr_set_partition_interleaved(rdr_renderer* r, uint32_t chunk, uint32_t rank, uint32_t world);
int rdr_set_shade_bands(rdr_renderer* r, uint32_t band, uint32_t rank, uint32_t world);
/* A few screen-filling triangles (a floor, a wall) are more pixel work than a rank's whole share, and they sit in
 * one meshlet.  With a threshold > 0 (and an interleaved share), a draw whose level-0 triangles are estimated to
 * cover more than `min_triangle_pixels` each on average this frame (object-space mean area x scale^2 x the
 * projection at the nearest point of its bounding sphere: a function of scene and view, the same on every rank)
 * is taken by EVERY rank, which rasterises its large triangles only on the rows it shades (rdr_set_shade_bands /
 * rdr_set_shade_rows).  A rank's buffer is then complete on its own rows, which is all the exchange reads of it,
 * and the minimum does not care that small triangles of such a draw arrive from several ranks.  0 = off
 * (default).  Requires the ranks' shade rows to partition the frame. */
int rdr_set_partition_replication(rdr_renderer* r, float min_triangle_pixels);

/* Device addresses of the W*H uint64 visibility buffer and the W*H*4 RGBA8
 * image, for the collective that composites them (RCCL through the caller). */
void* rdr_visbuffer_device_ptr(rdr_renderer* r);
void* rdr_color_device_ptr(rdr_renderer* r);

/* ------------------------------------------------------------------ */
/* Sort-last exchange, peer to peer over xGMI (no counterpart in the    */
/* reference, which is single-GPU).  One process per GPU; every rank     */
/* holds the whole scene, rasterises its rdr_set_partition share into a  */
/* full-frame visibility buffer and shades the rows of                    */
/* rdr_set_shade_rows.  Instead of a ring collective, each rank reads    */
/* the keys of its own rows straight out of every peer's buffer (shared  */
/* through HIP IPC handles) and takes the 64-bit minimum, which is       */
/* bit-identical to a uint64 MIN all-reduce restricted to those rows;    */
/* the shaded RGBA8 rows are then written straight into every peer's     */
/* image.  Per frame on every rank, in this order and on the renderer's  */
/* stream, without host synchronisation in between:                       */
/*   rdr_render_visbuffer, rdr_composite_exchange, rdr_shade,            */
/*   rdr_composite_gather.                                                */
/* Ranks are ordered against each other by phase counters in device      */
/* memory; a rank whose peer does not arrive within 5 s gets an error    */
/* from its next call instead of a hang.  Rows must be a partition of    */
/* the image among the ranks, and every rank has the same width and      */
/* height.  Both directions move 32x32-pixel tiles and leave out those   */
/* that cannot matter: keys come only from ranks whose rasterisers       */
/* touched the tile (a flag per tile, kept while connected), colour      */
/* tiles go out unless clear now and the frame before (with a skybox     */
/* every tile goes).  So: render the visibility buffer after connecting, */
/* and while connected let only the gathers write image rows outside     */
/* the own strip.                                                         */
/* ------------------------------------------------------------------ */
#define RDR_COMPOSITE_MAX_RANKS 16
typedef struct rdr_composite_handle { /* three hipIpcMemHandle_t: keys, image, control block */
    uint8_t vis[64];
    uint8_t color[64];
    uint8_t ctrl[64];
    uint8_t info[16]; /* bytes 0..3: the exporting renderer's exchange counter (little endian); the rest zero */
} rdr_composite_handle;
/* This renderer's handles, to be sent to every other rank (any transport: a file, MPI, a socket). */
int rdr_composite_export(rdr_renderer* r, rdr_composite_handle* out);
/* handles[world]: every rank's export in rank order.  Collective in the sense that every rank must connect
 * before any rank's first exchange completes.  The ranks' exchange counters are aligned here (every rank
 * continues from the largest counter among the handles), so renderers that ran different numbers of frames
 * before (one was recreated, one failed a frame) meet in the same phase again.  After a failure while
 * connected (a peer that did not arrive, a frame of this rank that dropped work) the connection is broken:
 * exchange and gather fail until rdr_composite_disconnect, a new export and a new connect on every rank. */
int rdr_composite_connect(rdr_renderer* r, uint32_t rank, uint32_t world, const rdr_composite_handle* handles);
int rdr_composite_exchange(rdr_renderer* r);
int rdr_composite_gather(rdr_renderer* r);
/* Waits for this renderer's work, then closes the peers' buffers.  Call on every rank before rdr_destroy. */
int rdr_composite_disconnect(rdr_renderer* r);

/* Run on the caller's HIP stream (hipStream_t) instead of the library's own.  NULL goes back to the
 * library's own (non-blocking) stream; to select HIP's null stream pass hipStreamLegacy, HIP's handle
 * for it ((hipStream_t)1), which like hipStreamPerThread is handed to the runtime unchanged. */
int rdr_set_stream(rdr_renderer* r, void* hip_stream);
int rdr_sync(rdr_renderer* r);

/* The reference never reads pixels back; these are the readback entry points
 * the replacement adds.  out_xy: W*H*2 uint32 in the reference's texel format
 * (x = meshlet_index << 8 | triangle, y = draw_id; 0xFFFFFFFF twice = no
 * geometry; assets/shaders/shader.frag.glsl:71-74, pass.rs:250). */
int rdr_read_visbuffer(rdr_renderer* r, uint32_t* out_xy);
int rdr_read_visbuffer_u64(rdr_renderer* r, uint64_t* out);
/* W*H*4 bytes, row 0 = top, sRGB-encoded as an *_SRGB swapchain image stores
 * it (lava/src/swapchain.rs:100-115). */
int rdr_read_color(rdr_renderer* r, uint8_t* out_rgba8_srgb);
int rdr_get_stats(rdr_renderer* r, rdr_stats* out);

/* Renderer::update_uniform_object (renderer.rs:1148-1197) on its own. */
int rdr_build_ubo(uint32_t width, uint32_t height, const float cam_pos[3],
                  const float cam_dir[3], const float sun_dir[3], uint32_t draw_count,
                  uint32_t lod_enabled, uint32_t cull_enabled, rdr_ubo* out);
int rdr_get_ubo(rdr_renderer* r, rdr_ubo* out);

void rdr_destroy(rdr_renderer* r);
const char* rdr_last_error(void);

/* ------------------------------------------------------------------ */
/* IBL bake: the inputs of rdr_env_desc, produced on the GPU (the      */
/* reference's offline `bake` tool: bake/src/brdf.rs:12-80 +           */
/* assets/shaders/brdf.comp.glsl, bake/src/brdf.rs:99-338 +            */
/* assets/shaders/filter_envmap.frag.glsl).  Stand-alone: no renderer. */
/* ------------------------------------------------------------------ */
/* size x size RGBA32F; column x = n.v (x + 0.5) / size, row y = roughness 1 - (y + 0.5) / size;
 * r, g = GGX split sum, b = Charlie, a = 0.  The reference bakes 256 x 256 with 10 000 samples. */
int rdr_bake_brdf_lut(int hip_device, uint32_t size, uint32_t samples, float* out_rgba32f);
/* Prefilter `skybox` into a cubemap of base `size` with `mips` levels (roughness = level /
 * (mips - 1)), mip-major, face-major RGBA32F.  distribution: 0 Lambertian, 1 GGX, 2 Charlie.
 * The reference bakes 256, log2(256) = 8 levels, 1024 samples. */
int rdr_bake_env_map(int hip_device, const rdr_cubemap_desc* skybox, uint32_t distribution, uint32_t size,
                     uint32_t mips, uint32_t samples, float* out_rgba32f);

/* ------------------------------------------------------------------ */
/* Mesh preparation on the host: Geometry::append_mesh                  */
/* (scenery/src/mesh.rs:197-374, which calls the meshoptimizer library): */
/* vertex de-duplication, vertex cache / fetch ordering, meshlets of at  */
/* most 64 vertices / 124 triangles (cone weight 0.25) with bounds, and  */
/* the LOD chain (each level simplified from level 0 to 0.65^k of its    */
/* indices, relative error limit 0.1, stop at >= 95 % or empty; error =  */
/* max(level error, previous level's) x mesh extent).  Same rules and    */
/* limits as the reference; the algorithms are this library's own, so    */
/* meshlet contents are not identical to meshoptimizer's.  fast_mode is   */
/* RendererConfig::fast_mode (scenery/src/lib.rs:11): index order for a   */
/* 16-entry FIFO vertex cache instead of the score-driven one.  No GPU.   */
/* ------------------------------------------------------------------ */
typedef struct rdr_mesh_builder rdr_mesh_builder;
typedef struct rdr_built_mesh { /* views into the builder, valid until rdr_mesh_free */
    const rdr_vertex* vertices;        /* de-duplicated, in fetch order */
    const uint32_t* indices;           /* level 0 triangle list over `vertices` */
    const uint32_t* meshlet_vertices;  /* all levels */
    const uint8_t* meshlet_triangles;  /* all levels; each meshlet padded to 4 bytes */
    const rdr_meshlet* meshlets;       /* offsets relative to the two arrays above */
    uint32_t vertex_count, index_count, meshlet_vertex_count, meshlet_triangle_bytes, meshlet_count, _pad;
    rdr_mesh mesh;                     /* offsets relative to these arrays (vertex_offset = index_offset = 0) */
} rdr_built_mesh;
int rdr_mesh_build(const rdr_vertex* vertices, uint32_t vertex_count, const uint32_t* indices, uint32_t index_count,
                   uint32_t max_lods, uint32_t fast_mode, rdr_mesh_builder** out);
int rdr_mesh_result(const rdr_mesh_builder* b, rdr_built_mesh* out);
void rdr_mesh_free(rdr_mesh_builder* b);

/* Replaces scenery::vertex::generate_tangents (scenery/src/vertex.rs:86-90, called for glTF primitives without a
 * TANGENT attribute at scenery/src/scene.rs:207), i.e. the mikktspace crate: tangents from the uv parametrisation,
 * angle-weighted over the triangles that share a vertex, an edge and a uv orientation, w = handedness.  Written into
 * vertices[indices[k]].t corner by corner in triangle order (a record shared by corners of different groups keeps
 * the last one, as in the reference).  Host code, no GPU. */
int rdr_generate_tangents(rdr_vertex* vertices, uint32_t vertex_count, const uint32_t* indices, uint32_t index_count);

#ifdef __cplusplus
}
#endif
#endif /* RDR_ABI_H */
