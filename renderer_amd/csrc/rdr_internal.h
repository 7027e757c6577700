// rdr_internal.h — shared declarations of librdr_hip.so (host side).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/rdr_abi.h"

// ---- error plumbing ---------------------------------------------------------
void rdr_set_error(const std::string& msg);
#define RDR_HIP(call)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            rdr_set_error(std::string(#call) + " failed: " + hipGetErrorString(e_) + " at " + \
                          __FILE__ + ":" + std::to_string(__LINE__));                        \
            return 1;                                                                        \
        }                                                                                    \
    } while (0)
#define RDR_CHECK(cond, msg)            \
    do {                                \
        if (!(cond)) {                  \
            rdr_set_error(msg);         \
            return 1;                   \
        }                               \
    } while (0)

// ---- device-side records ----------------------------------------------------

// One unit of per-meshlet culling work: up to 64 consecutive meshlets of one draw.
struct CullCommand {
    uint32_t draw_id;
    uint32_t meshlet_first;  // global meshlet index of lane 0
    uint32_t count;          // 1..64
    uint32_t instance_first; // meshlet-instance id of lane 0
    float max_scale;         // largest column length of the draw's model matrix
    uint32_t flags;          // kCmd* below
    // what the meshlet cull needs of the draw and its mesh, so that a command leads straight to the model matrix and
    // the meshlets (two dependent loads fewer in front of every wave's tests)
    uint32_t model_index;
    uint32_t mesh_vertex_offset;
};
// kCmdLateVisible: occlusion culling, late phase: the draw was visible last frame (so its meshlets with their bit
// set were already drawn by the early phase); kCmdReplicated: sort-last, every rank draws this draw
// (FrameView::replicate_pixels); kCmdMasked: the draw is alpha-tested.
constexpr uint32_t kCmdLateVisible = 1u, kCmdReplicated = 2u, kCmdMasked = 4u;
static_assert(sizeof(CullCommand) == 32, "CullCommand must be 32 bytes");

// A meshlet that survived culling, input of the rasteriser.
// Carries what the rasteriser needs to start fetching vertices at once (no draw ->
// mesh -> meshlet pointer chase per meshlet).
struct VisibleMeshlet {
    uint32_t draw_id;
    uint32_t meshlet_index;
    uint32_t instance;
    uint32_t model_index;        // of the draw
    uint32_t vertex_offset;      // of the meshlet, into meshlet_vertices
    uint32_t triangle_offset;    // of the meshlet, into meshlet_triangles
    uint32_t counts;             // vertex_count | triangle_count << 16 | bit 31: of a draw every sort-last rank draws
    uint32_t mesh_vertex_offset; // of the mesh, added to every meshlet vertex index
};
static_assert(sizeof(VisibleMeshlet) == 32, "VisibleMeshlet layout");

// A triangle the per-meshlet rasteriser hands to the clip / tile path: its three
// clip-space vertices (position + both uv sets) and its identity.
struct DeferredTriangle {
    float pos[3][4];
    float uv[3][4];
    uint32_t id;      // (instance << 7) | triangle
    uint32_t draw_id; // for the alpha test
    uint32_t replicated; // of a draw every sort-last rank draws: rasterised on this rank's rows only
    uint32_t _pad;
};
static_assert(sizeof(DeferredTriangle) == 112, "DeferredTriangle layout");

// A screen-space triangle ready for the tile rasteriser: snapped vertices already
// oriented so that the interior has positive edge values.
struct SetupTri {
    int32_t X0, Y0, X1, Y1, X2, Y2; // 24.8 fixed point, y down
    float z0, dz1, dz2, inv_area;
    uint32_t id;
    uint32_t draw_id;
    uint16_t px0, py0, px1, py1; // covered pixel box, clamped to the screen
    uint32_t masked;             // bit 0: SetupMask holds uv (alpha-tested draw with a texture); bit 1: alpha-tested draw;
                                 // bit 2: of a draw every sort-last rank draws (only this rank's rows are rasterised)
    uint32_t _pad;
};
static_assert(sizeof(SetupTri) == 64, "SetupTri layout");
struct SetupMask {
    float inv_w[3];
    float uv[3][4];
    uint32_t _pad;
};
static_assert(sizeof(SetupMask) == 64, "SetupMask layout");

// A small alpha-tested triangle queued for k_alpha_frag (already oriented for scanning), 96 B.
struct FragTri {
    int32_t X[3], Y[3];
    float z0, dz1, dz2, inv_area;
    float inv_w[3], u[3], v[3]; // uv of the set the material's albedo texture uses
    uint32_t origin;            // px0 | py0 << 16 of the pixel box
    uint32_t bw;                // width of the pixel box (box area <= 16)
    uint32_t id;
    uint32_t draw_id;
    uint32_t _pad;
};
static_assert(sizeof(FragTri) == 96, "FragTri layout");

// Heads of the per-frame work lists.
struct FrameCounters {
    uint32_t cmd_count;
    uint32_t vis_count;
    uint32_t deferred_count;
    uint32_t deferred_overflow;
    uint32_t pixel_count;     // covered pixels in the shaded rows (length of pixel_list)
    uint32_t shadow_overflow; // rays that overflowed k_shadow's LDS stack, left to k_shadow_walk (must follow pixel_count)
    uint32_t setup_count;     // SetupTri records written this frame
    uint32_t setup_overflow;
    uint32_t bin_total;       // tile-bin entries needed this frame
    uint32_t bin_overflow;
    uint32_t vis_mask_count;  // alpha-masked meshlets, stored from the back of `visible`
    uint32_t tile_nonempty;   // jobs of the tile rasteriser this pass (length of tile_list)
    uint32_t mid_overflow;
    uint32_t deep_overflow;   // shadow rays whose deep (scratch) stack dropped entries without finding a hit
    // Frame-wide (the reset between the phases leaves them alone): the RDR_FRAME_* flags of every geometry pass of the frame so far
    // (with two-phase culling the early pass's list heads are gone by the time the late pass reports), and whether
    // any pass had work for the tile path or the mid-sized queue (the host's hint for the next frame's launches).
    uint32_t sticky_flags;
    uint32_t tile_work;
    uint32_t rare_barrier;    // k_tile_path_rare: arrivals at its grid-wide barriers
    uint32_t rare_timeout;    // k_tile_path_rare: a barrier gave up waiting (reported as RDR_FRAME_DEFERRED_OVERFLOW)
    unsigned long long _pad1[15];
    // SetupTri records queued for k_mid_raster, per segment of the queue.  Every wave of k_raster
    // reserves its slots with an atomic that returns a value, and same-address atomics serialise
    // (about 10 ns each): the queue is cut into segments with counters 128 bytes apart (word
    // s * 32 counts segment s) and a workgroup always uses segment blockIdx % kMidSegments.
    uint32_t mid_seg[32 * 32];
    // Small alpha-tested triangles and their covered pixels, queued by k_raster<true> for k_alpha_frag:
    // segmented like the mid queue; one 64-bit counter per segment (128 bytes apart) so that a wave
    // reserves its triangle slots and its pixel slots with a single atomic.
    unsigned long long frag_seg[32 * 32]; // word s * 16: covered pixels (low half) and triangles (high half) of segment s
    uint32_t frag_overflow;
    uint32_t _pad_frag[31];
    // k_raster's meshlet tickets (opaque list, alpha-tested list): a wave takes its first two meshlets by its place in
    // the grid and draws a ticket for every further one.  The visible list is in scene order, so the meshlets a fixed
    // stride hands a wave are a regular sample of the scene (every so-manyth instance of a row, say) and the waves'
    // shares differ by a factor of two and more; with tickets a wave that got light meshlets simply takes more.  One
    // counter would serialise (same-address atomics, about 10 ns each): ticket t of segment s is meshlet
    // 2 * grid + t * segments + s, a wave uses segment blockIdx % segments, counters 128 bytes apart as above.
#ifndef RDR_RASTER_SEGMENTS
#define RDR_RASTER_SEGMENTS 128
#endif
    uint32_t raster_ticket[2][RDR_RASTER_SEGMENTS * 32];
};
constexpr uint32_t kTileJob = 8; // bin records per job of the tile rasteriser
constexpr uint32_t kMidSegments = 32;
constexpr uint32_t kFragSegments = 32;
constexpr uint32_t kRasterSegments = RDR_RASTER_SEGMENTS;
static_assert(sizeof(FrameCounters) == 192 + 4096 + 8192 + 128 + 2 * RDR_RASTER_SEGMENTS * 128, "FrameCounters layout");

struct GpuImage {
    uint32_t width, height, mip_count, format;
    uint32_t level_offset[17]; // byte offset of each level from `offset`
    uint32_t _pad[3];
    uint64_t offset;           // byte offset into the texel blob
    uint64_t _pad2;
};

// A cubemap in device memory: per level six faces of (n + 2) x (n + 2) texels, i.e. each face with
// a one-texel border (level_offset counts bytes of that padded layout).
struct GpuCubemap {
    uint32_t size, mip_count, format, _pad;
    uint32_t level_offset[17];
    uint32_t _pad1[3];
    uint64_t offset;
    uint64_t _pad2;
};

// Seamless cube edges: texel (i, j) of `face`, i and j in [-1, n], is the cube texel that contains
// its centre once the centre is projected back onto the cube.  In units of half a texel the centre
// is an integer vector (major component +-n, the others 2i+1-n and 2j+1-n), so face selection and
// texel index need no division and no rounding: texels inside the face map to themselves, edge
// texels to the neighbouring face's border row, the corner texel to the border corner of the face
// that wins the tie order (x before y before z).  Faces are +X, -X, +Y, -Y, +Z, -Z.
inline void rdr_cube_wrap_texel(int& face, int& i, int& j, int n) {
    const int a = 2 * i + 1 - n, b = 2 * j + 1 - n;
    int x, y, z;
    switch (face) {
    case 0: x = n; y = -b; z = -a; break;
    case 1: x = -n; y = -b; z = a; break;
    case 2: x = a; y = n; z = b; break;
    case 3: x = a; y = -n; z = -b; break;
    case 4: x = a; y = -b; z = n; break;
    default: x = -a; y = -b; z = -n; break;
    }
    const int ax = x < 0 ? -x : x, ay = y < 0 ? -y : y, az = z < 0 ? -z : z;
    int sc, tc;
    if (ax >= ay && ax >= az) {
        if (x >= 0) { face = 0; sc = -z; tc = -y; } else { face = 1; sc = z; tc = -y; }
    } else if (ay >= az) {
        if (y >= 0) { face = 2; sc = x; tc = z; } else { face = 3; sc = x; tc = -z; }
    } else {
        if (z >= 0) { face = 4; sc = x; tc = -y; } else { face = 5; sc = -x; tc = -y; }
    }
    const int ii = (sc + n - 1) >> 1, jj = (tc + n - 1) >> 1;
    i = ii < 0 ? 0 : (ii > n - 1 ? n - 1 : ii);
    j = jj < 0 ? 0 : (jj > n - 1 ? n - 1 : jj);
}

// Re-lay a caller's cubemap (levels of six tightly packed n x n faces) into the bordered device
// layout; fills g's size, mip_count, format and level_offset (g.offset is the caller's business).
inline void rdr_pack_cubemap(const rdr_cubemap_desc& cm, GpuCubemap& g, std::vector<uint8_t>& out) {
    const size_t tb = cm.format == RDR_FORMAT_RGBA32F ? 16 : 4;
    g.size = cm.size; g.mip_count = cm.mip_count; g.format = cm.format;
    size_t bytes = 0;
    for (uint32_t l = 0; l < cm.mip_count; ++l) {
        const size_t n = (cm.size >> l) ? (cm.size >> l) : 1u;
        g.level_offset[l] = (uint32_t)bytes;
        bytes += 6 * (n + 2) * (n + 2) * tb;
    }
    out.assign(bytes, 0);
    const uint8_t* src = static_cast<const uint8_t*>(cm.data);
    for (uint32_t l = 0; l < cm.mip_count; ++l) {
        const int n = (int)((cm.size >> l) ? (cm.size >> l) : 1u);
        const size_t pitch = (size_t)n + 2;
        uint8_t* dst = out.data() + g.level_offset[l];
        for (int face = 0; face < 6; ++face) {
            for (int j = -1; j <= n; ++j) {
                uint8_t* row = dst + (((size_t)face * pitch + (size_t)(j + 1)) * pitch) * tb;
                const bool interior = j >= 0 && j < n;
                if (interior) // one copy for the row's own texels, then only its two border texels
                    memcpy(row + tb, src + (((size_t)face * n + j) * n) * tb, (size_t)n * tb);
                for (int i = -1; i <= n; i += interior ? n + 1 : 1) {
                    int f = face, ii = i, jj = j;
                    rdr_cube_wrap_texel(f, ii, jj, n);
                    memcpy(row + (size_t)(i + 1) * tb, src + (((size_t)f * n + jj) * n + ii) * tb, tb);
                }
            }
        }
        src += (size_t)6 * n * n * tb;
    }
}

// Builder-side binary node.
struct BvhNode {
    float lo[3];
    int32_t left; // inner: index of left child (right = left + 1); leaf: first item
    float hi[3];
    int32_t count; // 0 = inner node
};
static_assert(sizeof(BvhNode) == 32, "BvhNode layout");

// Traversal-side node: kBvhWidth children, each a code and a box of 6 halfs
// (rounded outward so the box only grows), stored per axis as (min, max) pairs so
// that a node is four aligned 16-byte loads: codes, x pairs, y pairs, z pairs.  A child code is
// kBvhNone, a leaf (bit 31 | count << 24 | first item) or the index of the child's
// own node.  Wide nodes keep the dependent-load chain of a traversal short.
constexpr int kBvhWidth = 4;
struct BvhWide {
    uint32_t child[kBvhWidth];
    uint16_t box[3][kBvhWidth][2]; // [axis][child][min, max]
};
static_assert(sizeof(BvhWide) == kBvhWidth * 16, "BvhWide layout");
constexpr uint32_t kLaneMaxPixels = 16, kMidMaxPixels = 1024; // defaults of the routing thresholds
constexpr uint32_t kBvhNone = 0xFFFFFFFFu;
constexpr uint32_t kBvhLeaf = 0x80000000u;
constexpr uint32_t kBvhMaxLeaf = 127u;
constexpr uint32_t kSunCandidates = 8; // occluder candidates kept per instance

struct GpuInstance {
    float inv[12];      // rows of the inverse affine model matrix
    uint32_t blas_root; // index of the mesh's first BLAS node
    uint32_t tri_base;  // first triangle of the mesh in the BLAS triangle array
    // The mesh is the closed boundary of a convex body: a shadow ray that leaves one of its LOD-0 faces
    // outwards cannot meet the mesh again, so the instance's own bottom level is skipped for it.
    // 0 = not convex (never skip); otherwise |convex_cos| is the smallest cosine between the outward
    // face normal and the ray for which the skip is taken (it covers rounding of the ray's origin and of
    // the triangle tests, and the mesh's deviation from exact convexity), and the sign tells which way the
    // world-space cross product (w2 - w1) x (w3 - w1) of a face points (+ outwards, - inwards).
    float convex_cos;
    // Every mesh lies within outer_radius of a centre (SceneView::instance_ball.xyz, object space; 0 = unknown): a ray
    // meets the instance only between the parameters at which it enters and leaves that ball.  A closed mesh around a
    // convex body also contains a ball about the same centre (instance_ball.w, 0 = none): a ray that is inside it at a
    // parameter beyond tmin is inside the body there and outside it before tmax, so it crosses the surface in
    // between, and the ray / triangle test is watertight: some triangle is hit.  Such a ray is blocked without a walk.
    float outer_radius;
};

// Everything the kernels read, passed by value.
// What the shading passes need of a draw, written per frame by the draw cull next to draw_lod: with it a
// pixel's key leads to its triangle's vertices in three dependent loads (instance_draw -> draw_frame ->
// tri_table) instead of six (instance_draw -> draws -> meshes / draw_lod -> meshlets -> triangle bytes ->
// meshlet_vertices); those chains of round trips, not bandwidth, are what the two passes spend their time on.
struct DrawFrame {
    uint32_t meshlet_base;       // chosen LOD's meshlet_offset - the draw's meshlet_visibility_offset (wraps): + instance = meshlet
    uint32_t model_index;
    uint32_t material_lod;       // material_index | lod << 16
    uint32_t mesh_vertex_offset;
};
static_assert(sizeof(DrawFrame) == 16, "DrawFrame layout");
// The three vertices of triangle t of meshlet m, relative to the mesh's first vertex, at index m * 128 + t
// (built once per scene from meshlet_triangles and meshlet_vertices).
struct TriVertices {
    uint32_t v[3];
};
constexpr uint32_t kTriTableStride = 128;

// What the alpha test of a draw needs (make_alpha_test in rdr_geometry.hip; SceneView::draw_alpha holds the same per
// draw, written by the host at upload, for the kernel whose lanes each work on another draw).
struct AlphaTest {
    bool enabled, has_tex;
    uint32_t image, uv_set;
    float sx, sy, ox, oy, factor_a, cutoff;
};

struct SceneView {
    const rdr_vertex* vertices;
    const float* pos_x;
    const float* pos_y;
    const float* pos_z;
    const float4* uv_sets;      // uv0.xy | uv1.zw per vertex, apart from the 64-byte vertex (the rasteriser's alpha test)
    // The same four arrays in the order of meshlet_vertices (entry s = the vertex meshlet_vertices[s] names within
    // its mesh): a meshlet's 64 positions are then 64 consecutive floats per plane that need nothing but the
    // meshlet's record, not a gather behind an index load.  Null when two meshes with different vertex offsets
    // share meshlets (an entry would mean two vertices); the rasteriser then goes through the indices.
    const float* mpos_x;
    const float* mpos_y;
    const float* mpos_z;
    const float4* muv;
    const uint32_t* meshlet_vertices;
    const uint8_t* meshlet_triangles;
    const rdr_meshlet* meshlets;
    const rdr_mesh* meshes;
    const rdr_mesh_draw* draws;
    const rdr_material* materials;
    const float* transforms;
    const uint32_t* instance_draw; // meshlet instance -> draw id
    const AlphaTest* draw_alpha;   // draw -> its alpha test
    const float* mesh_tri_area;    // per mesh: mean area of a level-0 triangle in object space (FrameView::replicate_pixels)
    const float4* instance_ball;   // per draw: see GpuInstance::outer_radius
    // Sun occluder candidates (k_sun_candidates, for one sun direction and one set of transforms): the instances
    // that could block a sun ray starting anywhere on draw d are tlas_items[sun_cand_base + d * kSunCandidates ..]
    // (sun_cand_count[d] of them); 255 = too many or unknown, walk the top level.
    const float* instance_reach;   // per draw: object-space distance from instance_ball.xyz that holds every vertex of the mesh (all levels of detail)
    uint8_t* sun_cand_count;
    float4* world_ball;            // per draw (same kernel): world-space centre and radius of a ball around the LOD-0 mesh, w = 0: unknown
    uint32_t* sun_cand_items;      // == tlas_items + sun_cand_base
    uint32_t sun_cand_base;
    uint8_t* draw_lod;             // per frame: LOD each draw selected, 0xFF = not drawn
    DrawFrame* draw_frame;         // per frame, per draw (same writer)
    const TriVertices* tri_table;  // per (meshlet, triangle)
    const GpuImage* images;
    const uint8_t* texels;
    const float* srgb_lut;
    uint32_t draw_count, material_count, image_count, meshlet_count;
    uint32_t masked_meshlets; // meshlets of alpha-tested draws in the scene: 0 = their kernels are not launched
    // shadow-ray structure
    const BvhWide* bvh;  // every mesh's bottom-level nodes, then the top-level nodes
    uint32_t tlas_base;  // index of the top level's node 0 in `bvh`
    const uint32_t* tlas_items;
    const float* blas_tris; // 9 floats per triangle, leaf order
    const GpuInstance* instances;
    // environment
    GpuImage brdf_lut;
    GpuCubemap lambertian, ggx, charlie, skybox;
};

// Launch geometry of one frame and where each kernel's workgroups put their statistics.
struct GridPlan {
    uint32_t draw_blocks, cull_blocks, raster_blocks, raster_mask_blocks, mid_blocks, frag_blocks, deferred_blocks, shade_blocks_x, shade_blocks_y;
    uint32_t rare_blocks; // grid of k_tile_path_rare (0: never launched), launched instead of the three tile-path kernels when tile_path_rare is set
    uint32_t tile_path_rare;
    uint32_t slot_draw, slot_cull, slot_raster, slot_mid, slot_frag, slot_deferred, slot_shade, slot_total;
};
constexpr uint32_t kCullSlots = 4, kRasterSlots = 8;

struct FrameView {
    rdr_ubo ubo;
    unsigned long long* vis;
    uint8_t* color;
    CullCommand* commands;
    VisibleMeshlet* visible;
    DeferredTriangle* deferred;
    FrameCounters* counters;
    // The list heads and the per-tile counts exist twice and frames alternate between the sets: a frame's last
    // geometry kernel zeroes the set the next frame will use, so no frame starts with a launch that only clears.
    FrameCounters* counters_next;
    uint32_t* tile_count_next;
    uint32_t width, height;
    uint32_t part_lo, part_hi;
    // Interleaved sort-last share on top of the range (part_world > 1): chunk j of draw d (chunks of part_chunk
    // consecutive meshlets of the draw's level of detail) belongs to rank (j + d) % part_world.
    uint32_t part_chunk, part_rank, part_world;
    // > 0 (and part_world > 1): a draw whose triangles are estimated to cover more than this many pixels each on
    // average is rasterised by every rank (all its meshlets), and its large triangles (the tile path) only on the rows
    // the rank shades: what a rank's buffer is read for are its own rows, and the minimum does not care how many ranks
    // contribute a key.  Deals the pixel work of a few screen-filling triangles out by rows instead of leaving it with
    // the one rank whose share holds their meshlet.
    float replicate_pixels;
    // The rows this renderer shades, as `shade_rows` virtual rows: a contiguous range (band_world <= 1: virtual
    // row v is row row_lo + v) or every band_world-th band of (1 << band_shift) rows starting with band band_rank.
    uint32_t row_lo, row_hi;
    uint32_t band_shift, band_rank, band_world, shade_rows;
    uint32_t command_capacity, visible_capacity, deferred_capacity;
    uint32_t collect_stats;
    uint32_t* pixel_list; // the covered pixels of the shaded rows (k_classify), as y << 16 | x
    uint32_t* pixel_key;  // per pixel_list entry: the low word of the pixel's key, (instance << 7) | triangle
    uint8_t* shadow;      // one byte per pixel_list entry: 1 = the sun ray is blocked (padded to a multiple of 16 bytes)
    SetupTri* setup;
    SetupMask* setup_mask;
    uint32_t* tile_count;  // per tile: triangles binned
    uint32_t* tile_offset; // per tile: first entry in bin_entries
    uint32_t* tile_cursor; // per tile: fill cursor
    uint2* tile_list;      // the tile rasteriser's jobs: (tile, first record of its bin), kTileJob records each
    uint32_t tile_list_capacity;
    uint32_t* bin_entries; // SetupTri indices grouped by tile
    uint32_t setup_capacity, bin_capacity, tiles_x, tiles_y;
    SetupTri* mid;        // scan set-ups of mid-sized triangles (k_raster -> k_mid_raster)
    SetupMask* mid_mask;
    uint32_t mid_capacity;
    FragTri* frag_tri;          // small alpha-tested triangles (k_raster<true> -> k_alpha_frag), kFragSegments segments
    uint32_t* frag_px;          // their covered pixels: triangle slot << 4 | index of the pixel in its box
    uint32_t frag_tri_capacity, frag_px_capacity;
    // Routing of a triangle whose bounding box is at most 64 px on a side, by the pixels in
    // the box: up to lane_max_pixels one lane of k_raster scans it, up to mid_max_pixels it is
    // queued for k_mid_raster (a wave per triangle, 8x8 pixels per step), beyond that (and
    // for larger or clipped triangles) it goes to the tile path.
    uint32_t lane_max_pixels, mid_max_pixels;
    uint32_t skybox; // 1: uncovered pixels get the skybox (k_skybox) instead of the clear colour
    // Sort-last bookkeeping, only while a peer-to-peer exchange is connected (null otherwise): one word per
    // kTouchTile x kTouchTile pixel tile, non-zero when a rasteriser may have written a key into the tile this
    // frame.  It lives behind the visibility buffer in the same allocation, is zeroed with it, and lets the
    // peers skip the tiles this rank left clear.
    uint32_t* touched;
    uint32_t touched_x, touched_count; // tiles per row, tiles in all
    uint32_t list_heads_zero; // host side: pixel_count / shadow_overflow are still as the geometry pass's clear left them
    // Two-phase occlusion culling (result-neutral): state kept from the previous frame and the
    // depth pyramid of the early phase.  Level l holds, per 2^(l+1) x 2^(l+1) pixel block, the
    // smallest (farthest) depth bits of the block's pixels; the last level is 1 x 1.
    uint32_t occlusion;        // 1: run early phase -> pyramid -> late phase
    uint32_t stat_accumulate;  // 1: kernels add to their statistics slots (late phase) instead of storing
    uint8_t* draw_visible;     // per draw: passed the late phase's tests last frame
    uint32_t* meshlet_bits;    // per meshlet instance: 1 bit, accepted by the late phase last frame
    uint32_t* pyramid;
    uint32_t pyr_levels;
    uint32_t pyr_offset[16], pyr_w[16], pyr_h[16];
    uint32_t lds_stack_limit; // entries of the LDS traversal stack in use (<= 24; lowered only by tests)
    // Frame status in pinned host memory: word 0 is written by the last kernel of the geometry half, word 1 by
    // the last kernel of the shading half, each as frame_seq << 8 | RDR_FRAME_* flags, so that the host learns
    // of dropped work without synchronising (it looks at the words when the next frame is submitted).  Word 2:
    // frame_seq << 8 | work: bit 0 when the frame had work for the tile path or the mid-sized queue, bit 1 when it had
    // triangles for the tile rasteriser itself (FrameCounters::tile_work).
    uint32_t* host_status;
    uint32_t frame_seq;
    uint32_t convex_skip; // 1: shadow rays that leave a closed convex mesh skip that mesh (0 only as a measurement hook)
    uint32_t sun_candidates; // 1: SceneView::sun_cand_* hold this frame's sun direction and transforms (0 only as a measurement hook)
    // Per-frame values that are the same in every lane, worked out once on the host with the arithmetic the kernels
    // would use (same expressions, same rounding): the normalised sun direction, 2 / width, 2 / height, and the
    // reciprocals 1 / width, 1 / height that turn a pixel centre into its 0..1 screen coordinate with a product.
    float sun_unit[3];
    float two_over_width, two_over_height;
    float inv_width, inv_height;
    uint32_t shadow_defer; // 1: walks through a ray's own closed convex instance are left to k_shadow_walk (0 only as a measurement hook)
    // Per-workgroup statistics: every workgroup stores its partial counts in its own
    // slot with plain stores (no contended atomics); the host sums the slots on request.
    uint32_t* stat_slots;
    GridPlan grid;
};

// Frame row of the renderer's v-th shaded row (FrameView::shade_rows of them).
__host__ __device__ inline uint32_t rdr_shade_row(const FrameView& fv, uint32_t v) {
    if (fv.band_world <= 1u) return fv.row_lo + v;
    const uint32_t band = v >> fv.band_shift;
    return (((band * fv.band_world + fv.band_rank) << fv.band_shift) | (v & ((1u << fv.band_shift) - 1u)));
}
// Does the renderer shade any of the frame rows y0 .. y1 (inclusive)?
__host__ __device__ inline bool rdr_rows_owned(const FrameView& fv, uint32_t y0, uint32_t y1) {
    if (fv.band_world <= 1u) return y1 >= fv.row_lo && y0 < fv.row_hi;
    const uint32_t b0 = y0 >> fv.band_shift, b1 = y1 >> fv.band_shift;
    // the first band at or after b0 that belongs to band_rank
    return (fv.band_rank + fv.band_world - b0 % fv.band_world) % fv.band_world <= b1 - b0;
}
// Rows of a frame of `height` rows that belong to rank `rank` of `world` when bands of (1 << shift) rows are dealt round-robin.
inline uint32_t rdr_band_rows(uint32_t height, uint32_t shift, uint32_t rank, uint32_t world) {
    const uint32_t band = 1u << shift, cycle = band * world;
    const uint32_t full = height / cycle, rest = height % cycle;
    const uint32_t lo = rank * band;
    return full * band + (rest > lo ? (rest - lo < band ? rest - lo : band) : 0u);
}

// ---- host BVH builder (rdr_bvh.cpp) -----------------------------------------
struct HostBvh {
    std::vector<BvhWide> blas_pairs, tlas_pairs;
    std::vector<BvhNode> mesh_bounds; // per mesh: lo/hi of the whole BLAS
    std::vector<BvhNode> blas_nodes;
    std::vector<float> blas_tris;
    std::vector<uint32_t> mesh_root;     // per mesh
    std::vector<uint32_t> mesh_tri_base; // per mesh
    std::vector<int8_t> mesh_convex_sign;   // per mesh: 0 not a closed convex surface, +1 / -1 face winding outwards / inwards
    std::vector<float> mesh_protrusion;     // per mesh: how far a vertex lies outside a face plane at most (object units)
    std::vector<float> mesh_ball;           // per mesh: centre, inner radius (0: none), outer radius (GpuInstance::outer_radius)
    std::vector<float> instance_ball;       // per instance: centre and inner radius, for the kernels
    std::vector<float> mesh_reach;          // per mesh: distance from the ball's centre that holds every vertex of the mesh's vertex range
    std::vector<float> instance_reach;      // per instance: its mesh's
    std::vector<BvhNode> tlas_nodes;
    std::vector<uint32_t> tlas_items;
    std::vector<GpuInstance> instances;
};
void rdr_build_blas(const rdr_scene_desc& s, HostBvh& out);
void rdr_build_tlas(const rdr_scene_desc& s, const float* transforms, HostBvh& io);

// ---- kernel launchers (rdr_geometry.hip / rdr_shade.hip) ---------------------
int rdr_launch_geometry(const SceneView& sv, const FrameView& fv, hipStream_t stream, hipEvent_t* ev);
struct GeometryResidency {
    uint32_t meshlet_cull, raster, raster_mask, mid_raster, alpha_frag, tile_raster; // resident workgroups on the whole device
    uint32_t tile_path_rare; // k_tile_path_rare (0: does not fit, e.g. its LDS at this frame size)
};
int rdr_geometry_residency(GeometryResidency* out, uint32_t fill_lds_bytes);
// Dynamic LDS of k_mid_fill: one bin offset per 64x64 tile and the 256 partial sums of its prefix sum.
inline uint32_t rdr_tile_fill_lds(uint32_t n_tiles) { return ((n_tiles + 1u) & ~1u) * 4u + 256u * 8u; }
int rdr_launch_shadow(const SceneView& sv, const FrameView& fv, hipStream_t stream, hipEvent_t* ev); // classify + shadow rays
int rdr_launch_sun_candidates(const SceneView& sv, const FrameView& fv, hipStream_t stream); // SceneView::sun_cand_* for fv's sun direction
int rdr_launch_shade(const SceneView& sv, const FrameView& fv, hipStream_t stream, hipEvent_t* ev);  // the above, then k_shade
// ---- peer-to-peer sort-last exchange (rdr_composite.hip) ----------------------------
struct rdr_composite_state;
int rdr_composite_create(rdr_composite_state** out);
void rdr_composite_close(rdr_composite_state* c);
void rdr_composite_destroy(rdr_composite_state* c);
int rdr_composite_export_handles(rdr_composite_state* c, void* vis, void* color, rdr_composite_handle* out);
int rdr_composite_open(rdr_composite_state* c, uint32_t rank, uint32_t world, const rdr_composite_handle* handles, void* vis, void* color,
                       uint32_t width, uint32_t height);
bool rdr_composite_connected(const rdr_composite_state* c);
int rdr_composite_check(rdr_composite_state* c, const char* who);
void rdr_composite_mark_broken(rdr_composite_state* c); // a frame of this rank failed while connected
bool rdr_composite_timed_out(const rdr_composite_state* c); // a bounded wait gave up (not cleared by asking)
// The rows a rank composites and pushes: [row_lo, row_hi), or (band_world > 1) its bands of (1 << band_shift) rows.
struct RowSet {
    uint32_t row_lo, row_hi;
    uint32_t band_shift, band_rank, band_world;
};
int rdr_composite_launch_exchange(rdr_composite_state* c, const RowSet& rows, hipStream_t stream);
int rdr_composite_launch_gather(rdr_composite_state* c, const RowSet& rows, bool every_tile, hipStream_t stream);
// Edge of the tiles of FrameView::touched, and the words to reserve for them behind a width x height visibility buffer.
constexpr uint32_t kTouchTile = 32;
inline uint32_t rdr_touched_words(uint32_t width, uint32_t height) {
    return ((width + kTouchTile - 1) / kTouchTile) * ((height + kTouchTile - 1) / kTouchTile);
}
size_t rdr_bc_level_bytes(uint32_t format, uint32_t width, uint32_t height);
int rdr_launch_decode_bc(uint32_t format, uint32_t width, uint32_t height, const uint8_t* blocks, uint32_t* out, hipStream_t stream);
int rdr_launch_resolve(const SceneView& sv, const FrameView& fv, uint32_t* out_xy, hipStream_t stream);
