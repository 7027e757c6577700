// rdr_geometry.hip — cull -> vertex transform -> rasterise into the 64-bit
// depth|ID visibility buffer.
//
// What the kernels compute follows the reference's shaders (paths relative to
// the reference tree):
//   k_draw_cull      assets/shaders/cull.comp.glsl:54-130
//   k_meshlet_cull   assets/shaders/shader.task.glsl:59-137
//   k_raster         assets/shaders/shader.mesh.glsl:72-131 + the fixed-function
//                    state of lava/src/pipelines.rs:149-191 + the alpha test of
//                    assets/shaders/shader.frag.glsl:55-69
// How they compute it is specific to CDNA4: wave64 ballot/prefix compaction in
// place of per-item global atomics, persistent grids reading work-list lengths
// from memory in place of indirect dispatch, one wave per meshlet with the 64
// transformed vertices in LDS, and a 64-bit atomic min per covered sample in
// place of ROPs + a depth buffer.
//
// The reference's early/late occlusion split (cull.comp.glsl:60-95, shader.task.glsl:
// 96-128, depth_reduce.comp.glsl) is a conservative optimisation that does not change
// the image.  With occlusion culling off one pass over every draw is run; with it on,
// the same two phases run here: what was visible last frame, a depth pyramid of that,
// then everything else tested against the pyramid.
#include "rdr_device.h"

namespace {

constexpr float kGuardBand = 8.0f;
constexpr int kSmallExtent = 64 * 256; // bounding boxes up to 64 px on a side stay in k_raster / k_mid_raster (int32 exact); see FrameView::lane_max_pixels
constexpr int kTile = 64;              // screen tile of the binned path, pixels
constexpr int kTileBatch = 64;
constexpr uint32_t kJobGroup = 4;      // k_tile_raster: jobs a workgroup fetches together
constexpr int kTilePart = 32;           // k_tile_raster: a job covers a kTilePart x kTilePart piece of a tile
constexpr uint32_t kTileParts = (kTile / kTilePart) * (kTile / kTilePart);
static_assert(kTileParts <= kTileJob, "the part index travels in the low bits of the job's first record");
constexpr int kTileBlock = 2;          // k_tile_raster: a thread owns kTileBlock x kTileBlock pixels of the tile         // triangle records staged in LDS per round

enum : int { kPhaseSingle = 0, kPhaseEarly = 1, kPhaseLate = 2 };

// Store or, in the late phase, add a per-workgroup statistic.
RDR_DEV void stat_store(const FrameView& fv, uint32_t slot, uint32_t v) {
    fv.stat_slots[slot] = fv.stat_accumulate ? fv.stat_slots[slot] + v : v;
}

// Is everything inside the view-space sphere (c, r) behind what the early phase drew?
// Follows the reference's test (projectSphere, assets/shaders/math.inc.glsl:5-26; use at
// cull.comp.glsl:82-94 and shader.task.glsl:103-115) but is strictly conservative, because
// the result must not change a single key: the sphere is padded (1.6 % of the radius plus
// 0.1 % of the distance, far above the rounding of the transforms), its screen box is widened
// by a pixel, the pyramid level is chosen so that the box touches at most 2 x 2 texels, all of
// which are read, and only a sphere whose nearest depth is strictly below the block minimum
// is reported.  Spheres that reach the near plane are never reported.
// The pixel box [ix0, ix1] x [iy0, iy1] that holds everything inside the view-space sphere (c, r), padded and widened
// as described above and clipped to the frame (ix0 > ix1 or iy0 > iy1: nothing on screen); `r` comes back padded.
// False when the sphere reaches the near plane or the box is not a number: no statement.
RDR_DEV bool sphere_screen_box(const FrameView& fv, vec3 c, float& r, int& ix0, int& ix1, int& iy0, int& iy1) {
    const rdr_ubo& u = fv.ubo;
    r = r * 1.015625f + fabsf(c.z) * 0.0009765625f;
    if (!(c.z > r + u.near_plane)) return false;
    const float czr2 = c.z * c.z - r * r;
    const float vx = sqrtf(c.x * c.x + czr2), vy = sqrtf(c.y * c.y + czr2);
    const float ax = (vx * c.x - c.z * r) / (vx * c.z + c.x * r), bx = (vx * c.x + c.z * r) / (vx * c.z - c.x * r);
    const float ay = (vy * c.y - c.z * r) / (vy * c.z + c.y * r), by = (vy * c.y + c.z * r) / (vy * c.z - c.y * r);
    // view-space slopes -> NDC -> framebuffer (x_f = W/2 ndc.x + W/2, y_f = -H/2 ndc.y + H/2)
    const float W = (float)fv.width, H = (float)fv.height;
    const float xa = (ax * u.proj[0] * 0.5f + 0.5f) * W, xb = (bx * u.proj[0] * 0.5f + 0.5f) * W;
    const float ya = (0.5f - ay * u.proj[5] * 0.5f) * H, yb = (0.5f - by * u.proj[5] * 0.5f) * H;
    const float x_lo = fminf(xa, xb), x_hi = fmaxf(xa, xb), y_lo = fminf(ya, yb), y_hi = fmaxf(ya, yb);
    if (!(x_lo <= x_hi) || !(y_lo <= y_hi)) return false; // NaN: no statement
    ix0 = max((int)floorf(clampf(x_lo, -4.0f, W + 4.0f)) - 1, 0); ix1 = min((int)floorf(clampf(x_hi, -4.0f, W + 4.0f)) + 1, (int)fv.width - 1);
    iy0 = max((int)floorf(clampf(y_lo, -4.0f, H + 4.0f)) - 1, 0); iy1 = min((int)floorf(clampf(y_hi, -4.0f, H + 4.0f)) + 1, (int)fv.height - 1);
    return true;
}

RDR_DEV bool sphere_occluded(const FrameView& fv, vec3 c, float r) {
    const rdr_ubo& u = fv.ubo;
    int ix0, ix1, iy0, iy1;
    if (!sphere_screen_box(fv, c, r, ix0, ix1, iy0, iy1)) return false;
    if (ix0 > ix1 || iy0 > iy1) return false; // off screen: the frustum tests decide
    const uint32_t span = (uint32_t)max(ix1 - ix0, iy1 - iy0) + 1u;
    // smallest level whose texels (2^(l+1) pixels) are at least `span` wide: the box then touches <= 2 per axis
    uint32_t l = span <= 2u ? 0u : 31u - (uint32_t)__clz((int)(span - 1u));
    l = min(l, fv.pyr_levels - 1u);
    const uint32_t sh = l + 1u;
    const uint32_t tx0 = (uint32_t)ix0 >> sh, tx1 = (uint32_t)ix1 >> sh, ty0 = (uint32_t)iy0 >> sh, ty1 = (uint32_t)iy1 >> sh;
    const uint32_t* lv = fv.pyramid + fv.pyr_offset[l];
    const uint32_t pw = fv.pyr_w[l];
    uint32_t far_bits = 0xFFFFFFFFu;
    for (uint32_t ty = ty0; ty <= ty1; ++ty)
        for (uint32_t tx = tx0; tx <= tx1; ++tx) far_bits = min(far_bits, lv[ty * pw + tx]);
    const float nearest = u.near_plane / (c.z - r); // depth = near / view z (reverse-Z, infinite far plane)
    return nearest < __uint_as_float(far_bits);
}

// The meshlets [k_lo, k_hi) of draw `di` that belong to this renderer's sort-last share, as spans f(lo, hi) of at
// most one chunk each: everything (no interleaving, or a draw every rank draws), or (FrameView::part_world > 1) the
// chunks j with (j + di) % world == rank.
template <typename F>
RDR_DEV void own_spans(const FrameView& fv, uint32_t di, uint32_t k_lo, uint32_t k_hi, bool everything, F f) {
    if (fv.part_world <= 1u || everything) {
        if (k_hi > k_lo) f(k_lo, k_hi);
        return;
    }
    uint32_t j = k_lo / fv.part_chunk;
    j += (fv.part_rank + fv.part_world - (j + di) % fv.part_world) % fv.part_world; // the first own chunk at or after j
    for (; j * fv.part_chunk < k_hi; j += fv.part_world) f(max(k_lo, j * fv.part_chunk), min(k_hi, (j + 1u) * fv.part_chunk));
}

// ------------------------------------------------------------------------------
// Draw cull.  PHASE: single pass; early (draws visible last frame only, opaque only);
// late (every draw, opaque ones tested against the pyramid; masked draws are emitted
// here and only here, as in the reference's separate mask pass).
// ------------------------------------------------------------------------------
template <int PHASE>
__global__ __launch_bounds__(256) void k_draw_cull(SceneView sv, FrameView fv) {
    const uint32_t di = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = lane_id();
    const rdr_ubo& u = fv.ubo;
    // The sort-last tile flags start the frame at zero.  Here, in the frame's first kernel: the next one, the
    // meshlet cull, already sets them.
    if (PHASE != kPhaseLate && fv.touched)
        for (uint32_t i = di; i < fv.touched_count; i += gridDim.x * blockDim.x) fv.touched[i] = 0u;

    uint32_t ncmd = 0, k_lo = 0, k_hi = 0, lod_meshlet_offset = 0, vis_off = 0, was_visible = 0, replicated = 0;
    float max_scale = 0.0f;
    bool emitted = false;
    uint32_t draw_model_index = 0, draw_mesh_vertex_offset = 0;
    bool draw_masked = false;
    if (di < sv.draw_count) {
        const rdr_mesh_draw d = sv.draws[di];
        if (PHASE != kPhaseSingle) was_visible = fv.draw_visible[di];
        const rdr_mesh* mesh = &sv.meshes[d.mesh_index];
        draw_model_index = d.model_index;
        draw_mesh_vertex_offset = mesh->vertex_offset;
        draw_masked = d.alpha_mode == 1u;
        float model[16];
        load_m4(sv.transforms + (size_t)d.model_index * 16, model);
        const vec4 cw = mul_m4(model, mk4(mesh->center[0], mesh->center[1], mesh->center[2], 1.0f));
        const vec4 cv = mul_m4(u.view, cw);
        const vec3 c = mk3(cv.x, cv.y, cv.z);
        max_scale = model_max_scale(model);
        const float radius = max_scale * mesh->radius;

        bool visible = true;
        visible = visible && (c.z * -u.frustum[1] + fabsf(c.x) * u.frustum[0] < radius);
        visible = visible && (c.z * -u.frustum[3] + fabsf(c.y) * u.frustum[2] < radius);
        visible = visible && (c.z + radius > -u.near_plane) && (c.z + radius < u.far_plane);
        if (PHASE == kPhaseLate) {
            if (visible && d.alpha_mode != 1u) visible = !sphere_occluded(fv, c, radius);
            fv.draw_visible[di] = visible ? 1 : 0;
        }
        bool emit = visible || d.alpha_mode == 1u;
        if (PHASE == kPhaseEarly) emit = visible && was_visible != 0u && d.alpha_mode != 1u;

        uint32_t lod = 0xFFu;
        if (emit) {
            uint32_t lod_index = 0;
            const float threshold = fmaxf(len3(c) - radius, 0.0f) * u.lod_target / max_scale;
            const uint32_t lod_count = mesh->lod_count;
            for (uint32_t i = 1; i < lod_count; ++i)
                if (mesh->lods[i].error < threshold) lod_index = i;
            lod = u.lod_enabled == 1u ? lod_index : 0u;
            const rdr_mesh_lod L = mesh->lods[lod];
            lod_meshlet_offset = L.meshlet_offset;
            vis_off = d.meshlet_visibility_offset;
            // this renderer's sort-last share, in meshlet-instance ids
            const uint32_t lo = max(vis_off, fv.part_lo), hi = min(vis_off + L.meshlet_count, fv.part_hi);
            if (fv.part_world > 1u && fv.replicate_pixels > 0.0f) {
                // mean projected area of a triangle of the selected level, at the sphere's nearest point: object-space
                // area x scale^2 x (pixels per unit at that depth)^2; coarser levels have as many times larger triangles
                // as they have fewer meshlets
                const float zmin = fmaxf(c.z - radius, u.near_plane);
                const float ppu = 0.5f * u.screen_height * u.proj[5] / zmin;
                const float area = sv.mesh_tri_area[d.mesh_index] * (max_scale * max_scale) * (ppu * ppu) *
                                   ((float)mesh->lods[0].meshlet_count / (float)max(L.meshlet_count, 1u));
                replicated = area > fv.replicate_pixels ? 1u : 0u; // the same verdict on every rank: it depends on the scene and the view only
            }
            if (hi > lo) {
                k_lo = lo - vis_off;
                k_hi = hi - vis_off;
                own_spans(fv, di, k_lo, k_hi, replicated != 0u, [&](uint32_t a, uint32_t b) { ncmd += (b - a + 63u) / 64u; });
            }
            emitted = true;
        }
        if (PHASE != kPhaseEarly) { // the late phase sees every draw the early one saw
            sv.draw_lod[di] = (uint8_t)lod;
            DrawFrame df;
            df.meshlet_base = lod_meshlet_offset - d.meshlet_visibility_offset;
            df.model_index = d.model_index;
            df.material_lod = (uint32_t)d.material_index | (lod << 16);
            df.mesh_vertex_offset = mesh->vertex_offset;
            sv.draw_frame[di] = df;
        }
    }

    // wave-level compaction: one atomic per wave reserves the commands of 64 draws
    const uint32_t incl = wave_scan_inclusive(ncmd, lane);
    const uint32_t total = __shfl(incl, 63, 64);
    uint32_t base = 0;
    if (lane == 0 && total) base = atomicAdd(&fv.counters->cmd_count, total);
    base = __shfl(base, 0, 64);
    uint32_t slot = base + incl - ncmd;
    if (ncmd)
        own_spans(fv, di, k_lo, k_hi, replicated != 0u, [&](uint32_t a, uint32_t b) {
            for (uint32_t k = a; k < b; k += 64u, ++slot) {
                if (slot >= fv.command_capacity) return;
                CullCommand cmd;
                cmd.draw_id = di;
                cmd.meshlet_first = lod_meshlet_offset + k;
                cmd.count = min(64u, b - k);
                cmd.instance_first = vis_off + k;
                cmd.max_scale = max_scale;
                cmd.flags = (was_visible ? kCmdLateVisible : 0u) | (replicated ? kCmdReplicated : 0u) | (draw_masked ? kCmdMasked : 0u);
                cmd.model_index = draw_model_index;
                cmd.mesh_vertex_offset = draw_mesh_vertex_offset;
                fv.commands[slot] = cmd;
            }
        });
    if (fv.collect_stats) {
        __shared__ uint32_t wg_emitted[4];
        const uint32_t n = wave_sum(emitted ? 1u : 0u);
        if (lane == 0) wg_emitted[threadIdx.x >> 6] = n;
        __syncthreads();
        if (threadIdx.x == 0 && PHASE != kPhaseEarly) // draws_visible: what the single / late phase emitted
            fv.stat_slots[fv.grid.slot_draw + blockIdx.x] = wg_emitted[0] + wg_emitted[1] + wg_emitted[2] + wg_emitted[3];
    }
}

// ------------------------------------------------------------------------------
// Meshlet cull: one wave per command, one lane per meshlet; the 16 waves of a
// workgroup reserve their output slots with a single atomic.
// ------------------------------------------------------------------------------
// One workgroup's share of the visibility-buffer clear: 1024 threads, kClearPerThread 16-byte stores each,
// consecutive threads on consecutive addresses.
constexpr uint32_t kClearPerThread = 4;
RDR_DEV void clear_visbuffer(const FrameView& fv, uint32_t block, uint32_t blocks) {
    const size_t n = (size_t)fv.width * fv.height;
    const size_t base = (size_t)block * (1024u * kClearPerThread) + threadIdx.x;
#pragma unroll
    for (uint32_t k = 0; k < kClearPerThread; ++k) {
        const size_t i = (base + (size_t)k * 1024u) * 2;
        if (i + 1 < n) {
            *reinterpret_cast<ulonglong2*>(fv.vis + i) = make_ulonglong2(RDR_VIS_CLEAR, RDR_VIS_CLEAR);
        } else if (i < n) {
            fv.vis[i] = RDR_VIS_CLEAR;
        }
    }
}

template <int PHASE>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8))) void k_meshlet_cull(SceneView sv, FrameView fv) {
    __shared__ uint32_t wave_count[16];
    __shared__ uint32_t wave_masked[16];
    __shared__ uint32_t wg_base[2];
    // Workgroups past the culling grid clear the visibility buffer: nothing reads it before the rasterisers,
    // and the culling grid (a few hundred workgroups waiting on dependent loads) leaves the machine idle
    // enough that the 8 bytes per pixel go out underneath it instead of in a pass of their own.
    if (blockIdx.x >= fv.grid.cull_blocks) {
        clear_visbuffer(fv, blockIdx.x - fv.grid.cull_blocks, gridDim.x - fv.grid.cull_blocks);
        return;
    }
    const uint32_t lane = lane_id();
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // the same in the whole wave, and known to be
    const rdr_ubo& u = fv.ubo;
    const uint32_t cmd_count = min(fv.counters->cmd_count, fv.command_capacity);

    uint32_t st_tested = 0, st_visible = 0, st_tris = 0;
    for (uint32_t c0 = blockIdx.x * 16u; c0 < cmd_count; c0 += fv.grid.cull_blocks * 16u) {
        const uint32_t ci = c0 + wave;
        bool accept = false;
        uint32_t draw_id = 0, first = 0, instance_first = 0, tri_count = 0, count = 0, masked = 0;
        uint32_t model_index = 0, mesh_vertex_offset = 0, m_vertex_offset = 0, m_triangle_offset = 0, m_counts = 0, replicated = 0;
        bool emit = false;
        if (ci < cmd_count) {
            // the command and everything it leads to is the same for the whole wave: read through the scalar unit
            const auto* cp = &uniform_ref(&fv.commands[__builtin_amdgcn_readfirstlane(ci)]);
            draw_id = cp->draw_id; first = cp->meshlet_first; count = cp->count;
            instance_first = cp->instance_first;
            const uint32_t cmd_flags = cp->flags;
            replicated = (cmd_flags & kCmdReplicated) ? 1u : 0u;
            const float max_scale = cp->max_scale;
            model_index = cp->model_index;
            mesh_vertex_offset = cp->mesh_vertex_offset;
            masked = (cmd_flags & kCmdMasked) ? 1u : 0u;
            float model[16];
            load_m4_uniform(sv.transforms + (size_t)model_index * 16, model);
            accept = lane < count;
            if (accept) {
                const rdr_meshlet m = sv.meshlets[first + lane];
                tri_count = m.triangle_count;
                m_vertex_offset = m.vertex_offset; m_triangle_offset = m.triangle_offset;
                m_counts = (uint32_t)m.vertex_count | ((uint32_t)m.triangle_count << 16) | (replicated ? 0x80000000u : 0u);
                const vec4 cw = mul_m4(model, mk4(m.center[0], m.center[1], m.center[2], 1.0f));
                const vec4 cv = mul_m4(u.view, cw);
                const vec3 cc = mk3(cv.x, cv.y, cv.z);
                const vec4 aw = mul_m4(model, mk4((float)m.cone_axis[0] / 127.0f, (float)m.cone_axis[1] / 127.0f,
                                                  (float)m.cone_axis[2] / 127.0f, 0.0f));
                const vec4 av = mul_m4(u.view, aw);
                const vec3 axis = normalize3(mk3(av.x, av.y, av.z));
                const float radius = max_scale * m.radius;
                const float cutoff = (float)m.cone_cutoff / 127.0f;
                accept = accept && !(dot3(cc, axis) >= cutoff * len3(cc) + radius);
                accept = accept && (cc.z * -u.frustum[1] + cc.x * u.frustum[0] < radius);
                accept = accept && (cc.z * -u.frustum[3] + cc.y * u.frustum[2] < radius);
                accept = accept && (cc.z + radius > -u.near_plane) && (cc.z + radius < u.far_plane);
                emit = accept;
                if (PHASE != kPhaseSingle) {
                    // one bit per meshlet instance: accepted by last frame's late phase
                    const uint32_t bit_index = instance_first + lane;
                    uint32_t* word = &fv.meshlet_bits[bit_index >> 5];
                    const uint32_t bit = 1u << (bit_index & 31u);
                    const bool was = (*word & bit) != 0u;
                    if (PHASE == kPhaseEarly) {
                        emit = accept = accept && was;
                    } else {
                        if (accept && !masked) accept = !sphere_occluded(fv, cc, radius);
                        if (!masked && accept != was) { // steady state: no atomics at all
                            if (accept) atomicOr(word, bit);
                            else atomicAnd(word, ~bit);
                        }
                        // already drawn by the early phase: its draw was visible and its bit was set
                        emit = accept && !((cmd_flags & kCmdLateVisible) && was && !masked);
                    }
                }
                // Sort-last tile flags for what k_raster's lanes will scan of this meshlet: every tile under the screen box of
                // its bounding sphere (all its vertices are inside, so are the pixels of a triangle whose three corners need
                // no clipping; the other triangles go to kernels that mark for themselves).  Here and not in k_raster, where
                // the box of the 64 transformed vertices costs four wave-wide reductions per meshlet (a fifth of the kernel);
                // a meshlet whose sphere reaches the near plane or covers more than nine tiles is left to that (bit 30 of its
                // record's counts).
                if (fv.touched && emit) {
                    int ix0, ix1, iy0, iy1;
                    float padded = radius;
                    bool marked = false;
                    if (sphere_screen_box(fv, cc, padded, ix0, ix1, iy0, iy1)) {
                        marked = true;
                        if (ix0 <= ix1 && iy0 <= iy1) {
                            const uint32_t tx0 = (uint32_t)ix0 / kTouchTile, tx1 = (uint32_t)ix1 / kTouchTile, ty0 = (uint32_t)iy0 / kTouchTile, ty1 = (uint32_t)iy1 / kTouchTile;
                            marked = (tx1 - tx0 + 1u) * (ty1 - ty0 + 1u) <= 9u; // a lane's worth; a meshlet that fills the screen has a wave there
                            if (marked)
                                for (uint32_t ty = ty0; ty <= ty1; ++ty)
                                    for (uint32_t tx = tx0; tx <= tx1; ++tx) fv.touched[ty * fv.touched_x + tx] = 1u;
                        }
                    }
                    if (!marked) m_counts |= 0x40000000u;
                }
            }
        }
        accept = emit;
        const unsigned long long mask = __ballot(accept);
        const uint32_t n_acc = (uint32_t)__popcll(mask);
        if (lane == 0) { wave_count[wave] = n_acc; wave_masked[wave] = masked; }
        __syncthreads();
        if (threadIdx.x < 2) { // thread 0: opaque meshlets, thread 1: alpha-masked meshlets
            uint32_t total = 0;
            for (int w = 0; w < 16; ++w) total += wave_masked[w] == threadIdx.x ? wave_count[w] : 0u;
            uint32_t* head = threadIdx.x ? &fv.counters->vis_mask_count : &fv.counters->vis_count;
            wg_base[threadIdx.x] = total ? atomicAdd(head, total) : 0u;
        }
        __syncthreads();
        if (accept) {
            uint32_t rank = wg_base[masked] + ballot_rank(mask);
            for (uint32_t w = 0; w < wave; ++w) rank += wave_masked[w] == masked ? wave_count[w] : 0u;
            // the opaque list grows from the front of the buffer, the masked list from its back
            const uint32_t slot = masked ? fv.visible_capacity - 1u - rank : rank;
            if (rank < fv.visible_capacity) {
                VisibleMeshlet v;
                v.draw_id = draw_id;
                v.meshlet_index = first + lane;
                v.instance = instance_first + lane;
                v.model_index = model_index;
                v.vertex_offset = m_vertex_offset;
                v.triangle_offset = m_triangle_offset;
                v.counts = m_counts;
                v.mesh_vertex_offset = mesh_vertex_offset;
                fv.visible[slot] = v;
            }
        }
        __syncthreads(); // wave_count is rewritten next iteration
        if (PHASE != kPhaseEarly) { // the late phase tests everything the early one did
            st_tested += lane == 0 ? count : 0u;
            st_tris += tri_count;
        }
        st_visible += lane == 0 ? n_acc : 0u;
    }
    if (fv.collect_stats) {
        __shared__ uint32_t wg_stats[3];
        if (threadIdx.x < 3) wg_stats[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t a = wave_sum(st_tested), b = wave_sum(st_visible), c = wave_sum(st_tris);
        if (lane == 0) {
            atomicAdd(&wg_stats[0], a);
            atomicAdd(&wg_stats[1], b);
            atomicAdd(&wg_stats[2], c);
        }
        __syncthreads();
        if (threadIdx.x < 3) stat_store(fv, fv.grid.slot_cull + blockIdx.x * kCullSlots + threadIdx.x, wg_stats[threadIdx.x]);
    }
}

// ------------------------------------------------------------------------------
// Rasteriser
// ------------------------------------------------------------------------------

// `d`: the draw, as a plain reference or through uniform_ref where the whole wave works on one draw; the material and
// the image descriptor behind it are then scalar loads as well.
template <typename Draw>
RDR_DEV AlphaTest make_alpha_test(const SceneView& sv, const Draw& d) {
    AlphaTest at;
    at.enabled = d.alpha_mode == 1u;
    at.has_tex = false;
    at.image = 0; at.uv_set = 0;
    at.sx = at.sy = at.ox = at.oy = 0.0f;
    at.factor_a = 0.0f; at.cutoff = 0.0f; // material_index past the end reads zeros
    if (at.enabled && d.material_index < sv.material_count) {
        const auto* mat = &uniform_ref(&sv.materials[d.material_index]);
        at.factor_a = mat->base_color_factor[3];
        at.cutoff = mat->emissive_factor_alpha_cutoff[3];
        const uint32_t tex = mat->albedo_texture;
        if (tex != RDR_NO_TEXTURE && tex < sv.image_count && uniform_ref(&sv.images[tex]).mip_count == 0u) {
            at.factor_a = 0.0f; // an image without data samples as zero (as in the shading pass): alpha = 0 * factor
        } else if (tex != RDR_NO_TEXTURE && tex < sv.image_count) {
            at.has_tex = true;
            at.image = tex;
            at.uv_set = mat->albedo_texture_uv == 0u ? 0u : 1u;
            at.sx = mat->albedo_texture_transform[0]; at.sy = mat->albedo_texture_transform[1];
            at.ox = mat->albedo_texture_transform[2]; at.oy = mat->albedo_texture_transform[3];
        }
    }
    return at;
}

struct SnapVert {
    int32_t X, Y; // 24.8 framebuffer position, y down
    float z;      // z/w
    float inv_w;
    float u0, v0, u1, v1;
};

// rw = 1 / clip.w, correctly rounded: the perspective division of a vertex is one reciprocal and products (GLSL leaves
// the rounding of x / w to the implementation, and shader compilers lower a division by a shared scalar this way); every
// quotient by w of the geometry stage (NDC position, depth, the mesh shader's screen position) uses the same rw.
RDR_DEV void project_xy(float cx, float cy, float rw, float half_w, float half_h, int32_t& X, int32_t& Y) {
    const float ndcx = cx * rw, ndcy = cy * rw;
    const float xf = half_w * ndcx + half_w;
    const float yf = -half_h * ndcy + half_h; // viewport {y = H, height = -H}
    X = (int32_t)rintf(xf * 256.0f);
    Y = (int32_t)rintf(yf * 256.0f);
}

RDR_DEV bool vert_needs_clip(float x, float y, float z, float w) {
    const float g = kGuardBand * w;
    return !(w >= z) || !(fabsf(x) <= g) || !(fabsf(y) <= g);
}

// Sort-last tile flags (FrameView::touched): the tile of a pixel / every tile of a pixel box that is about to be
// scanned.  Plain stores of the same value: which lane or wave wins does not matter.
RDR_DEV void touch_pixel(const FrameView& fv, int px, int py) {
    if (fv.touched) fv.touched[(uint32_t)py / kTouchTile * fv.touched_x + (uint32_t)px / kTouchTile] = 1u;
}
// What k_raster's lanes scan is marked per meshlet: by k_meshlet_cull from the bounding sphere where that says enough,
// else here, outside the triangle loop: every tile under the pixel box of the meshlet's vertices that need no
// clipping.  Triangles scanned by a lane have three such vertices, so their pixels lie inside; the others go to
// kernels that mark for themselves.  (X, Y): the lane's snapped 24.8 vertex, `valid`: it has one.
// Wave-wide minimum as a scalar: four butterfly steps inside each row of 16 lanes on the data-parallel
// path (no LDS round trips, one temporary), then the four rows through scalar registers.
RDR_DEV int32_t wave_min_scalar(int32_t v) {
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false)); // row_half_mirror
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false)); // row_mirror
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

RDR_DEV void touch_meshlet(const FrameView& fv, bool valid, const int32_t* lds_x, const int32_t* lds_y, uint32_t lane) {
    const int32_t big = 0x7FFFFFFF;
    // one coordinate at a time: few registers live at once
    int32_t v = lds_x[lane];
    const int32_t x0 = wave_min_scalar(valid ? v : big), x1 = -wave_min_scalar(valid ? -v : big);
    if (x0 > x1) return; // no usable vertex
    v = lds_y[lane];
    const int32_t y0 = wave_min_scalar(valid ? v : big), y1 = -wave_min_scalar(valid ? -v : big);
    // a pixel box never reaches past floor(min / 256) .. floor(max / 256) (setup_triangle rounds inwards)
    const int32_t px0 = max(x0 >> 8, 0), py0 = max(y0 >> 8, 0);
    const int32_t px1 = min(x1 >> 8, (int32_t)fv.width - 1), py1 = min(y1 >> 8, (int32_t)fv.height - 1);
    if (px0 > px1 || py0 > py1) return;
    const uint32_t tx1 = (uint32_t)px1 / kTouchTile, ty1 = (uint32_t)py1 / kTouchTile;
    for (uint32_t ty = (uint32_t)py0 / kTouchTile; ty <= ty1; ++ty) // rows: the same in every lane; columns: a lane each
        for (uint32_t tx = (uint32_t)px0 / kTouchTile + lane; tx <= tx1; tx += 64u) fv.touched[ty * fv.touched_x + tx] = 1u;
}

RDR_DEV void emit_fragment(const FrameView& fv, int px, int py, float z, uint32_t id) {
    z = clampf(z, 0.0f, 1.0f);
    const unsigned long long key =
        ((unsigned long long)(0x7FFFFFFFu - __float_as_uint(z)) << 32) | (unsigned long long)id;
    unsigned long long* p = &fv.vis[(size_t)py * fv.width + px];
    // The result is not used: the atomic is issued and forgotten, so a lane that produces
    // several fragments never waits for a round trip to the L2 (a pre-test load would cost
    // one per fragment).
    __hip_atomic_fetch_min(p, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// (float)v for an edge-function value, correctly rounded like the plain conversion: below 2^53
// in magnitude (twice the area of any triangle inside the guard band is) the value is exact in
// a double, assembled from its two halves, and the double rounds to float once.
RDR_DEV float edge_to_float(long long v) {
    return (float)fma((double)(int)(v >> 32), 4294967296.0, (double)(uint32_t)v);
}
RDR_DEV float edge_value_to_float(long long v) { return edge_to_float(v); }
RDR_DEV float edge_value_to_float(int32_t v) { return (float)v; }

// Perspective-correct uv at the sample with edge values (e0, e1, e2).
template <typename I>
RDR_DEV void persp_uv(const SnapVert& a, const SnapVert& b, const SnapVert& c, I e0, I e1, I e2, uint32_t set,
                      float& u, float& v) {
    const float b0 = edge_value_to_float(e0) * a.inv_w, b1 = edge_value_to_float(e1) * b.inv_w, b2 = edge_value_to_float(e2) * c.inv_w;
    const float sum = (b0 + b1) + b2;
    if (set == 0u) {
        u = ((b0 * a.u0 + b1 * b.u0) + b2 * c.u0) / sum;
        v = ((b0 * a.v0 + b1 * b.v0) + b2 * c.v0) / sum;
    } else {
        u = ((b0 * a.u1 + b1 * b.u1) + b2 * c.u1) / sum;
        v = ((b0 * a.v1 + b1 * b.v1) + b2 * c.v1) / sum;
    }
}

// Triangle set up for scanning: oriented so that the interior has positive edge
// values, with the top-left rule folded into a bias.
template <typename I>
struct TriSetup {
    SnapVert v0, v1, v2;
    I dx01, dy01, dx12, dy12, dx20, dy20;
    int32_t bias01, bias12, bias20; // 0 for top-left edges, -1 otherwise
    float inv_area, dz1, dz2;
    int px0, py0, px1, py1;
    bool valid;
};

template <typename I>
RDR_DEV TriSetup<I> setup_triangle(const FrameView& fv, SnapVert v0, SnapVert v1, SnapVert v2, bool cull_front) {
    TriSetup<I> t;
    t.valid = false;
    // twice the signed area; in the 32-bit instantiation the caller has bounded the extents (kSmallExtent), so
    // the products fit 32 bits and (float)A rounds the same integer once either way
    static_assert(sizeof(I) == 8 || (long long)kSmallExtent * kSmallExtent * 2 < (1ll << 31), "32-bit set-up: area overflows");
    I A = (I)(v1.X - v0.X) * (I)(v2.Y - v0.Y) - (I)(v1.Y - v0.Y) * (I)(v2.X - v0.X);
    if (A == 0) return t;
    if (A > 0) {
        // clockwise on the y-down screen: front-facing under frontFace = CLOCKWISE
        if (cull_front) return t;
    } else {
        const SnapVert tmp = v1; v1 = v2; v2 = tmp;
        A = -A;
    }
    const int32_t minX = min(v0.X, min(v1.X, v2.X)), maxX = max(v0.X, max(v1.X, v2.X));
    const int32_t minY = min(v0.Y, min(v1.Y, v2.Y)), maxY = max(v0.Y, max(v1.Y, v2.Y));
    t.px0 = max((minX - 128 + 255) >> 8, 0);
    t.px1 = min((maxX - 128) >> 8, (int)fv.width - 1);
    t.py0 = max((minY - 128 + 255) >> 8, 0);
    t.py1 = min((maxY - 128) >> 8, (int)fv.height - 1);
    if (t.px0 > t.px1 || t.py0 > t.py1) return t;
    t.v0 = v0; t.v1 = v1; t.v2 = v2;
    t.dx01 = (I)(v1.X - v0.X); t.dy01 = (I)(v1.Y - v0.Y);
    t.dx12 = (I)(v2.X - v1.X); t.dy12 = (I)(v2.Y - v1.Y);
    t.dx20 = (I)(v0.X - v2.X); t.dy20 = (I)(v0.Y - v2.Y);
    t.bias01 = (t.dy01 < 0 || (t.dy01 == 0 && t.dx01 > 0)) ? 0 : -1;
    t.bias12 = (t.dy12 < 0 || (t.dy12 == 0 && t.dx12 > 0)) ? 0 : -1;
    t.bias20 = (t.dy20 < 0 || (t.dy20 == 0 && t.dx20 > 0)) ? 0 : -1;
    t.inv_area = 1.0f / (float)A;
    t.dz1 = v1.z - v0.z;
    t.dz2 = v2.z - v0.z;
    t.valid = true;
    return t;
}

// Edge values at the centre of pixel (px, py).
template <typename I>
RDR_DEV void edges_at(const TriSetup<I>& t, int px, int py, I& e01, I& e12, I& e20) {
    const I Px = (I)px * 256 + 128, Py = (I)py * 256 + 128;
    e01 = t.dx01 * (Py - (I)t.v0.Y) - t.dy01 * (Px - (I)t.v0.X);
    e12 = t.dx12 * (Py - (I)t.v1.Y) - t.dy12 * (Px - (I)t.v1.X);
    e20 = t.dx20 * (Py - (I)t.v2.Y) - t.dy20 * (Px - (I)t.v2.X);
}

// Coverage + alpha + depth + atomic for one sample. Returns 1 when a fragment was produced.
template <typename I, bool MASK>
RDR_DEV uint32_t shade_sample(const SceneView& sv, const FrameView& fv, const TriSetup<I>& t, const AlphaTest& at,
                              int px, int py, I e01, I e12, I e20, uint32_t id) {
    if (((e01 + (I)t.bias01) | (e12 + (I)t.bias12) | (e20 + (I)t.bias20)) < 0) return 0;
    if (MASK && at.enabled) {
        float alpha = 1.0f;
        if (at.has_tex) {
            float u, v, ux, vx, uy, vy;
            persp_uv<I>(t.v0, t.v1, t.v2, e12, e20, e01, at.uv_set, u, v);
            persp_uv<I>(t.v0, t.v1, t.v2, e12 - t.dy12 * 256, e20 - t.dy20 * 256, e01 - t.dy01 * 256, at.uv_set, ux, vx);
            persp_uv<I>(t.v0, t.v1, t.v2, e12 + t.dx12 * 256, e20 + t.dx20 * 256, e01 + t.dx01 * 256, at.uv_set, uy, vy);
            u = u * at.sx + at.ox; v = v * at.sy + at.oy;
            ux = ux * at.sx + at.ox; vx = vx * at.sy + at.oy;
            uy = uy * at.sx + at.ox; vy = vy * at.sy + at.oy;
            alpha = image_sample_grad(uniform_ref(&sv.images[at.image]), sv.texels, sv.srgb_lut, u, v, ux - u, vx - v, uy - u, vy - v).w;
        }
        alpha = alpha * at.factor_a;
        if (alpha < at.cutoff) return 0;
    }
    const float l1 = (float)e20 * t.inv_area, l2 = (float)e01 * t.inv_area;
    const float z = fmaf(l2, t.dz2, fmaf(l1, t.dz1, t.v0.z));
    emit_fragment(fv, px, py, z, id);
    return 1;
}

// One lane scans the whole bounding box.
template <typename I, bool MASK>
RDR_DEV uint32_t scan_lane(const SceneView& sv, const FrameView& fv, const TriSetup<I>& t, const AlphaTest& at, uint32_t id) {
    uint32_t frags = 0;
    I r01, r12, r20;
    edges_at<I>(t, t.px0, t.py0, r01, r12, r20);
    for (int py = t.py0; py <= t.py1; ++py) {
        I e01 = r01, e12 = r12, e20 = r20;
        for (int px = t.px0; px <= t.px1; ++px) {
            frags += shade_sample<I, MASK>(sv, fv, t, at, px, py, e01, e12, e20, id);
            e01 -= t.dy01 * 256; e12 -= t.dy12 * 256; e20 -= t.dy20 * 256;
        }
        r01 += t.dx01 * 256; r12 += t.dx12 * 256; r20 += t.dx20 * 256;
    }
    return frags;
}

// Orient, face-cull and box a snapped triangle. Returns false when nothing can be covered.
RDR_DEV bool make_setup(const FrameView& fv, SnapVert v0, SnapVert v1, SnapVert v2, bool cull_front, SetupTri& t, SetupMask& mk) {
    long long A = (long long)(v1.X - v0.X) * (v2.Y - v0.Y) - (long long)(v1.Y - v0.Y) * (v2.X - v0.X);
    if (A == 0) return false;
    if (A > 0) {
        if (cull_front) return false; // clockwise on the y-down screen: front-facing, culled in the opaque pipeline
    } else {
        const SnapVert tmp = v1; v1 = v2; v2 = tmp;
        A = -A;
    }
    const int32_t minX = min(v0.X, min(v1.X, v2.X)), maxX = max(v0.X, max(v1.X, v2.X));
    const int32_t minY = min(v0.Y, min(v1.Y, v2.Y)), maxY = max(v0.Y, max(v1.Y, v2.Y));
    const int px0 = max((minX - 128 + 255) >> 8, 0), px1 = min((maxX - 128) >> 8, (int)fv.width - 1);
    const int py0 = max((minY - 128 + 255) >> 8, 0), py1 = min((maxY - 128) >> 8, (int)fv.height - 1);
    if (px0 > px1 || py0 > py1) return false;
    t.X0 = v0.X; t.Y0 = v0.Y; t.X1 = v1.X; t.Y1 = v1.Y; t.X2 = v2.X; t.Y2 = v2.Y;
    t.z0 = v0.z; t.dz1 = v1.z - v0.z; t.dz2 = v2.z - v0.z;
    t.inv_area = 1.0f / (float)A;
    t.px0 = (uint16_t)px0; t.py0 = (uint16_t)py0; t.px1 = (uint16_t)px1; t.py1 = (uint16_t)py1;
    t._pad = 0;
    mk.inv_w[0] = v0.inv_w; mk.inv_w[1] = v1.inv_w; mk.inv_w[2] = v2.inv_w;
    const SnapVert* vs[3] = {&v0, &v1, &v2};
#pragma unroll
    for (int k = 0; k < 3; ++k) { mk.uv[k][0] = vs[k]->u0; mk.uv[k][1] = vs[k]->v0; mk.uv[k][2] = vs[k]->u1; mk.uv[k][3] = vs[k]->v1; }
    mk._pad = 0;
    return true;
}

// The whole wave scans the bounding box, 8x8 pixels per step.
template <typename I, bool MASK>
RDR_DEV uint32_t scan_wave(const SceneView& sv, const FrameView& fv, const TriSetup<I>& t, const AlphaTest& at, uint32_t id, uint32_t lane) {
    const int bw = t.px1 - t.px0 + 1, bh = t.py1 - t.py0 + 1;
    const int stamps_x = (bw + 7) >> 3, stamps = stamps_x * ((bh + 7) >> 3);
    uint32_t frags = 0;
    for (int sidx = 0; sidx < stamps; ++sidx) {
        const int px = t.px0 + (sidx % stamps_x) * 8 + (int)(lane & 7u);
        const int py = t.py0 + (sidx / stamps_x) * 8 + (int)(lane >> 3);
        if (px <= t.px1 && py <= t.py1) {
            touch_pixel(fv, px, py);
            I e01, e12, e20;
            edges_at<I>(t, px, py, e01, e12, e20);
            frags += shade_sample<I, MASK>(sv, fv, t, at, px, py, e01, e12, e20, id);
        }
    }
    return frags;
}

// Per-wave LDS: the transformed vertices of one meshlet.
// (One 16-byte element per vertex and phase instead of these planes — a corner in one LDS read — is slower: field
// 0.144 -> 0.154 ms; 16-byte gathers are served 16 lanes at a time and collide more.  So is keeping a vertex in the
// registers of its lane and fetching it with ds_bpermute, which needs no LDS writes and no waits between the phases:
// 0.137 -> 0.146 ms.  Wave-level work counters in scalar registers (four vector registers fewer) and compiler-only
// barriers between the phases (one wave per workgroup needs no more): 0.136 -> 0.138 ms, Field-XL 1.95 -> 2.00 ms.)
struct MeshletLds {
    float cx[64], cy[64], cz[64], cw[64]; // clip space
    float z[64];                          // z / w
    float sx[64], sy[64];                 // the mesh shader's y-up screen position
    int32_t X[64], Y[64];                 // snapped framebuffer position
    uint32_t clip[64];                    // vertex is outside the near plane / guard band
    uint32_t queue[128];                  // triangles that survived the cull tests: vertex bytes | index << 24
};

// What the alpha test needs per vertex besides the position: 1 / w and the uv set of the material's
// albedo texture, computed once per vertex (the opaque instantiation carries none of it).
template <bool MASK>
struct MaskVertexLds {};
template <>
struct MaskVertexLds<true> {
    float inv_w[64], u[64], v[64];
    FragTri rec[64]; // a lane's record for k_alpha_frag, staged here so that it does not sit in registers
};

// What one meshlet's pass reads from memory, fetched while the meshlet before it is being culled: a vertex position
// per lane, the lane's two triangles (three index bytes each, in the low 24 bits of an unaligned dword), the draw's
// model matrix (element k in lane k: a scalar load would be waited for by the very next LDS read, the two share a
// counter), and for alpha-tested draws both uv sets of the lane's vertex.
template <bool MASK>
struct MeshletOperands {
    float px, py, pz;
    uint32_t tri_a, tri_b;
    float model_lane;
    float4 uv;
};

// One record of the visible list at a wave-uniform index, through the scalar unit.
RDR_DEV VisibleMeshlet load_visible(const VisibleMeshlet* p) {
    const uint32_t* w = reinterpret_cast<const uint32_t*>(p);
    VisibleMeshlet r;
    r.draw_id = load_uniform(w + 0); r.meshlet_index = load_uniform(w + 1); r.instance = load_uniform(w + 2); r.model_index = load_uniform(w + 3);
    r.vertex_offset = load_uniform(w + 4); r.triangle_offset = load_uniform(w + 5); r.counts = load_uniform(w + 6); r.mesh_vertex_offset = load_uniform(w + 7);
    return r;
}

template <bool MASK>
RDR_DEV void fetch_meshlet(const SceneView& sv, const VisibleMeshlet& r, uint32_t lane, MeshletOperands<MASK>& o) {
    const uint32_t nv = r.counts & 0xFFu, nt = (r.counts >> 16) & 0xFFu;
    o.px = o.py = o.pz = 0.0f;
    o.uv = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (lane < nv) {
        if (sv.mpos_x) { // consecutive floats per plane, addressed by the record alone
            const uint32_t s = r.vertex_offset + lane;
            o.px = sv.mpos_x[s]; o.py = sv.mpos_y[s]; o.pz = sv.mpos_z[s];
            if (MASK) o.uv = sv.muv[s];
        } else {
            const uint32_t vi = sv.meshlet_vertices[r.vertex_offset + lane] + r.mesh_vertex_offset;
            o.px = sv.pos_x[vi]; o.py = sv.pos_y[vi]; o.pz = sv.pos_z[vi];
            if (MASK) o.uv = sv.uv_sets[vi];
        }
    }
    const uint8_t* tp = sv.meshlet_triangles + r.triangle_offset; // the buffer ends in 16 spare bytes
    o.tri_a = lane < nt ? load_u32_unaligned(tp + lane * 3u) : 0u;
    o.tri_b = lane + 64u < nt ? load_u32_unaligned(tp + (lane + 64u) * 3u) : 0u;
    o.model_lane = lane < 16u ? sv.transforms[(size_t)r.model_index * 16 + lane] : 0.0f;
}

// MASK = false: meshlets of opaque draws (front of the visible list); MASK = true:
// meshlets of alpha-tested draws (back of the list).  Two instantiations so that the
// opaque one carries no uv / texture state in registers.
// TOUCH: keep the sort-last tile flags for the meshlets k_meshlet_cull left to this kernel (an instantiation of its
// own: even never taken, the marking costs the plain one 2 %).
template <bool MASK, bool TOUCH>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MASK ? 4 : 5))) void k_raster(SceneView sv, FrameView fv) {
    __shared__ MeshletLds lds;
    __shared__ MaskVertexLds<MASK> mlds;
    const uint32_t lane = threadIdx.x;
    const rdr_ubo& u = fv.ubo;
    const uint32_t n_opaque = min(fv.counters->vis_count, fv.visible_capacity);
    const uint32_t n_masked = min(fv.counters->vis_mask_count, fv.visible_capacity - n_opaque);
    const uint32_t vis_count = MASK ? n_masked : n_opaque;
    const float half_w = (float)fv.width * 0.5f, half_h = (float)fv.height * 0.5f;

    uint32_t st_tested = 0, st_raster = 0, st_big = 0, st_frags = 0, st_verts = 0;
    struct { uint32_t vertex_offset, triangle_offset, vertex_count, triangle_count; } m;
    // Software pipeline over the wave's meshlets.  Every operand of a meshlet (MeshletOperands) is requested while
    // the meshlet before it runs its cull tests and is in registers before that meshlet's fragments are issued:
    // a meshlet's pass then starts without a memory round trip, and none of its waits sits behind the atomics of
    // the meshlet before (vector memory operations complete in order, so a wait placed after them would also wait
    // for their acknowledgements).  The record itself comes a whole meshlet ahead through the scalar unit.
    VisibleMeshlet vm_next{};
    MeshletOperands<MASK> next{};
    auto record = [&](uint32_t i) { return load_visible(fv.visible + (MASK ? fv.visible_capacity - 1u - i : i)); };
    // the loads of `next` have completed once their results have passed through here
    // Which meshlets: the first two by the wave's place in the grid, every further one by a ticket drawn a meshlet
    // ahead (FrameCounters::raster_ticket), so that waves with light meshlets take more of them.
    const uint32_t n_segments = min(kRasterSegments, gridDim.x), segment = blockIdx.x % n_segments;
    // (the counter's address through a lane offset the compiler cannot see is zero: with an address it knows to be
    // uniform it rewrites the atomic into a form that reads the result back at once, and the wave would wait for the
    // ticket, and for every fragment of the meshlet before, at the top of each pass)
    uint32_t lane_zero = 0;
    asm volatile("" : "+v"(lane_zero));
    uint32_t* const ticket_counter = &fv.counters->raster_ticket[MASK ? 1 : 0][segment * 32u] + lane_zero;
    uint32_t ticket = 0; // lane 0's is the one drawn
    auto settle = [&]() {
        asm volatile("" : "+v"(next.px), "+v"(next.py), "+v"(next.pz), "+v"(next.tri_a), "+v"(next.tri_b), "+v"(next.model_lane), "+v"(ticket));
        if constexpr (MASK) asm volatile("" : "+v"(next.uv.x), "+v"(next.uv.y), "+v"(next.uv.z), "+v"(next.uv.w));
    };
    if (blockIdx.x < vis_count) {
        vm_next = record(blockIdx.x);
        fetch_meshlet<MASK>(sv, vm_next, lane, next);
    }
    settle();
    uint32_t mi_next = blockIdx.x + gridDim.x;
    for (uint32_t mi = blockIdx.x; mi < vis_count;) {
        const VisibleMeshlet vm = vm_next;
        const MeshletOperands<MASK> op = next;
        const bool has_next = mi_next < vis_count;
        m.vertex_offset = vm.vertex_offset; m.triangle_offset = vm.triangle_offset;
        m.vertex_count = vm.counts & 0xFFu; m.triangle_count = (vm.counts >> 16) & 0xFFu;
        const uint32_t mesh_vertex_offset = vm.mesh_vertex_offset;
        AlphaTest at;
        at.enabled = false; at.has_tex = false;
        if (MASK) at = make_alpha_test(sv, uniform_ref(&sv.draws[vm.draw_id]));
        const bool cull_front = !MASK;

        __syncthreads(); // previous meshlet's readers are done
        // (after the barrier: the barrier waits for the scalar unit's loads together with the LDS operations)
        if (has_next) {
            vm_next = record(mi_next);
            if (lane == 0u) ticket = atomicAdd(ticket_counter, 1u); // for the meshlet after that one
        }
        float model[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) model[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, op.model_lane), k));
        bool vert_behind = false, vert_clipped = false; // w <= 0; outside the near plane / guard band
        if (lane < m.vertex_count) {
            st_verts++;
            const vec4 world = mul_m4(model, mk4(op.px, op.py, op.pz, 1.0f));
            const vec4 clip = mul_m4(u.view_proj, world);
            lds.cx[lane] = clip.x; lds.cy[lane] = clip.y; lds.cz[lane] = clip.z; lds.cw[lane] = clip.w;
            vert_behind = !(clip.w > 0.0f);
            const float rw = 1.0f / clip.w;
            lds.z[lane] = clip.z * rw;
            lds.sx[lane] = (clip.x * rw * 0.5f + 0.5f) * u.screen_width;
            lds.sy[lane] = (clip.y * rw * 0.5f + 0.5f) * u.screen_height;
            const bool nc = vert_needs_clip(clip.x, clip.y, clip.z, clip.w);
            int32_t X = 0, Y = 0;
            if (!nc) project_xy(clip.x, clip.y, rw, half_w, half_h, X, Y);
            lds.X[lane] = X; lds.Y[lane] = Y;
            // what the clipper's five planes say about the vertex (plane_dist < 0, the same expressions), bit 5: it needs clipping
            uint32_t outside = 0u;
            if (nc) {
                const float g = kGuardBand * clip.w;
                outside = 32u | (clip.w - clip.z < 0.0f ? 1u : 0u) | (g - clip.x < 0.0f ? 2u : 0u) | (g + clip.x < 0.0f ? 4u : 0u) |
                          (g - clip.y < 0.0f ? 8u : 0u) | (g + clip.y < 0.0f ? 16u : 0u);
            }
            lds.clip[lane] = outside;
            vert_clipped = nc;
            if constexpr (MASK) {
                if (at.has_tex) { // perspective-correct uv for the alpha test
                    mlds.inv_w[lane] = rw;
                    mlds.u[lane] = at.uv_set ? op.uv.z : op.uv.x;
                    mlds.v[lane] = at.uv_set ? op.uv.w : op.uv.y;
                }
            }
        }
        // Nearly every meshlet has all its vertices in front of the camera and inside the guard band: the triangle
        // phases then skip the per-corner reads that would tell them so (wave-uniform branches).
        const bool any_behind = __ballot(vert_behind) != 0ull, any_clipped = __ballot(vert_clipped) != 0ull;
        __syncthreads();
        // sort-last tile flags: k_meshlet_cull has marked the tiles under the meshlet's bounding sphere, unless that
        // reaches the near plane or is large on screen (bit 30 of the record's counts: the same in the whole wave)
        if (TOUCH && (vm.counts & 0x40000000u)) touch_meshlet(fv, lane < m.vertex_count && lds.clip[lane] == 0u, lds.X, lds.Y, lane);
        if (has_next) fetch_meshlet<MASK>(sv, vm_next, lane, next); // in flight during the cull tests

        // Triangle phase 1: the mesh shader's cull tests, two triangles per lane; the survivors (about
        // four in ten on a closed mesh) are packed into an LDS queue so that phase 2 runs on full lanes.
        // (Testing the lane's two triangles together, so that their LDS reads overlap, gains nothing: 0.138 -> 0.139 ms
        // on the field when it fits the registers, 0.144 -> 0.154 ms when it spills.)
        uint32_t n_queued = 0;
        for (uint32_t t0 = 0; t0 < m.triangle_count; t0 += 64) {
            const uint32_t ti = t0 + lane;
            bool keep = false;
            uint32_t packed = 0;
            if (ti < m.triangle_count) {
                const uint32_t tri_bytes = (t0 ? op.tri_b : op.tri_a) & 0xFFFFFFu;
                const uint32_t ia = tri_bytes & 255u, ib = (tri_bytes >> 8) & 255u, ic = tri_bytes >> 16;
                st_tested++;
                const float ax = lds.sx[ia], ay = lds.sy[ia], bx = lds.sx[ib], by = lds.sy[ib];
                const float cx = lds.sx[ic], cy = lds.sy[ic];
                const float ebax = bx - ax, ebay = by - ay, ecax = cx - ax, ecay = cy - ay;
                bool culled = ebax * ecay <= ebay * ecax;
                const float bminx = fminf(ax, fminf(bx, cx)), bminy = fminf(ay, fminf(by, cy));
                const float bmaxx = fmaxf(ax, fmaxf(bx, cx)), bmaxy = fmaxf(ay, fmaxf(by, cy));
                const float sbprec = 1.0f / 256.0f;
                culled = culled || (rintf(bminx - sbprec) == rintf(bmaxx) || rintf(bminy) == rintf(bmaxy + sbprec));
                if (any_behind) culled = culled && (lds.cw[ia] > 0.0f && lds.cw[ib] > 0.0f && lds.cw[ic] > 0.0f);
                keep = !culled;
                packed = tri_bytes | (ti << 24);
            }
            const unsigned long long km = __ballot(keep);
            if (keep) {
                st_raster++;
                lds.queue[n_queued + ballot_rank(km)] = packed;
            }
            n_queued += (uint32_t)__popcll(km);
        }
        __syncthreads(); // the queue is complete
        // the next meshlet's operands have arrived by now; taking them into registers here, before this meshlet's
        // atomics, keeps the wait for them from covering those atomics as well
        settle();
        const uint32_t mi_after = 2u * gridDim.x + (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket) * n_segments + segment;
        // Triangle phase 2: set-up and scan of the queued triangles, one per lane
        for (uint32_t q0 = 0; q0 < n_queued; q0 += 64) {
            bool defer = false; // clipped, large or pixel-heavy triangles go to the tile path
            bool mid = false;   // too many pixels for one lane: queued for k_mid_raster
            uint32_t ia = 0, ib = 0, ic = 0, ti = 0;
            uint32_t cmask = 0; // MASK: a lane-sized triangle's covered pixels (bit k: pixel k of its box, row-major)
                                // (what k_alpha_frag needs of the triangle is staged in mlds.rec[lane])
            // snapped vertices of the triangle with meshlet-local vertices (a, b, c), from the wave's LDS
            auto load_snap = [&](uint32_t a, uint32_t b, uint32_t c, SnapVert& s0, SnapVert& s1, SnapVert& s2, bool positions) {
                if (positions) {
                    s0.X = lds.X[a]; s0.Y = lds.Y[a]; s0.z = lds.z[a];
                    s1.X = lds.X[b]; s1.Y = lds.Y[b]; s1.z = lds.z[b];
                    s2.X = lds.X[c]; s2.Y = lds.Y[c]; s2.z = lds.z[c];
                }
                s0.inv_w = s1.inv_w = s2.inv_w = 0.0f;
                s0.u0 = s0.v0 = s0.u1 = s0.v1 = 0.0f;
                s1.u0 = s1.v0 = s1.u1 = s1.v1 = 0.0f;
                s2.u0 = s2.v0 = s2.u1 = s2.v1 = 0.0f;
                if constexpr (MASK) {
                    if (at.has_tex) { // both uv slots carry the set the albedo texture uses: whichever the test picks is right
                        s0.inv_w = mlds.inv_w[a]; s1.inv_w = mlds.inv_w[b]; s2.inv_w = mlds.inv_w[c];
                        s0.u0 = s0.u1 = mlds.u[a]; s0.v0 = s0.v1 = mlds.v[a];
                        s1.u0 = s1.u1 = mlds.u[b]; s1.v0 = s1.v1 = mlds.v[b];
                        s2.u0 = s2.u1 = mlds.u[c]; s2.v0 = s2.v1 = mlds.v[c];
                    }
                }
            };
            if (q0 + lane < n_queued) {
                const uint32_t packed = lds.queue[q0 + lane];
                ia = packed & 255u; ib = (packed >> 8) & 255u; ic = (packed >> 16) & 255u; ti = packed >> 24;
                bool nothing_left = false;
                if (any_clipped) {
                    const uint32_t oa = lds.clip[ia], ob = lds.clip[ib], oc = lds.clip[ic];
                    defer = (oa | ob | oc) != 0u;
                    // all three corners outside one of the clipper's planes: it would leave nothing (geometry beside and
                    // behind the camera, which the culling's spheres let through in bulk), so the triangle ends here
                    nothing_left = (oa & ob & oc & 31u) != 0u;
                    defer = defer && !nothing_left;
                }
                if (!defer && !nothing_left) {
                    // all nine reads at once: one LDS round trip in front of the set-up instead of two
                    const int32_t X0 = lds.X[ia], X1 = lds.X[ib], X2 = lds.X[ic];
                    const int32_t Y0 = lds.Y[ia], Y1 = lds.Y[ib], Y2 = lds.Y[ic];
                    const float z0 = lds.z[ia], z1 = lds.z[ib], z2 = lds.z[ic];
                    const int32_t ex = max(X0, max(X1, X2)) - min(X0, min(X1, X2));
                    const int32_t ey = max(Y0, max(Y1, Y2)) - min(Y0, min(Y1, Y2));
                    defer = ex > kSmallExtent || ey > kSmallExtent;
                    if (!defer) {
                        SnapVert s0, s1, s2;
                        s0.X = X0; s0.Y = Y0; s0.z = z0; s1.X = X1; s1.Y = Y1; s1.z = z1; s2.X = X2; s2.Y = Y2; s2.z = z2;
                        load_snap(ia, ib, ic, s0, s1, s2, false);
                        const TriSetup<int32_t> ts = setup_triangle<int32_t>(fv, s0, s1, s2, cull_front);
                        if (ts.valid) {
                            const int npx = (ts.px1 - ts.px0 + 1) * (ts.py1 - ts.py0 + 1);
                            if constexpr (MASK) {
                                // anything a lane cannot enumerate goes to the set-up kernel of the tile path, which
                                // hands the mid-sized ones on to k_mid_raster: the scan set-up with its uv planes
                                // would cost this kernel a quarter of its occupancy
                                defer = npx > (int)min(fv.lane_max_pixels, kLaneMaxPixels);
                            } else {
                                defer = npx > (int)fv.mid_max_pixels;
                                mid = !defer && npx > (int)fv.lane_max_pixels;
                            }
                            if (!defer && !mid) {
                                if constexpr (MASK) {
                                    int32_t r01, r12, r20;
                                    edges_at<int32_t>(ts, ts.px0, ts.py0, r01, r12, r20);
                                    uint32_t k = 0;
                                    for (int py = ts.py0; py <= ts.py1; ++py) {
                                        int32_t e01 = r01, e12 = r12, e20 = r20;
                                        for (int px = ts.px0; px <= ts.px1; ++px, ++k) {
                                            cmask |= (((e01 + ts.bias01) | (e12 + ts.bias12) | (e20 + ts.bias20)) >= 0) ? (1u << k) : 0u;
                                            e01 -= ts.dy01 * 256; e12 -= ts.dy12 * 256; e20 -= ts.dy20 * 256;
                                        }
                                        r01 += ts.dx01 * 256; r12 += ts.dx12 * 256; r20 += ts.dx20 * 256;
                                    }
                                    FragTri& r = mlds.rec[lane];
                                    r.X[0] = ts.v0.X; r.X[1] = ts.v1.X; r.X[2] = ts.v2.X;
                                    r.Y[0] = ts.v0.Y; r.Y[1] = ts.v1.Y; r.Y[2] = ts.v2.Y;
                                    r.z0 = ts.v0.z; r.dz1 = ts.dz1; r.dz2 = ts.dz2; r.inv_area = ts.inv_area;
                                    r.inv_w[0] = ts.v0.inv_w; r.inv_w[1] = ts.v1.inv_w; r.inv_w[2] = ts.v2.inv_w;
                                    r.u[0] = ts.v0.u0; r.u[1] = ts.v1.u0; r.u[2] = ts.v2.u0; // both uv slots hold the albedo set
                                    r.v[0] = ts.v0.v0; r.v[1] = ts.v1.v0; r.v[2] = ts.v2.v0;
                                    r.origin = (uint32_t)ts.px0 | ((uint32_t)ts.py0 << 16);
                                    r.bw = (uint32_t)(ts.px1 - ts.px0 + 1);
                                    r.id = (vm.instance << 7) | ti;
                                    r.draw_id = vm.draw_id;
                                    r._pad = 0;
                                } else {
                                    st_frags += scan_lane<int32_t, MASK>(sv, fv, ts, at, (vm.instance << 7) | ti);
                                }
                            }
                        }
                    }
                }
            }
            // Alpha-tested small triangles: a covered pixel costs three perspective uv evaluations and a
            // trilinear sample, and a lane's triangle covers anything from 0 to 16 pixels, so scanning here
            // would leave most of the wave waiting for its biggest triangles (and the sampler's registers
            // would halve this kernel's occupancy).  The lanes only find their covered pixels (cheap) and
            // queue (triangle, pixel) pairs; k_alpha_frag works through them with one fragment per lane.
            if constexpr (MASK) {
                const uint32_t n_cov = (uint32_t)__popc(cmask);
                const unsigned long long tm = __ballot(n_cov != 0u);
                if (tm) {
                    const uint32_t seg = blockIdx.x % kFragSegments;
                    const uint32_t cap_t = fv.frag_tri_capacity / kFragSegments, cap_p = fv.frag_px_capacity / kFragSegments;
                    const uint32_t incl = wave_scan_inclusive(n_cov, lane);
                    const uint32_t total = __shfl(incl, 63, 64);
                    const int leader = __ffsll((long long)tm) - 1;
                    unsigned long long base = 0;
                    if ((int)lane == leader)
                        base = atomicAdd(&fv.counters->frag_seg[seg * 16u], ((unsigned long long)__popcll(tm) << 32) | total);
                    base = __shfl(base, leader, 64);
                    const uint32_t base_t = (uint32_t)(base >> 32), base_p = (uint32_t)base;
                    if (n_cov) {
                        const uint32_t slot_t = base_t + ballot_rank(tm), first_p = base_p + incl - n_cov;
                        if (slot_t < cap_t && first_p + n_cov <= cap_p) {
                            const uint32_t gt = seg * cap_t + slot_t;
                            const uint4* src = reinterpret_cast<const uint4*>(&mlds.rec[lane]);
                            uint4* out = reinterpret_cast<uint4*>(&fv.frag_tri[gt]);
#pragma unroll
                            for (int c = 0; c < (int)(sizeof(FragTri) / 16); ++c) out[c] = src[c];
                            uint32_t* dst = fv.frag_px + (size_t)seg * cap_p + first_p;
                            for (uint32_t m2 = cmask; m2; m2 &= m2 - 1u) *dst++ = (gt << 4) | (uint32_t)(__ffs((int)m2) - 1);
                        } else {
                            atomicAdd(&fv.counters->frag_overflow, 1u);
                        }
                    }
                }
            }
            // mid-sized triangles: queue their scan set-up, slots reserved with one atomic per wave
            if constexpr (!MASK) {
                const unsigned long long mm = __ballot(mid);
                if (mm) {
                    const int leader = __ffsll((long long)mm) - 1;
                    uint32_t base = 0;
                    const uint32_t seg = blockIdx.x % kMidSegments, seg_capacity = fv.mid_capacity / kMidSegments;
                    if ((int)lane == leader) base = atomicAdd(&fv.counters->mid_seg[seg * 32u], (uint32_t)__popcll(mm));
                    base = __shfl(base, leader, 64);
                    if (mid) {
                        const uint32_t in_seg = base + ballot_rank(mm), slot = seg * seg_capacity + in_seg;
                        if (in_seg < seg_capacity) {
                            SnapVert s0, s1, s2;
                            load_snap(ia, ib, ic, s0, s1, s2, true);
                            SetupTri t;
                            SetupMask mk;
                            make_setup(fv, s0, s1, s2, cull_front, t, mk);
                            t.id = (vm.instance << 7) | ti; t.draw_id = vm.draw_id;
                            t.masked = (MASK ? 2u : 0u) | ((MASK && at.has_tex) ? 1u : 0u);
                            fv.mid[slot] = t;
                            if (MASK && at.has_tex) fv.mid_mask[slot] = mk;
                        } else {
                            atomicAdd(&fv.counters->mid_overflow, 1u);
                        }
                    }
                }
            }
            // deferred triangles: reserve slots with one atomic per wave
            {
                const unsigned long long dm = __ballot(defer);
                if (dm) {
                    const uint32_t n = (uint32_t)__popcll(dm);
                    uint32_t base = 0;
                    if (lane == (uint32_t)__ffsll((long long)dm) - 1u) base = atomicAdd(&fv.counters->deferred_count, n);
                    base = __shfl(base, __ffsll((long long)dm) - 1, 64);
                    if (defer) {
                        const uint32_t slot = base + ballot_rank(dm);
                        if (slot < fv.deferred_capacity) {
                            DeferredTriangle* dt = &fv.deferred[slot];
                            const uint32_t idx[3] = {ia, ib, ic};
                            const uint32_t* mv = sv.meshlet_vertices + m.vertex_offset;
#pragma unroll
                            for (int k = 0; k < 3; ++k) {
                                dt->pos[k][0] = lds.cx[idx[k]]; dt->pos[k][1] = lds.cy[idx[k]];
                                dt->pos[k][2] = lds.cz[idx[k]]; dt->pos[k][3] = lds.cw[idx[k]];
                                const rdr_vertex* vv = &sv.vertices[mv[idx[k]] + mesh_vertex_offset];
                                dt->uv[k][0] = vv->uv[0]; dt->uv[k][1] = vv->uv[1];
                                dt->uv[k][2] = vv->uv[2]; dt->uv[k][3] = vv->uv[3];
                            }
                            dt->id = (vm.instance << 7) | ti;
                            dt->draw_id = vm.draw_id;
                            dt->replicated = vm.counts >> 31;
                            dt->_pad = 0;
                        } else {
                            atomicAdd(&fv.counters->deferred_overflow, 1u);
                        }
                        st_big++;
                    }
                }
            }
        }
        mi = mi_next;
        mi_next = mi_after;
    }
    if (fv.collect_stats) {
        uint32_t a = wave_sum(st_tested), b = wave_sum(st_raster), c = wave_sum(st_big), f = wave_sum(st_frags);
        uint32_t v = wave_sum(st_verts);
        if (lane == 0) {
            uint32_t* slot = fv.stat_slots + fv.grid.slot_raster + (blockIdx.x + (MASK ? fv.grid.raster_blocks : 0u)) * kRasterSlots;
            if (fv.stat_accumulate) { a += slot[0]; b += slot[1]; c += slot[2]; f += slot[3]; v += slot[4]; }
            slot[0] = a; slot[1] = b; slot[2] = c; slot[3] = f; slot[4] = v;
        }
    }
}

// Mid-sized triangles queued by k_raster: one wave per record, every lane one pixel of each
// 8x8 stamp, fragments merged with the same atomic as the per-lane scan.  Spreads the pixel
// work of a meshlet that covers a lot of screen over many waves.
RDR_DEV void mid_raster_role(const SceneView& sv, const FrameView& fv, uint32_t block, uint32_t n_blocks, uint32_t stat_slot) {
    __shared__ uint32_t wg_frags[4];
    const uint32_t lane = lane_id();
    // the queue is segmented (FrameCounters::mid_seg): lane s holds the running total up to segment s
    const uint32_t seg_capacity = fv.mid_capacity / kMidSegments;
    const uint32_t seg_n = lane < kMidSegments ? min(fv.counters->mid_seg[lane * 32u], seg_capacity) : 0u;
    const uint32_t seg_end = wave_scan_inclusive(seg_n, lane);
    const uint32_t count = __shfl(seg_end, 63, 64);
    const uint32_t wave = (block * blockDim.x + threadIdx.x) >> 6, n_waves = (n_blocks * blockDim.x) >> 6;
    uint32_t frags = 0;
    for (uint32_t i = wave; i < count; i += n_waves) {
        const uint32_t seg = (uint32_t)__popcll(__ballot(lane < kMidSegments && seg_end <= i)); // segments that end at or before i
        const uint32_t r = seg * seg_capacity + (i - (__shfl(seg_end, (int)seg, 64) - __shfl(seg_n, (int)seg, 64)));
        SetupTri t; // the same record in every lane: through the scalar unit
        {
            const uint32_t* words = reinterpret_cast<const uint32_t*>(&fv.mid[__builtin_amdgcn_readfirstlane(r)]);
            uint32_t* out = reinterpret_cast<uint32_t*>(&t);
#pragma unroll
            for (uint32_t k = 0; k < sizeof(SetupTri) / 4u; ++k) out[k] = load_uniform(words + k);
        }
        TriSetup<int32_t> ts;
        ts.v0 = SnapVert{}; ts.v1 = SnapVert{}; ts.v2 = SnapVert{};
        ts.v0.X = t.X0; ts.v0.Y = t.Y0; ts.v1.X = t.X1; ts.v1.Y = t.Y1; ts.v2.X = t.X2; ts.v2.Y = t.Y2;
        ts.v0.z = t.z0;
        ts.dx01 = t.X1 - t.X0; ts.dy01 = t.Y1 - t.Y0;
        ts.dx12 = t.X2 - t.X1; ts.dy12 = t.Y2 - t.Y1;
        ts.dx20 = t.X0 - t.X2; ts.dy20 = t.Y0 - t.Y2;
        ts.bias01 = (ts.dy01 < 0 || (ts.dy01 == 0 && ts.dx01 > 0)) ? 0 : -1;
        ts.bias12 = (ts.dy12 < 0 || (ts.dy12 == 0 && ts.dx12 > 0)) ? 0 : -1;
        ts.bias20 = (ts.dy20 < 0 || (ts.dy20 == 0 && ts.dx20 > 0)) ? 0 : -1;
        ts.inv_area = t.inv_area; ts.dz1 = t.dz1; ts.dz2 = t.dz2;
        ts.px0 = t.px0; ts.py0 = t.py0; ts.px1 = t.px1; ts.py1 = t.py1;
        ts.valid = true;
        if (t.masked & 2u) { // alpha-tested draw (the record says so: no dependent load for the opaque majority)
            const AlphaTest at = make_alpha_test(sv, uniform_ref(&sv.draws[t.draw_id]));
            if (t.masked & 1u) {
                const SetupMask* mk = &fv.mid_mask[r];
                SnapVert* vs[3] = {&ts.v0, &ts.v1, &ts.v2};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    vs[k]->inv_w = mk->inv_w[k];
                    vs[k]->u0 = mk->uv[k][0]; vs[k]->v0 = mk->uv[k][1]; vs[k]->u1 = mk->uv[k][2]; vs[k]->v1 = mk->uv[k][3];
                }
            }
            frags += scan_wave<int32_t, true>(sv, fv, ts, at, t.id, lane);
        } else {
            AlphaTest at;
            at.enabled = false; at.has_tex = false;
            frags += scan_wave<int32_t, false>(sv, fv, ts, at, t.id, lane);
        }
    }
    if (fv.collect_stats) {
        const uint32_t f = wave_sum(frags);
        if (lane == 0) wg_frags[threadIdx.x >> 6] = f;
        __syncthreads();
        if (threadIdx.x == 0) stat_store(fv, stat_slot + block, wg_frags[0] + wg_frags[1] + wg_frags[2] + wg_frags[3]);
    }
}

// The covered pixels of small alpha-tested triangles queued by k_raster<true>: one fragment per
// lane (alpha test, depth, atomic), whatever triangle it belongs to.
__global__ __launch_bounds__(256) void k_alpha_frag(SceneView sv, FrameView fv) {
    __shared__ uint32_t seg_end[kFragSegments], seg_cnt[kFragSegments];
    __shared__ uint32_t wg_frags[4];
    const uint32_t lane = lane_id();
    const uint32_t cap_p = fv.frag_px_capacity / kFragSegments;
    if (threadIdx.x < 64) {
        const uint32_t n = lane < kFragSegments ? min((uint32_t)fv.counters->frag_seg[lane * 16u], cap_p) : 0u;
        const uint32_t incl = wave_scan_inclusive(n, lane);
        if (lane < kFragSegments) { seg_end[lane] = incl; seg_cnt[lane] = n; }
    }
    __syncthreads();
    const uint32_t count = seg_end[kFragSegments - 1];
    uint32_t frags = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        uint32_t seg = 0; // first segment whose running total exceeds i
#pragma unroll
        for (uint32_t step = kFragSegments / 2; step; step >>= 1)
            if (seg_end[seg + step - 1] <= i) seg += step;
        const uint32_t e = fv.frag_px[(size_t)seg * cap_p + (i - (seg_end[seg] - seg_cnt[seg]))];
        if ((e >> 4) >= fv.frag_tri_capacity) continue; // cannot happen unless a queue overflowed and left a hole
        const FragTri r = fv.frag_tri[e >> 4];
        const uint32_t k = e & 15u, dy = k / max(r.bw, 1u), dx = k - dy * r.bw;
        const int px = (int)(r.origin & 0xFFFFu) + (int)dx, py = (int)(r.origin >> 16) + (int)dy;
        if (px >= (int)fv.width || py >= (int)fv.height || r.draw_id >= sv.draw_count) continue; // same
        const AlphaTest at = sv.draw_alpha[r.draw_id]; // the draw's alpha test as one record (the host's make_alpha_test), not a chain draw -> material -> image
        TriSetup<int32_t> ft;
        ft.v0 = SnapVert{}; ft.v1 = SnapVert{}; ft.v2 = SnapVert{};
        ft.v0.X = r.X[0]; ft.v1.X = r.X[1]; ft.v2.X = r.X[2];
        ft.v0.Y = r.Y[0]; ft.v1.Y = r.Y[1]; ft.v2.Y = r.Y[2];
        ft.v0.z = r.z0; ft.dz1 = r.dz1; ft.dz2 = r.dz2; ft.inv_area = r.inv_area;
        ft.v0.inv_w = r.inv_w[0]; ft.v1.inv_w = r.inv_w[1]; ft.v2.inv_w = r.inv_w[2];
        ft.v0.u0 = ft.v0.u1 = r.u[0]; ft.v1.u0 = ft.v1.u1 = r.u[1]; ft.v2.u0 = ft.v2.u1 = r.u[2];
        ft.v0.v0 = ft.v0.v1 = r.v[0]; ft.v1.v0 = ft.v1.v1 = r.v[1]; ft.v2.v0 = ft.v2.v1 = r.v[2];
        ft.dx01 = ft.v1.X - ft.v0.X; ft.dy01 = ft.v1.Y - ft.v0.Y;
        ft.dx12 = ft.v2.X - ft.v1.X; ft.dy12 = ft.v2.Y - ft.v1.Y;
        ft.dx20 = ft.v0.X - ft.v2.X; ft.dy20 = ft.v0.Y - ft.v2.Y;
        ft.bias01 = (ft.dy01 < 0 || (ft.dy01 == 0 && ft.dx01 > 0)) ? 0 : -1;
        ft.bias12 = (ft.dy12 < 0 || (ft.dy12 == 0 && ft.dx12 > 0)) ? 0 : -1;
        ft.bias20 = (ft.dy20 < 0 || (ft.dy20 == 0 && ft.dx20 > 0)) ? 0 : -1;
        ft.px0 = ft.px1 = px; ft.py0 = ft.py1 = py;
        ft.valid = true;
        int32_t e01, e12, e20;
        edges_at<int32_t>(ft, px, py, e01, e12, e20);
        touch_pixel(fv, px, py);
        frags += shade_sample<int32_t, true>(sv, fv, ft, at, px, py, e01, e12, e20, r.id);
    }
    if (fv.collect_stats) {
        const uint32_t f = wave_sum(frags);
        if (lane == 0) wg_frags[threadIdx.x >> 6] = f;
        __syncthreads();
        if (threadIdx.x == 0) stat_store(fv, fv.grid.slot_frag + blockIdx.x, wg_frags[0] + wg_frags[1] + wg_frags[2] + wg_frags[3]);
    }
}

// ------------------------------------------------------------------------------
// Clipping helpers of the tile path.
// ------------------------------------------------------------------------------
struct ClipVert {
    float x, y, z, w, uv[4];
};

RDR_DEV float plane_dist(const ClipVert& v, int plane) {
    switch (plane) {
    case 0: return v.w - v.z;
    case 1: return kGuardBand * v.w - v.x;
    case 2: return kGuardBand * v.w + v.x;
    case 3: return kGuardBand * v.w - v.y;
    default: return kGuardBand * v.w + v.y;
    }
}

// New vertex on the plane, always interpolated from the inside vertex.
RDR_DEV ClipVert clip_lerp(const ClipVert& in, const ClipVert& out, float din, float dout) {
    const float t = din / (din - dout);
    ClipVert r;
    r.x = in.x + t * (out.x - in.x);
    r.y = in.y + t * (out.y - in.y);
    r.z = in.z + t * (out.z - in.z);
    r.w = in.w + t * (out.w - in.w);
#pragma unroll
    for (int k = 0; k < 4; ++k) r.uv[k] = in.uv[k] + t * (out.uv[k] - in.uv[k]);
    return r;
}

RDR_DEV SnapVert snap_clip_vert(const ClipVert& v, float half_w, float half_h) {
    SnapVert s;
    const float rw = 1.0f / v.w;
    project_xy(v.x, v.y, rw, half_w, half_h, s.X, s.Y);
    s.z = v.z * rw;
    s.inv_w = rw;
    s.u0 = v.uv[0]; s.v0 = v.uv[1]; s.u1 = v.uv[2]; s.v1 = v.uv[3];
    return s;
}

// ------------------------------------------------------------------------------
// Tile path: clipped, large or pixel-heavy triangles.
//   k_tile_setup  clip -> fan -> snap -> SetupTri records, counted per 64x64 tile as they are made
//   k_tile_fill   prefix sum of the per-tile counts -> bin offsets and the job list, then each
//                 record's index into the bins of the tiles it touches
//   k_tile_raster jobs of kTileJob records over a 32x32 piece of a tile: the records staged
//                 through LDS, each thread owns a 2x2 pixel block whose keys stay in registers,
//                 merged into the visibility buffer at the end.
// ------------------------------------------------------------------------------

struct EdgeEq {
    long long e;  // value at the reference pixel centre
    long long ax; // change per pixel in x
    long long by; // change per pixel in y
    int bias;     // 0 for top-left edges, -1 otherwise
};

RDR_DEV EdgeEq make_edge(int32_t Xa, int32_t Ya, int32_t Xb, int32_t Yb, int px, int py) {
    const int32_t dx = Xb - Xa, dy = Yb - Ya;
    EdgeEq q;
    const long long Px = (long long)px * 256 + 128, Py = (long long)py * 256 + 128;
    q.e = (long long)dx * (Py - Ya) - (long long)dy * (Px - Xa);
    q.ax = -(long long)dy * 256;
    q.by = (long long)dx * 256;
    q.bias = (dy < 0 || (dy == 0 && dx > 0)) ? 0 : -1;
    return q;
}

// Largest value of the edge function over the pixel box [0,w] x [0,h] from the reference pixel.
RDR_DEV long long edge_max(const EdgeEq& q, int w, int h) {
    return q.e + (q.ax > 0 ? q.ax * w : 0) + (q.by > 0 ? q.by * h : 0);
}

// Does the triangle possibly cover a pixel centre of tile (tx, ty)?  Box test plus the
// three edges evaluated at their most favourable tile corner.  Deterministic: the
// counting and the filling kernel must agree.
RDR_DEV bool tile_touched(const FrameView& fv, const SetupTri& t, uint32_t tx, uint32_t ty) {
    const int x0 = (int)tx * kTile, y0 = (int)ty * kTile;
    if ((int)t.px1 < x0 || (int)t.px0 > x0 + kTile - 1 || (int)t.py1 < y0 || (int)t.py0 > y0 + kTile - 1) return false;
    // a triangle every sort-last rank draws is rasterised on this rank's rows only
    if ((t.masked & 4u) && !rdr_rows_owned(fv, (uint32_t)max(y0, (int)t.py0), (uint32_t)min(y0 + kTile - 1, (int)t.py1))) return false;
    const EdgeEq e01 = make_edge(t.X0, t.Y0, t.X1, t.Y1, x0, y0);
    const EdgeEq e12 = make_edge(t.X1, t.Y1, t.X2, t.Y2, x0, y0);
    const EdgeEq e20 = make_edge(t.X2, t.Y2, t.X0, t.Y0, x0, y0);
    return edge_max(e01, kTile - 1, kTile - 1) + e01.bias >= 0 && edge_max(e12, kTile - 1, kTile - 1) + e12.bias >= 0 &&
           edge_max(e20, kTile - 1, kTile - 1) + e20.bias >= 0;
}

// tile_touched for the tiles of one tile row, left to right: the three edge maxima are linear in the tile's position,
// so after the first tile of the row each further one is three 64-bit additions and a sign test instead of three
// edge set-ups (records that span the screen are counted and binned tile by tile, and those set-ups were most of it).
// The caller walks tiles inside the record's own tile box, which is tile_touched's box test.
struct TileRowWalk {
    long long v[3];    // edge maximum over the current tile, top-left bias included: the tile is touched when none is negative
    long long step[3]; // change per tile in x
};
RDR_DEV TileRowWalk tile_row_walk(const SetupTri& t, uint32_t tx0, uint32_t ty) {
    const int x0 = (int)tx0 * kTile, y0 = (int)ty * kTile;
    const EdgeEq e[3] = {make_edge(t.X0, t.Y0, t.X1, t.Y1, x0, y0), make_edge(t.X1, t.Y1, t.X2, t.Y2, x0, y0), make_edge(t.X2, t.Y2, t.X0, t.Y0, x0, y0)};
    TileRowWalk r;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        r.v[k] = edge_max(e[k], kTile - 1, kTile - 1) + e[k].bias;
        r.step[k] = e[k].ax * kTile;
    }
    return r;
}
// The same for the tiles of one tile column, top to bottom.
RDR_DEV TileRowWalk tile_col_walk(const SetupTri& t, uint32_t tx, uint32_t ty0) {
    const int x0 = (int)tx * kTile, y0 = (int)ty0 * kTile;
    const EdgeEq e[3] = {make_edge(t.X0, t.Y0, t.X1, t.Y1, x0, y0), make_edge(t.X1, t.Y1, t.X2, t.Y2, x0, y0), make_edge(t.X2, t.Y2, t.X0, t.Y0, x0, y0)};
    TileRowWalk r;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        r.v[k] = edge_max(e[k], kTile - 1, kTile - 1) + e[k].bias;
        r.step[k] = e[k].by * kTile;
    }
    return r;
}
RDR_DEV bool tile_row_touched(const TileRowWalk& r) { return (r.v[0] | r.v[1] | r.v[2]) >= 0; }
RDR_DEV void tile_row_next(TileRowWalk& r) {
#pragma unroll
    for (int k = 0; k < 3; ++k) r.v[k] += r.step[k];
}
// Does this rank rasterise the record on the rows of tile row ty?  (Always, unless the record belongs to a draw every
// sort-last rank draws: those are rasterised on the rank's own rows only.)
RDR_DEV bool tile_row_wanted(const FrameView& fv, const SetupTri& t, uint32_t ty) {
    const int y0 = (int)ty * kTile;
    return !(t.masked & 4u) || rdr_rows_owned(fv, (uint32_t)max(y0, (int)t.py0), (uint32_t)min(y0 + kTile - 1, (int)t.py1));
}

// Binning of the set-up records into 64x64 tiles, two passes over the (record, tile) pairs where the record may
// cover a pixel of the tile.  Counting happens where the records are made: the wave of k_tile_setup that produced
// a record adds it to its tiles' counts (count_tiles).  k_tile_fill then turns the counts into bin offsets and
// writes the record indices into the bins.  In both, a lane walks the tiles of its own record while there are at
// most kLaneTiles of them; records spanning more tiles are taken one at a time by the whole wave, so that a
// full-screen triangle costs tens of steps rather than thousands.
constexpr uint32_t kLaneTiles = 4;

RDR_DEV void count_tiles(const FrameView& fv, const SetupTri& t, bool valid, uint32_t lane) {
    uint32_t tx0 = 0, ty0 = 0, tw = 1, n_tiles = 0;
    if (valid) {
        tx0 = t.px0 / kTile; ty0 = t.py0 / kTile;
        tw = t.px1 / kTile - tx0 + 1u;
        n_tiles = tw * (t.py1 / kTile - ty0 + 1u);
    }
    if (n_tiles <= kLaneTiles)
        for (uint32_t q = 0; q < n_tiles; ++q)
            if (tile_touched(fv, t, tx0 + q % tw, ty0 + q / tw)) atomicAdd(&fv.tile_count[(ty0 + q / tw) * fv.tiles_x + tx0 + q % tw], 1u);
    unsigned long long wide = __ballot(n_tiles > kLaneTiles);
    while (wide) {
        const int src = __ffsll((long long)wide) - 1;
        wide &= wide - 1ull;
        SetupTri w; // what tile_touched reads of the record in lane `src`
        w.X0 = __shfl(t.X0, src, 64); w.Y0 = __shfl(t.Y0, src, 64); w.X1 = __shfl(t.X1, src, 64);
        w.Y1 = __shfl(t.Y1, src, 64); w.X2 = __shfl(t.X2, src, 64); w.Y2 = __shfl(t.Y2, src, 64);
        const uint32_t lo = __shfl((uint32_t)t.px0 | ((uint32_t)t.py0 << 16), src, 64);
        const uint32_t hi = __shfl((uint32_t)t.px1 | ((uint32_t)t.py1 << 16), src, 64);
        w.px0 = (uint16_t)lo; w.py0 = (uint16_t)(lo >> 16); w.px1 = (uint16_t)hi; w.py1 = (uint16_t)(hi >> 16);
        w.masked = __shfl(t.masked, src, 64);
        const uint32_t wx0 = w.px0 / kTile, wy0 = w.py0 / kTile, ww = w.px1 / kTile - wx0 + 1u;
        const uint32_t rows = w.py1 / kTile - wy0 + 1u;
        if (ww > rows) { // a lane per tile column, walked top to bottom: the shorter way through a box that is wider than high
            const bool all_rows = !(w.masked & 4u);
            for (uint32_t col = lane; col < ww; col += 64u) {
                TileRowWalk r = tile_col_walk(w, wx0 + col, wy0);
                uint32_t* cnt = &fv.tile_count[wy0 * fv.tiles_x + wx0 + col];
                for (uint32_t i = 0; i < rows; ++i, tile_row_next(r), cnt += fv.tiles_x)
                    if (tile_row_touched(r) && (all_rows || tile_row_wanted(fv, w, wy0 + i))) atomicAdd(cnt, 1u);
            }
            continue;
        }
        for (uint32_t row = lane; row < rows; row += 64u) { // a lane per tile row, walked left to right
            const uint32_t ty = wy0 + row;
            if (!tile_row_wanted(fv, w, ty)) continue;
            TileRowWalk r = tile_row_walk(w, wx0, ty);
            uint32_t* cnt = &fv.tile_count[ty * fv.tiles_x + wx0];
            for (uint32_t i = 0; i < ww; ++i, tile_row_next(r))
                if (tile_row_touched(r)) atomicAdd(cnt + i, 1u);
        }
    }
}

// Set-up works on kSetupChunk triangles per wave at a time, sixteen lanes to a triangle.  The polygon being
// clipped lives in those lanes' registers, a vertex per lane: against one plane every lane looks at the edge from
// its vertex to the next (cross-lane reads), keeps its vertex when that is inside and adds the crossing when the
// edge changes sides, exactly what Sutherland-Hodgman emits for that edge and in the same order, at the index the
// counts of the lanes before it give (two ballots); the new polygon goes through a small LDS buffer to the lanes
// that own its vertices.  A plane is one step for all vertices of four triangles instead of a loop over the
// vertices of each (indexed per lane, which in registers means scratch memory, and in LDS a chain of dependent
// round trips per vertex: an interior scene's most clipped chunk of sixteen took 9 us to clip and 25 us to fan out,
// with the kernel waiting for it).  A triangle against five planes has at most eight vertices (ten are kept).
constexpr int kClipMaxVerts = 10;
constexpr uint32_t kSetupChunk = 4, kSetupLanes = 16;
static_assert(kSetupChunk * kSetupLanes == 64 && kClipMaxVerts <= (int)kSetupLanes, "a wave holds kSetupChunk polygons, a vertex per lane");
struct ClipLds {
    float v[8][64]; // [component][triangle of the chunk * kSetupLanes + vertex]
};
RDR_DEV ClipVert clip_load(const ClipLds& l, uint32_t at) {
    const float(*v)[64] = l.v;
    ClipVert c;
    c.x = v[0][at]; c.y = v[1][at]; c.z = v[2][at]; c.w = v[3][at];
#pragma unroll
    for (int k = 0; k < 4; ++k) c.uv[k] = v[4 + k][at];
    return c;
}
RDR_DEV void clip_store(ClipLds& l, uint32_t at, const ClipVert& c) {
    float(*v)[64] = l.v;
    v[0][at] = c.x; v[1][at] = c.y; v[2][at] = c.z; v[3][at] = c.w;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[4 + k][at] = c.uv[k];
}
RDR_DEV ClipVert clip_vert_of_lane(const ClipVert& c, uint32_t src) {
    ClipVert r;
    r.x = __shfl(c.x, (int)src, 64); r.y = __shfl(c.y, (int)src, 64); r.z = __shfl(c.z, (int)src, 64); r.w = __shfl(c.w, (int)src, 64);
#pragma unroll
    for (int k = 0; k < 4; ++k) r.uv[k] = __shfl(c.uv[k], (int)src, 64);
    return r;
}
RDR_DEV SnapVert snap_vert_of_lane(const SnapVert& c, uint32_t src) {
    SnapVert r;
    r.X = __shfl(c.X, (int)src, 64); r.Y = __shfl(c.Y, (int)src, 64); r.z = __shfl(c.z, (int)src, 64); r.inv_w = __shfl(c.inv_w, (int)src, 64);
    r.u0 = __shfl(c.u0, (int)src, 64); r.v0 = __shfl(c.v0, (int)src, 64); r.u1 = __shfl(c.u1, (int)src, 64); r.v1 = __shfl(c.v1, (int)src, 64);
    return r;
}

// One wave's share of the set-up: wave `wave_index` of `n_waves`, with an LDS polygon buffer of its own.  With
// kBlockWaves > 1 every wave of the workgroup is in here (wave threadIdx.x / 64 of kBlockWaves) and the workgroup
// reserves its waves' record slots with one atomic per round (`agg`: 2 * (kBlockWaves + 1) words of LDS): the
// reservations all go to one counter and return a value, about 11 ns apiece, and a frame's few thousand triangles
// dealt out one by one put a thousand waves in that queue at the same moment.
template <uint32_t kBlockWaves>
RDR_DEV void tile_setup_role(const SceneView& sv, const FrameView& fv, ClipLds& clip, uint32_t* agg, uint32_t wave_index, uint32_t n_waves) {
    const uint32_t lane = lane_id(), first_lane = lane & ~(kSetupLanes - 1u), j = lane & (kSetupLanes - 1u);
    const uint32_t count = min(fv.counters->deferred_count, fv.deferred_capacity);
    const float half_w = (float)fv.width * 0.5f, half_h = (float)fv.height * 0.5f;
    // A record that covers many tiles is counted into its tiles by the whole wave (count_tiles), one such record after
    // another, and the triangles that come here are mostly of that kind (walls, floors) and come in runs (a wall's
    // triangles follow each other in the list): a wave's triangles are taken n_waves apart, so that a run is dealt out
    // one triangle to a wave.
    const uint32_t stride = n_waves * kSetupChunk;
    // whole waves iterate together so that slots can be reserved with one atomic per wave
    uint32_t round = 0;
    for (uint32_t base_i = 0; base_i < count; base_i += stride, ++round) {
        const uint32_t di = base_i + (lane / kSetupLanes) * n_waves + wave_index;
        uint32_t n = 0; // vertices of this lane's polygon; lane j holds vertex j
        bool cull_front = true, masked = false, alpha_draw = false;
        uint32_t id = 0, draw_id = 0, replicated = 0;
        ClipVert v{};
        if (di < count) {
            const DeferredTriangle* dt = &fv.deferred[di];
            draw_id = dt->draw_id;
            id = dt->id;
            replicated = dt->replicated;
            const rdr_mesh_draw d = sv.draws[draw_id];
            cull_front = d.alpha_mode == 0u;
            alpha_draw = d.alpha_mode == 1u;
            const AlphaTest at = make_alpha_test(sv, d);
            masked = at.enabled && at.has_tex;
            n = 3;
            if (j < 3u) {
                v.x = dt->pos[j][0]; v.y = dt->pos[j][1]; v.z = dt->pos[j][2]; v.w = dt->pos[j][3];
                for (int k = 0; k < 4; ++k) v.uv[k] = dt->uv[j][k];
            }
        }
        const unsigned long long needing = __ballot(j < n && vert_needs_clip(v.x, v.y, v.z, v.w));
        const bool needs = ((needing >> first_lane) & 0xFFFFull) != 0ull, unclipped = !needs;
        if (needing) {
            // Sutherland-Hodgman against near + the four guard-band planes
            for (int plane = 0; plane < 5; ++plane) {
                const bool active = needs && j < n;
                const uint32_t next = first_lane | (j + 1u < n ? j + 1u : 0u);
                const ClipVert q = clip_vert_of_lane(v, next);
                const float dp = plane_dist(v, plane), dq = __shfl(dp, (int)next, 64);
                const bool pin = dp >= 0.0f, qin = dq >= 0.0f;
                const bool keep = active && pin, cross = active && pin != qin;
                const uint32_t keep_bits = (uint32_t)(__ballot(keep) >> first_lane) & 0xFFFFu, cross_bits = (uint32_t)(__ballot(cross) >> first_lane) & 0xFFFFu;
                const uint32_t before = (1u << j) - 1u;
                const uint32_t at = (uint32_t)__popc(keep_bits & before) + (uint32_t)__popc(cross_bits & before);
                if (keep && at < (uint32_t)kClipMaxVerts) clip_store(clip, first_lane + at, v);
                if (cross && at + (keep ? 1u : 0u) < (uint32_t)kClipMaxVerts)
                    clip_store(clip, first_lane + at + (keep ? 1u : 0u), pin ? clip_lerp(v, q, dp, dq) : clip_lerp(q, v, dq, dp));
                // a wave's LDS instructions execute in the order they were issued: the lanes' stores, then the loads, then
                // the next plane's stores need no wait between them, only to stay in this order
                asm volatile("" ::: "memory");
                if (needs) {
                    n = min((uint32_t)__popc(keep_bits) + (uint32_t)__popc(cross_bits), (uint32_t)kClipMaxVerts);
                    if (j < n) v = clip_load(clip, lane);
                }
                asm volatile("" ::: "memory");
            }
        }
        // fan triangle (0, j + 1, j + 2) in lane j
        SnapVert own{};
        if (j < n) own = snap_clip_vert(v, half_w, half_h);
        const SnapVert s0 = snap_vert_of_lane(own, first_lane), sp = snap_vert_of_lane(own, (lane + 1u) & 63u), sn = snap_vert_of_lane(own, (lane + 2u) & 63u);
        SetupTri t;
        SetupMask mk;
        bool ok = j + 2u < n && make_setup(fv, s0, sp, sn, cull_front, t, mk);
        t.id = id; t.draw_id = draw_id; t.masked = (masked ? 1u : 0u) | (alpha_draw ? 2u : 0u) | (replicated ? 4u : 0u);
        // Alpha-tested triangles that k_raster<true> could not enumerate in a lane but that fit a wave
        // (unclipped, small extent, box within the mid limit) are handed on to k_mid_raster, which runs
        // after this kernel: same record, same queue as the opaque ones k_raster<false> queues itself.
        bool to_mid = false;
        if (ok && alpha_draw && unclipped) {
            const int32_t ex = max(t.X0, max(t.X1, t.X2)) - min(t.X0, min(t.X1, t.X2));
            const int32_t ey = max(t.Y0, max(t.Y1, t.Y2)) - min(t.Y0, min(t.Y1, t.Y2));
            const uint32_t npx = ((uint32_t)t.px1 - t.px0 + 1u) * ((uint32_t)t.py1 - t.py0 + 1u);
            to_mid = ex <= kSmallExtent && ey <= kSmallExtent && npx <= fv.mid_max_pixels;
        }
        const unsigned long long mm = __ballot(to_mid);
        if (mm) {
            const int mid_leader = __ffsll((long long)mm) - 1;
            const uint32_t seg = wave_index % kMidSegments, seg_capacity = fv.mid_capacity / kMidSegments;
            uint32_t base = 0;
            if ((int)lane == mid_leader) base = atomicAdd(&fv.counters->mid_seg[seg * 32u], (uint32_t)__popcll(mm));
            base = __shfl(base, mid_leader, 64);
            if (to_mid) {
                const uint32_t in_seg = base + ballot_rank(mm);
                if (in_seg < seg_capacity) {
                    fv.mid[seg * seg_capacity + in_seg] = t;
                    if (masked) fv.mid_mask[seg * seg_capacity + in_seg] = mk;
                } else {
                    atomicAdd(&fv.counters->mid_overflow, 1u);
                }
            }
        }
        ok = ok && !to_mid;
        const unsigned long long m = __ballot(ok);
        uint32_t slot0 = 0;
        if (kBlockWaves == 1u) {
            if (!m) continue;
            const int leader = __ffsll((long long)m) - 1;
            if ((int)lane == leader) slot0 = atomicAdd(&fv.counters->setup_count, (uint32_t)__popcll(m));
            slot0 = __shfl(slot0, leader, 64);
        } else {
            // the rounds alternate between two sets of words: a wave may be a round ahead of one that still reads
            uint32_t* const a = agg + (round & 1u) * (kBlockWaves + 1u);
            const uint32_t wave = threadIdx.x >> 6;
            if (lane == 0u) a[wave] = (uint32_t)__popcll(m);
            __syncthreads();
            if (threadIdx.x == 0u) {
                uint32_t total = 0;
                for (uint32_t w = 0; w < kBlockWaves; ++w) total += a[w];
                a[kBlockWaves] = total ? atomicAdd(&fv.counters->setup_count, total) : 0u;
            }
            __syncthreads();
            slot0 = a[kBlockWaves];
            for (uint32_t w = 0; w < kBlockWaves; ++w) slot0 += w < wave ? a[w] : 0u;
        }
        const uint32_t slot = slot0 + ballot_rank(m);
        const bool stored = ok && slot < fv.setup_capacity;
        if (ok && !stored) atomicAdd(&fv.counters->setup_overflow, 1u);
        if (stored) {
            fv.setup[slot] = t;
            if (masked) fv.setup_mask[slot] = mk;
        }
        count_tiles(fv, t, stored, lane); // the whole wave: wide records are walked by all lanes
    }
}

constexpr uint32_t kSetupWaves = 4;
__global__ __launch_bounds__(64 * kSetupWaves) void k_tile_setup(SceneView sv, FrameView fv) {
    __shared__ ClipLds clip[kSetupWaves];
    __shared__ uint32_t agg[2 * (kSetupWaves + 1)];
    const uint32_t wave = threadIdx.x >> 6;
    tile_setup_role<kSetupWaves>(sv, fv, clip[wave], agg, blockIdx.x * kSetupWaves + wave, gridDim.x * kSetupWaves);
}

// Bin offsets and the bins.  Every workgroup first repeats the prefix sum over the per-tile counts for itself
// (a few thousand counts, served by the L2: cheaper than a launch in between, which costs 4-5 us of dependency
// latency even on a frame with nothing to bin; a last-workgroup ticket behind a release fence was slower still)
// and keeps the offsets in LDS.  What later kernels need in memory is written once, spread over the workgroups:
// the offsets, and the job list of k_tile_raster (a tile's bin is cut into jobs of kTileJob records over the
// kTileParts pieces of the tile, so that a tile under many expensive triangles is worked on by several
// workgroups instead of setting that kernel's length alone).  Both running sums travel in one 64-bit value:
// entries in the low 40 bits, jobs above.  Then the records are visited as in count_tiles and written into
// their tiles' bins.  FrameView::tile_cursor is zero on entry (k_tile_raster leaves it so).
constexpr uint32_t kBinChunk = 4;
RDR_DEV void tile_fill_role(const FrameView& fv, uint32_t block, uint32_t n_blocks) {
    extern __shared__ uint32_t tile_off[]; // n_tiles offsets, then the scan's 256 partial sums (64-bit)
    const uint32_t lane = lane_id();
    const uint32_t count = min(fv.counters->setup_count, fv.setup_capacity);
    if (count == 0u) { // nothing was set up: empty bins, no jobs
        if (block == 0 && threadIdx.x == 0) { fv.counters->bin_total = 0u; fv.counters->tile_nonempty = 0u; }
        return;
    }
    const uint32_t n_tiles = fv.tiles_x * fv.tiles_y;
    unsigned long long* partial = reinterpret_cast<unsigned long long*>(tile_off + ((n_tiles + 1u) & ~1u));
    {
        const uint32_t per = (n_tiles + 255u) / 256u;
        const uint32_t first = min(threadIdx.x * per, n_tiles), last = min(first + per, n_tiles);
        unsigned long long sum = 0;
        for (uint32_t i = first; i < last; ++i) {
            const uint32_t c = fv.tile_count[i];
            sum += (unsigned long long)c + ((unsigned long long)((c + kTileJob - 1u) / kTileJob * kTileParts) << 40);
        }
        partial[threadIdx.x] = sum;
        __syncthreads();
        for (uint32_t o = 1; o < 256u; o <<= 1) { // inclusive Hillis-Steele scan
            const unsigned long long v = threadIdx.x >= o ? partial[threadIdx.x - o] : 0ull;
            __syncthreads();
            partial[threadIdx.x] += v;
            __syncthreads();
        }
        const unsigned long long before = partial[threadIdx.x] - sum;
        uint32_t run = (uint32_t)(before & ((1ull << 40) - 1ull)), listed = (uint32_t)(before >> 40);
        const bool writer = threadIdx.x % n_blocks == block; // this thread's tiles go to memory from this workgroup
        for (uint32_t i = first; i < last; ++i) {
            const uint32_t c = fv.tile_count[i];
            tile_off[i] = run;
            if (writer) {
                fv.tile_offset[i] = run;
                for (uint32_t r = 0; r < c; r += kTileJob)
                    for (uint32_t part = 0; part < kTileParts; ++part, ++listed)
                        if (listed < fv.tile_list_capacity) fv.tile_list[listed] = make_uint2(i, r | part); // kTileJob >= kTileParts: the low bits are free
            }
            run += c;
        }
        if (block == 0 && threadIdx.x == 255u) {
            const unsigned long long total = partial[255] & ((1ull << 40) - 1ull);
            fv.counters->bin_total = (uint32_t)min(total, 0xFFFFFFFFull);
            fv.counters->tile_nonempty = (uint32_t)min(partial[255] >> 40, (unsigned long long)fv.tile_list_capacity);
            if (total > fv.bin_capacity) fv.counters->bin_overflow = 1u;
        }
        __syncthreads();
    }
    // A wave takes kBinChunk records at a time, not 64: records that cover many tiles are walked by
    // the whole wave one after another, and a frame usually has few records but some very wide ones
    // (walls, floors), so small chunks spread those over many more waves.
    const uint32_t wave = (block * blockDim.x + threadIdx.x) >> 6, n_waves = (n_blocks * blockDim.x) >> 6;
    // (a wave's records are taken n_waves apart: wide records come in runs, and a run is dealt out one record to a wave)
    for (uint32_t base_i = 0; base_i < count; base_i += n_waves * kBinChunk) {
        const uint32_t i = base_i + lane * n_waves + wave;
        uint32_t tx0 = 0, ty0 = 0, tw = 1, nt = 0;
        SetupTri t;
        if (lane < kBinChunk && i < count) {
            t = fv.setup[i];
            tx0 = t.px0 / kTile; ty0 = t.py0 / kTile;
            tw = t.px1 / kTile - tx0 + 1u;
            nt = tw * (t.py1 / kTile - ty0 + 1u);
        }
        if (nt <= kLaneTiles)
            for (uint32_t q = 0; q < nt; ++q) {
                const uint32_t tx = tx0 + q % tw, ty = ty0 + q / tw;
                if (!tile_touched(fv, t, tx, ty)) continue;
                const uint32_t tile = ty * fv.tiles_x + tx;
                const uint32_t slot = tile_off[tile] + atomicAdd(&fv.tile_cursor[tile], 1u);
                if (slot < fv.bin_capacity) fv.bin_entries[slot] = i;
            }
        unsigned long long wide = __ballot(nt > kLaneTiles);
        while (wide) {
            const int src = __ffsll((long long)wide) - 1;
            wide &= wide - 1ull;
            const uint32_t record = base_i + (uint32_t)src * n_waves + wave;
            const SetupTri w = fv.setup[record]; // the same address in every lane: one broadcast load
            const uint32_t wx0 = w.px0 / kTile, wy0 = w.py0 / kTile, ww = w.px1 / kTile - wx0 + 1u;
            const uint32_t rows = w.py1 / kTile - wy0 + 1u;
            // a lane per tile row, eight tiles per round: the atomics return a value (the slot), and it is the round
            // trips, not the tests, that a record spanning the screen pays for, so eight are in flight per lane
            constexpr int kPerRound = 8;
            for (uint32_t row = lane; row < rows; row += 64u) {
                const uint32_t ty = wy0 + row;
                if (!tile_row_wanted(fv, w, ty)) continue;
                TileRowWalk r = tile_row_walk(w, wx0, ty);
                const uint32_t tile0 = ty * fv.tiles_x + wx0;
                for (uint32_t i0 = 0; i0 < ww; i0 += kPerRound) {
                    bool touched[kPerRound];
                    uint32_t got[kPerRound];
#pragma unroll
                    for (int k = 0; k < kPerRound; ++k) {
                        touched[k] = i0 + k < ww && tile_row_touched(r);
                        tile_row_next(r);
                    }
#pragma unroll
                    for (int k = 0; k < kPerRound; ++k) got[k] = touched[k] ? atomicAdd(&fv.tile_cursor[tile0 + i0 + k], 1u) : 0u;
#pragma unroll
                    for (int k = 0; k < kPerRound; ++k) {
                        if (!touched[k]) continue;
                        const uint32_t slot = tile_off[tile0 + i0 + k] + got[k];
                        if (slot < fv.bin_capacity) fv.bin_entries[slot] = record;
                    }
                }
            }
        }
    }
}

// The two consumers of k_tile_setup's output in one launch, because neither needs the other: the first
// kFillBlocks workgroups fill the bins (first, so that they are running when the rest of the grid, a whole
// resident round, takes the machine), the other FrameView::grid.mid_blocks scan the mid-sized triangles.
// One dependent launch less (4-5 us when both have nothing to do), and where both have work it overlaps.
constexpr uint32_t kFillBlocks = 512;
__global__ __launch_bounds__(256) void k_mid_fill(SceneView sv, FrameView fv) {
    if (blockIdx.x < kFillBlocks) tile_fill_role(fv, blockIdx.x, kFillBlocks);
    else mid_raster_role(sv, fv, blockIdx.x - kFillBlocks, fv.grid.mid_blocks, fv.grid.slot_mid);
}

// A workgroup takes jobs: kTileJob records of one tile's bin over one kTilePart x kTilePart piece of the
// tile, a thread owning B x B pixels of it.  Small jobs in a one-resident-round grid, because what a tile
// costs is mostly the chain of dependent loads in front of the arithmetic (list -> count / offset -> bin ->
// records ... -> keys), which only other workgroups on the same CU can hide, and because a tile under many
// large alpha-tested triangles would otherwise set the kernel's length alone (interior scene: one
// 256-thread workgroup per tile with 4 x 4 pixels per thread 0.175 ms for the tile path, this 0.128 ms).
// A frame with empty bins pays for the launch of one resident round only.
// What the last kernel of a geometry pass owes the frame besides its own work: the sets of list heads and tile
// counts the next frame will use are zeroed (FrameView::counters_next; spread over the grid), and the pass's status
// goes where the host sees it without a synchronisation.  Every producer of the lists has finished (earlier
// kernels), so the overflow counters are final; with two-phase culling a frame has two passes and the flags and the
// tile-path hint accumulate in the frame's counters, which the reset between the phases (phase_reset) leaves alone.
RDR_DEV void pass_epilogue(const FrameView& fv, uint32_t block, uint32_t n_blocks) {
    const uint32_t n_words = (uint32_t)(sizeof(FrameCounters) / 4u), n_tiles = fv.tiles_x * fv.tiles_y;
    for (uint32_t w = block * blockDim.x + threadIdx.x; w < n_words + n_tiles; w += n_blocks * blockDim.x) {
        if (w < n_words) reinterpret_cast<uint32_t*>(fv.counters_next)[w] = 0u;
        else {
            fv.tile_count_next[w - n_words] = 0u;
            fv.tile_cursor[w - n_words] = 0u; // k_mid_fill is done with it and expects it zero the next time
        }
    }
    if (block == 0 && threadIdx.x == 0) {
        FrameCounters* c = fv.counters;
        uint32_t flags = c->sticky_flags;
        flags |= (c->deferred_overflow || c->rare_timeout) ? RDR_FRAME_DEFERRED_OVERFLOW : 0u;
        flags |= c->setup_overflow ? RDR_FRAME_SETUP_OVERFLOW : 0u;
        flags |= c->bin_overflow ? RDR_FRAME_BIN_OVERFLOW : 0u;
        flags |= c->mid_overflow ? RDR_FRAME_MID_OVERFLOW : 0u;
        flags |= c->frag_overflow ? RDR_FRAME_FRAG_OVERFLOW : 0u;
        flags |= (c->cmd_count > fv.command_capacity || (unsigned long long)c->vis_count + c->vis_mask_count > fv.visible_capacity)
                     ? RDR_FRAME_LIST_OVERFLOW : 0u;
        c->sticky_flags = flags;
        uint32_t work = c->tile_work | (c->deferred_count ? 3u : 0u); // bit 1: triangles for the tile rasteriser itself
        for (uint32_t sgm = 0; sgm < kMidSegments; ++sgm) work |= c->mid_seg[sgm * 32u] ? 1u : 0u;
        c->tile_work = work;
        __hip_atomic_store(&fv.host_status[0], (fv.frame_seq << 8) | flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&fv.host_status[2], (fv.frame_seq << 8) | work, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <int B>
RDR_DEV void tile_raster_role(const SceneView& sv, const FrameView& fv, uint32_t block, uint32_t n_blocks) {
    constexpr int kSide = kTilePart / B, kWaves = kSide * kSide / 64;
    __shared__ SetupTri batch[kTileBatch];
    __shared__ uint32_t batch_index[kTileBatch];
    __shared__ uint32_t wg_frags[kWaves];
    // The three edge functions of a staged record at the centre of the job's first pixel, in double precision (exact:
    // below 2^50 inside the guard band): value, change per pixel in x and in y, and the value the function has to
    // reach at a pixel for the pixel to be inside (0 on top-left edges, 1 on the others).
    __shared__ double edge_lds[kJobGroup * kTileJob][3][4];
    __shared__ uint4 job_lds[kJobGroup]; // tile, first record | part, first bin entry, records | bit 31: other jobs work on the same pixels
    static_assert(kJobGroup * kTileJob <= kTileBatch && kSide * kSide == 64 * (int)kJobGroup, "a group of jobs is staged at once, 64 threads to a job");
    uint32_t frags = 0;
    const double block_x = (double)((threadIdx.x % kSide) * B), block_y = (double)((threadIdx.x / kSide) * B); // this thread's block in the piece
    const uint32_t n_listed = min(fv.counters->tile_nonempty, fv.tile_list_capacity);
    // Jobs are dealt out by index, a grid's width apart: a shared counter would cost an atomic that returns a
    // value per job, and those serialise on their address (10 000 jobs: 0.1 ms).  A workgroup fetches kJobGroup of
    // its jobs at a time: what a job costs is mostly the chain of dependent loads in front of its arithmetic (list
    // -> count, offset -> bin -> records), and the jobs of a group walk that chain side by side.
    for (uint32_t li0 = block; li0 < n_listed; li0 += n_blocks * kJobGroup) {
        __syncthreads(); // the previous group has been consumed
        if (threadIdx.x < kJobGroup) {
            const uint32_t li = li0 + threadIdx.x * n_blocks;
            uint4 j = make_uint4(0u, 0u, 0u, 0u);
            if (li < n_listed) {
                const uint2 job = fv.tile_list[li];
                const uint32_t n_all = fv.tile_count[job.x], tile_off = fv.tile_offset[job.x];
                const uint32_t n_tile = tile_off >= fv.bin_capacity ? 0u : min(n_all, fv.bin_capacity - tile_off); // records that made it into the bin
                const uint32_t first = job.y & ~(kTileJob - 1u);
                j = make_uint4(job.x, job.y, tile_off + first, (first < n_tile ? min(kTileJob, n_tile - first) : 0u) | (n_all > kTileJob ? 0x80000000u : 0u));
            }
            job_lds[threadIdx.x] = j;
        }
        __syncthreads();
        {   // stage the group's records: 64 threads per job, 4 per 64-byte record (coalesced 16-byte loads)
            const uint32_t jk = threadIdx.x >> 6, r = (threadIdx.x & 63u) >> 2, part = threadIdx.x & 3u;
            const uint4 j = job_lds[jk];
            if (r < (j.w & 0x7FFFFFFFu)) {
                const uint32_t idx = fv.bin_entries[j.z + r], piece = j.y & (kTileJob - 1u);
                const int piece_x0 = (int)(j.x % fv.tiles_x) * kTile + (int)(piece % (kTile / kTilePart)) * kTilePart;
                const int piece_y0 = (int)(j.x / fv.tiles_x) * kTile + (int)(piece / (kTile / kTilePart)) * kTilePart;
                const SetupTri* src = &fv.setup[idx];
                reinterpret_cast<uint4*>(&batch[jk * kTileJob + r])[part] = reinterpret_cast<const uint4*>(src)[part];
                if (part == 0) batch_index[jk * kTileJob + r] = idx;
                if (part < 3u) { // three of the four set up an edge each: 0 from vertex 0 to 1 (the weight of vertex 2), 1 from 1 to 2, 2 from 2 to 0
                    const int32_t* xy = &src->X0;
                    const uint32_t a = part, b = part == 2u ? 0u : part + 1u;
                    const int32_t Xa = xy[2u * a], Ya = xy[2u * a + 1u], dx = xy[2u * b] - Xa, dy = xy[2u * b + 1u] - Ya;
                    double* e = edge_lds[jk * kTileJob + r][part];
                    e[0] = (double)dx * (double)(piece_y0 * 256 + 128 - Ya) - (double)dy * (double)(piece_x0 * 256 + 128 - Xa);
                    e[1] = -(double)dy * 256.0;
                    e[2] = (double)dx * 256.0;
                    e[3] = (dy < 0 || (dy == 0 && dx > 0)) ? 0.0 : 1.0;
                }
            }
        }
        __syncthreads();
    for (uint32_t jk = 0; jk < kJobGroup; ++jk) {
        const uint4 this_job = job_lds[jk];
        const uint32_t tile = this_job.x, part = this_job.y & (kTileJob - 1u), n = this_job.w & 0x7FFFFFFFu, rec0 = jk * kTileJob;
        const bool shared_tile = (this_job.w >> 31) != 0u; // other jobs work on the same pixels: merge with atomics
        if (n == 0u) continue;
        // the job's part of the tile, and this thread's B x B pixel block in it
        const int tile_x0 = (int)(tile % fv.tiles_x) * kTile + (int)(part % (kTile / kTilePart)) * kTilePart;
        const int tile_y0 = (int)(tile / fv.tiles_x) * kTile + (int)(part / (kTile / kTilePart)) * kTilePart;
        const int bx0 = tile_x0 + (int)(threadIdx.x % kSide) * B, by0 = tile_y0 + (int)(threadIdx.x / kSide) * B;
        const bool block_on_screen = bx0 < (int)fv.width && by0 < (int)fv.height;
        // triangles every sort-last rank draws are rasterised on this rank's rows only (a block's rows lie in one band)
        const bool rows_mine = rdr_rows_owned(fv, (uint32_t)by0, (uint32_t)min(by0 + B - 1, (int)fv.height - 1));
        unsigned long long keys[B * B];
        uint32_t on_screen = 0; // bit j * B + i: pixel (bx0 + i, by0 + j) exists
#pragma unroll
        for (int k = 0; k < B * B; ++k) {
            keys[k] = RDR_VIS_CLEAR;
            on_screen |= (bx0 + (k % B) < (int)fv.width && by0 + (k / B) < (int)fv.height) ? (1u << k) : 0u;
        }

        if (block_on_screen) {
            for (uint32_t k = 0; k < n; ++k) {
                const SetupTri& t = batch[rec0 + k];
                if ((int)t.px1 < bx0 || (int)t.px0 > bx0 + (B - 1) || (int)t.py1 < by0 || (int)t.py0 > by0 + (B - 1)) continue;
                if ((t.masked & 4u) && !rows_mine) continue;
                const float z0 = t.z0, dz1 = t.dz1, dz2 = t.dz2, inv_area = t.inv_area;
                const uint32_t id = t.id;
                if (!(t.masked & 2u)) {
                    // opaque: branch-free over the block's pixels; the depth is evaluated everywhere and kept where covered.
                    // The edge functions run in double precision from the staged set-up: the same integers as the 64-bit
                    // form, exactly, with one instruction per addition and per conversion to float.
                    const double* ea = edge_lds[rec0 + k][0];
                    const double* eb = edge_lds[rec0 + k][1];
                    const double* ec = edge_lds[rec0 + k][2];
                    const double a0 = fma(ea[1], block_x, fma(ea[2], block_y, ea[0])), b0 = fma(eb[1], block_x, fma(eb[2], block_y, eb[0])),
                                 c0 = fma(ec[1], block_x, fma(ec[2], block_y, ec[0]));
#pragma unroll
                    for (int j = 0; j < B; ++j) {
#pragma unroll
                        for (int i = 0; i < B; ++i) {
                            // (sums of integers below 2^53: exact in any order)
                            const double v01 = i ? (j ? (a0 + ea[1]) + ea[2] : a0 + ea[1]) : (j ? a0 + ea[2] : a0);
                            const double v12 = i ? (j ? (b0 + eb[1]) + eb[2] : b0 + eb[1]) : (j ? b0 + eb[2] : b0);
                            const double v20 = i ? (j ? (c0 + ec[1]) + ec[2] : c0 + ec[1]) : (j ? c0 + ec[2] : c0);
                            static_assert(B == 2, "the four pixels of a 2 x 2 block");
                            const bool in = (v01 >= ea[3]) & (v12 >= eb[3]) & (v20 >= ec[3]) & (bool)((on_screen >> (j * B + i)) & 1u);
                            const float l1 = (float)v20 * inv_area, l2 = (float)v01 * inv_area;
                            const float z = clampf(fmaf(l2, dz2, fmaf(l1, dz1, z0)), 0.0f, 1.0f);
                            const unsigned long long key = ((unsigned long long)(0x7FFFFFFFu - __float_as_uint(z)) << 32) | (unsigned long long)id;
                            keys[j * B + i] = (in && key < keys[j * B + i]) ? key : keys[j * B + i];
                            frags += in ? 1u : 0u;
                        }
                    }
                    continue;
                }
                const EdgeEq e01 = make_edge(t.X0, t.Y0, t.X1, t.Y1, bx0, by0); // weight of v2
                const EdgeEq e12 = make_edge(t.X1, t.Y1, t.X2, t.Y2, bx0, by0); // weight of v0
                const EdgeEq e20 = make_edge(t.X2, t.Y2, t.X0, t.Y0, bx0, by0); // weight of v1
                if (edge_max(e01, B - 1, B - 1) + e01.bias < 0 || edge_max(e12, B - 1, B - 1) + e12.bias < 0 || edge_max(e20, B - 1, B - 1) + e20.bias < 0) continue;
                const AlphaTest at = make_alpha_test(sv, uniform_ref(&sv.draws[__builtin_amdgcn_readfirstlane(t.draw_id)])); // one record for all threads
                const SetupMask* mk = (t.masked & 1u) ? &fv.setup_mask[batch_index[rec0 + k]] : nullptr;
                // alpha-tested: coverage of the block first, then the texture is consulted only where it matters
                uint32_t cov = 0;
#pragma unroll
                for (int j = 0; j < B; ++j) {
#pragma unroll
                    for (int i = 0; i < B; ++i) {
                        const long long v01 = e01.e + e01.ax * i + e01.by * j;
                        const long long v12 = e12.e + e12.ax * i + e12.by * j;
                        const long long v20 = e20.e + e20.ax * i + e20.by * j;
                        cov |= (((v01 + e01.bias) | (v12 + e12.bias) | (v20 + e20.bias)) >= 0) ? (1u << (j * B + i)) : 0u;
                    }
                }
                cov &= on_screen;
                if (cov == 0u) continue;
                const bool textured = at.enabled && at.has_tex && mk;
                SnapVert a{}, b{}, c{};
                if (textured) {
                    a.inv_w = mk->inv_w[0]; b.inv_w = mk->inv_w[1]; c.inv_w = mk->inv_w[2];
                    a.u0 = mk->uv[0][0]; a.v0 = mk->uv[0][1]; a.u1 = mk->uv[0][2]; a.v1 = mk->uv[0][3];
                    b.u0 = mk->uv[1][0]; b.v0 = mk->uv[1][1]; b.u1 = mk->uv[1][2]; b.v1 = mk->uv[1][3];
                    c.u0 = mk->uv[2][0]; c.v0 = mk->uv[2][1]; c.u1 = mk->uv[2][2]; c.v1 = mk->uv[2][3];
                }
                // Perspective-correct uv on the block's (B + 1)^2 sample grid, a row at a time: a pixel's forward
                // differences are its right and lower neighbours' own uv (25 evaluations serve 16 pixels at B = 4).
                auto uv_row = [&](int j, float* ru, float* rv) {
#pragma unroll
                    for (int i = 0; i <= B; ++i) {
                        float u, v;
                        persp_uv<long long>(a, b, c, e12.e + e12.ax * i + e12.by * j, e20.e + e20.ax * i + e20.by * j,
                                            e01.e + e01.ax * i + e01.by * j, at.uv_set, u, v);
                        ru[i] = u * at.sx + at.ox;
                        rv[i] = v * at.sy + at.oy;
                    }
                };
                float ru[B + 1] = {}, rv[B + 1] = {}, nu[B + 1] = {}, nv[B + 1] = {};
                if (textured) uv_row(0, ru, rv);
#pragma unroll
                for (int j = 0; j < B; ++j) {
                    if (textured) uv_row(j + 1, nu, nv);
#pragma unroll
                    for (int i = 0; i < B; ++i) {
                        if (!((cov >> (j * B + i)) & 1u)) continue;
                        if (at.enabled) {
                            float alpha = 1.0f;
                            if (textured)
                                alpha = image_sample_grad(uniform_ref(&sv.images[at.image]), sv.texels, sv.srgb_lut, ru[i], rv[i], ru[i + 1] - ru[i],
                                                          rv[i + 1] - rv[i], nu[i] - ru[i], nv[i] - rv[i]).w;
                            alpha = alpha * at.factor_a;
                            if (alpha < at.cutoff) continue;
                        }
                        const long long v01 = e01.e + e01.ax * i + e01.by * j, v20 = e20.e + e20.ax * i + e20.by * j;
                        const float l1 = edge_to_float(v20) * inv_area, l2 = edge_to_float(v01) * inv_area;
                        const float z = clampf(fmaf(l2, dz2, fmaf(l1, dz1, z0)), 0.0f, 1.0f);
                        const unsigned long long key = ((unsigned long long)(0x7FFFFFFFu - __float_as_uint(z)) << 32) | (unsigned long long)id;
                        keys[j * B + i] = min(keys[j * B + i], key);
                        frags++;
                    }
#pragma unroll
                    for (int i = 0; i <= B; ++i) { ru[i] = nu[i]; rv[i] = nv[i]; }
                }
            }
        }
        // merge the block into the visibility buffer: rows of B keys, contiguous
        if (block_on_screen) {
            bool any = false;
#pragma unroll
            for (int j = 0; j < B; ++j) {
                if (by0 + j >= (int)fv.height) continue;
                unsigned long long* row = fv.vis + (size_t)(by0 + j) * fv.width + bx0;
#pragma unroll
                for (int i = 0; i < B; ++i) {
                    if (bx0 + i >= (int)fv.width) continue;
                    const unsigned long long k = keys[j * B + i];
                    if (k == RDR_VIS_CLEAR) continue;
                    any = true;
                    if (shared_tile) atomicMin(&row[i], k);
                    else if (k < row[i]) row[i] = k;
                }
            }
            if (any) touch_pixel(fv, bx0, by0); // B divides the tile edge: the block lies in one tile
        }
    }
    }
    if (fv.collect_stats) {
        const uint32_t f = wave_sum(frags);
        if ((threadIdx.x & 63u) == 0) wg_frags[threadIdx.x >> 6] = f;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t total = 0;
            for (int w = 0; w < kWaves; ++w) total += wg_frags[w];
            stat_store(fv, fv.grid.slot_deferred + block, total);
        }
    }
}

template <int B>
__global__ __launch_bounds__((kTilePart / B) * (kTilePart / B)) void k_tile_raster(SceneView sv, FrameView fv) {
    pass_epilogue(fv, blockIdx.x, gridDim.x);
    tile_raster_role<B>(sv, fv, blockIdx.x, gridDim.x);
}

// The tile path as one launch, for frames that are not expected to need it (the frame before had neither a deferred
// triangle nor a mid-sized one: the host's hint, FrameCounters::tile_work): every workgroup looks at the frame's own
// counters and leaves at once when they are empty, which makes the all-small-triangles frame three dependent
// launches shorter (4-5 us each).  If this frame does have such triangles after all, the same three steps run here
// behind grid-wide barriers -- slower than the three launches (a small grid, barrier round trips), but correct, and
// the next frame is launched the usual way.  The grid (FrameView::grid.rare_blocks, a fraction of what fits the
// device) is resident as a whole, so the barriers cannot wait for a workgroup that has not started, unless other
// streams keep the device full for as long as kRareWaitTicks: then the pass gives up and the frame is reported
// as having dropped work (FrameCounters::rare_timeout), never hung.
constexpr long long kRareWaitTicks = 200000000; // wall_clock64 runs at 100 MHz: 2 s
RDR_DEV bool rare_grid_barrier(const FrameView& fv, uint32_t n_blocks, uint32_t& round) {
    __shared__ uint32_t ok;
    __threadfence(); // this thread's writes, device-wide
    __syncthreads();
    if (threadIdx.x == 0) {
        round += 1u;
        __hip_atomic_fetch_add(&fv.counters->rare_barrier, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const long long start = wall_clock64();
        uint32_t good = 1u;
        while (__hip_atomic_load(&fv.counters->rare_barrier, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < round * n_blocks) {
            if (__hip_atomic_load(&fv.counters->rare_timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || wall_clock64() - start > kRareWaitTicks) {
                __hip_atomic_store(&fv.counters->rare_timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0u;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        ok = good;
    }
    __syncthreads();
    __threadfence(); // the other workgroups' writes, for every thread of this one
    return ok != 0u;
}

__global__ __launch_bounds__(256) void k_tile_path_rare(SceneView sv, FrameView fv) {
    __shared__ ClipLds clip;
    __shared__ uint32_t any_work;
    if (threadIdx.x < 64) {
        const uint32_t lane = threadIdx.x;
        uint32_t w = lane == 0 ? fv.counters->deferred_count : 0u;
        if (lane < kMidSegments) w |= fv.counters->mid_seg[lane * 32u];
        const unsigned long long m = __ballot(w != 0u);
        if (lane == 0) any_work = m != 0ull ? 1u : 0u;
    }
    // the statistics slots of the kernels this launch stands in for, one per workgroup of this grid in each of their
    // regions: what this pass does not write must read as zero (workgroup b clears slot b and later writes it itself)
    if (fv.collect_stats && !fv.stat_accumulate && threadIdx.x == 0) {
        fv.stat_slots[fv.grid.slot_mid + blockIdx.x] = 0u;
        fv.stat_slots[fv.grid.slot_deferred + blockIdx.x] = 0u;
    }
    __syncthreads();
    if (!any_work) { // the expected case
        pass_epilogue(fv, blockIdx.x, gridDim.x);
        return;
    }
    uint32_t round = 0;
    if (threadIdx.x < 64) tile_setup_role<1>(sv, fv, clip, nullptr, blockIdx.x, gridDim.x);
    bool ok = rare_grid_barrier(fv, gridDim.x, round);
    const uint32_t fill_blocks = gridDim.x / 2u;
    if (ok) {
        if (blockIdx.x < fill_blocks) tile_fill_role(fv, blockIdx.x, fill_blocks);
        else mid_raster_role(sv, fv, blockIdx.x - fill_blocks, gridDim.x - fill_blocks, fv.grid.slot_mid + fill_blocks); // slot = this workgroup's index
    }
    ok = rare_grid_barrier(fv, gridDim.x, round) && ok;
    // every producer is done (or the pass has given up, which the epilogue reports): as in k_tile_raster
    pass_epilogue(fv, blockIdx.x, gridDim.x);
    if (ok) tile_raster_role<kTileBlock>(sv, fv, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------
// Depth pyramid of the early phase (the reference's depth_reduce.comp.glsl with a MIN
// sampler, renderer/src/renderer/pass.rs:273-393), built from the visibility keys: depth
// bits = 0x7FFFFFFF - (key >> 32), 0 where nothing was drawn.  Positive floats order like
// their bit patterns, so levels hold raw bits and reduce with an integer min.
// k_pyramid_tiles: one workgroup per 64x64 pixels writes levels 0..5 of its region (a thread
// reduces 2x2 pixels at a time, four times); k_pyramid_top: one workgroup finishes the remaining levels.
// ------------------------------------------------------------------------------
// Between the phases: empty the work lists the early phase filled.  Part of k_pyramid_tiles (which runs between the
// phases and touches none of this): workgroup b clears tile b's count, the first one the counters.
RDR_DEV void phase_reset(const FrameView& fv, uint32_t block, uint32_t gi) {
    if (gi == 0) fv.tile_count[block] = 0u;
    if (block != 0u) return;
    if (gi == 0) {
        FrameCounters* c = fv.counters;
        c->cmd_count = 0; c->vis_count = 0; c->vis_mask_count = 0; c->deferred_count = 0;
        c->setup_count = 0; c->bin_total = 0; c->tile_nonempty = 0;
        c->rare_barrier = 0; // the late pass's k_tile_path_rare counts its own arrivals
    }
    if (gi < kMidSegments) fv.counters->mid_seg[gi * 32u] = 0u;
    if (gi < kFragSegments) fv.counters->frag_seg[gi * 16u] = 0ull;
    for (uint32_t t = gi; t < 2u * kRasterSegments; t += 256u) fv.counters->raster_ticket[t / kRasterSegments][(t % kRasterSegments) * 32u] = 0u;
}
__global__ __launch_bounds__(256) void k_pyramid_tiles(FrameView fv) {
    __shared__ uint32_t lds[256];
    phase_reset(fv, blockIdx.y * gridDim.x + blockIdx.x, threadIdx.x);
    const uint32_t tx = threadIdx.x & 15u, ty = threadIdx.x >> 4;
    // Levels 0 and 1: a thread reduces 2x2 pixels (a level-0 texel) at a time, a wave two rows of 64 pixels and the
    // two below them: consecutive lanes read consecutive 16 bytes (two keys).  The four level-0 texels of a level-1
    // texel are in lanes l, l ^ 1, l ^ 32 and l ^ 33.
    const uint32_t cp = threadIdx.x & 31u, rp0 = threadIdx.x >> 5; // pair of columns in the tile, first of this thread's four pairs of rows
    const uint32_t x0 = blockIdx.x * 64u + cp * 2u;
    const bool aligned_pairs = (fv.width & 1u) == 0u; // every pair of keys is 16-byte aligned, and on the screen as a whole
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
        const uint32_t rp = rp0 + 8u * k, y0 = blockIdx.y * 64u + rp * 2u;
        uint32_t m = 0xFFFFFFFFu; // min of the depths = 0x7FFFFFFF - max of the keys' high words (at most 0x7FFFFFFF, the cleared key's)
#pragma unroll
        for (uint32_t r = 0; r < 2u; ++r) {
            if (x0 >= fv.width || y0 + r >= fv.height) continue;
            const unsigned long long* row = fv.vis + (size_t)(y0 + r) * fv.width + x0;
            if (aligned_pairs) {
                const uint4 a = *reinterpret_cast<const uint4*>(row);
                m = min(m, 0x7FFFFFFFu - max(a.y, a.w));
            } else {
                m = min(m, 0x7FFFFFFFu - (uint32_t)(row[0] >> 32));
                if (x0 + 1u < fv.width) m = min(m, 0x7FFFFFFFu - (uint32_t)(row[1] >> 32));
            }
        }
        {
            const uint32_t x = x0 >> 1, y = y0 >> 1;
            if (x < fv.pyr_w[0] && y < fv.pyr_h[0]) fv.pyramid[fv.pyr_offset[0] + y * fv.pyr_w[0] + x] = m;
        }
        uint32_t v = min(m, (uint32_t)__shfl_xor((int)m, 1, 64));
        v = min(v, (uint32_t)__shfl_xor((int)v, 32, 64));
        if (((cp | rp) & 1u) == 0u) { // level 1: 16x16 texels per workgroup
            lds[(rp >> 1) * 16u + (cp >> 1)] = v;
            const uint32_t x = x0 >> 2, y = y0 >> 2;
            if (fv.pyr_levels > 1u && x < fv.pyr_w[1] && y < fv.pyr_h[1]) fv.pyramid[fv.pyr_offset[1] + y * fv.pyr_w[1] + x] = v;
        }
    }
    // levels 2..5: 8x8, 4x4, 2x2, 1x1 texels per workgroup, reduced through LDS
    uint32_t n = 16u; // side of the grid currently held in lds
    __syncthreads();
    for (uint32_t l = 2; l <= 5u && l < fv.pyr_levels; ++l) {
        const uint32_t half = n >> 1;
        uint32_t r = 0xFFFFFFFFu;
        const bool mine = tx < half && ty < half;
        if (mine) {
            const uint32_t* p = &lds[(ty * 2u) * 16u + tx * 2u];
            r = min(min(p[0], p[1]), min(p[16], p[17]));
        }
        __syncthreads();
        if (mine) {
            lds[ty * 16u + tx] = r;
            const uint32_t x = blockIdx.x * half + tx, y = blockIdx.y * half + ty;
            if (x < fv.pyr_w[l] && y < fv.pyr_h[l]) fv.pyramid[fv.pyr_offset[l] + y * fv.pyr_w[l] + x] = r;
        }
        __syncthreads();
        n = half;
    }
}

__global__ __launch_bounds__(1024) void k_pyramid_top(FrameView fv) {
    for (uint32_t l = 6; l < fv.pyr_levels; ++l) {
        const uint32_t w = fv.pyr_w[l], h = fv.pyr_h[l], pw = fv.pyr_w[l - 1], ph = fv.pyr_h[l - 1];
        const uint32_t* src = fv.pyramid + fv.pyr_offset[l - 1];
        uint32_t* dst = fv.pyramid + fv.pyr_offset[l];
        for (uint32_t i = threadIdx.x; i < w * h; i += blockDim.x) {
            const uint32_t x = i % w, y = i / w;
            uint32_t r = 0xFFFFFFFFu;
            for (uint32_t dy = 0; dy < 2; ++dy)
                for (uint32_t dx = 0; dx < 2; ++dx) {
                    const uint32_t sx = 2u * x + dx, sy = 2u * y + dy;
                    if (sx < pw && sy < ph) r = min(r, src[sy * pw + sx]);
                }
            dst[i] = r;
        }
        __syncthreads(); // one workgroup: the level just written is visible to all its threads
    }
}


__global__ __launch_bounds__(256) void k_clear(unsigned long long* vis, size_t n, FrameCounters* counters, uint32_t* tile_count,
                                               uint32_t n_tiles) {
    const size_t gi = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi < n_tiles) tile_count[gi] = 0u;
    const size_t i = gi * 2;
    if (i + 1 < n) {
        *reinterpret_cast<ulonglong2*>(vis + i) = make_ulonglong2(RDR_VIS_CLEAR, RDR_VIS_CLEAR);
    } else if (i < n) {
        vis[i] = RDR_VIS_CLEAR;
    }
    if (blockIdx.x == 0)
        for (uint32_t w = threadIdx.x; w < sizeof(FrameCounters) / 4; w += blockDim.x) reinterpret_cast<uint32_t*>(counters)[w] = 0u;
}

// Packed key -> the reference's visibility-buffer texel
// (assets/shaders/shader.frag.glsl:71-74; clear value renderer/src/renderer/pass.rs:250).
__global__ __launch_bounds__(256) void k_resolve(SceneView sv, FrameView fv, uint32_t* out_xy) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)fv.width * fv.height) return;
    const unsigned long long key = fv.vis[i];
    uint32_t x = 0xFFFFFFFFu, y = 0xFFFFFFFFu;
    if (key != RDR_VIS_CLEAR) {
        const uint32_t id = (uint32_t)key, instance = id >> 7, tri = id & 127u;
        const uint32_t draw_id = sv.instance_draw[instance];
        const rdr_mesh_draw d = sv.draws[draw_id];
        const rdr_mesh_lod L = sv.meshes[d.mesh_index].lods[sv.draw_lod[draw_id]];
        x = ((L.meshlet_offset + (instance - d.meshlet_visibility_offset)) << 8) | tri;
        y = draw_id;
    }
    reinterpret_cast<uint2*>(out_xy)[i] = make_uint2(x, y);
}

} // namespace

// Workgroups of each persistent kernel that are resident at once on the current device: their grids are sized
// to one resident round, so no workgroup waits for a slot (the tile rasteriser's to four when it has work: its
// jobs are uneven, and the hardware then hands the later workgroups to whichever compute unit is free).
int rdr_geometry_residency(GeometryResidency* out, uint32_t fill_lds_bytes) {
    int device = 0, cus = 0, n = 0;
    RDR_HIP(hipGetDevice(&device));
    RDR_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    RDR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_meshlet_cull<kPhaseLate>, 1024, 0));
    out->meshlet_cull = (uint32_t)(n * cus);
    // (the instantiations that keep the sort-last tile flags are held to the same register budgets)
    RDR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (k_raster<false, false>), 64, 0));
    out->raster = (uint32_t)(n * cus);
    RDR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (k_raster<true, false>), 64, 0));
    out->raster_mask = (uint32_t)(n * cus);
    RDR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_mid_fill, 256, fill_lds_bytes));
    out->mid_raster = (uint32_t)(n * cus);
    RDR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_alpha_frag, 256, 0));
    out->alpha_frag = (uint32_t)(n * cus);
    RDR_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_tile_raster<kTileBlock>, (kTilePart / kTileBlock) * (kTilePart / kTileBlock), 0));
    out->tile_raster = (uint32_t)(n * cus);
    {   // the one-launch tile path: static LDS (the clipper's polygon buffer, the rasteriser's batch) + the bin offsets
        hipFuncAttributes attr;
        out->tile_path_rare = 0;
        if (hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(k_tile_path_rare)) == hipSuccess &&
            attr.sharedSizeBytes + fill_lds_bytes <= 65536 &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_tile_path_rare, 256, fill_lds_bytes) == hipSuccess)
            out->tile_path_rare = (uint32_t)(n * cus);
        (void)hipGetLastError();
    }
    return 0;
}

int rdr_launch_geometry(const SceneView& sv, const FrameView& fv, hipStream_t stream, hipEvent_t* ev) {
    const size_t npx = (size_t)fv.width * fv.height;
    // with draws, the visibility buffer is cleared by extra workgroups of the frame's first meshlet cull
    const bool clear_in_cull = sv.draw_count != 0;
    const uint32_t n_tiles = fv.tiles_x * fv.tiles_y;
    const uint32_t clear_blocks = clear_in_cull ? (n_tiles + 255u) / 256u : (uint32_t)((npx / 2 + 1 + 255) / 256);
    const uint32_t cull_clear_blocks = (uint32_t)((npx / 2 + 1 + 1024u * kClearPerThread - 1) / (1024u * kClearPerThread));
    if (ev) RDR_HIP(hipEventRecord(ev[0], stream));
    if (!clear_in_cull) {
        // a scene without draws: nothing else runs, so the clear is a launch of its own and zeroes both sets of heads
        hipLaunchKernelGGL(k_clear, dim3(std::max(clear_blocks, 1u)), dim3(256), 0, stream, fv.vis, npx, fv.counters, fv.tile_count, n_tiles);
        hipLaunchKernelGGL(k_clear, dim3(std::max((n_tiles + 255u) / 256u, 1u)), dim3(256), 0, stream, fv.vis, (size_t)0, fv.counters_next, fv.tile_count_next, n_tiles);
        if (fv.touched) RDR_HIP(hipMemsetAsync(fv.touched, 0, (size_t)fv.touched_count * 4, stream));
    }
    if (ev) RDR_HIP(hipEventRecord(ev[1], stream));
    // one pass of the rasterisers over the current work lists
    auto raster_lists = [&](const FrameView& f, hipEvent_t* mid_event) {
        // persistent grids: the work-list lengths live in device memory
        if (f.touched) hipLaunchKernelGGL((k_raster<false, true>), dim3(f.grid.raster_blocks), dim3(64), 0, stream, sv, f);
        else hipLaunchKernelGGL((k_raster<false, false>), dim3(f.grid.raster_blocks), dim3(64), 0, stream, sv, f);
        if (sv.masked_meshlets) { // a scene without alpha-tested draws has nothing for these two (an empty launch costs 4-5 us)
            if (f.touched) hipLaunchKernelGGL((k_raster<true, true>), dim3(f.grid.raster_mask_blocks), dim3(64), 0, stream, sv, f);
            else hipLaunchKernelGGL((k_raster<true, false>), dim3(f.grid.raster_mask_blocks), dim3(64), 0, stream, sv, f);
            hipLaunchKernelGGL(k_alpha_frag, dim3(f.grid.frag_blocks), dim3(256), 0, stream, sv, f);
        }
        if (f.grid.tile_path_rare && f.grid.rare_blocks) { // the frame before had nothing for these three: one launch that checks
            if (mid_event) (void)hipEventRecord(*mid_event, stream);
            hipLaunchKernelGGL(k_tile_path_rare, dim3(f.grid.rare_blocks), dim3(256), rdr_tile_fill_lds(n_tiles), stream, sv, f);
            return;
        }
        hipLaunchKernelGGL(k_tile_setup, dim3(1024 / kSetupWaves), dim3(64 * kSetupWaves), 0, stream, sv, f); // also feeds k_mid_raster
        hipLaunchKernelGGL(k_mid_fill, dim3(f.grid.mid_blocks + kFillBlocks), dim3(256), rdr_tile_fill_lds(n_tiles), stream, sv, f);
        if (mid_event) (void)hipEventRecord(*mid_event, stream);
        hipLaunchKernelGGL(k_tile_raster<kTileBlock>, dim3(f.grid.deferred_blocks), dim3((kTilePart / kTileBlock) * (kTilePart / kTileBlock)), 0, stream, sv, f);
    };
    if (sv.draw_count && !fv.occlusion) {
        hipLaunchKernelGGL(k_draw_cull<kPhaseSingle>, dim3(fv.grid.draw_blocks), dim3(256), 0, stream, sv, fv);
        if (ev) RDR_HIP(hipEventRecord(ev[2], stream));
        hipLaunchKernelGGL(k_meshlet_cull<kPhaseSingle>, dim3(fv.grid.cull_blocks + cull_clear_blocks), dim3(1024), 0, stream, sv, fv);
        if (ev) RDR_HIP(hipEventRecord(ev[3], stream));
        raster_lists(fv, ev ? &ev[4] : nullptr);
        if (ev) RDR_HIP(hipEventRecord(ev[5], stream));
    } else if (sv.draw_count) {
        // early phase: what last frame's late phase found visible (opaque draws only)
        hipLaunchKernelGGL(k_draw_cull<kPhaseEarly>, dim3(fv.grid.draw_blocks), dim3(256), 0, stream, sv, fv);
        if (ev) RDR_HIP(hipEventRecord(ev[2], stream));
        hipLaunchKernelGGL(k_meshlet_cull<kPhaseEarly>, dim3(fv.grid.cull_blocks + cull_clear_blocks), dim3(1024), 0, stream, sv, fv);
        if (ev) RDR_HIP(hipEventRecord(ev[3], stream));
        raster_lists(fv, ev ? &ev[4] : nullptr);
        // pyramid of its depth, then everything else against the pyramid; timed with the tile path
        hipLaunchKernelGGL(k_pyramid_tiles, dim3(fv.tiles_x, fv.tiles_y), dim3(256), 0, stream, fv);
        hipLaunchKernelGGL(k_pyramid_top, dim3(1), dim3(1024), 0, stream, fv);
        FrameView late = fv;
        late.stat_accumulate = 1u;
        hipLaunchKernelGGL(k_draw_cull<kPhaseLate>, dim3(late.grid.draw_blocks), dim3(256), 0, stream, sv, late);
        hipLaunchKernelGGL(k_meshlet_cull<kPhaseLate>, dim3(late.grid.cull_blocks), dim3(1024), 0, stream, sv, late);
        raster_lists(late, nullptr);
        if (ev) RDR_HIP(hipEventRecord(ev[5], stream));
    }
    RDR_HIP(hipGetLastError());
    return 0;
}

int rdr_launch_resolve(const SceneView& sv, const FrameView& fv, uint32_t* out_xy, hipStream_t stream) {
    const size_t npx = (size_t)fv.width * fv.height;
    hipLaunchKernelGGL(k_resolve, dim3((uint32_t)((npx + 255) / 256)), dim3(256), 0, stream, sv, fv, out_xy);
    RDR_HIP(hipGetLastError());
    return 0;
}
