// rdr_api.hip — the C ABI of include/rdr_abi.h: renderer state, scene upload,
// frame submission and readback.
//
// Mirrors the reference's Renderer (renderer/src/renderer.rs:780-1060):
//   rdr_create            Renderer::new              :793
//   rdr_set_scene         SceneResources::load_scene :426-698
//   rdr_update_transforms upload_transforms          :700-722
//   rdr_render            Renderer::render           :862-1002 with the pass order of :803-843
//   rdr_build_ubo         update_uniform_object      :1148-1197
// There is no CPU fallback: every entry point needs a HIP device.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "rdr_internal.h"

static thread_local std::string g_last_error;
void rdr_set_error(const std::string& msg) { g_last_error = msg; }
extern "C" const char* rdr_last_error(void) { return g_last_error.c_str(); }

namespace {

template <typename T>
struct DeviceBuffer {
    T* ptr = nullptr;
    size_t count = 0;
    ~DeviceBuffer() { release(); }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
    }
    int alloc(size_t n) {
        release();
        count = n;
        RDR_HIP(hipMalloc(reinterpret_cast<void**>(&ptr), std::max<size_t>(n, 1) * sizeof(T)));
        return 0;
    }
    int upload(const T* src, size_t n) {
        if (alloc(n)) return 1;
        if (n) RDR_HIP(hipMemcpy(ptr, src, n * sizeof(T), hipMemcpyHostToDevice));
        return 0;
    }
};

uint32_t mip_dim(uint32_t d, uint32_t l) {
    const uint32_t v = d >> l;
    return v ? v : 1u;
}
size_t texel_bytes(uint32_t format) { return format == RDR_FORMAT_RGBA32F ? 16 : 4; }

} // namespace

struct rdr_renderer {
    rdr_config cfg{};
    uint32_t width = 0, height = 0;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // time_kernels: a ring of per-frame event sets so that timing never forces a host sync
    static constexpr int kRing = 64;
    static constexpr int kEvents = 10;
    std::vector<hipEvent_t> ring; // kRing * kEvents
    bool slot_geometry[kRing]{}, slot_shade[kRing]{};
    uint64_t frames_started = 0; // frames since the last rdr_get_stats
    int cur_slot = 0;
    bool frame_rendered = false, shaded = false;
    bool force_tile_flags = false; // measurement hook RDR_TILE_FLAGS=1: keep the flags without an exchange
    bool frame_has_tile_flags = false; // the visibility buffer on the device was rendered with FrameView::touched kept
    bool timing_this_frame = false;

    // frame resources
    DeviceBuffer<unsigned long long> vis;
    DeviceBuffer<uint8_t> color;
    DeviceBuffer<FrameCounters> counters;
    DeviceBuffer<CullCommand> commands;
    DeviceBuffer<VisibleMeshlet> visible;
    DeviceBuffer<DeferredTriangle> deferred;
    DeviceBuffer<SetupTri> setup;
    DeviceBuffer<SetupMask> setup_mask;
    DeviceBuffer<SetupTri> mid;
    DeviceBuffer<SetupMask> mid_mask;
    DeviceBuffer<uint32_t> tile_count, tile_offset, tile_cursor, bin_entries;
    DeviceBuffer<uint2> tile_list;
    DeviceBuffer<uint32_t> resolve_out;
    DeviceBuffer<float> srgb_lut;
    DeviceBuffer<uint8_t> shadow;
    DeviceBuffer<uint32_t> pixel_list;
    DeviceBuffer<uint32_t> pixel_key;
    DeviceBuffer<FragTri> frag_tri;
    DeviceBuffer<uint32_t> frag_px;
    DeviceBuffer<uint32_t> stat_slots;

    // scene resources
    DeviceBuffer<rdr_vertex> vertices;
    DeviceBuffer<float> pos_x, pos_y, pos_z;
    DeviceBuffer<float4> uv_sets;
    DeviceBuffer<float> mpos_x, mpos_y, mpos_z; // positions and uv sets in meshlet_vertices order (SceneView::mpos_x)
    DeviceBuffer<float4> muv;
    bool meshlet_order = false;                 // those four are valid
    DeviceBuffer<uint32_t> meshlet_vertices;
    DeviceBuffer<uint8_t> meshlet_triangles;
    DeviceBuffer<rdr_meshlet> meshlets;
    DeviceBuffer<rdr_mesh> meshes;
    DeviceBuffer<rdr_mesh_draw> draws;
    DeviceBuffer<rdr_material> materials;
    DeviceBuffer<float> transforms;
    DeviceBuffer<uint32_t> instance_draw;
    DeviceBuffer<AlphaTest> draw_alpha;
    DeviceBuffer<uint8_t> draw_lod;
    DeviceBuffer<DrawFrame> draw_frame;
    DeviceBuffer<float> instance_reach;
    DeviceBuffer<uint8_t> sun_cand_count;
    DeviceBuffer<float4> world_ball;
    float sun_cand_dir[3] = {0, 0, 0}; // the sun direction the candidates were computed for
    bool sun_cand_valid = false;       // ... with the current transforms
    uint32_t sun_candidates = 1;       // measurement hook RDR_SUN_CANDIDATES=0: every shadow ray walks the top level
    DeviceBuffer<TriVertices> tri_table;
    DeviceBuffer<uint8_t> draw_visible;
    DeviceBuffer<uint32_t> meshlet_bits;
    DeviceBuffer<uint32_t> pyramid;
    bool occlusion = false; // cfg.cull_enabled, unless RDR_OCCLUSION=0
    DeviceBuffer<GpuImage> images;
    DeviceBuffer<uint8_t> texels;
    DeviceBuffer<BvhWide> bvh_dev; // bottom levels followed by room for the top level
    DeviceBuffer<uint32_t> tlas_items;
    DeviceBuffer<float> blas_tris;
    DeviceBuffer<GpuInstance> instances;
    DeviceBuffer<float4> instance_ball;

    // host copies needed to rebuild the TLAS when transforms change
    std::vector<rdr_mesh_draw> h_draws;
    std::vector<rdr_mesh> h_meshes;
    HostBvh bvh;
    bool has_scene = false;

    SceneView sv{};
    rdr_ubo ubo{};
    uint32_t part_lo = 0, part_hi = 0xFFFFFFFFu;
    uint32_t row_lo = 0, row_hi = 0xFFFFFFFFu;
    uint32_t part_chunk = 64, part_rank = 0, part_world = 1; // rdr_set_partition_interleaved
    float replicate_pixels = 0.0f;                           // rdr_set_partition_replication
    DeviceBuffer<float> mesh_tri_area;
    uint32_t band_shift = 5, band_rank = 0, band_world = 1;  // rdr_set_shade_bands
    uint32_t meshlet_visibility_count = 0;
    uint32_t lds_stack_limit = 24;
    GeometryResidency resident{};
    uint32_t lane_max_pixels = kLaneMaxPixels, mid_max_pixels = kMidMaxPixels;
    uint32_t masked_meshlets = 0; // LOD0 meshlet instances of alpha-masked draws
    rdr_stats stats{};
    // frame status words the kernels write (pinned host memory, see FrameView::host_status)
    volatile uint32_t* status_host = nullptr;
    uint32_t* status_dev = nullptr;
    uint32_t frame_seq = 0;
    uint32_t seen_status[2] = {0, 0};
    uint32_t debug_deferred_capacity = 0xFFFFFFFFu; // test hook: smaller deferred list
    rdr_composite_state* composite = nullptr; // peer-to-peer sort-last exchange, created on first use
    GridPlan last_grid{}; // launch geometry of the last rdr_render_visbuffer (which form of the tile path ran: the statistics slots follow it)
    uint32_t tile_path_rare_mode = 1; // RDR_TILE_PATH_RARE: 0 = always the three tile-path launches, 1 = the one-launch form after a frame without such work, 2 = always the one-launch form (tests)
    uint32_t shadow_defer = 1; // measurement hook RDR_SHADOW_DEFER=0: own-instance walks stay inside k_shadow
    uint32_t convex_skip = 1; // measurement hook RDR_CONVEX_SKIP=0: trace every shadow ray through its own instance; bit 1: inner-ball shortcut
};

namespace {

// ---- UBO -------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// glam Vec3::normalize(): multiply by the reciprocal of the length
V3 normalize_recip(V3 a) {
    const float r = 1.0f / std::sqrt(dot(a, a));
    return {a.x * r, a.y * r, a.z * r};
}
uint32_t power_2_floor(uint32_t x) { // renderer/src/math.rs:17-25
    uint32_t r = 1;
    while (r * 2 < x) r *= 2;
    return r;
}

void build_ubo(uint32_t width, uint32_t height, const float* cam_pos, const float* cam_dir, const float* sun_dir,
               uint32_t draw_count, uint32_t lod_enabled, uint32_t cull_enabled, rdr_ubo* u) {
    std::memset(u, 0, sizeof(*u));
    const V3 eye{cam_pos[0], cam_pos[1], cam_pos[2]};
    // Camera::view_mat = look_to_lh(position, -direction, Y) (scenery/src/camera.rs:12-14);
    // glam's look_to_lh(e, d, up) is look_to_rh(e, -d, up), so the forward axis is +direction.
    const V3 f = normalize_recip(V3{cam_dir[0], cam_dir[1], cam_dir[2]});
    const V3 s = normalize_recip(cross(f, V3{0.0f, 1.0f, 0.0f}));
    const V3 v = cross(s, f);
    float* V = u->view;
    V[0] = s.x; V[1] = v.x; V[2] = -f.x;
    V[4] = s.y; V[5] = v.y; V[6] = -f.y;
    V[8] = s.z; V[9] = v.z; V[10] = -f.z;
    V[12] = -dot(eye, s); V[13] = -dot(eye, v); V[14] = dot(eye, f); V[15] = 1.0f;

    // perspective_infinite_reverse_lh(45 deg, aspect, 0.01) then proj[0][0] *= -1 (renderer.rs:1160-1167)
    const float near_plane = 0.01f;
    const float fov = 45.0f * (3.14159265358979323846f / 180.0f);
    const float aspect = (float)width / (float)height;
    const float sn = std::sin(0.5f * fov), cs = std::cos(0.5f * fov);
    const float h = cs / sn;
    const float w = h / aspect;
    float* P = u->proj;
    P[0] = -w; P[5] = h; P[11] = 1.0f; P[14] = near_plane;

    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r)
            u->view_proj[c * 4 + r] =
                ((P[0 * 4 + r] * V[c * 4 + 0] + P[1 * 4 + r] * V[c * 4 + 1]) + P[2 * 4 + r] * V[c * 4 + 2]) + P[3 * 4 + r] * V[c * 4 + 3];

    u->camera_position[0] = eye.x; u->camera_position[1] = eye.y; u->camera_position[2] = eye.z;
    // normalize_plane(row3 + row0), normalize_plane(row3 + row1) (renderer.rs:1171-1177, math.rs:3-5)
    float fx[3], fy[3];
    for (int c = 0; c < 3; ++c) {
        fx[c] = P[c * 4 + 3] + P[c * 4 + 0];
        fy[c] = P[c * 4 + 3] + P[c * 4 + 1];
    }
    const float lx = std::sqrt(dot(V3{fx[0], fx[1], fx[2]}, V3{fx[0], fx[1], fx[2]}));
    const float ly = std::sqrt(dot(V3{fy[0], fy[1], fy[2]}, V3{fy[0], fy[1], fy[2]}));
    u->frustum[0] = fx[0] / lx; u->frustum[1] = fx[2] / lx;
    u->frustum[2] = fy[1] / ly; u->frustum[3] = fy[2] / ly;
    u->near_plane = near_plane;
    u->far_plane = 500.0f;
    u->lod_enabled = lod_enabled ? 1u : 0u;
    u->cull_enabled = cull_enabled ? 1u : 0u;
    u->draw_count = draw_count;
    u->depth_pyramid_width = (float)power_2_floor(width);
    u->depth_pyramid_height = (float)power_2_floor(height);
    u->lod_target = (1.0f / P[5]) * (1.0f / (float)height);
    u->screen_width = (float)width;
    u->screen_height = (float)height;
    u->sun_direction[0] = sun_dir[0]; u->sun_direction[1] = sun_dir[1]; u->sun_direction[2] = sun_dir[2];
}

// ---- scene validation ---------------------------------------------------------------
int validate_scene(const rdr_scene_desc& s) {
    RDR_CHECK(s.draw_count == 0 || (s.mesh_draws && s.meshes && s.global_transforms), "scene: null draw/mesh/transform array");
    RDR_CHECK(s.meshlet_count == 0 || (s.meshlets && s.meshlet_vertices && s.meshlet_triangles && s.vertices), "scene: null geometry array");
    RDR_CHECK(s.meshlet_visibility_count < (1u << 25), "scene: more than 2^25 meshlet instances do not fit the 32-bit triangle id");
    for (uint32_t mi = 0; mi < s.mesh_count; ++mi) {
        const rdr_mesh& m = s.meshes[mi];
        RDR_CHECK(m.lod_count >= 1 && m.lod_count <= 8, "scene: mesh lod_count out of range");
        RDR_CHECK((uint64_t)m.vertex_offset + m.vertex_count <= s.vertex_count, "scene: mesh vertex range out of bounds");
        RDR_CHECK(m.index_offset <= s.index_count, "scene: mesh index_offset out of bounds");
        if (mi + 1 < s.mesh_count) RDR_CHECK(s.meshes[mi + 1].index_offset >= m.index_offset, "scene: mesh index_offset not monotone");
        for (uint32_t l = 0; l < m.lod_count; ++l) {
            const rdr_mesh_lod& L = m.lods[l];
            RDR_CHECK((uint64_t)L.meshlet_offset + L.meshlet_count <= s.meshlet_count, "scene: lod meshlet range out of bounds");
            RDR_CHECK(L.meshlet_count <= m.lods[0].meshlet_count, "scene: a coarser lod has more meshlets than lod 0");
            for (uint32_t k = 0; k < L.meshlet_count; ++k) {
                const rdr_meshlet& ml = s.meshlets[L.meshlet_offset + k];
                RDR_CHECK(ml.vertex_count <= 64 && ml.triangle_count <= 124, "scene: meshlet exceeds 64 vertices / 124 triangles");
                RDR_CHECK((uint64_t)ml.vertex_offset + ml.vertex_count <= s.meshlet_vertex_count, "scene: meshlet vertex range out of bounds");
                RDR_CHECK((uint64_t)ml.triangle_offset + 3u * ml.triangle_count <= s.meshlet_triangle_bytes, "scene: meshlet triangle range out of bounds");
                for (uint32_t v = 0; v < ml.vertex_count; ++v)
                    RDR_CHECK(s.meshlet_vertices[ml.vertex_offset + v] < m.vertex_count, "scene: meshlet vertex id outside its mesh");
                for (uint32_t b = 0; b < 3u * ml.triangle_count; ++b)
                    RDR_CHECK(s.meshlet_triangles[ml.triangle_offset + b] < ml.vertex_count, "scene: meshlet triangle index outside the meshlet");
            }
        }
        const uint32_t end = mi + 1 < s.mesh_count ? s.meshes[mi + 1].index_offset : s.index_count;
        for (uint32_t i = m.index_offset; i < end; ++i) RDR_CHECK(s.indices[i] < m.vertex_count, "scene: index outside its mesh");
    }
    uint32_t expect = 0;
    for (uint32_t di = 0; di < s.draw_count; ++di) {
        const rdr_mesh_draw& d = s.mesh_draws[di];
        RDR_CHECK(d.mesh_index < s.mesh_count, "scene: draw mesh_index out of bounds");
        RDR_CHECK(d.model_index < s.transform_count, "scene: draw model_index out of bounds");
        RDR_CHECK(d.alpha_mode <= 1, "scene: draw alpha_mode must be 0 (opaque) or 1 (mask)");
        RDR_CHECK(d.meshlet_visibility_offset >= expect, "scene: meshlet_visibility_offset ranges overlap or are not in draw order");
        expect = d.meshlet_visibility_offset + s.meshes[d.mesh_index].lods[0].meshlet_count;
        RDR_CHECK(expect >= d.meshlet_visibility_offset, "scene: meshlet_visibility_offset overflow");
    }
    RDR_CHECK(expect <= s.meshlet_visibility_count, "scene: meshlet_visibility_count smaller than the draws need");
    for (uint32_t i = 0; i < s.image_count; ++i) {
        const rdr_image_desc& im = s.images[i];
        if (!im.data) continue;
        RDR_CHECK(im.width && im.height && im.mip_count >= 1 && im.mip_count <= 17, "scene: image dimensions / mip count invalid");
        RDR_CHECK(im.width <= 32768 && im.height <= 32768, "scene: image larger than 32768 x 32768 (the sampler indexes a level's texels with 31 bits)");
        RDR_CHECK(im.format <= RDR_FORMAT_BC7_SRGB, "scene: unknown image format");
    }
    return 0;
}

struct TexelPacker {
    std::vector<uint8_t> blob;
    // block-compressed images: room for the decoded texels is reserved in the blob and the blocks are
    // expanded on the device once the blob is there
    struct BlockImage {
        rdr_image_desc desc;
        GpuImage image;
    };
    std::vector<BlockImage> block_images;
    size_t append(const void* src, size_t bytes) { // src == nullptr: zero-filled
        const size_t off = (blob.size() + 255) & ~(size_t)255;
        blob.resize(off + bytes);
        if (bytes && src) std::memcpy(blob.data() + off, src, bytes);
        return off;
    }
};

GpuImage pack_image(TexelPacker& tp, const rdr_image_desc& im) {
    GpuImage g{};
    if (!im.data) return g; // mip_count 0: samples as zero
    const bool blocks = im.format >= RDR_FORMAT_BC1;
    g.width = im.width; g.height = im.height; g.mip_count = im.mip_count;
    g.format = blocks ? (im.format == RDR_FORMAT_BC7_SRGB ? RDR_FORMAT_RGBA8_SRGB : RDR_FORMAT_RGBA8_UNORM) : im.format;
    size_t bytes = 0;
    for (uint32_t l = 0; l < im.mip_count; ++l) {
        g.level_offset[l] = (uint32_t)bytes;
        bytes += (size_t)mip_dim(im.width, l) * mip_dim(im.height, l) * texel_bytes(g.format);
    }
    g.offset = tp.append(blocks ? nullptr : im.data, bytes);
    if (blocks) tp.block_images.push_back({im, g});
    return g;
}

// Expands the block-compressed images into the places pack_image reserved in the uploaded blob.
int decode_block_images(const TexelPacker& tp, uint8_t* texels_dev) {
    for (const TexelPacker::BlockImage& b : tp.block_images) {
        size_t total = 0;
        for (uint32_t l = 0; l < b.desc.mip_count; ++l)
            total += rdr_bc_level_bytes(b.desc.format, mip_dim(b.desc.width, l), mip_dim(b.desc.height, l));
        DeviceBuffer<uint8_t> src;
        if (src.upload(static_cast<const uint8_t*>(b.desc.data), total)) return 1;
        size_t off = 0;
        for (uint32_t l = 0; l < b.desc.mip_count; ++l) {
            const uint32_t w = mip_dim(b.desc.width, l), h = mip_dim(b.desc.height, l);
            uint32_t* out = reinterpret_cast<uint32_t*>(texels_dev + b.image.offset + b.image.level_offset[l]);
            if (rdr_launch_decode_bc(b.desc.format, w, h, src.ptr + off, out, nullptr)) return 1;
            off += rdr_bc_level_bytes(b.desc.format, w, h);
        }
        RDR_HIP(hipDeviceSynchronize()); // src is released at the end of the iteration
    }
    return 0;
}

int pack_cubemap(TexelPacker& tp, const rdr_cubemap_desc* cm, GpuCubemap* out) {
    static const uint8_t black[6 * 3 * 3 * 4] = {0}; // six bordered 1x1 faces
    GpuCubemap g{};
    if (!cm || !cm->data) {
        g.size = 1; g.mip_count = 1; g.format = RDR_FORMAT_RGBA8_UNORM;
        g.offset = tp.append(black, sizeof(black));
        *out = g;
        return 0;
    }
    RDR_CHECK(cm->size && cm->size <= 16384 && cm->mip_count >= 1 && cm->mip_count <= 17, "scene: cubemap dimensions / mip count invalid (faces of at most 16384 x 16384)");
    RDR_CHECK(cm->format == RDR_FORMAT_RGBA8_UNORM || cm->format == RDR_FORMAT_RGBA32F, "scene: cubemap format must be RGBA8_UNORM or RGBA32F");
    std::vector<uint8_t> padded;
    rdr_pack_cubemap(*cm, g, padded);
    g.offset = tp.append(padded.data(), padded.size());
    *out = g;
    return 0;
}

int upload_tlas(rdr_renderer* r) {
    const size_t blas_count = r->bvh.blas_pairs.size();
    RDR_CHECK(blas_count + r->bvh.tlas_pairs.size() <= r->bvh_dev.count, "internal: top-level BVH larger than reserved");
    RDR_HIP(hipMemcpy(r->bvh_dev.ptr + blas_count, r->bvh.tlas_pairs.data(), r->bvh.tlas_pairs.size() * sizeof(BvhWide), hipMemcpyHostToDevice));
    {
        // the top level's items, and behind them room for every instance's sun occluder candidates (k_sun_candidates)
        const size_t n = r->bvh.tlas_items.size();
        std::vector<uint32_t> items(n * (1 + kSunCandidates) + 1, 0u);
        std::copy(r->bvh.tlas_items.begin(), r->bvh.tlas_items.end(), items.begin());
        if (r->tlas_items.upload(items.data(), items.size())) return 1;
        if (r->instance_reach.upload(r->bvh.instance_reach.data(), r->bvh.instance_reach.size())) return 1;
        if (r->sun_cand_count.alloc(std::max<size_t>(n, 1))) return 1;
        RDR_HIP(hipMemset(r->sun_cand_count.ptr, 0xFF, std::max<size_t>(n, 1)));
        if (r->world_ball.alloc(std::max<size_t>(n, 1))) return 1;
        RDR_HIP(hipMemset(r->world_ball.ptr, 0, std::max<size_t>(n, 1) * sizeof(float4)));
        r->sun_cand_valid = false;
        r->sv.sun_cand_base = (uint32_t)n;
    }
    if (r->instances.upload(r->bvh.instances.data(), r->bvh.instances.size())) return 1;
    if (const char* e = std::getenv("RDR_INNER_BALL")) // measurement hook: 0 = no instance has an inner ball
        if (std::atoi(e) == 0)
            for (size_t i = 3; i < r->bvh.instance_ball.size(); i += 4) r->bvh.instance_ball[i] = 0.0f; // the radius only: the centre also serves the outer ball
    if (r->instance_ball.upload(reinterpret_cast<const float4*>(r->bvh.instance_ball.data()), r->bvh.instance_ball.size() / 4)) return 1;
    r->sv.bvh = r->bvh_dev.ptr;
    r->sv.tlas_base = (uint32_t)blas_count;
    r->sv.tlas_items = r->tlas_items.ptr;
    r->sv.sun_cand_items = r->tlas_items.ptr + r->sv.sun_cand_base;
    r->sv.sun_cand_count = r->sun_cand_count.ptr;
    r->sv.world_ball = r->world_ball.ptr;
    r->sv.instance_reach = r->instance_reach.ptr;
    r->sv.instances = r->instances.ptr;
    r->sv.instance_ball = r->instance_ball.ptr;
    return 0;
}

FrameView make_frame_view(rdr_renderer* r) {
    FrameView fv{};
    fv.ubo = r->ubo;
    fv.vis = r->vis.ptr;
    const bool flags = rdr_composite_connected(r->composite) || r->force_tile_flags;
    fv.touched = flags ? reinterpret_cast<uint32_t*>(r->vis.ptr + (size_t)r->width * r->height) : nullptr;
    fv.touched_x = (r->width + kTouchTile - 1) / kTouchTile;
    fv.touched_count = rdr_touched_words(r->width, r->height);
    fv.color = r->color.ptr;
    fv.commands = r->commands.ptr;
    fv.visible = r->visible.ptr;
    fv.deferred = r->deferred.ptr;
    {
        const uint32_t set = r->frame_seq & 1u;
        const size_t n_tiles = r->tile_count.count / 2;
        fv.counters = r->counters.ptr + set;
        fv.counters_next = r->counters.ptr + (set ^ 1u);
        fv.tile_count = r->tile_count.ptr + set * n_tiles;
        fv.tile_count_next = r->tile_count.ptr + (set ^ 1u) * n_tiles;
    }
    fv.width = r->width; fv.height = r->height;
    fv.part_lo = r->part_lo; fv.part_hi = r->part_hi;
    fv.part_chunk = r->part_chunk; fv.part_rank = r->part_rank; fv.part_world = r->part_world;
    fv.replicate_pixels = r->replicate_pixels;
    fv.row_lo = std::min(r->row_lo, r->height);
    fv.row_hi = std::min(r->row_hi, r->height);
    fv.band_shift = r->band_shift; fv.band_rank = r->band_rank; fv.band_world = r->band_world;
    fv.shade_rows = r->band_world > 1u ? rdr_band_rows(r->height, r->band_shift, r->band_rank, r->band_world) : fv.row_hi - fv.row_lo;
    fv.command_capacity = (uint32_t)r->commands.count;
    fv.visible_capacity = (uint32_t)r->visible.count;
    fv.deferred_capacity = (uint32_t)r->deferred.count;
    fv.collect_stats = r->cfg.collect_stats;
    fv.shadow = r->shadow.ptr;
    fv.pixel_list = r->pixel_list.ptr;
    fv.pixel_key = r->pixel_key.ptr;
    fv.lds_stack_limit = r->lds_stack_limit;
    fv.convex_skip = r->convex_skip;
    fv.shadow_defer = r->shadow_defer;
    {
        // normalize3() and div_exact() of rdr_device.h / rdr_pixel.h, term for term (this file is built without contraction too)
        const float x = fv.ubo.sun_direction[0], y = fv.ubo.sun_direction[1], z = fv.ubo.sun_direction[2];
        const float l = std::sqrt((x * x + y * y) + z * z);
        fv.sun_unit[0] = x / l; fv.sun_unit[1] = y / l; fv.sun_unit[2] = z / l;
        fv.two_over_width = 2.0f / fv.ubo.screen_width;
        fv.two_over_height = 2.0f / fv.ubo.screen_height;
        fv.inv_width = 1.0f / (float)fv.width;
        fv.inv_height = 1.0f / (float)fv.height;
    }
    fv.sun_candidates = r->sun_candidates;
    fv.host_status = r->status_dev;
    fv.frame_seq = r->frame_seq & 0xFFFFFFu;
    fv.deferred_capacity = std::min(fv.deferred_capacity, r->debug_deferred_capacity);
    fv.setup = r->setup.ptr;
    fv.setup_mask = r->setup_mask.ptr;
    fv.tile_offset = r->tile_offset.ptr;
    fv.tile_cursor = r->tile_cursor.ptr;
    fv.tile_list = r->tile_list.ptr;
    fv.tile_list_capacity = (uint32_t)r->tile_list.count;
    fv.bin_entries = r->bin_entries.ptr;
    fv.setup_capacity = (uint32_t)r->setup.count;
    fv.mid = r->mid.ptr;
    fv.mid_mask = r->mid_mask.ptr;
    fv.mid_capacity = (uint32_t)r->mid.count;
    fv.frag_tri = r->frag_tri.ptr; fv.frag_px = r->frag_px.ptr;
    fv.frag_tri_capacity = (uint32_t)r->frag_tri.count; fv.frag_px_capacity = (uint32_t)r->frag_px.count;
    fv.skybox = (r->cfg.skybox && r->sv.skybox.mip_count) ? 1u : 0u;
    fv.occlusion = r->occlusion ? 1u : 0u;
    fv.draw_visible = r->draw_visible.ptr;
    fv.meshlet_bits = r->meshlet_bits.ptr;
    fv.pyramid = r->pyramid.ptr;
    {
        uint32_t off = 0, l = 0;
        for (;; ++l) {
            const uint32_t step = 2u << l; // pixels per texel side at level l
            fv.pyr_w[l] = (r->width + step - 1u) / step;
            fv.pyr_h[l] = (r->height + step - 1u) / step;
            fv.pyr_offset[l] = off;
            off += fv.pyr_w[l] * fv.pyr_h[l];
            if ((fv.pyr_w[l] == 1u && fv.pyr_h[l] == 1u) || l == 15u) break;
        }
        fv.pyr_levels = l + 1u;
    }
    fv.lane_max_pixels = r->lane_max_pixels;
    fv.mid_max_pixels = r->mid_max_pixels;
    fv.bin_capacity = (uint32_t)r->bin_entries.count;
    fv.tiles_x = (r->width + 63u) / 64u;
    fv.tiles_y = (r->height + 63u) / 64u;
    fv.stat_slots = r->stat_slots.ptr;
    GridPlan& g = fv.grid;
    g.draw_blocks = std::max(1u, (r->sv.draw_count + 255u) / 256u);
    g.cull_blocks = std::max(1u, std::min((fv.command_capacity + 15u) / 16u, std::max(1u, r->resident.meshlet_cull)));
    g.raster_blocks = std::max(1u, std::min(fv.visible_capacity, std::max(1u, r->resident.raster)));
    g.raster_mask_blocks = std::max(1u, std::min(r->masked_meshlets, std::max(1u, r->resident.raster_mask)));
    // the tile rasteriser's workgroups take jobs from the list a grid's width apart: one resident round, or four when
    // the last finished frame had triangles for it, so that the hardware hands the later workgroups to whichever
    // compute unit is done first (jobs differ a lot in what they cover; with one round the workgroups' ends were
    // spread over half the kernel's length: interior scene 89 -> 79 us; an empty launch of four rounds costs 2 us more)
    const uint32_t work_hint = r->status_host ? r->status_host[2] : 0u;
    const uint32_t deferred_blocks_max = std::max(1u, std::min(fv.tiles_x * fv.tiles_y * 16u, 4u * r->resident.tile_raster));
    g.deferred_blocks = (work_hint & 2u) ? deferred_blocks_max : std::max(1u, std::min(fv.tiles_x * fv.tiles_y * 16u, r->resident.tile_raster));
    g.mid_blocks = std::max(1u, r->resident.mid_raster);
    g.frag_blocks = r->masked_meshlets ? std::max(1u, r->resident.alpha_frag) : 1u;
    g.shade_blocks_x = (r->width + 15u) / 16u;
    g.shade_blocks_y = (fv.shade_rows + 15u) / 16u;
    g.slot_draw = 0;
    g.slot_cull = g.slot_draw + g.draw_blocks;
    g.slot_raster = g.slot_cull + g.cull_blocks * kCullSlots;
    // the one-launch tile path: a fraction of what fits the device (its barriers need the whole grid resident, also
    // beside another renderer's), a multiple of the queue segments; on when the last finished frame had no such work
    g.rare_blocks = r->tile_path_rare_mode ? (std::min(128u, r->resident.tile_path_rare) / 32u) * 32u : 0u;
    {
        const uint32_t hint = work_hint;
        g.tile_path_rare = r->tile_path_rare_mode == 2u ? 1u : ((hint != 0u && (hint & 0xFFu) == 0u) ? 1u : 0u);
    }
    g.slot_mid = g.slot_raster + (g.raster_blocks + g.raster_mask_blocks) * kRasterSlots;
    g.slot_frag = g.slot_mid + std::max(g.mid_blocks, g.rare_blocks);
    g.slot_deferred = g.slot_frag + g.frag_blocks;
    g.slot_shade = g.slot_deferred + std::max(deferred_blocks_max, g.rare_blocks); // (the same from frame to frame: the slots are allocated once)
    g.slot_total = g.slot_shade;
    return fv;
}

int fetch_counters(rdr_renderer* r, FrameCounters* out) {
    RDR_HIP(hipStreamSynchronize(r->stream));
    RDR_HIP(hipMemcpy(out, r->counters.ptr + (r->frame_seq & 1u), sizeof(FrameCounters), hipMemcpyDeviceToHost));
    r->seen_status[0] = r->status_host[0]; // this call reports the frame itself
    r->seen_status[1] = r->status_host[1];
    RDR_CHECK(out->deferred_overflow == 0 && out->setup_overflow == 0 && out->bin_overflow == 0 && out->mid_overflow == 0 && out->frag_overflow == 0 &&
                  (out->sticky_flags & (RDR_FRAME_DEFERRED_OVERFLOW | RDR_FRAME_SETUP_OVERFLOW | RDR_FRAME_BIN_OVERFLOW | RDR_FRAME_MID_OVERFLOW | RDR_FRAME_FRAG_OVERFLOW)) == 0,
              "frame: a tile-path list (deferred triangles, setup records or bins) overflowed; triangles were dropped");
    RDR_CHECK(out->rare_timeout == 0, "frame: the one-launch tile path gave up waiting for its own workgroups (the device was kept full by other streams); triangles were dropped");
    RDR_CHECK((out->sticky_flags & RDR_FRAME_LIST_OVERFLOW) == 0, "frame: a work list overflowed");
    RDR_CHECK(out->deep_overflow == 0, "frame: a shadow ray's traversal stack overflowed; its pixel may be lit wrongly");
    RDR_CHECK(out->cmd_count <= r->commands.count && (uint64_t)out->vis_count + out->vis_mask_count <= r->visible.count,
              "frame: a work list overflowed");
    return 0;
}

RowSet shaded_rows(const rdr_renderer* r) {
    return RowSet{std::min(r->row_lo, r->height), std::min(r->row_hi, r->height), r->band_shift, r->band_rank, r->band_world};
}

std::string status_message(uint32_t flags) {
    std::string m;
    auto add = [&](uint32_t bit, const char* what) {
        if (flags & bit) m += (m.empty() ? "" : ", ") + std::string(what);
    };
    add(RDR_FRAME_DEFERRED_OVERFLOW, "deferred triangle list");
    add(RDR_FRAME_SETUP_OVERFLOW, "tile set-up records");
    add(RDR_FRAME_BIN_OVERFLOW, "tile bins");
    add(RDR_FRAME_MID_OVERFLOW, "mid-sized triangle queue");
    add(RDR_FRAME_FRAG_OVERFLOW, "alpha-tested fragment queue");
    add(RDR_FRAME_LIST_OVERFLOW, "cull command / visible meshlet list");
    add(RDR_FRAME_SHADOW_STACK, "shadow-ray traversal stack");
    return m;
}

// Looks at the status words the kernels of earlier frames left in host memory (no synchronisation: a frame
// still in flight is seen by a later call) and fails once for every finished frame that dropped work.
int check_finished_frames(rdr_renderer* r, const char* who) {
    uint32_t bad = 0;
    for (int h = 0; h < 2; ++h) {
        const uint32_t w = r->status_host[h];
        if (w != r->seen_status[h]) {
            r->seen_status[h] = w;
            bad |= w & 0xFFu;
        }
    }
    if (bad) {
        rdr_composite_mark_broken(r->composite); // this rank stops here while its peers go on: the phases part
        rdr_set_error(std::string(who) + ": an earlier frame dropped work (" + status_message(bad) + " overflowed); its image is incomplete");
        return 1;
    }
    return 0;
}

} // namespace

extern "C" {

int rdr_build_ubo(uint32_t width, uint32_t height, const float cam_pos[3], const float cam_dir[3],
                  const float sun_dir[3], uint32_t draw_count, uint32_t lod_enabled, uint32_t cull_enabled, rdr_ubo* out) {
    RDR_CHECK(out && cam_pos && cam_dir && sun_dir && width && height, "rdr_build_ubo: bad argument");
    build_ubo(width, height, cam_pos, cam_dir, sun_dir, draw_count, lod_enabled, cull_enabled, out);
    return 0;
}

int rdr_create(const rdr_config* cfg, uint32_t width, uint32_t height, const int* hip_devices, int n_devices, rdr_renderer** out) {
    RDR_CHECK(out, "rdr_create: out is null");
    *out = nullptr;
    RDR_CHECK(width >= 1 && height >= 1 && width <= 16384 && height <= 16384, "rdr_create: extent must be within 1..16384");
    RDR_CHECK(n_devices == 1 && hip_devices, "rdr_create: exactly one HIP device per renderer (one process per GPU)");
    int count = 0;
    RDR_HIP(hipGetDeviceCount(&count));
    RDR_CHECK(count > 0, "rdr_create: no HIP device available; this library has no CPU path");
    RDR_CHECK(hip_devices[0] >= 0 && hip_devices[0] < count, "rdr_create: HIP device index out of range");
    for (uint32_t b = 0; b < 256; ++b) { // the texel decode's division-free b / 255 must be exact
        const float c = 0x1.010102p-8f, x = (float)b, q = x * c;
        RDR_CHECK(std::fmaf(std::fmaf(-255.0f, q, x), c, q) == x / 255.0f, "rdr_create: unorm8 self-check failed");
    }
    std::unique_ptr<rdr_renderer> r(new rdr_renderer());
    if (cfg) r->cfg = *cfg;
    r->width = width; r->height = height; r->device = hip_devices[0];
    r->occlusion = r->cfg.cull_enabled != 0;
    if (const char* e = std::getenv("RDR_OCCLUSION")) r->occlusion = r->occlusion && std::atoi(e) != 0; // tuning hook
    if (const char* e = std::getenv("RDR_DEBUG_STACK_LIMIT")) // test hook: force the deep-stack shadow path
        r->lds_stack_limit = (uint32_t)std::min(24, std::max(1, std::atoi(e)));
    // tuning hooks: bounding-box pixel counts at which k_raster hands a triangle to k_mid_raster / the tile path
    if (const char* e = std::getenv("RDR_LANE_MAX_PIXELS")) r->lane_max_pixels = (uint32_t)std::min(4096, std::max(1, std::atoi(e)));
    if (const char* e = std::getenv("RDR_MID_MAX_PIXELS")) r->mid_max_pixels = (uint32_t)std::min(4096, std::max(1, std::atoi(e)));
    RDR_HIP(hipSetDevice(r->device));
    RDR_HIP(hipStreamCreateWithFlags(&r->own_stream, hipStreamNonBlocking));
    r->stream = r->own_stream;
    if (r->cfg.time_kernels) {
        r->ring.resize((size_t)rdr_renderer::kRing * rdr_renderer::kEvents);
        for (auto& e : r->ring) RDR_HIP(hipEventCreate(&e));
    }
    const size_t npx = (size_t)width * height;
    // the sort-last tile flags (FrameView::touched) live behind the keys, in the allocation the peers open
    if (r->vis.alloc(npx + (rdr_touched_words(width, height) + 1) / 2) || r->color.alloc(npx * 4) || r->counters.alloc(2) || r->shadow.alloc((npx + 15) / 16 * 16) ||
        r->pixel_list.alloc(npx) || r->pixel_key.alloc(npx)) return 1;
    // depth pyramid: W/2 x H/2 and its mips, a little over a third of a pixel count
    if (r->pyramid.alloc(npx / 2 + width + height + 64)) return 1; // sum over levels <= W*H/3 + W + H + levels
    {
        const size_t n_tiles = (size_t)((width + 63) / 64) * ((height + 63) / 64);
        if (r->tile_count.alloc(2 * n_tiles) || r->tile_offset.alloc(n_tiles) || r->tile_cursor.alloc(n_tiles)) return 1;
        RDR_HIP(hipMemset(r->tile_count.ptr, 0, 2 * n_tiles * 4));
        RDR_HIP(hipMemset(r->tile_cursor.ptr, 0, n_tiles * 4)); // k_tile_fill expects it zero; k_tile_raster leaves it so
        // k_tile_fill keeps one bin offset per tile in LDS (64 KiB: 15 872 tiles, e.g. 11 264 x 5 760 pixels)
        RDR_CHECK(rdr_tile_fill_lds((uint32_t)n_tiles) <= 65536, "rdr_create: frame too large for the tile path (more than 15 872 tiles of 64 x 64 pixels)");
        if (rdr_geometry_residency(&r->resident, rdr_tile_fill_lds((uint32_t)n_tiles))) return 1;
    }
    RDR_HIP(hipMemset(r->counters.ptr, 0, 2 * sizeof(FrameCounters)));
    {
        void* host = nullptr;
        void* dev = nullptr;
        RDR_HIP(hipHostMalloc(&host, 64, hipHostMallocMapped));
        std::memset(host, 0, 64);
        r->status_host = static_cast<volatile uint32_t*>(host);
        RDR_HIP(hipHostGetDevicePointer(&dev, host, 0));
        r->status_dev = static_cast<uint32_t*>(dev);
    }
    if (const char* e = std::getenv("RDR_TILE_FLAGS")) r->force_tile_flags = std::atoi(e) != 0;
    if (const char* e = std::getenv("RDR_TILE_PATH_RARE")) r->tile_path_rare_mode = (uint32_t)std::min(2, std::max(0, std::atoi(e)));
    if (const char* e = std::getenv("RDR_CONVEX_SKIP")) r->convex_skip = std::atoi(e) != 0 ? 1u : 0u;
    if (const char* e = std::getenv("RDR_SUN_CANDIDATES")) r->sun_candidates = std::atoi(e) != 0 ? 1u : 0u; // measurement hook
    if (const char* e = std::getenv("RDR_SHADOW_DEFER")) r->shadow_defer = std::atoi(e) != 0 ? 1u : 0u; // measurement hook
    if (const char* e = std::getenv("RDR_DEBUG_DEFERRED_CAPACITY")) // test hook: force the deferred list to overflow
        r->debug_deferred_capacity = (uint32_t)std::max(1, std::atoi(e));
    float lut[256];
    for (int i = 0; i < 256; ++i) {
        const float c = (float)i / 255.0f;
        lut[i] = c <= 0.04045f ? c / 12.92f : std::pow((c + 0.055f) / 1.055f, 2.4f);
    }
    if (r->srgb_lut.upload(lut, 256)) return 1;
    *out = r.release();
    return 0;
}

void rdr_destroy(rdr_renderer* r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    (void)hipStreamSynchronize(r->stream);
    for (auto& e : r->ring) (void)hipEventDestroy(e);
    if (r->own_stream) (void)hipStreamDestroy(r->own_stream);
    if (r->status_host) (void)hipHostFree(const_cast<uint32_t*>(r->status_host));
    rdr_composite_destroy(r->composite);
    delete r;
}

int rdr_set_stream(rdr_renderer* r, void* hip_stream) {
    RDR_CHECK(r, "rdr_set_stream: renderer is null");
    RDR_HIP(hipStreamSynchronize(r->stream));
    // null selects the library's own stream; hipStreamLegacy / hipStreamPerThread (HIP's special handles) select
    // the null stream and the per-thread stream and are passed through as they are
    r->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : r->own_stream;
    return 0;
}

int rdr_sync(rdr_renderer* r) {
    RDR_CHECK(r, "rdr_sync: renderer is null");
    RDR_HIP(hipStreamSynchronize(r->stream));
    return 0;
}

int rdr_set_scene(rdr_renderer* r, const rdr_scene_desc* scene) {
    RDR_CHECK(r && scene, "rdr_set_scene: null argument");
    RDR_HIP(hipSetDevice(r->device));
    RDR_HIP(hipStreamSynchronize(r->stream));
    const rdr_scene_desc& s = *scene;
    if (validate_scene(s)) return 1;
    r->has_scene = false;

    if (r->vertices.upload(s.vertices, s.vertex_count)) return 1;
    {
        // positions as three planes so the rasteriser's per-vertex gathers touch 12 useful bytes
        std::vector<float> px(s.vertex_count), py(s.vertex_count), pz(s.vertex_count);
        for (uint32_t i = 0; i < s.vertex_count; ++i) {
            px[i] = s.vertices[i].p[0]; py[i] = s.vertices[i].p[1]; pz[i] = s.vertices[i].p[2];
        }
        if (r->pos_x.upload(px.data(), px.size()) || r->pos_y.upload(py.data(), py.size()) || r->pos_z.upload(pz.data(), pz.size())) return 1;
    }
    { // both uv sets as one 16-byte element: the alpha-tested rasteriser reads uv per vertex and nothing else of it
        std::vector<float4> uv(s.vertex_count);
        for (uint32_t i = 0; i < s.vertex_count; ++i)
            uv[i] = make_float4(s.vertices[i].uv[0], s.vertices[i].uv[1], s.vertices[i].uv[2], s.vertices[i].uv[3]);
        if (r->uv_sets.upload(uv.data(), uv.size())) return 1;
    }
    if (r->meshlet_vertices.upload(s.meshlet_vertices, s.meshlet_vertex_count)) return 1;
    {
        // a few bytes of slack: triangle indices are fetched with byte loads only, but keep the tail mapped
        std::vector<uint8_t> tri(s.meshlet_triangle_bytes + 16, 0);
        if (s.meshlet_triangle_bytes) std::memcpy(tri.data(), s.meshlet_triangles, s.meshlet_triangle_bytes);
        if (r->meshlet_triangles.upload(tri.data(), tri.size())) return 1;
    }
    {
        // positions and uv sets once more, in the order of meshlet_vertices (see SceneView::mpos_x)
        const size_t n = (size_t)s.meshlet_vertex_count + 64;
        std::vector<float> mx(n, 0.0f), my(n, 0.0f), mz(n, 0.0f);
        std::vector<float4> muv(n, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        std::vector<uint32_t> owner(n, UINT32_MAX);
        const char* indexed = std::getenv("RDR_RASTER_INDEXED"); // 1: leave the arrays out (the path through the indices, for tests)
        bool ok = !(indexed && indexed[0] == '1');
        for (uint32_t me = 0; me < s.mesh_count && ok; ++me)
            for (uint32_t l = 0; l < s.meshes[me].lod_count && ok; ++l) {
                const rdr_mesh_lod& L = s.meshes[me].lods[l];
                for (uint32_t mi = L.meshlet_offset; mi < L.meshlet_offset + L.meshlet_count && ok; ++mi) {
                    const rdr_meshlet& ml = s.meshlets[mi];
                    for (uint32_t v = 0; v < ml.vertex_count; ++v) {
                        const uint32_t slot = ml.vertex_offset + v, vi = s.meshes[me].vertex_offset + s.meshlet_vertices[slot];
                        if (owner[slot] != UINT32_MAX && owner[slot] != vi) { ok = false; break; }
                        owner[slot] = vi;
                        const rdr_vertex& vx = s.vertices[vi];
                        mx[slot] = vx.p[0]; my[slot] = vx.p[1]; mz[slot] = vx.p[2];
                        muv[slot] = make_float4(vx.uv[0], vx.uv[1], vx.uv[2], vx.uv[3]);
                    }
                }
            }
        r->meshlet_order = ok;
        if (ok) {
            if (r->mpos_x.upload(mx.data(), n) || r->mpos_y.upload(my.data(), n) || r->mpos_z.upload(mz.data(), n) || r->muv.upload(muv.data(), n)) return 1;
        }
    }
    if (r->meshlets.upload(s.meshlets, s.meshlet_count)) return 1;
    if (r->meshes.upload(s.meshes, s.mesh_count)) return 1;
    if (r->draws.upload(s.mesh_draws, s.draw_count)) return 1;
    {
        // one all-zero material behind the scene's: an index past the end reads zeros (robust buffer access) through
        // the same pointer arithmetic as a valid one
        std::vector<rdr_material> mats((size_t)s.material_count + 1);
        std::memset(static_cast<void*>(mats.data()), 0, mats.size() * sizeof(rdr_material));
        if (s.material_count) std::memcpy(static_cast<void*>(mats.data()), s.materials, (size_t)s.material_count * sizeof(rdr_material));
        if (r->materials.upload(mats.data(), mats.size())) return 1;
    }
    if (r->transforms.upload(s.global_transforms, (size_t)s.transform_count * 16)) return 1;
    {
        // mean area of a level-0 triangle per mesh, in object space (what rdr_set_partition_replication's estimate starts from)
        std::vector<float> area(std::max<uint32_t>(s.mesh_count, 1u), 0.0f);
        for (uint32_t mi = 0; mi < s.mesh_count; ++mi) {
            const rdr_mesh& m = s.meshes[mi];
            const uint32_t end = mi + 1 < s.mesh_count ? s.meshes[mi + 1].index_offset : s.index_count;
            double sum = 0.0;
            uint32_t n = 0;
            for (uint32_t i = m.index_offset; i + 2 < end; i += 3, ++n) {
                const float* a = s.vertices[m.vertex_offset + s.indices[i]].p;
                const float* b = s.vertices[m.vertex_offset + s.indices[i + 1]].p;
                const float* c = s.vertices[m.vertex_offset + s.indices[i + 2]].p;
                const double e1[3] = {(double)b[0] - a[0], (double)b[1] - a[1], (double)b[2] - a[2]}, e2[3] = {(double)c[0] - a[0], (double)c[1] - a[1], (double)c[2] - a[2]};
                const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
                sum += 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
            }
            area[mi] = n ? (float)(sum / n) : 0.0f;
        }
        if (r->mesh_tri_area.upload(area.data(), area.size())) return 1;
    }
    if (r->draw_lod.alloc(s.draw_count) || r->draw_frame.alloc(std::max<size_t>(s.draw_count, 1)) || r->draw_visible.alloc(std::max<size_t>(s.draw_count, 1)) ||
        r->meshlet_bits.alloc(((size_t)s.meshlet_visibility_count + 31) / 32 + 1)) return 1;
    // nothing was visible "last frame": the first frame's early phase draws nothing, its late phase everything
    RDR_HIP(hipMemset(r->draw_visible.ptr, 0, r->draw_visible.count));
    RDR_HIP(hipMemset(r->meshlet_bits.ptr, 0, r->meshlet_bits.count * sizeof(uint32_t)));
    RDR_HIP(hipMemset(r->draw_lod.ptr, 0xFF, std::max<size_t>(s.draw_count, 1)));
    RDR_HIP(hipMemset(r->draw_frame.ptr, 0, std::max<size_t>(s.draw_count, 1) * sizeof(DrawFrame)));
    {
        // (meshlet, triangle) -> the triangle's three vertices relative to its mesh: what the shading passes would
        // otherwise find through the meshlet record, three index bytes and three meshlet_vertices entries
        std::vector<TriVertices> table((size_t)std::max<uint32_t>(s.meshlet_count, 1) * kTriTableStride, TriVertices{{0u, 0u, 0u}});
        for (uint32_t me = 0; me < s.mesh_count; ++me) // the meshlets the meshes use (the ones validate_scene has checked)
            for (uint32_t l = 0; l < s.meshes[me].lod_count; ++l) {
                const rdr_mesh_lod& L = s.meshes[me].lods[l];
                for (uint32_t mi = L.meshlet_offset; mi < L.meshlet_offset + L.meshlet_count; ++mi) {
                    const rdr_meshlet& ml = s.meshlets[mi];
                    const uint32_t nt = std::min<uint32_t>(ml.triangle_count, kTriTableStride);
                    for (uint32_t t = 0; t < nt; ++t)
                        for (int k = 0; k < 3; ++k)
                            table[(size_t)mi * kTriTableStride + t].v[k] = s.meshlet_vertices[ml.vertex_offset + s.meshlet_triangles[ml.triangle_offset + 3 * t + k]];
                }
            }
        if (r->tri_table.upload(table.data(), table.size())) return 1;
    }

    // meshlet instance -> draw, and work-list capacities
    uint64_t cmd_cap = 0, vis_cap = 0, tri_total = 0, masked_total = 0;
    {
        std::vector<uint32_t> inst(std::max<uint32_t>(s.meshlet_visibility_count, 1), 0u);
        for (uint32_t di = 0; di < s.draw_count; ++di) {
            const rdr_mesh_draw& d = s.mesh_draws[di];
            const rdr_mesh& m = s.meshes[d.mesh_index];
            const uint32_t n0 = m.lods[0].meshlet_count;
            for (uint32_t k = 0; k < n0; ++k) inst[d.meshlet_visibility_offset + k] = di;
            if (d.alpha_mode == 1u) masked_total += n0;
            cmd_cap += (n0 + 63) / 64;
            vis_cap += n0;
            tri_total += (uint64_t)n0 * 124;
        }
        if (r->instance_draw.upload(inst.data(), inst.size())) return 1;
        r->masked_meshlets = (uint32_t)masked_total;
    }
    if (r->commands.alloc(cmd_cap) || r->visible.alloc(vis_cap)) return 1;
    // The mid and alpha-fragment queues are cut into 32 segments and a workgroup of the producing kernel always
    // uses segment blockIdx % 32: a segment must hold everything the workgroups mapped to it can queue, i.e.
    // (meshlets per workgroup) x (workgroups per segment) x 124 triangles.
    auto segment_worst = [](uint64_t meshlets, uint32_t resident) -> uint64_t {
        const uint64_t grid = std::max<uint64_t>(1, std::min<uint64_t>(meshlets, std::max(1u, resident)));
        return ((meshlets + grid - 1) / grid) * ((grid + 31) / 32) * 124;
    };
    {
        // tile path: deferred clip-space triangles, their screen-space records (clipping can fan one
        // triangle into several) and the per-tile bins
        // at most every LOD-0 triangle, capped at 4 M per frame (about 1.6 GB of lists at the cap; a frame that
        // needs more reports an overflow error instead of dropping triangles silently)
        const size_t n_def = (size_t)std::min<uint64_t>(std::max<uint64_t>(tri_total, 1u << 16), 1u << 22);
        // mid queue: k_raster's opaque mid-sized triangles plus the alpha-tested ones k_tile_setup passes on from
        // the deferred list (its 1024 workgroups take deferred triangles a grid apart: n_def / 32 + 32 per segment)
        const size_t n_mid = (size_t)std::min<uint64_t>(32 * (segment_worst(vis_cap, r->resident.raster) + n_def / 32 + 32), (uint64_t)2 << 22);
        if (r->deferred.alloc(n_def) || r->setup.alloc(n_def + n_def / 2) || r->setup_mask.alloc(n_def + n_def / 2) ||
            r->mid.alloc(n_mid) || r->mid_mask.alloc(n_mid)) return 1;
        if (r->bin_entries.alloc(std::max<size_t>(n_def * 4, (size_t)1 << 22))) return 1;
        // jobs of the tile rasteriser: at most one partly filled job per tile plus the full ones
        if (r->tile_list.alloc(((size_t)r->tile_count.count / 2 + r->bin_entries.count / kTileJob + 1) * 4)) return 1; // x parts per tile
        // small alpha-tested triangles and their covered pixels (at most 16 each): every LOD-0 triangle of the
        // alpha-tested draws, capped like the lists above
        const size_t n_frag = (size_t)std::min<uint64_t>(std::max<uint64_t>(32 * segment_worst(r->masked_meshlets, r->resident.raster_mask), 1u << 12), 1u << 22);
        if (r->frag_tri.alloc(n_frag) || r->frag_px.alloc(n_frag * 16)) return 1;
        // zeroed once: a consumer that meets a never-written entry (only possible after an overflow, which is
        // reported as an error) then reads a valid index
        RDR_HIP(hipMemset(r->frag_tri.ptr, 0, n_frag * sizeof(FragTri)));
        RDR_HIP(hipMemset(r->frag_px.ptr, 0, n_frag * 16 * sizeof(uint32_t)));
    }
    {
        // statistics slots for the largest launch geometry this scene can produce
        r->sv.draw_count = s.draw_count;
        const uint32_t keep_lo = r->row_lo, keep_hi = r->row_hi, keep_world = r->band_world;
        r->row_lo = 0; r->row_hi = r->height; r->band_world = 1;
        const FrameView fv = make_frame_view(r);
        r->row_lo = keep_lo; r->row_hi = keep_hi; r->band_world = keep_world;
        if (r->stat_slots.alloc(fv.grid.slot_total)) return 1;
        RDR_HIP(hipMemset(r->stat_slots.ptr, 0, (size_t)fv.grid.slot_total * 4));
    }

    // textures and environment
    {
        TexelPacker tp;
        std::vector<GpuImage> imgs(s.image_count);
        for (uint32_t i = 0; i < s.image_count; ++i) imgs[i] = pack_image(tp, s.images[i]);
        static const float zero_lut[4] = {0, 0, 0, 0};
        rdr_image_desc lut_desc{zero_lut, 1, 1, 1, RDR_FORMAT_RGBA32F};
        if (s.env && s.env->brdf_lut.data) {
            lut_desc = s.env->brdf_lut;
            RDR_CHECK(lut_desc.width && lut_desc.height && lut_desc.width <= 32768 && lut_desc.height <= 32768 && lut_desc.mip_count >= 1 && lut_desc.mip_count <= 17 && lut_desc.format <= RDR_FORMAT_RGBA32F,
                      "scene: brdf_lut invalid");
        }
        r->sv.brdf_lut = pack_image(tp, lut_desc);
        if (pack_cubemap(tp, s.env ? &s.env->lambertian : nullptr, &r->sv.lambertian)) return 1;
        if (pack_cubemap(tp, s.env ? &s.env->ggx : nullptr, &r->sv.ggx)) return 1;
        if (pack_cubemap(tp, s.env ? &s.env->charlie : nullptr, &r->sv.charlie)) return 1;
        if (pack_cubemap(tp, s.env ? &s.env->skybox : nullptr, &r->sv.skybox)) return 1;
        if (r->images.upload(imgs.data(), imgs.size())) return 1;
        {   // make_alpha_test (rdr_geometry.hip) for every draw, term for term
            std::vector<AlphaTest> tests(std::max<uint32_t>(s.draw_count, 1u));
            for (uint32_t di = 0; di < s.draw_count; ++di) {
                const rdr_mesh_draw& d = s.mesh_draws[di];
                AlphaTest at{};
                at.enabled = d.alpha_mode == 1u;
                if (at.enabled && d.material_index < s.material_count) {
                    const rdr_material& mat = s.materials[d.material_index];
                    at.factor_a = mat.base_color_factor[3];
                    at.cutoff = mat.emissive_factor_alpha_cutoff[3];
                    const uint32_t tex = mat.albedo_texture;
                    if (tex != RDR_NO_TEXTURE && tex < s.image_count && imgs[tex].mip_count == 0u) {
                        at.factor_a = 0.0f;
                    } else if (tex != RDR_NO_TEXTURE && tex < s.image_count) {
                        at.has_tex = true;
                        at.image = tex;
                        at.uv_set = mat.albedo_texture_uv == 0u ? 0u : 1u;
                        at.sx = mat.albedo_texture_transform[0]; at.sy = mat.albedo_texture_transform[1];
                        at.ox = mat.albedo_texture_transform[2]; at.oy = mat.albedo_texture_transform[3];
                    }
                }
                tests[di] = at;
            }
            if (r->draw_alpha.upload(tests.data(), tests.size())) return 1;
        }
        tp.blob.resize(tp.blob.size() + 16); // slack: a bilinear tap at an image's last texel reads eight bytes from it
        if (r->texels.upload(tp.blob.data(), tp.blob.size())) return 1;
        if (decode_block_images(tp, r->texels.ptr)) return 1;
    }

    // shadow-ray structure
    rdr_build_blas(s, r->bvh);
    rdr_build_tlas(s, s.global_transforms, r->bvh);
    if (r->bvh_dev.alloc(r->bvh.blas_pairs.size() + (size_t)s.draw_count + 2)) return 1;
    if (!r->bvh.blas_pairs.empty())
        RDR_HIP(hipMemcpy(r->bvh_dev.ptr, r->bvh.blas_pairs.data(), r->bvh.blas_pairs.size() * sizeof(BvhWide), hipMemcpyHostToDevice));
    if (r->blas_tris.upload(r->bvh.blas_tris.data(), r->bvh.blas_tris.size())) return 1;
    r->h_draws.assign(s.mesh_draws, s.mesh_draws + s.draw_count);
    r->h_meshes.assign(s.meshes, s.meshes + s.mesh_count);

    SceneView& sv = r->sv;
    sv.vertices = r->vertices.ptr;
    sv.pos_x = r->pos_x.ptr; sv.pos_y = r->pos_y.ptr; sv.pos_z = r->pos_z.ptr;
    sv.uv_sets = r->uv_sets.ptr;
    sv.mpos_x = r->meshlet_order ? r->mpos_x.ptr : nullptr; sv.mpos_y = r->meshlet_order ? r->mpos_y.ptr : nullptr;
    sv.mpos_z = r->meshlet_order ? r->mpos_z.ptr : nullptr; sv.muv = r->meshlet_order ? r->muv.ptr : nullptr;
    sv.meshlet_vertices = r->meshlet_vertices.ptr;
    sv.meshlet_triangles = r->meshlet_triangles.ptr;
    sv.meshlets = r->meshlets.ptr;
    sv.meshes = r->meshes.ptr;
    sv.draws = r->draws.ptr;
    sv.materials = r->materials.ptr;
    sv.transforms = r->transforms.ptr;
    sv.instance_draw = r->instance_draw.ptr;
    sv.draw_alpha = r->draw_alpha.ptr;
    sv.mesh_tri_area = r->mesh_tri_area.ptr;
    sv.draw_lod = r->draw_lod.ptr;
    sv.draw_frame = r->draw_frame.ptr;
    sv.tri_table = r->tri_table.ptr;
    sv.images = r->images.ptr;
    sv.texels = r->texels.ptr;
    sv.srgb_lut = r->srgb_lut.ptr;
    sv.draw_count = s.draw_count;
    sv.material_count = s.material_count;
    sv.image_count = s.image_count;
    sv.meshlet_count = s.meshlet_count;
    sv.masked_meshlets = (uint32_t)std::min<uint64_t>(masked_total, 0xFFFFFFFFull);
    sv.blas_tris = r->blas_tris.ptr;
    if (upload_tlas(r)) return 1;
    r->meshlet_visibility_count = s.meshlet_visibility_count;
    RDR_HIP(hipDeviceSynchronize());
    r->has_scene = true;
    r->frame_rendered = false;
    return 0;
}

int rdr_update_transforms(rdr_renderer* r, const float* mat4s, uint32_t count) {
    RDR_CHECK(r && mat4s, "rdr_update_transforms: null argument");
    RDR_CHECK(r->has_scene, "rdr_update_transforms: no scene set");
    RDR_CHECK((size_t)count * 16 == r->transforms.count, "rdr_update_transforms: the whole transform array must be passed");
    RDR_HIP(hipSetDevice(r->device));
    RDR_HIP(hipStreamSynchronize(r->stream));
    RDR_HIP(hipMemcpy(r->transforms.ptr, mat4s, (size_t)count * 64, hipMemcpyHostToDevice));
    // the shadow ray sees the moved instances: rebuild the top level
    rdr_scene_desc s{};
    s.mesh_draws = r->h_draws.data();
    s.draw_count = (uint32_t)r->h_draws.size();
    s.meshes = r->h_meshes.data();
    s.mesh_count = (uint32_t)r->h_meshes.size();
    rdr_build_tlas(s, mat4s, r->bvh);
    if (upload_tlas(r)) return 1;
    RDR_HIP(hipDeviceSynchronize()); // the reference waits for device idle here (renderer.rs:718)
    return 0;
}

int rdr_set_partition(rdr_renderer* r, uint32_t instance_lo, uint32_t instance_hi) {
    RDR_CHECK(r, "rdr_set_partition: renderer is null");
    RDR_CHECK(instance_lo <= instance_hi, "rdr_set_partition: lo > hi");
    r->part_lo = instance_lo; r->part_hi = instance_hi;
    return 0;
}

int rdr_set_shade_rows(rdr_renderer* r, uint32_t row_lo, uint32_t row_hi) {
    RDR_CHECK(r, "rdr_set_shade_rows: renderer is null");
    RDR_CHECK(row_lo <= row_hi, "rdr_set_shade_rows: lo > hi");
    r->row_lo = row_lo; r->row_hi = row_hi;
    r->band_world = 1; r->band_rank = 0;
    return 0;
}

int rdr_set_partition_interleaved(rdr_renderer* r, uint32_t chunk, uint32_t rank, uint32_t world) {
    RDR_CHECK(r, "rdr_set_partition_interleaved: renderer is null");
    RDR_CHECK(world >= 1 && rank < world, "rdr_set_partition_interleaved: rank must be below world");
    RDR_CHECK(chunk >= 64 && chunk % 64 == 0 && chunk <= (1u << 20), "rdr_set_partition_interleaved: chunk must be a multiple of 64, at most 2^20");
    r->part_chunk = chunk; r->part_rank = rank; r->part_world = world;
    return 0;
}

int rdr_set_partition_replication(rdr_renderer* r, float min_triangle_pixels) {
    RDR_CHECK(r, "rdr_set_partition_replication: renderer is null");
    RDR_CHECK(min_triangle_pixels >= 0.0f, "rdr_set_partition_replication: the threshold must not be negative (0 switches it off)");
    r->replicate_pixels = min_triangle_pixels;
    return 0;
}

int rdr_set_shade_bands(rdr_renderer* r, uint32_t band, uint32_t rank, uint32_t world) {
    RDR_CHECK(r, "rdr_set_shade_bands: renderer is null");
    RDR_CHECK(world >= 1 && rank < world, "rdr_set_shade_bands: rank must be below world");
    RDR_CHECK(band >= 32 && band <= 16384 && (band & (band - 1)) == 0, "rdr_set_shade_bands: band must be a power of two between 32 and 16384");
    uint32_t shift = 5;
    while ((1u << shift) < band) ++shift;
    r->band_shift = shift; r->band_rank = rank; r->band_world = world;
    r->row_lo = 0; r->row_hi = 0xFFFFFFFFu;
    return 0;
}

int rdr_render_visbuffer(rdr_renderer* r, const float cam_pos[3], const float cam_dir[3], const float sun_dir[3]) {
    RDR_CHECK(r && cam_pos && cam_dir && sun_dir, "rdr_render_visbuffer: null argument");
    RDR_CHECK(r->has_scene, "rdr_render_visbuffer: no scene set");
    RDR_HIP(hipSetDevice(r->device));
    if (check_finished_frames(r, "rdr_render_visbuffer")) return 1;
    r->frame_seq++;
    build_ubo(r->width, r->height, cam_pos, cam_dir, sun_dir, r->sv.draw_count, r->cfg.lod_enabled, r->cfg.cull_enabled, &r->ubo);
    const FrameView fv = make_frame_view(r);
    hipEvent_t* ev = nullptr;
    // time_kernels = n: every n-th frame is bracketed (the brackets themselves cost about 3 us each, 33 us per frame)
    r->timing_this_frame = r->cfg.time_kernels && (r->frame_seq % r->cfg.time_kernels) == 0;
    if (r->timing_this_frame) {
        r->cur_slot = (int)(r->frames_started % rdr_renderer::kRing);
        r->frames_started++;
        r->slot_geometry[r->cur_slot] = true;
        r->slot_shade[r->cur_slot] = false;
        ev = r->ring.data() + (size_t)r->cur_slot * rdr_renderer::kEvents;
    }
    if (rdr_launch_geometry(r->sv, fv, r->stream, ev)) return 1;
    r->last_grid = fv.grid;
    r->frame_rendered = true;
    r->frame_has_tile_flags = fv.touched != nullptr;
    r->shaded = false;
    return 0;
}

int rdr_shade(rdr_renderer* r) {
    RDR_CHECK(r, "rdr_shade: renderer is null");
    RDR_CHECK(r->has_scene && r->frame_rendered, "rdr_shade: no visibility buffer rendered yet");
    RDR_HIP(hipSetDevice(r->device));
    if (check_finished_frames(r, "rdr_shade")) return 1;
    FrameView fv = make_frame_view(r);
    fv.list_heads_zero = r->shaded ? 0u : 1u; // first shading pass over this visibility buffer
    hipEvent_t* ev = r->timing_this_frame ? r->ring.data() + (size_t)r->cur_slot * rdr_renderer::kEvents : nullptr;
    // which instances could block a sun ray that starts on a given instance: a function of the sun direction and the
    // transforms, brought up to date here when either has changed since it was computed (a static scene pays once)
    if (!r->sun_cand_valid || std::memcmp(r->sun_cand_dir, fv.ubo.sun_direction, sizeof(r->sun_cand_dir)) != 0) {
        if (rdr_launch_sun_candidates(r->sv, fv, r->stream)) return 1;
        std::memcpy(r->sun_cand_dir, fv.ubo.sun_direction, sizeof(r->sun_cand_dir));
        r->sun_cand_valid = true;
    }
    if (rdr_launch_shade(r->sv, fv, r->stream, ev)) return 1;
    r->shaded = true;
    if (ev) r->slot_shade[r->cur_slot] = true;
    return 0;
}

int rdr_render(rdr_renderer* r, const float cam_pos[3], const float cam_dir[3], const float sun_dir[3]) {
    if (rdr_render_visbuffer(r, cam_pos, cam_dir, sun_dir)) return 1;
    return rdr_shade(r);
}

int rdr_composite_export(rdr_renderer* r, rdr_composite_handle* out) {
    RDR_CHECK(r && out, "rdr_composite_export: null argument");
    RDR_HIP(hipSetDevice(r->device));
    if (!r->composite && rdr_composite_create(&r->composite)) return 1;
    return rdr_composite_export_handles(r->composite, r->vis.ptr, r->color.ptr, out);
}

int rdr_composite_connect(rdr_renderer* r, uint32_t rank, uint32_t world, const rdr_composite_handle* handles) {
    RDR_CHECK(r && handles, "rdr_composite_connect: null argument");
    RDR_CHECK(r->composite, "rdr_composite_connect: call rdr_composite_export first (the peers need this rank's handles too)");
    RDR_HIP(hipSetDevice(r->device));
    return rdr_composite_open(r->composite, rank, world, handles, r->vis.ptr, r->color.ptr, r->width, r->height);
}

int rdr_composite_exchange(rdr_renderer* r) {
    RDR_CHECK(r && r->composite, "rdr_composite_exchange: not connected");
    RDR_CHECK(r->frame_rendered, "rdr_composite_exchange: no visibility buffer rendered yet");
    RDR_CHECK(r->frame_has_tile_flags, "rdr_composite_exchange: the visibility buffer was rendered before rdr_composite_connect; render it again");
    RDR_HIP(hipSetDevice(r->device));
    if (rdr_composite_check(r->composite, "rdr_composite_exchange")) return 1;
    return rdr_composite_launch_exchange(r->composite, shaded_rows(r), r->stream);
}

int rdr_composite_gather(rdr_renderer* r) {
    RDR_CHECK(r && r->composite, "rdr_composite_gather: not connected");
    RDR_CHECK(r->shaded, "rdr_composite_gather: nothing shaded yet");
    RDR_HIP(hipSetDevice(r->device));
    if (rdr_composite_check(r->composite, "rdr_composite_gather")) return 1;
    return rdr_composite_launch_gather(r->composite, shaded_rows(r), r->cfg.skybox != 0, r->stream);
}

int rdr_composite_disconnect(rdr_renderer* r) {
    RDR_CHECK(r, "rdr_composite_disconnect: renderer is null");
    if (!r->composite) return 0;
    RDR_HIP(hipSetDevice(r->device));
    RDR_HIP(hipStreamSynchronize(r->stream));
    // a wait that gave up and has not been reported yet is reported here; a connection already known to be broken
    // (the caller has seen that error) closes without another one
    const bool unreported_timeout = rdr_composite_timed_out(r->composite);
    rdr_composite_close(r->composite);
    if (unreported_timeout) {
        rdr_set_error("rdr_composite_disconnect: a peer rank did not reach the last exchange within 5 s (that frame's image is not valid)");
        return 1;
    }
    return 0;
}

int rdr_frame_status(rdr_renderer* r, uint32_t* flags) {
    RDR_CHECK(r && flags, "rdr_frame_status: null argument");
    RDR_HIP(hipSetDevice(r->device));
    RDR_HIP(hipStreamSynchronize(r->stream));
    r->seen_status[0] = r->status_host[0];
    r->seen_status[1] = r->status_host[1];
    // each word carries the frame it describes: the shading half's only counts if it ran for the last frame
    const uint32_t seq = r->frame_seq & 0xFFFFFFu;
    *flags = 0;
    if ((r->seen_status[0] >> 8) == seq) *flags |= r->seen_status[0] & 0x7Fu;
    if ((r->seen_status[1] >> 8) == seq) *flags |= r->seen_status[1] & 0x7Fu;
    if (rdr_composite_timed_out(r->composite)) *flags |= RDR_FRAME_PEER_TIMEOUT;
    return 0;
}

void* rdr_visbuffer_device_ptr(rdr_renderer* r) { return r ? r->vis.ptr : nullptr; }
void* rdr_color_device_ptr(rdr_renderer* r) { return r ? r->color.ptr : nullptr; }

int rdr_read_visbuffer_u64(rdr_renderer* r, uint64_t* out) {
    RDR_CHECK(r && out, "rdr_read_visbuffer_u64: null argument");
    RDR_CHECK(r->frame_rendered, "rdr_read_visbuffer_u64: nothing rendered yet");
    RDR_HIP(hipSetDevice(r->device));
    FrameCounters fc;
    if (fetch_counters(r, &fc)) return 1;
    RDR_HIP(hipMemcpy(out, r->vis.ptr, (size_t)r->width * r->height * 8, hipMemcpyDeviceToHost));
    return 0;
}

int rdr_read_visbuffer(rdr_renderer* r, uint32_t* out_xy) {
    RDR_CHECK(r && out_xy, "rdr_read_visbuffer: null argument");
    RDR_CHECK(r->frame_rendered, "rdr_read_visbuffer: nothing rendered yet");
    RDR_HIP(hipSetDevice(r->device));
    FrameCounters fc;
    if (fetch_counters(r, &fc)) return 1;
    const size_t npx = (size_t)r->width * r->height;
    if (r->resolve_out.count != npx * 2 && r->resolve_out.alloc(npx * 2)) return 1;
    const FrameView fv = make_frame_view(r);
    if (rdr_launch_resolve(r->sv, fv, r->resolve_out.ptr, r->stream)) return 1;
    RDR_HIP(hipStreamSynchronize(r->stream));
    RDR_HIP(hipMemcpy(out_xy, r->resolve_out.ptr, npx * 8, hipMemcpyDeviceToHost));
    return 0;
}

int rdr_read_color(rdr_renderer* r, uint8_t* out) {
    RDR_CHECK(r && out, "rdr_read_color: null argument");
    RDR_CHECK(r->frame_rendered, "rdr_read_color: nothing rendered yet");
    RDR_HIP(hipSetDevice(r->device));
    FrameCounters fc;
    if (fetch_counters(r, &fc)) return 1;
    RDR_HIP(hipMemcpy(out, r->color.ptr, (size_t)r->width * r->height * 4, hipMemcpyDeviceToHost));
    return 0;
}

int rdr_get_stats(rdr_renderer* r, rdr_stats* out) {
    RDR_CHECK(r && out, "rdr_get_stats: null argument");
    RDR_HIP(hipSetDevice(r->device));
    FrameCounters fc;
    if (fetch_counters(r, &fc)) return 1;
    rdr_stats st{};
    st.commands = fc.cmd_count;
    if (r->cfg.collect_stats && r->frame_rendered) {
        // sum the per-workgroup slots of the last frame
        const FrameView fv = make_frame_view(r);
        const GridPlan& g = r->last_grid;
        std::vector<uint32_t> slots(g.slot_total);
        RDR_HIP(hipMemcpy(slots.data(), r->stat_slots.ptr, (size_t)g.slot_total * 4, hipMemcpyDeviceToHost));
        for (uint32_t b = 0; b < g.draw_blocks; ++b) st.draws_visible += slots[g.slot_draw + b];
        for (uint32_t b = 0; b < g.cull_blocks; ++b) {
            const uint32_t* p = &slots[g.slot_cull + b * kCullSlots];
            st.meshlets_tested += p[0]; st.meshlets_visible += p[1]; st.triangles_submitted += p[2];
        }
        for (uint32_t b = 0; b < g.raster_blocks + g.raster_mask_blocks; ++b) {
            const uint32_t* p = &slots[g.slot_raster + b * kRasterSlots];
            st.triangles_tested += p[0]; st.triangles_rasterised += p[1]; st.triangles_big += p[2];
            st.fragments += p[3]; st.vertices_transformed += p[4];
        }
        // the tile path's two regions hold the slots of the form that ran: the one-launch form writes (or clears) one per
        // workgroup of its grid, the three launches one per workgroup of theirs
        const bool rare = g.tile_path_rare && g.rare_blocks;
        for (uint32_t b = 0; b < (rare ? g.rare_blocks : g.mid_blocks); ++b) st.fragments += slots[g.slot_mid + b];
        for (uint32_t b = 0; b < g.frag_blocks; ++b) st.fragments += slots[g.slot_frag + b];
        for (uint32_t b = 0; b < (rare ? g.rare_blocks : g.deferred_blocks); ++b) st.fragments += slots[g.slot_deferred + b];
        if (r->shaded) st.pixels_covered = fc.pixel_count;
        st.bin_entries = fc.bin_total;
        std::vector<uint32_t> bins((size_t)fv.tiles_x * fv.tiles_y);
        RDR_HIP(hipMemcpy(bins.data(), fv.tile_count, bins.size() * 4, hipMemcpyDeviceToHost));
        for (uint32_t n : bins) st.bin_max = std::max<uint64_t>(st.bin_max, n);
    }
    if (r->cfg.time_kernels && r->frames_started) {
        // events 0..5 bracket clear, draw cull, meshlet cull, raster, deferred; 6..9 bracket classification, shadow and shading.
        // Mean over the frames still in the ring (the stream was synchronised above).
        const int map[5] = {RDR_K_CLEAR, RDR_K_DRAW_CULL, RDR_K_MESHLET_CULL, RDR_K_RASTER, RDR_K_BIG_RASTER};
        const int n = (int)std::min<uint64_t>(r->frames_started, rdr_renderer::kRing);
        double sum[RDR_STAT_KERNELS] = {0};
        int n_geo = 0, n_shade = 0;
        for (int f = 0; f < n; ++f) {
            hipEvent_t* ev = r->ring.data() + (size_t)f * rdr_renderer::kEvents;
            if (r->slot_geometry[f]) {
                bool ok = true;
                float ms[5];
                for (int i = 0; i < 5 && ok; ++i) ok = hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]) == hipSuccess;
                if (ok) {
                    for (int i = 0; i < 5; ++i) sum[map[i]] += ms[i];
                    n_geo++;
                }
            }
            if (r->slot_shade[f]) {
                float ms0 = 0.0f, ms1 = 0.0f, ms2 = 0.0f;
                if (hipEventElapsedTime(&ms0, ev[6], ev[7]) == hipSuccess && hipEventElapsedTime(&ms1, ev[7], ev[8]) == hipSuccess &&
                    hipEventElapsedTime(&ms2, ev[8], ev[9]) == hipSuccess) {
                    sum[RDR_K_CLASSIFY] += ms0; sum[RDR_K_SHADOW] += ms1; sum[RDR_K_SHADE] += ms2; n_shade++;
                }
            }
            r->slot_geometry[f] = r->slot_shade[f] = false;
        }
        (void)hipGetLastError();
        for (int i = 0; i < 5; ++i) if (n_geo) st.kernel_ms[map[i]] = (float)(sum[map[i]] / n_geo);
        if (n_shade) {
            st.kernel_ms[RDR_K_SHADE] = (float)(sum[RDR_K_SHADE] / n_shade);
            st.kernel_ms[RDR_K_SHADOW] = (float)(sum[RDR_K_SHADOW] / n_shade);
            st.kernel_ms[RDR_K_CLASSIFY] = (float)(sum[RDR_K_CLASSIFY] / n_shade);
        }
        st.frames_timed = (uint64_t)n_geo;
        r->frames_started = 0;
    }
    *out = st;
    return 0;
}

int rdr_get_ubo(rdr_renderer* r, rdr_ubo* out) {
    RDR_CHECK(r && out, "rdr_get_ubo: null argument");
    *out = r->ubo;
    return 0;
}

} // extern "C"
