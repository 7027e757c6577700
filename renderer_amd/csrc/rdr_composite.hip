// rdr_composite.hip — the sort-last exchange step between the renderers of one node, peer to peer.
//
// The reference is single-GPU (renderer/src/renderer.rs:919-922 only keeps frames in flight); this is
// the one step the multi-GPU path adds between rdr_render_visbuffer and rdr_shade.  Every rank holds a
// full-frame 64-bit visibility buffer of its share of the scene and shades the rows of its own strip,
// so rank r needs, for its strip only, the minimum over all ranks' keys.  On MI355X the eight GPUs of a
// node are fully connected by point-to-point xGMI links: a ring collective moves every byte over one
// link per hop, whereas here each rank reads its strip straight out of the seven peers' buffers, all
// seven links at once, and takes the minimum in registers (k_composite_min); the shaded RGBA8 rows then
// go the other way, written straight into every peer's image (k_push_color).  Buffers are shared
// through HIP IPC handles, one process per GPU.
//
// Ordering between ranks is by monotonic phase numbers in a small control block per renderer: a rank
// tells every peer "I have reached phase n" by storing n into its own slot of the peer's block and
// waits by polling its own block until every slot has reached n (k_signal_wait does both).  Per frame f:
//   2f+1  my keys are complete             (peers may read my visibility buffer)
//   2f+2  my colour tiles are in your image (after k_push_tiles, which comes after this rank read the keys)
// A rank waits for 2f+1 before reading keys and for 2f+2 at the end of the frame; the latter also tells
// it that every peer has finished reading its keys (a peer pushes after it has read), so clearing the
// visibility buffer for the next frame is safe.  Every wait is bounded (about five seconds of wall
// clock): a peer that never arrives turns into an error flag, not a hang.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "rdr_internal.h"

namespace {

constexpr uint32_t kCtrlWords = 64;          // control block: arrived[rank] phase numbers, one word each
constexpr long long kWaitTicks = 500000000;  // wall_clock64 runs at 100 MHz: 5 s

struct PeerTable {
    unsigned long long* vis[RDR_COMPOSITE_MAX_RANKS];
    uint32_t* touched[RDR_COMPOSITE_MAX_RANKS]; // the tile flags behind each visibility buffer
    uint32_t* color[RDR_COMPOSITE_MAX_RANKS];
    uint32_t* ctrl[RDR_COMPOSITE_MAX_RANKS];
};

// Phase `phase` reached: written into this rank's slot of every peer's control block (the data the phase stands
// for was written by earlier kernels of this stream, which have completed); then wait until every peer's slot in
// this rank's block has reached the same phase.  One lane per peer.  The wait gives up after kWaitTicks and records
// that in *timed_out (pinned host memory).
__global__ void k_signal_wait(PeerTable peers, const uint32_t* ctrl, uint32_t rank, uint32_t world, uint32_t phase, uint32_t* timed_out) {
    const uint32_t p = threadIdx.x;
    if (p >= world || p == rank) return;
    __hip_atomic_store(&peers.ctrl[p][rank], phase, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const long long start = wall_clock64();
    for (;;) {
        const uint32_t seen = __hip_atomic_load(&ctrl[p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int32_t)(seen - phase) >= 0) return;
        if (wall_clock64() - start > kWaitTicks) {
            __hip_atomic_store(timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        __builtin_amdgcn_s_sleep(32);
    }
}

// The exchange works on tiles of kTouchTile x kTouchTile pixels.  Every rank's rasterisers leave one word per
// tile behind their visibility buffer (FrameView::touched): non-zero when the tile may hold a key.  A rank that
// renders an eighth of the scene leaves most of the frame clear, and a clear tile cannot lower a minimum, so a
// workgroup first reads the tile's word from every rank (a few bytes over each link) and then fetches the keys of
// the ranks that touched it only.  On the bench field at eight ranks that is about a tenth of the strips.
struct TileGrid {
    uint32_t width, height;   // of the frame
    uint32_t row_lo, row_hi;  // this rank's strip (the whole frame when its rows are bands)
    uint32_t tiles_x;         // tiles per row of the frame
    uint32_t tile_row0;       // strip: first tile row that overlaps it
    uint32_t band_tiles_shift, band_rank, band_world; // bands (band_world > 1): log2 of the tile rows per band, and which bands
};

// The tile row a workgroup row of the grid stands for: the strip's tile rows in order, or the rank's bands' tile rows.
__device__ __forceinline__ uint32_t tile_row(const TileGrid& g) {
    if (g.band_world <= 1u) return g.tile_row0 + blockIdx.y;
    const uint32_t band = blockIdx.y >> g.band_tiles_shift;
    return ((band * g.band_world + g.band_rank) << g.band_tiles_shift) | (blockIdx.y & ((1u << g.band_tiles_shift) - 1u));
}

__device__ __forceinline__ bool tile_rect(const TileGrid& g, uint32_t ty, uint32_t& x0, uint32_t& x1, uint32_t& y0, uint32_t& y1) {
    x0 = blockIdx.x * kTouchTile; x1 = min(x0 + kTouchTile, g.width);
    y0 = max(ty * kTouchTile, g.row_lo); y1 = min(ty * kTouchTile + kTouchTile, g.row_hi);
    return x0 < x1 && y0 < y1;
}

// This rank's strip of the composited visibility buffer: per tile, the minimum over the keys of every rank that
// touched the tile, two keys (16 bytes) per lane, all peers' loads in flight together.  `union_now[tile]` records
// whether any rank touched it (for the colour rows' way back).  use_flags == 0 fetches every tile from everyone.
__global__ __launch_bounds__(256) void k_composite_tiles(PeerTable peers, uint32_t rank, uint32_t world, TileGrid g, uint32_t* union_now,
                                                         uint32_t use_flags) {
    __shared__ uint32_t s_mask;
    const uint32_t ty = tile_row(g);
    if (ty * kTouchTile >= g.height) return; // the whole workgroup
    const uint32_t tile = ty * g.tiles_x + blockIdx.x;
    if (threadIdx.x < 64) {
        uint32_t f = 0;
        if (threadIdx.x < world) f = use_flags ? peers.touched[threadIdx.x][tile] : 1u;
        const unsigned long long m = __ballot(f != 0u);
        if (threadIdx.x == 0) {
            s_mask = (uint32_t)m;
            union_now[tile] = m != 0ull ? 1u : 0u;
        }
    }
    __syncthreads();
    const uint32_t others = __builtin_amdgcn_readfirstlane(s_mask) & ~(1u << rank);
    uint32_t x0, x1, y0, y1;
    if (!others || !tile_rect(g, ty, x0, x1, y0, y1)) return; // nobody else drew here: this rank's keys are the result
    if ((g.width & 1u) == 0u) {
        // even width: a tile row starts on a 16-byte boundary and holds whole pairs
        const uint32_t per_row = (x1 - x0) / 2u, n = per_row * (y1 - y0);
        for (uint32_t i = threadIdx.x; i < n; i += 256u) {
            const uint32_t ry = i / per_row, rx = i - ry * per_row;
            const size_t at = ((size_t)(y0 + ry) * g.width + x0) / 2u + rx;
            ulonglong2* mine = reinterpret_cast<ulonglong2*>(peers.vis[rank]) + at;
            ulonglong2 v[RDR_COMPOSITE_MAX_RANKS];
#pragma unroll
            for (uint32_t p = 0; p < RDR_COMPOSITE_MAX_RANKS; ++p)
                if ((others >> p) & 1u) v[p] = reinterpret_cast<const ulonglong2*>(peers.vis[p])[at];
            ulonglong2 m = *mine;
#pragma unroll
            for (uint32_t p = 0; p < RDR_COMPOSITE_MAX_RANKS; ++p)
                if ((others >> p) & 1u) {
                    m.x = v[p].x < m.x ? v[p].x : m.x;
                    m.y = v[p].y < m.y ? v[p].y : m.y;
                }
            *mine = m;
        }
    } else {
        const uint32_t per_row = x1 - x0, n = per_row * (y1 - y0);
        for (uint32_t i = threadIdx.x; i < n; i += 256u) {
            const uint32_t ry = i / per_row, rx = i - ry * per_row;
            const size_t at = (size_t)(y0 + ry) * g.width + x0 + rx;
            unsigned long long m = peers.vis[rank][at];
#pragma unroll
            for (uint32_t p = 0; p < RDR_COMPOSITE_MAX_RANKS; ++p)
                if ((others >> p) & 1u) {
                    const unsigned long long v = peers.vis[p][at];
                    m = v < m ? v : m;
                }
            peers.vis[rank][at] = m;
        }
    }
}

// This rank's shaded rows, written into every peer's image, 16 bytes (four pixels) per lane.  A tile that no
// rank touched is all clear colour; it is written only if it was not clear the last time (`union_prev`), so
// that a peer's image never keeps pixels of an earlier frame.  every_tile != 0 (skybox: no pixel is clear
// colour) writes all of them.
__global__ __launch_bounds__(256) void k_push_tiles(PeerTable peers, uint32_t rank, uint32_t world, TileGrid g, const uint32_t* union_now,
                                                    uint32_t* union_prev, uint32_t every_tile) {
    const uint32_t ty = tile_row(g);
    if (ty * kTouchTile >= g.height) return; // the whole workgroup
    const uint32_t tile = ty * g.tiles_x + blockIdx.x;
    const uint32_t now = union_now[tile], need = every_tile | now | union_prev[tile];
    __syncthreads(); // every thread has read union_prev before it changes
    if (threadIdx.x == 0) union_prev[tile] = now;
    uint32_t x0, x1, y0, y1;
    if (!need || !tile_rect(g, ty, x0, x1, y0, y1)) return;
    if ((g.width & 3u) == 0u) {
        const uint32_t per_row = (x1 - x0) / 4u, n = per_row * (y1 - y0);
        for (uint32_t i = threadIdx.x; i < n; i += 256u) {
            const uint32_t ry = i / per_row, rx = i - ry * per_row;
            const size_t at = ((size_t)(y0 + ry) * g.width + x0) / 4u + rx;
            const uint4 c = reinterpret_cast<const uint4*>(peers.color[rank])[at];
#pragma unroll
            for (uint32_t p = 0; p < RDR_COMPOSITE_MAX_RANKS; ++p)
                if (p < world && p != rank) reinterpret_cast<uint4*>(peers.color[p])[at] = c;
        }
    } else {
        const uint32_t per_row = x1 - x0, n = per_row * (y1 - y0);
        for (uint32_t i = threadIdx.x; i < n; i += 256u) {
            const uint32_t ry = i / per_row, rx = i - ry * per_row;
            const size_t at = (size_t)(y0 + ry) * g.width + x0 + rx;
            const uint32_t c = peers.color[rank][at];
#pragma unroll
            for (uint32_t p = 0; p < RDR_COMPOSITE_MAX_RANKS; ++p)
                if (p < world && p != rank) peers.color[p][at] = c;
        }
    }
}

} // namespace

struct rdr_composite_state {
    uint32_t rank = 0, world = 0;
    uint32_t frame = 0;
    PeerTable peers{};
    bool opened[RDR_COMPOSITE_MAX_RANKS][3] = {};
    uint32_t* ctrl = nullptr;              // this renderer's control block (fine-grained device memory)
    volatile uint32_t* timed_out = nullptr; // pinned host word
    uint32_t* timed_out_dev = nullptr;
    bool connected = false;
    bool broken = false; // a wait timed out or a frame of this rank failed while connected: the ranks' phases may differ
    // tiles: per tile of the frame, did any rank touch it this frame / the last time this rank pushed its rows
    uint32_t width = 0, height = 0;
    uint32_t* union_now = nullptr;
    uint32_t* union_prev = nullptr;
    RowSet pushed{0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0}; // the rows union_prev describes
    bool use_flags = true;
};

static bool same_rows(const RowSet& a, const RowSet& b) {
    return a.row_lo == b.row_lo && a.row_hi == b.row_hi && a.band_shift == b.band_shift && a.band_rank == b.band_rank && a.band_world == b.band_world;
}

// Grid over the tiles of the rank's rows; false when it has none.
static bool tile_grid(const rdr_composite_state* c, const RowSet& rows, TileGrid& g, dim3& grid) {
    g.width = c->width; g.height = c->height;
    g.tiles_x = (c->width + kTouchTile - 1) / kTouchTile;
    g.band_rank = rows.band_rank; g.band_world = rows.band_world;
    g.band_tiles_shift = 0; g.tile_row0 = 0;
    uint32_t tile_rows = 0;
    if (rows.band_world > 1u) {
        g.row_lo = 0; g.row_hi = c->height;
        uint32_t band_tiles = (1u << rows.band_shift) / kTouchTile; // a power of two (rdr_set_shade_bands)
        while ((1u << g.band_tiles_shift) < band_tiles) ++g.band_tiles_shift;
        tile_rows = (rdr_band_rows(c->height, rows.band_shift, rows.band_rank, rows.band_world) + kTouchTile - 1) / kTouchTile;
    } else {
        g.row_lo = rows.row_lo; g.row_hi = rows.row_hi;
        if (rows.row_hi <= rows.row_lo) return false;
        g.tile_row0 = rows.row_lo / kTouchTile;
        tile_rows = (rows.row_hi + kTouchTile - 1) / kTouchTile - g.tile_row0;
    }
    grid = dim3(g.tiles_x, tile_rows);
    return tile_rows != 0;
}

int rdr_composite_create(rdr_composite_state** out) {
    rdr_composite_state* c = new rdr_composite_state();
    void* p = nullptr;
    // fine-grained: peers' stores must become visible to a kernel that is already polling
    RDR_HIP(hipExtMallocWithFlags(&p, kCtrlWords * sizeof(uint32_t), hipDeviceMallocFinegrained));
    RDR_HIP(hipMemset(p, 0, kCtrlWords * sizeof(uint32_t)));
    c->ctrl = static_cast<uint32_t*>(p);
    void* host = nullptr;
    void* dev = nullptr;
    RDR_HIP(hipHostMalloc(&host, 64, hipHostMallocMapped));
    std::memset(host, 0, 64);
    RDR_HIP(hipHostGetDevicePointer(&dev, host, 0));
    c->timed_out = static_cast<volatile uint32_t*>(host);
    c->timed_out_dev = static_cast<uint32_t*>(dev);
    RDR_HIP(hipDeviceSynchronize());
    *out = c;
    return 0;
}

void rdr_composite_close(rdr_composite_state* c) {
    if (!c) return;
    for (uint32_t p = 0; p < RDR_COMPOSITE_MAX_RANKS; ++p) {
        if (c->opened[p][0]) (void)hipIpcCloseMemHandle(c->peers.vis[p]);
        if (c->opened[p][1]) (void)hipIpcCloseMemHandle(c->peers.color[p]);
        if (c->opened[p][2]) (void)hipIpcCloseMemHandle(c->peers.ctrl[p]);
        c->opened[p][0] = c->opened[p][1] = c->opened[p][2] = false;
    }
    c->connected = false;
    c->broken = false;
    if (c->timed_out) *c->timed_out = 0;
    if (c->union_now) (void)hipFree(c->union_now);
    if (c->union_prev) (void)hipFree(c->union_prev);
    c->union_now = c->union_prev = nullptr;
}

void rdr_composite_destroy(rdr_composite_state* c) {
    if (!c) return;
    rdr_composite_close(c);
    if (c->ctrl) (void)hipFree(c->ctrl);
    if (c->timed_out) (void)hipHostFree(const_cast<uint32_t*>(c->timed_out));
    delete c;
}

int rdr_composite_export_handles(rdr_composite_state* c, void* vis, void* color, rdr_composite_handle* out) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "an IPC handle is 64 bytes");
    hipIpcMemHandle_t h;
    RDR_HIP(hipIpcGetMemHandle(&h, vis));
    std::memcpy(out->vis, &h, 64);
    RDR_HIP(hipIpcGetMemHandle(&h, color));
    std::memcpy(out->color, &h, 64);
    RDR_HIP(hipIpcGetMemHandle(&h, c->ctrl));
    std::memcpy(out->ctrl, &h, 64);
    std::memset(out->info, 0, sizeof(out->info));
    std::memcpy(out->info, &c->frame, 4); // where this renderer's phase numbers stand (aligned across the ranks at connect)
    return 0;
}

bool rdr_composite_connected(const rdr_composite_state* c) { return c && c->connected; }

int rdr_composite_open(rdr_composite_state* c, uint32_t rank, uint32_t world, const rdr_composite_handle* handles, void* vis, void* color,
                       uint32_t width, uint32_t height) {
    RDR_CHECK(world >= 1 && world <= RDR_COMPOSITE_MAX_RANKS && rank < world, "rdr_composite_connect: rank / world out of range (at most 16 ranks)");
    RDR_CHECK(!c->connected, "rdr_composite_connect: already connected");
    c->rank = rank; c->world = world;
    // Phase numbers only ever grow, and every rank continues from the same one: the largest exchange counter among
    // the handles (this rank's own included).  Whatever an earlier connection left in a control block is smaller than
    // the first phase of this one, so a stale word cannot satisfy a wait.
    for (uint32_t p = 0; p < world; ++p) {
        uint32_t theirs = 0;
        std::memcpy(&theirs, handles[p].info, 4);
        if (p != rank && (int32_t)(theirs - c->frame) > 0) c->frame = theirs;
    }
    c->broken = false;
    *c->timed_out = 0;
    c->width = width; c->height = height;
    const size_t npx = (size_t)width * height, words = rdr_touched_words(width, height);
    RDR_HIP(hipMalloc(reinterpret_cast<void**>(&c->union_now), words * 4));
    RDR_HIP(hipMalloc(reinterpret_cast<void**>(&c->union_prev), words * 4));
    RDR_HIP(hipMemset(c->union_now, 0, words * 4));
    c->pushed = RowSet{0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0}; // nothing is known about the peers' images yet
    if (const char* e = std::getenv("RDR_COMPOSITE_TILE_FLAGS")) c->use_flags = std::atoi(e) != 0; // measurement hook: 0 = every tile
    for (uint32_t p = 0; p < world; ++p) {
        if (p == rank) {
            c->peers.vis[p] = static_cast<unsigned long long*>(vis);
            c->peers.touched[p] = reinterpret_cast<uint32_t*>(c->peers.vis[p] + npx);
            c->peers.color[p] = static_cast<uint32_t*>(color);
            c->peers.ctrl[p] = c->ctrl;
            continue;
        }
        hipIpcMemHandle_t h;
        void* ptr = nullptr;
        std::memcpy(&h, handles[p].vis, 64);
        RDR_HIP(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
        c->peers.vis[p] = static_cast<unsigned long long*>(ptr); c->opened[p][0] = true;
        c->peers.touched[p] = reinterpret_cast<uint32_t*>(c->peers.vis[p] + npx);
        std::memcpy(&h, handles[p].color, 64);
        RDR_HIP(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
        c->peers.color[p] = static_cast<uint32_t*>(ptr); c->opened[p][1] = true;
        std::memcpy(&h, handles[p].ctrl, 64);
        RDR_HIP(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
        c->peers.ctrl[p] = static_cast<uint32_t*>(ptr); c->opened[p][2] = true;
    }
    c->connected = true;
    return 0;
}

bool rdr_composite_timed_out(const rdr_composite_state* c) { return c && *c->timed_out != 0; }

void rdr_composite_mark_broken(rdr_composite_state* c) {
    if (c && c->connected) c->broken = true;
}

int rdr_composite_check(rdr_composite_state* c, const char* who) {
    if (*c->timed_out) {
        *c->timed_out = 0;
        c->broken = c->connected;
        rdr_set_error(std::string(who) + ": a peer rank did not reach the exchange within 5 s (the frame's image is not valid); "
                      "disconnect and connect again on every rank");
        return 1;
    }
    if (c->broken) {
        rdr_set_error(std::string(who) + ": the exchange is out of step with the other ranks after an earlier failure; "
                      "disconnect and connect again on every rank");
        return 1;
    }
    return 0;
}

// keys of the rank's rows: wait for every rank's keys, take the minimum into this rank's buffer
int rdr_composite_launch_exchange(rdr_composite_state* c, const RowSet& rows, hipStream_t stream) {
    RDR_CHECK(c->connected, "rdr_composite_exchange: not connected");
    c->frame++;
    const uint32_t base = 2u * c->frame;
    hipLaunchKernelGGL(k_signal_wait, dim3(1), dim3(64), 0, stream, c->peers, c->ctrl, c->rank, c->world, base + 1u, c->timed_out_dev);
    dim3 grid;
    TileGrid g;
    if (tile_grid(c, rows, g, grid))
        hipLaunchKernelGGL(k_composite_tiles, grid, dim3(256), 0, stream, c->peers, c->rank, c->world, g, c->union_now, c->use_flags ? 1u : 0u);
    RDR_HIP(hipGetLastError());
    return 0;
}

// the rank's colour rows into every peer's image, then wait until every peer's rows are here
int rdr_composite_launch_gather(rdr_composite_state* c, const RowSet& rows, bool every_tile, hipStream_t stream) {
    RDR_CHECK(c->connected, "rdr_composite_gather: not connected");
    RDR_CHECK(c->frame > 0, "rdr_composite_gather: no exchange has run for this frame");
    const uint32_t base = 2u * c->frame;
    dim3 grid;
    TileGrid g;
    if (tile_grid(c, rows, g, grid)) {
        if (!same_rows(rows, c->pushed)) {
            // other rows than the last time: what the peers' images hold in them is unknown, so every tile counts as written before
            RDR_HIP(hipMemsetAsync(c->union_prev, 0x01, (size_t)rdr_touched_words(c->width, c->height) * 4, stream));
            c->pushed = rows;
        }
        hipLaunchKernelGGL(k_push_tiles, grid, dim3(256), 0, stream, c->peers, c->rank, c->world, g, c->union_now, c->union_prev,
                           (every_tile || !c->use_flags) ? 1u : 0u);
    }
    hipLaunchKernelGGL(k_signal_wait, dim3(1), dim3(64), 0, stream, c->peers, c->ctrl, c->rank, c->world, base + 2u, c->timed_out_dev);
    RDR_HIP(hipGetLastError());
    return 0;
}
