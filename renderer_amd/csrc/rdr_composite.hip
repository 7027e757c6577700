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
// tells every peer "I have reached phase n" by storing n into its own slot of the peer's block
// (k_signal), and waits by polling its own block until every slot has reached n (k_wait).  Per frame f:
//   3f+1  my keys are complete            (peers may read my visibility buffer)
//   3f+2  I have finished reading yours   (after k_composite_min)
//   3f+3  my colour rows are in your image (after k_push_color)
// A rank waits for 3f+1 before reading keys and for 3f+3 at the end of the frame; the latter also tells
// it that every peer is past 3f+2, so clearing the visibility buffer for the next frame is safe.  Every
// wait is bounded (about five seconds of wall clock): a peer that never arrives turns into an error
// flag, not a hang.
#include <algorithm>
#include <cstring>

#include "rdr_internal.h"

namespace {

constexpr uint32_t kCtrlWords = 64;          // control block: arrived[rank] phase numbers, one word each
constexpr long long kWaitTicks = 500000000;  // wall_clock64 runs at 100 MHz: 5 s

struct PeerTable {
    unsigned long long* vis[RDR_COMPOSITE_MAX_RANKS];
    uint32_t* color[RDR_COMPOSITE_MAX_RANKS];
    uint32_t* ctrl[RDR_COMPOSITE_MAX_RANKS];
};

// Phase `phase` reached: written into this rank's slot of every peer's control block.  The data the
// phase stands for was written by earlier kernels of this stream, which have completed.
__global__ void k_signal(PeerTable peers, uint32_t rank, uint32_t world, uint32_t phase) {
    const uint32_t p = threadIdx.x;
    if (p < world && p != rank) __hip_atomic_store(&peers.ctrl[p][rank], phase, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Waits until every peer's slot in this rank's control block has reached `phase`; gives up after
// kWaitTicks and records that in *timed_out (pinned host memory).
__global__ void k_wait(const uint32_t* ctrl, uint32_t rank, uint32_t world, uint32_t phase, uint32_t* timed_out) {
    const uint32_t p = threadIdx.x;
    if (p >= world || p == rank) return;
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

// This rank's strip of the composited visibility buffer: minimum over every rank's keys, two keys (16
// bytes) per lane, all peers' loads in flight together.
__global__ __launch_bounds__(256) void k_composite_min(PeerTable peers, uint32_t rank, uint32_t world, size_t first_pair, size_t n_pairs) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    ulonglong2* mine = reinterpret_cast<ulonglong2*>(peers.vis[rank]) + first_pair + i;
    ulonglong2 v[RDR_COMPOSITE_MAX_RANKS];
#pragma unroll
    for (uint32_t p = 0; p < RDR_COMPOSITE_MAX_RANKS; ++p)
        if (p < world) v[p] = reinterpret_cast<const ulonglong2*>(peers.vis[p])[first_pair + i];
    ulonglong2 m = v[0];
#pragma unroll
    for (uint32_t p = 1; p < RDR_COMPOSITE_MAX_RANKS; ++p)
        if (p < world) {
            m.x = v[p].x < m.x ? v[p].x : m.x;
            m.y = v[p].y < m.y ? v[p].y : m.y;
        }
    *mine = m;
}

// This rank's shaded rows, written into every peer's image, 16 bytes (four pixels) per lane.
__global__ __launch_bounds__(256) void k_push_color(PeerTable peers, uint32_t rank, uint32_t world, size_t first_quad, size_t n_quads) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_quads) return;
    const uint4 c = reinterpret_cast<const uint4*>(peers.color[rank])[first_quad + i];
#pragma unroll
    for (uint32_t p = 0; p < RDR_COMPOSITE_MAX_RANKS; ++p)
        if (p < world && p != rank) reinterpret_cast<uint4*>(peers.color[p])[first_quad + i] = c;
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
};

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
    return 0;
}

int rdr_composite_open(rdr_composite_state* c, uint32_t rank, uint32_t world, const rdr_composite_handle* handles, void* vis, void* color) {
    RDR_CHECK(world >= 1 && world <= RDR_COMPOSITE_MAX_RANKS && rank < world, "rdr_composite_connect: rank / world out of range (at most 16 ranks)");
    RDR_CHECK(!c->connected, "rdr_composite_connect: already connected");
    c->rank = rank; c->world = world; // the frame number keeps counting across connections: phases only ever grow
    for (uint32_t p = 0; p < world; ++p) {
        if (p == rank) {
            c->peers.vis[p] = static_cast<unsigned long long*>(vis);
            c->peers.color[p] = static_cast<uint32_t*>(color);
            c->peers.ctrl[p] = c->ctrl;
            continue;
        }
        hipIpcMemHandle_t h;
        void* ptr = nullptr;
        std::memcpy(&h, handles[p].vis, 64);
        RDR_HIP(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
        c->peers.vis[p] = static_cast<unsigned long long*>(ptr); c->opened[p][0] = true;
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

int rdr_composite_check(rdr_composite_state* c, const char* who) {
    if (*c->timed_out) {
        *c->timed_out = 0;
        rdr_set_error(std::string(who) + ": a peer rank did not reach the exchange within 5 s (the frame's image is not valid)");
        return 1;
    }
    return 0;
}

// keys of rows [row_lo, row_hi): wait for every rank's keys, take the minimum into this rank's buffer
int rdr_composite_launch_exchange(rdr_composite_state* c, uint32_t width, uint32_t row_lo, uint32_t row_hi, hipStream_t stream) {
    RDR_CHECK(c->connected, "rdr_composite_exchange: not connected");
    c->frame++;
    const uint32_t base = 3u * c->frame;
    hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, stream, c->peers, c->rank, c->world, base + 1u);
    hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, stream, c->ctrl, c->rank, c->world, base + 1u, c->timed_out_dev);
    const size_t first = (size_t)row_lo * width, count = (size_t)(row_hi - row_lo) * width;
    // 16-byte units; an odd first key or count falls back to whole-buffer alignment by widening to even bounds
    const size_t first_pair = first / 2, end_pair = (first + count + 1) / 2;
    if (end_pair > first_pair)
        hipLaunchKernelGGL(k_composite_min, dim3((uint32_t)((end_pair - first_pair + 255) / 256)), dim3(256), 0, stream, c->peers, c->rank, c->world,
                           first_pair, end_pair - first_pair);
    hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, stream, c->peers, c->rank, c->world, base + 2u);
    RDR_HIP(hipGetLastError());
    return 0;
}

// colour rows [row_lo, row_hi) into every peer's image, then wait until every peer's rows are here
int rdr_composite_launch_gather(rdr_composite_state* c, uint32_t width, uint32_t row_lo, uint32_t row_hi, hipStream_t stream) {
    RDR_CHECK(c->connected, "rdr_composite_gather: not connected");
    RDR_CHECK(c->frame > 0, "rdr_composite_gather: no exchange has run for this frame");
    const uint32_t base = 3u * c->frame;
    const size_t first = (size_t)row_lo * width, count = (size_t)(row_hi - row_lo) * width;
    RDR_CHECK(first % 4 == 0 && count % 4 == 0, "rdr_composite_gather: strips must start and end on a multiple of four pixels (width x rows)");
    if (count)
        hipLaunchKernelGGL(k_push_color, dim3((uint32_t)((count / 4 + 255) / 256)), dim3(256), 0, stream, c->peers, c->rank, c->world, first / 4, count / 4);
    hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, stream, c->peers, c->rank, c->world, base + 3u);
    hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, stream, c->ctrl, c->rank, c->world, base + 3u, c->timed_out_dev);
    RDR_HIP(hipGetLastError());
    return 0;
}
