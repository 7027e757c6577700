// atomic_peak.hip — what the chip can do with the rasteriser's one memory operation: a 64-bit atomic minimum
// at agent scope whose result is not used, into a visibility buffer of 3840 x 2160 keys (66 MB).
//
// Three address patterns, each as a stand-alone kernel that does nothing else (addresses and keys come from a
// few integer instructions per atomic), so the rates are ceilings for k_raster / k_mid_raster, whose fragments
// are issued exactly like this (rdr_geometry.hip: emit_fragment):
//   random   every lane of a wave hits an unrelated key of the buffer
//   box      the per-lane scan's pattern: the lanes of a wave are the triangles of one meshlet, i.e. small
//            pixel boxes scattered over a region of a few dozen pixels; a lane walks its box row by row
//   same     every lane of the machine hits the same key (the serialisation limit of one address)
// Built as a shared library (profiles/Makefile) so that bench.py can measure the ceiling on the GPU it runs on;
// profiles/atomic_peak.py prints all three.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__device__ __forceinline__ uint32_t mix(uint32_t x) { // a cheap integer hash (three rounds of xorshift-multiply)
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

__device__ __forceinline__ void key_min(unsigned long long* p, uint32_t depth_bits, uint32_t id) {
    const unsigned long long key = ((unsigned long long)(depth_bits & 0x7FFFFFFFu) << 32) | id;
    __hip_atomic_fetch_min(p, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// mode 0: random; mode 2: same address
__global__ __launch_bounds__(256) void k_scatter(unsigned long long* vis, uint32_t n_keys, uint32_t per_lane, uint32_t mode, uint32_t seed) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t h = mix(gid * 0x9E3779B9u + seed);
    for (uint32_t i = 0; i < per_lane; ++i) {
        h = mix(h + i);
        const uint32_t at = mode == 2u ? 12345u : h % n_keys;
        key_min(vis + at, mix(h ^ 0xA5A5A5A5u), gid);
    }
}

// mode 1: one wave = one "meshlet" whose 64 lanes hold boxes of bw x bh pixels inside a region x region pixel
// window at a random place of the frame; `rounds` meshlets per wave.
__global__ __launch_bounds__(64) void k_boxes(unsigned long long* vis, uint32_t width, uint32_t height, uint32_t region, uint32_t bw, uint32_t bh,
                                              uint32_t rounds, uint32_t seed) {
    const uint32_t lane = threadIdx.x;
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t m = mix((blockIdx.x * rounds + r) * 0x9E3779B9u + seed); // the same in every lane: the meshlet's place
        const uint32_t ox = m % (width - region - bw), oy = (m >> 12) % (height - region - bh);
        const uint32_t t = mix(m + lane * 0x85EBCA6Bu);
        const uint32_t x0 = ox + t % region, y0 = oy + (t >> 10) % region;
        const uint32_t depth = mix(t ^ seed);
        for (uint32_t y = 0; y < bh; ++y)
            for (uint32_t x = 0; x < bw; ++x) key_min(vis + (size_t)(y0 + y) * width + (x0 + x), depth + x + y, (blockIdx.x << 7) | lane);
    }
}

} // namespace

// G atomics per second of one pattern; < 0 on a HIP error.  mode 0 random, 1 boxes (region, bw, bh), 2 same address.
extern "C" double atomic_peak_rate(int device, int mode, uint32_t region, uint32_t bw, uint32_t bh, int repeats) {
    const uint32_t W = 3840, H = 2160, n = W * H;
    if (hipSetDevice(device) != hipSuccess) return -1.0;
    unsigned long long* vis = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&vis), (size_t)n * 8) != hipSuccess) return -1.0;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    double best = 0.0;
    for (int rep = 0; rep < repeats + 1; ++rep) { // the first launch is a warm-up
        (void)hipMemset(vis, 0x7F, (size_t)n * 8);
        (void)hipDeviceSynchronize();
        double count = 0.0;
        (void)hipEventRecord(a, 0);
        if (mode == 1) {
            const uint32_t blocks = 65536, rounds = 8;
            hipLaunchKernelGGL(k_boxes, dim3(blocks), dim3(64), 0, 0, vis, W, H, region, bw, bh, rounds, 77u + (uint32_t)rep);
            count = (double)blocks * rounds * 64.0 * bw * bh;
        } else {
            const uint32_t blocks = mode == 2 ? 2048 : 16384, per_lane = mode == 2 ? 8 : 16;
            hipLaunchKernelGGL(k_scatter, dim3(blocks), dim3(256), 0, 0, vis, n, per_lane, (uint32_t)mode, 77u + (uint32_t)rep);
            count = (double)blocks * 256.0 * per_lane;
        }
        (void)hipEventRecord(b, 0);
        if (hipEventSynchronize(b) != hipSuccess) { best = -1.0; break; }
        float ms = 0.0f;
        (void)hipEventElapsedTime(&ms, a, b);
        if (rep > 0 && ms > 0.0f && count / (ms * 1e-3) / 1e9 > best) best = count / (ms * 1e-3) / 1e9;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    (void)hipFree(vis);
    return best;
}
