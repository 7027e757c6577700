// How fast a SIMD of this chip issues wave64 vector instructions: what the issue fractions that bench.py and
// pmc_traffic.py report for the frame's kernels are fractions of.  Two kernels of dependence-free chains, sixteen per
// lane (no wave ever waits for itself), plain v_fma_f32 and packed v_pk_fma_f32, each run with 1, 2, 4 and 8 waves per
// SIMD on every CU.  Prints SIMD cycles per instruction from the kernel's duration and the shader clock (s_memtime
// ticks of the one-wave run over its duration).
//   make -C profiles valu_rate && profiles/valu_rate        (built without the SLP vectoriser: it would pack the plain kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kChains = 16, kIters = 4096;
typedef float float2v __attribute__((ext_vector_type(2)));

template <bool PACKED>
__global__ void k_fma(float* out, unsigned long long* cycles, float a, float b) {
    float v[kChains];
#pragma unroll
    for (int i = 0; i < kChains; ++i) v[i] = (float)(threadIdx.x + i);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; ++it) {
        if (PACKED) {
#pragma unroll
            for (int i = 0; i < kChains; i += 2) {
                float2v x = {v[i], v[i + 1]};
                x = __builtin_elementwise_fma(x, float2v{a, a}, float2v{b, b});
                v[i] = x.x; v[i + 1] = x.y;
            }
        } else {
#pragma unroll
            for (int i = 0; i < kChains; ++i) v[i] = __builtin_fmaf(v[i], a, b);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < kChains; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

#define CHECK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__); return 1; } } while (0)

template <bool PACKED>
int run(int cus, double& clock_ghz) {
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int threads = 256 * waves_per_simd;                       // per CU: 4 SIMDs x waves x 64 lanes
        const int bt = threads > 1024 ? 1024 : threads, blocks = cus * (threads / bt);
        float* out; unsigned long long* cyc;
        CHECK(hipMalloc(&out, sizeof(float) * blocks * bt));
        CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * blocks * bt / 64));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        k_fma<PACKED><<<blocks, bt>>>(out, cyc, 1.0001f, 0.5f); // warm-up
        CHECK(hipEventRecord(e0));
        k_fma<PACKED><<<blocks, bt>>>(out, cyc, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(blocks * bt / 64);
        CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long longest = 0;
        for (auto c : h) longest = c > longest ? c : longest;
        if (waves_per_simd == 1 && !PACKED) clock_ghz = (double)longest / (ms * 1e6); // the wave spans the launch
        const double instructions = (double)kIters * (PACKED ? kChains / 2 : kChains) * waves_per_simd; // per SIMD
        printf("%-13s %d waves/SIMD: %.2f SIMD cycles per wave64 instruction (launch %.3f ms at %.2f GHz)\n", PACKED ? "v_pk_fma_f32" : "v_fma_f32",
               waves_per_simd, ms * 1e6 * clock_ghz / instructions, ms, clock_ghz);
        CHECK(hipFree(out)); CHECK(hipFree(cyc));
    }
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    double clock_ghz = 2.0;
    return run<false>(p.multiProcessorCount, clock_ghz) || run<true>(p.multiProcessorCount, clock_ghz);
}
