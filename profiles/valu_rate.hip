// How fast a SIMD of this chip issues wave64 vector instructions: what the issue fractions that bench.py and
// pmc_traffic.py report for the frame's kernels are fractions of.  Kernels of dependence-free chains, sixteen per lane
// (no wave ever waits for itself), one instruction class each (inline assembly, so that the compiler neither packs
// nor folds them), run with 1, 2, 4 and 8 waves per SIMD on every CU, and once more with the upper half of every
// wave switched off.  Prints nanoseconds and SIMD cycles per instruction from the kernel's duration and the shader
// clock (s_memtime ticks of a one-wave run over its duration).
//   make -C profiles valu_rate && profiles/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kChains = 16, kIters = 4096;

enum Op { kFma, kPkFma, kAddU32, kMov, kCndmask, kMulLo, kRcp, kReadlane, kLshlAdd64, kFmaSgpr, kMulSgpr, kAddSgpr, kFmaConst, kMulF32, kOps };
const char* const kNames[kOps] = {"v_fma_f32", "v_pk_fma_f32", "v_add_u32", "v_mov_b32", "v_cndmask_b32", "v_mul_lo_u32", "v_rcp_f32",
                                  "v_readlane_b32", "v_lshl_add_u64", "v_fma_f32 v,s,v", "v_mul_f32 s,v", "v_add_u32 s,v", "v_fma_f32 v,v,1.0", "v_mul_f32"};

template <int OP, bool HALF>
__global__ void k_chain(float* out, unsigned long long* cycles, float a, float b) {
    float v[kChains];
    double w[kChains];
    int s_sink = 0;
    const unsigned long long lane_mask = __ballot(threadIdx.x & 1u);
#pragma unroll
    for (int i = 0; i < kChains; ++i) { v[i] = (float)(threadIdx.x + i); w[i] = (double)v[i]; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (!HALF || (threadIdx.x & 32u) == 0u) {
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int i = 0; i < kChains; ++i) {
                if (OP == kFma) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
                if (OP == kPkFma) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[i]) : "v"((double)a), "v"((double)b));
                if (OP == kAddU32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
                if (OP == kMov) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(a));
                if (OP == kCndmask) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "s"(lane_mask));
                if (OP == kFmaSgpr) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "s"(a), "v"(b));
                if (OP == kMulSgpr) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v[i]) : "s"(a));
                if (OP == kAddSgpr) asm volatile("v_add_u32 %0, %1, %0" : "+v"(v[i]) : "s"(a));
                if (OP == kFmaConst) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(v[i]) : "v"(a));
                if (OP == kMulF32) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(v[i]) : "v"(a));
                if (OP == kMulLo) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
                if (OP == kRcp) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
                if (OP == kReadlane) asm volatile("v_readlane_b32 %0, %1, 3" : "+s"(s_sink) : "v"(v[i]));
                if (OP == kLshlAdd64) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(w[i]) : "v"((double)a));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = (float)s_sink;
#pragma unroll
    for (int i = 0; i < kChains; ++i) s += v[i] + (float)w[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

#define CHECK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__); return 1; } } while (0)

template <int OP, bool HALF>
int run(int cus, double& clock_ghz) {
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int threads = 256 * waves_per_simd;                       // per CU: 4 SIMDs x waves x 64 lanes
        const int bt = threads > 1024 ? 1024 : threads, blocks = cus * (threads / bt);
        float* out; unsigned long long* cyc;
        CHECK(hipMalloc(&out, sizeof(float) * blocks * bt));
        CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * blocks * bt / 64));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        k_chain<OP, HALF><<<blocks, bt>>>(out, cyc, 1.0001f, 0.5f); // warm-up
        CHECK(hipEventRecord(e0));
        k_chain<OP, HALF><<<blocks, bt>>>(out, cyc, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(blocks * bt / 64);
        CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long longest = 0;
        for (auto c : h) longest = c > longest ? c : longest;
        if (waves_per_simd == 1 && OP == kFma && !HALF) clock_ghz = (double)longest / (ms * 1e6); // the wave spans the launch
        const double instructions = (double)kIters * kChains * waves_per_simd; // per SIMD
        printf("%-15s %s %d waves/SIMD: %.2f ns = %.2f SIMD cycles per wave64 instruction (launch %.3f ms at %.2f GHz)\n", kNames[OP],
               HALF ? "lanes 0-31" : "all lanes ", waves_per_simd, ms * 1e6 / instructions, ms * 1e6 * clock_ghz / instructions, ms, clock_ghz);
        CHECK(hipFree(out)); CHECK(hipFree(cyc));
    }
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    double g = 2.0;
    const int c = p.multiProcessorCount;
    return run<kFma, false>(c, g) || run<kFma, true>(c, g) || run<kPkFma, false>(c, g) || run<kAddU32, false>(c, g) || run<kAddU32, true>(c, g) ||
           run<kMov, false>(c, g) || run<kCndmask, false>(c, g) || run<kMulLo, false>(c, g) || run<kRcp, false>(c, g) ||
           run<kReadlane, false>(c, g) || run<kLshlAdd64, false>(c, g) || run<kFmaSgpr, false>(c, g) || run<kMulSgpr, false>(c, g) ||
           run<kAddSgpr, false>(c, g) || run<kFmaConst, false>(c, g) || run<kMulF32, false>(c, g);
}
