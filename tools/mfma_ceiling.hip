// Bare MFMA loops on RANDOM operands (no memory traffic inside the loop): what the matrix pipes of THIS device deliver
// and at which clock, as the ceiling the linear kernels are priced against (MI355X_MICROARCH.md
// "DVFS give-back" items 5-7: devices differ, and bf16 MFMA loops on random data hold a clock well under 2.4 GHz).
//
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_ceiling.hip -o tools/mfma_ceiling.bin && tools/mfma_ceiling.bin
//
// Variants: bf16 16x16x32 (the pp kernel's instruction), bf16 32x32x16, fp32 32x32x2; 1 or 2 waves per SIMD.
// Per variant: >= 2 s of back-to-back launches, then TFLOP/s from HIP events and the in-kernel clock
// d(s_memtime) / d(s_memrealtime) x 100 MHz (median over workgroups).  One JSON line per variant.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));        \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float rnd(unsigned seed) { return (hash32(seed) >> 8) * (1.0f / 8388608.0f) - 1.0f; }

struct Stamps { unsigned long long cyc, rt; };

// kind 0: v_mfma_f32_16x16x32_bf16, 16 accumulators; 1: v_mfma_f32_32x32x16_bf16, 4 accumulators; 2: v_mfma_f32_32x32x2_f32, 4 accumulators
template <int KIND>
__global__ __launch_bounds__(512) void mfma_loop(int iters, float *sink, Stamps *stamps) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[4], b[4];
    float fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[i][e] = (__bf16)rnd(tid * 64 + i * 8 + e);
            b[i][e] = (__bf16)rnd(tid * 64 + 32 + i * 8 + e);
        }
        fa[i] = rnd(tid * 8 + i);
        fb[i] = rnd(tid * 8 + 4 + i);
    }
    f32x4 acc4[16];
    f32x16 acc16[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc4[i] = (f32x4){0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc16[i][r] = 0.0f;
    unsigned long long c0, c1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc4[4 * i + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc4[4 * i + j], 0, 0, 0);
        } else if (KIND == 1) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc16[2 * i + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2 * r + i], b[2 * r + j], acc16[2 * i + j], 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc16[2 * i + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2 * r + i], fb[2 * r + j], acc16[2 * i + j], 0, 0, 0);
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc4[i][0] + acc4[i][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc16[i][0] + acc16[i][15];
    if (s == 12345.678f) sink[tid] = s;  // keeps the loop alive, never true in practice
    if (threadIdx.x == 0) stamps[blockIdx.x] = Stamps{c1 - c0, r1 - r0};
}

template <int KIND>
static void run(const char *name, int waves_per_simd, double flop_per_mfma, int mfma_per_iter, double spec_tf) {
    int dev = 0, cus = 0;
    CHECK(hipGetDevice(&dev));
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int threads = 256 * waves_per_simd, blocks = cus;
    const int iters = KIND == 2 ? 20000 : 100000;
    float *sink;
    Stamps *stamps;
    CHECK(hipMalloc(&sink, (size_t)blocks * threads * sizeof(float)));
    CHECK(hipMalloc(&stamps, blocks * sizeof(Stamps)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto launch = [&]() { hipLaunchKernelGGL(mfma_loop<KIND>, dim3(blocks), dim3(threads), 0, 0, iters, sink, stamps); };
    const auto t_end = std::chrono::steady_clock::now() + std::chrono::milliseconds(2500);
    while (std::chrono::steady_clock::now() < t_end) {
        for (int i = 0; i < 4; ++i) launch();
        CHECK(hipDeviceSynchronize());
    }
    const int reps = 8;
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    std::vector<Stamps> h(blocks);
    CHECK(hipMemcpy(h.data(), stamps, blocks * sizeof(Stamps), hipMemcpyDeviceToHost));
    std::vector<double> clk(blocks), cyc(blocks);
    for (int i = 0; i < blocks; ++i) {
        clk[i] = (double)h[i].cyc / (double)std::max<unsigned long long>(h[i].rt, 1) * 0.1;
        cyc[i] = (double)h[i].cyc;
    }
    std::sort(clk.begin(), clk.end());
    std::sort(cyc.begin(), cyc.end());
    const double waves = (double)blocks * threads / 64.0;
    const double flops = waves * iters * mfma_per_iter * flop_per_mfma;
    const double tf = flops / (ms * 1e-3) / 1e12;
    const double cyc_per_mfma = cyc[blocks / 2] / ((double)iters * mfma_per_iter) / waves_per_simd;  // per SIMD issue slot
    printf("{\"variant\": \"%s\", \"waves_per_simd\": %d, \"tflops\": %.1f, \"frac_of_spec\": %.4f, \"in_kernel_clock_ghz\": %.3f, "
           "\"cycles_per_mfma_per_simd\": %.2f, \"launch_ms\": %.3f, \"cus\": %d}\n",
           name, waves_per_simd, tf, tf / spec_tf, clk[blocks / 2], cyc_per_mfma, ms, cus);
    CHECK(hipFree(sink));
    CHECK(hipFree(stamps));
}

int main() {
    run<0>("bf16_16x16x32", 1, 2.0 * 16 * 16 * 32, 16, 2500.0);
    run<0>("bf16_16x16x32", 2, 2.0 * 16 * 16 * 32, 16, 2500.0);
    run<1>("bf16_32x32x16", 1, 2.0 * 32 * 32 * 16, 8, 2500.0);
    run<1>("bf16_32x32x16", 2, 2.0 * 32 * 32 * 16, 8, 2500.0);
    run<2>("f32_32x32x2", 1, 2.0 * 32 * 32 * 2, 8, 157.3);
    return 0;
}
