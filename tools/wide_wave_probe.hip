// Would ONE wave per SIMD with a 128 x 128 wave tile lift the data-movement bound of the bf16 layer kernel?  The product kernel
// (pp16::linear_bf16_pp_kernel) runs the 256 x 256 x 64 K-step with 8 waves of 128 x 64: 24 ds_read_b128 per wave = 192 KiB of
// fragment reads per K-step and CU, 75 % of the LDS port time at full matrix rate (tools/loader_wave_probe.hip: 1.61 PF with the
// reads, 1.37-1.40 PF with reads + LDS-DMA).  Four waves of 128 x 128 (256 accumulators, the fp32 half-tile kernel's register
// budget) read 32 fragments each = 128 KiB per K-step, a third less.  Same probe discipline as loader_wave_probe: the K-step of
// that structure with NO barrier, epilogue or stores - an upper bound for a kernel built this way.
//   modes: 0 MFMA only | 1 + fragment reads | 2 + reads + the K-step's 64 LDS-DMA pieces issued by the four waves (16 each)
//   MF: 0 = v_mfma_f32_32x32x16_bf16 (64 per wave and K-step), 1 = v_mfma_f32_16x16x32_bf16 (128)
//   hipcc -O3 --offload-arch=gfx950 tools/wide_wave_probe.hip -o tools/wide_wave_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ float rnd(unsigned s) { return (hash32(s) >> 8) * (1.0f / 8388608.0f) - 1.0f; }

constexpr int kLds = 128 * 1024;

template <int MODE, int MF>
__global__ __launch_bounds__(256, 1) void probe_kernel(int ksteps, const float4 *__restrict__ big, const float4 *__restrict__ small_, float *sink,
                                                        unsigned long long *out) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr bool READS = MODE >= 1, SELF = MODE >= 2;
    for (int i = threadIdx.x; i < kLds / 4; i += blockDim.x) reinterpret_cast<unsigned *>(smem)[i] = hash32(tid * 977 + i) & 0xBF7FBF7Fu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs_big = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(big), 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_small = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(small_), 0, 0x7fffffff, 0x00020000);
    unsigned long long c0 = 0, c1 = 0, r0 = 0, r1 = 0;
    const unsigned agroup = (blockIdx.x % 8u) + 8u * ((blockIdx.x / 8u) / 4u);  // activation stream shared by the 4 column tiles on one XCD
    // fragments of one 32-deep half of the K-step: 8 A (4 row blocks of 32 x 2 chunks of 16, or 8 row blocks of 16 x 1 chunk of 32)
    // and 8 B, double-buffered: 2 x 16 x 4 = 128 registers beside the 256 accumulators
    bf16x8 a[2][8], b[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) { a[h][i][e] = (__bf16)rnd(tid * 512 + h * 64 + i * 8 + e); b[h][i][e] = (__bf16)rnd(tid * 512 + 256 + h * 64 + i * 8 + e); }
    f32x16 acc[MF == 0 ? 16 : 1];
    f32x4 acc4[MF == 0 ? 1 : 64];
#pragma unroll
    for (int i = 0; i < (MF == 0 ? 16 : 1); ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
#pragma unroll
    for (int i = 0; i < (MF == 0 ? 1 : 64); ++i) acc4[i] = (f32x4){0, 0, 0, 0};
    const unsigned lds_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem + (wave & 1) * 16384 + lane * 16;
    const unsigned voff = (unsigned)(lane * 16);
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
    for (int ks = 0; ks < ksteps; ++ks) {
        const unsigned stage = (ks & 1) * 65536;
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            constexpr int NM = MF == 0 ? 32 : 64;  // MFMAs per half
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                if (MF == 0) {
                    const int kc = m >> 4, i = (m >> 2) & 3, j = m & 3;
                    acc[4 * i + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k2][kc * 4 + i], b[k2][kc * 4 + j], acc[4 * i + j], 0, 0, 0);
                } else {
                    const int i = m >> 3, j = m & 7;   // 8 x 8 blocks of 16 x 16
                    acc4[8 * i + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k2][i], b[k2][j], acc4[8 * i + j], 0, 0, 0);
                }
                constexpr int RS = NM / 16;  // one fragment read every RS MFMAs: the other half's 16 fragments
                if (READS && m % RS == 0) {
                    const int q = m / RS;
                    if (q < 8) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[k2 ^ 1][q]) : "v"(lds_addr + stage), "n"(q * 1024 + (k2 ^ 1) * 8192));
                    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[k2 ^ 1][q - 8]) : "v"(lds_addr + stage), "n"(32768 + (q - 8) * 1024 + (k2 ^ 1) * 8192));
                }
                constexpr int DS = NM / 8;   // 8 pieces per half = 16 per K-step and wave
                if (SELF && m % DS == DS / 2) {
                    const int q = m / DS, p = ks * 16 + k2 * 8 + q;
                    lds_ptr_t dst = (lds_ptr_t)(smem + (stage ^ 65536) + wave * 16384 + (k2 * 8 + q) * 1024);
                    if (p & 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_big, dst, 16, voff, (unsigned)(((agroup * 4099u + p * 4u + wave) & 0xFFFFF) * 1024u), 0, 0);
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_small, dst, 16, voff, (unsigned)(((p * 4u + wave) & 2047) * 1024u), 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (READS) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
        }
        if (SELF) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
    float s = 0;
#pragma unroll
    for (int i = 0; i < (MF == 0 ? 16 : 1); ++i) s += acc[i][0] + acc[i][7] + acc[i][15];
#pragma unroll
    for (int i = 0; i < (MF == 0 ? 1 : 64); ++i) s += acc4[i][0] + acc4[i][3];
    if (s == 12345.678f) sink[tid] = s;
    if (threadIdx.x == 0) { out[blockIdx.x * 4 + 0] = c1 - c0; out[blockIdx.x * 4 + 1] = r1 - r0; }
}

template <int MODE, int MF>
static void run(const char *name, int ksteps, const float4 *big, const float4 *small_) {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    float *sink; unsigned long long *out;
    CHECK(hipMalloc(&sink, (size_t)cus * 256 * 4)); CHECK(hipMalloc(&out, cus * 32)); CHECK(hipMemset(out, 0, cus * 32));
    auto kfn = probe_kernel<MODE, MF>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int r = 0; r < 4; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kfn, dim3(cus), dim3(256), kLds, 0, ksteps, big, small_, sink, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> h(cus * 4);
    CHECK(hipMemcpy(h.data(), out, cus * 32, hipMemcpyDeviceToHost));
    std::vector<double> cyc, clk;
    for (int i = 0; i < cus; ++i) { cyc.push_back((double)h[4 * i] / ksteps); clk.push_back((double)h[4 * i] / (double)h[4 * i + 1] * 0.1); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double flops = (double)cus * 256.0 * 256.0 * 64.0 * 2.0 * ksteps;
    printf("{\"mode\": %d, \"mfma\": \"%s\", \"what\": \"%s\", \"ksteps\": %d, \"wave0_cycles_per_kstep_median\": %.1f, "
           "\"in_kernel_clock_ghz_median_wave0\": %.3f, \"kernel_ms\": %.3f, \"tflops\": %.1f}\n",
           MODE, MF == 0 ? "32x32x16" : "16x16x32", name, ksteps, cyc[cus / 2], clk[cus / 2], ms, flops / ms / 1e9);
    fflush(stdout);
    CHECK(hipFree(sink)); CHECK(hipFree(out));
}

int main() {
    float4 *big, *small_;
    const size_t big_bytes = (size_t)1 << 30, small_bytes = (size_t)2 << 20;
    CHECK(hipMalloc(&big, big_bytes + (1 << 20))); CHECK(hipMalloc(&small_, small_bytes + (1 << 20)));
    std::vector<unsigned> rndv((1 << 20) / 4);
    for (size_t i = 0; i < rndv.size(); ++i) rndv[i] = (unsigned)(i * 2654435761u) & 0xBF7FBF7Fu;
    for (size_t off = 0; off < big_bytes; off += (1 << 20)) CHECK(hipMemcpy((char *)big + off, rndv.data(), 1 << 20, hipMemcpyHostToDevice));
    for (size_t off = 0; off < small_bytes; off += (1 << 20)) CHECK(hipMemcpy((char *)small_ + off, rndv.data(), 1 << 20, hipMemcpyHostToDevice));
    const int ksteps = 6000;
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0>("4 waves of 128 x 128: mfma only", ksteps, big, small_);
        run<1, 0>("+ fragment reads (32 per wave and K-step)", ksteps, big, small_);
        run<2, 0>("+ fragment reads + LDS-DMA issued by the same waves (16 pieces each)", ksteps, big, small_);
        run<0, 1>("4 waves of 128 x 128: mfma only", ksteps, big, small_);
        run<1, 1>("+ fragment reads (32 per wave and K-step)", ksteps, big, small_);
        run<2, 1>("+ fragment reads + LDS-DMA issued by the same waves (16 pieces each)", ksteps, big, small_);
    }
    return 0;
}
