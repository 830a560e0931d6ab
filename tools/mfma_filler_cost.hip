// What one non-matrix instruction COSTS a wave that otherwise issues back-to-back bf16 MFMAs (16x16x32, random operands):
// the loop body is 16 MFMAs + F fillers of one kind, every wave of the chip running it, 1 or 2 waves per SIMD.
//   kind 0 none | 1 buffer_load_dwordx4 ... lds (LDS-DMA, 1 KiB) | 2 global_load_dwordx4 -> VGPRs | 3 ds_read_b128 | 4 ds_write_b128
// Prints cycles per 16-MFMA group (s_memtime) and the derived cost per filler = (cycles - cycles(kind 0)) / F.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_filler_cost.hip -o tools/mfma_filler_cost.bin
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

template <int KIND, int F, int BIG = 0>
__global__ __launch_bounds__(512) void loop_kernel(int iters, const float4 *__restrict__ src, float *sink, unsigned long long *cyc) {
    __shared__ __attribute__((aligned(1024))) char smem[64 * 1024];
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[i][e] = (__bf16)rnd(tid * 64 + i * 8 + e); b[i][e] = (__bf16)rnd(tid * 64 + 32 + i * 8 + e); }
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    f32x16 accb[8];  // BIG: 8 MFMAs 32x32x16 (32 cycles each) per body instead of 16 MFMAs 16x16x32 (16 cycles each)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) accb[i][e] = 0;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(src), 0, 0x7fffffff, 0x00020000);
    const unsigned voff = (unsigned)(lane * 16 + wave * 1024);
    const unsigned lds_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem + wave * 4096 + lane * 16;
    float4 g[F > 0 ? F : 1];
    f32x4 d[F > 0 ? F : 1];
#pragma unroll
    for (int f = 0; f < (F > 0 ? F : 1); ++f) { g[f] = make_float4(0, 0, 0, 0); d[f] = (f32x4){0, 0, 0, 0}; }
    unsigned long long c0, c1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int slot = 4 * i + j;
                if (!BIG) acc[slot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[slot], 0, 0, 0);
                else if ((slot & 1) == 0) accb[slot >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], accb[slot >> 1], 0, 0, 0);
                if (F > 0 && slot % (16 / F) == 0) {
                    const int f = slot / (16 / F);
                    const int off = ((it * F + f) & 31) * 8192;  // 256 KiB footprint per CU: L2-resident
                    if (KIND == 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + wave * 4096 + (f & 3) * 1024), 16, voff, off, 0, 0);
                    if (KIND == 2) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(g[f]) : "v"(voff), "s"(rs), "s"(off) : "memory");
                    if (KIND == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(d[f]) : "v"(lds_addr + (f & 3) * 1024));
                    if (KIND == 4) asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr + (f & 3) * 1024), "v"(d[f]) : "memory");
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        if ((it & 3) == 3) {  // keep the queues bounded like a real kernel does (counted waits every few groups)
            if (KIND == 1 || KIND == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (F > 0 ? F : 1)) : "memory");
            if (KIND == 3 || KIND == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1)::"memory");
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += accb[i][0] + accb[i][15];
#pragma unroll
    for (int f = 0; f < (F > 0 ? F : 1); ++f) s += g[f].x + d[f][0];
    if (s == 12345.678f) sink[tid] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = c1 - c0;
}

template <int KIND, int F, int BIG = 0>
static double run(int wps) {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int iters = 4000, threads = 256 * wps;
    float4 *src; float *sink; unsigned long long *cyc;
    CHECK(hipMalloc(&src, 64 << 20)); CHECK(hipMemset(src, 0x11, 64 << 20));
    CHECK(hipMalloc(&sink, (size_t)cus * threads * 4)); CHECK(hipMalloc(&cyc, cus * 8));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((loop_kernel<KIND, F, BIG>), dim3(cus), dim3(threads), 0, 0, iters, src, sink, cyc);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(cus);
    CHECK(hipMemcpy(h.data(), cyc, cus * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    CHECK(hipFree(src)); CHECK(hipFree(sink)); CHECK(hipFree(cyc));
    return (double)h[cus / 2] / iters;
}

int main() {
    const char *names[5] = {"none", "lds_dma_1KiB", "global_load_x4", "ds_read_b128", "ds_write_b128"};
    for (int wps = 1; wps <= 2; ++wps) {
        const double base = run<0, 0>(wps);
        printf("{\"waves_per_simd\": %d, \"kind\": \"none\", \"cycles_per_16_mfma_per_wave\": %.1f}\n", wps, base);
#define ROW(K, F) { const double c = run<K, F>(wps); printf("{\"waves_per_simd\": %d, \"kind\": \"%s\", \"fillers_per_16_mfma\": %d, \"cycles_per_16_mfma_per_wave\": %.1f, \"cost_per_filler_cycles\": %.1f}\n", wps, names[K], F, c, (c - base) / F); }
        ROW(1, 1) ROW(1, 2) ROW(1, 4) ROW(2, 1) ROW(2, 2) ROW(2, 4) ROW(3, 2) ROW(3, 4) ROW(3, 8) ROW(4, 2) ROW(4, 4)
    }
    for (int wps = 1; wps <= 2; ++wps) {  // the same fillers beside 8 x v_mfma_f32_32x32x16_bf16 (same 256 cycles of matrix work per body)
        const double base = run<0, 0, 1>(wps);
        printf("{\"mfma\": \"32x32x16\", \"waves_per_simd\": %d, \"kind\": \"none\", \"cycles_per_body\": %.1f}\n", wps, base);
#define ROWB(K, F) { const double c = run<K, F, 1>(wps); printf("{\"mfma\": \"32x32x16\", \"waves_per_simd\": %d, \"kind\": \"%s\", \"fillers_per_body\": %d, \"cycles_per_body\": %.1f, \"cost_per_filler_cycles\": %.1f}\n", wps, names[K], F, c, (c - base) / F); }
        ROWB(1, 1) ROWB(1, 2) ROWB(1, 4) ROWB(2, 2) ROWB(2, 4) ROWB(3, 4) ROWB(3, 8) ROWB(4, 2) ROWB(4, 4)
    }
    return 0;
}
