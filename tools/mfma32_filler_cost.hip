// What a non-matrix instruction costs a wave that otherwise issues back-to-back fp32 MFMAs (v_mfma_f32_32x32x2_f32, 64 cycles
// each, 16 independent accumulators = the whole 256-register accumulator file, ONE wave per SIMD: the shape of m360's fp32
// linear kernels).  The loop body is 16 MFMAs; N fillers of one kind go into the gap after MFMA 0 (kinds "x1": all in that
// one gap) or one per gap ("spread").  Prints cycles per 16-MFMA body (ideal 1024) and the cost per filler.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma32_filler_cost.hip -o tools/mfma32_filler_cost.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { NONE, VALU_MAX_1GAP, VALU_MAX_SPREAD, VALU_PKADD_1GAP, STORE_SADDR, STORE_VADDR, LDS_DMA, DS_WRITE_B32_AGPR, DS_READ_B128,
       VALU_MAX_LATE, DS_ADD_F32, SALU_BRANCH, VALU_MAX_2GAPS, EPI_1GAP, EPI_4GAPS, EPI_1GAP_LDS, EPI_4GAPS_LDS };

template <int KIND, int N>
__global__ __launch_bounds__(256, 1) void loop_kernel(int iters, const float4 *__restrict__ src, float *dst, float *sink, unsigned long long *cyc) {
    __shared__ __attribute__((aligned(1024))) char smem[96 * 1024];  // one workgroup per CU
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = (float)((tid * 7 + i) % 13) * 0.01f; b[i] = (float)((tid * 3 + i) % 11) * 0.01f; }
    f32x16 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(src), 0, 0x7fffffff, 0x00020000);
    const unsigned voff = (unsigned)(lane * 16 + wave * 1024);
    const unsigned lds_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem + wave * 8192 + lane * 16;
    float *wdst = dst + (size_t)(blockIdx.x * 4 + wave) * 16384;  // 64 KiB per wave, rewritten: stays in L2
    float *vdst = wdst + lane * 4;
    constexpr int NV = N > 0 ? N : 1;
    f32x4 d[4];
    float v[24];
    f32x2 p[8], q[8];
#pragma unroll
    for (int f = 0; f < 4; ++f) d[f] = (f32x4){1.0f * lane, 2, 3, 4};
#pragma unroll
    for (int f = 0; f < 24; ++f) v[f] = lane - 31.5f + f;
#pragma unroll
    for (int f = 0; f < 8; ++f) { p[f] = (f32x2){1.0f * lane, 2.0f + f}; q[f] = (f32x2){0.5f, 0.25f}; }
    float zero_v;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero_v));
    f32x4 bqv = {0.25f * lane, 1, 2, 3};
    asm volatile("" : "+v"(bqv));
    unsigned long long c0, c1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0)::"memory");
    for (int it = 0; it < iters; ++it) {
        const int off = (it & 15) * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int slot = 4 * i + j;
                acc[slot] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[slot], 0, 0, 0);
                if (KIND == VALU_MAX_1GAP && slot == 0) {
#pragma unroll
                    for (int f = 0; f < NV; ++f) asm volatile("v_max_f32 %0, %1, %0" : "+v"(v[f]) : "v"(zero_v));
                }
                if (KIND == VALU_MAX_2GAPS && (slot == 0 || slot == 1)) {
#pragma unroll
                    for (int f = 0; f < NV / 2; ++f) asm volatile("v_max_f32 %0, %1, %0" : "+v"(v[f + slot * (NV / 2)]) : "v"(zero_v));
                }
                if (KIND == VALU_MAX_LATE && slot == 0) {  // the same after 40 idle cycles of the gap
                    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7");
#pragma unroll
                    for (int f = 0; f < NV; ++f) asm volatile("v_max_f32 %0, %1, %0" : "+v"(v[f]) : "v"(zero_v));
                }
                if (KIND == VALU_MAX_SPREAD && slot < NV) asm volatile("v_max_f32 %0, %1, %0" : "+v"(v[slot]) : "v"(zero_v));
                if (KIND == VALU_PKADD_1GAP && slot == 0) {
#pragma unroll
                    for (int f = 0; f < NV; ++f) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[f & 7]) : "v"(q[f & 7]));
                }
                if (KIND == STORE_SADDR && slot < NV) asm volatile("global_store_dwordx4 %0, %1, %2 offset:0" ::"v"(voff + off), "v"(d[slot & 3]), "s"(wdst) : "memory");
                if (KIND == STORE_VADDR && slot < NV) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(vdst + off / 4), "v"(d[slot & 3]) : "memory");
                if (KIND == LDS_DMA && slot < NV) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + wave * 8192 + (slot & 3) * 1024), 16, voff, ((it * 16 + slot) & 31) * 8192, 0, 0);
                if (KIND == DS_WRITE_B32_AGPR && slot < NV) asm volatile("ds_write_b32 %0, %1" ::"v"(lds_addr), "a"(acc[15][slot & 15]) : "memory");
                if (KIND == DS_ADD_F32 && slot < NV) asm volatile("ds_add_f32 %0, %1" ::"v"(lds_addr), "v"(zero_v) : "memory");
                if (KIND == DS_READ_B128 && slot < NV) asm volatile("ds_read_b128 %0, %1" : "=v"(d[slot & 3]) : "v"(lds_addr + (slot & 3) * 1024));
                // the epilogue arithmetic of m360_linear_hd.hip.h: per 16 bytes 2 packed bias adds + 4 ReLU max (N = blocks of 4 x 16 B)
                if ((KIND == EPI_1GAP_LDS || KIND == EPI_4GAPS_LDS) && slot >= 2 && slot < 6) asm volatile("ds_read_b128 %0, %1" : "=v"(d[slot - 2]) : "v"(lds_addr + (slot - 2) * 1024));
                if ((KIND == EPI_1GAP_LDS || KIND == EPI_4GAPS_LDS) && slot == 7) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
                if (((KIND == EPI_1GAP || KIND == EPI_1GAP_LDS) && slot == 7) || ((KIND == EPI_4GAPS || KIND == EPI_4GAPS_LDS) && slot >= 7 && slot < 11)) {
#pragma unroll
                    for (int rep = 0; rep < NV; ++rep)
#pragma unroll
                        for (int pp = 0; pp < 4; ++pp) {
                            if ((KIND == EPI_4GAPS || KIND == EPI_4GAPS_LDS) && pp != slot - 7) continue;
                            d[pp] += bqv;
                            d[pp][0] = fmaxf(d[pp][0], 0.0f); d[pp][1] = fmaxf(d[pp][1], 0.0f); d[pp][2] = fmaxf(d[pp][2], 0.0f); d[pp][3] = fmaxf(d[pp][3], 0.0f);
                        }
                    asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
                }
                if (KIND == SALU_BRANCH && slot < NV) { if (iters == 12345 + slot) asm volatile("s_nop 0"); }
                __builtin_amdgcn_sched_barrier(0);
            }
        if (KIND == STORE_SADDR || KIND == STORE_VADDR || KIND == LDS_DMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NV) : "memory");
        if (KIND == DS_WRITE_B32_AGPR || KIND == DS_READ_B128 || KIND == DS_ADD_F32 || KIND == EPI_1GAP_LDS || KIND == EPI_4GAPS_LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1)::"memory");
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][15];
#pragma unroll
    for (int f = 0; f < 24; ++f) s += v[f];
#pragma unroll
    for (int f = 0; f < 8; ++f) s += p[f][0] + p[f][1];
#pragma unroll
    for (int f = 0; f < 4; ++f) s += d[f][0];
    if (s == 12345.678f) sink[tid] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = c1 - c0;
}

template <int KIND, int N>
static double run() {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int iters = 2000;
    float4 *src; float *dst, *sink; unsigned long long *cyc;
    CHECK(hipMalloc(&src, 64 << 20)); CHECK(hipMemset(src, 0x11, 64 << 20));
    CHECK(hipMalloc(&dst, (size_t)cus * 4 * 65536 * 2)); CHECK(hipMalloc(&sink, (size_t)cus * 256 * 4)); CHECK(hipMalloc(&cyc, cus * 8));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((loop_kernel<KIND, N>), dim3(cus), dim3(256), 0, 0, iters, src, dst, sink, cyc);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(cus);
    CHECK(hipMemcpy(h.data(), cyc, cus * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    CHECK(hipFree(src)); CHECK(hipFree(dst)); CHECK(hipFree(sink)); CHECK(hipFree(cyc));
    return (double)h[cus / 2] / iters;
}

int main() {
    const double base = run<NONE, 0>();
    printf("{\"kind\": \"none\", \"cycles_per_16_mfma\": %.1f, \"ideal\": 1024}\n", base);
#define ROW(K, N) { const double c = run<K, N>(); printf("{\"kind\": \"%s\", \"fillers_per_16_mfma\": %d, \"cycles_per_16_mfma\": %.1f, \"cost_per_filler_cycles\": %.1f}\n", #K, N, c, (c - base) / N); fflush(stdout); }
    ROW(VALU_MAX_1GAP, 1) ROW(VALU_MAX_1GAP, 2) ROW(VALU_MAX_1GAP, 4) ROW(VALU_MAX_1GAP, 6) ROW(VALU_MAX_1GAP, 8) ROW(VALU_MAX_1GAP, 12) ROW(VALU_MAX_1GAP, 16) ROW(VALU_MAX_1GAP, 24)
    ROW(VALU_MAX_2GAPS, 8) ROW(VALU_MAX_2GAPS, 16) ROW(VALU_MAX_2GAPS, 24)
    ROW(VALU_MAX_LATE, 1) ROW(VALU_MAX_LATE, 4) ROW(VALU_MAX_LATE, 8)
    ROW(VALU_MAX_SPREAD, 4) ROW(VALU_MAX_SPREAD, 8) ROW(VALU_MAX_SPREAD, 16)
    ROW(VALU_PKADD_1GAP, 2) ROW(VALU_PKADD_1GAP, 4) ROW(VALU_PKADD_1GAP, 8)
    ROW(STORE_SADDR, 1) ROW(STORE_SADDR, 2) ROW(STORE_SADDR, 4)
    ROW(STORE_VADDR, 1) ROW(STORE_VADDR, 2) ROW(STORE_VADDR, 4)
    ROW(LDS_DMA, 1) ROW(LDS_DMA, 2) ROW(LDS_DMA, 4)
    ROW(DS_WRITE_B32_AGPR, 4) ROW(DS_WRITE_B32_AGPR, 16)
    ROW(DS_ADD_F32, 1) ROW(DS_ADD_F32, 4)
    ROW(DS_READ_B128, 4) ROW(DS_READ_B128, 8)
    ROW(SALU_BRANCH, 16)
    ROW(EPI_1GAP, 1) ROW(EPI_4GAPS, 1) ROW(EPI_1GAP_LDS, 1) ROW(EPI_4GAPS_LDS, 1)
    return 0;
}
