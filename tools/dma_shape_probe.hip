// How fast does the L2 -> LDS path (buffer_load ... lds, 16 bytes per lane) deliver the operand stream of a 1024 x 1024 bf16 layer,
// and does the SHAPE of a 1-KiB piece matter?  The one-wave ring kernel (m360_linear_bf16_w16.hip.h) stages 32-deep slabs:
// a piece is 16 rows x 64 B - HALF of sixteen 128-byte lines, whose other halves are fetched one slab later.  Its ablations put the
// kernel at the delivery rate of that stream (32 KiB per slab and CU in 0.73-0.78 us = 42-45 GB/s per CU, with or without the
// fragment reads).  This probe streams exactly the layer's operands (activation tile rows shared by the 4 column tiles of an XCD,
// weights L2-resident), no matrix work, a 4-deep ring of counted waits, in two piece shapes:
//   shape 0: 16 rows x 64 B  per piece (32-deep slabs: 512 half lines per slab and CU)
//   shape 1:  8 rows x 128 B per piece (64-deep stages: the same bytes as whole lines)
//   shape 2:  4 rows x 256 B per piece
// Reported: GB/s per CU and in total.   hipcc -O3 --offload-arch=gfx950 tools/dma_shape_probe.hip -o tools/dma_shape_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(3))) void *lds_ptr_t;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kK = 1024, kN = 1024;        // layer
constexpr long kM = 524288;
constexpr int kRowBytes = kK * 2;          // bf16 rows

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void stream_kernel(const char *__restrict__ X, const char *__restrict__ W, int ntiles, unsigned long long *out) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];  // 128 KiB ring: 4 x 32 KiB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int ROWS = SHAPE == 0 ? 16 : SHAPE == 1 ? 8 : 4;     // rows per piece
    constexpr int SEG = 1024 / ROWS;                                // bytes per row segment
    constexpr int LPR = SEG / 16;                                   // lanes per row
    // a "stage" = SEG bytes of k for all 256 + 256 rows: SEG * 512 bytes; the ring holds 128 KiB / that many stages
    constexpr int kStageBytes = SEG * 512;
    constexpr int kStages = 131072 / kStageBytes;                   // 4, 2, 1
    constexpr int kPiecesPerWave = kStageBytes / 1024 / 4;          // 8, 16, 32 (half X, half W)
    const unsigned voff = (unsigned)((lane / LPR) * kRowBytes + (lane % LPR) * 16);
    unsigned long long t0 = 0, t1 = 0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    const int G = gridDim.x;
    long issued = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += G) {
        const int full = (ntiles / 8) * 8;
        int lin = tile;
        if (tile < full) lin = (tile % 8) * (full / 8) + tile / 8;
        const long m0 = (long)(lin / 4) * 256;
        const int n0 = (lin % 4) * 256;
        __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(X + m0 * kRowBytes), 0, 0x7fffffff, 0x00020000);
        __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(W + (long)n0 * kRowBytes), 0, 0x7fffffff, 0x00020000);
        for (int kb = 0; kb < kRowBytes; kb += SEG) {
            const int slot = (int)(issued % kStages);
            char *base = smem + slot * kStageBytes + wave * (kStageBytes / 4);
#pragma unroll
            for (int q = 0; q < kPiecesPerWave / 2; ++q) {
                // this wave's rows [64 wave, 64 wave + 64) of both operands, ROWS at a time
                const unsigned roff = (unsigned)((64 * wave + ROWS * q) * kRowBytes);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(base + q * 1024), 16, voff + roff, kb, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(base + (kPiecesPerWave / 2 + q) * 1024), 16, voff + roff, kb, 0, 0);
            }
            ++issued;
            // three stages of 32 KiB (or their equivalent in bytes) stay in flight behind the one just issued: 24 pieces per wave
            asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

// Second question: with whole-line pieces a 64-deep stage is 64 KiB and a 128-KiB ring holds two of them, so only ~64 KiB per CU can
// be in flight - is that enough?  INFLIGHT = pieces per wave that may stay outstanding behind the stage just issued (16 pieces = one
// 64-KiB stage per CU).  Measured: 32 KiB in flight already sustain 64 GB/s per CU (latency ~0.5 us), so the early 4-byte "touch"
// loads (TOUCH stages ahead, one instruction per wave and stage) this kernel can also issue are not needed and not run.
template <int INFLIGHT, int TOUCH>
__global__ __launch_bounds__(256, 1) void stream2_kernel(const char *__restrict__ X, const char *__restrict__ W, int ntiles, unsigned long long *out) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned voff = (unsigned)((lane / 8) * kRowBytes + (lane % 8) * 16);
    const unsigned toff = (unsigned)((64 * wave + lane) * kRowBytes);  // touch: one lane per row of this wave's 64 activation rows
    unsigned long long t0 = 0, t1 = 0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    const int G = gridDim.x;
    long issued = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += G) {
        const int full = (ntiles / 8) * 8;
        int lin = tile;
        if (tile < full) lin = (tile % 8) * (full / 8) + tile / 8;
        const long m0 = (long)(lin / 4) * 256;
        const int n0 = (lin % 4) * 256;
        __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(X + m0 * kRowBytes), 0, 0x7fffffff, 0x00020000);
        __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(W + (long)n0 * kRowBytes), 0, 0x7fffffff, 0x00020000);
        for (int kb = 0; kb < kRowBytes; kb += 128) {
            char *base = smem + (int)(issued & 1) * 65536 + wave * 16384;
            if (TOUCH && kb + TOUCH * 128 < kRowBytes) {  // the lines of stage kb + TOUCH * 128 of this tile
                unsigned tmp;
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(tmp) : "v"(toff), "s"(rx), "s"(kb + TOUCH * 128) : "memory");
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const unsigned roff = (unsigned)((64 * wave + 8 * q) * kRowBytes);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(base + q * 1024), 16, voff + roff, kb, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(base + (8 + q) * 1024), 16, voff + roff, kb, 0, 0);
            }
            ++issued;
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT + (TOUCH ? INFLIGHT / 16 : 0)) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int INFLIGHT, int TOUCH>
static void run2(const char *X, const char *W) {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    unsigned long long *out;
    CHECK(hipMalloc(&out, cus * 8));
    auto kfn = stream2_kernel<INFLIGHT, TOUCH>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    const int ntiles = (int)(kM / 256) * (kN / 256);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0, best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kfn, dim3(cus), dim3(256), 131072, 0, X, W, ntiles, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    const double bytes = (double)ntiles * 2.0 * 256 * kRowBytes;
    printf("{\"piece\": \"8 rows x 128 B\", \"KiB_in_flight_per_CU\": %d, \"touch_stages_ahead\": %d, \"kernel_ms\": %.4f, \"GBps_per_CU\": %.1f, "
           "\"TBps_total\": %.2f}\n", INFLIGHT * 4, TOUCH, best, bytes / best / 1e6 / cus, bytes / best / 1e9);
    fflush(stdout);
    CHECK(hipFree(out));
}

template <int SHAPE>
static void run(const char *X, const char *W) {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    unsigned long long *out;
    CHECK(hipMalloc(&out, cus * 8));
    auto kfn = stream_kernel<SHAPE>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    const int ntiles = (int)(kM / 256) * (kN / 256);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0, best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kfn, dim3(cus), dim3(256), 131072, 0, X, W, ntiles, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    const double bytes = (double)ntiles * 2.0 * 256 * kRowBytes;  // X tile + W tile per output tile
    const char *names[3] = {"16 rows x 64 B", "8 rows x 128 B", "4 rows x 256 B"};
    printf("{\"piece\": \"%s\", \"kernel_ms\": %.4f, \"staged_GB\": %.3f, \"GBps_per_CU\": %.1f, \"TBps_total\": %.2f, "
           "\"layer_time_bound_ms_at_this_rate\": %.4f}\n",
           names[SHAPE], best, bytes / 1e9, bytes / best / 1e6 / cus, bytes / best / 1e9, best);
    fflush(stdout);
    CHECK(hipFree(out));
}

int main() {
    char *X, *W;
    CHECK(hipMalloc(&X, (size_t)kM * kRowBytes));
    CHECK(hipMalloc(&W, (size_t)kN * kRowBytes));
    CHECK(hipMemset(X, 0x11, (size_t)kM * kRowBytes));
    CHECK(hipMemset(W, 0x22, (size_t)kN * kRowBytes));
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(X, W);
        run<1>(X, W);
        run<2>(X, W);
    }
    for (int rep = 0; rep < 2; ++rep) {
        run2<8, 0>(X, W); run2<16, 0>(X, W); run2<24, 0>(X, W); run2<32, 0>(X, W);
    }
    return 0;
}
