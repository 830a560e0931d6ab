// How fast can ONE CU store, and is the limit per CU or chip-wide?  The one-wave ring kernel (m360_linear_bf16_w16.hip.h) issues the
// 128 KiB of a tile's output in one burst per CU; the burst and the wait behind it cost ~12 k cycles per tile = 11 B/clk per CU.
// This probe lets G workgroups (one per CU, 256 threads) store 32 MiB each in 1-KiB instructions of two shapes, back to back:
//   shape 0: 16 rows x 64 B per instruction (the ring kernel's: half of sixteen 128-byte lines; the other half follows 1 instruction later)
//   shape 1:  8 rows x 128 B per instruction (whole lines)
// and reports GB/s per CU for G = 1, 8, 64, 256.   hipcc -O3 --offload-arch=gfx950 tools/store_rate_probe.hip -o tools/store_rate_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr long kBytesPerWG = 32l << 20;
constexpr int kRowBytes = 2048;  // output rows of a 1024-wide bf16 layer

template <int SHAPE>
__global__ __launch_bounds__(256) void store_kernel(char *__restrict__ y, unsigned long long *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char *base = y + (long)blockIdx.x * kBytesPerWG;
    // the workgroup's region = 16384 rows x 2048 B; a wave walks 128-row x 256-byte wave tiles like the kernel's epilogue
    u32x4 v = {threadIdx.x, 1u, 2u, blockIdx.x};
    unsigned long long t0, t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (long tile = wave; tile < kBytesPerWG / (128 * 256); tile += 4) {
        char *tb = base + (tile / 8) * 128 * kRowBytes + (tile % 8) * 256;
        if (SHAPE == 0) {
            const int r = lane & 15, c = (lane >> 4) * 16;
#pragma unroll 4
            for (int i = 0; i < 32; ++i) {  // (p = i / 8: 64-byte column piece, ib = i % 8: 16-row block)
                char *p = tb + (long)((i % 8) * 16 + r) * kRowBytes + (i / 8) * 64 + c;
                asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
            }
        } else {
            const int r = lane >> 3, c = (lane & 7) * 16;
#pragma unroll 4
            for (int i = 0; i < 32; ++i) {  // (half = i / 16: 128-byte column piece, rb = i % 16: 8-row block)
                char *p = tb + (long)((i % 16) * 8 + r) * kRowBytes + (i / 16) * 128 + c;
                asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int SHAPE>
static void run(char *y, int G) {
    unsigned long long *out;
    CHECK(hipMalloc(&out, 256 * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0, best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(store_kernel<SHAPE>, dim3(G), dim3(256), 0, 0, y, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    printf("{\"store_instruction\": \"%s\", \"workgroups\": %d, \"kernel_ms\": %.4f, \"GBps_per_CU\": %.1f, \"TBps_total\": %.3f}\n",
           SHAPE == 0 ? "16 rows x 64 B" : "8 rows x 128 B", G, best, kBytesPerWG / best / 1e6, (double)G * kBytesPerWG / best / 1e9);
    fflush(stdout);
    CHECK(hipFree(out));
}

int main() {
    char *y;
    CHECK(hipMalloc(&y, 256 * kBytesPerWG));
    CHECK(hipMemset(y, 0, 256 * kBytesPerWG));
    const int gs[5] = {1, 8, 32, 64, 256};
    for (int gi = 0; gi < 5; ++gi) { run<0>(y, gs[gi]); run<1>(y, gs[gi]); }
    return 0;
}
