// Does a dedicated LOADER wave disturb the MFMA waves it shares a SIMD with?  (producer / consumer wave
// specialisation for the bf16 layer kernel.)  One 768-thread workgroup per CU = 12 waves = 3 per SIMD: waves 0-7 run the
// matrix stream of the 256 x 256 x 64 K-step of the product kernel (64 x v_mfma_f32_16x16x32_bf16 per wave and K-step,
// optionally with the 24 ds_read_b128 of its fragments), waves 8-11 (one per SIMD) issue the K-step's 64 LDS-DMA pieces
// (16 each; half of them streamed from a 1 GiB buffer like the activations, half from a 2 MiB one like the weights), bounded
// by a counted vmcnt.  No handshake, no epilogue: the probe prices ISSUE interference, LDS port sharing and the clock the
// chip holds - an upper bound for any kernel of this structure.  Reported per mode: kernel time and its TFLOP/s equivalent
// (THE result), the in-kernel clock (s_memtime / s_memrealtime) and cycles per K-step of wave 0 (the oldest wave wins the
// SIMD's arbitration, so wave 0 alone finishes early in the MFMA-only modes: read the kernel time, not this).
//   modes: 0 MFMA only | 1 + fragment reads | 2 + loader waves (no reads) | 3 + reads + loader waves
//          4 = mode 3 but the 8 MFMA waves issue the DMA themselves (8 pieces each per K-step, the ping-pong kernel's split)
//   hipcc -O3 --offload-arch=gfx950 tools/loader_wave_probe.hip -o tools/loader_wave_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ float rnd(unsigned s) { return (hash32(s) >> 8) * (1.0f / 8388608.0f) - 1.0f; }

constexpr int kLds = 128 * 1024;  // two K-step stages of 64 KiB, as in the product kernel

template <int MODE, int SHARE>
__global__ __launch_bounds__(768) void probe_kernel(int ksteps, const float4 *__restrict__ big, const float4 *__restrict__ small_, float *sink,
                                                     unsigned long long *out) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr bool READS = (MODE == 1 || MODE == 3 || MODE == 4), LOADERS = (MODE == 2 || MODE == 3), SELF = (MODE == 4);
    // random bf16 bits into LDS (fragments read from it must be random: the clock depends on the operand bits)
    for (int i = threadIdx.x; i < kLds / 4; i += blockDim.x) reinterpret_cast<unsigned *>(smem)[i] = hash32(tid * 977 + i) & 0xBF7FBF7Fu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs_big = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(big), 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_small = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(small_), 0, 0x7fffffff, 0x00020000);
    unsigned long long c0 = 0, c1 = 0, r0 = 0, r1 = 0;
    // the activation tile of a row block is shared by the 4 column tiles of a 1024-wide layer, which the product kernel places on ONE
    // XCD (block ids equal mod 8): SHARE = 1 streams each activation piece through that XCD's L2 once for 4 workgroups, SHARE = 0
    // gives every workgroup its own stream (4 x the HBM bytes of the real layer)
    const unsigned agroup = SHARE ? (blockIdx.x % 8u) + 8u * ((blockIdx.x / 8u) / 4u) : blockIdx.x;
    if (wave < 8) {
        bf16x8 a[2], b[4];  // 3 waves per SIMD leave 168 registers: 128 accumulators + B fragments + a double-buffered A fragment
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)rnd(tid * 128 + i * 8 + e);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) b[i][e] = (__bf16)rnd(tid * 128 + 64 + i * 8 + e);
        f32x4 acc[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        const unsigned lds_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem + (wave & 3) * 4096 + lane * 16;
        const unsigned voff = (unsigned)(lane * 16);
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
        for (int ks = 0; ks < ksteps; ++ks) {
            const unsigned stage = (ks & 1) * 65536;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {  // two 32-deep halves of the 64-deep K-step: 32 MFMAs each
                if (READS) {  // 4 B fragments + the first A fragment of this half
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[i]) : "v"(lds_addr + stage), "n"(32768 + i * 1024 + k2 * 4096));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[0]) : "v"(lds_addr + stage), "n"(k2 * 8192));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (READS && i < 7) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[(i + 1) & 1]) : "v"(lds_addr + stage), "n"((i + 1) * 1024 + k2 * 8192));
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[4 * i + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 1], b[j], acc[4 * i + j], 0, 0, 0);
                    if (READS && i < 7) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
                    if (SELF && (i & 1) == 0) {  // 4 pieces per half = 8 per K-step per wave, spread over the MFMAs
                        const int p = ks * 8 + k2 * 4 + (i >> 1);
                        lds_ptr_t dst = (lds_ptr_t)(smem + (stage ^ 65536) + wave * 8192 + (k2 * 4 + (i >> 1)) * 1024);
                        if (p & 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_big, dst, 16, voff, (unsigned)(((agroup * 4099u + p * 8u + wave) & 0xFFFFF) * 1024u), 0, 0);
                        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_small, dst, 16, voff, (unsigned)(((p * 8u + wave) & 2047) * 1024u), 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (SELF) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
        float s = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
        if (s == 12345.678f) sink[tid] = s;
        if (threadIdx.x == 0) { out[blockIdx.x * 4 + 0] = c1 - c0; out[blockIdx.x * 4 + 1] = r1 - r0; }
    } else if (LOADERS) {
        const int lw = wave - 8;
        const unsigned voff = (unsigned)(lane * 16);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0)::"memory");
        for (int ks = 0; ks < ksteps; ++ks) {
            const unsigned stage = ((ks + 1) & 1) * 65536;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const unsigned p = (unsigned)ks * 64u + lw * 16u + q;
                lds_ptr_t dst = (lds_ptr_t)(smem + stage + lw * 16384 + q * 1024);
                if (q & 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_big, dst, 16, voff, (unsigned)(((agroup * 4099u + p) & 0xFFFFF) * 1024u), 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_small, dst, 16, voff, (unsigned)((p & 2047) * 1024u), 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // one K-step in flight behind the one being issued
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1)::"memory");
        if (threadIdx.x == 512) out[blockIdx.x * 4 + 2] = c1 - c0;
    }
}

template <int MODE, int SHARE>
static void run(const char *name, int ksteps, const float4 *big, const float4 *small_) {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    float *sink; unsigned long long *out;
    CHECK(hipMalloc(&sink, (size_t)cus * 768 * 4)); CHECK(hipMalloc(&out, cus * 32)); CHECK(hipMemset(out, 0, cus * 32));
    { auto kfn = probe_kernel<MODE, SHARE>; CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, kLds)); }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int r = 0; r < 4; ++r) {  // the last of four back-to-back launches is reported (clock settled)
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe_kernel<MODE, SHARE>), dim3(cus), dim3(768), kLds, 0, ksteps, big, small_, sink, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> h(cus * 4);
    CHECK(hipMemcpy(h.data(), out, cus * 32, hipMemcpyDeviceToHost));
    std::vector<double> cyc, clk, ld;
    for (int i = 0; i < cus; ++i) { cyc.push_back((double)h[4 * i] / ksteps); clk.push_back((double)h[4 * i] / (double)h[4 * i + 1] * 0.1); ld.push_back((double)h[4 * i + 2] / ksteps); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end()); std::sort(ld.begin(), ld.end());
    const double flops = (double)cus * 8 * 64 * 16384.0 * ksteps;  // 8 waves x 64 MFMAs x 16*16*32*2
    printf("{\"mode\": %d, \"activation_stream_shared_by_4_workgroups\": %d, \"what\": \"%s\", \"ksteps\": %d, \"wave0_cycles_per_kstep_median\": %.1f, "
           "\"in_kernel_clock_ghz_median_wave0\": %.3f, \"loader_cycles_per_kstep_median\": %.1f, \"kernel_ms\": %.3f, \"tflops\": %.1f}\n",
           MODE, SHARE, name, ksteps, cyc[cus / 2], clk[cus / 2], ld[cus / 2], ms, flops / ms / 1e9);
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
    const int ksteps = 6000;  // ~ 6 ms at full rate
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 1>("mfma only (12 waves resident, 4 idle)", ksteps, big, small_);
        run<1, 1>("mfma + fragment reads", ksteps, big, small_);
        run<2, 1>("mfma + loader waves", ksteps, big, small_);
        run<3, 1>("mfma + fragment reads + loader waves", ksteps, big, small_);
        run<4, 1>("mfma + fragment reads + DMA issued by the mfma waves themselves", ksteps, big, small_);
        run<3, 0>("mfma + fragment reads + loader waves", ksteps, big, small_);
        run<4, 0>("mfma + fragment reads + DMA issued by the mfma waves themselves", ksteps, big, small_);
    }
    return 0;
}
