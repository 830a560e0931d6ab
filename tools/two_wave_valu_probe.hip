// Can a SECOND wave on the same SIMD run vector work in the shadow of a wave that issues back-to-back bf16 MFMAs?
// One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run the matrix loop (v_mfma_f32_16x16x32_bf16, 16 independent accumulators),
// waves 4-7 (their SIMD partners) run MODE: 0 exit at once | 1 a stream of independent v_pk_fma_f32 / v_cvt_pk_bf16_f32 / v_pk_max_i16
// (the ring kernel's epilogue mix) | 2 the same + one 16-byte global store per 24 vector instructions.
// Prints the matrix waves' cycles per 16-MFMA group (256 cycles of pure issue) and the vector instructions per 16-MFMA group the
// partner got through.  (The ring kernel itself cannot host such a partner: it needs 384 of the SIMD's 512 registers per wave and all
// waves of a kernel get the same allocation - this probe prices the structure)
//   hipcc -O3 --offload-arch=gfx950 tools/two_wave_valu_probe.hip -o tools/two_wave_valu_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ float rnd(unsigned s) { return (hash32(s) >> 8) * (1.0f / 8388608.0f) - 1.0f; }

template <int MODE>
__global__ __launch_bounds__(512) void probe_kernel(int iters, float *sink, unsigned *out16, unsigned long long *cyc, unsigned long long *valu_done) {
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ volatile int stop;
    if (threadIdx.x == 0) stop = 0;
    __syncthreads();
    if (wave < 4) {  // matrix waves
        bf16x8 a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) { a[i][e] = (__bf16)rnd(tid * 64 + i * 8 + e); b[i][e] = (__bf16)rnd(tid * 64 + 32 + i * 8 + e); }
        f32x4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        unsigned long long c0, c1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0)::"memory");
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[4 * i + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[4 * i + j], 0, 0, 0);
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1)::"memory");
        if ((threadIdx.x & 63) == 0) {
            cyc[blockIdx.x * 4 + wave] = c1 - c0;
            stop = 1;
        }
        float s = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
        if (s == 123.456f) sink[tid] = s;
    } else {         // partner waves
        if (MODE == 0) return;
        f32x2 v[8], bias = {0.25f, -0.5f};
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (f32x2){rnd(tid * 16 + i), rnd(tid * 16 + 8 + i)};
        unsigned long long n = 0;
        unsigned packed = 0;
        while (!stop) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {  // 3 vector instructions per pair, as the epilogue's bias add / conversion / ReLU
                v[i] = __builtin_elementwise_fma(v[i], (f32x2){0.999f, 1.001f}, bias);
                const bf16x2 h = __builtin_convertvector(v[i], bf16x2);
                s16x2 p = __builtin_bit_cast(s16x2, h);
                p = __builtin_elementwise_max(p, (s16x2){0, 0});
                packed ^= __builtin_bit_cast(unsigned, p);
            }
            n += 24;
            if (MODE == 2) out16[(size_t)tid * 4 + (n & 3)] = packed;
        }
        if ((threadIdx.x & 63) == 0) valu_done[blockIdx.x * 4 + (wave - 4)] = n;
        if (packed == 0x12345678u) sink[tid] = v[0][0];
    }
}

int main() {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int iters = 20000;
    float *sink; unsigned *out16; unsigned long long *cyc, *done;
    CHECK(hipMalloc(&sink, (size_t)cus * 512 * 4)); CHECK(hipMalloc(&out16, (size_t)cus * 512 * 16));
    CHECK(hipMalloc(&cyc, (size_t)cus * 4 * 8)); CHECK(hipMalloc(&done, (size_t)cus * 4 * 8));
    for (int mode = 0; mode < 3; ++mode) {
        CHECK(hipMemset(done, 0, (size_t)cus * 4 * 8));
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(probe_kernel<0>, dim3(cus), dim3(512), 0, 0, iters, sink, out16, cyc, done);
            if (mode == 1) hipLaunchKernelGGL(probe_kernel<1>, dim3(cus), dim3(512), 0, 0, iters, sink, out16, cyc, done);
            if (mode == 2) hipLaunchKernelGGL(probe_kernel<2>, dim3(cus), dim3(512), 0, 0, iters, sink, out16, cyc, done);
            CHECK(hipDeviceSynchronize());
        }
        unsigned long long *h = (unsigned long long *)malloc((size_t)cus * 4 * 8), *hd = (unsigned long long *)malloc((size_t)cus * 4 * 8);
        CHECK(hipMemcpy(h, cyc, (size_t)cus * 4 * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hd, done, (size_t)cus * 4 * 8, hipMemcpyDeviceToHost));
        double c = 0, d = 0;
        for (int i = 0; i < cus * 4; ++i) { c += (double)h[i]; d += (double)hd[i]; }
        c /= cus * 4; d /= cus * 4;
        printf("{\"mode\": %d, \"matrix_cycles_per_16_mfma\": %.1f, \"pure_issue\": 256, \"partner_vector_instructions_per_16_mfma\": %.1f}\n", mode, c / iters, d / iters);
        free(h); free(hd);
    }
    return 0;
}
