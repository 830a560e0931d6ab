// Second probe of the dependency gap: a LONG first kernel (the host is far ahead: the packet behind it is queued long before it ends) followed by a
// dependent second kernel of varying shape, on the null stream and on a non-blocking stream.  Read with tools/diag/launch_gap_probe2.py.
//   pairs (first, second):  0: (small-LDS, tiny 1 workgroup)   1: (small-LDS, small-LDS big grid)   2: (small-LDS, 64 KB-LDS big grid)
//                           3: (160 KB-LDS, small-LDS big grid) 4: (160 KB-LDS, 160 KB-LDS)          5: (small-LDS, 512-VGPR kernel)
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int LDSB>
__global__ __launch_bounds__(256) void work(f32x4_t *__restrict__ buf, long n16, int spin) {
    __shared__ char lds[LDSB > 0 ? LDSB : 16];
    if (LDSB > 0 && threadIdx.x == 0) lds[(spin * 7) % LDSB] = 1;
    __syncthreads();
    float acc = LDSB > 0 ? (float)lds[0] : 0.0f;
    for (int s = 0; s < spin; ++s) acc = acc * 1.0001f + 0.5f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) buf[i] = (f32x4_t){acc, 1.0f, 2.0f, 3.0f};
}
__global__ __launch_bounds__(256) void fat(f32x4_t *__restrict__ buf, long n16) {  // many live registers
    float r[200];
#pragma unroll
    for (int i = 0; i < 200; ++i) r[i] = (float)(threadIdx.x + i);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 200; ++i) r[i] = r[i] * r[(i + 1) % 200] + 1.0f;
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 200; ++i) s += r[i];
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) buf[i] = (f32x4_t){s, 0.0f, 0.0f, 0.0f};
}

static void pair(int which, hipStream_t st, f32x4_t *buf) {
    const long big = (64l << 20) / 16;  // the first kernel writes 64 MB and spins: ~40-60 us
    for (int r = 0; r < 12; ++r) {
        if (which >= 3 && which <= 4) hipLaunchKernelGGL(work<160 * 1024>, dim3(256), dim3(256), 0, st, buf, big, 4000);
        else hipLaunchKernelGGL(work<1024>, dim3(2048), dim3(256), 0, st, buf, big, 4000);
        switch (which) {
            case 0: hipLaunchKernelGGL(work<0>, dim3(1), dim3(256), 0, st, buf, 256l, 0); break;
            case 1: hipLaunchKernelGGL(work<1024>, dim3(2048), dim3(256), 0, st, buf, 1l << 18, 0); break;
            case 2: hipLaunchKernelGGL(work<64 * 1024>, dim3(2048), dim3(256), 0, st, buf, 1l << 18, 0); break;
            case 3: hipLaunchKernelGGL(work<1024>, dim3(2048), dim3(256), 0, st, buf, 1l << 18, 0); break;
            case 4: hipLaunchKernelGGL(work<160 * 1024>, dim3(256), dim3(256), 0, st, buf, 1l << 18, 0); break;
            default: hipLaunchKernelGGL(fat, dim3(2048), dim3(256), 0, st, buf, 1l << 18); break;
        }
    }
    (void)hipStreamSynchronize(st);
}

int main() {
    f32x4_t *buf;
    if (hipMalloc(&buf, 256l << 20) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, 256l << 20);
    hipStream_t nb;
    (void)hipStreamCreateWithFlags(&nb, hipStreamNonBlocking);
    for (int s = 0; s < 2; ++s)
        for (int w = 0; w < 6; ++w) pair(w, s ? nb : (hipStream_t)0, buf);
    printf("done\n");
    return 0;
}
