// What is the dependency gap between two launches of one stream made of?  writer<MODE>(n bytes) then tiny(), many times per size; run under
// `rocprofv3 --kernel-trace` and read tiny.start - writer.end per (mode, size) (tools/diag/launch_gap_probe.py).
//   MODE 0: plain global stores   1: non-temporal stores   2: sc0 sc1 (system-scope, write-through) stores   3: loads only (no dirty line)
// hipcc --offload-arch=gfx950 -O2 tools/diag/launch_gap_probe.hip -o /tmp/launch_gap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4_t __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void writer(f32x4_t *__restrict__ buf, long n16, float *__restrict__ sink) {
    float acc = 0.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
        const f32x4_t v = {(float)i, 1.0f, 2.0f, 3.0f};
        if (MODE == 0) buf[i] = v;
        else if (MODE == 1) __builtin_nontemporal_store(v, buf + i);
        else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(buf + i), "v"(v) : "memory");
        else acc += buf[i].x;
    }
    if (MODE == 3 && acc == 12345.678f) *sink = acc;
}
template <int ID> __global__ void tiny(float *p) { if (threadIdx.x == 0 && p[1] == 42.0f) p[0] = 1.0f; }

template <int MODE> static void run(f32x4_t *buf, float *flag, long bytes, int reps) {
    const long n16 = bytes / 16;
    long blocks = (n16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(writer<MODE>, dim3((unsigned)blocks), dim3(256), 0, 0, buf, n16, flag + 8);
        hipLaunchKernelGGL(tiny<MODE>, dim3(1), dim3(64), 0, 0, flag);
    }
    (void)hipDeviceSynchronize();
}

int main() {
    f32x4_t *buf;
    float *flag;
    const long maxb = 256l << 20;
    if (hipMalloc(&buf, maxb) != hipSuccess || hipMalloc(&flag, 4096) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, maxb);
    (void)hipMemset(flag, 0, 4096);
    const long sizes[] = {64, 4096, 64 << 10, 1 << 20, 2 << 20, 8 << 20, 32 << 20, 128 << 20};
    for (long b : sizes) {  // the launches of one (mode, size) are told apart in the trace by their order: mode-major inside a size
        run<0>(buf, flag, b, 20);
        run<1>(buf, flag, b, 20);
        run<2>(buf, flag, b, 20);
        run<3>(buf, flag, b, 20);
    }
    printf("done\n");
    return 0;
}
