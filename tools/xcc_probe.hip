// Probe: which XCD (HW_REG_XCC_ID) does each workgroup of a launch land on?  (speed-only knowledge)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int LDS_BYTES>
__global__ void probe(int *xcc, int *cu) {
    __shared__ char lds[LDS_BYTES];
    unsigned x, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 8, 4)" : "=s"(hw));
    if (threadIdx.x == 0) { lds[0] = 1; xcc[blockIdx.x] = (int)x; cu[blockIdx.x] = (int)hw; }
    // keep the block resident for a while so one block per CU really occupies distinct CUs
    long t0 = clock64();
    while (clock64() - t0 < 200000) {}
    if (threadIdx.x == 1 && lds[0] == 7) xcc[0] = -1;
}
template <int LDS_BYTES>
void run(int grid, int block, const char *what) {
    int *dx, *dc;
    hipMalloc(&dx, grid * 4); hipMalloc(&dc, grid * 4);
    hipLaunchKernelGGL(probe<LDS_BYTES>, dim3(grid), dim3(block), 0, 0, dx, dc);
    std::vector<int> x(grid), c(grid);
    hipMemcpy(x.data(), dx, grid * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, grid * 4, hipMemcpyDeviceToHost);
    printf("%s grid=%d block=%d: xcc of blocks 0..31: ", what, grid, block);
    for (int i = 0; i < 32 && i < grid; ++i) printf("%d ", x[i]);
    int rr = 0; for (int i = 0; i < grid; ++i) rr += (x[i] == x[i % 8]);
    int hist[8] = {0}; for (int i = 0; i < (grid < 256 ? grid : 256); ++i) hist[x[i] & 7]++;
    printf("\n   blocks with xcc[b]==xcc[b%%8]: %d/%d ; first-256 histogram:", rr, grid);
    for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
    printf("\n");
    hipFree(dx); hipFree(dc);
}
int main() {
    run<131072>(256, 256, "persist-like (128KB LDS)");
    run<131072>(256, 256, "persist-like again");
    run<147456>(8192, 512, "v1-like (144KB LDS)");
    run<1024>(2048, 256, "small LDS");
    return 0;
}
