// Third probe of the dependency gap: libm360's own ring kernel (m360_linear_bf16, 256 x 256 layer) 24 times in a row from a PLAIN C++ program
// (no torch, no Python) - does the 10 us gap of the kernel traces come with the process or with the kernel?
//   hipcc tools/diag/launch_gap_lib_probe.cpp -Iinclude -Lmipnerf360_amd -lm360 -Wl,-rpath,$PWD/mipnerf360_amd -o /tmp/lgp3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "m360.h"

int main() {
    const long M = 262144;
    const int W = 256;
    void *x, *y, *wp;
    float *w, *b, *bp;
    if (hipMalloc(&x, M * W * 2) || hipMalloc(&y, M * W * 2) || hipMalloc(&wp, W * W * 2) || hipMalloc(&w, W * W * 4) || hipMalloc(&b, W * 4) || hipMalloc(&bp, W * 4)) return 1;
    (void)hipMemset(x, 0, M * W * 2);
    (void)hipMemset(w, 0, W * W * 4);
    (void)hipMemset(b, 0, W * 4);
    if (m360_pack_linear_bf16(w, b, W, W, W, W, wp, bp, nullptr) != M360_OK) { printf("pack: %s\n", m360_last_error()); return 2; }
    for (int r = 0; r < 24; ++r) {
        const int rc = m360_linear_bf16(r & 1 ? y : x, M, W, wp, bp, W, W, M360_ACT_RELU, r & 1 ? x : y, W, nullptr);
        if (rc != M360_OK) { printf("linear: %s\n", m360_last_error()); return 3; }
    }
    (void)hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
