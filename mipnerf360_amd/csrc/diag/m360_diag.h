/* Private to the diagnostics build (make diag -> libm360_diag.so); NOT part of the C-ABI in include/m360.h. */
#pragma once
#include "../../../include/m360.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Diagnostics build only (make -C mipnerf360_amd/csrc diag -> libm360_diag.so; never shipped, never loaded by the
 * package): kernels instrumented with s_memtime / s_memrealtime stamps, see tools/. */
int m360_diag_linear(const float *x, long M, int ldx, const float *w_packed, const float *b_packed, int n_pad, int k_pad,
                     float *y, int ldy, m360_stream_t stream);
/* variant: 0 = ping-pong kernel with stamps, 1 = software-pipelined kernel with stamps, 2 / 3 = the same two without */
int m360_diag_linear_bf16(const void *x, long M, int ldx, const void *w_packed, const float *b_packed, int n_pad,
                          int k_pad, void *y, int ldy, int variant, int ldw /* row stride of w_packed, elements */, m360_stream_t stream);
/* the half-tile / double-accumulator fp32 kernel (m360_linear_hd.hip.h) while it is evaluated against the product kernel */
int m360_diag_linear_hd(const float *x, long M, int ldx, const float *w_packed, const float *b_packed, int n_pad, int k_pad,
                        int act, float *y, int ldy, int ablate /* 0, or ABL bits of m360_linear_hd.hip.h: timing only */, unsigned *queue /* zeroed word or NULL */,
                        m360_stream_t stream);
/* 4 x uint64 per workgroup of the last m360_diag_linear_bf16 launch with variant 20 + ABL (the w32 kernel): cycles, 100 MHz ticks, slabs, epilogue cycles */
int m360_diag_read_w32_stamps(unsigned long long *out_host, int n);
int m360_diag_read_w16_stamps(unsigned long long *out_host, int n);
/* 4 x uint64 per workgroup of the last (ReLU or ablated) m360_diag_linear_hd launch: cycles, 100 MHz ticks, K-steps */
int m360_diag_read_hd_stamps(unsigned long long *out_host, int n);
/* m360_linear of THIS (diagnostics) library: 0 = the shape rule, 1 = always the 256 x 256 kernel, 2 = half tiles where they apply */
int m360_diag_force_linear_kernel(int which);
/* 16 x uint64 per workgroup, the first 256 workgroups (slot meaning: see the STAMP blocks of the two kernels) */
int m360_diag_read_stamps(unsigned long long *out_host, int n);
#ifdef __cplusplus
}
#endif
