/*
 * m360.h — C-ABI of libm360.so: the MI355X (gfx950) native ray-marching hot path of
 * mip-NeRF 360 as implemented by zhangkai0425/mipnerf360.
 *
 * Every entry point replaces one function (or one fused group of functions) of the
 * reference's Python hot path; the reference file:line is cited on each declaration
 * (paths relative to the reference repository root).
 *
 * Conventions
 *   - all pointers are DEVICE pointers (HBM) unless the name ends in `_host`;
 *   - all tensors are fp32, contiguous, row-major, shapes as written in brackets;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - the library never allocates device memory and never synchronises: scratch is a
 *     caller-provided `workspace` (size from the matching *_workspace_bytes query);
 *   - return value: M360_OK (0) or a negative M360_ERR_* code; `m360_last_error()` gives
 *     a thread-local human readable message for the last failure on the calling thread;
 *   - inputs are never written (the reference's in-place `g()` mutation of near/far/t_vals,
 *     intern/parameterization.py:15-21, is reproduced numerically, not physically).
 */
#ifndef M360_H_
#define M360_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M360_VERSION 200 /* 0.2.0: no process-wide switches any more (m360_set_* gone): per-call m360_hyper_t.tuning, caller-owned m360_side_t; test hooks in the diagnostics build only */

typedef void *m360_stream_t;

enum {
    M360_OK = 0,
    M360_ERR_INVALID_ARGUMENT = -1,
    M360_ERR_LAUNCH = -2,
    M360_ERR_WORKSPACE_TOO_SMALL = -3,
    M360_ERR_NO_DEVICE = -4
};

enum { M360_ACT_NONE = 0, M360_ACT_RELU = 1, M360_ACT_SIGMOID = 2, M360_ACT_RELU_MASK = 3 /* internal: m360_linear_dgrad */ };
/* "Paired rows": the layout of bf16 / bf16x3 hidden activations BETWEEN two layers that both run on the one-wave ring kernel
 * (model.py:44-50,132-146: the hidden activations never leave the MLP).  OR these flags into `act` of m360_linear_bf16 /
 * m360_linear_bf16x3 (both), m360_linear_bf16_split / m360_linear_bf16x3_bf16out (OUT: first layers), m360_linear_heads_bf16 /
 * m360_linear_heads_bf16x3 (IN: last layers, store_y = 0).  Inside every block of 2 rows x 64 columns of the full 256-row tiles the
 * four 64-byte quarters are transposed: the 128 bytes at row 2R hold [row 2R, columns 0-31 | row 2R + 1, columns 0-31], the 128
 * bytes at row 2R + 1 the columns 32-63 of both rows ([hi | lo] rows: each half the same way; same buffer, same strides).  The
 * kernel's epilogue holds exactly these lines in its lanes, so its stores need no exchange between lanes (16 of 65 instructions
 * per 16 x 64 block); its loader reads either layout at the same rate.  Rows beyond the last full 256-row tile stay plain.  A call
 * whose shape is not the ring kernel's (m360_linear_bf16_rows_pairable) fails with M360_ERR_INVALID_ARGUMENT; OUT needs ReLU. */
#define M360_ROWS_PAIRED_IN 0x100
#define M360_ROWS_PAIRED_OUT 0x200
#define M360_ROWS_PAIRED_MASK 0x300
/* With M360_ROWS_PAIRED_OUT: temporal instead of non-temporal output stores - for a caller that runs a BLOCK of rows through all layers of
 * an MLP before the next block, so that the block's ping / pong pair stays in the 256 MiB Infinity Cache between the layer that writes it
 * and the layer that reads it (non-temporal stores stream past it; what m360_nerf_forward / m360_forward do in the bf16 modes, below). */
#define M360_STORES_TEMPORAL 0x400
#define M360_ACT_FLAGS_MASK 0x700
enum { M360_PAIRABLE_LINEAR = 0 /* m360_linear_bf16 */, M360_PAIRABLE_X3 = 1 /* m360_linear_bf16x3 */, M360_PAIRABLE_SPLIT = 2 /* m360_linear_bf16_split */,
       M360_PAIRABLE_X3_BF16OUT = 3 /* m360_linear_bf16x3_bf16out */, M360_PAIRABLE_HEADS = 4 /* m360_linear_heads_bf16 */, M360_PAIRABLE_HEADS_X3 = 5 /* m360_linear_heads_bf16x3 */ };

int m360_version(void);
const char *m360_last_error(void);
/* number of HIP devices visible (0 when none) — lets hosts fail loudly and early */
int m360_device_count(void);

/* ------------------------------------------------------------------ sampling ---------- */

/* t_vals[B,N+1] = g(s*g(far) + (1-s)*g(near)), s = linspace(0,1,N+1), each g() adding 1e-6
 * first; optional stratified jitter with caller-supplied uniforms t_rand[B,N+1] (NULL =
 * deterministic).  Replaces intern/ray.py:99-110 (sample_along_rays, t part). */
int m360_sample_t(const float *near /*[B]*/, const float *far /*[B]*/, const float *t_rand,
                  int B, int N, float *t_vals, m360_stream_t stream);

/* The same with the jitter's uniforms drawn inside the kernel (randomized=True, intern/ray.py:103-108, without a [B,N+1] tensor of
 * torch.rand): element e = b (N + 1) + i uses the Philox4x32-10 uniform of (seed, offset, stream 0, e) - see m360_hyper_t.rng_seed. */
int m360_sample_t_philox(const float *near, const float *far, int B, int N, unsigned long long seed, unsigned long long offset,
                         float *t_vals, m360_stream_t stream);
/* out[e] = the uniform the kernels draw for element e of stream `stream_id` (0 = t_rand, 1 = u_rand) under (seed, offset): what a
 * launch used, written out - tests hand these to the CPU oracle as its t_rand / u_rand (replaces torch.rand at intern/ray.py:31,104). */
int m360_philox_uniform(unsigned long long seed, unsigned long long offset, int stream_id, long n, float *out, m360_stream_t stream);

/* y = g(x) = 1/(x + 1e-6), elementwise, x untouched.  Replaces intern/parameterization.py:15-21. */
int m360_g(const float *x, long n, float *y, m360_stream_t stream);

/* t_vals[B,M] = g(s*g(far) + (1-s)*g(near)) for caller-provided s_vals[B,M].
 * Replaces intern/parameterization.py:10-13 (s_to_t). */
int m360_s_to_t(const float *s_vals, const float *near, const float *far, int B, int M,
                float *t_vals, m360_stream_t stream);

/* s_vals[B,M] of t_to_s as evaluated by nerf_net.forward: near/far have already been through
 * g() `near_calls`/`far_calls` times.  Replaces intern/parameterization.py:5-8 at model.py:196. */
int m360_t_to_s(const float *t_vals /*[B,M]*/, const float *near, const float *far, int B, int M,
                int near_calls, int far_calls, float *s_vals, m360_stream_t stream);

/* ------------------------------------------------------------------ gaussians --------- */

/* stable conical-frustum moments.  Replaces intern/parameterization.py:99-107. */
int m360_frustum_moments(const float *t0 /*[B,N]*/, const float *t1 /*[B,N]*/,
                         const float *radii /*[B]*/, int B, int N, float *t_mean, float *t_var,
                         float *r_var, m360_stream_t stream);

/* the reference's other two public branches (never taken by its own hot path, kept for API completeness):
 * direct ("unstable") frustum moments, intern/parameterization.py:108-113 (stable=False), and the diagonal lift,
 * intern/parameterization.py:48-54 (diag=True): cov_diag[B,N,3]. */
int m360_frustum_moments_unstable(const float *t0, const float *t1, const float *radii, int B, int N,
                                  float *t_mean, float *t_var, float *r_var, m360_stream_t stream);
int m360_gaussian_to_xyz_diag(const float *d /*[B,3]*/, const float *t_mean /*[B,N]*/, const float *t_var,
                              const float *r_var, int B, int N, float *mean /*[B,N,3]*/,
                              float *cov_diag /*[B,N,3]*/, m360_stream_t stream);

/* lift to xyz, full covariance.  Replaces intern/parameterization.py:31-62 (diag=False). */
int m360_gaussian_to_xyz(const float *d /*[B,3]*/, const float *t_mean /*[B,N]*/,
                         const float *t_var, const float *r_var, int B, int N,
                         float *mean /*[B,N,3]*/, float *cov /*[B,N,3,3]*/, m360_stream_t stream);

size_t m360_contract_workspace_bytes(void);

/* y = contract(x): x if ||x||_F <= 1 else (2 - 1/||x||)(x/||x||), norm over ALL n elements.
 * Replaces intern/parameterization.py:23-29. */
int m360_contract(const float *x, long n, float *y, void *workspace, size_t workspace_bytes,
                  m360_stream_t stream);

/* whole-tensor-norm contraction of the means + J cov J^T with the closed-form Jacobian.
 * Replaces intern/parameterization.py:23-29,64-83 (the per-sample autograd loop). */
int m360_gaussian_contract(const float *mean_in /*[S,3]*/, const float *cov_in /*[S,3,3]*/, long S,
                           float *mean_out, float *cov_out, void *workspace, size_t workspace_bytes,
                           m360_stream_t stream);

/* t_vals -> contracted gaussians (+origins).  Replaces intern/parameterization.py:119-135. */
int m360_para_rays(const float *t_vals /*[B,N+1]*/, const float *origins /*[B,3]*/,
                   const float *directions /*[B,3]*/, const float *radii /*[B]*/, int B, int N,
                   float *means /*[B,N,3]*/, float *covs /*[B,N,3,3]*/, void *workspace,
                   size_t workspace_bytes, m360_stream_t stream);

/* ------------------------------------------------------------------ encodings --------- */

/* 21-direction integrated positional encoding -> enc[S,42]; cov == NULL gives the plain
 * positional encoding branch.  Replaces intern/encoding.py:33-61. */
int m360_ipe(const float *mean /*[S,3]*/, const float *cov /*[S,3,3] or NULL*/, long S,
             float *enc /*[S,42]*/, m360_stream_t stream);

/* theta/phi view-direction encoding -> enc[B,4*(max_deg-min_deg)].
 * Replaces intern/encoding.py:69-90. */
int m360_viewdir_enc(const float *viewdirs /*[B,3]*/, int B, int min_deg, int max_deg, float *enc,
                     m360_stream_t stream);

/* fused: t_vals + rays (+ per-ray view-direction encoding vdenc[B,vd_ch]) -> MLP input rows
 * feat[B*N, ld_feat] = [ipe(42) | vdenc(vd_ch) | zero pad].  No means/covs are materialised.
 * A NaN feature (NaN rays, |viewdir z| > 1) is written as the positive quiet NaN 0x7FC00000 whatever its source's sign.
 * Replaces model.py:82-88 / :167-176 minus the t sampling (fused para_rays + IPE + repeat + cat). */
int m360_encode_features(const float *t_vals /*[B,N+1]*/, const float *origins,
                         const float *directions, const float *radii, const float *vdenc, int vd_ch,
                         int B, int N, float *feat, int ld_feat, void *workspace,
                         size_t workspace_bytes, m360_stream_t stream);

/* ------------------------------------------------------------------ MLP --------------- */

/* zero-pad a PyTorch Linear ([n_out,k_in] weight, [n_out] bias or NULL) to [n_pad,k_pad] /
 * [n_pad]; n_pad, k_pad multiples of 32.  The packed layout keeps k contiguous. */
int m360_pack_linear(const float *w, const float *b, int n_out, int k_in, int n_pad, int k_pad,
                     float *w_packed, float *b_packed, m360_stream_t stream);

/* y[M,n_pad] = act(x[M,k_pad] * w_packed^T + b_packed) on fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * One nn.Linear + activation of model.py:43-53 / :131-148.
 * NaN: like torch.relu, the ReLU epilogue lets NaN through - a row of x that holds a NaN comes out NaN in every unit
 * (fixture G18).  It is a signed-integer max on the bit pattern, so this holds for NaNs with a CLEAR sign bit
 * (0x7FC00000, what numpy / torch write for float('nan') and what m360_encode_features writes for every NaN feature);
 * the matrix pipe hands a NaN operand on with its sign, and a NaN with the sign bit set (x86's default result of 0/0)
 * is treated as a negative number by the ReLU.  Sigmoid / none epilogues keep either. */
int m360_linear(const float *x, long M, int ldx, const float *w_packed, const float *b_packed,
                int n_pad, int k_pad, int act, float *y, int ldy, m360_stream_t stream);

/* The same layer with load balancing across the chip's 8 XCDs, whose clocks differ by 1-2 %: the last few percent of the
 * 128-row tiles are handed out through `tile_queue`, a caller-owned 4-byte device word that is ZERO when the launch starts
 * (the launch leaves it non-zero: zero it again - or pass a fresh word - before the next launch; the forward entry points
 * below keep 16 such words in their workspace).  NULL, or a layer the balanced kernel does not take, = m360_linear.
 * Results are bit-identical to m360_linear. */
int m360_linear_balanced(const float *x, long M, int ldx, const float *w_packed, const float *b_packed,
                         int n_pad, int k_pad, int act, float *y, int ldy, unsigned *tile_queue, m360_stream_t stream);

/* ---- training path (row f3): the two gradient GEMMs autograd runs for every nn.Linear of model.py:43-53 /
 * :131-158 under train.py:62,80.  All fp32 MFMA, deterministic (no atomics). */

/* wt_packed[k_pad,n_pad] (n contiguous) = zero-padded transpose of a PyTorch [n_out,k_in] weight. */
int m360_pack_linear_transposed(const float *w, int n_out, int k_in, int n_pad, int k_pad, float *wt_packed,
                                m360_stream_t stream);

/* input gradient: dx[M,k_pad] = dz[M,n_pad] * W[n_pad,k_pad], optionally masked by ReLU' of the previous layer:
 * relu_out[M,ldx] is that layer's forward OUTPUT (dx is zeroed where it is <= 0), NULL = no mask. */
int m360_linear_dgrad(const float *dz, long M, int ldz, const float *wt_packed, int k_pad, int n_pad,
                      const float *relu_out, float *dx, int ldx, m360_stream_t stream);

/* weight and bias gradient: grad_w[n_pad,k_pad] = dz[M,n_pad]^T * x[M,k_pad] (the packed layout of
 * m360_pack_linear), grad_b[n_pad] = column sums of dz (NULL to skip).  The M rows are split over the workgroups
 * and the partial tiles are added in a fixed order. */
size_t m360_linear_wgrad_workspace_bytes(long M, int n_pad, int k_pad);
int m360_linear_wgrad(const float *dz, int ldz, const float *x, int ldx, long M, int n_pad, int k_pad,
                      float *grad_w, float *grad_b, void *workspace, size_t workspace_bytes,
                      m360_stream_t stream);

/* flag[0] = 1 if any of the n_tensors fp32 tensors (host arrays of device pointers / element counts) holds a NaN, else 0 - ONE launch per
 * 32 tensors.  The host side of the bf16 modes uses it to refuse NaN parameters (the bf16 matrix pipe's NaN has its sign bit set, and the
 * packed integer-max ReLU of those modes reads it as a negative number: model.py:43-53,131-148 render NaN through nn.ReLU, these modes
 * could not). */
int m360_params_nan_flag(const float *const *tensors, const long *counts, int n_tensors, unsigned *flag, m360_stream_t stream);

/* ---- opt-in bf16 MLP (BASELINE configs[4]): bf16 inputs, fp32 accumulate on v_mfma_f32_16x16x32_bf16.
 * bf16 tensors are passed as raw 16-bit storage (void*).  k_pad multiple of 64, ldx/ldy multiples of 8.  Which kernel takes the
 * full 256 x 256 tiles depends on the call's shape alone: bias + {none, ReLU} with k_pad a multiple of 128 (>= 256) or k_pad = 64
 * -> the one-wave ring kernel (m360_linear_bf16_w16.hip.h); other contractions >= 128 and sigmoid -> the 8-wave ping-pong kernel;
 * ragged rows / widths -> the generic kernel.  The ring and the ping-pong kernel accumulate the same 32-deep MFMA k-steps in the
 * same order (bit-identical results); all kernels accumulate in fp32 and round once to bf16. */
int m360_pack_linear_bf16(const float *w, const float *b, int n_out, int k_in, int n_pad, int k_pad,
                          void *w_packed_bf16, float *b_packed, m360_stream_t stream);
int m360_linear_bf16(const void *x_bf16, long M, int ldx, const void *w_packed_bf16, const float *b_packed,
                     int n_pad, int k_pad, int act, void *y_bf16, int ldy, m360_stream_t stream);
/* ---- bf16 training path (round 5; SURVEY.md §8 row f3 in the precision BASELINE configs[4] runs at): what autograd computes for
 * nn.Linear under train.py:62,80 (model.py:43-53,131-158) with activations and their gradients carried in bf16 - fp32 accumulation,
 * fp32 gradients out, fp32 master weights / AdamW untouched.
 *   m360_pack_linear_bf16_transposed: fp32 [n_out, k_in] -> bf16 transpose [k_pad, n_pad] (n_pad a multiple of 64), the operand of the
 *     input gradient;
 *   m360_linear_dgrad_bf16: dX[M, k_pad] = dZ[M, n_pad] * W as the forward layer kernel on that packing (bf16 in, bf16 out); relu_out (the
 *     forward OUTPUT of the layer below, bf16, leading dimension ldx; NULL = no mask) clears dX where it is <= 0 (ReLU');
 *   m360_linear_wgrad_bf16: dW[n_pad, k_pad] = dZ^T X and db[n_pad] = column sums of dZ (NULL to skip) in fp32 from bf16 rows.  n_pad, k_pad
 *     multiples of 256 run on v_mfma_f32_16x16x32_bf16 (both operands transposed on their way out of the LDS by ds_read_b64_tr_b16; row splits
 *     reduced in a fixed order: deterministic); other pads (multiples of 32: reduced-width models) are widened to fp32 for m360_linear_wgrad. */
/* Which MFMA form m360_linear_wgrad_bf16 takes is a per-call choice (`tuning`, M360_TUNE_* below): 0 (default) = one wave per SIMD with
 * 128 x 128 wave tiles and the LDS filled four k-steps ahead, M360_TUNE_WGRAD_FORM0 = the 8-wave kernel; both deterministic, results equal up to
 * fp32 summation order.  m360_prop_backward / m360_nerf_backward read the same bit from m360_hyper_t.tuning. */
int m360_pack_linear_bf16_transposed(const float *w, int n_out, int k_in, int n_pad, int k_pad, void *wt_packed_bf16 /*[k_pad, n_pad]*/,
                                     m360_stream_t stream);
int m360_linear_dgrad_bf16(const void *dz_bf16, long M, int ldz, const void *wt_packed_bf16, int k_pad, int n_pad, const void *relu_out_bf16,
                           void *dx_bf16, int ldx, m360_stream_t stream);
size_t m360_linear_wgrad_bf16_workspace_bytes(long M, int n_pad, int k_pad);
int m360_linear_wgrad_bf16(const void *dz_bf16, int ldz, const void *x_bf16, int ldx, long M, int n_pad, int k_pad, float *grad_w,
                           float *grad_b, void *workspace, size_t workspace_bytes, unsigned tuning, m360_stream_t stream);
/* ---- opt-in "bf16x3" MLP: near-fp32 accuracy on the bf16 matrix pipe.  Every activation and weight is carried as TWO bf16
 * terms (hi = bf16(v), lo = bf16(v - hi): 16 significant bits) and a product x w is formed as xh wh + xl wh + xh wl with
 * fp32 accumulation (the xl wl term, 2^-16 of the product, is dropped): three bf16 MFMA passes per 64-deep block, in the order
 * xl wh, xh wh, xh wl (bias + {none, ReLU} layers on the one-wave ring kernel, sigmoid layers on the ping-pong kernel).  Layouts: activations [M, 2 K] = [hi | lo]; packed weights [n_pad, 3 k_pad] = [Wh | Wh | Wl]; the output is
 * written as [M, 2 n_pad] = [hi | lo] again.  Measured against the fp32 path: hidden activations differ by <= 2e-5, rendered
 * colours by <= 2e-5 against the reference's own outputs (fixture G8; the stated fp32 tolerance is 1e-4) at ~3x the
 * fp32 rays/s.  Never the default. */
int m360_pack_linear_bf16x3(const float *w, const float *b, int n_out, int k_in, int n_pad, int k_pad,
                            void *w_packed3_bf16 /*[n_pad, 3 k_pad]*/, float *b_packed, m360_stream_t stream);
int m360_linear_bf16x3(const void *x_hi_lo_bf16 /*[M, ldx >= 2 k_pad]*/, long M, int ldx, const void *w_packed3_bf16,
                       const float *b_packed, int n_pad, int k_pad, int act, void *y_hi_lo_bf16 /*[M, ldy >= 2 n_pad]*/,
                       int ldy, m360_stream_t stream);

/* ---- first layers of the bf16 / bf16x3 modes at fp32 accuracy ("x6", round 4).  The reference contracts a whole chunk by its
 * Frobenius norm (intern/parameterization.py:23-29, called at :75): a ray's samples end up within ~1e-2 of each other in the
 * encoder's coordinates, so a network that resolves anything along a ray has first-layer gains of 1e3-1e4 on differences of the
 * sin / cos features (intern/encoding.py:33-56) and needs all 24 bits of them.  Features go to the first layer as THREE bf16 terms
 * (hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid); exact) in "x6" rows [lo | mid | hi | mid | hi | hi] of k_pad
 * columns each (m360_encode_features_grouped / _ext_norm with bf16 = 3), the first-layer weights (model.py:44,132) are packed as
 * [n_pad, 6 k_pad] = [Wh | Wm | Wl | Wh | Wm | Wh], and ONE plain bf16 contraction of length 6 k_pad forms
 * xl wh + xm wm + xh wl + xm wh + xh wm + xh wh, small terms first, fp32 accumulation: the fp32 product up to 2^-24 terms.
 * m360_linear_bf16_split ([hi | lo] pair rows out, the row format of m360_linear_bf16x3; bias + {none, ReLU}) runs it in the bf16x3
 * mode (m360_linear_bf16 takes x6 rows too: bf16 rows out); 6 k_pad = 384 takes the one-wave ring kernel.
 * Non-finite values: a NaN feature or weight makes its row NaN as in fp32 (hi carries it, mid = lo = 0).  An INFINITE weight
 * differs: the fp32 product x * Inf is +-Inf, the split products contain 0 * Inf = NaN whenever a term of x is zero (any x
 * that bf16 represents exactly) - such a checkpoint renders NaN one layer earlier than the fp32 path (which turns it into NaN
 * in the next layer's mixed-sign sum); the same holds for the two-term products of m360_linear_bf16x3. */
/* The bf16 mode's first layers need 16 of those bits, not 24 (their output is rounded to bf16 anyway; fixture G19: PSNR within
 * 0.013 dB of the reference's): two-term features [hi | lo] and weights [Wh | Wh | Wl] as for m360_linear_bf16x3, ONE bf16 term out -
 * m360_linear_bf16x3_bf16out (bias + {none, ReLU}; the ring kernel's three-product loop with the plain bf16 epilogue: a 64-deep layer
 * stays one block per tile and store-bound, 0.30 instead of the 0.53 ms of the x6 form at 1024 x 58 on 524 288 rows). */
/* `layers` (1..8) equally shaped hidden layers (model.py:134-146: Linear(width, width) + ReLU; bf16 weights of m360_pack_linear_bf16, PAIRED
 * rows in and out) in ONE launch: layer j reads act[j & 1] and writes act[(j + 1) & 1] (the result is in act[layers & 1]), the activations
 * handed over through the L2 of the XCD whose four CUs own a row block (a counter per row block and layer in `workspace`, zeroed by the
 * call) instead of a kernel boundary - same bits as `layers` calls of m360_linear_bf16.  Shapes: width 1024 with M a multiple of 32768,
 * or width 256 (a workgroup then owns whole rows: its own previous tile is all its next layer needs) with M a multiple of 131072; a
 * 256-CU device: m360_mlp_chain_bf16_supported answers.
 * The hand-over relies on two things no API promises, and the kernel checks both itself (round 5): (1) all workgroups b with the same b % 8 run
 * on ONE XCD - every workgroup reads HW_REG_XCC_ID and compares it with its slot's; (2) all 256 workgroups are resident at the same time - a
 * kernel on another stream that holds CUs breaks that while it runs; every wait is bounded by wall-clock time (0.1 s), so a launch always
 * ends.  A launch that found either violated sets the `error` word of the status block at the start of `workspace`
 * (m360_workspace_status).  m360_mlp_chain_bf16 leaves it at that (its rows are then not to be trusted: the caller must check);
 * m360_mlp_chain_bf16_safe - what m360_forward / m360_nerf_forward use - reads layer 0's input from a third buffer `x_in` that no layer
 * writes and queues `layers` GATED launches of m360_linear_bf16's own kernel behind the chain: their workgroups return at once while the
 * error word is 0 and redo every row layer by layer when it is not (same bits, no host round trip, nothing to check before using the result;
 * the status block's sticky counters say how often it happened).  workspace: m360_mlp_chain_bf16_workspace(M, layers) bytes,
 * 16-byte aligned = [status block, 128 bytes | counters]. */
int m360_mlp_chain_bf16_supported(long M, int width, int layers);
size_t m360_mlp_chain_bf16_workspace(long M, int layers);
/* opts (may be NULL = defaults): only its per-call switches are read - m360_hyper_t.tuning (M360_TUNE_CHAIN_COOPERATIVE) and, in the diagnostics
 * build, the chain_debug_* test hooks. */
struct m360_hyper;
int m360_mlp_chain_bf16(void *act0_bf16, void *act1_bf16, long M, int ld, const void *const *w_packed_bf16 /*[layers]*/, const float *const *b_packed /*[layers]*/,
                        int layers, int width, void *workspace, const struct m360_hyper *opts, m360_stream_t stream);
int m360_mlp_chain_bf16_safe(const void *x_in_bf16, void *act0_bf16, void *act1_bf16, long M, int ld, const void *const *w_packed_bf16 /*[layers]*/,
                             const float *const *b_packed /*[layers]*/, int layers, int width, void *workspace, const struct m360_hyper *opts,
                             m360_stream_t stream);
/* The same for the bf16x3 mode's hidden layers (round 6; model.py:134-146 with every value as two bf16 terms): rows are [hi | lo] pairs
 * (ld >= 2 width), weights those of m360_pack_linear_bf16x3 ([Wh | Wh | Wl]: 3 width per row), width 1024, M a multiple of 32768 - same bits as
 * `layers` calls of m360_linear_bf16x3 on paired rows, same self-checks, same gated re-run. */
int m360_mlp_chain_bf16x3_safe(const void *x_in_hi_lo_bf16, void *act0_hi_lo_bf16, void *act1_hi_lo_bf16, long M, int ld,
                               const void *const *w_packed3_bf16 /*[layers]*/, const float *const *b_packed /*[layers]*/, int layers, int width,
                               void *workspace, const struct m360_hyper *opts, m360_stream_t stream);
/* The status block: the first 128 bytes of a chain workspace AND of every forward workspace (m360_forward_workspace_bytes).
 * m360_workspace_init zeroes its sticky counters (once, after allocating the workspace; m360_mlp_chain_bf16 does it per call);
 * m360_workspace_status copies it out and waits for the stream: out5 = {chain launches that ran, launches repaired by the gated re-run,
 * waves whose wait ran out, workgroups off their slot's XCD, error word of the LAST launch}.  Counters of a workspace that was never
 * initialised are meaningless; no result ever depends on them. */
int m360_workspace_init(void *workspace, m360_stream_t stream);
int m360_workspace_status(const void *workspace, unsigned *out5_host, m360_stream_t stream);
/* 1 when the call `kind` (M360_PAIRABLE_*) with these pads runs its full tiles on the one-wave ring kernel, i.e. takes paired rows */
int m360_linear_bf16_rows_pairable(int kind, int n_pad, int k_pad);
/* m360_forward / m360_prop_forward / m360_nerf_forward (bf16 modes) use paired rows between the layers of an MLP whose layers are all
 * pairable, and run the six hidden NeRF layers as ONE launch (m360_mlp_chain_bf16_safe) for the rows it takes (multiples of 32768; width 1024;
 * paired rows; the proposal MLP's 256-wide layers stay launches: the chain is slower there).  Both can be switched off PER CALL through
 * m360_hyper_t.tuning (M360_TUNE_PLAIN_ROWS, M360_TUNE_NO_HIDDEN_CHAIN): A/B runs on one box - the outputs are the same bits.
 * (Until 0.2.0 these were process-wide m360_set_* switches; the library now holds no mutable state at all.  The row-block form of the NeRF
 * MLP is gone with its switch: measured in round 4, superseded by the chain, profiles/HISTORY.md.) */
int m360_linear_bf16x3_bf16out(const void *x_hi_lo_bf16 /*[M, ldx >= 2 k_pad]*/, long M, int ldx, const void *w_packed3_bf16,
                               const float *b_packed, int n_pad, int k_pad, int act, void *y_bf16 /*[M, ldy >= n_pad]*/, int ldy,
                               m360_stream_t stream);
int m360_pack_linear_bf16x6(const float *w, const float *b, int n_out, int k_in, int n_pad, int k_pad,
                            void *w_packed6_bf16 /*[n_pad, 6 k_pad]*/, float *b_packed, m360_stream_t stream);

/* Every packing of a parameter set in ONE launch (round 6).  A training step re-packs both networks on every forward and their transposes on
 * every backward: a 4 us kernel per layer behind a NaN scan of the same tensors, each with its launch gap, in front of a 6 ms forward.  An item
 * is the argument list of one per-layer entry point: M360_PACK_F32 = m360_pack_linear, _BF16 = m360_pack_linear_bf16, _BF16X3 =
 * m360_pack_linear_bf16x3, _BF16X6 = m360_pack_linear_bf16x6 (w_packed [n_pad, k_pad] / [n_pad, 3 k_pad] / [n_pad, 6 k_pad]; b NULL = zero bias,
 * b_packed NULL = none written), _F32_T = m360_pack_linear_transposed, _BF16_T = m360_pack_linear_bf16_transposed (w_packed [k_pad, n_pad]; no
 * bias).  Same pads required, same bits written as by those calls.  nan_flag (device, or NULL): set to 1 when any source weight or bias read is
 * NaN, else 0 - what m360_params_nan_flag reports for the same tensors (the bf16 modes refuse NaN parameters).  The list is checked as a whole
 * before anything is launched; lists longer than 16 items take one launch per 16. */
#define M360_PACK_F32 0
#define M360_PACK_BF16 1
#define M360_PACK_BF16X3 2
#define M360_PACK_BF16X6 3
#define M360_PACK_F32_T 4
#define M360_PACK_BF16_T 5
typedef struct m360_pack_item {
    const float *w;   /* [n_out, k_in] fp32, row-major (nn.Linear.weight) */
    const float *b;   /* [n_out] or NULL */
    void *w_packed;
    float *b_packed;  /* [n_pad] or NULL */
    int n_out, k_in, n_pad, k_pad;
    int format;       /* M360_PACK_* */
    int reserved;     /* 0 */
} m360_pack_item_t;
int m360_pack_many(const m360_pack_item_t *items_host, int count, unsigned *nan_flag, m360_stream_t stream);
int m360_linear_bf16_split(const void *x_bf16 /*[M, ldx >= k_pad]*/, long M, int ldx, const void *w_packed_bf16,
                           const float *b_packed, int n_pad, int k_pad, int act, void *y_hi_lo_bf16 /*[M, ldy >= 2 n_pad]*/,
                           int ldy, m360_stream_t stream);

/* m360_encode_features writing bf16 rows */
int m360_encode_features_bf16(const float *t_vals, const float *origins, const float *directions,
                              const float *radii, const float *vdenc, int vd_ch, int B, int N,
                              void *feat_bf16, int ld_feat, void *workspace, size_t workspace_bytes,
                              m360_stream_t stream);

/* m360_encode_features[_bf16] with per-chunk contraction norms: rays [g*group_rays, (g+1)*group_rays) form group g
 * and are contracted with the norm of their own means (what parameterization.py:25 sees when render_image,
 * model.py:262-264, feeds that chunk alone).  group_rays = 0: one norm for the batch.  group_rays * N <= 131072 and
 * at most 1024 groups.  A group's norm is reduced by one workgroup in an order that depends only on the group, so
 * grouped and chunk-by-chunk launches agree bit for bit. */
int m360_encode_features_grouped(const float *t_vals, const float *origins, const float *directions,
                                 const float *radii, const float *vdenc, int vd_ch, int B, int N, void *feat,
                                 int ld_feat, int bf16, int group_rays, void *workspace, size_t workspace_bytes,
                                 m360_stream_t stream);

/* ------------------------------------------------------------------ per-ray scans ----- */

/* model.py:59-78 (prop_net.density_to_weight); density[B,N]. */
int m360_density_to_weight(const float *t_vals /*[B,N+1]*/, const float *density,
                           const float *dirs /*[B,3]*/, int B, int N, float *weights,
                           m360_stream_t stream);

/* intern/ray.py:12-57; bins[B,nb], weights[B,nb-1], samples[B,num_samples];
 * u_rand[B,num_samples] uniforms for the randomized branch or NULL. */
int m360_sorted_pdf(const float *bins, const float *weights, const float *u_rand, int B, int nb,
                    int num_samples, float *samples, m360_stream_t stream);

/* intern/ray.py:136-149: max-blur + padding + inverse-CDF -> t_new[B,N+1]. */
int m360_resample_t(const float *t_vals /*[B,N+1]*/, const float *weights /*[B,N]*/,
                    const float *u_rand, int B, int N, float resample_padding, float *t_new,
                    m360_stream_t stream);

/* Extension of m360_resample_t: num_out resampled values per ray instead of N+1 (the reference cannot
 * express different proposal / NeRF sample counts, intern/ray.py:147; used for "64+128" rendering). */
int m360_resample_t_n(const float *t_vals, const float *weights, const float *u_rand, int B, int N,
                      int num_out, float resample_padding, float *t_new /*[B,num_out]*/,
                      m360_stream_t stream);

/* m360_sorted_pdf / m360_resample_t_n in the randomized branch (intern/ray.py:30-35) with the uniforms drawn inside the kernel: element
 * e = b num_samples + j uses the Philox uniform of (seed, offset, stream 1, e). */
int m360_sorted_pdf_philox(const float *bins, const float *weights, int B, int nb, int num_samples, unsigned long long seed,
                           unsigned long long offset, float *samples, m360_stream_t stream);
int m360_resample_t_philox(const float *t_vals, const float *weights, int B, int N, int num_out, float resample_padding,
                           unsigned long long seed, unsigned long long offset, float *t_new, m360_stream_t stream);

/* intern/ray.py:155-191; rgb[B,N,3], density[B,N]; weights may be NULL. */
int m360_volumetric_rendering(const float *rgb, const float *density, const float *t_vals,
                              const float *dirs, int B, int N, int white_bkgd, float *comp_rgb,
                              float *distance, float *acc, float *weights, m360_stream_t stream);

/* float -> uint8 image quantisation.  Replaces intern/utils.py:17-20. */
int m360_to8b(const float *x, long n, uint8_t *out, m360_stream_t stream);

/* ------------------------------------------------------------------ ray generation ---- */

/* Pinhole rays for every pixel of n_cams cameras (cam_to_world[n_cams,3,4], row-major), flattened to
 * [n_cams*h*w, .] like NeRFDataset.flatten_to_pytorch: origins, un-normalised directions, unit viewdirs,
 * radii (distance to the next-row neighbour * 2/sqrt(12)), near, far.  ndc != 0 additionally converts
 * origins/directions to NDC (near plane ndc_near, 1.0 in the reference) and takes the radii from the NDC
 * origins of the row and column neighbours.  The last row / column repeats the SECOND-to-last neighbour
 * distance, as the reference's `dx[:, -2:-1]` padding does; h, w >= 3.
 * Replaces dataset.py:109-145 (NeRFDataset.generate_rays), dataset.py:364-387 (LLFF.generate_rays). */
int m360_generate_rays(const float *cam_to_world, int n_cams, int h, int w, float focal, float near,
                       float far, int ndc, float ndc_near, float *origins, float *directions,
                       float *viewdirs, float *radii, float *near_out, float *far_out,
                       m360_stream_t stream);

/* The same rays for the flat pixel span [first, first+count) of the n_cams*h*w pixels only (output row 0 = pixel
 * `first`): what one rank of a ray-sharded frame render needs (each GPU generates the rays of its own
 * block of chunks, nothing but the pose is broadcast).  Values are bit-identical to the rows of the full call. */
int m360_generate_rays_span(const float *cam_to_world, int n_cams, int h, int w, float focal, float near,
                            float far, int ndc, float ndc_near, long first, long count, float *origins,
                            float *directions, float *viewdirs, float *radii, float *near_out, float *far_out,
                            m360_stream_t stream);

/* NDC conversion of n rays.  Replaces intern/ray.py:59-79 (convert_to_ndc). */
int m360_convert_to_ndc(const float *origins /*[n,3]*/, const float *directions /*[n,3]*/, long n,
                        float focal, int w, int h, float near, float *origins_out, float *directions_out,
                        m360_stream_t stream);

/* ------------------------------------------------------------------ visualisation ----- */

size_t m360_visualize_workspace_bytes(void);

/* fake normals of an orthographic depth map: 3x3 blur/edge convolution (scipy convolve2d 'same', zero
 * fill) -> normals[h,w,3].  Replaces intern/pose.py:112-121 (depth_to_normals). */
int m360_depth_to_normals(const float *depth /*[h,w]*/, int h, int w, float *normals, m360_stream_t stream);

/* cyclic colormap rgb[n,3] = sin^2(pi (k/6 - h)), k = 3,5,7.  Replaces intern/pose.py:122-125 (sinebow). */
int m360_sinebow(const float *h, long n, float *rgb, m360_stream_t stream);

/* vis[h,w,3] of the isotropically scaled fake normals, NaN -> 1, blended to white with acc (acc may be
 * NULL).  Replaces intern/pose.py:127-146 (visualize_normals, scaling=None). */
int m360_visualize_normals(const float *depth, const float *acc, int h, int w, float *vis,
                           void *workspace, size_t workspace_bytes, m360_stream_t stream);

/* vis[h,w,3] of a depth map: curve -log(x + eps32), near/far given or taken from the map (near_auto /
 * far_auto != 0: lowest / highest depth -/+ eps, i.e. the reference's ignore_frac = 0 behaviour),
 * modulus == 0: matplotlib 'turbo' lookup, modulus > 0: sinebow of mod(value, modulus)/modulus; blended
 * to white with acc (NULL = ones; NaN depth -> acc 0).  Replaces intern/pose.py:148-212 (visualize_depth
 * with its default curve_fn / colormap). */
int m360_visualize_depth(const float *depth, const float *acc, int h, int w, float near, float far,
                         int near_auto, int far_auto, float modulus, float *vis, void *workspace,
                         size_t workspace_bytes, m360_stream_t stream);

/* ------------------------------------------------------------------ losses (row f3) --- */

/* workspace for the three loss entry points; N = number of proposal intervals (0 for loss_dist / loss_nerf) */
size_t m360_loss_workspace_bytes(int B, int N);

/* proposal ("envelope") loss, intern/loss.py:6-21 = distillation.py:4-51.  The reference's bounds() selects
 * fine_weights[..., mask] with a [B,Nf] mask (distillation.py:29), which flattens over the rays, so
 *   bounds[b,i] = sum over ALL rays b' and fine intervals j of w[b',j] [not (t0[b',j] > T1[b',i] or t1[b',j] < T0[b',i])]
 * (the batch total, identical for every b) - reproduced here: per-ray overlap sums, then an fp64 column sum.
 * loss[0] = sum relu(bounds - w_hat)^2 / (w_hat + 1e-6) / B; optional outputs bounds[B,Np] and
 * grad_w_hat[B,Np] = d loss / d w_hat (bounds detached, as in the reference).  t == NULL: `w` [B,Np] already
 * holds the bounds (the reference's two-step bounds() -> loss_prop() use), t_hat is ignored, Nf must equal Np. */
int m360_loss_prop(const float *t /*[B,Nf+1]*/, const float *w /*[B,Nf]*/, const float *t_hat /*[B,Np+1]*/,
                   const float *w_hat /*[B,Np]*/, int B, int Nf, int Np, float *bounds, float *loss,
                   float *grad_w_hat, void *workspace, size_t workspace_bytes, m360_stream_t stream);

/* distortion loss, summed over rays: sum_ij w_i w_j |m_i - m_j| + 1/3 sum_i w_i^2 (s_{i+1} - s_i);
 * optional gradients grad_w[B,N], grad_s[B,N+1].  Replaces intern/regularization.py:3-19 (O(N^2) Python
 * double loop) and intern/loss.py:42-54. */
int m360_loss_dist(const float *s_vals /*[B,N+1]*/, const float *weights /*[B,N]*/, int B, int N, float *loss,
                   float *grad_w, float *grad_s, void *workspace, size_t workspace_bytes,
                   m360_stream_t stream);

/* reconstruction loss: mse = sum (input - target)^2 / B; out3 = {10 log10(mse) + 30, psnr, mse}; optional
 * grad_input[B,C] of out3[0].  Replaces intern/loss.py:23-40,57-59 (Loss_nerf, mse_to_psnr). */
int m360_loss_nerf(const float *input /*[B,C]*/, const float *target, int B, int C, float *out3,
                   float *grad_input, void *workspace, size_t workspace_bytes, m360_stream_t stream);

/* ------------------------------------------------------------------ fused stages ------ */

/* last proposal layer (hidden -> 1) + softplus(raw + density_bias) + density_to_weight +
 * resample: act[B*N, ld] is the sigmoid output of the 4th proposal layer.
 * Replaces model.py:52 (Linear(h,1)), :92-93 and intern/ray.py:136-149. */
int m360_prop_finish(const float *act, int ld, const float *head_w /*[k_pad]*/,
                     const float *head_b /*[1]*/, int k_pad, float density_bias,
                     const float *t_vals, const float *dirs, const float *u_rand, int B, int N,
                     float resample_padding, float *weights /*[B,N]*/, float *t_new /*[B,N+1] or NULL*/,
                     m360_stream_t stream);

/* m360_prop_finish with num_out resampled values per ray (t_new[B,num_out]) instead of N+1. */
int m360_prop_finish_n(const float *act, int ld, const float *head_w, const float *head_b, int k_pad,
                       float density_bias, const float *t_vals, const float *dirs, const float *u_rand,
                       int B, int N, int num_out, float resample_padding, float *weights, float *t_new,
                       m360_stream_t stream);

/* m360_prop_finish_n / m360_nerf_finish reading bf16 activations (heads and scans stay fp32) */
int m360_prop_finish_bf16(const void *act_bf16, int ld, const float *head_w, const float *head_b, int k_pad,
                          float density_bias, const float *t_vals, const float *dirs, const float *u_rand,
                          int B, int N, int num_out, float resample_padding, float *weights, float *t_new,
                          m360_stream_t stream);
int m360_nerf_finish_bf16(const void *act_bf16, int ld, const float *head_w, const float *head_b, int k_pad,
                          float density_bias, float rgb_padding, const float *t_vals, const float *dirs,
                          int B, int N, int white_bkgd, float *comp_rgb, float *distance, float *acc,
                          float *weights, m360_stream_t stream);

/* density/colour heads + activations + alpha composite: head_w[4,k_pad] rows = (density, r, g, b).
 * Replaces model.py:150-158,180-186 and intern/ray.py:155-191. */
int m360_nerf_finish(const float *act, int ld, const float *head_w, const float *head_b /*[4]*/,
                     int k_pad, float density_bias, float rgb_padding, const float *t_vals,
                     const float *dirs, int B, int N, int white_bkgd, float *comp_rgb,
                     float *distance, float *acc, float *weights /*or NULL*/, m360_stream_t stream);

/* visualize_depth with the reference's remaining options (intern/pose.py:148-212):
 *   ignore_frac > 0: the automatic near / far planes are the first / last depth of the depth-SORTED map whose running sum
 *     of acc lies inside [ignore_frac, 1 - ignore_frac] of the total (device radix sort + numpy's sequential float32 cumsum);
 *   curved != 0: `depth`, `near`, `far` already went through the caller's own curve_fn (a host callable in the reference's
 *     API) - the default -log(x + eps) is then skipped; automatic planes cannot be combined with it;
 *   value_out != NULL: write the normalised value [h,w] (the argument of the colormap) instead of colours, for a
 *     caller-supplied colormap callable, whose colours m360_visualize_composite then blends with acc;
 *   planes_out != NULL (device float[2]): the automatic near / far planes (with their -/+ eps), e.g. to curve them on the host.
 * vis and value_out may both be NULL (planes only). */
size_t m360_visualize_depth_ex_workspace_bytes(int h, int w);
int m360_visualize_depth_ex(const float *depth, const float *acc, int h, int w, float near, float far, int near_auto,
                            int far_auto, float ignore_frac, int curved, float modulus, float *vis /*[h,w,3]*/,
                            float *value_out /*[h,w]*/, float *planes_out /*[2]*/, void *workspace,
                            size_t workspace_bytes, m360_stream_t stream);
/* vis[h,w,3] = colors[h,w,3] * acc + (1 - acc), acc := 0 where depth is NaN (depth may be NULL).  intern/pose.py:207-210. */
int m360_visualize_composite(const float *colors, const float *acc, const float *depth, int h, int w, float *vis,
                             m360_stream_t stream);

/* ------------------------------------------------------------------ fused last layer + heads ----- */

/* The LAST hidden layer of a stage (sigmoid, model.py:50 / :146) fused with the stage's output heads (model.py:52:
 * hidden -> 1; model.py:150-158: hidden -> 1 + 3): while a 256 x 256 output tile is still in registers its activated
 * values are multiplied with the `heads` (1 or 4) head rows head_w[heads,n_pad] (fp32, the m360_model_t layout) and
 * per-row PARTIAL sums go to head_part[fused_rows][slots][heads] (slots = m360_linear_heads_slots(n_pad, bf16), one per
 * wave tile of 128 (fp32) or 64 (bf16) columns; no bias).  store_y = 0: the fused rows of y are NOT written (rendering: the 2.15 GB activation
 * of the last NeRF layer never reaches HBM); store_y = 1: also written (training tape).  Rows beyond
 * fused_rows = m360_linear_heads_fused_rows(M, n_pad, bf16) (ragged tail; every row when n_pad % 256 != 0 or n_pad > 1024)
 * are computed as by m360_linear into y, and the *_finish_fused entry points take their head products from there.
 * Replaces the last nn.Linear + nn.Sigmoid of model.py:43-53 / :131-148 plus the matrix product of the heads. */
/* (bf16 = 1 with k_pad < 128 and full tiles - a shape the stage drivers never produce, their last layers are square - is rejected by
 * m360_linear_heads_bf16 with M360_ERR_INVALID_ARGUMENT: no kernel fuses the heads of a single 64-deep K-step.  bf16 / bf16x3: the
 * ragged tail rows are formed by the generic kernel (v_mfma_f32_32x32x16_bf16, one 3K-deep contraction in bf16x3) and the full
 * tiles by the ring / ping-pong kernels (v_mfma_f32_16x16x32_bf16, xl wh -> xh wh -> xh wl per 64-deep block): the same products,
 * a different fp32 summation order - a row's low-order bits depend on which side of fused_rows it lies; only the fp32 path
 * promises the same bits from every kernel.) */
long m360_linear_heads_fused_rows(long M, int n_pad, int bf16);
int m360_linear_heads_slots(int n_pad, int bf16);
/* bf16 / bf16x3 (bf16 = 1 / 2): the number of slots m360_linear_heads_bf16 / _bf16x3 writes for THESE arguments - 2 per 256 columns
 * when the one-wave ring kernel takes the layer (store_y = 0 and a contraction of its shape: one partial sum per 128-column wave
 * tile), else 8 (ping-pong kernel).  Never more than m360_linear_heads_slots(n_pad, bf16): size head_part with that. */
int m360_linear_heads_slots_bf16(int n_pad, int k_pad, int bf16, int store_y);
int m360_linear_heads(const float *x, long M, int ldx, const float *w_packed, const float *b_packed, int n_pad,
                      int k_pad, int act /* M360_ACT_SIGMOID */, float *y, int ldy, int store_y, const float *head_w,
                      int heads, float *head_part, m360_stream_t stream);
/* bf16 counterpart (x, w_packed, y: bf16 as for m360_linear_bf16; head_w, head_part: fp32).  Round 3: the head products of
 * the full 256-row tiles (widths 256 / 512 / 768 / 1024) are formed on the matrix pipe inside the layer's epilogue - the
 * packed bf16 row segments the epilogue holds are B fragments of v_mfma_f32_16x16x32_bf16, the head rows (as two bf16 terms)
 * the A fragments - from the bf16-ROUNDED activations, i.e. from what m360_nerf_finish_bf16 would read back;
 * the call writes m360_linear_heads_slots_bf16(n_pad, k_pad, mode, store_y) slots per row: 8 per 256 columns on the ping-pong
 * kernel (one partial sum per wave column group and 8-column half), 2 per 256 columns on the one-wave ring kernel (store_y = 0
 * and a contraction of its shape: one per 128-column wave tile) - pass that number to the *_finish_fused calls.
 * The bf16x3 form takes / writes [hi | lo] pair rows like m360_linear_bf16x3 and adds the lo activations' term. */
int m360_linear_heads_bf16(const void *x, long M, int ldx, const void *w_packed, const float *b_packed, int n_pad,
                           int k_pad, int act, void *y, int ldy, int store_y, const float *head_w, int heads,
                           float *head_part, m360_stream_t stream);
int m360_linear_heads_bf16x3(const void *x_hi_lo, long M, int ldx, const void *w_packed3, const float *b_packed, int n_pad,
                             int k_pad, int act, void *y_hi_lo, int ldy, int store_y, const float *head_w, int heads,
                             float *head_part, m360_stream_t stream);

/* m360_prop_finish_n / m360_nerf_finish whose head products of the samples [0, fused_rows) come from
 * head_part[fused_rows][slots][heads] (summed slot 0, 1, ... + bias) and of the remaining samples from the activation
 * rows `act` (fp32, or bf16 when act_bf16 != 0) as before.  fused_rows = 0 is exactly the unfused finisher. */
int m360_prop_finish_fused(const void *act, int act_bf16, int ld, const float *head_part, long fused_rows, int slots,
                           const float *head_w, const float *head_b, int k_pad, float density_bias,
                           const float *t_vals, const float *dirs, const float *u_rand, int B, int N, int num_out,
                           float resample_padding, float *weights, float *t_new, m360_stream_t stream);
int m360_nerf_finish_fused(const void *act, int act_bf16, int ld, const float *head_part, long fused_rows, int slots,
                           const float *head_w, const float *head_b, int k_pad, float density_bias, float rgb_padding,
                           const float *t_vals, const float *dirs, int B, int N, int white_bkgd, float *comp_rgb,
                           float *distance, float *acc, float *weights, m360_stream_t stream);

/* m360_nerf_finish_fused that also writes, in the same launch, the two tensors nerf_net.forward returns beside the composite
 * (model.py:194-196): t_vals_out[B,N+1] = t_vals + 1e-6 (what the in-place g() inside t_to_s leaves in the stored t_vals,
 * intern/parameterization.py:5-8,15-21) and s_vals_out[B,N+1] = t_to_s(t_vals, near, far) with near / far having gone through
 * g() `near_far_calls` times already (m360_t_to_s's near_calls = far_calls); either may be NULL.  Same arithmetic as m360_t_to_s. */
int m360_nerf_finish_outputs(const void *act, int act_bf16, int ld, const float *head_part, long fused_rows, int slots,
                             const float *head_w, const float *head_b, int k_pad, float density_bias, float rgb_padding,
                             const float *t_vals, const float *dirs, const float *near, const float *far, int near_far_calls,
                             int B, int N, int white_bkgd, float *comp_rgb, float *distance, float *acc, float *weights,
                             float *t_vals_out, float *s_vals_out, m360_stream_t stream);

/* ------------------------------------------------------------------ whole forward ----- */

typedef struct {
    const float *origins;    /* [B,3] */
    const float *directions; /* [B,3] */
    const float *viewdirs;   /* [B,3] */
    const float *radii;      /* [B]   */
    const float *near;       /* [B]   */
    const float *far;        /* [B]   */
} m360_rays_t; /* intern/ray.py:6 */

typedef struct {
    int in_ch;  /* 42 + 4*(viewdir_max_deg - viewdir_min_deg) */
    int in_pad; /* in_ch rounded up to 32 */
    int hp_pad; /* proposal width rounded up to 32 */
    int hn_pad; /* nerf width rounded up to 32 */
    const float *prop_w[4];   /* packed [hp_pad,in_pad], [hp_pad,hp_pad] x3 */
    const float *prop_b[4];   /* [hp_pad] */
    const float *prop_head_w; /* [hp_pad] */
    const float *prop_head_b; /* [1] */
    const float *nerf_w[8];   /* packed [hn_pad,in_pad], [hn_pad,hn_pad] x7 */
    const float *nerf_b[8];
    const float *nerf_head_w; /* [4,hn_pad]: final_density row, final_color rows */
    const float *nerf_head_b; /* [4] */
    int mlp_bf16; /* extension: 0 = fp32 MLP (default, the parity path); 1 = prop_w / nerf_w point to bf16
                     weights from m360_pack_linear_bf16 (pads multiples of 64), hidden activations
                     are bf16, accumulation / biases / heads stay fp32; 2 = "bf16x3": weights from
                     m360_pack_linear_bf16x3, hidden activations as [hi | lo] bf16 pairs (m360_linear_bf16x3).
                     The FIRST layers (prop_w[0] / nerf_w[0]) see more bits of the features than the hidden layers do (see "x6"
                     above): mode 1: weights from m360_pack_linear_bf16x3, features as [hi | lo] pairs (4 in_pad bytes per sample,
                     m360_linear_bf16x3_bf16out); mode 2: weights from m360_pack_linear_bf16x6, features as x6 rows (three bf16
                     terms per value, 12 in_pad bytes per sample, m360_linear_bf16_split) */
    int packed_layout; /* which packing prop_w / nerf_w follow: must be M360_PACKED_LAYOUT when mlp_bf16 != 0 (fp32: 0 is accepted too).
                     The first-layer packings of the bf16 modes changed in 0.1.0 (layout 2: [n_pad, 3 in_pad] / [n_pad, 6 in_pad] instead of
                     [n_pad, in_pad] / [n_pad, 3 in_pad]) and the kernels cannot see a buffer's size: a caller (or a packed file,
                     checkpoint.load_packed) of the older layout is refused instead of being read past its end */
} m360_model_t; /* packed form of the state_dict of model.py:43-53,131-158 */
#define M360_PACKED_LAYOUT 2

/* Per-call switches (m360_hyper_t.tuning; 0 = the defaults).  Every one leaves the results bit-identical (M360_TUNE_WGRAD_FORM0: up to fp32
 * summation order): they exist for A/B measurements on one box and for tests, and they are per CALL so that two threads, streams or models of
 * one process never see each other's choice. */
#define M360_TUNE_NO_HIDDEN_CHAIN 1u   /* bf16 mode: six launches instead of the one-launch chain for the hidden NeRF layers */
#define M360_TUNE_PLAIN_ROWS 2u        /* bf16 modes: plain instead of paired rows between the layers (which also means no chain) */
#define M360_TUNE_WGRAD_FORM0 4u       /* bf16 backward: the 8-wave weight-gradient kernel instead of the one-wave form */
#define M360_TUNE_CHAIN_COOPERATIVE 8u /* the chain through hipLaunchCooperativeKernel (co-residency asked of the runtime) */
#define M360_TUNE_CHAIN_UNGATED 16u    /* diagnostics build only (refused otherwise): no gated re-run behind the chain - A/B of its cost, results unchecked */

/* A caller-owned second stream with its fork / join events (bf16 backward: the ReLU mask of a layer's input gradient runs there beside that
 * layer's weight gradient, forked from and joined to the caller's stream inside the call - the caller still sees one stream).  Created for the
 * device that is current at m360_side_create; put into m360_hyper_t.side of the backward calls that may use it (NULL = everything on the
 * caller's stream: same weight gradients bit for bit, the lower layers' bias gradients the same sums in another fixed order).  One call at a
 * time per handle - use one per (device, stream) or per thread, like a workspace.  The library keeps no stream, event or handle of its own. */
typedef struct m360_side m360_side_t;
int m360_side_create(m360_side_t **out);
void m360_side_destroy(m360_side_t *side);

typedef struct m360_hyper {
    int num_samples;
    int viewdir_min_deg, viewdir_max_deg;
    int white_bkgd;
    float density_bias, rgb_padding, resample_padding;
    int num_samples_fine; /* extension: NeRF-stage samples per ray; 0 = num_samples (the reference's behaviour) */
    int norm_group_rays;  /* extension: > 0 = the batch is a run of render_image chunks (model.py:262-264) of this many
                             rays, each contracted with ITS OWN global norm (parameterization.py:25): one launch
                             sequence renders many chunks, bit-identical to launching them one by one; 0 = one norm
                             for the whole batch (the reference's forward) */
    void *prof;           /* optional m360_prof_t* (see "measurement" below); NULL = no timing */
    int rays_mutated;     /* 0 (default): rays.near / rays.far are the caller's ORIGINAL values and the +1e-6 offsets the
                             reference's in-place g() accumulates inside one (prop, nerf) forward pair are applied
                             numerically.  1: the caller mutates near / far physically like the reference does
                             (intern/parameterization.py:15-21: +1e-6 each after the proposal stage, so the NeRF stage's
                             t_to_s starts from the values it is handed); used by the mirrors'
                             mutate_like_reference mode to reproduce train.py's drift over its three pairs */
    int randomized;       /* `randomized=True` (model.py:27,112; the reference's CLI default, config.py:15) with the uniforms drawn INSIDE the
                             kernels (round 5): bit 0 = stratified jitter of the proposal samples (intern/ray.py:103-108), bit 1 = randomized
                             inverse CDF incl. its `u + u` (intern/ray.py:30-35).  A bit applies wherever the entry point's own t_rand /
                             u_rand tensor is NULL (m360_forward has none: a randomized model runs the fused forward); a tensor, when
                             given, wins.  0 = deterministic. */
    unsigned long long rng_seed;   /* Philox4x32-10 key: the torch device generator's seed */
    unsigned long long rng_offset; /* first 64 counter bits: the generator's offset / 4 at this call (the caller advances the generator by 4
                                      per call); element e of stream s (0 = t_rand, 1 = u_rand, [B, N + 1] row-major) draws
                                      philox(key, counter = (rng_offset, e, s)).x >> 8 as a 24-bit uniform in [0, 1) - m360_philox_uniform
                                      writes out exactly these numbers (tests replay them through the oracle) */
    unsigned tuning;      /* M360_TUNE_* bits, 0 = defaults: per-call A/B switches with bit-identical results */
    void *side;           /* optional m360_side_t* (see above): m360_prop_backward / m360_nerf_backward in the bf16 mode; NULL = one stream */
    long chain_debug_wait_ticks; /* test hooks of the layer chain, honoured by the DIAGNOSTICS build only (libm360_diag.so; the product library */
    int chain_debug_fault;       /* refuses non-zero values): bound of one wait in 100 MHz ticks (0 = the default 0.1 s); fault 1 = one workgroup
                                    reports a foreign XCD, 2 = every wave treats its first wait as run out (wrong rows, error set: what the gated
                                    re-run must repair) */
} m360_hyper_t; /* ctor arguments of model.py:203-215 */

typedef struct {
    float *rgb;      /* [B,3] */
    float *distance; /* [B]   */
    float *acc;      /* [B]   */
    /* optional (NULL to skip): */
    float *t_hat;    /* [B,N+1] proposal t_vals            (model.py:94)  */
    float *w_hat;    /* [B,N]   proposal weights           (model.py:94)  */
    float *t_vals;   /* [B,Nf+1] resampled t_vals + 1e-6   (model.py:194,196); Nf = fine sample count */
    float *fine_w;   /* [B,Nf]   nerf weights              (model.py:193) */
    float *s_vals;   /* [B,Nf+1]                           (model.py:196) */
} m360_outputs_t;

size_t m360_forward_workspace_bytes(int B, int N, const m360_model_t *model_host);
/* Every forward workspace starts with the 128-byte status block of the bf16 mode's layer chain (see m360_mlp_chain_bf16_safe above):
 * m360_workspace_init(workspace, stream) once after allocating it, m360_workspace_status(workspace, out5, stream) to read the counters.
 * A forward's OUTPUTS never need checking: a chain launch that reports an error is repaired on the device by the gated launches behind it. */

/* prop_net.forward, model.py:80-94 -> out->t_hat, out->w_hat (both required).
 * t_rand: optional uniforms [B,N+1] for randomized=True. */
int m360_prop_forward(const m360_rays_t *rays_host, const m360_model_t *model_host,
                      const m360_hyper_t *hyper_host, int B, const float *t_rand, float *t_hat,
                      float *w_hat, void *workspace, size_t workspace_bytes, m360_stream_t stream);

/* nerf_net.forward, model.py:163-200 (t_hat/w_hat from the proposal stage). */
int m360_nerf_forward(const m360_rays_t *rays_host, const m360_model_t *model_host,
                      const m360_hyper_t *hyper_host, int B, const float *t_hat, const float *w_hat,
                      const float *u_rand, const m360_outputs_t *out_host, void *workspace,
                      size_t workspace_bytes, m360_stream_t stream);

/* mipNeRF360.forward, model.py:247-252: both stages back to back on one stream. */
int m360_forward(const m360_rays_t *rays_host, const m360_model_t *model_host,
                 const m360_hyper_t *hyper_host, int B, const m360_outputs_t *out_host,
                 void *workspace, size_t workspace_bytes, m360_stream_t stream);

/* ------------------------------------------------------------------ training path (row f3) ---
 * Backward of the two stages with respect to the network parameters - what `loss.backward()` computes in
 * train.py:62,80.  As in the reference, the sample positions carry no gradient (resampling runs under
 * torch.no_grad(), intern/ray.py:136; train.py:70-71 detaches t_hat / w_hat), so the gradient stops at the MLP
 * inputs.  mlp_bf16 = 0 (fp32) or 1 (bf16, round 5: the tape holds bf16 layer outputs and the [hi | lo] feature rows, dz travels in bf16,
 * products accumulate in fp32; m360_mlp_transposed_t then carries m360_pack_linear_bf16_transposed packings); the bf16x3 mode is
 * forward-only.  Gradients are fp32 in the packed [n_pad, k_pad] layouts of the fp32 model struct (layer 0: [n_pad, in_pad] in every mode),
 * zero in the padding. */

typedef struct m360_mlp_transposed {
    const float *w_t[8]; /* layer l >= 1: m360_pack_linear_transposed of that layer's weight; [0] unused */
} m360_mlp_transposed_t;

typedef struct m360_mlp_grads {
    float *w[8];    /* [n_pad,k_pad] per layer (4 proposal / 8 NeRF layers) */
    float *b[8];    /* [n_pad] */
    float *head_w;  /* proposal [hp_pad]; NeRF [4,hn_pad]: density, r, g, b */
    float *head_b;  /* [1] / [4] */
} m360_mlp_grads_t;

/* per-ray backward of the finishers: from the gradients of the stage outputs to dz[B*N,ld] = gradient at the
 * pre-activation of the last hidden (sigmoid) layer, plus the head gradients.  act = that layer's output.
 * Replaces autograd through model.py:52,92-93,59-77 (proposal) and model.py:150-158,180-186 +
 * intern/ray.py:171-191 (NeRF; grad_distance / grad_acc / grad_weights / grad_rgb may each be NULL = zero). */
size_t m360_finish_backward_workspace_bytes(int B, int heads, int k_pad);
int m360_prop_finish_backward(const float *act, int ld, const float *head_w, const float *head_b, int k_pad,
                              float density_bias, const float *t_vals, const float *dirs, int B, int N,
                              const float *grad_weights, float *dz, float *grad_head_w, float *grad_head_b,
                              void *workspace, size_t workspace_bytes, m360_stream_t stream);
int m360_nerf_finish_backward(const float *act, int ld, const float *head_w, const float *head_b, int k_pad,
                              float density_bias, float rgb_padding, const float *t_vals, const float *dirs, int B,
                              int N, int white_bkgd, const float *grad_rgb, const float *grad_distance,
                              const float *grad_acc, const float *grad_weights, float *dz, float *grad_head_w,
                              float *grad_head_b, void *workspace, size_t workspace_bytes, m360_stream_t stream);

/* stage = 0 proposal (N = num_samples), 1 NeRF (N = number of fine intervals).  The tape keeps the sample
 * positions, the encoded features, every layer output and the head sums (1 / 4 floats per sample, as the forward's
 * finisher formed them: the backward starts from those instead of re-reading the last layer's rows) of ONE forward
 * for its backward.  Opaque: its size and layout belong to the library build that wrote it. */
size_t m360_train_tape_bytes(int B, int N, const m360_model_t *model_host, int stage);
size_t m360_backward_workspace_bytes(int B, int N, const m360_model_t *model_host, int stage);

/* m360_prop_forward / m360_nerf_forward that also fill the tape (workspace as for the plain forward). */
int m360_prop_forward_train(const m360_rays_t *rays_host, const m360_model_t *model_host,
                            const m360_hyper_t *hyper_host, int B, const float *t_rand, float *t_hat,
                            float *w_hat, void *tape, size_t tape_bytes, void *workspace,
                            size_t workspace_bytes, m360_stream_t stream);
int m360_nerf_forward_train(const m360_rays_t *rays_host, const m360_model_t *model_host,
                            const m360_hyper_t *hyper_host, int B, const float *t_hat, const float *w_hat,
                            const float *u_rand, const m360_outputs_t *out_host, void *tape, size_t tape_bytes,
                            void *workspace, size_t workspace_bytes, m360_stream_t stream);

/* d loss / d parameters of prop_net given grad_w_hat[B,N] = d loss / d w_hat (train.py:60-62). */
int m360_prop_backward(const m360_rays_t *rays_host, const m360_model_t *model_host,
                       const m360_mlp_transposed_t *wt_host, const m360_hyper_t *hyper_host, int B,
                       const void *tape, size_t tape_bytes, const float *grad_w_hat,
                       const m360_mlp_grads_t *grads_host, void *workspace, size_t workspace_bytes,
                       m360_stream_t stream);
/* d loss / d parameters of nerf_net given the gradients of comp_rgb[B,3], distance[B], acc[B] and
 * fine weights[B,N] (train.py:75-80 uses comp_rgb and the weights). */
int m360_nerf_backward(const m360_rays_t *rays_host, const m360_model_t *model_host,
                       const m360_mlp_transposed_t *wt_host, const m360_hyper_t *hyper_host, int B,
                       const void *tape, size_t tape_bytes, const float *grad_rgb, const float *grad_distance,
                       const float *grad_acc, const float *grad_weights, const m360_mlp_grads_t *grads_host,
                       void *workspace, size_t workspace_bytes, m360_stream_t stream);

/* ------------------------------------------------------------------ one batch over several devices ---
 * when ONE logical batch (e.g. BASELINE configs[4], 8192 rays) is split over ranks, the
 * contraction norm of parameterization.py:25 spans all ranks' rays.  Each rank computes the fp64 sum of squares
 * of its own un-contracted means (m360_mean_sumsq), the host all-reduces that one double per stage (RCCL) and
 * hands sqrt(sum) back as a device float.  Everything else is rank-local. */
int m360_mean_sumsq(const float *t_vals /*[B,N+1]*/, const float *directions, const float *radii, int B, int N,
                    double *sumsq /*device, 1*/, void *workspace, size_t workspace_bytes, m360_stream_t stream);
int m360_encode_features_ext_norm(const float *t_vals, const float *origins, const float *directions,
                                  const float *radii, const float *vdenc, int vd_ch, int B, int N, void *feat,
                                  int ld_feat, int bf16, const float *norm /*device, 1*/, void *workspace,
                                  size_t workspace_bytes, m360_stream_t stream);
/* proposal stage on given sample positions t_hat[B,N+1] (m360_sample_t) -> w_hat[B,N], t_new[B,n_fine+1] (or NULL) */
int m360_prop_forward_from_t(const m360_rays_t *rays_host, const m360_model_t *model_host,
                             const m360_hyper_t *hyper_host, int B, const float *t_hat, const float *norm,
                             float *w_hat, float *t_new, void *workspace, size_t workspace_bytes,
                             m360_stream_t stream);
/* NeRF stage on given resampled positions t_new[B,n_fine+1] */
int m360_nerf_forward_from_t(const m360_rays_t *rays_host, const m360_model_t *model_host,
                             const m360_hyper_t *hyper_host, int B, const float *t_new, const float *norm,
                             const m360_outputs_t *out_host, void *workspace, size_t workspace_bytes,
                             m360_stream_t stream);

/* ------------------------------------------------------------------ measurement ------ */

/* Optional HIP-event timing of the kernels of a stage driver (m360_forward, m360_prop_forward, m360_nerf_forward,
 * their *_train / *_from_t forms and the two backward drivers), recorded on the launch stream itself.  Not part of
 * the reference; used by bench.py's roofline.  The recorder is a CALLER-OWNED host object (no state lives in the
 * library): create one with `capacity` record slots, put it into m360_hyper_t.prof for the calls to be timed (NULL =
 * no timing), read the records afterwards.  m360_prof_read() synchronises on record i's stop event - the only call
 * in this library that blocks.  One recorder must not be used by two threads at once. */
typedef struct m360_prof m360_prof_t;
enum {
    M360_K_LINEAR = 0,      /* one fp32 layer   (M rows, n_pad, k_pad) */
    M360_K_LINEAR_BF16 = 1, /* one bf16 layer   */
    M360_K_ENCODE = 2,      /* sample -> contract -> IPE -> feature rows (M = B*N samples, n_pad = ld_feat, k_pad = bf16?) */
    M360_K_PROP_FINISH = 3, /* proposal head + weights + resample (M = B*N, n_pad = width, k_pad = bf16?) */
    M360_K_NERF_FINISH = 4, /* NeRF heads + composite             (M = B*N, n_pad = width, k_pad = bf16?) */
    M360_K_WGRAD = 5,
    M360_K_DGRAD = 6,
    M360_K_LINEAR_HEADS = 7 /* last hidden layer fused with the heads (k_pad < 0: bf16) */
};
m360_prof_t *m360_prof_create(int capacity); /* NULL on failure */
void m360_prof_destroy(m360_prof_t *prof);
int m360_prof_count(const m360_prof_t *prof);
int m360_prof_reset(m360_prof_t *prof);
int m360_prof_read(m360_prof_t *prof, int i, float *ms, int *kind, long *M, int *n_pad, int *k_pad);

#ifdef __cplusplus
}
#endif
#endif /* M360_H_ */
