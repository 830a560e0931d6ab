// y[M,Np] = act(x[M,Kp] * W^T + b) on the CDNA4 matrix cores with exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32: bitwise a k-ordered fmaf chain, 64 FLOP/clk/SIMD => 157.3 TFLOP/s).
// This kernel carries ~100 % of the path's FLOPs (model.py:43-53 and :131-148 of the reference).
//
// Tiling (wave64, 8 waves = 2 per SIMD, one workgroup per CU):
//   workgroup tile 256(M) x 256(N), K-step 32, waves arranged 2(M) x 4(N), wave tile 128 x 64
//   = 4 x 2 MFMA tiles of 32x32 -> 128 accumulator registers per lane.
//   A (activations, [M][K] k-contiguous) and B (PyTorch weight, [N][K] k-contiguous) tiles are
//   staged global -> registers -> LDS with 16-byte accesses, double-buffered in LDS, ONE barrier
//   per K-step; the global loads of step t+1 are issued before the 128 MFMAs of step t.
//   LDS rows are padded 128 B -> 144 B so every ds_read_b128 lane group hits 16 distinct
//   4-bank slots (conflict-free; MI355X LDS: 64 banks for b128).
//   The K index inside each group of 8 is permuted (lane half h takes k = 8g+4h+s for MFMA
//   step s) so one ds_read_b128 feeds four consecutive MFMAs for A and for B alike.
//   Workgroup ids are remapped so the 4 N-tiles of one M-tile run on the same XCD (shared L2).
#include <stdlib.h>
#include <atomic>

#include "m360_common.hip.h"
#include "m360_linear_persist.hip.h"
#include "m360_linear_bf16.hip.h"
#include "m360_linear_bf16_pp.hip.h"
#include "m360_linear_bf16_w16.hip.h"
#ifndef M360_W16_HEADS_ON
#define M360_W16_HEADS_ON 1  // fused-heads last layers of the rendering forward on the one-wave ring kernel (0: ping-pong kernel)
#endif
#ifndef M360_W16_K64
#define M360_W16_K64 1  // 64-deep bf16 first layers on the one-wave ring kernel (0: the first one-wave kernel)
#endif
#ifndef M360_W16_X3
#define M360_W16_X3 1  // bf16x3 hidden layers on the one-wave ring kernel (0: all on the ping-pong kernel)
#endif
#ifndef M360_W16_MIN_K
#define M360_W16_MIN_K 256  // narrowest contraction the bf16 hidden layers hand to the one-wave ring kernel
#endif
#include "m360_linear_tn.hip.h"
#include "m360_linear_tn_bf16.hip.h"
#include "m360_linear_tn_bf16_w.hip.h"
#ifdef M360_DIAG  // diagnostics build only: stamped twins of the product kernels + the two bf16 structures that lost the A/B
#include "diag/m360_diag.h"
#include "diag/m360_linear_bf16_sp.hip.h"
#include "diag/m360_linear_bf16_rg.hip.h"
#include "diag/m360_linear_bf16_w32.hip.h"
#endif
#include "m360_linear_hd.hip.h"

namespace m360 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 256, BN = 256, BK = 32;
constexpr int LDS_LD = BK + 4;  // floats per padded LDS row (144 B)
constexpr int kThreads = 512;
constexpr int TM = 4, TN = 2;   // 32x32 MFMA tiles per wave (M, N)

template <int ACT>
__device__ __forceinline__ float activate(float v) {
    if (ACT == M360_ACT_RELU) return relu_nanf_(v);
    if (ACT == M360_ACT_SIGMOID) return persist::act_fn<M360_ACT_SIGMOID>(v);  // the same hardware exp / rcp form as the persistent kernel (fp32: all three kernels give the same bits for a row)
    return v;
}

template <int ACT>
__global__ __launch_bounds__(kThreads, 2) void linear_f32_mfma_kernel(
    const float *__restrict__ X, long M, int ldx, const float *__restrict__ W,
    const float *__restrict__ bias, int Np, int Kp, float *__restrict__ Y, int ldy, int tiles_n,
    const float *__restrict__ aux = nullptr) {
    __shared__ float As[2][BM * LDS_LD];
    __shared__ float Bs[2][BN * LDS_LD];

    // ---- XCD-aware tile mapping: consecutive ids on one XCD walk the N-tiles of one M-tile
    const long tiles_m = (M + BM - 1) / BM;
    const long nwg = tiles_m * tiles_n;
    long tile_m, tile_n;
    {
        const long bid = blockIdx.x;
        const long full = (nwg / 8) * 8;  // ids covered by the bijective 8-way remap
        if (bid < full) {
            const long xcd = bid % 8, seq = bid / 8;        // seq-th workgroup of this XCD
            const long lin = xcd * (full / 8) + seq;        // contiguous id range per XCD
            tile_m = lin / tiles_n;
            tile_n = lin % tiles_n;
        } else {
            tile_m = bid / tiles_n;
            tile_n = bid % tiles_n;
        }
    }
    const long m0 = tile_m * BM;
    const int n0 = (int)tile_n * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;  // 2 x 4 waves
    const int l31 = lane & 31, h = lane >> 5;

    // ---- global -> register staging: thread handles float4 #c4 of rows r, r+64, r+128, r+192.
    // Uniform tile base (SGPRs) + small per-thread 32-bit offsets; tail rows are clamped (they
    // are computed but never stored).
    const long rows_left = M - m0;       // >= 1
    const int cols_left = Np - n0;       // >= 1
    const float *__restrict__ Xt = X + m0 * ldx;
    const float *__restrict__ Wt = W + (long)n0 * Kp;
    const int c4 = tid & 7, r0 = tid >> 3;
    int oa[4], ob[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + 64 * i;
        const int ra = r < rows_left ? r : (int)rows_left - 1;
        const int rb = r < cols_left ? r : cols_left - 1;
        oa[i] = ra * ldx + 4 * c4;
        ob[i] = rb * Kp + 4 * c4;
    }
    const int st_off = r0 * LDS_LD + 4 * c4;
    const int a_off = (wm * 128 + l31) * LDS_LD + 4 * h;
    const int b_off = (wn * 64 + l31) * LDS_LD + 4 * h;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define M360_LOAD_GLOBAL(k0)                                                   \
    do {                                                                       \
        ra0 = *reinterpret_cast<const float4 *>(Xt + oa[0] + (k0));            \
        ra1 = *reinterpret_cast<const float4 *>(Xt + oa[1] + (k0));            \
        ra2 = *reinterpret_cast<const float4 *>(Xt + oa[2] + (k0));            \
        ra3 = *reinterpret_cast<const float4 *>(Xt + oa[3] + (k0));            \
        rb0 = *reinterpret_cast<const float4 *>(Wt + ob[0] + (k0));            \
        rb1 = *reinterpret_cast<const float4 *>(Wt + ob[1] + (k0));            \
        rb2 = *reinterpret_cast<const float4 *>(Wt + ob[2] + (k0));            \
        rb3 = *reinterpret_cast<const float4 *>(Wt + ob[3] + (k0));            \
    } while (0)
#define M360_STORE_LDS(buf)                                                              \
    do {                                                                                 \
        *reinterpret_cast<float4 *>(&As[buf][st_off]) = ra0;                             \
        *reinterpret_cast<float4 *>(&As[buf][st_off + 64 * LDS_LD]) = ra1;               \
        *reinterpret_cast<float4 *>(&As[buf][st_off + 128 * LDS_LD]) = ra2;              \
        *reinterpret_cast<float4 *>(&As[buf][st_off + 192 * LDS_LD]) = ra3;              \
        *reinterpret_cast<float4 *>(&Bs[buf][st_off]) = rb0;                             \
        *reinterpret_cast<float4 *>(&Bs[buf][st_off + 64 * LDS_LD]) = rb1;               \
        *reinterpret_cast<float4 *>(&Bs[buf][st_off + 128 * LDS_LD]) = rb2;              \
        *reinterpret_cast<float4 *>(&Bs[buf][st_off + 192 * LDS_LD]) = rb3;              \
    } while (0)

    // 128 MFMAs on one staged K-step
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const float *Ab = &As[buf][a_off];
        const float *Bb = &Bs[buf][b_off];
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            float4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4 *>(Ab + i * 32 * LDS_LD + 8 * g);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const float4 *>(Bb + j * 32 * LDS_LD + 8 * g);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float av = s == 0 ? a[i].x : s == 1 ? a[i].y : s == 2 ? a[i].z : a[i].w;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const float bv = s == 0 ? b[j].x : s == 1 ? b[j].y : s == 2 ? b[j].z : b[j].w;
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
    };

    const int ksteps = Kp / BK;
    M360_LOAD_GLOBAL(0);
    M360_STORE_LDS(0);
    __syncthreads();
    int buf = 0;
    for (int t = 0; t < ksteps - 1; ++t) {
        M360_LOAD_GLOBAL((t + 1) * BK);  // in flight behind the MFMAs below
        __builtin_amdgcn_sched_barrier(0);  // keep the issue point: hipcc otherwise sinks the loads
        compute(buf);
        __builtin_amdgcn_sched_barrier(0);
        M360_STORE_LDS(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    compute(buf);
#undef M360_LOAD_GLOBAL
#undef M360_STORE_LDS

    // ---- epilogue: bias + activation; lane holds column (n) l31, rows (r&3)+8(r>>2)+4h
    const bool interior = rows_left >= BM && cols_left >= BN;  // wave-uniform
    float *__restrict__ Yt = Y + m0 * ldy + n0;
    if (ACT == M360_ACT_RELU_MASK) {  // backward of ReLU: no bias, keep where the forward output (aux, same ld) was > 0
        const float *__restrict__ At = aux + m0 * ldy + n0;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn * 64 + j * 32 + l31;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int rbase = wm * 128 + i * 32 + 4 * h;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (col < cols_left && row < rows_left)
                        Yt[(long)row * ldy + col] = At[(long)row * ldy + col] > 0.0f ? acc[i][j][r] : 0.0f;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = wn * 64 + j * 32 + l31;
        const bool col_ok = col < cols_left;
        const float bj = col_ok ? bias[n0 + col] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rbase = wm * 128 + i * 32 + 4 * h;
            if (interior) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Yt[(long)(rbase + (r & 3) + 8 * (r >> 2)) * ldy + col] = activate<ACT>(acc[i][j][r] + bj);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (col_ok && row < rows_left) Yt[(long)row * ldy + col] = activate<ACT>(acc[i][j][r] + bj);
                }
            }
        }
    }
}

constexpr int kZeroBias = 8192;
__device__ float g_zero_bias[kZeroBias];  // zero-initialised

__global__ void pack_linear_kernel(const float *__restrict__ w, const float *__restrict__ b, int n_out,
                                   int k_in, int n_pad, int k_pad, float *__restrict__ wp,
                                   float *__restrict__ bp) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)n_pad * k_pad) {
        const int n = (int)(idx / k_pad), k = (int)(idx % k_pad);
        // NaN parameters are packed as +NaN (0x7FC00000): the ReLU epilogues keep exactly the NaNs with a clear sign bit
        // (relu_nanf_), and a checkpoint may hold -NaN (x86's 0/0) where nn.ReLU propagates every NaN
        wp[idx] = (n < n_out && k < k_in) ? canon_nanf_(w[(long)n * k_in + k]) : 0.0f;
    }
    if (bp != nullptr && idx < n_pad) bp[idx] = (b != nullptr && idx < n_out) ? canon_nanf_(b[idx]) : 0.0f;
}

int launch_colsum(const float *in, long R, int C, int ld, float *scratch, int slices, float *out, hipStream_t st) {
    if (slices < 1) slices = 1;
    long per_slice = (R + slices - 1) / slices;
    if (per_slice < 1) per_slice = 1;
    const dim3 block(tn::kColWaves * kWave);
    hipLaunchKernelGGL(tn::colsum_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)slices), block, 0, st, in, R, C, ld, per_slice, scratch);
    hipLaunchKernelGGL(tn::colsum_kernel, dim3((unsigned)((C + 63) / 64), 1), block, 0, st, scratch, (long)slices, C, C, (long)slices, out);
    return check_launch("colsum");
}

}  // namespace m360

using namespace m360;

// CU count of the CURRENT device (the persistent kernels launch one workgroup per CU); cached per device
static int cu_count() {
    // a memo of an immutable hardware property, not state: zero-initialised, relaxed atomics (two threads at worst both query and store the same number)
    static std::atomic<int> cached[64];
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    if (dev >= 0 && dev < 64) {
        const int c = cached[dev].load(std::memory_order_relaxed);
        if (c > 0) return c;
    }
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    if (dev >= 0 && dev < 64) cached[dev].store(n, std::memory_order_relaxed);
    return n;
}

extern "C" {

int m360_pack_linear(const float *w, const float *b, int n_out, int k_in, int n_pad, int k_pad,
                     float *w_packed, float *b_packed, m360_stream_t stream) {
    if (!w || !w_packed || n_out < 1 || k_in < 1 || n_pad < n_out || k_pad < k_in)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_pack_linear: bad argument (n_out=%d k_in=%d n_pad=%d k_pad=%d)", n_out, k_in, n_pad, k_pad);
    const long n = (long)n_pad * k_pad;
    hipLaunchKernelGGL(pack_linear_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, b, n_out, k_in, n_pad, k_pad, w_packed, b_packed);
    return check_launch("pack_linear");
}

// Half-tile kernel (m360_linear_hd.hip.h) on M rows (a multiple of 128) of a 256-multiple width, bias + {none, ReLU}.
static int launch_linear_hd(const float *x, long M, int ldx, const float *w_packed, const float *b_packed, int n_pad, int k_pad,
                            int act, float *y, int ldy, unsigned *queue, hipStream_t st) {
    const int cus = cu_count();
    if (cus <= 0) return fail(M360_ERR_NO_DEVICE, "m360_linear: no HIP device");
    const long nt = (M / hd::BM) * (n_pad / hd::BN);
    if (nt > 0x7fffffffL - 4L * cus) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear: grid too large");
    dim3 grid((unsigned)(nt < cus ? nt : cus)), block(hd::kThreads);
    // with a queue word: all but the last max(2, 1/16) of a workgroup's share of tiles stay static (XCD-aware order), the
    // rest is handed out by ticket; launches of fewer than 4 tiles per workgroup have nothing to balance
    const long share = nt / cus;
    int n_static = 0;
    if (queue && share >= 4) n_static = (int)(share - (share / 16 > 2 ? share / 16 : 2));
    else queue = nullptr;
    const bool even = (k_pad / hd::BK) % 2 == 0;  // static LDS stages: no vector address updates in the K loop
#define M360_HD_LAUNCH(A, E) hipLaunchKernelGGL((hd::linear_f32_hd_kernel<A, E>), grid, block, 0, st, x, M, ldx, w_packed, b_packed, n_pad, k_pad, y, ldy, n_pad / hd::BN, (int)nt, queue, n_static)
    if (act == M360_ACT_RELU) {
        if (even) M360_HD_LAUNCH(M360_ACT_RELU, true); else M360_HD_LAUNCH(M360_ACT_RELU, false);
    } else {
        if (even) M360_HD_LAUNCH(M360_ACT_NONE, true); else M360_HD_LAUNCH(M360_ACT_NONE, false);
    }
#undef M360_HD_LAUNCH
    return check_launch("linear_hd");
}

// Which of the two persistent kernels takes the full tiles of a layer (both give the same bits).  The half-tile kernel hides
// the epilogue of a tile behind the next tile's matrix work, balances the last round of tiles better and leaves at most 127
// ragged rows: it takes every layer it can express (bias + {none, ReLU}, width a multiple of 256, contraction >= 64).
// Whole forward at BASELINE configs[1]: 54.42 ms against 54.61 ms with the 256 x 256 kernel (profiles/r02/bench_kernel_ab.txt).
#ifdef M360_DIAG
// diagnostics build only (A/B of the two kernels): 0 = the rule, 1 = 256 x 256, 2 = half tiles; M360_DIAG_FORCE_KERNEL presets it
static int g_diag_force_kernel = [] { const char *e = getenv("M360_DIAG_FORCE_KERNEL"); return e ? atoi(e) : 0; }();
static int g_diag_stagger = [] { const char *e = getenv("M360_DIAG_STAGGER"); return e ? atoi(e) : 0; }();  // ring kernel, variant 140: start stagger (x ~1 k cycles per row-block phase)
#endif
static bool prefer_half_tiles(long M, int n_pad, int k_pad, int act) {
    if ((act != M360_ACT_NONE && act != M360_ACT_RELU) || n_pad % hd::BN != 0 || n_pad > hd::kMaxBias || k_pad < 2 * hd::BK || M < hd::BM) return false;
#ifdef M360_DIAG
    if (g_diag_force_kernel == 1) return false;
#endif
    return true;
}

static int launch_linear(const float *x, long M, int ldx, const float *w_packed, const float *b_packed, int n_pad,
                         int k_pad, int act, float *y, int ldy, const float *aux, m360_stream_t stream, unsigned *queue = nullptr) {
    if ((uintptr_t)queue & 3) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_balanced: tile_queue must be a 4-byte aligned device pointer");
    if (!x || !w_packed || (!b_packed && act != M360_ACT_RELU_MASK) || !y || M < 0) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear: null pointer or negative M");
    if (act == M360_ACT_RELU_MASK && (!aux || ((uintptr_t)aux & 15))) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear: the ReLU-mask epilogue needs a 16-byte aligned mask source");
    if (n_pad < 1 || k_pad < BK || k_pad % BK != 0 || ldx < k_pad || ldy < n_pad || ldx % 4 != 0 || ldy % 4 != 0)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear: k_pad=%d must be a positive multiple of %d, ldx=%d >= k_pad, ldy=%d >= n_pad=%d, both multiples of 4", k_pad, BK, ldx, ldy, n_pad);
    if (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)b_packed | (uintptr_t)y) & 15) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear: x, w_packed, b_packed and y must be 16-byte aligned");
    if (M == 0) return M360_OK;
    const long tiles_m = (M + BM - 1) / BM;
    const int tiles_n = (n_pad + BN - 1) / BN;
    const long nwg = tiles_m * tiles_n;
    if (nwg > 0x7fffffffL) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear: grid too large");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (act != M360_ACT_NONE && act != M360_ACT_RELU && act != M360_ACT_SIGMOID && act != M360_ACT_RELU_MASK) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear: unknown activation %d", act);
    // Full tiles go to one of the two persistent LDS-DMA kernels when the width is a multiple of 256; ragged rows (and
    // narrow layers) go to the workgroup-per-tile kernel.  All three produce bit-identical results.
    const int cus = cu_count();
    if (cus <= 0) return fail(M360_ERR_NO_DEVICE, "m360_linear: no HIP device");
    long M_full = 0;
    if (prefer_half_tiles(M, n_pad, k_pad, act)) {
        M_full = (M / hd::BM) * hd::BM;
        const int rc = launch_linear_hd(x, M_full, ldx, w_packed, b_packed, n_pad, k_pad, act, y, ldy, queue, st);
        if (rc != M360_OK) return rc;
    } else if (n_pad % persist::BN == 0 && M >= persist::BM) {
        M_full = (M / persist::BM) * persist::BM;
        const long nt = (M_full / persist::BM) * (n_pad / persist::BN);
        const int ntiles = (int)nt;
        const bool even_k = (k_pad / persist::BK) % 2 == 0;  // static LDS buffers (sigmoid and masked epilogues: the layers this kernel still serves)
        dim3 grid((unsigned)(nt < cus ? nt : cus)), block(persist::kThreads);
        switch (act) {
            case M360_ACT_NONE: hipLaunchKernelGGL(persist::linear_f32_mfma_persist_kernel<M360_ACT_NONE>, grid, block, 0, st, x, M_full, ldx, w_packed, b_packed, n_pad, k_pad, y, ldy, n_pad / persist::BN, ntiles); break;
            case M360_ACT_RELU: hipLaunchKernelGGL(persist::linear_f32_mfma_persist_kernel<M360_ACT_RELU>, grid, block, 0, st, x, M_full, ldx, w_packed, b_packed, n_pad, k_pad, y, ldy, n_pad / persist::BN, ntiles); break;
            case M360_ACT_RELU_MASK:
                if (even_k) hipLaunchKernelGGL((persist::linear_f32_mfma_persist_kernel<M360_ACT_RELU_MASK, false, 0, true, true>), grid, block, 0, st, x, M_full, ldx, w_packed, b_packed, n_pad, k_pad, y, ldy, n_pad / persist::BN, ntiles, aux);
                else hipLaunchKernelGGL(persist::linear_f32_mfma_persist_kernel<M360_ACT_RELU_MASK>, grid, block, 0, st, x, M_full, ldx, w_packed, b_packed, n_pad, k_pad, y, ldy, n_pad / persist::BN, ntiles, aux);
                break;
            default:
                if (even_k) hipLaunchKernelGGL((persist::linear_f32_mfma_persist_kernel<M360_ACT_SIGMOID, false, 0, true, true>), grid, block, 0, st, x, M_full, ldx, w_packed, b_packed, n_pad, k_pad, y, ldy, n_pad / persist::BN, ntiles);
                else hipLaunchKernelGGL(persist::linear_f32_mfma_persist_kernel<M360_ACT_SIGMOID>, grid, block, 0, st, x, M_full, ldx, w_packed, b_packed, n_pad, k_pad, y, ldy, n_pad / persist::BN, ntiles);
                break;
        }
    }
    if (M > M_full) {
        const float *xt = x + M_full * ldx;
        float *yt = y + M_full * ldy;
        const long Mt = M - M_full;
        const long nwg_t = ((Mt + BM - 1) / BM) * tiles_n;
        dim3 grid((unsigned)nwg_t), block(kThreads);
        switch (act) {
            case M360_ACT_NONE: hipLaunchKernelGGL(linear_f32_mfma_kernel<M360_ACT_NONE>, grid, block, 0, st, xt, Mt, ldx, w_packed, b_packed, n_pad, k_pad, yt, ldy, tiles_n); break;
            case M360_ACT_RELU: hipLaunchKernelGGL(linear_f32_mfma_kernel<M360_ACT_RELU>, grid, block, 0, st, xt, Mt, ldx, w_packed, b_packed, n_pad, k_pad, yt, ldy, tiles_n); break;
            case M360_ACT_RELU_MASK: hipLaunchKernelGGL(linear_f32_mfma_kernel<M360_ACT_RELU_MASK>, grid, block, 0, st, xt, Mt, ldx, w_packed, b_packed, n_pad, k_pad, yt, ldy, tiles_n, aux + M_full * ldy); break;
            default: hipLaunchKernelGGL(linear_f32_mfma_kernel<M360_ACT_SIGMOID>, grid, block, 0, st, xt, Mt, ldx, w_packed, b_packed, n_pad, k_pad, yt, ldy, tiles_n); break;
        }
    }
    return check_launch("linear");
}

int m360_linear(const float *x, long M, int ldx, const float *w_packed, const float *b_packed,
                int n_pad, int k_pad, int act, float *y, int ldy, m360_stream_t stream) {
    if (act == M360_ACT_RELU_MASK) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear: use m360_linear_dgrad for the masked epilogue");
    return launch_linear(x, M, ldx, w_packed, b_packed, n_pad, k_pad, act, y, ldy, nullptr, stream);
}

int m360_linear_balanced(const float *x, long M, int ldx, const float *w_packed, const float *b_packed,
                         int n_pad, int k_pad, int act, float *y, int ldy, unsigned *tile_queue, m360_stream_t stream) {
    if (act == M360_ACT_RELU_MASK) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_balanced: use m360_linear_dgrad for the masked epilogue");
    return launch_linear(x, M, ldx, w_packed, b_packed, n_pad, k_pad, act, y, ldy, nullptr, stream, tile_queue);
}

#ifndef M360_X3_MODE
#define M360_X3_MODE 2  // 2 = the three products of a 64-deep block share operand tiles (product); 1 = one 3K-deep contraction (A/B builds)
#endif
// ---- last hidden layer of a stage fused with its heads
long m360_linear_heads_fused_rows(long M, int n_pad, int bf16) {
    if (bf16) {  // bf16 / bf16x3: the ping-pong kernel's MFMA-side heads (full 256-row tiles, widths of 256..1024 in steps of 256)
        if (M < 0 || n_pad < pp16::BN || n_pad % pp16::BN != 0 || n_pad > pp16::kHeadMaxN) return 0;
        return (M / pp16::BM) * pp16::BM;
    }
    if (M < 0 || n_pad < persist::BN || n_pad % persist::BN != 0 || n_pad > persist::kHeadMaxN) return 0;
    return (M / persist::BM) * persist::BM;  // full 256-row tiles of a 256-multiple width <= 1024
}

int m360_linear_heads_slots(int n_pad, int bf16) {  // partial sums per row: fp32 one per 128-column wave tile, bf16 at most one per 32 columns
    return n_pad >= persist::BN ? (bf16 ? 8 : 2) * (n_pad / persist::BN) : 0;
}

// which kernel forms the fused heads of a bf16 / bf16x3 last layer: the one-wave ring kernel (rendering forward: the layer's own
// output is not kept) when the contraction has its shape, else the ping-pong kernel
static bool heads_on_ring(int k_pad, int x3, int store_y) {
    // (store_y - the tape-keeping forward of the bf16 training path - since round 5 in the plain bf16 form: KEEP_Y; bf16x3 is forward-only)
    return M360_W16_HEADS_ON && !(store_y && x3) && (x3 ? (k_pad % w16::BKS == 0 && k_pad >= 2 * w16::BKS) : (k_pad % (2 * w16::BKS) == 0 && k_pad >= 4 * w16::BKS));
}

// paired rows (m360.h: M360_ROWS_PAIRED_IN / _OUT): the layout only the one-wave ring kernel reads and writes - does the call `kind`
// put the full tiles of an n_pad-wide layer with this contraction on it?  (The predicates of the dispatchers below, in one place.)
int m360_linear_bf16_rows_pairable(int kind, int n_pad, int k_pad) {
    if (n_pad < w16::BN || n_pad % w16::BN != 0 || k_pad < w16::BKS || k_pad % w16::BKS != 0) return 0;
    switch (kind) {
        case M360_PAIRABLE_LINEAR:
            if (k_pad == w16::BKS) return M360_W16_K64 && n_pad <= w16::kMaxBias;
            return n_pad <= pp16::kMaxBias && k_pad % (2 * w16::BKS) == 0 && k_pad >= M360_W16_MIN_K && k_pad >= 2 * pp16::BK;
        case M360_PAIRABLE_X3:
        case M360_PAIRABLE_X3_BF16OUT: return M360_W16_X3 && n_pad <= pp16::kMaxBias;
        case M360_PAIRABLE_SPLIT: return n_pad <= w16::kMaxBias && k_pad % (2 * w16::BKS) == 0 && k_pad >= M360_W16_MIN_K;
        case M360_PAIRABLE_HEADS: return n_pad <= pp16::kHeadMaxN && heads_on_ring(k_pad, 0, 0);
        case M360_PAIRABLE_HEADS_X3: return n_pad <= pp16::kHeadMaxN && heads_on_ring(k_pad, 1, 0);
        default: return 0;
    }
}

int m360_linear_heads_slots_bf16(int n_pad, int k_pad, int bf16, int store_y) {  // slots the call with these arguments writes
    if (n_pad < pp16::BN) return 0;
    return (heads_on_ring(k_pad, bf16 == 2, store_y) ? 2 : 8) * (n_pad / pp16::BN);
}

int m360_linear_heads(const float *x, long M, int ldx, const float *w_packed, const float *b_packed, int n_pad, int k_pad,
                      int act, float *y, int ldy, int store_y, const float *head_w, int heads, float *head_part,
                      m360_stream_t stream) {
    if (heads != 1 && heads != 4) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_heads: heads=%d (1 or 4)", heads);
    if (act != M360_ACT_SIGMOID) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_heads: the last hidden layer is a sigmoid layer (model.py:50,146), act=%d", act);
    if (!x || !w_packed || !b_packed || !y || !head_w || M < 0) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_heads: null pointer or negative M");
    const long M_fused = m360_linear_heads_fused_rows(M, n_pad, 0);
    if (M_fused > 0) {
        if (!head_part || ((uintptr_t)head_part & 15) || ((uintptr_t)head_w & 15)) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_heads: head_part / head_w must be 16-byte aligned device pointers");
        if (k_pad < BK || k_pad % BK != 0 || ldx < k_pad || ldy < n_pad || ldx % 4 != 0 || ldy % 4 != 0)
            return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_heads: k_pad=%d must be a positive multiple of %d, ldx=%d >= k_pad, ldy=%d >= n_pad=%d, both multiples of 4", k_pad, BK, ldx, ldy, n_pad);
        if (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)b_packed | (uintptr_t)y) & 15) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_heads: x, w_packed, b_packed and y must be 16-byte aligned");
        const int cus = cu_count();
        if (cus <= 0) return fail(M360_ERR_NO_DEVICE, "m360_linear_heads: no HIP device");
        const long nt = (M_fused / persist::BM) * (n_pad / persist::BN);
        dim3 grid((unsigned)(nt < cus ? nt : cus)), block(persist::kThreads);
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
        const int tn = n_pad / persist::BN;
#define M360_LAUNCH_HEADS(H, SY)                                                                                              \
    do {                                                                                                                       \
        if ((k_pad / persist::BK) % 2 == 0) hipLaunchKernelGGL((persist::linear_f32_mfma_persist_kernel<M360_ACT_SIGMOID, false, H, SY, true>), grid, block, 0, st, x, M_fused, ldx, w_packed, b_packed, n_pad, k_pad, y, ldy, tn, (int)nt, nullptr, head_w, head_part); \
        else hipLaunchKernelGGL((persist::linear_f32_mfma_persist_kernel<M360_ACT_SIGMOID, false, H, SY, false>), grid, block, 0, st, x, M_fused, ldx, w_packed, b_packed, n_pad, k_pad, y, ldy, tn, (int)nt, nullptr, head_w, head_part); \
    } while (0)
        if (heads == 1) {
            if (store_y) M360_LAUNCH_HEADS(1, true); else M360_LAUNCH_HEADS(1, false);
        } else {
            if (store_y) M360_LAUNCH_HEADS(4, true); else M360_LAUNCH_HEADS(4, false);
        }
#undef M360_LAUNCH_HEADS
        const int rc = check_launch("linear_heads");
        if (rc != M360_OK) return rc;
    }
    if (M > M_fused)  // ragged tail rows / widths the fused epilogue does not take: the plain layer; the finisher reads y there
        return launch_linear(x + M_fused * ldx, M - M_fused, ldx, w_packed, b_packed, n_pad, k_pad, act, y + M_fused * ldy, ldy, nullptr, stream);
    return M360_OK;
}

// bf16 (x3 = 0) and bf16x3 (x3 = 1) last hidden layer + heads: the full 256-row tiles on the ping-pong kernel with its MFMA-side
// head products (y is written only when store_y), the ragged tail rows on the plain layer (the finisher reads y there)
static int linear_heads_bf16_any(const void *x, long M, int ldx, const void *w_packed, const float *b_packed, int n_pad,
                                 int k_pad, int act, void *y, int ldy, int store_y, const float *head_w, int heads,
                                 float *head_part, m360_stream_t stream, int x3) {
    const char *who = x3 ? "m360_linear_heads_bf16x3" : "m360_linear_heads_bf16";
    if (heads != 1 && heads != 4) return fail(M360_ERR_INVALID_ARGUMENT, "%s: heads=%d (1 or 4)", who, heads);
    const int layout = act & M360_ACT_FLAGS_MASK;  // paired INPUT rows (m360.h): where the one-wave ring kernel forms the heads
    act &= ~M360_ACT_FLAGS_MASK;
    if (act != M360_ACT_SIGMOID) return fail(M360_ERR_INVALID_ARGUMENT, "%s: the last hidden layer is a sigmoid layer, act=%d", who, act);
    if (layout && (layout != M360_ROWS_PAIRED_IN || store_y || !m360_linear_bf16_rows_pairable(x3 ? M360_PAIRABLE_HEADS_X3 : M360_PAIRABLE_HEADS, n_pad, k_pad)))
        return fail(M360_ERR_INVALID_ARGUMENT, "%s: paired rows: input only, store_y = 0, a shape of the one-wave ring kernel (n_pad=%d k_pad=%d)", who, n_pad, k_pad);
    const int xin = layout ? 1 : 0;
    if (!x || !w_packed || !b_packed || !y || !head_w || M < 0) return fail(M360_ERR_INVALID_ARGUMENT, "%s: null pointer or negative M", who);
    const int xm = x3 ? 2 : 1;  // row-length multiplier of the [hi | lo] layout
    const long M_fused = m360_linear_heads_fused_rows(M, n_pad, 1);
    // callers (and the finishers) take the fused row count from m360_linear_heads_fused_rows(M, n_pad, bf16), which does not see the
    // contraction: a shape whose tiles this call could not fuse is an error, never a silently different row count
    if (M_fused > 0 && !x3 && k_pad < 2 * pp16::BK)
        return fail(M360_ERR_INVALID_ARGUMENT, "%s: k_pad=%d < %d with full %d-row tiles of a %d-wide layer: no kernel fuses the heads of a single 64-deep K-step (use m360_linear_bf16 + the unfused finisher)", who, k_pad, 2 * pp16::BK, pp16::BM, n_pad);
    if (M_fused > 0) {
        if (!head_part || ((uintptr_t)head_part & 15) || ((uintptr_t)head_w & 15)) return fail(M360_ERR_INVALID_ARGUMENT, "%s: head_part / head_w must be 16-byte aligned device pointers", who);
        if (k_pad < pbf16::BK || k_pad % pbf16::BK != 0 || ldx < xm * k_pad || ldy < xm * n_pad || ldx % 8 != 0 || ldy % 8 != 0)
            return fail(M360_ERR_INVALID_ARGUMENT, "%s: k_pad=%d must be a positive multiple of %d, ldx=%d, ldy=%d", who, k_pad, pbf16::BK, ldx, ldy);
        if (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)b_packed | (uintptr_t)y) & 15) return fail(M360_ERR_INVALID_ARGUMENT, "%s: pointers must be 16-byte aligned", who);
        const int cus = cu_count();
        if (cus <= 0) return fail(M360_ERR_NO_DEVICE, "%s: no HIP device", who);
        const long nt = (M_fused / pp16::BM) * (n_pad / pp16::BN);
        dim3 grid((unsigned)(nt < cus ? nt : cus)), block(pp16::kThreads);
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
        const __bf16 *xb = static_cast<const __bf16 *>(x), *wb = static_cast<const __bf16 *>(w_packed);
        __bf16 *yb = static_cast<__bf16 *>(y);
        const int kk = x3 ? 3 * k_pad : k_pad, tn = n_pad / pp16::BN;
#define M360_PP_HEADS(X3M, H, SY) hipLaunchKernelGGL((pp16::linear_bf16_pp_kernel<M360_ACT_SIGMOID, false, X3M, H, SY>), grid, block, 0, st, xb, M_fused, ldx, wb, b_packed, n_pad, kk, yb, ldy, tn, (int)nt, head_w, head_part)
#define M360_W16_HEADS(X3B, H) hipLaunchKernelGGL((w16::linear_bf16_w16_kernel<M360_ACT_SIGMOID, 0, false, X3B, false, H>), grid, dim3(w16::kThreads), 0, st, xb, M_fused, ldx, wb, b_packed, n_pad, kk, yb, ldy, tn, (int)nt, head_w, head_part, xin)
        // the rendering forward (the layer's own output is not kept): the one-wave ring kernel - its exposed sigmoid epilogue costs
        // less than its K loop wins (0.85 against 1.08-1.12 ms for the 1024^2 NeRF layer); with store_y (the tape-keeping forward of the bf16
        // training path) its KEEP_Y form: 32 row stores + 8 partial-sum stores per tile (the ping-pong kernel's store_y forms spill 44-75
        // registers: 1.7 ms for that layer, 0.27 ms for the proposal net's)
        const bool ring = heads_on_ring(k_pad, x3, store_y);
#define M360_W16_HEADS_KEEP(H) hipLaunchKernelGGL((w16::linear_bf16_w16_kernel<M360_ACT_SIGMOID, 0, false, false, false, H, false, false, false, false, true>), grid, dim3(w16::kThreads), 0, st, xb, M_fused, ldx, wb, b_packed, n_pad, kk, yb, ldy, tn, (int)nt, head_w, head_part, xin)
        if (ring && store_y) {
            if (heads == 1) M360_W16_HEADS_KEEP(1); else M360_W16_HEADS_KEEP(4);
        } else if (ring) {
            if (x3) { if (heads == 1) M360_W16_HEADS(true, 1); else M360_W16_HEADS(true, 4); }
            else { if (heads == 1) M360_W16_HEADS(false, 1); else M360_W16_HEADS(false, 4); }
        } else if (x3) {
            if (heads == 1) { if (store_y) M360_PP_HEADS(M360_X3_MODE, 1, true); else M360_PP_HEADS(M360_X3_MODE, 1, false); }
            else { if (store_y) M360_PP_HEADS(M360_X3_MODE, 4, true); else M360_PP_HEADS(M360_X3_MODE, 4, false); }
        } else {
            if (heads == 1) { if (store_y) M360_PP_HEADS(0, 1, true); else M360_PP_HEADS(0, 1, false); }
            else { if (store_y) M360_PP_HEADS(0, 4, true); else M360_PP_HEADS(0, 4, false); }
        }
#undef M360_PP_HEADS
#undef M360_W16_HEADS
#undef M360_W16_HEADS_KEEP
        const int rc = check_launch(who);
        if (rc != M360_OK) return rc;
    }
    if (M > M_fused) {  // ragged tail rows / shapes the fused epilogue does not take: the plain layer; the finisher reads y there
        const char *xt = static_cast<const char *>(x) + (size_t)M_fused * ldx * 2;
        char *yt = static_cast<char *>(y) + (size_t)M_fused * ldy * 2;
        if (x3) return m360_linear_bf16x3(xt, M - M_fused, ldx, w_packed, b_packed, n_pad, k_pad, act, yt, ldy, stream);
        return m360_linear_bf16(xt, M - M_fused, ldx, w_packed, b_packed, n_pad, k_pad, act, yt, ldy, stream);
    }
    return M360_OK;
}

int m360_linear_heads_bf16(const void *x, long M, int ldx, const void *w_packed, const float *b_packed, int n_pad,
                           int k_pad, int act, void *y, int ldy, int store_y, const float *head_w, int heads,
                           float *head_part, m360_stream_t stream) {
    return linear_heads_bf16_any(x, M, ldx, w_packed, b_packed, n_pad, k_pad, act, y, ldy, store_y, head_w, heads, head_part, stream, 0);
}

int m360_linear_heads_bf16x3(const void *x, long M, int ldx, const void *w_packed3, const float *b_packed, int n_pad,
                             int k_pad, int act, void *y, int ldy, int store_y, const float *head_w, int heads,
                             float *head_part, m360_stream_t stream) {
    return linear_heads_bf16_any(x, M, ldx, w_packed3, b_packed, n_pad, k_pad, act, y, ldy, store_y, head_w, heads, head_part, stream, 1);
}

// ---- training path: input gradient, weight gradient, transposed packing
int m360_linear_dgrad(const float *dz, long M, int ldz, const float *wt_packed, int k_pad, int n_pad,
                      const float *relu_out, float *dx, int ldx, m360_stream_t stream) {
    // dX[M, k_pad] = dZ[M, n_pad] * W[n_pad, k_pad]: the forward kernel with the roles of n and k exchanged on the
    // transposed packing wt[k_pad][n_pad]; relu_out (the forward OUTPUT of the previous layer, leading dimension ldx)
    // masks the result (ReLU'), NULL = no mask.  Needs n_pad % 32 == 0 (it is the contraction length here).
    float *zero_bias = nullptr;  // the unmasked epilogue adds a bias: a zero vector that lives in the code object
    if (!relu_out) {
        if (k_pad > kZeroBias) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_dgrad: k_pad=%d > %d without a mask", k_pad, kZeroBias);
        if (hipGetSymbolAddress(reinterpret_cast<void **>(&zero_bias), HIP_SYMBOL(g_zero_bias)) != hipSuccess) return fail(M360_ERR_LAUNCH, "m360_linear_dgrad: zero bias symbol not found");
    }
    return launch_linear(dz, M, ldz, wt_packed, relu_out ? nullptr : zero_bias, k_pad, n_pad, relu_out ? M360_ACT_RELU_MASK : M360_ACT_NONE, dx, ldx, relu_out, stream);
}

static void wgrad_plan(long M, int n_pad, int k_pad, int *ntiles, int *nsplit, long *total_steps, long *steps_per_split) {
    *ntiles = ((n_pad + tn::BT - 1) / tn::BT) * ((k_pad + tn::BT - 1) / tn::BT);
    *total_steps = M / tn::BKM;
    long ns = tn::kMaxWorkgroups / *ntiles;
    if (ns < 1) ns = 1;
    if (ns > *total_steps) ns = *total_steps;
    *nsplit = (int)ns;
    *steps_per_split = ns > 0 ? (*total_steps + ns - 1) / ns : 0;
}
static inline size_t up256_(size_t v) { return (v + 255) & ~(size_t)255; }

size_t m360_linear_wgrad_workspace_bytes(long M, int n_pad, int k_pad) {
    if (M < 0 || n_pad < 1 || k_pad < 1) return 0;
    int ntiles, nsplit;
    long total, per;
    wgrad_plan(M, n_pad, k_pad, &ntiles, &nsplit, &total, &per);
    return up256_((size_t)(nsplit > 0 ? nsplit : 1) * n_pad * k_pad * sizeof(float)) + up256_((size_t)tn::kMaxWorkgroups * n_pad * sizeof(float));
}

int m360_linear_wgrad(const float *dz, int ldz, const float *x, int ldx, long M, int n_pad, int k_pad, float *grad_w,
                      float *grad_b, void *workspace, size_t workspace_bytes, m360_stream_t stream) {
    if (!dz || !x || !grad_w || M < 0 || n_pad < 32 || k_pad < 32 || n_pad % 32 || k_pad % 32 || ldz < n_pad || ldx < k_pad || ldz % 4 || ldx % 4)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_wgrad: bad argument (M=%ld n_pad=%d k_pad=%d ldz=%d ldx=%d)", M, n_pad, k_pad, ldz, ldx);
    if (((uintptr_t)dz | (uintptr_t)x | (uintptr_t)grad_w) & 15) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_wgrad: pointers must be 16-byte aligned");
    const size_t need = m360_linear_wgrad_workspace_bytes(M, n_pad, k_pad);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 255)) return fail(M360_ERR_WORKSPACE_TOO_SMALL, "m360_linear_wgrad: workspace %zu < %zu bytes (or not 256-byte aligned)", workspace_bytes, need);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int ntiles, nsplit;
    long total, per;
    wgrad_plan(M, n_pad, k_pad, &ntiles, &nsplit, &total, &per);
    float *partial = static_cast<float *>(workspace);
    float *bias_part = reinterpret_cast<float *>(static_cast<char *>(workspace) + up256_((size_t)(nsplit > 0 ? nsplit : 1) * n_pad * k_pad * sizeof(float)));
    if (nsplit > 0)
        hipLaunchKernelGGL(tn::linear_tn_kernel, dim3((unsigned)(ntiles * nsplit)), dim3(tn::kThreads), 0, st, dz, ldz, x, ldx, n_pad, k_pad, partial, (k_pad + tn::BT - 1) / tn::BT, ntiles, nsplit, total, per, grad_b ? bias_part : nullptr);
    const long count4 = (long)n_pad * k_pad / 4;
    hipLaunchKernelGGL(tn::tn_reduce_kernel, dim3((unsigned)((count4 + 255) / 256)), dim3(256), 0, st, partial, nsplit, n_pad, k_pad, dz, ldz, x, ldx, total * tn::BKM, M, grad_w);
    if (grad_b) hipLaunchKernelGGL(tn::tn_bias_reduce_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, st, bias_part, nsplit, n_pad, dz, ldz, total * tn::BKM, M, grad_b);
    return check_launch("linear_wgrad");
}

__global__ void pack_linear_t_kernel(const float *__restrict__ w, int n_out, int k_in, int n_pad, int k_pad,
                                     float *__restrict__ wt) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n_pad * k_pad) return;
    const int k = (int)(idx / n_pad), n = (int)(idx % n_pad);
    wt[idx] = (n < n_out && k < k_in) ? w[(long)n * k_in + k] : 0.0f;
}

int m360_pack_linear_transposed(const float *w, int n_out, int k_in, int n_pad, int k_pad, float *wt_packed,
                                m360_stream_t stream) {
    if (!w || !wt_packed || n_out < 1 || k_in < 1 || n_pad < n_out || k_pad < k_in)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_pack_linear_transposed: bad argument (n_out=%d k_in=%d n_pad=%d k_pad=%d)", n_out, k_in, n_pad, k_pad);
    const long n = (long)n_pad * k_pad;
    hipLaunchKernelGGL(pack_linear_t_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, n_out, k_in, n_pad, k_pad, wt_packed);
    return check_launch("pack_linear_transposed");
}

// ---- NaN scan of a parameter set in ONE launch (the bf16 modes refuse NaN parameters: m360.h) - block (x, t) strides over tensor t
}  // extern "C"
namespace m360 {
constexpr int kNanScanMax = 32, kNanScanBlocks = 256;  // (16 workgroups per tensor: 56 us for the set of a 1024-wide model - a 1024 x 1024 matrix was 256 serial loads per thread; 256: 16)
struct nan_scan_t {
    const float *p[kNanScanMax];
    long n[kNanScanMax];
};
__global__ __launch_bounds__(256) void nan_scan_kernel(nan_scan_t a, unsigned *flag) {
    const float *__restrict__ p = a.p[blockIdx.y];
    const long n = a.n[blockIdx.y];
    bool bad = false;
#pragma unroll 4
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)kNanScanBlocks * 256) bad |= p[i] != p[i];
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
}  // namespace m360
extern "C" {
int m360_params_nan_flag(const float *const *tensors, const long *counts, int n_tensors, unsigned *flag, m360_stream_t stream) {
    if (!flag || n_tensors < 0 || (n_tensors > 0 && (!tensors || !counts))) return fail(M360_ERR_INVALID_ARGUMENT, "m360_params_nan_flag: null pointer or negative count");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(flag, 0, sizeof(unsigned), st) != hipSuccess) return fail(M360_ERR_LAUNCH, "m360_params_nan_flag: memset failed");
    for (int t0 = 0; t0 < n_tensors; t0 += kNanScanMax) {
        nan_scan_t a{};
        const int nt = n_tensors - t0 < kNanScanMax ? n_tensors - t0 : kNanScanMax;
        for (int t = 0; t < nt; ++t) {
            if (counts[t0 + t] < 0 || (counts[t0 + t] > 0 && !tensors[t0 + t])) return fail(M360_ERR_INVALID_ARGUMENT, "m360_params_nan_flag: tensor %d: null pointer or negative count", t0 + t);
            a.p[t] = tensors[t0 + t];
            a.n[t] = counts[t0 + t];
        }
        hipLaunchKernelGGL(nan_scan_kernel, dim3(kNanScanBlocks, (unsigned)nt), dim3(256), 0, st, a, flag);
    }
    return check_launch("params_nan_flag");
}

// ---- opt-in bf16 MLP (fp32 accumulate): weights and activations as raw 16-bit bf16
int m360_pack_linear_bf16(const float *w, const float *b, int n_out, int k_in, int n_pad, int k_pad,
                          void *w_packed_bf16, float *b_packed, m360_stream_t stream) {
    if (!w || !w_packed_bf16 || n_out < 1 || k_in < 1 || n_pad < n_out || k_pad < k_in || k_pad % 64 != 0)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_pack_linear_bf16: bad argument (n_out=%d k_in=%d n_pad=%d k_pad=%d; k_pad must be a multiple of 64)", n_out, k_in, n_pad, k_pad);
    const long n = (long)n_pad * k_pad;
    hipLaunchKernelGGL(pbf16::pack_linear_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, b, n_out, k_in, n_pad, k_pad, static_cast<__bf16 *>(w_packed_bf16), b_packed);
    return check_launch("pack_linear_bf16");
}

int m360_linear_bf16(const void *x, long M, int ldx, const void *w_packed, const float *b_packed, int n_pad,
                     int k_pad, int act, void *y, int ldy, m360_stream_t stream) {
    if (!x || !w_packed || !b_packed || !y || M < 0) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16: null pointer or negative M");
    if (n_pad < 1 || k_pad < pbf16::BK || k_pad % pbf16::BK != 0 || ldx < k_pad || ldy < n_pad || ldx % 8 != 0 || ldy % 8 != 0)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16: k_pad=%d must be a positive multiple of %d, ldx=%d >= k_pad, ldy=%d >= n_pad=%d, both multiples of 8", k_pad, pbf16::BK, ldx, ldy, n_pad);
    if (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)b_packed | (uintptr_t)y) & 15) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16: pointers must be 16-byte aligned");
    const int layout = act & M360_ROWS_PAIRED_MASK;  // paired rows in / out (m360.h): the ring kernel's shapes only
    const bool temporal = (act & M360_STORES_TEMPORAL) != 0;
    act &= ~M360_ACT_FLAGS_MASK;
    if (act != M360_ACT_NONE && act != M360_ACT_RELU && act != M360_ACT_SIGMOID) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16: unknown activation %d", act);
    if (temporal && !(layout & M360_ROWS_PAIRED_OUT)) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16: M360_STORES_TEMPORAL goes with M360_ROWS_PAIRED_OUT");
    if (layout && !m360_linear_bf16_rows_pairable(M360_PAIRABLE_LINEAR, n_pad, k_pad))
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16: paired rows with n_pad=%d k_pad=%d: not a shape of the one-wave ring kernel (m360_linear_bf16_rows_pairable)", n_pad, k_pad);
    if ((layout & M360_ROWS_PAIRED_OUT) && act != M360_ACT_RELU) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16: paired output rows: ReLU layers only (act=%d)", act);
    if ((layout & M360_ROWS_PAIRED_IN) && k_pad == w16::BKS) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16: paired input rows: not for the 64-deep first layers (their input is the encoder's)");
    if (M == 0) return M360_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const __bf16 *xb = static_cast<const __bf16 *>(x), *wb = static_cast<const __bf16 *>(w_packed);
    __bf16 *yb = static_cast<__bf16 *>(y);
    const int xin = (layout & M360_ROWS_PAIRED_IN) ? 1 : 0;
    const long M_full = (n_pad % pbf16::BN == 0) ? (M / pbf16::BM) * pbf16::BM : 0;
    if (M_full > 0) {
        const int cus = cu_count();
        if (cus <= 0) return fail(M360_ERR_NO_DEVICE, "m360_linear_bf16: no HIP device");
        const long nt = (M_full / pbf16::BM) * (n_pad / pbf16::BN);
        const bool pp_ok = k_pad >= 2 * pp16::BK && n_pad <= pp16::kMaxBias;  // else (the K = 64 first layer) the one-wave-per-SIMD kernel
        // hidden layers (bias + {none, ReLU}) with a contraction that is a multiple of 128: the one-wave ring kernel with 128 x 128
        // wave tiles (1.22-1.28 PF against the ping-pong kernel's 1.13 on a 1024^2 layer, 0.114 against 0.144 ms on a 256^2 one)
        const bool w16_ok = pp_ok && act != M360_ACT_SIGMOID && k_pad % (2 * w16::BKS) == 0 && k_pad >= M360_W16_MIN_K;
        const bool w16_one = M360_W16_K64 && act != M360_ACT_SIGMOID && k_pad == w16::BKS && n_pad <= w16::kMaxBias;  // the 64-deep first layers
        if (w16_ok || w16_one) {
            dim3 grid((unsigned)(nt < cus ? nt : cus)), block(w16::kThreads);
#define M360_W16(A, ONE, PR) hipLaunchKernelGGL((w16::linear_bf16_w16_kernel<A, 0, false, false, ONE, 0, false, false, PR>), grid, block, 0, st, xb, M_full, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / w16::BN, (int)nt, nullptr, nullptr, xin)
            if (temporal) {
                if (w16_one) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16: M360_STORES_TEMPORAL: hidden layers (k_pad >= %d)", M360_W16_MIN_K);
                hipLaunchKernelGGL((w16::linear_bf16_w16_kernel<M360_ACT_RELU, 128, false, false, false, 0, false, false, true>), grid, block, 0, st, xb, M_full, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / w16::BN, (int)nt, nullptr, nullptr, xin);
            } else if (layout & M360_ROWS_PAIRED_OUT) { if (w16_one) M360_W16(M360_ACT_RELU, true, true); else M360_W16(M360_ACT_RELU, false, true); }
            else if (w16_one) { if (act == M360_ACT_RELU) M360_W16(M360_ACT_RELU, true, false); else M360_W16(M360_ACT_NONE, true, false); }
            else { if (act == M360_ACT_RELU) M360_W16(M360_ACT_RELU, false, false); else M360_W16(M360_ACT_NONE, false, false); }
#undef M360_W16
        } else if (pp_ok) {  // 8-wave ping-pong kernel, persistent
            dim3 grid((unsigned)(nt < cus ? nt : cus)), block(pp16::kThreads);
            switch (act) {
                case M360_ACT_NONE: hipLaunchKernelGGL(pp16::linear_bf16_pp_kernel<M360_ACT_NONE>, grid, block, 0, st, xb, M_full, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / pp16::BN, (int)nt); break;
                case M360_ACT_RELU: hipLaunchKernelGGL(pp16::linear_bf16_pp_kernel<M360_ACT_RELU>, grid, block, 0, st, xb, M_full, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / pp16::BN, (int)nt); break;
                default: hipLaunchKernelGGL(pp16::linear_bf16_pp_kernel<M360_ACT_SIGMOID>, grid, block, 0, st, xb, M_full, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / pp16::BN, (int)nt); break;
            }
        } else {
            dim3 grid((unsigned)(nt < cus ? nt : cus)), block(pbf16::kThreads);
            switch (act) {
                case M360_ACT_NONE: hipLaunchKernelGGL(pbf16::linear_bf16_mfma_persist_kernel<M360_ACT_NONE>, grid, block, 0, st, xb, M_full, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / pbf16::BN, (int)nt); break;
                case M360_ACT_RELU: hipLaunchKernelGGL(pbf16::linear_bf16_mfma_persist_kernel<M360_ACT_RELU>, grid, block, 0, st, xb, M_full, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / pbf16::BN, (int)nt); break;
                default: hipLaunchKernelGGL(pbf16::linear_bf16_mfma_persist_kernel<M360_ACT_SIGMOID>, grid, block, 0, st, xb, M_full, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / pbf16::BN, (int)nt); break;
            }
        }
    }
    if (M > M_full) {
        const long Mt = M - M_full;
        dim3 grid((unsigned)((n_pad + 31) / 32), (unsigned)((Mt + 31) / 32)), block(64);
        const __bf16 *xt = xb + M_full * ldx;
        __bf16 *yt = yb + M_full * ldy;
        switch (act) {
            case M360_ACT_NONE: hipLaunchKernelGGL(pbf16::linear_bf16_mfma_simple_kernel<M360_ACT_NONE>, grid, block, 0, st, xt, Mt, ldx, wb, b_packed, n_pad, k_pad, yt, ldy); break;
            case M360_ACT_RELU: hipLaunchKernelGGL(pbf16::linear_bf16_mfma_simple_kernel<M360_ACT_RELU>, grid, block, 0, st, xt, Mt, ldx, wb, b_packed, n_pad, k_pad, yt, ldy); break;
            default: hipLaunchKernelGGL(pbf16::linear_bf16_mfma_simple_kernel<M360_ACT_SIGMOID>, grid, block, 0, st, xt, Mt, ldx, wb, b_packed, n_pad, k_pad, yt, ldy); break;
        }
    }
    return check_launch("linear_bf16");
}

// ---- the chain of equally shaped bf16 ReLU layers in one launch (w16::chain_t, m360_linear_bf16_w16.hip.h)
// Per-call options (m360_hyper_t: tuning bit M360_TUNE_CHAIN_COOPERATIVE; diagnostics build only: the bound of one wait and a fault to inject -
// the product library refuses those hooks, it never injects a fault).  The library holds no switch of its own.
// Shapes and device the chain takes.  Nothing is probed: the placement the hand-over relies on (workgroup b on the XCD of slot b % 8) is
// checked by every launch itself, inside the kernel, and a launch that finds it violated is re-run layer by layer (mlp_chain_bf16_rerun).
int m360_mlp_chain_bf16_supported(long M, int width, int layers) {
    if (M <= 0 || layers < 1 || layers > 8 || (width != 4 * w16::BN && width != w16::BN)) return 0;
    if (M % ((width == w16::BN ? 512l : 128l) * w16::BM) != 0) return 0;  // an even number of row blocks per group of workgroups
    return cu_count() == 256;
}

// [chain_status_t (128 bytes) | one counter per row block and layer]
size_t m360_mlp_chain_bf16_workspace(long M, int layers) { return sizeof(w16::chain_status_t) + (size_t)(M / w16::BM) * (size_t)layers * sizeof(unsigned); }

int m360_workspace_init(void *workspace, m360_stream_t stream) {
    if (!workspace || ((uintptr_t)workspace & 15)) return fail(M360_ERR_INVALID_ARGUMENT, "m360_workspace_init: null or unaligned workspace");
    if (hipMemsetAsync(workspace, 0, sizeof(w16::chain_status_t), reinterpret_cast<hipStream_t>(stream)) != hipSuccess)
        return fail(M360_ERR_LAUNCH, "m360_workspace_init: hipMemsetAsync failed: %s", hipGetErrorString(hipGetLastError()));
    return M360_OK;
}

int m360_workspace_status(const void *workspace, unsigned *out5, m360_stream_t stream) {
    if (!workspace || !out5) return fail(M360_ERR_INVALID_ARGUMENT, "m360_workspace_status: null pointer");
    w16::chain_status_t h;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(&h, workspace, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return fail(M360_ERR_LAUNCH, "m360_workspace_status: copy failed: %s", hipGetErrorString(hipGetLastError()));
    out5[0] = h.launches; out5[1] = h.recoveries; out5[2] = h.timeouts; out5[3] = h.xcc_mismatch; out5[4] = h.error;
    return M360_OK;
}

}  // extern "C"

namespace m360 {

// One chain launch.  ws = [chain_status_t | counters]; the launch part of the status and the counters are zeroed here (one memset), the
// sticky counters are left alone.  x_in (may be NULL: act0) is what layer 0 reads.
int chain_opts_ok(const m360_hyper_t *opts, const char *who) {
#ifndef M360_DIAG
    if (opts && (opts->chain_debug_wait_ticks != 0 || opts->chain_debug_fault != 0 || (opts->tuning & M360_TUNE_CHAIN_UNGATED)))
        return fail(M360_ERR_INVALID_ARGUMENT, "%s: chain_debug_* / M360_TUNE_CHAIN_UNGATED are test hooks of the diagnostics build (libm360_diag.so); this library has none", who);
#endif
    (void)opts; (void)who;
    return M360_OK;
}

int mlp_chain_bf16_launch(const void *x_in, void *act0, void *act1, long M, int ld, const void *const *w_packed, const float *const *b_packed,
                          int layers, int width, void *ws, m360_stream_t stream, const m360_hyper_t *opts, bool x3) {
    const int rc_opts = chain_opts_ok(opts, "m360_mlp_chain_bf16");
    if (rc_opts != M360_OK) return rc_opts;
    w16::chain_t ch;
    for (int l = 0; l < 8; ++l) {
        ch.w[l] = static_cast<const __bf16 *>(w_packed[l < layers ? l : layers - 1]);
        ch.b[l] = b_packed[l < layers ? l : layers - 1];
        if (!ch.w[l] || !ch.b[l] || (((uintptr_t)ch.w[l] | (uintptr_t)ch.b[l]) & 15)) return fail(M360_ERR_INVALID_ARGUMENT, "m360_mlp_chain_bf16: layer %d: null or unaligned weights / bias", l);
    }
    ch.x_in = static_cast<const __bf16 *>(x_in);
    ch.act[0] = static_cast<__bf16 *>(act0);
    ch.act[1] = static_cast<__bf16 *>(act1);
    ch.status = static_cast<w16::chain_status_t *>(ws);
    ch.done = reinterpret_cast<unsigned *>(static_cast<char *>(ws) + sizeof(w16::chain_status_t));
    ch.layers = layers;
    ch.row_blocks = (int)(M / w16::BM);
    ch.wait_ticks = w16::kChainWaitTicks;
    ch.fault = 0;
#ifdef M360_DIAG
    if (opts && opts->chain_debug_wait_ticks > 0) ch.wait_ticks = (unsigned)opts->chain_debug_wait_ticks;
    if (opts) ch.fault = opts->chain_debug_fault;
#endif
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(static_cast<char *>(ws) + w16::kChainStatusLaunchOffset, 0, m360_mlp_chain_bf16_workspace(M, layers) - w16::kChainStatusLaunchOffset, st) != hipSuccess)
        return fail(M360_ERR_LAUNCH, "m360_mlp_chain_bf16: hipMemsetAsync failed");
    // bf16: rows of `width` bf16, weights [width, width]; bf16x3: [hi | lo] rows of 2 width, weights [Wh | Wh | Wl] of 3 width (m360_linear_bf16x3)
    auto kern = x3 ? w16::linear_bf16_w16_kernel<M360_ACT_RELU, 128, false, true, false, 0, true, false, true, true>
                   : w16::linear_bf16_w16_kernel<M360_ACT_RELU, 128, false, false, false, 0, false, false, true, true>;
    int kp = x3 ? 3 * width : width;
    const __bf16 *X = ch.x_in ? ch.x_in : ch.act[0], *W0 = ch.w[0];
    const float *B0 = ch.b[0], *hw = nullptr;
    __bf16 *Y = ch.act[1];
    float *hp = nullptr;
    int tiles_n = width / w16::BN, ntiles = ch.row_blocks * tiles_n, xpair = 1, stagger = 0, gate_first = 0;
    unsigned *gate = nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool coop = opts && (opts->tuning & M360_TUNE_CHAIN_COOPERATIVE) && hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone;
    if (coop) {  // co-residency of the 256 workgroups asked of the runtime (A/B: M360_TUNE_CHAIN_COOPERATIVE)
        void *args[] = {&X, &M, &ld, &W0, &B0, &width, &kp, &Y, &ld, &tiles_n, &ntiles, &hw, &hp, &xpair, &stagger, &ch, &gate, &gate_first};
        if (hipLaunchCooperativeKernel(reinterpret_cast<void *>(kern), dim3(256), dim3(w16::kThreads), args, 0, st) != hipSuccess)
            return fail(M360_ERR_LAUNCH, "m360_mlp_chain_bf16: cooperative launch failed: %s", hipGetErrorString(hipGetLastError()));
    } else {
        hipLaunchKernelGGL(kern, dim3(256), dim3(w16::kThreads), 0, st, X, M, ld, W0, B0, width, kp, Y, ld, tiles_n, ntiles, hw, hp, xpair, stagger, ch, gate, gate_first);
    }
    return check_launch("mlp_chain_bf16");
}

// The same layers, layer by layer, as GATED launches behind a chain launch: every workgroup first reads the launch's error word and returns
// at once when it is 0 (the normal case: `layers` empty launches); when the chain kernel reported a wait that ran out or a workgroup off
// its XCD, these launches redo all of its rows from x_in (which the chain never writes) - the very kernel m360_linear_bf16 runs for a
// hidden layer on paired rows, hence the same bits.  No host round trip, nothing for a caller to check before using the result.
int mlp_chain_bf16_rerun(const void *x_in, void *act0, void *act1, long M, int ld, const void *const *w_packed, const float *const *b_packed,
                         int layers, int width, void *ws, m360_stream_t stream, const m360_hyper_t *opts, bool x3) {
#ifdef M360_DIAG
    if (opts && (opts->tuning & M360_TUNE_CHAIN_UNGATED)) return M360_OK;  // A/B of the gated launches' cost: the chain alone, unchecked
#endif
    (void)opts;
    const int cus = cu_count();
    if (cus <= 0) return fail(M360_ERR_NO_DEVICE, "m360_mlp_chain_bf16: no HIP device");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    unsigned *gate = reinterpret_cast<unsigned *>(static_cast<char *>(ws) + w16::kChainStatusLaunchOffset);
    const long nt = (M / w16::BM) * (width / w16::BN);
    dim3 grid((unsigned)(nt < cus ? nt : cus)), block(w16::kThreads);
    __bf16 *act[2] = {static_cast<__bf16 *>(act0), static_cast<__bf16 *>(act1)};
    for (int j = 0; j < layers; ++j) {
        const __bf16 *src = j == 0 ? static_cast<const __bf16 *>(x_in) : act[j & 1];
        if (x3)
            hipLaunchKernelGGL((w16::linear_bf16_w16_kernel<M360_ACT_RELU, 0, false, true, false, 0, true, false, true>), grid, block, 0, st, src, M, ld,
                               static_cast<const __bf16 *>(w_packed[j]), b_packed[j], width, 3 * width, act[(j + 1) & 1], ld, width / w16::BN, (int)nt, nullptr, nullptr, 1, 0,
                               w16::chain_t(), gate, j == 0 ? 1 : 0);
        else
        hipLaunchKernelGGL((w16::linear_bf16_w16_kernel<M360_ACT_RELU, 0, false, false, false, 0, false, false, true>), grid, block, 0, st, src, M, ld,
                           static_cast<const __bf16 *>(w_packed[j]), b_packed[j], width, width, act[(j + 1) & 1], ld, width / w16::BN, (int)nt, nullptr, nullptr, 1, 0,
                           w16::chain_t(), gate, j == 0 ? 1 : 0);
    }
    return check_launch("mlp_chain_bf16 (gated re-run)");
}

static int chain_args_ok(const char *who, const void *x_in, void *act0, void *act1, long M, int ld, const void *const *w_packed, const float *const *b_packed,
                         int layers, int width, void *workspace, bool x3 = false) {
    if (!act0 || !act1 || !w_packed || !b_packed || !workspace) return fail(M360_ERR_INVALID_ARGUMENT, "%s: null pointer", who);
    if (x3 && (ld < 2 * width || width != 4 * w16::BN)) return fail(M360_ERR_INVALID_ARGUMENT, "%s: [hi | lo] rows need ld=%d >= 2 width=%d, width 1024", who, ld, width);
    if (ld < width || ld % 8 != 0 || (((uintptr_t)x_in | (uintptr_t)act0 | (uintptr_t)act1 | (uintptr_t)workspace) & 15))
        return fail(M360_ERR_INVALID_ARGUMENT, "%s: ld=%d >= width=%d, a multiple of 8; 16-byte aligned pointers", who, ld, width);
    if (!m360_mlp_chain_bf16_supported(M, width, layers))
        return fail(M360_ERR_INVALID_ARGUMENT, "%s: M=%ld (a multiple of 32768 at width 1024, of 131072 at width 256), width=%d, layers=%d (1..8), a 256-CU device (m360_mlp_chain_bf16_supported)", who, M, width, layers);
    return M360_OK;
}

}  // namespace m360

extern "C" {

int m360_mlp_chain_bf16(void *act0, void *act1, long M, int ld, const void *const *w_packed, const float *const *b_packed, int layers,
                        int width, void *workspace, const m360_hyper_t *opts, m360_stream_t stream) {
    const int rc = chain_args_ok("m360_mlp_chain_bf16", nullptr, act0, act1, M, ld, w_packed, b_packed, layers, width, workspace);
    if (rc != M360_OK) return rc;
    const int rc2 = m360_workspace_init(workspace, stream);  // a standalone call's status is its own
    if (rc2 != M360_OK) return rc2;
    return mlp_chain_bf16_launch(nullptr, act0, act1, M, ld, w_packed, b_packed, layers, width, workspace, stream, opts, false);
}

int m360_mlp_chain_bf16_safe(const void *x_in, void *act0, void *act1, long M, int ld, const void *const *w_packed, const float *const *b_packed,
                             int layers, int width, void *workspace, const m360_hyper_t *opts, m360_stream_t stream) {
    if (!x_in || x_in == act0 || x_in == act1) return fail(M360_ERR_INVALID_ARGUMENT, "m360_mlp_chain_bf16_safe: x_in must be a third buffer (the re-run reads it again)");
    const int rc = chain_args_ok("m360_mlp_chain_bf16_safe", x_in, act0, act1, M, ld, w_packed, b_packed, layers, width, workspace);
    if (rc != M360_OK) return rc;
    if (width != 4 * w16::BN) return fail(M360_ERR_INVALID_ARGUMENT, "m360_mlp_chain_bf16_safe: width=%d (1024: the NeRF MLP's hidden layers)", width);
    const int rc2 = mlp_chain_bf16_launch(x_in, act0, act1, M, ld, w_packed, b_packed, layers, width, workspace, stream, opts, false);
    if (rc2 != M360_OK) return rc2;
    return mlp_chain_bf16_rerun(x_in, act0, act1, M, ld, w_packed, b_packed, layers, width, workspace, stream, opts, false);
}

int m360_mlp_chain_bf16x3_safe(const void *x_in, void *act0, void *act1, long M, int ld, const void *const *w_packed3, const float *const *b_packed,
                               int layers, int width, void *workspace, const m360_hyper_t *opts, m360_stream_t stream) {
    if (!x_in || x_in == act0 || x_in == act1) return fail(M360_ERR_INVALID_ARGUMENT, "m360_mlp_chain_bf16x3_safe: x_in must be a third buffer (the re-run reads it again)");
    const int rc = chain_args_ok("m360_mlp_chain_bf16x3_safe", x_in, act0, act1, M, ld, w_packed3, b_packed, layers, width, workspace, true);
    if (rc != M360_OK) return rc;
    const int rc2 = mlp_chain_bf16_launch(x_in, act0, act1, M, ld, w_packed3, b_packed, layers, width, workspace, stream, opts, true);
    if (rc2 != M360_OK) return rc2;
    return mlp_chain_bf16_rerun(x_in, act0, act1, M, ld, w_packed3, b_packed, layers, width, workspace, stream, opts, true);
}

int m360_pack_linear_bf16x6(const float *w, const float *b, int n_out, int k_in, int n_pad, int k_pad, void *w_packed6,
                            float *b_packed, m360_stream_t stream) {
    if (!w || !w_packed6 || n_out < 1 || k_in < 1 || n_pad < n_out || k_pad < k_in || k_pad % pbf16::BK != 0)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_pack_linear_bf16x6: n_pad=%d >= n_out=%d, k_pad=%d >= k_in=%d (multiple of %d)", n_pad, n_out, k_pad, k_in, pbf16::BK);
    const long n = (long)n_pad * k_pad;
    hipLaunchKernelGGL(pbf16::pack_linear_bf16x6_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, b, n_out, k_in, n_pad, k_pad, static_cast<__bf16 *>(w_packed6), b_packed);
    return check_launch("pack_linear_bf16x6");
}

// plain bf16 rows in, [hi | lo] pair rows out: the x6 first layer of the bf16x3 mode (K = 6 in_pad = 384: the one-wave ring kernel's
// plain K loop with the split epilogue); any other shape on the generic kernel
int m360_linear_bf16_split(const void *x, long M, int ldx, const void *w_packed, const float *b_packed, int n_pad, int k_pad,
                           int act, void *y, int ldy, m360_stream_t stream) {
    if (!x || !w_packed || !b_packed || !y || M < 0) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16_split: null pointer or negative M");
    if (n_pad < 1 || k_pad < pbf16::BK || k_pad % pbf16::BK != 0 || ldx < k_pad || ldy < 2 * n_pad || ldx % 8 != 0 || ldy % 8 != 0 || n_pad % 8 != 0)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16_split: k_pad=%d must be a positive multiple of %d, ldx=%d >= k_pad, ldy=%d >= 2 n_pad=%d, all multiples of 8", k_pad, pbf16::BK, ldx, ldy, 2 * n_pad);
    if (((uintptr_t)x | (uintptr_t)w_packed | (uintptr_t)b_packed | (uintptr_t)y) & 15) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16_split: pointers must be 16-byte aligned");
    const int layout = act & M360_ROWS_PAIRED_MASK;  // paired OUTPUT rows (the input rows are the encoder's)
    const bool temporal = (act & M360_STORES_TEMPORAL) != 0;
    act &= ~M360_ACT_FLAGS_MASK;
    if (temporal && !layout) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16_split: M360_STORES_TEMPORAL goes with M360_ROWS_PAIRED_OUT");
    if (act != M360_ACT_NONE && act != M360_ACT_RELU) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16_split: activation %d (none or ReLU)", act);
    if (layout && (layout != M360_ROWS_PAIRED_OUT || act != M360_ACT_RELU || !m360_linear_bf16_rows_pairable(M360_PAIRABLE_SPLIT, n_pad, k_pad)))
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16_split: paired rows: output only, ReLU, a shape of the one-wave ring kernel (n_pad=%d k_pad=%d)", n_pad, k_pad);
    if (M == 0) return M360_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const __bf16 *xb = static_cast<const __bf16 *>(x), *wb = static_cast<const __bf16 *>(w_packed);
    __bf16 *yb = static_cast<__bf16 *>(y);
    const bool ring = n_pad % w16::BN == 0 && n_pad <= w16::kMaxBias && k_pad % (2 * w16::BKS) == 0 && k_pad >= M360_W16_MIN_K;
    const long M_full = ring ? (M / w16::BM) * w16::BM : 0;
    if (M_full > 0) {
        const int cus = cu_count();
        if (cus <= 0) return fail(M360_ERR_NO_DEVICE, "m360_linear_bf16_split: no HIP device");
        const long nt = (M_full / w16::BM) * (n_pad / w16::BN);
        dim3 grid((unsigned)(nt < cus ? nt : cus)), block(w16::kThreads);
        if (temporal) hipLaunchKernelGGL((w16::linear_bf16_w16_kernel<M360_ACT_RELU, 128, false, false, false, 0, true, false, true>), grid, block, 0, st, xb, M_full, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / w16::BN, (int)nt);
        else if (layout) hipLaunchKernelGGL((w16::linear_bf16_w16_kernel<M360_ACT_RELU, 0, false, false, false, 0, true, false, true>), grid, block, 0, st, xb, M_full, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / w16::BN, (int)nt);
        else if (act == M360_ACT_RELU) hipLaunchKernelGGL((w16::linear_bf16_w16_kernel<M360_ACT_RELU, 0, false, false, false, 0, true>), grid, block, 0, st, xb, M_full, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / w16::BN, (int)nt);
        else hipLaunchKernelGGL((w16::linear_bf16_w16_kernel<M360_ACT_NONE, 0, false, false, false, 0, true>), grid, block, 0, st, xb, M_full, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / w16::BN, (int)nt);
    }
    if (M > M_full) {
        const long Mt = M - M_full;
        dim3 grid((unsigned)((n_pad + 31) / 32), (unsigned)((Mt + 31) / 32)), block(64);
        const __bf16 *xt = xb + M_full * ldx;
        __bf16 *yt = yb + M_full * ldy;
        if (act == M360_ACT_RELU) hipLaunchKernelGGL((pbf16::linear_bf16_mfma_simple_kernel<M360_ACT_RELU, false, true>), grid, block, 0, st, xt, Mt, ldx, wb, b_packed, n_pad, k_pad, yt, ldy);
        else hipLaunchKernelGGL((pbf16::linear_bf16_mfma_simple_kernel<M360_ACT_NONE, false, true>), grid, block, 0, st, xt, Mt, ldx, wb, b_packed, n_pad, k_pad, yt, ldy);
    }
    return check_launch("linear_bf16_split");
}

// ---- bf16 training path (round 5): transposed bf16 packing, input gradient, weight gradient
}  // extern "C"
namespace m360 {
int linear_wgrad_bf16_rows(const void *dz, int ldz, const void *x, int ldx, long M, int n_pad, int k_pad, float *grad_w, float *grad_b,
                           void *workspace, size_t workspace_bytes, unsigned tuning, m360_stream_t stream, bool x_rows_overlap);
// dx[m, k] = relu_out[m, k] > 0 ? dx[m, k] : 0 in place (the second half of m360_linear_dgrad_bf16; m360_capi.hip runs it on a second stream
// beside the layer's weight gradient, which does not read dx)
// can the throttled form with `blocks` workgroups also form the column sums of the masked rows (mlp_backward_bf16: the next layer's bias gradient)?
bool relu_mask_bf16_sums_ok(long M, int k_pad, int blocks) {
    const int c8 = k_pad / 8;
    return k_pad % 8 == 0 && c8 >= 1 && c8 <= 256 && 256 % c8 == 0 && blocks > 0 && (M * c8 + 255) / 256 > blocks;
}
// bytes of its part[blocks * 256 / (k_pad / 8)][k_pad] (one row of sums per thread group: 8 floats per thread)
constexpr int kMaskSumGroups = 32;  // row groups of the first reduction level
size_t relu_mask_bf16_sums_bytes(int blocks) { return (size_t)blocks * 256 * 8 * sizeof(float) + (size_t)kMaskSumGroups * 2048 * sizeof(float); }
// ... and the reduction of those rows into grad_b[k_pad] (ascending)
int relu_mask_bf16_sums_reduce(const float *part, int blocks, int k_pad, float *grad_b, m360_stream_t stream) {
    if (!part || !grad_b || blocks < 1 || k_pad < 8) return fail(M360_ERR_INVALID_ARGUMENT, "relu_mask_bf16_sums_reduce: bad argument");
    // two levels (round 6): kMaskSumGroups groups of rows by as many workgroups per 256 columns, then those sums - one thread per column over all
    // 1024 - 2048 rows was 70 us of dependent load rounds at the end of the second stream, which the caller's stream waits for
    const int rows = blocks * (256 / (k_pad / 8)), per = (rows + kMaskSumGroups - 1) / kMaskSumGroups;
    float *level1 = const_cast<float *>(part) + (size_t)blocks * 256 * 8;  // behind the mask kernel's rows (relu_mask_bf16_sums_bytes)
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(tn16::tn16_bias_reduce_rows_kernel, dim3((unsigned)((k_pad + 255) / 256), (unsigned)kMaskSumGroups), dim3(256), 0, st, part, rows, per, k_pad, level1);
    hipLaunchKernelGGL(tn16::tn16_bias_reduce_kernel, dim3((unsigned)((k_pad + 255) / 256)), dim3(256), 0, st, static_cast<const float *>(level1), kMaskSumGroups, k_pad, static_cast<const __bf16 *>(nullptr), 0, 0l, 0l, grad_b);
    return check_launch("linear_dgrad_bf16 (column sums of the masked rows)");
}
int relu_mask_bf16(void *dx, const void *relu_out, long M, int k_pad, int ldx, m360_stream_t stream, int blocks, float *sums_part) {
    if (!dx || !relu_out || M < 0 || k_pad < 8 || k_pad % 8 || ldx < k_pad || ldx % 8) return fail(M360_ERR_INVALID_ARGUMENT, "relu_mask_bf16: bad argument (M=%ld k_pad=%d ldx=%d)", M, k_pad, ldx);
    if (((uintptr_t)relu_out | (uintptr_t)dx) & 15) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_dgrad_bf16: relu_out / dx must be 16-byte aligned");
    if (M == 0) return M360_OK;
    const long n = M * (k_pad / 8);
    if (sums_part && !relu_mask_bf16_sums_ok(M, k_pad, blocks)) return fail(M360_ERR_INVALID_ARGUMENT, "relu_mask_bf16: column sums need the throttled form on a width whose eighth divides 256 (M=%ld k_pad=%d blocks=%d)", M, k_pad, blocks);
    if (sums_part)
        hipLaunchKernelGGL((tn16::relu_mask_bf16_stride_kernel<2, true>), dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), static_cast<__bf16 *>(dx), static_cast<const __bf16 *>(relu_out), M, k_pad, ldx, sums_part);
    else if (blocks > 0 && (n + 255) / 256 > blocks)  // beside another kernel: a fixed number of workgroups (m360_capi.hip: mlp_backward_bf16)
        hipLaunchKernelGGL((tn16::relu_mask_bf16_stride_kernel<2, false>), dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), static_cast<__bf16 *>(dx), static_cast<const __bf16 *>(relu_out), M, k_pad, ldx, static_cast<float *>(nullptr));
    else
        hipLaunchKernelGGL(tn16::relu_mask_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), static_cast<__bf16 *>(dx), static_cast<const __bf16 *>(relu_out), M, k_pad, ldx);
    return check_launch("linear_dgrad_bf16 (ReLU mask)");
}
}  // namespace m360
extern "C" {
int m360_pack_linear_bf16_transposed(const float *w, int n_out, int k_in, int n_pad, int k_pad, void *wt_packed_bf16, m360_stream_t stream) {
    if (!w || !wt_packed_bf16 || n_out < 1 || k_in < 1 || n_pad < n_out || k_pad < k_in || n_pad % 64 != 0)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_pack_linear_bf16_transposed: bad argument (n_out=%d k_in=%d n_pad=%d k_pad=%d; n_pad - the contraction of the input gradient - must be a multiple of 64)", n_out, k_in, n_pad, k_pad);
    const long n = (long)n_pad * k_pad;
    hipLaunchKernelGGL(tn16::pack_linear_bf16_t_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, n_out, k_in, n_pad, k_pad, static_cast<__bf16 *>(wt_packed_bf16));
    return check_launch("pack_linear_bf16_transposed");
}

int m360_linear_dgrad_bf16(const void *dz, long M, int ldz, const void *wt_packed_bf16, int k_pad, int n_pad, const void *relu_out,
                           void *dx, int ldx, m360_stream_t stream) {
    // dX[M, k_pad] = dZ[M, n_pad] * W[n_pad, k_pad] in bf16 with fp32 accumulation: the forward layer kernel (m360_linear_bf16, act none,
    // zero bias) on the transposed packing wt[k_pad][n_pad]; then, where relu_out (the forward OUTPUT of the layer below, bf16, leading
    // dimension ldx) is <= 0, the row's entry is cleared (ReLU').  The mask is a pass of its own here: the forward kernels' epilogues
    // are counted-wait schedules that a load in the epilogue would stall
    if (!dz || !wt_packed_bf16 || !dx || M < 0) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_dgrad_bf16: null pointer or negative M");
    if (k_pad > kZeroBias || k_pad % 8 != 0 || ldx < k_pad || ldx % 8 != 0) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_dgrad_bf16: k_pad=%d (<= %d, a multiple of 8), ldx=%d", k_pad, kZeroBias, ldx);
    float *zero_bias = nullptr;
    if (hipGetSymbolAddress(reinterpret_cast<void **>(&zero_bias), HIP_SYMBOL(g_zero_bias)) != hipSuccess) return fail(M360_ERR_LAUNCH, "m360_linear_dgrad_bf16: zero bias symbol not found");
    const int rc = m360_linear_bf16(dz, M, ldz, wt_packed_bf16, zero_bias, k_pad, n_pad, M360_ACT_NONE, dx, ldx, stream);
    if (rc != M360_OK || !relu_out || M == 0) return rc;
    return m360::relu_mask_bf16(dx, relu_out, M, k_pad, ldx, stream, 0, nullptr);
}

// which MFMA form m360_linear_wgrad_bf16 runs is the caller's per-call choice (`tuning`): default = one wave per SIMD, 128 x 128 wave tiles,
// five LDS quarters (m360_linear_tn_bf16_w.hip.h), M360_TUNE_WGRAD_FORM0 = the 8-wave kernel of m360_linear_tn_bf16.hip.h.  Both deterministic; their row
// splits differ, so their results agree to fp32 summation order, not bit for bit.
// The path is a function of (M, n_pad, k_pad) ALONE - it sizes the workspace (m360_linear_wgrad_bf16_workspace_bytes has nothing else) and it
// picks the kernel; what the MFMA kernels additionally need of a call (leading dimensions that are multiples of 8, 16-byte aligned pointers) is
// REQUIRED of a call with such pads, not a reason to take the other path with a workspace sized for this one.
static bool wgrad_bf16_on_mfma(long M, int n_pad, int k_pad) {
    return n_pad % tn16::BT == 0 && k_pad % tn16::BT == 0 && M >= tn16::BKM;
}
static void wgrad_bf16_plan(long M, int n_pad, int k_pad, int *ntiles, int *nsplit, long *total_steps, long *steps_per_split) {
    *ntiles = (n_pad / tn16::BT) * (k_pad / tn16::BT);
    *total_steps = M / tn16::BKM;
    long ns = tn16::kMaxWorkgroups / *ntiles;
    if (ns < 1) ns = 1;
    if (ns > *total_steps) ns = *total_steps;
    *nsplit = (int)ns;
    *steps_per_split = ns > 0 ? (*total_steps + ns - 1) / ns : 0;
}

size_t m360_linear_wgrad_bf16_workspace_bytes(long M, int n_pad, int k_pad) {
    if (M < 0 || n_pad < 1 || k_pad < 1) return 0;
    if (wgrad_bf16_on_mfma(M, n_pad, k_pad)) {
        int ntiles, nsplit;
        long total, per;
        wgrad_bf16_plan(M, n_pad, k_pad, &ntiles, &nsplit, &total, &per);
        return up256_((size_t)(nsplit > 0 ? nsplit : 1) * n_pad * k_pad * sizeof(float)) + up256_((size_t)tn16::kMaxWorkgroups * n_pad * sizeof(float));
    }
    // other shapes: both operands widened to fp32 for the fp32 kernel
    return up256_((size_t)M * n_pad * sizeof(float)) + up256_((size_t)M * k_pad * sizeof(float)) + m360_linear_wgrad_workspace_bytes(M, n_pad, k_pad);
}

int m360_linear_wgrad_bf16(const void *dz, int ldz, const void *x, int ldx, long M, int n_pad, int k_pad, float *grad_w, float *grad_b,
                           void *workspace, size_t workspace_bytes, unsigned tuning, m360_stream_t stream) {
    return m360::linear_wgrad_bf16_rows(dz, ldz, x, ldx, M, n_pad, k_pad, grad_w, grad_b, workspace, workspace_bytes, tuning, stream, false);
}
}  // extern "C"
namespace m360 {
// m360_linear_wgrad_bf16 - and, for the stage backwards (m360_capi.hip: the first layer, whose operand rows are 2 in_pad = 128 bf16 where the MFMA
// kernels tile the contraction in 256 columns), x_rows_overlap: ldx < k_pad is allowed, columns ldx .. k_pad - 1 of a row ARE
// the next row's first columns.  The caller owns the (k_pad - ldx) elements behind the last row and ignores columns >= ldx of grad_w (each
// element of dW depends on its own column of X only).  Round 6: until then a zero-padded [M, 256] copy was made for every backward - a 268 MB
// memset and a strided copy, 0.1 ms - for the same number of MFMAs; the overlapped rows' second halves are the lines the next row reads anyway.
int linear_wgrad_bf16_rows(const void *dz, int ldz, const void *x, int ldx, long M, int n_pad, int k_pad, float *grad_w, float *grad_b,
                           void *workspace, size_t workspace_bytes, unsigned tuning, m360_stream_t stream, bool x_rows_overlap) {
    if (x_rows_overlap && ldx < 8) return fail(M360_ERR_INVALID_ARGUMENT, "linear_wgrad_bf16_rows: ldx=%d", ldx);
    if (!dz || !x || !grad_w || M < 0 || n_pad < 32 || k_pad < 32 || n_pad % 32 || k_pad % 32 || ldz < n_pad || (ldx < k_pad && !x_rows_overlap))
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_wgrad_bf16: bad argument (M=%ld n_pad=%d k_pad=%d ldz=%d ldx=%d)", M, n_pad, k_pad, ldz, ldx);
    const size_t need = m360_linear_wgrad_bf16_workspace_bytes(M, n_pad, k_pad);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 255)) return fail(M360_ERR_WORKSPACE_TOO_SMALL, "m360_linear_wgrad_bf16: workspace %zu < %zu bytes (or not 256-byte aligned)", workspace_bytes, need);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const __bf16 *dzb = static_cast<const __bf16 *>(dz), *xb = static_cast<const __bf16 *>(x);
    if (wgrad_bf16_on_mfma(M, n_pad, k_pad)) {
        if (ldz % 8 != 0 || ldx % 8 != 0 || (((uintptr_t)dz | (uintptr_t)x | (uintptr_t)grad_w) & 15))
            return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_wgrad_bf16: n_pad=%d, k_pad=%d (multiples of %d) run on the MFMA kernels: ldz=%d and ldx=%d must be multiples of 8 and dz, x, grad_w 16-byte aligned", n_pad, k_pad, tn16::BT, ldz, ldx);
        int ntiles, nsplit;
        long total, per;
        wgrad_bf16_plan(M, n_pad, k_pad, &ntiles, &nsplit, &total, &per);
        // a split's rows are addressed by 32-bit offsets inside a 2 GB buffer descriptor (M x ld beyond ~16 G elements: split the batch)
        if ((per + 1) * tn16::BKM * (long)(ldz > ldx ? ldz : ldx) * 2 >= (1l << 31))
            return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_wgrad_bf16: M=%ld rows of %d elements exceed the 2 GB a row split addresses (%d splits); call it on blocks of rows", M, ldz > ldx ? ldz : ldx, nsplit);
        float *partial = static_cast<float *>(workspace);
        float *bias_part = reinterpret_cast<float *>(static_cast<char *>(workspace) + up256_((size_t)(nsplit > 0 ? nsplit : 1) * n_pad * k_pad * sizeof(float)));
        const long total32 = M / tn16w::KS, per32 = (total32 + nsplit - 1) / nsplit;
        // one wave per SIMD: k-steps of 32 rows, the same number of splits (>= 4 k tiles: its bias sums; a split's rows within the 2 GB its
        // buffer descriptors address: 32-bit offsets)
        if (!(tuning & M360_TUNE_WGRAD_FORM0) && k_pad >= 4 * tn16w::BT && (per32 + 1) * tn16w::KS * (long)(ldz > ldx ? ldz : ldx) * 2 < (1l << 31)) {
#define M360_TNW_LAUNCH(A) hipLaunchKernelGGL(tn16w::linear_tn_bf16_w_kernel<A>, dim3((unsigned)(ntiles * nsplit)), dim3(tn16w::kThreads), 0, st, dzb, ldz, xb, ldx, n_pad, k_pad, partial, k_pad / tn16w::BT, ntiles, nsplit, total32, per32, grad_b ? bias_part : nullptr)
#ifdef M360_DIAG  // diagnostics build only (tools/diag/wgrad_bf16_probe.py with M360_LIB=libm360_diag.so): ablations of the one-wave form, wrong results when != 0
            static const int wabl = getenv("M360_TNW_ABL") ? atoi(getenv("M360_TNW_ABL")) : 0;
            switch (wabl) {
                case 1: M360_TNW_LAUNCH(1); break;
                case 2: M360_TNW_LAUNCH(2); break;
                case 3: M360_TNW_LAUNCH(3); break;
                case 4: M360_TNW_LAUNCH(4); break;
                case 5: M360_TNW_LAUNCH(5); break;
                case 6: M360_TNW_LAUNCH(6); break;
                case 8: M360_TNW_LAUNCH(8); break;
                case 16: M360_TNW_LAUNCH(16); break;
                default: M360_TNW_LAUNCH(0); break;
            }
#else
            M360_TNW_LAUNCH(0);
#endif
#undef M360_TNW_LAUNCH
            const long count4w = (long)n_pad * k_pad / 4;
            hipLaunchKernelGGL(tn16::tn16_reduce_kernel, dim3((unsigned)((count4w + 255) / 256)), dim3(256), 0, st, partial, nsplit, n_pad, k_pad, dzb, ldz, xb, ldx, total32 * tn16w::KS, M, grad_w);
            if (grad_b) hipLaunchKernelGGL(tn16::tn16_bias_reduce_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, st, bias_part, nsplit, n_pad, dzb, ldz, total32 * tn16w::KS, M, grad_b);
            return check_launch("linear_wgrad_bf16 (one-wave form)");
        }
#ifdef M360_DIAG  // diagnostics build only (make diag; tools/diag/wgrad_bf16_probe.py with M360_LIB=libm360_diag.so): ablations, wrong results when != 0
        static const int abl = getenv("M360_TN16_ABL") ? atoi(getenv("M360_TN16_ABL")) : 0;
#else
        const int abl = 0;
#endif
        hipLaunchKernelGGL(tn16::linear_tn_bf16_kernel, dim3((unsigned)(ntiles * nsplit)), dim3(tn16::kThreads), 0, st, dzb, ldz, xb, ldx, n_pad, k_pad, partial, k_pad / tn16::BT, ntiles, nsplit, total, per, grad_b ? bias_part : nullptr, abl);
        const long count4 = (long)n_pad * k_pad / 4;
        hipLaunchKernelGGL(tn16::tn16_reduce_kernel, dim3((unsigned)((count4 + 255) / 256)), dim3(256), 0, st, partial, nsplit, n_pad, k_pad, dzb, ldz, xb, ldx, total * tn16::BKM, M, grad_w);
        if (grad_b) hipLaunchKernelGGL(tn16::tn16_bias_reduce_kernel, dim3((unsigned)((n_pad + 255) / 256)), dim3(256), 0, st, bias_part, nsplit, n_pad, dzb, ldz, total * tn16::BKM, M, grad_b);
        return check_launch("linear_wgrad_bf16");
    }
    if (M == 0) {
        if (hipMemsetAsync(grad_w, 0, (size_t)n_pad * k_pad * sizeof(float), st) != hipSuccess || (grad_b && hipMemsetAsync(grad_b, 0, (size_t)n_pad * sizeof(float), st) != hipSuccess))
            return fail(M360_ERR_LAUNCH, "m360_linear_wgrad_bf16: memset failed");
        return M360_OK;
    }
    float *dzf = static_cast<float *>(workspace);
    float *xf = reinterpret_cast<float *>(static_cast<char *>(workspace) + up256_((size_t)M * n_pad * sizeof(float)));
    char *rest = reinterpret_cast<char *>(xf) + up256_((size_t)M * k_pad * sizeof(float));
    const size_t used = (size_t)(rest - static_cast<char *>(workspace));
    if (used > workspace_bytes) return fail(M360_ERR_WORKSPACE_TOO_SMALL, "m360_linear_wgrad_bf16: workspace %zu < %zu bytes for the widened operands", workspace_bytes, used);
    hipLaunchKernelGGL(tn16::widen_bf16_kernel, dim3((unsigned)((M * n_pad + 255) / 256)), dim3(256), 0, st, dzb, M, n_pad, ldz, dzf);
    hipLaunchKernelGGL(tn16::widen_bf16_kernel, dim3((unsigned)((M * k_pad + 255) / 256)), dim3(256), 0, st, xb, M, k_pad, ldx, xf);
    const int rc = check_launch("linear_wgrad_bf16 (widen)");
    if (rc != M360_OK) return rc;
    return m360_linear_wgrad(dzf, n_pad, xf, k_pad, M, n_pad, k_pad, grad_w, grad_b, rest, workspace_bytes - used, stream);
}
}  // namespace m360
extern "C" {

int m360_pack_linear_bf16x3(const float *w, const float *b, int n_out, int k_in, int n_pad, int k_pad, void *w_packed3,
                            float *b_packed, m360_stream_t stream) {
    if (!w || !w_packed3 || n_out < 1 || k_in < 1 || n_pad < n_out || k_pad < k_in || k_pad % pbf16::BK != 0 || n_pad % 32 != 0)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_pack_linear_bf16x3: n_pad=%d >= n_out=%d (multiple of 32), k_pad=%d >= k_in=%d (multiple of %d)", n_pad, n_out, k_pad, k_in, pbf16::BK);
    const long n = (long)n_pad * k_pad;
    hipLaunchKernelGGL(pbf16::pack_linear_bf16x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), w, b, n_out, k_in, n_pad, k_pad, static_cast<__bf16 *>(w_packed3), b_packed);
    return check_launch("pack_linear_bf16x3");
}

// split_out = true: m360_linear_bf16x3 ([hi | lo] rows out); false: m360_linear_bf16x3_bf16out (one bf16 term out: the first layer of the
// bf16 mode - bias + {none, ReLU}, full tiles on the ring kernel's X3 loop with the plain epilogue, ragged rows on the generic kernel)
static int linear_bf16x3_any(const void *x, long M, int ldx, const void *w_packed3, const float *b_packed, int n_pad, int k_pad,
                             int act, void *y, int ldy, m360_stream_t stream, bool split_out);

int m360_linear_bf16x3(const void *x, long M, int ldx, const void *w_packed3, const float *b_packed, int n_pad, int k_pad,
                       int act, void *y, int ldy, m360_stream_t stream) {
    return linear_bf16x3_any(x, M, ldx, w_packed3, b_packed, n_pad, k_pad, act, y, ldy, stream, true);
}

int m360_linear_bf16x3_bf16out(const void *x, long M, int ldx, const void *w_packed3, const float *b_packed, int n_pad, int k_pad,
                               int act, void *y, int ldy, m360_stream_t stream) {
    if ((act & ~M360_ACT_FLAGS_MASK) != M360_ACT_NONE && (act & ~M360_ACT_FLAGS_MASK) != M360_ACT_RELU) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16x3_bf16out: activation %d (none or ReLU)", act);
    return linear_bf16x3_any(x, M, ldx, w_packed3, b_packed, n_pad, k_pad, act, y, ldy, stream, false);
}

static int linear_bf16x3_any(const void *x, long M, int ldx, const void *w_packed3, const float *b_packed, int n_pad, int k_pad,
                             int act, void *y, int ldy, m360_stream_t stream, bool split_out) {
    if (!x || !w_packed3 || !b_packed || !y || M < 0) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16x3: null pointer or negative M");
    if (n_pad < 1 || k_pad < pbf16::BK || k_pad % pbf16::BK != 0 || ldx < 2 * k_pad || ldy < (split_out ? 2 : 1) * n_pad || ldx % 8 != 0 || ldy % 8 != 0 || n_pad % 8 != 0)
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16x3: k_pad=%d must be a positive multiple of %d, ldx=%d >= 2 k_pad, ldy=%d >= %s n_pad=%d, all multiples of 8", k_pad, pbf16::BK, ldx, ldy, split_out ? "2" : "1", n_pad);
    if (((uintptr_t)x | (uintptr_t)w_packed3 | (uintptr_t)b_packed | (uintptr_t)y) & 15) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16x3: pointers must be 16-byte aligned");
    const int layout = act & M360_ROWS_PAIRED_MASK;  // paired rows in / out (m360.h); the bf16-out first layer: out only
    const bool temporal = (act & M360_STORES_TEMPORAL) != 0;
    act &= ~M360_ACT_FLAGS_MASK;
    if (temporal && !(layout & M360_ROWS_PAIRED_OUT)) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16x3: M360_STORES_TEMPORAL goes with M360_ROWS_PAIRED_OUT");
    if (act != M360_ACT_NONE && act != M360_ACT_RELU && act != M360_ACT_SIGMOID) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16x3: unknown activation %d", act);
    if (layout && !m360_linear_bf16_rows_pairable(split_out ? M360_PAIRABLE_X3 : M360_PAIRABLE_X3_BF16OUT, n_pad, k_pad))
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16x3: paired rows with n_pad=%d k_pad=%d: not a shape of the one-wave ring kernel (m360_linear_bf16_rows_pairable)", n_pad, k_pad);
    if ((layout & M360_ROWS_PAIRED_OUT) && act != M360_ACT_RELU) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16x3: paired output rows: ReLU layers only (act=%d)", act);
    if ((layout & M360_ROWS_PAIRED_IN) && !split_out) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16x3_bf16out: its input rows are the encoder's: paired output only");
    const int xin = (layout & M360_ROWS_PAIRED_IN) ? 1 : 0;
    if (M == 0) return M360_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const __bf16 *xb = static_cast<const __bf16 *>(x), *wb = static_cast<const __bf16 *>(w_packed3);
    __bf16 *yb = static_cast<__bf16 *>(y);
    const int k3 = 3 * k_pad;  // >= 192: always at least two K-steps of the ping-pong kernel
    const long M_full = (n_pad % pp16::BN == 0 && n_pad <= pp16::kMaxBias) ? (M / pp16::BM) * pp16::BM : 0;
    if (M_full > 0) {
        const int cus = cu_count();
        if (cus <= 0) return fail(M360_ERR_NO_DEVICE, "m360_linear_bf16x3: no HIP device");
        const long nt = (M_full / pp16::BM) * (n_pad / pp16::BN);
        dim3 grid((unsigned)(nt < cus ? nt : cus)), block(pp16::kThreads);
        // hidden layers (bias + {none, ReLU}): the one-wave ring kernel (same accumulation order)
        if (M360_W16_X3 && act != M360_ACT_SIGMOID && k_pad % w16::BKS == 0) {
            dim3 blk(w16::kThreads);
#define M360_W16X(A, ONE, SP) hipLaunchKernelGGL((w16::linear_bf16_w16_kernel<A, 0, false, true, ONE, 0, SP>), grid, blk, 0, st, xb, M_full, ldx, wb, b_packed, n_pad, k3, yb, ldy, n_pad / w16::BN, (int)nt, nullptr, nullptr, xin)
#define M360_W16XP(ONE, SP) hipLaunchKernelGGL((w16::linear_bf16_w16_kernel<M360_ACT_RELU, 0, false, true, ONE, 0, SP, false, true>), grid, blk, 0, st, xb, M_full, ldx, wb, b_packed, n_pad, k3, yb, ldy, n_pad / w16::BN, (int)nt, nullptr, nullptr, xin)
#define M360_W16XT(ONE, SP) hipLaunchKernelGGL((w16::linear_bf16_w16_kernel<M360_ACT_RELU, 128, false, true, ONE, 0, SP, false, true>), grid, blk, 0, st, xb, M_full, ldx, wb, b_packed, n_pad, k3, yb, ldy, n_pad / w16::BN, (int)nt, nullptr, nullptr, xin)
            if (temporal) {  // temporal stores: the hidden layers of the bf16x3 mode, the first layer of the bf16 mode
                if (split_out) { if (k_pad == w16::BKS) return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16x3: M360_STORES_TEMPORAL: k_pad > %d", w16::BKS); M360_W16XT(false, true); }
                else { if (k_pad == w16::BKS) M360_W16XT(true, false); else M360_W16XT(false, false); }
            } else if (layout & M360_ROWS_PAIRED_OUT) {
                if (split_out) { if (k_pad == w16::BKS) M360_W16XP(true, true); else M360_W16XP(false, true); }
                else { if (k_pad == w16::BKS) M360_W16XP(true, false); else M360_W16XP(false, false); }
            } else if (split_out) {
                if (k_pad == w16::BKS) { if (act == M360_ACT_RELU) M360_W16X(M360_ACT_RELU, true, true); else M360_W16X(M360_ACT_NONE, true, true); }
                else { if (act == M360_ACT_RELU) M360_W16X(M360_ACT_RELU, false, true); else M360_W16X(M360_ACT_NONE, false, true); }
            } else {
                if (k_pad == w16::BKS) { if (act == M360_ACT_RELU) M360_W16X(M360_ACT_RELU, true, false); else M360_W16X(M360_ACT_NONE, true, false); }
                else { if (act == M360_ACT_RELU) M360_W16X(M360_ACT_RELU, false, false); else M360_W16X(M360_ACT_NONE, false, false); }
            }
#undef M360_W16X
#undef M360_W16XP
#undef M360_W16XT
        } else if (!split_out) {
            return fail(M360_ERR_INVALID_ARGUMENT, "m360_linear_bf16x3_bf16out: no full-tile kernel for this shape");  // (unreachable: k_pad is a multiple of 64)
        } else
        switch (act) {
            case M360_ACT_NONE: hipLaunchKernelGGL((pp16::linear_bf16_pp_kernel<M360_ACT_NONE, false, M360_X3_MODE>), grid, block, 0, st, xb, M_full, ldx, wb, b_packed, n_pad, k3, yb, ldy, n_pad / pp16::BN, (int)nt); break;
            case M360_ACT_RELU: hipLaunchKernelGGL((pp16::linear_bf16_pp_kernel<M360_ACT_RELU, false, M360_X3_MODE>), grid, block, 0, st, xb, M_full, ldx, wb, b_packed, n_pad, k3, yb, ldy, n_pad / pp16::BN, (int)nt); break;
            default: hipLaunchKernelGGL((pp16::linear_bf16_pp_kernel<M360_ACT_SIGMOID, false, M360_X3_MODE>), grid, block, 0, st, xb, M_full, ldx, wb, b_packed, n_pad, k3, yb, ldy, n_pad / pp16::BN, (int)nt); break;
        }
    }
    if (M > M_full) {
        const long Mt = M - M_full;
        dim3 grid((unsigned)((n_pad + 31) / 32), (unsigned)((Mt + 31) / 32)), block(64);
        const __bf16 *xt = xb + M_full * ldx;
        __bf16 *yt = yb + M_full * ldy;
        if (!split_out) {
            if (act == M360_ACT_RELU) hipLaunchKernelGGL((pbf16::linear_bf16_mfma_simple_kernel<M360_ACT_RELU, true, false>), grid, block, 0, st, xt, Mt, ldx, wb, b_packed, n_pad, k3, yt, ldy);
            else hipLaunchKernelGGL((pbf16::linear_bf16_mfma_simple_kernel<M360_ACT_NONE, true, false>), grid, block, 0, st, xt, Mt, ldx, wb, b_packed, n_pad, k3, yt, ldy);
        } else
        switch (act) {
            case M360_ACT_NONE: hipLaunchKernelGGL((pbf16::linear_bf16_mfma_simple_kernel<M360_ACT_NONE, true>), grid, block, 0, st, xt, Mt, ldx, wb, b_packed, n_pad, k3, yt, ldy); break;
            case M360_ACT_RELU: hipLaunchKernelGGL((pbf16::linear_bf16_mfma_simple_kernel<M360_ACT_RELU, true>), grid, block, 0, st, xt, Mt, ldx, wb, b_packed, n_pad, k3, yt, ldy); break;
            default: hipLaunchKernelGGL((pbf16::linear_bf16_mfma_simple_kernel<M360_ACT_SIGMOID, true>), grid, block, 0, st, xt, Mt, ldx, wb, b_packed, n_pad, k3, yt, ldy); break;
        }
    }
    return check_launch("linear_bf16x3");
}

#ifdef M360_DIAG
// ---- diagnostics build only (libm360_diag.so): the two hot kernels instrumented with cycle stamps (ReLU epilogue)
int m360_diag_linear(const float *x, long M, int ldx, const float *w_packed, const float *b_packed, int n_pad, int k_pad,
                     float *y, int ldy, m360_stream_t stream) {
    if (!x || !w_packed || !b_packed || !y || M < persist::BM || M % persist::BM || n_pad % persist::BN || k_pad % BK) return fail(M360_ERR_INVALID_ARGUMENT, "m360_diag_linear: full 256 x 256 tiles only");
    const int cus = cu_count();
    const long nt = (M / persist::BM) * (n_pad / persist::BN);
    dim3 grid((unsigned)(nt < cus ? nt : cus)), block(persist::kThreads);
    hipLaunchKernelGGL((persist::linear_f32_mfma_persist_kernel<M360_ACT_RELU, true>), grid, block, 0, reinterpret_cast<hipStream_t>(stream), x, M, ldx, w_packed, b_packed, n_pad, k_pad, y, ldy, n_pad / persist::BN, (int)nt);
    return check_launch("diag_linear");
}

int m360_diag_linear_bf16(const void *x, long M, int ldx, const void *w_packed, const float *b_packed, int n_pad,
                          int k_pad, void *y, int ldy, int variant, int ldw, m360_stream_t stream) {
    if (!x || !w_packed || !b_packed || !y || M < pp16::BM || M % pp16::BM || n_pad % pp16::BN || k_pad % pp16::BK || k_pad < pp16::BK) return fail(M360_ERR_INVALID_ARGUMENT, "m360_diag_linear_bf16: full 256 x 256 tiles, k_pad a multiple of 64 only");
    const int cus = cu_count();
    const long nt = (M / pp16::BM) * (n_pad / pp16::BN);
    dim3 grid((unsigned)(nt < cus ? nt : cus)), block(pp16::kThreads);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const __bf16 *xb = static_cast<const __bf16 *>(x), *wb = static_cast<const __bf16 *>(w_packed);
    __bf16 *yb = static_cast<__bf16 *>(y);
    if (variant >= 20 && variant < 100) {  // the one-wave-per-SIMD 32x32x16 ring kernel (m360_linear_bf16_w32.hip.h), stamped; 20 + ABL bits
        if (k_pad % 128 || n_pad > w32::kMaxBias) return fail(M360_ERR_INVALID_ARGUMENT, "m360_diag_linear_bf16: variant 20 needs k_pad %% 128 == 0");
        dim3 g4((unsigned)(nt < cus ? nt : cus)), b4(w32::kThreads);
#define M360_W32_ABL(A) hipLaunchKernelGGL((w32::linear_bf16_w32_kernel<M360_ACT_RELU, A, true>), g4, b4, 0, st, xb, M, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / w32::BN, (int)nt)
        switch (variant - 20) {
            case 0: M360_W32_ABL(0); break;
            case 1: M360_W32_ABL(1); break;
            case 2: M360_W32_ABL(2); break;
            case 4: M360_W32_ABL(4); break;
            case 7: M360_W32_ABL(7); break;
            case 16: M360_W32_ABL(16); break;
            case 32: M360_W32_ABL(32); break;
            case 39: M360_W32_ABL(39); break;
            default: return fail(M360_ERR_INVALID_ARGUMENT, "m360_diag_linear_bf16: variant %d", variant);
        }
#undef M360_W32_ABL
        return check_launch("diag_linear_bf16_w32");
    }
    if (variant == 300 || variant == 301) {  // bf16x3 on either kernel (ReLU): x = [hi | lo] rows (ldx >= 2 k_pad), w = [Wh | Wh | Wl], y = [hi | lo]
        if (k_pad < 2 * w16::BKS || ldx < 2 * k_pad || ldy < 2 * n_pad || n_pad > w16::kMaxBias) return fail(M360_ERR_INVALID_ARGUMENT, "m360_diag_linear_bf16: variant 300 needs k_pad >= 128, [hi | lo] strides");
        if (variant == 300) hipLaunchKernelGGL((w16::linear_bf16_w16_kernel<M360_ACT_RELU, 0, false, true, false>), dim3((unsigned)(nt < cus ? nt : cus)), dim3(w16::kThreads), 0, st, xb, M, ldx, wb, b_packed, n_pad, 3 * k_pad, yb, ldy, n_pad / w16::BN, (int)nt);
        else hipLaunchKernelGGL((pp16::linear_bf16_pp_kernel<M360_ACT_RELU, false, 2>), grid, block, 0, st, xb, M, ldx, wb, b_packed, n_pad, 3 * k_pad, yb, ldy, n_pad / pp16::BN, (int)nt);
        return check_launch("diag_linear_bf16x3");
    }
    if (variant >= 100 && variant <= 200) {  // the one-wave-per-SIMD 16x16x32 ring kernel (m360_linear_bf16_w16.hip.h): 100 + ABL bits stamped,
                                             // 200 = the product instantiation
        if (k_pad % 128 || k_pad < 256 || n_pad > w16::kMaxBias) return fail(M360_ERR_INVALID_ARGUMENT, "m360_diag_linear_bf16: variant 100 needs k_pad %% 128 == 0, >= 256");
        dim3 g4((unsigned)(nt < cus ? nt : cus)), b4(w16::kThreads);
#define M360_W16_ABL(A, S) hipLaunchKernelGGL((w16::linear_bf16_w16_kernel<M360_ACT_RELU, A, S>), g4, b4, 0, st, xb, M, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / w16::BN, (int)nt)
        switch (variant - 100) {
            case 0: M360_W16_ABL(0, true); break;
            case 1: M360_W16_ABL(1, true); break;
            case 2: M360_W16_ABL(2, true); break;
            case 4: M360_W16_ABL(4, true); break;
            case 7: M360_W16_ABL(7, true); break;
            case 16: M360_W16_ABL(16, true); break;
            case 32: M360_W16_ABL(32, true); break;
            case 39: M360_W16_ABL(39, true); break;
            case 64: M360_W16_ABL(64, true); break;
            case 28: M360_W16_ABL(128, true); break;  // variant 128: plain instead of non-temporal stores
            case 29: M360_W16_ABL(256, true); break;  // variant 129: non-temporal activation pieces
            case 30: M360_W16_ABL(512, true); break;  // variant 130: non-temporal weight pieces
            case 31: M360_W16_ABL(1024, true); break;        // variant 131: pieces issued, never awaited: what does WAITING for LDS-DMA cost?
            case 33: M360_W16_ABL(1024 + 16, true); break;   // variant 133: the same without the stores
            case 34: M360_W16_ABL(2048, true); break;        // variant 134: the stores rewrite the workgroup's first 256 rows (L2-resident output)
            case 35: M360_W16_ABL(2048 + 128, true); break;  // variant 135: the same with plain (temporal) stores
            case 36: hipLaunchKernelGGL((w16::linear_bf16_w16_kernel<M360_ACT_RELU, 0, true, false, false, 0, false, true>), g4, b4, 0, st, xb, M, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / w16::BN, (int)nt); break;  // variant 136: the LDS epilogue, stamped (results correct)
            case 37: hipLaunchKernelGGL((w16::linear_bf16_w16_kernel<M360_ACT_RELU, 16, true, false, false, 0, false, true>), g4, b4, 0, st, xb, M, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / w16::BN, (int)nt); break;  // variant 137: ... without its stores
            case 40: hipLaunchKernelGGL((w16::linear_bf16_w16_kernel<M360_ACT_RELU, 0, true, false, false, 0, false, false, true>), g4, b4, 0, st, xb, M, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / w16::BN, (int)nt, nullptr, nullptr, 1, g_diag_stagger); break;   // variant 140: paired rows in and out (y comes out paired, x is read as if it were)
            case 41: hipLaunchKernelGGL((w16::linear_bf16_w16_kernel<M360_ACT_RELU, 16, true, false, false, 0, false, false, true>), g4, b4, 0, st, xb, M, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / w16::BN, (int)nt, nullptr, nullptr, 1); break;  // variant 141: ... without the stores
            case 46: hipLaunchKernelGGL((w16::linear_bf16_w16_kernel<M360_ACT_RELU, 16384, true, false, false, 0, false, false, true>), g4, b4, 0, st, xb, M, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / w16::BN, (int)nt, nullptr, nullptr, 1); break;  // variant 146: paired rows, epilogue of stores only
            case 48: hipLaunchKernelGGL((w16::linear_bf16_w16_kernel<M360_ACT_RELU, 16384 + 128, true, false, false, 0, false, false, true>), g4, b4, 0, st, xb, M, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / w16::BN, (int)nt, nullptr, nullptr, 1); break;  // variant 148: ... of plain (temporal) stores only
            case 47: M360_W16_ABL(16384, true); break;       // variant 147: plain rows, epilogue of exchange + stores only
            case 42: M360_W16_ABL(4096, true); break;        // variant 142: agent-scope non-temporal stores (sc1 nt)
            case 43: M360_W16_ABL(4096 + 128, true); break;  // variant 143: agent-scope stores (sc1)
            case 44: M360_W16_ABL(8192, true); break;        // variant 144: system-scope non-temporal stores (sc0 sc1 nt)
            case 45: M360_W16_ABL(8192 + 128, true); break;  // variant 145: system-scope stores (sc0 sc1)
            case 100: M360_W16_ABL(0, false); break;
            case 51: M360_W16_ABL(32768, true); break;  // variant 151: + 8 selects per piece on an SGPR-pair mask (a ReLU bit tape APPLIED in the epilogue; results correct)
            case 52: M360_W16_ABL(65536, true); break;  // variant 152: + 8 v_cmp per piece into SGPR pairs (the tape PRODUCED by the epilogue; results correct)
            case 50: hipLaunchKernelGGL((w16::linear_bf16_w16_kernel<M360_ACT_SIGMOID, 16, true>), g4, b4, 0, st, xb, M, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / w16::BN, (int)nt); break;  // sigmoid epilogue, no stores: what would a last layer cost here?
            default: return fail(M360_ERR_INVALID_ARGUMENT, "m360_diag_linear_bf16: variant %d", variant);
        }
#undef M360_W16_ABL
        return check_launch("diag_linear_bf16_w16");
    }
    switch (variant) {  // ReLU epilogue throughout
        case 0: hipLaunchKernelGGL((pp16::linear_bf16_pp_kernel<M360_ACT_RELU, true>), grid, block, 0, st, xb, M, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / pp16::BN, (int)nt); break;
        case 1: hipLaunchKernelGGL((sp16::linear_bf16_sp_kernel<M360_ACT_RELU, true>), grid, block, 0, st, xb, M, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / sp16::BN, (int)nt, ldw); break;
        case 2: hipLaunchKernelGGL((sp16::linear_bf16_sp_kernel<M360_ACT_RELU, false>), grid, block, 0, st, xb, M, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / sp16::BN, (int)nt, ldw); break;
        case 4: hipLaunchKernelGGL((sp16::linear_bf16_sp_kernel<M360_ACT_RELU, false, 1>), grid, block, 0, st, xb, M, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / sp16::BN, (int)nt, ldw); break;
        case 5: hipLaunchKernelGGL((sp16::linear_bf16_sp_kernel<M360_ACT_RELU, false, 2>), grid, block, 0, st, xb, M, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / sp16::BN, (int)nt, ldw); break;
        case 6: hipLaunchKernelGGL((sp16::linear_bf16_sp_kernel<M360_ACT_RELU, false, 4>), grid, block, 0, st, xb, M, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / sp16::BN, (int)nt, ldw); break;
        case 7: hipLaunchKernelGGL((sp16::linear_bf16_sp_kernel<M360_ACT_RELU, false, 6>), grid, block, 0, st, xb, M, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / sp16::BN, (int)nt, ldw); break;
#define M360_RG(ST, AB) hipLaunchKernelGGL((rg16::linear_bf16_rg_kernel<M360_ACT_RELU, ST, AB>), grid, block, 0, st, xb, M, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / rg16::BN, (int)nt, ldw)
        case 8: M360_RG(true, 0); break;
        case 9: M360_RG(false, 0); break;
        case 10: M360_RG(false, 1); break;
        case 11: M360_RG(false, 2); break;
        case 12: M360_RG(false, 4); break;
        case 13: M360_RG(false, 6); break;
#undef M360_RG
        case 3: hipLaunchKernelGGL((pp16::linear_bf16_pp_kernel<M360_ACT_RELU, false>), grid, block, 0, st, xb, M, ldx, wb, b_packed, n_pad, k_pad, yb, ldy, n_pad / pp16::BN, (int)nt); break;
        default: return fail(M360_ERR_INVALID_ARGUMENT, "m360_diag_linear_bf16: variant %d", variant);
    }
    return check_launch("diag_linear_bf16");
}

int m360_diag_linear_hd(const float *x, long M, int ldx, const float *w_packed, const float *b_packed, int n_pad, int k_pad,
                        int act, float *y, int ldy, int ablate, unsigned *queue, m360_stream_t stream) {
    if (!x || !w_packed || !b_packed || !y || M < hd::BM || M % hd::BM || n_pad % hd::BN || k_pad % hd::BK || k_pad < 2 * hd::BK || n_pad > hd::kMaxBias || (act != M360_ACT_NONE && act != M360_ACT_RELU))
        return fail(M360_ERR_INVALID_ARGUMENT, "m360_diag_linear_hd: full 128 x 256 tiles, k_pad >= 64, act none / ReLU only");
    const int cus = cu_count();
    const long nt = (M / hd::BM) * (n_pad / hd::BN);
    dim3 grid((unsigned)(nt < cus ? nt : cus)), block(hd::kThreads);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool even_ = (k_pad / hd::BK) % 2 == 0;
    const long share = nt / (cus > 0 ? cus : 1);  // the split of launch_linear_hd
    int n_static = 0;
    if (queue && share >= 4) n_static = (int)(share - (share / 16 > 2 ? share / 16 : 2));
    else queue = nullptr;
#define M360_HD_ABL(A) do { if (even_) hipLaunchKernelGGL((hd::linear_f32_hd_kernel<M360_ACT_RELU, true, A, true>), grid, block, 0, st, x, M, ldx, w_packed, b_packed, n_pad, k_pad, y, ldy, n_pad / hd::BN, (int)nt, queue, n_static); else hipLaunchKernelGGL((hd::linear_f32_hd_kernel<M360_ACT_RELU, false, A, true>), grid, block, 0, st, x, M, ldx, w_packed, b_packed, n_pad, k_pad, y, ldy, n_pad / hd::BN, (int)nt, queue, n_static); } while (0)
    if (ablate || act == M360_ACT_RELU) {  // stamped; timing-only ablations: bits as listed in m360_linear_hd.hip.h (the combinations instantiated here)
        switch (ablate) {
            case 0: M360_HD_ABL(0); break;
            case 1: M360_HD_ABL(1); break;
            case 2: M360_HD_ABL(2); break;
            case 7: M360_HD_ABL(7); break;
            case 16: M360_HD_ABL(16); break;
            case 64: M360_HD_ABL(64); break;
            case 128: M360_HD_ABL(128); break;
            default: return fail(M360_ERR_INVALID_ARGUMENT, "m360_diag_linear_hd: ablate=%d", ablate);
        }
        return check_launch("diag_linear_hd");
    }
#undef M360_HD_ABL
    return launch_linear_hd(x, M, ldx, w_packed, b_packed, n_pad, k_pad, act, y, ldy, queue, st);
}

int m360_diag_force_linear_kernel(int which) {
    if (which < 0 || which > 2) return fail(M360_ERR_INVALID_ARGUMENT, "m360_diag_force_linear_kernel: 0 rule, 1 full tiles, 2 half tiles");
    g_diag_force_kernel = which;
    return M360_OK;
}

int m360_diag_read_w32_stamps(unsigned long long *out_host, int n) {
    if (!out_host || n < 0 || n > 256 * 4) return fail(M360_ERR_INVALID_ARGUMENT, "m360_diag_read_w32_stamps: bad argument");
    if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(w32::g_w32_stamps), sizeof(unsigned long long) * n) != hipSuccess)
        return fail(M360_ERR_LAUNCH, "m360_diag_read_w32_stamps: copy failed");
    return M360_OK;
}

int m360_diag_read_w16_stamps(unsigned long long *out_host, int n) {
    if (!out_host || n < 0 || n > 256 * 4) return fail(M360_ERR_INVALID_ARGUMENT, "m360_diag_read_w16_stamps: bad argument");
    if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(w16::g_w16_stamps), sizeof(unsigned long long) * n) != hipSuccess)
        return fail(M360_ERR_LAUNCH, "m360_diag_read_w16_stamps: copy failed");
    return M360_OK;
}

int m360_diag_read_hd_stamps(unsigned long long *out_host, int n) {
    if (!out_host || n < 0 || n > 2 * 256 * 4) return fail(M360_ERR_INVALID_ARGUMENT, "m360_diag_read_hd_stamps: bad argument");
    if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(hd::g_hd_stamps), sizeof(unsigned long long) * n) != hipSuccess)
        return fail(M360_ERR_LAUNCH, "m360_diag_read_hd_stamps: copy failed");
    return M360_OK;
}

int m360_diag_read_stamps(unsigned long long *out_host, int n) {
    if (!out_host || n < 0 || n > 256 * 16) return fail(M360_ERR_INVALID_ARGUMENT, "m360_diag_read_stamps: bad argument");
    if (hipMemcpyFromSymbol(out_host, HIP_SYMBOL(persist::g_stamps), sizeof(unsigned long long) * n) != hipSuccess)
        return fail(M360_ERR_LAUNCH, "m360_diag_read_stamps: copy failed");
    return M360_OK;
}
#endif

}  // extern "C"
