// bf16-input / fp32-accumulate linear kernel (opt-in reduced-precision MLP, BASELINE configs[4]):
// y[M,Np] (bf16) = act(x[M,Kp] (bf16) * W^T (bf16) + b (fp32)) on v_mfma_f32_32x32x16_bf16.
//
// Same skeleton as the fp32 kernel (m360_linear_persist.hip.h): persistent workgroups of 4 waves (one per
// SIMD, 512 registers each), 256 x 256 tiles, 128-byte LDS rows filled by global_load_lds_dwordx4 with the
// source-side XOR swizzle, inline-asm ds_read_b128 with waits tied to the fragment registers, LDS-transposed
// wide-store epilogue.  Differences: a K-step is 64 bf16 (the same 128 bytes per row), one 16-byte chunk per
// lane IS one MFMA operand (lane (r,h) holds k = 8h..8h+7 of row r), so a K-group is 16 MFMAs of 32 cycles,
// and everything that is not an MFMA is 8x more expensive relative to the matrix work than in fp32.
#pragma once
#include "m360_common.hip.h"

namespace m360 {
namespace pbf16 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 256, BK = 64;  // BK in bf16 elements = 128 bytes per row
constexpr int kThreads = 256;
constexpr int TM = 4, TN = 4;
constexpr int kTileBytes = 256 * 128;      // one operand tile of one K-step
constexpr int kBufBytes = 2 * kTileBytes;  // A + B
constexpr int kDma = 8;
typedef __attribute__((address_space(3))) void *lds_ptr_t;

#define B16_INL __attribute__((always_inline))

template <int ACT>
__device__ __forceinline__ float act_fn(float v) {
    if (ACT == M360_ACT_RELU) return relu_nanf_(v);
    if (ACT == M360_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    return v;
}

template <int ACT>
__global__ __launch_bounds__(kThreads, 1) void linear_bf16_mfma_persist_kernel(
    const __bf16 *__restrict__ X, long M, int ldx, const __bf16 *__restrict__ W, const float *__restrict__ bias,
    int Np, int Kp, __bf16 *__restrict__ Y, int ldy, int tiles_n, int ntiles) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * kBufBytes];  // 128 KiB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int G = gridDim.x;
    const int ksteps = Kp / BK;

    auto tile_coords = [&](int lin_id, long &m0, int &n0) B16_INL {
        const int full = (ntiles / 8) * 8;
        int lin = lin_id;
        if (lin_id < full) lin = (lin_id % 8) * (full / 8) + lin_id / 8;  // XCD-aware (speed only)
        m0 = (long)(lin / tiles_n) * BM;
        n0 = (lin % tiles_n) * BN;
    };

    // ---- LDS-DMA: wave w fills rows [64w, 64w+64) of A and of B (8 instructions of 8 rows x 128 B each)
    const int st_r = wave * 64 + (lane >> 3);
    const __bf16 *ga[kDma];
    const __bf16 *gb[kDma];
    auto set_load_tile = [&](long m0, int n0) B16_INL {
#pragma unroll
        for (int q = 0; q < kDma; ++q) {
            const int r = st_r + 8 * q;
            const int chunk = (lane & 7) ^ ((r >> 1) & 7);
            ga[q] = X + (m0 + r) * ldx + 8 * chunk;
            gb[q] = W + (long)(n0 + r) * Kp + 8 * chunk;
        }
    };
    char *const dma_dst = smem + wave * 64 * 128;
    auto issue_dma = [&](int buf, int k0, int q, int which) B16_INL {
        char *dstA = dma_dst + buf * kBufBytes + q * 8 * 128;
        if (which & 1) __builtin_amdgcn_global_load_lds(ga[q] + k0, (lds_ptr_t)dstA, 16, 0, 0);
        if (which & 2) __builtin_amdgcn_global_load_lds(gb[q] + k0, (lds_ptr_t)(dstA + kTileBytes), 16, 0, 0);
    };

    // ---- operand reads: lane (l31, h), K-group g (= MFMA k-substep) reads chunk 2g+h of its rows
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const int fsw = (l31 >> 1) & 7;
    unsigned a_addr[4], b_addr[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int slot = ((2 * g + h) ^ fsw) * 16;
        a_addr[g] = lds0 + (wm * 128 + l31) * 128 + slot;
        b_addr[g] = lds0 + kTileBytes + (wn * 128 + l31) * 128 + slot;
    }

    f32x16 acc[TM][TN];
    bf16x8 fa_a[TM], fa_b[TN], fb_a[TM], fb_b[TN];

#define B16_DS128(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr))
#define B16_READ(FA, FB, aa_, bb_)    \
    do {                              \
        B16_DS128(FA[0], aa_, 0);     \
        B16_DS128(FA[1], aa_, 4096);  \
        B16_DS128(FA[2], aa_, 8192);  \
        B16_DS128(FA[3], aa_, 12288); \
        B16_DS128(FB[0], bb_, 0);     \
        B16_DS128(FB[1], bb_, 4096);  \
        B16_DS128(FB[2], bb_, 8192);  \
        B16_DS128(FB[3], bb_, 12288); \
    } while (0)
#define B16_WAIT_FRAG(FA, FB)                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                             \
                 : "+v"(FA[0]), "+v"(FA[1]), "+v"(FA[2]), "+v"(FA[3]), "+v"(FB[0]), "+v"(FB[1]),    \
                   "+v"(FB[2]), "+v"(FB[3])::"memory")
#define B16_SB() __builtin_amdgcn_sched_barrier(0)
// 2 MFMAs: row-block i, column-blocks j0, j0+1
#define B16_MFMA2(FA, FB, i, j0)                                                                        \
    do {                                                                                                \
        acc[i][j0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[i], FB[j0], acc[i][j0], 0, 0, 0);       \
        acc[i][(j0) + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[i], FB[(j0) + 1], acc[i][(j0) + 1], 0, 0, 0); \
    } while (0)
// one K-group: 8 units of [2 MFMA -> 1 ds_read_b128 of the next group (-> DMA row-block q = unit when DMA != 0)]
#define B16_UNIT(FA, FB, i, j0, RD, DMA, q)              \
    do {                                                 \
        B16_MFMA2(FA, FB, i, j0);                        \
        RD;                                              \
        if (DMA) issue_dma(buf ^ 1, next_k0, q, DMA);    \
        B16_SB();                                        \
    } while (0)
#define B16_GROUP_PIPE(FA, FB, NA, NB, na, nb, DMA)                            \
    do {                                                                       \
        B16_UNIT(FA, FB, 0, 0, B16_DS128(NA[0], na, 0), DMA, 0);               \
        B16_UNIT(FA, FB, 0, 2, B16_DS128(NA[1], na, 4096), DMA, 1);            \
        B16_UNIT(FA, FB, 1, 0, B16_DS128(NA[2], na, 8192), DMA, 2);            \
        B16_UNIT(FA, FB, 1, 2, B16_DS128(NA[3], na, 12288), DMA, 3);           \
        B16_UNIT(FA, FB, 2, 0, B16_DS128(NB[0], nb, 0), DMA, 4);               \
        B16_UNIT(FA, FB, 2, 2, B16_DS128(NB[1], nb, 4096), DMA, 5);            \
        B16_UNIT(FA, FB, 3, 0, B16_DS128(NB[2], nb, 8192), DMA, 6);            \
        B16_UNIT(FA, FB, 3, 2, B16_DS128(NB[3], nb, 12288), DMA, 7);           \
    } while (0)

    int lin_id = blockIdx.x;
    if (lin_id >= ntiles) return;
    long m0;
    int n0;
    tile_coords(lin_id, m0, n0);
    set_load_tile(m0, n0);
#pragma unroll
    for (int q = 0; q < kDma; ++q) issue_dma(0, 0, q, 3);
    int buf = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    B16_SB();
    B16_READ(fa_a, fa_b, a_addr[0], b_addr[0]);  // loop invariant: group 0 of the current step is in flight
    B16_SB();

    for (; lin_id < ntiles; lin_id += G) {
        tile_coords(lin_id, m0, n0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

        for (int kt = 0; kt < ksteps; ++kt) {
            int next_k0 = (kt + 1) * BK;
            if (kt + 1 == ksteps) {  // next K-step = first one of this workgroup's next tile (if any)
                next_k0 = 0;
                if (lin_id + G < ntiles) {
                    long nm0;
                    int nn0;
                    tile_coords(lin_id + G, nm0, nn0);
                    set_load_tile(nm0, nn0);
                }
            }
            const unsigned boff = buf ? (unsigned)kBufBytes : 0u;
            const unsigned noff = buf ? 0u : (unsigned)kBufBytes;
            const unsigned a1 = a_addr[1] + boff, b1 = b_addr[1] + boff, a2 = a_addr[2] + boff, b2 = b_addr[2] + boff;
            const unsigned a3 = a_addr[3] + boff, b3 = b_addr[3] + boff, a0n = a_addr[0] + noff, b0n = b_addr[0] + noff;
            B16_SB();
            B16_WAIT_FRAG(fa_a, fa_b);
            B16_SB();
            B16_GROUP_PIPE(fa_a, fa_b, fb_a, fb_b, a1, b1, 3);  // group 0 (+ reads of group 1, + DMA of step t+1)
            B16_WAIT_FRAG(fb_a, fb_b);
            B16_SB();
            B16_GROUP_PIPE(fb_a, fb_b, fa_a, fa_b, a2, b2, 0);  // group 1 (+ reads of group 2)
            B16_WAIT_FRAG(fa_a, fa_b);
            B16_SB();
            B16_GROUP_PIPE(fa_a, fa_b, fb_a, fb_b, a3, b3, 0);  // group 2 (+ reads of group 3)
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier"
                         : "+v"(fb_a[0]), "+v"(fb_a[1]), "+v"(fb_a[2]), "+v"(fb_a[3]), "+v"(fb_b[0]), "+v"(fb_b[1]),
                           "+v"(fb_b[2]), "+v"(fb_b[3])::"memory");
            B16_SB();
            B16_GROUP_PIPE(fb_a, fb_b, fa_a, fa_b, a0n, b0n, 0);  // group 3 (+ reads of the next step's group 0)
            buf ^= 1;
        }

        // ---- epilogue: two 32 x 32 accumulator blocks (64 columns) are transposed through the wave's private
        // slice of the idle LDS buffer; each lane then owns 8 consecutive columns of one row -> bias, activation,
        // bf16 conversion, one 16-byte store (8 rows x 128 B per instruction)
        {
            int ldy_t = ldy;
            asm volatile("" : "+s"(ldy_t));
            __bf16 *__restrict__ Yt = Y + m0 * ldy_t + n0;
            float *stg = reinterpret_cast<float *>(dma_dst + (buf ^ 1) * kBufBytes);  // [32][64] fp32 = 8 KiB
            const int rrow = lane >> 3, rcol = 8 * (lane & 7);
#pragma unroll
            for (int jp = 0; jp < TN / 2; ++jp) {
                const float *bp = bias + n0 + wn * 128 + jp * 64 + rcol;
                const float4 b0 = *reinterpret_cast<const float4 *>(bp), b1 = *reinterpret_cast<const float4 *>(bp + 4);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            stg[((r & 3) + 8 * (r >> 2) + 4 * h) * 64 + jj * 32 + l31] = acc[i][2 * jp + jj][r];
                    __bf16 *__restrict__ Yc = Yt + (long)(wm * 128 + i * 32 + rrow) * ldy_t + wn * 128 + jp * 64 + rcol;
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const float *sp = stg + (p * 8 + rrow) * 64 + rcol;
                        const float4 v0 = *reinterpret_cast<const float4 *>(sp), v1 = *reinterpret_cast<const float4 *>(sp + 4);
                        bf16x8 o;
                        o[0] = (__bf16)act_fn<ACT>(v0.x + b0.x);
                        o[1] = (__bf16)act_fn<ACT>(v0.y + b0.y);
                        o[2] = (__bf16)act_fn<ACT>(v0.z + b0.z);
                        o[3] = (__bf16)act_fn<ACT>(v0.w + b0.w);
                        o[4] = (__bf16)act_fn<ACT>(v1.x + b1.x);
                        o[5] = (__bf16)act_fn<ACT>(v1.y + b1.y);
                        o[6] = (__bf16)act_fn<ACT>(v1.z + b1.z);
                        o[7] = (__bf16)act_fn<ACT>(v1.w + b1.w);
                        *reinterpret_cast<bf16x8 *>(Yc + (long)(p * 8) * ldy_t) = o;
                    }
                    B16_SB();
                }
            }
        }
    }
#undef B16_DS128
#undef B16_READ
#undef B16_WAIT_FRAG
#undef B16_SB
#undef B16_MFMA2
#undef B16_UNIT
#undef B16_GROUP_PIPE
}

// Generic (slow, any shape) bf16 kernel for ragged rows / narrow layers: one wave per 32 x 32 output block,
// operands straight from global memory (16 B per lane per MFMA), rows clamped, stores predicated.
// X3: the bf16x3 contraction of m360_linear_bf16_pp.hip.h (Kp = 3K, activation column wraps at 2K, output [hi(Np) | lo(Np)]).
// SPLIT: a plain contraction over bf16 rows whose output is written as [hi(Np) | lo(Np)] (m360_linear_bf16_split: the x6 first
// layer of the bf16x3 mode).
template <int ACT, bool X3 = false, bool SPLIT = X3>
__global__ __launch_bounds__(64) void linear_bf16_mfma_simple_kernel(
    const __bf16 *__restrict__ X, long M, int ldx, const __bf16 *__restrict__ W, const float *__restrict__ bias,
    int Np, int Kp, __bf16 *__restrict__ Y, int ldy) {
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    const long m0 = (long)blockIdx.y * 32;
    const int n0 = blockIdx.x * 32;
    long ra = m0 + l31;
    if (ra > M - 1) ra = M - 1;
    int rb = n0 + l31;
    if (rb > Np - 1) rb = Np - 1;
    const __bf16 *xa = X + ra * ldx + 8 * h;
    const __bf16 *wb = W + (long)rb * Kp + 8 * h;
    const int x_wrap = X3 ? 2 * (Kp / 3) : 0;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int k = 0; k < Kp; k += 16) {
        const bf16x8 a = *reinterpret_cast<const bf16x8 *>(xa + ((X3 && k >= x_wrap) ? k - x_wrap : k));
        const bf16x8 b = *reinterpret_cast<const bf16x8 *>(wb + k);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    const int col = n0 + l31;
    if (col >= Np) return;
    const float bj = bias[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long row = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < M) {
            const float v = act_fn<ACT>(acc[r] + bj);
            if (SPLIT) {
                __bf16 hi, lo;
                split_bf16_(v, hi, lo);
                Y[row * ldy + col] = hi;
                Y[row * ldy + Np + col] = lo;
            } else {
                Y[row * ldy + col] = (__bf16)v;
            }
        }
    }
}

// bf16x3 weights: [n_pad, 3 k_pad] = [Wh | Wh | Wl] (zero padded), Wh = bf16(W), Wl = bf16(W - Wh); fp32 bias
__global__ void pack_linear_bf16x3_kernel(const float *__restrict__ w, const float *__restrict__ b, int n_out, int k_in,
                                          int n_pad, int k_pad, __bf16 *__restrict__ wp, float *__restrict__ bp) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)n_pad * k_pad) {
        const int n = (int)(idx / k_pad), k = (int)(idx % k_pad);
        const float v = (n < n_out && k < k_in) ? canon_nanf_(w[(long)n * k_in + k]) : 0.0f;
        __bf16 hi, lo;
        split_bf16_(v, hi, lo);
        __bf16 *row = wp + (long)n * 3 * k_pad;
        row[k] = hi;
        row[k_pad + k] = hi;
        row[2 * k_pad + k] = lo;
    }
    if (bp != nullptr && idx < n_pad) bp[idx] = (b != nullptr && idx < n_out) ? canon_nanf_(b[idx]) : 0.0f;
}

// "x6" weights of a first layer: [n_pad, 6 k_pad] = [Wh | Wm | Wl | Wh | Wm | Wh] (three bf16 terms per weight, split3_bf16_),
// the blocks matched to the encoder's x6 rows [xl | xm | xh | xm | xh | xh]: one plain bf16 contraction of length 6 k_pad forms
// xl wh + xm wm + xh wl + xm wh + xh wm + xh wh, small terms first
__global__ void pack_linear_bf16x6_kernel(const float *__restrict__ w, const float *__restrict__ b, int n_out, int k_in,
                                          int n_pad, int k_pad, __bf16 *__restrict__ wp, float *__restrict__ bp) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)n_pad * k_pad) {
        const int n = (int)(idx / k_pad), k = (int)(idx % k_pad);
        const float v = (n < n_out && k < k_in) ? canon_nanf_(w[(long)n * k_in + k]) : 0.0f;
        __bf16 hi, mid, lo;
        split3_bf16_(v, hi, mid, lo);
        __bf16 *row = wp + (long)n * 6 * k_pad + k;
        row[0] = hi;
        row[k_pad] = mid;
        row[2 * k_pad] = lo;
        row[3 * k_pad] = hi;
        row[4 * k_pad] = mid;
        row[5 * k_pad] = hi;
    }
    if (bp != nullptr && idx < n_pad) bp[idx] = (b != nullptr && idx < n_out) ? canon_nanf_(b[idx]) : 0.0f;
}

__global__ void pack_linear_bf16_kernel(const float *__restrict__ w, const float *__restrict__ b, int n_out, int k_in,
                                        int n_pad, int k_pad, __bf16 *__restrict__ wp, float *__restrict__ bp) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)n_pad * k_pad) {
        const int n = (int)(idx / k_pad), k = (int)(idx % k_pad);
        wp[idx] = (__bf16)((n < n_out && k < k_in) ? canon_nanf_(w[(long)n * k_in + k]) : 0.0f);
    }
    if (bp != nullptr && idx < n_pad) bp[idx] = (b != nullptr && idx < n_out) ? canon_nanf_(b[idx]) : 0.0f;
}

}  // namespace pbf16
}  // namespace m360
