// Weight-gradient GEMM of the training path:  dW[Np, Kp] = dZ[M, Np]^T * X[M, Kp]  in exact
// fp32 on the matrix cores (v_mfma_f32_32x32x2_f32).  What autograd computes for nn.Linear's weight in the reference
// (model.py:43-53,131-158 under train.py:62,80).
//
// The contraction runs over the M = rays x samples rows (524 288 at the BASELINE shape) while the output is only
// Np x Kp (<= 1024 x 1024 = 16 tiles of 256 x 256), so the rows are split: workgroup (tile, split) reduces its slice
// of rows into a private partial tile, a second kernel adds the partials in a FIXED order (deterministic, no atomics).
//
//   * 256 threads = 4 waves (one per SIMD, whole register file), wave tile 128 x 128 = 4 x 4 MFMA tiles.
//   * a K-step is 32 rows of dZ and of X (256 columns each): 32 + 32 LDS-DMA instructions of one full 1 KiB row each
//     (global_load_lds_dwordx4, no staging registers), double-buffered in 128 KiB of LDS.
//   * both operands are "row = contraction index" in memory, which the MFMA wants transposed.  No transposition is
//     needed: lane (i = l & 31, h = l >> 5) reads ONE ds_read_b128 at row 2p+h, columns 4i..4i+3 and uses component q
//     as MFMA block q, i.e. block q of the wave tile holds output rows/columns {4i + q}.  That is a permutation of the
//     output only; the epilogue undoes it for free (4 column blocks of one lane = 4 consecutive columns = one 16-byte
//     store).  Reads are conflict-free (16 lanes cover 256 contiguous bytes).
//   * ids are XCD-aware: the workgroups of one split (same rows, all tiles) are neighbours on one XCD's L2.
#pragma once
#include "m360_common.hip.h"
#include "m360_linear_persist.hip.h"

namespace m360 {
namespace tn {

using persist::f32x16;
using persist::f32x4;
using persist::lds_ptr_t;

constexpr int BT = 256;                    // output tile edge (Np and Kp direction)
constexpr int BKM = 32;                    // rows (contraction) per K-step
constexpr int kThreads = 256;
constexpr int kTileFloats = BKM * BT;      // one operand of one K-step: 32 KiB
constexpr int kBufFloats = 2 * kTileFloats;
constexpr int kMaxWorkgroups = 256;        // tiles x splits target (one per CU)

__global__ __launch_bounds__(kThreads, 1) void linear_tn_kernel(
    const float *__restrict__ dZ, int ldz, const float *__restrict__ X, int ldx, int Np, int Kp,
    float *__restrict__ partial /*[nsplit][Np][Kp]*/, int tiles_k, int ntiles, int nsplit, long total_steps,
    long steps_per_split, float *__restrict__ bias_partial /*[nsplit][Np] or nullptr*/) {
    __shared__ __attribute__((aligned(1024))) float smem[2 * kBufFloats];  // 128 KiB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

    // XCD-aware id: ids sharing id % 8 (one XCD) take a contiguous range of (split, tile) pairs, tile fastest
    const int total = ntiles * nsplit;
    const int full = (total / 8) * 8;
    int lin = blockIdx.x;
    if (lin < full) lin = (lin % 8) * (full / 8) + lin / 8;
    const int split = lin / ntiles, tile = lin % ntiles;
    const int n0 = (tile / tiles_k) * BT, k0 = (tile % tiles_k) * BT;
    const long s_begin = (long)split * steps_per_split;
    long s_end = s_begin + steps_per_split;
    if (s_end > total_steps) s_end = total_steps;

    // bias gradient (column sums of dZ) for free: the workgroups of the first tile column (k0 == 0) also add up the
    // dZ tile they stage in LDS anyway, thread t = column n0 + t, two rows per MFMA pair (ds_read_b32 in the MFMA shadow)
    const bool do_bias = bias_partial != nullptr && k0 == 0;
    float bsum = 0.0f, c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, c3 = 0.0f;

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    if (s_begin < s_end) {
        // ---- LDS-DMA: wave w fills rows [8w, 8w+8) of the dZ tile and of the X tile, one 1 KiB row per instruction
        // (lane L = 16-byte chunk L of the row).  Columns beyond the matrix read column 0 instead (finite values that
        // only reach accumulators which are never stored).
        const int ca = (n0 + 4 * lane < Np) ? n0 + 4 * lane : 0;
        const int cb = (k0 + 4 * lane < Kp) ? k0 + 4 * lane : 0;
        // buffer descriptors based at this workgroup's first row (SGPRs), one 32-bit lane offset per operand, the row
        // advance in the instruction's scalar offset: no vector address arithmetic in the loop
        __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(dZ + (s_begin * BKM + wave * 8) * ldz), 0, 0x7fffffff, 0x00020000);
        __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X + (s_begin * BKM + wave * 8) * ldx), 0, 0x7fffffff, 0x00020000);
        const unsigned va = 4u * ca, vb = 4u * cb;
        unsigned soff_a = 0, soff_b = 0;  // bytes: row offset of the K-step being staged
        float *const dma_dst = smem + wave * 8 * BT;
        auto issue_dma = [&](int buf, int q) __attribute__((always_inline)) {
            float *dst = dma_dst + buf * kBufFloats + q * BT;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)dst, 16, va, soff_a + 4u * q * ldz, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(dst + kTileFloats), 16, vb, soff_b + 4u * q * ldx, 0, 0);
        };

        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)smem;
        const unsigned a_base = lds0 + 4u * (h * BT + wm * 128 + 4 * l31);
        const unsigned b_base = lds0 + 4u * (kTileFloats + h * BT + wn * 128 + 4 * l31);

        f32x4 a0, b0, a1, b1;  // fragment double buffer: (a0, b0) even pairs, (a1, b1) odd pairs
        const unsigned c_base = lds0 + 4u * tid;

#define TN_DS128(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr))
#define TN_DS32(dst, addr, imm) asm volatile("ds_read_b32 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr))
#define TN_WAIT(FA, FB, CX, CY) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(FA), "+v"(FB), "+v"(CX), "+v"(CY)::"memory")
#define TN_SB() __builtin_amdgcn_sched_barrier(0)
#define TN_MFMA16(FA, FB)                                                                         \
    do {                                                                                          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[i], FB[j], acc[i][j], 0, 0, 0);   \
    } while (0)
// pair p of the K-step: wait for its fragment, issue the reads of pair p+1 (byte offset imm = (p+1) * 2048), optionally
// one DMA row pair of the next K-step, then 16 MFMAs that cover the latency of both.  Every thread also loads rows
// 2p, 2p+1 of ITS column of the dZ tile (LX, LY; byte offsets r0, r1) and consumes the two values (CX, CY) loaded by the
// previous pair: the bias gradient, unconditionally (no branch in the MFMA stream; only k0 == 0 workgroups store it).
#define TN_PAIR(FA, FB, NA, NB, imm, DMAQ, CX, CY, LX, LY, r0, r1) \
    do {                                                 \
        TN_WAIT(FA, FB, CX, CY);                         \
        TN_SB();                                         \
        TN_DS128(NA, a_cur, imm);                        \
        TN_DS128(NB, b_cur, imm);                        \
        TN_DS32(LX, c_cur, r0);                          \
        TN_DS32(LY, c_cur, r1);                          \
        if (DMAQ >= 0) issue_dma(buf ^ 1, DMAQ);         \
        TN_SB();                                         \
        bsum += CX + CY;                                 \
        TN_MFMA16(FA, FB);                               \
        TN_SB();                                         \
    } while (0)

#pragma unroll
        for (int q = 0; q < 8; ++q) issue_dma(0, q);
        int buf = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        TN_SB();
        TN_DS128(a0, a_base, 0);
        TN_DS128(b0, b_base, 0);
        TN_SB();

        for (long s = s_begin; s < s_end; ++s) {
            // the DMA issued during this step loads step s + 1; after the last step it harmlessly re-loads the last rows
            // (nobody reads them): no branch inside the MFMA stream
            const unsigned adv = (s + 1 < s_end) ? 4u * BKM : 0u;
            soff_a += adv * ldz;
            soff_b += adv * ldx;
            const unsigned a_cur = a_base + (buf ? 4u * kBufFloats : 0u), b_cur = b_base + (buf ? 4u * kBufFloats : 0u);
            const unsigned a_nxt = a_base + (buf ? 0u : 4u * kBufFloats), b_nxt = b_base + (buf ? 0u : 4u * kBufFloats);
            const unsigned c_cur = c_base + (buf ? 4u * kBufFloats : 0u);
            TN_SB();
            TN_PAIR(a0, b0, a1, b1, 2048, 0, c2, c3, c0, c1, 0, 1024);
            TN_PAIR(a1, b1, a0, b0, 4096, 1, c0, c1, c2, c3, 2048, 3072);
            TN_PAIR(a0, b0, a1, b1, 6144, 2, c2, c3, c0, c1, 4096, 5120);
            TN_PAIR(a1, b1, a0, b0, 8192, 3, c0, c1, c2, c3, 6144, 7168);
            TN_PAIR(a0, b0, a1, b1, 10240, 4, c2, c3, c0, c1, 8192, 9216);
            TN_PAIR(a1, b1, a0, b0, 12288, 5, c0, c1, c2, c3, 10240, 11264);
            TN_PAIR(a0, b0, a1, b1, 14336, 6, c2, c3, c0, c1, 12288, 13312);
            TN_PAIR(a1, b1, a0, b0, 16384, 7, c0, c1, c2, c3, 14336, 15360);
            TN_PAIR(a0, b0, a1, b1, 18432, -1, c2, c3, c0, c1, 16384, 17408);
            TN_PAIR(a1, b1, a0, b0, 20480, -1, c0, c1, c2, c3, 18432, 19456);
            TN_PAIR(a0, b0, a1, b1, 22528, -1, c2, c3, c0, c1, 20480, 21504);
            TN_PAIR(a1, b1, a0, b0, 24576, -1, c0, c1, c2, c3, 22528, 23552);
            TN_PAIR(a0, b0, a1, b1, 26624, -1, c2, c3, c0, c1, 24576, 25600);
            TN_PAIR(a1, b1, a0, b0, 28672, -1, c0, c1, c2, c3, 26624, 27648);
            TN_PAIR(a0, b0, a1, b1, 30720, -1, c2, c3, c0, c1, 28672, 29696);
            // pair 15: every read of `buf` by this wave has landed and its DMA of the next step too -> barrier, then
            // the first reads of the next step go out before the last 16 MFMAs (bias: rows 30, 31 are read BEFORE the
            // barrier, they belong to the buffer that the next step's DMA overwrites)
            TN_DS32(c2, c_cur, 30720);
            TN_DS32(c3, c_cur, 31744);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" : "+v"(a1), "+v"(b1), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3)::"memory");
            TN_SB();
            TN_DS128(a0, a_nxt, 0);
            TN_DS128(b0, b_nxt, 0);
            TN_SB();
            bsum += c0 + c1;
            c0 = 0.0f;
            c1 = 0.0f;
            TN_MFMA16(a1, b1);
            TN_SB();
            buf ^= 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(b0)::"memory");  // drain the speculative last reads
        bsum += c2 + c3;  // rows 30, 31 of the last step
#undef TN_PAIR
#undef TN_MFMA16
#undef TN_SB
#undef TN_WAIT
#undef TN_DS32
#undef TN_DS128
    }

    // ---- epilogue: block (q, q2) register r of lane (l31, h) is output row n0 + wm*128 + 4*i + q with
    // i = (r&3) + 8(r>>2) + 4h, column k0 + wn*128 + 4*l31 + q2: the four q2 blocks form one 16-byte store
    if (do_bias && n0 + tid < Np) bias_partial[(long)split * Np + n0 + tid] = bsum;
    float *__restrict__ P = partial + (long)split * Np * Kp;
    const int col = k0 + wn * 128 + 4 * l31;
    if (col < Kp) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm * 128 + 4 * ((r & 3) + 8 * (r >> 2) + 4 * h) + q;
                if (n < Np) {
                    float4 v;
                    v.x = acc[q][0][r];
                    v.y = acc[q][1][r];
                    v.z = acc[q][2][r];
                    v.w = acc[q][3][r];
                    *reinterpret_cast<float4 *>(P + (long)n * Kp + col) = v;
                }
            }
        }
    }
}

// grad_w[n][k] = sum_s partial[s][n][k] (s ascending) + the < 32 tail rows the K-steps did not cover
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float *__restrict__ partial, int nsplit, int Np, int Kp,
                                                        const float *__restrict__ dZ, int ldz,
                                                        const float *__restrict__ X, int ldx, long m_begin, long M,
                                                        float *__restrict__ grad_w) {
    const long idx4 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long count = (long)Np * Kp;
    if (idx4 * 4 >= count) return;
    const int n = (int)((idx4 * 4) / Kp), k = (int)((idx4 * 4) % Kp);
    float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int p = 0; p < nsplit; ++p) {
        const float4 v = *reinterpret_cast<const float4 *>(partial + (long)p * count + idx4 * 4);
        s.x += v.x;
        s.y += v.y;
        s.z += v.z;
        s.w += v.w;
    }
    for (long m = m_begin; m < M; ++m) {
        const float g = dZ[m * ldz + n];
        const float4 x = *reinterpret_cast<const float4 *>(X + m * ldx + k);
        s.x += g * x.x;
        s.y += g * x.y;
        s.z += g * x.z;
        s.w += g * x.w;
    }
    *reinterpret_cast<float4 *>(grad_w + idx4 * 4) = s;
}

// grad_b[n] = sum_s bias_partial[s][n] (s ascending) + the tail rows
__global__ __launch_bounds__(256) void tn_bias_reduce_kernel(const float *__restrict__ bias_partial, int nsplit, int Np,
                                                             const float *__restrict__ dZ, int ldz, long m_begin, long M,
                                                             float *__restrict__ grad_b) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= Np) return;
    float s = 0.0f;
    for (int p = 0; p < nsplit; ++p) s += bias_partial[(long)p * Np + n];
    for (long m = m_begin; m < M; ++m) s += dZ[m * ldz + n];
    grad_b[n] = s;
}

// Deterministic column sums of a row-major [R, C] matrix (bias gradients, head-weight gradients): grid (ceil(C/64),
// slices); block = 16 waves; wave w of slice y adds rows y*rows_per_slice + w, + 16, ... in fp32, the 16 waves are
// combined in fp64 in a fixed order.  Called twice (slices, then 1) by the host.
constexpr int kColWaves = 16;
__global__ __launch_bounds__(kColWaves * kWave) void colsum_kernel(const float *__restrict__ in, long R, int C, int ld,
                                                                    long rows_per_slice, float *__restrict__ out /*[slices][C]*/) {
    __shared__ double red[kColWaves][kWave];
    const int wave = threadIdx.x >> 6, l = lane_id();
    const int c = blockIdx.x * kWave + l;
    const long r0 = (long)blockIdx.y * rows_per_slice;
    long r1 = r0 + rows_per_slice;
    if (r1 > R) r1 = R;
    float acc = 0.0f;
    if (c < C)
        for (long r = r0 + wave; r < r1; r += kColWaves) acc += in[r * ld + c];
    red[wave][l] = (double)acc;
    __syncthreads();
    if (wave == 0 && c < C) {
        double s = 0.0;
        for (int k = 0; k < kColWaves; ++k) s += red[k][l];
        out[(long)blockIdx.y * C + c] = (float)s;
    }
}

}  // namespace tn
}  // namespace m360
