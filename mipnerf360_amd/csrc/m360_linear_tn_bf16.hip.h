// Weight-gradient GEMM of the bf16 training path (SURVEY.md §8 row f3 in the reduced-precision mode, round 5):
//     dW[Np, Kp] = dZ[M, Np]^T * X[M, Kp]      bf16 operands, fp32 accumulation on v_mfma_f32_16x16x32_bf16, fp32 result
// - what autograd computes for nn.Linear's weight (model.py:43-53,131-158 under train.py:62,80) when activations and their gradients are
// carried in bf16 (master weights, the gradient itself and AdamW stay fp32).
//
// Same decomposition as the fp32 kernel (m360_linear_tn.hip.h): the contraction runs over the M = rays x samples rows while the output is
// at most 1024 x 1024 = 16 tiles of 256 x 256, so the rows are split - workgroup (tile, split) reduces its slice of rows into a private
// partial tile, a second kernel adds the partials in a FIXED order (deterministic, no atomics).
//   * BOTH operands are "row = contraction index" in memory and the MFMA wants 8 consecutive contraction indices per lane: a column access.
//     gfx950's ds_read_b64_tr_b16 does that transposition on the way out of the LDS (a 16-lane group reads 4 rows x 16 columns and every
//     lane receives one column): two of them (rows 8g .. 8g+3 and 8g+4 .. 8g+7 of a 32-row k-step) ARE one operand fragment.  The tiles go
//     into the LDS as they lie in memory, by LDS-DMA (no staging registers, no transposing pass).
//   * a stage is 64 rows of dZ and of X (256 columns = 512 bytes each), 2 x 32 KiB, double-buffered; an LDS-DMA piece is two whole rows.
//     LDS image: row r at r x 512 bytes, its sixteen 32-byte units XOR-swizzled on the source side, unit u in slot u ^ f(r) with
//     f(r) = (r & 3) | ((r >> 3) & 1) << 2: the 8 row segments a 32-lane half of a transposed read touches (rows R .. R+3 and R+8 .. R+11, one
//     16-column block) then fall into 8 different bank octets - conflict-free (MI355X_MICROARCH.md §LDS: banks count per 32-lane half).
//   * 512 threads = 8 waves, wave tile 128 (n) x 64 (k) = 8 x 4 blocks, 128 accumulator registers; operands swapped (A := X columns,
//     B := dZ columns) so that a lane's 4 accumulator registers are 4 consecutive k of one n: 16-byte stores of the partial tile.
//   * bias gradient db[n] = sum_m dZ[m, n] on the matrix pipe as well: one MFMA per dZ block and k-step against a fragment of ones (exact:
//     bf16 values added in fp32), the 16 blocks of a tile row spread over its workgroups and waves (below).
//   * ids are XCD-aware: the 16 tiles of one split (same rows) run on one XCD, so a row of dZ / X comes from HBM once per XCD and from its L2
//     for the other tiles of the split.
#pragma once
#include "m360_common.hip.h"

namespace m360 {
namespace tn16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef __attribute__((address_space(3))) s16x4 *lds_s16x4_p;

constexpr int BT = 256;                        // output tile edge (Np and Kp direction)
constexpr int BKM = 64;                        // rows (contraction) per stage
constexpr int kThreads = 512;
constexpr int kRowBytes = BT * 2;              // 512
constexpr int kOperandBytes = BKM * kRowBytes; // 32 KiB
constexpr int kStageBytes = 2 * kOperandBytes; // dZ tile | X tile
constexpr int kMaxWorkgroups = 256;            // tiles x splits target (one per CU)

__device__ __forceinline__ int swz(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

__global__ __launch_bounds__(kThreads, 1) void linear_tn_bf16_kernel(
    const __bf16 *__restrict__ dZ, int ldz, const __bf16 *__restrict__ X, int ldx, int Np, int Kp,
    float *__restrict__ partial /*[nsplit][Np][Kp]*/, int tiles_k, int ntiles, int nsplit, long total_steps,
    long steps_per_split, float *__restrict__ bias_partial /*[nsplit][Np] or nullptr*/,
    int abl = 0 /* diagnostics (M360_TN16_ABL, wrong results): 1 = no MFMAs, 2 = no fragment reads, 4 = no LDS-DMA, 8 = no barrier */) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * kStageBytes];  // 128 KiB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 2, wk = wave & 3;
    const int g = lane >> 4, l15 = lane & 15, q = l15 >> 2, p = l15 & 3;

    // all tiles of a split on ONE XCD (workgroup b runs on XCD b % 8: speed only)
    const int total = ntiles * nsplit;
    int split, tile;
    if (total % 8 == 0 && (total / 8) % ntiles == 0) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile = j % ntiles;
        split = x * ((total / 8) / ntiles) + j / ntiles;
    } else {
        split = blockIdx.x / ntiles;
        tile = blockIdx.x % ntiles;
    }
    const int n0 = (tile / tiles_k) * BT, k0 = (tile % tiles_k) * BT;
    const long s_begin = (long)split * steps_per_split;
    long s_end = s_begin + steps_per_split;
    if (s_end > total_steps) s_end = total_steps;
    // bias gradient: the column sums of the dZ tile's 2 x 8 blocks of 16 columns are spread over the tile row's workgroups and waves - slot
    // 4 kt + wk of k tile kt, wave (wn, wk) adds up `bias_n` blocks of its n half from `bias_first` on (one block per slot with 4 k tiles: ONE
    // extra MFMA and fragment read per k-step in the waves of the first two k tiles; two blocks per wave when the layer has a single k tile).
    // When all of it sat in the first k tile's wk = 0 waves - 8 extra MFMAs per 32 - those workgroups finished 12 % after the rest
    // (1.31 -> 1.55 ms per 1024^2 layer).
    const int nslots = 4 * tiles_k < 8 ? 4 * tiles_k : 8, per = 8 / nslots, slot = 4 * (tile % tiles_k) + wk;
    const int bias_n = (bias_partial != nullptr && slot < nslots) ? per : 0, bias_first = slot * per;  // wave-uniform
    f32x4 acc[8][4], bacc[2];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 2; ++i) bacc[i] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    if (s_begin < s_end) {
        // ---- LDS-DMA: wave w stages pieces 4w .. 4w+3 (rows 8w .. 8w+7) of the dZ tile and of the X tile; lane L of a piece lands at byte
        // 16 L of the piece = row 2 pi + (L >> 5), 16-byte slot L & 31, and fetches the chunk whose 32-byte unit the swizzle puts there
        unsigned va[4], vb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 2 * (4 * wave + i) + (lane >> 5), s = lane & 31;
            const int c = 2 * ((s >> 1) ^ swz(r)) + (s & 1);  // 16-byte chunk of the row (0 .. 31)
            va[i] = (unsigned)(r * ldz + 8 * c) * 2u;
            vb[i] = (unsigned)(r * ldx + 8 * c) * 2u;
        }
        __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(dZ + s_begin * BKM * ldz + n0), 0, 0x7fffffff, 0x00020000);
        __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(X + s_begin * BKM * ldx + k0), 0, 0x7fffffff, 0x00020000);
        char *const dma_dst = smem + wave * 4 * 1024;
        auto issue = [&](int buf, unsigned soff_a, unsigned soff_b) __attribute__((always_inline)) {
            char *dst = dma_dst + buf * kStageBytes;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(dst + i * 1024), 16, va[i], soff_a, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(dst + kOperandBytes + i * 1024), 16, vb[i], soff_b, 0, 0);
            }
        };

        // ---- transposed fragment reads: lane (group g, q, p) supplies row 8g + q (+ 4 for the second half, + 32 per k-step) of column block cb,
        // bytes 8p of its 32-byte unit; the unit's slot is (cb & 7) ^ f with f = q | (g & 1) << 2 (= swz of every row this lane touches)
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
        const int f = q | ((g & 1) << 2);
        const unsigned lane_off = (unsigned)((8 * g + q) * kRowBytes + 8 * p);
        unsigned a_off[8], b_off[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) a_off[i] = lds0 + lane_off + (unsigned)((i ^ f) + 8 * wn) * 32u;                                   // dZ block 8 wn + i
#pragma unroll
        for (int j = 0; j < 4; ++j) b_off[j] = lds0 + kOperandBytes + lane_off + (unsigned)(((4 * (wk & 1) + j) ^ f) + 8 * (wk >> 1)) * 32u;  // X block 4 wk + j
        unsigned bias_off[2];  // the blocks this wave sums for the bias gradient (read once more: no dynamic choice among the fragments in registers)
#pragma unroll
        for (int e = 0; e < 2; ++e) bias_off[e] = lds0 + lane_off + (unsigned)((((bias_first + e) & 7) ^ f) + 8 * wn) * 32u;
        // one piece pair (dZ + X piece i) of the next stage: issued BETWEEN the MFMAs of the stage's first k-step (an LDS-DMA issue - M0 write +
        // buffer_load .. lds - costs a wave 100-185 cycles when it stands alone and nothing in an MFMA gap; the eight of a stage in one burst
        // behind the barrier: 1.28 instead of 1.15 ms per 1024^2 layer) and still most of a stage ahead of their use
        auto issue_pair = [&](int buf, int i, unsigned soff_a, unsigned soff_b) __attribute__((always_inline)) {
            char *dst = dma_dst + buf * kStageBytes;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(dst + i * 1024), 16, va[i], soff_a, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(dst + kOperandBytes + i * 1024), 16, vb[i], soff_b, 0, 0);
        };
        auto frag = [&](unsigned addr) __attribute__((always_inline)) -> bf16x8 {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(uintptr_t)addr);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(uintptr_t)(addr + 4 * kRowBytes));
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        const bf16x8 ones = {(__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f};

        // ONE barrier per stage: behind it every wave has its pieces of stage s in the LDS and has finished reading the other buffer (stage
        // s - 1), so the pieces of stage s + 1 may go out (issue_pair, in the first k-step's MFMA gaps) and have most of the stage to land.
        // What the 1.15-1.19 ms of a 1024^2 layer are made of (M360_TN16_ABL ablations of the diagnostics build, tools/diag/wgrad_ablations.sh,
        // profiles/r05/wgrad_bf16_ablation_8wave_form.txt; times include the 0.02 ms reduce kernel): LDS-DMA + barriers alone 0.84 ms (8.6 GB of
        // operand tiles: the L2s deliver ~10.5 TB/s into the LDSs), MFMAs + barriers alone 0.72 (without the barrier 0.69 = 1.59 PF: the matrix
        // pipe's own ceiling at the clock the chip holds), fragment reads + barriers alone 0.39; SQ_LDS_BANK_CONFLICT = 0; MFMA pipe busy 37 -> 41 %.
        // The three overlap only partly in an 8-wave, barrier-per-stage loop scheduled by the compiler; the remedy is the ring kernel's
        // structure (one wave per SIMD, 128 x 128 wave tiles: m360_linear_tn_bf16_w.hip.h, round 5 - 1.04-1.12 ms; a generated schedule: not built).
        // Also measured: the transposed reads as inline assembly with counted lgkmcnt waits (the compiler puts s_waitcnt vmcnt(0) in front of
        // LDS reads that follow an LDS-DMA issue): 1.17 ms, no gain - kept as builtins.
        // (Measured and rejected, profiles/r05/bf16_wgrad_pipelined_halves_SLOWER.jsonl: the loop over 32-row halves in four LDS quarters with
        // the fragments of half i + 1 read into a second register set "under" the MFMAs of half i and one barrier per half - 1.41 instead of
        // 1.27 ms per 1024^2 layer: lgkmcnt counts at most 15 LDS operations, so the 24 reads in front of the MFMAs are waited for anyway,
        // and the barrier count doubles.)
        issue(0, 0u, 0u);
        int buf = 0;
        for (long s = s_begin; s < s_end; ++s) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(abl & 8)) __syncthreads();
            const bool more = s + 1 < s_end && !(abl & 4);  // wave-uniform
            const unsigned nrows = (unsigned)(s + 1 - s_begin) * BKM;
            const unsigned soff_a = nrows * (unsigned)ldz * 2u, soff_b = nrows * (unsigned)ldx * 2u;
            const unsigned boff = buf ? (unsigned)kStageBytes : 0u;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 fa[8], fb[4];
                bf16x8 fbias[2] = {ones, ones};  // read WITH the other fragments: a read right in front of its MFMA exposes the LDS latency twice per stage
                if (bias_n > 0) fbias[0] = frag(bias_off[0] + boff + ks * 32 * kRowBytes);
                if (bias_n > 1) fbias[1] = frag(bias_off[1] + boff + ks * 32 * kRowBytes);
                if (!(abl & 2)) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[j] = frag(b_off[j] + boff + ks * 32 * kRowBytes);
#pragma unroll
                    for (int i = 0; i < 8; ++i) fa[i] = frag(a_off[i] + boff + ks * 32 * kRowBytes);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[j] = ones;
#pragma unroll
                    for (int i = 0; i < 8; ++i) fa[i] = ones;
                }
                if (abl & 1) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(fa[i]));
#pragma unroll
                    for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(fb[j]));
                    if (more && ks == 0) issue(buf ^ 1, soff_a, soff_b);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                        if (ks == 0 && (i & 1)) {  // the next stage's four piece pairs, each behind 8 MFMAs of the first k-step
                            __builtin_amdgcn_sched_barrier(0);
                            if (more) issue_pair(buf ^ 1, i >> 1, soff_a, soff_b);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
                if (bias_n > 0) {
                    bacc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fbias[0], bacc[0], 0, 0, 0);
                    if (bias_n > 1) bacc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fbias[1], bacc[1], 0, 0, 0);
                }
                // one k-step's 12 fragments at a time: with both k-steps' reads hoisted to the top of the stage the kernel needs 300+ registers
                // (256 per wave with two waves per SIMD) and spilled 62 of them into the loop
                __builtin_amdgcn_sched_barrier(0);
            }
            buf ^= 1;
        }
    }

    // ---- epilogue: acc[i][j][r] of lane (l15, g) = dW[n0 + 128 wn + 16 i + l15][k0 + 64 wk + 16 j + 4 g + r]
    float *__restrict__ P = partial + (long)split * Np * Kp;
    const int nrow = n0 + wn * 128 + l15, kcol = k0 + wk * 64 + 4 * g;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4 *>(P + (long)(nrow + 16 * i) * Kp + kcol + 16 * j) = acc[i][j];
    if (bias_n > 0 && g == 0) {  // every row of the ones product holds the column sums: take row 0 (lanes 0 .. 15, register 0)
        bias_partial[(long)split * Np + nrow + 16 * bias_first] = bacc[0][0];
        if (bias_n > 1) bias_partial[(long)split * Np + nrow + 16 * (bias_first + 1)] = bacc[1][0];
    }
}

// grad_w[n][k] = sum_s partial[s][n][k] (s ascending) + the < 64 tail rows the stages did not cover
__global__ __launch_bounds__(256) void tn16_reduce_kernel(const float *__restrict__ partial, int nsplit, int Np, int Kp,
                                                          const __bf16 *__restrict__ dZ, int ldz, const __bf16 *__restrict__ X, int ldx,
                                                          long m_begin, long M, float *__restrict__ grad_w) {
    const long idx4 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long count = (long)Np * Kp;
    if (idx4 * 4 >= count) return;
    const int n = (int)((idx4 * 4) / Kp), k = (int)((idx4 * 4) % Kp);
    float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    // 16 loads in flight, then added in ascending order (a thread that loads and adds one split at a time waits out 16 memory latencies:
    // 38-77 us per 1024 x 1024 gradient)
    for (int s0 = 0; s0 < nsplit; s0 += 16) {
        float4 v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e)
            v[e] = s0 + e < nsplit ? *reinterpret_cast<const float4 *>(partial + (long)(s0 + e) * count + idx4 * 4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (s0 + e < nsplit) {
                s.x += v[e].x;
                s.y += v[e].y;
                s.z += v[e].z;
                s.w += v[e].w;
            }
        }
    }
    for (long m = m_begin; m < M; ++m) {
        const float gz = (float)dZ[m * ldz + n];
        const bf16x4 x = *reinterpret_cast<const bf16x4 *>(X + m * ldx + k);
        s.x += gz * (float)x[0];
        s.y += gz * (float)x[1];
        s.z += gz * (float)x[2];
        s.w += gz * (float)x[3];
    }
    *reinterpret_cast<float4 *>(grad_w + idx4 * 4) = s;
}

__global__ __launch_bounds__(256) void tn16_bias_reduce_kernel(const float *__restrict__ bias_partial, int nsplit, int Np,
                                                               const __bf16 *__restrict__ dZ, int ldz, long m_begin, long M,
                                                               float *__restrict__ grad_b) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= Np) return;
    float s = 0.0f;
    for (int s0 = 0; s0 < nsplit; s0 += 16) {  // 16 loads in flight, added in ascending order
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = s0 + e < nsplit ? bias_partial[(long)(s0 + e) * Np + n] : 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (s0 + e < nsplit) s += v[e];
    }
    for (long m = m_begin; m < M; ++m) s += (float)dZ[m * ldz + n];
    grad_b[n] = s;
}

// First level of a two-level column sum of part[rows][Np] (the throttled mask kernel leaves 1024 - 2048 rows; one thread per column walking all of
// them is a 70 us chain of dependent load rounds at the end of the second stream): workgroup (x, y) adds rows [y per, (y + 1) per) of its 256
// columns in ascending order into out[y][Np]; tn16_bias_reduce_kernel then adds the gridDim.y rows of `out`.  No LDS (it runs beside a kernel
// that owns all of it), fixed order: deterministic.
__global__ __launch_bounds__(256) void tn16_bias_reduce_rows_kernel(const float *__restrict__ part, int rows, int per, int Np, float *__restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= Np) return;
    const int r0 = blockIdx.y * per, r1 = r0 + per < rows ? r0 + per : rows;
    float s = 0.0f;
    for (int r = r0; r < r1; r += 16) {
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = r + e < r1 ? part[(long)(r + e) * Np + n] : 0.0f;
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (r + e < r1) s += v[e];
    }
    out[(long)blockIdx.y * Np + n] = s;
}

// shapes the MFMA kernel does not take (reduced-width test models: pads of 64 / 128): the operands widened to fp32 for the fp32 kernel
__global__ __launch_bounds__(256) void widen_bf16_kernel(const __bf16 *__restrict__ in, long rows, int cols, int ld, float *__restrict__ out) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const long r = idx / cols;
    const int c = (int)(idx % cols);
    out[idx] = (float)in[r * ld + c];
}

// dx = relu_out > 0 ? dx : 0 (ReLU' of the layer below, from its stored OUTPUT: model.py:44-49,132-145), 8 values per thread
__global__ __launch_bounds__(256) void relu_mask_bf16_kernel(__bf16 *__restrict__ dx, const __bf16 *__restrict__ relu_out, long M, int cols, int ld) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8 = cols / 8;
    if (idx >= M * c8) return;
    const long r = idx / c8;
    const int c = (int)(idx % c8) * 8;
    s16x8 v = *reinterpret_cast<const s16x8 *>(dx + r * ld + c);
    const s16x8 a = *reinterpret_cast<const s16x8 *>(relu_out + r * ld + c);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a[i] > 0 ? v[i] : (short)0;  // a is a ReLU output: >= +0 or +NaN; bits > 0 <=> value > 0 (or NaN, kept like autograd's 1 * g)
    *reinterpret_cast<s16x8 *>(dx + r * ld + c) = v;
}

// The same mask by a FIXED number of workgroups striding over the rows (U pieces per thread in flight): the form m360_capi.hip runs on a second
// stream beside the layer's weight gradient - throttled so that it takes about as long as that kernel instead of saturating the HBM for half of it.
// SUMS: the masked rows ARE the next layer's dz, whose column sums are that layer's bias gradient - a thread always meets the same 8 columns
// (the stride is a multiple of cols / 8: the host checks), so it adds them up in registers on the way and writes them to part[workgroups x 256 /
// (cols / 8)][cols]; tn16_bias_reduce_kernel adds those rows in ascending order: deterministic.
template <int U, bool SUMS>
__global__ __launch_bounds__(256) void relu_mask_bf16_stride_kernel(__bf16 *__restrict__ dx, const __bf16 *__restrict__ relu_out, long M, int cols, int ld,
                                                                    float *__restrict__ part) {
    const int c8 = cols / 8;
    const long total = M * c8, stride = (long)gridDim.x * blockDim.x;
    float acc[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += U * stride) {
        long o[U];
        s16x8 v[U], a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long i = idx + u * stride < total ? idx + u * stride : idx;
            o[u] = (i / c8) * ld + (i % c8) * 8;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const s16x8 *>(dx + o[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) a[u] = *reinterpret_cast<const s16x8 *>(relu_out + o[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[u][i] = a[u][i] > 0 ? v[u][i] : (short)0;
            if (u == 0 || idx + u * stride < total) {
                *reinterpret_cast<s16x8 *>(dx + o[u]) = v[u];
                if (SUMS) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] += __builtin_bit_cast(float, (unsigned)(unsigned short)v[u][i] << 16);  // bf16 -> fp32: exact
                }
            }
        }
    }
    if (SUMS) {  // (256 % c8 == 0 and c8 <= 256: thread t's column group is t % c8 in every workgroup.)  No LDS: the kernel runs beside one that
        // owns all 160 KB of every CU - with a shared array its workgroups wait for that kernel to END (measured: +1.9 ms per layer).  Every
        // thread writes its own sums: part[(workgroup * 256 / c8 + t / c8)][cols]
        float *row = part + ((long)blockIdx.x * (256 / c8) + threadIdx.x / c8) * cols + 8 * (threadIdx.x % c8);
        *reinterpret_cast<float4 *>(row) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4 *>(row + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
}

__global__ void pack_linear_bf16_t_kernel(const float *__restrict__ w, int n_out, int k_in, int n_pad, int k_pad, __bf16 *__restrict__ wt) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n_pad * k_pad) return;
    const int k = (int)(idx / n_pad), n = (int)(idx % n_pad);
    wt[idx] = (__bf16)((n < n_out && k < k_in) ? w[(long)n * k_in + k] : 0.0f);
}

}  // namespace tn16
}  // namespace m360
