// Second-generation fp32-MFMA linear kernel: persistent workgroups, one wave per SIMD, LDS-DMA.
//
//   * grid = one 256-thread workgroup (4 waves = one per SIMD, each owning the whole 512-entry
//     register file) per CU; each workgroup walks 256 x 256 output tiles with stride gridDim.x,
//     XCD-aware (the 32 workgroups of one XCD sweep 8 M-tiles x 4 N-tiles that share activation
//     rows in that XCD's L2).  Wave tile 128 x 128 = 4 x 4 MFMA tiles -> 256 accumulator registers.
//   * A/B K-step tiles (256 rows x 32 floats each) go global -> LDS directly with
//     buffer_load_dwordx4 ... lds (LDS-DMA: no staging VGPRs, no ds_write pass; tile base in a buffer descriptor,
//     one tile-independent 32-bit lane offset per instruction, K offset = scalar offset), double-buffered.  LDS rows are
//     unpadded 128 B; bank conflicts are removed by an XOR swizzle applied on the per-lane SOURCE
//     address (16-B chunk c of row r lands in slot c ^ ((r>>1)&7)) and mirrored on the
//     ds_read_b128 side.
//   * operand fragments are double-buffered in registers: the 8 ds_read_b128 of K-group g+1 are
//     issued before the 64 MFMAs of group g; the last group of a K-step runs AFTER the barrier and
//     after the first reads of the next K-step were issued, so barrier skew and read latency hide
//     behind matrix work.  The DMA of the next K-step is issued in between MFMA slices.
//   * the first K-step of the NEXT tile is already in flight while a tile's epilogue runs.
#pragma once
#include "m360_common.hip.h"

namespace m360 {
namespace persist {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: valid "v" asm operand (HIP float4 is a struct)

constexpr int BM = 256, BN = 256, BK = 32;
constexpr int kThreads = 256;
constexpr int TM = 4, TN = 4;
constexpr int kTileFloats = 256 * BK;          // one operand tile (A or B) of one K-step
constexpr int kBufFloats = 2 * kTileFloats;    // A + B
constexpr int kDma = 8;                        // DMA instructions per operand per wave per K-step
typedef __attribute__((address_space(3))) void *lds_ptr_t;

#define M360_INL __attribute__((always_inline))

template <int ACT>
__device__ __forceinline__ float act_fn(float v) {
    if (ACT == M360_ACT_RELU) return relu_nanf_(v);
    // hardware exp2 / rcp (v_exp_f32, v_rcp_f32: ~1 ulp each): |error| of the sigmoid <= ~2e-7 absolute,
    // against the 1e-4 render tolerance; the IEEE divide + full expf cost 8 % of a 1024x1024 layer
    if (ACT == M360_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    return v;
}

// diagnostic cycle stamps (STAMP builds only; never used by the product path): per workgroup, summed
// over all K-steps: [0] group0+DMA issue, [1] group1, [2] group2, [3] DMA-wait+barrier, [4] group3,
// [5] whole K-step, [6] K-steps, [7] epilogue
#ifdef M360_DIAG
__device__ unsigned long long g_stamps[256 * 16];
#define M360_STAMP_STORE(i, v) g_stamps[blockIdx.x * 16 + (i)] = (v)
#else
#define M360_STAMP_STORE(i, v) ((void)(v))
#endif

constexpr int kHeadMaxN = 1024;  // widest layer whose heads can be fused (bias + head rows live in LDS: (1 + HEADS) x 4 KiB)

// HEADS > 0 (the LAST hidden layer of a stage): the epilogue also multiplies the activated outputs with the HEADS head
// rows (model.py:52 / :150-158: hidden -> 1 density, hidden -> 1 + 3 density / colour) while they are in registers and
// writes per-row PARTIAL head sums head_part[M][2 * tiles_n][HEADS] (one slot per 128-column wave tile; the finisher
// adds the slots in a fixed order).  With STORE_Y = false the layer output itself never goes to HBM (rendering);
// the training tape keeps it (STORE_Y = true) - both produce the same partial sums bit for bit.
// EVENK: Kp / 32 is even -> K-step kt always works on LDS buffer kt & 1: the fragment addresses of both buffers are kept in
// registers and the 8 vector address updates per K-step disappear (vector instructions are not hidden behind MFMAs)
template <int ACT, bool STAMP = false, int HEADS = 0, bool STORE_Y = true, bool EVENK = false>
__global__ __launch_bounds__(kThreads, 1) void linear_f32_mfma_persist_kernel(
    const float *__restrict__ X, long M, int ldx, const float *__restrict__ W,
    const float *__restrict__ bias, int Np, int Kp, float *__restrict__ Y, int ldy, int tiles_n,
    int ntiles, const float *__restrict__ aux = nullptr, const float *__restrict__ head_w = nullptr,
    float *__restrict__ head_part = nullptr) {
    __shared__ __attribute__((aligned(1024))) float smem[2 * kBufFloats];  // 128 KiB
    __shared__ __attribute__((aligned(16))) float s_hw[HEADS > 0 ? (1 + HEADS) * kHeadMaxN : 4];  // bias, then the head rows

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: keeps the DMA's LDS base (M0) in SGPRs
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int G = gridDim.x;
    const int ksteps = Kp / BK;

    auto tile_coords = [&](int lin_id, long &m0, int &n0) M360_INL {
        // XCD-aware: ids that share id % 8 (one XCD under round-robin dispatch) cover a contiguous
        // range of tiles, N-tiles of one M-tile adjacent (speed only, never correctness)
        const int full = (ntiles / 8) * 8;
        int lin = lin_id;
        if (lin_id < full) lin = (lin_id % 8) * (full / 8) + lin_id / 8;
        m0 = (long)(lin / tiles_n) * BM;
        n0 = (lin % tiles_n) * BN;
    };

    // ---- LDS-DMA staging: wave w fills rows [64w, 64w+64) of A and of B, 8 instructions each
    // (8 rows x 128 B per instruction: lane L -> row L>>3, slot L&7 holding chunk slot ^ f(row)).
    // The kernel only ever sees FULL 256 x 256 tiles (M and Np multiples of 256): the host sends ragged
    // rows / columns to the first-generation kernel, which is bit-identical.
    const int st_r = wave * 64 + (lane >> 3);  // row of DMA instruction q = st_r + 8 q
    // per-lane byte offsets inside a tile (tile-independent); the tile base lives in two buffer descriptors (SGPRs) and
    // the K offset is the instruction's scalar offset: no vector address arithmetic per K-step or per tile
    unsigned a_voff[kDma], b_voff[kDma];
#pragma unroll
    for (int q = 0; q < kDma; ++q) {
        const int r = st_r + 8 * q;
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        a_voff[q] = (unsigned)(r * ldx + 4 * chunk) * 4u;
        b_voff[q] = (unsigned)(r * Kp + 4 * chunk) * 4u;
    }
    __amdgpu_buffer_rsrc_t rsrc_a, rsrc_b;
    auto set_load_tile = [&](long m0, int n0) M360_INL {  // full tiles only: no row clamping needed
        rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X + m0 * ldx), 0, 0x7fffffff, 0x00020000);
        rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(W + (long)n0 * Kp), 0, 0x7fffffff, 0x00020000);
    };
    float *const dma_dst = smem + st_r * 0 + wave * 64 * BK;  // + buf * kBufFloats + q * 8 * BK
    // which: 1 = A rows, 2 = B rows, 3 = both (row-block q of this wave's 64-row slice)
    auto issue_dma = [&](int buf, int k0, int q, int which) M360_INL {
        float *dstA = dma_dst + buf * kBufFloats + q * 8 * BK;
        if (which & 1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)dstA, 16, a_voff[q], 4 * k0, 0, 0);
        if (which & 2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(dstA + kTileFloats), 16, b_voff[q], 4 * k0, 0, 0);
    };

    // ---- operand reads: lane (l31, h), K-group g reads chunk (2g+h) of its rows = slot (2g+h)^f.
    // The reads are inline asm with hand-counted lgkmcnt waits: with LDS-DMA in flight hipcc would
    // otherwise put s_waitcnt vmcnt(0) / lgkmcnt(0) in front of every compiler-visible LDS read.
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)smem;
    const int fsw = (l31 >> 1) & 7;
    unsigned a_addr[4], b_addr[4];  // byte addresses in buffer 0
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int slot = ((2 * g + h) ^ fsw) * 4;
        a_addr[g] = lds0 + 4u * ((wm * 128 + l31) * BK + slot);
        b_addr[g] = lds0 + 4u * (kTileFloats + (wn * 128 + l31) * BK + slot);
    }
    unsigned a_hi[4], b_hi[4];  // the same in buffer 1 (EVENK only: 65536 does not fit the 16-bit offset of ds_read)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        a_hi[g] = a_addr[g] + 4u * kBufFloats;
        b_hi[g] = b_addr[g] + 4u * kBufFloats;
    }

    f32x16 acc[TM][TN];
    f32x4 fa_a[TM], fa_b[TN], fb_a[TM], fb_b[TN];

#define M360_DS128(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr))
#define M360_READ(FA, FB, bufoff, g)                                   \
    do {                                                               \
        const unsigned aa_ = a_addr[g] + (bufoff), bb_ = b_addr[g] + (bufoff); \
        M360_DS128(FA[0], aa_, 0);                                     \
        M360_DS128(FA[1], aa_, 4096);                                  \
        M360_DS128(FA[2], aa_, 8192);                                  \
        M360_DS128(FA[3], aa_, 12288);                                 \
        M360_DS128(FB[0], bb_, 0);                                     \
        M360_DS128(FB[1], bb_, 4096);                                  \
        M360_DS128(FB[2], bb_, 8192);                                  \
        M360_DS128(FB[3], bb_, 12288);                                 \
    } while (0)
// the wait takes the fragment it guards as in/out operands: every later use depends on the wait,
// so hipcc can neither hoist a use nor place a register copy of not-yet-landed data above it
#define M360_WAIT_FRAG(n, FA, FB)                                                                   \
    asm volatile("s_waitcnt lgkmcnt(" #n ")"                                                        \
                 : "+v"(FA[0]), "+v"(FA[1]), "+v"(FA[2]), "+v"(FA[3]), "+v"(FB[0]), "+v"(FB[1]),    \
                   "+v"(FB[2]), "+v"(FB[3])::"memory")
#define M360_SB() __builtin_amdgcn_sched_barrier(0)
// 8 MFMAs: rows i0, i0+1 of the wave tile x 4 columns, k-pair s of the current K-group
#define M360_MFMA8(FA, FB, s, i0)                                                                      \
    do {                                                                                               \
        _Pragma("unroll") for (int i = (i0); i < (i0) + 2; ++i) _Pragma("unroll") for (int j = 0; j < TN; ++j) \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[i][s], FB[j][s], acc[i][j], 0, 0, 0);  \
    } while (0)
// One K-group = 8 units of 8 MFMAs on fragment (FA, FB).  After each unit ONE ds_read_b128 of the NEXT
// K-group (into the other fragment NA/NB, byte addresses na/nb) and, when DMA is 1, the two LDS-DMA
// instruction(s) of row-block q = unit of the next K-step are issued (DMA: 1 = A half, 2 = B half,
// 3 = both): every non-matrix instruction gets
// its own >= 512-cycle MFMA shadow instead of being issued in bursts.
#define M360_UNIT(FA, FB, s, i0, RD, DMA, q)            \
    do {                                                \
        M360_MFMA8(FA, FB, s, i0);                      \
        RD;                                             \
        if (DMA) issue_dma(buf ^ 1, next_k0, q, DMA);   \
        M360_SB();                                      \
    } while (0)
#define M360_GROUP_PIPE(FA, FB, NA, NB, na, nb, DMA)                                 \
    do {                                                                             \
        M360_UNIT(FA, FB, 0, 0, M360_DS128(NA[0], na, 0), DMA, 0);                   \
        M360_UNIT(FA, FB, 0, 2, M360_DS128(NA[1], na, 4096), DMA, 1);                \
        M360_UNIT(FA, FB, 1, 0, M360_DS128(NA[2], na, 8192), DMA, 2);                \
        M360_UNIT(FA, FB, 1, 2, M360_DS128(NA[3], na, 12288), DMA, 3);               \
        M360_UNIT(FA, FB, 2, 0, M360_DS128(NB[0], nb, 0), DMA, 4);                   \
        M360_UNIT(FA, FB, 2, 2, M360_DS128(NB[1], nb, 4096), DMA, 5);                \
        M360_UNIT(FA, FB, 3, 0, M360_DS128(NB[2], nb, 8192), DMA, 6);                \
        M360_UNIT(FA, FB, 3, 2, M360_DS128(NB[3], nb, 12288), DMA, 7);               \
    } while (0)

#define M360_STAMP(var)                                                             \
    do {                                                                            \
        if (STAMP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); \
    } while (0)
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;

    unsigned long long mt0 = 0, mt1 = 0, rt0 = 0, rt1 = 0;  // whole tile loop: s_memtime (shader clock) / s_memrealtime (100 MHz)
    int lin_id = blockIdx.x;
    if (lin_id >= ntiles) return;
    if (HEADS > 0) {  // before any LDS-DMA is in flight; made visible by the __syncthreads() below
        for (int c = tid; c < Np; c += kThreads) {
            s_hw[c] = bias[c];
#pragma unroll
            for (int hh = 0; hh < HEADS; ++hh) s_hw[(1 + hh) * kHeadMaxN + c] = head_w[(long)hh * Np + c];
        }
    }
    long m0;
    int n0;
    tile_coords(lin_id, m0, n0);
    set_load_tile(m0, n0);
#pragma unroll
    for (int q = 0; q < kDma; ++q) issue_dma(0, 0, q, 3);
    int buf = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // first K-step of the first tile has landed
    __syncthreads();
    M360_SB();
    M360_READ(fa_a, fa_b, 0u, 0);  // loop invariant from here on: group 0 of the current step in flight
    M360_SB();

    if (STAMP) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(mt0), "=s"(rt0)::"memory");
    for (; lin_id < ntiles; lin_id += G) {
        tile_coords(lin_id, m0, n0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

        // the K-step loaded next: kt+1 of this tile, else step 0 of this workgroup's next tile
        // (else, harmlessly, step 0 of the current tile again: nobody reads it)
#define M360_KSETUP(KT)                                  \
    int next_k0 = ((KT) + 1) * BK;                       \
    if ((KT) + 1 == ksteps) {                            \
        next_k0 = 0;                                     \
        if (lin_id + G < ntiles) {                       \
            long nm0;                                    \
            int nn0;                                     \
            tile_coords(lin_id + G, nm0, nn0);           \
            set_load_tile(nm0, nn0);                     \
        }                                                \
    }
        // every group: wait for its own operands (the only LDS reads outstanding), then run its 64 MFMAs with the next
        // group's 8 reads (and, in group 0, the next K-step's 16 DMA instructions) in between; the barrier comes before
        // group 3, which hides its skew and whose interleaved reads are group 0 of the NEXT step / tile
#define M360_KSTEP(A1, B1, A2, B2, A3, B3, A0N, B0N)                                                                     \
    do {                                                                                                                 \
        M360_SB();                                                                                                       \
        M360_STAMP(c0);                                                                                                  \
        M360_WAIT_FRAG(0, fa_a, fa_b); /* R0 landed */                                                                   \
        M360_SB();                                                                                                       \
        M360_GROUP_PIPE(fa_a, fa_b, fb_a, fb_b, A1, B1, 3); /* group 0 (+ reads R1, + the 16 DMA instructions of step t+1) */ \
        M360_STAMP(c1);                                                                                                  \
        M360_WAIT_FRAG(0, fb_a, fb_b);                                                                                   \
        M360_SB();                                                                                                       \
        M360_GROUP_PIPE(fb_a, fb_b, fa_a, fa_b, A2, B2, 0);                                                              \
        M360_STAMP(c2);                                                                                                  \
        M360_WAIT_FRAG(0, fa_a, fa_b);                                                                                   \
        M360_SB();                                                                                                       \
        M360_GROUP_PIPE(fa_a, fa_b, fb_a, fb_b, A3, B3, 0);                                                              \
        M360_STAMP(c3);                                                                                                  \
        /* R3 landed => every read of `buf` by this wave is done; own DMA of the next step landed */                    \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier"                                                        \
                     : "+v"(fb_a[0]), "+v"(fb_a[1]), "+v"(fb_a[2]), "+v"(fb_a[3]), "+v"(fb_b[0]), "+v"(fb_b[1]),         \
                       "+v"(fb_b[2]), "+v"(fb_b[3])::"memory");                                                          \
        M360_SB();                                                                                                       \
        M360_STAMP(c4);                                                                                                  \
        M360_GROUP_PIPE(fb_a, fb_b, fa_a, fa_b, A0N, B0N, 0);                                                            \
        M360_STAMP(c5);                                                                                                  \
        if (STAMP) {                                                                                                     \
            st[0] += c1 - c0; st[1] += c2 - c1; st[2] += c3 - c2; st[3] += c4 - c3; st[4] += c5 - c4;                    \
            st[5] += c5 - c0; st[6] += 1;                                                                                \
        }                                                                                                                \
    } while (0)
        if (EVENK) {  // `buf` stays 0 across tiles; the inner `buf` names the K-step's buffer for the DMA target
            for (int kt = 0; kt < ksteps; kt += 2) {
                {
                    M360_KSETUP(kt);
                    const int buf = 0;
                    M360_KSTEP(a_addr[1], b_addr[1], a_addr[2], b_addr[2], a_addr[3], b_addr[3], a_hi[0], b_hi[0]);
                }
                {
                    M360_KSETUP(kt + 1);
                    const int buf = 1;
                    M360_KSTEP(a_hi[1], b_hi[1], a_hi[2], b_hi[2], a_hi[3], b_hi[3], a_addr[0], b_addr[0]);
                }
            }
        } else {
            for (int kt = 0; kt < ksteps; ++kt) {
                M360_KSETUP(kt);
                const unsigned boff = buf ? 4u * kBufFloats : 0u;
                const unsigned noff = buf ? 0u : 4u * kBufFloats;  // the other buffer (next K-step)
                const unsigned a1 = a_addr[1] + boff, b1 = b_addr[1] + boff, a2 = a_addr[2] + boff, b2 = b_addr[2] + boff;
                const unsigned a3 = a_addr[3] + boff, b3 = b_addr[3] + boff, a0n = a_addr[0] + noff, b0n = b_addr[0] + noff;
                M360_KSTEP(a1, b1, a2, b2, a3, b3, a0n, b0n);
                buf ^= 1;
            }
        }
#undef M360_KSTEP
#undef M360_KSETUP
        M360_STAMP(c0);

        // ---- epilogue: bias + activation.  The accumulator layout (lane = column l31, 16 registers =
        // rows (r&3)+8(r>>2)+4h) would store 4 B per lane; interior tiles are instead transposed
        // through the idle LDS buffer (wave-private 32 x 36 floats) so that every lane stores 16 B and one
        // instruction writes 8 full 128-B row segments: 4x fewer, 4x wider global stores.
        {
            int ldy_t = ldy;
            asm volatile("" : "+s"(ldy_t));  // keep the address math inside the tile loop (LICM would spill it)
            float *__restrict__ Yt = Y + m0 * ldy_t + n0;
            // staging lives in the idle buffer (`buf` already holds the next tile), INSIDE the 8 KiB slice that only
            // this wave's own DMA instructions write: a faster wave that already streams the next K-step into the
            // idle buffer can therefore never overwrite another wave's staging rows, and program order protects ours
            float *stg = dma_dst + (buf ^ 1) * kBufFloats;
            const int rrow = lane >> 3, rcol = 4 * (lane & 7);
            if constexpr (HEADS > 0) {
                // row-major walk (all 4 column blocks of a 32-row block before the next rows): a lane carries the head
                // sums of its 4 rows (p) across the 128 columns of the wave tile, then the 8 lanes of a row are reduced
                const int slots = 2 * tiles_n, slot = 2 * (n0 / BN) + wn;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    float hacc[4][HEADS];
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                        for (int hh = 0; hh < HEADS; ++hh) hacc[p][hh] = 0.0f;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int col = n0 + wn * 128 + j * 32 + rcol;
                        const float4 b4 = *reinterpret_cast<const float4 *>(s_hw + col);
                        float4 hw4[HEADS];
#pragma unroll
                        for (int hh = 0; hh < HEADS; ++hh) hw4[hh] = *reinterpret_cast<const float4 *>(s_hw + (1 + hh) * kHeadMaxN + col);
#pragma unroll
                        for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * h) * 36 + l31] = acc[i][j][r];
                        const long yoff = (long)(wm * 128 + i * 32 + rrow) * ldy_t + wn * 128 + j * 32 + rcol;
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            float4 v = *reinterpret_cast<const float4 *>(stg + (p * 8 + rrow) * 36 + rcol);
                            v.x = act_fn<ACT>(v.x + b4.x);
                            v.y = act_fn<ACT>(v.y + b4.y);
                            v.z = act_fn<ACT>(v.z + b4.z);
                            v.w = act_fn<ACT>(v.w + b4.w);
                            if (STORE_Y) *reinterpret_cast<float4 *>(Yt + yoff + (long)(p * 8) * ldy_t) = v;
#pragma unroll
                            for (int hh = 0; hh < HEADS; ++hh)
                                hacc[p][hh] = fmaf(v.w, hw4[hh].w, fmaf(v.z, hw4[hh].z, fmaf(v.y, hw4[hh].y, fmaf(v.x, hw4[hh].x, hacc[p][hh]))));
                        }
                        M360_SB();
                    }
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
#pragma unroll
                        for (int hh = 0; hh < HEADS; ++hh) hacc[p][hh] = row8_sum(hacc[p][hh]);
                        if ((lane & 7) == 0) {
                            float *dst = head_part + ((m0 + wm * 128 + i * 32 + p * 8 + rrow) * slots + slot) * HEADS;
#pragma unroll
                            for (int hh = 0; hh < HEADS; ++hh) dst[hh] = hacc[p][hh];
                        }
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float4 b4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    if (ACT != M360_ACT_RELU_MASK) b4 = *reinterpret_cast<const float4 *>(bias + n0 + wn * 128 + j * 32 + rcol);
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * h) * 36 + l31] = acc[i][j][r];
                        const long yoff = (long)(wm * 128 + i * 32 + rrow) * ldy_t + wn * 128 + j * 32 + rcol;
                        float *__restrict__ Yc = Yt + yoff;
#pragma unroll
                        for (int p = 0; p < 4; ++p) {
                            float4 v = *reinterpret_cast<const float4 *>(stg + (p * 8 + rrow) * 36 + rcol);
                            if (ACT == M360_ACT_RELU_MASK) {  // backward of ReLU: keep where the forward output (aux, same ld) was > 0
                                const float4 a4 = *reinterpret_cast<const float4 *>(aux + m0 * ldy_t + n0 + yoff + (long)(p * 8) * ldy_t);
                                v.x = a4.x > 0.0f ? v.x : 0.0f;
                                v.y = a4.y > 0.0f ? v.y : 0.0f;
                                v.z = a4.z > 0.0f ? v.z : 0.0f;
                                v.w = a4.w > 0.0f ? v.w : 0.0f;
                            } else {
                                v.x = act_fn<ACT>(v.x + b4.x);
                                v.y = act_fn<ACT>(v.y + b4.y);
                                v.z = act_fn<ACT>(v.z + b4.z);
                                v.w = act_fn<ACT>(v.w + b4.w);
                            }
                            *reinterpret_cast<float4 *>(Yc + (long)(p * 8) * ldy_t) = v;
                        }
                        M360_SB();
                    }
                }
            }
        }
        M360_STAMP(c1);
        if (STAMP) st[7] += c1 - c0;
    }
    if (STAMP) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(mt1), "=s"(rt1)::"memory");
    if (STAMP && threadIdx.x == 0 && blockIdx.x < 256) {
#pragma unroll
        for (int i = 0; i < 8; ++i) M360_STAMP_STORE(i, st[i]);
        M360_STAMP_STORE(8, mt1 - mt0);  // in-kernel clock = [8] / [9] x 100 MHz (MI355X_MICROARCH.md, DVFS item 6)
        M360_STAMP_STORE(9, rt1 - rt0);
    }
#undef M360_STAMP
#undef M360_READ
#undef M360_DS128
#undef M360_WAIT_FRAG
#undef M360_SB
}

}  // namespace persist
}  // namespace m360
