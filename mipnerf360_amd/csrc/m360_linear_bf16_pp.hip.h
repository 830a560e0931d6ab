// Second bf16 linear kernel (opt-in reduced-precision MLP, BASELINE configs[4]): 8 waves = two per SIMD in "ping-pong".
//
// Why a second structure: with ONE wave per SIMD (m360_linear_bf16.hip.h) every LDS-DMA issue (~100-185 cycles while the
// phase also carries ds_reads) is exposed against only 2048 MFMA cycles per K-step, and the kernel stops at ~36 % of the
// bf16 peak.  Here the 256 x 256 tile is shared by 8 waves (2 along M x 4 along N, wave tile 128 x 64 = 8 x 4 blocks of
// v_mfma_f32_16x16x32_bf16 -> 128 accumulator registers).  A K-step (64 bf16) is 4 phases; every phase is
//       [ds_read the operand sub-tile | stage one 16 KiB unit of the NEXT K-step by LDS-DMA | s_waitcnt vmcnt(4)]
//       s_barrier   [s_waitcnt lgkmcnt(0) | 16 MFMAs = one 64 x 32 quadrant over the whole K-step]   s_barrier
// and the waves of the second M half run ONE barrier behind the first, so on every SIMD one wave computes while its
// partner loads.  vmcnt is counted (never 0): a unit (= what one phase reads) is staged >= 3 phases before its first read, retired by the
// wait one phase before that read, and re-staged >= 4 phases after its last read.
// Operands are swapped (A := weight rows, B := activation rows), so a lane's 4 accumulator registers are 4 consecutive
// output columns of one row; in addition MFMA row i = 4a + b of N-block jb is weight row 16a + 4jb + b of the wave's 64,
// so the four N-blocks of a lane are 16 CONSECUTIVE columns: bias + activation + bf16 pack + two 16-byte stores per
// row, whole 128-byte lines per store instruction, no LDS transposition in the epilogue.
// Persistent: workgroups walk tiles with stride gridDim.x; the last K-step of a tile stages the first K-step of the next.
// LDS rows are 128 B with a source-side XOR swizzle (chunk c of row r in slot c ^ f(r)); f = (r>>1)&7 for activation
// rows and f = 2((r>>4)&3) + ((r>>1)&1) for weight rows (their 16-lane read groups touch rows 16a + b): either way a
// 16-lane ds_read_b128 group covers 16 distinct 16-byte slots.
#pragma once
#include "m360_common.hip.h"
#include "m360_linear_persist.hip.h"  // diagnostic stamp buffer

namespace m360 {
namespace pp16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int kThreads = 512;
constexpr int kHalfBytes = 128 * 128;       // half-tile: 128 rows x 128 B
constexpr int kTileBytes = 4 * kHalfBytes;  // A0 A1 B0 B1 of one K-step
constexpr int kMaxBias = 4096;              // widest layer this kernel takes (bias is served from LDS)

template <int ACT>
__device__ __forceinline__ float act_fn(float v) {
    if (ACT == M360_ACT_RELU) return relu_nanf_(v);
    if (ACT == M360_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    return v;
}

// X3 ("bf16x3", near-fp32 accuracy on the bf16 pipe): activations are stored as TWO bf16 terms per value, rows [hi(K) | lo(K)],
// the weights as W3 = [Wh | Wh | Wl] (rows of 3K); the kernel forms xh wh + xl wh + xh wl in one fp32 accumulator and writes its
// output as [hi(Np) | lo(Np)].
//   X3 = 1: ONE contraction over Kp = 3K, the activation column wrapping back to 0 at 2K (6 operand tiles staged per 64-deep block);
//   X3 = 2: per 64-deep block the three products run as  xl wh -> xh wh -> xh wl,  so consecutive K-steps SHARE an operand tile:
//           the second keeps Wh, the third keeps Xh - 4 operand tiles staged per block instead of 6 (a third less L2 -> LDS
//           traffic and LDS-DMA issue, the two things that bound this kernel: DESIGN.md 4.3).  The A and B halves of the two LDS
//           buffers are switched independently: T1 reads (A0, B0) and stages Xh into A1; T2 reads (A1, B0) and stages Wl into
//           B1; T3 reads (A1, B1) and stages the next block's Xl / Wh into A0 / B0.
// HEADS > 0 (the last hidden layer of a stage, round 3): the output heads' dot products are formed ON THE MATRIX PIPE from the packed
// bf16 rows the epilogue holds anyway - a lane's 8 consecutive output columns of one row ARE a B fragment of
// v_mfma_f32_16x16x32_bf16, the head rows (as hi / lo bf16 terms, from LDS) the A fragment, so one MFMA per row block and term
// gives D[head][row]: lanes 0-15 then hold the 4 heads of their row.  No register growth (the fragment registers are free while
// a quadrant is stored), 8-12 MFMAs per quadrant against 1024 per tile.  One partial sum per row, wave column group and 8-column
// half goes to head_part[row][(Np / 256) * 8][HEADS]; with STORE_Y = false the layer's own output is never written.
constexpr int kHeadMaxN = 1024;  // widest layer the fused heads take (their hi / lo rows live in 16 KiB of LDS)
template <int ACT, bool STAMP = false, int X3 = 0, int HEADS = 0, bool STORE_Y = true>
__global__ __launch_bounds__(kThreads, 1) void linear_bf16_pp_kernel(
    const __bf16 *__restrict__ X, long M, int ldx, const __bf16 *__restrict__ W, const float *__restrict__ bias,
    int Np, int Kp, __bf16 *__restrict__ Y, int ldy, int tiles_n, int ntiles, const float *__restrict__ head_w = nullptr,
    float *__restrict__ head_part = nullptr) {
    // 128 KiB of stages + the bias vector (+ with HEADS the head rows as bf16 hi / lo terms: 2 x HEADS x Np <= 16 KiB)
    __shared__ __attribute__((aligned(1024))) char smem[2 * kTileBytes + kMaxBias * 4 + (HEADS ? 2 * 4 * kHeadMaxN * 2 : 0)];
    static_assert(HEADS == 0 || HEADS == 1 || HEADS == 4, "1 (proposal) or 4 (NeRF) heads");
#if defined(PP_PHASES) && PP_PHASES != 4
    static_assert(!X3, "the split (bf16x3) epilogue is counted for the 4-phase K-step only");
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int ksteps = Kp / BK;
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, rt1 = 0, rt2 = 0;  // rt*: s_memrealtime (100 MHz) around the main loop
#define PP_STAMP(var)                                                                           \
    do {                                                                                        \
        if (STAMP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");  \
    } while (0)
    PP_STAMP(ts0);

    // XCD-aware tile id (speed only): ids sharing id % 8 take a contiguous range, N-tiles of one M-tile adjacent
    auto tile_coords = [&](int id, long &tm0, int &tn0) __attribute__((always_inline)) {
        const int full = (ntiles / 8) * 8;
        int lin = id;
        if (id < full) lin = (id % 8) * (full / 8) + id / 8;
        tm0 = (long)(lin / tiles_n) * BM;
        tn0 = (lin % tiles_n) * BN;
    };
    int tile_id = blockIdx.x;
    if (tile_id >= ntiles) return;
    const int G = gridDim.x;
    long m0;
    int n0;
    tile_coords(tile_id, m0, n0);

    // ---- staging units = what one phase reads (all waves together), 128 rows x 128 B = 16 KiB each:
    //   unit 0 "XA": activation rows {0..63, 128..191}    (first 64 rows of each M half, read in phase 0)
    //   unit 1 "WA": weight rows with (row & 8) == 0       (N blocks 0, 1 of every wave,  read in phase 0)
    //   unit 2 "WB": weight rows with (row & 8) != 0       (N blocks 2, 3,                read in phase 1)
    //   unit 3 "XB": activation rows {64..127, 192..255}   (last 64 rows of each M half,  read in phase 2)
    // wave w stages unit-rows [16w, 16w+16) as two instructions of 8 rows x 128 B; rows keep their natural place in LDS
    // per-lane byte offsets inside a tile (the same for every tile) + two wave-uniform tile base pointers
    unsigned src_off[4][2];
    int dst_off[4][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int ub = 16 * wave + 8 * q;  // first unit-row of this instruction (wave-uniform, multiple of 8)
        const int row0[4] = {ub < 64 ? ub : ub + 64, (ub >> 3) * 16, (ub >> 3) * 16 + 8, ub < 64 ? ub + 64 : ub + 128};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = row0[u] + (lane >> 3);
            const bool is_x = (u == 0 || u == 3);
            const int f = is_x ? ((r >> 1) & 7) : (2 * ((r >> 4) & 3) + ((r >> 1) & 1));
            const int chunk = (lane & 7) ^ f;
            src_off[u][q] = (unsigned)(r * (is_x ? ldx : Kp) + 8 * chunk) * 2u;
            dst_off[u][q] = (is_x ? 0 : 2 * kHalfBytes) + row0[u] * 128;
        }
    }
    // tile bases live in two buffer descriptors (SGPRs); the K offset is the instruction's scalar offset
    __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(X + m0 * ldx), 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(W + (long)n0 * Kp), 0, 0x7fffffff, 0x00020000);
#ifndef PP_ABLATE
#define PP_ABLATE 0  // diagnostics only (timing experiments, results are wrong when != 0): 1 = no LDS-DMA, 2 = no ds_reads
#endif
    const int x_wrap = X3 ? 2 * (Kp / 3) : 0;  // X3 = 1: activation columns [0, 2K) serve k in [0, 2K) and, again from 0, k in [2K, 3K)
    // one unit (two instructions) into tile buffer `buf`: X units (0, 3) at activation column kx, W units (1, 2) at weight column kw
    auto stage_x = [&](int buf, int unit, int kx) __attribute__((always_inline)) {
        if (PP_ABLATE & 1) return;
        char *base = smem + buf * kTileBytes;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(base + dst_off[unit][0]), 16, src_off[unit][0], 2 * kx, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(base + dst_off[unit][1]), 16, src_off[unit][1], 2 * kx, 0, 0);
    };
    auto stage_w = [&](int buf, int unit, int kw) __attribute__((always_inline)) {
        if (PP_ABLATE & 1) return;
        char *base = smem + buf * kTileBytes;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(base + dst_off[unit][0]), 16, src_off[unit][0], 2 * kw, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(base + dst_off[unit][1]), 16, src_off[unit][1], 2 * kw, 0, 0);
    };
    auto stage = [&](int buf, int unit, int k0) __attribute__((always_inline)) {
        if (unit == 0 || unit == 3) stage_x(buf, unit, (X3 == 1 && k0 >= x_wrap) ? k0 - x_wrap : k0);
        else stage_w(buf, unit, k0);
    };

    // ---- fragment addresses: lane (row l15 of a 16-row block, k-chunk g4), K-substep s: chunk 4s + g4
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    unsigned x_addr[2], w_addr[2];  // buffer 0; + kTileBytes for buffer 1
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        x_addr[s] = lds0 + wm * kHalfBytes + l15 * 128 + (((4 * s + g4) ^ (l15 >> 1)) * 16);  // + ib * 2048
        // weight row of MFMA row l15 = 4a + b in N-block jb: 16a + 4jb + b;  f = 2a + (b >> 1) does not depend on jb
        const int wr = 16 * (l15 >> 2) + (l15 & 3);
        const int fw_ = 2 * (l15 >> 2) + ((l15 >> 1) & 1);
        w_addr[s] = lds0 + 2 * kHalfBytes + wn * 64 * 128 + wr * 128 + (((4 * s + g4) ^ fw_) * 16);  // + jb * 512
    }

    f32x4 acc[8][4];
    bf16x8 fx[4][2];   // activation rows of the current M quadrant half (4 blocks x 2 K-substeps)
    bf16x8 fw[4][2];   // weight rows: blocks 0,1 = N half 0, blocks 2,3 = N half 1

#if PP_ABLATE & 2
#define PP_DS128(dst, addr, imm) asm volatile("; no read %0 %1" : "=v"(dst) : "v"(addr))
#else
#define PP_DS128(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr))
#endif
#define PP_BAR() asm volatile("s_barrier" ::: "memory")
#define PP_SB() __builtin_amdgcn_sched_barrier(0)
#define PP_READ_X(boff, IMM0, IMM1, IMM2, IMM3)            \
    do {                                                   \
        const unsigned x0_ = x_addr[0] + (boff), x1_ = x_addr[1] + (boff); \
        PP_DS128(fx[0][0], x0_, IMM0);                     \
        PP_DS128(fx[0][1], x1_, IMM0);                     \
        PP_DS128(fx[1][0], x0_, IMM1);                     \
        PP_DS128(fx[1][1], x1_, IMM1);                     \
        PP_DS128(fx[2][0], x0_, IMM2);                     \
        PP_DS128(fx[2][1], x1_, IMM2);                     \
        PP_DS128(fx[3][0], x0_, IMM3);                     \
        PP_DS128(fx[3][1], x1_, IMM3);                     \
    } while (0)
#define PP_READ_W(J0, boff, IMM0, IMM1)                    \
    do {                                                   \
        const unsigned w0_ = w_addr[0] + (boff), w1_ = w_addr[1] + (boff); \
        PP_DS128(fw[J0][0], w0_, IMM0);                    \
        PP_DS128(fw[J0][1], w1_, IMM0);                    \
        PP_DS128(fw[(J0) + 1][0], w0_, IMM1);              \
        PP_DS128(fw[(J0) + 1][1], w1_, IMM1);              \
    } while (0)
// the fragments a phase has just loaded are in/out operands of its wait, so no use can be scheduled above it (and
// only those: naming dead fragments would keep them alive and push the accumulators out of the register file)
#define PP_WAIT_X()                                                                                                \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                            \
                 : "+v"(fx[0][0]), "+v"(fx[0][1]), "+v"(fx[1][0]), "+v"(fx[1][1]), "+v"(fx[2][0]), "+v"(fx[2][1]), \
                   "+v"(fx[3][0]), "+v"(fx[3][1])::"memory")
#define PP_WAIT_W(J0)                                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fw[J0][0]), "+v"(fw[J0][1]), "+v"(fw[(J0) + 1][0]), "+v"(fw[(J0) + 1][1])::"memory")
// one quadrant: M blocks I0..I0+3 (fragments fx[0..3]) x N blocks J0, J0+1 (fragments fw[J0], fw[J0+1]) x 2 K-substeps
#define PP_MFMA16(I0, J0)                                                                                        \
    do {                                                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                           \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) _Pragma("unroll") for (int i = 0; i < 4; ++i)              \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                        \
                acc[(I0) + i][(J0) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[(J0) + j][s], fx[i][s],     \
                                                                                  acc[(I0) + i][(J0) + j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                           \
    } while (0)
#define PP_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define PP_VMCNT_C(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")  // N: a constant expression
    // vector-memory operations one PP_STORE_Q issues: the layer's own rows (4 stores, 8 as [hi | lo] pairs) and / or the head sums (4)
    constexpr int SPQ = (STORE_Y ? (X3 ? 8 : 4) : 0) + (HEADS ? 4 : 0);
// Deferred epilogue of the PREVIOUS tile, one quadrant: bias + activation + bf16 pack, one 16-byte store per row (the
// N-blocks J0, J0+1 of a lane are 8 consecutive columns), then the accumulators restart from zero.  Runs in the load
// half of a phase, i.e. while the partner wave on this SIMD is in its MFMA half.
// one output element: activation, bf16 (and, X3, the second term bf16(value - hi))
#define PP_E1(IDX, VAL)                                         \
    do {                                                        \
        const float t_ = act_fn<ACT>(VAL);                      \
        o_[IDX] = (__bf16)t_;                                   \
        if (X3) l_[IDX] = bf16_lo_(t_, o_[IDX]);                \
    } while (0)
#define PP_STORE_Q(I0, J0)                                                                              \
    do {                                                                                                \
        bf16x8 hh_, hl_;                                                                                \
        if (HEADS) { /* this lane's head row (hi and lo terms) over the 8 columns of this half: 2 LDS reads, transient */ \
            const unsigned ha_ = hfrag_addr + ((J0) ? 16u : 0u);                                        \
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"           \
                         : "=&v"(hh_), "=&v"(hl_) : "v"(ha_), "v"(ha_ + hlo_off) : "memory");            \
        }                                                                                               \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                 \
            const f32x4 v_ = acc[(I0) + i][(J0)], w_ = acc[(I0) + i][(J0) + 1];                         \
            bf16x8 o_, l_;                                                                              \
            PP_E1(0, v_[0] + bq[(J0)][0]); PP_E1(1, v_[1] + bq[(J0)][1]);                               \
            PP_E1(2, v_[2] + bq[(J0)][2]); PP_E1(3, v_[3] + bq[(J0)][3]);                               \
            PP_E1(4, w_[0] + bq[(J0) + 1][0]); PP_E1(5, w_[1] + bq[(J0) + 1][1]);                       \
            PP_E1(6, w_[2] + bq[(J0) + 1][2]); PP_E1(7, w_[3] + bq[(J0) + 1][3]);                       \
            if (STORE_Y) {                                                                              \
                __bf16 *const yp_ = Yp + (long)(((I0) + i) * 16) * ldy_t + ((J0) ? 8 : 0);              \
                *reinterpret_cast<bf16x8 *>(yp_) = o_;                                                  \
                if (X3) *reinterpret_cast<bf16x8 *>(yp_ + Np) = l_; /* the lo terms: columns [Np, 2 Np) */ \
            }                                                                                           \
            if (HEADS) { /* D[head][row] = sum over this lane group's 8 columns: hi and lo head terms (X3: and the lo activations) */ \
                f32x4 hq_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hh_, o_, (f32x4){0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0); \
                hq_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hl_, o_, hq_, 0, 0, 0);                   \
                if (X3) hq_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hh_, l_, hq_, 0, 0, 0);           \
                if (g4 == 0) {                                                                          \
                    float *const hp_ = Hp + (long)(((I0) + i) * 16) * hstride + ((J0) ? HEADS : 0);     \
                    if (HEADS == 4) *reinterpret_cast<f32x4 *>(hp_) = hq_;                              \
                    else *hp_ = hq_[0];                                                                 \
                }                                                                                       \
            }                                                                                           \
            acc[(I0) + i][(J0)] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};                                      \
            acc[(I0) + i][(J0) + 1] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};                                  \
            if (X3) PP_SB(); /* one row at a time: the split epilogue's temporaries must not pile up across rows */ \
        }                                                                                               \
        asm volatile("" ::: "memory");                                                                  \
        PP_SB();                                                                                        \
    } while (0)
// One K-step = 4 phases.  ST: also store the previous tile (4 stores per phase, issued BEFORE the phase's reads - while
// the fragment registers they fill are still free - and before its 2 LDS-DMA instructions).  V0..V3: the counted vmcnt of each phase = number of vector-memory operations issued after the DMA pair
// of two phases ago, which must have landed before the next phase reads it.
#ifndef PP_PHASES
#define PP_PHASES 4  // phases per K-step: 4 (16 MFMAs each) or 2 (32 MFMAs each, half the barrier hand-offs): measured equal (1110 TF)
#endif
#if PP_PHASES == 4
#define PP_KSTEP(ST)                                        \
    do {                                                    \
        PP_SB();                                            \
        if (ST) PP_STORE_Q(0, 0);                           \
        PP_READ_W(0, boff, 0, 512);                         \
        PP_READ_X(boff, 0, 2048, 4096, 6144);               \
        stage(nbuf, 0, k_next);                             \
        if (ST) PP_VMCNT_C(SPQ + 4); else PP_VMCNT(4);      \
        PP_BAR();                                           \
        PP_WAIT_W(0);                                       \
        PP_WAIT_X();                                        \
        PP_SB();                                            \
        PP_MFMA16(0, 0);                                    \
        PP_SB();                                            \
        PP_BAR();                                           \
        if (ST) PP_STORE_Q(0, 2);                           \
        PP_READ_W(2, boff, 1024, 1536);                     \
        stage(nbuf, 1, k_next);                             \
        if (ST) PP_VMCNT_C(2 * SPQ + 4); else PP_VMCNT(4);  \
        PP_BAR();                                           \
        PP_WAIT_W(2);                                       \
        PP_SB();                                            \
        PP_MFMA16(0, 2);                                    \
        PP_SB();                                            \
        PP_BAR();                                           \
        if (ST) PP_STORE_Q(4, 2);                           \
        PP_READ_X(boff, 8192, 10240, 12288, 14336);         \
        stage(nbuf, 2, k_next);                             \
        if (ST) PP_VMCNT_C(2 * SPQ + 4); else PP_VMCNT(4);  \
        PP_BAR();                                           \
        PP_WAIT_X();                                        \
        PP_SB();                                            \
        PP_MFMA16(4, 2);                                    \
        PP_SB();                                            \
        PP_BAR();                                           \
        if (ST) PP_STORE_Q(4, 0);                           \
        stage(nbuf, 3, k_next);                             \
        if (ST) PP_VMCNT_C(2 * SPQ + 4); else PP_VMCNT(4);  \
        PP_BAR();                                           \
        PP_SB();                                            \
        PP_MFMA16(4, 0);                                    \
        PP_SB();                                            \
        PP_BAR();                                           \
    } while (0)
#define PP_PROLOGUE_VMCNT() PP_VMCNT(4)
#else
// Two phases per K-step (M half 0, then M half 1), 32 MFMAs each.  Phase 0 stages the three units the next K-step's
// phase 0 reads (XA, WA, WB: 6 instructions), phase 1 the one its phase 1 reads (XB: 2).  Counted waits, placed after
// the phase's own DMA issue: phase 0 retires XB of THIS K-step (6 younger operations, + 8 stores in a store K-step),
// phase 1 retires XA / WA / WB of the next one (2 younger, + 8 stores) - each one phase before the first read.
#define PP_KSTEP(ST)                                        \
    do {                                                    \
        PP_SB();                                            \
        if (ST) {                                           \
            PP_STORE_Q(0, 0);                               \
            PP_STORE_Q(0, 2);                               \
        }                                                   \
        PP_READ_W(0, boff, 0, 512);                         \
        PP_READ_W(2, boff, 1024, 1536);                     \
        PP_READ_X(boff, 0, 2048, 4096, 6144);               \
        stage(nbuf, 0, k_next);                             \
        stage(nbuf, 1, k_next);                             \
        stage(nbuf, 2, k_next);                             \
        if (ST) PP_VMCNT(14); else PP_VMCNT(6);             \
        PP_BAR();                                           \
        PP_WAIT_W(0);                                       \
        PP_WAIT_W(2);                                       \
        PP_WAIT_X();                                        \
        PP_SB();                                            \
        PP_MFMA16(0, 0);                                    \
        PP_MFMA16(0, 2);                                    \
        PP_SB();                                            \
        PP_BAR();                                           \
        if (ST) {                                           \
            PP_STORE_Q(4, 2);                               \
            PP_STORE_Q(4, 0);                               \
        }                                                   \
        PP_READ_X(boff, 8192, 10240, 12288, 14336);         \
        stage(nbuf, 3, k_next);                             \
        if (ST) PP_VMCNT(10); else PP_VMCNT(2);             \
        PP_BAR();                                           \
        PP_WAIT_X();                                        \
        PP_SB();                                            \
        PP_MFMA16(4, 2);                                    \
        PP_MFMA16(4, 0);                                    \
        PP_SB();                                            \
        PP_BAR();                                           \
    } while (0)
#define PP_PROLOGUE_VMCNT() PP_VMCNT(2)
#endif

// X3 = 2: the three K-steps of one 64-deep block.  TYPE 1: xl wh (stages Xh for TYPE 2), TYPE 2: xh wh (keeps Wh, stages Wl for
// TYPE 3), TYPE 3: xh wl (keeps Xh, stages the next block's Xl and Wh).  A phase stages a unit only when the next K-step does not
// reuse it; the counted waits retire exactly what the NEXT phase reads and was staged (ops issued after that unit's pair):
//   T1: p0 WB of this K-step (previous T3 p2) 4 | p1 its XB (T3 p3) 2 | p3 XA' (T1 p0) 2;  with the previous tile's stores (SPQ per
//       phase, issued before the phase's pair) SPQ + 4 | 2 SPQ + 2 | 3 SPQ + 2
//   T2: p1 XB' (T1 p3) 2 | p3 WA'' (T2 p1) 2          T3: p0 WB'' (T2 p2) 2 | p3 XA*, WA* (T3 p0, p1) 4
// Staging distances as in the 4-unit K-step: every unit is staged >= 3 phases before its first read and re-staged >= 4 phases
// after its last one.
#define PP_KSTEP_T(TYPE, ST)                                                                    \
    do {                                                                                        \
        PP_SB();                                                                                \
        if (ST) PP_STORE_Q(0, 0);                                                               \
        PP_READ_W(0, woff, 0, 512);                                                             \
        PP_READ_X(aoff, 0, 2048, 4096, 6144);                                                   \
        if ((TYPE) != 2) stage_x(sx_buf, 0, sx_k);                                              \
        if ((TYPE) == 1) { if (ST) PP_VMCNT_C(SPQ + 4); else PP_VMCNT(4); } else if ((TYPE) == 3) PP_VMCNT(2); \
        PP_BAR();                                                                               \
        PP_WAIT_W(0);                                                                           \
        PP_WAIT_X();                                                                            \
        PP_SB();                                                                                \
        PP_MFMA16(0, 0);                                                                        \
        PP_SB();                                                                                \
        PP_BAR();                                                                               \
        if (ST) PP_STORE_Q(0, 2);                                                               \
        PP_READ_W(2, woff, 1024, 1536);                                                         \
        if ((TYPE) != 1) stage_w(sw_buf, 1, sw_k);                                              \
        if ((TYPE) == 1) { if (ST) PP_VMCNT_C(2 * SPQ + 2); else PP_VMCNT(2); } else if ((TYPE) == 2) PP_VMCNT(2); \
        PP_BAR();                                                                               \
        PP_WAIT_W(2);                                                                           \
        PP_SB();                                                                                \
        PP_MFMA16(0, 2);                                                                        \
        PP_SB();                                                                                \
        PP_BAR();                                                                               \
        if (ST) PP_STORE_Q(4, 2);                                                               \
        PP_READ_X(aoff, 8192, 10240, 12288, 14336);                                             \
        if ((TYPE) != 1) stage_w(sw_buf, 2, sw_k);                                              \
        PP_BAR();                                                                               \
        PP_WAIT_X();                                                                            \
        PP_SB();                                                                                \
        PP_MFMA16(4, 2);                                                                        \
        PP_SB();                                                                                \
        PP_BAR();                                                                               \
        if (ST) PP_STORE_Q(4, 0);                                                               \
        if ((TYPE) != 2) stage_x(sx_buf, 3, sx_k);                                              \
        if ((TYPE) == 1) { if (ST) PP_VMCNT_C(3 * SPQ + 2); else PP_VMCNT(2); } else if ((TYPE) == 2) PP_VMCNT(2); else PP_VMCNT(4); \
        PP_BAR();                                                                               \
        PP_SB();                                                                                \
        PP_MFMA16(4, 0);                                                                        \
        PP_SB();                                                                                \
        PP_BAR();                                                                               \
    } while (0)

    // ---- bias -> LDS once (before any LDS-DMA is in flight)
    float *const bias_lds = reinterpret_cast<float *>(smem + 2 * kTileBytes);
    for (int i = tid; i < Np; i += kThreads) bias_lds[i] = bias[i];
    // head rows as two bf16 terms: hfrag[0][h][n] = bf16(w), hfrag[1][h][n] = bf16(w - hi) (HEADS x Np <= 4 x 1024 values each)
    __bf16 *const hfrag = reinterpret_cast<__bf16 *>(smem + 2 * kTileBytes + kMaxBias * 4);
    if (HEADS) {
        for (int i = tid; i < HEADS * Np; i += kThreads) {
            const float w_ = head_w[i];
            const __bf16 hi_ = (__bf16)w_;
            hfrag[i] = hi_;
            hfrag[HEADS * Np + i] = bf16_lo_(w_, hi_);
        }
    }
    __syncthreads();
    // this lane's head row for MFMA A-fragment row l15 (rows >= HEADS repeat the last head: their D rows are never stored)
    const unsigned hlo_off = 2u * (unsigned)(HEADS * Np);
    const int hstride = HEADS ? (Np / BN) * 8 * HEADS : 0;  // floats per row of head_part: [(Np / 256) * 8 slots][HEADS]
    unsigned hfrag_addr = 0;  // LDS byte address of the fragment for the tile being stored (set with Yp)
    float *Hp = head_part;
    const unsigned bias_addr = lds0 + 2 * kTileBytes + 4u * (wn * 64 + 16 * g4);  // + 4 * n0 of the tile, + 16 * jb

    // ---- prologue: the four units of K-step 0 in the order they are first read (X3 = 2: K-step 0 is xl wh of block 0)
    const int K1 = X3 ? Kp / 3 : Kp;  // the layer's contraction length
    if (X3 == 2) {
        stage_x(0, 0, K1);
        stage_w(0, 1, 0);
        stage_w(0, 2, 0);
        stage_x(0, 3, K1);
    } else {
        stage(0, 0, 0);
        stage(0, 1, 0);
        stage(0, 2, 0);
        stage(0, 3, 0);
    }
    PP_PROLOGUE_VMCNT();  // what phase 0 of the first K-step reads has landed (this wave's rows)
    PP_BAR();
    if (wm == 1) PP_BAR();  // the second M half runs one barrier behind the first

#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 bq[4];            // bias of the 16 columns this lane stores, for the tile being written
    __bf16 *Yp = Y;         // this lane's first output element of that tile
    int ldy_t = ldy;
    int buf = 0;
    PP_STAMP(ts1);
    if (STAMP) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1)::"memory");
    bool have_prev = false;
    for (; tile_id < ntiles; tile_id += G) {
        tile_coords(tile_id, m0, n0);
        if (X3 == 2) {
            const int nblk = K1 / BK;
            for (int kb = 0; kb < nblk; ++kb) {
                {   // T1: xl wh from (A0, B0); Xh of this block -> A1
                    const unsigned aoff = 0u, woff = 0u;
                    const int sx_buf = 1, sx_k = kb * BK, sw_buf = 0, sw_k = 0;
                    (void)sw_buf; (void)sw_k;
                    const bool st = have_prev && kb == 0;
                    PP_KSTEP_T(1, st);
                }
                {   // T2: xh wh from (A1, B0); Wl of this block -> B1
                    const unsigned aoff = (unsigned)kTileBytes, woff = 0u;
                    const int sx_buf = 0, sx_k = 0, sw_buf = 1, sw_k = 2 * K1 + kb * BK;
                    (void)sx_buf; (void)sx_k;
                    PP_KSTEP_T(2, false);
                }
                {   // T3: xh wl from (A1, B1); the next block's Xl / Wh -> (A0, B0): of this tile, of the workgroup's next tile, else
                    // (harmlessly) block 0 of this tile again
                    int nb = (kb + 1) * BK;
                    if (kb + 1 == nblk) {
                        nb = 0;
                        if (tile_id + G < ntiles) {
                            long nm0;
                            int nn0;
                            tile_coords(tile_id + G, nm0, nn0);
                            rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(X + nm0 * ldx), 0, 0x7fffffff, 0x00020000);
                            rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(W + (long)nn0 * Kp), 0, 0x7fffffff, 0x00020000);
                        }
                    }
                    const unsigned aoff = (unsigned)kTileBytes, woff = (unsigned)kTileBytes;
                    const int sx_buf = 0, sx_k = K1 + nb, sw_buf = 0, sw_k = nb;
                    PP_KSTEP_T(3, false);
                }
            }
        } else
        for (int kt = 0; kt < ksteps; ++kt) {
            // what this K-step stages: the next K-step of this tile, else K-step 0 of this workgroup's next tile, else
            // (harmlessly) K-step 0 of the current tile again - nobody reads it
            int k_next = (kt + 1) * BK;
            if (kt + 1 == ksteps) {
                k_next = 0;
                if (tile_id + G < ntiles) {
                    long nm0;
                    int nn0;
                    tile_coords(tile_id + G, nm0, nn0);
                    rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(X + nm0 * ldx), 0, 0x7fffffff, 0x00020000);
                    rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(W + (long)nn0 * Kp), 0, 0x7fffffff, 0x00020000);
                }
            }
            const unsigned boff = buf ? (unsigned)kTileBytes : 0u;
            const int nbuf = buf ^ 1;
            // K-step 0 of every tile but the first also carries the stores of the tile just finished (wave-uniform
            // branches in the load halves only).  The K-step after it could allow 8 outstanding operations in its
            // phase 0; 4 merely also waits for the last four stores, issued a whole phase earlier.
            const bool st = have_prev && kt == 0;
            PP_KSTEP(st);
            buf ^= 1;
        }
        PP_STAMP(ts2);
        // the finished tile is written during the first K-step of the next one (or below, if it is the last): its bias
        {
            const unsigned ba = bias_addr + 4u * n0;
            PP_DS128(bq[0], ba, 0);
            PP_DS128(bq[1], ba, 16);
            PP_DS128(bq[2], ba, 32);
            PP_DS128(bq[3], ba, 48);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3])::"memory");
            Yp = Y + (m0 + wm * 128 + l15) * ldy_t + n0 + wn * 64 + 16 * g4;
            if (HEADS) {
                const int hrow = l15 < HEADS ? l15 : HEADS - 1;
                hfrag_addr = lds0 + 2 * kTileBytes + kMaxBias * 4 + 2u * (unsigned)(hrow * Np + n0 + wn * 64 + 16 * g4);
                Hp = head_part + (m0 + wm * 128 + l15) * hstride + ((n0 / BN) * 8 + wn * 2) * HEADS;
            }
            asm volatile("" : "+s"(ldy_t));
            have_prev = true;
        }
    }
    if (STAMP) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt2)::"memory");
    // ---- the last tile of this workgroup: plain epilogue (no barrier: the groups keep their one-barrier stagger)
    PP_STORE_Q(0, 0);
    PP_STORE_Q(0, 2);
    PP_STORE_Q(4, 2);
    PP_STORE_Q(4, 0);
    if (wm == 0) PP_BAR();  // equalise the barrier count of the two groups
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (STAMP) {  // diagnostic build only: [0] prologue, [1] main loop, [2] epilogue (cycles, wave 0 of the first 256 workgroups)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PP_STAMP(ts3);
        if (tid == 0 && blockIdx.x < 256) {
            using namespace persist;
            M360_STAMP_STORE(0, ts1 - ts0);
            M360_STAMP_STORE(1, ts2 - ts1);
            M360_STAMP_STORE(2, ts3 - ts2);
            M360_STAMP_STORE(3, ts0);
            M360_STAMP_STORE(4, ts3);
            M360_STAMP_STORE(5, rt2 - rt1);  // in-kernel clock = [1] / [5] x 100 MHz (MI355X_MICROARCH.md, DVFS item 6)
        }
    }
#undef PP_STAMP
#undef PP_DS128
#undef PP_BAR
#undef PP_SB
#undef PP_READ_X
#undef PP_READ_W
#undef PP_WAIT_X
#undef PP_WAIT_W
#undef PP_MFMA16
#undef PP_VMCNT
#undef PP_STORE_Q
#undef PP_E1
#undef PP_KSTEP
#undef PP_KSTEP_T
#undef PP_PROLOGUE_VMCNT
}

}  // namespace pp16
}  // namespace m360
