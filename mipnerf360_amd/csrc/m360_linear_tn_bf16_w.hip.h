// Weight-gradient GEMM of the bf16 training path, second form (round 5): ONE wave per SIMD with 128 x 128 wave tiles - the structure of the
// forward ring kernel (m360_linear_bf16_w16.hip.h) for the "row = contraction index" operand layout of dW = dZ^T X.
//
// Why a second form: the 8-wave kernel (m360_linear_tn_bf16.hip.h) reads 24 fragments per 32 MFMAs and wave, meets at one barrier per 64-row
// stage and prefetches ONE stage ahead; its ablations (profiles/r05/wgrad_bf16_ablation_8wave_form.txt) show LDS-DMA, fragment reads and matrix
// work overlapping only partly (0.84 / 0.39 / 0.72 ms alone, 1.15-1.19 ms together per 1024^2 layer; MFMA pipe busy 41 %).  Here:
//   * 4 waves, wave tile 128 (n) x 128 (k) = 8 x 8 blocks of v_mfma_f32_16x16x32_bf16: 256 accumulators in AccVGPRs (MFMAs as inline assembly with
//     "+a" operands, as in the ring kernel), 16 fragments = 32 ds_read_b64_tr_b16 per 64 MFMAs: two thirds of the LDS read traffic per flop;
//   * the unit of the loop is a k-step of 32 rows; the LDS holds FIVE such "quarters" (5 x 32 KiB = all 160 KiB), filled four k-steps ahead of the
//     matrix work by LDS-DMA (8 pieces per wave and k-step, one per 8 MFMAs) - the lead the L2 -> LDS latency needs (~1.5 us under load against
//     ~0.45 us of matrix work per k-step);
//   * the fragments of k-step i + 1 are read BETWEEN the MFMAs of k-step i (two per 8 MFMAs: the X fragment a group has just finished with is
//     overwritten in place, the dZ fragments alternate between two register sets), behind the one barrier of the k-step that makes quarter i + 1
//     complete for every wave;
//   * the LDS-DMA instructions are inline assembly (M0 write + buffer_load_dwordx4 .. lds): the compiler must not know that they write the LDS, or
//     it puts s_waitcnt vmcnt(0) in front of every LDS read behind one (which waits for pieces issued four k-steps ahead); the transposed reads are
//     the ds_read_tr16_b64 builtin (register pairs assembled without copies, lgkmcnt waits placed by the compiler per fragment).  The orderings
//     that matter are kept by hand: vmcnt(16) + s_barrier in front of the first read of a quarter, and a quarter is refilled only behind the
//     barrier after its last read.
// (Measured, no gain: FOUR quarters in flight instead of three - k-step i + 5 into the quarter whose fragments k-step i - 1 already took into
// registers, vmcnt(24) - 1.10 against 1.06-1.09 ms on a box whose other kernels ran 2-3 % slow: the L2 -> LDS delivery rate bounds the loop, not the
// bytes in flight.)
// Same LDS image (512-byte rows, 32-byte units XOR-swizzled by f(r): conflict-free transposed reads), same (tile, split) decomposition, same
// deterministic split reduction and the same bits in the partial sums' layout as the 8-wave kernel.
#pragma once
#include "m360_linear_tn_bf16.hip.h"

namespace m360 {
namespace tn16w {

using tn16::bf16x8;
using tn16::f32x4;
using tn16::lds_s16x4_p;
using tn16::s16x4;
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int BT = 256;                 // output tile edge
constexpr int KS = 32;                  // rows (contraction) per k-step = per LDS quarter
constexpr int kThreads = 256;
constexpr int kRowBytes = BT * 2;       // 512
constexpr int kOpBytes = KS * kRowBytes;  // 16 KiB: one operand of one k-step
constexpr int kQuarterBytes = 2 * kOpBytes;  // dZ rows | X rows
constexpr int kQuarters = 5;

// ABL (diagnostics build only, M360_TNW_ABL; results are wrong unless 0): 1 = no LDS-DMA pieces in the loop, 2 = no fragment reads in the loop,
// 4 = no MFMAs, 8 = no counted wait + barrier per k-step, 16 = no epilogue stores - which of the loop's three streams overlap badly?
template <int ABL = 0>
__global__ __launch_bounds__(kThreads, 1) void linear_tn_bf16_w_kernel(
    const __bf16 *__restrict__ dZ, int ldz, const __bf16 *__restrict__ X, int ldx, int Np, int Kp,
    float *__restrict__ partial /*[nsplit][Np][Kp]*/, int tiles_k, int ntiles, int nsplit, long total_steps /* k-steps of 32 rows */,
    long steps_per_split, float *__restrict__ bias_partial /*[nsplit][Np] or nullptr*/) {
    __shared__ __attribute__((aligned(1024))) char smem[kQuarters * kQuarterBytes];  // 160 KiB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wk = wave & 1;
    const int g = lane >> 4, l15 = lane & 15, q = l15 >> 2, p = l15 & 3;

    const int total = ntiles * nsplit;
    int split, tile;
    if (total % 8 == 0 && (total / 8) % ntiles == 0) {  // all tiles of a split on ONE XCD (speed only)
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile = j % ntiles;
        split = x * ((total / 8) / ntiles) + j / ntiles;
    } else {
        split = blockIdx.x / ntiles;
        tile = blockIdx.x % ntiles;
    }
    const int kt = tile % tiles_k;
    const int n0 = (tile / tiles_k) * BT, k0 = kt * BT;
    const long s_begin = (long)split * steps_per_split;
    long s_end = s_begin + steps_per_split;
    if (s_end > total_steps) s_end = total_steps;
    const long nk = s_end > s_begin ? s_end - s_begin : 0;  // k-steps of this workgroup
    // bias gradient (the host sends layers with fewer than 4 k tiles to the 8-wave kernel): in the workgroups of k tiles kt < 4 wave (wn, wk) adds up dZ
    // block 2 kt + wk of its n half against a fragment of ones - 1 extra MFMA per 64 and wave.  (Measured: the bias gradient costs the kernel ~0.1 ms of
    // 1.0 whichever way its two MFMAs per k-step are issued - selected by v_cndmask on two waves, by a wave-uniform switch, or one per wave as here.)
    const bool bias_wave = bias_partial != nullptr && kt < 4;
    const int bias_blk = 2 * kt + wk;  // wave-uniform: the dZ block (of this wave's n half) whose column sums this wave forms - ONE extra MFMA per k-step and wave
    f32x4 acc[8][8], bacc = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[a][j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    if (nk > 0) {
        // ---- LDS-DMA: wave w stages pieces 4w .. 4w+3 of each operand of a quarter (rows 8w .. 8w+7 of its 32); lane L of a piece lands at byte 16 L =
        // row 2 pi + (L >> 5), 16-byte slot L & 31, and fetches the chunk whose 32-byte unit the swizzle puts there
        unsigned va[4], vb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = 2 * (4 * wave + e) + (lane >> 5), s = lane & 31;
            const int c = 2 * ((s >> 1) ^ tn16::swz(r)) + (s & 1);
            va[e] = (unsigned)(r * ldz + 8 * c) * 2u;
            vb[e] = (unsigned)(r * ldx + 8 * c) * 2u;
        }
        const unsigned long long pa = (unsigned long long)(dZ + s_begin * KS * ldz + n0), pb = (unsigned long long)(X + s_begin * KS * ldx + k0);
        i32x4 ra, rb;
        ra[0] = __builtin_amdgcn_readfirstlane((int)pa); ra[1] = __builtin_amdgcn_readfirstlane((int)(pa >> 32)); ra[2] = 0x7fffffff; ra[3] = 0x00020000;
        rb[0] = __builtin_amdgcn_readfirstlane((int)pb); rb[1] = __builtin_amdgcn_readfirstlane((int)(pb >> 32)); rb[2] = 0x7fffffff; rb[3] = 0x00020000;
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
        const unsigned dma0 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 4u * 1024u);
        const unsigned step_a = (unsigned)KS * (unsigned)ldz * 2u, step_b = (unsigned)KS * (unsigned)ldx * 2u;  // bytes per k-step
        // one piece of k-step `ks` (clamped to the last one: the pipeline's tail re-loads rows nobody reads) into quarter `qd`; e = 0..3: dZ, 4..7: X
#define TNW_DMA(E, QD_OFF, SOFF_A, SOFF_B)                                                                                                   \
    do {                                                                                                                                     \
        if ((E) < 4) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dma0 + (QD_OFF) + (unsigned)((E) & 3) * 1024u), "v"(va[(E) & 3]), "s"(ra), "s"(SOFF_A) : "memory"); \
        else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(dma0 + (QD_OFF) + (unsigned)kOpBytes + (unsigned)((E) & 3) * 1024u), "v"(vb[(E) & 3]), "s"(rb), "s"(SOFF_B) : "memory"); \
    } while (0)

        // ---- transposed fragment reads (tn16): lane (g, q, p) supplies row 8g + q (+ 4) of block c, bytes 8p of its 32-byte unit in slot (c & 7) ^ f;
        // block a of this wave's operand = base ^ (a << 5) (the slot bits 5 - 7 are nobody else's)
        const int f = q | ((g & 1) << 2);
        const unsigned lane_off = (unsigned)((8 * g + q) * kRowBytes + 8 * p);
        const unsigned a_base = lds0 + lane_off + (unsigned)(f + 8 * wn) * 32u;
        const unsigned b_base = lds0 + (unsigned)kOpBytes + lane_off + (unsigned)(f + 8 * wk) * 32u;
        auto frag = [&](unsigned addr) __attribute__((always_inline)) -> bf16x8 {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(uintptr_t)addr);
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(uintptr_t)(addr + 4 * kRowBytes));
            return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        };
        const bf16x8 ones = {(__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f};
#define TNW_MFMA(ACC, A, B) do { if (!(ABL & 4)) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B)); } while (0)
// the bias sum's MFMA: accumulator in ArchVGPRs; s_nop 1 in front: one operand is a VALU result (the ones), and an MFMA must not read a VGPR within 2
// wait states of the VALU write - behind inline assembly the hazard recogniser cannot insert them; 2 x s_nop 15 behind: the MFMA's own latency (above)
#define TNW_MFMA_V1(ACC, A, B)                                                                                                          \
    asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 15" : "+v"(ACC) : "v"(A), "v"(B))
#define TNW_SB() __builtin_amdgcn_sched_barrier(0)

        // ---- prologue: k-steps 0 .. 3 on their way (8 pieces each), quarter 0 awaited and read
        const long last = nk - 1;
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const unsigned ks = (unsigned)(k4 < last ? k4 : last);
#pragma unroll
            for (int e = 0; e < 8; ++e) TNW_DMA(e, (unsigned)k4 * (unsigned)kQuarterBytes, ks * step_a, ks * step_b);
        }
        asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        __syncthreads();
        // Round 6: every non-matrix instruction of a k-step in its OWN MFMA gap, like the forward ring kernel's generated stages.  The ablations of
        // round 5's loop (profiles/r06/wgrad_bf16_ablation_one_wave_form.txt: M360_TNW_ABL in the diagnostics build) showed its three streams adding
        // up instead of overlapping - matrix work alone 0.63 ms, + fragment reads 0.89, + LDS-DMA 0.88, all three 1.11 - because the 4 transposed
        // reads and the 3 instructions of a piece were issued in one clump behind every 8 MFMAs (the matrix pipe drains while a lone wave issues
        // 7 other instructions), and the last fragments of a k-step were read at its very end and awaited at the top of the next one.  Now: 64
        // gaps per k-step; transposed read r (0 .. 31: the 16 halves of the next k-step's dZ fragments, then the X fragments') in gap
        // floor(1.5 r) - all issued 17 MFMAs before the k-step ends; LDS-DMA piece q in gap 2 + 6 q (free: 2 mod 3).  Both fragment sets are double
        // buffered (an MFMA reads its sources at issue, the LDS returns data later: a read must not target a register a later MFMA of the same
        // k-step still reads).  Same MFMA order, same accumulation order: same bits.
        bf16x8 fa[2][8], fb[2][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) fb[0][j] = frag(b_base ^ (unsigned)(j << 5));
#pragma unroll
        for (int a = 0; a < 8; ++a) fa[0][a] = frag(a_base ^ (unsigned)(a << 5));

        unsigned qn = 1;  // quarter of k-step i + 1
        for (long i = 0; i < nk; i += 2) {
#pragma unroll
            for (int par = 0; par < 2; ++par) {  // the two fragment sets alternate: unrolled so that both are static
                const long ii = i + par;
                if (ii < nk) {
                    // quarter ii + 1 complete (this wave's pieces: all but the 16 youngest of the 24 in flight; everyone's: the barrier), quarter ii - 1 free
                    if (!(ABL & 8)) {
                        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                        __syncthreads();
                    }
                    const unsigned qoff = qn * (unsigned)kQuarterBytes;                       // where k-step ii + 1 lies
                    const unsigned qd = (qn + 3 >= (unsigned)kQuarters ? qn + 3 - kQuarters : qn + 3) * (unsigned)kQuarterBytes;  // quarter of k-step ii + 4 (= of ii - 1)
                    const unsigned ksn = (unsigned)(ii + 4 < last ? ii + 4 : last);
                    const unsigned soff_a = ksn * step_a, soff_b = ksn * step_b;
                    const unsigned aq = a_base + qoff, bq = b_base + qoff;
                    // (the last k-step reads ahead too - a quarter that holds stale rows, into fragments nobody uses: no branch in the loop)
                    if (bias_wave) {
                        switch (bias_blk) {
                            case 0: TNW_MFMA_V1(bacc, ones, fa[par][0]); break;
                            case 1: TNW_MFMA_V1(bacc, ones, fa[par][1]); break;
                            case 2: TNW_MFMA_V1(bacc, ones, fa[par][2]); break;
                            case 3: TNW_MFMA_V1(bacc, ones, fa[par][3]); break;
                            case 4: TNW_MFMA_V1(bacc, ones, fa[par][4]); break;
                            case 5: TNW_MFMA_V1(bacc, ones, fa[par][5]); break;
                            case 6: TNW_MFMA_V1(bacc, ones, fa[par][6]); break;
                            default: TNW_MFMA_V1(bacc, ones, fa[par][7]); break;
                        }
                    }
                    s16x4 ha[8][2], hb[8][2];  // the 32 halves read in this k-step
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
#pragma unroll
                        for (int a = 0; a < 8; ++a) {
                            TNW_MFMA(acc[a][j], fb[par][j], fa[par][a]);
                            const int gap = 8 * j + a;
                            if (gap % 3 != 2 && gap <= 46 && !(ABL & 2)) {  // transposed read r = gap - gap / 3
                                const int r = gap - gap / 3, f_ = (r & 15) >> 1, hf = r & 1;
                                const unsigned addr = ((r < 16 ? aq : bq) ^ (unsigned)(f_ << 5)) + (unsigned)(hf * 4 * kRowBytes);
                                const s16x4 v_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(uintptr_t)addr);
                                if (r < 16) ha[f_][hf] = v_; else hb[f_][hf] = v_;
                            }
#ifndef M360_TNW_DMA_LATE
                            if (gap % 6 == 2 && gap <= 44 && !(ABL & 1)) TNW_DMA(gap / 6, qd, soff_a, soff_b);
#else  // A/B: the pieces behind the reads, every other gap of the k-step's tail
                            if (gap >= 47 && gap % 2 == 1 && gap <= 61 && !(ABL & 1)) TNW_DMA((gap - 47) / 2, qd, soff_a, soff_b);
#endif
                            TNW_SB();
                        }
                    }
                    if (!(ABL & 2)) {
#pragma unroll
                        for (int f_ = 0; f_ < 8; ++f_) {
                            fa[par ^ 1][f_] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(ha[f_][0], ha[f_][1], 0, 1, 2, 3, 4, 5, 6, 7));
                            fb[par ^ 1][f_] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(hb[f_][0], hb[f_][1], 0, 1, 2, 3, 4, 5, 6, 7));
                        }
                    } else {
#pragma unroll
                        for (int f_ = 0; f_ < 8; ++f_) { fa[par ^ 1][f_] = fa[par][f_]; fb[par ^ 1][f_] = fb[par][f_]; }
                    }
                    TNW_SB();
                    qn = qn + 1 == (unsigned)kQuarters ? 0u : qn + 1;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA of this wave may land after the workgroup is gone
        // the last MFMAs (inline assembly: the hazard recogniser does not see them) have written their accumulators before anything reads them
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#undef TNW_DMA
#undef TNW_MFMA
#undef TNW_MFMA_V1
#undef TNW_SB
    }

    // ---- epilogue: acc[a][j][r] of lane (l15, g) = dW[n0 + 128 wn + 16 a + l15][k0 + 128 wk + 16 j + 4 g + r]
    float *__restrict__ P = partial + (long)split * Np * Kp;
    const int nrow = n0 + wn * 128 + l15, kcol = k0 + wk * 128 + 4 * g;
    if (!(ABL & 16))
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4 *>(P + (long)(nrow + 16 * a) * Kp + kcol + 16 * j) = acc[a][j];
    if (bias_wave && g == 0)  // every row of the ones product holds the column sums: row 0 = lanes 0 .. 15, register 0
        bias_partial[(long)split * Np + nrow + 16 * bias_blk] = bacc[0];
}

}  // namespace tn16w
}  // namespace m360
