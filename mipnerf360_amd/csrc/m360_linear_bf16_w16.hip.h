// Third bf16 linear kernel (round 3): ONE wave per SIMD, 128 x 128 wave tiles on v_mfma_f32_16x16x32_bf16, 4-slab LDS ring.
//
// Why: the 8-wave ping-pong kernel (m360_linear_bf16_pp.hip.h) reads 192 KiB of fragments and writes 64 KiB of LDS-DMA per 64-deep
// K-step and CU - the whole 128 B/clk of the LDS in the 2048 cycles the matrix work takes.  A probe of the bare K-step
// (tools/wide_wave_probe.hip, profiles/r03/bf16_wide_wave_probe.jsonl, same box) puts that geometry's bound at 1.32-1.34 PF
// and the bound of FOUR waves with 128 x 128 wave tiles (128 KiB of fragment reads) at 1.57-1.58 PF - with the 16-cycle MFMA:
// the 32-cycle v_mfma_f32_32x32x16_bf16 of round 2's attempt at this geometry (diag/m360_linear_bf16_w32.hip.h, 0.98-1.03 PF)
// draws more power per flop (1.46 PF at 1.46 GHz in the same probe).  That attempt lost 24 % of its time in an epilogue whose
// stores all 256 workgroups issued in the same microseconds, 32 bytes per row and instruction; its skeleton is kept:
//   * a RING of four 32-deep slabs (32 KiB each: 256 activation + 256 weight rows x 64 B, source-side XOR swizzle), a 1-KiB LDS-DMA
//     piece issued 2.5-3.5 slabs before its first read, one per 8th MFMA gap, counted vmcnt(16), ONE barrier per slab;
//   * a GENERATED schedule (tools/gen_w16_slab.py -> m360_linear_bf16_w16_gen.inc): every ds_read_b128 / LDS-DMA piece in its own
//     MFMA gap, ring positions static (slab offsets are instruction immediates), the counted waits from a simulation of the issue
//     order;
//   * operands swapped (MFMA A := weight rows, B := activation rows): D[i][j], lane (j = lane & 15, g4 = lane >> 4) holds rows
//     i = 4 g4 + r of 16 output columns for ONE activation row.
// New here:
//   * a slab is ONE k-step of the 16x16x32 MFMA: 8 x 8 blocks = 64 MFMAs in two halves (activation blocks 0-3 | 4-7); fragment
//     registers: one set of 8 activation fragments (refilled half by half) + two sets of 8 weight fragments = 96 VGPRs;
//   * weight rows permuted so that a lane's accumulators of weight blocks 2p, 2p+1 are 8 CONSECUTIVE output columns and the four
//     lanes of a row cover 32 consecutive columns: MFMA row i of block jb is column 32 (jb >> 1) + 8 (i >> 2) + 4 (jb & 1) + (i & 3)
//     of the wave's 128.  Epilogue per (activation block, p): 4 packed bias adds, 4 v_cvt_pk_bf16_f32, ReLU as v_pk_max_i16 on the
//     packed pairs, ONE 16-byte store - 16 rows x 64 contiguous bytes per instruction, no LDS transposition;
//   * the last slab of a tile also issues the weight pieces its successor would issue in slab 0, so every piece the next tile
//     needs through ITS slab 3 is older than the epilogue's 32 stores: the first counted wait that covers the stores comes 3.5
//     slabs (3.8 k cycles) after the epilogue;
//   * the workgroups of an XCD start up to `stagger` x 3 x 64 cycles apart (4 classes), so that the chip never sees all 256
//     epilogues in the same microsecond (32 MiB of stores at once: 6 us at the 5.35 TB/s of a pure store kernel).
// Takes full 256 x 256 tiles of layers with K a multiple of 128 (>= 256), bias + {none, ReLU}; the sigmoid / fused-heads layers
// stay with the ping-pong kernel (its partner wave hides the transcendental epilogue).
#pragma once
#include "m360_common.hip.h"

namespace m360 {
namespace w16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;

constexpr int BM = 256, BN = 256, BKS = 32;  // slab depth
constexpr int kThreads = 256;
constexpr int kXBytes = 256 * 64;            // activation rows of one slab
constexpr int kSlabBytes = 2 * kXBytes;      // + weight rows
constexpr int kMaxBias = 4096;

#ifdef M360_DIAG
// diagnostics build, per workgroup: [0] cycles (s_memtime) and [1] 100 MHz ticks of the tile loop, [2] slabs, [3] cycles in epilogues
__device__ unsigned long long g_w16_stamps[256 * 4];
#endif
// ABL (diagnostic builds; results are wrong unless 0): 1 = no barrier, 2 = no LDS-DMA, 4 = no fragment reads, 16 = no stores,
// 32 = no epilogue at all
template <int ACT, int ABL = 0, bool STAMP = false>
__global__ __launch_bounds__(kThreads, 1) void linear_bf16_w16_kernel(
    const __bf16 *__restrict__ X, long M, int ldx, const __bf16 *__restrict__ W, const float *__restrict__ bias, int Np,
    int Kp, __bf16 *__restrict__ Y, int ldy, int tiles_n, int ntiles, int stagger) {
    __shared__ __attribute__((aligned(1024))) char smem[4 * kSlabBytes + kMaxBias * 4];  // 144 KiB
    static_assert(ACT == M360_ACT_NONE || ACT == M360_ACT_RELU, "bias + {none, ReLU} only");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int G = gridDim.x;
    const int nslabs = Kp / BKS;  // a multiple of 4, >= 8
    const int kbytes = 2 * Kp;

    auto tile_coords = [&](int id, long &tm0, int &tn0) __attribute__((always_inline)) {
        const int full = (ntiles / 8) * 8;  // XCD-aware (speed only): ids sharing id % 8 cover a contiguous range of tiles
        int lin = id;
        if (id < full) lin = (id % 8) * (full / 8) + id / 8;
        tm0 = (long)(lin / tiles_n) * BM;
        tn0 = (lin % tiles_n) * BN;
    };
    int tile_id = blockIdx.x;
    if (tile_id >= ntiles) return;

    // ---- LDS-DMA: wave w stages rows [64w, 64w + 64) of both operands, 4 pieces of 16 rows x 64 B each.  LDS slot of chunk c of
    // row r: c ^ f(r), f = (r >> 2) & 3 for activation rows (a 16-lane read group touches 16 consecutive rows) and (r >> 3) & 3 for
    // weight rows (a read group touches rows C + 8 a + b): either way 16 distinct 16-byte slots of the 256-byte bank line
    unsigned x_voff[4], w_voff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 64 * wave + 16 * q + (lane >> 2);
        x_voff[q] = (unsigned)(r * ldx + 8 * ((lane & 3) ^ ((r >> 2) & 3))) * 2u;
        w_voff[q] = (unsigned)(r * Kp + 8 * ((lane & 3) ^ ((r >> 3) & 3))) * 2u;
    }
    char *const dma_x = smem + 64 * wave * 64;            // + slot * kSlabBytes + q * 1024
    char *const dma_w = smem + kXBytes + 64 * wave * 64;
    // two cursors run ahead of the matrix work, across tile boundaries: the activation pieces of slab t + 4 and the weight
    // pieces of slab t + 3 (scalar state: buffer descriptor of the cursor's tile + byte offset of its slab in a row)
    __amdgpu_buffer_rsrc_t rsrc_xd, rsrc_wd;
    int kx = 0, kw = 0, tile_xd = tile_id, tile_wd = tile_id;
    {
        long m0_;
        int n0_;
        tile_coords(tile_id, m0_, n0_);
        rsrc_xd = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(X + m0_ * ldx), 0, 0x7fffffff, 0x00020000);
        rsrc_wd = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(W + (long)n0_ * Kp), 0, 0x7fffffff, 0x00020000);
    }
#define W16_ADV_X()                                                                                                          \
    do {                                                                                                                     \
        kx += 2 * BKS;                                                                                                       \
        if (kx == kbytes) { /* next tile of this workgroup (past the last one: harmlessly the same rows again) */            \
            kx = 0;                                                                                                          \
            tile_xd += G;                                                                                                    \
            if (tile_xd < ntiles) {                                                                                          \
                long m0_;                                                                                                    \
                int n0_;                                                                                                     \
                tile_coords(tile_xd, m0_, n0_);                                                                              \
                rsrc_xd = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(X + m0_ * ldx), 0, 0x7fffffff, 0x00020000);  \
            }                                                                                                                \
        }                                                                                                                    \
    } while (0)
#define W16_ADV_W()                                                                                                          \
    do {                                                                                                                     \
        kw += 2 * BKS;                                                                                                       \
        if (kw == kbytes) {                                                                                                  \
            kw = 0;                                                                                                          \
            tile_wd += G;                                                                                                    \
            if (tile_wd < ntiles) {                                                                                          \
                long m0_;                                                                                                    \
                int n0_;                                                                                                     \
                tile_coords(tile_wd, m0_, n0_);                                                                              \
                rsrc_wd = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(W + (long)n0_ * Kp), 0, 0x7fffffff, 0x00020000); \
            }                                                                                                                \
        }                                                                                                                    \
    } while (0)
#define W16_DMA_X(SLOT, Q) if (!(ABL & 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_xd, (lds_ptr_t)(dma_x + (SLOT) * kSlabBytes + (Q) * 1024), 16, x_voff[Q], kx, 0, 0)
#define W16_DMA_W(SLOT, Q) if (!(ABL & 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_wd, (lds_ptr_t)(dma_w + (SLOT) * kSlabBytes + (Q) * 1024), 16, w_voff[Q], kw, 0, 0)

    // ---- fragment reads: lane (l15, g4) reads chunk g4 (k = 8 g4 .. 8 g4 + 7) of its row.  Activation block ib: row 16 ib + l15
    // of the wave's 128.  Weight block jb: MFMA row i = l15 is LDS row 32 (jb >> 1) + 8 (i >> 2) + 4 (jb & 1) + (i & 3) - so the
    // accumulators acc[.][2p][0..3], acc[.][2p + 1][0..3] of lane group g4 are output columns 32 p + 8 g4 + 0..7
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const int wr0 = 8 * (l15 >> 2) + (l15 & 3);  // weight row of block 0
    // byte addresses in ring slots 0 / 1 (l) and 2 / 3 (h); odd slots and the blocks are instruction immediates
    const unsigned xal = lds0 + (wm * 128 + l15) * 64 + ((g4 ^ ((l15 >> 2) & 3)) * 16);
    const unsigned wal = lds0 + kXBytes + (wn * 128 + wr0) * 64 + ((g4 ^ ((wr0 >> 3) & 3)) * 16);
    const unsigned xah = xal + 2 * kSlabBytes, wah = wal + 2 * kSlabBytes;

    f32x4 acc[8][8];  // [activation block][weight block]
    bf16x8 fx[8], fw0[8], fw1[8];

#define W16_RD(dst, addr, imm)                                                                          \
    do {                                                                                                \
        if (!(ABL & 4)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm)); \
        else asm volatile("" : "=v"(dst) : "v"(addr));                                                  \
    } while (0)
#define W16_SB() __builtin_amdgcn_sched_barrier(0)
// The MFMAs are inline assembly with the accumulator tied to an AccVGPR ("+a"): with the builtin the register allocator kept half
// of the 64 accumulator tuples in ArchVGPRs across the slab bodies and copied them in and out around every MFMA (4 v_accvgpr_write
// + s_nop per MFMA in the K loop).  The hazard recogniser does not see these MFMAs: the epilogue waits out the last one itself.
#define W16_MFMA(ACC, A, B) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(A), "v"(B))
#define W16_MFMA_Z(ACC, A, B) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(ACC) : "v"(A), "v"(B))
// the fragments a wait has just covered are in/out operands of it, so no use can be scheduled above it
#define W16_TIE_HI "+v"(fx[4]), "+v"(fx[5]), "+v"(fx[6]), "+v"(fx[7])
#define W16_WAIT_NEXT(FW)                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                     \
                 : "+v"(fx[0]), "+v"(fx[1]), "+v"(fx[2]), "+v"(fx[3]), "+v"(FW[0]), "+v"(FW[1]), "+v"(FW[2]), "+v"(FW[3]),  \
                   "+v"(FW[4]), "+v"(FW[5]), "+v"(FW[6]), "+v"(FW[7])::"memory")
// this wave's pieces of the NEXT slab have landed (N younger operations may stay in flight), every wave's reads of this slab are done
#define W16_BARRIER(N)                                                                                  \
    do {                                                                                                \
        if (!(ABL & 1)) asm volatile("s_waitcnt vmcnt(%4) lgkmcnt(0)\n\ts_barrier" : W16_TIE_HI : "n"(N) : "memory"); \
        else asm volatile("s_waitcnt vmcnt(%4) lgkmcnt(0)" : W16_TIE_HI : "n"(N) : "memory");           \
    } while (0)
// the first three slabs after an epilogue: its 32 stores are younger than the pieces waited for (a bare counted wait under the
// branch: two register-tied variants would meet in a join and cost copies)
#define W16_BARRIER_E(N, NS)                                                                            \
    do {                                                                                                \
        if (have_prev) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS) : "memory");                        \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");                                   \
        if (!(ABL & 1)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : W16_TIE_HI::"memory");       \
        else asm volatile("s_waitcnt lgkmcnt(0)" : W16_TIE_HI::"memory");                               \
    } while (0)
#include "m360_linear_bf16_w16_gen.inc"

    // ---- bias -> LDS once (before any LDS-DMA is in flight)
    float *const bias_lds = reinterpret_cast<float *>(smem + 4 * kSlabBytes);
    for (int i = tid; i < Np; i += kThreads) bias_lds[i] = bias[i];
    __syncthreads();
    const unsigned bias_addr = lds0 + 4 * kSlabBytes + 4u * (wn * 128 + 8 * g4);  // + 4 * n0 of the tile, + 128 * p

    // ---- start stagger: class c = (workgroup / 8) & 3 of an XCD waits 3 c x stagger x 64 cycles
    {
        const int cls = (blockIdx.x >> 3) & 3;
        for (int i = 0; i < cls * stagger; ++i) { __builtin_amdgcn_s_sleep(3); }
    }

    // ---- prologue: slabs 0..3 of the first tile in the order of their first reads
    W16_DMA_X(0, 0); W16_DMA_X(0, 1); W16_DMA_X(0, 2); W16_DMA_X(0, 3); W16_ADV_X();
    W16_DMA_W(0, 0); W16_DMA_W(0, 1); W16_DMA_W(0, 2); W16_DMA_W(0, 3); W16_ADV_W();
    W16_DMA_X(1, 0); W16_DMA_X(1, 1); W16_DMA_X(1, 2); W16_DMA_X(1, 3); W16_ADV_X();
    W16_DMA_W(1, 0); W16_DMA_W(1, 1); W16_DMA_W(1, 2); W16_DMA_W(1, 3); W16_ADV_W();
    W16_DMA_X(2, 0); W16_DMA_X(2, 1); W16_DMA_X(2, 2); W16_DMA_X(2, 3); W16_ADV_X();
    W16_DMA_W(2, 0); W16_DMA_W(2, 1); W16_DMA_W(2, 2); W16_DMA_W(2, 3); W16_ADV_W();
    W16_DMA_X(3, 0); W16_DMA_X(3, 1); W16_DMA_X(3, 2); W16_DMA_X(3, 3); W16_ADV_X();
    W16_DMA_W(3, 0); W16_DMA_W(3, 1); W16_DMA_W(3, 2); W16_DMA_W(3, 3); W16_ADV_W();
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");  // slab 0 has landed (this wave's rows)
    __builtin_amdgcn_s_barrier();
    W16_SB();
    W16_RD(fw0[0], wal, 0); W16_RD(fw0[1], wal, 256); W16_RD(fw0[2], wal, 2048); W16_RD(fw0[3], wal, 2304);
    W16_RD(fw0[4], wal, 4096); W16_RD(fw0[5], wal, 4352); W16_RD(fw0[6], wal, 6144); W16_RD(fw0[7], wal, 6400);
    W16_RD(fx[0], xal, 0); W16_RD(fx[1], xal, 1024); W16_RD(fx[2], xal, 2048); W16_RD(fx[3], xal, 3072);
    W16_WAIT_NEXT(fw0);
    W16_SB();

    const unsigned y_voff = (unsigned)(l15 * ldy + 8 * g4) * 2u;  // this lane's 16 bytes inside a 16-row x 32-column piece
    bool have_prev = false;
    long m0;
    int n0;
    unsigned long long mt0 = 0, rt0 = 0, mt1 = 0, rt1 = 0, e0 = 0, e1 = 0, te = 0, nsl = 0;
    (void)mt1; (void)rt1; (void)e0; (void)e1; (void)te; (void)nsl;
    if (STAMP) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(mt0), "=s"(rt0)::"memory");
    for (; tile_id < ntiles; tile_id += G) {
        tile_coords(tile_id, m0, n0);
        W16_SLAB0Z();
        W16_SLAB1E();
        W16_SLAB2E();
        W16_SLAB3();
        for (int s = 4; s < nslabs - 4; s += 4) {
            W16_SLAB0();
            W16_SLAB1();
            W16_SLAB2();
            W16_SLAB3();
        }
        W16_SLAB0();
        W16_SLAB1();
        W16_SLAB2();
        W16_SLAB3L();
        // ---- epilogue (exposed; the pieces of the next tile's slabs 0-3 are in flight or landed meanwhile)
        if (STAMP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(e0)::"memory");
        // the last MFMAs (inline assembly: the hazard recogniser does not see them) have written their accumulators before anything
        // reads them: the 8 tuples of the last 8 MFMAs are redefined by this statement, every other tuple is >= 128 cycles old
        asm volatile("s_nop 15\n\ts_nop 15"
                     : "+a"(acc[7][0]), "+a"(acc[7][1]), "+a"(acc[7][2]), "+a"(acc[7][3]), "+a"(acc[7][4]), "+a"(acc[7][5]),
                       "+a"(acc[7][6]), "+a"(acc[7][7])::"memory");
        if (!(ABL & 32)) {
            const __bf16 *yt = Y + (m0 + wm * 128) * ldy + n0 + wn * 128;  // wave-uniform corner of the wave tile
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                f32x4 b0, b1;  // bias of the lane's columns 32 p + 8 g4 + 0..7
                const unsigned ba = bias_addr + 4u * (n0 + 32 * p);
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(b0), "=&v"(b1) : "v"(ba) : "memory");
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    // explicit AccVGPR reads, one (activation block, piece) at a time: left to the register allocator, 150 of the
                    // 256 accumulators were copied out at the top of the epilogue and an address register was spilled
                    f32x4 v, w;
                    asm volatile("v_accvgpr_read_b32 %0, %8\n\tv_accvgpr_read_b32 %1, %9\n\tv_accvgpr_read_b32 %2, %10\n\t"
                                 "v_accvgpr_read_b32 %3, %11\n\tv_accvgpr_read_b32 %4, %12\n\tv_accvgpr_read_b32 %5, %13\n\t"
                                 "v_accvgpr_read_b32 %6, %14\n\tv_accvgpr_read_b32 %7, %15"
                                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
                                 : "a"(acc[i][2 * p][0]), "a"(acc[i][2 * p][1]), "a"(acc[i][2 * p][2]), "a"(acc[i][2 * p][3]),
                                   "a"(acc[i][2 * p + 1][0]), "a"(acc[i][2 * p + 1][1]), "a"(acc[i][2 * p + 1][2]), "a"(acc[i][2 * p + 1][3]));
                    u32x4 o;
#define W16_PACK(e, a0, a1, bb, be)                                                                            \
    do {                                                                                                       \
        f32x2 t_ = {a0 + bb[be], a1 + bb[(be) + 1]};                                                           \
        s16x2 p_ = __builtin_bit_cast(s16x2, __builtin_convertvector(t_, bf16x2));                             \
        if (ACT == M360_ACT_RELU) p_ = __builtin_elementwise_max(p_, (s16x2){0, 0});                           \
        o[e] = __builtin_bit_cast(unsigned, p_);                                                               \
    } while (0)
                    W16_PACK(0, v[0], v[1], b0, 0); W16_PACK(1, v[2], v[3], b0, 2);
                    W16_PACK(2, w[0], w[1], b1, 0); W16_PACK(3, w[2], w[3], b1, 2);
#undef W16_PACK
                    const __bf16 *row = yt + (long)(16 * i) * ldy + 32 * p;
                    // s_nop: a store of more than 8 bytes still reads its data registers in the cycle after issue, and the next
                    // instruction here (an AccVGPR read, invisible to the hazard recogniser like this store) may write them
                    if (!(ABL & 16)) asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(y_voff), "v"(o), "s"(row) : "memory");
                    else asm volatile("" ::"v"(o));
                }
                W16_SB();  // one column piece at a time: 256 accumulator reads hoisted together would not fit the register file
            }
        }
        if (STAMP) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(e1)::"memory"); te += e1 - e0; nsl += nslabs; }
        have_prev = !(ABL & 48);
    }
    if (STAMP) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(mt1), "=s"(rt1)::"memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA of this wave may land after the workgroup is gone
#ifdef M360_DIAG
    if (STAMP && tid == 0 && blockIdx.x < 256) {
        g_w16_stamps[blockIdx.x * 4 + 0] = mt1 - mt0;
        g_w16_stamps[blockIdx.x * 4 + 1] = rt1 - rt0;
        g_w16_stamps[blockIdx.x * 4 + 2] = nsl;
        g_w16_stamps[blockIdx.x * 4 + 3] = te;
    }
#endif
#undef W16_ADV_X
#undef W16_ADV_W
#undef W16_DMA_X
#undef W16_DMA_W
#undef W16_RD
#undef W16_SB
#undef W16_MFMA
#undef W16_MFMA_Z
#undef W16_TIE_HI
#undef W16_WAIT_NEXT
#undef W16_BARRIER
#undef W16_BARRIER_E
}

}  // namespace w16
}  // namespace m360
