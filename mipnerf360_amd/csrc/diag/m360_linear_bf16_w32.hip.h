// Fourth bf16 structure (diagnostics build only - it lost the A/B against the ping-pong kernel, see the end of this comment):
// ONE wave per SIMD, v_mfma_f32_32x32x16_bf16, 4-slab LDS ring.
//
// What round 2's clocks and microbenchmarks said about the 8-wave ping-pong kernel (m360_linear_bf16_pp.hip.h, 0.45 of the
// peak): beside 16-cycle MFMAs (16x16x32) every LDS-DMA issue costs 15-38 cycles of matrix time and a ds_write 20-27, beside
// 32-cycle MFMAs (32x32x16) the same fillers are nearly free and the bare issue efficiency is 92.5 % instead of 87.4 %
// (tools/mfma_filler_cost.hip); its 128 x 64 wave tiles read 256 KB of LDS per 64-deep K-step and CU - exactly the 128 B/clk
// the LDS delivers in the 2048 cycles the matrix work takes - and a two-stage buffer cannot keep the 1 us of L2 / HBM latency
// times the 30 B/clk/CU the kernel needs in flight.  This kernel therefore
//   * gives each of 4 waves (one per SIMD, the whole register file) a 128 x 128 wave tile = 4 x 4 blocks of 32x32x16: 256
//     accumulator registers, 192 KB of LDS traffic per 64 deep instead of 256;
//   * streams the operands through a RING of four 32-deep slabs (32 KiB each: 256 activation + 256 weight rows x 64 B,
//     source-side XOR swizzle slot = chunk ^ ((row >> 2) & 3)): a 1-KiB LDS-DMA piece is issued 2.5-3.5 slabs (2600-3600
//     cycles) before its first read, one every 4th MFMA gap, counted vmcnt(16), ONE barrier per slab (1024 MFMA cycles);
//   * has a GENERATED schedule (tools/gen_w32_slab.py -> m360_linear_bf16_w32_gen.inc): every ds_read_b128 / LDS-DMA piece in
//     its own MFMA gap, ring positions static (slab offsets are instruction immediates: no address arithmetic in the loop);
//   * swaps the operands (MFMA A := weight rows, B := activation rows) and permutes the weight rows of a 32-block so that a
//     lane's 16 accumulators of a block are two runs of 8 consecutive output columns: bias (2 packed adds per 4 values),
//     v_cvt_pk_bf16_f32, ReLU as v_pk_max_i16 on the packed pair, 16-byte stores with a scalar row base - no LDS transposition.
//     The epilogue is exposed (nothing hides vector work behind the same wave's MFMAs): ~2.6 k cycles per
//     32.8 k-cycle tile at K = 1024, which is why the sigmoid layers stay with the ping-pong kernel (its partner wave hides them).
// Takes full 256 x 256 tiles of layers with K a multiple of 128 (>= 128), bias + {none, ReLU}.
// Measured (profiles/r02/bf16_w32_ablations.jsonl, M = 524288, 1024 x 1024): right on the first run, 979-1032 TF against the
// ping-pong kernel's 1130.  The matrix stream alone needs 1038 cycles per 1024-cycle slab and the fragment reads + barrier
// 10 more (as planned), but the LDS-DMA stream adds 357 (the L2 -> LDS path delivers ~40-50 GB/s per CU here), the 32-byte
// row pieces of the direct stores make the epilogue 10.4 k cycles per tile instead of 2.6 k, and the chip holds 1.64 GHz under
// this kernel against 1.87 GHz under the ping-pong kernel (2.38 GHz with the loads ablated): 73 % of the issue slots at 1.64
// GHz and 58 % at 1.87 GHz are the same 1.1 PF.
#pragma once
#include "../m360_common.hip.h"

namespace m360 {
namespace w32 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
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
__device__ unsigned long long g_w32_stamps[256 * 4];
#endif
// ABL (diagnostic builds; results are wrong unless 0): 1 = no barrier, 2 = no LDS-DMA, 4 = no fragment reads, 16 = no stores,
// 32 = no epilogue at all
template <int ACT, int ABL = 0, bool STAMP = false>
__global__ __launch_bounds__(kThreads, 1) void linear_bf16_w32_kernel(
    const __bf16 *__restrict__ X, long M, int ldx, const __bf16 *__restrict__ W, const float *__restrict__ bias, int Np,
    int Kp, __bf16 *__restrict__ Y, int ldy, int tiles_n, int ntiles) {
    __shared__ __attribute__((aligned(1024))) char smem[4 * kSlabBytes + kMaxBias * 4];  // 144 KiB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int G = gridDim.x;
    const int nslabs = Kp / BKS;  // a multiple of 4, >= 4
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

    // ---- LDS-DMA: wave w stages rows [64w, 64w + 64) of both operands, 4 pieces of 16 rows x 64 B each
    unsigned x_voff[4], w_voff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 64 * wave + 16 * q + (lane >> 2);
        const int chunk = (lane & 3) ^ ((r >> 2) & 3);
        x_voff[q] = (unsigned)(r * ldx + 8 * chunk) * 2u;
        w_voff[q] = (unsigned)(r * Kp + 8 * chunk) * 2u;
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
#define W32_ADV_X()                                                                                                          \
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
#define W32_ADV_W()                                                                                                          \
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
#define W32_DMA_X(SLOT, Q) if (!(ABL & 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_xd, (lds_ptr_t)(dma_x + (SLOT) * kSlabBytes + (Q) * 1024), 16, x_voff[Q], kx, 0, 0)
#define W32_DMA_W(SLOT, Q) if (!(ABL & 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_wd, (lds_ptr_t)(dma_w + (SLOT) * kSlabBytes + (Q) * 1024), 16, w_voff[Q], kw, 0, 0)

    // ---- fragment reads: lane (l31, h), slice s of a slab: chunk 2s + h of its row = slot (2s + h) ^ ((row >> 2) & 3).
    // MFMA row i of a weight block is weight row n(i) of the block: lane (m, h) then holds columns 16 (r >> 3) + 8 h + (r & 7)
    const int pr = (l31 & 3) + 4 * (l31 >> 3);                       // accumulator register of MFMA row l31 (in its h half)
    const int pl = 16 * (pr >> 3) + 8 * ((l31 >> 2) & 1) + (pr & 7);  // weight row of MFMA row l31
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const unsigned xrow = lds0 + (wm * 128 + l31) * 64, wrow = lds0 + kXBytes + (wn * 128 + pl) * 64;
    const int fxs = (l31 >> 2) & 3, fws = (pl >> 2) & 3;
    // byte addresses in ring slots 0 / 1 (l) and 2 / 3 (h); odd slots and the 32-row blocks are instruction immediates
    const unsigned xa0l = xrow + ((0 + h) ^ fxs) * 16, xa1l = xrow + ((2 + h) ^ fxs) * 16;
    const unsigned wa0l = wrow + ((0 + h) ^ fws) * 16, wa1l = wrow + ((2 + h) ^ fws) * 16;
    const unsigned xa0h = xa0l + 2 * kSlabBytes, xa1h = xa1l + 2 * kSlabBytes, wa0h = wa0l + 2 * kSlabBytes, wa1h = wa1l + 2 * kSlabBytes;

    f32x16 acc[4][4];  // [activation block I][weight block J]: rows of the MFMA = output columns
    bf16x8 fx0[4], fw0[4], fx1[4], fw1[4];
    const f32x16 kZero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

#define W32_RD(dst, addr, imm)                                                                          \
    do {                                                                                                \
        if (!(ABL & 4)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm)); \
        else asm volatile("" : "=v"(dst) : "v"(addr));                                                  \
    } while (0)
#define W32_SB() __builtin_amdgcn_sched_barrier(0)
#define W32_WAIT(FX, FW)                                                                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                     \
                 : "+v"(FX[0]), "+v"(FX[1]), "+v"(FX[2]), "+v"(FX[3]), "+v"(FW[0]), "+v"(FW[1]), "+v"(FW[2]), "+v"(FW[3])::"memory")
// this wave's pieces of the NEXT slab have landed (16 younger pieces may stay in flight), every wave's reads of this slab are done
#define W32_BARRIER(FX, FW)                                                                                                 \
    do {                                                                                                                    \
        if (!(ABL & 1))                                                                                                     \
            asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier"                                                      \
                         : "+v"(FX[0]), "+v"(FX[1]), "+v"(FX[2]), "+v"(FX[3]), "+v"(FW[0]), "+v"(FW[1]), "+v"(FW[2]), "+v"(FW[3])::"memory"); \
        else                                                                                                                \
            asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)"                                                                   \
                         : "+v"(FX[0]), "+v"(FX[1]), "+v"(FX[2]), "+v"(FX[3]), "+v"(FW[0]), "+v"(FW[1]), "+v"(FW[2]), "+v"(FW[3])::"memory"); \
    } while (0)
// the first two slabs after an epilogue: its 32 stores are younger than the pieces waited for (a bare counted wait under the
// branch: two register-tied variants would meet in a join and cost copies)
#define W32_BARRIER_E(FX, FW)                                                                                               \
    do {                                                                                                                    \
        if (have_prev) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");                                                    \
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");                                                              \
        if (!(ABL & 1))                                                                                                     \
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier"                                                                \
                         : "+v"(FX[0]), "+v"(FX[1]), "+v"(FX[2]), "+v"(FX[3]), "+v"(FW[0]), "+v"(FW[1]), "+v"(FW[2]), "+v"(FW[3])::"memory"); \
        else                                                                                                                \
            asm volatile("s_waitcnt lgkmcnt(0)"                                                                             \
                         : "+v"(FX[0]), "+v"(FX[1]), "+v"(FX[2]), "+v"(FX[3]), "+v"(FW[0]), "+v"(FW[1]), "+v"(FW[2]), "+v"(FW[3])::"memory"); \
    } while (0)
#include "m360_linear_bf16_w32_gen.inc"

    // ---- bias -> LDS once (before any LDS-DMA is in flight)
    float *const bias_lds = reinterpret_cast<float *>(smem + 4 * kSlabBytes);
    for (int i = tid; i < Np; i += kThreads) bias_lds[i] = bias[i];
    __syncthreads();
    const unsigned bias_addr = lds0 + 4 * kSlabBytes + 4u * (wn * 128 + 8 * h);  // + 4 * n0 of the tile, + 128 * J, + {0, 16, 64, 80}

    // ---- prologue: slabs 0..3 of the first tile (weights of slab 3 come with body 0), in the order of their first reads
    W32_DMA_X(0, 0); W32_DMA_X(0, 1); W32_DMA_X(0, 2); W32_DMA_X(0, 3); W32_ADV_X();
    W32_DMA_W(0, 0); W32_DMA_W(0, 1); W32_DMA_W(0, 2); W32_DMA_W(0, 3); W32_ADV_W();
    W32_DMA_X(1, 0); W32_DMA_X(1, 1); W32_DMA_X(1, 2); W32_DMA_X(1, 3); W32_ADV_X();
    W32_DMA_W(1, 0); W32_DMA_W(1, 1); W32_DMA_W(1, 2); W32_DMA_W(1, 3); W32_ADV_W();
    W32_DMA_X(2, 0); W32_DMA_X(2, 1); W32_DMA_X(2, 2); W32_DMA_X(2, 3); W32_ADV_X();
    W32_DMA_W(2, 0); W32_DMA_W(2, 1); W32_DMA_W(2, 2); W32_DMA_W(2, 3); W32_ADV_W();
    W32_DMA_X(3, 0); W32_DMA_X(3, 1); W32_DMA_X(3, 2); W32_DMA_X(3, 3); W32_ADV_X();
    asm volatile("s_waitcnt vmcnt(20)" ::: "memory");  // slab 0 has landed (this wave's rows)
    __builtin_amdgcn_s_barrier();
    W32_SB();
    W32_RD(fx0[0], xa0l, 0); W32_RD(fx0[1], xa0l, 2048); W32_RD(fx0[2], xa0l, 4096); W32_RD(fx0[3], xa0l, 6144);
    W32_RD(fw0[0], wa0l, 0); W32_RD(fw0[1], wa0l, 2048); W32_RD(fw0[2], wa0l, 4096); W32_RD(fw0[3], wa0l, 6144);
    W32_WAIT(fx0, fw0);
    W32_SB();

    const unsigned y_voff = (unsigned)(l31 * ldy + 8 * h) * 2u;  // this lane's 16 bytes inside a 32-row x 16-column piece
    bool have_prev = false;
    long m0;
    int n0;
    unsigned long long mt0 = 0, rt0 = 0, mt1 = 0, rt1 = 0, e0 = 0, e1 = 0, te = 0, nsl = 0;
    (void)mt1; (void)rt1; (void)e0; (void)e1; (void)te; (void)nsl;
    if (STAMP) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(mt0), "=s"(rt0)::"memory");
    for (; tile_id < ntiles; tile_id += G) {
        tile_coords(tile_id, m0, n0);
        W32_SLAB0Z(W32_BARRIER_E);
        W32_SLAB1(W32_BARRIER_E);
        W32_SLAB2(W32_BARRIER);
        W32_SLAB3(W32_BARRIER);
        for (int s = 4; s < nslabs; s += 4) {
            W32_SLAB0(W32_BARRIER);
            W32_SLAB1(W32_BARRIER);
            W32_SLAB2(W32_BARRIER);
            W32_SLAB3(W32_BARRIER);
        }
        // ---- epilogue (exposed; the DMA pieces of the next tile's first slabs are in flight meanwhile)
        if (STAMP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(e0)::"memory");
        if (!(ABL & 32)) {
            const __bf16 *yt = Y + (m0 + wm * 128) * ldy + n0 + wn * 128;  // wave-uniform corner of the wave tile
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 b0, b1, b2, b3;  // bias of the lane's columns 8h..8h+7 and 16+8h..16+8h+7 of block j
                const unsigned ba = bias_addr + 4u * (n0 + 32 * j);
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:64\n\t"
                             "ds_read_b128 %3, %4 offset:80\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(ba) : "memory");
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x16 v = acc[i][j];
                    u32x4 o0, o1;
#define W32_PACK(dst, e, lo, hi, bb, be)                                                                       \
    do {                                                                                                       \
        f32x2 t_ = {v[lo] + bb[be], v[hi] + bb[(be) + 1]};                                                     \
        s16x2 p_ = __builtin_bit_cast(s16x2, __builtin_convertvector(t_, bf16x2));                             \
        if (ACT == M360_ACT_RELU) p_ = __builtin_elementwise_max(p_, (s16x2){0, 0});                           \
        dst[e] = __builtin_bit_cast(unsigned, p_);                                                             \
    } while (0)
                    W32_PACK(o0, 0, 0, 1, b0, 0); W32_PACK(o0, 1, 2, 3, b0, 2); W32_PACK(o0, 2, 4, 5, b1, 0); W32_PACK(o0, 3, 6, 7, b1, 2);
                    W32_PACK(o1, 0, 8, 9, b2, 0); W32_PACK(o1, 1, 10, 11, b2, 2); W32_PACK(o1, 2, 12, 13, b3, 0); W32_PACK(o1, 3, 14, 15, b3, 2);
#undef W32_PACK
                    const __bf16 *row = yt + (long)(32 * i) * ldy + 32 * j;
                    if (!(ABL & 16)) {
                        asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(y_voff), "v"(o0), "s"(row) : "memory");
                        asm volatile("global_store_dwordx4 %0, %1, %2 offset:32" ::"v"(y_voff), "v"(o1), "s"(row) : "memory");
                    } else asm volatile("" ::"v"(o0), "v"(o1));
                    W32_SB();  // one block at a time: 256 accumulator reads hoisted together would not fit the register file
                }
            }
            W32_SB();
        }
        if (STAMP) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(e1)::"memory"); te += e1 - e0; nsl += nslabs; }
        have_prev = !(ABL & 48);
    }
    if (STAMP) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(mt1), "=s"(rt1)::"memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA of this wave may land after the workgroup is gone
#ifdef M360_DIAG
    if (STAMP && tid == 0 && blockIdx.x < 256) {
        g_w32_stamps[blockIdx.x * 4 + 0] = mt1 - mt0;
        g_w32_stamps[blockIdx.x * 4 + 1] = rt1 - rt0;
        g_w32_stamps[blockIdx.x * 4 + 2] = nsl;
        g_w32_stamps[blockIdx.x * 4 + 3] = te;
    }
#endif
#undef W32_ADV_X
#undef W32_ADV_W
#undef W32_DMA_X
#undef W32_DMA_W
#undef W32_RD
#undef W32_SB
#undef W32_WAIT
#undef W32_BARRIER
#undef W32_BARRIER_E
}

}  // namespace w32
}  // namespace m360
