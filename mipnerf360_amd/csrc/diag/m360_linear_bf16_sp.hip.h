// Third bf16 linear kernel (opt-in reduced-precision MLP, BASELINE configs[4]): 8 waves = two per SIMD, SOFTWARE-PIPELINED.
//
// Why a third structure.  Round-2 evidence (profiles/r02): the bare v_mfma_f32_16x16x32_bf16 loop delivers 2.01 PFLOP/s
// at 2.11 GHz on this device, the 8-wave ping-pong kernel (m360_linear_bf16_pp.hip.h) 1.12 PFLOP/s at an in-kernel clock
// of 1.85 GHz = 58 % of what the MFMA pipes could issue at the clock the chip holds: 55.0 k cycles per 256 x 256 x 1024
// tile against 32.8 k of MFMA issue.  It is stall-bound, not power-bound: in a ping-pong pairing a K-step costs the SUM of
// each phase's max(load segment, MFMA segment), and the load segments (reads + LDS-DMA issue + counted waits + 8 barrier
// hand-offs per K-step) are the longer ones.
//
// Here every wave runs ONE instruction stream in which the non-matrix work sits in MFMA shadows (as in the fp32 kernel,
// m360_linear_persist.hip.h), and the two waves of a SIMD simply interleave: while one wave is held by an LDS-DMA issue
// or a wait, the other one's MFMAs keep the pipe busy.  Same tile (256 x 256, K-step 64 bf16), same wave grid (2 along M
// x 4 along N, wave tile 128 x 64 = 8 x 4 blocks of 16x16x32 -> 128 accumulators), same LDS image (128-byte rows, source
// side XOR swizzle), same operand swap / weight-row permutation (a lane owns 16 consecutive output columns) as the
// ping-pong kernel.  A K-step is 4 units of 16 MFMAs:
//       u0 = (k-substep 0, M blocks 0-3)   u1 = (0, M blocks 4-7)   u2 = (1, 0-3)   u3 = (1, 4-7)
// with operand fragments double-buffered in registers (XA / XB: 4 activation blocks each; WA / WB: the 4 weight blocks
// of one k-substep): the reads of unit u+1 are issued between the MFMAs of unit u.  ONE barrier per K-step, between u2
// and u3: before it a wave has read everything it needs from the current LDS buffer and its own LDS-DMA pieces of the
// next K-step have landed (vmcnt(0)); after it u3 runs from registers while the fragments of the next K-step's u0 are
// read from the other buffer and the K-step after next is staged into the buffer just freed.
#pragma once
#include "../m360_common.hip.h"
#include "../m360_linear_persist.hip.h"  // diagnostic stamp buffer

namespace m360 {
namespace sp16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int kThreads = 512;
constexpr int kHalfBytes = 128 * 128;       // half-tile: 128 rows x 128 B
constexpr int kTileBytes = 4 * kHalfBytes;  // X rows 0-255, W rows 0-255 of one K-step
constexpr int kMaxBias = 4096;              // widest layer this kernel takes (bias is served from LDS)

template <int ACT>
__device__ __forceinline__ float act_fn(float v) {
    if (ACT == M360_ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == M360_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    return v;
}

// ABL (diagnostics build only, results are WRONG when != 0): timing ablations - 1 = never wait for the LDS-DMA (vmcnt),
// 2 = issue no LDS-DMA in the main loop, 4 = issue no ds_reads in the main loop
template <int ACT, bool STAMP = false, int ABL = 0>
__global__ __launch_bounds__(kThreads, 1) void linear_bf16_sp_kernel(
    const __bf16 *__restrict__ X, long M, int ldx, const __bf16 *__restrict__ W, const float *__restrict__ bias,
    int Np, int Kp, __bf16 *__restrict__ Y, int ldy, int tiles_n, int ntiles, int ldw) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * kTileBytes + kMaxBias * 4];  // 128 KiB + the bias vector

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int ksteps = Kp / BK;  // >= 2 (the host sends K = 64 layers to the one-wave-per-SIMD kernel)
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, rt1 = 0, rt2 = 0;
#define SP_STAMP(var)                                                                           \
    do {                                                                                        \
        if (STAMP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");  \
    } while (0)
    SP_STAMP(ts0);

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

    // ---- staging: 4 units of 128 rows x 128 B per K-step (X rows 0-127, X rows 128-255, W rows 0-127, W rows 128-255);
    // wave w stages rows [16w, 16w+16) of each unit as two LDS-DMA instructions of 8 rows x 128 B.  Per-lane byte offsets
    // inside a tile are tile-independent; the tile base sits in buffer descriptors (SGPRs), the K offset is scalar.
    unsigned src_off[4][2];
    int dst_off[4][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool is_x = u < 2;
            const int row0 = (u & 1) * 128 + 16 * wave + 8 * q;  // first row of this instruction inside its operand tile
            const int r = row0 + (lane >> 3);
            const int f = is_x ? ((r >> 1) & 7) : (2 * ((r >> 4) & 3) + ((r >> 1) & 1));
            const int chunk = (lane & 7) ^ f;
            src_off[u][q] = (unsigned)(r * (is_x ? ldx : ldw) + 8 * chunk) * 2u;
            dst_off[u][q] = (is_x ? 0 : 2 * kHalfBytes) + row0 * 128;
        }
    }
    auto make_x = [&](long tm0) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(X + tm0 * ldx), 0, 0x7fffffff, 0x00020000);
    };
    auto make_w = [&](int tn0) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(W + (long)tn0 * ldw), 0, 0x7fffffff, 0x00020000);
    };
    // one staging unit (2 instructions) of the K-step at element offset k0 of the tile behind (rx, rw) into LDS buffer `buf`
    auto stage = [&](int buf, int unit, __amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t rw, int k0) __attribute__((always_inline)) {
        char *base = smem + buf * kTileBytes;
        if (unit < 2) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(base + dst_off[unit][0]), 16, src_off[unit][0], 2 * k0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr_t)(base + dst_off[unit][1]), 16, src_off[unit][1], 2 * k0, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(base + dst_off[unit][0]), 16, src_off[unit][0], 2 * k0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)(base + dst_off[unit][1]), 16, src_off[unit][1], 2 * k0, 0, 0);
        }
    };

    // ---- fragment addresses: lane (row l15 of a 16-row block, k-chunk g4), K-substep s: chunk 4s + g4
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    unsigned x_addr[2], w_addr[2];  // buffer 0; + kTileBytes for buffer 1
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        x_addr[s] = lds0 + wm * kHalfBytes + l15 * 128 + (((4 * s + g4) ^ (l15 >> 1)) * 16);  // + block * 2048
        // weight row of MFMA row l15 = 4a + b in N-block jb: 16a + 4jb + b;  f = 2a + (b >> 1) does not depend on jb
        const int wr = 16 * (l15 >> 2) + (l15 & 3);
        const int fw_ = 2 * (l15 >> 2) + ((l15 >> 1) & 1);
        w_addr[s] = lds0 + 2 * kHalfBytes + wn * 64 * 128 + wr * 128 + (((4 * s + g4) ^ fw_) * 16);  // + jb * 512
    }

    f32x4 acc[8][4];
    bf16x8 XA[4], XB[4], WA[4], WB[4];  // register double buffers of the operand fragments

#define SP_DS128(dst, addr, imm)                                                                   \
    do {                                                                                           \
        if (ABL & 4) asm volatile("; no read %0 %1" : "=v"(dst) : "v"(addr));                      \
        else asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr));             \
    } while (0)
#define SP_SB() __builtin_amdgcn_sched_barrier(0)
// the fragments a unit is about to use are in/out operands of its wait: no use can be scheduled above it
#define SP_WAIT8(n, F, Gf)                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(" #n ")"                                                                            \
                 : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(Gf[0]), "+v"(Gf[1]), "+v"(Gf[2]), "+v"(Gf[3])::"memory")
#define SP_WAIT4(n, F) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3])::"memory")
#define SP_MFMA(I, J, XF, WF) acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[J], XF[(I) & 3], acc[I][J], 0, 0, 0)
#define SP_MFMA_B(I, J, XF, WF) acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[J], XF[(I) & 3], bq[J], 0, 0, 0)
// One unit = 16 MFMAs (M blocks I0..I0+3 x N blocks 0..3 of one k-substep); after every second MFMA one "slot" S0..S7
// (a ds_read_b128 of the next unit's fragments and / or an LDS-DMA instruction) is issued in the MFMA shadow.
#define SP_UNIT(MF, I0, XF, WF, S0, S1, S2, S3, S4, S5, S6, S7)   \
    do {                                                          \
        MF((I0) + 0, 0, XF, WF); MF((I0) + 0, 1, XF, WF); S0; SP_SB(); \
        MF((I0) + 0, 2, XF, WF); MF((I0) + 0, 3, XF, WF); S1; SP_SB(); \
        MF((I0) + 1, 0, XF, WF); MF((I0) + 1, 1, XF, WF); S2; SP_SB(); \
        MF((I0) + 1, 2, XF, WF); MF((I0) + 1, 3, XF, WF); S3; SP_SB(); \
        MF((I0) + 2, 0, XF, WF); MF((I0) + 2, 1, XF, WF); S4; SP_SB(); \
        MF((I0) + 2, 2, XF, WF); MF((I0) + 2, 3, XF, WF); S5; SP_SB(); \
        MF((I0) + 3, 0, XF, WF); MF((I0) + 3, 1, XF, WF); S6; SP_SB(); \
        MF((I0) + 3, 2, XF, WF); MF((I0) + 3, 3, XF, WF); S7; SP_SB(); \
    } while (0)
#define SP_NOP ((void)0)

    // ---- bias -> LDS once (before any LDS-DMA is in flight)
    float *const bias_lds = reinterpret_cast<float *>(smem + 2 * kTileBytes);
    for (int i = tid; i < Np; i += kThreads) bias_lds[i] = bias[i];
    __syncthreads();
    const unsigned bias_addr = lds0 + 2 * kTileBytes + 4u * (wn * 64 + 16 * g4);  // + 4 * n0 of the tile, + 16 * jb

    // ---- prologue: K-step 0 -> buffer 0 (all 4 units), the first two units of K-step 1 -> buffer 1
    __amdgpu_buffer_rsrc_t rx_c = make_x(m0), rw_c = make_w(n0), rx_n = rx_c, rw_n = rw_c;
    stage(0, 0, rx_c, rw_c, 0);
    stage(0, 1, rx_c, rw_c, 0);
    stage(0, 2, rx_c, rw_c, 0);
    stage(0, 3, rx_c, rw_c, 0);
    stage(1, 0, rx_c, rw_c, BK);
    stage(1, 1, rx_c, rw_c, BK);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // K-step 0 has landed (this wave's rows)
    __builtin_amdgcn_s_barrier();
    SP_SB();
    {   // fragments of u0 of the first K-step
        const unsigned w0 = w_addr[0], x0 = x_addr[0];
        SP_DS128(WA[0], w0, 0); SP_DS128(WA[1], w0, 512); SP_DS128(WA[2], w0, 1024); SP_DS128(WA[3], w0, 1536);
        SP_DS128(XA[0], x0, 0); SP_DS128(XA[1], x0, 2048); SP_DS128(XA[2], x0, 4096); SP_DS128(XA[3], x0, 6144);
    }
    SP_SB();
    f32x4 bq[4];  // bias of this lane's 16 output columns, current tile
    int ldy_t = ldy;
    int buf = 0;
    SP_STAMP(ts1);
    if (STAMP) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1)::"memory");
    for (; tile_id < ntiles; tile_id += G) {
        tile_coords(tile_id, m0, n0);
        rx_c = make_x(m0);
        rw_c = make_w(n0);
        if (tile_id + G < ntiles) {  // descriptors of this workgroup's next tile (else: harmless re-staging of this one)
            long nm0;
            int nn0;
            tile_coords(tile_id + G, nm0, nn0);
            rx_n = make_x(nm0);
            rw_n = make_w(nn0);
        } else {
            rx_n = rx_c;
            rw_n = rw_c;
        }
        {
            const unsigned ba = bias_addr + 4u * n0;
            SP_DS128(bq[0], ba, 0);
            SP_DS128(bq[1], ba, 16);
            SP_DS128(bq[2], ba, 32);
            SP_DS128(bq[3], ba, 48);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3])::"memory");
        }
        for (int kt = 0; kt < ksteps; ++kt) {
            // staging targets: u0 issues units 2, 3 of K-step kt+1, u3 units 0, 1 of K-step kt+2 (past the end of this tile:
            // the first K-steps of the workgroup's next tile)
            const bool nB = kt + 1 >= ksteps, nA = kt + 2 >= ksteps;
            const int kB = nB ? 0 : (kt + 1) * BK;
            const int kA = nA ? (kt + 2 - ksteps) * BK : (kt + 2) * BK;
            const __amdgpu_buffer_rsrc_t rwB = nB ? rw_n : rw_c;
            const __amdgpu_buffer_rsrc_t rxA = nA ? rx_n : rx_c;
            const unsigned boff = buf ? (unsigned)kTileBytes : 0u, noff = buf ? 0u : (unsigned)kTileBytes;
            const int nbuf = buf ^ 1;
            const unsigned x0 = x_addr[0] + boff, x1 = x_addr[1] + boff, w1 = w_addr[1] + boff;
            const unsigned x0n = x_addr[0] + noff, w0n = w_addr[0] + noff;
            char *const base_n = smem + nbuf * kTileBytes, *const base_c = smem + buf * kTileBytes;
            SP_SB();
            // ---- u0: (s0, blocks 0-3) on XA x WA; reads XB <- X(s0, 4-7), WB <- W(s1); stages units 2, 3 of K-step kt+1
            SP_WAIT8(0, XA, WA);
            SP_SB();
#define SP_DMA_W(Q, U) do { if (!(ABL & 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rwB, (lds_ptr_t)(base_n + dst_off[U][Q]), 16, src_off[U][Q], 2 * kB, 0, 0); } while (0)
#define SP_DMA_X(Q, U) do { if (!(ABL & 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rxA, (lds_ptr_t)(base_c + dst_off[U][Q]), 16, src_off[U][Q], 2 * kA, 0, 0); } while (0)
            if (kt == 0) {  // first K-step of a tile: the accumulators start from the bias (C operand), nothing to zero
                SP_UNIT(SP_MFMA_B, 0, XA, WA,
                        SP_DS128(XB[0], x0, 8192), do { SP_DS128(XB[1], x0, 10240); SP_DMA_W(0, 2); } while (0),
                        SP_DS128(XB[2], x0, 12288), do { SP_DS128(XB[3], x0, 14336); SP_DMA_W(1, 2); } while (0),
                        SP_DS128(WB[0], w1, 0), do { SP_DS128(WB[1], w1, 512); SP_DMA_W(0, 3); } while (0),
                        SP_DS128(WB[2], w1, 1024), do { SP_DS128(WB[3], w1, 1536); SP_DMA_W(1, 3); } while (0));
                SP_WAIT4(4, XB);
                SP_SB();
                SP_UNIT(SP_MFMA_B, 4, XB, WA,
                        SP_DS128(XA[0], x1, 0), SP_NOP, SP_DS128(XA[1], x1, 2048), SP_NOP,
                        SP_DS128(XA[2], x1, 4096), SP_NOP, SP_DS128(XA[3], x1, 6144), SP_NOP);
            } else {
                SP_UNIT(SP_MFMA, 0, XA, WA,
                        SP_DS128(XB[0], x0, 8192), do { SP_DS128(XB[1], x0, 10240); SP_DMA_W(0, 2); } while (0),
                        SP_DS128(XB[2], x0, 12288), do { SP_DS128(XB[3], x0, 14336); SP_DMA_W(1, 2); } while (0),
                        SP_DS128(WB[0], w1, 0), do { SP_DS128(WB[1], w1, 512); SP_DMA_W(0, 3); } while (0),
                        SP_DS128(WB[2], w1, 1024), do { SP_DS128(WB[3], w1, 1536); SP_DMA_W(1, 3); } while (0));
                // ---- u1: (s0, blocks 4-7) on XB x WA; reads XA <- X(s1, 0-3)
                SP_WAIT4(4, XB);  // the 4 W(s1) reads may still be in flight
                SP_SB();
                SP_UNIT(SP_MFMA, 4, XB, WA,
                        SP_DS128(XA[0], x1, 0), SP_NOP, SP_DS128(XA[1], x1, 2048), SP_NOP,
                        SP_DS128(XA[2], x1, 4096), SP_NOP, SP_DS128(XA[3], x1, 6144), SP_NOP);
            }
            // ---- u2: (s1, blocks 0-3) on XA x WB; reads XB <- X(s1, 4-7)
            SP_WAIT8(0, XA, WB);
            SP_SB();
            SP_UNIT(SP_MFMA, 0, XA, WB,
                    SP_DS128(XB[0], x1, 8192), SP_NOP, SP_DS128(XB[1], x1, 10240), SP_NOP,
                    SP_DS128(XB[2], x1, 12288), SP_NOP, SP_DS128(XB[3], x1, 14336), SP_NOP);
            // every read of the current buffer by this wave is done, its pieces of K-step kt+1 have landed
            if (ABL & 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(XB[0]), "+v"(XB[1]), "+v"(XB[2]), "+v"(XB[3])::"memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" : "+v"(XB[0]), "+v"(XB[1]), "+v"(XB[2]), "+v"(XB[3])::"memory");
            SP_SB();
            // ---- u3: (s1, blocks 4-7) on XB x WB; reads the next K-step's u0 fragments from the other buffer; stages
            // units 0, 1 of K-step kt+2 into the buffer just freed
            SP_UNIT(SP_MFMA, 4, XB, WB,
                    SP_DS128(WA[0], w0n, 0), do { SP_DS128(WA[1], w0n, 512); SP_DMA_X(0, 0); } while (0),
                    SP_DS128(WA[2], w0n, 1024), do { SP_DS128(WA[3], w0n, 1536); SP_DMA_X(1, 0); } while (0),
                    SP_DS128(XA[0], x0n, 0), do { SP_DS128(XA[1], x0n, 2048); SP_DMA_X(0, 1); } while (0),
                    SP_DS128(XA[2], x0n, 4096), do { SP_DS128(XA[3], x0n, 6144); SP_DMA_X(1, 1); } while (0));
#undef SP_DMA_W
#undef SP_DMA_X
            buf ^= 1;
        }
        // ---- epilogue: activation (the bias came in through the C operand of the first K-step), bf16 pack, two 16-byte
        // stores per row: whole 128-byte lines per store instruction, no LDS transposition
        {
            asm volatile("" : "+s"(ldy_t));
            __bf16 *Yp = Y + (m0 + wm * 128 + l15) * ldy_t + n0 + wn * 64 + 16 * g4;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int jh = 0; jh < 2; ++jh) {
                    const f32x4 v_ = acc[i][2 * jh], w_ = acc[i][2 * jh + 1];
                    bf16x8 o_;
                    o_[0] = (__bf16)act_fn<ACT>(v_[0]);
                    o_[1] = (__bf16)act_fn<ACT>(v_[1]);
                    o_[2] = (__bf16)act_fn<ACT>(v_[2]);
                    o_[3] = (__bf16)act_fn<ACT>(v_[3]);
                    o_[4] = (__bf16)act_fn<ACT>(w_[0]);
                    o_[5] = (__bf16)act_fn<ACT>(w_[1]);
                    o_[6] = (__bf16)act_fn<ACT>(w_[2]);
                    o_[7] = (__bf16)act_fn<ACT>(w_[3]);
                    *reinterpret_cast<bf16x8 *>(Yp + (long)(i * 16) * ldy_t + 8 * jh) = o_;
                }
                SP_SB();
            }
        }
    }
    SP_STAMP(ts2);
    if (STAMP) {
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt2)::"memory");
        if (tid == 0 && blockIdx.x < 256) {
            using namespace persist;
            M360_STAMP_STORE(0, ts1 - ts0);
            M360_STAMP_STORE(1, ts2 - ts1);  // all tiles of this workgroup: main loops + epilogues
            M360_STAMP_STORE(5, rt2 - rt1);  // in-kernel clock = [1] / [5] x 100 MHz
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of this wave's LDS-DMA may land after the workgroup is gone
#undef SP_STAMP
#undef SP_DS128
#undef SP_SB
#undef SP_WAIT8
#undef SP_WAIT4
#undef SP_MFMA
#undef SP_MFMA_B
#undef SP_UNIT
#undef SP_NOP
}

}  // namespace sp16
}  // namespace m360
