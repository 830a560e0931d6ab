// bf16 linear kernel, fourth structure: 8 waves (two per SIMD), software-pipelined, FOUR-STAGE LDS RING of 32-deep K slabs.
//
// What the measurements of the earlier structures said (profiles/r02, same 524288 x 1024 x 1024 layer, random data):
//   bare v_mfma_f32_16x16x32_bf16 loop                         2.01 PFLOP/s at 2.11 GHz   (what the pipes can do on this device)
//   ping-pong kernel (pp16) / software-pipelined, 2 stages     1.12 / 1.09 PFLOP/s       (55-58 k cycles per tile, 32.8 k of MFMA)
//   software-pipelined without waiting for the LDS-DMA         1.17   (+7 %: latency is NOT what costs)
//   software-pipelined without issuing the LDS-DMA             1.45   (+34 %: the L2 -> LDS stream itself costs)
//   ... without LDS-DMA and without ds_reads                   1.82
// A 256 x 256 tile moves 64 KB of operands per 64-deep K-step from L2 into LDS: 8.6 GB per launch, i.e. 30 B/clk/CU at
// full MFMA rate - about what the L2 -> LDS path delivers when it is kept full (MI355X_MICROARCH.md: 66-73 GB/s per CU).
// With two 64 KB LDS buffers the loads of a K-step can only be issued inside ONE K-step window and in bursts, so the path
// idles half of the time.  Here a stage is a 32-deep slab (X 256 rows x 64 B + W 256 rows x 64 B = 32 KB) and the ring
// holds four of them: the pieces of stage q+4 are issued the moment stage q's buffer is free, three stages (>= 3000 MFMA
// cycles) before they are needed, with counted vmcnt (8 younger pieces may stay in flight at every barrier) - the
// memory path sees a continuous stream of up to 96 KB in flight per CU.
//
// Everything else as in the other bf16 kernels: 256 x 256 tile, waves 2 (M) x 4 (N), wave tile 128 x 64 = 8 x 4 blocks of
// 16x16x32, operands swapped / weight rows permuted so that a lane owns 16 consecutive output columns, bias through the C
// operand of a tile's first MFMAs, persistent workgroups with the ring running across tile boundaries.
// LDS rows are 64 B (4 chunks of 16 B); chunk c of row r lands in slot c ^ 2((r>>3)&1) (activations) resp.
// c ^ 2((r>>5)&1) (weights): with the ds_read_b128 lane groups of gfx950 ({0-3,12-15,20-27}, ...) every group then covers
// 16 distinct 16-byte bank slots (brute-forced).
#pragma once
#include "../m360_common.hip.h"
#include "../m360_linear_persist.hip.h"  // diagnostic stamp buffer

namespace m360 {
namespace rg16 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;

constexpr int BM = 256, BN = 256, BKS = 32;  // BKS: K depth of one ring stage
constexpr int kThreads = 512;
constexpr int kStages = 4;
constexpr int kWOff = 256 * 64;              // byte offset of the weight rows inside a stage
constexpr int kStageBytes = 2 * kWOff;       // 32 KiB
constexpr int kMaxBias = 4096;               // widest layer this kernel takes (bias is served from LDS)

template <int ACT>
__device__ __forceinline__ float act_fn(float v) {
    if (ACT == M360_ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == M360_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.0f + __expf(-v));
    return v;
}

// ABL (diagnostics build only, results are WRONG when != 0): 1 = never wait for the LDS-DMA, 2 = issue no LDS-DMA in the
// main loop, 4 = issue no ds_reads in the main loop
template <int ACT, bool STAMP = false, int ABL = 0>
__global__ __launch_bounds__(kThreads, 1) void linear_bf16_rg_kernel(
    const __bf16 *__restrict__ X, long M, int ldx, const __bf16 *__restrict__ W, const float *__restrict__ bias,
    int Np, int Kp, __bf16 *__restrict__ Y, int ldy, int tiles_n, int ntiles, int ldw) {
    __shared__ __attribute__((aligned(1024))) char smem[kStages * kStageBytes + kMaxBias * 4];  // 128 KiB ring + the bias vector

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int kstages = Kp / BKS;  // even, >= 4 (k_pad is a multiple of 64, >= 128)
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, rt1 = 0, rt2 = 0;
#define RG_STAMP(var)                                                                           \
    do {                                                                                        \
        if (STAMP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");  \
    } while (0)
    RG_STAMP(ts0);

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

    // ---- staging: a stage is 32 pieces of 16 rows x 64 B (1 KiB = one LDS-DMA instruction); wave w stages activation rows
    // [32w, 32w+32) and weight rows [32w, 32w+32) as two pieces each.  Lane L of a piece: row L>>2, LDS slot L&3 holding
    // global chunk slot ^ f(row).  Per-lane byte offsets inside a tile are tile-independent.
    unsigned src_x[2], src_w[2];
    int dst_x[2], dst_w[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int row0 = 32 * wave + 16 * p;
        const int r = row0 + (lane >> 2), slot = lane & 3;
        src_x[p] = (unsigned)(r * ldx + 8 * (slot ^ (2 * ((r >> 3) & 1)))) * 2u;
        src_w[p] = (unsigned)(r * ldw + 8 * (slot ^ (2 * ((r >> 5) & 1)))) * 2u;
        dst_x[p] = row0 * 64;
        dst_w[p] = kWOff + row0 * 64;
    }
    auto make_x = [&](long tm0) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(X + tm0 * ldx), 0, 0x7fffffff, 0x00020000);
    };
    auto make_w = [&](int tn0) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(W + (long)tn0 * ldw), 0, 0x7fffffff, 0x00020000);
    };

    // ---- fragment addresses (stage 0 of the ring; + stage * kStageBytes): lane (row l15 of a 16-row block, k-chunk g4)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const unsigned x_addr = lds0 + (wm * 128 + l15) * 64 + 16 * (g4 ^ (2 * ((l15 >> 3) & 1)));  // + block * 1024
    // weight row of MFMA row l15 = 4a + b in N-block jb: 16a + 4jb + b of the wave's 64; its swizzle 2(a>>1) = 2(l15>>3)
    const unsigned w_addr = lds0 + kWOff + (wn * 64 + 16 * (l15 >> 2) + (l15 & 3)) * 64 + 16 * (g4 ^ (2 * ((l15 >> 3) & 1)));  // + jb * 256

    f32x4 acc[8][4];
    bf16x8 XA[4], XB[4], WA[4], WB[4];

#define RG_DS128(dst, addr, imm)                                                                   \
    do {                                                                                           \
        if (ABL & 4) asm volatile("; no read %0 %1" : "=v"(dst) : "v"(addr));                      \
        else asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr));             \
    } while (0)
#define RG_SB() __builtin_amdgcn_sched_barrier(0)
#define RG_WAIT8(F, Gf)                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                   \
                 : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]), "+v"(F[3]), "+v"(Gf[0]), "+v"(Gf[1]), "+v"(Gf[2]), "+v"(Gf[3])::"memory")
#define RG_MFMA(I, J, XF, WF) acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[J], XF[(I) & 3], acc[I][J], 0, 0, 0)
#define RG_MFMA_B(I, J, XF, WF) acc[I][J] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(WF[J], XF[(I) & 3], bq[J], 0, 0, 0)
// One unit = 16 MFMAs (M blocks I0..I0+3 x N blocks 0..3); after every second MFMA one slot S0..S7 is issued in the shadow
#define RG_UNIT(MF, I0, XF, WF, S0, S1, S2, S3, S4, S5, S6, S7)   \
    do {                                                          \
        MF((I0) + 0, 0, XF, WF); MF((I0) + 0, 1, XF, WF); S0; RG_SB(); \
        MF((I0) + 0, 2, XF, WF); MF((I0) + 0, 3, XF, WF); S1; RG_SB(); \
        MF((I0) + 1, 0, XF, WF); MF((I0) + 1, 1, XF, WF); S2; RG_SB(); \
        MF((I0) + 1, 2, XF, WF); MF((I0) + 1, 3, XF, WF); S3; RG_SB(); \
        MF((I0) + 2, 0, XF, WF); MF((I0) + 2, 1, XF, WF); S4; RG_SB(); \
        MF((I0) + 2, 2, XF, WF); MF((I0) + 2, 3, XF, WF); S5; RG_SB(); \
        MF((I0) + 3, 0, XF, WF); MF((I0) + 3, 1, XF, WF); S6; RG_SB(); \
        MF((I0) + 3, 2, XF, WF); MF((I0) + 3, 3, XF, WF); S7; RG_SB(); \
    } while (0)
#define RG_NOP ((void)0)
// a piece is issued by the waves whose group (wave & 3) equals GRP: at most two of the CU's eight waves hit the texture
// addresser in the same slot (it takes a 1-KiB piece per 16 cycles; eight at once made every issuing wave wait)
#define RG_DMA(GRP, RS, BASE, SRC, DST, K0)                                                                                     \
    do {                                                                                                                     \
        if (!(ABL & 2) && wgrp == (GRP)) __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, (lds_ptr_t)((BASE) + (DST)), 16, SRC, 2 * (K0), 0, 0); \
    } while (0)
// One ring stage.  WC: weight fragments of this stage (in registers), WN: receives the next stage's.
//   unit A: blocks 0-3 on XA x WC; reads XB <- blocks 4-7 of this stage; the two WEIGHT pieces of the stage whose
//           activation pieces the previous unit B issued (ring slot freed one barrier ago)
//   wait: own reads done, this wave's pieces of the NEXT stage landed (8 younger pieces may be in flight); barrier
//   unit B: blocks 4-7 on XB x WC; reads WN, XA <- next stage; the two ACTIVATION pieces of stage +4 into the slot just freed
// LDS-DMA slots are staggered over the wave groups (wave & 3): slot k of a unit belongs to group k & 3.
#define RG_STAGE(MF, WC, WN)                                                                                    \
    do {                                                                                                        \
        RG_WAIT8(XA, WC);                                                                                       \
        RG_SB();                                                                                                \
        RG_UNIT(MF, 0, XA, WC,                                                                                  \
                do { RG_DS128(XB[0], xc, 4096); RG_DMA(0, rw_p, ring_p, src_w[0], dst_w[0], k_p); } while (0),  \
                RG_DMA(1, rw_p, ring_p, src_w[0], dst_w[0], k_p),                                               \
                do { RG_DS128(XB[1], xc, 5120); RG_DMA(2, rw_p, ring_p, src_w[0], dst_w[0], k_p); } while (0),  \
                RG_DMA(3, rw_p, ring_p, src_w[0], dst_w[0], k_p),                                               \
                do { RG_DS128(XB[2], xc, 6144); RG_DMA(0, rw_p, ring_p, src_w[1], dst_w[1], k_p); } while (0),  \
                RG_DMA(1, rw_p, ring_p, src_w[1], dst_w[1], k_p),                                               \
                do { RG_DS128(XB[3], xc, 7168); RG_DMA(2, rw_p, ring_p, src_w[1], dst_w[1], k_p); } while (0),  \
                RG_DMA(3, rw_p, ring_p, src_w[1], dst_w[1], k_p));                                              \
        if (ABL & 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+v"(XB[0]), "+v"(XB[1]), "+v"(XB[2]), "+v"(XB[3])::"memory"); \
        else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" : "+v"(XB[0]), "+v"(XB[1]), "+v"(XB[2]), "+v"(XB[3])::"memory"); \
        RG_SB();                                                                                                \
        RG_UNIT(MF, 4, XB, WC,                                                                                  \
                do { RG_DS128(WN[0], wnx, 0); RG_DMA(0, rx4, ring_free, src_x[0], dst_x[0], k4); } while (0),   \
                do { RG_DS128(WN[1], wnx, 256); RG_DMA(1, rx4, ring_free, src_x[0], dst_x[0], k4); } while (0), \
                do { RG_DS128(WN[2], wnx, 512); RG_DMA(2, rx4, ring_free, src_x[0], dst_x[0], k4); } while (0), \
                do { RG_DS128(WN[3], wnx, 768); RG_DMA(3, rx4, ring_free, src_x[0], dst_x[0], k4); } while (0), \
                do { RG_DS128(XA[0], xnx, 0); RG_DMA(0, rx4, ring_free, src_x[1], dst_x[1], k4); } while (0),   \
                do { RG_DS128(XA[1], xnx, 1024); RG_DMA(1, rx4, ring_free, src_x[1], dst_x[1], k4); } while (0), \
                do { RG_DS128(XA[2], xnx, 2048); RG_DMA(2, rx4, ring_free, src_x[1], dst_x[1], k4); } while (0), \
                do { RG_DS128(XA[3], xnx, 3072); RG_DMA(3, rx4, ring_free, src_x[1], dst_x[1], k4); } while (0)); \
        rw_p = rw4;                                                                                             \
        k_p = k4;                                                                                               \
        ring_p = ring_free;                                                                                     \
    } while (0)

    // ---- bias -> LDS once (before any LDS-DMA is in flight)
    float *const bias_lds = reinterpret_cast<float *>(smem + kStages * kStageBytes);
    for (int i = tid; i < Np; i += kThreads) bias_lds[i] = bias[i];
    __syncthreads();
    const unsigned bias_addr = lds0 + kStages * kStageBytes + 4u * (wn * 64 + 16 * g4);  // + 4 * n0 of the tile, + 16 * jb

    // ---- prologue: stages 0..2 of the first tile and the activation pieces of stage 3 (its weight pieces follow in unit
    // A of stage 0, like in the steady state)
    const int wgrp = wave & 3;
    __amdgpu_buffer_rsrc_t rx_c = make_x(m0), rw_c = make_w(n0), rx_n = rx_c, rw_n = rw_c;
#pragma unroll
    for (int q = 0; q < kStages; ++q) {
        char *ring_free = smem + q * kStageBytes;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_c, (lds_ptr_t)(ring_free + dst_x[0]), 16, src_x[0], 2 * q * BKS, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx_c, (lds_ptr_t)(ring_free + dst_x[1]), 16, src_x[1], 2 * q * BKS, 0, 0);
        if (q < kStages - 1) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_c, (lds_ptr_t)(ring_free + dst_w[0]), 16, src_w[0], 2 * q * BKS, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw_c, (lds_ptr_t)(ring_free + dst_w[1]), 16, src_w[1], 2 * q * BKS, 0, 0);
        }
    }
    __amdgpu_buffer_rsrc_t rw_p = rw_c;  // target of the weight pieces the next unit A issues: stage 3 of the first tile
    int k_p = 3 * BKS;
    char *ring_p = smem + 3 * kStageBytes;
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // stage 0 has landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    RG_SB();
    RG_DS128(WA[0], w_addr, 0); RG_DS128(WA[1], w_addr, 256); RG_DS128(WA[2], w_addr, 512); RG_DS128(WA[3], w_addr, 768);
    RG_DS128(XA[0], x_addr, 0); RG_DS128(XA[1], x_addr, 1024); RG_DS128(XA[2], x_addr, 2048); RG_DS128(XA[3], x_addr, 3072);
    RG_SB();
    f32x4 bq[4];  // bias of this lane's 16 output columns, current tile
    int ldy_t = ldy;
    int ring = 0;  // ring slot of the current stage
    RG_STAMP(ts1);
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
            RG_DS128(bq[0], ba, 0);
            RG_DS128(bq[1], ba, 16);
            RG_DS128(bq[2], ba, 32);
            RG_DS128(bq[3], ba, 48);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3])::"memory");
        }
        for (int ks = 0; ks < kstages; ks += 2) {
#define RG_SETUP(KS)                                                                                           \
            const bool nx_ = (KS) + 4 >= kstages;                                                              \
            const int k4 = (nx_ ? (KS) + 4 - kstages : (KS) + 4) * BKS;                                        \
            const __amdgpu_buffer_rsrc_t rx4 = nx_ ? rx_n : rx_c, rw4 = nx_ ? rw_n : rw_c;                     \
            char *const ring_free = smem + ring * kStageBytes;                                                 \
            const unsigned xc = x_addr + ring * kStageBytes;                                                   \
            const unsigned nring_ = (ring + 1) & (kStages - 1);                                                \
            const unsigned xnx = x_addr + nring_ * kStageBytes, wnx = w_addr + nring_ * kStageBytes
            {
                RG_SETUP(ks);
                RG_SB();
                if (ks == 0) RG_STAGE(RG_MFMA_B, WA, WB);  // first stage of a tile: accumulators start from the bias
                else RG_STAGE(RG_MFMA, WA, WB);
                ring = nring_;
            }
            {
                RG_SETUP(ks + 1);
                RG_SB();
                RG_STAGE(RG_MFMA, WB, WA);
                ring = nring_;
            }
#undef RG_SETUP
        }
        // ---- epilogue: activation (the bias came in through the C operand), bf16 pack, two 16-byte stores per row
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
                RG_SB();
            }
        }
    }
    RG_STAMP(ts2);
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
#undef RG_STAMP
#undef RG_DS128
#undef RG_SB
#undef RG_WAIT8
#undef RG_MFMA
#undef RG_MFMA_B
#undef RG_UNIT
#undef RG_NOP
#undef RG_DMA
#undef RG_STAGE
}

}  // namespace rg16
}  // namespace m360
