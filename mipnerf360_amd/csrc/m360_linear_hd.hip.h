// Third-generation fp32-MFMA linear kernel: HALF tile (128 x 256), DOUBLE accumulator set, epilogue in MFMA shadows.
//
// The persistent kernel (m360_linear_persist.hip.h) spends its whole accumulator file (256 registers per lane) on ONE
// 256 x 256 tile, so a tile's epilogue (LDS transposition + bias + activation + stores: 10.3 k cycles) cannot overlap
// matrix work: 1.9 % of a 1024-deep tile, 7 % of a 256-deep one (proposal layers: 0.81-0.84 of the roofline) and 24 % of a
// 64-deep one (first layers: 0.64).  Here a workgroup's tile is 128 x 256 (wave tile 64 x 128 = 2 x 4 MFMA blocks = 128
// accumulators) and the file holds TWO sets: while tile t accumulates into one set, the 8 blocks of tile t-1 in the
// other set are staged through a wave-private LDS area, activated and stored - one block per K-group during the first
// two K-steps of tile t, every instruction in its own MFMA gap (the schedule is GENERATED: tools/gen_hd_kstep.py ->
// m360_linear_hd_gen.inc).  Everything else follows the persistent kernel: one wave per SIMD, persistent workgroups
// with XCD-aware tile order, K-step 32 staged by LDS-DMA (buffer descriptors, scalar K offset, source-side XOR swizzle,
// double-buffered), operand fragments by ds_read_b128 feeding four MFMAs each, hand-counted waits tied to the fragment
// registers, one barrier per K-step placed before the last K-group.  The k order of every output element is the
// persistent kernel's, so results are bit-identical to the other two fp32 kernels.
// Bias + {none, ReLU} epilogues only (the sigmoid / fused-heads / ReLU-mask layers stay with the persistent kernel).
// Only the matrix pipe, the LDS and the scalar unit work inside the K loop: with an even number of K-steps (EVENK) the LDS
// stage of every read is an instruction immediate; the epilogue's arithmetic (6 vector instructions per 16 bytes, which no
// schedule hides behind the same wave's MFMAs) is all the vector work there is.  With a queue word the last tiles of a launch
// are handed out by ticket so that XCDs with different clocks finish together (m360_linear_balanced).  DESIGN.md 4.1b.
#pragma once
#include "m360_common.hip.h"

// Non-temporal epilogue stores: the output lines of a tile are not read again before the next launch and otherwise displace the
// operands from L2.  Measured (profiles/r03/hd_nontemporal_stores_ab.jsonl, hd_nontemporal_stores_step_ab.txt): the 64-deep first
// NeRF layer 0.666 -> 0.61-0.62 ms, 256^2 and 1024^2 layers unchanged, step 53.21-53.25 -> 53.11-53.20 ms alternating on one box.
#ifndef M360_HD_NT_STORES
#define M360_HD_NT_STORES 1
#endif

namespace m360 {
namespace hd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void *lds_ptr_t;

constexpr int BM = 128, BN = 256, BK = 32;
constexpr int kThreads = 256;
constexpr int kAFloats = BM * BK;                    // 16 KiB
constexpr int kBFloats = BN * BK;                    // 32 KiB
constexpr int kStageFloats = kAFloats + kBFloats;    // one K-step: 48 KiB
constexpr int kStgFloats = 32 * 36;                  // per-wave epilogue staging (32 x 32 block, rows padded to 36)
constexpr int kMaxBias = 4096;                       // widest layer (bias is served from LDS)

#ifdef M360_DIAG
// diagnostics build, per workgroup: [0] shader-clock cycles (s_memtime) and [1] 100 MHz ticks (s_memrealtime) of the whole tile
// loop, [2] K-steps executed (cycles per K-step against the 128 x 64 = 8192 the matrix pipe needs, and the clock held);
// second half: cycles summed over the first / second / further K-steps of every tile
__device__ unsigned long long g_hd_stamps[2 * 256 * 4];
#endif
// ABL (diagnostic builds only; results are wrong unless 0): 1 = no workgroup barrier, 2 = no LDS-DMA, 4 = no operand reads,
// 8 = epilogue fillers unguarded, 16 = no epilogue stores, 32 = no epilogue arithmetic, 64 = tiles in natural order (no XCD-aware remap; results stay right), 128 = non-temporal stores (results stay right)
// EVENK: Kp / 32 is even -> static LDS stages (see the generated K-steps)
template <int ACT, bool EVENK = false, int ABL = 0, bool STAMP = false>
__global__ __launch_bounds__(kThreads, 1) void linear_f32_hd_kernel(
    const float *__restrict__ X, long M, int ldx, const float *__restrict__ W, const float *__restrict__ bias, int Np,
    int Kp, float *__restrict__ Y, int ldy, int tiles_n, int ntiles, unsigned *__restrict__ queue, int n_static) {
    __shared__ __attribute__((aligned(1024))) float smem[2 * kStageFloats + 4 * kStgFloats + kMaxBias + 64];  // 130 KiB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int G = gridDim.x;
    const int ksteps = Kp / BK;  // >= 2

    auto tile_coords = [&](int lin_id, long &m0, int &n0) __attribute__((always_inline)) {
        const int full = (ntiles / 8) * 8;  // XCD-aware (speed only): ids sharing id % 8 cover a contiguous range of tiles
        int lin = lin_id;
        if (lin_id < full && !(ABL & 64)) lin = (lin_id % 8) * (full / 8) + lin_id / 8;
        m0 = (long)(lin / tiles_n) * BM;
        n0 = (lin % tiles_n) * BN;
    };

    // ---- LDS-DMA staging: wave w fills A rows [32w, 32w+32) (4 pieces) and B rows [64w, 64w+64) (8 pieces) of a K-step
    unsigned a_voff[4], b_voff[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 32 * wave + 8 * q + (lane >> 3);
        a_voff[q] = (unsigned)(r * ldx + 4 * ((lane & 7) ^ ((r >> 1) & 7))) * 4u;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int r = 64 * wave + 8 * q + (lane >> 3);
        b_voff[q] = (unsigned)(r * Kp + 4 * ((lane & 7) ^ ((r >> 1) & 7))) * 4u;
    }
    __amdgpu_buffer_rsrc_t rsrc_a, rsrc_b;
    auto set_load_tile = [&](long m0, int n0) __attribute__((always_inline)) {
        rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X + m0 * ldx), 0, 0x7fffffff, 0x00020000);
        rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(W + (long)n0 * Kp), 0, 0x7fffffff, 0x00020000);
    };
    float *const dma_a = smem + wave * 32 * BK;             // + buf * kStageFloats + q * 8 * BK
    float *const dma_b = smem + kAFloats + wave * 64 * BK;

    // ---- operand reads: lane (l31, h), K-group g reads chunk (2g + h) of its rows = slot (2g + h) ^ f
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float *)smem;
    const int fsw = (l31 >> 1) & 7;
    unsigned a_addr[4], b_addr[4];  // byte addresses in buffer 0
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int slot = ((2 * g + h) ^ fsw) * 4;
        a_addr[g] = lds0 + 4u * ((wm * 64 + l31) * BK + slot);
        b_addr[g] = lds0 + 4u * (kAFloats + (wn * 128 + l31) * BK + slot);
    }
    // ---- epilogue staging (wave-private) and bias
    const unsigned stg0 = lds0 + 4u * (2 * kStageFloats + wave * kStgFloats);
    const unsigned stw = stg0 + 4u * (4 * h * 36 + l31);                       // + ((r&3) + 8(r>>2)) * 144 per register r
    const int rrow = lane >> 3, rcol = 4 * (lane & 7);
    const unsigned str = stg0 + 4u * (rrow * 36 + rcol);                       // + p * 8 * 144
    float *const bias_lds = smem + 2 * kStageFloats + 4 * kStgFloats;
    for (int c = tid; c < Np; c += kThreads) bias_lds[c] = bias[c];
    const unsigned bias_addr = lds0 + 4u * (2 * kStageFloats + 4 * kStgFloats + wn * 128 + rcol);  // + 4 * n0 + 128 * j

    f32x16 acc[2][2][4];
    f32x4 fa0[2], fb0[4], fa1[2], fb1[4];
    f32x4 ev[4] = {};  // one staged 32 x 32 block as this lane reads it back: 4 row groups x 4 consecutive columns
    // A VALU instruction is never hidden behind this wave's own MFMAs on gfx950 (both go through the SIMD's vector issue port,
    // which an MFMA holds for its 64 cycles): n of them in one MFMA gap cost 4 n + 8 cycles of matrix time, one per gap 12
    // each (tools/mfma32_filler_cost.hip, profiles/r02/mfma32_filler_cost.jsonl).  So the epilogue keeps its arithmetic to
    // the minimum (bias: 2 packed adds, ReLU: 4 max per 1 KiB) and issues a block's 24 instructions in ONE gap; everything
    // else is LDS / scalar / VMEM work, which is hidden: ds_write_b32 straight from the accumulator registers, stores with a
    // scalar row base + one lane-offset register + immediate column offset (no address arithmetic).
    f32x4 bq[4];            // bias of this lane's 4 columns in each of the 4 column blocks, tile being stored
    const float *Yt = Y;    // wave-uniform: first element of this wave's 64 x 128 corner of the tile being stored
    const unsigned y_voff = (unsigned)(rrow * ldy + rcol) * 4u;  // this lane's 16 bytes inside an 8-row slab of that corner
    const f32x16 kZero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

#define HD_DS128(dst, addr, imm)                                                            \
    do {                                                                                    \
        if (!(ABL & 4)) asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr)); \
        else asm volatile("" : "=v"(dst) : "v"(addr));                                      \
    } while (0)
#define HD_SB() __builtin_amdgcn_sched_barrier(0)
#define HD_WAIT_FRAG(FA, FB) \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(FA[0]), "+v"(FA[1]), "+v"(FB[0]), "+v"(FB[1]), "+v"(FB[2]), "+v"(FB[3])::"memory")
#define HD_BARRIER(VM, FA, FB)                                                                                        \
    do {                                                                                                              \
        if (!(ABL & 1))                                                                                               \
            asm volatile("s_waitcnt vmcnt(" #VM ") lgkmcnt(0)\n\ts_barrier"                                            \
                         : "+v"(FA[0]), "+v"(FA[1]), "+v"(FB[0]), "+v"(FB[1]), "+v"(FB[2]), "+v"(FB[3])::"memory");   \
        else                                                                                                          \
            asm volatile("s_waitcnt vmcnt(" #VM ") lgkmcnt(0)"                                                        \
                         : "+v"(FA[0]), "+v"(FA[1]), "+v"(FB[0]), "+v"(FB[1]), "+v"(FB[2]), "+v"(FB[3])::"memory");   \
    } while (0)
#define HD_DMA_A(Q) if (!(ABL & 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (lds_ptr_t)(dma_a + nbuf * kStageFloats + (Q) * 8 * BK), 16, a_voff[Q], 4 * next_k0, 0, 0)
#define HD_DMA_B(Q) if (!(ABL & 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (lds_ptr_t)(dma_b + nbuf * kStageFloats + (Q) * 8 * BK), 16, b_voff[Q], 4 * next_k0, 0, 0)
// epilogue pieces: register r of block (I, J) of accumulator set P -> staging row (r&3) + 8(r>>2) + 4h, column l31
#define HD_EW(P, I, J, R) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(stw), "a"(acc[P][I][J][R]), "n"((((R) & 3) + 8 * ((R) >> 2)) * 144) : "memory")
#define HD_ER(PP) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ev[PP]) : "v"(str), "n"((PP) * 8 * 144) : "memory")
#define HD_EWAIT() asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ev[0]), "+v"(ev[1]), "+v"(ev[2]), "+v"(ev[3])::"memory")
// bias + activation of the staged block (column block J) in registers: the block's whole vector arithmetic, one MFMA gap
#define HD_EV(J, PA, PB)                                                                                     \
    do {                                                                                                     \
        if (!(ABL & 32)) {                                                                                   \
            _Pragma("unroll") for (int pp_ = PA; pp_ < PB; ++pp_) {                                          \
                /* bias: two packed adds, ReLU: four v_max_i32 (NaN-preserving ReLU, relu_nanf_ of m360_common) - spelled out: left to itself hipcc emits four v_add_f32 in some \
                   instantiations, and a canonicalising v_max in front of every fmaxf on an asm result */   \
                f32x2 lo_ = __builtin_shufflevector(ev[pp_], ev[pp_], 0, 1), hi_ = __builtin_shufflevector(ev[pp_], ev[pp_], 2, 3); \
                const f32x2 blo_ = __builtin_shufflevector(bq[J], bq[J], 0, 1), bhi_ = __builtin_shufflevector(bq[J], bq[J], 2, 3); \
                asm("v_pk_add_f32 %0, %1, %2" : "=v"(lo_) : "v"(lo_), "v"(blo_));                            \
                asm("v_pk_add_f32 %0, %1, %2" : "=v"(hi_) : "v"(hi_), "v"(bhi_));                            \
                float e0_ = lo_[0], e1_ = lo_[1], e2_ = hi_[0], e3_ = hi_[1];                                \
                if (ACT == M360_ACT_RELU) {                                                                  \
                    asm("v_max_i32 %0, 0, %1" : "=v"(e0_) : "v"(e0_));                                       \
                    asm("v_max_i32 %0, 0, %1" : "=v"(e1_) : "v"(e1_));                                       \
                    asm("v_max_i32 %0, 0, %1" : "=v"(e2_) : "v"(e2_));                                       \
                    asm("v_max_i32 %0, 0, %1" : "=v"(e3_) : "v"(e3_));                                       \
                }                                                                                            \
                ev[pp_] = (f32x4){e0_, e1_, e2_, e3_};                                                       \
            }                                                                                                \
            /* pins the instructions here (machine sinking would move them next to the stores) */           \
            asm volatile("" : "+v"(ev[0]), "+v"(ev[1]), "+v"(ev[2]), "+v"(ev[3]));                          \
        }                                                                                                    \
    } while (0)
#define HD_ES(I, J, PP)                                                                                      \
    do {                                                                                                     \
        if (!(ABL & 16) && !(ABL & 128) && !M360_HD_NT_STORES)                                               \
            asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3" ::"v"(y_voff), "v"(ev[PP]),            \
                         "s"(Yt + (long)((I) * 32 + (PP) * 8) * ldy), "n"((J) * 128) : "memory");            \
        else if (!(ABL & 16)) /* 128 / M360_HD_NT_STORES: non-temporal stores (results stay right) */        \
            asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3 nt" ::"v"(y_voff), "v"(ev[PP]),         \
                         "s"(Yt + (long)((I) * 32 + (PP) * 8) * ldy), "n"((J) * 128) : "memory");            \
    } while (0)
// an epilogue filler of the generated schedule: skipped (wave-uniform branch) while there is no previous tile
#define HD_G(X)                                   \
    do {                                          \
        if (ABL & 8) { X; }                       \
        else if (have_prev) { X; }                \
    } while (0)
// barrier of an epilogue K-step: the 12 stores younger than this K-step's LDS-DMA pieces may stay in flight.  The counted
// wait is a bare instruction under the branch (no register ties: two tied variants would meet in a join and cost copies)
#define HD_BARRIER_E(FA, FB)                                                         \
    do {                                                                             \
        if ((ABL & 8) || have_prev) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        \
        HD_BARRIER(63, FA, FB);                                                      \
    } while (0)
#include "m360_linear_hd_gen.inc"

    // Tile ids of a workgroup: b, b + G, ... (n_static of them), then - when the caller supplied a zeroed queue word - ids
    // n_static * G + (ticket from the queue): the XCDs of one chip hold clocks 1-2 % apart (profiles/r02/hd_xcd_spread.jsonl),
    // with equal shares the launch ends with the slowest one.  The static part keeps the XCD-aware order (worth 0.8 %), the
    // last ~6 % of the tiles go to whoever is free.  `cur` runs, `nxt` is prefetched by cur's last K-step, the id after that
    // is fetched (thread 0, returning atomic) at the start of cur and handed to the other waves through LDS.
    int lin_id = blockIdx.x, nxt_id = blockIdx.x + G, tcount = 0;
    unsigned ticket = 0;
    int *const id_slot = reinterpret_cast<int *>(smem + 2 * kStageFloats + 4 * kStgFloats + kMaxBias);
    if (lin_id >= ntiles) return;
    long m0;
    int n0;
    tile_coords(lin_id, m0, n0);
    set_load_tile(m0, n0);
    {
        const int nbuf = 0, next_k0 = 0;
        HD_DMA_A(0); HD_DMA_A(1); HD_DMA_A(2); HD_DMA_A(3);
        HD_DMA_B(0); HD_DMA_B(1); HD_DMA_B(2); HD_DMA_B(3); HD_DMA_B(4); HD_DMA_B(5); HD_DMA_B(6); HD_DMA_B(7);
    }
    int buf = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // first K-step of the first tile has landed; bias_lds written
    __syncthreads();
    HD_SB();
    HD_DS128(fa0[0], a_addr[0], 0); HD_DS128(fa0[1], a_addr[0], 4096);
    HD_DS128(fb0[0], b_addr[0], 0); HD_DS128(fb0[1], b_addr[0], 4096); HD_DS128(fb0[2], b_addr[0], 8192); HD_DS128(fb0[3], b_addr[0], 12288);
    HD_SB();

    bool have_prev = false;
    unsigned long long mt0 = 0, rt0 = 0, mt1 = 0, rt1 = 0, nk = 0;
    unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, tF = 0, tS = 0, tP = 0;  // cycles in the first / second / further K-steps
    (void)nk; (void)c0; (void)c1; (void)c2; (void)c3; (void)tF; (void)tS; (void)tP; (void)mt1; (void)rt1;
    if (STAMP) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(mt0), "=s"(rt0)::"memory");
    // one tile into accumulator set S while (have_prev) the previous tile's set P is stored
#define HD_KSETUP(KT)                                                                                              \
    int next_k0 = ((KT) + 1) * BK;                                                                                \
    if ((KT) + 1 == ksteps) { /* stage K-step 0 of this workgroup's next tile (else, harmlessly, of this one) */   \
        next_k0 = 0;                                                                                              \
        if (nxt_id < ntiles) {                                                                                    \
            long nm0;                                                                                             \
            int nn0;                                                                                              \
            tile_coords(nxt_id, nm0, nn0);                                                                        \
            set_load_tile(nm0, nn0);                                                                              \
        }                                                                                                         \
    }                                                                                                             \
    const int nbuf = buf ^ 1;                                                                                     \
    const unsigned boff = buf ? 4u * kStageFloats : 0u, noff = buf ? 0u : 4u * kStageFloats;                      \
    const unsigned a1 = a_addr[1] + boff, b1 = b_addr[1] + boff, a2 = a_addr[2] + boff, b2 = b_addr[2] + boff;    \
    const unsigned a3 = a_addr[3] + boff, b3 = b_addr[3] + boff, a0n = a_addr[0] + noff, b0n = b_addr[0] + noff;  \
    HD_SB()
// static stages: K-step KT works on stage BUF = KT & 1, nothing to compute but the scalar offset of the next K-step
#define HD_KSETUP_S(KT, BUF)                                                                                      \
    int next_k0 = ((KT) + 1) * BK;                                                                                \
    if ((KT) + 1 == ksteps) {                                                                                     \
        next_k0 = 0;                                                                                              \
        if (nxt_id < ntiles) {                                                                                    \
            long nm0;                                                                                             \
            int nn0;                                                                                              \
            tile_coords(nxt_id, nm0, nn0);                                                                        \
            set_load_tile(nm0, nn0);                                                                              \
        }                                                                                                         \
    }                                                                                                             \
    const int nbuf = 1 - (BUF);                                                                                   \
    HD_SB()
#define HD_STAMP(var)                                                                             \
    do {                                                                                          \
        if (STAMP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");    \
    } while (0)
#define HD_TILE(S, P)                                                                                             \
    do {                                                                                                          \
        const bool dyn = queue != nullptr && tcount + 2 >= n_static;                                              \
        nn_id = blockIdx.x + (tcount + 2) * G;                                                                    \
        if (dyn && tid == 0) ticket = atomicAdd(queue, 1u);                                                       \
        HD_STAMP(c0);                                                                                             \
        if (EVENK) {                                                                                              \
            {                                                                                                     \
                HD_KSETUP_S(0, 0);                                                                                \
                HD_KSTEP_F0(S, P);                                                                                \
            }                                                                                                     \
            HD_STAMP(c1);                                                                                         \
            if (dyn && tid == 0) id_slot[0] = n_static * G + (int)ticket; /* the wait for the ticket lands here: free */ \
            {                                                                                                     \
                HD_KSETUP_S(1, 1);                                                                                \
                HD_KSTEP_S1(S, P);                                                                                \
            }                                                                                                     \
            HD_STAMP(c2);                                                                                         \
            for (int kt = 2; kt < ksteps; kt += 2) {                                                              \
                {                                                                                                 \
                    HD_KSETUP_S(kt, 0);                                                                           \
                    HD_KSTEP_P0(S, P);                                                                            \
                }                                                                                                 \
                {                                                                                                 \
                    HD_KSETUP_S(kt + 1, 1);                                                                       \
                    HD_KSTEP_P1(S, P);                                                                            \
                }                                                                                                 \
            }                                                                                                     \
        } else {                                                                                                  \
            {                                                                                                     \
                HD_KSETUP(0);                                                                                     \
                HD_KSTEP_F(S, P);                                                                                 \
                buf ^= 1;                                                                                         \
            }                                                                                                     \
            HD_STAMP(c1);                                                                                         \
            if (dyn && tid == 0) id_slot[0] = n_static * G + (int)ticket;                                         \
            {                                                                                                     \
                HD_KSETUP(1);                                                                                     \
                HD_KSTEP_S(S, P);                                                                                 \
                buf ^= 1;                                                                                         \
            }                                                                                                     \
            HD_STAMP(c2);                                                                                         \
            for (int kt = 2; kt < ksteps; ++kt) {                                                                 \
                HD_KSETUP(kt);                                                                                    \
                HD_KSTEP_P(S, P);                                                                                 \
                buf ^= 1;                                                                                         \
            }                                                                                                     \
        }                                                                                                         \
        HD_STAMP(c3);                                                                                             \
        if (STAMP) { tF += c1 - c0; tS += c2 - c1; tP += c3 - c2; }                                               \
        if (dyn) nn_id = __builtin_amdgcn_readfirstlane(id_slot[0]); /* written before this tile's last barrier */ \
    } while (0)
    // what the epilogue of the tile just finished needs: its bias slices and this lane's first output element
#define HD_SET_EPILOGUE()                                                                                         \
    do {                                                                                                          \
        const unsigned ba_ = bias_addr + 4u * n0;                                                                 \
        HD_DS128(bq[0], ba_, 0); HD_DS128(bq[1], ba_, 128); HD_DS128(bq[2], ba_, 256); HD_DS128(bq[3], ba_, 384); \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3])::"memory");      \
        Yt = Y + (m0 + wm * 64) * ldy + n0 + wn * 128;                                                            \
        have_prev = true;                                                                                         \
    } while (0)
    // the last tile of this workgroup: its set is stored without matrix work to hide behind
#define HD_EBLOCK(P, I, J)                                                                                        \
    do {                                                                                                          \
        HD_EW(P, I, J, 0); HD_EW(P, I, J, 1); HD_EW(P, I, J, 2); HD_EW(P, I, J, 3); HD_EW(P, I, J, 4); HD_EW(P, I, J, 5);   \
        HD_EW(P, I, J, 6); HD_EW(P, I, J, 7); HD_EW(P, I, J, 8); HD_EW(P, I, J, 9); HD_EW(P, I, J, 10); HD_EW(P, I, J, 11); \
        HD_EW(P, I, J, 12); HD_EW(P, I, J, 13); HD_EW(P, I, J, 14); HD_EW(P, I, J, 15);                            \
        HD_ER(0); HD_ER(1); HD_ER(2); HD_ER(3);                                                                   \
        HD_EWAIT();                                                                                               \
        HD_EV(J, 0, 4);                                                                                           \
        HD_ES(I, J, 0); HD_ES(I, J, 1); HD_ES(I, J, 2); HD_ES(I, J, 3);                                           \
        HD_SB();                                                                                                  \
    } while (0)
#define HD_FINAL_EPILOGUE(P)                                                                                      \
    do {                                                                                                          \
        HD_EBLOCK(P, 0, 0); HD_EBLOCK(P, 0, 1); HD_EBLOCK(P, 0, 2); HD_EBLOCK(P, 0, 3);                           \
        HD_EBLOCK(P, 1, 0); HD_EBLOCK(P, 1, 1); HD_EBLOCK(P, 1, 2); HD_EBLOCK(P, 1, 3);                           \
    } while (0)

    int nn_id = 0;
    for (;;) {
        tile_coords(lin_id, m0, n0);
        HD_TILE(0, 1);
        HD_SET_EPILOGUE();
        lin_id = nxt_id; nxt_id = nn_id; ++tcount;
        nk += ksteps;
        if (lin_id >= ntiles) {
            if (STAMP) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(mt1), "=s"(rt1)::"memory");
            HD_FINAL_EPILOGUE(0);
            break;
        }
        tile_coords(lin_id, m0, n0);
        HD_TILE(1, 0);
        HD_SET_EPILOGUE();
        lin_id = nxt_id; nxt_id = nn_id; ++tcount;
        nk += ksteps;
        if (lin_id >= ntiles) {
            if (STAMP) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(mt1), "=s"(rt1)::"memory");
            HD_FINAL_EPILOGUE(1);
            break;
        }
    }
#ifdef M360_DIAG
    if (STAMP && tid == 0 && blockIdx.x < 256) {
        g_hd_stamps[blockIdx.x * 4 + 0] = mt1 - mt0;
        g_hd_stamps[blockIdx.x * 4 + 1] = rt1 - rt0;
        g_hd_stamps[blockIdx.x * 4 + 2] = nk;
        g_hd_stamps[1024 + blockIdx.x * 4 + 0] = tF;
        g_hd_stamps[1024 + blockIdx.x * 4 + 1] = tS;
        g_hd_stamps[1024 + blockIdx.x * 4 + 2] = tP;
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA of this wave may land after the workgroup is gone
#undef HD_DS128
#undef HD_SB
#undef HD_WAIT_FRAG
#undef HD_BARRIER
#undef HD_DMA_A
#undef HD_DMA_B
#undef HD_EW
#undef HD_ER
#undef HD_EWAIT
#undef HD_ES
#undef HD_EV
#undef HD_TILE
#undef HD_STAMP
#undef HD_G
#undef HD_BARRIER_E
#undef HD_KSTEP_F
#undef HD_KSTEP_S
#undef HD_KSTEP_P
#undef HD_KSTEP_F0
#undef HD_KSTEP_S1
#undef HD_KSTEP_P0
#undef HD_KSTEP_P1
#undef HD_KSETUP
#undef HD_KSETUP_S
#undef HD_SET_EPILOGUE
#undef HD_FINAL_EPILOGUE
#undef HD_EBLOCK
}

}  // namespace hd
}  // namespace m360
