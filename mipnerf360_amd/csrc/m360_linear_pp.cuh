// Third-generation fp32-MFMA linear kernel: 8 waves = two per SIMD in "ping-pong" (the structure of
// m360_linear_bf16_pp.cuh applied to v_mfma_f32_32x32x2_f32).
//
// The one-wave-per-SIMD kernel (m360_linear_persist.cuh) leaves three things exposed: the issue of the 16 LDS-DMA
// instructions per K-step (~3 %), the barrier skew and the epilogue (~2 %) - a lone wave cannot hide its own non-matrix
// work.  Here the 256 x 256 tile belongs to 8 waves (2 along M x 4 along N, wave tile 128 x 64 = 4 x 2 MFMA blocks,
// 128 accumulator registers).  A K-step (32 floats) is 2 phases, phase p = M-blocks 2p, 2p+1 of the wave tile:
//       [stores of the PREVIOUS tile | ds_read the operands | LDS-DMA of the NEXT K-step + vmcnt(0) (phase 1)]
//       s_barrier   [64 MFMAs = 4096 cycles: 4 rotating accumulators over the whole K-step]   s_barrier
// and the second M half runs one barrier behind the first: on every SIMD one wave computes while its partner does
// everything else.  The epilogue of a tile is folded into the load halves of the next tile's first K-step.
// Operands are swapped (A := weight rows, B := activation rows): a lane's registers 4t..4t+3 of a block are 4
// consecutive output columns -> 16-byte stores without an LDS transposition.  Every accumulator sees the same
// k-pairs in the same order as in the other two kernels (K-groups 0..3, steps 0..3, K-steps ascending), and products
// commute exactly, so the three kernels are bit-identical.
#pragma once
#include "m360_common.cuh"
#include "m360_linear_persist.cuh"

namespace m360 {
namespace pp32 {

using persist::act_fn;
using persist::f32x16;
using persist::f32x4;
using persist::lds_ptr_t;

constexpr int BM = 256, BN = 256, BK = 32;
constexpr int kThreads = 512;
constexpr int kTileBytes = 256 * 128;      // one operand of one K-step: 256 rows x 128 B
constexpr int kBufBytes = 2 * kTileBytes;  // A + B
constexpr int kMaxBias = 4096;             // widest layer (bias is served from LDS)

template <int ACT>
__global__ __launch_bounds__(kThreads, 1) void linear_f32_pp_kernel(
    const float *__restrict__ X, long M, int ldx, const float *__restrict__ W, const float *__restrict__ bias, int Np,
    int Kp, float *__restrict__ Y, int ldy, int tiles_n, int ntiles) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * kBufBytes + kMaxBias * 4];  // 128 KiB + bias

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, h = lane >> 5;
    const int ksteps = Kp / BK;

    auto tile_coords = [&](int id, long &tm0, int &tn0) __attribute__((always_inline)) {
        const int full = (ntiles / 8) * 8;  // XCD-aware (speed only), as in the persistent kernel
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

    // ---- staging: wave w fills rows [32w, 32w+32) of the A tile and of the B tile, 4 + 4 instructions of 8 rows x 128 B
    unsigned a_off[4], b_off[4];  // per-lane byte offsets inside a tile (the same for every tile)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 32 * wave + 8 * q + (lane >> 3);
        const int chunk = (lane & 7) ^ ((r >> 1) & 7);
        a_off[q] = (unsigned)(r * ldx + 4 * chunk) * 4u;
        b_off[q] = (unsigned)(r * Kp + 4 * chunk) * 4u;
    }
    __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X + m0 * ldx), 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(W + (long)n0 * Kp), 0, 0x7fffffff, 0x00020000);
    char *const dma_dst = smem + wave * 32 * 128;
    auto stage_a = [&](int buf, int k0) __attribute__((always_inline)) {
        char *dst = dma_dst + buf * kBufBytes;
#pragma unroll
        for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (lds_ptr_t)(dst + q * 1024), 16, a_off[q], 4 * k0, 0, 0);
    };
    auto stage_b = [&](int buf, int k0) __attribute__((always_inline)) {
        char *dst = dma_dst + buf * kBufBytes + kTileBytes;
#pragma unroll
        for (int q = 0; q < 4; ++q) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_ptr_t)(dst + q * 1024), 16, b_off[q], 4 * k0, 0, 0);
    };

    // ---- operand reads: lane (l31, h), K-group g reads chunk 2g + h of its row = slot (2g + h) ^ ((row >> 1) & 7)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const int fsw = (l31 >> 1) & 7;
    unsigned x_addr[4], w_addr[4];  // buffer 0; + kBufBytes for buffer 1; + block * 4096
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const unsigned slot = ((2 * g + h) ^ fsw) * 16;
        x_addr[g] = lds0 + (wm * 128 + l31) * 128 + slot;
        w_addr[g] = lds0 + kTileBytes + (wn * 64 + l31) * 128 + slot;
    }

    f32x16 acc[4][2];
    f32x4 fw[2][4];  // weight rows: N-block jb, K-group g (held for the whole K-step)
    f32x4 fx[2][4];  // activation rows of the phase's two M-blocks, K-group g

#define P32_DS128(dst, addr, imm) asm volatile("ds_read_b128 %0, %1 offset:" #imm : "=v"(dst) : "v"(addr))
#define P32_BAR() asm volatile("s_barrier" ::: "memory")
#define P32_SB() __builtin_amdgcn_sched_barrier(0)
#define P32_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define P32_READ_W(boff)                                      \
    do {                                                      \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {       \
            const unsigned wa_ = w_addr[g] + (boff);          \
            P32_DS128(fw[0][g], wa_, 0);                      \
            P32_DS128(fw[1][g], wa_, 4096);                   \
        }                                                     \
    } while (0)
#define P32_READ_X(boff, IMM0, IMM1)                          \
    do {                                                      \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {       \
            const unsigned xa_ = x_addr[g] + (boff);          \
            P32_DS128(fx[0][g], xa_, IMM0);                   \
            P32_DS128(fx[1][g], xa_, IMM1);                   \
        }                                                     \
    } while (0)
// only the fragments a phase has just loaded are tied to its wait (dead ones would be kept alive)
#define P32_WAIT_X()                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                            \
                 : "+v"(fx[0][0]), "+v"(fx[0][1]), "+v"(fx[0][2]), "+v"(fx[0][3]), "+v"(fx[1][0]), "+v"(fx[1][1]), \
                   "+v"(fx[1][2]), "+v"(fx[1][3])::"memory")
#define P32_WAIT_W()                                                                                               \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                            \
                 : "+v"(fw[0][0]), "+v"(fw[0][1]), "+v"(fw[0][2]), "+v"(fw[0][3]), "+v"(fw[1][0]), "+v"(fw[1][1]), \
                   "+v"(fw[1][2]), "+v"(fw[1][3])::"memory")
// M-blocks I0, I0+1 x N-blocks 0, 1 over the whole K-step: K-groups ascending, steps ascending (the order of the other
// kernels for every accumulator); four independent accumulators rotate, so no MFMA waits for its predecessor
#define P32_MFMA64(I0)                                                                                            \
    do {                                                                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) _Pragma("unroll") for (int s = 0; s < 4; ++s) {             \
            acc[I0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[0][g][s], fx[0][g][s], acc[I0][0], 0, 0, 0);     \
            acc[I0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[1][g][s], fx[0][g][s], acc[I0][1], 0, 0, 0);     \
            acc[(I0) + 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[0][g][s], fx[1][g][s], acc[(I0) + 1][0], 0, 0, 0); \
            acc[(I0) + 1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fw[1][g][s], fx[1][g][s], acc[(I0) + 1][1], 0, 0, 0); \
        }                                                                                                         \
        __builtin_amdgcn_s_setprio(0);                                                                            \
    } while (0)
// deferred epilogue of the PREVIOUS tile, M-block Q: bias + activation, 8 stores of 16 bytes, accumulators back to zero
#define P32_STORE_Q(Q)                                                                                \
    do {                                                                                              \
        _Pragma("unroll") for (int jb = 0; jb < 2; ++jb) {                                            \
            f32x4 b0_, b1_, b2_, b3_; /* bias of this lane's 16 columns of N-block jb, from LDS */    \
            const unsigned ba_ = bias_tile + 128u * jb;                                               \
            P32_DS128(b0_, ba_, 0);                                                                   \
            P32_DS128(b1_, ba_, 32);                                                                  \
            P32_DS128(b2_, ba_, 64);                                                                  \
            P32_DS128(b3_, ba_, 96);                                                                  \
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b0_), "+v"(b1_), "+v"(b2_), "+v"(b3_)::"memory"); \
            const f32x4 bb_[4] = {b0_, b1_, b2_, b3_};                                                \
            _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                           \
                float4 o_;                                                                            \
                o_.x = act_fn<ACT>(acc[Q][jb][4 * t + 0] + bb_[t][0]);                                \
                o_.y = act_fn<ACT>(acc[Q][jb][4 * t + 1] + bb_[t][1]);                                \
                o_.z = act_fn<ACT>(acc[Q][jb][4 * t + 2] + bb_[t][2]);                                \
                o_.w = act_fn<ACT>(acc[Q][jb][4 * t + 3] + bb_[t][3]);                                \
                *reinterpret_cast<float4 *>(Yp + (long)((Q) * 32) * ldy_t + jb * 32 + 8 * t) = o_;    \
            }                                                                                         \
        }                                                                                             \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                              \
            acc[Q][0][r] = 0.0f;                                                                      \
            acc[Q][1][r] = 0.0f;                                                                      \
        }                                                                                             \
        asm volatile("" ::: "memory");                                                                \
        P32_SB();                                                                                     \
    } while (0)
// One K-step = 2 phases of 64 MFMAs (4096 cycles).  ST (wave-uniform, load halves only): also store the previous tile,
// 16 stores per phase, BEFORE the reads.  The whole next K-step is staged in phase 0 after the phase's own reads (the
// other wave group issued its last reads of that buffer before the barrier this phase started from; the DMA data lands
// hundreds of cycles after those reads have returned) and retired by phase 1's counted vmcnt, one phase before its
// first read: a budget of > 8000 cycles, like the one-wave kernel's.
#define P32_KSTEP(ST)                                         \
    do {                                                      \
        P32_SB();                                             \
        if (ST) {                                             \
            P32_STORE_Q(0);                                   \
            P32_STORE_Q(1);                                   \
        }                                                     \
        P32_READ_W(boff);                                     \
        P32_READ_X(boff, 0, 4096);                            \
        stage_a(nbuf, k_next);                                \
        stage_b(nbuf, k_next);                                \
        P32_BAR();                                            \
        P32_WAIT_W();                                         \
        P32_WAIT_X();                                         \
        P32_SB();                                             \
        P32_MFMA64(0);                                        \
        P32_SB();                                             \
        P32_BAR();                                            \
        if (ST) {                                             \
            P32_STORE_Q(2);                                   \
            P32_STORE_Q(3);                                   \
        }                                                     \
        P32_READ_X(boff, 8192, 12288);                        \
        if (ST) P32_VMCNT(16); else P32_VMCNT(0);             \
        P32_BAR();                                            \
        P32_WAIT_X();                                         \
        P32_SB();                                             \
        P32_MFMA64(2);                                        \
        P32_SB();                                             \
        P32_BAR();                                            \
    } while (0)

    // ---- bias -> LDS once (before any LDS-DMA is in flight)
    float *const bias_lds = reinterpret_cast<float *>(smem + 2 * kBufBytes);
    for (int i = tid; i < Np; i += kThreads) bias_lds[i] = bias[i];
    __syncthreads();
    const unsigned bias_addr = lds0 + 2 * kBufBytes + 4u * (wn * 64 + 4 * h);  // + 4 * n0, + 128 * jb, + 32 * t

    // ---- prologue: K-step 0 of the first tile
    stage_a(0, 0);
    stage_b(0, 0);
    P32_VMCNT(0);
    P32_BAR();
    if (wm == 1) P32_BAR();  // the second M half runs one barrier behind the first

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    unsigned bias_tile = bias_addr;  // LDS address of the bias of this lane's first column of the tile being written
    float *Yp = Y;      // this lane's first output element of that tile
    int ldy_t = ldy;
    bool have_prev = false;
    int buf = 0;
    for (; tile_id < ntiles; tile_id += G) {
        tile_coords(tile_id, m0, n0);
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
                    rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(X + nm0 * ldx), 0, 0x7fffffff, 0x00020000);
                    rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(W + (long)nn0 * Kp), 0, 0x7fffffff, 0x00020000);
                }
            }
            const unsigned boff = buf ? (unsigned)kBufBytes : 0u;
            const int nbuf = buf ^ 1;
            const bool st = have_prev && kt == 0;
            P32_KSTEP(st);
            buf ^= 1;
        }
        // the finished tile is written during the first K-step of the next one (or below, if it is the last): its bias
        {
            bias_tile = bias_addr + 4u * n0;
            Yp = Y + (m0 + wm * 128 + l31) * ldy_t + n0 + wn * 64 + 4 * h;
            asm volatile("" : "+s"(ldy_t));
            have_prev = true;
        }
    }
    // ---- the last tile of this workgroup: plain epilogue (no barrier: the groups keep their one-barrier stagger)
    P32_STORE_Q(0);
    P32_STORE_Q(1);
    P32_STORE_Q(2);
    P32_STORE_Q(3);
    if (wm == 0) P32_BAR();  // equalise the barrier count of the two groups
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef P32_DS128
#undef P32_BAR
#undef P32_SB
#undef P32_VMCNT
#undef P32_READ_W
#undef P32_READ_X
#undef P32_WAIT_X
#undef P32_WAIT_W
#undef P32_MFMA64
#undef P32_STORE_Q
#undef P32_KSTEP
}

}  // namespace pp32
}  // namespace m360
