// Third bf16 linear kernel (round 3): ONE wave per SIMD, 128 x 128 wave tiles on v_mfma_f32_16x16x32_bf16, whole-line LDS-DMA.
//
// Why: the 8-wave ping-pong kernel (m360_linear_bf16_pp.hip.h) reads 192 KiB of fragments and writes 64 KiB of LDS-DMA per 64-deep
// K-step and CU - the whole 128 B/clk of the LDS in the 2048 cycles the matrix work takes.  A probe of the bare K-step
// (tools/wide_wave_probe.hip, profiles/r03/bf16_wide_wave_probe.jsonl, same box) puts that geometry's bound at 1.32-1.34 PF
// and the bound of FOUR waves with 128 x 128 wave tiles (128 KiB of fragment reads) at 1.57-1.58 PF - with the 16-cycle MFMA:
// the 32-cycle v_mfma_f32_32x32x16_bf16 of round 2's attempt at this geometry (diag/m360_linear_bf16_w32.hip.h, 0.98-1.03 PF)
// draws more power per flop (1.46 PF at 1.46 GHz in the same probe).  That attempt, and the first form of this kernel, staged
// 32-deep slabs: an LDS-DMA piece of 16 rows x 64 B is HALF of sixteen 128-byte lines, and such pieces stream at 42 GB/s per CU -
// exactly the rate both kernels ran at (0.79 ms per 1024^2 layer) - where whole-line pieces reach 65 GB/s with as little as 32 KiB
// in flight (tools/dma_shape_probe.hip, profiles/r03/dma_piece_shape_probe.jsonl).  Hence:
//   * a stage is 64 deep (128-byte LDS rows, pieces of 8 rows x 128 B, source-side XOR swizzle), the LDS holds two stages.  65 GB/s
//     per CU is one piece per ~31 cycles: the 64 pieces of a stage need all of the stage's 2048 matrix cycles, so they are issued
//     at a UNIFORM rate (4 per wave and half k-step, one per 8th MFMA gap) and every REGION of a buffer (weight rows, activation
//     rows of blocks 0-3, of blocks 4-7) is refilled as soon as its last reader is done - three barriers per stage, every piece
//     issued >= 1024 matrix cycles before the barrier that needs it (a first form that filled whole buffers in bursts between one
//     barrier per stage waited 40 % of its time for the last burst: 1.09 PF);
//   * a GENERATED schedule (tools/gen_w16_slab.py -> m360_linear_bf16_w16_gen.inc): every ds_read_b128 / LDS-DMA piece / store in
//     its own MFMA gap, buffer offsets static, the counted waits from a simulation of the issue order;
//   * operands swapped (MFMA A := weight rows, B := activation rows): D[i][j], lane (j = lane & 15, g4 = lane >> 4) holds rows
//     i = 4 g4 + r of 16 output columns for ONE activation row;
//   * a k-step is 8 x 8 blocks = 64 MFMAs in two halves (activation blocks 0-3 | 4-7); fragment registers: one set of 8
//     activation fragments (refilled half by half) + two sets of 8 weight fragments = 96 VGPRs; the 256 accumulators live in
//     AccVGPRs (the MFMAs are inline assembly with "+a" operands);
//   * weight rows permuted so that a lane's accumulators of weight blocks 2p, 2p+1 are 8 CONSECUTIVE output columns and the four
//     lanes of a row cover 32 consecutive columns: MFMA row i of block jb is column 32 (jb >> 1) + 8 (i >> 2) + 4 (jb & 1) + (i & 3)
//     of the wave's 128.  Epilogue per (activation block, p): 4 packed bias adds, 4 v_cvt_pk_bf16_f32, ReLU as v_pk_max_i16 on the
//     packed pairs -> one packed 16-byte row piece; lanes l15 = 2j, 2j + 1 then swap one piece each (v_cmp + v_cndmask_b32_dpp) so that a
//     store instruction writes 8 rows x 128 B = whole lines (142 GB/s per CU against 38 for 16 rows x 64 B); no LDS transposition;
//   * the epilogue is exposed (vector work is not hidden behind the same wave's MFMAs): ~5.5 k cycles of conversion per tile with
//     the 32 stores issued as the pieces are packed.  vmcnt retires in order and a store takes ~2 k cycles to complete, so a
//     counted wait for a piece issued after a store cannot complete before that store has: the LAST stage of a tile therefore also
//     issues the activation pieces of the next tile's stage 1 - everything the next tile's first stage waits for is older than the
//     stores, which stay in flight behind it.  Carrying the packed pieces in 128 VGPRs and storing them during the next tile was
//     built and measured WORSE the thinner the stores were spread (8 per stage: 600 cycles per k-step of those stages, one per
//     k-step: 670 cycles in every k-step - tools/gen_w16_slab.py).  With the stores the K loop still runs 10 % slower than without
//     (1260 against 1136 cycles per 32 deep); starting the workgroups of an XCD up to 30 k cycles apart, so that the 256 epilogues
//     do not store in the same microseconds, changes nothing (profiles/r03/bf16_w16_start_stagger_REJECTED.jsonl), and the stores
//     retire within ~270 cycles (ABL 64): what costs is their lines in L2 - the stores are non-temporal (+2 %, see W16_SWAP_STORE).
// (Round 4 added: paired rows between layers - PAIR, no lane exchange before the stores; 96 of the 256 accumulators in ArchVGPRs; and
// CHAIN - several equally shaped layers in one launch, see chain_t below.)
// Takes full 256 x 256 tiles of layers with K a multiple of 128 (>= 256) or K = 64 (ONE_BLOCK: the first layers - all epilogue,
// 1.07 GB of whole-line stores), bias + {none, ReLU}; in its X3 form every K that is a multiple of 64; with HEADS the sigmoid /
// fused-heads last layers of the rendering forward.  Everything else (K = 128 / 192, sigmoid without heads, fused heads that also
// keep the layer's output) stays with the ping-pong kernel.
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

constexpr int BM = 256, BN = 256, BKS = 64;  // stage depth
constexpr int kThreads = 256;
constexpr int kXBytes = 256 * 128;           // activation rows of one stage
constexpr int kStageBytes = 2 * kXBytes;     // + weight rows
constexpr int kMaxBias = 4096;
#ifndef M360_W16_PLAIN_STORES
#define M360_W16_PLAIN_STORES 0  // A/B builds: 1 = temporal instead of non-temporal output stores in every instantiation
#endif
#ifndef M360_W16_ACC_V_BLOCKS
#define M360_W16_ACC_V_BLOCKS 3
#endif
constexpr int kAccVBlocks = M360_W16_ACC_V_BLOCKS;  // activation blocks (of 8) whose accumulators live in ArchVGPRs (W16_ACC_V)

struct cursor_t { __amdgpu_buffer_rsrc_t rsrc; int k; int tile; };  // an LDS-DMA cursor: descriptor of its tile, byte offset in a row

// CHAIN (round 4; hardened in round 5): up to 8 equally shaped ReLU layers (width x width, paired rows) in ONE launch of exactly 256 workgroups,
// the hidden activations handed from layer to layer through the XCD's L2 instead of a kernel boundary.  Workgroup b is column tile c = (b / 8) & 3 of
// quartet Q = 8 (b % 8) + (b / 8) / 4: the four workgroups of a quartet run on four CUs of the XCD that serves slot b % 8 and own the row blocks
// Q + 64 i, which they walk in pairs - (Ra, 0), (Rb, 0), (Ra, 1), (Rb, 1), ... - so that the tile whose activation pieces a last stage prefetches
// never depends on the running one.  Layer 0 reads x_in (never written: the launch can be repeated), layer j >= 1 reads act[j & 1]; layer j
// writes act[(j + 1) & 1]; tile number t of the sequence - (R, j) - may fetch its activations once all 128 waves of the XCD have stored their
// tile t - 2, which for its own quartet is (R, j - 1): done[slot][t - 2] == 128 (a wave adds 1 once its tile's stores have retired - behind the
// counted waits of the NEXT tile's second stage, the last tile behind a drain: the stores are then in the L2 both sides share; the readers'
// pieces carry sc1 and bypass the CU's L1).  That wait also covers the write-after-read on act[(j + 1) & 1].
// The two things the hand-over RELIES on are checked by the kernel itself and reported in chain_status_t::error, which gates a layer-by-layer
// re-run queued behind every launch (m360_linear.hip: mlp_chain_bf16_safe) - a launch whose assumptions did not hold costs time, never rows:
//   * placement: all workgroups of slot s = b % 8 run on ONE XCD (one L2).  Every workgroup reads HW_REG_XCC_ID and compares it with what the
//     first workgroup of its slot published (xcc_slot[s]); a difference sets kChainErrXcc.  (The dispatcher's round-robin over the XCDs is a
//     hardware habit, not a contract: MI355X_MICROARCH.md uses it for speed only.)
//   * residency: all 256 workgroups run at the same time (one per CU: 144 KiB of LDS).  A kernel on another stream that holds CUs - an RCCL
//     all-gather waiting for its peers, a second renderer - breaks that for as long as it runs.  Waits are bounded by wall-clock time
//     (wait_ticks of s_memrealtime, 0.1 s): a wait that runs out sets kChainErrTimeout and ends all further waiting of its wave, so the launch
//     always ends, with wrong rows that the gated re-run then overwrites.
// Status block of a chain launch (device memory, 128 bytes; the chain's counters follow it directly, so ONE memset prepares a launch).
// First line: sticky counters, only ever incremented (m360_workspace_init zeroes them once): what a caller reads to learn that launches were
// repaired, and why.  Second line: zeroed by the host before EVERY launch - `error` is also the gate of the layer-by-layer re-run the host
// queues behind the launch (gated launches: a workgroup that reads 0 there returns at once), `xcc_slot[s]` the XCD id + 1 the first
// workgroup of slot s = blockIdx.x & 7 found itself on.
struct chain_status_t {
    unsigned launches;      // chain launches that ran (workgroup 0 counts)
    unsigned timeouts;      // waves that gave up waiting
    unsigned xcc_mismatch;  // workgroups that were not on the XCD of their slot
    unsigned recoveries;    // launches re-run layer by layer (counted by the first gated launch that found `error` set)
    unsigned pad0[12];
    unsigned error;         // != 0: this launch's rows are not to be trusted (bit 0: a wait ran out, bit 1: a workgroup off its slot's XCD)
    unsigned xcc_slot[8];
    unsigned pad1[7];
};
static_assert(sizeof(chain_status_t) == 128, "two 64-byte lines");
constexpr int kChainStatusLaunchOffset = 64;  // offsetof(chain_status_t, error): what the per-launch memset starts at
constexpr unsigned kChainErrTimeout = 1u, kChainErrXcc = 2u;
constexpr unsigned kChainWaitTicks = 10u * 1000u * 1000u;  // bound of one wait in s_memrealtime ticks (100 MHz): 0.1 s where a tile takes 26 us

struct chain_t {
    const __bf16 *w[8];
    const float *b[8];
    const __bf16 *x_in;  // layer 0 reads this (paired rows); it is never written, so the layers can be re-run (NULL: act[0])
    __bf16 *act[2];      // layer j writes act[(j + 1) & 1]; layer j >= 1 reads act[j & 1]
    unsigned *done;      // [8 XCDs][tiles of a workgroup's sequence] (<= row blocks x layers words), zeroed by the host
    chain_status_t *status;
    int layers;
    int row_blocks;      // a multiple of 128 (width 1024) / 512 (width 256): every group of workgroups owns an even number
    unsigned wait_ticks; // bound of one wait (kChainWaitTicks; tests shrink it to force the give-up path)
    int fault;           // test hook, read by the DIAGNOSTICS build only: 1 = workgroup 9 reports a foreign XCD, 2 = every wave treats its first wait as run out
};

#ifdef M360_DIAG
// diagnostics build, per workgroup: [0] cycles (s_memtime) and [1] 100 MHz ticks of the tile loop, [2] stages, [3] cycles in epilogues
__device__ unsigned long long g_w16_stamps[256 * 4];
#endif
// ABL (diagnostic builds; results are wrong unless 0 or 64): 1 = no barrier, 2 = no LDS-DMA, 4 = no fragment reads, 16 = no stores,
// 32 = no epilogue at all, 64 = wait for every outstanding operation at the end of the epilogue (how long do the stores take?),
// 128 = plain instead of non-temporal stores (results correct), 1024 = no counted vmcnt waits in the K loop (wrong results),
// 4096 / 8192 = agent- / system-scope stores (results correct), 16384 = epilogue of stores only (no AccVGPR reads, no packing)
// X3 ("bf16x3", m360_linear_bf16_pp.hip.h): activations [hi(K) | lo(K)], weights [Wh | Wh | Wl] (rows of Kp = 3K), output
// [hi(Np) | lo(Np)]; per 64-deep block three stages xl wh -> xh wh -> xh wl that share an operand with their neighbour
// (tools/gen_w16_slab.py, second half): 4 operand tiles staged per block instead of 6, the same accumulation order as the ping-pong
// kernel's X3 = 2.
// HEADS > 0 (the last hidden layer of a stage, sigmoid, as in the ping-pong kernel): the output heads' dot products are formed on
// the matrix pipe from the packed bf16 pieces the epilogue holds - a lane's 8 consecutive output columns of one row ARE a B fragment
// of v_mfma_f32_16x16x32_bf16, the head rows (as hi / lo bf16 terms, from 16 KiB of LDS) the A fragment: D[head][row], lanes 0-15
// hold the heads of their row over the piece's 32 columns; the four pieces of a wave are accumulated through the C operand, so ONE
// partial sum per row and 128-column wave tile goes to head_part[row][(Np / 256) * 2][HEADS] (slot = 2 (n0 / 256) + wn: 128 bytes
// per sample at width 1024 - the ping-pong kernel writes 8 slots per 256 columns); the layer's own output is NOT written
// (the rendering forward; the tape-keeping one stays on the ping-pong kernel).  The sigmoid is exposed here (11.5 k cycles of epilogue
// per tile instead of 5.4 k), and still the layer takes 0.85 ms against 1.08-1.12 ms (tools/linear_bench.py --variant 150).
constexpr int kHeadMaxN = 1024;  // widest layer the fused heads take (their hi / lo rows live in 16 KiB of LDS)
// SPLIT (defaults to X3): the output rows are [hi(Np) | lo(Np)] pairs.  SPLIT without X3 is the x6 first layer of the bf16x3 mode
// (m360_linear_bf16_split): a plain contraction over bf16 rows whose fp32 result goes out as two bf16 terms.  X3 without SPLIT is the
// first layer of the bf16 mode (m360_linear_bf16x3_bf16out): the three products of two-term features and weights, one bf16 term out.
// LDSEPI (round 4, plain bf16 output only): the epilogue moves the accumulators through a wave-private LDS tile instead of the vector
// pipe - ds_write_b128 straight from the AccVGPRs (16 rows x 64 fp32 columns, XOR-swizzled 16-byte chunks: no bank conflicts),
// ds_read_b128 back with lane (row L >> 3, columns 8 (L & 7) ..+7), so that 8 lanes hold one 128-byte line of bf16 output: no
// v_accvgpr_read, no DPP exchange - 24 instead of 48 vector instructions per 16 outputs, same bias add / conversion / ReLU: same bits.
// PAIR (round 4) / xpair: "paired rows" - the layout the hidden activations have BETWEEN two launches of this kernel (m360.h:
// M360_ROWS_PAIRED_OUT / _IN).  A lane of the epilogue holds two 16-byte pieces of ONE row (columns 32 (2P) + 8 g4.. and
// 32 (2P + 1) + 8 g4..), and a store instruction reaches the chip's store rate only when it writes whole 128-byte lines with the two
// lanes of a line next to each other (W16_SWAP_STORE) - which is why the plain layout needs the lanes' exchange: 16 of the block's
// 65 instructions.  Paired rows need none: inside every block of 2 rows x 64 columns the four 64-byte quarters are transposed - the
// line at row 2R holds [row 2R, columns 0-31 | row 2R + 1, columns 0-31], the line at row 2R + 1 the columns 32-63 of both rows - so
// the first pieces of a lane pair ARE one whole line, the second pieces the other: same store instructions, same addresses, no
// exchange.  The reader is this kernel's LDS-DMA, whose lanes carry their own source addresses: xpair != 0 fetches chunk c of row r
// from line 2 (r >> 1) + (c >> 2), bytes 64 (r & 1) + 16 (c & 3) - the same eight whole lines per piece, a different lane order.
// Rows beyond the last full 256-row tile are other kernels' rows and stay plain in both layouts.
template <int ACT, int ABL = 0, bool STAMP = false, bool X3 = false, bool ONE_BLOCK = false, int HEADS = 0, bool SPLIT = X3, bool LDSEPI = false, bool PAIR = false, bool CHAIN = false, bool KEEP_Y = false>
__global__ __launch_bounds__(kThreads, 1) void linear_bf16_w16_kernel(
    const __bf16 *__restrict__ X, long M, int ldx, const __bf16 *__restrict__ W, const float *__restrict__ bias, int Np,
    int Kp, __bf16 *__restrict__ Y, int ldy, int tiles_n, int ntiles, const float *__restrict__ head_w = nullptr,
    float *__restrict__ head_part = nullptr, int xpair = 0, int stagger = 0, chain_t ch = chain_t(), unsigned *gate = nullptr, int gate_first = 0) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * kStageBytes + kMaxBias * 4 + (HEADS ? 2 * 4 * kHeadMaxN * 2 : 0) + (LDSEPI ? 4 * 4096 : 0)];  // 144 (160) KiB
    static_assert(!LDSEPI || (HEADS == 0 && !SPLIT && !X3 && !PAIR), "the LDS epilogue: plain bf16 output (its 16 KiB sit where the head rows would)");
    static_assert(!PAIR || HEADS == 0, "paired rows are a layout of the layer's own output");
    // (round 6: the chain takes the bf16x3 hidden layers too - X3 with its [hi | lo] split output)
    static_assert(!CHAIN || (PAIR && !ONE_BLOCK && SPLIT == X3 && HEADS == 0 && ACT == M360_ACT_RELU), "the layer chain: bf16 / bf16x3 ReLU layers on paired rows");
    static_assert(HEADS == 0 || HEADS == 1 || HEADS == 4, "1 (proposal) or 4 (NeRF) heads");
    // X3 loop + plain bf16 output: the first layer of the bf16 mode (two-term features and weights in, one bf16 term out)
    static_assert(SPLIT == X3 || HEADS == 0, "X3 loop and split output go together wherever the heads are fused");
    // KEEP_Y (round 5): fused heads AND the layer's own output (plain rows) - the last hidden layer of the tape-keeping bf16 forward (training)
    static_assert(!KEEP_Y || (HEADS > 0 && !X3 && !SPLIT && !PAIR && !LDSEPI && !CHAIN && !ONE_BLOCK), "KEEP_Y: the plain bf16 fused-heads layer");
    constexpr bool STORE_Y = HEADS == 0 || KEEP_Y;
    // accumulators in ArchVGPRs (W16_ACC_V): not in the plain 64-deep form, whose single stage keeps more fragments live (256 ArchVGPRs + spills)
    constexpr int kAccV = (ONE_BLOCK && !X3) ? 0 : kAccVBlocks;
    // vector-memory operations of a tile's epilogue: 32 whole-line stores (64 as [hi | lo]) or, with fused heads, 8 partial-sum stores
    constexpr int W16_STORES = (STORE_Y ? (SPLIT ? 64 : 32) : 0) + (HEADS ? 8 : 0);
    static_assert(ACT == M360_ACT_NONE || ACT == M360_ACT_RELU || (ACT == M360_ACT_SIGMOID && (ABL != 0 || HEADS > 0)), "bias + {none, ReLU}; sigmoid with fused heads");

    // gated launch (the layer-by-layer re-run behind a chain launch): nothing to do unless the chain reported an error; the first gated
    // launch that finds one counts the recovery (gate = &chain_status_t::error: the sticky counters are the 64 bytes in front of it)
    if (gate != nullptr) {
        if (__hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
        if (gate_first && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_fetch_add(&reinterpret_cast<chain_status_t *>(reinterpret_cast<char *>(gate) - kChainStatusLaunchOffset)->recoveries, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, g4 = lane >> 4;
    const int G = gridDim.x;
    const int K1 = X3 ? Kp / 3 : Kp;  // the layer's contraction length
    const int nstages = K1 / BKS;     // 64-deep blocks per tile.  plain: even, >= 4 (first, generic and last stage of a tile are
                                      // different bodies; ONE_BLOCK: == 1); X3: >= 2 (ONE_BLOCK: == 1), three stages each
    const int kbytes = 2 * K1;

    auto tile_coords = [&](int id, long &tm0, int &tn0) __attribute__((always_inline)) {
        const int full = (ntiles / 8) * 8;  // XCD-aware (speed only): ids sharing id % 8 cover a contiguous range of tiles
        int lin = id;
        if (id < full) lin = (id % 8) * (full / 8) + id / 8;
        tm0 = (long)(lin / tiles_n) * BM;
        tn0 = (lin % tiles_n) * BN;
    };
    int tile_id = blockIdx.x;
    if (tile_id >= ntiles) return;
    // CHAIN: this workgroup's column tile and quartet; its sequence of tiles t -> (row block, layer)
    // (tiles_n = 4: quartets as above; tiles_n = 1 - a 256-wide layer: every workgroup is its own "quartet" and owns the row blocks b + 256 i)
    const int ch_nq = 256 / tiles_n;  // groups of tiles_n workgroups
    const int ch_col = tiles_n == 4 ? (blockIdx.x >> 3) & 3 : 0;
    const int ch_q = tiles_n == 4 ? 8 * (blockIdx.x & 7) + (blockIdx.x >> 5) : 32 * (blockIdx.x & 7) + (blockIdx.x >> 3);
    const int ch_T = CHAIN ? (ch.row_blocks / ch_nq) * ch.layers : 0;
    auto chain_seq = [&](int t, int &R, int &j) __attribute__((always_inline)) {
        const int two_l = 2 * ch.layers, p_ = t / two_l, rem = t - p_ * two_l;
        j = rem >> 1;
        R = ch_q + ch_nq * (2 * p_ + (rem & 1));
    };
    if (CHAIN) tile_id = 0;
    auto chain_src = [&](int j) __attribute__((always_inline)) -> const __bf16 * { return (j == 0 && ch.x_in) ? ch.x_in : ch.act[j & 1]; };
    if (CHAIN) {  // placement check: is this workgroup on the XCD the first workgroup of its slot ran on?
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
#ifdef M360_DIAG  // test hook (diagnostics build only): one workgroup reports a foreign XCD
        if (ch.fault == 1 && blockIdx.x == 9) xcc ^= 1u;
#endif
        if (threadIdx.x == 0) {
            unsigned seen = 0u;
            __hip_atomic_compare_exchange_strong(&ch.status->xcc_slot[blockIdx.x & 7], &seen, xcc + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (seen != 0u && seen != xcc + 1u) {
                __hip_atomic_fetch_or(&ch.status->error, kChainErrXcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(&ch.status->xcc_mismatch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (blockIdx.x == 0) __hip_atomic_fetch_add(&ch.status->launches, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#ifdef M360_DIAG
    // diagnostics (round 4): start the eight row blocks an XCD works on at once `stagger` x ~1 k cycles apart (the four column tiles of a
    // row block stay together: they share its activation rows through the L2), so that the 256 CUs do not store their tiles in the same
    // microseconds - does the epilogue's store floor (32 MB from all CUs at once) give way?
    if (stagger > 0) {
        const int phase = ((blockIdx.x >> 3) / tiles_n) & 7;
        for (int i = 0; i < phase * stagger; ++i) __builtin_amdgcn_s_sleep(16);
    }
#endif

    // ---- LDS-DMA: 8 activation + 8 weight pieces of 8 rows x 128 B per wave and stage.  Weight rows [64w, 64w + 64).  Activation
    // rows: 32 "lo" rows (blocks 0-3 of a wave tile: pieces 0-3) + 32 "hi" rows (blocks 4-7: pieces 4-7) of wave-tile row
    // w >> 1, so that the lo and hi REGIONS of a buffer can be refilled at different times.  LDS slot of chunk c of row r:
    // c ^ f(r), f = (r >> 1) & 7 for activation rows (a 16-lane read group touches 16 consecutive rows) and
    // 2 ((r >> 3) & 3) + ((r >> 1) & 1) for weight rows (a read group touches rows C + 8 a + b, C even): either way 16 distinct
    // 16-byte slots of the 256-byte bank line
    const int xrow0 = (wave >> 1) * 128 + (wave & 1) * 32;  // first lo row of this wave; its hi rows start 64 further
    unsigned x_voff[8], w_voff[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int rx = xrow0 + (q >> 2) * 64 + 8 * (q & 3) + (lane >> 3);
        const int rw = 64 * wave + 8 * q + (lane >> 3);
        const int cx_ = (lane & 7) ^ ((rx >> 1) & 7);  // the 16-byte chunk of row rx this lane fetches
        x_voff[q] = xpair ? (unsigned)((2 * (rx >> 1) + (cx_ >> 2)) * ldx + 32 * (rx & 1) + 8 * (cx_ & 3)) * 2u : (unsigned)(rx * ldx + 8 * cx_) * 2u;
        w_voff[q] = (unsigned)(rw * Kp + 8 * ((lane & 7) ^ (2 * ((rw >> 3) & 3) + ((rw >> 1) & 1)))) * 2u;
    }
    char *const dma_x = smem + xrow0 * 128;               // + buffer * kStageBytes + (q & 3) * 1024 + (q >> 2) * 8192
    char *const dma_w = smem + kXBytes + 64 * wave * 128;
    // cursors run ahead of the matrix work, across tile boundaries (scalar state: buffer descriptor of the cursor's tile + byte
    // offset of its 64-deep block in a row).  plain: cx = activation pieces of stage s + 1, cw = weight pieces of stage s + 2.
    // X3: cx = Xl, cx2 = Xh, cw = Wh of block b + 1, cw2 = Wl of block b (columns K.., 0.., 0.., 2K.. of their rows).
    cursor_t cx, cx2, cw, cw2;
    {
        long m0_;
        int n0_;
        tile_coords(tile_id, m0_, n0_);
        const __bf16 *x0 = X + m0_ * ldx, *w0 = W + (long)n0_ * Kp;
        if (CHAIN) {
            int R_, j_;
            chain_seq(0, R_, j_);
            x0 = chain_src(0) + (long)R_ * BM * ldx;
            w0 = ch.w[0] + (long)(ch_col * BN) * Kp;
        }
        cx.rsrc = cx2.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(x0), 0, 0x7fffffff, 0x00020000);
        cw.rsrc = cw2.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(w0), 0, 0x7fffffff, 0x00020000);
        cx.k = cx2.k = cw.k = cw2.k = 0;
        cx.tile = cx2.tile = cw.tile = cw2.tile = tile_id;
    }
    // CHAIN: wait (bounded) until all 128 waves of this XCD have stored their tile number t of the sequence: the quartet's own (whose rows
    // tile t + 2 fetches) and the seven others' - the 32 workgroups of an XCD stay on the same layer, i.e. on the same 2 MB of weights in
    // their L2 (without this the quartets drift apart over a long sequence: 8192 x 256 rows ran no faster than layer by layer)
    // A wait that runs out (wait_ticks of the 100 MHz clock: 0.1 s where a tile takes 26 us; only workgroups that are NOT all resident can get
    // there) sets the error bit and ends all further waiting of this wave: one bounded delay per launch, wrong rows, and the gated re-run behind
    // the launch puts them right.
    bool ch_gave_up = false;
    auto chain_wait = [&](int t) __attribute__((always_inline)) {
        if (ch_gave_up) return;
        unsigned *flag = ch.done + (long)(blockIdx.x & 7) * ch_T + t;
#ifdef M360_DIAG  // test hook (diagnostics build only): every wave treats its first wait as run out
        bool lost = ch.fault == 2;
#else
        bool lost = false;
#endif
        if (!lost && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 128u) {
            unsigned long long t0, t1;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 128u) {
                __builtin_amdgcn_s_sleep(8);
                asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
                if (t1 - t0 > (unsigned long long)ch.wait_ticks) { lost = true; break; }
            }
        }
        if (lost) ch_gave_up = true;  // scalar state only: the wave reports it once, behind its tile loop (nothing divergent in the K loop)
    };
#define W16_CUR_ADV(C, IS_X)                                                                                                 \
    do {                                                                                                                     \
        C.k += 2 * BKS;                                                                                                      \
        if (C.k == kbytes) { /* next tile of this workgroup (past the last one: harmlessly the same rows again) */           \
            C.k = 0;                                                                                                         \
            if (CHAIN) {                                                                                                     \
                C.tile += 1;                                                                                                 \
                if (C.tile < ch_T) {                                                                                         \
                    int R_, j_;                                                                                              \
                    chain_seq(C.tile, R_, j_);                                                                               \
                    if (IS_X) {                                                                                              \
                        if (C.tile >= 2) chain_wait(C.tile - 2);                                                             \
                        C.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(chain_src(j_)) + (long)R_ * BM * ldx, 0, 0x7fffffff, 0x00020000); \
                    } else {                                                                                                 \
                        C.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(ch.w[j_]) + (long)(ch_col * BN) * Kp, 0, 0x7fffffff, 0x00020000); \
                    }                                                                                                        \
                }                                                                                                            \
            } else {                                                                                                         \
            C.tile += G;                                                                                                     \
            if (C.tile < ntiles) {                                                                                           \
                long m0_;                                                                                                    \
                int n0_;                                                                                                     \
                tile_coords(C.tile, m0_, n0_);                                                                               \
                C.rsrc = (IS_X) ? __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(X + m0_ * ldx), 0, 0x7fffffff, 0x00020000) \
                                : __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16 *>(W + (long)n0_ * Kp), 0, 0x7fffffff, 0x00020000); \
            }                                                                                                                \
            }                                                                                                                \
        }                                                                                                                    \
    } while (0)
// (ABL 256 / 512, diagnostics: non-temporal activation / weight pieces; results stay right.  One layer launched back to back gains
// 1.3 % from non-temporal activation pieces (1278-1285 against 1263-1268 TF) and loses 4 % from non-temporal weight pieces
// (profiles/r03/bf16_w16_nontemporal_loads_ab_NOT_ADOPTED.jsonl) - but in the step, where a layer reads what the previous one has just written,
// the activation variant measured 7.04 against 6.96-7.02 ms (bf16x3: 17.5 against 16.9-17.1): not adopted.)
#define W16_PIECE_X(C, COL, SLOT, Q) if (!(ABL & 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(C.rsrc, (lds_ptr_t)(dma_x + (SLOT) * kStageBytes + ((Q) & 3) * 1024 + ((Q) >> 2) * 8192), 16, x_voff[Q], C.k + (COL), 0, CHAIN ? 16 : (ABL & 256) ? 2 : 0)
#define W16_PIECE_W(C, COL, SLOT, Q) if (!(ABL & 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(C.rsrc, (lds_ptr_t)(dma_w + (SLOT) * kStageBytes + (Q) * 1024), 16, w_voff[Q], C.k + (COL), 0, (ABL & 512) ? 2 : 0)
#define W16_ADV_X() W16_CUR_ADV(cx, true)
#define W16_ADV_W() W16_CUR_ADV(cw, false)
#define W16_DMA_X(SLOT, Q) W16_PIECE_X(cx, 0, SLOT, Q)
#define W16_DMA_W(SLOT, Q) W16_PIECE_W(cw, 0, SLOT, Q)
// X3: the four operands live in fixed halves of the LDS: Xl -> activation buffer 0, Xh -> 1, Wh -> weight buffer 0, Wl -> 1
#define W16X_ADV_XL() W16_CUR_ADV(cx, true)
#define W16X_ADV_XH() W16_CUR_ADV(cx2, true)
#define W16X_ADV_WH() W16_CUR_ADV(cw, false)
#define W16X_ADV_WL() W16_CUR_ADV(cw2, false)
#define W16X_DMA_XL(Q) W16_PIECE_X(cx, kbytes, 0, Q)
#define W16X_DMA_XH(Q) W16_PIECE_X(cx2, 0, 1, Q)
#define W16X_DMA_WH(Q) W16_PIECE_W(cw, 0, 0, Q)
#define W16X_DMA_WL(Q) W16_PIECE_W(cw2, 2 * kbytes, 1, Q)

    // ---- fragment reads: lane (l15, g4), k-step kk of a stage: chunk 4 kk + g4 of its row.  Activation block ib: row 16 ib + l15
    // of the wave's 128.  Weight block jb: MFMA row i = l15 is LDS row 32 (jb >> 1) + 8 (i >> 2) + 4 (jb & 1) + (i & 3) - so the
    // accumulators acc[.][2p][0..3], acc[.][2p + 1][0..3] of lane group g4 are output columns 32 p + 8 g4 + 0..7
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
    const int wr0 = 8 * (l15 >> 2) + (l15 & 3);                 // weight row of block 0
    const int fxs = (l15 >> 1) & 7, fws = 2 * (l15 >> 2) + ((l15 >> 1) & 1);
    // byte addresses x/wa<kk><buffer>; the blocks are instruction immediates
    const unsigned xa00 = lds0 + (wm * 128 + l15) * 128 + (((0 + g4) ^ fxs) * 16), xa10 = lds0 + (wm * 128 + l15) * 128 + (((4 + g4) ^ fxs) * 16);
    const unsigned wa00 = lds0 + kXBytes + (wn * 128 + wr0) * 128 + (((0 + g4) ^ fws) * 16), wa10 = lds0 + kXBytes + (wn * 128 + wr0) * 128 + (((4 + g4) ^ fws) * 16);
    const unsigned xa01 = xa00 + kStageBytes, xa11 = xa10 + kStageBytes, wa01 = wa00 + kStageBytes, wa11 = wa10 + kStageBytes;

    f32x4 acc[8][8];  // [activation block][weight block]
    bf16x8 fx[8], fw0[8], fw1[8];

#define W16_RD(dst, addr, imm)                                                                          \
    do {                                                                                                \
        if (!(ABL & 4)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm)); \
        else asm volatile("" : "=v"(dst) : "v"(addr));                                                  \
    } while (0)
#define W16_SB() __builtin_amdgcn_sched_barrier(0)
// The MFMAs are inline assembly with the accumulator tied to an AccVGPR ("+a"): with the builtin the register allocator kept half
// of the 64 accumulator tuples in ArchVGPRs across the stage bodies and copied them in and out around every MFMA (4 v_accvgpr_write
// + s_nop per MFMA in the K loop).  The hazard recogniser does not see these MFMAs: the epilogue waits out the last one itself.
// Where the 256 accumulators live: activation blocks [0, 8 - kAccVBlocks) in AccVGPRs, the last kAccVBlocks blocks (8 tuples each) in
// ArchVGPRs - the kernel needs ~130 of the 256 ArchVGPRs for everything else, and an accumulator the epilogue finds in an ArchVGPR costs
// it no v_accvgpr_read_b32 (one issue slot each, tools/valu_issue_probe.hip).  MFMA C and D share a register file: the tuple's own.
#define W16_ACC_V(I) ((I) >= 8 - kAccV)
#define W16_MFMA(I, J, A, B)                                                                                                  \
    do {                                                                                                                      \
        if constexpr (W16_ACC_V(I)) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[I][J]) : "v"(A), "v"(B)); \
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[I][J]) : "v"(A), "v"(B));                      \
    } while (0)
#define W16_MFMA_Z(I, J, A, B)                                                                                                \
    do {                                                                                                                      \
        if constexpr (W16_ACC_V(I)) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(acc[I][J]) : "v"(A), "v"(B)); \
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc[I][J]) : "v"(A), "v"(B));                       \
    } while (0)
// the fragments a wait has just covered are in/out operands of it, so no use can be scheduled above it
#define W16_TIE_HI "+v"(fx[4]), "+v"(fx[5]), "+v"(fx[6]), "+v"(fx[7])
#define W16_WAIT_HI() asm volatile("s_waitcnt lgkmcnt(0)" : W16_TIE_HI::"memory")
#define W16_WAIT_NEXT(FW)                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                     \
                 : "+v"(fx[0]), "+v"(fx[1]), "+v"(fx[2]), "+v"(fx[3]), "+v"(FW[0]), "+v"(FW[1]), "+v"(FW[2]), "+v"(FW[3]),  \
                   "+v"(FW[4]), "+v"(FW[5]), "+v"(FW[6]), "+v"(FW[7])::"memory")
// The three barriers of a stage.  E0 (end of k-step 0): every wave has read the weight and lo rows of this stage's buffer for the
// last time.  M1 (middle of k-step 1): this wave's weight and lo pieces of the NEXT stage have landed, every wave has read the hi rows
// of this buffer for the last time.  E1 (end of k-step 1): the hi pieces of the next stage have landed.  N younger pieces (NS: pieces
// + stores of the previous tile, if it exists) may stay in flight - a bare counted wait under the wave-uniform branch: two
// register-tied variants would meet in a join and cost copies.
#define W16_VMCNT(N, NS)                                                                                \
    do {                                                                                                \
        if (ABL & 1024) break; /* diagnostics: the pieces are issued but never awaited (wrong results): ISSUE cost vs WAITING */ \
        if ((NS) > (N) && have_prev) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS) > 63 ? 63 : (NS)) : "memory"); /* 6-bit counter */ \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");                                   \
    } while (0)
#define W16_BAR() do { if (!(ABL & 1)) asm volatile("s_barrier" ::: "memory"); } while (0)
#define W16_BARRIER_E0() W16_BAR()
#define W16_BARRIER_M1(N, NS)                                                                           \
    do {                                                                                                \
        W16_VMCNT(N, NS);                                                                               \
        if (!(ABL & 1)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : W16_TIE_HI::"memory");       \
        else asm volatile("s_waitcnt lgkmcnt(0)" : W16_TIE_HI::"memory");                               \
    } while (0)
#define W16_BARRIER_E1(N, NS) do { W16_VMCNT(N, NS); W16_BAR(); } while (0)
#include "m360_linear_bf16_w16_gen.inc"
#include "m360_linear_bf16_w16x3_gen.inc"

    // ---- bias -> LDS once (before any LDS-DMA is in flight)
    float *const bias_lds = reinterpret_cast<float *>(smem + 2 * kStageBytes);
    if (CHAIN) {  // [layer][the 256 columns of this workgroup's column tile]
        for (int i = tid; i < ch.layers * BN; i += kThreads) bias_lds[i] = ch.b[i / BN][ch_col * BN + (i % BN)];
    } else
    for (int i = tid; i < Np; i += kThreads) bias_lds[i] = bias[i];
    // head rows as two bf16 terms: hfrag[0][h][n] = bf16(w), hfrag[1][h][n] = bf16(w - hi) (HEADS x Np <= 4 x 1024 values each)
    __bf16 *const hfrag = reinterpret_cast<__bf16 *>(smem + 2 * kStageBytes + kMaxBias * 4);
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
    const unsigned hfrag_addr = lds0 + 2 * kStageBytes + kMaxBias * 4 + 2u * (unsigned)((l15 < HEADS ? l15 : (HEADS ? HEADS - 1 : 0)) * Np + wn * 128 + 8 * g4);
    const unsigned hlo_off = 2u * (unsigned)(HEADS * Np);
    const int hstride = HEADS ? (Np / BN) * 2 * HEADS : 0;  // floats per row of head_part: [(Np / 256) * 2 slots][HEADS]
    const unsigned bias_addr = lds0 + 2 * kStageBytes + 4u * (wn * 128 + 8 * g4);  // + 4 * n0 of the tile, + 128 * p

    // ---- prologue: stages 0 and 1 of the first tile (the first stage of a tile issues no activation pieces: the last stage of its
    // predecessor has).  X3: Xl, Wh, Xh, Wl of block 0.
#define W16_ALL8(M, ...) M(__VA_ARGS__, 0); M(__VA_ARGS__, 1); M(__VA_ARGS__, 2); M(__VA_ARGS__, 3); M(__VA_ARGS__, 4); M(__VA_ARGS__, 5); M(__VA_ARGS__, 6); M(__VA_ARGS__, 7)
    if (X3) {
        W16_ALL8(W16_PIECE_X, cx, kbytes, 0); W16X_ADV_XL();
        W16_ALL8(W16_PIECE_W, cw, 0, 0); W16X_ADV_WH();
        W16_ALL8(W16_PIECE_X, cx2, 0, 1); W16X_ADV_XH();
        W16_ALL8(W16_PIECE_W, cw2, 2 * kbytes, 1); W16X_ADV_WL();
    } else {
        W16_ALL8(W16_PIECE_X, cx, 0, 0); W16_ADV_X();
        W16_ALL8(W16_PIECE_W, cw, 0, 0); W16_ADV_W();
        W16_ALL8(W16_PIECE_W, cw, 0, 1); W16_ADV_W();
        W16_ALL8(W16_PIECE_X, cx, 0, 1); W16_ADV_X();
    }
#undef W16_ALL8
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // stage 0 has landed (this wave's rows)
    __builtin_amdgcn_s_barrier();
    W16_SB();
    W16_RD(fw0[0], wa00, 0); W16_RD(fw0[1], wa00, 512); W16_RD(fw0[2], wa00, 4096); W16_RD(fw0[3], wa00, 4608);
    W16_RD(fw0[4], wa00, 8192); W16_RD(fw0[5], wa00, 8704); W16_RD(fw0[6], wa00, 12288); W16_RD(fw0[7], wa00, 12800);
    W16_RD(fx[0], xa00, 0); W16_RD(fx[1], xa00, 2048); W16_RD(fx[2], xa00, 4096); W16_RD(fx[3], xa00, 6144);
    W16_WAIT_NEXT(fw0);
    W16_SB();

    // this lane's 16 bytes inside a 16-row x 64-column piece, first of its two stores (row 2 (l15 >> 1); the second: one row further)
    const unsigned y_voff = (unsigned)(2 * (l15 >> 1) * ldy + 32 * (l15 & 1) + 8 * g4) * 2u;
    const unsigned lane_odd = (unsigned)(l15 & 1);  // which of a pair of lanes keeps which piece
    bool have_prev = false, odd_tile = false;
    long m0;
    int n0;
    unsigned long long mt0 = 0, rt0 = 0, mt1 = 0, rt1 = 0, e0 = 0, e1 = 0, te = 0, nsl = 0;
    (void)mt1; (void)rt1; (void)e0; (void)e1; (void)te; (void)nsl;
    if (STAMP) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(mt0), "=s"(rt0)::"memory");
    bool ch_pending = false;  // CHAIN: the previous tile of the sequence is stored but not yet counted
    int nbias = 0, ch_R = 0, ch_j = 0;  // where the tile's bias starts in the LDS (n0; CHAIN: 256 x layer), CHAIN: the tile's row block and layer
    __bf16 *Yt = Y;
    for (; tile_id < (CHAIN ? ch_T : ntiles); tile_id += (CHAIN ? 1 : G)) {
        if (CHAIN) {
            chain_seq(tile_id, ch_R, ch_j);
            m0 = (long)ch_R * BM;
            n0 = ch_col * BN;
            nbias = ch_j * BN;
            Yt = ch.act[(ch_j + 1) & 1];
        } else {
            tile_coords(tile_id, m0, n0);
            nbias = n0;
        }
        if (X3) {
            W16X_T1Z_S0();
            W16X_T2_S1();
            if (ONE_BLOCK) {  // a 64-deep layer: the only block of the tile is its first and its last.  (A template parameter: with
                W16X_T3L();   // this body behind a run-time branch of the general kernel its 1024^2 layer took 2.42 instead of 2.21 ms)
            } else {
                W16X_T3();
                // CHAIN: stages 0 and 1 of a tile are the only ones whose counted waits still let the previous tile's stores stay in flight
                // (the generator's "@0" / "@1" bodies): behind the first T3 they have retired - the tile is counted here
                if (CHAIN && ch_pending) {
                    if (lane == 0) __hip_atomic_fetch_add(ch.done + (long)(blockIdx.x & 7) * ch_T + (tile_id - 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ch_pending = false;
                }
                for (int b = 1; b < nstages - 1; ++b) {
                    W16X_T1();
                    W16X_T2();
                    W16X_T3();
                }
                W16X_T1();
                W16X_T2();
                W16X_T3L();
            }
        } else if (ONE_BLOCK) {  // a 64-deep layer: one stage per tile, the buffers alternate by tile
            if (odd_tile) W16_STAGE1ZL();
            else W16_STAGE0ZL();
            odd_tile = !odd_tile;
        } else {
            W16_STAGE0Z();
            W16_STAGE1();
            // CHAIN: the previous tile's stores are complete by now - this stage's counted waits let only its own pieces stay in flight
            // (vmcnt retires in order) - so the tile is counted here, not behind a drain at the end of its epilogue
            if (CHAIN && ch_pending) {
                if (lane == 0) __hip_atomic_fetch_add(ch.done + (long)(blockIdx.x & 7) * ch_T + (tile_id - 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ch_pending = false;
            }
            for (int s = 2; s < nstages - 2; s += 2) {
                W16_STAGE0();
                W16_STAGE1();
            }
            W16_STAGE0();
            W16_STAGE1L();
        }
        // ---- epilogue (exposed): conversion + stores, while the pieces of the next tile's first stages are in flight
        if (STAMP) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(e0)::"memory");
        // the last MFMAs (inline assembly: the hazard recogniser does not see them) have written their accumulators before anything
        // reads them: the 8 tuples of the last 8 MFMAs are redefined by this statement, every other tuple is >= 128 cycles old
        if constexpr (W16_ACC_V(7))
            asm volatile("s_nop 15\n\ts_nop 15"
                         : "+v"(acc[7][0]), "+v"(acc[7][1]), "+v"(acc[7][2]), "+v"(acc[7][3]), "+v"(acc[7][4]), "+v"(acc[7][5]),
                           "+v"(acc[7][6]), "+v"(acc[7][7])::"memory");
        else
            asm volatile("s_nop 15\n\ts_nop 15"
                         : "+a"(acc[7][0]), "+a"(acc[7][1]), "+a"(acc[7][2]), "+a"(acc[7][3]), "+a"(acc[7][4]), "+a"(acc[7][5]),
                           "+a"(acc[7][6]), "+a"(acc[7][7])::"memory");
        // the other accumulators that live in ArchVGPRs are ordinary values to the compiler, whose uses it may hoist: pin them behind the
        // statement above (an MFMA result read too early is a software hazard, not an interlock)
#pragma unroll
        for (int i = 8 - kAccV; i < 7; ++i)
            asm volatile("" : "+v"(acc[i][0]), "+v"(acc[i][1]), "+v"(acc[i][2]), "+v"(acc[i][3]), "+v"(acc[i][4]), "+v"(acc[i][5]),
                              "+v"(acc[i][6]), "+v"(acc[i][7]));
        if (LDSEPI && !(ABL & 32)) {
            const __bf16 *yt = Y + (m0 + wm * 128) * ldy + n0 + wn * 128;  // wave-uniform corner of the wave tile
            // wave-private staging tile: 16 rows x 256 B; 16-byte chunk c of row r lives in slot c ^ r
            const unsigned stg = lds0 + 2 * kStageBytes + kMaxBias * 4 + (unsigned)wave * 4096u;
            unsigned wad[4], rad[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // accumulator tuple q of a column half: chunk 8 (q >> 1) + 2 g4 + (q & 1) of row l15
                const int c = 8 * (q >> 1) + 2 * g4 + (q & 1);
                wad[q] = stg + (unsigned)(l15 * 256 + ((c ^ l15) * 16));
            }
            const int rr = lane >> 3, c0 = 2 * (lane & 7);
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // rows rr (q < 2) and rr + 8 (q >= 2), chunks c0, c0 + 1
                const int r = rr + 8 * (q >> 1), c = c0 + (q & 1);
                rad[q] = stg + (unsigned)(r * 256 + ((c ^ r) * 16));
            }
            const unsigned y_voff2 = (unsigned)(rr * ldy + 8 * (lane & 7)) * 2u;
            const unsigned bias2 = lds0 + 2 * kStageBytes + 4u * (unsigned)(wn * 128 + 8 * (lane & 7));
#pragma unroll
            for (int P = 0; P < 2; ++P) {
                f32x4 b0, b1;  // bias of this lane's 8 columns of the half
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(b0), "=&v"(b1) : "v"(bias2 + 4u * (unsigned)(n0 + 64 * P)) : "memory");
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    asm volatile("ds_write_b128 %0, %4\n\tds_write_b128 %1, %5\n\tds_write_b128 %2, %6\n\tds_write_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                                 ::"v"(wad[0]), "v"(wad[1]), "v"(wad[2]), "v"(wad[3]), "a"(acc[i][4 * P + 0]), "a"(acc[i][4 * P + 1]),
                                 "a"(acc[i][4 * P + 2]), "a"(acc[i][4 * P + 3]) : "memory");
                    f32x4 v0, v1, v2, v3;
                    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(rad[0]), "v"(rad[1]), "v"(rad[2]), "v"(rad[3]) : "memory");
                    u32x4 s1, s2;
#define W16_PK(DST, e, A, be, B_)                                                                     \
    do {                                                                                              \
        f32x2 t_ = {A[be] + B_[be], A[(be) + 1] + B_[(be) + 1]};                                     \
        const bf16x2 h_ = __builtin_convertvector(t_, bf16x2);                                        \
        s16x2 p_ = __builtin_bit_cast(s16x2, h_);                                                     \
        if (ACT == M360_ACT_RELU) p_ = __builtin_elementwise_max(p_, (s16x2){0, 0});                  \
        DST[e] = __builtin_bit_cast(unsigned, p_);                                                    \
    } while (0)
                    W16_PK(s1, 0, v0, 0, b0); W16_PK(s1, 1, v0, 2, b0); W16_PK(s1, 2, v1, 0, b1); W16_PK(s1, 3, v1, 2, b1);
                    W16_PK(s2, 0, v2, 0, b0); W16_PK(s2, 1, v2, 2, b0); W16_PK(s2, 2, v3, 0, b1); W16_PK(s2, 3, v3, 2, b1);
#undef W16_PK
                    const __bf16 *row = yt + (long)(16 * i) * ldy + 64 * P;
                    if (!(ABL & 16))
                        asm volatile("global_store_dwordx4 %0, %1, %3 nt\n\tglobal_store_dwordx4 %0, %2, %4 nt\n\ts_nop 1"
                                     ::"v"(y_voff2), "v"(s1), "v"(s2), "s"(row), "s"(row + 8 * (long)ldy) : "memory");
                    else asm volatile("" ::"v"(s1), "v"(s2));
                }
                W16_SB();
            }
        } else if (!(ABL & 32)) {
            // (ABL 2048, diagnostics, wrong results: every tile of a workgroup is written over the SAME 256 rows - the stores keep their
            // count, shape and TA occupancy but their lines stay in the XCD's L2: what does the HBM side of the output cost the K loop?)
            const __bf16 *yt = Yt + (((ABL & 2048) ? (long)blockIdx.x * BM : m0) + wm * 128) * ldy + n0 + wn * 128;  // wave-uniform corner of the wave tile
            f32x4 hacc[8];  // HEADS: the head sums of this lane's rows (one per activation block) over the pieces done so far
            const unsigned row_step = 16u * (unsigned)ldy;  // elements per activation block
#pragma unroll
            for (int P = 0; P < 2; ++P) {  // column pieces p = 2P, 2P + 1: 64 output columns = one 128-byte line per row
                const __bf16 *rowp = yt;   // rows 16 i .. of the wave tile, i = 0
                asm volatile("" : "+s"(rowp));
                f32x4 bb[4];  // bias of the lane's columns 32 p + 8 g4 + 0..7, p = 2P (bb[0], bb[1]) and 2P + 1 (bb[2], bb[3])
                const unsigned ba = bias_addr + 4u * (nbias + 64 * P);
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:128\n\t"
                             "ds_read_b128 %3, %4 offset:144\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(bb[0]), "=&v"(bb[1]), "=&v"(bb[2]), "=&v"(bb[3]) : "v"(ba) : "memory");
                bf16x8 hh[2], hl[2];  // this lane's head row (hi and lo terms) over the 8 columns of pieces 2P, 2P + 1
                if (HEADS) {
                    const unsigned ha = hfrag_addr + 2u * (unsigned)(n0 + 64 * P);
                    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %5\n\t"
                                 "ds_read_b128 %3, %5 offset:64\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(hh[0]), "=&v"(hh[1]), "=&v"(hl[0]), "=&v"(hl[1]) : "v"(ha), "v"(ha + hlo_off) : "memory");
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    u32x4 ab[2];  // packed pieces (row l15 of block i): ab[0] = columns 32 (2P) + 8 g4.., ab[1] = 32 (2P + 1) + 8 g4..
                    u32x4 lo[2];  // X3: their second bf16 terms
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int p = 2 * P + h;
                        // explicit AccVGPR reads, one (activation block, piece) at a time: left to the register allocator, 150 of
                        // the 256 accumulators were copied out at the top of the epilogue and an address register was spilled
                        f32x4 v, w;
                        if (ABL & 16384) {  // diagnostics (wrong results): no reads, no packing - what do the tile's stores alone take?
                            ab[h] = lo[h] = (u32x4){(unsigned)lane, (unsigned)lane, (unsigned)lane, (unsigned)lane};
                            continue;
                        }
                        if (W16_ACC_V(i)) { v = acc[i][2 * p]; w = acc[i][2 * p + 1]; }  // already in ArchVGPRs
                        else
                        asm volatile("v_accvgpr_read_b32 %0, %8\n\tv_accvgpr_read_b32 %1, %9\n\tv_accvgpr_read_b32 %2, %10\n\t"
                                     "v_accvgpr_read_b32 %3, %11\n\tv_accvgpr_read_b32 %4, %12\n\tv_accvgpr_read_b32 %5, %13\n\t"
                                     "v_accvgpr_read_b32 %6, %14\n\tv_accvgpr_read_b32 %7, %15"
                                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
                                     : "a"(acc[i][2 * p][0]), "a"(acc[i][2 * p][1]), "a"(acc[i][2 * p][2]), "a"(acc[i][2 * p][3]),
                                       "a"(acc[i][2 * p + 1][0]), "a"(acc[i][2 * p + 1][1]), "a"(acc[i][2 * p + 1][2]), "a"(acc[i][2 * p + 1][3]));
                        // Diagnostics (round 6; VERDICT r5 item 2: a ReLU bit tape applied in / produced by this epilogue instead of the
                        // backward's mask pass).  ABL 32768: what APPLYING it costs the input gradient - 8 selects per piece on a wave-level
                        // mask in an SGPR pair (here: exec, i.e. all ones - the results stay right; the real thing fetches 16 SGPRs per piece
                        // with s_load_dwordx16, scalar memory that the counted vmcnt schedule does not see).  ABL 65536: what PRODUCING it
                        // costs the tape-keeping forward - 8 v_cmp per piece into SGPR pairs (the 4 s_store_dwordx4 that would carry them to
                        // memory are scalar instructions and left out).  tools/linear_bench.py --variant 151 / 152 against 100.
                        if constexpr ((ABL & 32768) != 0) {
                            unsigned long long mk_;
                            asm volatile("s_mov_b64 %0, exec" : "=s"(mk_));
                            asm volatile("v_cndmask_b32 %0, 0, %0, %8\n\tv_cndmask_b32 %1, 0, %1, %8\n\tv_cndmask_b32 %2, 0, %2, %8\n\tv_cndmask_b32 %3, 0, %3, %8\n\t"
                                         "v_cndmask_b32 %4, 0, %4, %8\n\tv_cndmask_b32 %5, 0, %5, %8\n\tv_cndmask_b32 %6, 0, %6, %8\n\tv_cndmask_b32 %7, 0, %7, %8"
                                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]) : "s"(mk_));
                        }
                        if constexpr ((ABL & 65536) != 0) {
                            unsigned long long m0_, m1_, m2_, m3_, m4_, m5_, m6_, m7_;
                            asm volatile("v_cmp_lt_f32 %0, 0, %8\n\tv_cmp_lt_f32 %1, 0, %9\n\tv_cmp_lt_f32 %2, 0, %10\n\tv_cmp_lt_f32 %3, 0, %11\n\t"
                                         "v_cmp_lt_f32 %4, 0, %12\n\tv_cmp_lt_f32 %5, 0, %13\n\tv_cmp_lt_f32 %6, 0, %14\n\tv_cmp_lt_f32 %7, 0, %15"
                                         : "=s"(m0_), "=s"(m1_), "=s"(m2_), "=s"(m3_), "=s"(m4_), "=s"(m5_), "=s"(m6_), "=s"(m7_)
                                         : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]));
                        }
// plain: bias, bf16 conversion, ReLU on the packed pair.  X3: bias, ReLU in fp32, hi = bf16(t), lo = bf16(t - hi)
#define W16_PACK(e, a0, a1, bb_, be)                                                                           \
    do {                                                                                                       \
        f32x2 t_ = {a0 + bb_[be], a1 + bb_[(be) + 1]};                                                         \
        if (SPLIT && ACT == M360_ACT_RELU) { t_[0] = relu_nanf_(t_[0]); t_[1] = relu_nanf_(t_[1]); }              \
        if (ACT == M360_ACT_SIGMOID) { /* rcp(1 + __expf(-t)), __expf(-t) = v_exp_f32(t * -log2 e): the same operations, the   \
               multiply and the two adds as PACKED instructions (written per element the compiler splits the bias add as well: 8   \
               instead of 5 instructions per pair, and every instruction costs the lone wave an issue slot) */                      \
            f32x2 u_ = t_ * (f32x2){-1.44269502162933349609375f, -1.44269502162933349609375f};                \
            f32x2 e_ = {__builtin_amdgcn_exp2f(u_[0]), __builtin_amdgcn_exp2f(u_[1])};                         \
            e_ = e_ + (f32x2){1.0f, 1.0f};                                                                     \
            t_[0] = __builtin_amdgcn_rcpf(e_[0]);                                                              \
            t_[1] = __builtin_amdgcn_rcpf(e_[1]);                                                              \
        }                                                                                                      \
        const bf16x2 h_ = __builtin_convertvector(t_, bf16x2);                                                 \
        s16x2 p_ = __builtin_bit_cast(s16x2, h_);                                                              \
        if (!SPLIT && ACT == M360_ACT_RELU) p_ = __builtin_elementwise_max(p_, (s16x2){0, 0});                    \
        ab[h][e] = __builtin_bit_cast(unsigned, p_);                                                           \
        if (SPLIT) {                                                                                           \
            const f32x2 r_ = {t_[0] - (float)h_[0], t_[1] - (float)h_[1]};                                     \
            lo[h][e] = __builtin_bit_cast(unsigned, __builtin_convertvector(r_, bf16x2));                      \
        }                                                                                                      \
    } while (0)
                        W16_PACK(0, v[0], v[1], bb[2 * h], 0); W16_PACK(1, v[2], v[3], bb[2 * h], 2);
                        W16_PACK(2, w[0], w[1], bb[2 * h + 1], 0); W16_PACK(3, w[2], w[3], bb[2 * h + 1], 2);
#undef W16_PACK
                    }
                    // Whole-line stores: a lane holds 2 x 16 bytes of ONE row, four lanes 2 x 64 bytes - a store of that shape
                    // (16 rows x 64 B per instruction) runs at 38 GB/s per CU, whole lines (8 rows x 128 B) at 142
                    // (tools/store_rate_probe.hip).  Lanes l15 = 2j, 2j + 1 swap one piece each through DPP: s1 = row 2j
                    // (even lane: its own first piece, odd lane: the even lane's second piece), s2 = row 2j + 1.
                    // Round 4: one wave per SIMD issues ONE instruction of any kind per 4.3-5 cycles (tools/valu_issue_probe.hip,
                    // profiles/r04/valu_issue_probe.jsonl), so the epilogue is priced by its instruction count - and a v_cndmask_b32
                    // whose VCC was NOT written by the vector instruction just before it holds VCC's read path for ~19 cycles (64
                    // selects in a row: 19.4 cycles each, DPP or not, scalar write of VCC before each or not; 4.8 behind a v_cmp; 5.0
                    // on an SGPR pair, which DPP cannot encode): round 3's 8 selects on two masks moved into VCC cost a block of 48
                    // vector instructions ~65 of its 336 cycles.  Hence a v_cmp on the lane's parity in front of EVERY select (8 more
                    // instructions, 34 cycles).  Measured and rejected (profiles/r04/bf16_w16_epilogue_*_REJECTED.jsonl): the
                    // exchange as 4 copies + 8 v_mov_b32_dpp with bank masks (lanes r / r + 8 as partners: no VCC, packing 5.4 k ->
                    // 4.4 k cycles per tile, but stores whose lanes of a quad lie in four lines instead of two: 0.5 k -> 4.0 k), and
                    // the selects issued one per pair of AccVGPR reads of the NEXT block (each in its own statement with its own
                    // s_mov_b64 vcc: 26 more scalar instructions per block than it saves in waiting).
                    // s_nop 0 + the first v_cmp: VALU write -> DPP read of the same register needs 2 wait states.
                    // s_nop after the stores: a store of more than 8 bytes still reads its data registers in the cycle after
                    // issue, and the next instruction here (an AccVGPR read, invisible to the hazard recogniser like the
                    // store) may write them.
#define W16_SEL_(D, OTHER, OWN, CMP)                                                                                             \
    "v_cmp_" CMP "_u32_e32 vcc, 0, %16\n\tv_cndmask_b32_dpp " D ", " OTHER ", " OWN ", vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
#define W16_STORE2_(S1, S2, ROW, IMM, POLICY)                                                                                   \
    asm volatile("global_store_dwordx4 %0, %2, %4 offset:" IMM POLICY "\n\tglobal_store_dwordx4 %1, %3, %4 offset:" IMM POLICY "\n\ts_nop 1" \
                 ::"v"(y_voff), "v"(y_voff + 2u * (unsigned)ldy), "v"(S1), "v"(S2), "s"(ROW) : "memory")
#define W16_STORE2(S1, S2, ROW, IMM)                                                                                            \
    do {                                                                                                                        \
        /* non-temporal: the tile's 128 KiB of output lines otherwise displace the activation tile the 4 column tiles of an XCD share  \
           from its L2 (1263-1270 against 1241-1242 TF, alternating launches on one box; ABL & 128: the plain stores;                 \
           ABL & 4096 / 8192, diagnostics: agent- / system-scope stores (sc1 / sc0 sc1), with and without nt: all 1-2 % slower,      \
           profiles/r04/bf16_w16_store_scope_NOT_ADOPTED.jsonl) */                                                                    \
        if (ABL & 16) asm volatile("" ::"v"(S1), "v"(S2));                                                                      \
        else if ((ABL & 4096) && (ABL & 128)) W16_STORE2_(S1, S2, ROW, IMM, " sc1");                                            \
        else if (ABL & 4096) W16_STORE2_(S1, S2, ROW, IMM, " sc1 nt");                                                          \
        else if ((ABL & 8192) && (ABL & 128)) W16_STORE2_(S1, S2, ROW, IMM, " sc0 sc1");                                        \
        else if (ABL & 8192) W16_STORE2_(S1, S2, ROW, IMM, " sc0 sc1 nt");                                                      \
        else if (CHAIN && ch_j + 1 == ch.layers) W16_STORE2_(S1, S2, ROW, IMM, " nt"); /* the chain's result leaves the die */   \
        else if ((ABL & 128) || M360_W16_PLAIN_STORES) W16_STORE2_(S1, S2, ROW, IMM, "");                                       \
        else W16_STORE2_(S1, S2, ROW, IMM, " nt");                                                                              \
    } while (0)
#define W16_SWAP_STORE(A0, A1, ROW, IMM)                                                                                        \
    do {                                                                                                                        \
        u32x4 s1, s2;                                                                                                           \
        asm volatile("s_nop 0\n\t"                                                                                             \
                     W16_SEL_("%0", "%12", "%8", "eq") W16_SEL_("%1", "%13", "%9", "eq")                                         \
                     W16_SEL_("%2", "%14", "%10", "eq") W16_SEL_("%3", "%15", "%11", "eq")                                       \
                     W16_SEL_("%4", "%8", "%12", "ne") W16_SEL_("%5", "%9", "%13", "ne")                                         \
                     W16_SEL_("%6", "%10", "%14", "ne") W16_SEL_("%7", "%11", "%15", "ne")                                       \
                     : "=&v"(s1[0]), "=&v"(s1[1]), "=&v"(s1[2]), "=&v"(s1[3]), "=&v"(s2[0]), "=&v"(s2[1]), "=&v"(s2[2]), "=&v"(s2[3]) \
                     : "v"(A0[0]), "v"(A0[1]), "v"(A0[2]), "v"(A0[3]), "v"(A1[0]), "v"(A1[1]), "v"(A1[2]), "v"(A1[3]), "v"(lane_odd) \
                     : "vcc");                                                                                                  \
        W16_STORE2(s1, s2, ROW, IMM);                                                                                           \
    } while (0)
                    if (STORE_Y) {
                        // rows 16 i .. of the wave tile: a running pointer (one 64-bit scalar add per block, pinned by the empty
                        // statement below - every instruction of the epilogue costs the lone wave an issue slot); column half P is
                        // the stores' immediate offset
                        if (PAIR) {  // paired rows: the lanes' own pieces are whole lines
                            if (P == 0) W16_STORE2(ab[0], ab[1], rowp, "0"); else W16_STORE2(ab[0], ab[1], rowp, "128");
                            if (SPLIT) { if (P == 0) W16_STORE2(lo[0], lo[1], rowp + Np, "0"); else W16_STORE2(lo[0], lo[1], rowp + Np, "128"); }
                        } else {
                            if (P == 0) W16_SWAP_STORE(ab[0], ab[1], rowp, "0"); else W16_SWAP_STORE(ab[0], ab[1], rowp, "128");
                            if (SPLIT) {  // the second terms: columns [Np, 2 Np)
                                if (P == 0) W16_SWAP_STORE(lo[0], lo[1], rowp + Np, "0"); else W16_SWAP_STORE(lo[0], lo[1], rowp + Np, "128");
                            }
                        }
                        rowp += row_step;
                        asm volatile("" : "+s"(rowp));
                    }
#undef W16_SWAP_STORE
#undef W16_STORE2
#undef W16_STORE2_
#undef W16_SEL_
                    if (HEADS) {  // D[head][row] += over each piece's 32 columns: hi and lo head terms (X3: and the lo activations)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            f32x4 hq = (P == 0 && h == 0) ? (f32x4){0.0f, 0.0f, 0.0f, 0.0f} : hacc[i];
                            hq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hh[h], __builtin_bit_cast(bf16x8, ab[h]), hq, 0, 0, 0);
                            hq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hl[h], __builtin_bit_cast(bf16x8, ab[h]), hq, 0, 0, 0);
                            if (X3) hq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hh[h], __builtin_bit_cast(bf16x8, lo[h]), hq, 0, 0, 0);
                            hacc[i] = hq;
                        }
                        if (P == 1 && g4 == 0 && !(ABL & 16)) {  // the wave tile's 128 columns are in: one store per row
                            float *hp = head_part + (m0 + wm * 128 + 16 * i + l15) * hstride + ((n0 / BN) * 2 + wn) * HEADS;
                            if (HEADS == 4) *reinterpret_cast<f32x4 *>(hp) = hacc[i];
                            else *hp = hacc[i][0];
                        }
                    }
                }
                W16_SB();  // one pair of column pieces at a time
            }
        }
        if (CHAIN) {  // this wave's rows of tile (R, j): counted once its stores are in the L2 - during the next tile's second stage (above),
            ch_pending = true;  // the sequence's last tile behind a drain
            if (tile_id + 1 >= ch_T) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(ch.done + (long)(blockIdx.x & 7) * ch_T + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (ABL & 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // diagnostics: drain the stores (and everything else) inside the stamped epilogue
        if (STAMP) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(e1)::"memory"); te += e1 - e0; nsl += nstages; }
        have_prev = !(ABL & 32);
    }
    if (STAMP) asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(mt1), "=s"(rt1)::"memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA of this wave may land after the workgroup is gone
    if (CHAIN && ch_gave_up && lane == 0) {  // a wait of this wave ran out: the launch's rows are not to be trusted (the gated re-run redoes them)
        __hip_atomic_fetch_or(&ch.status->error, kChainErrTimeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&ch.status->timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef M360_DIAG
    if (STAMP && tid == 0 && blockIdx.x < 256) {
        g_w16_stamps[blockIdx.x * 4 + 0] = mt1 - mt0;
        g_w16_stamps[blockIdx.x * 4 + 1] = rt1 - rt0;
        g_w16_stamps[blockIdx.x * 4 + 2] = nsl;
        g_w16_stamps[blockIdx.x * 4 + 3] = te;
    }
#endif
#undef W16_CUR_ADV
#undef W16_PIECE_X
#undef W16_PIECE_W
#undef W16_ADV_X
#undef W16_ADV_W
#undef W16_DMA_X
#undef W16_DMA_W
#undef W16X_ADV_XL
#undef W16X_ADV_XH
#undef W16X_ADV_WH
#undef W16X_ADV_WL
#undef W16X_DMA_XL
#undef W16X_DMA_XH
#undef W16X_DMA_WH
#undef W16X_DMA_WL
#undef W16_RD
#undef W16_SB
#undef W16_MFMA
#undef W16_MFMA_Z
#undef W16_TIE_HI
#undef W16_WAIT_NEXT
#undef W16_WAIT_HI
#undef W16_VMCNT
#undef W16_BAR
#undef W16_BARRIER_E0
#undef W16_BARRIER_M1
#undef W16_BARRIER_E1
}

}  // namespace w16
}  // namespace m360
