// What does ONE wave per SIMD pay per vector instruction of the bf16 ring kernel's epilogue?
// The epilogue (m360_linear_bf16_w16.hip.h) is 768 vector instructions per lane and tile - v_accvgpr_read_b32, v_pk_add_f32,
// v_cvt_pk_bf16_f32, v_pk_max_i16, v_cndmask_b32_dpp - and measures 5.4 k cycles: 7 per instruction where a 16-lane SIMD needs 4.
// This probe times blocks of 64 instructions of each kind (independent registers, and dependent chains), the epilogue's block
// as the compiler emits it, and reorderings of it, with one and with two waves per SIMD.  s_memtime around `iters` repetitions.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_issue_probe.hip -o tools/valu_issue_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// registers the blocks use: v16..v95, a0..a63 (all clobbered)
#define CLOB "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35", \
    "v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57", \
    "v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79", \
    "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95", \
    "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23", \
    "a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45", \
    "a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63", "vcc", "memory"

#define REPT(N, BODY) ".set pi, 0\n\t.rept " #N "\n\t" BODY "\n\t.set pi, pi+1\n\t.endr\n\t"

// one (activation block, column pair) of the epilogue: two halves of 8 reads + 4 adds + 4 conversions + 4 max, then 8 selects
#define HALF_AS_EMITTED(A0)                                                                                                     \
    REPT(8, "v_accvgpr_read_b32 v[16+pi], a[" #A0 "+pi]")                                                                        \
    REPT(4, "v_pk_add_f32 v[16+2*pi:17+2*pi], v[16+2*pi:17+2*pi], v[80+2*pi:81+2*pi]")                                            \
    REPT(4, "v_cvt_pk_bf16_f32 v[16+2*pi], v[16+2*pi], v[17+2*pi]")
#define SELECTS                                                                                                                 \
    "s_nop 1\n\ts_mov_b64 vcc, %0\n\ts_nop 1\n\t"                                                                               \
    REPT(4, "v_cndmask_b32_dpp v[40+pi], v[32+pi], v[24+pi], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")                  \
    "s_mov_b64 vcc, %1\n\ts_nop 1\n\t"                                                                                           \
    REPT(4, "v_cndmask_b32_dpp v[44+pi], v[24+pi], v[32+pi], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")

template <int T>
__device__ __forceinline__ void block() {
    const unsigned long long m0 = 0x5555555555555555ull, m1 = 0xAAAAAAAAAAAAAAAAull;
    if (T == 0) asm volatile(REPT(64, "v_mov_b32 v[16+pi], v[80]") ::"s"(m0), "s"(m1) : CLOB);
    if (T == 1) asm volatile(REPT(64, "v_accvgpr_read_b32 v[16+pi], a[pi]") ::"s"(m0), "s"(m1) : CLOB);
    if (T == 2) asm volatile(REPT(32, "v_pk_add_f32 v[16+2*pi:17+2*pi], v[80:81], v[82:83]") REPT(32, "v_pk_add_f32 v[16+2*pi:17+2*pi], v[80:81], v[82:83]") ::"s"(m0), "s"(m1) : CLOB);
    if (T == 3) asm volatile(REPT(64, "v_pk_add_f32 v[16:17], v[16:17], v[82:83]") ::"s"(m0), "s"(m1) : CLOB);
    if (T == 4) asm volatile(REPT(64, "v_cvt_pk_bf16_f32 v[16+pi], v[80], v[81]") ::"s"(m0), "s"(m1) : CLOB);
    if (T == 5) asm volatile(REPT(64, "v_pk_max_i16 v[16+pi], v[80], 0") ::"s"(m0), "s"(m1) : CLOB);
    if (T == 6) asm volatile("s_mov_b64 vcc, %0\n\ts_nop 1\n\t" REPT(64, "v_cndmask_b32_dpp v[16+pi], v[80], v[81], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") ::"s"(m0), "s"(m1) : CLOB);
    if (T == 7) asm volatile(REPT(64, "v_add_f32 v16, v16, v80") ::"s"(m0), "s"(m1) : CLOB);
    if (T == 8) asm volatile(REPT(64, "v_add_f32 v[16+pi], v80, v81") ::"s"(m0), "s"(m1) : CLOB);
    // 9: the block as the compiler emits it (48 vector instructions)
    if (T == 9) asm volatile(HALF_AS_EMITTED(0) REPT(4, "v_pk_max_i16 v[24+pi], v[16+2*pi], 0") HALF_AS_EMITTED(8) REPT(4, "v_pk_max_i16 v[32+pi], v[16+2*pi], 0") SELECTS ::"s"(m0), "s"(m1) : CLOB);
    // 10: the same with v_mov_b32 in place of the AccVGPR reads
    if (T == 10) asm volatile(REPT(8, "v_mov_b32 v[16+pi], v[84+pi]") REPT(4, "v_pk_add_f32 v[16+2*pi:17+2*pi], v[16+2*pi:17+2*pi], v[80+2*pi:81+2*pi]") REPT(4, "v_cvt_pk_bf16_f32 v[16+2*pi], v[16+2*pi], v[17+2*pi]") REPT(4, "v_pk_max_i16 v[24+pi], v[16+2*pi], 0")
                              REPT(8, "v_mov_b32 v[16+pi], v[84+pi]") REPT(4, "v_pk_add_f32 v[16+2*pi:17+2*pi], v[16+2*pi:17+2*pi], v[80+2*pi:81+2*pi]") REPT(4, "v_cvt_pk_bf16_f32 v[16+2*pi], v[16+2*pi], v[17+2*pi]") REPT(4, "v_pk_max_i16 v[32+pi], v[16+2*pi], 0") SELECTS ::"s"(m0), "s"(m1) : CLOB);
    // 11: the block without the selects (40 vector instructions)
    if (T == 11) asm volatile(HALF_AS_EMITTED(0) REPT(4, "v_pk_max_i16 v[24+pi], v[16+2*pi], 0") HALF_AS_EMITTED(8) REPT(4, "v_pk_max_i16 v[32+pi], v[16+2*pi], 0") ::"s"(m0), "s"(m1) : CLOB);
    // 12: 16 reads first (two register sets), then 8 adds, 8 conversions, 8 max: every instruction >= 8 instructions behind its producer
    if (T == 12) asm volatile(REPT(16, "v_accvgpr_read_b32 v[48+pi], a[pi]") REPT(8, "v_pk_add_f32 v[48+2*pi:49+2*pi], v[48+2*pi:49+2*pi], v[80+2*pi:81+2*pi]") REPT(8, "v_cvt_pk_bf16_f32 v[48+2*pi], v[48+2*pi], v[49+2*pi]") REPT(8, "v_pk_max_i16 v[24+pi], v[48+2*pi], 0") SELECTS ::"s"(m0), "s"(m1) : CLOB);
    // 13: reads only as in the block (16), 14: adds only (8 dependent on nothing recent), 15: conversions only
    if (T == 13) asm volatile(REPT(16, "v_accvgpr_read_b32 v[48+pi], a[pi]") ::"s"(m0), "s"(m1) : CLOB);
    // 14: AccVGPR reads alternating with packed max (does the read share an issue port?)
    if (T == 14) asm volatile(REPT(32, "v_accvgpr_read_b32 v[16+pi], a[pi]\n\tv_pk_max_i16 v[48+pi], v[80], 0") ::"s"(m0), "s"(m1) : CLOB);
    // 15-19, 21, 22: is it the DPP path, the select, or the VCC read?
    if (T == 15) asm volatile(REPT(64, "v_mov_b32_dpp v[16+pi], v[80] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") ::"s"(m0), "s"(m1) : CLOB);
    if (T == 16) asm volatile(REPT(64, "v_mov_b32_dpp v[16+pi], v[80] row_ror:8 row_mask:0xf bank_mask:0xc") ::"s"(m0), "s"(m1) : CLOB);
    if (T == 17) asm volatile("s_mov_b64 vcc, %0\n\ts_nop 1\n\t" REPT(64, "v_cndmask_b32_e32 v[16+pi], v[80], v[81], vcc") ::"s"(m0), "s"(m1) : CLOB);
    if (T == 18) asm volatile(REPT(64, "v_cndmask_b32_e64 v[16+pi], v[80], v[81], %0") ::"s"(m0), "s"(m1) : CLOB);
    if (T == 19) asm volatile(REPT(64, "v_add_u32_dpp v[16+pi], v[80], v[81] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") ::"s"(m0), "s"(m1) : CLOB);
    // 20: the block with the exchange as 4 copies + 8 masked DPP moves (lanes l and l ^ 8 of a row of 16 as partners)
    if (T == 20) asm volatile(HALF_AS_EMITTED(0) REPT(4, "v_pk_max_i16 v[24+pi], v[16+2*pi], 0") HALF_AS_EMITTED(8) REPT(4, "v_pk_max_i16 v[32+pi], v[16+2*pi], 0")
                              REPT(4, "v_mov_b32 v[40+pi], v[32+pi]") "s_nop 1\n\t"
                              REPT(4, "v_mov_b32_dpp v[32+pi], v[24+pi] row_ror:8 row_mask:0xf bank_mask:0x3")
                              REPT(4, "v_mov_b32_dpp v[24+pi], v[40+pi] row_ror:8 row_mask:0xf bank_mask:0xc") ::"s"(m0), "s"(m1) : CLOB);
    if (T == 21) asm volatile(REPT(64, "v_mov_b32_dpp v[16+pi], v[80] row_shr:1 row_mask:0xf bank_mask:0xf") ::"s"(m0), "s"(m1) : CLOB);
    if (T == 22) asm volatile("s_mov_b64 vcc, %0\n\ts_nop 1\n\t" REPT(64, "v_cndmask_b32_dpp v[16+pi], v[80], v[81], vcc row_ror:8 row_mask:0xf bank_mask:0xf") ::"s"(m0), "s"(m1) : CLOB);
    // 23, 24: selects spread between other vector instructions (1 in 4, 1 in 8)
    if (T == 23) asm volatile("s_mov_b64 vcc, %0\n\ts_nop 1\n\t" REPT(16, "v_cndmask_b32_dpp v[16+pi], v[80], v[81], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_pk_max_i16 v[32+pi], v[82], 0\n\tv_pk_max_i16 v[48+pi], v[83], 0\n\tv_pk_max_i16 v[64+pi], v[84], 0") ::"s"(m0), "s"(m1) : CLOB);
    if (T == 24) asm volatile("s_mov_b64 vcc, %0\n\ts_nop 1\n\t" REPT(8, "v_cndmask_b32_dpp v[16+pi], v[80], v[81], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_pk_max_i16 v[24+pi], v[82], 0\n\tv_pk_max_i16 v[32+pi], v[83], 0\n\tv_pk_max_i16 v[40+pi], v[84], 0\n\tv_pk_max_i16 v[48+pi], v[85], 0\n\tv_pk_max_i16 v[56+pi], v[86], 0\n\tv_pk_max_i16 v[64+pi], v[87], 0\n\tv_pk_max_i16 v[72+pi], v[88], 0") ::"s"(m0), "s"(m1) : CLOB);
    // 25: the block with the selects of the PREVIOUS block spread over it: one after every 5th of the 40 other instructions
    //     (both masks live in SGPR pairs: v_cndmask_b32_dpp takes VCC only, so the two masks alternate through VCC every 4 selects)
    if (T == 25) asm volatile("s_mov_b64 vcc, %0\n\t"
                              REPT(4, "v_accvgpr_read_b32 v[16+2*pi], a[2*pi]\n\tv_accvgpr_read_b32 v[17+2*pi], a[2*pi+1]\n\tv_cndmask_b32_dpp v[40+pi], v[60+pi], v[64+pi], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                                      "v_pk_add_f32 v[16+2*pi:17+2*pi], v[16+2*pi:17+2*pi], v[80+2*pi:81+2*pi]\n\tv_cvt_pk_bf16_f32 v[16+2*pi], v[16+2*pi], v[17+2*pi]\n\tv_pk_max_i16 v[24+pi], v[16+2*pi], 0")
                              "s_mov_b64 vcc, %1\n\t"
                              REPT(4, "v_accvgpr_read_b32 v[48+2*pi], a[8+2*pi]\n\tv_accvgpr_read_b32 v[49+2*pi], a[9+2*pi]\n\tv_cndmask_b32_dpp v[44+pi], v[64+pi], v[60+pi], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                                      "v_pk_add_f32 v[48+2*pi:49+2*pi], v[48+2*pi:49+2*pi], v[80+2*pi:81+2*pi]\n\tv_cvt_pk_bf16_f32 v[48+2*pi], v[48+2*pi], v[49+2*pi]\n\tv_pk_max_i16 v[32+pi], v[48+2*pi], 0")
                              REPT(4, "v_mov_b32 v[60+pi], v[24+pi]\n\tv_mov_b32 v[64+pi], v[32+pi]") ::"s"(m0), "s"(m1) : CLOB);
    // 26-30: the compiler's own compare / select patterns: does every reader of VCC pay, and does an SGPR pair avoid it?
    if (T == 26) asm volatile(REPT(32, "v_cmp_lt_f32_e32 vcc, v[80], v[81+pi]\n\tv_cndmask_b32_e32 v[16+pi], v[82], v[83], vcc") ::"s"(m0), "s"(m1) : CLOB);
    if (T == 27) asm volatile(REPT(32, "v_cmp_lt_f32_e64 s[40:41], v[80], v[81+pi]\n\tv_cndmask_b32_e64 v[16+pi], v[82], v[83], s[40:41]") ::"s"(m0), "s"(m1) : "s40", "s41", CLOB);
    if (T == 28) asm volatile(REPT(64, "v_cmp_lt_f32_e32 vcc, v[80], v[81]") ::"s"(m0), "s"(m1) : CLOB);
    if (T == 29) asm volatile(REPT(32, "v_add_co_u32_e32 v[16+pi], vcc, v[80], v[81]\n\tv_addc_co_u32_e32 v[48+pi], vcc, v[82], v[83], vcc") ::"s"(m0), "s"(m1) : CLOB);
    if (T == 30) asm volatile(REPT(32, "v_add_co_u32_e64 v[16+pi], s[40:41], v[80], v[81]\n\tv_addc_co_u32_e64 v[48+pi], s[40:41], v[82], v[83], s[40:41]") ::"s"(m0), "s"(m1) : "s40", "s41", CLOB);
    // 32, 33: does a SCALAR write of VCC before every select make its read fast, as a v_cmp does (26)?
    if (T == 32) asm volatile(REPT(32, "s_mov_b64 vcc, %0\n\tv_cndmask_b32_dpp v[16+pi], v[80], v[81], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                                       "s_mov_b64 vcc, %1\n\tv_cndmask_b32_dpp v[48+pi], v[81], v[80], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") ::"s"(m0), "s"(m1) : CLOB);
    if (T == 33) asm volatile(HALF_AS_EMITTED(0) REPT(4, "v_pk_max_i16 v[24+pi], v[16+2*pi], 0") HALF_AS_EMITTED(8) REPT(4, "v_pk_max_i16 v[32+pi], v[16+2*pi], 0") "s_nop 1\n\t"
                              REPT(4, "s_mov_b64 vcc, %0\n\tv_cndmask_b32_dpp v[40+pi], v[32+pi], v[24+pi], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                                      "s_mov_b64 vcc, %1\n\tv_cndmask_b32_dpp v[44+pi], v[24+pi], v[32+pi], vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") ::"s"(m0), "s"(m1) : CLOB);
    // 34-37: the sigmoid of the fused-heads layers: do v_exp_f32 / v_rcp_f32 (quarter rate) overlap with other vector instructions?
    if (T == 34) asm volatile(REPT(64, "v_exp_f32_e32 v[16+pi], v[80]") ::"s"(m0), "s"(m1) : CLOB);
    if (T == 35) asm volatile(REPT(16, "v_exp_f32_e32 v[16+pi], v[80]\n\tv_pk_max_i16 v[32+pi], v[82], 0\n\tv_pk_max_i16 v[48+pi], v[83], 0\n\tv_pk_max_i16 v[64+pi], v[84], 0") ::"s"(m0), "s"(m1) : CLOB);
    if (T == 36) asm volatile(REPT(32, "v_exp_f32_e32 v[16+pi], v[80]\n\tv_pk_max_i16 v[48+pi], v[82], 0") ::"s"(m0), "s"(m1) : CLOB);
    if (T == 37) asm volatile(REPT(32, "v_exp_f32_e32 v[16+pi], v[80]\n\tv_rcp_f32_e32 v[48+pi], v[81]") ::"s"(m0), "s"(m1) : CLOB);
    // 31: one compare into VCC, then 3 selects on it (the 4-wide vector selects the compiler emits), x 16
    if (T == 31) asm volatile(REPT(16, "v_cmp_lt_f32_e32 vcc, v[80], v[81+pi]\n\tv_cndmask_b32_e32 v[16+pi], v[82], v[83], vcc\n\tv_cndmask_b32_e32 v[32+pi], v[82], v[83], vcc\n\tv_cndmask_b32_e32 v[48+pi], v[82], v[83], vcc") ::"s"(m0), "s"(m1) : CLOB);
}

constexpr int kTests = 38;
__host__ __device__ constexpr int valu_in(int t) { return t == 9 || t == 10 || t == 33 ? 48 : t == 11 ? 40 : t == 12 ? 48 : t == 13 ? 16 : t == 20 ? 52 : t == 25 ? 56 : 64; }

template <int T>
__global__ __launch_bounds__(512) void probe_kernel(int iters, unsigned long long *cyc) {
    unsigned long long c0, c1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0)::"memory");
    for (int it = 0; it < iters; ++it) block<T>();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1)::"memory");
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = c1 - c0;
}

template <int T>
void run(int cus, int iters, unsigned long long *cyc, unsigned long long *h) {
    static const char *names[kTests] = {"v_mov_b32, independent", "v_accvgpr_read_b32, independent", "v_pk_add_f32, independent", "v_pk_add_f32, dependent chain",
        "v_cvt_pk_bf16_f32, independent", "v_pk_max_i16, independent", "v_cndmask_b32_dpp quad_perm, independent", "v_add_f32, dependent chain",
        "v_add_f32, independent", "epilogue block as emitted (48 vector instructions)", "the block with v_mov_b32 for the AccVGPR reads",
        "the block without its 8 selects (40)", "the block reordered: 16 reads, 8 adds, 8 conversions, 8 max, 8 selects", "16 AccVGPR reads alone", "AccVGPR read / v_pk_max_i16 alternating",
        "v_mov_b32_dpp quad_perm, independent", "v_mov_b32_dpp row_ror:8 bank_mask:0xc, independent", "v_cndmask_b32_e32 (VCC, no DPP), independent", "v_cndmask_b32_e64 (SGPR pair, no DPP), independent",
        "v_add_u32_dpp quad_perm, independent", "the block with 4 copies + 8 masked DPP moves for the 8 selects (52)", "v_mov_b32_dpp row_shr:1, independent", "v_cndmask_b32_dpp row_ror:8, independent",
        "1 select + 3 v_pk_max_i16, x 16", "1 select + 7 v_pk_max_i16, x 8", "the block with the previous block's selects spread over it, + 8 copies (56)",
        "v_cmp -> VCC + v_cndmask on VCC, x 32", "v_cmp -> SGPR pair + v_cndmask on it, x 32", "v_cmp -> VCC, independent", "v_add_co / v_addc_co through VCC, x 32", "v_add_co / v_addc_co through an SGPR pair, x 32",
        "v_cmp -> VCC + 3 v_cndmask on VCC, x 16", "s_mov_b64 vcc before EVERY select (two masks alternating), x 64", "the block with s_mov_b64 vcc before every select (48)",
        "v_exp_f32, independent", "1 v_exp_f32 + 3 v_pk_max_i16, x 16", "v_exp_f32 / v_pk_max_i16 alternating", "v_exp_f32 / v_rcp_f32 alternating"};
    for (int waves = 4; waves <= 8; waves += 4) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(probe_kernel<T>, dim3(cus), dim3(64 * waves), 0, 0, iters, cyc);
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipMemcpy(h, cyc, (size_t)cus * 8 * 8, hipMemcpyDeviceToHost));
        double c = 0;
        for (int b = 0; b < cus; ++b) for (int w = 0; w < waves; ++w) c += (double)h[b * 8 + w];
        c /= (double)cus * waves * iters;
        printf("{\"test\": %d, \"what\": \"%s\", \"waves_per_simd\": %d, \"vector_instructions_per_block\": %d, \"cycles_per_block\": %.1f, \"cycles_per_vector_instruction\": %.2f}\n",
               T, names[T], waves / 4, valu_in(T), c, c / valu_in(T));
    }
}

int main() {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int iters = 4000;
    unsigned long long *cyc, *h = (unsigned long long *)malloc((size_t)cus * 8 * 8);
    CHECK(hipMalloc(&cyc, (size_t)cus * 8 * 8));
    run<0>(cus, iters, cyc, h); run<1>(cus, iters, cyc, h); run<2>(cus, iters, cyc, h); run<3>(cus, iters, cyc, h); run<4>(cus, iters, cyc, h);
    run<5>(cus, iters, cyc, h); run<6>(cus, iters, cyc, h); run<7>(cus, iters, cyc, h); run<8>(cus, iters, cyc, h); run<9>(cus, iters, cyc, h);
    run<10>(cus, iters, cyc, h); run<11>(cus, iters, cyc, h); run<12>(cus, iters, cyc, h); run<13>(cus, iters, cyc, h); run<14>(cus, iters, cyc, h);
    run<15>(cus, iters, cyc, h); run<16>(cus, iters, cyc, h); run<17>(cus, iters, cyc, h); run<18>(cus, iters, cyc, h); run<19>(cus, iters, cyc, h); run<20>(cus, iters, cyc, h);
    run<21>(cus, iters, cyc, h); run<22>(cus, iters, cyc, h); run<23>(cus, iters, cyc, h); run<24>(cus, iters, cyc, h); run<25>(cus, iters, cyc, h);
    run<26>(cus, iters, cyc, h); run<27>(cus, iters, cyc, h); run<28>(cus, iters, cyc, h); run<29>(cus, iters, cyc, h); run<30>(cus, iters, cyc, h); run<31>(cus, iters, cyc, h); run<32>(cus, iters, cyc, h); run<33>(cus, iters, cyc, h);
    run<34>(cus, iters, cyc, h); run<35>(cus, iters, cyc, h); run<36>(cus, iters, cyc, h); run<37>(cus, iters, cyc, h);
    return 0;
}
