#!/usr/bin/env python3
"""Generates mipnerf360_amd/csrc/m360_linear_bf16_w16_gen.inc: the stage bodies of the one-wave-per-SIMD bf16 linear kernel with
128 x 128 wave tiles on v_mfma_f32_16x16x32_bf16 (m360_linear_bf16_w16.hip.h), every non-matrix instruction in ONE MFMA gap.

A stage = 64 bf16 of the contraction (128-byte LDS rows: an LDS-DMA piece is 8 rows x 128 B = eight WHOLE lines - pieces of 16 rows
x 64 B, i.e. 32-deep stages, stream at 42 instead of 65 GB/s per CU: profiles/r03/dma_piece_shape_probe.jsonl) = TWO k-steps of the
MFMA, each 8 x 8 blocks of 16 x 16 = 64 MFMAs of 16 cycles in two halves (activation blocks 0-3 "lo", then 4-7 "hi"; all 8 weight
blocks).  The LDS holds two stages (64 KiB each: 256 activation + 256 weight rows).  65 GB/s per CU is one 1-KiB piece per ~31
cycles and CU: the 64 pieces of a stage need the whole 2048 matrix cycles of a stage, so the pieces are issued at a UNIFORM rate -
4 per wave and half k-step, one per 8th MFMA gap - and each region of a buffer is refilled as soon as its last reader is done:
  k-step 0 (weight fragment set fw0) of stage s in buffer B
    half 0   MFMAs fx[0..3] x fw0      gaps 0,2,4,6      ds_read_b128 of fx[4..7] of (s, 0)                          (buffer B)
                                       gaps 1,9,17,25    activation pieces "lo" of stage s+1 -> buffer B^1 (free since end of (s-1, 0))
    wait fx[4..7]
    half 1   MFMAs fx[4..7] x fw0      gaps 0,2,..,22    ds_read_b128 of fw1[0..7], fx[0..3] of (s, 1)               (buffer B)
                                       gaps 1,9,17,25    activation pieces "hi" of stage s+1 -> buffer B^1 (free since mid (s-1, 1))
    wait fw1, fx[0..3] | s_barrier E0: every wave has read the weight and "lo" rows of buffer B for the last time
  k-step 1 (fw1)
    half 0   MFMAs fx[0..3] x fw1      gaps 0,2,4,6      fx[4..7] of (s, 1)
                                       gaps 1,9,17,25    weight pieces 0-3 of stage s+2 -> buffer B
    wait fx[4..7] | vmcnt: weight + "lo" pieces of stage s+1 have landed | s_barrier M1 (the "hi" rows of B are free now)
    half 1   MFMAs fx[4..7] x fw1      gaps 0,2,..,22    fw0[0..7], fx[0..3] of (s+1, 0)                             (buffer B^1)
                                       gaps 1,9,17,25    weight pieces 4-7 of stage s+2 -> buffer B
    wait fw0, fx[0..3] | vmcnt: "hi" pieces of stage s+1 have landed | s_barrier E1
Every piece is issued >= 2 half k-steps (1024 matrix cycles) before the barrier that needs it.  The counted vmcnt of M1 and E1 comes
from a simulation of the issue order (`simulate()`); the stores of the previous tile (STORES per lane, one per k-step of the first
STORE_STAGES stages of the next tile) are counted like pieces: those younger than the awaited pieces may stay in flight.
Macros: W16_STAGE<B>() generic, W16_STAGE0Z() first stage of a tile (accumulators restart from 0 through the C operand); with
STORE_STAGES > 0 also W16_STAGE<B>[Z]S<n>(): stage n of a tile carrying stores of the previous tile (guarded by `have_prev`).
"""
import os

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mipnerf360_amd", "csrc", "m360_linear_bf16_w16_gen.inc")
STORES = 32             # 16-byte stores per lane and tile (8 activation blocks x 4 column pieces)
STORE_STAGES = 0        # stages of the NEXT tile that carry the stores of a tile (one per k-step, in its second half, gap 5).
                        # 0 = all stores in the epilogue, which is what the product does: vmcnt retires in order and a store takes
                        # ~2.2 k cycles to complete, so a counted wait for a piece issued after a store cannot complete before the
                        # store has.  Measured per 32 deep (1024 matrix cycles): no stores 1150-1170 cycles; 8 stores per stage over
                        # 4 stages 1760; 2 per stage over 16 stages 1836 (every k-step waits ~670 cycles for its store); all 32 in
                        # the epilogue: one wait of ~5 k cycles per tile (= 170 per 32 deep).


def half_ops(s, r, kk, half):
    """vector-memory operations of one wave in half `half` of k-step kk of stage s (stage r of its tile), in issue order"""
    what = {(0, 0): ("XL", s + 1), (0, 1): ("XH", s + 1), (1, 0): ("WA", s + 2), (1, 1): ("WB", s + 2)}[(kk, half)]
    ops = []
    for g in range(32):
        if g % 8 == 1:
            ops.append(what)
        if r < STORE_STAGES and half == 1 and g == 5:
            ops.append(("S", None))
    return ops


def simulate(nst=24):
    """-> {(r, 'M1' | 'E1'): (pieces, stores) that may stay outstanding at that barrier of stage r of a tile (steady state)}"""
    ops, mark = [], {}
    for s in range(3 * nst):
        r = s % nst
        for kk in range(2):
            for half in range(2):
                ops += half_ops(s, r, kk, half)
                if (kk, half) == (1, 0):
                    mark[(s, "M1")] = len(ops)
                if (kk, half) == (1, 1):
                    mark[(s, "E1")] = len(ops)
    res = {}
    for s in range(nst, 2 * nst):
        for bar, needs in (("M1", ("XL", "WA", "WB")), ("E1", ("XH",))):
            upto = mark[(s, bar)]
            need = max(i for i, o in enumerate(ops[:upto]) if o[0] in needs and o[1] == s + 1)
            younger = ops[need + 1:upto]
            res[(s % nst, bar)] = (sum(1 for o in younger if o[0] != "S"), sum(1 for o in younger if o[0] == "S"))
    return res


def w_off(jb):
    # weight block jb = LDS rows 32 (jb >> 1) + 4 (jb & 1) + {8 a + b}: 128-byte rows
    return 4096 * (jb >> 1) + 512 * (jb & 1)


def stage(B, kind, vm, r=None):
    """r: stage of the tile when it carries stores of the previous tile, else None"""
    name = f"W16_STAGE{B}{kind}"
    L = [f"#define {name}() do {{"]
    nb = 1 - B
    for kk in range(2):
        cur, nxt = kk, 1 - kk
        for half in range(2):
            if half == 0:
                reads = [f"W16_RD(fx[{i}], xa{kk}{B}, {i * 2048})" for i in range(4, 8)]
                read_gaps = {0: 0, 2: 1, 4: 2, 6: 3}
            else:
                rb, rk = (B, 1) if kk == 0 else (nb, 0)   # fragments of the next k-step: (s, 1) from B, or (s + 1, 0) from B^1
                reads = [f"W16_RD(fw{nxt}[{j}], wa{rk}{rb}, {w_off(j)})" for j in range(8)] + \
                        [f"W16_RD(fx[{i}], xa{rk}{rb}, {i * 2048})" for i in range(4)]
                read_gaps = {2 * k: k for k in range(12)}
            if kk == 0:
                dma = {1 + 8 * q: f"W16_DMA_X({nb}, {4 * half + q})" for q in range(4)}
            else:
                dma = {1 + 8 * q: f"W16_DMA_W({B}, {4 * half + q})" for q in range(4)}
            store_gaps = {5: 2 * r + kk} if (r is not None and half == 1) else {}
            for m in range(32):
                ib, jb = 4 * half + m // 8, m % 8
                z = "_Z" if (kind.startswith("Z") and kk == 0) else ""
                L.append(f"    W16_MFMA{z}(acc[{ib}][{jb}], fw{cur}[{jb}], fx[{ib}]);")
                if m in read_gaps:
                    L.append(f"    {reads[read_gaps[m]]};")
                if m in dma:
                    L.append(f"    {dma[m]};")
                if m in store_gaps:
                    L.append(f"    W16_STORE({store_gaps[m]});")
                L.append("    W16_SB();")
            if half == 0:
                if kk == 0:
                    L.append("    W16_WAIT_HI();")
                else:
                    p, st = vm[(r if r is not None else STORE_STAGES, "M1")]
                    L.append(f"    W16_BARRIER_M1({p}, {p + st});")
                L.append("    W16_SB();")
            else:
                if kk == 0:
                    L.append("    W16_ADV_X();")
                    L.append(f"    W16_WAIT_NEXT(fw{nxt});")
                    L.append("    W16_BARRIER_E0();")
                else:
                    L.append("    W16_ADV_W();")
                    L.append(f"    W16_WAIT_NEXT(fw{nxt});")
                    p, st = vm[(r if r is not None else STORE_STAGES, "E1")]
                    L.append(f"    W16_BARRIER_E1({p}, {p + st});")
                L.append("    W16_SB();")
    L.append("} while (0)")
    return " \\\n".join(L) + "\n"


def main():
    vm = simulate()
    assert all(vm[(r, b)] == vm[(STORE_STAGES, b)] and vm[(r, b)][1] == 0 for r in range(STORE_STAGES, 24) for b in ("M1", "E1")), vm
    out = ["// GENERATED by tools/gen_w16_slab.py - do not edit.  Stage bodies of m360_linear_bf16_w16.hip.h.\n",
           f"#define W16_STORE_STAGES {STORE_STAGES}\n"]
    for n in range(STORE_STAGES):   # stages 0..15 of a tile: stores of the previous tile ride along
        out.append(stage(n % 2, ("Z" if n == 0 else "") + f"S{n}", vm, r=n))
    assert vm[(STORE_STAGES, "M1")][1] == 0 and vm[(STORE_STAGES, "E1")][1] == 0, vm   # no store younger than what that stage awaits
    if STORE_STAGES == 0:
        out.append(stage(0, "Z", vm))   # first stage of a tile: accumulators restart from 0
    out.append(stage(0, "", vm))
    out.append(stage(1, "", vm))
    with open(OUT, "w") as f:
        f.write("\n".join(out))
    print("wrote", OUT, sum(len(o) for o in out), "bytes; barrier counts (pieces, stores) by stage of a tile:",
          {k: v for k, v in sorted(vm.items()) if k[0] in (0, 1, STORE_STAGES - 1, STORE_STAGES)})


if __name__ == "__main__":
    main()
