#!/usr/bin/env python3
"""Generates mipnerf360_amd/csrc/m360_linear_hd_gen.inc: the K-step bodies of the half-tile / double-accumulator fp32 linear
kernel (m360_linear_hd.hip.h) with every non-matrix instruction assigned to ONE MFMA gap.

A K-step = 4 K-groups x 32 MFMAs (v_mfma_f32_32x32x2_f32, 64 cycles each) on a 64 x 128 wave tile (TM = 2, TN = 4).
Fillers per K-step: 24 ds_read_b128 of the next group's fragments, 12 LDS-DMA pieces of the next K-step (one every
P_STRIDE gaps - back to back the four waves of a CU would keep the texture addresser 100 % busy and the issue stalls show up
as MFMA bubbles; in the epilogue K-steps all of them inside group 0, so that everything issued later is younger than
them and a counted vmcnt can leave the epilogue's stores in flight), one
barrier between groups 2 and 3 - and, in the first two K-steps of a tile, the EPILOGUE OF THE PREVIOUS TILE: its 8 blocks
of 32 x 32 accumulators (the other accumulator set) go through a wave-private LDS staging area, one block per K-group
(16 ds_write_b32 + 4 ds_read_b128 + the 24 vector instructions of bias and activation in ONE gap + 4 16-byte stores).

Macros emitted (S = accumulator set of the running tile, P = set of the previous tile):
  HD_KSTEP_F(S, P)  first K-step of a tile: the accumulators of S start from 0 through the C operand; epilogue blocks 0-3 of P
  HD_KSTEP_S(S, P)  second K-step: epilogue blocks 4-7 of P
  HD_KSTEP_P(S, P)  every further K-step: no epilogue
  HD_KSTEP_F0 / S1 / P0 / P1: the same for layers with an even number of K-steps - K-step kt then always works on LDS stage
  kt & 1, the stage offset moves into the immediates of the ds_read_b128 and the 8 vector address updates per K-step (32-40
  cycles of matrix time: vector instructions are not hidden behind MFMAs) disappear
The epilogue fillers are guarded by the wave-uniform `have_prev` (the first tile of a workgroup has nothing to store): ONE
straight-line MFMA stream per K-step, so the 256 accumulators never meet a control-flow join inside a tile.
"""
import os

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mipnerf360_amd", "csrc", "m360_linear_hd_gen.inc")


EV_MODE = "four" # gaps the bias + activation instructions of a block are spread over
P_STRIDE = 4   # MFMA gaps between two LDS-DMA pieces in the K-steps without epilogue (12 pieces: groups 0-1)
FS_STRIDE = 2  # ... in the two epilogue K-steps (all 12 inside group 0, before the first epilogue stores)


STAGE_BYTES = 49152  # one LDS stage (A + B tile of a K-step)


def group(lines, g, first, epi_block, dma, buf=None):
    cur, nxt = ("0", "1") if g % 2 == 0 else ("1", "0")   # fragment sets: G0 F0, G1 F1, G2 F0, G3 F1
    # addresses of the reads issued in this group (fragments of the NEXT group; group 3: group 0 of the next K-step)
    if buf is None:   # LDS stage known at run time only: per-K-step address registers (8 vector adds per K-step)
        ra, rb = {0: ("a1", "b1"), 1: ("a2", "b2"), 2: ("a3", "b3"), 3: ("a0n", "b0n")}[g]
        off = 0
    else:             # static stage (even number of K-steps): the stage offset is part of the instruction's immediate
        gi = (g + 1) % 4
        ra, rb = f"a_addr[{gi}]", f"b_addr[{gi}]"
        off = (buf if g < 3 else 1 - buf) * STAGE_BYTES
    reads = [f"HD_DS128(fa{nxt}[0], {ra}, {off})", f"HD_DS128(fa{nxt}[1], {ra}, {off + 4096})",
             f"HD_DS128(fb{nxt}[0], {rb}, {off})", f"HD_DS128(fb{nxt}[1], {rb}, {off + 4096})",
             f"HD_DS128(fb{nxt}[2], {rb}, {off + 8192})", f"HD_DS128(fb{nxt}[3], {rb}, {off + 12288})"]
    m = 0
    for s in range(4):
        for i in range(2):
            for j in range(4):
                if first and g == 0 and s == 0:
                    lines.append(f"    acc[S][{i}][{j}] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa{cur}[{i}][{s}], fb{cur}[{j}][{s}], kZero16, 0, 0, 0);")
                else:
                    lines.append(f"    acc[S][{i}][{j}] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa{cur}[{i}][{s}], fb{cur}[{j}][{s}], acc[S][{i}][{j}], 0, 0, 0);")
                fill = []
                gm = 32 * g + m                          # gap index inside the K-step
                if dma and gm % dma == 0 and gm // dma < 12:   # 4 A pieces, then 8 B pieces of the next K-step
                    q = gm // dma
                    fill.append(f"HD_DMA_A({q})" if q < 4 else f"HD_DMA_B({q - 4})")
                if m in (1, 3, 5, 7, 9, 11):
                    fill.append(reads.pop(0))
                for f in fill:
                    lines.append(f"    {f};")
                if epi_block is not None:
                    e = epi_block
                    g_ = None
                    if m < 16:
                        g_ = f"HD_EW(P, {e >> 2}, {e & 3}, {m})"
                    elif 18 <= m <= 21:
                        g_ = f"HD_ER({m - 18})"
                    elif m == 23:
                        g_ = "HD_EWAIT()"
                    elif m in (24, 26, 28, 30):
                        g_ = f"HD_ES({e >> 2}, {e & 3}, {(m - 24) // 2})"
                    if g_:
                        lines.append(f"    HD_G({g_});")
                    # the block's 24 vector instructions; unguarded: no control-flow join on the staged registers (first tile: harmless)
                    if EV_MODE == "one" and m == 23:
                        lines.append(f"    HD_EV({e & 3}, 0, 4);")
                    if EV_MODE == "two" and m in (23, 25):
                        lines.append(f"    HD_EV({e & 3}, {m - 23}, {m - 21});")
                    if EV_MODE == "four" and m in (23, 25, 27, 29):
                        lines.append(f"    HD_EV({e & 3}, {(m - 23) // 2}, {(m - 23) // 2 + 1});")
                lines.append("    HD_SB();")
                m += 1
    assert not reads


def kstep(name, first, epi, buf=None):
    lines = [f"#define HD_KSTEP_{name}(S, P) do {{"]
    blocks = {0: [None] * 4, 1: [0, 1, 2, 3], 2: [4, 5, 6, 7]}[epi]
    for g in range(4):
        cur = "0" if g % 2 == 0 else "1"
        if g < 3:
            lines.append(f"    HD_WAIT_FRAG(fa{cur}, fb{cur});")
            lines.append("    HD_SB();")
        else:
            # stores of the blocks of groups 0-2 (12) are younger than the 12 LDS-DMA pieces: they may stay in flight
            if epi:
                lines.append(f"    HD_BARRIER_E(fa{cur}, fb{cur});")
            else:
                lines.append(f"    HD_BARRIER(0, fa{cur}, fb{cur});")
            lines.append("    HD_SB();")
        stride = FS_STRIDE if epi else P_STRIDE
        assert 11 * stride < (32 if epi else 64)  # epilogue K-steps: every piece older than the first store; else: landed by group 3
        group(lines, g, first, blocks[g], dma=stride, buf=buf)
    lines.append("} while (0)")
    return " \\\n".join(lines) + "\n"


def main():
    global P_STRIDE, FS_STRIDE, EV_MODE
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--p-stride", type=int, default=P_STRIDE)
    ap.add_argument("--fs-stride", type=int, default=FS_STRIDE)
    ap.add_argument("--ev-mode", default=EV_MODE, choices=["one", "two", "four"])
    a = ap.parse_args()
    EV_MODE = a.ev_mode
    P_STRIDE, FS_STRIDE = a.p_stride, a.fs_stride
    out = ["// GENERATED by tools/gen_hd_kstep.py - do not edit.  K-step bodies of m360_linear_hd.hip.h.\n"]
    out.append(kstep("F", True, 1))
    out.append(kstep("S", False, 2))
    out.append(kstep("P", False, 0))
    # the same with the LDS stage static (K-step kt works on stage kt & 1: layers with an even number of K-steps)
    out.append(kstep("F0", True, 1, buf=0))
    out.append(kstep("S1", False, 2, buf=1))
    out.append(kstep("P0", False, 0, buf=0))
    out.append(kstep("P1", False, 0, buf=1))
    with open(OUT, "w") as f:
        f.write("\n".join(out))
    print("wrote", OUT, sum(len(o) for o in out), "bytes")


if __name__ == "__main__":
    main()
