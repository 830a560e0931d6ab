#!/usr/bin/env python3
"""Generates mipnerf360_amd/csrc/m360_linear_bf16_w16_gen.inc: the slab bodies of the one-wave-per-SIMD bf16 linear kernel with
128 x 128 wave tiles on v_mfma_f32_16x16x32_bf16 (m360_linear_bf16_w16.hip.h), every non-matrix instruction in ONE MFMA gap.

A slab = 32 bf16 of the contraction = ONE k-step of the MFMA: 8 x 8 blocks of 16 x 16 = 64 MFMAs of 16 cycles on the wave tile,
in two halves of 32 (activation blocks 0-3, then 4-7; all 8 weight blocks each).  The LDS holds a RING of 4 slabs (32 KiB each:
256 activation rows + 256 weight rows x 64 B).  Body of slab t at ring position P = t % 4, weight fragment set P % 2:
  half 0  (gaps 0-31)   MFMAs on fx[0..3] x fw{cur}[0..7]
                        gaps 0,2,4,6    ds_read_b128 of fx[4..7] of THIS slab                       (slot P)
                        gaps 5,13,21,29 LDS-DMA of the 4 WEIGHT pieces of slab t+3 -> slot (P+3)%4  (free since barrier t-1)
  wait fx[4..7] | counted vmcnt: this wave's pieces of slab t+1 have landed | s_barrier
  half 1  (gaps 32-63)  MFMAs on fx[4..7] x fw{cur}[0..7]
                        gaps 32,34,..,54 ds_read_b128 of fw{nxt}[0..7], fx[0..3] of slab t+1        (slot (P+1)%4)
                        gaps 37,45,53,61 LDS-DMA of the 4 ACTIVATION pieces of slab t+4 -> slot P   (free since this barrier)
  wait fw{nxt}, fx[0..3]
A piece is issued 2.5-3.5 slabs before its first read; one barrier per 1024 cycles of matrix work.
Variants (the counted vmcnt of every barrier comes from a simulation of the issue order, `simulate()`):
  W16_SLAB<P>      generic
  W16_SLAB3L       last slab of a tile: half 1 also issues the weight pieces of slab t+4 (gaps 33,41,49,57), so that every piece
                   the next tile needs through ITS slab 3 is older than the epilogue's stores
  W16_SLAB0Z       slab 0 of a tile: accumulators restart from 0 through the C operand, no weight pieces in half 0
  W16_SLAB0Z/1E/2E after an epilogue: the 32 stores are younger than the pieces waited for (+32 when the workgroup has stored)
"""
import os

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mipnerf360_amd", "csrc", "m360_linear_bf16_w16_gen.inc")
STORES = 32  # 16-byte stores per lane in an epilogue (8 activation blocks x 4 column pieces)


def simulate():
    """Issue order of one wave around a tile boundary -> outstanding-operation count allowed at the barrier of each slab variant.
    Slab t's barrier needs every piece of slab t+1 landed: N = operations issued after the youngest of them."""
    n = 12                       # slabs per tile in the model (>= 8)
    ops = []                     # ("X"/"W", absolute slab) or ("S",)
    bar = {}                     # absolute slab -> index into ops at its barrier
    for t in range(3 * n):
        r = t % n
        if r != 0:
            ops += [("W", t + 3)] * 4          # half 0 (slab 0 of a tile: none, the last slab of the previous tile issued them)
        bar[t] = len(ops)
        ops += [("X", t + 4)] * 4              # half 1
        if r == n - 1:
            ops += [("W", t + 4)] * 4
            ops += [("S",)] * STORES           # epilogue
    res = {}
    for t in range(n, 2 * n):                  # the middle tile: steady state on both sides
        need = max(i for i, o in enumerate(ops[:bar[t]]) if o[0] in "XW" and o[1] == t + 1)
        younger = ops[need + 1:bar[t]]
        res[t % n] = (sum(1 for o in younger if o[0] != "S"), sum(1 for o in younger if o[0] == "S"))
    return res, n


def rd(dst, which, slot, blk):
    # address register: lo / hi half of the ring; immediate: block offset + odd slot * 32768
    reg = f"{which}a{'h' if slot >= 2 else 'l'}"
    # activation block ib = rows 16 ib..: 1024 B apart; weight block jb = LDS rows 32 (jb >> 1) + 4 (jb & 1) + {8 a + b} (the row
    # permutation that gives a lane 8 consecutive output columns per pair of blocks, see the kernel)
    off = blk * 1024 if which == "x" else 2048 * (blk >> 1) + 256 * (blk & 1)
    return f"W16_RD({dst}, {reg}, {off + (slot & 1) * 32768})"


def slab(P, kind, vm):
    """kind: '' generic, 'Z' first of a tile (zero C, no W pieces), 'E' after an epilogue, 'L' last of a tile"""
    name = f"W16_SLAB{P}{kind}"
    cur, nxt = P % 2, 1 - P % 2
    L = [f"#define {name}() do {{"]
    for half in range(2):
        if half == 0:
            reads = [rd(f"fx[{i}]", "x", P, i) for i in range(4, 8)]
            read_gaps = {0: 0, 2: 1, 4: 2, 6: 3}
            dma = {} if kind == "Z" else {5 + 8 * q: f"W16_DMA_W({(P + 3) % 4}, {q})" for q in range(4)}
        else:
            ns = (P + 1) % 4
            reads = [rd(f"fw{nxt}[{j}]", "w", ns, j) for j in range(8)] + [rd(f"fx[{i}]", "x", ns, i) for i in range(4)]
            read_gaps = {2 * k: k for k in range(12)}
            dma = {5 + 8 * q: f"W16_DMA_X({P}, {q})" for q in range(4)}
            if kind == "L":
                dma.update({1 + 8 * q: f"W16_DMA_W({P}, {q})" for q in range(4)})
        for m in range(32):
            ib, jb = 4 * half + m // 8, m % 8
            L.append(f"    W16_MFMA{'_Z' if kind == 'Z' else ''}(acc[{ib}][{jb}], fw{cur}[{jb}], fx[{ib}]);")
            if m in read_gaps:
                L.append(f"    {reads[read_gaps[m]]};")
            if m in dma:
                L.append(f"    {dma[m]};")
            L.append("    W16_SB();")
        if half == 0:
            if kind != "Z":
                L.append("    W16_ADV_W();")
            pieces, stores = vm
            if stores:
                L.append(f"    W16_BARRIER_E({pieces}, {pieces + stores});")
            else:
                L.append(f"    W16_BARRIER({pieces});")
            L.append("    W16_SB();")
        else:
            L.append("    W16_ADV_X();")
            if kind == "L":
                L.append("    W16_ADV_W();")
            L.append(f"    W16_WAIT_NEXT(fw{nxt});")
            L.append("    W16_SB();")
    L.append("} while (0)")
    return " \\\n".join(L) + "\n"


def main():
    vm, n = simulate()
    # steady state must not depend on the position, and only slabs 0-2 of a tile see the stores
    assert all(vm[r] == (16, 0) for r in range(3, n)), vm
    assert all(vm[r][1] == STORES for r in range(3)), vm
    out = ["// GENERATED by tools/gen_w16_slab.py - do not edit.  Slab bodies of m360_linear_bf16_w16.hip.h.\n"]
    out.append(slab(0, "Z", vm[0]))
    out.append(slab(1, "E", vm[1]))
    out.append(slab(2, "E", vm[2]))
    for p in range(4):
        out.append(slab(p, "", vm[4 + p]))
    out.append(slab(3, "L", vm[n - 1]))
    with open(OUT, "w") as f:
        f.write("\n".join(out))
    print("wrote", OUT, sum(len(o) for o in out), "bytes; barrier counts (pieces, stores) by slab of a tile:", {r: vm[r] for r in range(5)})


if __name__ == "__main__":
    main()
