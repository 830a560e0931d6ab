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
from a simulation of the issue order (`simulate()`).  The tile's 32 stores are issued in its (exposed) epilogue; the LAST stage of a
tile (kind L) also issues the activation pieces of the next tile's stage 1, which its first stage (kind Z) then does not: everything
stage Z waits for is older than the stores, which may stay in flight (counted like pieces).
Macros: W16_STAGE0Z() first stage of a tile (accumulators restart from 0 through the C operand), W16_STAGE<B>() generic,
W16_STAGE1L() last stage of a tile.
"""
import os

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mipnerf360_amd", "csrc", "m360_linear_bf16_w16_gen.inc")
JB_OUTER = False        # experiment: weight block outer, activation block inner inside a half (same accumulation order per tuple): 1246-1249 against
                        # 1241-1244 TF alternating on one box (profiles/r03/bf16_w16_mfma_order_ab_NO_EFFECT.txt, tools/diag/w16_order_ab.sh)
TWO_BARRIERS = False   # experiment: all 8 activation pieces in half 0, the hi rows awaited at barrier M1 as well, no barrier E1 - correct
                        # (bitwise soak), 1228 TF against 1222-1243 with three barriers (profiles/r03/bf16_w16_two_barriers_REJECTED.jsonl)
STORES = 32             # 16-byte stores per lane and tile (8 activation blocks x 4 column pieces x 2 rows / 2), all in the epilogue;
                        # the bodies name them W16_STORES: the kernel's constant (32; bf16x3: 64; fused heads without the layer's own
                        # output: 32 small ones) - only "all of a tile's stores sit between these two pieces" is simulated here


def dma_plan(kind):
    """{(kk, half): [(gap, operand, part, stage offset)]}: LDS-DMA pieces of one wave in a stage of the given kind.
    '' generic; 'L' last stage of a tile: also issues the activation pieces its successor would issue in k-step 0, so that every
    piece the next tile's FIRST stage waits for is older than the epilogue's stores (vmcnt retires in order and a store takes
    ~2 k cycles to complete: with the generic schedule the first barrier of every tile waited for the stores, 115 cycles per
    32 deep on average); 'Z' first stage of a tile: accumulators restart from 0, no activation pieces; 'ZL' the only stage of a
    tile (64-deep layers): both."""
    plan = {(0, 0): [], (0, 1): [], (1, 0): [], (1, 1): []}
    if kind not in ("Z", "ZL"):
        if TWO_BARRIERS:   # all 8 activation pieces in half 0 (one per 4th gap): the hi rows are awaited at M1 as well, no barrier E1
            plan[(0, 0)] += [(1 + 4 * q, "X", q, 1) for q in range(8)]
        else:
            plan[(0, 0)] += [(1 + 8 * q, "X", q, 1) for q in range(4)]          # lo rows of stage s + 1
            plan[(0, 1)] += [(1 + 8 * q, "X", 4 + q, 1) for q in range(4)]      # hi rows
    plan[(1, 0)] += [(1 + 8 * q, "W", q, 2) for q in range(4)]
    plan[(1, 1)] += [(1 + 8 * q, "W", 4 + q, 2) for q in range(4)]
    if kind in ("L", "ZL"):
        plan[(1, 0)] += [(5 + 8 * q, "X", q, 2) for q in range(4)]          # lo rows of stage s + 2 (free since barrier E0)
        plan[(1, 1)] += [(5 + 8 * q, "X", 4 + q, 2) for q in range(4)]      # hi rows (free since barrier M1)
    return {k: sorted(v) for k, v in plan.items()}


def simulate(nst=8):
    """-> {(kind, 'M1' | 'E1'): (pieces, stores) that may stay outstanding at that barrier (steady state)}"""
    ops, mark, kinds = [], {}, {}
    total = max(3 * nst, 8)
    for s in range(total):
        r = s % nst
        kind = "ZL" if nst == 1 else "Z" if r == 0 else "L" if r == nst - 1 else ""
        kinds[s] = kind
        plan = dma_plan(kind)
        for kk in range(2):
            for half in range(2):
                for _, what, part, off in plan[(kk, half)]:
                    ops.append((what + ("L" if part < 4 else "H"), s + off))
                mark[(s, "M1" if half == 0 else "E1")] = len(ops) if kk == 1 else mark.get((s, "M1" if half == 0 else "E1"))
        if kind in ("L", "ZL"):
            ops += [("S", None)] * STORES
    res = {}
    for s in range(max(nst, 3), max(2 * nst, 6)):
        for bar, needs in ((("M1", ("XL", "XH", "WL", "WH")),) if TWO_BARRIERS else (("M1", ("XL", "WL", "WH")), ("E1", ("XH",)))):
            upto = mark[(s, bar)]
            need = max(i for i, o in enumerate(ops[:upto]) if o[0] in needs and o[1] == s + 1)
            younger = ops[need + 1:upto]
            key = (kinds[s] if kinds[s] else ("after Z" if kinds[s - 1] == "Z" else ""), bar)
            val = (sum(1 for o in younger if o[0] != "S"), sum(1 for o in younger if o[0] == "S"))
            assert res.setdefault(key, val) == val, (key, val, res)
    return res


def w_off(jb):
    # weight block jb = LDS rows 32 (jb >> 1) + 4 (jb & 1) + {8 a + b}: 128-byte rows
    return 4096 * (jb >> 1) + 512 * (jb & 1)


def stage(B, kind, vm):
    name = f"W16_STAGE{B}{kind}"
    L = [f"#define {name}() do {{"]
    nb = 1 - B
    plan = dma_plan(kind)
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
            # stage s + 1 lives in the other buffer, stage s + 2 in this one
            dma = {g: f"W16_DMA_{what}({nb if off == 1 else B}, {part})" for g, what, part, off in plan[(kk, half)]}
            for m in range(32):
                ib, jb = (4 * half + m // 8, m % 8) if not JB_OUTER else (4 * half + m % 4, m // 4)
                z = "_Z" if (kind in ("Z", "ZL") and kk == 0) else ""
                L.append(f"    W16_MFMA{z}({ib}, {jb}, fw{cur}[{jb}], fx[{ib}]);")
                if m in read_gaps:
                    L.append(f"    {reads[read_gaps[m]]};")
                if m in dma:
                    L.append(f"    {dma[m]};")
                L.append("    W16_SB();")
            advx = sum(1 for g, what, part, off in plan[(kk, half)] if what == "X" and part == 7)   # the last hi piece: advance
            if half == 0:
                for _ in range(advx):
                    L.append("    W16_ADV_X();")
                if kk == 0:
                    L.append("    W16_WAIT_HI();")
                else:
                    p, st = vm[(kind if kind in ("Z", "L", "ZL") else "", "M1")]
                    assert st in (0, STORES)
                    L.append(f"    W16_BARRIER_M1({p}, {p}{' + W16_STORES' if st else ''});")
                L.append("    W16_SB();")
            else:
                for _ in range(advx):
                    L.append("    W16_ADV_X();")
                if kk == 0:
                    L.append(f"    W16_WAIT_NEXT(fw{nxt});")
                    L.append("    W16_BARRIER_E0();")
                else:
                    L.append("    W16_ADV_W();")
                    L.append(f"    W16_WAIT_NEXT(fw{nxt});")
                    if not TWO_BARRIERS:
                        p, st = vm[(kind if kind in ("Z", "L", "ZL") else "", "E1")]
                        assert st in (0, STORES)
                        L.append(f"    W16_BARRIER_E1({p}, {p}{' + W16_STORES' if st else ''});")
                L.append("    W16_SB();")
    L.append("} while (0)")
    return " \\\n".join(L) + "\n"


# ---------------------------------------------------------------------------------------------------------------------------------
# bf16x3 ("X3"): every value is two bf16 terms, a product is xl wh + xh wh + xh wl.  Per 64-deep block b three stages run, sharing
# an operand with their neighbour, so the activation and weight halves of the two LDS buffers are switched independently:
#     T1: Xl_b (XA) x Wh_b (WA)      T2: Xh_b (XB) x Wh_b (WA)      T3: Xh_b (XB) x Wl_b (WB)
# Four operand tiles per block instead of six; each is staged as soon as its region is free:
#     T1 issues Wl_b -> WB (k-step 0; not the first block of a tile) and Xl_{b+1} -> XA (k-step 1: lo rows after barrier E0, hi after M1)
#     T2 issues Wh_{b+1} -> WA (k-step 1, after E0)          T3 issues Xh_{b+1} -> XB (k-step 1: lo after E0, hi after M1)
#     the last T3 of a tile also issues Wl_0 of the next tile, so that what its first stages wait for is older than the stores
OUT_X3 = OUT.replace("_w16_gen.inc", "_w16x3_gen.inc")
STORES_X3 = 64
X3_TYPES = {
    "T1": dict(xb=0, wb=0, nxb=1, nwb=0, e1=True,
               dma={(0, 0): [("WL", q) for q in range(4)], (0, 1): [("WL", 4 + q) for q in range(4)],
                    (1, 0): [("XL", q) for q in range(4)], (1, 1): [("XL", 4 + q) for q in range(4)]}),
    "T2": dict(xb=1, wb=0, nxb=1, nwb=1, e1=False,
               dma={(1, 0): [("WH", q) for q in range(4)], (1, 1): [("WH", 4 + q) for q in range(4)]}),
    "T3": dict(xb=1, wb=1, nxb=0, nwb=0, e1=True,
               dma={(1, 0): [("XH", q) for q in range(4)], (1, 1): [("XH", 4 + q) for q in range(4)]}),
}
X3_DST = {"XL": 0, "XH": 1, "WH": 0, "WL": 1}   # LDS buffer of each operand


def x3_plan(t, variant):
    """[(kk, half, gap, operand, piece)] of one stage; variant: '' | 'Z' (first T1 of a tile) | 'L' (last T3 of a tile)"""
    plan = []
    for (kk, half), lst in X3_TYPES[t]["dma"].items():
        for i, (op, q) in enumerate(lst):
            if variant == "Z" and op == "WL":
                continue
            plan.append((kk, half, 1 + 8 * i, op, q))
    if variant == "L":
        plan += [(1, 0, 5 + 8 * q, "WL", q) for q in range(4)] + [(1, 1, 5 + 8 * q, "WL", 4 + q) for q in range(4)]
    return sorted(plan)


def x3_simulate(nb=4):
    """-> {(type + variant, 'M1' | 'E1'): (pieces, stores) that may stay outstanding at that barrier (steady state)}"""
    ops, mark, names = [], {}, {}
    # which block's operand a stage of block g issues: XL, WH, XH of block g + 1; WL of block g (L: of block g + 1)
    for g in range(3 * nb):
        b = g % nb
        for t in ("T1", "T2", "T3"):
            variant = "Z" if (t == "T1" and b == 0) else "L" if (t == "T3" and b == nb - 1) else ""
            names[(g, t)] = t + variant
            for kk in range(2):
                for half in range(2):
                    for (k_, h_, gap, op, q) in x3_plan(t, variant):
                        if (k_, h_) != (kk, half):
                            continue
                        blk = g + 1 if (op != "WL" or (variant == "L" and gap % 8 == 5)) else g
                        ops.append((op + (".lo" if q < 4 else ".hi"), blk))
                    if kk == 1:
                        mark[(g, t, "M1" if half == 0 else "E1")] = len(ops)
            if variant == "L":
                ops += [("S", None)] * STORES_X3
    need = {"T1": {"M1": lambda g: [("XH.lo", g)], "E1": lambda g: [("XH.hi", g)]},
            "T2": {"M1": lambda g: [("WL.lo", g), ("WL.hi", g)]},
            "T3": {"M1": lambda g: [("WH.lo", g + 1), ("WH.hi", g + 1), ("XL.lo", g + 1)], "E1": lambda g: [("XL.hi", g + 1)]}}
    res = {}
    for g in range(nb, 2 * nb):
        for t in ("T1", "T2", "T3"):
            for bar, fn in need[t].items():
                upto = mark[(g, t, bar)]
                idx = [i for i, o in enumerate(ops[:upto]) if o in fn(g)]
                assert idx, (g, t, bar)
                younger = ops[max(idx) + 1:upto]
                key = (names[(g, t)] if names[(g, t)] != "T1" or names[(g - 0, "T1")] == "T1" else "T1", bar)
                val = (sum(1 for o in younger if o[0] != "S"), sum(1 for o in younger if o[0] == "S"))
                # the stages right after a tile boundary may see the stores: keep them apart from the steady-state ones
                after = 3 * (g % nb) + ("T1", "T2", "T3").index(t)   # stage index inside the tile
                key = (key[0] + (f"@{after}" if val[1] else ""), bar)
                assert res.setdefault(key, val) == val, (key, val, res)
    return res


def x3_stage(t, variant, vm, at=None):
    """at: stage index inside the tile for the variants that still see the previous tile's stores"""
    T = X3_TYPES[t]
    name = f"W16X_{t}{variant}" + (f"_S{at}" if at is not None else "")
    L = [f"#define {name}() do {{"]
    plan = x3_plan(t, variant)
    key = t + variant + (f"@{at}" if at is not None else "")
    for kk in range(2):
        cur, nxt = kk, 1 - kk
        for half in range(2):
            if half == 0:
                reads = [f"W16_RD(fx[{i}], xa{kk}{T['xb']}, {i * 2048})" for i in range(4, 8)]
                read_gaps = {0: 0, 2: 1, 4: 2, 6: 3}
            else:
                (rx, rw, rk) = (T["xb"], T["wb"], 1) if kk == 0 else (T["nxb"], T["nwb"], 0)
                reads = [f"W16_RD(fw{nxt}[{j}], wa{rk}{rw}, {w_off(j)})" for j in range(8)] + \
                        [f"W16_RD(fx[{i}], xa{rk}{rx}, {i * 2048})" for i in range(4)]
                read_gaps = {2 * k: k for k in range(12)}
            dma = {gap: f"W16X_DMA_{op}({q})" for (k_, h_, gap, op, q) in plan if (k_, h_) == (kk, half)}
            for m in range(32):
                ib, jb = 4 * half + m // 8, m % 8
                z = "_Z" if (variant == "Z" and kk == 0) else ""
                L.append(f"    W16_MFMA{z}({ib}, {jb}, fw{cur}[{jb}], fx[{ib}]);")
                if m in read_gaps:
                    L.append(f"    {reads[read_gaps[m]]};")
                if m in dma:
                    L.append(f"    {dma[m]};")
                L.append("    W16_SB();")
            for op in sorted({op for (k_, h_, gap, op, q) in plan if (k_, h_) == (kk, half) and q == 7}):
                L.append(f"    W16X_ADV_{op}();")
            if half == 0:
                if kk == 0:
                    L.append("    W16_WAIT_HI();")
                else:
                    p, st = vm[(key, "M1")]
                    assert st in (0, STORES_X3)
                    L.append(f"    W16_BARRIER_M1({p}, {p}{' + W16_STORES' if st else ''});")
                L.append("    W16_SB();")
            else:
                L.append(f"    W16_WAIT_NEXT(fw{nxt});")
                if kk == 0:
                    L.append("    W16_BARRIER_E0();")
                elif T["e1"]:
                    p, st = vm[(key, "E1")]
                    assert st in (0, STORES_X3)
                    L.append(f"    W16_BARRIER_E1({p}, {p}{' + W16_STORES' if st else ''});")
                L.append("    W16_SB();")
    L.append("} while (0)")
    return " \\\n".join(L) + "\n"


def main_x3():
    vm = x3_simulate()
    out = ["// GENERATED by tools/gen_w16_slab.py - do not edit.  bf16x3 stage bodies of m360_linear_bf16_w16.hip.h.\n"]
    keys = sorted({k[0] for k in vm})
    for k in keys:   # e.g. "T1", "T1Z@0", "T2@1", "T3L"
        base, _, at = k.partition("@")
        t, variant = base[:2], base[2:]
        out.append(x3_stage(t, variant, vm, int(at) if at else None))
    with open(OUT_X3, "w") as f:
        f.write("\n".join(out))
    print("wrote", OUT_X3, sum(len(o) for o in out), "bytes; bodies:", keys, "| barrier counts (pieces, stores):", vm)


def main():
    vm = simulate()
    # the stage after Z awaits what Z issued (nothing but weights) and what itself issued: same counts as a generic stage
    assert vm[("after Z", "M1")] == vm[("", "M1")] and vm[("", "M1")][1] == 0 and vm[("L", "M1")][1] == 0, vm
    if not TWO_BARRIERS:
        assert vm[("after Z", "E1")] == vm[("", "E1")] and vm[("", "E1")][1] == 0 and vm[("L", "E1")][1] == 0, vm
    out = ["// GENERATED by tools/gen_w16_slab.py - do not edit.  Stage bodies of m360_linear_bf16_w16.hip.h.\n"]
    out.append(stage(0, "Z", vm))
    out.append(stage(0, "", vm))
    out.append(stage(1, "", vm))
    out.append(stage(1, "L", vm))
    vm1 = simulate(nst=1)   # 64-deep layers: every stage is the first and the last of its tile, the buffers alternate by tile
    out.append(stage(0, "ZL", vm1))
    out.append(stage(1, "ZL", vm1))
    vm.update(vm1)
    with open(OUT, "w") as f:
        f.write("\n".join(out))
    print("wrote", OUT, sum(len(o) for o in out), "bytes; barrier counts (pieces, stores):", vm)


if __name__ == "__main__":
    main()
    main_x3()
