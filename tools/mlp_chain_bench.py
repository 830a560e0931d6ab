#!/usr/bin/env python3
"""Layer-by-layer vs row-blocked execution of the hidden-layer chain (1024 x 1024 ReLU layers, M = 4096 x 128 rows).

Layer by layer (what the stage drivers did through round 2) every layer streams its whole input (M x 1024) from HBM and
its whole output back: the ping / pong activation buffers are 1-2 GB each, far beyond the 256 MiB Infinity Cache.
Row-blocked: the SAME kernels, launched per block of `rows` rows through all layers before the next block is touched, so a
block's ping / pong pair (2 x rows x 1024 x elt bytes) stays resident on die between the layer that writes it and the
layer that reads it (MI355X_MICROARCH.md, Infinity Cache residency rule; cdna_hip_programming.md rule 28: streamed data
served from on-die raises the clock the chip holds under an MFMA-dense loop).  Results are bit-identical: the layers are
row-independent.

  python tools/mlp_chain_bench.py --dtype bf16 --json gpurun_out/x/chain.jsonl
"""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from mipnerf360_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", choices=("fp32", "bf16"), default="bf16")
    ap.add_argument("--m", type=int, default=4096 * 128)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--blocks", type=str, default="0,262144,131072,65536,32768,16384,8192")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reuse", action="store_true", help="every row block writes the SAME rows [0, rows) of the ping / pong buffers (timing only: "
                    "what the stage drivers do - the pair then never has to leave the Infinity Cache)")
    ap.add_argument("--balanced", action="store_true", help="fp32: m360_linear_balanced (ticketed tail) as the product uses")
    ap.add_argument("--json", type=str, default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    bf16 = args.dtype == "bf16"
    g = torch.Generator(device=dev).manual_seed(0)
    W, M, L = args.width, args.m, args.layers
    x0 = torch.rand(M, W, device=dev, generator=g) * 2 - 1
    packs = []
    for _ in range(L):
        w = (torch.rand(W, W, device=dev, generator=g) * 2 - 1) * (6.0 / W) ** 0.5
        b = torch.rand(W, device=dev, generator=g) * 0.2 - 0.1
        packs.append(ops.pack_linear_bf16(w, b, W, W) if bf16 else ops.pack_linear(w, b, W, W))
    if bf16:
        x0 = x0.bfloat16()
    a, bb = torch.empty_like(x0), torch.empty_like(x0)
    lin = ops.linear_bf16 if bf16 else ops.linear

    def chain(rows):
        """x0 -> L layers, ping-pong between a and bb; rows = 0: whole batch per launch."""
        step = rows or M
        for r0 in range(0, M, step):
            r1 = min(r0 + step, M)
            src = x0[r0:r1]
            for li, (wp, bp) in enumerate(packs):
                dst = (a if li % 2 == 0 else bb)[(0 if args.reuse else r0):(r1 - r0 if args.reuse else r1)]
                if bf16:
                    lin(src, wp, bp, _lib.ACT_RELU, out=dst)
                else:
                    lin(src, wp, bp, _lib.ACT_RELU, out=dst, balanced=args.balanced)
                src = dst
        return (a if (L - 1) % 2 == 0 else bb)

    ref = chain(0).clone()
    flops = 2.0 * M * W * W * L
    results = []
    blocks = [int(v) for v in args.blocks.split(",")]
    for rows in blocks:
        out = chain(rows)
        same = bool(torch.equal(out, ref)) if not args.reuse else None
        times = []
        for _ in range(args.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            chain(rows)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        ms = statistics.median(times)
        elt = 2 if bf16 else 4
        res = {"dtype": args.dtype, "M": M, "width": W, "layers": L, "rows_per_block": rows or M,
               "launches": L * ((M + (rows or M) - 1) // (rows or M)), "ping_pong_MiB": round(2 * (rows or M) * W * elt / 2 ** 20, 1),
               "ms": round(ms, 4), "ms_per_layer": round(ms / L, 4), "tflops": round(flops / ms / 1e9, 1),
               "bit_identical_to_layer_by_layer": same, "min_ms": round(min(times), 4), "reuse": bool(args.reuse)}
        results.append(res)
        print(json.dumps(res), flush=True)
        if args.json:
            with open(args.json, "a") as f:
                f.write(json.dumps(res) + "\n")


if __name__ == "__main__":
    main()
