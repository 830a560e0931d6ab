#!/usr/bin/env python3
"""m360_mlp_chain_bf16 (six 1024 x 1024 ReLU layers in ONE launch, hand-over through the XCD's L2) against six launches of
m360_linear_bf16 on paired rows: ms per chain of six at M = 4096 x 128 rows (and 8192 x 256 with --c5), alternating, bitwise compared."""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from mipnerf360_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--c5", action="store_true")
ap.add_argument("--layers", type=int, default=6)
ap.add_argument("--width", type=int, default=1024)
ap.add_argument("--rounds", type=int, default=9)
a = ap.parse_args()
dev = torch.device("cuda:0")
M = 8192 * 256 if a.c5 else 4096 * 128
g = torch.Generator(device=dev).manual_seed(0)
packs = []
for _ in range(a.layers):
    w = (torch.rand(a.width, a.width, device=dev, generator=g) * 2 - 1) * (6.0 / a.width) ** 0.5
    b = torch.rand(a.width, device=dev, generator=g) * 0.2 - 0.1
    packs.append(ops.pack_linear_bf16(w, b, a.width, a.width))
x = ops.pair_rows((torch.rand(M, a.width, device=dev, generator=g) * 2 - 1).bfloat16())
flags = _lib.ACT_RELU | _lib.ROWS_PAIRED_IN | _lib.ROWS_PAIRED_OUT
p0, p1 = torch.empty_like(x), torch.empty_like(x)


def layer_by_layer():
    src = x
    for i, (wp, bp) in enumerate(packs):
        dst = p0 if i % 2 == 0 else p1
        ops.linear_bf16(src, wp, bp, flags, out=dst)
        src = dst
    return src


c0, c1 = torch.empty_like(x), torch.empty_like(x)


def chain():
    c0.copy_(x)
    return ops.mlp_chain_bf16(c0, c1, packs)


want = layer_by_layer().clone()
got = chain()
same = bool(torch.equal(got, want))


def timed(fn, pre=None):
    ts = []
    for _ in range(a.rounds):
        if pre:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return statistics.median(ts), min(ts)


def chain_only():  # the launch alone (the copy of the input is the caller's first layer in the product)
    import ctypes as C
    ops.mlp_chain_bf16(c0, c1, packs)


for rep in range(2):
    m1, b1 = timed(layer_by_layer)
    m2, b2 = timed(chain_only, pre=lambda: c0.copy_(x))
    print(json.dumps({"M": M, "layers": a.layers, "layer_by_layer_ms": round(m1, 4), "chain_ms": round(m2, 4), "best": [round(b1, 4), round(b2, 4)],
                      "per_layer_ms": [round(m1 / a.layers, 4), round(m2 / a.layers, 4)], "width": a.width, "tflops": [round(2.0 * M * a.width * a.width * a.layers / m1 / 1e9, 1), round(2.0 * M * a.width * a.width * a.layers / m2 / 1e9, 1)],
                      "same_bits": same}), flush=True)
