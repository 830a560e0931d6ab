#!/usr/bin/env python3
"""Stress of m360_mlp_chain_bf16's hand-over through the L2: many launches of the six-layer chain on fresh inputs, each compared bit for bit
with six launches of m360_linear_bf16 (paired rows); other work (a large copy on a second stream) runs alongside every other launch."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from mipnerf360_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--rows", type=int, default=4096 * 128)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
packs = []
for _ in range(6):
    w = (torch.rand(1024, 1024, device=dev, generator=g) * 2 - 1) * (6.0 / 1024) ** 0.5
    b = torch.rand(1024, device=dev, generator=g) * 0.2 - 0.1
    packs.append(ops.pack_linear_bf16(w, b, 1024, 1024))
flags = _lib.ACT_RELU | _lib.ROWS_PAIRED_IN | _lib.ROWS_PAIRED_OUT
M = a.rows
side = torch.cuda.Stream(device=dev)
junk_a, junk_b = torch.empty(64 << 20, device=dev, dtype=torch.uint8), torch.empty(64 << 20, device=dev, dtype=torch.uint8)
bad = 0
for rep in range(a.reps):
    x = (torch.rand(M, 1024, device=dev, generator=g) * 2 - 1).bfloat16()
    want = x
    for wp, bp in packs:
        want = ops.linear_bf16(want, wp, bp, flags)
    c0, c1 = x.clone(), torch.empty_like(x)
    if rep & 1:
        with torch.cuda.stream(side):
            junk_b.copy_(junk_a)
    got = ops.mlp_chain_bf16(c0, c1, packs)   # raises when a wait ran out
    torch.cuda.synchronize()
    if not torch.equal(got, want):
        bad += 1
        print(f"rep {rep}: {int((got != want).sum())} elements differ", flush=True)
print(f"rows {M}: {a.reps} launches of the chain, {bad} with a difference")
sys.exit(1 if bad else 0)
