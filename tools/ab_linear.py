#!/usr/bin/env python3
"""A/B timing of the m360_linear kernel variants on the dominant GEMM shape of the path
(M = 4096 rays x 128 samples, 1024 x 1024 hidden layer, fp32), interleaved rounds in ONE process
(cdna_hip_programming.md §5.4 rule 24), random data, HIP events on the launch stream."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from mipnerf360_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=4096 * 128)
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--k", type=int, default=1024)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--variants", type=str, default="1,2")
    ap.add_argument("--act", type=int, default=1)
    ap.add_argument("--ld-pad", type=int, default=0, help="extra floats of row stride for x and y (breaks the 4 KiB stride)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    g = torch.Generator(device=dev).manual_seed(0)
    x_full = torch.rand(args.m, args.k + args.ld_pad, device=dev, generator=g) * 2 - 1
    x = x_full[:, :args.k]
    w = (torch.rand(args.n, args.k, device=dev, generator=g) * 2 - 1) * (6.0 / args.k) ** 0.5
    b = torch.rand(args.n, device=dev, generator=g) - 0.5
    y_full = torch.empty(args.m, args.n + args.ld_pad, device=dev)
    y = y_full[:, :args.n]

    def run_linear():
        _lib.check(lib.m360_linear(x.data_ptr(), args.m, args.k + args.ld_pad, w.data_ptr(), b.data_ptr(), args.n, args.k,
                                   args.act, y.data_ptr(), args.n + args.ld_pad, torch.cuda.current_stream().cuda_stream), "linear")
    variants = [int(v) for v in args.variants.split(",")]
    outs = {}
    for v in variants:
        _lib.check(lib.m360_debug_set_linear_variant(v), "variant")
        run_linear()
        torch.cuda.synchronize()
        outs[v] = y[:: max(1, args.m // 4096)].clone()
    base = outs[variants[0]]
    ref = torch.relu(x[:: max(1, args.m // 4096)].double() @ w.double().T + b.double()) if args.act == 1 else None
    for v in variants:
        same = torch.equal(outs[v], base)
        err = float((outs[v].double() - ref).abs().max()) if ref is not None else float("nan")
        print(f"variant {v}: bit-identical to variant {variants[0]}: {same}; max |err| vs fp64 = {err:.3e}")
    flops = 2.0 * args.m * args.n * args.k
    times = {v: [] for v in variants}
    for _ in range(args.rounds):
        for v in variants:
            _lib.check(lib.m360_debug_set_linear_variant(v), "variant")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run_linear()
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / args.iters)
    for v in variants:
        t = sorted(times[v])
        med, best = t[len(t) // 2], t[0]
        print(f"variant {v}: median {med:.3f} ms = {flops / med / 1e9:.1f} TFLOP/s, best {best:.3f} ms = "
              f"{flops / best / 1e9:.1f} TFLOP/s  ({100 * flops / med / 1e9 / 157.3:.1f}% of 157.3)")
    if 3 in variants:
        import ctypes as C
        _lib.check(lib.m360_debug_set_linear_variant(3), "variant")
        run_linear()
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * (256 * 8))()
        _lib.check(lib.m360_debug_read_stamps(buf, 256 * 8), "stamps")
        import numpy as np
        st = np.array(buf[:], dtype=np.float64).reshape(256, 8)
        ks = st[:, 6].mean()
        names = ["group0+DMA issue", "group1", "group2", "DMA wait+barrier", "group3", "whole K-step"]
        print(f"stamps (cycles per K-step, mean over 256 workgroups; {ks:.0f} K-steps each; ideal group = 4096):")
        for i, nme in enumerate(names):
            print(f"   {nme:18s} {st[:, i].mean() / ks:9.0f}   (min {st[:, i].min() / ks:.0f} max {st[:, i].max() / ks:.0f})")
        tiles = ks / (args.k / 32)
        print(f"   epilogue per tile  {st[:, 7].mean() / tiles:9.0f}")
    lib.m360_debug_set_linear_variant(2)


if __name__ == "__main__":
    main()
