#!/usr/bin/env python3
"""Timing of the opt-in bf16 linear kernel on the dominant shape (M = 4096 x 128, 1024 x 1024), random data,
HIP events on the launch stream; prints TFLOP/s against the 2.5 PFLOP/s dense bf16 MFMA peak."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from mipnerf360_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=4096 * 128)
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--k", type=int, default=1024)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--act", type=int, default=1)
    ap.add_argument("--variant", type=int, default=1, help="1 = persistent one-wave-per-SIMD kernel, 2 = 8-wave ping-pong")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    from mipnerf360_amd import _lib
    _lib.check(_lib.lib().m360_debug_set_linear_variant(10 + args.variant), "variant")
    g = torch.Generator(device=dev).manual_seed(0)
    x = (torch.rand(args.m, args.k, device=dev, generator=g) * 2 - 1).bfloat16()
    w = (torch.rand(args.n, args.k, device=dev, generator=g) * 2 - 1) * (6.0 / args.k) ** 0.5
    b = torch.rand(args.n, device=dev, generator=g) - 0.5
    wp, bp = ops.pack_linear_bf16(w, b, args.n, args.k)
    y = torch.empty(args.m, args.n, device=dev, dtype=torch.bfloat16)
    ops.linear_bf16(x, wp, bp, args.act, out=y)
    torch.cuda.synchronize()
    sub = slice(None, None, max(1, args.m // 2048))
    ref = x[sub].double() @ wp.double().T + bp.double()
    ref = ref.clamp_min(0) if args.act == 1 else ref
    print(f"max |err| vs fp64 on a row sample = {float((y[sub].double() - ref).abs().max()):.3e} (bf16 ulp ~ 4e-3 x |y|)")
    flops = 2.0 * args.m * args.n * args.k
    times = []
    for _ in range(args.rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            ops.linear_bf16(x, wp, bp, args.act, out=y)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / args.iters)
    t = sorted(times)
    med = t[len(t) // 2]
    if args.variant == 3:
        import ctypes
        import numpy as np
        buf = (ctypes.c_ulonglong * (256 * 8))()
        _lib.check(_lib.lib().m360_debug_read_stamps(buf, 256 * 8), "stamps")
        st = np.array(buf[:], dtype=np.float64).reshape(256, 8)
        print("stamps (median cycles over the first 256 workgroups): prologue %.0f, main loop %.0f (%.0f per K-step), epilogue %.0f; "
              "workgroup lifetime %.0f" % (np.median(st[:, 0]), np.median(st[:, 1]), np.median(st[:, 1]) / (args.k / 64),
                                           np.median(st[:, 2]), np.median(st[:, 4] - st[:, 3])))
    print(f"bf16 linear {args.m}x{args.n}x{args.k}: median {med:.3f} ms = {flops / med / 1e9:.1f} TFLOP/s "
          f"({100 * flops / med / 1e9 / 2500:.1f}% of 2.5 PF), best {t[0]:.3f} ms")


if __name__ == "__main__":
    main()
