#!/usr/bin/env python3
"""Timing of one linear layer (fp32 exact-MFMA or bf16 kernel) on the dominant shape of the path
(M = 4096 x 128 rows, 1024 x 1024), random data, HIP events on the launch stream; optional in-kernel evidence from the
diagnostics build (make -C mipnerf360_amd/csrc diag -> libm360_diag.so):

  python tools/linear_bench.py --dtype bf16 --clock

`--clock` follows MI355X_MICROARCH.md (DVFS item 6): >= 2 s of back-to-back launches of the PRODUCT kernel on random
data, then ONE launch of the same kernel with s_memtime / s_memrealtime stamps around its main loop; the in-kernel clock
is d(s_memtime) / d(s_memrealtime) x 100 MHz, median over workgroups.  Prints TFLOP/s against the spec peak AND against
what the MFMA pipe could deliver at the clock the chip actually held."""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mipnerf360_amd import _lib, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", choices=("fp32", "bf16"), default="fp32")
    ap.add_argument("--m", type=int, default=4096 * 128)
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--k", type=int, default=1024)
    ap.add_argument("--act", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--clock", action="store_true", help="in-kernel clock + cycle stamps from libm360_diag.so")
    ap.add_argument("--variant", type=int, default=-1,
                    help="bf16 only: time a kernel of the diagnostics library instead of the product one (2 = software-pipelined, "
                         "3 = ping-pong); its stamped twin (variant - 2) serves --clock; the output is checked against fp64")
    ap.add_argument("--ld-pad", type=int, default=0, help="bf16 diag variants: pad the row stride of x / y / w by this many elements")
    ap.add_argument("--no-check", action="store_true", help="timing-only ablation variants produce wrong results by design")
    ap.add_argument("--soak-s", type=float, default=2.5)
    ap.add_argument("--json", type=str, default=None, help="append the result as one JSON line to this file")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    bf16 = args.dtype == "bf16"
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.rand(args.m, args.k, device=dev, generator=g) * 2 - 1
    w = (torch.rand(args.n, args.k, device=dev, generator=g) * 2 - 1) * (6.0 / args.k) ** 0.5
    b = torch.rand(args.n, device=dev, generator=g) - 0.5
    diag = None
    if args.clock or args.variant >= 0:
        path = os.path.join(os.path.dirname(_lib.LIB_PATH), "libm360_diag.so")
        if not os.path.exists(path):
            raise SystemExit(f"{path} missing: make -C mipnerf360_amd/csrc diag")
        diag = ctypes.CDLL(path)
        vp = ctypes.c_void_p
        diag.m360_diag_linear_bf16.argtypes = [vp, ctypes.c_long, ctypes.c_int, vp, vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
        diag.m360_diag_linear.argtypes = [vp, ctypes.c_long, ctypes.c_int, vp, vp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp]
        diag.m360_diag_read_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    if bf16:
        x = x.bfloat16()
        wp, bp = ops.pack_linear_bf16(w, b, args.n, args.k)
        y = torch.empty(args.m, args.n, device=dev, dtype=torch.bfloat16)
        if args.variant >= 0:
            pad = args.ld_pad
            xs, ys, ws = (torch.zeros(args.m, args.k + pad, device=dev, dtype=torch.bfloat16),
                          torch.zeros(args.m, args.n + pad, device=dev, dtype=torch.bfloat16),
                          torch.zeros(args.n, args.k + pad, device=dev, dtype=torch.bfloat16))
            xs[:, :args.k] = x
            ws[:, :args.k] = wp
            x, wp, y = xs[:, :args.k], ws[:, :args.k], ys[:, :args.n]  # strided views: rows (k + pad) / (n + pad) elements apart

            def run():
                rc = diag.m360_diag_linear_bf16(x.data_ptr(), args.m, args.k + pad, wp.data_ptr(), bp.data_ptr(), args.n, args.k,
                                                y.data_ptr(), args.n + pad, args.variant, args.k + pad,
                                                torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc
        else:
            run = lambda: ops.linear_bf16(x, wp, bp, args.act, out=y)  # noqa: E731
        peak, cyc_per_flop = 2500.0, 1.0 / (256 * 4 * 2 * 16 * 16 * 32 / 16.0)  # 16x16x32: 16 cycles per MFMA per SIMD
    else:
        wp, bp = ops.pack_linear(w, b, args.n, args.k)
        y = torch.empty(args.m, args.n, device=dev)
        run = lambda: ops.linear(x, wp, bp, args.act, out=y)  # noqa: E731
        peak, cyc_per_flop = 157.3, 1.0 / (256 * 4 * 2 * 32 * 32 * 2 / 64.0)   # 32x32x2 f32: 64 cycles per MFMA per SIMD
    run()
    torch.cuda.synchronize()
    if bf16 and not args.no_check:  # against the exact product of the same bf16 operands (a strided row sample), ReLU
        sub = slice(None, None, max(1, args.m // 4096))
        ref = (x[sub].double() @ wp.double().T + bp.double()).clamp_min(0)
        err = (y[sub].double() - ref).abs()
        bad = float((err - 2.0 ** -8 * ref.abs()).max())
        print(f"max |err| vs fp64 on a row sample: {float(err.max()):.3e}; worst excess over one bf16 ulp: {bad:.3e}")
        assert bad <= 2e-3, "wrong results"
        y2 = y.clone()
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        assert torch.equal(y, y2), "non-deterministic results (race)"
    flops = 2.0 * args.m * args.n * args.k
    times = []
    for _ in range(args.rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / args.iters)
    med = float(np.median(times))
    tf = flops / med / 1e9
    res = {"dtype": args.dtype, "variant": args.variant, "ld_pad": args.ld_pad, "M": args.m, "N": args.n, "K": args.k, "median_ms": round(med, 4), "best_ms": round(min(times), 4),
           "tflops": round(tf, 1), "frac_of_spec_peak": round(tf / peak, 4), "spec_peak_tflops": peak}
    print(f"{args.dtype} linear {args.m}x{args.n}x{args.k}: median {med:.3f} ms = {tf:.1f} TFLOP/s ({100 * tf / peak:.1f}% of "
          f"{peak} TF), best {min(times):.3f} ms")
    if bf16 and 20 <= args.variant < 200:  # in-kernel stamps of the w32 / w16 kernels (after the timed launches: a loaded chip)
        stt = (ctypes.c_ulonglong * 1024)()
        reader = diag.m360_diag_read_w32_stamps if args.variant < 100 else diag.m360_diag_read_w16_stamps
        reader.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
        assert reader(stt, 1024) == 0
        a = np.array(stt[:], dtype=np.float64).reshape(256, 4)
        unit = 32 if args.variant < 100 else 64   # contraction depth of one counted unit: w32 slabs, w16 stages
        res.update(in_kernel_clock_ghz=round(float(np.median(a[:, 0] / np.maximum(a[:, 1], 1)) * 0.1), 3),
                   cycles_per_32_deep=round(float(np.median((a[:, 0] - a[:, 3]) / np.maximum(a[:, 2], 1))) * 32 / unit, 1),
                   epilogue_cycles_per_tile=round(float(np.median(a[:, 3] / np.maximum(a[:, 2] / (args.k // unit), 1))), 1))
        print({k: res[k] for k in ("in_kernel_clock_ghz", "cycles_per_32_deep", "epilogue_cycles_per_tile")})
    if args.clock:
        t_end = time.time() + args.soak_s
        while time.time() < t_end:  # back-to-back product launches: the chip settles at its loaded clock
            for _ in range(20):
                run()
            torch.cuda.synchronize()
        for _ in range(40):
            run()
        st = torch.cuda.current_stream().cuda_stream
        if bf16:
            stamped = {2: 1, 3: 0, 9: 8}.get(args.variant, 0)
            rc = diag.m360_diag_linear_bf16(x.data_ptr(), args.m, x.stride(0), wp.data_ptr(), bp.data_ptr(), args.n, args.k, y.data_ptr(), y.stride(0), stamped, wp.stride(0), st)
        else:
            rc = diag.m360_diag_linear(x.data_ptr(), args.m, args.k, wp.data_ptr(), bp.data_ptr(), args.n, args.k, y.data_ptr(), args.n, st)
        assert rc == 0, rc
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (256 * 16))()
        assert diag.m360_diag_read_stamps(buf, 256 * 16) == 0
        s = np.array(buf[:], dtype=np.float64).reshape(256, 16)
        if bf16:
            cyc, rt = s[:, 1], s[:, 5]
            tiles = (args.m // 256) * (args.n // 256) / 256.0
            per_tile = np.median(cyc) / tiles
            pure = (256 * 256 * args.k * 2) * cyc_per_flop * 256  # MFMA issue cycles of one tile on one CU's 4 SIMDs
            res.update(stamp_main_loop_cycles_per_tile=round(per_tile), pure_mfma_cycles_per_tile=round(pure))
        else:
            cyc, rt = s[:, 8], s[:, 9]
            ksteps = np.median(s[:, 6])
            res.update(stamp_cycles_per_kstep=round(float(np.median(s[:, 5]) / max(ksteps, 1)), 1), pure_mfma_cycles_per_kstep=16384,
                       stamp_epilogue_cycles_per_tile=round(float(np.median(s[:, 7]) / (ksteps / (args.k / 32))), 1))
        clock = float(np.median(cyc / np.maximum(rt, 1)) * 0.1)  # GHz
        ceiling = flops / 1e12 / (flops * cyc_per_flop / (clock * 1e9))  # TFLOP/s if every MFMA slot were used at this clock
        res.update(in_kernel_clock_ghz=round(clock, 3), mfma_ceiling_at_held_clock_tflops=round(ceiling, 1),
                   frac_of_held_clock_ceiling=round(tf / ceiling, 4))
        print(f"in-kernel clock {clock:.3f} GHz (median over {len(cyc)} workgroups, after {args.soak_s:.1f} s of back-to-back "
              f"launches); MFMA ceiling at that clock {ceiling:.1f} TFLOP/s -> the kernel reaches {100 * tf / ceiling:.1f}% of it")
    print(json.dumps(res))
    if args.json:
        with open(args.json, "a") as f:
            f.write(json.dumps(res) + "\n")


if __name__ == "__main__":
    main()
