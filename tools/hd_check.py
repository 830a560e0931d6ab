#!/usr/bin/env python3
"""A/B of the two persistent fp32 kernels behind m360_linear - the 256 x 256 tile kernel and the half-tile / double-accumulator
kernel (m360_linear_hd.hip.h) - through the diagnostics library's kernel switch: bit-identity of every element, median launch
time over a sweep of layer shapes, and (--ablate) the timing-only ablations of the half-tile kernel."""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mipnerf360_amd import _lib, ops  # noqa: E402


def timed(fn, reps=5, inner=10):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--ablate", action="store_true")
    ap.add_argument("--diag-lib", default="libm360_diag.so")
    ap.add_argument("--quick", action="store_true", help="three shapes only")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    diag = ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), args.diag_lib))
    vp, ci, cl = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
    diag.m360_diag_linear_hd.argtypes = [vp, cl, ci, vp, vp, ci, ci, ci, vp, ci, ci, vp, vp]
    diag.m360_linear_balanced.argtypes = [vp, cl, ci, vp, vp, ci, ci, ci, vp, ci, vp, vp]
    diag.m360_linear.argtypes = [vp, cl, ci, vp, vp, ci, ci, ci, vp, ci, vp]
    diag.m360_diag_force_linear_kernel.argtypes = [ci]
    diag.m360_diag_read_hd_stamps.argtypes = [vp, ci]
    big = 4096 * 128
    shapes = [(big, n, k, 1) for n in (256, 1024) for k in (64, 96, 128, 256, 384, 512, 1024)]
    shapes += [(128 * 37, 768, 96, 0), (128 * 3 + 5, 256, 64, 1), (128 * 513 + 77, 512, 160, 1), (256 * 96, 1024, 1024, 1),
               (256 * 64, 1024, 1024, 1), (256 * 100 + 128, 1024, 1024, 0), (8192 * 256, 1024, 1024, 1)]
    if args.quick:
        shapes = [(big, 1024, 1024, 1), (big, 256, 256, 1), (big, 1024, 96, 1)]
    out = []
    for M, n, k, act in shapes:
        g = torch.Generator(device=dev).manual_seed(M + n + k)
        x = torch.rand(M, k, device=dev, generator=g) * 2 - 1
        w = (torch.rand(n, k, device=dev, generator=g) * 2 - 1) * (6.0 / k) ** 0.5
        b = torch.rand(n, device=dev, generator=g) - 0.5
        wp, bp = ops.pack_linear(w, b, n, k)
        ys = {}

        def run(which, y):
            assert diag.m360_diag_force_linear_kernel(which) == 0
            rc = diag.m360_linear(x.data_ptr(), M, k, wp.data_ptr(), bp.data_ptr(), n, k, act, y.data_ptr(), n,
                                  torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc

        times = {}
        stable = True
        for name, which in (("full", 1), ("half", 2)):
            y = torch.full((M, n), float("nan"), device=dev)
            run(which, y)
            torch.cuda.synchronize()
            ys[name] = y.clone()
            times[name] = timed(lambda: run(which, y))
            stable = stable and bool(torch.equal(y, ys[name]))
        y_rule = ops.linear(x, wp, bp, act)   # the product library with its shape rule
        t_rule = timed(lambda: ops.linear(x, wp, bp, act, out=y_rule))
        queue = torch.zeros(16, dtype=torch.int32, device=dev)
        y_bal = torch.full((M, n), float("nan"), device=dev)

        def run_bal():   # half tiles + the dynamic pool for the last tiles (queue word zeroed before every launch)
            diag.m360_diag_force_linear_kernel(2)
            queue.zero_()
            rc = diag.m360_linear_balanced(x.data_ptr(), M, k, wp.data_ptr(), bp.data_ptr(), n, k, act, y_bal.data_ptr(), n,
                                           queue.data_ptr(), torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc

        run_bal()
        torch.cuda.synchronize()
        t_bal = timed(run_bal)
        same = bool(torch.equal(ys["full"], ys["half"])) and bool(torch.equal(ys["full"], y_rule)) and bool(torch.equal(ys["full"], y_bal))
        fl = 2.0 * M * n * k
        rec = {"M": M, "N": n, "K": k, "act": act, "bit_identical": same, "repeatable": stable,
               "full_ms": round(times["full"], 4), "half_ms": round(times["half"], 4), "rule_ms": round(t_rule, 4), "balanced_ms": round(t_bal, 4), "tickets": int(queue[0]),
               "full_tflops": round(fl / times["full"] / 1e9, 1), "half_tflops": round(fl / times["half"] / 1e9, 1)}
        if args.ablate and M % 128 == 0 and act == 1 and (n, k) in ((1024, 1024), (256, 256), (1024, 64)):
            y = torch.empty(M, n, device=dev)
            for abl in (0, "q", 1, 2, 7, 16):
                def f():
                    qp = None
                    if abl == "q":
                        queue.zero_()
                        qp = queue.data_ptr()
                    rc = diag.m360_diag_linear_hd(x.data_ptr(), M, k, wp.data_ptr(), bp.data_ptr(), n, k, act, y.data_ptr(), n,
                                                  0 if abl == "q" else abl, qp, torch.cuda.current_stream().cuda_stream)
                    assert rc == 0, rc
                f()
                torch.cuda.synchronize()
                rec[f"abl{abl}_ms"] = round(timed(f, reps=5, inner=max(10, int(150 / max(rec["half_ms"], 0.05)))), 4)  # ~0.75 s soak
                st = (ctypes.c_ulonglong * 2048)()
                assert diag.m360_diag_read_hd_stamps(st, 2048) == 0
                a = np.array(st[:1024], dtype=np.float64).reshape(256, 4)
                ph = np.array(st[1024:], dtype=np.float64).reshape(256, 4)
                tiles = np.maximum(a[:, 2] / (k // 32), 1)
                rec[f"abl{abl}_cycles_F_S_P"] = [round(float(np.median(ph[:, 0] / tiles)), 1), round(float(np.median(ph[:, 1] / tiles)), 1),
                                                round(float(np.median(ph[:, 2] / tiles / max(k // 32 - 2, 1))), 1)]
                if abl in (0, "q"):  # spread of the tile-loop duration over the workgroups (100 MHz ticks -> us), and by XCD (workgroup id % 8)
                    us = a[:, 1] / 100.0
                    sfx = "" if abl == 0 else "_balanced"
                    rec["loop_us_min_median_max" + sfx] = [round(float(us.min()), 1), round(float(np.median(us)), 1), round(float(us.max()), 1)]
                    rec["loop_us_median_by_xcd" + sfx] = [round(float(np.median(us[x::8])), 1) for x in range(8)]
                    rec["clock_ghz_by_xcd" + sfx] = [round(float(np.median(a[x::8, 0] / np.maximum(a[x::8, 1], 1)) * 0.1), 3) for x in range(8)]
                rec[f"abl{abl}_clock_ghz"] = round(float(np.median(a[:, 0] / np.maximum(a[:, 1], 1)) * 0.1), 3)
                rec[f"abl{abl}_cycles_per_kstep"] = round(float(np.median(a[:, 0] / np.maximum(a[:, 2], 1))), 1)
        print(json.dumps(rec), flush=True)
        out.append(rec)
        del x, y, ys, y_rule
    diag.m360_diag_force_linear_kernel(0)
    if args.out:
        with open(args.out, "w") as f:
            for r in out:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
