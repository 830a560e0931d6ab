#!/usr/bin/env python3
"""A/B of the half-tile / double-accumulator fp32 kernel (libm360_diag.so: m360_diag_linear_hd) against the product
m360_linear on the layer shapes of the path: bit-identity of every element and median launch time."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mipnerf360_amd import _lib, ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    diag = ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libm360_diag.so"))
    vp = ctypes.c_void_p
    diag.m360_diag_linear_hd.argtypes = [vp, ctypes.c_long, ctypes.c_int, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp]
    shapes = [(4096 * 128, 1024, 1024, 1), (4096 * 128, 1024, 64, 1), (4096 * 128, 256, 256, 1), (4096 * 128, 256, 64, 1),
              (128 * 37, 768, 96, 0), (128 * 3, 256, 64, 1), (128 * 513, 512, 160, 1)]
    out = []
    for M, n, k, act in shapes:
        g = torch.Generator(device=dev).manual_seed(M + n + k)
        x = torch.rand(M, k, device=dev, generator=g) * 2 - 1
        w = (torch.rand(n, k, device=dev, generator=g) * 2 - 1) * (6.0 / k) ** 0.5
        b = torch.rand(n, device=dev, generator=g) - 0.5
        wp, bp = ops.pack_linear(w, b, n, k)
        y_ref = ops.linear(x, wp, bp, act)
        y = torch.full((M, n), float("nan"), device=dev)

        def run_hd():
            rc = diag.m360_diag_linear_hd(x.data_ptr(), M, k, wp.data_ptr(), bp.data_ptr(), n, k, act, y.data_ptr(), n,
                                          torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc

        run_hd()
        torch.cuda.synchronize()
        same = bool(torch.equal(y, y_ref))
        nbad = int((y != y_ref).sum()) if not same else 0
        for _ in range(4):
            y.fill_(float("nan"))
            run_hd()
            torch.cuda.synchronize()
            same = same and bool(torch.equal(y, y_ref))
        times = {}
        for name, fn in (("product", lambda: ops.linear(x, wp, bp, act, out=y)), ("hd", run_hd)):
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10)
            times[name] = float(np.median(ts))
        fl = 2.0 * M * n * k
        rec = {"M": M, "N": n, "K": k, "act": act, "bit_identical": same, "mismatches": nbad,
               "product_ms": round(times["product"], 4), "hd_ms": round(times["hd"], 4),
               "product_tflops": round(fl / times["product"] / 1e9, 1), "hd_tflops": round(fl / times["hd"] / 1e9, 1)}
        print(json.dumps(rec), flush=True)
        out.append(rec)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            for r in out:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
