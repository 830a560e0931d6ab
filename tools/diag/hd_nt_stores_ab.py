import ctypes, os, sys, json, numpy as np, torch
sys.path.insert(0, '/root/repo')
from mipnerf360_amd import _lib, ops
dev = torch.device("cuda:0")
diag = ctypes.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libm360_diag.so"))
vp = ctypes.c_void_p
diag.m360_diag_linear_hd.argtypes = [vp, ctypes.c_long, ctypes.c_int, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, ctypes.c_int, vp, vp]
for (M, n, k) in ((524288, 1024, 1024), (524288, 256, 256), (524288, 1024, 64)):
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.rand(M, k, device=dev, generator=g) * 2 - 1
    w = (torch.rand(n, k, device=dev, generator=g) * 2 - 1) * (6.0 / k) ** 0.5
    b = torch.rand(n, device=dev, generator=g) - 0.5
    wp, bp = ops.pack_linear(w, b, n, k)
    y = torch.empty(M, n, device=dev)
    res = {}
    outs = {}
    for rep in range(2):
        for abl in (0, 128):
            def f():
                rc = diag.m360_diag_linear_hd(x.data_ptr(), M, k, wp.data_ptr(), bp.data_ptr(), n, k, 1, y.data_ptr(), n, abl, None, torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc
            f(); torch.cuda.synchronize()
            outs[abl] = y.clone()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): f()
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10)
            res.setdefault(f"abl{abl}_ms", []).append(round(float(np.median(ts)), 4))
    res.update(M=M, N=n, K=k, equal=bool(torch.equal(outs[0], outs[128])))
    print(json.dumps(res), flush=True)
