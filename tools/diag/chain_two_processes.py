#!/usr/bin/env python3
"""Two processes on ONE GPU, both running bf16 forwards (the hidden-layer chain wants all 256 CUs for its 256 workgroups: with two tenants
its workgroups are not all resident).  What must hold: every forward returns (bounded waits: no hang), a forward whose chain gave up says
so (model.chain_error()), and forwards that did not are bit-identical to the single-tenant result."""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

if len(sys.argv) > 1 and sys.argv[1] == "worker":
    import torch
    from mipnerf360_amd import synthetic
    from mipnerf360_amd.intern.ray import Rays
    from mipnerf360_amd.model import mipNeRF360
    dev = torch.device("cuda:0")
    sd = synthetic.make_state_dict(256, 1024, seed=0)
    r = synthetic.make_rays("garden", 4096, seed=1)
    rays = Rays(*[torch.from_numpy(r[k]).to(dev) for k in synthetic.RAY_FIELDS])
    m = mipNeRF360(num_samples=128, device=dev, mlp_dtype="bf16")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m.eval()
    n = int(sys.argv[2])
    errs, t0 = 0, time.perf_counter()
    sums = []
    with torch.no_grad():
        for i in range(n):
            out = m(rays)
            bad = m.chain_error()
            errs += int(bad)
            if not bad:
                sums.append(float(out[0].double().sum()))
    torch.cuda.synchronize()
    print(json.dumps({"forwards": n, "chain_errors": errs, "seconds": round(time.perf_counter() - t0, 3), "distinct_results_among_good_forwards": len(set(sums)),
                      "rgb_sum": sums[0] if sums else None}), flush=True)
    sys.exit(0)

me = os.path.abspath(__file__)
solo = subprocess.run([sys.executable, me, "worker", "20"], capture_output=True, text=True, timeout=600)
print("solo:", solo.stdout.strip().splitlines()[-1] if solo.stdout.strip() else solo.stderr[-500:])
ps = [subprocess.Popen([sys.executable, me, "worker", "20"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(2)]
for i, p in enumerate(ps):
    try:
        out, err = p.communicate(timeout=900)
        print(f"tenant {i}:", out.strip().splitlines()[-1] if out.strip() else err[-500:])
    except subprocess.TimeoutExpired:
        p.kill()
        print(f"tenant {i}: TIMEOUT")
