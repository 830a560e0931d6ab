#!/usr/bin/env python3
"""Per-tensor error of the bf16 mirrors' parameter gradients on fixture G13 (reference fp32 autograd) - which tensors carry the largest
relative error, and how the fp32 mirrors do on the same tensors."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from mipnerf360_amd import synthetic  # noqa: E402
from mipnerf360_amd.intern.loss import Loss_dist, Loss_nerf  # noqa: E402
from mipnerf360_amd.intern.ray import Rays  # noqa: E402
from mipnerf360_amd.model import mipNeRF360  # noqa: E402

dev = torch.device("cuda:0")
g = dict(np.load(os.path.join(ROOT, "tests", "golden", "g13_train_gradients.npz")))
for kind in ("lego", "garden"):
    B, n, wb = (int(v) for v in g[f"{kind}_cfg"])
    sd = {k[3:]: g[k] for k in g if k.startswith("sd.")}
    rays = Rays(*[torch.from_numpy(np.ascontiguousarray(g[f"{kind}_rays_{f}"])).float().to(dev) for f in synthetic.RAY_FIELDS])
    res = {}
    for dtype in ("fp32", "bf16"):
        m = mipNeRF360(randomized=False, num_samples=n, hidden_proposal=32, hidden_nerf=64, white_bkgd=bool(wb), device=dev, mlp_dtype=dtype).train()
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        with torch.no_grad():
            t_hat, w_hat = m.prop_net.forward(rays)
        rgb, _, _, _, fw, sv = m.nerf_net.forward(rays, t_vals=t_hat, coarse_weights=w_hat)
        ln, _ = Loss_nerf(input=rgb, target=torch.from_numpy(g[f"{kind}_pixels"]).to(dev))
        m.zero_grad()
        (ln + 0.01 * Loss_dist(s_vals=sv, weights=fw)).backward()
        res[dtype] = {name: p.grad.detach().float().cpu().numpy() for name, p in m.named_parameters() if name.startswith("nerf_net")}
    print(kind)
    for name in res["fp32"]:
        want = g[f"{kind}_nerfstep.{name}"].astype(np.float64)
        sc = np.abs(want).max()
        e32 = np.abs(res["fp32"][name] - want).max() / sc
        e16 = np.abs(res["bf16"][name] - want).max() / sc
        rms16 = np.sqrt(((res["bf16"][name] - want) ** 2).mean()) / max(np.sqrt((want ** 2).mean()), 1e-30)
        print(f"  {name:38s} scale {sc:9.3e}  fp32 max-rel {e32:8.2e}  bf16 max-rel {e16:8.2e}  bf16 rms-rel {rms16:8.2e}")
