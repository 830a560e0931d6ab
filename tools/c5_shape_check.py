import sys, time, torch
sys.path.insert(0, '.')
from mipnerf360_amd import synthetic
from mipnerf360_amd.intern.ray import Rays
from mipnerf360_amd.model import mipNeRF360
dev = torch.device('cuda:0')
sd = {k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(256, 1024, seed=0).items()}
r = synthetic.make_rays('garden', 8192, seed=1)
rays = Rays(*[torch.from_numpy(r[k]).to(dev) for k in synthetic.RAY_FIELDS])
for dt in ('bf16', 'fp32'):
    m = mipNeRF360(num_samples=256, hidden_proposal=256, hidden_nerf=1024, device=dev, mlp_dtype=dt)
    m.load_state_dict(sd)
    torch.set_grad_enabled(False)  # rendering (with grad enabled the mirrors keep a training tape)
    for _ in range(2): out = m(rays)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5 if dt == 'bf16' else 2
    for _ in range(n): out = m(rays)
    torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / n
    print(f"configs[4] shape 8192 rays x 256 samples, {dt}: {dtm*1e3:.1f} ms/step = {8192/dtm:.0f} rays/s, finite={bool(torch.isfinite(out[0]).all())}")
    del m
