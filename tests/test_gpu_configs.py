"""GPU tests of the BASELINE.json configurations that round 1 left without a `-m gpu` test:

  configs[2]  nerf_360/garden full 1237x822 frame, hierarchical 64+128 samples, 1 MI355X
  configs[3]  ray-sharded render on >1 ranks (here: two ranks of the real HIP renderer on the ONE GPU of the box)
  configs[4]  nerf_360/bicycle, 8192-ray batch, 256 samples/ray, bf16 MLP

At sizes the CPU oracle finishes in seconds the HIP path is compared with it (fp32: the stated 1e-4 tolerance; bf16:
<= 6e-3 against the oracle's bf16 emulation, <= 2e-2 and PSNR > 45 dB against the fp32 oracle); at full size through
size-independent properties (determinism, ranges, sortedness, ray-permutation equivariance, bit-identity of a sub-range
with the chunk loop).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from mipnerf360_amd import synthetic

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test needs a HIP device")
    return torch.device("cuda:0")


def dev_rays(d, dev):
    from mipnerf360_amd.intern.ray import Rays
    return Rays(*[torch.from_numpy(np.ascontiguousarray(d[k])).float().to(dev) for k in synthetic.RAY_FIELDS])


def make_model(dev, n, n_fine=None, mlp_dtype="fp32", hp=256, hn=1024, seed=0):
    from mipnerf360_amd.model import mipNeRF360
    sd = synthetic.make_state_dict(hp, hn, seed=seed)
    m = mipNeRF360(num_samples=n, hidden_proposal=hp, hidden_nerf=hn, device=dev, num_samples_fine=n_fine,
                   mlp_dtype=mlp_dtype)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.eval(), sd


def psnr(a, b):
    return -10.0 * np.log10(max(float(((a - b) ** 2).mean()), 1e-20))


# ------------------------------------------------------------------------------- configs[4]
def test_c5_shard_bf16_256_samples_vs_oracle(dev):
    """configs[4] as ONE of its 8 per-GPU shards: 1024 rays x 256 samples, full-width 256 / 1024 MLPs,
    mlp_dtype="bf16", against the oracle rounding at the same points (<= 6e-3) and against the fp32 oracle
    (<= 2e-2, PSNR > 45 dB)."""
    from oracle import ref_path as O
    B, N = 1024, 256
    m, sd = make_model(dev, N, mlp_dtype="bf16")
    r = synthetic.make_rays("garden", B, seed=31)
    with torch.no_grad():
        rgb, dist, acc = (t.cpu() for t in m(dev_rays(r, dev)))
    sdt = O.to_torch_state_dict(sd)
    emu = O.forward(O.rays_from_numpy(r), sdt, O.Hyper(num_samples=N, mlp_bf16=True))
    ref = O.forward(O.rays_from_numpy(r), sdt, O.Hyper(num_samples=N))
    assert float((rgb - emu[0]).abs().max()) <= 6e-3 and float((acc - emu[2]).abs().max()) <= 6e-3
    assert float((rgb - ref[0]).abs().max()) <= 2e-2 and float((acc - ref[2]).abs().max()) <= 2e-2
    assert psnr(rgb.numpy(), ref[0].numpy()) > 45.0
    assert float((dist - ref[1]).abs().max()) <= 2e-2


def test_c5_shard_fp32_256_samples_full_width_vs_oracle(dev):
    """the same shard shape in the exact-fp32 parity mode: the stated fp32 tolerance at 256 samples / full width"""
    from oracle import ref_path as O
    B, N = 256, 256
    m, sd = make_model(dev, N)
    r = synthetic.make_rays("garden", B, seed=32)
    with torch.no_grad():
        rgb, dist, acc = (t.cpu() for t in m(dev_rays(r, dev)))
    ref = O.forward(O.rays_from_numpy(r), O.to_torch_state_dict(sd), O.Hyper(num_samples=N))
    assert float((rgb - ref[0]).abs().max()) <= 1e-4 and float((acc - ref[2]).abs().max()) <= 1e-4
    assert bool(torch.all((dist - ref[1]).abs() <= 1e-4 * torch.clamp(ref[1].abs(), min=1.0)))


def test_c5_full_size_bf16_properties(dev):
    """configs[4] at its own size on one GPU: 8192 rays x 256 samples, bf16 MLP, full width - determinism, ranges,
    sorted resampled t, distance inside [t_0, t_N], ray-permutation equivariance (rays couple only through the
    permutation-invariant contraction norm; bf16 tolerance)."""
    from mipnerf360_amd.intern.ray import Rays
    B, N = 8192, 256
    m, _ = make_model(dev, N, mlp_dtype="bf16")
    rays = dev_rays(synthetic.make_rays("garden", B, seed=33), dev)
    with torch.no_grad():
        rgb, dist, acc = m(rays)
        rgb2, dist2, acc2 = m(rays)
    assert torch.equal(rgb, rgb2) and torch.equal(dist, dist2) and torch.equal(acc, acc2)
    assert torch.isfinite(rgb).all() and torch.isfinite(dist).all() and torch.isfinite(acc).all()
    assert float(acc.min()) >= 0 and float(acc.max()) <= 1 + 1e-5
    assert float(rgb.min()) >= -0.001 - 1e-6 and float(rgb.max()) <= 1.001 + 1e-6 and float(rgb.std()) > 1e-3
    tv = m.nerf_net.t_vals
    assert tv.shape == (B, N + 1) and m.nerf_net.fine_weights.shape == (B, N)
    assert torch.all(tv[:, 1:] >= tv[:, :-1])
    assert torch.all(dist >= tv[:, 0] - 2e-6) and torch.all(dist <= tv[:, -1])
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).to(dev)
    with torch.no_grad():
        rgb_p, dist_p, acc_p = m(Rays(*[f[perm].contiguous() for f in rays]))
    assert float((rgb_p - rgb[perm]).abs().max()) <= 1e-2 and float((acc_p - acc[perm]).abs().max()) <= 1e-2
    assert float((rgb_p - rgb[perm]).abs().mean()) <= 1e-4


def test_c5_full_size_fp32_properties(dev):
    """The exact-fp32 kernels at the largest single-GPU shape of BASELINE.json (configs[4]'s 8192 rays x 256 samples; the fp32
    activation buffers are [2 097 152, 1024] = exactly 2^31 elements each - the index arithmetic of the half-tile / persistent
    linear kernels, the encoder and the finishers past 32 bits): determinism, ranges, sorted resampled t, distance inside
    [t_0, t_N], ray-permutation equivariance, and PARITY: the first and the last 128 rays of the batch, rendered by the oracle with
    the whole batch's two contraction norms (what m360_*_forward_from_t take: intern/parameterization.py:23-29 spans all 8192
    rays), against the rows of the full-size forward - the stated fp32 tolerance (model.py:163-200)."""
    from mipnerf360_amd.intern.ray import Rays
    from oracle import ref_path as O
    B, N = 8192, 256
    m, sd = make_model(dev, N)
    r = synthetic.make_rays("garden", B, seed=33)
    rays = dev_rays(r, dev)
    with torch.no_grad():
        rgb, dist, acc = m(rays)
        tv = m.nerf_net.t_vals.clone()
        rgb2, dist2, acc2 = m(rays)
    assert torch.equal(rgb, rgb2) and torch.equal(dist, dist2) and torch.equal(acc, acc2)
    assert torch.isfinite(rgb).all() and torch.isfinite(dist).all() and torch.isfinite(acc).all()
    assert float(acc.min()) >= 0 and float(acc.max()) <= 1 + 1e-5
    assert float(rgb.min()) >= -0.001 - 1e-6 and float(rgb.max()) <= 1.001 + 1e-6 and float(rgb.std()) > 1e-3
    assert tv.shape == (B, N + 1) and torch.all(tv[:, 1:] >= tv[:, :-1])
    assert torch.all(dist >= tv[:, 0] - 2e-6) and torch.all(dist <= tv[:, -1])
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(0)).to(dev)
    with torch.no_grad():
        rgb_p, dist_p, acc_p = m(Rays(*[f[perm].contiguous() for f in rays]))
    assert float((rgb_p - rgb[perm]).abs().max()) <= 2e-5 and float((acc_p - acc[perm]).abs().max()) <= 2e-5
    # the two whole-batch norms, from the device (fp64 sums of squares), then the oracle on two 128-ray slices with them
    with torch.no_grad():
        t_hat = m.sharded_sample(rays)
        norm0 = m.sharded_sumsq(rays, t_hat).sqrt().float()
        _, t_new = m.sharded_prop(rays, t_hat, norm0)
        norm1 = m.sharded_sumsq(rays, t_new).sqrt().float()
        again = m.sharded_nerf(rays, t_new, norm1)
    assert float((again[0] - rgb).abs().max()) <= 2e-6             # the sharded entry points ARE the forward, up to the norm's bits
    o_sd, hp = O.to_torch_state_dict(sd), O.Hyper(num_samples=N)
    for sl in (slice(0, 128), slice(B - 128, B)):
        sub = O.rays_from_numpy({k: v[sl] for k, v in r.items()})
        with torch.no_grad():
            _, o_t_new = O.prop_forward_from_t(sub, o_sd, hp, t_hat[sl].cpu(), norm0.cpu())
            o_rgb, o_dist, o_acc = O.nerf_forward_from_t(sub, o_sd, hp, o_t_new, norm1.cpu())
        assert float((rgb[sl].cpu() - o_rgb).abs().max()) <= 1e-4 and float((acc[sl].cpu() - o_acc).abs().max()) <= 1e-4
        assert bool(torch.all((dist[sl].cpu() - o_dist).abs() <= 1e-4 * torch.clamp(o_dist.abs(), min=1.0)))


def test_bench_c5_named_workload(dev):
    """`bench.py --config c5`: the configs[4] shape as a named bench workload (dtype bf16), never the default line."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c5", "--steps", "3", "--warmup", "1"],
                         cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["dtype"] == "bf16" and line["config"]["name"] == "c5" and line["config"]["samples_per_ray"] == 256
    assert line["config"]["rays_per_gpu"] == 8192 and "256 samples" in line["metric"]
    assert line["value"] > 5e4 and line["roofline"]["peak"] == 2500.0 and 0.1 < line["roofline"]["frac"] < 1.0


# ------------------------------------------------------------------------------- configs[2]
@pytest.mark.parametrize("mlp_dtype", ["fp32", "bf16x3"])
def test_c3_hierarchical_64_128_full_width_vs_oracle(dev, mlp_dtype):
    """configs[2]'s sampling ("64+128": 64 proposal, 128 NeRF samples) at FULL width against the oracle's same
    extension (the reference itself cannot express unequal counts, intern/ray.py:147), 256 rays as one chunk; fp32 and bf16x3."""
    from oracle import ref_path as O
    B = 256
    m, sd = make_model(dev, 64, n_fine=128, mlp_dtype=mlp_dtype)
    r = synthetic.make_rays("garden", B, seed=34)
    with torch.no_grad():
        rgb, dist, acc = (t.cpu() for t in m(dev_rays(r, dev)))
    assert m.nerf_net.t_vals.shape == (B, 129)
    ref = O.forward(O.rays_from_numpy(r), O.to_torch_state_dict(sd), O.Hyper(num_samples=64, num_samples_fine=128))
    assert float((rgb - ref[0]).abs().max()) <= 1e-4 and float((acc - ref[2]).abs().max()) <= 1e-4
    assert bool(torch.all((dist - ref[1]).abs() <= 1e-4 * torch.clamp(ref[1].abs(), min=1.0)))


def test_c3_full_frame_render_view(dev):
    """configs[2] at its own size: one full 1237 x 822 frame, 64+128 samples, full width, rendered from a pose with
    `render_view` (rays generated on the device, 249 chunks of 4096).  Properties: shapes / dtypes of the reference's
    render_image, finite values, ranges; and bit-identity of two chunks (an interior one and the ragged last one)
    with the plain per-chunk forward on the same rays."""
    from mipnerf360_amd.intern.ray import Rays, generate_rays
    H, W, focal, chunks = 822, 1237, 1100.0, 4096
    m, _ = make_model(dev, 64, n_fine=128)
    pose = torch.tensor([[1.0, 0.0, 0.0, 0.1], [0.0, 1.0, 0.0, -0.05], [0.0, 0.0, 1.0, 0.2]])
    rgb8, dist, acc = m.render_view(pose, H, W, focal, 0.0, 1.0, ndc=True, chunks=chunks)
    assert rgb8.shape == (H, W, 3) and rgb8.dtype == np.uint8
    assert dist.shape == (H, W) and dist.dtype == np.float32 and acc.shape == (H, W) and acc.dtype == np.float32
    assert np.isfinite(dist).all() and np.isfinite(acc).all()
    assert acc.min() >= 0 and acc.max() <= 1 + 1e-5 and dist.min() >= 0 and dist.max() <= 1.0 + 1e-4
    assert rgb8.std() > 1.0  # not a constant image
    rays = generate_rays(pose.to(dev), H, W, focal, 0.0, 1.0, True)
    n = H * W
    n_chunks = (n + chunks - 1) // chunks
    assert n_chunks == 249
    from mipnerf360_amd import ops
    for c in (100, n_chunks - 1):
        lo, hi = c * chunks, min((c + 1) * chunks, n)
        with torch.no_grad():
            r, d, a = m(Rays(*[f[lo:hi].contiguous() for f in rays]))
        assert np.array_equal(d.cpu().numpy(), dist.reshape(-1)[lo:hi])
        assert np.array_equal(a.cpu().numpy(), acc.reshape(-1)[lo:hi])
        assert np.array_equal(ops.to8b(r).cpu().numpy(), rgb8.reshape(-1, 3)[lo:hi])


def test_render_image_with_tiny_chunks(dev):
    """chunks = 1, 2, 3 (more than 1024 contraction-norm groups per 4096-ray super-batch: the grouped launch is capped
    to 1024 groups) render like the one-launch-per-chunk loop, bit for bit."""
    from mipnerf360_amd.intern.ray import Rays
    m, _ = make_model(dev, 16, hp=32, hn=64, seed=2)
    h, w = 47, 53
    r = synthetic.make_rays("garden", h * w, seed=35)
    rays_cpu = Rays(*[torch.from_numpy(r[k]) for k in synthetic.RAY_FIELDS])
    for chunks in (1, 3):
        grouped = m.render_image(rays_cpu, h, w, chunks=chunks)
        m.super_batch_rays = chunks  # one launch per chunk
        loop = m.render_image(rays_cpu, h, w, chunks=chunks)
        m.super_batch_rays = 4096
        assert all(np.array_equal(a, b) for a, b in zip(grouped, loop)), chunks


# ------------------------------------------------------------------------------- configs[3]
@pytest.mark.parametrize("full_width", [False, True])
def test_two_ranks_of_the_hip_renderer_on_one_gpu(dev, full_width):
    """configs[3]: the real HIP renderer with world_size 2.  The box has one GPU and RCCL refuses two ranks on one
    device, so both ranks use cuda:0 and the collectives travel through gloo (host staged) - sharding, chunk
    partition, per-rank compute and the gather logic are the production code.  `render_image_sharded` must equal the
    single-process `render_image` bit for bit and `forward_sharded` the whole-batch forward to <= 2e-6."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29551" if full_width else "29550", os.path.join(ROOT, "tools", "dist_check.py"),
           "--backend", "gloo", "--same-gpu"] + (["--full-width"] if full_width else [])
    res = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert res.returncode == 0 and "dist_check world=2 sharded==single: True" in res.stdout, res.stdout[-3000:]


def _run_bench(*argv, timeout=900):
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    return res, (json.loads(lines[-1]) if lines else None)


def test_bench_launches_its_own_ranks_weak_scaling_line(dev):
    """`python bench.py --gpus 2` with no launcher around it (how the driver starts it): the parent starts two fresh rank
    processes, rank 0's single JSON line comes back through the parent.  One GPU here, so the diagnostics backend (gloo,
    ranks share cuda:0) stands in for RCCL: launcher, per-rank batches, PixelGather, replica check and the
    strong_scaling_frame leg are the production code."""
    res, line = _run_bench("--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--frame-size", "160x120")
    assert res.returncode == 0 and line is not None, (res.stdout[-1500:], res.stderr[-3000:])
    assert len([ln for ln in res.stdout.splitlines() if ln.strip()]) == 1, "exactly ONE line on stdout"
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak" and line["dtype"] == "f32"
    assert line["rccl"]["ranks"] == 2 and line["rccl"]["backend"] == "gloo"
    assert len(line["per_rank"]) == 2 and all(r["compute_ms_median"] > 0 for r in line["per_rank"])
    assert line["value"] > 0 and line["roofline"]["frac"] > 0.2     # two ranks share one GPU: no performance claim
    fr = line["strong_scaling_frame"]
    assert fr["finite"] and fr["rays_per_frame"] == 160 * 120 and fr["n_chunks"] == 5 and fr["chunks_per_rank"] == 3
    assert [r["rays"] for r in fr["per_rank"]] == [3 * 4096, 160 * 120 - 3 * 4096]
    assert abs(fr["partition_efficiency_bound"] - 160 * 120 / (2 * 3 * 4096)) < 1e-4
    assert "cpu_baseline" not in line  # N = 1 only


def test_bench_eight_ranks_share_the_gpu_over_gloo(dev):
    """The driver's N = 8 invocation (`python bench.py --gpus 8`), as far as ONE GPU can go: eight fresh rank processes through bench.py's own
    launcher, each with its own 4096 x 128 batch (weak line) and its block of whole chunks of one frame (configs[3]: 30 chunks -> 4 per rank,
    the last rank 2), pixel blocks gathered over gloo - the per-rank code, partition and gather logic of the RCCL run; no performance claim."""
    res, line = _run_bench("--gpus", "8", "--backend", "gloo", "--steps", "1", "--warmup", "1", "--frame-size", "400x300", timeout=900)
    assert res.returncode == 0 and line is not None, (res.stdout[-1500:], res.stderr[-3000:])
    assert line["n_gpus"] == 8 and line["rccl"]["ranks"] == 8 and len(line["per_rank"]) == 8 and line["scaling"] == "weak"
    assert line["config"]["global_rays"] == 8 * 4096 if "global_rays" in line["config"] else True
    fr = line["strong_scaling_frame"]
    assert fr["finite"] and fr["rays_per_frame"] == 400 * 300 and fr["n_chunks"] == 30 and fr["chunks_per_rank"] == 4
    assert [r["rays"] for r in fr["per_rank"]] == [4 * 4096] * 7 + [400 * 300 - 7 * 4 * 4096]
    assert abs(fr["partition_efficiency_bound"] - 400 * 300 / (8 * 4 * 4096)) < 1e-4


@pytest.mark.parametrize("bad_rank", [0, 1])
def test_bench_frame_leg_is_skipped_by_all_ranks_when_one_cannot_set_it_up(dev, bad_rank):
    """the frame leg used to be wrapped in a per-rank try / except - a rank that failed recorded an error and moved on
    while the others waited for it in the pixel all-gather: a hang instead of a failure.  Now the ranks agree on the set-up's outcome (one
    MIN all-reduce everybody takes part in) before the leg's first collective: either rank failing makes BOTH skip it, the run ends with
    its headline line and rc 0, within the timeout."""
    res, line = _run_bench("--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--frame-size", "160x120",
                           "--fail-frame-setup-on-rank", str(bad_rank), timeout=420)
    assert res.returncode == 0 and line is not None, (res.stdout[-1500:], res.stderr[-3000:])
    assert line["value"] > 0 and line["n_gpus"] == 2
    assert "error" in line["strong_scaling_frame"] and "rays_per_s" not in line["strong_scaling_frame"], line["strong_scaling_frame"]
    assert ("injected" in line["strong_scaling_frame"]["error"]) == (bad_rank == 0)   # rank 0's line: its own error, or "another rank"
    assert "injected by --fail-frame-setup-on-rank" in res.stderr


def test_bench_c4_frames_sharded_over_two_ranks(dev):
    """bench.py --config c4 (BASELINE configs[3]) through its own launcher: frames sharded by whole chunks, each rank
    generating the rays of its own span, overlapped and serial pixel gathers."""
    res, line = _run_bench("--gpus", "2", "--backend", "gloo", "--config", "c4", "--steps", "3", "--warmup", "1",
                           "--frame-size", "200x123")
    assert res.returncode == 0 and line is not None, (res.stdout[-1500:], res.stderr[-3000:])
    n = 200 * 123
    assert line["scaling"] == "strong" and line["n_gpus"] == 2 and line["config"]["name"] == "c4"
    assert line["config"]["rays_per_frame"] == n and line["config"]["n_chunks"] == 7 and line["config"]["chunks_per_rank"] == 4
    for mode in ("overlapped", "serial"):
        fr = line["frame"][mode]
        assert fr["finite"] and fr["frames"] == 3 and sum(r["rays"] for r in fr["per_rank"]) == n
        assert fr["rays_per_s"] > 0
    assert line["frame"]["overlapped"]["overlap"] and not line["frame"]["serial"]["overlap"]
    assert abs(line["value"] - line["frame"]["overlapped"]["rays_per_s"]) < 1.0
    assert line["roofline"] is not None and line["rccl"]["ranks"] == 2


def test_bench_single_gpu_frame_workload(dev):
    """--config c4 at N = 1 = BASELINE configs[2] (no collective): the same frame loop, one process."""
    res, line = _run_bench("--config", "c4", "--steps", "1", "--warmup", "1", "--frame-size", "300x137")
    assert res.returncode == 0 and line is not None, (res.stdout[-1500:], res.stderr[-3000:])
    assert line["n_gpus"] == 1 and line["rccl"] is None and line["frame"]["serial"] is None
    assert line["frame"]["overlapped"]["finite"] and line["frame"]["overlapped"]["per_rank"][0]["rays"] == 300 * 137


def test_model_on_explicit_device_without_set_device(dev):
    """mipNeRF360(device='cuda:0') must work whatever torch's current device/stream bookkeeping says: every C-ABI call
    runs under a device guard on the device its tensors live on, on that device's current stream (side stream too)."""
    from oracle import ref_path as O
    m, sd = make_model(dev, 16, hp=32, hn=64, seed=3)
    r = synthetic.make_rays("lego", 64, seed=36)
    rays = dev_rays(r, dev)
    ref = O.forward(O.rays_from_numpy(r), O.to_torch_state_dict(sd), O.Hyper(num_samples=16))
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side), torch.no_grad():
        a = m(rays)
    with torch.no_grad():
        b = m(rays)  # default stream, its own workspace: may overlap with the side-stream call
    torch.cuda.synchronize()
    for x, y, z in zip(a, b, ref):
        assert torch.equal(x, y) and float((x.cpu() - z).abs().max()) <= 1e-4
    from mipnerf360_amd import ops
    with pytest.raises(RuntimeError, match="CPU"):
        ops.call("m360_to8b", torch.zeros(4), 4, torch.zeros(4, dtype=torch.uint8), ops.STREAM)


def test_data_writes_are_seen_after_invalidate_or_in_training_mode(dev):
    """`p.data.mul_()` does not bump the version counter.  In eval mode the cached packing is reused until
    `invalidate_packed()`; in training mode every forward re-packs."""
    m, _ = make_model(dev, 16, hp=32, hn=64, seed=4)
    rays = dev_rays(synthetic.make_rays("garden", 32, seed=37), dev)
    with torch.no_grad():
        base = m(rays)[0].clone()
        m.nerf_net.final_color[0].weight.data.mul_(0.5)
        stale = m(rays)[0].clone()
        assert torch.equal(stale, base)  # documented limitation of the eval-mode cache
        m.invalidate_packed()
        fresh = m(rays)[0].clone()
        assert not torch.equal(fresh, base)
        m.train()
        m.nerf_net.final_color[0].weight.data.mul_(2.0)
        again = m(rays)[0].clone()
    assert float((again - base).abs().max()) <= 1e-6


# ------------------------------------------------------------------------------- fused last layer + heads
@pytest.mark.parametrize("B,N,width,heads", [(8, 128, 1024, 4), (6, 100, 1024, 4), (16, 64, 256, 1), (5, 77, 256, 1),
                                              (3, 50, 128, 4), (4, 64, 512, 4)])
def test_fused_last_layer_heads_match_the_unfused_path(dev, B, N, width, heads):
    """the last hidden layer's epilogue forms the head products (m360_linear_heads) and the finisher
    only adds the per-wave-tile partial sums (m360_*_finish_fused).  Against the unfused pair m360_linear +
    m360_*_finish on the same inputs: layer output bit-identical where it is stored, rendered values within 1e-6 (the
    head dot product is summed in a different order); ragged row counts (tail rows take the unfused route), widths
    that cannot fuse (128: fused_rows = 0) and store_y = 0 (the activation is never written) included."""
    from mipnerf360_amd import _lib, ops
    g = torch.Generator(device=dev).manual_seed(B * N + width)
    S = B * N
    x = torch.rand(S, width, device=dev, generator=g)
    w = (torch.rand(width, width, device=dev, generator=g) * 2 - 1) * (6.0 / width) ** 0.5
    b = torch.rand(width, device=dev, generator=g) - 0.5
    hw = (torch.rand(heads, width, device=dev, generator=g) * 2 - 1) * (6.0 / width) ** 0.5
    hb = torch.rand(heads, device=dev, generator=g) - 0.5
    t = torch.sort(torch.rand(B, N + 1, device=dev, generator=g), dim=1).values
    dirs = torch.rand(B, 3, device=dev, generator=g) + 0.5
    wp, bp = ops.pack_linear(w, b)
    y_ref = ops.linear(x, wp, bp, _lib.ACT_SIGMOID)
    y1, part1, fused = ops.linear_heads(x, wp, bp, hw, store_y=True)
    y0, part0, fused0 = ops.linear_heads(x, wp, bp, hw, store_y=False)
    expect = (S // 256) * 256 if width % 256 == 0 else 0
    assert fused == fused0 == expect
    assert torch.equal(y1, y_ref)                               # stored: bit-identical to the plain layer
    assert torch.equal(y0[fused:], y_ref[fused:])               # tail rows always go through y
    assert fused == 0 or float(y0[:fused].abs().max()) == 0.0   # fused rows never written when store_y = 0
    assert torch.equal(part0, part1)                            # same partial sums with and without the store
    if fused:
        raw = part1[:fused].sum(1) + hb
        want = y_ref[:fused].double() @ hw.double().T + hb.double()
        assert float((raw.double() - want).abs().max()) <= 2e-5
    for y, part in ((y1, part1), (y0, part0)):
        if heads == 4:
            a = ops.nerf_finish_fused(y, part, fused, hw, hb, -1.0, 0.001, t, dirs, False)
            r = ops.nerf_finish(y_ref, hw, hb, -1.0, 0.001, t, dirs, False)
        else:
            a = ops.prop_finish_fused(y, part, fused, hw, hb, -1.0, t, dirs, 0.01)
            r = ops.prop_finish(y_ref, hw, hb, -1.0, t, dirs, 0.01)
        for u, v in zip(a, r):
            assert float((u - v).abs().max()) <= 1e-6, float((u - v).abs().max())
    for _ in range(3):  # deterministic: fixed summation order, no atomics
        assert torch.equal(ops.linear_heads(x, wp, bp, hw, store_y=False)[1], part0)
    # rows the fused layer does not cover are summed by the finisher in the SAME order as the fused epilogue: a row's result
    # does not depend on which side of `fused_rows` it fell (chunk alone == chunk inside a larger launch, bit for bit)
    if width % 256 == 0 and width <= 1024:
        if heads == 4:
            fused_out = ops.nerf_finish_fused(y1, part1, fused, hw, hb, -1.0, 0.001, t, dirs, True)
            tail_out = ops.nerf_finish_fused(y_ref, part1, 0, hw, hb, -1.0, 0.001, t, dirs, True)
        else:
            fused_out = ops.prop_finish_fused(y1, part1, fused, hw, hb, -1.0, t, dirs, 0.01)
            tail_out = ops.prop_finish_fused(y_ref, part1, 0, hw, hb, -1.0, t, dirs, 0.01)
        for u, v in zip(fused_out, tail_out):
            assert torch.equal(u, v)


def test_grouped_chunks_bit_identical_at_fusable_widths(dev):
    """Full-width MLPs (the last hidden layer of each stage is fused with its heads for whole 256-row tiles) with a sample
    count and chunk size whose products are NOT multiples of 256: a chunk rendered by its own launch has a ragged tail of
    unfused rows in other places than the same chunk inside a many-chunk launch.  The finisher adds those rows' head
    products in the fused epilogue's order, so both renders are still bit-identical."""
    m, _ = make_model(dev, 33, seed=12)
    r = synthetic.make_rays("garden", 700, seed=38)
    r["origins"] = r["origins"] * 3.0
    rays = dev_rays(r, dev)
    m.super_batch_rays = 512
    a = [t.clone() for t in m.render_rays(rays, 100)]      # 5 chunks per launch (500 rays x 33 samples), then 200 rays
    m.super_batch_rays = 0
    b = m.render_rays(rays, 100)                           # one launch sequence per chunk (3300 rows: 12 tiles + 228 tail rows)
    m.super_batch_rays = 4096
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_bench_default_line_carries_every_single_gpu_config(dev):
    """the DEFAULT `python bench.py` line (what the driver records) carries a number for every single-GPU
    configuration of BASELINE.json - the headline (configs[1], fp32) untouched, `strong_scaling_frame` at 64 proposal + 128 NeRF samples
    (configs[2] as written; here a small diagnostic frame), and `named_workloads`: configs[4]'s shape in bf16, configs[1] in bf16 and
    bf16x3, one train.py iteration - each with its own dtype, rays/s, ms/step and roofline fraction.  (Short: 3 steps, no CPU pass.)"""
    res, line = _run_bench("--steps", "3", "--warmup", "1", "--cpu-rays", "0", "--frame-size", "200x160")
    assert res.returncode == 0 and line is not None, (res.stdout[-1500:], res.stderr[-3000:])
    assert line["dtype"] == "f32" and line["config"]["name"] == "c2" and line["n_gpus"] == 1 and line["vs_baseline"] is None
    assert line["metric"] == "rendered rays/sec at 128 samples/ray" and line["unit"] == "rays/s" and line["higher_is_better"] is True
    assert line["roofline"]["bound"] == "mfma" and 0.9 < line["roofline"]["frac"] < 1.0 and line["value"] > 6e4
    for k in ("encode_features", "prop_finish", "nerf_finish"):
        assert 0 < line["hbm_kernels"][k]["frac_of_8TBps"] <= 1.0
        assert abs(line["hbm_kernels"][k]["frac_of_measured_copy_6p29TBps"] - line["hbm_kernels"][k]["frac_of_8TBps"] * 8000 / 6290) < 2e-3
    fr = line["strong_scaling_frame"]
    assert "error" not in fr and not any("error" in v for v in line["named_workloads"].values() if isinstance(v, dict))  # (the extra legs are fail-soft)
    assert fr["finite"] and "64 proposal + 128 NeRF" in fr["samples_per_ray"] and fr["flops_per_ray"] == 423424 * 64 + 14807040 * 128
    nw = line["named_workloads"]
    assert set(nw) >= {"c5_bf16", "c2_bf16", "c2_bf16x3", "c2_training_iteration", "c2_training_iteration_bf16", "c3_frame_64+128", "seconds"}
    assert nw["c5_bf16"]["dtype"] == "bf16" and nw["c5_bf16"]["config"] == "c5" and "8192 rays x 256 samples" in nw["c5_bf16"]["workload"]
    assert nw["c2_bf16"]["dtype"] == "bf16" and nw["c2_bf16x3"]["dtype"] == "bf16x3"
    for k in ("c5_bf16", "c2_bf16", "c2_bf16x3"):
        assert nw[k]["finite"] and nw[k]["rays_per_s"] > 5e4 and 0.3 < nw[k]["roofline"]["frac"] < 1.0 and nw[k]["roofline"]["peak"] == 2500.0
    assert nw["c2_bf16"]["rays_per_s"] > 3 * line["value"] and nw["c2_bf16x3"]["rays_per_s"] > 2 * line["value"]
    tr = nw["c2_training_iteration"]
    assert tr["dtype"] == "f32" and 100 < tr["iteration_ms"] < 1000 and tr["wgrad_1024x1024"]["frac"] > 0.7 and tr["dgrad_1024x1024"]["frac"] > 0.7
    tb = nw["c2_training_iteration_bf16"]   # round 5: row f3 in the precision configs[4] runs at
    assert tb["dtype"] == "bf16" and tb["finite"] and tb["iteration_ms"] < 0.5 * tr["iteration_ms"] and tb["wgrad_1024x1024"]["frac"] > 0.15
    for k in ("c5_bf16", "c2_bf16"):        # the layer chain's self-checks, in the record
        assert nw[k]["chain"]["launches"] > 0 and nw[k]["chain"]["chain_error"] is False
    fr128 = nw["c1_frame_chunks128"]        # the reference's CLI default chunk size, through the grouped-chunk path
    assert fr128["finite"] and fr128["chunks"] == 128 and fr128["rays_per_s"] > 5e4
    assert nw["seconds"] < 45
