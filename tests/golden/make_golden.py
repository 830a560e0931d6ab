#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE itself.

Run in the authoring container only (needs /root/reference, CPU is enough):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference (zhangkai0425/mipnerf360, pure PyTorch) is imported unmodified
from /root/reference with device=cpu; this script only calls its public
functions on seeded inputs and stores inputs + outputs as small .npz files.
Nothing of the reference's source travels: fixtures are data.

Because the reference's `g()` mutates near/far/t_vals in place
(intern/parameterization.py:15-21) every call below receives fresh clones and
the fixture stores the ORIGINAL (un-mutated) inputs.

Fixture ids follow SURVEY.md §8c (G1..G9).
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(1, ROOT)

import torch  # noqa: E402

import model as ref_model  # noqa: E402  (the reference)
from intern import encoding as ref_enc  # noqa: E402
from intern import parameterization as ref_par  # noqa: E402
from intern import ray as ref_ray  # noqa: E402

from mipnerf360_amd import synthetic  # noqa: E402  (build-owned generators)

CPU = torch.device("cpu")
torch.set_num_threads(8)


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x)).float().clone()


def T64(x):
    return torch.from_numpy(np.ascontiguousarray(x)).double().clone()


class reference_in_fp64:
    """Run the reference in double precision: its fp32-ness comes from torch's default dtype alone (`Jf = torch.zeros(..)`
    in gaussian_contract, `torch.linspace`, `torch.eye`: intern/parameterization.py:75,57, intern/ray.py:99), so with the
    default switched and fp64 inputs the SAME code evaluates the same formulas in fp64.  Used to measure how far the
    reference's own fp32 covariances are from their exact values (the t_var cancellation, parameterization.py:102-105)."""

    def __enter__(self):
        torch.set_default_dtype(torch.float64)

    def __exit__(self, *a):
        torch.set_default_dtype(torch.float32)


def N(x):
    return x.detach().cpu().numpy().copy()


def ref_rays(d):
    return ref_ray.Rays(*[T(d[k]) for k in synthetic.RAY_FIELDS])


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ----------------------------------------------------------------------------- G1 + G2
def g1_g2():
    out = {}
    for kind in ("lego", "garden"):
        for n in (8, 64, 128):
            r = synthetic.make_rays(kind, 6, seed=11)
            t0 = time.time()
            t_vals, (means, covs) = ref_ray.sample_along_rays(
                T(r["origins"]), T(r["directions"]), T(r["radii"]), n, T(r["near"]), T(r["far"]), False)
            key = f"{kind}_{n}"
            out[key + "_t"] = N(t_vals)
            # the reference's own conical_frustum_to_gaussian + gaussian_contract + origin shift (para_rays) for EVERY
            # sample count, so the near = 0 / N = 128 near-denormal t_var / r_var are pinned directly (768 Jacobian calls)
            out[key + "_means"] = N(means)
            out[key + "_covs"] = N(covs)
            with reference_in_fp64():  # the same call in double precision, from the same fp32 ray values
                _, (means64, covs64) = ref_ray.sample_along_rays(
                    T64(r["origins"]), T64(r["directions"]), T64(r["radii"]), n, T64(r["near"]), T64(r["far"]), False)
            assert means64.dtype == torch.float64 and covs64.dtype == torch.float64
            out[key + "_means64"], out[key + "_covs64"] = N(means64), N(covs64)
            # pre-contraction lift (G2)
            # the frustum moments are only INPUTS of the reference's gaussian_to_xyz here; they
            # are pinned end-to-end by the contracted means/covs of sample_along_rays above
            tv = t_vals.clone()
            d = T(r["directions"])
            mu = (tv[..., :-1] + tv[..., 1:]) / 2
            hw = (tv[..., 1:] - tv[..., :-1]) / 2
            t_mean = mu + (2 * mu * hw ** 2) / (3 * mu ** 2 + hw ** 2)
            t_var = (hw ** 2) / 3 - (4 / 15) * ((hw ** 4 * (12 * mu ** 2 - hw ** 2)) / (3 * mu ** 2 + hw ** 2) ** 2)
            r_var = T(r["radii"]) ** 2 * ((mu ** 2) / 4 + (5 / 12) * hw ** 2 - 4 / 15 * (hw ** 4) / (3 * mu ** 2 + hw ** 2))
            xyz_mean, xyz_cov = ref_par.gaussian_to_xyz(d, t_mean, t_var, r_var, diag=False)
            out[key + "_tmean"], out[key + "_tvar"], out[key + "_rvar"] = N(t_mean), N(t_var), N(r_var)
            out[key + "_xyzmean"], out[key + "_xyzcov"] = N(xyz_mean), N(xyz_cov)
            print(f"  G1/G2 {key}: {time.time() - t0:.2f}s")
        for k in synthetic.RAY_FIELDS:
            out[f"{kind}_{k}"] = synthetic.make_rays(kind, 6, seed=11)[k]
    save("g1_g2_sampling", **out)


# ----------------------------------------------------------------------------- G3
def g3():
    g = np.random.Generator(np.random.PCG64(33))
    out = {}

    def spd(shape):
        a = g.normal(size=shape + (3, 3)).astype(np.float32) * 0.1
        return a @ np.swapaxes(a, -1, -2) + 1e-3 * np.eye(3, dtype=np.float32)

    cases = {
        "big": g.normal(size=(8, 16, 3)).astype(np.float32) * 2.0,          # ||M||_F >> 1 -> J = I
        "tiny": g.normal(size=(1, 2, 3)).astype(np.float32) * 40.0,         # contracted points with |y| > 1 -> J != I
        "inside": g.normal(size=(2, 3, 3)).astype(np.float32) * 0.05,       # ||M||_F <= 1 -> identity
    }
    for name, mean in cases.items():
        cov = spd(mean.shape[:2])
        m, c = ref_par.gaussian_contract(T(mean), T(cov))
        out[name + "_mean_in"], out[name + "_cov_in"] = mean, cov
        out[name + "_mean_out"], out[name + "_cov_out"] = N(m), N(c)
    save("g3_contract", **out)


# ----------------------------------------------------------------------------- G4
def g4():
    g = np.random.Generator(np.random.PCG64(44))
    mean = g.normal(size=(5, 7, 3)).astype(np.float32)
    a = g.normal(size=(5, 7, 3, 3)).astype(np.float32) * 0.3
    cov = a @ np.swapaxes(a, -1, -2)
    pe = ref_enc.PositionalEncoding()
    enc = pe(T(mean), T(cov))
    v = g.normal(size=(12, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    v = v.astype(np.float32)
    # edge rows: x ~ 0 (atan blow-up guarded by +1e-6), z = +-1
    v[0] = [0.0, 1.0, 0.0]
    v[1] = [0.0, 0.0, 1.0]
    v[2] = [0.0, 0.0, -1.0]
    v[3] = [-1e-6, 0.6, 0.8]
    out = dict(mean=mean, cov=cov, ipe=N(enc), viewdirs=v)
    for lo, hi in ((0, 4), (1, 3)):
        vd = ref_enc.ViewdirectionEncoding(lo, hi)
        out[f"vd_{lo}_{hi}"] = N(vd(T(v)))
    save("g4_encoding", **out)


# ----------------------------------------------------------------------------- G5
def g5():
    g = np.random.Generator(np.random.PCG64(55))
    B, n = 7, 32
    out = {}
    for kind in ("lego", "garden"):
        r = synthetic.make_rays(kind, B, seed=5)
        t_vals = ref_ray.sample_along_rays(T(r["origins"]), T(r["directions"]), T(r["radii"]), 2,
                                           T(r["near"]), T(r["far"]), False)[0]  # cheap call, only for near/far eps
        s = np.sort(g.uniform(0, 1, size=(B, n + 1)), axis=1).astype(np.float32)
        lo, hi = float(r["near"][0, 0]) + 1e-3, float(r["far"][0, 0])
        t = (lo + s * (hi - lo)).astype(np.float32)
        density = (g.gamma(0.6, 4.0, size=(B, n, 1))).astype(np.float32)
        density[0] = 0.0            # zero-density ray -> acc 0, distance NaN -> nan_to_num -> clamp
        density[1] = 1e4            # saturated ray
        rgb = g.uniform(0, 1, size=(B, n, 3)).astype(np.float32)
        dirs = r["directions"]
        pn = ref_model.prop_net(num_samples=n, hidden_proposal=8, device=CPU)
        w = pn.density_to_weight(T(t), T(density), T(dirs))
        out[f"{kind}_t"], out[f"{kind}_density"], out[f"{kind}_rgb"], out[f"{kind}_dirs"] = t, density, rgb, dirs
        out[f"{kind}_w"] = N(w)
        for wb in (False, True):
            c, d, a, ww = ref_ray.volumetric_rendering(T(rgb), T(density), T(t), T(dirs), wb)
            tag = f"{kind}_wb{int(wb)}"
            out[tag + "_rgb"], out[tag + "_dist"], out[tag + "_acc"], out[tag + "_w"] = N(c), N(d), N(a), N(ww)
    save("g5_weights_composite", **out)


# ----------------------------------------------------------------------------- G6
def g6():
    g = np.random.Generator(np.random.PCG64(66))
    out = {}
    B, n = 6, 48
    t = np.sort(g.uniform(2, 6, size=(B, n + 1)), axis=1).astype(np.float32)
    w = g.uniform(0, 1, size=(B, n)).astype(np.float32)
    w[0] = 1.0 / n                                  # uniform
    w[1] = 0.0
    w[1, 17] = 1.0                                  # single peak
    w[2] = 0.0                                      # all-zero -> eps padding branch
    w[3] = np.exp(-0.5 * ((np.arange(n) - 30) / 2.0) ** 2)
    out["t"], out["w"] = t, w
    # bare pdf sampler (weights are modified in place by the reference -> clone)
    for ns in (n + 1, 16):
        smp = ref_ray.sorted_piecewise_constant_pdf(T(t), T(w) + 0.01, ns, randomized=False)
        out[f"pdf_samples_{ns}"] = N(smp)
    out["pdf_zero_samples"] = N(ref_ray.sorted_piecewise_constant_pdf(T(t), T(np.zeros_like(w)), n + 1, randomized=False))
    # blur + padding + sampler + gaussians (small: python jacobian loop)
    r = synthetic.make_rays("lego", B, seed=6)
    for pad in (0.01, 0.0):
        new_t, (means, covs) = ref_ray.resample_along_rays(T(r["origins"]), T(r["directions"]), T(r["radii"]),
                                                           T(t), T(w), False, pad)
        out[f"resample_t_pad{pad}"] = N(new_t)
        if pad == 0.01:
            out["resample_means"], out["resample_covs"] = N(means), N(covs)
            with reference_in_fp64():
                t64, (means64, covs64) = ref_ray.resample_along_rays(T64(r["origins"]), T64(r["directions"]), T64(r["radii"]),
                                                                     T64(t), T64(w), False, pad)
            out["resample_t64"], out["resample_means64"], out["resample_covs64"] = N(t64), N(means64), N(covs64)
    for k in synthetic.RAY_FIELDS:
        out["rays_" + k] = r[k]
    save("g6_resample", **out)


# ----------------------------------------------------------------------------- G7 / G8 / G9 helpers
def build_ref_model(sd_np, n, hp, hn, white_bkgd, **kw):
    m = ref_model.mipNeRF360(randomized=False, num_samples=n, hidden_proposal=hp, hidden_nerf=hn,
                             white_bkgd=white_bkgd, device=CPU, **kw)
    m.load_state_dict({k: T(v) for k, v in sd_np.items()})
    return m


def g7():
    out = {}
    hp_, hn_ = 32, 64
    sd = synthetic.make_state_dict(hp_, hn_, seed=7)
    for k, v in sd.items():
        out["sd." + k] = v
    for kind, B, n, wb in (("lego", 12, 16, True), ("garden", 10, 24, False)):
        r = synthetic.make_rays(kind, B, seed=70 + n)
        m = build_ref_model(sd, n, hp_, hn_, wb)
        t0 = time.time()
        with torch.no_grad():
            rays = ref_rays(r)
            t_hat, w_hat = m.prop_net.forward(rays)
            t_hat_np, w_hat_np = N(t_hat), N(w_hat)
            o = m.nerf_net.forward(rays, t_vals=t_hat, coarse_weights=w_hat)
        tag = f"{kind}"
        for k in synthetic.RAY_FIELDS:
            out[f"{tag}_rays_{k}"] = r[k]
        out[tag + "_cfg"] = np.array([B, n, int(wb)])
        out[tag + "_t_hat"], out[tag + "_w_hat"] = t_hat_np, w_hat_np
        for nm, v in zip(("rgb", "dist", "acc", "t_vals", "fine_w", "s_vals"), o):
            out[f"{tag}_{nm}"] = N(v)
        print(f"  G7 {kind}: {time.time() - t0:.1f}s")
    save("g7_stages_small", **out)


def g8():
    out = {"weights_seed": np.array([0]), "rays_seed": np.array([1])}
    sd = synthetic.make_state_dict(256, 1024, seed=0)
    for kind, B, n, wb in (("lego", 256, 64, True), ("garden", 256, 128, False)):
        r = synthetic.make_rays(kind, B, seed=1)
        m = build_ref_model(sd, n, 256, 1024, wb)
        t0 = time.time()
        with torch.no_grad():
            rgb, dist, acc = m(ref_rays(r))
        out[f"{kind}_{n}_rgb"], out[f"{kind}_{n}_dist"], out[f"{kind}_{n}_acc"] = N(rgb), N(dist), N(acc)
        out[f"{kind}_{n}_cfg"] = np.array([B, n, int(wb)])
        print(f"  G8 {kind} B={B} N={n}: {time.time() - t0:.1f}s  ({B / (time.time() - t0):.1f} rays/s reference CPU)")
    save("g8_end_to_end_fullwidth", **out)


def g9():
    import contextlib
    import io
    out = {}
    hp_, hn_, n = 32, 64, 16
    h, w = 24, 32
    sd = synthetic.make_state_dict(hp_, hn_, seed=9)
    for k, v in sd.items():
        out["sd." + k] = v
    r = synthetic.make_rays("garden", h * w, seed=9)
    for k in synthetic.RAY_FIELDS:
        out["rays_" + k] = r[k]
    out["cfg"] = np.array([h, w, n])
    m = build_ref_model(sd, n, hp_, hn_, False)
    for chunks in (128, 4096):
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            rgb8, dist, acc = m.render_image(ref_rays(r), h, w, chunks=chunks)
        assert rgb8.dtype == np.uint8 and dist.dtype == np.float32
        out[f"c{chunks}_rgb8"], out[f"c{chunks}_dist"], out[f"c{chunks}_acc"] = rgb8, dist, acc
        print(f"  G9 chunks={chunks}: {time.time() - t0:.1f}s")
    save("g9_render_image", **out)


def g10():
    """Ray generation: NeRFDataset.generate_rays (dataset.py:109-145),
    LLFF.generate_rays (dataset.py:364-387) and convert_to_ndc (intern/ray.py:59-79) on synthetic poses.
    dataset.py imports cv2 (absent here, only used for Blender down-scaling): stubbed with an empty module."""
    import types
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    import dataset as ref_dataset
    g = np.random.Generator(np.random.PCG64(1010))
    out = {}
    n_cams, h, w = 3, 13, 17

    def random_pose():
        a = g.normal(size=(3, 3))
        q, _ = np.linalg.qr(a)
        return np.concatenate([q, g.normal(size=(3, 1)) * 2.0], axis=1)

    c2w = np.stack([random_pose() for _ in range(n_cams)], 0).astype(np.float32)
    # LLFF-style forward-facing poses: small rotations about the identity, cameras looking down -z
    c2w_ff = np.stack([np.concatenate([np.linalg.qr(np.eye(3) + 0.05 * g.normal(size=(3, 3)))[0] * np.array([1, 1, 1]),
                                       g.normal(size=(3, 1)) * 0.2], axis=1) for _ in range(n_cams)], 0).astype(np.float32)
    for i in range(n_cams):  # make sure z column points so that directions have dz < 0 (valid NDC projection)
        if c2w_ff[i, 2, 2] < 0:
            c2w_ff[i, :, 2] *= -1
    out["c2w"], out["c2w_ff"] = c2w, c2w_ff
    out["cfg"] = np.array([n_cams, h, w])
    for name, cls, poses, focal, near, far in (("pinhole", ref_dataset.NeRFDataset, c2w, 555.5, 2.0, 6.0),
                                               ("llff", ref_dataset.LLFF, c2w_ff, 38.25, 0.0, 1.0)):
        obj = object.__new__(cls)
        obj.h, obj.w, obj.focal, obj.cam_to_world, obj.near, obj.far = h, w, focal, poses, near, far
        obj.generate_rays()
        out[name + "_focal_near_far"] = np.array([focal, near, far], dtype=np.float64)
        for k in synthetic.RAY_FIELDS:
            out[f"{name}_{k}"] = np.asarray(getattr(obj.rays, k), dtype=np.float32).reshape(-1, getattr(obj.rays, k).shape[-1])
    o, d = ref_ray.convert_to_ndc(out["pinhole_origins"].reshape(n_cams, h, w, 3) * 0 + c2w_ff[:, None, None, :, 3],
                                  out["llff_directions"].reshape(n_cams, h, w, 3) * 0 + 1.0 * np.array([0.1, -0.2, -1.0], dtype=np.float32),
                                  38.25, w, h, near=1.0)
    out["ndc_in_o"] = (out["pinhole_origins"].reshape(n_cams, h, w, 3) * 0 + c2w_ff[:, None, None, :, 3]).astype(np.float32)
    out["ndc_in_d"] = (out["llff_directions"].reshape(n_cams, h, w, 3) * 0 + np.array([0.1, -0.2, -1.0], dtype=np.float32)).astype(np.float32)
    out["ndc_out_o"], out["ndc_out_d"] = o.astype(np.float32), d.astype(np.float32)
    save("g10_ray_generation", **out)


def g11():
    """Visualisation: visualize_depth / visualize_normals / depth_to_normals / sinebow of
    intern/pose.py:112-212 on a synthetic depth + accumulation map.  The reference calls `cm.get_cmap('turbo')`,
    removed in matplotlib >= 3.9 (3.10.8 here): patched to the equivalent `matplotlib.colormaps['turbo']`."""
    import matplotlib
    import matplotlib.cm as cm
    if not hasattr(cm, "get_cmap"):
        cm.get_cmap = lambda name: matplotlib.colormaps[name]
    from intern import pose as ref_pose
    g = np.random.Generator(np.random.PCG64(1111))
    h, w = 23, 31
    yy, xx = np.meshgrid(np.linspace(-1, 1, h), np.linspace(-1, 1, w), indexing="ij")
    depth = (3.0 + np.sin(3 * xx) * np.cos(2 * yy) + 0.05 * g.normal(size=(h, w))).astype(np.float32)
    acc = np.clip(g.uniform(0.2, 1.1, size=(h, w)), 0, 1).astype(np.float32)
    out = dict(depth=depth, acc=acc)
    out["normals"] = ref_pose.depth_to_normals(depth)
    out["vis_normals"] = ref_pose.visualize_normals(depth, acc)
    out["vis_normals_noacc"] = ref_pose.visualize_normals(depth, None)
    out["vis_depth_given"] = ref_pose.visualize_depth(depth, acc, 2.0, 6.0)          # blender-style near/far
    out["vis_depth_auto"] = ref_pose.visualize_depth(depth, acc, 0.0, 1.0)           # near = 0 is falsy -> automatic near
    out["vis_depth_auto2"] = ref_pose.visualize_depth(depth, None, None, None)
    out["vis_depth_mod"] = ref_pose.visualize_depth(depth, acc, 2.0, 6.0, modulus=0.25)
    out["sinebow"] = ref_pose.sinebow(np.linspace(-0.5, 1.5, 41))
    depth_nan = depth.copy()
    depth_nan[3, 4] = np.nan
    depth_nan[h - 1, 0] = np.nan
    out["depth_nan"] = depth_nan
    out["vis_depth_nan"] = ref_pose.visualize_depth(depth_nan, acc, 2.0, 6.0)
    out["vis_normals_nan"] = ref_pose.visualize_normals(depth_nan, acc)
    save("g11_visualisation", **{k: np.asarray(v) for k, v in out.items()})


def g12():
    """Losses: Loss_prop / Loss_nerf / Loss_dist of intern/loss.py with the autograd
    gradients w.r.t. their direct inputs, plus the bounds()/loss_prop() pair of intern/distillation.py.
    Inputs: sorted interval edges (t in [2, 6], s in [0, 1]) and positive weights that sum to < 1 per ray."""
    from intern import distillation as ref_dis
    from intern import loss as ref_loss
    g = np.random.Generator(np.random.PCG64(1212))
    out = {}
    for tag, B, n in (("a", 5, 16), ("b", 9, 128), ("c", 1, 1), ("d", 3, 70)):
        t = np.sort(g.uniform(2.0, 6.0, size=(B, n + 1)), -1).astype(np.float32)
        t_hat = np.sort(g.uniform(2.0, 6.0, size=(B, n + 1)), -1).astype(np.float32)
        s = np.sort(g.uniform(0.0, 1.0, size=(B, n + 1)), -1).astype(np.float32)
        w = g.dirichlet(np.full(n + 1, 0.4), size=B)[:, :n].astype(np.float32)
        w_hat = g.dirichlet(np.full(n + 1, 0.6), size=B)[:, :n].astype(np.float32)
        rgb = g.uniform(0, 1, size=(B, 3)).astype(np.float32)
        pix = g.uniform(0, 1, size=(B, 4)).astype(np.float32)  # the 4th column must be ignored ([..., :3])
        if tag == "a":
            t_hat[0] = t[0]            # identical partitions: the <,> comparisons see touching edges
            w_hat[1, 3] = 0.0          # division by eps alone
        out.update({f"{tag}.t": t, f"{tag}.t_hat": t_hat, f"{tag}.s": s, f"{tag}.w": w, f"{tag}.w_hat": w_hat,
                    f"{tag}.rgb": rgb, f"{tag}.pix": pix})
        wh = T(w_hat).requires_grad_(True)
        lp = ref_loss.Loss_prop(t=T(t), w=T(w), t_hat=T(t_hat), w_hat=wh)
        lp.backward()
        out[f"{tag}.loss_prop"], out[f"{tag}.loss_prop.grad_w_hat"] = N(lp), N(wh.grad)
        bnd = ref_dis.bounds(t_vals_fine=T(t), fine_weights=T(w), t_vals_coarse=T(t_hat))
        out[f"{tag}.bounds"] = N(bnd)
        out[f"{tag}.loss_prop_split"] = N(ref_dis.loss_prop(coarse_weights=T(w_hat), bounds=bnd))
        sv, wv = T(s).requires_grad_(True), T(w).requires_grad_(True)
        ld = ref_loss.Loss_dist(s_vals=sv, weights=wv)
        ld.backward()
        out[f"{tag}.loss_dist"], out[f"{tag}.loss_dist.grad_s"], out[f"{tag}.loss_dist.grad_w"] = N(ld), N(sv.grad), N(wv.grad)
        rv = T(rgb).requires_grad_(True)
        ln, psnr = ref_loss.Loss_nerf(input=rv, target=T(pix))
        ln.backward()
        out[f"{tag}.loss_nerf"], out[f"{tag}.psnr"], out[f"{tag}.loss_nerf.grad"] = N(ln), N(psnr), N(rv.grad)
    out["mse_to_psnr.in"] = np.array([1e-4, 0.02, 0.5, 1.0], np.float32)
    out["mse_to_psnr.out"] = N(ref_loss.mse_to_psnr(T(out["mse_to_psnr.in"])))
    save("g12_losses", **out)


def g13():
    """Training step: the parameter gradients of the reference's own train.py loop body on a
    reduced-width model (hidden 32 / 64, the G7 weights).
      prop step (train.py:55-62): Loss_prop(t, w, t_hat, w_hat).backward()        -> gradients of prop_net.*
      nerf step (train.py:69-80): (Loss_nerf + 0.01 * Loss_dist).backward()       -> gradients of nerf_net.*
      extra: a loss on acc (sum(acc * b)), not in train.py, to pin that path.  (A loss on `distance` cannot be
             differentiated in the reference: t_to_s -> g() bumps t_vals in place after the clamp of
             intern/ray.py:187 saved them, and autograd raises "modified by an inplace operation".)
    g() mutates rays.near / rays.far in place, so every (prop forward, nerf forward) PAIR shares one fresh clone of the
    rays - the sequence of mipNeRF360.forward (model.py:247-252) and of G7; the fixture keeps the originals.  (train.py
    reuses one rays object for all three pairs of an iteration, so there near/far drift by a few 1e-6 more per pair;
    that drift is not part of the fixture.)"""
    from intern import loss as ref_loss
    out = {}
    hp_, hn_ = 32, 64
    sd = synthetic.make_state_dict(hp_, hn_, seed=7)
    for k, v in sd.items():
        out["sd." + k] = v
    gen = np.random.Generator(np.random.PCG64(1313))
    for kind, B, n, wb in (("lego", 12, 16, True), ("garden", 10, 24, False), ("garden70", 3, 70, False)):
        r = synthetic.make_rays(kind[:6], B, seed=130 + n)
        pixels = gen.uniform(0, 1, size=(B, 3)).astype(np.float32)
        ca, cb = gen.normal(size=B).astype(np.float32), gen.normal(size=B).astype(np.float32)
        m = build_ref_model(sd, n, hp_, hn_, wb)
        m.train()
        for k in synthetic.RAY_FIELDS:
            out[f"{kind}_rays_{k}"] = r[k]
        out[f"{kind}_cfg"], out[f"{kind}_pixels"], out[f"{kind}_ca"], out[f"{kind}_cb"] = np.array([B, n, int(wb)]), pixels, ca, cb

        # --- prop step
        m.zero_grad()
        rays = ref_rays(r)
        t_hat, w_hat = m.prop_net.forward(rays)
        _, _, _, t, w, _ = m.nerf_net.forward(rays, t_vals=t_hat, coarse_weights=w_hat)
        loss_prop = ref_loss.Loss_prop(t=t.detach(), w=w.detach(), t_hat=t_hat, w_hat=w_hat)
        loss_prop.backward()
        out[f"{kind}_loss_prop"] = N(loss_prop)
        for name, p in m.named_parameters():
            if name.startswith("prop_net"):
                out[f"{kind}_propstep.{name}"] = N(p.grad)
            else:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, name   # the NeRF net gets nothing here

        # --- nerf step
        m.zero_grad()
        rays = ref_rays(r)
        t_hat, w_hat = m.prop_net.forward(rays)
        rgb, _, _, _, fine_w, s_vals = m.nerf_net.forward(rays, t_vals=t_hat.detach(), coarse_weights=w_hat.detach())
        loss_nerf, psnr = ref_loss.Loss_nerf(input=rgb, target=T(pixels))
        loss_dist = ref_loss.Loss_dist(s_vals=s_vals, weights=fine_w)
        (loss_nerf + 0.01 * loss_dist).backward()
        out[f"{kind}_loss_nerf"], out[f"{kind}_loss_dist"] = N(loss_nerf), N(loss_dist)
        for name, p in m.named_parameters():
            if name.startswith("nerf_net"):
                out[f"{kind}_nerfstep.{name}"] = N(p.grad)

        # --- acc path
        m.zero_grad()
        rays = ref_rays(r)
        t_hat, w_hat = m.prop_net.forward(rays)
        _, _, acc, _, _, _ = m.nerf_net.forward(rays, t_vals=t_hat.detach(), coarse_weights=w_hat.detach())
        (acc * T(cb)).sum().backward()
        for name, p in m.named_parameters():
            if name.startswith("nerf_net"):
                out[f"{kind}_acc.{name}"] = N(p.grad) if p.grad is not None else np.zeros(tuple(p.shape), np.float32)  # colour head: no path
    save("g13_train_gradients", **out)


def g14():
    """Reference-mutation fidelity (train.py:51-71): the reference's g() adds 1e-6 IN PLACE, so
    within one training iteration - three (prop forward, nerf forward) pairs on ONE rays object - rays.near gains
    3e-6 and rays.far 2e-6 per pair.  For nerf_360 (near = 0) the 2nd and 3rd pair therefore sample from
    near ~ 3e-6 / 6e-6 instead of 1e-6.  This fixture runs the loop body of train.py:53-80 (without optimizer steps:
    weights stay fixed) on ONE rays object and stores what every pair produced, the gradients of the steps and the
    final near / far.  Mirrors reproduce it with mutate_like_reference=True."""
    from intern import loss as ref_loss
    out = {}
    hp_, hn_ = 32, 64
    sd = synthetic.make_state_dict(hp_, hn_, seed=7)
    for k, v in sd.items():
        out["sd." + k] = v
    gen = np.random.Generator(np.random.PCG64(1414))
    for kind, B, n, wb in (("garden", 10, 24, False), ("lego", 6, 16, True)):
        r = synthetic.make_rays(kind, B, seed=140 + n)
        pixels = gen.uniform(0, 1, size=(B, 3)).astype(np.float32)
        m = build_ref_model(sd, n, hp_, hn_, wb)
        m.train()
        for k in synthetic.RAY_FIELDS:
            out[f"{kind}_rays_{k}"] = r[k]
        out[f"{kind}_cfg"], out[f"{kind}_pixels"] = np.array([B, n, int(wb)]), pixels
        rays = ref_rays(r)  # ONE object for the three pairs, as in train.py:52
        for pair in range(2):  # train.py:53-65
            m.zero_grad()
            t_hat, w_hat = m.prop_net.forward(rays)
            _, _, _, t, w, s = m.nerf_net.forward(rays, t_vals=t_hat, coarse_weights=w_hat)
            loss_prop = ref_loss.Loss_prop(t=t.detach(), w=w.detach(), t_hat=t_hat, w_hat=w_hat)
            loss_prop.backward()
            out[f"{kind}_p{pair}_t_hat"], out[f"{kind}_p{pair}_w_hat"] = N(t_hat), N(w_hat)
            out[f"{kind}_p{pair}_t"], out[f"{kind}_p{pair}_w"], out[f"{kind}_p{pair}_s"] = N(t), N(w), N(s)
            out[f"{kind}_p{pair}_loss_prop"] = N(loss_prop)
            out[f"{kind}_p{pair}_near"], out[f"{kind}_p{pair}_far"] = N(rays.near), N(rays.far)
            for name, p in m.named_parameters():
                if name.startswith("prop_net"):
                    out[f"{kind}_p{pair}_grad.{name}"] = N(p.grad)
        m.zero_grad()  # train.py:68-80
        t_hat, w_hat = m.prop_net.forward(rays)
        rgb, dist, acc, t, fine_w, s_vals = m.nerf_net.forward(rays, t_vals=t_hat.detach(), coarse_weights=w_hat.detach())
        loss_nerf, psnr = ref_loss.Loss_nerf(input=rgb, target=T(pixels))
        loss_dist = ref_loss.Loss_dist(s_vals=s_vals, weights=fine_w)
        (loss_nerf + 0.01 * loss_dist).backward()
        out[f"{kind}_p2_t_hat"], out[f"{kind}_p2_w_hat"] = N(t_hat), N(w_hat)
        out[f"{kind}_p2_t"], out[f"{kind}_p2_w"], out[f"{kind}_p2_s"] = N(t), N(fine_w), N(s_vals)
        out[f"{kind}_p2_rgb"], out[f"{kind}_p2_dist"], out[f"{kind}_p2_acc"] = N(rgb), N(dist), N(acc)
        out[f"{kind}_p2_loss_nerf"], out[f"{kind}_p2_loss_dist"] = N(loss_nerf), N(loss_dist)
        out[f"{kind}_p2_near"], out[f"{kind}_p2_far"] = N(rays.near), N(rays.far)
        for name, p in m.named_parameters():
            if name.startswith("nerf_net"):
                out[f"{kind}_p2_grad.{name}"] = N(p.grad)
    save("g14_mutation", **out)


def g15():
    """Row f4: a checkpoint written by the REFERENCE's own class exactly as train.py:98-103 writes it
    (`torch.save(model.state_dict(), path)`; tiny widths, the reference's own kaiming init under a fixed torch seed),
    plus what that reference model renders on a small seeded batch - `load_reference_checkpoint` must load the file
    and reproduce the render."""
    torch.manual_seed(1515)
    n = 16
    m = ref_model.mipNeRF360(randomized=False, num_samples=n, hidden_proposal=16, hidden_nerf=24, white_bkgd=True, device=CPU)
    path = os.path.join(HERE, "g15_reference_checkpoint.pt")
    torch.save(m.state_dict(), path)
    print(f"  wrote g15_reference_checkpoint.pt  ({os.path.getsize(path) / 1024:.1f} KiB)")
    r = synthetic.make_rays("lego", 12, seed=150)
    m.eval()
    with torch.no_grad():
        rgb, dist, acc = m(ref_rays(r))
    out = {f"rays_{k}": r[k] for k in synthetic.RAY_FIELDS}
    out.update(rgb=N(rgb), dist=N(dist), acc=N(acc), cfg=np.array([n, 16, 24, 1]))
    save("g15_reference_render", **out)


def g16():
    """The reference's two public branches its own hot path never takes: gaussian_to_xyz(diag=True)
    (intern/parameterization.py:48-54) and conical_frustum_to_gaussian(stable=False) (:108-113)."""
    g = np.random.Generator(np.random.PCG64(1616))
    B, n = 5, 7
    d = g.normal(size=(B, 3)).astype(np.float32)
    t0 = (2.0 + np.sort(g.uniform(0, 4, size=(B, n + 1)), axis=1)).astype(np.float32)
    radii = g.uniform(1e-3, 5e-2, size=(B, 1)).astype(np.float32)
    tm, tv, rv = (g.uniform(0.5, 3, size=(B, n)).astype(np.float32), g.uniform(1e-4, 1e-2, size=(B, n)).astype(np.float32),
                  g.uniform(1e-5, 1e-3, size=(B, n)).astype(np.float32))
    mean, cov = ref_par.gaussian_to_xyz(T(d), T(tm), T(tv), T(rv), diag=True)
    m2, c2 = ref_par.conical_frustum_to_gaussian(T(d), T(t0[:, :-1]), T(t0[:, 1:]), T(radii), diag=False, stable=False)
    save("g16_dead_branches", d=d, t=t0, radii=radii, tm=tm, tv=tv, rv=rv, diag_mean=N(mean), diag_cov=N(cov),
         unstable_mean=N(m2), unstable_cov=N(c2))


# build-owned callables for G17 (the tests pass the same two to the mirrors)
def g17_curve(x):
    return 1.0 / (x + np.finfo(np.float32).eps)


def g17_colormap(v):
    return np.stack([v, 1.0 - v, v * v, np.ones_like(v)], -1)


def g17():
    """visualize_depth's remaining options (intern/pose.py:148-212): ignore_frac > 0 (weighted-quantile planes), a custom
    curve_fn and a custom colormap callable."""
    import matplotlib
    import matplotlib.cm as cm
    if not hasattr(cm, "get_cmap"):
        cm.get_cmap = lambda name: matplotlib.colormaps[name]
    from intern import pose as ref_pose
    g = np.random.Generator(np.random.PCG64(1717))
    h, w = 37, 41
    yy, xx = np.meshgrid(np.linspace(-1, 1, h), np.linspace(-1, 1, w), indexing="ij")
    depth = (3.0 + np.sin(3 * xx) * np.cos(2 * yy) + 0.3 * g.normal(size=(h, w))).astype(np.float32)
    acc = np.clip(g.uniform(0.1, 1.2, size=(h, w)), 0, 1).astype(np.float32)
    depth_nan = depth.copy()
    depth_nan[5, 6] = np.nan
    out = dict(depth=depth, acc=acc, depth_nan=depth_nan)
    out["ignore"] = ref_pose.visualize_depth(depth, acc, ignore_frac=0.1)
    out["ignore_nan"] = ref_pose.visualize_depth(depth_nan, acc, ignore_frac=0.05)
    out["ignore_noacc_far"] = ref_pose.visualize_depth(depth, None, None, 5.0, ignore_frac=0.2)
    out["curve"] = ref_pose.visualize_depth(depth, acc, 2.0, 6.0, curve_fn=g17_curve)
    out["curve_auto"] = ref_pose.visualize_depth(depth, acc, curve_fn=g17_curve, ignore_frac=0.1)
    out["colormap"] = ref_pose.visualize_depth(depth, acc, 2.0, 6.0, colormap=g17_colormap)
    out["colormap_mod"] = ref_pose.visualize_depth(depth, acc, 2.0, 6.0, modulus=0.3, colormap=g17_colormap)
    out["both"] = ref_pose.visualize_depth(depth, acc, curve_fn=g17_curve, colormap=g17_colormap, ignore_frac=0.05)
    save("g17_visualize_depth_options", **{k: np.asarray(v) for k, v in out.items()})


def g18():
    """NaN / Inf / degenerate inputs on the hot path, as the REFERENCE treats them: where it produces NaN, where
    nan_to_num / clamp swallow it, where a degenerate value stays finite.
      vd        ViewdirectionEncoding with |z| = 1 + 2^-23 (acos -> NaN), NaN / Inf / zero components, x + 1e-6 == 0
                (intern/encoding.py:69-90)
      sample    sample_along_rays with far == near, near = far = 0, far < near, directions = 0, radii = 0
                (intern/ray.py:81-116, intern/parameterization.py:46 clamp 1e-10)
      lift      gaussian_to_xyz with d = 0 rows (parameterization.py:31-62)
      resample  sorted_piecewise_constant_pdf / resample_along_rays with a NaN weight, an all-NaN row, +Inf, a negative
                weight, weights whose sum overflows (intern/ray.py:12-57,118-153)
      render    density_to_weight / volumetric_rendering with NaN, +Inf, 1e38 and -1e4 densities (exp overflow), a NaN
                colour, a zero direction (model.py:59-78, intern/ray.py:155-191)
      e2e       both stage forwards of a small model on a batch holding such rays (NaN stays local to its ray), and on
                a batch with one NaN origin (local: origins are added after the contraction) and one NaN direction (the
                whole-chunk contraction norm turns every ray of the chunk NaN)"""
    out = {}
    nan, inf = np.float32(np.nan), np.float32(np.inf)
    one_up = np.float32(1.0) + np.float32(2.0 ** -23)
    # ---- vd
    v = np.array([[0.0, 0.0, one_up], [0.0, 0.0, -one_up], [0.6, 0.0, 0.8], [nan, 0.5, 0.5], [inf, 0.0, 0.5],
                  [0.0, 0.0, 0.0], [-1e-6, 0.0, 0.5], [-1e-6, 0.3, 0.5], [0.3, nan, 0.5], [0.3, 0.4, inf]], dtype=np.float32)
    out["vd_in"] = v
    for lo, hi in ((0, 4), (1, 3)):
        out[f"vd_{lo}_{hi}"] = N(ref_enc.ViewdirectionEncoding(lo, hi)(T(v)))
    # ---- sample
    r = synthetic.make_rays("lego", 7, seed=181)
    r["near"][:, 0] = [1.0, 0.0, 3.0, 2.0, 2.0, 2.0, 0.0]
    r["far"][:, 0] = [1.0, 0.0, 2.0, 6.0, 6.0, 6.0, 1.0]
    r["directions"][4] = 0.0
    r["radii"][5] = 0.0
    n = 4
    t_vals, (means, covs) = ref_ray.sample_along_rays(T(r["origins"]), T(r["directions"]), T(r["radii"]), n, T(r["near"]),
                                                      T(r["far"]), False)
    for k in synthetic.RAY_FIELDS:
        out["sample_rays_" + k] = r[k]
    out["sample_t"], out["sample_means"], out["sample_covs"] = N(t_vals), N(means), N(covs)
    # ---- lift
    g = np.random.Generator(np.random.PCG64(1818))
    d = g.normal(size=(4, 3)).astype(np.float32)
    d[1] = 0.0
    d[2] = [1e-6, 0.0, 0.0]          # |d|^2 = 1e-12 < the 1e-10 clamp
    tm, tv, rv = (g.uniform(0.5, 3, size=(4, 5)).astype(np.float32), g.uniform(1e-4, 1e-2, size=(4, 5)).astype(np.float32),
                  g.uniform(1e-5, 1e-3, size=(4, 5)).astype(np.float32))
    m_, c_ = ref_par.gaussian_to_xyz(T(d), T(tm), T(tv), T(rv), diag=False)
    out.update(lift_d=d, lift_tm=tm, lift_tv=tv, lift_rv=rv, lift_mean=N(m_), lift_cov=N(c_))
    # ---- resample
    B, n = 8, 8
    t = np.sort(g.uniform(2, 6, size=(B, n + 1)), axis=1).astype(np.float32)
    w = g.uniform(0.05, 1, size=(B, n)).astype(np.float32)
    w[0, 3] = nan
    w[1, :] = nan
    w[2, 5] = inf
    w[3, 2] = -1.0
    w[4, :] = 3e38                   # the sum overflows to +Inf
    w[5, 0] = nan                    # NaN in the first / last interval (the replicate padding of the blur)
    w[6, n - 1] = nan
    out["resample_t"], out["resample_w"] = t, w
    out["pdf_samples"] = N(ref_ray.sorted_piecewise_constant_pdf(T(t), T(w), n + 1, randomized=False))
    out["pdf_samples_5"] = N(ref_ray.sorted_piecewise_constant_pdf(T(t), T(w), 5, randomized=False))
    rr = synthetic.make_rays("lego", B, seed=182)
    new_t, (means, covs) = ref_ray.resample_along_rays(T(rr["origins"]), T(rr["directions"]), T(rr["radii"]), T(t), T(w),
                                                       False, 0.01)
    for k in synthetic.RAY_FIELDS:
        out["resample_rays_" + k] = rr[k]
    out["resample_new_t"], out["resample_means"], out["resample_covs"] = N(new_t), N(means), N(covs)
    # ---- render
    B, n = 8, 8
    t = np.sort(g.uniform(2, 6, size=(B, n + 1)), axis=1).astype(np.float32)
    dens = g.gamma(0.6, 4.0, size=(B, n, 1)).astype(np.float32)
    dens[0, 3] = nan
    dens[1, 2] = inf
    dens[2, :] = 1e38                # density * delta overflows
    dens[3, 4] = -1e4                # exp(+big) = Inf -> alpha = -Inf
    dens[4, 0] = nan                 # first sample: every transmittance after it is NaN
    dens[5, n - 1] = nan             # last sample: only its own weight
    rgb = g.uniform(0, 1, size=(B, n, 3)).astype(np.float32)
    rgb[6, 2, 1] = nan
    dirs = g.normal(size=(B, 3)).astype(np.float32)
    dirs[7] = 0.0
    dens[7, 1] = inf                 # Inf * 0 = NaN
    out.update(render_t=t, render_density=dens, render_rgb=rgb, render_dirs=dirs)
    pn = ref_model.prop_net(num_samples=n, hidden_proposal=8, device=CPU)
    out["render_w_prop"] = N(pn.density_to_weight(T(t), T(dens), T(dirs)))
    for wb in (False, True):
        c, dd, a, ww = ref_ray.volumetric_rendering(T(rgb), T(dens), T(t), T(dirs), wb)
        tag = f"render_wb{int(wb)}"
        out[tag + "_rgb"], out[tag + "_dist"], out[tag + "_acc"], out[tag + "_w"] = N(c), N(dd), N(a), N(ww)
    # ---- e2e
    hp_, hn_, n = 32, 64, 8
    sd = synthetic.make_state_dict(hp_, hn_, seed=18)
    for k, v_ in sd.items():
        out["sd." + k] = v_
    out["e2e_cfg"] = np.array([n, hp_, hn_])
    for tag in ("local", "nan_origin", "nan_direction"):
        r = synthetic.make_rays("garden", 8, seed=183)
        r["viewdirs"][0] = [0.0, 0.0, one_up]
        r["far"][1] = r["near"][1]
        r["directions"][2] = 0.0
        r["radii"][3] = 0.0
        r["viewdirs"][4] = [0.0, 0.0, -one_up]
        if tag == "nan_origin":      # origins are added AFTER the contraction (parameterization.py:135): stays local
            r["origins"][5, 1] = nan
        if tag == "nan_direction":   # enters the whole-chunk norm of contract(): every ray of the chunk turns NaN
            r["directions"][5, 1] = nan
        m = build_ref_model(sd, n, hp_, hn_, False)
        with torch.no_grad():
            rays = ref_rays(r)
            t_hat, w_hat = m.prop_net.forward(rays)
            t_hat_np, w_hat_np = N(t_hat), N(w_hat)
            o = m.nerf_net.forward(rays, t_vals=t_hat, coarse_weights=w_hat)
        for k in synthetic.RAY_FIELDS:
            out[f"e2e_{tag}_rays_{k}"] = r[k]
        out[f"e2e_{tag}_t_hat"], out[f"e2e_{tag}_w_hat"] = t_hat_np, w_hat_np
        for nm, v_ in zip(("rgb", "dist", "acc", "t_vals", "fine_w", "s_vals"), o):
            out[f"e2e_{tag}_{nm}"] = N(v_)
        print(f"  G18 e2e {tag}: NaN rays in rgb = {np.isnan(N(o[0])).any(1).nonzero()[0].tolist()}")
    save("g18_degenerate", **out)


def g19():
    """Whole-path parity OUTSIDE the flat regime of Kaiming-initialised weights: build-owned
    "trained-like" weights (`synthetic.make_structured_state_dict`: scaled layers, density shells routed through the
    proposal net, colour / density units routed through the NeRF net) on three ray kinds - lego, garden and `mixed`
    (per-ray near / far: short rays render nearly empty, long rays saturate; the NeRF density is bounded by the
    reference's sigmoid head, model.py:150-158,185, so opacity is a function of ray length).
      small.*   reduced width (32 / 64, weights stored), all six stage outputs + (t_hat, w_hat), G7-style
      full.*    full width (256 / 1024, weights regenerated from the seed; per-tensor checksums stored), B = 256
      frame.*   a 32 x 24 render_image with chunks in {128, 4096}, reduced width
    The regime is asserted here (and again on the stored data by tests/test_oracle_golden.py): rgb std over rays >= 0.2,
    proposal weights with max / mean >= 20 on most rays of the pinhole kinds, saturated and nearly empty rays in `mixed`."""
    import contextlib
    import io
    out = {}
    seed = 19

    def run(tag, kind, B, n, wb, hp_, hn_, store_sd):
        r = synthetic.make_rays(kind, B, seed=seed)
        sd = synthetic.make_structured_state_dict(hp_, hn_, seed, r, n)
        m = build_ref_model(sd, n, hp_, hn_, wb)
        t0 = time.time()
        with torch.no_grad():
            rays = ref_rays(r)
            t_hat, w_hat = m.prop_net.forward(rays)
            t_hat_np, w_hat_np = N(t_hat), N(w_hat)
            o = m.nerf_net.forward(rays, t_vals=t_hat, coarse_weights=w_hat)
        # the same model and rays in double precision (the reference's fp32-ness is torch's default dtype plus the dtype of
        # what it is handed): in this regime the path is ill-conditioned - the shells turn a rounding error of the sample
        # position into a weight error 1e3 times larger - so the tests take their tolerances from the reference's OWN fp32
        # error against these values instead of asserting one (stored rounded to fp32: 6e-8 relative, far below the errors)
        with reference_in_fp64():
            m64 = build_ref_model(sd, n, hp_, hn_, wb)
            m64.load_state_dict({k: T64(v) for k, v in sd.items()})
            m64 = m64.double()
            with torch.no_grad():
                rays64 = ref_ray.Rays(*[T64(r[k]) for k in synthetic.RAY_FIELDS])
                t_hat64, w_hat64 = m64.prop_net.forward(rays64)
                w_hat64_np = N(w_hat64)
                o64 = m64.nerf_net.forward(rays64, t_vals=t_hat64, coarse_weights=w_hat64)
            assert o64[0].dtype == torch.float64 and w_hat64.dtype == torch.float64
        out[tag + "_w_hat64"] = w_hat64_np.astype(np.float32)
        for nm, v in zip(("rgb", "dist", "acc", "t_vals", "fine_w", "s_vals"), o64):
            out[f"{tag}_{nm}64"] = N(v).astype(np.float32)
        if store_sd:
            for k, v in sd.items():
                out[f"{tag}_sd." + k] = v
        else:
            out[tag + "_sdsum"] = synthetic.state_dict_checksum(sd)
        for k in synthetic.RAY_FIELDS:
            out[f"{tag}_rays_{k}"] = r[k]
        out[tag + "_cfg"] = np.array([B, n, int(wb), hp_, hn_, seed])
        out[tag + "_t_hat"], out[tag + "_w_hat"] = t_hat_np, w_hat_np
        for nm, v in zip(("rgb", "dist", "acc", "t_vals", "fine_w", "s_vals"), o):
            out[f"{tag}_{nm}"] = N(v)
        rgb, acc = N(o[0]), N(o[2])
        peak = w_hat_np.max(1) / np.maximum(w_hat_np.mean(1), 1e-30)
        print(f"  G19 {tag}: {time.time() - t0:.1f}s  rgb std over rays {rgb.std(0).round(3)}  acc [{acc.min():.3f}, {acc.max():.3f}]  "
              f"w_hat max/mean median {np.median(peak):.1f} (>= 20 on {np.mean(peak >= 20):.2f} of the rays)")
        return rgb, acc, peak

    for kind, B, n, wb in (("lego", 12, 16, True), ("garden", 10, 24, False), ("mixed", 16, 32, True)):
        run("small." + kind, kind, B, n, wb, 32, 64, True)
    for kind, B, n, wb in (("lego", 256, 64, True), ("garden", 256, 128, False), ("mixed", 256, 128, True)):
        rgb, acc, peak = run("full." + kind, kind, B, n, wb, 256, 1024, False)
        assert rgb.std(0).mean() >= 0.2, "not a high-contrast render"
        assert np.mean(peak >= 20) >= 0.6, "proposal weights are not peaked"
        if kind == "mixed":
            assert (acc > 0.99).mean() >= 0.1 and (acc < 0.15).mean() >= 0.1, "no saturated / empty rays"
    hp_, hn_, n = 32, 64, 16
    h, w = 24, 32
    r = synthetic.make_rays("mixed", h * w, seed=seed + 1)
    sd = synthetic.make_structured_state_dict(hp_, hn_, seed + 1, r, n)
    for k, v in sd.items():
        out["frame_sd." + k] = v
    for k in synthetic.RAY_FIELDS:
        out["frame_rays_" + k] = r[k]
    out["frame_cfg"] = np.array([h, w, n, hp_, hn_, seed + 1])
    m = build_ref_model(sd, n, hp_, hn_, True)
    for chunks in (128, 4096):
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            rgb8, dist, acc = m.render_image(ref_rays(r), h, w, chunks=chunks)
        out[f"frame_c{chunks}_rgb8"], out[f"frame_c{chunks}_dist"], out[f"frame_c{chunks}_acc"] = rgb8, dist, acc
        with reference_in_fp64(), contextlib.redirect_stdout(io.StringIO()):   # the same frame in double precision (see run())
            m64 = build_ref_model(sd, n, hp_, hn_, True)
            m64.load_state_dict({k: T64(v) for k, v in sd.items()})
            _, dist64, acc64 = m64.double().render_image(ref_ray.Rays(*[T64(r[k]) for k in synthetic.RAY_FIELDS]), h, w, chunks=chunks)
        assert dist64.dtype == np.float64
        out[f"frame_c{chunks}_dist64"], out[f"frame_c{chunks}_acc64"] = dist64.astype(np.float32), acc64.astype(np.float32)
        print(f"  G19 frame chunks={chunks}: {time.time() - t0:.1f}s  rgb8 std {rgb8.reshape(-1, 3).std(0).round(1)}")
    save("g19_structured_weights", **out)


def g20():
    """A checkpoint TRAINED by the build's own training loop, rendered by the REFERENCE (optional part; rows
    f3 + f4 + the whole path).  `tools/train_demo.py --steps 500 --rays 1024 --samples 32 --hidden 32 64 --lr 3e-3 --kind lego
    --teacher structured --white-bkgd --save ...` ran on an MI355X: the reference's loop body (train.py:53-82) on the HIP mirrors
    (tape-keeping forwards, hand-written backward, AdamW), a Kaiming-initialised student fitted to the pixels a structured
    (G19) teacher renders for 1024 lego rays; the student's state_dict was written with torch.save in the reference's checkpoint
    layout (train.py:98-103) and is committed as tests/golden/g20_trained_checkpoint.pt.  Here the reference's own class loads
    that file and renders the same 1024 rays as ONE chunk (the chunk size it was trained at), in fp32 and in fp64."""
    path = os.path.join(HERE, "g20_trained_checkpoint.pt")
    sd_t = torch.load(path, map_location="cpu")
    sd = {k: N(v) for k, v in sd_t.items()}
    meta = __import__("json").load(open(os.path.join(HERE, "g20_training_run.json")))
    B, n, hp_, hn_, wb = int(meta["rays"]), int(meta["samples"]), int(meta["hidden"][0]), int(meta["hidden"][1]), bool(meta["white_bkgd"])
    r = synthetic.make_rays(meta["kind"], B, seed=int(meta["rays_seed"]))
    out = {"cfg": np.array([B, n, int(wb), hp_, hn_, int(meta["rays_seed"])])}
    for k in synthetic.RAY_FIELDS:
        out["rays_" + k] = r[k]
    m = build_ref_model(sd, n, hp_, hn_, wb)
    t0 = time.time()
    with torch.no_grad():
        rays = ref_rays(r)
        t_hat, w_hat = m.prop_net.forward(rays)
        t_hat_np, w_hat_np = N(t_hat), N(w_hat)
        o = m.nerf_net.forward(rays, t_vals=t_hat, coarse_weights=w_hat)
    print(f"  G20 fp32: {time.time() - t0:.1f}s")
    t0 = time.time()
    with reference_in_fp64():
        m64 = build_ref_model(sd, n, hp_, hn_, wb)
        m64.load_state_dict({k: T64(v) for k, v in sd.items()})
        m64 = m64.double()
        with torch.no_grad():
            rays64 = ref_ray.Rays(*[T64(r[k]) for k in synthetic.RAY_FIELDS])
            t_hat64, w_hat64 = m64.prop_net.forward(rays64)
            w_hat64_np = N(w_hat64)
            o64 = m64.nerf_net.forward(rays64, t_vals=t_hat64, coarse_weights=w_hat64)
    print(f"  G20 fp64: {time.time() - t0:.1f}s")
    out["t_hat"], out["w_hat"], out["w_hat64"] = t_hat_np, w_hat_np, w_hat64_np.astype(np.float32)
    for nm, v, v64 in zip(("rgb", "dist", "acc", "t_vals", "fine_w", "s_vals"), o, o64):
        out[nm], out[nm + "64"] = N(v), N(v64).astype(np.float32)
    rgb, acc = out["rgb"], out["acc"]
    peak = w_hat_np.max(1) / np.maximum(w_hat_np.mean(1), 1e-30)
    print(f"  G20: rgb std over rays {rgb.std(0).round(3)}  acc [{acc.min():.3f}, {acc.max():.3f}]  w_hat max/mean median {np.median(peak):.1f}; "
          f"student on the GPU after training: rgb std {meta.get('student_rgb_std_over_rays')}, PSNR trajectory "
          f"{[t['psnr'] for t in meta['trajectory']][:3]} ... {[t['psnr'] for t in meta['trajectory']][-2:]}")
    save("g20_trained_checkpoint_render", **out)


def g21():
    """Training gradients OUTSIDE the flat regime (row f3 on fixture G19's weights): the parameter gradients of the reference's
    train.py loop body (train.py:55-62 proposal step, :69-80 NeRF step) on the reduced-width structured weights of G19, fp32 and
    fp64.  Through a density shell the proposal loss - relu(bound - w_hat)^2 / (w_hat + 1e-6) with w_hat over decades - is
    ill-conditioned, so the tests take their tolerance from the reference's own fp32 error (as for G19)."""
    from intern import loss as ref_loss
    out = {}
    hp_, hn_, seed = 32, 64, 19
    gen = np.random.Generator(np.random.PCG64(2121))
    for kind, B, n, wb in (("lego", 12, 16, True), ("mixed", 16, 32, True)):
        r = synthetic.make_rays(kind, B, seed=seed)
        sd = synthetic.make_structured_state_dict(hp_, hn_, seed, r, n)
        pixels = gen.uniform(0, 1, size=(B, 3)).astype(np.float32)
        for k, v in sd.items():
            out[f"{kind}_sd." + k] = v
        for k in synthetic.RAY_FIELDS:
            out[f"{kind}_rays_{k}"] = r[k]
        out[f"{kind}_cfg"], out[f"{kind}_pixels"] = np.array([B, n, int(wb), hp_, hn_]), pixels
        for dt, tag in ((torch.float32, ""), (torch.float64, "64")):
            ctx = reference_in_fp64() if dt == torch.float64 else __import__("contextlib").nullcontext()
            with ctx:
                m = build_ref_model(sd, n, hp_, hn_, wb)
                m.load_state_dict({k: torch.from_numpy(v).to(dt) for k, v in sd.items()})
                m = m.to(dt)
                m.train()
                mk = lambda: ref_ray.Rays(*[torch.from_numpy(r[k]).to(dt).clone() for k in synthetic.RAY_FIELDS])  # noqa: E731
                m.zero_grad()
                rays = mk()
                t_hat, w_hat = m.prop_net.forward(rays)
                _, _, _, t, w, _ = m.nerf_net.forward(rays, t_vals=t_hat, coarse_weights=w_hat)
                loss_prop = ref_loss.Loss_prop(t=t.detach(), w=w.detach(), t_hat=t_hat, w_hat=w_hat)
                loss_prop.backward()
                out[f"{kind}_loss_prop{tag}"] = N(loss_prop).astype(np.float64)
                for name, p in m.named_parameters():
                    if name.startswith("prop_net"):
                        out[f"{kind}_propstep{tag}.{name}"] = N(p.grad).astype(np.float32)
                m.zero_grad()
                rays = mk()
                t_hat, w_hat = m.prop_net.forward(rays)
                rgb, _, _, _, fine_w, s_vals = m.nerf_net.forward(rays, t_vals=t_hat.detach(), coarse_weights=w_hat.detach())
                loss_nerf, _ = ref_loss.Loss_nerf(input=rgb, target=torch.from_numpy(pixels).to(dt))
                loss_dist = ref_loss.Loss_dist(s_vals=s_vals, weights=fine_w)
                (loss_nerf + 0.01 * loss_dist).backward()
                out[f"{kind}_loss_nerf{tag}"], out[f"{kind}_loss_dist{tag}"] = N(loss_nerf).astype(np.float64), N(loss_dist).astype(np.float64)
                for name, p in m.named_parameters():
                    if name.startswith("nerf_net"):
                        out[f"{kind}_nerfstep{tag}.{name}"] = N(p.grad).astype(np.float32)
        worst = max(np.abs(out[f"{kind}_propstep.{nm}"] - out[f"{kind}_propstep64.{nm}"]).max() / max(np.abs(out[f"{kind}_propstep64.{nm}"]).max(), 1e-30)
                    for nm in sd if nm.startswith("prop_net"))
        worst_n = max(np.abs(out[f"{kind}_nerfstep.{nm}"] - out[f"{kind}_nerfstep64.{nm}"]).max() / max(np.abs(out[f"{kind}_nerfstep64.{nm}"]).max(), 1e-30)
                      for nm in sd if nm.startswith("nerf_net"))
        print(f"  G21 {kind}: loss_prop {float(out[kind + '_loss_prop']):.6g} loss_nerf {float(out[kind + '_loss_nerf']):.6g}; the reference's own fp32 error of the "
              f"gradients (relative to each tensor's max): proposal step {worst:.2e}, NeRF step {worst_n:.2e}")
    save("g21_structured_gradients", **out)


# ----------------------------------------------------------------------------- G22: the randomized branches, draws RECORDED
class record_draws:
    """Record - not replace - what the reference draws in its randomized branches: `torch.rand` (the stratified jitter,
    intern/ray.py:104) and `Tensor.uniform_(to=s - eps)` (the inverse CDF, intern/ray.py:33).  The reference runs unmodified on
    torch's own CPU generator; every tensor those two calls return is kept.  For `uniform_` the UNIT uniform behind the scaled value
    is recovered too: torch's CPU kernel computes x * (to - from) + from with x = (random & (2^24 - 1)) * 2^-24 from the same
    generator stream `torch.rand` reads, so replaying the generator state it started from through `torch.rand` gives x - checked
    bit for bit against the scaled tensor the reference received (x * float32(to) == scaled), then the generator is put back where the
    reference left it."""

    def __enter__(self):
        self.rand, self.uniform_scaled, self.uniform_unit, self.uniform_to = [], [], [], []
        self._rand, self._uniform = torch.rand, torch.Tensor.uniform_
        rec = self

        def rand(*a, **k):
            out = rec._rand(*a, **k)
            rec.rand.append(out.clone())
            return out

        def uniform_(t, *a, **k):
            assert not a and set(k) == {"to"}, (a, k)   # the reference's only form: uniform_(to=...)
            before = torch.get_rng_state()
            out = rec._uniform(t, **k)
            after = torch.get_rng_state()
            torch.set_rng_state(before)
            unit = rec._rand(t.shape, dtype=t.dtype)
            torch.set_rng_state(after)
            to = torch.tensor(k["to"], dtype=t.dtype)
            assert torch.equal(unit * to, out), "uniform_(to=) is not rand * to on this torch build"
            rec.uniform_scaled.append(out.clone())
            rec.uniform_unit.append(unit)
            rec.uniform_to.append(float(k["to"]))
            return out

        torch.rand, torch.Tensor.uniform_ = rand, uniform_
        return self

    def __exit__(self, *a):
        torch.rand, torch.Tensor.uniform_ = self._rand, self._uniform


def g22():
    """Fixture G22: intern/ray.py:103-108 (jitter) and :30-35 (randomized inverse CDF: `u + u`, the scaled
    uniform, the 1 - eps clamp) - the reference's CLI default (config.py:15) - run with randomized=True, draws recorded."""
    out = {}
    torch.manual_seed(2222)
    # (1) sample_along_rays: jittered t, and the Gaussians built from them
    for kind in ("lego", "garden"):
        for n in (8, 64, 128):
            r = synthetic.make_rays(kind, 5, seed=22)
            with record_draws() as rec:
                t_vals, (means, covs) = ref_ray.sample_along_rays(
                    T(r["origins"]), T(r["directions"]), T(r["radii"]), n, T(r["near"]), T(r["far"]), True)
            assert len(rec.rand) == 1 and not rec.uniform_scaled
            key = f"sample_{kind}_{n}"
            out[key + "_t_rand"], out[key + "_t"] = N(rec.rand[0]), N(t_vals)
            out[key + "_means"], out[key + "_covs"] = N(means), N(covs)
        for k in synthetic.RAY_FIELDS:
            out[f"sample_{kind}_rays_{k}"] = synthetic.make_rays(kind, 5, seed=22)[k]
    # (2) the bare sampler: uniform, peaked, all-zero (padding branch), near-empty (sum < 1e-5: padding branch with a remainder), and
    # an output count that differs from the bin count
    g = np.random.Generator(np.random.PCG64(2266))
    B, n = 7, 48
    t = np.sort(g.uniform(2, 6, size=(B, n + 1)), axis=1).astype(np.float32)
    w = g.uniform(0, 1, size=(B, n)).astype(np.float32)
    w[0] = 1.0 / n
    w[1] = 0.0
    w[1, 17] = 1.0
    w[2] = 0.0
    w[3] = np.exp(-0.5 * ((np.arange(n) - 30) / 2.0) ** 2)
    w[4] = 1e-8 * g.uniform(0, 1, size=n)            # sums to ~2.4e-7 < 1e-5: the pad is added to unequal weights
    out["pdf_t"], out["pdf_w"] = t, w
    for ns in (n + 1, 16, 128):
        with record_draws() as rec:
            smp = ref_ray.sorted_piecewise_constant_pdf(T(t), T(w), ns, randomized=True)
        assert len(rec.uniform_unit) == 1 and not rec.rand
        out[f"pdf_{ns}_u_unit"], out[f"pdf_{ns}_u_scaled"] = N(rec.uniform_unit[0]), N(rec.uniform_scaled[0])
        out[f"pdf_{ns}_to"] = np.array(rec.uniform_to[0], dtype=np.float64)
        out[f"pdf_{ns}_samples"] = N(smp)
    # the same rows the way the path hands them over (+ resample_padding, intern/ray.py:142): no flat stretch of the cdf near u = 1 - eps
    for ns in (n + 1, 16, 128):
        with record_draws() as rec:
            smp = ref_ray.sorted_piecewise_constant_pdf(T(t), T(w) + 0.01, ns, randomized=True)
        out[f"pdfpad_{ns}_u_unit"], out[f"pdfpad_{ns}_samples"] = N(rec.uniform_unit[0]), N(smp)
    # (3) resample_along_rays (blur + padding + randomized sampler + Gaussians), N in {8, 64, 128}
    for n2 in (8, 64, 128):
        r = synthetic.make_rays("lego" if n2 != 64 else "garden", 4, seed=220 + n2)
        lo, hi = float(r["near"][0, 0]), float(r["far"][0, 0])
        t2 = np.sort(g.uniform(lo + 1e-3, hi, size=(4, n2 + 1)), axis=1).astype(np.float32)
        w2 = (g.uniform(0, 1, size=(4, n2)) ** 4).astype(np.float32)
        w2[1] = 0.0                                   # all-zero ray: only the padding remains
        w2[2] = 0.0
        w2[2, n2 // 3] = 0.9                          # a single surface
        for pad in (0.01, 0.0):
            with record_draws() as rec:
                new_t, (means, covs) = ref_ray.resample_along_rays(T(r["origins"]), T(r["directions"]), T(r["radii"]), T(t2), T(w2), True, pad)
            key = f"resample_{n2}_pad{pad}"
            out[key + "_u_unit"], out[key + "_u_scaled"] = N(rec.uniform_unit[0]), N(rec.uniform_scaled[0])
            out[key + "_t"] = N(new_t)
            if pad == 0.01:
                out[key + "_means"], out[key + "_covs"] = N(means), N(covs)
        out[f"resample_{n2}_t_in"], out[f"resample_{n2}_w_in"] = t2, w2
        for k in synthetic.RAY_FIELDS:
            out[f"resample_{n2}_rays_{k}"] = r[k]
    # (4) both stage forwards of a randomized model (reduced width; model.py:80-94,163-200), and eval() leaving the sub-nets
    # randomized (model.py:281-283 switches only the outer flag)
    hp_, hn_ = 32, 64
    sd = synthetic.make_state_dict(hp_, hn_, seed=22)
    for k, v in sd.items():
        out["sd." + k] = v
    for kind, Bm, nm, wb in (("lego", 10, 16, True), ("garden", 8, 64, False), ("lego", 3, 128, False)):
        r = synthetic.make_rays(kind, Bm, seed=2200 + nm)
        m = ref_model.mipNeRF360(randomized=True, num_samples=nm, hidden_proposal=hp_, hidden_nerf=hn_, white_bkgd=wb, device=CPU)
        m.load_state_dict({k: T(v) for k, v in sd.items()})
        m.eval()
        assert m.prop_net.randomized and m.nerf_net.randomized   # eval() leaves the sub-nets' flags alone
        out[f"stage_{kind}_{nm}_outer_randomized_after_eval"] = np.array(int(bool(m.randomized)))  # nn.Module.eval() -> the overridden train(False)
        t0 = time.time()
        with torch.no_grad(), record_draws() as rec:
            rays = ref_rays(r)
            t_hat, w_hat = m.prop_net.forward(rays)
            t_hat_np, w_hat_np = N(t_hat), N(w_hat)
            o = m.nerf_net.forward(rays, t_vals=t_hat, coarse_weights=w_hat)
        assert len(rec.rand) == 1 and len(rec.uniform_unit) == 1
        tag = f"stage_{kind}_{nm}"
        for k in synthetic.RAY_FIELDS:
            out[f"{tag}_rays_{k}"] = r[k]
        out[tag + "_cfg"] = np.array([Bm, nm, int(wb)])
        out[tag + "_t_rand"], out[tag + "_u_unit"], out[tag + "_u_scaled"] = N(rec.rand[0]), N(rec.uniform_unit[0]), N(rec.uniform_scaled[0])
        out[tag + "_t_hat"], out[tag + "_w_hat"] = t_hat_np, w_hat_np
        for nm_, v in zip(("rgb", "dist", "acc", "t_vals", "fine_w", "s_vals"), o):
            out[f"{tag}_{nm_}"] = N(v)
        print(f"  G22 {tag}: {time.time() - t0:.1f}s")
    # (5) one iteration's two kinds of update with randomized=True - what `python train.py` runs by default (config.py:15): the proposal
    # step (train.py:55-62) and the NeRF step (train.py:69-80) on a randomized model in train() mode, every draw recorded
    from intern import loss as ref_loss
    gen = np.random.Generator(np.random.PCG64(2213))
    for kind, Bm, nm, wb in (("lego", 10, 16, True), ("garden", 6, 64, False)):
        r = synthetic.make_rays(kind, Bm, seed=2300 + nm)
        pixels = gen.uniform(0, 1, size=(Bm, 3)).astype(np.float32)
        m = ref_model.mipNeRF360(randomized=True, num_samples=nm, hidden_proposal=hp_, hidden_nerf=hn_, white_bkgd=wb, device=CPU)
        m.load_state_dict({k: T(v) for k, v in sd.items()})
        m.train()
        tag = f"train_{kind}_{nm}"
        for k in synthetic.RAY_FIELDS:
            out[f"{tag}_rays_{k}"] = r[k]
        out[tag + "_cfg"], out[tag + "_pixels"] = np.array([Bm, nm, int(wb)]), pixels
        m.zero_grad()
        with record_draws() as rec:
            rays = ref_rays(r)
            t_hat, w_hat = m.prop_net.forward(rays)
            _, _, _, t, w, _ = m.nerf_net.forward(rays, t_vals=t_hat, coarse_weights=w_hat)
        loss_prop = ref_loss.Loss_prop(t=t.detach(), w=w.detach(), t_hat=t_hat, w_hat=w_hat)
        loss_prop.backward()
        out[tag + "_prop_t_rand"], out[tag + "_prop_u_unit"], out[tag + "_loss_prop"] = N(rec.rand[0]), N(rec.uniform_unit[0]), N(loss_prop)
        for name, p in m.named_parameters():
            if name.startswith("prop_net"):
                out[f"{tag}_propstep.{name}"] = N(p.grad)
        m.zero_grad()
        with record_draws() as rec:
            rays = ref_rays(r)
            t_hat, w_hat = m.prop_net.forward(rays)
            rgb, _, _, _, fine_w, s_vals = m.nerf_net.forward(rays, t_vals=t_hat.detach(), coarse_weights=w_hat.detach())
        loss_nerf, _ = ref_loss.Loss_nerf(input=rgb, target=T(pixels))
        loss_dist = ref_loss.Loss_dist(s_vals=s_vals, weights=fine_w)
        (loss_nerf + 0.01 * loss_dist).backward()
        out[tag + "_nerf_t_rand"], out[tag + "_nerf_u_unit"] = N(rec.rand[0]), N(rec.uniform_unit[0])
        out[tag + "_loss_nerf"], out[tag + "_loss_dist"], out[tag + "_rgb"] = N(loss_nerf), N(loss_dist), N(rgb)
        for name, p in m.named_parameters():
            if name.startswith("nerf_net"):
                out[f"{tag}_nerfstep.{name}"] = N(p.grad)
        print(f"  G22 {tag}: loss_prop {float(loss_prop):.5g} loss_nerf {float(loss_nerf):.5g} loss_dist {float(loss_dist):.5g}")
    save("g22_randomized", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15", "g16", "g17", "g18", "g19", "g20", "g21", "g22"]
    table = dict(g1=g1_g2, g3=g3, g4=g4, g5=g5, g6=g6, g7=g7, g8=g8, g9=g9, g10=g10, g11=g11, g12=g12, g13=g13, g14=g14,
                 g15=g15, g16=g16, g17=g17, g18=g18, g19=g19, g20=g20, g21=g21, g22=g22)
    for k in which:
        print(k)
        table[k]()
