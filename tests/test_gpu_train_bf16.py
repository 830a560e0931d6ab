"""GPU tests (`-m gpu`) of TRAINING in the bf16 mode (round 5; SURVEY.md §8 row f3 in the precision BASELINE configs[4] runs at): the
reference's loop body (train.py:53-82) on the mirrors with mlp_dtype="bf16" - tape-keeping bf16 forwards, dz in bf16, the input gradient
through the forward layer kernel on a transposed bf16 packing, the weight gradient on v_mfma_f32_16x16x32_bf16 with both operands
transposed out of the LDS (ds_read_b64_tr_b16), fp32 accumulation, fp32 gradients and master weights.

Gradients are held against the REFERENCE's own gradients (fixtures G13 / G21: its autograd in fp32 and fp64).  bf16 carries 8
significant bits, so the bound is relative to each tensor's scale and stated where it is used; what it is derived from - the error of
the bf16 FORWARD on the same weights - is printed beside it.
"""
import numpy as np
import pytest
import torch
from conftest import g21_case

from mipnerf360_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test needs a HIP device")
    return torch.device("cuda:0")


def H(t):
    return t.detach().float().cpu().numpy()


def D(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).float().to(dev)


def dev_rays(d, dev):
    from mipnerf360_amd.intern.ray import Rays
    return Rays(*[D(d[k], dev) for k in synthetic.RAY_FIELDS])


@pytest.mark.parametrize("M,n,k", [(4096 + 37, 256, 256), (65536, 1024, 256), (32768 + 64, 256, 1024), (700, 64, 128), (64, 256, 256), (50, 256, 256),
                                   # the one-wave form (k >= 1024): fewer k-steps than its four-deep prologue, one split, ragged tails, many splits
                                   (64, 256, 1024), (100, 256, 1024), (96, 1024, 1024), (577, 256, 1024), (8192 + 31, 1024, 1024), (262144, 256, 1024)])
def test_linear_wgrad_bf16_against_fp64(dev, M, n, k):
    """dW = dZ^T X, db = column sums of dZ from bf16 rows: the MFMA kernel (pads of 256; ragged tail rows, a single stage, fewer rows than a
    stage) and the widening fall-back (other pads) against fp64 products of the SAME bf16 values - only the fp32 summation order differs;
    twice the same bits (row splits are reduced in a fixed order)."""
    from mipnerf360_amd import ops
    g = torch.Generator().manual_seed(M + n + k)
    dz = (torch.randn(M, n, generator=g) * torch.rand(M, 1, generator=g)).to(dev).bfloat16()
    x = torch.relu(torch.randn(M, k, generator=g)).to(dev).bfloat16()
    gw, gb = ops.linear_wgrad_bf16(dz, x)
    ref_w = dz.double().t() @ x.double()
    ref_b = dz.double().sum(0)
    scale_w, scale_b = float(ref_w.abs().max()), float(ref_b.abs().max())
    assert float((gw.double() - ref_w).abs().max()) <= 2e-5 * scale_w, (float((gw.double() - ref_w).abs().max()), scale_w)
    assert float((gb.double() - ref_b).abs().max()) <= 2e-5 * scale_b
    gw2, gb2 = ops.linear_wgrad_bf16(dz, x)
    assert torch.equal(gw, gw2) and torch.equal(gb, gb2)
    gw3, none = ops.linear_wgrad_bf16(dz, x, want_bias=False)
    assert none is None and torch.equal(gw3, gw)


@pytest.mark.parametrize("M,n,k,mask", [(512, 256, 256, True), (300, 64, 128, True), (256 * 3 + 5, 1024, 256, True), (131072, 1024, 1024, True),
                                        (512, 256, 256, False)])
def test_linear_dgrad_bf16_against_fp64(dev, M, n, k, mask):
    """dx = (dz W) * [relu_out > 0] in bf16: the forward layer kernels on the transposed bf16 packing, then the mask - against fp64 on the
    bf16-rounded operands, within one bf16 rounding of the result."""
    from mipnerf360_amd import ops
    g = torch.Generator().manual_seed(M * 3 + n + k)
    n_out, k_in = n - 3, k - 5
    w = (torch.randn(n_out, k_in, generator=g) / np.sqrt(n_out)).to(dev)
    dz = torch.randn(M, n, generator=g).to(dev)
    dz[:, n_out:] = 0
    dz = dz.bfloat16()
    relu_out = torch.relu(torch.randn(M, k, generator=g)).to(dev).bfloat16() if mask else None
    wt = ops.pack_linear_bf16_transposed(w, n_pad=n, k_pad=k)
    assert torch.equal(wt[:k_in, :n_out], w.t().bfloat16()) and float(wt[k_in:].float().abs().sum()) == 0 and float(wt[:, n_out:].float().abs().sum()) == 0
    dx = ops.linear_dgrad_bf16(dz, wt, relu_out)
    ref = dz.double() @ wt.double().t()
    if mask:
        ref = ref * (relu_out > 0)
    err = (dx.double() - ref).abs()
    assert float((err - (2.0 ** -8) * ref.abs()).max()) <= 1e-3 * float(ref.abs().max()), float(err.max())
    if mask:
        assert float(dx.float()[relu_out <= 0].abs().sum()) == 0.0


def _rel(got, want):
    want = np.asarray(want, dtype=np.float64)
    return float(np.abs(H(got).astype(np.float64) - want).max()) / max(float(np.abs(want).max()), 1e-30)


def _cos(pairs):
    a = np.concatenate([H(g).ravel().astype(np.float64) for g, _ in pairs])
    b = np.concatenate([np.asarray(w, dtype=np.float64).ravel() for _, w in pairs])
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


def _bf16_model(sd, dev, n, hp_, hn_, wb):
    from mipnerf360_amd.model import mipNeRF360
    m = mipNeRF360(randomized=False, num_samples=n, hidden_proposal=hp_, hidden_nerf=hn_, white_bkgd=wb, device=dev, mlp_dtype="bf16")
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.train()


# How the bounds are derived.  A gradient error is quoted per tensor as max |error| / max |value|.  Training in bf16 changes the gradient
# twice: (1) the bf16 FORWARD is another function than the fp32 one (activations rounded to 8 bits: ReLU units near zero switch, every
# layer's input carries 2^-9 relative noise) - inherent to the mode, measured here by the oracle's bf16-emulating forward differentiated
# by torch autograd in fp32 (O.*_step_gradients with Hyper(mlp_bf16=1): exact backward of the rounded forward) against the reference's
# gradients; it grows towards the first layers (G13 lego: 0.4 % at the last hidden layer, ~10 % at the first, on 32 / 64-wide Kaiming
# nets whose units are nearly constant over the batch); (2) the bf16 BACKWARD rounds dz to bf16 after every layer and reads bf16
# weights - what the HIP path adds on top: held to BACKWARD_REL against the oracle's bf16 gradients.  Against the reference itself the
# HIP gradients may then be off by the inherent part + that; their DIRECTION - what AdamW consumes - must agree to COS_MIN.
BACKWARD_REL, COS_MIN = 4e-2, 0.999


def _step_errors(hip, oracle16, ref, what):
    names = list(ref)
    inherent = max(_rel(torch.from_numpy(np.asarray(oracle16[n])), ref[n]) for n in names)
    added = max(_rel(hip[n], oracle16[n].numpy()) for n in names)
    total = max(_rel(hip[n], ref[n]) for n in names)
    cos = _cos([(hip[n], ref[n]) for n in names])
    print(f"{what}: bf16 forward alone (oracle, fp32 backward) {inherent:.2e} | HIP bf16 backward on top {added:.2e} | HIP vs reference {total:.2e}, cos {cos:.6f}")
    return inherent, added, total, cos


@pytest.mark.parametrize("kind", ["lego", "garden"])
def test_g13_train_step_gradients_bf16(golden, dev, kind):
    """Fixture G13 (every parameter gradient of the reference's loop body, reduced width 32 / 64 -> the widening weight-gradient path):
    the bf16 mirrors against the reference's fp32 autograd, with the share of the bf16 forward taken from the oracle."""
    from mipnerf360_amd.intern.loss import Loss_dist, Loss_nerf, Loss_prop
    from oracle import ref_path as O
    g = golden("g13_train_gradients")
    B, n, wb = (int(v) for v in g[f"{kind}_cfg"])
    sd = {k[3:]: g[k] for k in g if k.startswith("sd.")}
    model = _bf16_model(sd, dev, n, 32, 64, bool(wb))
    r = {f: g[f"{kind}_rays_{f}"] for f in synthetic.RAY_FIELDS}
    rays = dev_rays(r, dev)
    sdt, hp16 = O.to_torch_state_dict(sd), O.Hyper(num_samples=n, white_bkgd=bool(wb), mlp_bf16=1)
    t_hat, w_hat = model.prop_net.forward(rays)
    assert w_hat.requires_grad
    _, _, _, t, w, _ = model.nerf_net.forward(rays, t_vals=t_hat, coarse_weights=w_hat)
    loss_prop = Loss_prop(t=t.detach(), w=w.detach(), t_hat=t_hat, w_hat=w_hat)
    model.zero_grad()
    loss_prop.backward()
    hip = {name: p.grad.clone() for name, p in model.named_parameters() if name.startswith("prop_net")}
    _, og = O.prop_step_gradients(O.rays_from_numpy(r), sdt, hp16)
    inh_p, add_p, tot_p, cos_p = _step_errors(hip, og, {nm: g[f"{kind}_propstep.{nm}"] for nm in hip}, f"G13 bf16 {kind} proposal step")
    t_hat, w_hat = model.prop_net.forward(rays)
    rgb, _, _, _, fine_w, s_vals = model.nerf_net.forward(rays, t_vals=t_hat.detach(), coarse_weights=w_hat.detach())
    loss_nerf, _ = Loss_nerf(input=rgb, target=D(g[f"{kind}_pixels"], dev))
    loss_dist = Loss_dist(s_vals=s_vals, weights=fine_w)
    model.zero_grad()
    (loss_nerf + 0.01 * loss_dist).backward()
    hip = {name: p.grad.clone() for name, p in model.named_parameters() if name.startswith("nerf_net")}
    _, _, og = O.nerf_step_gradients(O.rays_from_numpy(r), sdt, hp16, torch.from_numpy(g[f"{kind}_pixels"]))
    inh_n, add_n, tot_n, cos_n = _step_errors(hip, og, {nm: g[f"{kind}_nerfstep.{nm}"] for nm in hip}, f"G13 bf16 {kind} NeRF step")
    for inh, add, tot, cos in ((inh_p, add_p, tot_p, cos_p), (inh_n, add_n, tot_n, cos_n)):
        assert add <= BACKWARD_REL and tot <= inh + BACKWARD_REL and cos >= COS_MIN, (inh, add, tot, cos)
    assert abs(float(loss_nerf.detach()) - float(g[f"{kind}_loss_nerf"])) <= 2e-2 * abs(float(g[f"{kind}_loss_nerf"]))


@pytest.mark.parametrize("kind", ["lego", "mixed"])
def test_g21_structured_gradients_bf16(golden, dev, kind):
    """Fixture G21 (trained-like weights: density shells, spread colours): the bf16 gradients against the reference's fp64 gradients,
    same decomposition.  (The reference's own fp32 run is up to 1 % of a tensor's scale from its fp64 run on the proposal step here.)"""
    from mipnerf360_amd.intern.loss import Loss_dist, Loss_nerf, Loss_prop
    from oracle import ref_path as O
    g = golden("g21_structured_gradients")
    (B, n, wb, hp_, hn_), r, sd, pixels = g21_case(g, kind)
    model = _bf16_model(sd, dev, n, hp_, hn_, wb)
    rays = dev_rays(r, dev)
    sdt, hp16 = O.to_torch_state_dict(sd), O.Hyper(num_samples=n, white_bkgd=bool(wb), mlp_bf16=1)
    t_hat, w_hat = model.prop_net.forward(rays)
    _, _, _, t, w, _ = model.nerf_net.forward(rays, t_vals=t_hat, coarse_weights=w_hat)
    loss_prop = Loss_prop(t=t.detach(), w=w.detach(), t_hat=t_hat, w_hat=w_hat)
    model.zero_grad()
    loss_prop.backward()
    hip = {name: p.grad.clone() for name, p in model.named_parameters() if name.startswith("prop_net")}
    _, og = O.prop_step_gradients(O.rays_from_numpy(r), sdt, hp16)
    inh_p, add_p, tot_p, cos_p = _step_errors(hip, og, {nm: g[f"{kind}_propstep64.{nm}"] for nm in hip}, f"G21 bf16 {kind} proposal step")
    model.zero_grad()
    t_hat, w_hat = model.prop_net.forward(rays)
    rgb, _, _, _, fine_w, s_vals = model.nerf_net.forward(rays, t_vals=t_hat.detach(), coarse_weights=w_hat.detach())
    loss_nerf, _ = Loss_nerf(rgb, D(pixels, dev))
    (loss_nerf + 0.01 * Loss_dist(s_vals, fine_w)).backward()
    hip = {name: p.grad.clone() for name, p in model.named_parameters() if name.startswith("nerf_net")}
    _, _, og = O.nerf_step_gradients(O.rays_from_numpy(r), sdt, hp16, torch.from_numpy(np.ascontiguousarray(pixels)).float())
    inh_n, add_n, tot_n, cos_n = _step_errors(hip, og, {nm: g[f"{kind}_nerfstep64.{nm}"] for nm in hip}, f"G21 bf16 {kind} NeRF step")
    # the proposal loss through a density shell is ill-conditioned: twice the bound there, direction to 0.99
    assert add_n <= BACKWARD_REL and tot_n <= inh_n + BACKWARD_REL and cos_n >= COS_MIN, (inh_n, add_n, tot_n, cos_n)
    assert add_p <= 2 * BACKWARD_REL and tot_p <= inh_p + 2 * BACKWARD_REL and cos_p >= 0.99, (inh_p, add_p, tot_p, cos_p)


@pytest.mark.parametrize("tag", ["lego_16", "garden_64"])
def test_g22_randomized_training_steps_bf16(golden, dev, tag):
    """Fixture G22's training steps - randomized=True, what `python train.py` runs by default (config.py:15) - in bf16, on the reference's own
    recorded draws (module.replay_uniforms), with the same decomposition as G13: what the bf16 forward costs is taken from the oracle's
    bf16-emulating forward on the same draws, what the HIP bf16 backward adds is held to BACKWARD_REL, the direction to COS_MIN."""
    from mipnerf360_amd.intern.loss import Loss_dist, Loss_nerf, Loss_prop
    from oracle import ref_path as O
    g = golden("g22_randomized")
    pre = f"train_{tag}_"
    B, n, wb = (int(v) for v in g[pre + "cfg"])
    sd = {k[3:]: g[k] for k in g if k.startswith("sd.")}
    r = {f: g[f"{pre}rays_{f}"] for f in synthetic.RAY_FIELDS}
    rays = dev_rays(r, dev)
    model = _bf16_model(sd, dev, n, 32, 64, bool(wb))
    model.prop_net.randomized = model.nerf_net.randomized = True
    sdt, hp16 = O.to_torch_state_dict(sd), O.Hyper(num_samples=n, white_bkgd=bool(wb), mlp_bf16=1)
    T_ = lambda a: torch.from_numpy(np.ascontiguousarray(a)).float()  # noqa: E731
    U = lambda a: T_(a).to(dev)  # noqa: E731
    model.prop_net.replay_uniforms, model.nerf_net.replay_uniforms = U(g[pre + "prop_t_rand"]), U(g[pre + "prop_u_unit"])
    t_hat, w_hat = model.prop_net.forward(rays)
    with torch.no_grad():
        _, _, _, t, w, _ = model.nerf_net.forward(rays, t_vals=t_hat.detach(), coarse_weights=w_hat.detach())
    model.zero_grad()
    Loss_prop(t=t, w=w, t_hat=t_hat, w_hat=w_hat).backward()
    hip = {name: p.grad.clone() for name, p in model.named_parameters() if name.startswith("prop_net")}
    _, og = O.prop_step_gradients(O.rays_from_numpy(r), sdt, hp16, t_rand=T_(g[pre + "prop_t_rand"]), u_rand=T_(g[pre + "prop_u_unit"]))
    inh_p, add_p, tot_p, cos_p = _step_errors(hip, og, {nm: g[f"{pre}propstep.{nm}"] for nm in hip}, f"G22 bf16 {tag} proposal step")
    model.prop_net.replay_uniforms, model.nerf_net.replay_uniforms = U(g[pre + "nerf_t_rand"]), U(g[pre + "nerf_u_unit"])
    with torch.no_grad():
        t_hat, w_hat = model.prop_net.forward(rays)
    rgb, _, _, _, fine_w, s_vals = model.nerf_net.forward(rays, t_vals=t_hat, coarse_weights=w_hat)
    loss_nerf, _ = Loss_nerf(input=rgb, target=U(g[pre + "pixels"]))
    model.zero_grad()
    (loss_nerf + 0.01 * Loss_dist(s_vals=s_vals, weights=fine_w)).backward()
    hip = {name: p.grad.clone() for name, p in model.named_parameters() if name.startswith("nerf_net")}
    _, _, og = O.nerf_step_gradients(O.rays_from_numpy(r), sdt, hp16, T_(g[pre + "pixels"]), t_rand=T_(g[pre + "nerf_t_rand"]), u_rand=T_(g[pre + "nerf_u_unit"]))
    inh_n, add_n, tot_n, cos_n = _step_errors(hip, og, {nm: g[f"{pre}nerfstep.{nm}"] for nm in hip}, f"G22 bf16 {tag} NeRF step")
    # the proposal loss divides by w_hat + 1e-6 (near = 0: weights of 1e-6): twice the bound there, as for G21.  Direction: against the
    # reference to COS_MIN - or, where the bf16 FORWARD alone already turns the gradient further than that (lego_16's 64-wide NeRF net on 10
    # jittered rays: 38 % of a tensor's scale inherent), no further than the oracle's bf16 gradients are
    cos_inherent = _cos([(og[nm], g[f"{pre}nerfstep.{nm}"]) for nm in hip])
    assert add_n <= BACKWARD_REL and tot_n <= inh_n + BACKWARD_REL and cos_n >= min(COS_MIN, cos_inherent - 1e-3), (inh_n, add_n, tot_n, cos_n, cos_inherent)
    assert _cos([(hip[nm], og[nm].numpy()) for nm in hip]) >= COS_MIN
    assert add_p <= 2 * BACKWARD_REL and tot_p <= inh_p + 2 * BACKWARD_REL and cos_p >= 0.99, (inh_p, add_p, tot_p, cos_p)
    assert abs(float(loss_nerf.detach()) - float(g[pre + "loss_nerf"])) <= 2e-2 * abs(float(g[pre + "loss_nerf"]))


NERF_REL, PROP_REL = 3e-2, 8e-2  # full width against the fp32 mirrors (measured 1.2e-2 / 1.8e-2: wide layers average the rounding noise)


@pytest.mark.parametrize("B,N,Nf", [(77, 37, None), (130, 32, 48), (3, 11, None)])
def test_ragged_full_width_gradients_bf16_vs_fp32_mirrors(dev, B, N, Nf):
    """Row counts that are no multiple of anything (77 x 37 = 2849 rows: 11 full 256-row tiles + 33 ragged rows; the weight-gradient
    kernel's 64-row stages + a tail added by its reduce kernel), a NeRF stage with its own sample count (`num_samples_fine`), and fewer rows
    than one stage of the MFMA weight-gradient kernel (3 x 11 = 33: the fp32 kernel on widened operands): the bf16 gradients against the fp32
    mirrors', full width.  The first layer's weight gradient reads the 128-wide [hi | lo] feature rows as OVERLAPPING 256-wide rows
    (linear_wgrad_bf16_rows, round 6) on every one of these paths: its last row runs into the tape's next buffer, its result's upper columns
    are never read."""
    from mipnerf360_amd.intern.loss import Loss_dist, Loss_nerf
    from mipnerf360_amd.model import mipNeRF360
    sd = synthetic.make_state_dict(256, 1024, seed=14)
    rays = dev_rays(synthetic.make_rays("lego", B, seed=16), dev)
    pixels = torch.rand(B, 3, generator=torch.Generator().manual_seed(2)).to(dev)
    grads = {}
    for dtype in ("fp32", "bf16"):
        m = mipNeRF360(randomized=False, num_samples=N, num_samples_fine=Nf, hidden_proposal=256, hidden_nerf=1024, white_bkgd=True, device=dev, mlp_dtype=dtype).train()
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        with torch.no_grad():
            t_hat, w_hat = m.prop_net.forward(rays)
        rgb, dist, acc, _, fw, sv = m.nerf_net.forward(rays, t_vals=t_hat, coarse_weights=w_hat)
        assert fw.shape == (B, Nf or N)
        ln, _ = Loss_nerf(input=rgb, target=pixels)
        m.zero_grad()
        (ln + 0.01 * Loss_dist(s_vals=sv, weights=fw) + 0.1 * acc.sum() + 0.05 * dist.sum()).backward()
        grads[dtype] = {n: p.grad.clone() for n, p in m.named_parameters() if n.startswith("nerf_net")}
        del m
    pairs = [(grads["bf16"][n], H(grads["fp32"][n])) for n in grads["fp32"]]
    rel, cos = max(_rel(a, b) for a, b in pairs), _cos(pairs)
    print(f"ragged {B} x {N} (+{Nf}) full width, NeRF step incl. acc / distance gradients: max rel {rel:.2e} cos {cos:.6f}")
    # (33 samples: a tensor's bf16 error is a few samples' error and does not average out - 0.100 measured, the same on the build that still
    # made the zero-padded copy of the first layer's operand; the direction bound is the same for every size)
    bound = 2 * NERF_REL if B * N >= 64 else 5 * NERF_REL
    assert rel <= bound and cos >= COS_MIN and all(torch.isfinite(a).all() for a, _ in pairs), (rel, cos)


def test_full_width_gradients_bf16_vs_fp32_mirrors(dev):
    """Full width (256 / 1024: the MFMA weight-gradient kernel, the first layer's overlapping operand rows, the ring kernel as input-gradient
    kernel) at 512 rays x 64 samples: the bf16 gradients of one NeRF update and one proposal update against the fp32 mirrors' (which
    G13 / G21 pin to the reference), tensor by tensor."""
    from mipnerf360_amd.intern.loss import Loss_dist, Loss_nerf, Loss_prop
    from mipnerf360_amd.model import mipNeRF360
    sd = synthetic.make_state_dict(256, 1024, seed=4)
    r = synthetic.make_rays("garden", 512, seed=6)
    rays = dev_rays(r, dev)
    pixels = torch.rand(512, 3, generator=torch.Generator().manual_seed(1)).to(dev)
    grads = {}
    for dtype in ("fp32", "bf16"):
        m = mipNeRF360(randomized=False, num_samples=64, hidden_proposal=256, hidden_nerf=1024, device=dev, mlp_dtype=dtype).train()
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        t_hat, w_hat = m.prop_net.forward(rays)
        with torch.no_grad():
            _, _, _, t, w, _ = m.nerf_net.forward(rays, t_vals=t_hat.detach(), coarse_weights=w_hat.detach())
        m.zero_grad()
        Loss_prop(t=t, w=w, t_hat=t_hat, w_hat=w_hat).backward()
        gp = {n: p.grad.clone() for n, p in m.named_parameters() if n.startswith("prop_net")}
        rgb, _, _, _, fw, sv = m.nerf_net.forward(rays, t_vals=t_hat.detach(), coarse_weights=w_hat.detach())
        ln, _ = Loss_nerf(input=rgb, target=pixels)
        m.zero_grad()
        (ln + 0.01 * Loss_dist(s_vals=sv, weights=fw)).backward()
        gn = {n: p.grad.clone() for n, p in m.named_parameters() if n.startswith("nerf_net")}
        grads[dtype] = (gp, gn)
        del m
    for which, bound in ((1, NERF_REL), (0, PROP_REL)):
        pairs = [(grads["bf16"][which][n], H(grads["fp32"][which][n])) for n in grads["fp32"][which]]
        rel, cos = max(_rel(a, b) for a, b in pairs), _cos(pairs)
        print(f"full width bf16 vs fp32 mirrors, {'NeRF' if which else 'proposal'} step: max rel {rel:.2e} cos {cos:.6f}")
        assert rel <= bound and cos >= COS_MIN, (which, rel, cos)
        assert all(torch.isfinite(a).all() for a, _ in pairs)


def test_training_descends_end_to_end_bf16(dev):
    """tools/train_demo.py --mlp-dtype bf16: the reference's loop body with the student in bf16 fits the teacher's pixels like the fp32
    student does (PSNR up by > 3 dB in 40 steps, within 1.5 dB of the fp32 run's end point), deterministic and randomized."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import train_demo
    kw = dict(steps=40, rays_n=512, samples=16, hp=32, hn=64, lr=3e-3, log_every=39)
    f32 = train_demo.run(**kw)
    b16 = train_demo.run(**kw, mlp_dtype="bf16")
    first, last = b16["trajectory"][0]["psnr"], b16["trajectory"][-1]["psnr"]
    print("train_demo fp32:", f32["trajectory"], "bf16:", b16["trajectory"])
    assert np.isfinite(last) and last > first + 3.0 and last > f32["trajectory"][-1]["psnr"] - 1.5, (f32, b16)
    rnd = train_demo.run(**kw, mlp_dtype="bf16", randomized=True)
    assert rnd["trajectory"][-1]["psnr"] > rnd["trajectory"][0]["psnr"] + 3.0, rnd


def test_nan_parameter_is_refused_in_training_mode_one_forward_late(dev):
    """the NaN-parameter guard of the bf16 modes used to cost ~30 host syncs per forward in training mode (one per
    tensor, every forward re-packs).  Now: one reduction on the device, its flag copied to pinned memory behind an event and read by the
    NEXT re-pack - no stall in a training step, and a NaN parameter is still refused (the bf16 pipe's ReLU would drop it silently)."""
    sd = synthetic.make_state_dict(64, 128, seed=3)
    m = _bf16_model(sd, dev, 16, 64, 128, False)
    rays = dev_rays(synthetic.make_rays("lego", 32, seed=4), dev)
    m.prop_net.forward(rays)
    with torch.no_grad():
        m.prop_net.model[2].weight[1, 1] = float("nan")
    m.prop_net.forward(rays)  # this packing's flag is looked at by the next one
    with pytest.raises(RuntimeError, match="parameters hold NaN"):
        m.prop_net.forward(rays)
    m.eval()
    with pytest.raises(RuntimeError, match="parameters hold NaN"):
        with torch.no_grad():
            m(rays)


def test_nan_parameter_never_reaches_the_optimizer(dev):
    """the FIRST tape-keeping forward on NaN parameters used to hand finite-looking gradients to the optimizer (the bf16
    ReLU drops the NaN) - the flag was only read by the next forward, and the last step's never.  Now the backward of that very forward
    raises before returning gradients, and `flush_nan_check()` reads the flag of a tape-keeping forward that nothing follows.  A no-grad
    forward in training mode (a model never put into eval(), or the other net's forward inside a training step) still refuses at once."""
    sd = synthetic.make_state_dict(64, 128, seed=3)
    rays = dev_rays(synthetic.make_rays("lego", 32, seed=4), dev)
    m = _bf16_model(sd, dev, 16, 64, 128, False)
    with torch.no_grad():
        m.prop_net.model[2].weight[1, 1] = float("nan")
    t_hat, w_hat = m.prop_net.forward(rays)       # first forward ever: nothing before it could have looked
    with pytest.raises(RuntimeError, match="parameters hold NaN"):
        w_hat.sum().backward()
    assert all(p.grad is None for p in m.prop_net.parameters())
    m2 = _bf16_model(sd, dev, 16, 64, 128, False)
    with torch.no_grad():
        m2.nerf_net.final_color[0].bias[0] = float("nan")
    with torch.no_grad():
        t_hat, w_hat = m2.prop_net.forward(rays)
    m2.flush_nan_check()                          # the proposal net is clean: nothing to report
    out = m2.nerf_net.forward(rays, t_vals=t_hat, coarse_weights=w_hat)   # tape-keeping: the flag is deferred ...
    assert out[0].requires_grad
    with pytest.raises(RuntimeError, match="parameters hold NaN"):
        m2.flush_nan_check()                      # ... and read here although no backward and no further forward ever runs
    m2.flush_nan_check()                          # reported once
    with pytest.raises(RuntimeError, match="parameters hold NaN"), torch.no_grad():
        m2.nerf_net.forward(rays, t_vals=t_hat, coarse_weights=w_hat)     # training mode, no grad: refused at once


def test_params_nan_flag_one_launch(dev):
    """m360_params_nan_flag: one launch over a parameter set (more than 32 tensors: two), NaNs of either sign, at the first / last element."""
    from mipnerf360_amd import ops
    g = torch.Generator().manual_seed(5)
    ts = [torch.randn(n, generator=g).to(dev) for n in (1, 7, 256, 1024 * 1024 + 3, 4097)] + [torch.randn(33, 5, generator=g).to(dev) for _ in range(35)]
    assert int(ops.params_nan_flag(ts).item()) == 0
    neg_nan = np.frombuffer(np.uint32(0xFFC00000).tobytes(), dtype=np.float32)[0]
    for which, where, val in ((0, 0, float("nan")), (3, 1024 * 1024 + 2, neg_nan), (39, 164, float("nan")), (1, 6, neg_nan)):
        keep = ts[which].reshape(-1)[where].item()
        ts[which].reshape(-1)[where] = float(val)
        assert int(ops.params_nan_flag(ts).item()) == 1, (which, where)
        ts[which].reshape(-1)[where] = keep
    ts[2][5] = float("inf")  # an infinity is not a NaN
    assert int(ops.params_nan_flag(ts).item()) == 0


def test_pack_many_same_bits_as_the_per_layer_calls(dev):
    """m360_pack_many (round 6): every packing of a parameter set in ONE launch - each format against its per-layer entry point bit for bit
    (ragged sizes, with and without bias, the heads written side by side into one buffer, more than 16 items = two launches), the NaN flag
    against m360_params_nan_flag, and a bad item anywhere in the list refusing the whole list."""
    from mipnerf360_amd import _lib, ops
    g = torch.Generator().manual_seed(11)
    neg_nan = float(np.frombuffer(np.uint32(0xFFC00000).tobytes(), dtype=np.float32)[0])

    def lin(n, k, bias=True):
        return torch.randn(n, k, generator=g).to(dev), (torch.randn(n, generator=g).to(dev) if bias else None)

    single = {_lib.PACK_F32: ops.pack_linear, _lib.PACK_BF16: ops.pack_linear_bf16, _lib.PACK_BF16X3: ops.pack_linear_bf16x3,
              _lib.PACK_BF16X6: ops.pack_linear_bf16x6}
    single_t = {_lib.PACK_F32_T: ops.pack_linear_transposed, _lib.PACK_BF16_T: ops.pack_linear_bf16_transposed}
    items, want = [], []
    for fmt in single:
        for (n, k, n_pad, k_pad, bias) in ((5, 58, 64, 64, True), (256, 256, 256, 256, True), (33, 100, 64, 128, False), (1, 1, 32, 64, True)):
            w, b = lin(n, k, bias)
            items.append((w, b, n_pad, k_pad, fmt))
            want.append(single[fmt](w, b, n_pad, k_pad))
    for fmt in single_t:
        for (n, k, n_pad, k_pad) in ((5, 58, 64, 64), (256, 256, 256, 256), (100, 33, 128, 64)):
            w, _ = lin(n, k, False)
            items.append((w, None, n_pad, k_pad, fmt))
            want.append((single_t[fmt](w, n_pad, k_pad), None))
    assert len(items) > 16
    outs, flag = ops.pack_many(items, nan_flag=True)
    assert int(flag.item()) == 0
    for i, ((gw, gb), (ww, wb)) in enumerate(zip(outs, want)):
        assert gw.shape == ww.shape and gw.dtype == ww.dtype and torch.equal(gw.view(torch.uint8), ww.view(torch.uint8)), i
        if wb is not None:
            assert torch.equal(gb.view(torch.int32), wb.view(torch.int32)), i
        else:
            assert gb is None, i
    # the heads side by side in one buffer, as model.py packs them
    (w1, b1), (w3, b3) = lin(1, 200), lin(3, 200)
    hw, hb = torch.full((4, 256), 7.0, device=dev), torch.full((4,), 7.0, device=dev)
    ops.pack_many([(w1, b1, 1, 256, _lib.PACK_F32, hw[0:1], hb[0:1]), (w3, b3, 3, 256, _lib.PACK_F32, hw[1:4], hb[1:4])])
    ww, wb = ops.pack_linear(torch.cat([w1, w3]), torch.cat([b1, b3]), 4, 256)
    assert torch.equal(hw, ww) and torch.equal(hb, wb)
    # NaN of either sign in a weight or a bias raises the flag (and is packed as the per-layer calls pack it); an infinity does not
    for what, val in (("w", float("nan")), ("w", neg_nan), ("b", neg_nan), ("inf", float("inf"))):
        w, b = lin(40, 70)
        (w if what != "b" else b).reshape(-1)[-1] = val
        for fmt in (_lib.PACK_F32, _lib.PACK_BF16, _lib.PACK_BF16X3, _lib.PACK_BF16X6):
            (o,), f = ops.pack_many([(w, b, 64, 128, fmt)], nan_flag=True)
            assert int(f.item()) == (0 if what == "inf" else 1), (what, fmt)
            ww, wb = single[fmt](w, b, 64, 128)
            assert torch.equal(o[0].view(torch.uint8), ww.view(torch.uint8)) and torch.equal(o[1].view(torch.int32), wb.view(torch.int32)), (what, fmt)
        assert int(ops.params_nan_flag([w, b]).item()) == (0 if what == "inf" else 1)
    # a NaN OUTSIDE what an item reads is not reported; a bad item (k_pad not a multiple of 64 for a bf16 packing) refuses the list as a whole
    w, b = lin(8, 64)
    canary = torch.full((64, 64), 3.0, device=dev)
    with pytest.raises(RuntimeError, match="item 1"):
        ops.pack_many([(w, b, 64, 64, _lib.PACK_F32, canary, None), (w, b, 64, 96, _lib.PACK_BF16)])
    torch.cuda.synchronize()
    assert bool((canary == 3.0).all()), "a refused list must pack nothing"
    assert ops.pack_many([]) == ([], None)


def test_backward_overlap_same_bits(dev):
    """m360_hyper_t.side (ops.set_backward_overlap): the ReLU mask of a layer's input gradient on a CALLER-OWNED second stream (m360_side_t) beside the weight gradient (the
    default; one striding workgroup per CU) against everything on the caller's stream - the same weight gradients bit for bit (bias gradients: the
    same sums in another order), NeRF and proposal update, at a size where the overlapped form runs (>= 32768 rows: 300 rays x 128 samples, ragged)."""
    from mipnerf360_amd import ops
    from mipnerf360_amd.intern.loss import Loss_dist, Loss_nerf, Loss_prop
    sd = synthetic.make_state_dict(64, 256, seed=21)
    rays = dev_rays(synthetic.make_rays("garden", 300, seed=22), dev)
    pixels = torch.rand(300, 3, generator=torch.Generator().manual_seed(3)).to(dev)
    out = {}
    was = ops.set_backward_overlap(True)
    try:
        for mode in (1, 0, 1):
            ops.set_backward_overlap(bool(mode))
            m = _bf16_model(sd, dev, 128, 64, 256, True)
            t_hat, w_hat = m.prop_net.forward(rays)
            rgb, dist, acc, t, fw, sv = m.nerf_net.forward(rays, t_vals=t_hat.detach(), coarse_weights=w_hat.detach())
            ln, _ = Loss_nerf(input=rgb, target=pixels)
            m.zero_grad()
            (ln + 0.01 * Loss_dist(s_vals=sv, weights=fw)).backward()
            Loss_prop(t=t.detach(), w=fw.detach(), t_hat=t_hat, w_hat=w_hat).backward()
            torch.cuda.synchronize()
            g = {n: p.grad.clone() for n, p in m.named_parameters()}
            assert all(torch.isfinite(v).all() for v in g.values())
            if mode in out:
                assert all(torch.equal(g[n], out[mode][n]) for n in g), "the overlapped backward is not deterministic"
            out[mode] = g
            del m
    finally:
        ops.set_backward_overlap(was)
    # weights: the same bits.  Bias gradients of the layers whose dz came out of a mask pass: in the overlapped form they are the column sums the
    # (throttled) mask kernel forms on its way, on one stream the weight-gradient kernel's ones-product - the same fp32 sums in another order
    diff = [n for n in out[0] if not n.endswith(".bias") and not torch.equal(out[0][n], out[1][n])]
    assert not diff, diff
    for n in out[0]:
        if n.endswith(".bias"):
            scale = float(out[0][n].abs().max())
            assert float((out[0][n] - out[1][n]).abs().max()) <= 2e-5 * scale + 1e-30, (n, float((out[0][n] - out[1][n]).abs().max()), scale)
    assert sum(int(not torch.equal(out[0][n], out[1][n])) for n in out[0] if n.endswith(".bias")) > 0, "the mask kernel's column sums were not used"
    assert float(max(v.abs().max() for v in out[1].values())) > 0


def test_two_threads_train_two_models_on_two_streams(dev):
    """libm360 keeps no state of its own - no switch, no stream, no event (0.2.0).  Two host threads each train their own
    bf16 model on their own stream (own workspace, own m360_side_t for the overlapped backward), one of them with per-call switches flipped
    (plain rows, no chain, 8-wave weight gradient) that the other must never see: every thread's gradients are the bits a serial run of
    the same thread's configuration gives, step after step."""
    import threading

    from mipnerf360_amd import ops
    from mipnerf360_amd.intern.loss import Loss_dist, Loss_nerf
    sds = [synthetic.make_state_dict(64, 256, seed=31), synthetic.make_state_dict(64, 256, seed=32)]
    rays = [dev_rays(synthetic.make_rays("garden", 300, seed=40 + i), dev) for i in range(2)]
    pixels = [torch.rand(300, 3, generator=torch.Generator().manual_seed(50 + i)).to(dev) for i in range(2)]

    def configure(i):
        if i == 1:  # thread-local: the other thread keeps the defaults
            ops.set_paired_rows(False), ops.set_hidden_chain(False), ops.set_wgrad_bf16_form(0)

    def steps(i, stream, n, out):
        try:
            configure(i)
            with torch.cuda.stream(stream):
                for _ in range(n):
                    m = _bf16_model(sds[i], dev, 128, 64, 256, True)
                    t_hat, w_hat = m.prop_net.forward(rays[i])
                    rgb, dist, acc, t, fw, sv = m.nerf_net.forward(rays[i], t_vals=t_hat.detach(), coarse_weights=w_hat.detach())
                    ln, _ = Loss_nerf(input=rgb, target=pixels[i])
                    (ln + 0.01 * Loss_dist(s_vals=sv, weights=fw)).backward()
                    stream.synchronize()
                    out.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        except BaseException as e:  # noqa: BLE001  (reported by the main thread)
            out.append(e)

    def run_thread(i, stream, n):
        out = []
        th = threading.Thread(target=steps, args=(i, stream, n, out))
        return th, out

    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    serial = []
    for i in range(2):  # each configuration alone, in a thread of its own (thread-local switches start from the defaults)
        th, out = run_thread(i, streams[i], 1)
        th.start(), th.join()
        assert out and not isinstance(out[0], BaseException), out
        serial.append(out[0])
    assert ops.tuning_bits() == 0  # nothing leaked into this thread
    pairs = [run_thread(i, streams[i], 4) for i in range(2)]
    for th, _ in pairs:
        th.start()
    for th, _ in pairs:
        th.join()
    for i, (_, out) in enumerate(pairs):
        assert len(out) == 4 and not any(isinstance(o, BaseException) for o in out), out
        for step in out:
            assert step.keys() == serial[i].keys()
            for k in step:
                assert torch.equal(step[k], serial[i][k]), (i, k)
