"""Row (f4): checkpoint tooling — the reference's torch.save'd state_dict loads unchanged; architecture is inferred
from tensor shapes; the packed (padded / bf16) layout for C-ABI consumers round-trips."""
import os

import numpy as np
import pytest
import torch

from mipnerf360_amd import checkpoint, synthetic


def _save_reference_style(tmp_path, hp, hn, max_deg=4, seed=0):
    sd = synthetic.make_state_dict(hp, hn, seed=seed, viewdir_max_deg=max_deg)
    path = tmp_path / "model_100.pt"  # train.py:99 naming
    torch.save({k: torch.from_numpy(v) for k, v in sd.items()}, str(path))
    return sd, str(path)


def test_infer_config_and_cpu_roundtrip(tmp_path):
    sd, path = _save_reference_style(tmp_path, 96, 160, max_deg=3)
    cfg = checkpoint.infer_config(torch.load(path))
    assert cfg == dict(hidden_proposal=96, hidden_nerf=160, viewdir_min_deg=0, viewdir_max_deg=3)
    m = checkpoint.load_reference_checkpoint(path, device=torch.device("cpu"), num_samples=32, white_bkgd=True)
    assert m.num_samples == 32 and m.white_bkgd is True and not m.training
    back = checkpoint.to_reference_state_dict(m)
    assert list(back) == list(sd) and all(np.array_equal(back[k].numpy(), sd[k]) for k in sd)
    with pytest.raises(KeyError):
        checkpoint.infer_config({"foo": torch.zeros(1)})


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_packed_export_matches_padded_weights(tmp_path, dtype):
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    sd, path = _save_reference_style(tmp_path, 100, 200, seed=3)  # widths that need padding (128 / 256 for bf16: 128, 256)
    m = checkpoint.load_reference_checkpoint(path, device=dev, num_samples=16, mlp_dtype=dtype)
    packed = checkpoint.export_packed(m)
    pad = 64 if dtype == "bf16" else 32
    in_ch, in_pad, hp_pad, hn_pad, is_bf16, layout, version = (int(x) for x in packed["meta"])
    assert layout == checkpoint.PACKED_LAYOUT == 2 and version >= 101
    assert (in_ch, in_pad, is_bf16) == (58, 64, int(dtype == "bf16"))
    assert hp_pad == -(-100 // pad) * pad and hn_pad == -(-200 // pad) * pad
    w0 = packed["nerf.w0"]
    ref = np.zeros((hn_pad, in_pad), np.float32)
    ref[:200, :58] = sd["nerf_net.model.0.weight"]
    if dtype == "bf16":
        # the first layer of the bf16 mode carries 16 bits of each weight: the [Wh | Wh | Wl] packing of m360_pack_linear_bf16x3
        assert w0.dtype == np.uint16 and w0.shape == (hn_pad, 3 * in_pad)
        got = torch.from_numpy(w0.view(np.int16).copy()).view(torch.bfloat16).float().numpy()
        hi, lo = got[:, :in_pad], got[:, 2 * in_pad:]
        assert np.array_equal(hi, torch.from_numpy(ref).bfloat16().float().numpy()) and np.array_equal(got[:, in_pad:2 * in_pad], hi)
        assert np.array_equal(lo, (torch.from_numpy(ref) - torch.from_numpy(ref).bfloat16().float()).bfloat16().float().numpy())
        w1 = packed["nerf.w1"]              # hidden layers: plain bf16 [n_pad, k_pad]
        ref1 = np.zeros((hn_pad, hn_pad), np.float32)
        ref1[:200, :200] = sd["nerf_net.model.2.weight"]
        got1 = torch.from_numpy(w1.view(np.int16).copy()).view(torch.bfloat16).float().numpy()
        assert w1.shape == (hn_pad, hn_pad) and np.array_equal(got1, torch.from_numpy(ref1).bfloat16().float().numpy())
    else:
        assert w0.shape == (hn_pad, in_pad) and np.array_equal(w0, ref)
    assert packed["nerf.head_w"].shape == (4, hn_pad) and packed["prop.head_w"].shape == (1, hp_pad)
    assert np.array_equal(packed["nerf.head_w"][0, :200], sd["nerf_net.final_density.0.weight"][0])
    assert np.array_equal(packed["nerf.head_w"][1:, :200], sd["nerf_net.final_color.0.weight"])
    checkpoint.save_packed(m, str(tmp_path / "packed.npz"))
    again = checkpoint.load_packed(str(tmp_path / "packed.npz"))
    assert all(np.array_equal(again[k], packed[k]) for k in packed)
    # a file of another layout is refused, never read past the end of its first-layer matrices
    old = dict(packed, meta=packed["meta"][:5])
    np.savez(str(tmp_path / "old.npz"), **old)
    with pytest.raises(ValueError, match="layout"):
        checkpoint.load_packed(str(tmp_path / "old.npz"))
    if dtype == "bf16":  # round 3's first layer: [n_pad, in_pad]
        bad = dict(packed)
        bad["nerf.w0"] = packed["nerf.w0"][:, :in_pad].copy()
        with pytest.raises(ValueError, match="nerf.w0"):
            checkpoint.validate_packed(bad)
    # the loaded model renders (padding path: 100 -> 128, 200 -> 224/256 columns)
    from mipnerf360_amd.intern.ray import Rays
    from oracle import ref_path as O
    r = synthetic.make_rays("lego", 40, seed=2)
    with torch.no_grad():
        rgb, dist, acc = m(Rays(*[torch.from_numpy(r[k]).to(dev) for k in synthetic.RAY_FIELDS]))
    o = O.forward(O.rays_from_numpy(r), O.to_torch_state_dict(sd), O.Hyper(num_samples=16, mlp_bf16=(dtype == "bf16")))
    tol = 6e-3 if dtype == "bf16" else 1e-4
    assert float((rgb.cpu() - o[0]).abs().max()) <= tol and float((acc.cpu() - o[2]).abs().max()) <= tol


@pytest.mark.gpu
def test_render_checkpoint_tool_writes_frames(tmp_path):
    """tools/render_checkpoint.py: the flow of the reference's test.py (checkpoint -> frames + depth / normal maps as PNG)
    on the mirrors, from a state_dict saved the way train.py saves it."""
    import subprocess
    import sys
    dev = torch.device("cuda:0")
    sd = {k: torch.from_numpy(v) for k, v in synthetic.make_state_dict(64, 96, seed=9).items()}
    ck = tmp_path / "model_0.pt"
    torch.save(sd, str(ck))
    out = tmp_path / "frames"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, os.path.join(root, "tools", "render_checkpoint.py"), str(ck), str(out), "--width", "40",
                          "--height", "30", "--views", "2", "--num-samples", "16", "--chunks", "128", "--ndc"],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:]
    names = sorted(os.listdir(out))
    assert names == ["dist_0000.png", "dist_0001.png", "norm_0000.png", "norm_0001.png", "rgb_0000.png", "rgb_0001.png"]
    for n in names:
        raw = open(out / n, "rb").read()
        assert raw[:8] == b"\x89PNG\r\n\x1a\n" and b"IHDR" in raw[:32] and len(raw) > 100


def test_validate_packed_on_cpu():
    """checkpoint.validate_packed needs no device: shapes by mode, layout entry required."""
    def fake(mode, layout=checkpoint.PACKED_LAYOUT, in_pad=64, hp=64, hn=128):
        first = {0: 1, 1: 3, 2: 6}[mode] * in_pad
        hid = 3 if mode == 2 else 1
        dt = np.float32 if mode == 0 else np.uint16
        p = {"meta": np.array([58, in_pad, hp, hn, mode, layout, 101], np.int32)}
        for name, w, layers in (("prop", hp, 4), ("nerf", hn, 8)):
            for i in range(layers):
                p[f"{name}.w{i}"] = np.zeros((w, first if i == 0 else hid * w), dt)
                p[f"{name}.b{i}"] = np.zeros(w, np.float32)
        return p
    for mode in (0, 1, 2):
        checkpoint.validate_packed(fake(mode))
    with pytest.raises(ValueError, match="layout 1"):
        checkpoint.validate_packed(fake(1, layout=1))
    p = fake(2)
    p["prop.w0"] = p["prop.w0"][:, :3 * 64]  # the bf16x3 first layer of round 3
    with pytest.raises(ValueError, match="prop.w0"):
        checkpoint.validate_packed(p)
    p = fake(1)
    p["meta"] = p["meta"][:5]
    with pytest.raises(ValueError, match="without a layout entry"):
        checkpoint.validate_packed(p)
