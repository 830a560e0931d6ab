"""Checkpoint tooling.

The reference stores `model.state_dict()` with `torch.save` (train.py:99,102) and restores it with
`model.load_state_dict(torch.load(path))` (test.py:34, video.py:29).  The mirrors keep that wire format
(30 tensors), so reference checkpoints load unchanged; this module adds

  * `infer_config(state_dict)`          - architecture (widths, view-direction degrees) from the tensor shapes,
  * `load_reference_checkpoint(path)`   - build a `mipNeRF360` of the right shape and load the file,
  * `export_packed(model)` / `save_packed` / `load_packed`
                                        - the zero-padded, k-contiguous (optionally bf16) layout that the C-ABI
                                          consumes (`m360_model_t`), for C/C++ callers that do not run Python,
  * `to_reference_state_dict(model)`    - back to the reference's layout (CPU fp32 tensors) for `torch.save`.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Mapping

import numpy as np
import torch

PROP_KEYS = [f"prop_net.model.{i}" for i in (0, 2, 4, 6, 8)]
NERF_KEYS = [f"nerf_net.model.{i}" for i in range(0, 16, 2)] + ["nerf_net.final_density.0", "nerf_net.final_color.0"]


def infer_config(state_dict: Mapping[str, torch.Tensor]) -> Dict[str, int]:
    """Constructor kwargs that the tensor shapes determine: hidden widths and the number of view-direction
    octaves (input size = 42 + 4 * (max_deg - min_deg), model.py:39,127; min_deg itself is not recoverable,
    the reference default 0 is assumed)."""
    missing = [k + s for k in PROP_KEYS + NERF_KEYS for s in (".weight", ".bias") if k + s not in state_dict]
    if missing:
        raise KeyError(f"not a mipNeRF360 state_dict of the reference layout, missing {missing[:4]}...")
    hp, in_p = state_dict["prop_net.model.0.weight"].shape
    hn, in_n = state_dict["nerf_net.model.0.weight"].shape
    if in_p != in_n or (in_p - 42) % 4 or in_p < 42:
        raise ValueError(f"unexpected input sizes {in_p} / {in_n}")
    return dict(hidden_proposal=int(hp), hidden_nerf=int(hn), viewdir_min_deg=0, viewdir_max_deg=(in_p - 42) // 4)


def load_reference_checkpoint(path: str, device=torch.device("cuda"), **model_kwargs):
    """`torch.load` a file written by the reference's train.py and return a ready `mipNeRF360` (eval mode).
    Keyword arguments that the file cannot know (num_samples, white_bkgd, ...) are passed through."""
    from .model import mipNeRF360
    sd = torch.load(path, map_location="cpu")
    cfg = infer_config(sd)
    cfg.update(model_kwargs)
    model = mipNeRF360(device=device, **cfg)
    model.load_state_dict(sd)
    return model.eval()


def to_reference_state_dict(model) -> "OrderedDict[str, torch.Tensor]":
    """The reference's checkpoint layout (CPU, fp32), ready for `torch.save` (train.py:102)."""
    return OrderedDict((k, v.detach().float().cpu().clone()) for k, v in model.state_dict().items())


def export_packed(model) -> Dict[str, np.ndarray]:
    """Packed weights exactly as the kernels read them (include/m360.h, m360_model_t): per layer a zero-padded
    [n_pad, k_pad] matrix (k contiguous; float32, or bf16 bit patterns as uint16 for mlp_dtype='bf16'; for
    mlp_dtype='bf16x3' the [n_pad, 3 k_pad] = [Wh | Wh | Wl] layout of m360_pack_linear_bf16x3; in both bf16 modes layer 0 is
    the [n_pad, 6 in_pad] "x6" layout of m360_pack_linear_bf16x6) and a [n_pad] fp32 bias;
    heads as fp32 [H, h_pad] + [H]; meta[4] = m360_model_t.mlp_bf16 (0 / 1 / 2).  Needs a HIP device (packing runs in
    m360_pack_linear*)."""
    prop, nerf = model.prop_net._pack(), model.nerf_net._pack()
    out: Dict[str, np.ndarray] = {"meta": np.array([model.prop_net.input_size, prop.in_pad, prop.h_pad, nerf.h_pad,
                                                    int(prop.bf16)], dtype=np.int32)}

    def host(t: torch.Tensor) -> np.ndarray:
        return (t.view(torch.int16) if t.dtype == torch.bfloat16 else t).detach().cpu().numpy().copy()

    for name, p in (("prop", prop), ("nerf", nerf)):
        for i, (w, b) in enumerate(zip(p.w, p.b)):
            out[f"{name}.w{i}"] = host(w).view(np.uint16) if w.dtype == torch.bfloat16 else host(w)
            out[f"{name}.b{i}"] = host(b)
        out[f"{name}.head_w"], out[f"{name}.head_b"] = host(p.head_w), host(p.head_b)
    return out


def save_packed(model, path: str) -> None:
    np.savez(path, **export_packed(model))


def load_packed(path: str) -> Dict[str, np.ndarray]:
    with np.load(path) as z:
        return {k: z[k] for k in z.files}
