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


PACKED_LAYOUT = 2  # include/m360.h: M360_PACKED_LAYOUT (meta[5] of a packed file)


def export_packed(model) -> Dict[str, np.ndarray]:
    """Packed weights exactly as the kernels read them (include/m360.h, m360_model_t): per layer a zero-padded
    [n_pad, k_pad] matrix (k contiguous; float32, or bf16 bit patterns as uint16 in the bf16 modes) and a [n_pad] fp32 bias; heads as
    fp32 [H, h_pad] + [H].  Layouts by mode (meta[4] = m360_model_t.mlp_bf16):
      0 fp32:    every layer [n_pad, k_pad] float32;
      1 bf16:    hidden layers [n_pad, k_pad] (m360_pack_linear_bf16); layer 0 [n_pad, 3 in_pad] = [Wh | Wh | Wl] (m360_pack_linear_bf16x3);
      2 bf16x3:  hidden layers [n_pad, 3 k_pad] = [Wh | Wh | Wl]; layer 0 [n_pad, 6 in_pad] = "x6" (m360_pack_linear_bf16x6).
    meta = [in_ch, in_pad, hp_pad, hn_pad, mlp_bf16, packed layout (= m360_model_t.packed_layout), M360_VERSION of the writer].
    Needs a HIP device (packing runs in m360_pack_linear*)."""
    from . import _lib
    prop, nerf = model.prop_net._pack(), model.nerf_net._pack()
    out: Dict[str, np.ndarray] = {"meta": np.array([model.prop_net.input_size, prop.in_pad, prop.h_pad, nerf.h_pad,
                                                    int(prop.bf16), PACKED_LAYOUT, int(_lib.lib().m360_version())], dtype=np.int32)}

    def host(t: torch.Tensor) -> np.ndarray:
        return (t.view(torch.int16) if t.dtype == torch.bfloat16 else t).detach().cpu().numpy().copy()

    for name, p in (("prop", prop), ("nerf", nerf)):
        for i, (w, b) in enumerate(zip(p.w, p.b)):
            out[f"{name}.w{i}"] = host(w).view(np.uint16) if w.dtype == torch.bfloat16 else host(w)
            out[f"{name}.b{i}"] = host(b)
        out[f"{name}.head_w"], out[f"{name}.head_b"] = host(p.head_w), host(p.head_b)
    validate_packed(out)
    return out


def validate_packed(packed: Mapping[str, np.ndarray]) -> None:
    """Refuse a packed model whose layout this library does not read (the first-layer packings of the bf16 modes changed and
    the kernels cannot see a buffer's size): the layout entry must be there and current, and every matrix must have the shape its
    mode says."""
    meta = np.asarray(packed["meta"]).astype(np.int64).ravel()
    if meta.size < 6:
        raise ValueError("packed model without a layout entry (meta has %d < 6 values): written before layout %d - its first-layer "
                         "packings are not what the bf16 kernels read today; export it again" % (meta.size, PACKED_LAYOUT))
    in_ch, in_pad, hp, hn, mode, layout = (int(v) for v in meta[:6])
    if layout != PACKED_LAYOUT:
        raise ValueError(f"packed model of layout {layout}; this library reads layout {PACKED_LAYOUT}: export it again")
    if mode not in (0, 1, 2):
        raise ValueError(f"packed model: unknown mlp_bf16 mode {mode}")
    first = {0: 1, 1: 3, 2: 6}[mode] * in_pad
    hid = 3 if mode == 2 else 1
    want_dtype = np.float32 if mode == 0 else np.uint16
    for name, width, layers in (("prop", hp, 4), ("nerf", hn, 8)):
        for i in range(layers):
            w = packed[f"{name}.w{i}"]
            shape = (width, first if i == 0 else hid * width)
            if tuple(w.shape) != shape or w.dtype != want_dtype:
                raise ValueError(f"packed model: {name}.w{i} is {w.dtype}{tuple(w.shape)}, mode {mode} needs {np.dtype(want_dtype)}{shape}")
            if tuple(packed[f"{name}.b{i}"].shape) != (width,):
                raise ValueError(f"packed model: {name}.b{i} has shape {tuple(packed[f'{name}.b{i}'].shape)}, expected ({width},)")


def save_packed(model, path: str) -> None:
    np.savez(path, **export_packed(model))


def load_packed(path: str) -> Dict[str, np.ndarray]:
    """-> the arrays of a `save_packed` file; raises ValueError for a file of another layout (validate_packed)."""
    with np.load(path) as z:
        packed = {k: z[k] for k in z.files}
    validate_packed(packed)
    return packed
