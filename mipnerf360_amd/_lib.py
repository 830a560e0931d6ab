"""ctypes binding of libm360.so (C-ABI declared in include/m360.h).

There is NO fallback: if the shared object is missing or a call fails, a
RuntimeError is raised.  Build with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C mipnerf360_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("M360_LIB") or os.path.join(_HERE, "libm360.so")  # M360_LIB: A/B a diagnostic build
CSRC_DIR = os.path.join(_HERE, "csrc")

M360_OK = 0
ACT_NONE, ACT_RELU, ACT_SIGMOID = 0, 1, 2
# include/m360.h: "paired rows" flags (OR-ed into act of the bf16 linear calls) and the call kinds of m360_linear_bf16_rows_pairable
ROWS_PAIRED_IN, ROWS_PAIRED_OUT, STORES_TEMPORAL = 0x100, 0x200, 0x400
PAIRABLE_LINEAR, PAIRABLE_X3, PAIRABLE_SPLIT, PAIRABLE_X3_BF16OUT, PAIRABLE_HEADS, PAIRABLE_HEADS_X3 = range(6)
# record kinds of the event recorder (include/m360.h, "measurement")
K_LINEAR, K_LINEAR_BF16, K_ENCODE, K_PROP_FINISH, K_NERF_FINISH, K_WGRAD, K_DGRAD, K_LINEAR_HEADS = range(8)

_f = C.POINTER(C.c_float)
_vp = C.c_void_p


class RaysStruct(C.Structure):  # m360_rays_t
    _fields_ = [(n, _vp) for n in ("origins", "directions", "viewdirs", "radii", "near", "far")]


class ModelStruct(C.Structure):  # m360_model_t
    _fields_ = [("in_ch", C.c_int), ("in_pad", C.c_int), ("hp_pad", C.c_int), ("hn_pad", C.c_int),
                ("prop_w", _vp * 4), ("prop_b", _vp * 4), ("prop_head_w", _vp), ("prop_head_b", _vp),
                ("nerf_w", _vp * 8), ("nerf_b", _vp * 8), ("nerf_head_w", _vp), ("nerf_head_b", _vp),
                ("mlp_bf16", C.c_int), ("packed_layout", C.c_int)]


PACKED_LAYOUT = 2  # include/m360.h: M360_PACKED_LAYOUT


# m360_hyper_t.tuning: per-call A/B switches (include/m360.h, M360_TUNE_*)
TUNE_NO_HIDDEN_CHAIN, TUNE_PLAIN_ROWS, TUNE_WGRAD_FORM0, TUNE_CHAIN_COOPERATIVE, TUNE_CHAIN_UNGATED = 1, 2, 4, 8, 16


PACK_F32, PACK_BF16, PACK_BF16X3, PACK_BF16X6, PACK_F32_T, PACK_BF16_T = range(6)  # include/m360.h: M360_PACK_*


class PackItem(C.Structure):  # include/m360.h: m360_pack_item_t
    _fields_ = [("w", C.c_void_p), ("b", C.c_void_p), ("w_packed", C.c_void_p), ("b_packed", C.c_void_p),
                ("n_out", C.c_int), ("k_in", C.c_int), ("n_pad", C.c_int), ("k_pad", C.c_int), ("format", C.c_int), ("reserved", C.c_int)]


class HyperStruct(C.Structure):  # m360_hyper_t
    _fields_ = [("num_samples", C.c_int), ("viewdir_min_deg", C.c_int), ("viewdir_max_deg", C.c_int),
                ("white_bkgd", C.c_int), ("density_bias", C.c_float), ("rgb_padding", C.c_float),
                ("resample_padding", C.c_float), ("num_samples_fine", C.c_int), ("norm_group_rays", C.c_int),
                ("prof", C.c_void_p), ("rays_mutated", C.c_int), ("randomized", C.c_int), ("rng_seed", C.c_ulonglong),
                ("rng_offset", C.c_ulonglong), ("tuning", C.c_uint), ("side", C.c_void_p),
                ("chain_debug_wait_ticks", C.c_long), ("chain_debug_fault", C.c_int)]


class OutputsStruct(C.Structure):  # m360_outputs_t
    _fields_ = [(n, _vp) for n in ("rgb", "distance", "acc", "t_hat", "w_hat", "t_vals", "fine_w", "s_vals")]


class MlpTransposedStruct(C.Structure):  # m360_mlp_transposed_t
    _fields_ = [("w_t", _vp * 8)]


class MlpGradsStruct(C.Structure):  # m360_mlp_grads_t
    _fields_ = [("w", _vp * 8), ("b", _vp * 8), ("head_w", _vp), ("head_b", _vp)]


_i, _l, _fl, _sz = C.c_int, C.c_long, C.c_float, C.c_size_t
_P = C.POINTER

# name -> (restype, argtypes); mirrors include/m360.h one to one
SIGNATURES = {
    "m360_version": (_i, []),
    "m360_last_error": (C.c_char_p, []),
    "m360_device_count": (_i, []),
    "m360_sample_t": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "m360_sample_t_philox": (_i, [_vp, _vp, _i, _i, C.c_ulonglong, C.c_ulonglong, _vp, _vp]),
    "m360_philox_uniform": (_i, [C.c_ulonglong, C.c_ulonglong, _i, _l, _vp, _vp]),
    "m360_sorted_pdf_philox": (_i, [_vp, _vp, _i, _i, _i, C.c_ulonglong, C.c_ulonglong, _vp, _vp]),
    "m360_resample_t_philox": (_i, [_vp, _vp, _i, _i, _i, _fl, C.c_ulonglong, C.c_ulonglong, _vp, _vp]),
    "m360_g": (_i, [_vp, _l, _vp, _vp]),
    "m360_s_to_t": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "m360_contract": (_i, [_vp, _l, _vp, _vp, _sz, _vp]),
    "m360_t_to_s": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "m360_frustum_moments": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "m360_gaussian_to_xyz": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "m360_frustum_moments_unstable": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "m360_gaussian_to_xyz_diag": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp]),
    "m360_contract_workspace_bytes": (_sz, []),
    "m360_gaussian_contract": (_i, [_vp, _vp, _l, _vp, _vp, _vp, _sz, _vp]),
    "m360_para_rays": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "m360_ipe": (_i, [_vp, _vp, _l, _vp, _vp]),
    "m360_viewdir_enc": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "m360_encode_features": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _sz, _vp]),
    "m360_encode_features_grouped": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "m360_pack_linear": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "m360_linear": (_i, [_vp, _l, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    "m360_linear_balanced": (_i, [_vp, _l, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp]),
    "m360_pack_linear_transposed": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "m360_linear_dgrad": (_i, [_vp, _l, _i, _vp, _i, _i, _vp, _vp, _i, _vp]),
    "m360_linear_wgrad_workspace_bytes": (_sz, [_l, _i, _i]),
    "m360_linear_wgrad": (_i, [_vp, _i, _vp, _i, _l, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "m360_side_create": (_i, [_P(_vp)]),
    "m360_side_destroy": (None, [_vp]),
    "m360_params_nan_flag": (_i, [_vp, _vp, _i, _vp, _vp]),
    "m360_pack_many": (_i, [_vp, _i, _vp, _vp]),
    "m360_pack_linear_bf16_transposed": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "m360_linear_dgrad_bf16": (_i, [_vp, _l, _i, _vp, _i, _i, _vp, _vp, _i, _vp]),
    "m360_linear_wgrad_bf16_workspace_bytes": (_sz, [_l, _i, _i]),
    "m360_linear_wgrad_bf16": (_i, [_vp, _i, _vp, _i, _l, _i, _i, _vp, _vp, _vp, _sz, C.c_uint, _vp]),
    "m360_pack_linear_bf16": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "m360_linear_bf16": (_i, [_vp, _l, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    "m360_pack_linear_bf16x3": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "m360_linear_bf16x3": (_i, [_vp, _l, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    "m360_linear_bf16x3_bf16out": (_i, [_vp, _l, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    "m360_mlp_chain_bf16_supported": (_i, [_l, _i, _i]),
    "m360_mlp_chain_bf16_workspace": (C.c_size_t, [_l, _i]),
    "m360_mlp_chain_bf16": (_i, [_vp, _vp, _l, _i, _vp, _vp, _i, _i, _vp, _P(HyperStruct), _vp]),
    "m360_linear_bf16_rows_pairable": (_i, [_i, _i, _i]),
    "m360_mlp_chain_bf16_safe": (_i, [_vp, _vp, _vp, _l, _i, _vp, _vp, _i, _i, _vp, _P(HyperStruct), _vp]),
    "m360_mlp_chain_bf16x3_safe": (_i, [_vp, _vp, _vp, _l, _i, _vp, _vp, _i, _i, _vp, _P(HyperStruct), _vp]),
    "m360_workspace_init": (_i, [_vp, _vp]),
    "m360_workspace_status": (_i, [_vp, _P(C.c_uint), _vp]),
    "m360_pack_linear_bf16x6": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "m360_linear_bf16_split": (_i, [_vp, _l, _i, _vp, _vp, _i, _i, _i, _vp, _i, _vp]),
    "m360_encode_features_bf16": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _sz, _vp]),
    "m360_prop_finish_bf16": (_i, [_vp, _i, _vp, _vp, _i, _fl, _vp, _vp, _vp, _i, _i, _i, _fl, _vp, _vp, _vp]),
    "m360_nerf_finish_bf16": (_i, [_vp, _i, _vp, _vp, _i, _fl, _fl, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "m360_density_to_weight": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "m360_sorted_pdf": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "m360_resample_t": (_i, [_vp, _vp, _vp, _i, _i, _fl, _vp, _vp]),
    "m360_resample_t_n": (_i, [_vp, _vp, _vp, _i, _i, _i, _fl, _vp, _vp]),
    "m360_prop_finish_n": (_i, [_vp, _i, _vp, _vp, _i, _fl, _vp, _vp, _vp, _i, _i, _i, _fl, _vp, _vp, _vp]),
    "m360_volumetric_rendering": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "m360_to8b": (_i, [_vp, _l, _vp, _vp]),
    "m360_loss_workspace_bytes": (_sz, [_i, _i]),
    "m360_loss_prop": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "m360_loss_dist": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "m360_loss_nerf": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "m360_visualize_workspace_bytes": (_sz, []),
    "m360_depth_to_normals": (_i, [_vp, _i, _i, _vp, _vp]),
    "m360_sinebow": (_i, [_vp, _l, _vp, _vp]),
    "m360_visualize_normals": (_i, [_vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "m360_visualize_depth": (_i, [_vp, _vp, _i, _i, _fl, _fl, _i, _i, _fl, _vp, _vp, _sz, _vp]),
    "m360_visualize_depth_ex_workspace_bytes": (_sz, [_i, _i]),
    "m360_visualize_depth_ex": (_i, [_vp, _vp, _i, _i, _fl, _fl, _i, _i, _fl, _i, _fl, _vp, _vp, _vp, _vp, _sz, _vp]),
    "m360_visualize_composite": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp]),
    "m360_generate_rays": (_i, [_vp, _i, _i, _i, _fl, _fl, _fl, _i, _fl, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "m360_generate_rays_span": (_i, [_vp, _i, _i, _i, _fl, _fl, _fl, _i, _fl, _l, _l, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "m360_convert_to_ndc": (_i, [_vp, _vp, _l, _fl, _i, _i, _fl, _vp, _vp, _vp]),
    "m360_prop_finish": (_i, [_vp, _i, _vp, _vp, _i, _fl, _vp, _vp, _vp, _i, _i, _fl, _vp, _vp, _vp]),
    "m360_nerf_finish": (_i, [_vp, _i, _vp, _vp, _i, _fl, _fl, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "m360_linear_heads_fused_rows": (_l, [_l, _i, _i]),
    "m360_linear_heads_slots": (_i, [_i, _i]),
    "m360_linear_heads_slots_bf16": (_i, [_i, _i, _i, _i]),
    "m360_linear_heads": (_i, [_vp, _l, _i, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _i, _vp, _vp]),
    "m360_linear_heads_bf16": (_i, [_vp, _l, _i, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _i, _vp, _vp]),
    "m360_linear_heads_bf16x3": (_i, [_vp, _l, _i, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _i, _vp, _vp]),
    "m360_prop_finish_fused": (_i, [_vp, _i, _i, _vp, _l, _i, _vp, _vp, _i, _fl, _vp, _vp, _vp, _i, _i, _i, _fl, _vp, _vp, _vp]),
    "m360_nerf_finish_fused": (_i, [_vp, _i, _i, _vp, _l, _i, _vp, _vp, _i, _fl, _fl, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp,
                                    _vp]),
    "m360_nerf_finish_outputs": (_i, [_vp, _i, _i, _vp, _l, _i, _vp, _vp, _i, _fl, _fl, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp,
                                      _vp, _vp, _vp, _vp, _vp]),
    "m360_forward_workspace_bytes": (_sz, [_i, _i, _P(ModelStruct)]),
    "m360_prop_forward": (_i, [_P(RaysStruct), _P(ModelStruct), _P(HyperStruct), _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "m360_nerf_forward": (_i, [_P(RaysStruct), _P(ModelStruct), _P(HyperStruct), _i, _vp, _vp, _vp,
                               _P(OutputsStruct), _vp, _sz, _vp]),
    "m360_prof_create": (_vp, [_i]),
    "m360_prof_destroy": (None, [_vp]),
    "m360_prof_count": (_i, [_vp]),
    "m360_prof_reset": (_i, [_vp]),
    "m360_prof_read": (_i, [_vp, _i, _P(C.c_float), _P(_i), _P(C.c_long), _P(_i), _P(_i)]),
    "m360_forward": (_i, [_P(RaysStruct), _P(ModelStruct), _P(HyperStruct), _i, _P(OutputsStruct), _vp, _sz, _vp]),
    "m360_mean_sumsq": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "m360_encode_features_ext_norm": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "m360_prop_forward_from_t": (_i, [_P(RaysStruct), _P(ModelStruct), _P(HyperStruct), _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "m360_nerf_forward_from_t": (_i, [_P(RaysStruct), _P(ModelStruct), _P(HyperStruct), _i, _vp, _vp, _P(OutputsStruct), _vp,
                                      _sz, _vp]),
    "m360_finish_backward_workspace_bytes": (_sz, [_i, _i, _i]),
    "m360_prop_finish_backward": (_i, [_vp, _i, _vp, _vp, _i, _fl, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "m360_nerf_finish_backward": (_i, [_vp, _i, _vp, _vp, _i, _fl, _fl, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                                       _vp, _vp, _sz, _vp]),
    "m360_train_tape_bytes": (_sz, [_i, _i, _P(ModelStruct), _i]),
    "m360_backward_workspace_bytes": (_sz, [_i, _i, _P(ModelStruct), _i]),
    "m360_prop_forward_train": (_i, [_P(RaysStruct), _P(ModelStruct), _P(HyperStruct), _i, _vp, _vp, _vp, _vp, _sz, _vp, _sz,
                                     _vp]),
    "m360_nerf_forward_train": (_i, [_P(RaysStruct), _P(ModelStruct), _P(HyperStruct), _i, _vp, _vp, _vp, _P(OutputsStruct),
                                     _vp, _sz, _vp, _sz, _vp]),
    "m360_prop_backward": (_i, [_P(RaysStruct), _P(ModelStruct), _P(MlpTransposedStruct), _P(HyperStruct), _i, _vp, _sz, _vp,
                                _P(MlpGradsStruct), _vp, _sz, _vp]),
    "m360_nerf_backward": (_i, [_P(RaysStruct), _P(ModelStruct), _P(MlpTransposedStruct), _P(HyperStruct), _i, _vp, _sz, _vp,
                                _vp, _vp, _vp, _P(MlpGradsStruct), _vp, _sz, _vp]),
}

_lib = None
_loaded = {}  # path -> bound CDLL
DIAG_LIB_PATH = os.path.join(_HERE, "libm360_diag.so")  # make -C mipnerf360_amd/csrc diag: the same sources with -DM360_DIAG (test hooks, stamped kernels)


def build(verbose: bool = False, diag: bool = False) -> str:
    """Compile libm360.so for gfx950 with hipcc (cross-compiles without a GPU); diag=True: libm360_diag.so as well (the same sources with
    -DM360_DIAG: stamped kernels for tools/, the layer chain's fault-injection hooks for tests/test_gpu_chain.py - never loaded by the package)."""
    for target in ([], ["diag"]) if diag else ([],):
        cmd = ["make", "-C", CSRC_DIR, "-j4"] + target
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if verbose or res.returncode != 0:
            print(res.stdout)
        if res.returncode != 0:
            raise RuntimeError(f"building {'libm360_diag.so' if target else 'libm360.so'} failed (see output above)")
    return LIB_PATH


def lib() -> C.CDLL:
    """The loaded library; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is required and there is no CPU fallback. "
                "Build it with `make -C mipnerf360_amd/csrc` (needs hipcc, gfx950).")
        # libm360 and PyTorch must share ONE HIP runtime (torch owns the device memory and the
        # streams we launch on): load torch's bundled libamdhip64 first so the loader reuses it.
        _lib = _load(LIB_PATH)
    return _lib


def _load(path: str) -> C.CDLL:
    if path not in _loaded:
        import torch  # noqa: F401
        handle = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        handle.m360_path = path
        _loaded[path] = handle
    return _loaded[path]


class use_library:
    """Tests / diagnostics: route every call of this process through another build of the library for the duration of the block -
    `with _lib.use_library(_lib.DIAG_LIB_PATH): ...` runs the diagnostics build, the only one that honours the layer chain's fault-injection
    hooks (m360_hyper_t.chain_debug_*).  Handles made by one build (event recorders, second streams) must not be handed to the other."""

    def __init__(self, path: str):
        self.path = path

    def __enter__(self):
        global _lib
        if not os.path.exists(self.path):
            raise RuntimeError(f"{self.path} not found (make -C mipnerf360_amd/csrc diag)")
        self.prev = lib()
        _lib = _load(self.path)
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.prev


class Prof:
    """Caller-owned HIP-event recorder (m360_prof_t): attach to a model with `model.prof = Prof(n)`; the stage drivers
    then time every kernel they launch on the launch stream.  `records()` blocks until the recorded work is done."""

    def __init__(self, capacity: int):
        self.handle = lib().m360_prof_create(int(capacity))
        if not self.handle:
            raise RuntimeError(f"m360_prof_create({capacity}) failed: {last_error()}")

    def reset(self) -> None:
        lib().m360_prof_reset(self.handle)

    def records(self):
        """-> list of dicts {kind, ms, M, n_pad, k_pad}"""
        out = []
        ms, kind, M_, n_, k_ = C.c_float(), C.c_int(), C.c_long(), C.c_int(), C.c_int()
        for i in range(lib().m360_prof_count(self.handle)):
            check(lib().m360_prof_read(self.handle, i, C.byref(ms), C.byref(kind), C.byref(M_), C.byref(n_), C.byref(k_)),
                  "m360_prof_read")
            out.append(dict(kind=kind.value, ms=ms.value, M=M_.value, n_pad=n_.value, k_pad=k_.value))
        return out

    def close(self) -> None:
        if getattr(self, "handle", None):
            lib().m360_prof_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def last_error() -> str:
    msg = lib().m360_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, what: str = "") -> None:
    if rc != M360_OK:
        raise RuntimeError(f"libm360 {what} failed with code {rc}: {last_error()}")
