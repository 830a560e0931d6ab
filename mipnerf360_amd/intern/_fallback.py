"""Names the mirrors do not provide fall through to the reference's own files.

`install_dropin(reference_root=...)` aliases the mirrors to the reference's module names.  The reference's scripts also
import host-side helpers that are out of the hot path (camera paths in intern/pose.py:1-110, `normalize` / `to_float` in
intern/utils.py:4-15, the LR schedule in intern/scheduler.py): those keep running from the reference's own source, loaded
lazily from `<reference_root>/intern/<module>.py` under a private module name.  Without a reference root the missing
names raise AttributeError, as for any module.
"""
from __future__ import annotations

import importlib.util
import os
import sys
from types import ModuleType
from typing import Callable, Dict, Optional

_REF_ROOT: Optional[str] = None
_LOADED: Dict[str, ModuleType] = {}

# The ONLY names that may resolve to the reference's own files: host-side helpers outside the hot path (SURVEY.md §8
# "out of scope").  Every hot-path name (§8 rows a / f) must come from the mirror: if one is ever dropped from a
# mirror it raises AttributeError instead of silently running the reference's CPU code.
ALLOWED: Dict[str, frozenset] = {
    # camera-path generation, intern/pose.py:6-110 (host NumPy, once per video)
    "pose": frozenset({"generate_spiral_cam_to_world", "generate_spherical_cam_to_world", "recenter_poses", "poses_avg",
                       "look_at"}),
    # intern/utils.py:4-15; convolve2d is the scipy wrapper the reference's own pose.py imports at load time (the
    # mirrors' depth_to_normals runs its stencils on the device and never calls it)
    "utils": frozenset({"normalize", "to_float", "convolve2d"}),
}


def set_reference_root(path: Optional[str]) -> None:
    global _REF_ROOT
    _REF_ROOT = os.path.abspath(path) if path else None
    _LOADED.clear()


def reference_root() -> Optional[str]:
    return _REF_ROOT


def _load(modname: str) -> Optional[ModuleType]:
    if _REF_ROOT is None:
        return None
    if modname in _LOADED:
        return _LOADED[modname]
    path = os.path.join(_REF_ROOT, "intern", modname + ".py")
    if not os.path.isfile(path):
        return None
    spec = importlib.util.spec_from_file_location(f"_m360_reference_intern_{modname}", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    prev = sys.dont_write_bytecode
    sys.dont_write_bytecode = True  # never leave __pycache__ files inside the reference checkout
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.dont_write_bytecode = prev
    _LOADED[modname] = mod
    return mod


def make_getattr(modname: str, mirror_name: str) -> Callable[[str], object]:
    """module-level __getattr__ (PEP 562) for the mirror of intern/<modname>.py"""

    def __getattr__(name: str):
        if name.startswith("__"):
            raise AttributeError(name)
        if name not in ALLOWED.get(modname, frozenset()):
            raise AttributeError(f"module {mirror_name!r} has no attribute {name!r} (and {name!r} is not one of the "
                                 f"out-of-scope host helpers that may fall through to the reference)")
        ref = _load(modname)
        if ref is not None and hasattr(ref, name):
            return getattr(ref, name)
        hint = "" if _REF_ROOT else " (call mipnerf360_amd.install_dropin(reference_root=...) to fall back to the reference's own helper)"
        raise AttributeError(f"module {mirror_name!r} has no attribute {name!r}{hint}")

    return __getattr__
