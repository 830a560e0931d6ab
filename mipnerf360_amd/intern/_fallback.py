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
    spec.loader.exec_module(mod)
    _LOADED[modname] = mod
    return mod


def make_getattr(modname: str, mirror_name: str) -> Callable[[str], object]:
    """module-level __getattr__ (PEP 562) for the mirror of intern/<modname>.py"""

    def __getattr__(name: str):
        if name.startswith("__"):
            raise AttributeError(name)
        ref = _load(modname)
        if ref is not None and hasattr(ref, name):
            return getattr(ref, name)
        hint = "" if _REF_ROOT else " (call mipnerf360_amd.install_dropin(reference_root=...) to fall back to the reference's own helper)"
        raise AttributeError(f"module {mirror_name!r} has no attribute {name!r}{hint}")

    return __getattr__
