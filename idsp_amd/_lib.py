"""Loader for ``libidsp_hip.so`` (the HIP engine).  There is no fallback: if the
shared object is missing or does not export the full C ABI the import of the
processing layer fails loudly."""
from __future__ import annotations

import ctypes
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# IDSP_HIP_LIB overrides the path (A/B-testing a differently built engine); still no fallback.
LIB_PATH = os.environ.get("IDSP_HIP_LIB") or os.path.join(_HERE, "lib", "libidsp_hip.so")


class IdspError(RuntimeError):
    """A negative ``idsp_status`` returned by the C ABI."""

    def __init__(self, code: int, text: str):
        super().__init__(f"idsp status {code}: {text}")
        self.code = code
        self.text = text


_lib = None
_fn = None


def load():
    """Return ({name: ctypes function}, CDLL); raises if the engine is not built."""
    global _lib, _fn
    if _fn is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build the HIP engine first (`make lib` or "
                "`python -c 'import __graft_entry__ as g; g.build()'`). idsp_amd has no CPU fallback."
            )
        _lib = ctypes.CDLL(LIB_PATH)
        _fn = _abi.bind(_lib, "idsp_", with_stream=True, utils=True)
        if _fn["version"]() < _abi.ABI_VERSION:  # versions only add symbols (include/idsp_hip.h): a newer library is fine
            raise ImportError(f"libidsp_hip.so reports ABI version {_fn['version']()}, this package binds version {_abi.ABI_VERSION}: rebuild (`make lib`)")
    return _fn, _lib


def call(name: str, *args):
    """Invoke ``idsp_<name>`` and raise IdspError on a negative status."""
    fn, _ = load()
    rc = fn[name](*args)
    if isinstance(rc, int) and rc < 0:
        raise IdspError(rc, fn["last_error"]().decode(errors="replace"))
    return rc
