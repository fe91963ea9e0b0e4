"""CPU oracle package — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  ``load()`` returns a ctypes handle on
``oracle/_build/libidsp_oracle.so`` (built by ``make oracle``) whose
``idsp_ref_*`` functions are the host-pointer twins of the C ABI in
``include/idsp_hip.h``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
_LIB = None


def lib_path(native: bool = False) -> str:
    name = "libidsp_oracle_native.so" if native else "libidsp_oracle.so"
    return os.path.join(_HERE, "_build", name)


def build(native: bool = False) -> str:
    """Compile the C restatement with gcc (portable x86-64 by default;
    ``native=True`` adds -march=native for the timed CPU baseline)."""
    target = "oracle-native" if native else "oracle"
    subprocess.run(["make", "-s", target], cwd=_ROOT, check=True)
    return lib_path(native)


def load(native: bool = False) -> ctypes.CDLL:
    global _LIB
    if native:
        path = lib_path(True)
        if not os.path.exists(path):
            build(True)
        return ctypes.CDLL(path)
    if _LIB is None:
        path = lib_path()
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB
