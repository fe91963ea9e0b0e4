"""Cic parity cases shared by the CPU (oracle vs spec) and GPU (HIP vs oracle) suites.
Test infrastructure only."""
from __future__ import annotations

import ctypes as C

import numpy as np

from idsp_amd import _abi

FM, LM = 0, 1


def samples(rng, dtype, n):
    info = np.iinfo(dtype)
    x = rng.integers(info.min, info.max, size=n, dtype=np.int64, endpoint=True).astype(dtype) if dtype == np.int32 else \
        rng.integers(info.min, info.max, size=n, dtype=np.int64, endpoint=True)
    if n >= 8:
        x[rng.integers(0, n, size=max(1, n // 8))] = rng.choice(np.array([info.min, info.max, 0, 1, -1], dtype=dtype))
    return x


def run(be, kind, dtype, cfg, st, x, lanes, frames, layout):
    R = cfg.rate + 1
    n_out = lanes * frames * (1 if kind == "dec" else R)
    return be.cfgcall(f"cic_{kind}_{'i64' if dtype == np.int64 else 'i32'}", cfg, st, x, (n_out,), dtype, lanes, frames, layout)


def state_words(be, cfg, dtype):
    return be.helper("cic_state_words", C.byref(cfg), 64 if dtype == np.int64 else 32)


def random_state(rng, words, lanes):
    return rng.integers(0, 1 << 32, size=(words, lanes), dtype=np.uint64).astype(np.uint32)


CONFIGS = [(1, 1, 0), (1, 1, 3), (2, 1, 1), (3, 1, 3), (3, 1, 15), (3, 2, 7), (4, 3, 4), (5, 4, 2), (6, 1, 9), (3, 1, 16), (2, 4, 11)]
