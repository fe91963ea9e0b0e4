"""Normal-form and Wdf section cases shared by the CPU and GPU suites.  Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from idsp_amd import _abi
from tests import _harness as H

FM, LM = 0, 1
I32_MIN, I32_MAX = -(1 << 31), (1 << 31) - 1

# the architectures the reference's embedded bench instantiates (tests/embedded/src/bin/biquad.rs:126-164)
WDF_BENCH = [(0xAD, [-0.9, 0.9]), (0xAD, [-0.6, 0.7]), (0xAD, [-0.7, 0.6]), (0xA, [0.8]), (0x1, [0.0]),
             (0x1C, [-0.226119, 0.0]), (0x1D, [-0.602422, 0.0]), (0x1D, [-0.83932, 0.0]), (0x1D, [-0.950847, 0.0]),
             (0x1C, [-0.063978, 0.0]), (0x1C, [-0.423068, 0.0]), (0x1D, [-0.741327, 0.0]), (0x1D, [-0.905567, 0.0]),
             (0x1D, [-0.984721, 0.0])]


def wdf_section(be, m, g):
    out = _abi.Wdf()
    rc = be.helper("wdf_quantize", len(g), m, (C.c_double * len(g))(*g), C.byref(out))
    return rc, out


def wdf_array(secs):
    arr = (_abi.Wdf * max(len(secs), 1))()
    for d, s in zip(arr, secs):
        d.n, d.m = s.n, s.m
        d.a[:] = list(s.a)
    return arr


def random_wdf(rng, n_sections):
    """Arbitrary adaptor types (incl. Z and unknown nibbles) and arbitrary coefficient bits."""
    secs = []
    for _ in range(n_sections):
        s = _abi.Wdf()
        s.n = int(rng.integers(1, 9))
        nibs = rng.choice([0xA, 0xB, 0xE, 0x1, 0xC, 0xF, 0xD, 0x0, 0x7], size=8)
        s.m = int(sum(int(v) << (4 * i) for i, v in enumerate(nibs[:s.n])))
        s.a[:] = rng.integers(I32_MIN, 1, size=8, dtype=np.int64).tolist()
        secs.append(s)
    return secs


def normal_rows(rng, n, dtype, frac=None):
    """Stable conjugate pole pairs inside the unit circle + arbitrary zeros; some sections with arbitrary bits."""
    rows = []
    for _ in range(n):
        r, th = rng.uniform(0.3, 0.98), rng.uniform(0.05, 3.0)
        b = (rng.standard_normal(3) * 0.3).tolist()
        ba = b + [r * math.cos(th), r * math.sin(th)]
        if dtype == np.int32:
            if rng.integers(0, 4) == 0:
                q = rng.integers(I32_MIN, I32_MAX, size=5, dtype=np.int64, endpoint=True).tolist()
            else:
                q = [int(np.clip(round(v * (1 << frac)), I32_MIN, I32_MAX)) for v in ba]
            rows.append((q, frac))
        else:
            rows.append([float(dtype(v)) for v in ba])
    return rows


def normal_cfg(rows, dtype):
    return {np.int32: H.biquad_i32, np.float32: H.biquad_f32, np.float64: H.biquad_f64}[dtype](rows)
