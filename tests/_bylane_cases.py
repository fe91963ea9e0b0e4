"""Per-lane coefficient (`ByLane<[C; N]>`, dsp-process/src/compose.rs:363-390) cases
shared by the CPU (oracle) and GPU (HIP) suites.  Test infrastructure only."""
from __future__ import annotations

import numpy as np

from tests import _harness as H

FM, LM = 0, 1
I32_MIN, I32_MAX = -(1 << 31), (1 << 31) - 1

# op (without `_bylane`), numpy dtype, state words per section, clamp
OPS = [
    ("biquad_i32_df1", np.int32, 4, False), ("biquad_i32_df1_clamp", np.int32, 4, True),
    ("biquad_i32_dither", np.int32, 5, False), ("biquad_i32_dither_clamp", np.int32, 5, True),
    ("biquad_i32_wide", np.int32, 6, False), ("biquad_i32_wide_clamp", np.int32, 6, True),
    ("biquad_f32_df1", np.float32, 4, False), ("biquad_f32_df1_clamp", np.float32, 4, True),
    ("biquad_f32_df2t", np.float32, 2, False), ("biquad_f32_df2t_clamp", np.float32, 2, True),
    ("biquad_f64_df1", np.float64, 8, False), ("biquad_f64_df1_clamp", np.float64, 8, True),
    ("biquad_f64_df2t", np.float64, 4, False), ("biquad_f64_df2t_clamp", np.float64, 4, True),
]


def coef_planes(rng, dtype, n, lanes, clamp, frac):
    """[n, CV, lanes]: a different (mostly stable) section per lane, some lanes with arbitrary bits."""
    cv = 8 if clamp else 5
    coef = np.zeros((n, cv, lanes), dtype=dtype)
    f0 = rng.uniform(0.002, 0.45, size=(n, lanes))
    q = rng.uniform(0.3, 4.0, size=(n, lanes))
    gain = rng.uniform(0.1, 1.0, size=(n, lanes))
    w0 = 2 * np.pi * f0
    alpha = 0.5 * np.sin(w0) / q
    b = gain * 0.5 * (1 - np.cos(w0))
    a0 = 1 + alpha
    ba = np.stack([b / a0, 2 * b / a0, b / a0, 2 * np.cos(w0) / a0, -(1 - alpha) / a0], 1)  # [n, 5, lanes]
    if dtype == np.int32:
        v = np.clip(np.round(ba * float(1 << frac)), I32_MIN, I32_MAX).astype(np.int64)
        wild = rng.integers(0, 5, size=(n, 1, lanes)) == 0
        rnd = rng.integers(I32_MIN, I32_MAX, size=v.shape, dtype=np.int64, endpoint=True)
        coef[:, :5] = np.where(wild, rnd, v).astype(np.int32)
        if clamp:
            lim = np.sort(rng.integers(I32_MIN, I32_MAX, size=(2, n, lanes), dtype=np.int64, endpoint=True), axis=0)
            open_ = rng.integers(0, 3, size=(n, lanes)) == 0
            coef[:, 5] = rng.integers(-1000, 1000, size=(n, lanes))
            coef[:, 6] = np.where(open_, I32_MIN, lim[0]).astype(np.int32)
            coef[:, 7] = np.where(open_, I32_MAX, lim[1]).astype(np.int32)
    else:
        coef[:, :5] = ba.astype(dtype)
        if clamp:
            lim = np.sort(rng.standard_normal(size=(2, n, lanes)) * 2, axis=0)
            open_ = rng.integers(0, 3, size=(n, lanes)) == 0
            coef[:, 5] = (rng.standard_normal(size=(n, lanes)) * 0.1).astype(dtype)
            coef[:, 6] = np.where(open_, -np.inf, lim[0]).astype(dtype)
            coef[:, 7] = np.where(open_, np.inf, lim[1]).astype(dtype)
    return coef


def samples(rng, dtype, size):
    if dtype == np.int32:
        x = rng.integers(I32_MIN, I32_MAX, size=size, dtype=np.int64, endpoint=True).astype(np.int32)
        x[rng.integers(0, size, size=max(1, size // 8))] = rng.choice(np.array([I32_MIN, I32_MAX, 0, 1, -1], np.int32))
        return x
    x = rng.standard_normal(size=size).astype(dtype)
    tiny = np.array([0.0, -0.0, 1e-41, -3e-42, 1e30, -1e30, 1.0]) if dtype == np.float32 else np.array([0.0, -0.0, 1e-310, -3e-320, 1e300, -1e300, 1.0])
    x[rng.integers(0, size, size=max(1, size // 10))] = rng.choice(tiny.astype(dtype))
    return x


def init_state(rng, dtype, words, lanes):
    if dtype == np.int32:
        return rng.integers(0, 1 << 32, size=(words, lanes), dtype=np.uint64).astype(np.uint32)
    if dtype == np.float32:
        return rng.standard_normal(size=(words, lanes)).astype(np.float32).view(np.uint32)
    v = rng.standard_normal(size=(words // 2, lanes)).view(np.uint32).reshape(words // 2, lanes, 2)
    return np.ascontiguousarray(v.transpose(0, 2, 1)).reshape(words, lanes)


def shared_cfg(op, dtype, clamp, coef_lane, frac):
    """ctypes cfg array for the shared-coefficient entry from one lane's [n, CV] coefficients."""
    n = coef_lane.shape[0]
    if dtype == np.int32:
        rows = [(coef_lane[k, :5].tolist(), frac) + ((int(coef_lane[k, 5]), int(coef_lane[k, 6]), int(coef_lane[k, 7])) if clamp else ())
                for k in range(n)]
        return H.biquad_clamp_i32(rows) if clamp else H.biquad_i32(rows)
    mk = {(np.float32, False): H.biquad_f32, (np.float32, True): H.biquad_clamp_f32,
          (np.float64, False): H.biquad_f64, (np.float64, True): H.biquad_clamp_f64}[(dtype, clamp)]
    if clamp:
        return mk([(coef_lane[k, :5].tolist(), float(coef_lane[k, 5]), float(coef_lane[k, 6]), float(coef_lane[k, 7])) for k in range(n)])
    return mk([coef_lane[k, :5].tolist() for k in range(n)])


def bits(a):
    return a.view({4: np.uint32, 8: np.uint64}[a.dtype.itemsize])
