"""Shared test plumbing: call the CPU oracle (numpy, host pointers) and the HIP
engine (torch device tensors, through the C ABI) with one signature.

Test infrastructure only."""
from __future__ import annotations

import ctypes as C
import functools

import numpy as np

from idsp_amd import _abi

FM, LM = _abi.FRAME_MAJOR, _abi.LANE_MAJOR


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return C.c_void_p(a.ctypes.data)
    return C.c_void_p(a.data_ptr())  # torch tensor


class Oracle:
    def __init__(self):
        import oracle

        self.lib = oracle.load()
        self.fn = _abi.bind(self.lib, "idsp_ref_", with_stream=False, utils=False)
        self.lib.idsp_ref_cossin.argtypes = [C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        self.lib.idsp_ref_cossin.restype = None
        self.lib.idsp_ref_cossin_table.restype = C.POINTER(C.c_uint32)
        self.lib.idsp_ref_quantize_f64.argtypes = [C.c_double, C.c_int]
        self.lib.idsp_ref_quantize_f64.restype = C.c_int32
        for n in ("idsp_ref_filter_lowpass", "idsp_ref_filter_highpass"):
            f = getattr(self.lib, n)
            f.argtypes = [C.c_double, C.c_double, C.c_double, C.c_void_p]
            f.restype = None
        self.lib.idsp_ref_biquad_i32_df1_mt.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                                        C.c_size_t, C.c_size_t, C.c_int, C.c_int]
        self.lib.idsp_ref_biquad_i32_df1_mt.restype = C.c_int

    # -- generic calls on numpy arrays (state: uint32 [words, lanes]) ---------
    def stream(self, name, cfg, n, state, x, y, lanes, frames, layout):
        return self.fn[name](C.cast(cfg, C.c_void_p) if cfg is not None else None, n, _ptr(state), _ptr(x), _ptr(y),
                             lanes, frames, layout)

    def cfgcall(self, name, cfg, state, x, y, lanes, frames, layout):
        return self.fn[name](C.byref(cfg), _ptr(state), _ptr(x), _ptr(y), lanes, frames, layout)

    def cossin(self, phase: int):
        c, s = C.c_int32(), C.c_int32()
        self.lib.idsp_ref_cossin(phase, C.byref(c), C.byref(s))
        return c.value, s.value

    def cossin_table(self):
        t = self.lib.idsp_ref_cossin_table()
        return [t[i] for i in range(128)]

    def lowpass_sos(self, f0, gain=1.0, q=2 ** -0.5, highpass=False):
        import math

        sos = (C.c_double * 6)()
        (self.lib.idsp_ref_filter_highpass if highpass else self.lib.idsp_ref_filter_lowpass)(math.tau * f0, gain, q, sos)
        return list(sos)


class Engine:
    """The product library, called exactly as a foreign host would."""

    def __init__(self):
        from idsp_amd._lib import load

        self.fn, self.lib = load()

    def err(self):
        return self.fn["last_error"]().decode()

    def last_kernel(self):
        return self.fn["last_kernel"]().decode()

    def stream(self, name, cfg, n, state, x, y, lanes, frames, layout, stream=None):
        return self.fn[name](C.cast(cfg, C.c_void_p) if cfg is not None else None, n, _ptr(state), _ptr(x), _ptr(y),
                             lanes, frames, layout, stream)

    def cfgcall(self, name, cfg, state, x, y, lanes, frames, layout, stream=None):
        return self.fn[name](C.byref(cfg), _ptr(state), _ptr(x), _ptr(y), lanes, frames, layout, stream)


@functools.lru_cache(maxsize=None)
def oracle() -> Oracle:
    return Oracle()


@functools.lru_cache(maxsize=None)
def engine() -> Engine:
    return Engine()


# ---------------------------------------------------------------- cfg builders
def biquad_i32(rows):
    arr = (_abi.BiquadI32 * max(len(rows), 1))()
    for a, (ba, frac) in zip(arr, rows):
        a.ba[:] = [int(v) for v in ba]
        a.frac = frac
    return arr


def biquad_clamp_i32(rows):
    arr = (_abi.BiquadClampI32 * max(len(rows), 1))()
    for a, (ba, frac, u, lo, hi) in zip(arr, rows):
        a.ba[:] = [int(v) for v in ba]
        a.frac, a.u, a.min, a.max = frac, u, lo, hi
    return arr


def biquad_f32(rows):
    arr = (_abi.BiquadF32 * max(len(rows), 1))()
    for a, ba in zip(arr, rows):
        a.ba[:] = [float(v) for v in ba]
    return arr


def biquad_clamp_f32(rows):
    arr = (_abi.BiquadClampF32 * max(len(rows), 1))()
    for a, (ba, u, lo, hi) in zip(arr, rows):
        a.ba[:] = [float(v) for v in ba]
        a.u, a.min, a.max = u, lo, hi
    return arr


def biquad_f64(rows):
    arr = (_abi.BiquadF64 * max(len(rows), 1))()
    for a, ba in zip(arr, rows):
        a.ba[:] = [float(v) for v in ba]
    return arr


def biquad_clamp_f64(rows):
    arr = (_abi.BiquadClampF64 * max(len(rows), 1))()
    for a, (ba, u, lo, hi) in zip(arr, rows):
        a.ba[:] = [float(v) for v in ba]
        a.u, a.min, a.max = u, lo, hi
    return arr


def ulp_diff_f64(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    ia, ib = a.view(np.int64), b.view(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFFFFFFFFFF), ia).astype(object)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFFFFFFFFFF), ib).astype(object)
    d = np.abs(ia - ib)
    return np.where(np.isnan(a) & np.isnan(b), 0, d)


def hbf_cfg(taps_list):
    cfg = _abi.HbfCascadeF32()
    cfg.stages = len(taps_list)
    for s, t in enumerate(taps_list):
        cfg.m[s] = len(t)
        for k, v in enumerate(t):
            cfg.taps[s][k] = v
    return cfg


def lockin_cfg(ks):
    cfg = _abi.LockinI32()
    cfg.order = len(ks[0])
    cfg.cascade = len(ks)
    for c, k in enumerate(ks):
        for j, v in enumerate(k):
            cfg.k[c][j] = int(v)
    return cfg


def ulp_diff_f32(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Distance in units in the last place between two float32 arrays
    (monotone integer mapping; NaN == NaN counts as 0)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia)
    ib = np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    d = np.abs(ia - ib)
    both_nan = np.isnan(a) & np.isnan(b)
    return np.where(both_nan, 0, d)
