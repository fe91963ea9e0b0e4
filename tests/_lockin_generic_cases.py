"""`Lockin<C>` beyond `[Lowpass<N>; K]` arms and the phase form (src/lockin.rs:16-39): case drivers shared by the oracle
tests (CPU) and the HIP parity tests.  `lib` is tests._harness.oracle() or .engine(); arrays are numpy (oracle) or device
tensors moved by the caller.  Test infrastructure only."""
import ctypes as C
import math

import numpy as np

from tests import _harness as H

FM, LM = H.FM, H.LM


def sections_i32(n, rng):
    """(ctypes array of n idsp_biquad_i32, [(ba, frac)]): stable Q30 lowpass sections with different corner frequencies,
    quantised by the oracle's `idsp_ref_biquad_i32_from_sos` (src/iir/biquad.rs:545-576)."""
    from idsp_amd import _abi

    o = H.oracle()
    arr = (_abi.BiquadI32 * n)()
    rows = []
    for k in range(n):
        sos = (C.c_double * 6)(*o.lowpass_sos(0.01 * (k + 1) + 0.002 * float(rng.random())))
        assert o.fn["biquad_i32_from_sos"](sos, 30, C.byref(arr[k])) == 0
        rows.append(([int(v) for v in arr[k].ba], 30))
    return arr, rows


def sections_f32(n, rng, f0=None):
    from idsp_amd import _abi

    o = H.oracle()
    arr = (_abi.BiquadF32 * n)()
    rows = []
    for k in range(n):
        sos = (C.c_double * 6)(*o.lowpass_sos(f0 if f0 is not None else 0.01 * (k + 1) + 0.002 * float(rng.random())))
        assert o.fn["biquad_f32_from_sos_f64"](sos, C.byref(arr[k])) == 0
        rows.append([np.float32(v) for v in arr[k].ba])
    return arr, rows


def call_lo(lib, name, cfg, n, state, x, lo, y, lanes, frames, layout, with_stream):
    """The `_lo` entries: (cfg[, n], state, x, lo, y, lanes, frames, layout[, stream])."""
    args = [C.cast(cfg, C.c_void_p) if n is not None else C.byref(cfg)]
    if n is not None:
        args.append(n)
    args += [H._ptr(state), H._ptr(x), H._ptr(lo), H._ptr(y), lanes, frames, layout]
    if with_stream:
        args.append(None)
    return lib.fn[name](*args)


def ddc_fixture(n=16384, lo_freq=0.173, phi=0.37):
    """examples/ddc_lockin.rs:58-62,91-98: tone cos(2 pi f i + phi) in f32 and the mixer's LO (cos, -sin) of a phase that
    advances by 2 pi f per sample (rem_euclid 2 pi), all in f32 as `QuadratureMix` computes them."""
    tau = np.float32(math.tau)
    x = np.cos(tau * np.float32(lo_freq) * np.arange(n, dtype=np.float32) + np.float32(phi)).astype(np.float32)
    ph = np.empty(n, np.float32)
    p = np.float32(0.0)
    step = tau * np.float32(lo_freq)
    for i in range(n):
        ph[i] = p
        p = np.float32(np.fmod(p + step, tau))
    lo = np.stack([np.cos(ph), -np.sin(ph)], axis=1).astype(np.float32)
    return x, lo, (0.5 * math.cos(phi), 0.5 * math.sin(phi))
