"""ctypes view of the C ABI declared in ``include/idsp_hip.h``.

The structures and prototypes below are shared by the product library
(``libidsp_hip.so``, symbols ``idsp_*``); ``bind()`` takes the symbol prefix as an
argument so that the test suite can attach the same prototypes to its checker
library (host pointers, no ``stream`` argument).  Nothing
in here computes anything.
"""
from __future__ import annotations

import ctypes as C

ABI_VERSION = 4  # IDSP_ABI_VERSION of include/idsp_hip.h

IDSP_OK = 0
IDSP_EINVAL = -1
IDSP_EHIP = -2
IDSP_ENODEV = -3

FRAME_MAJOR = 0
LANE_MAJOR = 1

MAX_SECTIONS = 64
HBF_MAX_STAGES = 5
HBF_MAX_TAPS = 32
LOCKIN_MAX_CASCADE = 4
LOCKIN_MAX_SECTIONS = 4


class BiquadI32(C.Structure):
    _fields_ = [("ba", C.c_int32 * 5), ("frac", C.c_int32)]


class BiquadClampI32(C.Structure):
    _fields_ = [("ba", C.c_int32 * 5), ("frac", C.c_int32), ("u", C.c_int32), ("min", C.c_int32), ("max", C.c_int32)]


class BiquadF32(C.Structure):
    _fields_ = [("ba", C.c_float * 5)]


class BiquadClampF32(C.Structure):
    _fields_ = [("ba", C.c_float * 5), ("u", C.c_float), ("min", C.c_float), ("max", C.c_float)]


class BiquadF64(C.Structure):
    _fields_ = [("ba", C.c_double * 5)]


class BiquadClampF64(C.Structure):
    _fields_ = [("ba", C.c_double * 5), ("u", C.c_double), ("min", C.c_double), ("max", C.c_double)]


class HbfCascadeF32(C.Structure):
    _fields_ = [
        ("stages", C.c_int32),
        ("m", C.c_int32 * HBF_MAX_STAGES),
        ("taps", (C.c_float * HBF_MAX_TAPS) * HBF_MAX_STAGES),
    ]


class FirSymF32(C.Structure):
    _fields_ = [("kind", C.c_int32), ("m", C.c_int32), ("taps", C.c_float * HBF_MAX_TAPS)]


class HbfCascadeF64(C.Structure):
    _fields_ = [
        ("stages", C.c_int32),
        ("m", C.c_int32 * HBF_MAX_STAGES),
        ("taps", (C.c_double * HBF_MAX_TAPS) * HBF_MAX_STAGES),
    ]


class FirSymF64(C.Structure):
    _fields_ = [("kind", C.c_int32), ("m", C.c_int32), ("taps", C.c_double * HBF_MAX_TAPS)]


class LockinI32(C.Structure):
    _fields_ = [("order", C.c_int32), ("cascade", C.c_int32), ("k", (C.c_int32 * 2) * LOCKIN_MAX_CASCADE)]


WDF_MAX_ORDER = 8


class Wdf(C.Structure):
    _fields_ = [("n", C.c_int32), ("m", C.c_uint32), ("a", C.c_int32 * WDF_MAX_ORDER)]


class FmDisc(C.Structure):
    _fields_ = [("carrier", C.c_int32), ("deemph", BiquadI32)]


FM_DISC_STATE_WORDS = 7


class Cic(C.Structure):
    _fields_ = [("order", C.c_int32), ("comb_delay", C.c_int32), ("rate", C.c_uint32)]


class Filter(C.Structure):
    _fields_ = [("frequency", C.c_double), ("gain", C.c_double), ("shelf", C.c_double), ("shape", C.c_double),
                ("shape_kind", C.c_int32), ("f32", C.c_int32)]


class PidBuilder(C.Structure):
    _fields_ = [("order", C.c_int32), ("f32", C.c_int32), ("gain", C.c_double * 5), ("limit", C.c_double * 5)]


class Units(C.Structure):
    _fields_ = [("t", C.c_double), ("x", C.c_double), ("y", C.c_double)]


class Pid(C.Structure):
    _fields_ = [("builder", PidBuilder), ("setpoint", C.c_double), ("min", C.c_double), ("max", C.c_double)]


class BaConfig(C.Structure):
    _fields_ = [("ba", C.c_double * 6), ("offset", C.c_double), ("min", C.c_double), ("max", C.c_double),
                ("f32", C.c_int32)]


class FilterConfig(C.Structure):
    _fields_ = [("typ", C.c_int32), ("shape_kind", C.c_int32), ("frequency", C.c_double), ("gain_db", C.c_double),
                ("shelf_db", C.c_double), ("shape", C.c_double), ("offset", C.c_double), ("min", C.c_double),
                ("max", C.c_double), ("f32", C.c_int32)]


# idsp_status of the builder validation errors (`iir::Error`, src/iir/error.rs:5-16)
IDSP_ENONFINITE, IDSP_ENONPOSITIVE, IDSP_EOUTOFRANGE, IDSP_EINVERTED, IDSP_ESIGN = -10, -11, -12, -13, -14
FILTER_TYPES = ("lowpass", "highpass", "bandpass", "allpass", "notch", "peaking", "lowshelf", "highshelf", "iho")
SHAPE_Q, SHAPE_BANDWIDTH, SHAPE_SLOPE = 0, 1, 2

_P = C.c_void_p
_SZ = C.c_size_t
_I = C.c_int

# name -> (restype, argtypes) for the processing entry points; `stream` (the
# last void*) is dropped when binding the checker library.
_STREAM_SIG = [_P, _SZ, _P, _P, _P, _SZ, _SZ, _I, _P]  # cfg, n, state, x, y, lanes, frames, layout, stream
_CFG_SIG = [_P, _P, _P, _P, _SZ, _SZ, _I, _P]          # cfg, state, x, y, lanes, frames, layout, stream
_STREAM_LO_SIG = [_P, _SZ, _P, _P, _P, _P, _SZ, _SZ, _I, _P]  # sections, n, state, x, lo, y, lanes, frames, layout, stream
_CFG_LO_SIG = [_P, _P, _P, _P, _P, _SZ, _SZ, _I, _P]          # cfg, state, x, lo, y, lanes, frames, layout, stream

_BYLANE_I32_SIG = [_P, _I, _SZ, _P, _P, _P, _SZ, _SZ, _I, _P]  # coef, frac, n, state, x, y, lanes, frames, layout, stream
_BYLANE_F_SIG = [_P, _SZ, _P, _P, _P, _SZ, _SZ, _I, _P]        # coef, n, state, x, y, lanes, frames, layout, stream

PROCESSING = {
    "biquad_i32_df1": _STREAM_SIG,
    "biquad_i32_df1_clamp": _STREAM_SIG,
    "biquad_i32_dither": _STREAM_SIG,
    "biquad_i32_dither_clamp": _STREAM_SIG,
    "biquad_i32_wide": _STREAM_SIG,
    "biquad_i32_wide_clamp": _STREAM_SIG,
    "cascade_i32_df1": _STREAM_SIG,
    "biquad_f32_df1": _STREAM_SIG,
    "biquad_f32_df1_clamp": _STREAM_SIG,
    "biquad_f32_df2t": _STREAM_SIG,
    "biquad_f32_df2t_clamp": _STREAM_SIG,
    "cascade_f32_df1": _STREAM_SIG,
    "biquad_f64_df1": _STREAM_SIG,
    "biquad_f64_df1_clamp": _STREAM_SIG,
    "biquad_f64_df2t": _STREAM_SIG,
    "biquad_f64_df2t_clamp": _STREAM_SIG,
    "cascade_f64_df1": _STREAM_SIG,
    "biquad_i32_df1_bylane": _BYLANE_I32_SIG,
    "biquad_i32_df1_clamp_bylane": _BYLANE_I32_SIG,
    "biquad_i32_dither_bylane": _BYLANE_I32_SIG,
    "biquad_i32_dither_clamp_bylane": _BYLANE_I32_SIG,
    "biquad_i32_wide_bylane": _BYLANE_I32_SIG,
    "biquad_i32_wide_clamp_bylane": _BYLANE_I32_SIG,
    "biquad_f32_df1_bylane": _BYLANE_F_SIG,
    "biquad_f32_df1_clamp_bylane": _BYLANE_F_SIG,
    "biquad_f32_df2t_bylane": _BYLANE_F_SIG,
    "biquad_f32_df2t_clamp_bylane": _BYLANE_F_SIG,
    "biquad_f64_df1_bylane": _BYLANE_F_SIG,
    "biquad_f64_df1_clamp_bylane": _BYLANE_F_SIG,
    "biquad_f64_df2t_bylane": _BYLANE_F_SIG,
    "biquad_f64_df2t_clamp_bylane": _BYLANE_F_SIG,
    "hbf_dec_f32": _CFG_SIG,
    "hbf_int_f32": _CFG_SIG,
    "fir_sym_f32_process": _CFG_SIG,
    "hbf_dec_f64": _CFG_SIG,
    "hbf_int_f64": _CFG_SIG,
    "fir_sym_f64_process": _CFG_SIG,
    "normal_i32_df1": _STREAM_SIG,
    "normal_f32_df1": _STREAM_SIG,
    "normal_f64_df1": _STREAM_SIG,
    "wdf_i32": _STREAM_SIG,
    "cic_dec_i32": _CFG_SIG,
    "cic_dec_i64": _CFG_SIG,
    "cic_int_i32": _CFG_SIG,
    "cic_int_i64": _CFG_SIG,
    "cossin_i32": [_P, _P, _SZ, _P],
    "atan2_i32": [_P, _P, _SZ, _P],
    "dds_i32": [_P, _P, _SZ, _SZ, _I, _P],
    "lockin_i32_process": _CFG_SIG,
    "lockin_i32_arg": _CFG_SIG,
    "lockin_i32_norm_sqr": _CFG_SIG,
    "lockin_i32_biquad_process": _STREAM_SIG,
    "lockin_i32_lo_process": _CFG_LO_SIG,
    "lockin_i32_biquad_lo_process": _STREAM_LO_SIG,
    "lockin_f32_biquad_lo_process": _STREAM_LO_SIG,
    "lowpass_i32": _CFG_SIG,
    "fm_disc_i32": _CFG_SIG,
}

# `<entry>_pitch` twins (product only: explicit x / y row pitches after the x and y pointers)
def _pitched(sig):
    sig = list(sig)
    i = len(sig) - 6  # ..., x, y, lanes, frames, layout, stream
    return sig[:i + 1] + [_SZ] + [sig[i + 1]] + [_SZ] + sig[i + 2:]


PITCHED = {name + "_pitch": _pitched(sig) for name, sig in PROCESSING.items()
           if name.startswith(("biquad_", "cascade_"))}

# host-side helpers present in both libraries (same signature)
HELPERS = {
    "biquad_i32_from_sos": (_I, [_P, _I, _P]),
    "biquad_f32_from_sos": (_I, [_P, _P]),
    "biquad_f32_from_sos_f64": (_I, [_P, _P]),
    "biquad_f64_from_sos": (_I, [_P, _P]),
    "hbf_dec_cascade": (_I, [_I, _I, _P]),
    "hbf_int_cascade": (_I, [_I, _I, _P]),
    "hbf_dec_response_length": (_I, [_P]),
    "hbf_int_response_length": (_I, [_P]),
    "hbf_dec_state_words": (_SZ, [_P]),
    "hbf_int_state_words": (_SZ, [_P]),
    "lockin_state_words": (_SZ, [_P]),
    "lockin_biquad_state_words": (_SZ, [_SZ, _I]),
    "fir_sym_state_words": (_SZ, [_P]),
    "hbf_dec_cascade_f64": (_I, [_I, _I, _P]),
    "hbf_int_cascade_f64": (_I, [_I, _I, _P]),
    "hbf_dec_state_words_f64": (_SZ, [_P]),
    "hbf_int_state_words_f64": (_SZ, [_P]),
    "fir_sym_state_words_f64": (_SZ, [_P]),
    "normal_from_sos": (_I, [_P, _P]),
    "wdf_quantize": (_I, [_I, C.c_uint32, _P, _P]),
    "wdf_state_words": (_SZ, [_P, _SZ]),
    "cic_gain": (C.c_int64, [_P]),
    "cic_gain_log2": (_I, [_P]),
    "cic_response_length": (_SZ, [_P]),
    "cic_state_words": (_SZ, [_P, _I]),
}

_D = C.c_double
# host-side coefficient front-end (product only; checked against oracle/spec.py)
FRONTEND = {
    "filter_build": (_I, [_P, _I, _I, _P]),
    "pid_build_i32": (_I, [_P, _D, _I, _I, _P]),
    "pid_build_f32": (_I, [_P, _D, _I, _P]),
    "pid_build_f64": (_I, [_P, _D, _I, _P]),
    "pid_build_clamp_i32": (_I, [_P, _P, _I, _I, _P]),
    "pid_build_clamp_f32": (_I, [_P, _P, _I, _P]),
    "pid_build_clamp_f64": (_I, [_P, _P, _I, _P]),
    "config_ba_build_i32": (_I, [_P, _P, _I, _I, _P]),
    "config_ba_build_f32": (_I, [_P, _P, _I, _P]),
    "config_ba_build_f64": (_I, [_P, _P, _I, _P]),
    "config_filter_build_i32": (_I, [_P, _P, _I, _I, _P]),
    "config_filter_build_f32": (_I, [_P, _P, _I, _P]),
    "config_filter_build_f64": (_I, [_P, _P, _I, _P]),
}

# product-only utilities
UTILS = {
    "version": (_I, []),
    "last_error": (C.c_char_p, []),
    "last_kernel": (C.c_char_p, []),
    "device_count": (_I, []),
    "device_set": (_I, [_I]),
    "device_alloc": (_I, [C.POINTER(_P), _SZ]),
    "device_free": (_I, [_P]),
    "device_memset": (_I, [_P, _I, _SZ, _P]),
    "device_h2d": (_I, [_P, _P, _SZ, _P]),
    "device_d2h": (_I, [_P, _P, _SZ, _P]),
    "device_copy": (_I, [_P, _P, _SZ, _P]),
    "stream_sync": (_I, [_P]),
    "device_sync": (_I, []),
    # single-process lane split over several devices
    "multi_create": (_I, [_P, _I, C.POINTER(_P)]),
    "multi_destroy": (_I, [_P]),
    "multi_size": (_I, [_P]),
    "multi_device": (_I, [_P, _I]),
    "multi_stream": (_P, [_P, _I]),
    "multi_shard": (_I, [_P, _SZ, _I, C.POINTER(_SZ), C.POINTER(_SZ)]),
    "multi_for_each": (_I, [_P, _SZ, _P, _P]),
    "multi_sync": (_I, [_P]),
    "multi_last_block": (_I, []),
    "multi_alloc": (_I, [_P, _SZ, _SZ, C.POINTER(_P)]),
    "multi_free": (_I, [_P, C.POINTER(_P)]),
    "multi_copy": (_I, [_P, _SZ, _SZ, C.POINTER(_P), _P, _I]),
    "multi_biquad_i32_df1": (_I, [_P, _P, _SZ, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), _SZ, _SZ, _I]),
    "multi_biquad_f32_df2t": (_I, [_P, _P, _SZ, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), _SZ, _SZ, _I]),
}

SHARD_FN = C.CFUNCTYPE(_I, _P, _I, _SZ, _SZ, _P)  # idsp_shard_fn


def exported_names() -> list:
    """Every symbol include/idsp_hip.h declares (without the ``idsp_`` prefix)."""
    return sorted(list(PROCESSING) + list(PITCHED) + list(HELPERS) + list(UTILS) + list(FRONTEND))


def bind(lib: C.CDLL, prefix: str, *, with_stream: bool, utils: bool):
    """Attach prototypes to `lib`; returns {short name: function}."""
    out = {}
    for name, sig in PROCESSING.items():
        fn = getattr(lib, prefix + name)
        fn.restype = _I
        fn.argtypes = list(sig) if with_stream else list(sig[:-1])
        out[name] = fn
    for name, (res, args) in HELPERS.items():
        fn = getattr(lib, prefix + name)
        fn.restype = res
        fn.argtypes = list(args)
        out[name] = fn
    if utils:
        for name, sig in PITCHED.items():
            fn = getattr(lib, prefix + name)
            fn.restype = _I
            fn.argtypes = list(sig)
            out[name] = fn
        for name, (res, args) in list(UTILS.items()) + list(FRONTEND.items()):
            fn = getattr(lib, prefix + name)
            fn.restype = res
            fn.argtypes = list(args)
            out[name] = fn
    return out
