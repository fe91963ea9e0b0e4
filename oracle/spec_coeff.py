"""TEST INFRASTRUCTURE — independent Python restatement of the reference's host-side
coefficient front-end, used only to check `libidsp_hip.so`'s `idsp_filter_build`,
`idsp_pid_build_*`, `idsp_pid_build_clamp_*` and `idsp_config_*_build_*`.

Follows src/iir/coefficients.rs:240-495 (`Filter`), src/iir/pid.rs:195-313,497-567
(`Builder`, `Pid`), src/iir/config.rs:309-427 (`BiquadConfig::{build, try_build}`),
src/iir/biquad.rs:224-226,253-255,545-588 and the float -> Q conversion of
dsp-fixedpoint/src/num_traits_impl.rs:32-46.

The builder float type T is `np.float64` or `np.float32`; every arithmetic operation is
carried out on numpy scalars of that type so it rounds where the reference's generic code
rounds.  Transcendentals go through `math.*` (the platform libm, as Rust's std does for
f64); for T = f32 they are the f64 libm result rounded to f32, which can differ from
`sinf`/`powf` by an ULP — tests allow for that on f32 trigonometric paths only.

Pinned by the reference's own tests: coefficients.rs:289-300,316-326 (Q30 lowpass and
highpass doctests), :695-845 (transfer-function masks of all nine types before and
after Q30 quantisation), pid.rs:104-112,144-158,251-255,574-590,606-619.
"""
from __future__ import annotations

import math

import numpy as np

I32_MIN, I32_MAX = -(1 << 31), (1 << 31) - 1
LOWPASS, HIGHPASS, BANDPASS, ALLPASS, NOTCH, PEAKING, LOWSHELF, HIGHSHELF, IHO = range(9)
SHAPE_Q, SHAPE_BANDWIDTH, SHAPE_SLOPE = 0, 1, 2
ORDER_P, ORDER_I, ORDER_I2 = 2, 1, 0


class BuildError(Exception):
    """`iir::Error` (src/iir/error.rs:5-30)."""

    def __init__(self, kind: str, name: str):
        text = {"NonFinite": "parameter `%s` must be finite", "NonPositive": "parameter `%s` must be positive",
                "OutOfRange": "parameter `%s` is out of range", "InvertedRange": "range `%s` is inverted",
                "SignMismatch": "parameter `%s` has incompatible sign"}[kind] % name
        super().__init__(text)
        self.kind, self.name, self.text = kind, name, text


def _fn(T, f, *a):
    with np.errstate(all="ignore"):
        try:
            return T(f(*[float(v) for v in a]))
        except (ValueError, OverflowError):  # math domain/range -> IEEE result
            return T(getattr(np, f.__name__ if f.__name__ != "pow" else "power")(*[np.float64(v) for v in a]))


def as_i32(v) -> int:
    """Rust `as i32` from a float."""
    v = float(v)
    if math.isnan(v):
        return 0
    if v >= 2147483648.0:
        return I32_MAX
    if v <= -2147483648.0:
        return I32_MIN
    return int(v)


def round_half_away(v: float) -> float:
    if math.isnan(v) or math.isinf(v):
        return v
    r = float(math.trunc(v))
    if abs(v - r) >= 0.5:
        r += math.copysign(1.0, v)
    return r


def to_q(T, v, frac: int) -> int:
    with np.errstate(all="ignore"):
        s = T(v) * T(float(1 << frac))
    return as_i32(round_half_away(float(s)))


class Out:
    """C / Y adaptor: kind 'i32' (C = Q32<frac>, Y = i32), 'f32', 'f64'."""

    def __init__(self, kind: str, frac: int = 0):
        self.kind, self.frac = kind, frac
        self.F = {"f32": np.float32, "f64": np.float64}.get(kind)

    def coef(self, T, v):
        return to_q(T, v, self.frac) if self.kind == "i32" else self.F(v)

    def samp(self, T, v):
        return as_i32(v) if self.kind == "i32" else self.F(v)

    def add(self, a, b):
        if self.kind == "i32":
            return ((a + b + (1 << 31)) & 0xFFFFFFFF) - (1 << 31)
        with np.errstate(all="ignore"):
            return a + b

    def sub(self, a, b):
        if self.kind == "i32":
            return ((a - b + (1 << 31)) & 0xFFFFFFFF) - (1 << 31)
        with np.errstate(all="ignore"):
            return a - b

    def mul(self, y, c):
        if self.kind == "i32":  # dsp-fixedpoint/src/lib.rs:449-456
            return (((y * c) >> self.frac) + (1 << 31) & 0xFFFFFFFF) - (1 << 31)
        with np.errstate(all="ignore"):
            return y * c

    def ymin(self):
        return I32_MIN if self.kind == "i32" else self.F(-math.inf)

    def ymax(self):
        return I32_MAX if self.kind == "i32" else self.F(math.inf)


# ---------------------------------------------------------------------------- Filter
def filter_validate(T, frequency, gain, shelf, shape_kind, shape):
    """src/iir/coefficients.rs:240-263"""
    frequency, gain, shelf, shape = T(frequency), T(gain), T(shelf), T(shape)
    if not np.isfinite(frequency):
        raise BuildError("NonFinite", "frequency")
    if frequency < T(0) or frequency > T(math.pi):
        raise BuildError("OutOfRange", "frequency")
    if not np.isfinite(gain) or gain <= T(0):
        raise BuildError("NonPositive", "gain")
    if not np.isfinite(shelf) or shelf <= T(0):
        raise BuildError("NonPositive", "shelf")
    name = {SHAPE_Q: "q", SHAPE_BANDWIDTH: "bandwidth", SHAPE_SLOPE: "slope"}[shape_kind]
    if not np.isfinite(shape):
        raise BuildError("NonFinite", name)
    if shape_kind != SHAPE_BANDWIDTH and shape <= T(0):
        raise BuildError("NonPositive", name)


def filter_build(T, typ, frequency, gain, shelf, shape_kind, shape):
    """src/iir/coefficients.rs:266-495 -> [b0, b1, b2, a0, a1, a2] as T scalars."""
    w, g, sh, shape = T(frequency), T(gain), T(shelf), T(shape)
    one, two, half = T(1), T(2), T(0.5)
    with np.errstate(all="ignore"):
        if shape_kind == SHAPE_Q:
            qi = one / shape
        elif shape_kind == SHAPE_BANDWIDTH:
            qi = two * _fn(T, math.sinh, T(math.log(2.0)) / two * shape * w / _fn(T, math.sin, w))
        else:
            qi = _fn(T, math.sqrt, (sh + one / sh) * (one / shape - one) + two)
        fsin, fcos = _fn(T, math.sin, w), _fn(T, math.cos, w)
        alpha = half * fsin * qi
        a = [one + alpha, T(-2) * fcos, one - alpha]
        if typ == LOWPASS:
            v = g * half * (one - fcos)
            b = [v, two * v, v]
        elif typ == HIGHPASS:
            v = g * half * (one + fcos)
            b = [v, T(-2) * v, v]
        elif typ == BANDPASS:
            v = g * alpha
            b = [v, T(0), -v]
        elif typ == NOTCH:
            f2 = T(-2) * fcos
            b = [g, f2 * g, g]
            a[1] = f2
        elif typ == ALLPASS:
            f2 = T(-2) * fcos
            b = [(one - alpha) * g, f2 * g, (one + alpha) * g]
            a[1] = f2
        elif typ == PEAKING:
            s, f2 = _fn(T, math.sqrt, sh), T(-2) * fcos
            b = [(one + alpha * s) * g, f2 * g, (one - alpha * s) * g]
            a = [one + alpha / s, f2, one - alpha / s]
        elif typ in (LOWSHELF, HIGHSHELF):
            s = _fn(T, math.sqrt, sh)
            tsa = two * _fn(T, math.sqrt, s) * alpha
            sp1, sm1 = s + one, s - one
            if typ == LOWSHELF:
                b = [s * g * (sp1 - sm1 * fcos + tsa), two * s * g * (sm1 - sp1 * fcos), s * g * (sp1 - sm1 * fcos - tsa)]
                a = [sp1 + sm1 * fcos + tsa, T(-2) * (sm1 + sp1 * fcos), sp1 + sm1 * fcos - tsa]
            else:
                b = [s * g * (sp1 + sm1 * fcos + tsa), T(-2) * s * g * (sm1 + sp1 * fcos), s * g * (sp1 + sm1 * fcos - tsa)]
                a = [sp1 - sm1 * fcos + tsa, two * (sm1 - sp1 * fcos), sp1 - sm1 * fcos - tsa]
        else:  # IHO
            hs = half * _fn(T, math.sin, w)
            av = (one + fcos) / (two * sh)
            b = [g * (one + alpha), T(-2) * g * fcos, g * (one - alpha)]
            a = [av + hs, T(-2) * av, av - hs]
    return b + a


def normalize(T, sos, out: Out):
    """src/iir/biquad.rs:545-576"""
    with np.errstate(all="ignore"):
        a0 = T(1) / sos[3]
        vals = [sos[0] * a0, sos[1] * a0, sos[2] * a0, -sos[4] * a0, -sos[5] * a0]
    return [out.coef(T, v) for v in vals]


# ---------------------------------------------------------------------------- pid::Builder
def powi(T, a, b: int):
    """llvm.powi / compiler-rt __powi?f2"""
    recip, r, a = b < 0, T(1), T(a)
    b = abs(b)  # C's b /= 2 truncates toward zero: identical bit pattern walk on |b|
    with np.errstate(all="ignore"):
        while True:
            if b & 1:
                r = r * a
            b //= 2
            if b == 0:
                break
            a = a * a
        return T(1) / r if recip else r


def builder_validate(T, order, gain, limit, period):
    """src/iir/pid.rs:195-222"""
    period = T(period)
    if not np.isfinite(period):
        raise BuildError("NonFinite", "period")
    if period <= T(0):
        raise BuildError("NonPositive", "period")
    for name, values in (("gain", gain), ("limit", limit)):
        for v in values:
            if np.isnan(T(v)):
                raise BuildError("NonFinite", name)
    for action in (0, 1, 3, 4):
        g, l = T(gain[action]), T(limit[action])
        if np.isfinite(l):
            if l == T(0):
                raise BuildError("NonPositive", "limit")
            if g != T(0) and np.signbit(g) != np.signbit(l):
                raise BuildError("SignMismatch", "gain/limit")


def builder_build(T, order, gain, limit, period, out: Out):
    """src/iir/pid.rs:256-313 -> [b0, b1, b2, a1, a2] in C."""
    period = T(period)
    with np.errstate(all="ignore"):
        z = powi(T, period, -order)
        gl = [[T(0), T(0)] for _ in range(3)]
        for j in (2, 1, 0):
            i = order + j
            gl[j][0] = T(gain[i]) * z
            gl[j][1] = T(1) if i == 2 else gl[j][0] / T(limit[i])
            z = z * period
        a0i = T(1) / (gl[0][1] + gl[1][1] + gl[2][1])
        kernels = [[1, 0, 0], [1, -1, 0], [1, -2, 1]]
        zero = out.coef(T, T(0))
        ba = [[zero, zero] for _ in range(3)]
        for j in range(3):
            g0, g1 = out.coef(T, gl[j][0] * a0i), out.coef(T, gl[j][1] * a0i)
            for m in range(3):
                k = kernels[j][m]
                for _ in range(abs(k)):
                    if k > 0:
                        ba[m][0], ba[m][1] = out.add(ba[m][0], g0), out.sub(ba[m][1], g1)
                    else:
                        ba[m][0], ba[m][1] = out.sub(ba[m][0], g0), out.add(ba[m][1], g1)
    return [ba[0][0], ba[1][0], ba[2][0], ba[1][1], ba[2][1]]


# ---------------------------------------------------------------------------- Pid / BiquadConfig
def pid_validate(T, order, gain, limit, setpoint, mn, mx, units):
    """src/iir/pid.rs:497-518"""
    if T(mn) > T(mx):
        raise BuildError("InvertedRange", "output_limits")
    for name, v in zip("txy", units):
        if not np.isfinite(T(v)):
            raise BuildError("NonFinite", name)
        if T(v) <= T(0):
            raise BuildError("NonPositive", name)
    builder_validate(T, order, gain, limit, units[0])


def pid_build_clamp(T, order, gain, limit, setpoint, mn, mx, units, out: Out):
    """src/iir/pid.rs:533-567 -> (ba[5], u, min, max)"""
    t, x, y = (T(v) for v in units)
    with np.errstate(all="ignore"):
        yu = T(1) / y
        yx = x * yu
        p = T(gain[2])
        g = [yx * np.copysign(T(v), p) for v in gain]
        l = [yx * np.copysign(T(math.inf) if np.isnan(T(v)) else T(v), p) for v in limit]
        ba = builder_build(T, order, g, l, t, out)
        i = out.samp(T, -T(setpoint) * (T(1) / x))
        fg = out.add(out.add(ba[0], ba[1]), ba[2])
        return ba, out.mul(i, fg), out.samp(T, T(mn) * yu), out.samp(T, T(mx) * yu)


def check_offset_limits(T, offset, mn, mx):
    """src/iir/config.rs:309-326"""
    if not np.isfinite(T(offset)):
        raise BuildError("NonFinite", "offset")
    if np.isnan(T(mn)) or np.isnan(T(mx)):
        raise BuildError("NonFinite", "output_limits")
    if T(mn) > T(mx):
        raise BuildError("InvertedRange", "output_limits")


def check_units(T, units, check_t: bool):
    """src/iir/config.rs:328-344"""
    for name, v in (("x", units[1]), ("y", units[2])) + ((("t", units[0]),) if check_t else ()):
        if not np.isfinite(T(v)):
            raise BuildError("NonFinite", name)
        if T(v) <= T(0):
            raise BuildError("NonPositive", name)


def _finish(T, sos, units, offset, mn, mx, out: Out):
    with np.errstate(all="ignore"):
        yu = T(1) / T(units[2])
        yx = T(units[1]) * yu
        sos = [v * yx for v in sos[:3]] + list(sos[3:])
        return normalize(T, sos, out), out.samp(T, T(offset) * yu), out.samp(T, T(mn) * yu), out.samp(T, T(mx) * yu)


def config_ba_build(T, ba6, offset, mn, mx, units, out: Out, validate=False):
    """`BiquadConfig::Ba` (src/iir/config.rs:359-367, 389-407)"""
    sos = [T(v) for v in ba6]
    if validate:
        check_units(T, units, False)
        check_offset_limits(T, offset, mn, mx)
        if not all(np.isfinite(v) for v in sos):
            raise BuildError("NonFinite", "ba")
    return _finish(T, sos, units, offset, mn, mx, out)


def config_filter_build(T, typ, frequency, gain_db, shelf_db, shape_kind, shape, offset, mn, mx, units, out: Out,
                        validate=False):
    """`BiquadConfig::Filter` (src/iir/config.rs:372-385, 409-427)"""
    if validate:
        check_units(T, units, True)
        check_offset_limits(T, offset, mn, mx)
    with np.errstate(all="ignore"):
        gain = _fn(T, math.pow, T(10), T(gain_db) / T(20))
        w0 = T(math.tau) * (T(frequency) * T(units[0]))
        shelf = _fn(T, math.pow, T(10), T(shelf_db) / T(20))
    if validate:
        filter_validate(T, w0, gain, shelf, shape_kind, shape)
    sos = filter_build(T, typ, w0, gain, shelf, shape_kind, shape)
    return _finish(T, sos, units, offset, mn, mx, out)


def freqz(b, a, f: float) -> complex:
    """src/iir/response.rs: H(e^{j 2 pi f}) of [[b],[a]] in cookbook sign convention."""
    z = complex(math.cos(-2 * math.pi * f), math.sin(-2 * math.pi * f))
    num = b[0] + b[1] * z + b[2] * z * z
    den = a[0] + a[1] * z + a[2] * z * z
    return num / den
