"""Executable spec model — CPU ORACLE, TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A second, independent restatement of the reference hot path (quartiq/idsp
0.22.0) in pure Python: exact ``int`` arithmetic with explicit two's-complement
wrapping for the fixed-point paths and ``numpy.float32`` scalars (every
operation individually rounded, no FMA) for the float paths.  It is slow and
only meant for small cases; it exists so that the components for which the
reference holds no asserted value ("parity unpinned", see idsp_oracle.h) are
pinned by the agreement of two restatements written from the cited lines
independently (this file is structured like the reference: one state object
and one ``process`` per type; the C oracle is structured around flat word
records).

Only ``tests/`` may import this module.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np

f32 = np.float32
I32_MIN, I32_MAX = -(1 << 31), (1 << 31) - 1


# ----------------------------------------------------------------------------
# integer helpers: Rust release-mode semantics
# ----------------------------------------------------------------------------
def wrap(v: int, bits: int) -> int:
    """Two's-complement wrap of a Python int to a signed `bits`-wide value."""
    m = 1 << bits
    v &= m - 1
    return v - m if v >> (bits - 1) else v


def i32(v: int) -> int:
    return wrap(v, 32)


def i64(v: int) -> int:
    return wrap(v, 64)


def u32(v: int) -> int:
    return v & 0xFFFFFFFF


# ----------------------------------------------------------------------------
# dsp-fixedpoint: float -> Q (num_traits_impl.rs:32-46), biquad.rs:545-576
# ----------------------------------------------------------------------------
def round_half_away(v: float) -> float:
    return math.floor(v + 0.5) if v >= 0 else -math.floor(-v + 0.5)


def quantize(v: float, frac: int) -> int:
    """`(v * 2^F).round() as i32`: half away from zero, saturating, NaN -> 0."""
    if math.isnan(v):
        return 0
    s = v * (2.0 ** frac)
    if math.isinf(s):
        return I32_MAX if s > 0 else I32_MIN
    r = round_half_away(s)
    return max(I32_MIN, min(I32_MAX, int(r)))


def ba_from_sos_f64(sos: Sequence[float]) -> List[float]:
    """`From<[[f64;3];2]>` (src/iir/biquad.rs:545-566)."""
    a0 = 1.0 / sos[3]
    return [sos[0] * a0, sos[1] * a0, sos[2] * a0, -sos[4] * a0, -sos[5] * a0]


def biquad_i32_from_sos(sos: Sequence[float], frac: int) -> List[int]:
    return [quantize(v, frac) for v in ba_from_sos_f64(sos)]


def filter_lowpass(f0: float, gain: float = 1.0, q: float = 1 / math.sqrt(2.0)) -> List[float]:
    """coefficients::Filter lowpass in f64 (src/iir/coefficients.rs:259-283);
    `critical_frequency(f0)` sets w0 = TAU*f0 (:132-134); default Shape::Q(1/sqrt2) (:19-22)."""
    w0 = math.tau * f0
    fsin, fcos = math.sin(w0), math.cos(w0)
    alpha = 0.5 * fsin * (1.0 / q)
    b = gain * 0.5 * (1.0 - fcos)
    return [b, 2.0 * b, b, 1.0 + alpha, -2.0 * fcos, 1.0 - alpha]


def filter_highpass(f0: float, gain: float = 1.0, q: float = 1 / math.sqrt(2.0)) -> List[float]:
    """src/iir/coefficients.rs:302-335."""
    w0 = math.tau * f0
    fsin, fcos = math.sin(w0), math.cos(w0)
    alpha = 0.5 * fsin * (1.0 / q)
    b = gain * 0.5 * (1.0 + fcos)
    return [b, -2.0 * b, b, 1.0 + alpha, -2.0 * fcos, 1.0 - alpha]


# ----------------------------------------------------------------------------
# iir::Biquad, fixed point (src/iir/biquad.rs)
# ----------------------------------------------------------------------------
class DirectForm1:
    """`DirectForm1<T>` (biquad.rs:260-269,319): x = [x0, x1], y = [[y0, y1]]."""

    def __init__(self, zero=0):
        self.x = [zero, zero]
        self.y = [zero, zero]


def clamp(v, lo, hi):
    """num_traits::clamp."""
    if v < lo:
        return lo
    if v > hi:
        return hi
    return v


def biquad_i32_df1(ba: Sequence[int], frac: int, st: DirectForm1, x0: int) -> int:
    """biquad.rs:366-383 with C = Q32<F>."""
    acc = i64(ba[0] * x0)
    acc = i64(acc + ba[1] * st.x[0])
    acc = i64(acc + ba[2] * st.x[1])
    acc = i64(acc + ba[3] * st.y[0])
    acc = i64(acc + ba[4] * st.y[1])
    y0 = i32(acc >> frac)
    st.x = [x0, st.x[0]]
    st.y = [y0, st.y[0]]
    return y0


def biquad_i32_df1_clamp(ba, frac, u, lo, hi, st: DirectForm1, x0: int) -> int:
    """biquad.rs:394-404."""
    y0 = clamp(i32(biquad_i32_df1(ba, frac, st, x0) + u), lo, hi)
    st.y[0] = y0
    return y0


class DirectForm1Dither:
    """biquad.rs:484-491."""

    def __init__(self):
        self.xy = DirectForm1()
        self.e = 0


def biquad_i32_dither(ba, frac, st: DirectForm1Dither, x0: int) -> int:
    """biquad.rs:511-530."""
    xy = st.xy
    acc = i64(ba[0] * x0)
    acc = i64(acc + ba[1] * xy.x[0])
    acc = i64(acc + ba[2] * xy.x[1])
    acc = i64(acc + ba[3] * xy.y[0])
    acc = i64(acc + ba[4] * xy.y[1])
    acc = i64(st.e + acc)
    acc = i64(acc << (32 - frac))
    st.e = u32(acc) >> ((32 - frac) & 31)  # release-mode shift masking for F = 0
    y0 = i32(acc >> 32)
    xy.x = [x0, xy.x[0]]
    xy.y = [y0, xy.y[0]]
    return y0


def biquad_i32_dither_clamp(ba, frac, u, lo, hi, st: DirectForm1Dither, x0: int) -> int:
    """biquad.rs:532-538."""
    y0 = clamp(i32(biquad_i32_dither(ba, frac, st, x0) + u), lo, hi)
    st.xy.y[0] = y0
    return y0


class DirectForm1Wide:
    """biquad.rs:445-454: x: [i32;2], y: [i64;2]."""

    def __init__(self):
        self.x = [0, 0]
        self.y = [0, 0]


def biquad_i32_wide(ba, frac, st: DirectForm1Wide, x0: int) -> int:
    """biquad.rs:456-472."""
    acc = i64(ba[0] * x0)
    acc = i64(acc + ba[1] * st.x[0])
    acc = i64(acc + ba[2] * st.x[1])
    st.x = [x0, st.x[0]]
    acc = i64(acc + ((u32(st.y[0]) * ba[3]) >> 32))
    acc = i64(acc + i32(st.y[0] >> 32) * ba[3])
    acc = i64(acc + ((u32(st.y[1]) * ba[4]) >> 32))
    acc = i64(acc + i32(st.y[1] >> 32) * ba[4])
    acc = i64(acc << (32 - frac))
    st.y = [acc, st.y[0]]
    return i32(acc >> 32)


def biquad_i32_wide_clamp(ba, frac, u, lo, hi, st: DirectForm1Wide, x0: int) -> int:
    """biquad.rs:474-480."""
    y0 = clamp(i32(biquad_i32_wide(ba, frac, st, x0) + u), lo, hi)
    st.y[0] = i64((y0 << 32) | u32(st.y[0]))
    return y0


def cascade_df1(ba_list, fracs, st_x: list, st_y: List[list], x0, *, is_float=False):
    """`Cascade<[Biquad;N]>` x `DirectForm<T,N>` (biquad.rs:339-364). st_x=[x0,x1], st_y=[[y0,y1]]*N."""
    x = st_x
    for k, ba in enumerate(ba_list):
        y = st_y[k]
        if is_float:
            acc = f32(ba[0]) * x0
            acc = acc + f32(ba[1]) * x[0]
            acc = acc + f32(ba[2]) * x[1]
            acc = acc + f32(ba[3]) * y[0]
            acc = acc + f32(ba[4]) * y[1]
            y0 = acc
        else:
            acc = i64(ba[0] * x0)
            acc = i64(acc + ba[1] * x[0])
            acc = i64(acc + ba[2] * x[1])
            acc = i64(acc + ba[3] * y[0])
            acc = i64(acc + ba[4] * y[1])
            y0 = i32(acc >> fracs[k])
        x[1] = x[0]
        x[0] = x0
        x0, x = y0, y
    x[1] = x[0]
    x[0] = x0
    return x0


# ----------------------------------------------------------------------------
# iir::Biquad, f32
# ----------------------------------------------------------------------------
def biquad_f32_df1(ba, st: DirectForm1, x0):
    """biquad.rs:366-383 with C = T = A = f32, left-to-right, no fusing."""
    ba = [f32(c) for c in ba]
    x0 = f32(x0)
    y0 = ba[0] * x0
    y0 = y0 + ba[1] * st.x[0]
    y0 = y0 + ba[2] * st.x[1]
    y0 = y0 + ba[3] * st.y[0]
    y0 = y0 + ba[4] * st.y[1]
    st.x = [x0, st.x[0]]
    st.y = [y0, st.y[0]]
    return y0


def biquad_f32_df1_clamp(ba, u, lo, hi, st: DirectForm1, x0):
    y0 = clamp(biquad_f32_df1(ba, st, x0) + f32(u), f32(lo), f32(hi))
    st.y[0] = y0
    return y0


def biquad_f32_df2t(ba, s: list, x0):
    """biquad.rs:418-428; s = DirectForm2Transposed.x."""
    ba = [f32(c) for c in ba]
    x0 = f32(x0)
    y0 = s[0] + ba[0] * x0
    s[0] = s[1] + ba[1] * x0 + ba[3] * y0
    s[1] = ba[2] * x0 + ba[4] * y0
    return y0


def biquad_f32_df2t_clamp(ba, u, lo, hi, s: list, x0):
    """biquad.rs:430-440."""
    ba = [f32(c) for c in ba]
    x0 = f32(x0)
    y0 = clamp(s[0] + ba[0] * x0 + f32(u), f32(lo), f32(hi))
    s[0] = s[1] + ba[1] * x0 + ba[3] * y0
    s[1] = ba[2] * x0 + ba[4] * y0
    return y0


# ----------------------------------------------------------------------------
# hbf (src/hbf.rs)
# ----------------------------------------------------------------------------
HBF_TAPS = (
    (7.60375795e-07, -3.77494111e-06, 1.26458559e-05, -3.43188253e-05, 8.10687478e-05,
     -1.72971467e-04, 3.40845059e-04, -6.29522864e-04, 1.10128831e-03, -1.83933299e-03,
     2.95124926e-03, -4.57290964e-03, 6.87374176e-03, -1.00656257e-02, 1.44199840e-02,
     -2.03025100e-02, 2.82462332e-02, -3.91128509e-02, 5.44795658e-02, -7.77002672e-02,
     1.17523452e-01, -2.06185388e-01, 6.34588695e-01),
    (-1.12811343e-05, 1.12724671e-04, -6.07439343e-04, 2.31904511e-03, -7.00322950e-03,
     1.78225473e-02, -4.01209836e-02, 8.43315989e-02, -1.83189521e-01, 6.26346521e-01),
    (0.0007686, -0.00768669, 0.0386536, -0.14002434, 0.60828885),
    (-0.00261331, 0.02476858, -0.12112638, 0.59897111),
    (0.01186105, -0.09808109, 0.58622005),
)
HBF_TAPS_98 = (
    (7.02144012e-05, -2.43279582e-04, 6.35026936e-04, -1.39782541e-03, 2.74613582e-03,
     -4.96403839e-03, 8.41806912e-03, -1.35827601e-02, 2.11004053e-02, -3.19267647e-02,
     4.77024289e-02, -7.18014345e-02, 1.12942004e-01, -2.03279594e-01, 6.33592923e-01),
    (-0.00086943, 0.00577837, -0.02201674, 0.06357869, -0.16627679, 0.61979312),
    (0.01414651, -0.10439639, 0.59026742),
    (0.01227974, -0.09930782, 0.58702834),
    (-0.06291796, 0.5629161),
)
HBF_CASCADE_BLOCK = 32


def _get(taps, x: Sequence) -> list:
    """`get::<_,_,M,false,true>` (hbf.rs:46-68) over all windows of x."""
    m = len(taps)
    out = []
    for i in range(len(x) - 2 * m + 1):
        w = x[i:i + 2 * m]
        acc = f32(-0.0)  # f32::sum neutral element
        for k in range(m):
            acc = acc + (w[2 * m - 1 - k] + w[k]) * f32(taps[k])
        out.append(acc)
    return out


def fir_sym(taps, odd: bool, sym: bool, hist: list, x: Sequence) -> list:
    """`type_fir!` same-rate FIR (hbf.rs:70-138) via `get::<_,_,M,ODD,SYM>` (hbf.rs:46-68);
    `hist` holds the last LEN = 2M-1+odd inputs and is updated in place."""
    m = len(taps)
    ln = 2 * m - 1 + int(odd)
    assert len(hist) == ln
    buf = [f32(v) for v in hist] + [f32(v) for v in x]
    out = []
    for i in range(len(x)):
        w = buf[i:i + 2 * m + int(odd)]
        old, new = w[:m], w[len(w) - m:]
        acc = f32(-0.0)
        for k in range(m):
            nw, od = new[m - 1 - k], old[k]
            acc = acc + ((nw + od) if sym else (nw - od)) * f32(taps[k])
        out.append(acc + w[m] if (odd and sym) else acc)
    hist[:] = buf[len(x):len(x) + ln]
    return out


class HbfDec:
    """`HbfDec<[f32; N]>` (hbf.rs:142-155) with the reference's array sizes."""

    def __init__(self, m: int, n: int):
        assert n > 2 * m - 1
        self.m, self.n = m, n
        self.even = [f32(0)] * n
        self.odd = [f32(0)] * n


def hbf_dec_block(taps, st: HbfDec, x_pairs: Sequence[Tuple], y: list):
    """`SplitProcess<[T;2],T,HbfDec>::block` for EvenSymmetric (hbf.rs:163-185)."""
    m, n = st.m, st.n
    ln = 2 * m - 1
    pos = 0
    step = n - ln
    while pos < len(x_pairs):
        xc = x_pairs[pos:pos + step]
        for i, p in enumerate(xc):
            st.even[m - 1 + i] = f32(p[0])
            st.odd[ln + i] = f32(p[1])
        odd = _get(taps, st.odd)
        for i in range(len(xc)):
            y[pos + i] = odd[i] + st.even[i]
        c = len(xc)
        st.even[0:m - 1] = st.even[c:c + m - 1]
        st.odd[0:ln] = st.odd[c:c + ln]
        pos += c


class HbfInt:
    def __init__(self, m: int, n: int):
        assert n > 2 * m - 1
        self.m, self.n = m, n
        self.x = [f32(0)] * n


def hbf_int_block(taps, st: HbfInt, x: Sequence, y_pairs: list):
    """`SplitProcess<T,[T;2],HbfInt>::block` (hbf.rs:207-227)."""
    m, n = st.m, st.n
    ln = 2 * m - 1
    pos = 0
    step = n - ln
    while pos < len(x):
        xc = x[pos:pos + step]
        for i, v in enumerate(xc):
            st.x[ln + i] = f32(v)
        even = _get(taps, st.x)
        for i in range(len(xc)):
            y_pairs[pos + i] = (even[i], st.x[m + i])
        c = len(xc)
        st.x[0:ln] = st.x[c:c + ln]
        pos += c


def hbf_dec_states(taps_seq) -> list:
    """State sizes of `HbfDec2..32` (hbf.rs:363-383): stage at depth d from the
    output has N = LEN + (BLOCK << d)."""
    n = len(taps_seq)
    return [HbfDec(len(t), 2 * len(t) - 1 + (HBF_CASCADE_BLOCK << (n - 1 - s))) for s, t in enumerate(taps_seq)]


def hbf_dec_cascade_block(taps_seq, states, x: Sequence, block: int = HBF_CASCADE_BLOCK) -> list:
    """`Major<(ChunkIn<stage,2>, Major<...>), [[f32; R/2]; B]>::block`
    (compose.rs:581-593 + adapters.rs:333-339): x is the flat high-rate stream,
    processed in chunks of `block` OUTPUT frames through per-level scratch."""
    stages = len(taps_seq)
    r = 1 << stages
    frames = len(x) // r
    y = [f32(0)] * frames

    def level(s: int, xin: Sequence) -> list:
        # xin: flat samples at this stage's input rate; returns its final outputs
        rs = 1 << (stages - s)  # samples per frame at this level
        nfr = len(xin) // rs
        out = []
        if s == stages - 1:
            yy = [f32(0)] * nfr
            hbf_dec_block(taps_seq[s], states[s], [(xin[2 * i], xin[2 * i + 1]) for i in range(nfr)], yy)
            return yy
        for c0 in range(0, nfr, block):
            c1 = min(nfr, c0 + block)
            chunk = xin[c0 * rs:c1 * rs]
            u = [f32(0)] * (len(chunk) // 2)
            hbf_dec_block(taps_seq[s], states[s], [(chunk[2 * i], chunk[2 * i + 1]) for i in range(len(u))], u)
            out.extend(level(s + 1, u))
        return out

    res = level(0, [f32(v) for v in x])
    assert len(res) == frames
    return res


def hbf_int_states(taps_seq) -> list:
    """`HbfInt2..32` (hbf.rs:454-474): stage s (0 = lowest rate) has N = LEN + (BLOCK << s)."""
    return [HbfInt(len(t), 2 * len(t) - 1 + (HBF_CASCADE_BLOCK << s)) for s, t in enumerate(taps_seq)]


def hbf_int_cascade_block(taps_seq, states, x: Sequence) -> list:
    """Interpolator cascade, stage-major over the whole buffer (values are chunk independent)."""
    cur = [f32(v) for v in x]
    for s, t in enumerate(taps_seq):
        pairs = [None] * len(cur)
        hbf_int_block(t, states[s], cur, pairs)
        cur = [v for p in pairs for v in p]
    return cur


# ----------------------------------------------------------------------------
# cossin (src/cossin.rs:14-67, build.rs:8-41)
# ----------------------------------------------------------------------------
COSSIN_DEPTH = 7


def cossin_table() -> List[int]:
    amp = float(0xFFFF)
    tab = []
    for i in range(1 << COSSIN_DEPTH):
        th = math.pi / 4.0 * ((i + 0.5) / (1 << COSSIN_DEPTH))
        s, c = math.sin(th), math.cos(th)
        ci = int(round_half_away((c * 2.0 - 1.0) * amp - 1.0))
        si = int(round_half_away(s * amp))
        tab.append((ci + (si << 16)) & 0xFFFFFFFF)
    return tab


_COSSIN = cossin_table()


def cossin(phase: int) -> Tuple[int, int]:
    octant = u32(phase)
    if octant & (1 << 29):
        phase = i32(~phase)
    align = 32 - 16 - 1
    phase = i32((u32(u32(phase) << 3)) >> (32 - COSSIN_DEPTH - align))
    lookup = _COSSIN[phase >> align]
    phase &= (1 << align) - 1
    phase -= 1 << (align - 1)
    pi4 = int(math.pi / 4 * (1 << 16))
    dphi = i32(phase * pi4) >> 16
    cos = (lookup & 0xFFFF) + (1 << 16)
    sin = lookup >> 16
    dcos = i32(sin * dphi) >> COSSIN_DEPTH
    dsin = i32(cos * dphi) >> (COSSIN_DEPTH + 1)
    cos = i32((cos << (align - 1)) - dcos)
    sin = i32((sin << align) + dsin)
    octant ^= octant >> 1
    if octant & (1 << 29):
        cos, sin = sin, cos
    if octant & (1 << 30):
        cos = i32(-cos)
    if octant & (1 << 31):
        sin = i32(-sin)
    return cos, sin


# ----------------------------------------------------------------------------
# atan2 (src/atan2.rs, build.rs:43-66)
# ----------------------------------------------------------------------------
def atan2_table():
    q31 = float(1 << 31)
    out = []
    for i in range(16):
        x0, x1 = 1.0 + i / 16.0, 1.0 + (i + 1) / 16.0
        out.append((int(round_half_away(q31 / x0)) & 0xFFFFFFFF, int(round_half_away((1.0 / x1 - 1.0 / x0) * q31))))
    return out


_ATAN2 = atan2_table()
_ATANI = (0x0517c2cd, -0x06c6496b, 0x0fbdb021, -0x25b32e0a, 0x43b34c81, -0x3bc823dd)


def _mul_q31(x: int, y: int) -> int:
    return u32((x * y) >> 31)


def atan2(y: int, x: int) -> int:
    """src/atan2.rs:66-82 with divi (:12-29) and atani (:32-49)."""
    k = 0
    if y < 0:
        y = I32_MAX if y == I32_MIN else -y
        k ^= 0xFFFFFFFF
    if x < 0:
        x = I32_MAX if x == I32_MIN else -x
        k ^= 0x7FFFFFFF
    if y > x:
        y, x = x, y
        k ^= 0x3FFFFFFF
    if x == 0:
        q = 0
    else:
        shift = 32 - x.bit_length()
        yn, xn = u32(y << shift), u32(x << shift)
        rem = xn & ((1 << 27) - 1)
        idx = u32(xn << 1) >> 28
        base, slope = _ATAN2[idx]
        r0 = u32(base + u32((slope * rem) >> 27))
        q = _mul_q31(yn, _mul_q31(r0, u32(-_mul_q31(xn, r0))))
    x2 = i32((q * q) >> 32)
    r = 0
    for a in reversed(_ATANI):
        r = i32(i32((r * x2) >> 32) + a)
    return i32(u32((r * q) >> 28) ^ k)


# ----------------------------------------------------------------------------
# Normal form (src/iir/normal.rs) and wave digital allpass sections (src/iir/wdf.rs)
# ----------------------------------------------------------------------------
def normal_i32(ba, frac: int, st: DirectForm1, x0: int) -> int:
    """`Normal<Q32<F>>` x `DirectForm1<i32>` (normal.rs:43-57); ba = [b0, b1, b2, p.re, p.im];
    st.y = [y0 (in-phase), y1 (quadrature)]."""
    b0, b1, b2, re, im = ba
    y0o, y1o = st.y
    acc = i64(b0 * x0)
    for c, v in ((b1, st.x[0]), (b2, st.x[1]), (re, y1o), (i32(-im), y0o)):
        acc = i64(acc + c * v)
    y1 = i32(acc >> frac)
    y0 = i32(i64(im * y1o + re * y0o) >> frac)
    st.x = [x0, st.x[0]]
    st.y = [y0, y1]
    return y0


def normal_float(ba, st: DirectForm1, x0, F=np.float32):
    """`Normal<f32|f64>` x `DirectForm1`: every product and sum rounded in F, left to right."""
    b0, b1, b2, re, im = (F(v) for v in ba)
    y0o, y1o = F(st.y[0]), F(st.y[1])
    with np.errstate(all="ignore"):
        acc = b0 * F(x0)
        acc = acc + b1 * F(st.x[0])
        acc = acc + b2 * F(st.x[1])
        acc = acc + re * y1o
        acc = acc + (-im) * y0o
        y0 = im * y1o + re * y0o
    st.x = [F(x0), F(st.x[0])]
    st.y = [y0, acc]
    return y0


def normal_from_sos(sos):
    """`From<&[[f64; 3]; 2]> for Normal<C>` (normal.rs:62-76) -> [b0, b1, b2, p.re, p.im] or None"""
    a0 = 1.0 / sos[3]
    p2 = -0.5 * sos[4]
    pq = sos[3] * sos[5] - p2 * p2
    if not pq >= 0.0:
        return None
    return [sos[0] * a0, sos[1] * a0, sos[2] * a0, p2 * a0, math.sqrt(pq) * a0]


TPA = {"Z": 0x0, "A": 0xA, "B": 0xB, "B1": 0xE, "X": 0x1, "C": 0xC, "C1": 0xF, "D": 0xD}  # wdf.rs:14-32


def _mulq32(c: int, a: int) -> int:
    return i32((c * a) >> 32)


def tpa_adapt(nib: int, a: int, x):
    """`Tpa::adapt` (wdf.rs:65-100)"""
    x0, x1 = x
    if nib == 0xA:
        c = i32(x1 - x0); y = i32(_mulq32(c, a) + x1); return [i32(y + c), y]
    if nib == 0xB:
        c = i32(x0 - x1); y = i32(_mulq32(c, a) + x1); return [y, i32(y + c)]
    if nib == 0xE:
        c = i32(x0 - x1); y = _mulq32(c, a); return [i32(y + x1), i32(y + x0)]
    if nib == 0x1:
        return [x1, x0]
    if nib == 0xC:
        c = i32(x1 - x0); y = i32(_mulq32(c, a) - x1); return [y, i32(y + c)]
    if nib == 0xF:
        c = i32(x1 - x0); y = _mulq32(c, a); return [i32(y - x1), i32(y - x0)]
    if nib == 0xD:
        c = i32(x0 - x1); y = i32(_mulq32(c, a) - x1); return [i32(y + c), y]
    return [x0, x1]


def tpa_quantize(nib: int, g: float):
    """`Tpa::quantize` (wdf.rs:50-62) -> raw Q32<32> bits or None"""
    a = {0xA: g - 1.0, 0xB: -g, 0xE: -g, 0xC: g, 0xF: g, 0xD: -1.0 - g}.get(nib, 0.0)
    if not (-0.5 <= a <= 0.0):
        return None
    return quantize(a, 32)


def wdf_process(n: int, m: int, a, z: list, x: int) -> int:
    """`SplitProcess<i32, i32, WdfState<N>> for Wdf<N, M>` (wdf.rs:153-169): the fold writes the first
    adaptor output into the PREVIOUS slot (the result for adaptor 0), the second output travels on."""
    y = 0
    for i in range(n):
        o = tpa_adapt((m >> (4 * i)) & 0xF, a[i], [x, z[i]])
        if i == 0:
            y = o[0]
        else:
            z[i - 1] = o[0]
        x = o[1]
    z[n - 1] = x
    return y


# ----------------------------------------------------------------------------
# Cic (src/cic.rs) and the modular composition its tests compare it with
# ----------------------------------------------------------------------------
class Cic:
    """`Cic<T, N, M>` (src/cic.rs:13-28) with T = i32 or i64 (`bits`), release (wrapping) arithmetic."""

    def __init__(self, order: int, comb_delay: int, rate: int, bits: int = 64):
        assert comb_delay > 0, "Comb delay must be non-zero"  # cic.rs:36
        self.n, self.m, self.rate, self.bits = order, comb_delay, rate, bits
        self.index, self.zoh = 0, 0
        self.combs = [[0] * comb_delay for _ in range(order)]
        self.integrators = [0] * order

    def _w(self, v):
        return wrap(v, self.bits)

    def tick(self):  # cic.rs:88-90
        return self.index == 0

    def gain(self):  # cic.rs:103-105
        return self._w((self.m * (self.rate + 1)) ** self.n)

    def gain_log2(self):  # cic.rs:111-113
        return (self.m * self.rate + self.m - 1).bit_length() * self.n

    def response_length(self):  # cic.rs:116-118
        return self.rate * self.n

    def _combs(self, x):  # cic.rs:166-171 / :197-203
        for c in self.combs:
            y = self._w(x - c[0])
            c[:] = c[1:] + [x]
            x = y
        return x

    def interpolate(self, x):
        """`Process<Option<T>, T>` (cic.rs:160-182); x = None or a sample"""
        if x is not None:
            assert self.index == 0
            self.index = self.rate
            self.zoh = self._combs(x)
        else:
            self.index -= 1
        v = self.zoh
        for i in range(self.n):
            self.integrators[i] = self._w(self.integrators[i] + v)
            v = self.integrators[i]
        return v

    def decimate(self, x):
        """`Process<T, Option<T>>` (cic.rs:186-207)"""
        for i in range(self.n):
            self.integrators[i] = self._w(self.integrators[i] + x)
            x = self.integrators[i]
        if self.index > 0:
            self.index -= 1
            return None
        self.index = self.rate
        self.zoh = self._combs(x)
        return self.zoh


def cic_modular_decimator(order: int, r: int, m: int, chunks, bits: int = 64):
    """src/cic.rs:313-320: `Integrator` x N (dsp-process/src/basic.rs:457-466) -> `Downsample(R - 1)`
    (adapters.rs:71-83) -> `Comb<[T; M]>` x N (basic.rs:475-486) under `Decimator` (adapters.rs:158-167)."""
    ints, ds, combs = [0] * order, 0, [[0] * m for _ in range(order)]
    out = []
    for chunk in chunks:
        assert len(chunk) == r
        y = None
        for x in chunk:
            for i in range(order):
                ints[i] = wrap(ints[i] + x, bits)
                x = ints[i]
            if ds > 0:
                ds -= 1
                continue
            ds = r - 1
            for c in combs:
                v = wrap(x - c[m - 1], bits)
                c[:] = [x] + c[:m - 1]
                x = v
            assert y is None  # exactly one tick per chunk
            y = x
        out.append(y)
    return out


def cic_modular_interpolator(order: int, r: int, m: int, xs, bits: int = 64):
    """src/cic.rs:330-336: `Comb` x N (mapped over Option) -> `Hold` (adapters.rs:109-118) -> `Integrator` x N
    under `Interpolator` (adapters.rs:27-35)."""
    combs, hold, ints = [[0] * m for _ in range(order)], 0, [0] * order
    out = []
    for x in xs:
        row = []
        for k in range(r):
            if k == 0:
                v = x
                for c in combs:
                    w = wrap(v - c[m - 1], bits)
                    c[:] = [v] + c[:m - 1]
                    v = w
                hold = v
            v = hold
            for i in range(order):
                ints[i] = wrap(ints[i] + v, bits)
                v = ints[i]
            row.append(v)
        out.append(row)
    return out


# ----------------------------------------------------------------------------
# FM discriminator graph (examples/fm_disc.rs:25-50)
# ----------------------------------------------------------------------------
def fm_disc(carrier: int, ba, frac: int, prev, st: DirectForm1, x):
    """One sample of `(disc * deemph).minor()`; `prev` is a 1-element list holding None or (re, im)."""
    p = prev[0]
    prev[0] = (x[0], x[1])
    d = 0
    if p is not None:
        cim = i32(-p[1])
        re = i64(x[0] * p[0] - x[1] * cim)
        im = i64(x[0] * cim + x[1] * p[0])
        d = i32(atan2(i32(im >> 32), i32(re >> 32)) - carrier)
    return biquad_i32_df1(ba, frac, st, d)


# ----------------------------------------------------------------------------
# Accu, Lowpass, Lockin
# ----------------------------------------------------------------------------
class Accu:
    """src/accu.rs:16-41 on Wrapping<i32>."""

    def __init__(self, state: int, step: int):
        self.state, self.step = i32(state), i32(step)

    def next(self) -> int:
        self.state = i32(self.state + self.step)
        return self.state


def sat_sub_i32(a: int, b: int) -> int:
    return max(I32_MIN, min(I32_MAX, a - b))


def lowpass(k: Sequence[int], s: list, x: int) -> int:
    """`Lowpass<N>` (src/lowpass.rs:47-78); s = LowpassState<N>.0."""
    n = len(k)
    d = sat_sub_i32(x, i32(s[0] >> 32)) * k[0]
    if n == 1:
        s[0] = i64(s[0] + d)
        y = i32(s[0] >> 32)
        s[0] = i64(s[0] + d)
    elif n == 2:
        d = i64(d + (s[1] >> 32) * k[1])
        s[1] = i64(s[1] + d)
        s[0] = i64(s[0] + s[1])
        y = i32(s[0] >> 32)
        s[0] = i64(s[0] + s[1])
        s[1] = i64(s[1] + d)
    else:
        raise NotImplementedError
    return y


def lowpass_cascade(ks: Sequence[Sequence[int]], states: List[list], x: int) -> int:
    """`[Lowpass<N>; K]` (compose.rs:84-93)."""
    for k, s in zip(ks, states):
        x = lowpass(k, s, x)
    return x


def lockin(ks, states_iq: List[List[list]], x: int, phase: int) -> Tuple[int, int]:
    """`Lockin<C>` on (sample, phase) (src/lockin.rs:30-39,17-27)."""
    c, s = cossin(phase)
    xi = i32((c * x) >> 32)
    xq = i32((s * x) >> 32)
    return lowpass_cascade(ks, states_iq[0], xi), lowpass_cascade(ks, states_iq[1], xq)


def lockin_lo(arm, states_iq, x, lo):
    """`Lockin<C>` on (sample, LO) (src/lockin.rs:17-27): `Complex::new(C(state[0], x * lo.re), C(state[1], x * lo.im))`.
    `arm(state, v)` is the arm filter C; for i32 samples `x * Q32<32>` = ((q as i64 * x as i64) >> 32) as i32
    (dsp-fixedpoint/src/lib.rs:449-456), for f32 one rounded multiply."""
    if isinstance(x, (int, np.integer)):
        mix = [i32((int(lo[0]) * int(x)) >> 32), i32((int(lo[1]) * int(x)) >> 32)]
    else:
        mix = [f32(x) * f32(lo[0]), f32(x) * f32(lo[1])]
    return arm(states_iq[0], mix[0]), arm(states_iq[1], mix[1])


def lockin_phase(arm, states_iq, x: int, phase: int):
    """`Lockin<C>` on (sample, phase) for any arm filter (src/lockin.rs:30-39): LO = `Complex::<i32>::from_angle(phase)` read
    as `Complex<Q32<32>>` bits."""
    return lockin_lo(arm, states_iq, x, cossin(phase))


def biquad_chain_i32(sections, states: List[DirectForm1], x0: int) -> int:
    """`[Biquad<Q32<F>>; n]` x `[DirectForm1<i32>; n]`, sample-major (dsp-process/src/compose.rs:84-93); sections = [(ba, frac)]."""
    for (ba, frac), st in zip(sections, states):
        x0 = biquad_i32_df1(ba, frac, st, x0)
    return x0


def biquad_chain_f32(sections, states: List[DirectForm1], x0):
    """`[Biquad<f32>; n]` x `[DirectForm1<f32>; n]`; sections = [ba]."""
    for ba, st in zip(sections, states):
        x0 = biquad_f32_df1(ba, st, x0)
    return x0
