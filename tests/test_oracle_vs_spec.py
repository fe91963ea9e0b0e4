"""Second pin for the components the reference holds no asserted values for
("parity unpinned": DirectForm1Wide, clamp on Dither/Wide, Lowpass, Lockin,
HbfInt, HBF_TAPS_98): the C oracle (flat word records, whole-buffer stage-major)
must agree bit for bit with the independently written Python spec model
(reference-shaped state objects, `Major`-style chunked cascades) on seeded
adversarial inputs.  Also cross-checks the pinned components."""
import ctypes as C
import os

import numpy as np
import pytest

from idsp_amd import _abi
from oracle import spec
from tests import _harness as H

FM, LM = H.FM, H.LM
I32_MIN, I32_MAX = -(1 << 31), (1 << 31) - 1


@pytest.fixture(scope="module")
def o():
    return H.oracle()


def rand_i32(rng, n):
    x = rng.integers(I32_MIN, I32_MAX, size=n, dtype=np.int64, endpoint=True)
    x[rng.integers(0, n, size=max(1, n // 6))] = rng.choice([I32_MIN, I32_MAX, 0, -1, 1])
    return x.astype(np.int32)


def rand_ba(rng):
    return rand_i32(rng, 5).tolist() if rng.integers(0, 2) else rng.integers(-(1 << 30), 1 << 30, size=5).tolist()


@pytest.mark.parametrize("frac", [0, 1, 15, 29, 30, 31])
def test_i32_sections(o, frac):
    rng = np.random.default_rng(frac)
    n = 200
    for trial in range(6):
        ba = rand_ba(rng)
        u = int(rng.integers(-(1 << 20), 1 << 20))
        lo, hi = sorted(rand_i32(rng, 2).tolist())
        x = rand_i32(rng, n)
        cases = [
            ("biquad_i32_df1", 4, H.biquad_i32([(ba, frac)]), lambda st, v: spec.biquad_i32_df1(ba, frac, st, v), spec.DirectForm1),
            ("biquad_i32_df1_clamp", 4, H.biquad_clamp_i32([(ba, frac, u, lo, hi)]),
             lambda st, v: spec.biquad_i32_df1_clamp(ba, frac, u, lo, hi, st, v), spec.DirectForm1),
            ("biquad_i32_dither", 5, H.biquad_i32([(ba, frac)]), lambda st, v: spec.biquad_i32_dither(ba, frac, st, v), spec.DirectForm1Dither),
            ("biquad_i32_dither_clamp", 5, H.biquad_clamp_i32([(ba, frac, u, lo, hi)]),
             lambda st, v: spec.biquad_i32_dither_clamp(ba, frac, u, lo, hi, st, v), spec.DirectForm1Dither),
            ("biquad_i32_wide", 6, H.biquad_i32([(ba, frac)]), lambda st, v: spec.biquad_i32_wide(ba, frac, st, v), spec.DirectForm1Wide),
            ("biquad_i32_wide_clamp", 6, H.biquad_clamp_i32([(ba, frac, u, lo, hi)]),
             lambda st, v: spec.biquad_i32_wide_clamp(ba, frac, u, lo, hi, st, v), spec.DirectForm1Wide),
        ]
        for op, words, cfg, step, State in cases:
            st = np.zeros((words, 1), np.uint32)
            y = np.empty_like(x)
            assert o.stream(op, cfg, 1, st, x, y, 1, n, LM) == 0
            s = State()
            want = [step(s, int(v)) for v in x]
            assert y.tolist() == want, (op, frac, trial)
            # final state words agree with the reference-shaped state object
            if State is spec.DirectForm1:
                assert st[:, 0].view(np.int32).tolist() == s.x + s.y
            elif State is spec.DirectForm1Dither:
                assert st[:4, 0].view(np.int32).tolist() == s.xy.x + s.xy.y and int(st[4, 0]) == s.e
            else:
                y64 = [int(st[2, 0]) | (int(st[3, 0]) << 32), int(st[4, 0]) | (int(st[5, 0]) << 32)]
                assert st[:2, 0].view(np.int32).tolist() == s.x and [spec.i64(v) for v in y64] == s.y


def test_cascade_and_slice_composition(o):
    rng = np.random.default_rng(3)
    n, k = 150, 5
    bas = [rand_ba(rng) for _ in range(k)]
    fracs = [int(f) for f in rng.integers(0, 32, size=k)]
    x = rand_i32(rng, n)
    cfg = H.biquad_i32(list(zip(bas, fracs)))
    y = np.empty_like(x)
    st = np.zeros((2 + 2 * k, 1), np.uint32)
    assert o.stream("cascade_i32_df1", cfg, k, st, x, y, 1, n, LM) == 0
    sx, sy = [0, 0], [[0, 0] for _ in range(k)]
    want = [spec.cascade_df1(bas, fracs, sx, sy, int(v)) for v in x]
    assert y.tolist() == want
    # `[Biquad; N] x [DirectForm1; N]` gives the same samples (shared vs separate delay lines)
    y2 = np.empty_like(x)
    assert o.stream("biquad_i32_df1", cfg, k, np.zeros((4 * k, 1), np.uint32), x, y2, 1, n, LM) == 0
    assert np.array_equal(y, y2)


def test_f32_sections(o):
    rng = np.random.default_rng(4)
    n = 300
    for trial in range(5):
        ba = (rng.standard_normal(5) * 0.5).astype(np.float32).tolist()
        u, lo, hi = float(np.float32(rng.standard_normal() * 0.1)), -0.7, 0.9
        x = rng.standard_normal(n).astype(np.float32)
        x[::17] = np.float32(1e-41)
        for op, words, cfg, mk, step in [
            ("biquad_f32_df1", 4, H.biquad_f32([ba]), lambda: spec.DirectForm1(spec.f32(0)), lambda s, v: spec.biquad_f32_df1(ba, s, v)),
            ("biquad_f32_df1_clamp", 4, H.biquad_clamp_f32([(ba, u, lo, hi)]), lambda: spec.DirectForm1(spec.f32(0)),
             lambda s, v: spec.biquad_f32_df1_clamp(ba, u, lo, hi, s, v)),
            ("biquad_f32_df2t", 2, H.biquad_f32([ba]), lambda: [spec.f32(0), spec.f32(0)], lambda s, v: spec.biquad_f32_df2t(ba, s, v)),
            ("biquad_f32_df2t_clamp", 2, H.biquad_clamp_f32([(ba, u, lo, hi)]), lambda: [spec.f32(0), spec.f32(0)],
             lambda s, v: spec.biquad_f32_df2t_clamp(ba, u, lo, hi, s, v)),
        ]:
            y = np.empty_like(x)
            assert o.stream(op, cfg, 1, np.zeros((words, 1), np.uint32), x, y, 1, n, FM) == 0
            s = mk()
            with np.errstate(all="ignore"):
                want = np.array([step(s, v) for v in x], dtype=np.float32)
            assert np.array_equal(y.view(np.uint32), want.view(np.uint32)), (op, trial)


def test_f64_sections(o):
    """`Biquad<f64>` (same generic impls): Python floats are IEEE f64, every operation rounded once."""
    rng = np.random.default_rng(6)
    n = 300
    ba = (rng.standard_normal(5) * 0.5).tolist()
    u, lo, hi = 0.0625, -0.7, 0.9
    x = rng.standard_normal(n)
    x[::13] = 1e-310

    def df1(s, v, clamp):
        acc = ba[0] * v
        acc = acc + ba[1] * s[0]
        acc = acc + ba[2] * s[1]
        acc = acc + ba[3] * s[2]
        acc = acc + ba[4] * s[3]
        if clamp:
            acc = spec.clamp(acc + u, lo, hi)
        s[1], s[0], s[3], s[2] = s[0], v, s[2], acc
        return acc

    def df2t(s, v, clamp):
        y0 = s[0] + ba[0] * v
        if clamp:
            y0 = spec.clamp(y0 + u, lo, hi)
        s[0] = s[1] + ba[1] * v + ba[3] * y0
        s[1] = ba[2] * v + ba[4] * y0
        return y0

    for op, words, cfg, fn, clamp in [
        ("biquad_f64_df1", 8, H.biquad_f64([ba]), df1, False),
        ("biquad_f64_df1_clamp", 8, H.biquad_clamp_f64([(ba, u, lo, hi)]), df1, True),
        ("biquad_f64_df2t", 4, H.biquad_f64([ba]), df2t, False),
        ("biquad_f64_df2t_clamp", 4, H.biquad_clamp_f64([(ba, u, lo, hi)]), df2t, True),
    ]:
        y = np.empty_like(x)
        assert o.stream(op, cfg, 1, np.zeros((words, 1), np.uint32), x, y, 1, n, LM) == 0
        s = [0.0] * 4
        want = np.array([fn(s, float(v), clamp) for v in x])
        assert np.array_equal(y.view(np.uint64), want.view(np.uint64)), op


@pytest.mark.parametrize("taps_name,stages", [("HBF_TAPS", 4), ("HBF_TAPS", 5), ("HBF_TAPS_98", 3), ("HBF_TAPS_98", 5)])
def test_hbf_cascades(o, taps_name, stages):
    tap_set = 0 if taps_name == "HBF_TAPS" else 1
    table = getattr(spec, taps_name)
    rng = np.random.default_rng(stages)
    R = 1 << stages
    # decimator: processing order = tuple index stages-1 .. 0, reference chunking (block 32) vs whole buffer
    cfg = _abi.HbfCascadeF32()
    assert o.fn["hbf_dec_cascade"](tap_set, stages, C.byref(cfg)) == 0
    frames = 75
    x = rng.standard_normal(frames * R).astype(np.float32)
    st = np.zeros((o.fn["hbf_dec_state_words"](C.byref(cfg)), 1), np.uint32)
    y = np.empty(frames, np.float32)
    assert o.cfgcall("hbf_dec_f32", cfg, st, x, y, 1, frames, LM) == 0
    seq = [table[t] for t in range(stages - 1, -1, -1)]
    want = np.array(spec.hbf_dec_cascade_block(seq, spec.hbf_dec_states(seq), x), dtype=np.float32)
    assert np.array_equal(y.view(np.uint32), want.view(np.uint32))
    # interpolator: tuple index 0 .. stages-1
    cfg = _abi.HbfCascadeF32()
    assert o.fn["hbf_int_cascade"](tap_set, stages, C.byref(cfg)) == 0
    frames = 40
    x = rng.standard_normal(frames).astype(np.float32)
    st = np.zeros((o.fn["hbf_int_state_words"](C.byref(cfg)), 1), np.uint32)
    y = np.empty(frames * R, np.float32)
    assert o.cfgcall("hbf_int_f32", cfg, st, x, y, 1, frames, LM) == 0
    seq = [table[t] for t in range(stages)]
    want = np.array(spec.hbf_int_cascade_block(seq, spec.hbf_int_states(seq), x), dtype=np.float32)
    assert np.array_equal(y.view(np.uint32), want.view(np.uint32))


def test_cossin_exhaustive_sample(o):
    rng = np.random.default_rng(9)
    ph = np.concatenate([rand_i32(rng, 20000), (np.arange(-2048, 2048, dtype=np.int64) << 20).astype(np.int32)])
    out = np.empty((ph.size, 2), np.int32)
    assert o.fn["cossin_i32"](H._ptr(ph), H._ptr(out), ph.size) == 0
    for p, (c, s) in zip(ph.tolist(), out.tolist()):
        assert spec.cossin(p) == (c, s)


def test_atan2_all_octants_and_edges(o):
    rng = np.random.default_rng(10)
    edge = np.array([0, 1, -1, 2, 3, I32_MAX, I32_MIN, I32_MIN + 1, 1 << 30, -(1 << 30), (1 << 27) - 1, 1 << 27], np.int64)
    xy = np.concatenate([
        np.stack([rand_i32(rng, 20000), rand_i32(rng, 20000)], 1),
        rng.integers(-40, 41, size=(4000, 2)),  # tiny vectors: large clz normalisation shifts
        np.stack([np.repeat(edge, edge.size), np.tile(edge, edge.size)], 1),
    ]).astype(np.int32)
    out = np.empty(xy.shape[0], np.int32)
    assert o.fn["atan2_i32"](H._ptr(xy), H._ptr(out), out.size) == 0
    for (x, y), a in zip(xy.tolist(), out.tolist()):
        assert spec.atan2(y, x) == a, (y, x)


def test_atan2_table_matches_generated_header():
    """build.rs:43-66 recomputed here == the constexpr table the HIP kernel compiles in."""
    import re

    src = open(os.path.join(os.path.dirname(__file__), "..", "idsp_amd", "csrc", "atan2_table.h")).read()
    base = [int(v) for v in re.findall(r"(\d+)u", src.split("kAtan2Base")[1].split("}")[0])]
    slope = [int(v) for v in re.findall(r"-?\d+", src.split("kAtan2Slope[16]")[1].split("}")[0])]
    assert [(b, s) for b, s in zip(base, slope)] == spec.atan2_table()


@pytest.mark.parametrize("order,cascade", [(1, 1), (1, 3), (2, 1), (2, 2), (2, 4)])
def test_lowpass_and_lockin(o, order, cascade):
    rng = np.random.default_rng(10 * order + cascade)
    n = 400
    ks = []
    for _ in range(cascade):
        k = int(rng.integers(1 << 16, 1 << 30))
        ks.append([int(rng.integers(1, I32_MAX))] if order == 1 else [max(1, (k * k) >> 32), -int(k * 1.4142135623730951)])
    if cascade > 1:
        ks[-1] = rand_i32(rng, order).tolist()  # arbitrary (even unstable) gains: wrapping must agree
    cfg = H.lockin_cfg(ks)
    x = rand_i32(rng, n)
    # Lowpass cascade
    st = np.zeros((2 * order * cascade, 1), np.uint32)
    y = np.empty_like(x)
    assert o.cfgcall("lowpass_i32", cfg, st, x, y, 1, n, FM) == 0
    states = [[0] * order for _ in range(cascade)]
    assert y.tolist() == [spec.lowpass_cascade(ks, states, int(v)) for v in x]
    # Lockin with Accu
    st = np.zeros((2 + 4 * order * cascade, 1), np.uint32)
    st[0, 0], st[1, 0] = 12345, np.uint32(0x9E3779B1)
    y = np.empty(2 * n, np.int32)
    assert o.cfgcall("lockin_i32_process", cfg, st, x, y, 1, n, LM) == 0
    acc = spec.Accu(12345, 0x9E3779B1)
    siq = [[[0] * order for _ in range(cascade)] for _ in range(2)]
    want = [spec.lockin(ks, siq, int(v), acc.next()) for v in x]
    assert y.reshape(-1, 2).tolist() == [list(w) for w in want]
    assert spec.i32(int(st[0, 0])) == acc.state
    # polar read-outs fused into the lock-in == Complex::arg / norm_sqr of the Complex<i32> output
    # (src/complex.rs:254-256 -> atan2.rs:66-82; complex.rs:214-217, wrapping sum)
    for layout in (FM, LM):
        lanes, fr = 3, 50
        x3 = rand_i32(rng, lanes * fr)
        st3 = rng.integers(0, 1 << 32, size=(2 + 4 * order * cascade, lanes), dtype=np.uint64).astype(np.uint32)
        s_iq, s_arg, s_pow = st3.copy(), st3.copy(), st3.copy()
        iq = np.empty(2 * lanes * fr, np.int32)
        arg = np.empty(lanes * fr, np.int32)
        pw = np.empty(lanes * fr, np.int64)
        assert o.cfgcall("lockin_i32_process", cfg, s_iq, x3, iq, lanes, fr, layout) == 0
        assert o.cfgcall("lockin_i32_arg", cfg, s_arg, x3, arg, lanes, fr, layout) == 0
        assert o.cfgcall("lockin_i32_norm_sqr", cfg, s_pow, x3, pw, lanes, fr, layout) == 0
        assert np.array_equal(s_iq, s_arg) and np.array_equal(s_iq, s_pow)
        z = iq.reshape(-1, 2)
        assert arg.tolist() == [spec.atan2(int(im), int(re)) for re, im in z]
        assert pw.tolist() == [spec.wrap(int(re) * int(re) + int(im) * int(im), 64) for re, im in z]


@pytest.mark.parametrize("kind", [0, 1, 2, 3])
def test_fir_sym(o, kind):
    """Same-rate `type_fir!` FIRs (src/hbf.rs:70-138): C oracle vs spec, incl. chunked continuation."""
    rng = np.random.default_rng(40 + kind)
    odd, sym = kind in (0, 2), kind in (0, 1)
    for m in (1, 3, 8, 23):
        taps = (rng.standard_normal(m) * 0.3).astype(np.float32)
        cfg = _abi.FirSymF32()
        cfg.kind, cfg.m = kind, m
        for k, v in enumerate(taps):
            cfg.taps[k] = v
        words = o.fn["fir_sym_state_words"](C.byref(cfg))
        assert words == 2 * m - 1 + int(odd)
        st = np.zeros((words, 1), np.uint32)
        hist = [np.float32(0)] * words
        for n in (1, 50, 33):
            x = rng.standard_normal(n).astype(np.float32)
            y = np.empty_like(x)
            assert o.cfgcall("fir_sym_f32_process", cfg, st, x, y, 1, n, LM) == 0
            want = np.array(spec.fir_sym(taps.tolist(), odd, sym, hist, x), dtype=np.float32)
            assert np.array_equal(y.view(np.uint32), want.view(np.uint32))
            assert np.array_equal(st[:, 0], np.array(hist, np.float32).view(np.uint32))


def test_fir_even_symmetric_equals_hbf_odd_branch(o):
    """The half-band decimator is built on the EvenSymmetric FIR (src/hbf.rs:177): feeding the odd
    samples through the same-rate FIR and adding the delayed even samples reproduces HbfDec."""
    rng = np.random.default_rng(77)
    taps = list(spec.HBF_TAPS[3])
    m, n = len(taps), 40
    x = rng.standard_normal(2 * n).astype(np.float32)
    cfg = _abi.FirSymF32()
    cfg.kind, cfg.m = 1, m
    for k, v in enumerate(taps):
        cfg.taps[k] = v
    st = np.zeros((2 * m - 1, 1), np.uint32)
    odd = np.ascontiguousarray(x[1::2])
    yo = np.empty_like(odd)
    assert o.cfgcall("fir_sym_f32_process", cfg, st, odd, yo, 1, n, LM) == 0
    even_delayed = np.concatenate([np.zeros(m - 1, np.float32), x[0::2]])[:n]
    h = H.hbf_cfg([taps])
    yd = np.empty(n, np.float32)
    assert o.cfgcall("hbf_dec_f32", h, np.zeros((3 * m - 2, 1), np.uint32), x, yd, 1, n, LM) == 0
    assert np.array_equal((yo + even_delayed).view(np.uint32), yd.view(np.uint32))
